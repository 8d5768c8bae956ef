"""In-tree builds of the native libraries (no JIT cache: the .so files travel with the tree).

  raytracing_amd/lib/libqdhost.so   host front end (g++, plain C++17)
  raytracing_amd/lib/libqdtrace.so  HIP trace kernels + C-ABI (hipcc, gfx950)
  oracle/liboracle.so               CPU restatement used only as the checker (g++)

FP discipline: -ffp-contract=off everywhere and no -march/-ffast-math, so that host, oracle and
device evaluate the same IEEE operations in the same order (the reference is a baseline x86-64
-O2 build: SSE2 doubles, no FMA).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "raytracing_amd")
LIB = os.path.join(PKG, "lib")
INC = os.path.join(ROOT, "include")

HOST_SRCS = ["image.cpp", "mesh.cpp", "heightfield.cpp", "scene.cpp", "api.cpp"]
CXXFLAGS = ["-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _headers(*dirs):
    out = []
    for d in dirs:
        for f in os.listdir(d):
            if f.endswith((".h", ".hpp", ".cuh")):
                out.append(os.path.join(d, f))
    return out


def build_host(force=False):
    os.makedirs(LIB, exist_ok=True)
    src_dir = os.path.join(PKG, "csrc", "host")
    srcs = [os.path.join(src_dir, s) for s in HOST_SRCS]
    out = os.path.join(LIB, "libqdhost.so")
    if force or _newer(out, srcs + _headers(src_dir, INC)):
        _run(["g++", *CXXFLAGS, "-shared", "-I", INC, "-I", src_dir, *srcs, "-o", out])
    return out


def build_oracle(force=False):
    src_dir = os.path.join(ROOT, "oracle")
    srcs = [os.path.join(src_dir, "oracle.cpp")]
    out = os.path.join(src_dir, "liboracle.so")
    if force or _newer(out, srcs + _headers(src_dir, INC)):
        _run(["g++", *CXXFLAGS, "-shared", "-pthread", "-I", INC, *srcs, "-o", out])
    return out


def build_device(force=False):
    os.makedirs(LIB, exist_ok=True)
    src_dir = os.path.join(PKG, "csrc", "device")
    srcs = [os.path.join(src_dir, "qdtrace.hip")]
    out = os.path.join(LIB, "libqdtrace.so")
    if force or _newer(out, srcs + _headers(src_dir, INC)):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        _run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
              "-ffp-contract=off", "-fno-fast-math",
              # machine LICM hoists the polynomial constants of rarely taken paths (asin/atan2 of the sphere's uv, the
              # bump sines) to the top of trace_nodes and spills them there on every call: 164 -> 45 spilled VGPRs, +13 %
              "-mllvm", "-disable-machine-licm",
              "-I", INC, "-I", src_dir, *srcs, "-o", out])
    return out


def build_all(force=False):
    return build_host(force), build_oracle(force), build_device(force)


if __name__ == "__main__":
    what = sys.argv[1:] or ["all"]
    force = "--force" in what
    what = [w for w in what if not w.startswith("--")] or ["all"]
    for w in what:
        {"host": build_host, "oracle": build_oracle, "device": build_device, "all": build_all}[w](force)
