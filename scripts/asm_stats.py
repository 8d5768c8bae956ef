#!/usr/bin/env python3
"""Development aid: compile qdtrace.hip to gfx950 assembly (few kernel copies) and report, per kernel of interest, the
register budget and the scratch / LDS / global traffic inside its loops.  usage: scripts/asm_stats.py [-Dmacro ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "raytracing_amd/csrc/device/qdtrace.hip")
out = os.path.join(tempfile.gettempdir(), "qd_asm.s")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-mllvm", "-disable-machine-licm",
       "-DQD_DEV_BUILD", "-I", os.path.join(ROOT, "include"), "-I", os.path.dirname(src), "-S", "--cuda-device-only", src, "-o", out] + sys.argv[1:]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
names = [l.split(":")[0] for l in lines if re.match(r"^_ZN12_GLOBAL__N_1\d+(wf_walk_kernelILb0E|wf_shade_kernelILb0ELi0ELb0E|trace_kernelILb0ELi0ELb0E)", l)]
for name in names:
    start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    end = next(i for i in range(start + 10, len(lines)) if lines[i].startswith(".Lfunc_end"))
    b = lines[start:end]
    meta = {}
    for l in lines[end:end + 400]:
        m = re.match(r"\s*\.amdhsa_next_free_vgpr (\d+)|; ScratchSize: (\d+)|; NumVgprs: (\d+)|; SGPRSpill.*: (\d+)|; VGPRSpill.*: (\d+)", l) if False else None
    tail = "\n".join(lines[end:end + 120])
    info = {k: re.search(k + r":?\s+(\d+)", tail) for k in ("NumVgprs", "ScratchSize", "sgpr_spill_count", "vgpr_spill_count", "Occupancy")}
    print(name[:60], {k: (v.group(1) if v else None) for k, v in info.items()}, "instructions", sum(1 for l in b if l.startswith("\t") and not l.strip().startswith((".", ";"))))
    labels = {}
    for i, l in enumerate(b):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m: labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(b):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i: loops.append((labels[m.group(1)], i))
    for a, e in sorted(set(loops)):
        seg = b[a:e]
        if e - a < 200: continue
        c = lambda p: sum(p in x for x in seg)
        print(f"   loop {a}-{e} len {e-a}: scratch ld/st {c('scratch_load')}/{c('scratch_store')} global ld/st {c('global_load')}/{c('global_store')} ds {c('ds_')} readlane {c('v_readlane')} f64 {c('_f64')}")
