#!/usr/bin/env python3
"""Regenerates tests/golden/ref/*: golden vectors computed by the REFERENCE'S OWN object code.

Build-container only (needs /root/reference).  oracle/ref_build.sh compiles the reference translation units that need
neither SDL nor OpenEXR, unmodified, plus oracle/ref_driver.cpp; this script writes the inputs (a 160 x 120 frame of camera
rays of tests/golden/data/ref_pin.qdmg, 2048 random rays, 2048 function-level records), runs the driver and keeps its
outputs.  Inputs and outputs are committed; tests/test_ref_vectors.py holds the oracle, the host front end and (on the
GPU box) the HIP path to them.

    python tests/golden/make_ref_vectors.py
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "ref")
SCENE = os.path.join(ROOT, "tests", "golden", "data", "ref_pin.qdmg")
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")


def random_rays(n, seed):
    """Rays that do not share an origin: starts in a box around and inside the scene, directions uniform on the sphere,
    plus rays with exactly axis-parallel directions (zero components take the branches of Plane / Cube / BBox that camera
    rays never reach)."""
    rng = np.random.default_rng(seed)
    start = rng.uniform([-260, -20, -260], [260, 220, 260], size=(n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    axis = rng.integers(0, 3, size=n // 8)
    sign = rng.choice([-1.0, 1.0], size=n // 8)
    d[: n // 8] = 0.0
    d[np.arange(n // 8), axis] = sign
    d[n // 8: n // 4, 1] = 0.0                      # horizontal rays (dir.y == 0)
    nrm = np.linalg.norm(d, axis=1, keepdims=True)
    d = np.where(nrm > 0, d / np.where(nrm > 0, nrm, 1), [0, 0, 1.0])
    return np.concatenate([start, d], axis=1)


def func_records(n, seed):
    """{a[3], b[3], c[3], d[3], s, 0, 0, 0}: a, b box corners / matrix rows / scale & rotation, c a point, d a direction."""
    rng = np.random.default_rng(seed)
    r = np.zeros((n, 16))
    r[:, 0:3] = rng.uniform(-3, 3, size=(n, 3))
    r[:, 3:6] = rng.uniform(-3, 3, size=(n, 3))
    r[:, 6:9] = rng.uniform(-5, 5, size=(n, 3))
    r[:, 9:12] = rng.normal(size=(n, 3))
    r[:, 12] = rng.uniform(0.3, 2.5, size=n)           # ior ratio / angle
    k = n // 8
    lo, hi = np.minimum(r[:, 0:3], r[:, 3:6]), np.maximum(r[:, 0:3], r[:, 3:6])
    r[:k, 6:9] = lo[:k] + (hi[:k] - lo[:k]) * rng.uniform(0, 1, size=(k, 3))       # ray starts inside the box
    r[k:2 * k, 6] = lo[k:2 * k, 0]                                                   # ... exactly on a face
    ax = rng.integers(0, 3, size=k)
    r[2 * k:3 * k, 9:12] = 0.0
    r[np.arange(2 * k, 3 * k), 9 + ax] = rng.choice([-1.0, 1.0], size=k)            # axis-parallel directions
    r[3 * k:4 * k, 9] *= 1e-10                                                       # |dir.x| below the 1e-9 cut of the slab test
    r[4 * k:5 * k, 0:3] = np.abs(r[4 * k:5 * k, 0:3]) + 0.2                          # positive scales (Transform)
    return r


def main():
    if not os.path.isdir("/root/reference/src"):
        raise SystemExit("needs /root/reference (build container only)")
    subprocess.check_call(["bash", os.path.join(ROOT, "oracle", "ref_build.sh")])
    import oracle
    from raytracing_amd import host
    os.makedirs(OUT, exist_ok=True)
    s = host.Scene.load(SCENE)
    W, H = s.settings.frame_width, s.settings.frame_height
    cam = np.zeros((W * H, 6))
    assert oracle.lib().orc_camera_rays(s.desc_ptr, 0, 0, W, H, cam.ctypes.data) == 0
    rays = np.concatenate([cam, random_rays(2048, 7)], axis=0)
    rays.astype("<f8").tofile(os.path.join(OUT, "pin_rays.f64"))
    subprocess.check_call([DRIVER, "scene", SCENE, os.path.join(OUT, "pin_rays.f64"), os.path.join(OUT, "pin_records.bin")])
    func_records(2048, 11).astype("<f8").tofile(os.path.join(OUT, "funcs_in.f64"))
    subprocess.check_call([DRIVER, "funcs", os.path.join(OUT, "funcs_in.f64"), os.path.join(OUT, "funcs_out.f64")])
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
