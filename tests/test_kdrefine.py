"""Device KD construction (kdrefine.h, host code inside libqdtrace): built with g++ and checked on the CPU.

The check program builds the two-level records for a fixture scene and walks them with the kernel's rules
(mesh_query in dgeom.h) against a brute-force closest hit over all triangles: distances must be bit-identical.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "golden", "data")


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("kdrefine") / "kdrefine_check")
    lib = os.path.join(ROOT, "raytracing_amd", "lib")
    if not os.path.exists(os.path.join(lib, "libqdhost.so")):
        from raytracing_amd import build
        build.build_host()
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.join(ROOT, "raytracing_amd", "csrc", "device"), os.path.join(ROOT, "tests", "native", "kdrefine_check.cpp"),
                    "-o", exe, "-L", lib, "-lqdhost", "-Wl,-rpath," + lib], check=True)
    return exe


def run(exe, scene, rays, max_leaf, cost, rebuild=0):
    target = scene if scene.startswith("synth:") else os.path.join(DATA, scene)
    p = subprocess.run([exe, target, str(rays), str(max_leaf), str(cost), str(rebuild)], capture_output=True, text=True, timeout=300)
    m = re.search(r"rays (\d+) hits (\d+) mismatches (\d+) nodes/ray ([\d.]+) tris/ray ([\d.]+) tree nodes (\d+) refs (\d+) max_leaf (\d+)", p.stdout)
    assert m, p.stdout + p.stderr
    return p.returncode, [float(x) for x in m.groups()]


@pytest.mark.parametrize("scene", ["kdtree_teapot_lowres_noenv.qdmg", "kdtree_1low_noenv.qdmg"])
def test_refined_records_give_the_brute_force_hit(checker, scene):
    rc, (rays, hits, mism, nodes, tris, *_rest) = run(checker, scene, 2000, 2, 0.6)
    assert rc == 0 and mism == 0
    assert hits > rays * 0.3          # the ray generator aims at the mesh box
    assert tris < 10                  # refined leaves: a handful of triangle tests per ray, not 20 per leaf


def test_unrefined_records_keep_the_callers_leaves(checker):
    rc, (rays, hits, mism, nodes, tris, tree_nodes, refs, max_leaf) = run(checker, "kdtree_teapot_lowres_noenv.qdmg", 500, 0, 0.6)
    assert rc == 0 and mism == 0
    from raytracing_amd import host
    st = host.Scene.load(os.path.join(DATA, "kdtree_teapot_lowres_noenv.qdmg")).kd_stats(0)
    assert refs == round(st["avg_tris"] * st["leaves"]) and max_leaf <= 20      # src/mesh.cpp:218: leaf when fewer than 20 triangles


@pytest.mark.parametrize("scene", ["kdtree_teapot_lowres_noenv.qdmg", "kdtree_1low_noenv.qdmg", "synth:60", "synth:330"])
def test_tree_built_from_the_triangles_gives_the_brute_force_hit(checker, scene):
    """qd_set_kd_build(1): the whole tree by the surface-area heuristic.  synth:330 (217 800 triangles) is large enough
    for the parallel top of the build (classification slices, one task per node) and the parallel stitch."""
    rc, (rays, hits, mism, nodes, tris, tree_nodes, refs, max_leaf) = run(checker, scene, 400 if scene == "synth:330" else 2000, 2, 0.6, rebuild=1)
    assert rc == 0 and mism == 0
    assert hits > rays * 0.3 and tris < 10 and max_leaf <= 32


def test_tree_build_does_not_depend_on_thread_timing(checker):
    a = run(checker, "synth:330", 50, 2, 0.6, rebuild=1)
    b = run(checker, "synth:330", 50, 2, 0.6, rebuild=1)
    assert a == b                      # same node / reference counts, same walk statistics
