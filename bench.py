#!/usr/bin/env python3
"""bench.py -- rays/s and ms/frame of the trace path on N MI355X GPUs of one node.

A "step" is one full frame of the workload: every rank traces its interleaved tile set with the HIP
kernels (scene buffers replicated per GPU, already resident in HBM), the compacted shards are
gathered to rank 0 over RCCL and de-interleaved.  `value` is whole-job rays/s, where a ray is a
traced Raytrace call (src/main.cpp:20) or a shadow query (src/shadinghelper.cpp:20), counted on
the device.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def build_scene(args):
    from raytracing_amd import host
    data = os.path.join(ROOT, "tests", "golden", "data")
    if args.workload == "synth":
        env = os.path.join(data, "env", "forest")
        scene = host.Scene.synthetic(args.mesh_n, seed=42, width=args.width, height=args.height,
                                     max_trace_depth=args.depth, env_folder=env)
        name = (f"synthetic displaced-grid glass mesh, {scene.desc.n_tris} triangles, midpoint KD-tree from the host builder + device SAH refinement, "
                f"{args.width}x{args.height}, maxTraceDepth {args.depth}, forest cubemap (BASELINE configs[4])")
    else:
        scene = host.Scene.load(os.path.join(data, args.workload))
        st = scene.settings
        st.want_aa = 0
        if args.width_set:
            st.frame_width, st.frame_height = args.width, args.height
        name = f"{args.workload} at {st.frame_width}x{st.frame_height}, AA off"
    return scene, name


def cpu_baseline(scene, budget_s=12.0):
    """The CPU oracle (a port of the reference's path) timed on one host core over a strided pixel
    sample of the same frame; reported only, never the thing measured as `value`."""
    import numpy as np
    import oracle
    W, H = scene.settings.frame_width, scene.settings.frame_height

    def sample(n):
        step = max(1, (W * H) // n)
        idx = np.arange(0, W * H, step, dtype=np.int64)[:n]
        return np.stack([idx % W, idx // W], 1).astype(np.int32)

    xy = sample(4096)
    t = time.perf_counter(); _, _, st = oracle.render_pixels(scene, xy, rng_mode=1, threads=1); dt = time.perf_counter() - t
    n = int(min(W * H, max(4096, 4096 * budget_s / max(dt, 1e-3))))
    xy = sample(n)
    t = time.perf_counter(); _, _, st = oracle.render_pixels(scene, xy, rng_mode=1, threads=1); dt = time.perf_counter() - t
    return {"value": st["rays"] / dt, "unit": "rays/s", "cores": 1, "kind": "port",
            "sample": f"{len(xy)} pixels on a uniform stride over the same frame ({st['rays']} rays, {dt:.1f} s, 1 thread)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="synth", help="'synth' or a scene under tests/golden/data (e.g. kdtree_fixed.qdmg)")
    ap.add_argument("--mesh-n", type=int, default=2240, help="synthetic grid size n (2*n*n triangles)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--tile", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kd-refine", default=None, help="'max_leaf,step_cost' for qd_set_kd_refine (tuning; default = library default)")
    args = ap.parse_args()
    args.width_set = args.width is not None
    args.width = args.width or 3840
    args.height = args.height or 2160

    import torch
    import torch.distributed as dist
    from raytracing_amd import device, renderer

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the trace path has no CPU fallback")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    if args.kd_refine:
        ml, sc = args.kd_refine.split(",")
        device.set_kd_refine(int(ml), float(sc))
    scene, workload = build_scene(args)
    fr = renderer.FrameRenderer(scene, tile=args.tile, rank=rank, world=world, local_device=local)
    W, H = fr.W, fr.H

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        fr.render_frame(antialias=False)
    barrier()
    fr.check()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        fr.render_shard()          # the dominant kernel, on torch's current stream
        ev[i][1].record()
        fr.gather()
    barrier()
    elapsed = time.perf_counter() - t0
    fr.check()
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, args.steps)

    stats = fr.stats_dict()
    rays_local = torch.tensor([stats["rays"], stats["primary_rays"], stats["shadow_rays"]], dtype=torch.float64, device=fr.dev)
    t_max = torch.tensor([elapsed], dtype=torch.float64, device=fr.dev)
    if world > 1:
        dist.all_reduce(rays_local, op=dist.ReduceOp.SUM)
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    elapsed = float(t_max.item())
    rays_frame = float(rays_local[0].item())

    # one untimed counting launch for the algorithmic bytes of this rank's kernel launch
    device.set_counting(True)
    fr.render_shard()
    torch.cuda.synchronize()
    cstats = fr.stats_dict()
    device.set_counting(False)
    owned_px = sum((x1 - x0) * (y1 - y0) for (x0, y0, x1, y1) in
                   __import__("raytracing_amd.sharding", fromlist=["x"]).owned_tile_rects(W, H, args.tile, rank, world))
    # 16 B per KD node visited + 76 B per triangle tested (72 B vertices + 4 B index) + 12 B per pixel written
    alg_bytes = 16.0 * cstats["kd_nodes_visited"] + 76.0 * cstats["tris_tested"] + 12.0 * owned_px

    if rank == 0:
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath) and world == 1 and args.workload == "synth" and args.mesh_n == 2240:
            try:
                traffic = json.load(open(tpath)).get("bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "rays_per_sec", "value": rays_frame * args.steps / elapsed, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "width": W, "height": H, "rays_per_frame": int(rays_frame),
                       "primary_rays": int(rays_local[1].item()), "shadow_rays": int(rays_local[2].item()),
                       "parallelism": f"tiles{args.tile}x{args.tile} interleaved over {world} rank(s), RCCL gather to rank 0"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "trace_kernel", "kernel_ms": kernel_ms, "launches_per_step": 3,
                         "note": "one frame = 3 launches of trace_kernel (pixels, then two passes over handed-over ray tasks, usually empty); kernel_ms, achieved and traffic are per frame = the sum over the three (rocprofv3: TotalDurationNs / frames)",
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "kd_nodes_visited": cstats["kd_nodes_visited"], "tris_tested": cstats["tris_tested"]},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
