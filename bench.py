#!/usr/bin/env python3
"""bench.py -- rays/s and ms/frame of the trace path on N MI355X GPUs of one node.

A "step" is one full frame of the workload: every rank traces its interleaved tile set with the HIP
kernels (scene buffers replicated per GPU, already resident in HBM), the compacted shards are
gathered to rank 0 over RCCL and de-interleaved.  Up to --frames-in-flight frames are in flight at a time,
each on its own HIP stream with its own buffers (the thin tail of one frame overlaps the start of the next);
all K frames are complete when the clock stops.  `value` is whole-job rays/s, where a ray is a
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
# One hardware queue per frame in flight (plus RCCL's): HIP gives a process 4 by default and streams beyond that share
# a queue and serialise.  Read by the HIP runtime when it is loaded, i.e. before `import torch` below.
if int(os.environ.get("GPU_MAX_HW_QUEUES") or 0) < 8:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
L2_PEAK_GBS = 34500.0     # MI355X_MICROARCH.md: L2, 4 MiB per XCD, ~34.5 TB/s aggregate
MAX_CLOCK_HZ = 2.4e9      # MI355X_MICROARCH.md: max clock (the chip holds less under load: the issue bound below is priced on the maximum)


def lib_sha256():
    """Identity of the device library build: sha256 over its sources (csrc/device/*, include/qdtrace.h) -- the same for every
    rebuild of the same tree, unlike the bytes of the .so.  None for a development build loaded through device.use_library."""
    import hashlib
    from raytracing_amd import device
    if device._LIB_OVERRIDE:
        return None
    h = hashlib.sha256()
    src = os.path.join(ROOT, "raytracing_amd", "csrc", "device")
    for f in sorted(os.listdir(src)) + [os.path.join(ROOT, "include", "qdtrace.h")]:
        p = f if os.path.isabs(f) else os.path.join(src, f)
        if os.path.isfile(p):
            h.update(os.path.basename(p).encode()); h.update(open(p, "rb").read())
    return h.hexdigest()


def profiled_counters(key):
    """PMC sums per frame of a separate `rocprofv3 --pmc` run of this workload (scripts/gpu_profile.sh ->
    scripts/update_traffic.py -> profiles/hbm_traffic.json), valid only for the library build they were collected on:
    (entry or None, source string)."""
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not key or not os.path.exists(tpath):
        return None, "no PMC profile of this workload under profiles/"
    try:
        e = json.load(open(tpath)).get(key)
    except Exception as ex:
        return None, f"profiles/hbm_traffic.json unreadable: {ex}"
    if not e:
        return None, "no PMC profile of this workload under profiles/"
    if e.get("lib_sha256") != lib_sha256():
        return None, f"stale: profiles/hbm_traffic.json[{key}] was collected on build {e.get('build')} (other device sources than this run's)"
    return e, f"profiles/hbm_traffic.json[{key}], build {e.get('build')}, device sources sha256 {e['lib_sha256'][:12]}: {e.get('source', '')}"


def build_scene(args):
    from raytracing_amd import host
    data = os.path.join(ROOT, "tests", "golden", "data")
    if args.workload == "synth":
        env = os.path.join(data, "env", "forest")
        scene = host.Scene.synthetic(args.mesh_n, seed=42, width=args.width, height=args.height,
                                     max_trace_depth=args.depth, env_folder=env)
        name = (f"synthetic displaced-grid glass mesh, {scene.desc.n_tris} triangles, device KD-tree built from the triangles by SAH, "
                f"{args.width}x{args.height}, maxTraceDepth {args.depth}, forest cubemap (BASELINE configs[4])")
    else:
        scene = host.Scene.load(os.path.join(data, args.workload))
        st = scene.settings
        st.want_aa = 0
        if args.width_set:
            st.frame_width, st.frame_height = args.width, args.height
        name = f"{args.workload} at {st.frame_width}x{st.frame_height}, AA off"
    return scene, name


def cpu_baseline(scene, budget_s=12.0, gpu_frame=None):
    """The CPU oracle (a port of the reference's path) timed on one host core over a strided pixel
    sample of the same frame; reported only, never the thing measured as `value`.  The colours the oracle computes
    for that sample are compared (untimed) with the frame the GPU rendered: second return value."""
    import numpy as np
    import oracle
    W, H = scene.settings.frame_width, scene.settings.frame_height

    def sample(n):
        step = max(1, (W * H) // n)
        idx = np.arange(0, W * H, step, dtype=np.int64)[:n]
        return np.stack([idx % W, idx // W], 1).astype(np.int32)

    scene.ensure_kd()              # the reference-shaped tree the oracle walks is built on request (2 s for 10 M triangles): not part of the timing
    xy = sample(4096)
    t = time.perf_counter(); _, _, st = oracle.render_pixels(scene, xy, rng_mode=1, threads=1); dt = time.perf_counter() - t
    n = int(min(W * H, max(4096, 4096 * budget_s / max(dt, 1e-3))))
    xy = sample(n)
    t = time.perf_counter(); orgb, _, st = oracle.render_pixels(scene, xy, rng_mode=1, threads=1); dt = time.perf_counter() - t
    out = {"value": st["rays"] / dt, "unit": "rays/s", "cores": 1, "kind": "port",
           "sample": f"{len(xy)} pixels on a uniform stride over the same frame ({st['rays']} rays, {dt:.1f} s, 1 thread)"}
    parity = None
    if gpu_frame is not None:
        err = np.abs(gpu_frame[xy[:, 1], xy[:, 0]].astype(np.float64) - orgb.astype(np.float64)).max(axis=1)
        parity = {"checked_px": int(len(xy)), "max_err": float(np.nanmax(err)), "bad_px": int((~(err <= 1e-4)).sum()), "tol": 1e-4,
                  "note": "GPU frame vs the oracle's colours of the cpu_baseline sample (untimed); bad_px = pixels off by more than tol, "
                          "budget 0.01 % (near-tie rays)"}
    # also one thread per core of this rank's CPU share (the reference itself has one render thread)
    nthr = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "16")), os.cpu_count() or 1))
    if nthr > 1:
        xy = sample(int(min(W * H, len(xy) * min(nthr, 8))))
        t = time.perf_counter(); _, _, st = oracle.render_pixels(scene, xy, rng_mode=1, threads=nthr); dt = time.perf_counter() - t
        out["all_cores"] = {"value": st["rays"] / dt, "cores": nthr, "sample": f"{len(xy)} pixels, {st['rays']} rays, {dt:.1f} s"}
    return out, parity


def leg_1080p(args, scene, fr, rank, world, local, barrier):
    """The scene of the timed region once more at 1920x1080 (same camera, same device scene and form choice, same frames in
    flight): ms per frame and whole-job rays/s.  Untimed as far as `value` goes."""
    import torch
    import torch.distributed as dist
    from raytracing_amd import device, renderer
    st = scene.settings
    old = (st.frame_width, st.frame_height)
    if args.pipeline is None or args.pipeline == 2:
        device.set_pipeline(2)                                    # the library's own choice, as in the timed region
    st.frame_width, st.frame_height = 1920, 1080
    try:
        f2 = renderer.FrameRenderer(scene, tile=args.tile, rank=rank, world=world, local_device=local, frames_in_flight=args.frames_in_flight,
                                    dscene=fr.dscene, comm=fr.comms, streams=[sl.stream for sl in fr.slots], collectives=args.collectives)
        f2.settle()
        barrier()
        for _ in range(args.warmup):
            f2.submit_frame()
        barrier()
        f2.check()
        steps = max(args.steps, 20)
        t0 = time.perf_counter()
        for _ in range(steps):
            f2.submit_frame()
        barrier()
        dt = time.perf_counter() - t0
        f2.check()
        sched = f2.dscene.schedule_info()
        v = torch.tensor([float(f2.stats_dict()["rays"]), dt], dtype=torch.float64)
        if world > 1:
            r = v[:1].clone(); t = v[1:].clone()
            dist.all_reduce(r, op=dist.ReduceOp.SUM); dist.all_reduce(t, op=dist.ReduceOp.MAX)
            v = torch.cat([r, t])
        rays, dt = float(v[0].item()), float(v[1].item())
        return {"width": 1920, "height": 1080, "steps": steps, "ms_per_frame": dt / steps * 1e3, "rays_per_frame": int(rays),
                "rays_per_sec": rays * steps / dt, "form": sched["form"], "frames_in_flight": args.frames_in_flight}
    finally:
        st.frame_width, st.frame_height = old


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="synth", help="'synth' or a scene under tests/golden/data (e.g. kdtree_fixed.qdmg)")
    ap.add_argument("--mesh-n", type=int, default=2240, help="synthetic grid size n (2*n*n triangles)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--tile", type=int, default=32)
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frames enqueued at a time, one HIP stream each (1 = strictly one after the other; 0 = by shard size: 4 up to 4 GPUs, 6 beyond)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--collectives", default="rccl", choices=["rccl", "host"],
                    help="'host': the exchange staged through host memory over gloo -- a rehearsal of N ranks sharing ONE GPU (RCCL refuses two ranks per device); not a measurement")
    ap.add_argument("--no-1080p", action="store_true", help="skip the 1920x1080 leg that follows a 3840x2160 run")
    ap.add_argument("--pipeline", type=int, default=None, help="qd_set_pipeline: 0 = one kernel with per-lane queues, 1 = wavefront, 2 = the faster of the two, measured (library default)")
    ap.add_argument("--frame-parts", type=int, default=None, help="qd_set_frame_parts: concurrent parts of a lone wavefront frame (1 = off; default = library default, 4)")
    ap.add_argument("--kd-build", type=int, default=None, help="qd_set_kd_build: 1 = device tree by SAH from the triangles, 0 = the caller's planes + refined leaves")
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
    if args.collectives == "host":
        local = 0                                         # every rank on the one GPU
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the trace path has no CPU fallback")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # host-side rendezvous only (barriers, three scalars, the 128-byte RCCL id): gloo.  The frames travel over the library's
        # own RCCL communicator (qd_comm_*); a second RCCL communicator inside torch would only compete for its channels.
        dist.init_process_group("gloo", rank=rank, world_size=world)

    if args.frames_in_flight == 0:
        # frames needed to keep the chip full ~ critical path of a frame (4-5 ms of dependent ray generations) / work per shard
        args.frames_in_flight = 4 if world <= 4 else 6      # world 1, final build: 11.3 / 11.1 / 11.1 ms per frame at F = 2 / 3 / 4; wavefront-form scenes gain 15-25 % from 2 -> 4
    if args.pipeline is not None:
        device.set_pipeline(args.pipeline)
    if args.kd_build is not None:
        device.set_kd_build(args.kd_build)
    if args.frame_parts is not None:
        device.set_frame_parts(args.frame_parts)
    if args.kd_refine:
        ml, sc = args.kd_refine.split(",")
        device.set_kd_refine(int(ml), float(sc))
    device.init(local)
    from raytracing_amd import host
    host.prep_on_device(True)      # heightfield tables / bump-map differentiation of the scene by the device kernels (same bits as the CPU code)
    t_setup = time.perf_counter()
    scene, workload = build_scene(args)
    t_host_scene = time.perf_counter() - t_setup
    t_setup = time.perf_counter()
    fr = renderer.FrameRenderer(scene, tile=args.tile, rank=rank, world=world, local_device=local,
                                frames_in_flight=args.frames_in_flight, collectives=args.collectives)
    t_device_scene = time.perf_counter() - t_setup     # qd_scene_create: upload + device KD tree build, untimed set-up
    W, H = fr.W, fr.H

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fr.settle()                    # set-up, untimed: both scheduling forms are timed on this scene, the faster one renders from here on
    barrier()
    for _ in range(args.warmup):
        fr.submit_frame()
    barrier()
    fr.check()

    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        fr.submit_frame()          # trace + gather + de-interleave of one frame, on the stream of slot i % F
    barrier()                      # torch.cuda.synchronize() waits for every stream: all K frames are complete here
    elapsed = time.perf_counter() - t0
    fr.check()

    # roofline leg, untimed: the dominant kernel alone, launches strictly one after the other on torch's current
    # stream with HIP events around each frame's launches (overlapped launches would stretch each other's duration)
    sched = fr.dscene.schedule_info()        # the form the timed frames ran in: the roofline leg and the counting launch are pinned to it
    form = sched["form"]
    if args.pipeline is None or args.pipeline == 2:
        device.set_pipeline(form)
    n_roof = max(3, min(args.steps, 10))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_roof)]
    fr.render_shard()                        # (untimed: the first frame after frames in flight; the library treats the ones after it as a serial caller's)
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        fr.render_shard()
        b.record()
        torch.cuda.synchronize()             # every frame starts on a drained GPU: what a caller that renders one image gets
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / n_roof
    lone_parts = fr.dscene.frame_parts()     # a lone frame of the wavefront form is enqueued as concurrent parts (qd_set_frame_parts)

    stats = fr.stats_dict()
    rays_local = torch.tensor([stats["rays"], stats["primary_rays"], stats["shadow_rays"]], dtype=torch.float64)
    t_max = torch.tensor([elapsed], dtype=torch.float64)
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
    # 16 B per KD node visited + 76 B per triangle tested (72 B vertices + 4 B index) + 12 B per pixel written;
    # a scene whose traversal is a heightfield DDA instead: 68 B per DDA step (4 B maxHeight + one 64 B highmap line) + 16 B
    # per cell tested (four heights).  The counting build tallies steps with the node visits and cells with the triangle tests.
    hf_scene = scene.desc.n_hfields > 0 and scene.desc.n_tris == 0
    per_visit, per_test = (68.0, 16.0) if hf_scene else (16.0, 76.0)
    alg_bytes = per_visit * cstats["kd_nodes_visited"] + per_test * cstats["tris_tested"] + 12.0 * owned_px

    if rank == 0:
        n_kd = sum(1 for i in range(scene.desc.n_nodes) if scene.desc.geoms[scene.desc.nodes[i].geom].type == 5 and (scene.desc.meshes[scene.desc.geoms[scene.desc.nodes[i].geom].a].kd_root >= 0 or scene.desc.meshes[scene.desc.geoms[scene.desc.nodes[i].geom].a].flags & 4))
        launches = 3 if form == 0 else (scene.settings.max_trace_depth + 2) * (n_kd + 1) * lone_parts
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        ms_step = elapsed / args.steps * 1e3
        achieved_in_flight = alg_bytes / (ms_step * 1e-3) / 1e9         # the same bytes over what a frame costs with F frames in flight
        tkey = {"synth": "synth10m_4k" if (args.mesh_n, W, H) == (2240, 3840, 2160) else None}.get(args.workload, f"{os.path.basename(args.workload).split('.')[0]}_{W}x{H}")
        prof, traffic_source = profiled_counters(tkey if world == 1 else None)
        traffic = prof["bytes_per_frame"] if prof else None
        # Which roof?  HBM is the contract's bound.  A working set that lives in L2 / MALL (counter traffic below the
        # algorithmic bytes) is not bounded by it, and the KD walk on this chip runs into instruction issue first: both are
        # priced beside it.  issue: a SIMD issues one wave-instruction per 4 cycles at most -> wave-instructions x 4 / (SIMDs x clock).
        n_simd = 4 * torch.cuda.get_device_properties(local).multi_processor_count
        other = {"l2": {"achieved": achieved, "peak": L2_PEAK_GBS, "unit": "GB/s", "frac_lone": achieved / L2_PEAK_GBS, "frac_in_flight": achieved_in_flight / L2_PEAK_GBS,
                        "applies": bool(traffic is not None and traffic < alg_bytes),
                        "note": "algorithmic bytes / kernel time against the aggregate L2 bandwidth; applies when the PMC traffic is below the algorithmic bytes (the working set is served by L2 / MALL)"}}
        if prof and prof.get("wave_insts_per_frame"):
            wi = float(prof["wave_insts_per_frame"]["total"])
            t_issue_ms = wi * 4.0 / (n_simd * MAX_CLOCK_HZ) * 1e3
            other["issue"] = {"wave_insts_per_frame": wi, "by_kind": prof["wave_insts_per_frame"], "min_ms_per_frame": t_issue_ms, "simds": n_simd, "clock_hz": MAX_CLOCK_HZ,
                              "frac_lone": t_issue_ms / kernel_ms, "frac_in_flight": t_issue_ms / ms_step,
                              "valu_lane_util": prof.get("valu_lane_util"), "l2_hit": prof.get("l2_hit"),
                              "note": "time the frame's wave-instructions need at one per SIMD per 4 cycles and the maximum clock, over the measured time: the fraction of the issue roof"}
        out = {
            "metric": "rays_per_sec", "value": rays_frame * args.steps / elapsed, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "width": W, "height": H, "rays_per_frame": int(rays_frame),
                       "primary_rays": int(rays_local[1].item()), "shadow_rays": int(rays_local[2].item()),
                       "parallelism": f"tiles{args.tile}x{args.tile} interleaved over {world} rank(s), " + ("RCCL gather to rank 0" if args.collectives == "rccl" else "HOST-STAGED gloo gather (rehearsal on one GPU, not a measurement)"),
                       "frames_in_flight": args.frames_in_flight, "schedule": sched,
                       "setup_s": {"host_scene": round(t_host_scene, 2), "device_scene": round(t_device_scene, 2), **{k: round(v, 2) for k, v in fr.dscene.setup_info().items()}}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "frac_lone": achieved / HBM_PEAK_GBS, "frac_in_flight": achieved_in_flight / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source, "other_bounds": other,
                         "kernel": "trace_kernel" if form == 0 else "wf_walk_kernel + wf_shade_kernel", "kernel_ms": kernel_ms,
                         "launches_per_step": launches, "launch_avg_ms": kernel_ms / launches, "lone_frame_parts": lone_parts,
                         "note": ("one frame = 3 launches of trace_kernel (pixels, then two passes over handed-over ray tasks, usually empty)" if form == 0 else
                                  "one frame = (maxTraceDepth + 2) x (one wf_walk_kernel launch per KD-mesh node + one wf_shade_kernel launch)"
                                  + (f", a lone frame as {lone_parts} concurrent parts on streams of the scene's own (kernel_ms is the frame's span on the caller's stream; the parts' launches overlap, so their durations in a rocprofv3 trace sum to more)" if lone_parts > 1 else "")) +
                                 "; kernel_ms, achieved, frac (= frac_lone) and traffic are per frame = the sum over its launches, measured on frames that run alone (a separate untimed loop after the timed region; rocprofv3 of `bench.py --frames-in-flight 1`: TotalDurationNs / frames); the timed region overlaps frames, so ms_per_step can be below kernel_ms: frac_in_flight prices the same bytes on ms_per_step",
                         "algorithmic_bytes_per_launch": alg_bytes, "bytes_per_visit_and_test": [per_visit, per_test],
                         "kd_nodes_visited": cstats["kd_nodes_visited"], "tris_tested": cstats["tris_tested"]},
        }
        if world == 1 and not args.no_cpu_baseline:
            fr.render_shard()
            gpu_frame = fr.gather().cpu().numpy()
            fr.check()
            out["cpu_baseline"], out["parity"] = cpu_baseline(scene, gpu_frame=gpu_frame)
    # BASELINE.json's metric also names ms/frame at 1080p: the same scene, same form choice, same frames in flight at 1920x1080,
    # after the timed region (never part of `value`); every rank takes part (the frames are gathered)
    small = None
    if (W, H) == (3840, 2160) and not args.no_1080p:
        small = leg_1080p(args, scene, fr, rank, world, local, barrier)
    if rank == 0:
        if small:
            out["config"]["ms_per_frame_1080p"] = small["ms_per_frame"]
            out["config"]["rays_per_sec_1080p"] = small["rays_per_sec"]
            out["config"]["leg_1080p"] = small
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
