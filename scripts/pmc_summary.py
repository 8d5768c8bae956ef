#!/usr/bin/env python3
"""Per-frame sums of rocprofv3 --pmc counters over the trace kernels of a bench.py run (counting launches excluded).
usage: pmc_summary.py <dir with one sub-directory per pass> <out.json> [bench json for the depth]"""
import collections, csv, glob, json, sys
src, out = sys.argv[1], sys.argv[2]
depth = 8
if len(sys.argv) > 3:
    try:
        d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
        launches = d["roofline"]["launches_per_step"]
    except Exception:
        launches = None
res = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(src + "/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "trace_kernel<true" in k or "wf_walk_kernel<true" in k or "wf_shade_kernel<true" in k:
            continue          # the counting launch
        fam = "wf_walk_kernel" if "wf_walk_kernel" in k else "wf_shade_kernel" if "wf_shade_kernel" in k else "trace_kernel" if "trace_kernel" in k else None
        if fam is None:
            continue
        res[fam][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[fam][row["Counter_Name"]] += 1
summary = {"per": "frame (sum over the frame's launches of the kernel)", "kernels": {}}
tot_fetch = tot_write = 0.0
for fam, d in res.items():
    n_launch = max(cnt[fam].values())
    per_frame = 3 if fam == "trace_kernel" else None
    if fam == "trace_kernel" and "wf_shade_kernel" in res:
        continue              # the in-stream fallback launches behind the wavefront bands (they return at once): not a kernel of this frame
    if fam != "trace_kernel":
        # launches per frame of this family: from the bench line (walk launches = shade launches x KD-mesh nodes)
        shade = max(cnt["wf_shade_kernel"].values()) if "wf_shade_kernel" in cnt else n_launch
        walk = max(cnt["wf_walk_kernel"].values()) if cnt.get("wf_walk_kernel") else 0      # a scene without KD meshes has no walk launches
        per_frame = n_launch / (shade / (launches / (1 + walk / shade))) if launches else None
    frames = n_launch / per_frame if per_frame else 1
    g = {c: v / frames for c, v in d.items()}
    g["_launches_in_run"] = n_launch; g["_frames_in_run"] = frames
    if g.get("SQ_WAVE_CYCLES"): g["_wait_frac"] = g.get("SQ_WAIT_ANY", 0) / g["SQ_WAVE_CYCLES"]
    if g.get("SQ_ACTIVE_INST_VALU"): g["_valu_lane_util"] = g.get("SQ_THREAD_CYCLES_VALU", 0) / (64 * g["SQ_ACTIVE_INST_VALU"])
    if g.get("TCC_HIT_sum") is not None and g.get("TCC_MISS_sum") is not None and g["TCC_HIT_sum"] + g["TCC_MISS_sum"] > 0:
        g["_l2_hit"] = g["TCC_HIT_sum"] / (g["TCC_HIT_sum"] + g["TCC_MISS_sum"])
    # MI355X_MICROARCH.md, HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 B -> x2
    g["_hbm_read_bytes"] = 2.0 * g.get("FETCH_SIZE", 0.0) * 1024.0
    g["_hbm_write_bytes"] = g.get("WRITE_SIZE", 0.0) * 1024.0
    tot_fetch += g["_hbm_read_bytes"]; tot_write += g["_hbm_write_bytes"]
    summary["kernels"][fam] = g
summary["hbm_bytes_per_frame"] = {"read": tot_fetch, "write": tot_write, "total": tot_fetch + tot_write}
json.dump(summary, open(out, "w"), indent=1)
print(json.dumps({k: {c: (round(v, 4) if c.startswith("_") and abs(v) < 10 else f"{v:.4g}") for c, v in g.items() if c.startswith("_")} for k, g in summary["kernels"].items()}), summary["hbm_bytes_per_frame"])
