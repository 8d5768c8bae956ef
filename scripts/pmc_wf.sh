#!/bin/bash
# PMC passes over one bench.py run of a pipeline; sums per kernel family and per frame.  usage: scripts/pmc_wf.sh <tag> <pipeline> [bench args]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; PIPE=$2; shift; shift
mkdir -p "$OUT"
run() { name=$1; shift; counters=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc $counters --output-format csv -d "$OUT/$name" -- python3 bench.py --pipeline $PIPE --steps 2 --warmup 1 --no-cpu-baseline --frames-in-flight 1 "$@" > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"; }
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" "$@" || exit 1
run sq2 "SQ_WAVES SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "$@" || exit 1
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "$@" || exit 1
run fetch "FETCH_SIZE" "$@" || exit 1
run write "WRITE_SIZE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "$@" || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        fam = "walk" if "wf_walk" in k else "shade" if "wf_shade" in k else "onekernel" if "trace_kernel" in k else None
        if fam is None: continue
        res[fam][row["Counter_Name"]] += float(row["Counter_Value"]); n[fam][row["Counter_Name"]] += 1
FRAMES = 5   # warmup 1 + steps 2 + roofline leg 3 - but the counting launch differs; report totals and launches, ratios are what matter
summary = {fam: {c: v for c, v in d.items()} for fam, d in res.items()}
for fam, d in summary.items():
    g = lambda c: d.get(c, 0.0)
    d["_launches"] = max(n[fam].values())
    if g("SQ_WAVE_CYCLES"): d["_wait_frac"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
    if g("SQ_ACTIVE_INST_VALU"): d["_valu_lane_util"] = g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_ACTIVE_INST_VALU")) if g("SQ_ACTIVE_INST_VALU") else None
    if g("SQ_BUSY_CYCLES"): d["_valu_busy"] = g("SQ_ACTIVE_INST_VALU") / g("SQ_BUSY_CYCLES")
    if g("TCC_REQ_sum"): d["_l2_hit"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
json.dump(summary, open(out + "/pmc_summary.json", "w"), indent=1)
for fam, d in summary.items():
    print(fam, {k: (round(v, 4) if k.startswith("_") else f"{v:.4g}") for k, v in sorted(d.items())})
PY
find "$OUT" -name "*.csv" -size +1M -delete
