#!/bin/bash
# PMC passes over one bench.py frame (separate rocprofv3 runs per counter group; no tracing combined with --pmc).
# usage: scripts/pmc_passes.sh <outdir-under-gpurun_out> [bench args...]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
run() { name=$1; shift; counters=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc $counters --output-format csv -d "$OUT/$name" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --frames-in-flight 1 "$@" > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"; }
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" "$@" || exit 1
run sq2 "SQ_WAVES SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "$@" || exit 1
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "$@" || exit 1
run fetch "FETCH_SIZE" "$@" || exit 1
run write "WRITE_SIZE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "$@" || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "trace_kernel<false" not in k: continue
        res[row["Counter_Name"]].setdefault("vals", []).append(float(row["Counter_Value"]))
LAUNCHES_PER_FRAME = 3   # one frame = 3 launches of trace_kernel (launch() in qdtrace.hip); counters are summed per frame
summary = {c: sum(v["vals"]) / (len(v["vals"]) / LAUNCHES_PER_FRAME) for c, v in res.items()}
json.dump(summary, open(out + "/pmc_summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
