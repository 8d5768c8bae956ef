#!/bin/bash
# how many frames of an in-flight loop are enqueued in parts?  (kernel trace: wf_band_end_kernel runs once per band and part)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_parts; mkdir -p "$OUT"; rm -rf "$OUT/kt"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 bench.py --workload lecture7_fixed.qdmg --width 3840 --height 2160 --steps 60 --no-cpu-baseline --no-1080p > "$OUT/kt_bench.json" 2> "$OUT/kt.err" || { tail -5 "$OUT/kt.err"; exit 1; }
find "$OUT/kt" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/lecture7_parts_kernel_stats.csv"
rm -rf "$OUT/kt"
cut -c1-200 "$OUT/lecture7_parts_kernel_stats.csv" | head -12
python - "$OUT/kt_bench.json" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("in flight %.2f ms/frame, lone %.2f ms, form %s" % (d["ms_per_step"], r["kernel_ms"], d["config"]["schedule"]))
P
