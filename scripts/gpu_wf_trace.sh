#!/bin/bash
# kernel trace of the wavefront pipeline, one frame after the other.  usage: scripts/gpu_wf_trace.sh <tag> [bench args]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; mkdir -p "$OUT"; shift
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktrace" -- python3 bench.py --pipeline 1 --steps 3 --warmup 1 --no-cpu-baseline --frames-in-flight 1 "$@" > "$OUT/ktrace_bench.json" 2> "$OUT/ktrace.err" || { tail -5 "$OUT/ktrace.err"; exit 1; }
find "$OUT/ktrace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
find "$OUT/ktrace" -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_trace.csv"
python3 - "$OUT" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] + "/kernel_trace.csv")))
rows = [r for r in rows if "wf_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last frame: find the last trace kernel whose predecessor gap is large ... simply print the last 2*D+3 launches
n = 20
t0 = int(rows[-n]["Start_Timestamp"])
for r in rows[-n:]:
    print(("walk " if "wf_walk" in r["Kernel_Name"] else "shade"), "start %8.3f ms  dur %8.3f ms" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
rm -rf "$OUT/ktrace"
