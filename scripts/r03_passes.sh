#!/bin/bash
# per-launch durations of the one-kernel form's three passes on lone frames of the bench scene (kernel trace)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_passes; mkdir -p "$OUT"; rm -rf "$OUT/kt"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-1080p --frames-in-flight 1 "$@" > "$OUT/bench.json" 2> "$OUT/kt.err" || { tail -5 "$OUT/kt.err"; exit 1; }
find "$OUT/kt" -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_trace.csv"
rm -rf "$OUT/kt"
python3 - "$OUT/kernel_trace.csv" <<'P'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "trace_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-30:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("start %10.3f ms  dur %8.3f ms  grid %s" % ((s - t0) / 1e6, (e - s) / 1e6, r["Grid_Size_X"]))
P
