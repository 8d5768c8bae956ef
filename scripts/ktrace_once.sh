#!/bin/bash
# per-launch kernel durations of one prof_run.py: scripts/ktrace_once.sh <lib> [mesh-n] [rect]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ktrace_once; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 scripts/prof_run.py ${2:-2240} ${3:-0,0,3840,2160} $1 > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
rows = [r for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv") for r in csv.DictReader(open(f)) if "trace_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows: print(f'start {(int(r["Start_Timestamp"]) - t0) / 1e6:9.3f} ms  dur {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6:8.3f} ms  grid {r.get("Grid_Size_X", r.get("Grid_Size", "?"))}')
PY
