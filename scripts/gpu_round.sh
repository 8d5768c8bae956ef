#!/bin/bash
# Full GPU check for a milestone: parity suite, bench, rocprofv3 kernel trace of the same bench command, PMC passes.
# usage: scripts/gpu_round.sh <tag>     (results under gpurun_out/<tag>/)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; mkdir -p "$OUT"
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > "$OUT/pytest_gpu.log" 2>&1; rc=$?
tail -3 "$OUT/pytest_gpu.log"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python -u bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
tail -1 "$OUT/bench.json"
# kernel trace twice: frames strictly one after the other (launch durations comparable with the roofline leg's HIP events),
# and the default command (frames in flight: launches overlap and stretch each other)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktrace" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --frames-in-flight 1 > "$OUT/ktrace_bench.json" 2> "$OUT/ktrace.err" || { tail -5 "$OUT/ktrace.err"; exit 1; }
find "$OUT/ktrace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
head -5 "$OUT/kernel_stats.csv"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktrace_default" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/ktrace_default_bench.json" 2> "$OUT/ktrace_default.err" || { tail -5 "$OUT/ktrace_default.err"; exit 1; }
find "$OUT/ktrace_default" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_default.csv"
head -3 "$OUT/kernel_stats_default.csv"
scripts/pmc_passes.sh "$1/pmc" > "$OUT/pmc.log" 2>&1; tail -30 "$OUT/pmc.log"
# FETCH_SIZE calibration on a known number of random 128-byte line fetches (scripts/micro/gather.hip)
if [ -x build_variants/gather ]; then
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/cal" -- build_variants/gather cal > "$OUT/cal.log" 2>&1
  python3 - "$OUT" <<'PY'
import csv, glob, sys
rows = [r for f in glob.glob(sys.argv[1] + "/cal/*/*counter_collection.csv") for r in csv.DictReader(open(f)) if "chase" in r["Kernel_Name"]]
for r in rows: print("calibration", r["Kernel_Name"][:40], r["Counter_Name"], r["Counter_Value"])
PY
fi
