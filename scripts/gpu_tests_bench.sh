#!/bin/bash
# GPU parity suite + default bench line.  usage: scripts/gpu_tests_bench.sh <tag>   (results under gpurun_out/<tag>/)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; mkdir -p "$OUT"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 300 --timeout-method thread > "$OUT/pytest_gpu.log" 2>&1; rc=$?
tail -5 "$OUT/pytest_gpu.log"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python -u bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
tail -1 "$OUT/bench.json"
