#!/bin/bash
# usage: scripts/gpu_wip.sh <tag> : the no-alloc WIP test alone, then the GPU suite (minus what is known to be open), then the bench line
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; mkdir -p "$OUT"
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -x -v --timeout 120 --timeout-method thread > "$OUT/wip.log" 2>&1; rc=$?
tail -5 "$OUT/wip.log"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python -u -m pytest tests -m gpu -v --timeout 300 --timeout-method thread -k "not ref_pin and not hip_path_matches" > "$OUT/pytest_gpu.log" 2>&1; rc=$?
tail -8 "$OUT/pytest_gpu.log"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python -u bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
tail -1 "$OUT/bench.json"
