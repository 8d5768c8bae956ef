#!/bin/bash
# usage: scripts/gpu_suite.sh <tag> [extra pytest args] : the whole GPU suite in one process, then the last test file alone
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; mkdir -p "$OUT"; shift
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 300 --timeout-method thread "$@" > "$OUT/pytest_gpu.log" 2>&1; rc=$?
tail -6 "$OUT/pytest_gpu.log"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u -m pytest tests/test_zz_no_alloc.py -x -v --timeout 120 --timeout-method thread > "$OUT/zz_alone.log" 2>&1; rc=$?
tail -3 "$OUT/zz_alone.log"; exit $rc
