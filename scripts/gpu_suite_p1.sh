#!/bin/bash
# the whole GPU suite on the wavefront pipeline, then both pipelines over the workloads.  usage: scripts/gpu_suite_p1.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; mkdir -p "$OUT"
QD_PIPELINE=1 timeout -k 10 900 python -u -m pytest tests -m gpu -v --timeout 300 --timeout-method thread > "$OUT/pytest_gpu_p1.log" 2>&1; rc=$?
grep -E "FAILED|ERROR|passed|failed" "$OUT/pytest_gpu_p1.log" | tail -12
exit $rc
