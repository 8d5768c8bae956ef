#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_parts; mkdir -p "$OUT"
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 120 --timeout-method thread -k "parts or bands or cannot_fail or begun_ahead or first_frame" > "$OUT/pytest_parts.log" 2>&1 || { tail -30 "$OUT/pytest_parts.log"; exit 1; }
tail -2 "$OUT/pytest_parts.log"
bash scripts/r03_parts2.sh
