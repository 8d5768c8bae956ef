#!/bin/bash
# wavefront pipeline bring-up: parity subset with QD_PIPELINE=1, then bench both pipelines.  usage: scripts/gpu_wf.sh <tag> [pytest -k expr]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; mkdir -p "$OUT"
K=${2:-"scene_matches_oracle or synthetic_mesh"}
QD_PIPELINE=1 timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -v --timeout 120 --timeout-method thread -k "$K" > "$OUT/pytest_wf.log" 2>&1; rc=$?
grep -E "PASSED|FAILED|Error|assert" "$OUT/pytest_wf.log" | head -30; [ $rc -ne 0 ] && exit $rc
for p in 1 0; do
  for f in 1 2; do
    timeout -k 10 300 python -u bench.py --pipeline $p --frames-in-flight $f --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/bench_p${p}_f$f.json" 2> "$OUT/bench_p${p}_f$f.err" || { tail -5 "$OUT/bench_p${p}_f$f.err"; exit 1; }
    python3 -c "import json,sys; d=json.loads(open('$OUT/bench_p${p}_f$f.json').read().strip().splitlines()[-1]); print('pipeline $p F $f:', round(d['value']/1e6,1),'Mrays/s', round(d['ms_per_step'],2),'ms/frame', round(d['roofline']['kernel_ms'],2),'ms lone', d['config']['rays_per_frame'])"
  done
done
