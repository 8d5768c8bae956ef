#!/bin/bash
# wavefront pipeline: quick parity subset, bench F=1/2, per-generation kernel trace.  usage: scripts/gpu_wf2.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; mkdir -p "$OUT"
QD_PIPELINE=1 timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread -k "${2:-scene_matches_oracle or synthetic_mesh}" > "$OUT/pytest_wf.log" 2>&1; rc=$?
tail -3 "$OUT/pytest_wf.log"; [ $rc -ne 0 ] && exit $rc
for f in 1 2; do
  timeout -k 10 300 python -u bench.py --pipeline 1 --frames-in-flight $f --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/bench_p1_f$f.json" 2> "$OUT/bench_p1_f$f.err" || { tail -5 "$OUT/bench_p1_f$f.err"; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('$OUT/bench_p1_f$f.json').read().strip().splitlines()[-1]); print('pipeline 1 F $f:', round(d['value']/1e6,1),'Mrays/s', round(d['ms_per_step'],2),'ms/frame', round(d['roofline']['kernel_ms'],2),'ms lone', d['config']['rays_per_frame'])"
done
scripts/gpu_wf_trace.sh $1
