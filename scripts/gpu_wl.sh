#!/bin/bash
# both pipelines over several workloads.  usage: scripts/gpu_wl.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; mkdir -p "$OUT"
for wl in "--workload kdtree_fixed.qdmg" "--workload kdtree_fixed.qdmg --width 3840 --height 2160" "--workload heightfield_fixed.qdmg --width 3840 --height 2160" "--workload lecture7_fixed.qdmg --width 3840 --height 2160" "--workload window_fixed.qdmg --width 3840 --height 2160" "--workload hw9/dragon.qdmg --width 3840 --height 2160"; do
  for p in 0 1; do
    for f in 1 2; do
      timeout -k 10 300 python -u bench.py --pipeline $p --frames-in-flight $f --steps 8 --warmup 2 --no-cpu-baseline $wl > "$OUT/b.json" 2> "$OUT/b.err" || { tail -5 "$OUT/b.err"; exit 1; }
      python3 -c "import json,sys; d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]); print('$wl | pipeline $p F $f:', round(d['value']/1e6,1),'Mrays/s', round(d['ms_per_step'],2),'ms/frame', round(d['roofline']['kernel_ms'],2),'ms lone', d['config']['rays_per_frame'])"
    done
  done
done
