#!/bin/bash
# default settings (adaptive form) on the workloads + suite.  usage: scripts/gpu_auto.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; mkdir -p "$OUT"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > "$OUT/pytest_gpu.log" 2>&1; rc=$?
tail -3 "$OUT/pytest_gpu.log"; [ $rc -ne 0 ] && exit $rc
for wl in "" "--workload kdtree_fixed.qdmg" "--workload heightfield_fixed.qdmg --width 3840 --height 2160" "--workload lecture7_fixed.qdmg --width 3840 --height 2160" "--workload window_fixed.qdmg --width 3840 --height 2160" "--workload bumpmap_fixed.qdmg --width 3840 --height 2160" "--workload hw9/dragon.qdmg --width 3840 --height 2160" "--workload kdtree_fixed.qdmg --width 3840 --height 2160" "--width 1920 --height 1080"; do
  timeout -k 10 300 python -u bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-1080p $wl > "$OUT/b.json" 2> "$OUT/b.err" || { tail -5 "$OUT/b.err"; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$wl |', round(d['value']/1e6,1),'Mrays/s', round(d['ms_per_step'],2),'ms/frame', round(r['kernel_ms'],2),'ms lone frac', round(r['frac'],3), d['config']['schedule'])"
done
