#!/bin/bash
# several builds x (synthetic, kdtree_fixed), pipeline 1, F = 1 and 2.  usage: scripts/gpu_libs_wl2.sh lib1.so lib2.so ...
cd "$GRAFT_REPO_ROOT"
for wl in "" "--workload kdtree_fixed.qdmg"; do
  for lib in "$@"; do
    for f in 1 2; do
    timeout -k 10 200 python -u scripts/ab_bench.py $PWD/$lib --pipeline 1 --frames-in-flight $f --steps 8 --warmup 2 --no-cpu-baseline $wl > gpurun_out/libs.log 2>gpurun_out/libs.err || { tail -5 gpurun_out/libs.err; exit 1; }
    tail -1 gpurun_out/libs.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl | $lib F $f', round(d['value']/1e6,1),'Mrays/s', round(d['ms_per_step'],2),'ms/frame')"
    done
  done
done
