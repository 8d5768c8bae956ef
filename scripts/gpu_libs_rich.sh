#!/bin/bash
# several builds x the analytic / textured workloads (PIPE = scheduling form, default 0; FIF = frames in flight, default 1).
# usage: scripts/gpu_libs_rich.sh lib1.so lib2.so ...
cd "$GRAFT_REPO_ROOT"
for wl in "--workload window_fixed.qdmg --width 3840 --height 2160" "--workload bumpmap_fixed.qdmg --width 3840 --height 2160" "--workload hw9/dragon.qdmg --width 3840 --height 2160" "--workload lecture7_fixed.qdmg --width 3840 --height 2160"; do
  for lib in "$@"; do
    timeout -k 10 200 python -u scripts/ab_bench.py $PWD/$lib --pipeline ${PIPE:-0} --frames-in-flight ${FIF:-1} --steps 6 --warmup 2 --no-cpu-baseline $wl > gpurun_out/libs.log 2>gpurun_out/libs.err || { tail -5 gpurun_out/libs.err; exit 1; }
    tail -1 gpurun_out/libs.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl | $lib', round(d['value']/1e6,1),'Mrays/s', round(d['ms_per_step'],2),'ms/frame')"
  done
done
