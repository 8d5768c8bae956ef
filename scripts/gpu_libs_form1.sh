#!/bin/bash
# several builds on kdtree_fixed and the bench scene, wavefront form, one frame at a time.  usage: scripts/gpu_libs_form1.sh lib1.so ...
cd "$GRAFT_REPO_ROOT"
for wl in "--workload kdtree_fixed.qdmg" "--workload heightfield_fixed.qdmg --width 3840 --height 2160" ""; do
  for lib in "$@"; do
    timeout -k 10 200 python -u scripts/ab_bench.py $PWD/$lib --pipeline 1 --frames-in-flight 1 --steps 6 --warmup 2 --no-cpu-baseline $wl > gpurun_out/libs.log 2>gpurun_out/libs.err || { tail -5 gpurun_out/libs.err; exit 1; }
    tail -1 gpurun_out/libs.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('form 1 $wl | $lib', round(d['value']/1e6,1),'Mrays/s', round(d['ms_per_step'],2),'ms/frame')"
  done
done
