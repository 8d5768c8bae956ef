#!/bin/bash
# several builds x the textured mesh workloads, one-kernel form, two frames in flight.  usage: scripts/gpu_libs_mesh.sh lib1.so ...
cd "$GRAFT_REPO_ROOT"
for wl in "--workload hw9/dragon.qdmg --width 3840 --height 2160" "--workload kdtree_teapot_lowres_env.qdmg --width 3840 --height 2160" "--workload hw9/testmesh.qdmg --width 3840 --height 2160" "--workload hw9/axe_test.qdmg --width 3840 --height 2160"; do
  for lib in "$@"; do
    timeout -k 10 200 python -u scripts/ab_bench.py $PWD/$lib --pipeline 0 --frames-in-flight 2 --steps 8 --warmup 2 --no-cpu-baseline $wl > gpurun_out/libs.log 2>gpurun_out/libs.err || { tail -5 gpurun_out/libs.err; exit 1; }
    tail -1 gpurun_out/libs.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl | $lib', round(d['value']/1e6,1),'Mrays/s', round(d['ms_per_step'],2),'ms/frame', d['roofline']['kernel'])"
  done
done
