#!/bin/bash
# several builds on the bench scene, one-kernel form, one frame at a time and two in flight.  usage: scripts/gpu_libs_synth.sh lib1.so ...
cd "$GRAFT_REPO_ROOT"
for fif in 1 2; do
  for lib in "$@"; do
    timeout -k 10 200 python -u scripts/ab_bench.py $PWD/$lib --pipeline ${PIPE:-0} --frames-in-flight $fif --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/libs.log 2>gpurun_out/libs.err || { tail -5 gpurun_out/libs.err; exit 1; }
    tail -1 gpurun_out/libs.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('F $fif | $lib', round(d['value']/1e6,1),'Mrays/s', round(d['ms_per_step'],2),'ms/frame')"
  done
done
