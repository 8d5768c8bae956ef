#!/bin/bash
# bench the same workload with several builds of libqdtrace.so: scripts/gpu_libs.sh build_variants/a.so build_variants/b.so ...
cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  timeout -k 10 200 python -u scripts/ab_bench.py $PWD/$lib --steps 8 --warmup 2 --no-cpu-baseline $BENCH_ARGS > gpurun_out/libs.log 2>gpurun_out/libs.err || { tail -5 gpurun_out/libs.err; exit 1; }
  tail -1 gpurun_out/libs.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']/1e6,1),'Mrays/s', round(d['ms_per_step'],2),'ms/frame in flight', round(d['roofline']['kernel_ms'],2),'ms lone frame')"
  grep "qd profile" gpurun_out/libs.err | tail -1
done
