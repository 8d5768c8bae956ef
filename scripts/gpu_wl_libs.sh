#!/bin/bash
# bench several builds of libqdtrace.so on several workloads in one GPU session: scripts/gpu_wl_libs.sh "<wl args>;<wl args>" lib...
cd "$GRAFT_REPO_ROOT"
IFS=';' read -ra WLS <<< "$1"; shift
for wl in "${WLS[@]}"; do
  for lib in "$@"; do
    timeout -k 10 300 python -u scripts/ab_bench.py $PWD/$lib --steps 8 --warmup 2 --no-cpu-baseline $wl > gpurun_out/wl.log 2>gpurun_out/wl.err || { tail -5 gpurun_out/wl.err; exit 1; }
    tail -1 gpurun_out/wl.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$wl] $lib', round(d['value']/1e6,1),'Mrays/s', round(d['ms_per_step'],2),'ms in flight', round(d['roofline']['kernel_ms'],2),'ms lone', d['config']['rays_per_frame'], 'rays')"
  done
done
