#!/bin/bash
# GPU tests then one short bench (development loop)
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -3 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/bench_quick.log 2>&1 || { tail -5 gpurun_out/bench_quick.log; exit 1; }
tail -1 gpurun_out/bench_quick.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']/1e6,1),'Mrays/s', round(d['ms_per_step'],2),'ms  nodes',r['kd_nodes_visited'],'tris',r['tris_tested'],'frac',round(r['frac'],3))"
