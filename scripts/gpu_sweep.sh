#!/bin/bash
# bench over several KD refinement settings
cd "$GRAFT_REPO_ROOT"
for kr in "$@"; do
  timeout -k 10 200 python -u bench.py --steps 4 --warmup 1 --no-cpu-baseline --kd-refine "$kr" > gpurun_out/sweep.log 2>&1 || { tail -5 gpurun_out/sweep.log; exit 1; }
  tail -1 gpurun_out/sweep.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$kr', round(d['value']/1e6,1),'Mrays/s', round(d['ms_per_step'],2),'ms  nodes',r['kd_nodes_visited'],'tris',r['tris_tested'])"
done
