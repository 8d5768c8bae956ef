#!/bin/bash
# bench the 10 M triangle scene for several device-tree refinement settings in one GPU session
cd "$GRAFT_REPO_ROOT"
for kr in "$@"; do
  timeout -k 10 300 python -u bench.py --steps 8 --warmup 2 --no-cpu-baseline --kd-refine $kr > gpurun_out/refine.log 2>gpurun_out/refine.err || { tail -5 gpurun_out/refine.err; exit 1; }
  tail -1 gpurun_out/refine.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('refine $kr', round(d['value']/1e6,1),'Mrays/s', round(d['ms_per_step'],2),'ms in flight', round(r['kernel_ms'],2),'ms lone  nodes', r['kd_nodes_visited'], 'tris', r['tris_tested'])"
done
