#!/bin/bash
# a few bench.py command lines, one result line each.  usage: scripts/gpu_q.sh "<args 1>" "<args 2>" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/q
for a in "$@"; do
  timeout -k 10 300 python -u bench.py --no-cpu-baseline $a > gpurun_out/q/b.json 2> gpurun_out/q/b.err || { tail -5 gpurun_out/q/b.err; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('gpurun_out/q/b.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$a |', round(d['value']/1e6,1),'Mrays/s', round(d['ms_per_step'],2),'ms/frame', round(r['kernel_ms'],2),'ms lone', d['config']['schedule'])"
done
