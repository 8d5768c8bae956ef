#!/bin/bash
# rehearsal of bench.py's N-rank control flow with two ranks on ONE GPU (host-staged gloo collectives; not a measurement)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_two; mkdir -p "$OUT"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 6 --warmup 2 --collectives host --mesh-n 600 "$@" > "$OUT/bench2.json" 2> "$OUT/bench2.err"; rc=$?
echo rc=$rc; tail -c 1500 "$OUT/bench2.json"; [ $rc -ne 0 ] && tail -20 "$OUT/bench2.err"
timeout -k 10 200 python -u -m pytest tests/test_gpu_two_ranks.py -x -q --timeout 150 --timeout-method thread 2>&1 | tail -3
