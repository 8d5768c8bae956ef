#!/bin/bash
# bench lines for several frames-in-flight settings in ONE gpurun call (boxes differ by several percent), plus the
# single-GPU rehearsal of an 8-way shard.  usage: scripts/flight_bench.sh "1 2 3 4"
set -e
mkdir -p gpurun_out/flight
for f in ${1:-1 2 3 4}; do
  timeout -k 10 200 python -u bench.py --steps 12 --warmup 4 --frames-in-flight $f --no-cpu-baseline | tee gpurun_out/flight/bench_f$f.json
done
timeout -k 10 300 python -u scripts/flight_probe.py 2240 1,2,4,8 4 | tee gpurun_out/flight/probe.txt
