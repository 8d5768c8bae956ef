#!/bin/bash
# Profile evidence for one workload: bench line, rocprofv3 kernel-trace stats (frames one after the other, and the default
# command), PMC passes (separate runs per counter group) with per-frame sums.  Results under gpurun_out/<tag>/<name>_*.
# usage: scripts/gpu_profile.sh <tag> <name> <form 0|1> [bench args...]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; NAME=$2; FORM=$3; shift; shift; shift
mkdir -p "$OUT"
timeout -k 10 400 python -u bench.py "$@" > "$OUT/${NAME}_bench.json" 2> "$OUT/${NAME}_bench.err" || { tail -5 "$OUT/${NAME}_bench.err"; exit 1; }
tail -1 "$OUT/${NAME}_bench.json" | cut -c1-400
# the kernel statistics and counters below are taken on WHOLE frames (qd_set_frame_parts 1: one launch sequence per frame, so a
# kernel's durations add up to the frame); this is the bench line that goes with them (wavefront form: lone frames unsplit)
if [ "$FORM" = "1" ]; then
  timeout -k 10 400 python -u bench.py --frame-parts 1 --no-cpu-baseline --no-1080p "$@" > "$OUT/${NAME}_bench_whole.json" 2> "$OUT/${NAME}_bench.err" || { tail -5 "$OUT/${NAME}_bench.err"; exit 1; }
fi
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-1080p --frames-in-flight 1 --frame-parts 1 "$@" > "$OUT/${NAME}_ktrace_bench.json" 2> "$OUT/kt.err" || { tail -5 "$OUT/kt.err"; exit 1; }
find "$OUT/kt" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/${NAME}_kernel_stats_serial.csv"
rm -rf "$OUT/kt"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-1080p "$@" > "$OUT/${NAME}_ktrace_default_bench.json" 2> "$OUT/kt.err" || { tail -5 "$OUT/kt.err"; exit 1; }
find "$OUT/kt" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/${NAME}_kernel_stats_default.csv"
rm -rf "$OUT/kt"
head -4 "$OUT/${NAME}_kernel_stats_serial.csv"
run() { name=$1; shift; counters=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc $counters --output-format csv -d "$OUT/pmc/$name" -- python3 bench.py --pipeline $FORM --frame-parts 1 --steps 2 --warmup 1 --no-cpu-baseline --no-1080p --frames-in-flight 1 "$@" > "$OUT/pmc_$name.log" 2>&1
  echo "$name rc=$?"; }
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" "$@" || exit 1
run sq2 "SQ_WAVES SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "$@" || exit 1
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "$@" || exit 1
run fetch "FETCH_SIZE" "$@" || exit 1
run write "WRITE_SIZE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "$@" || exit 1
python3 scripts/pmc_summary.py "$OUT/pmc" "$OUT/${NAME}_pmc_summary.json" "$OUT/${NAME}_ktrace_bench.json"
rm -rf "$OUT/pmc"
