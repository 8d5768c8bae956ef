#!/bin/bash
# the three bench lines that go with profiles/r03_*_pmc_summary.json (run after scripts/update_traffic.py has stamped the table)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_lines; mkdir -p "$OUT"
run() { name=$1; shift
  timeout -k 10 400 python -u bench.py "$@" > "$OUT/$name.json" 2> "$OUT/$name.err" || { tail -5 "$OUT/$name.err"; exit 1; }
  tail -1 "$OUT/$name.json" | cut -c1-160; }
run synth10m_4k_bench
run kdtree_fixed_1200x900_bench --workload kdtree_fixed.qdmg
run kdtree_fixed_1200x900_bench_whole --workload kdtree_fixed.qdmg --frame-parts 1 --no-cpu-baseline
run heightfield_3840x2160_bench --workload heightfield_fixed.qdmg --width 3840 --height 2160
run heightfield_3840x2160_bench_whole --workload heightfield_fixed.qdmg --width 3840 --height 2160 --frame-parts 1 --no-cpu-baseline --no-1080p
