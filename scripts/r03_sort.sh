#!/bin/bash
# experiment: coherence sort of secondary generations (scripts/r03_sort.py), three workloads x modes 3 (control), 1, 2
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_sort; mkdir -p "$OUT"
# small case first: parity of the sorted path before anything large runs
QD_WF_SORT=1 timeout -k 10 120 python -u scripts/r03_sort.py lecture7_fixed.qdmg 640 480 2 > "$OUT/small.log" 2>&1 || { tail -5 "$OUT/small.log"; exit 1; }
tail -2 "$OUT/small.log"
for wl in "kdtree_fixed.qdmg" "synth" "kdtree_fixed.qdmg 3840 2160"; do
  for m in 3 1 2; do
    tag=$(echo "$wl" | tr ' ./' '___')_m$m
    QD_WF_SORT=$m timeout -k 10 200 python -u scripts/r03_sort.py $wl > "$OUT/$tag.log" 2>&1 || { tail -5 "$OUT/$tag.log"; exit 1; }
    tail -2 "$OUT/$tag.log"
  done
done
