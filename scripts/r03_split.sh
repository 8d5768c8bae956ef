#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_split; mkdir -p "$OUT"
for wl in "synth" "synth 1920 1080" "kdtree_fixed.qdmg" "lecture7_fixed.qdmg 3840 2160" "heightfield_fixed.qdmg 3840 2160" "kdtree_fixed.qdmg 3840 2160"; do
  tag=$(echo "$wl" | tr ' ./' '___')
  timeout -k 10 200 python -u scripts/r03_split.py $wl > "$OUT/$tag.log" 2>&1 || { tail -5 "$OUT/$tag.log"; exit 1; }
  tail -2 "$OUT/$tag.log"
done
