#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_parts; mkdir -p "$OUT"
for parts in 1 4; do
  rm -rf "$OUT/kt"
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -- python3 scripts/r03_parts5.py $parts > "$OUT/burst_$parts.log" 2> "$OUT/kt.err" || { tail -5 "$OUT/kt.err"; exit 1; }
  cat "$OUT/burst_$parts.log"
  find "$OUT/kt" -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} "$OUT/burst_${parts}_kernel_trace.csv"
  rm -rf "$OUT/kt"
done
ls -la "$OUT"
