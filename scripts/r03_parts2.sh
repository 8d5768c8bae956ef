#!/bin/bash
# in-flight throughput with frame parts off / on, repeated (is the 2-3 % difference noise?)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_parts; mkdir -p "$OUT"
for wl in "--workload kdtree_fixed.qdmg" "--workload lecture7_fixed.qdmg --width 3840 --height 2160"; do
  for rep in 1 2 3; do for parts in 1 4; do
    timeout -k 10 200 python bench.py $wl --frame-parts $parts --steps 60 --no-cpu-baseline --no-1080p > "$OUT/b.json" 2> "$OUT/b.err" || { tail -5 "$OUT/b.err"; exit 1; }
    python - "$OUT/b.json" "$wl parts $parts" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(sys.argv[2], "| in flight %.2f ms/frame, lone %.2f ms, form %s" % (d["ms_per_step"], r["kernel_ms"], d["config"]["schedule"]["form"]), flush=True)
P
  done; done
done
