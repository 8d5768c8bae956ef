#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_dragon; mkdir -p "$OUT"
for p in 0 1 2; do
  timeout -k 10 200 python bench.py --workload hw9/dragon.qdmg --width 3840 --height 2160 --pipeline $p --steps 20 --no-cpu-baseline --no-1080p > "$OUT/p$p.json" 2> "$OUT/p$p.err" || { tail -3 "$OUT/p$p.err"; exit 1; }
  python - "$OUT/p$p.json" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(sys.argv[1], "ms/step %.2f Grays/s %.3f kernel_ms %.2f form %s" % (d["ms_per_step"], d["value"]/1e9, r["kernel_ms"], d["config"]["schedule"]))
P
done
