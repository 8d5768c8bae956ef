#!/bin/bash
# development aid: the bench scene at maxTraceDepth 0 / 1 / 8 -- what a ray costs when every ray of a wave is a camera ray
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_depth; mkdir -p "$OUT"
for d in 0 1 2 8; do for p in 0 1; do
  timeout -k 10 200 python bench.py --depth $d --pipeline $p --steps 40 --no-cpu-baseline --no-1080p > "$OUT/d${d}_p$p.json" 2> "$OUT/d${d}_p$p.err" || { echo failed; tail -3 "$OUT/d${d}_p$p.err"; continue; }
  python - "$OUT/d${d}_p$p.json" $d $p <<'P'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; c=d["config"]
print("depth %s form %s: rays/frame %.2fM (primary %.2fM shadow %.2fM) ms/step %.2f lone %.2f Grays/s %.3f | visits/ray %.1f tests/ray %.2f" % (sys.argv[2], sys.argv[3], c["rays_per_frame"]/1e6, c["primary_rays"]/1e6, c["shadow_rays"]/1e6, d["ms_per_step"], r["kernel_ms"], d["value"]/1e9, r["kd_nodes_visited"]/c["rays_per_frame"], r["tris_tested"]/c["rays_per_frame"]))
P
done; done
