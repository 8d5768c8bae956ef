#!/bin/bash
# experiment: a lone frame of the one-kernel form as 2 / 4 concurrent parts (contiguous bands of the image, a share of the waves each)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_form0_parts; mkdir -p "$OUT"
for wl in "" "--width 1920 --height 1080" "--workload hw9/dragon.qdmg --width 3840 --height 2160 --pipeline 0" "--workload bumpmap_fixed.qdmg --width 3840 --height 2160"; do
  for k in 1 2 4; do
    QD_FORM0_PARTS=$k timeout -k 10 200 python bench.py $wl --steps 30 --no-cpu-baseline --no-1080p > "$OUT/b.json" 2> "$OUT/b.err" || { tail -5 "$OUT/b.err"; exit 1; }
    python - "$OUT/b.json" "$wl form0 parts $k" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; p=d.get("parity") or {}
print(sys.argv[2], "| in flight %.2f ms/frame, lone %.2f ms, form %s, lone parts %s, rays %d" % (d["ms_per_step"], r["kernel_ms"], d["config"]["schedule"]["form"], r.get("lone_frame_parts"), d["config"]["rays_per_frame"]), flush=True)
P
  done
done
