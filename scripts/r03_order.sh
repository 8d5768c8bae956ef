#!/bin/bash
# experiment: hand-out order of the units of a frame (QD_UNIT_ORDER 0 = as is, 1 = bottom rows first, 2 = rows dealt over the XCD runs, bottom to top)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_order; mkdir -p "$OUT"
for wl in "" "--width 1920 --height 1080" "--workload kdtree_fixed.qdmg" "--workload hw9/dragon.qdmg --width 1920 --height 1080 --pipeline 0"; do
  for m in 0 1 2; do
    QD_UNIT_ORDER=$m timeout -k 10 300 python bench.py $wl --steps 30 --no-1080p > "$OUT/b.json" 2> "$OUT/b.err" || { tail -5 "$OUT/b.err"; exit 1; }
    python - "$OUT/b.json" "$wl order $m" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; p=d.get("parity") or {}
print(sys.argv[2], "| in flight %.2f ms/frame, lone %.2f ms, form %s, rays %d, parity max_err %s bad %s" % (d["ms_per_step"], r["kernel_ms"], d["config"]["schedule"]["form"], d["config"]["rays_per_frame"], p.get("max_err"), p.get("bad_px")), flush=True)
P
  done
done
