#!/bin/bash
# sweep of the device KD builder's leaf size / step cost (bench.py --kd-refine) on the bench scene and kdtree_fixed
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_kdparams; mkdir -p "$OUT"
for wl in "" "--workload kdtree_fixed.qdmg"; do
  for kr in "2,0.6" "1,0.6" "3,0.6" "4,0.6" "2,0.3" "2,0.45" "2,0.8" "2,1.0" "3,0.45" "1,0.45"; do
    timeout -k 10 200 python bench.py $wl --kd-refine $kr --steps 30 --no-cpu-baseline --no-1080p > "$OUT/b.json" 2> "$OUT/b.err" || { tail -5 "$OUT/b.err"; exit 1; }
    python - "$OUT/b.json" "$wl kd-refine $kr" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(sys.argv[2], "| in flight %.2f ms/frame, lone %.2f ms, form %s, visits %.3f G tests %.1f M, kd %s" % (d["ms_per_step"], r["kernel_ms"], d["config"]["schedule"]["form"], r["kd_nodes_visited"]/1e9, r["tris_tested"]/1e6, d["config"]["setup_s"]), flush=True)
P
  done
done
