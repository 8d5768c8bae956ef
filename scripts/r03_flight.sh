#!/bin/bash
# frames in flight 4..7 on the wavefront-form workloads
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_flight; mkdir -p "$OUT"
for wl in "--workload kdtree_fixed.qdmg" "--workload heightfield_fixed.qdmg --width 3840 --height 2160" "--workload lecture7_fixed.qdmg --width 3840 --height 2160" ""; do
  for f in 3 4 5 6 7; do
    timeout -k 10 200 python bench.py $wl --frames-in-flight $f --steps 60 --no-cpu-baseline --no-1080p > "$OUT/b.json" 2> "$OUT/b.err" || { tail -5 "$OUT/b.err"; exit 1; }
    python - "$OUT/b.json" "$wl F=$f" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(sys.argv[2], "| in flight %.2f ms/frame, form %s" % (d["ms_per_step"], d["config"]["schedule"]["form"]), flush=True)
P
  done
done
