#!/bin/bash
# lone frames of the wavefront form enqueued as concurrent parts (qd_set_frame_parts): GPU suite first, then lone / in-flight
# figures with parts off (1) and on (4) per workload
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_parts; mkdir -p "$OUT"
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 120 --timeout-method thread -k "parts or bands or cannot_fail or begun_ahead or first_frame" > "$OUT/pytest_parts.log" 2>&1 || { tail -30 "$OUT/pytest_parts.log"; exit 1; }
tail -2 "$OUT/pytest_parts.log"
for wl in "--workload kdtree_fixed.qdmg" "--workload lecture7_fixed.qdmg --width 3840 --height 2160" "--workload heightfield_fixed.qdmg --width 3840 --height 2160" "--workload window_fixed.qdmg --width 3840 --height 2160" "--workload kdtree_fixed.qdmg --width 3840 --height 2160" "--workload lecture7_fixed.qdmg" "--workload hw9/dragon.qdmg --width 1920 --height 1080"; do
  for parts in 1 4; do
    timeout -k 10 200 python bench.py $wl --frame-parts $parts --steps 30 --no-cpu-baseline --no-1080p > "$OUT/b.json" 2> "$OUT/b.err" || { tail -5 "$OUT/b.err"; exit 1; }
    python - "$OUT/b.json" "$wl parts $parts" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(sys.argv[2], "| in flight %.2f ms/frame, lone %.2f ms, form %s" % (d["ms_per_step"], r["kernel_ms"], d["config"]["schedule"]["form"]), flush=True)
P
  done
done
