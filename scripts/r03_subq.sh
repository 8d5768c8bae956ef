#!/bin/bash
# experiment: sub-queues per XCD label in the one-kernel form (waves that share a SIMD take pixels from different runs of the frame)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_subq; mkdir -p "$OUT"
# parity first, on a small frame
QD_SUBQ=3 timeout -k 10 200 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 120 --timeout-method thread -k "scene_matches_oracle and (kdtree or lecture7)" > "$OUT/pytest.log" 2>&1 || { tail -20 "$OUT/pytest.log"; exit 1; }
tail -1 "$OUT/pytest.log"
for cfg in "1 3" "2 3" "3 3" "4 3" "3 10" "2 10" "4 10" "3 7"; do
  set -- $cfg
  QD_SUBQ=$1 QD_SUBQ_SHIFT=$2 timeout -k 10 200 python bench.py --pipeline 0 --steps 30 --no-1080p > "$OUT/b.json" 2> "$OUT/b.err" || { tail -5 "$OUT/b.err"; exit 1; }
  python - "$OUT/b.json" "subq $1 shift $2" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; p=d.get("parity") or {}
print(sys.argv[2], "| in flight %.2f ms/frame, lone %.2f ms, rays %d, parity max_err %s bad %s" % (d["ms_per_step"], r["kernel_ms"], d["config"]["rays_per_frame"], p.get("max_err"), p.get("bad_px")), flush=True)
P
done
