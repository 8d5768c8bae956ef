#!/bin/bash
# development aid: the bench scene in both scheduling forms, frames in flight 1 and 4
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_forms; mkdir -p "$OUT"
for p in 0 1; do for f in 1 4; do
  timeout -k 10 200 python bench.py --pipeline $p --frames-in-flight $f --steps 30 --no-cpu-baseline "$@" > "$OUT/p${p}_f${f}.json" 2> "$OUT/p${p}_f${f}.err" || exit 1
  python - "$OUT/p${p}_f${f}.json" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(sys.argv[1], "ms/step %.2f Grays/s %.3f kernel_ms %.2f setup %s" % (d["ms_per_step"], d["value"]/1e9, r["kernel_ms"], d["config"]["setup_s"]))
P
done; done
