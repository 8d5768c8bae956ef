#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_forms; mkdir -p "$OUT"
for pf in "1 2" "1 3" "1 6" "1 7" "0 6"; do set -- $pf; p=$1; f=$2
  timeout -k 10 200 python bench.py --pipeline $p --frames-in-flight $f --steps 42 --no-cpu-baseline > "$OUT/p${p}_f${f}.json" 2> "$OUT/p${p}_f${f}.err" || exit 1
  python - "$OUT/p${p}_f${f}.json" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(sys.argv[1], "ms/step %.2f Grays/s %.3f kernel_ms %.2f" % (d["ms_per_step"], d["value"]/1e9, r["kernel_ms"]))
P
done
