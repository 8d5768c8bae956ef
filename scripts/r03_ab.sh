#!/bin/bash
# development aid: A/B of kernel variants on the bench scene.  usage: scripts/r03_ab.sh <variant> ... [-- bench args]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_ab; mkdir -p "$OUT"
VARS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VARS+=("$1"); shift; done; [ $# -gt 0 ] && shift
for v in "${VARS[@]}"; do
  timeout -k 10 240 python scripts/ab_bench.py build_variants/libqdtrace_$v.so --steps 40 --no-1080p "$@" > "$OUT/$v.json" 2> "$OUT/$v.err" || { echo "$v FAILED"; tail -5 "$OUT/$v.err"; exit 1; }
  python - "$OUT/$v.json" "$v" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; p=d.get("parity") or {}
print("%-10s ms/step %.2f Grays/s %.3f lone %.2f ms | visits %.3fG tests %.1fM | parity max_err %s bad %s | form %s" % (sys.argv[2], d["ms_per_step"], d["value"]/1e9, r["kernel_ms"], r["kd_nodes_visited"]/1e9, r["tris_tested"]/1e6, p.get("max_err"), p.get("bad_px"), d["config"]["schedule"]["form"]))
P
done
