#!/bin/bash
# what the driver runs at the end of a round, in its order: the GPU suite, smoke(), bench.py with its defaults
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_final; mkdir -p "$OUT"
timeout -k 10 600 python -u -m pytest tests/ -x -q -m gpu --timeout 300 --timeout-method thread > "$OUT/pytest_gpu.log" 2>&1 || { tail -30 "$OUT/pytest_gpu.log"; exit 1; }
tail -2 "$OUT/pytest_gpu.log"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1 || { tail -20 "$OUT/smoke.log"; exit 1; }
tail -1 "$OUT/smoke.log"
t0=$(date +%s)
timeout -k 10 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
echo "bench.py wall $(( $(date +%s) - t0 )) s"
tail -1 "$OUT/bench.json" | cut -c1-330
