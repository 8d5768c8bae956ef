#!/bin/bash
# one PMC pass over scripts/prof_run.py: scripts/pmc_one.sh <name> "<counters>" [mesh-n]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$1
mkdir -p "$OUT"
timeout -s KILL 240 rocprofv3 --pmc $2 --output-format csv -d "$OUT" -- python3 scripts/prof_run.py ${3:-2240} > "$OUT/run.log" 2>&1
echo "rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
res = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "trace_kernel" in row["Kernel_Name"]:
            res[row["Counter_Name"]].append(float(row["Counter_Value"]))
for c, v in sorted(res.items()): print(c, sum(v) / len(v), len(v))
PY
