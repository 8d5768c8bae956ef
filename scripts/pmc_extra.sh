#!/bin/bash
# One extra PMC pass over a bench workload: where the L2's writes to memory come from (write-backs vs atomics vs
# uncached writes).  usage: scripts/pmc_extra.sh <tag> <form> "<counters>" [bench args...]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; FORM=$2; CTR=$3; shift; shift; shift
mkdir -p "$OUT"; rm -rf "$OUT/pmcx"
timeout -s KILL 240 rocprofv3 --pmc $CTR --output-format csv -d "$OUT/pmcx" -- python3 bench.py --pipeline $FORM --steps 2 --warmup 1 --no-cpu-baseline --frames-in-flight 1 "$@" > "$OUT/pmcx.log" 2>&1 || { tail -5 "$OUT/pmcx.log"; exit 1; }
python3 - "$OUT/pmcx" <<'PY'
import collections, csv, glob, sys
res = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "<true" in k: continue
        fam = "walk" if "wf_walk" in k else "shade" if "wf_shade" in k else "trace" if "trace_kernel" in k else None
        if fam is None: continue
        res[fam][row["Counter_Name"]] += float(row["Counter_Value"])
for fam, d in res.items(): print(fam, {c: f"{v:.4g}" for c, v in d.items()})
PY
rm -rf "$OUT/pmcx"
