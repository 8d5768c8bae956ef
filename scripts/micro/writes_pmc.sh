#!/bin/bash
# PMC passes over scripts/micro/writes.hip (built as build_variants/writes).  usage: scripts/micro/writes_pmc.sh
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/writes
for k in 0 1 2; do
  rm -rf gpurun_out/writes/p
  timeout -s KILL 60 rocprofv3 --pmc TCC_WRITE_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d gpurun_out/writes/p -- build_variants/writes $k > gpurun_out/writes/run$k.log 2>&1 || { tail -3 gpurun_out/writes/run$k.log; exit 1; }
  grep "^kernel" gpurun_out/writes/run$k.log
  python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/writes/p/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if any(s in row["Kernel_Name"] for s in ("scratch_rw", "global_rw", "atomic_f32")):
            print("   ", row["Kernel_Name"][:20], row["Counter_Name"], row["Counter_Value"])
PY
done
rm -rf gpurun_out/writes/p
