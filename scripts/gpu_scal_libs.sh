#!/bin/bash
# strong-scaling rehearsal (world 1 and 8) for several builds of libqdtrace.so
cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  echo "== $lib"
  QDTRACE_LIB=$PWD/$lib timeout -k 10 300 python -u scripts/shard_scaling.py 2240 1,8 2>&1 | grep world || exit 1
done
