#!/bin/bash
# Development aid: build_variants/libqdtrace_<name>.so with extra -D flags.  usage: scripts/build_variant.sh <name> [-DX=Y ...]
cd "$(dirname "$0")/.." && mkdir -p build_variants
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -mllvm -disable-machine-licm \
  -DQD_DEV_BUILD "$@" -I include -I raytracing_amd/csrc/device raytracing_amd/csrc/device/qdtrace.hip -o build_variants/libqdtrace_$name.so && echo "built $name"
