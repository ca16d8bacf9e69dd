#!/bin/bash
# Phase timing of the attention kernel (audiotoken_amd/csrc/attention_bf16x3.hip, AX_DEBUG_STAMPS): builds a second library with the stamps compiled
# in (here, before gpurun), then on the GPU box runs one semantic_m bench step with it and prints the per-tile averages of the first launches.
#   build:  bash tools/ax_stamps.sh build          run (gpurun):  bash tools/ax_stamps.sh run
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
C=$R/audiotoken_amd/csrc
if [ "$1" = "build" ]; then
  make -C $C -j8 > /dev/null || exit 1
  hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -ffp-contract=on -DAX_DEBUG_STAMPS -c $C/attention_bf16x3.hip -o $C/build/attention_bf16x3_dbg.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o $R/audiotoken_amd/lib/libaudiotoken_hip_axdbg.so $(ls $C/build/*.o | grep -v -e attention_bf16x3.o -e _dbg.o) $C/build/attention_bf16x3_dbg.o
else
  export AUDIOTOKEN_HIP_LIB=$R/audiotoken_amd/lib/libaudiotoken_hip_axdbg.so
  timeout 300 python3 $R/bench.py --full-line --workload semantic_m --steps 1 --warmup 0 --no-cpu-baseline --no-verify 2>&1 >/dev/null | grep "ax stamps" | head -8
fi
