#!/bin/bash
# Phase timing of the round-4 attention kernel (audiotoken_amd/csrc/attention_f16x2_w8.hip, W8_DEBUG_STAMPS): builds a second library with the stamps
# compiled in (here, before gpurun), then on the GPU box runs tools/attn_bench.py with it and prints the per-interval averages of the first launches.
#   build:  bash tools/w8_stamps.sh build          run (gpurun):  bash tools/w8_stamps.sh run
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
C=$R/audiotoken_amd/csrc
if [ "$1" = "build" ]; then
  make -C $C -j8 > /dev/null || exit 1
  hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -ffp-contract=on -DW8_DEBUG_STAMPS -c $C/attention_f16x2_w8.hip -o $C/build/attention_f16x2_w8_dbg.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o $R/audiotoken_amd/lib/libaudiotoken_hip_w8dbg.so $(ls $C/build/*.o | grep -v -e attention_f16x2_w8.o -e _dbg.o) $C/build/attention_f16x2_w8_dbg.o
else
  export AUDIOTOKEN_HIP_LIB=$R/audiotoken_amd/lib/libaudiotoken_hip_w8dbg.so
  timeout 300 python3 $R/tools/attn_bench.py 2>&1 | grep "w8 stamps\|w8=" | head -12
fi
