#!/bin/bash
# Segment timing of the two-group GEMM (audiotoken_amd/csrc/gemm_f16x2_tg.hip, TG_DEBUG_STAMPS): builds a second library with the stamps compiled
# in (here, before gpurun), then on the GPU box runs one semantic_m bench step with it and prints the per-K-step averages of the first launches.
#   build:  bash tools/tg_stamps.sh build          run (gpurun):  bash tools/tg_stamps.sh run
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
C=$R/audiotoken_amd/csrc
if [ "$1" = "build" ]; then
  make -C $C -j8 > /dev/null || exit 1
  hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -ffp-contract=on -DTG_DEBUG_STAMPS -c $C/gemm_f16x2_tg.hip -o $C/build/gemm_f16x2_tg_dbg.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o $R/audiotoken_amd/lib/libaudiotoken_hip_dbg.so $(ls $C/build/*.o | grep -v -e gemm_f16x2_tg.o -e _dbg.o) $C/build/gemm_f16x2_tg_dbg.o
else
  export AUDIOTOKEN_HIP_LIB=$R/audiotoken_amd/lib/libaudiotoken_hip_dbg.so
  timeout 300 python3 $R/bench.py --full-line --workload semantic_m --steps 1 --warmup 0 --no-cpu-baseline 2>&1 >/dev/null | grep "tg stamps\|first tile\|all .* tiles" | head -96
fi
