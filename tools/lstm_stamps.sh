#!/bin/bash
# Phase breakdown of the persistent LSTM step (audiotoken_amd/csrc/lstm_seq_x3.hip, LX_DEBUG_STAMPS): builds a second library with the stamps
# compiled in (here, before gpurun: hipcc cross-compiles), then on the GPU box runs two acoustic bench steps with it and prints the averages.
#   build:  bash tools/lstm_stamps.sh build          run (gpurun):  bash tools/lstm_stamps.sh run
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
C=$R/audiotoken_amd/csrc
if [ "$1" = "build" ]; then
  make -C $C -j8 > /dev/null || exit 1
  hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -ffp-contract=on -DLX_DEBUG_STAMPS -c $C/lstm_seq_x3.hip -o $C/build/lstm_seq_x3_dbg.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o $R/audiotoken_amd/lib/libaudiotoken_hip_dbg.so $(ls $C/build/*.o | grep -v -e lstm_seq_x3.o -e lstm_seq_x3_dbg.o) $C/build/lstm_seq_x3_dbg.o
else
  export AUDIOTOKEN_HIP_LIB=$R/audiotoken_amd/lib/libaudiotoken_hip_dbg.so
  timeout 300 python3 $R/bench.py --full-line --steps 2 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep "lstm stamps" | tail -4
fi
