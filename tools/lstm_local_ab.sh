#!/bin/bash
# A/B of the persistent LSTM's XCD-local hand-off (AUDIOTOKEN_LSTM_LOCAL=0: memory-side protocol everywhere) on one box + the acoustic tests on it
mkdir -p gpurun_out/lstm_local
for v in 0 1 0 1; do
  AUDIOTOKEN_LSTM_LOCAL=$v timeout 300 python3 bench.py --workload acoustic --steps 10 --warmup 2 --no-cpu-baseline --no-verify > gpurun_out/lstm_local/l$v.json 2> gpurun_out/lstm_local/l$v.err
  python3 - <<PY
import json
d = json.load(open("gpurun_out/lstm_local/l$v.json"))
s = d["acoustic"]
print("lstm_local $v:", s["ms_per_step"], "lstm_rec", s["breakdown"]["lstm_rec"]["ms_per_step"], "pinned", s.get("checksum_pinned"), "status", s.get("lstm_handoff_status"))
PY
done
timeout 900 python -m pytest tests/test_acoustic_gpu.py -m gpu -q 2>&1 | tail -3
