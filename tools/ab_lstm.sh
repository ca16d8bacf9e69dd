# one-box A/B of the split-bf16 LSTM's two shapes (8 or 4 waves per workgroup) at full batch and single-clip latency
for w in 1 0 1 0; do
  AUDIOTOKEN_LSTM_X3_WAVES8=$w timeout 300 python bench.py --workload acoustic --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab_l$w.json
  python -c "
import json; d=json.load(open('gpurun_out/ab_l$w.json')); print('waves8=$w', d['ms_per_step'], d['breakdown']['lstm_rec']['ms_per_step'])"
  AUDIOTOKEN_LSTM_X3_WAVES8=$w timeout 200 python tools/latency_probe.py 2>&1 | grep acoustic
done
