# A/B of the split-bf16 SEANet kernels on ONE box (box-to-box spread is 2-4 %): acoustic bench per $AUDIOTOKEN_X3_KERNELS mask
for m in 511 0 511 0; do
  AUDIOTOKEN_X3_KERNELS=$m timeout 300 python bench.py --full-line --workload acoustic --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab_$m.json
  python -c "
import json; d=json.load(open('gpurun_out/ab_$m.json')); print('mask $m', d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['breakdown'].items()})"
done
