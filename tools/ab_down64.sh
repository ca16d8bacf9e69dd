# one-box A/B of seanet_down64x3_kernel's two shapes (8 or 4 waves per workgroup)
for w in 1 0 1 0; do
  AUDIOTOKEN_DOWN64_WAVES8=$w timeout 300 python bench.py --workload acoustic --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab_d$w.json
  python -c "
import json; d=json.load(open('gpurun_out/ab_d$w.json')); print('waves8=$w', d['ms_per_step'], d['breakdown']['down1']['ms_per_step'])"
done
