timeout 600 python -m pytest tests/test_acoustic_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -3; timeout 300 python bench.py --full-line --workload acoustic --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/b_ac.json; python -c "
import json; d=json.load(open('gpurun_out/b_ac.json')); print(d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['breakdown'].items()})"
