#!/bin/bash
# Acoustic inner loop: the acoustic GPU tests, then the acoustic bench twice (ms per step, kernel groups, pinned checksum).
#   gpurun -- bash tools/acoustic_check.sh [bench args...]
out=gpurun_out/acoustic_check; mkdir -p $out
timeout 1200 python -m pytest tests/test_acoustic_gpu.py -m gpu -q > $out/pytest.log 2>&1; tail -4 $out/pytest.log
for v in 1 2; do
  timeout 600 python bench.py --full-line --workload acoustic --steps 10 --warmup 2 --no-cpu-baseline --no-verify "$@" > $out/b$v.json 2> $out/b$v.err
  python - <<PY
import json
try:
    d = json.load(open("$out/b$v.json")); a = d["acoustic"]
    print("run $v:", a["ms_per_step"], {g: v["ms_per_step"] for g, v in a["breakdown"].items()}, "pinned", a.get("checksum_pinned"))
except Exception as e:
    print("parse failed", e); print(open("$out/b$v.err").read()[-2000:])
PY
done
