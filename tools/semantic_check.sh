#!/bin/bash
# Semantic inner loop: the semantic / ops GPU tests, then the semantic_m bench (ms per step, kernel groups, pinned checksum).
#   gpurun -- bash tools/semantic_check.sh [pytest -k expression]
out=gpurun_out/semantic_check; mkdir -p $out
timeout 1500 python -m pytest tests/test_semantic_gpu.py tests/test_hubert.py tests/test_packed_gpu.py tests/test_ops_gpu.py -m gpu -q ${1:+-k "$1"} > $out/pytest.log 2>&1; tail -4 $out/pytest.log
for v in 1 2; do
  timeout 900 python bench.py --full-line --workload semantic_m --steps 5 --warmup 1 --no-cpu-baseline --no-verify > $out/b$v.json 2> $out/b$v.err
  python - <<PY
import json
try:
    d = json.load(open("$out/b$v.json")); a = d.get("semantic_m", d)
    print("run $v:", a["ms_per_step"], {g: v["ms_per_step"] for g, v in a["breakdown"].items()}, "pinned", a.get("checksum_pinned"))
except Exception as e:
    print("parse failed", e); print(open("$out/b$v.err").read()[-2000:])
PY
done
