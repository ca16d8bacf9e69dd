#!/bin/bash
# A/B of the fused stage-1 kernel (seanet_res64down.hip) against the res64x3 + down64x3 pair: bit-identity test, then the acoustic bench with the option on / off.
#   gpurun -- bash tools/stage1_ab.sh
out=gpurun_out/stage1; mkdir -p $out
if [ -f tools/stage1_dbg.py ]; then timeout 300 python tools/stage1_dbg.py 2>&1 | tail -4; fi
timeout 900 python -m pytest tests/test_acoustic_gpu.py -m gpu -q -k "fused_stage1 or repeated or golden or range" > $out/pytest.log 2>&1; tail -5 $out/pytest.log
for v in 1 0 1; do
  timeout 600 python bench.py --full-line --acoustic-option fused_stage1=$v --workload acoustic --steps 10 --warmup 2 --no-cpu-baseline --no-verify > $out/b$v.json 2> $out/b$v.err
  python - <<PY
import json
try:
    d = json.load(open("$out/b$v.json")); a = d["acoustic"]
    print("fused_stage1 $v:", a["ms_per_step"], {g: v["ms_per_step"] for g, v in a["breakdown"].items() if g in ("res1", "down1", "res1_down1", "stage0_fused")}, "pinned", a.get("checksum_pinned"))
except Exception as e:
    print("parse failed", e); print(open("$out/b$v.err").read()[-2000:])
PY
done
