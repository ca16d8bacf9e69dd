#!/bin/bash
# One GPU-box call of the inner loop: the -m gpu suite (or a subset: $2...), a short bench and, optionally, the GEMM segment stamps.
#   gpurun -- bash tools/gpu_check.sh <tag> [pytest args...]
tag=${1:-check}; shift
out=gpurun_out/$tag; mkdir -p $out
if [ $# -eq 0 ]; then set -- tests -m gpu; fi
timeout 1500 python -m pytest "$@" -q -s > $out/pytest.log 2>&1; echo "pytest rc $?" | tee -a $out/pytest.log
grep -E "passed|failed|error" $out/pytest.log | tail -3
grep -E "parity-at-size|\[sweep\]|\[edge\]|\[N2\]|split gemm|conv split" $out/pytest.log > $out/parity_counts.txt
timeout 600 python bench.py --full-line --steps 10 --warmup 2 --workload both --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc $?"
python - <<PY
import json
try:
    d = json.load(open("$out/bench.json"))
    print("value", d["value"], "ms", d["ms_per_step"], "fallback_batches", d.get("fallback_batches"))
    for n in ("acoustic", "semantic_m"):
        s = d.get(n)
        if s:
            print(n, s.get("value"), s.get("ms_per_step"), "pinned", s.get("checksum_pinned"), {g: v["ms_per_step"] for g, v in s.get("breakdown", {}).items()})
    r = d.get("roofline", {})
    print("roofline", r.get("workload"), r.get("kernel"), "alg", r.get("achieved"), "frac_executed", r.get("frac_executed"))
    print("verify", d.get("verify"))
except Exception as e:
    print("bench parse failed", e); print(open("$out/bench.err").read()[-3000:])
PY
if [ -f audiotoken_amd/lib/libaudiotoken_hip_dbg.so ]; then bash tools/tg_stamps.sh run > $out/tg_stamps.txt 2>&1; head -60 $out/tg_stamps.txt; fi
if [ -f audiotoken_amd/lib/libaudiotoken_hip_axdbg.so ]; then bash tools/ax_stamps.sh run > $out/ax_stamps.txt 2>&1; cat $out/ax_stamps.txt; fi
