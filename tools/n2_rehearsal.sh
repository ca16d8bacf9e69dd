#!/bin/bash
# Rehearsal of bench.py's N > 1 path on a one-GPU box: two ranks, both on cuda:0, gloo collectives through host memory (--shared-device). Exercises the
# self-launch, the packed-model export / broadcast / import of the semantic tokenizers, the per-rank clip shards, the barrier-bracketed timed region,
# max-over-ranks and the rank all-gather. The numbers are NOT a measurement (two processes share one device).
#   gpurun -- bash tools/n2_rehearsal.sh
out=gpurun_out/n2; mkdir -p $out
timeout 1200 python bench.py --full-line --gpus 2 --backend gloo --shared-device --steps 2 --warmup 1 --no-cpu-baseline --no-verify --batch 32 --sem-batch 8 --sem-layers 3 --workload both > $out/both.json 2> $out/both.err; echo "both rc $?"
timeout 1200 python bench.py --full-line --gpus 2 --backend gloo --shared-device --steps 2 --warmup 1 --no-cpu-baseline --no-verify --hub-batch 8 --workload semantic_s > $out/hub.json 2> $out/hub.err; echo "semantic_s rc $?"
python - <<PY
import json
for n in ("both", "hub"):
    try:
        d = json.load(open("$out/%s.json" % n))
        print(n, "n_gpus", d["n_gpus"], "value", d["value"], "ms", d["ms_per_step"], "ranks", d.get("rccl_ranks"), "per_rank_ms", d.get("per_rank_ms"))
        for k in ("acoustic", "semantic_m", "semantic_s"):
            s = d.get(k)
            if isinstance(s, dict):
                print("  ", k, "ms", s.get("ms_per_step"), "checksum", s.get("token_checksum"), "finalize_ms", s.get("finalize_ms"), "export_ms", s.get("export_ms"), "broadcast_ms", s.get("broadcast_ms"))
    except Exception as e:
        print(n, "parse failed", e); print(open("$out/%s.err" % n).read()[-3000:])
PY
# the files leg on two ranks (every rank its own generated files; LPT sharding is exercised by tests/test_distributed_cpu.py)
timeout 1200 python bench.py --full-line --gpus 2 --backend gloo --shared-device --workload files --files-acoustic 256 --files-acoustic-batch 64 --files-semantic 0 --no-cpu-baseline --no-verify > $out/files.json 2> $out/files.err; echo "files rc $?"
python - <<PY
import json
try:
    d = json.load(open("$out/files.json"))
    for leg in d["files"]["legs"]:
        print("files", leg["tokenizer"], leg["file"][-28:], "value", leg["value"], "token files", leg["token_files_written"])
except Exception as e:
    print("files parse failed", e); print(open("$out/files.err").read()[-3000:])
PY
# the default workload ("all": both tokenizers + decode + semantic_s + files) at reduced sizes
timeout 1200 python bench.py --full-line --gpus 2 --backend gloo --shared-device --steps 2 --warmup 1 --no-cpu-baseline --no-verify --batch 32 --sem-batch 8 --sem-layers 3 --hub-batch 8 --files-acoustic 64 --files-acoustic-batch 32 --files-semantic 0 > $out/all.json 2> $out/all.err; echo "all rc $?"
python - <<PY
import json
try:
    d = json.load(open("$out/all.json"))
    print("all: n_gpus", d["n_gpus"], "value", d["value"], "keys", [k for k in ("acoustic", "semantic_m", "semantic_s", "acoustic_decode", "files") if k in d],
          "decode ms", d.get("acoustic_decode", {}).get("ms_per_step"), "decode err", d.get("acoustic_decode", {}).get("error"))
except Exception as e:
    print("all parse failed", e); print(open("$out/all.err").read()[-3000:])
PY
