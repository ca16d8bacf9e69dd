#!/bin/bash
# 8-rank readiness on a ONE-GPU box. Eight ranks, all on cuda:0, gloo collectives through host memory (--shared-device): bench.py's
# self-launch of 8 children, the EnCodec weight broadcast and the packed-model export -> broadcast to 7 receivers -> import of both semantic tokenizers, the
# start-up probe (every rank's checksum == rank 0's), per-rank clip shards, the barrier-bracketed timed region, max over ranks, the rank all-gather, the files
# leg with 8 processes x 8 decode threads; then tools/n8_shared_dir.py: LPT sharding of ONE shared directory of mixed-length files, token files byte-identical
# to world 1. The numbers are NOT a measurement (8 processes share one device); the result goes to profiles/r05_n8_rehearsal.txt.
#   gpurun --timeout 2400 -- bash tools/n8_rehearsal.sh
out=gpurun_out/n8; mkdir -p $out
N=${N:-8}
run() { name=$1; shift; timeout 1500 python bench.py --full-line --gpus $N --backend gloo --shared-device --no-cpu-baseline --no-verify "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc $?"; }
run both --steps 2 --warmup 1 --batch 16 --sem-batch 4 --sem-layers 3 --workload both
run hub --steps 2 --warmup 1 --hub-batch 4 --workload semantic_s
run files --workload files --files-acoustic 64 --files-acoustic-batch 16 --files-semantic 8 --files-semantic-batch 4 --files-semantic-s 16 --files-semantic-s-batch 8
python - <<PY
import json
for n in ("both", "hub", "files"):
    try:
        d = json.load(open("$out/%s.json" % n))
        print(n, "n_gpus", d["n_gpus"], "value", d.get("value"), "ms", d.get("ms_per_step"), "ranks", d.get("rccl_ranks"), "per_rank_ms", d.get("per_rank_ms"))
        for k in ("acoustic", "semantic_m", "semantic_s"):
            s = d.get(k)
            if isinstance(s, dict):
                print("  ", k, "ms", s.get("ms_per_step"), "checksum", s.get("token_checksum"), "rank_probe", s.get("rank_probe"), "finalize_ms", s.get("finalize_ms"), "export_ms", s.get("export_ms"), "broadcast_ms", s.get("broadcast_ms"))
        for leg in (d.get("files") or {}).get("legs", []):
            print("   files", leg["tokenizer"], leg["file"][-30:], "value", leg["value"], "token files per rank", leg["token_files_written"], "host_seconds", leg["host_seconds"])
    except Exception as e:
        print(n, "parse failed", e); print(open("$out/%s.err" % n).read()[-3000:])
PY
for tk in acoustic semantic_s; do
  timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 tools/n8_shared_dir.py /dev/shm/n8_$tk $tk 384 8 2> $out/shared_$tk.err | grep "shared-dir" ; echo "shared_dir $tk rc ${PIPESTATUS[0]}"
  rm -rf /dev/shm/n8_$tk
done
