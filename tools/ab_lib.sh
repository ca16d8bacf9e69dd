#!/bin/bash
# A/B of two library builds on ONE box: tools/_lib_old.so (built from the stashed tree) vs the in-tree build, interleaved semantic_m benches
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
W=${1:-semantic_m}
for v in old new old new; do
  if [ $v = old ]; then export AUDIOTOKEN_HIP_LIB=$R/tools/_lib_old.so; else unset AUDIOTOKEN_HIP_LIB; fi
  python3 $R/bench.py --full-line --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$v', d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['$W']['breakdown'].items()})"
done
