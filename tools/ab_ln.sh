#!/bin/bash
# A/B of two library builds on one box: kernel-trace stats of a short semantic_m bench per library (tools/_lib_old.so vs the in-tree build)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
export TMPDIR=/tmp; cd /tmp
for v in old new old new; do
  if [ $v = old ]; then export AUDIOTOKEN_HIP_LIB=$R/tools/_lib_old.so; else unset AUDIOTOKEN_HIP_LIB; fi
  python3 $R/bench.py --workload semantic_m --steps 5 --warmup 2 --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$v', d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['semantic_m']['breakdown'].items()})"
done
for v in old new; do
  if [ $v = old ]; then export AUDIOTOKEN_HIP_LIB=$R/tools/_lib_old.so; else unset AUDIOTOKEN_HIP_LIB; fi
  rm -rf /tmp/ab_$v; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$v -- python3 $R/bench.py --workload semantic_m --steps 3 --warmup 1 --no-cpu-baseline --no-verify > /dev/null 2>&1
  echo "== $v"; grep -h "layernorm\|split_blocked" $(find /tmp/ab_$v -name "*kernel_stats.csv") | cut -d, -f1-5 | cut -c1-70,150-400
done
