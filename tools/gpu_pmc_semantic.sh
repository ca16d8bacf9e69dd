#!/bin/bash
# PMC passes over one semantic_m step (64 x 30 s, 19 layers): HBM traffic and the SQ busy / wait / LDS counters of the GEMM,
# attention and depthwise-conv kernels. Each --pmc set is its own run with kernel-trace only (MI355X_MICROARCH.md, PMC slots).
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
W=${1:-semantic_m}
TAG=${2:-r03_$W}
export TMPDIR=/tmp
cd /tmp
O=$R/gpurun_out/$TAG
mkdir -p $O
rocprofv3 -L > $O/counters_list.txt 2>&1
run() { # name, counters...
  n=$1; shift
  keep=""
  for c in "$@"; do if grep -qw "$c" $O/counters_list.txt; then keep="$keep $c"; else echo "counter $c not offered on this box"; fi; done
  set -- $keep
  rm -rf $O/$n
  timeout 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -- python3 $R/bench.py --full-line --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-verify > $O/$n.json 2> $O/$n.err
  echo "$n rc=$?"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq_busy SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
run sq_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_BUSY_CU_CYCLES
cd $R
python3 tools/collect_pmc.py $O/pmc_summary.csv $O/fetch $O/write $O/sq_busy $O/sq_lds > $O/collect.log 2>&1
# per-ROLE traffic of the one GEMM kernel symbol (needs the per-dispatch rows: before the large CSVs are deleted)
if [ "$W" = "semantic_m" ]; then python3 tools/gemm_roles_pmc.py $O/gemm_roles_traffic.json $O/fetch $O/write > $O/gemm_roles.log 2>&1; cat $O/gemm_roles.log; fi
find $O -name "*.csv" -size +1M -delete
find $O -name "*.db" -delete
tail -3 $O/*.err | tail -40
head -c 3000 $O/pmc_summary.csv
