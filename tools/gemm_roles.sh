#!/bin/bash
# Per-role times of the two-group GEMM from a kernel trace of two semantic_m steps (tools/gemm_groups_from_trace.py):
#   gpurun -- bash tools/gemm_roles.sh <tag>     -> gpurun_out/<tag>/gemm_roles_from_trace.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r03_final}; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --full-line --workload semantic_m --steps 2 --warmup 1 --no-cpu-baseline --no-verify > $O/trace_bench.json 2> $O/trace.err
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/gemm_groups_from_trace.py $T > $O/gemm_roles_from_trace.txt
cat $O/gemm_roles_from_trace.txt
rm -rf $O/trace
