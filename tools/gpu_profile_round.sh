#!/bin/bash
# Round evidence run on the GPU box: full bench, rocprofv3 kernel-trace stats of the same command, and the PMC passes (counters are
# collected in their own runs, kernel-trace only) for the acoustic and the semantic_m workloads. Outputs under gpurun_out/<tag>/.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
TAG=${1:-r04_final}
export TMPDIR=/tmp
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp
# exactly what the driver runs: stdout = the compact line alone (kept as bench_line.json), full object -> bench.json (and stderr)
python3 $R/bench.py --steps 20 --warmup 5 --detail-out $O/bench.json > $O/bench_stdout.txt 2> $O/bench.err
tail -n 1 $O/bench_stdout.txt > $O/bench_line.json; rm -f $O/bench_stdout.txt
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --full-line --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $O/stats_bench.json 2> $O/stats.err
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
cd $R
bash tools/gpu_pmc_semantic.sh acoustic $TAG/pmc_acoustic > $O/pmc_acoustic.log 2>&1
bash tools/gpu_pmc_semantic.sh semantic_m $TAG/pmc_semantic_m > $O/pmc_semantic_m.log 2>&1
bash tools/gpu_pmc_semantic.sh semantic_s $TAG/pmc_semantic_s > $O/pmc_semantic_s.log 2>&1
bash tools/gpu_pmc_semantic.sh decode $TAG/pmc_decode > $O/pmc_decode.log 2>&1
python3 tools/gemm_groups_from_trace.py $(find $O/stats -name "*kernel_trace.csv" | head -1) > $O/gemm_roles_from_trace.txt 2>&1
# idle time between kernels inside an encode: a kernel trace of each workload alone (3 encodes), tools/trace_gaps.py
for wm in semantic_m:vq_argmax_kernel acoustic:rvq_encode semantic_s:vq_argmax_kernel; do
  w=${wm%%:*}; m=${wm##*:}
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/gaps_$w -- python3 $R/bench.py --full-line --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-verify > /dev/null 2> $O/gaps_$w.err)
  echo "== $w" >> $O/trace_gaps.txt
  python3 tools/trace_gaps.py $(find $O/gaps_$w -name "*kernel_trace.csv" | head -1) $m >> $O/trace_gaps.txt 2>&1
done
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
cut -c1-400 $O/bench.json
