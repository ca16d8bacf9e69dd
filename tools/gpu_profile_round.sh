#!/bin/bash
# Round-end evidence run on the GPU box: full bench, rocprofv3 kernel-trace stats of the same command, and the two PMC
# passes for HBM traffic (counters collected in their own runs, kernel-trace only). Outputs under gpurun_out/.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
export TMPDIR=/tmp
cd /tmp
python3 $R/bench.py --steps 3 --warmup 1 > $R/gpurun_out/bench_v4.json 2> $R/gpurun_out/bench_v4.err
rm -rf $R/gpurun_out/prof_v4 $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_v4 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_v4.json 2> $R/gpurun_out/prof_v4.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --workload acoustic --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --workload acoustic --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc_write.err
cd $R
python3 tools/collect_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/r01 | tail -40
# the raw per-dispatch counter files are large; keep only the summaries
find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*.csv" -size +2M -delete
cat gpurun_out/bench_v4.json | cut -c1-600
