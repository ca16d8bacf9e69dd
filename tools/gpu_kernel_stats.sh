#!/bin/bash
# rocprofv3 kernel-trace stats of one bench workload; prints the top kernels. usage: tools/gpu_kernel_stats.sh <workload> <tag> [steps]
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
W=${1:-semantic_m}; TAG=${2:-stats}; STEPS=${3:-2}
export TMPDIR=/tmp
cd /tmp
rm -rf $R/gpurun_out/$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -- python3 $R/bench.py --full-line --workload $W --steps $STEPS --warmup 1 --no-cpu-baseline > $R/gpurun_out/$TAG.json 2> $R/gpurun_out/$TAG.err
cd $R
f=$(find gpurun_out/$TAG -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/${TAG}_kernel_stats.csv
head -14 $f | cut -c1-150
find gpurun_out/$TAG -name "*.csv" -size +1M -delete; find gpurun_out/$TAG -name "*.db" -delete
python3 -c "
import json;d=json.load(open('gpurun_out/$TAG.json'));print(d['ms_per_step'],d['token_checksum']);[print(k,v['ms_per_step']) for k,v in d['breakdown'].items()]"
