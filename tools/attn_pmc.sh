#!/bin/bash
# PMC passes (kernel-trace only, one counter set per run) over tools/attn_bench.py: both f16x2 attention kernels on the semantic_m shape.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
TAG=${1:-r4_attn_pmc}
export TMPDIR=/tmp
cd /tmp
O=$R/gpurun_out/$TAG
mkdir -p $O
run() { n=$1; shift
  rm -rf $O/$n
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -- python3 $R/tools/attn_bench.py > $O/$n.out 2> $O/$n.err
  echo "$n rc=$?"; }
run sq_busy SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
run sq_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_BUSY_CU_CYCLES
cd $R
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for n in ("sq_busy", "sq_lds"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(f"{O}/{n}/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (r["Dispatch_Id"], k)
            if key not in seen: seen.add(key); cnt[k] += 1
    for k in acc:
        if "attention" in k or "split" in k:
            print(n, k, "launches", cnt[k], {c: round(v / cnt[k]) for c, v in acc[k].items()})
    for f in glob.glob(f"{O}/{n}/**/*kernel_trace.csv", recursive=True):
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            d[r["Kernel_Name"][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, v in d.items():
            if "attention" in k or "split" in k:
                v.sort(); print(n, "trace", k, "n", len(v), "median us", v[len(v)//2] / 1000.0)
PY
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
