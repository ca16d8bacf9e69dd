#!/bin/bash
# Round 6 experiment (VERDICT round 5, next #5b): P.V with the probabilities as ONE fp16 piece (tools/_lib_p1.so = the library with attention_f16x2_w8.hip built
# with -DW8_P_PIECES=1: `hipcc <Makefile CXXFLAGS> -DW8_P_PIECES=1 -c attention_f16x2_w8.hip -o /tmp/a.o && hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_lib_p1.so
# $(ls audiotoken_amd/csrc/build/*.o | grep -v attention_f16x2_w8) /tmp/a.o`, built here before gpurun; the .so is git-ignored). Oracle in the loop: the attention operator against its oracle, the bench batches of semantic_m on both families, the fitted code book; then
# the time, interleaved with the product build.   gpurun --timeout 2400 -- bash tools/p1_experiment.sh
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/p1; mkdir -p $O
export AUDIOTOKEN_HIP_LIB=$R/tools/_lib_p1.so
timeout 1500 python -m pytest tests/test_semantic_gpu.py tests/test_fullsize_gpu.py::test_bench_batch_semantic_m_vs_oracle tests/test_fullsize_gpu.py::test_semantic_m_data_fitted_codebook tests/test_fullsize_gpu.py::test_semantic_m_full_depth_properties -m gpu -q -s > $O/tests.log 2>&1
echo "P1 tests rc $?"; tail -15 $O/tests.log | cut -c1-300
grep -i "differ\|hidden\|max err\|unexplained" $O/tests.log | cut -c1-260 | head -60
unset AUDIOTOKEN_HIP_LIB
for v in p2 p1 p2 p1; do
  if [ $v = p1 ]; then export AUDIOTOKEN_HIP_LIB=$R/tools/_lib_p1.so; else unset AUDIOTOKEN_HIP_LIB; fi
  python3 $R/bench.py --full-line --workload semantic_m --steps 5 --warmup 2 --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$v', d['ms_per_step'], 'checksum', d['semantic_m']['token_checksum'], d['semantic_m']['checksum_pinned'], {k:v['ms_per_step'] for k,v in d['semantic_m']['breakdown'].items()})"
done
