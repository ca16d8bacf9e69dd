#!/bin/bash
# Copies the evidence of `tools/gpu_profile_round.sh <tag>` (merged back under gpurun_out/<tag>/) into profiles/<tag>_* and derives the PMC tables
# and the traffic file bench.py reads:   bash tools/refresh_profiles.sh r03_final
T=${1:-r04_final}; G=gpurun_out/$T
cp $G/bench.json profiles/${T}_bench.json
[ -f $G/bench_line.json ] && cp $G/bench_line.json profiles/${T}_bench_line.json
cp $G/kernel_stats.csv profiles/${T}_kernel_stats.csv
cp $G/pmc_acoustic/pmc_summary.csv profiles/${T}_acoustic_pmc_summary.csv
cp $G/pmc_semantic_m/pmc_summary.csv profiles/${T}_semantic_m_pmc_summary.csv
[ -f $G/pmc_semantic_s/pmc_summary.csv ] && cp $G/pmc_semantic_s/pmc_summary.csv profiles/${T}_semantic_s_pmc_summary.csv
[ -f $G/pmc_decode/pmc_summary.csv ] && cp $G/pmc_decode/pmc_summary.csv profiles/${T}_decode_pmc_summary.csv
[ -f $G/pmc_semantic_m/gemm_roles_traffic.json ] && cp $G/pmc_semantic_m/gemm_roles_traffic.json profiles/${T}_gemm_roles_traffic.json
[ -f $G/gemm_roles_from_trace.txt ] && cp $G/gemm_roles_from_trace.txt profiles/${T}_gemm_roles_from_trace.txt
[ -f $G/trace_gaps.txt ] && cp $G/trace_gaps.txt profiles/${T}_trace_gaps.txt
python3 tools/pmc_report.py profiles/$T acoustic=profiles/${T}_acoustic_pmc_summary.csv semantic_m=profiles/${T}_semantic_m_pmc_summary.csv $( [ -f profiles/${T}_semantic_s_pmc_summary.csv ] && echo semantic_s=profiles/${T}_semantic_s_pmc_summary.csv ) $( [ -f profiles/${T}_decode_pmc_summary.csv ] && echo acoustic_decode=profiles/${T}_decode_pmc_summary.csv ) > /dev/null
python3 - "$T" <<'PY'
import csv, json, sys
T = sys.argv[1]
for w in ('acoustic', 'semantic_m'):   # (HBM GB per step)
    rows = list(csv.DictReader(open(f'profiles/{T}_{w}_pmc_derived.csv')))
    calls = [int(r['launches']) for r in rows if ('stage0' in r['kernel'] or 'fbank_stats' in r['kernel'])][0]
    tot = sum(int(r['launches']) * float(r['hbm_bytes_per_launch']) for r in rows if r['hbm_bytes_per_launch'] not in ('None', ''))
    print(w, 'encode calls in the PMC run', calls, '-> HBM GB per step', round(tot / calls / 1e9, 2))
d = json.loads(open(f'profiles/{T}_bench.json').readline())
print('step', d['ms_per_step'], 'ms', d['value'], d['unit'], '(' + d['metric'] + ')')
for k in ('acoustic', 'semantic_m', 'semantic_s', 'acoustic_decode'):
    if k in d:
        print(k, {kk: d[k][kk] for kk in d[k] if kk in ('value', 'ms_per_step')}, {g: v['ms_per_step'] for g, v in d[k].get('breakdown', {}).items()})
PY
