"""Reconciles bench.py's per-GROUP GEMM timings with rocprofv3's per-KERNEL average: the two-group GEMM is one kernel symbol for eight launches per
conformer layer (ffn1 a/b, ffn2 a/b, q/k/v, out, pointwise 1, pointwise 2) and, being persistent, one grid size — `rocprofv3 --stats` can only average
over all of them. This reads the kernel TRACE of the same run, walks the launch sequence of the semantic_m encodes (each layer: LayerNorm, GEMM, GEMM,
LayerNorm, GEMM, attention, GEMM, LayerNorm, GEMM, depthwise conv, GEMM, LayerNorm, GEMM, GEMM, LayerNorm) and prints the average duration per role.

    python tools/gemm_groups_from_trace.py <kernel_trace.csv>
"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
roles = defaultdict(list)
ROLE = ["ffn1_a (swish -> pieces)", "ffn1_b (residual)", "qkv (fused projection)", "out projection (residual)", "pointwise 1 (GLU)", "pointwise 2 (residual)",
        "ffn2_a (swish -> pieces)", "ffn2_b (residual)"]
i, n = 0, len(rows)
while i < n:
    # a conformer layer starts at a layernorm_split launch followed by two <false, 4, 8> GEMMs
    if "layernorm_split_kernel" in names[i] and i + 14 < n and "gemm_f16x2_tg_kernel<false, 4, 8>" in names[i + 1] and "gemm_f16x2_tg_kernel<false, 4, 8>" in names[i + 2] \
            and "relpos_attention" in names[i + 5] and "dwconv" in names[i + 9]:
        idx = [i + 1, i + 2, i + 4, i + 6, i + 8, i + 10, i + 12, i + 13]
        if all("gemm_f16x2_tg_kernel<false, 4, 8>" in names[j] for j in idx):
            for k, j in enumerate(idx):
                roles[ROLE[k]].append(dur[j])
            roles["attention"].append(dur[i + 5])
            roles["depthwise conv + LN + swish"].append(dur[i + 9])
            i += 14
            continue
    i += 1
tot = 0
print(f"{'role':32s} {'launches':>9s} {'avg ms':>9s}")
for k in ROLE + ["attention", "depthwise conv + LN + swish"]:
    v = roles[k]
    if v:
        print(f"{k:32s} {len(v):9d} {sum(v) / len(v) / 1e6:9.4f}")
g = [x for k in ROLE for x in roles[k]]
f = [x for k in ROLE if k.startswith("ffn") for x in roles[k]]
if g:
    print(f"all eight GEMM roles: {len(g)} launches, average {sum(g) / len(g) / 1e6:.4f} ms (what `--stats` averages, semantic_m share)")
    print(f"the four FFN GEMMs (bench.py group `ffn`): {len(f)} launches, average {sum(f) / len(f) / 1e6:.4f} ms")
