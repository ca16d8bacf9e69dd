"""Role-resolved HBM traffic of the split GEMM: `gemm_f16x2_tg_kernel<false, 4, 8>` is ONE kernel symbol for the eight GEMM
launches of a conformer layer (ffn1 a / b, q/k/v, out, pointwise 1 / 2, ffn2 a / b), so a per-kernel PMC table can only average over all of them. This reads
the PER-DISPATCH counter rows of the FETCH_SIZE and WRITE_SIZE passes (tools/gpu_pmc_semantic.sh: separate --pmc runs, kernel-trace only), walks the
dispatch sequence of the semantic_m encodes exactly like tools/gemm_groups_from_trace.py (a layer = LayerNorm, GEMM, GEMM, LayerNorm, GEMM, attention, GEMM,
LayerNorm, GEMM, depthwise conv, GEMM, LayerNorm, GEMM, GEMM) and writes bytes per launch per ROLE and per bench.py group:
    bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   (gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream; MI355X_MICROARCH.md)

    python tools/gemm_roles_pmc.py <out.json> <fetch pass dir> <write pass dir>
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROLE = ["ffn1_a", "ffn1_b", "qkv", "out", "pw1", "pw2", "ffn2_a", "ffn2_b"]
GROUP = {"ffn": ["ffn1_a", "ffn1_b", "ffn2_a", "ffn2_b"], "attn_proj": ["qkv", "out"], "conv_module": ["pw1", "pw2"]}
OPERAND_BYTES = {   # algorithmic bytes per launch at M = 96 000: A pieces (4 B / element) + W pieces + output (+ residual)
    "ffn1_a": 96000 * 1024 * 4 + 4096 * 1024 * 4 + 96000 * 4096 * 4, "ffn1_b": 96000 * 4096 * 4 + 1024 * 4096 * 4 + 2 * 96000 * 1024 * 4,
    "qkv": 96000 * 1024 * 4 + 3072 * 1024 * 4 + 96000 * 1024 * 4 + 2 * 96000 * 1024 * 4, "out": 96000 * 1024 * 4 + 1024 * 1024 * 4 + 2 * 96000 * 1024 * 4,
    "pw1": 96000 * 1024 * 4 + 2048 * 1024 * 4 + 96000 * 1024 * 4, "pw2": 96000 * 1024 * 4 + 1024 * 1024 * 4 + 2 * 96000 * 1024 * 4,
}
OPERAND_BYTES["ffn2_a"], OPERAND_BYTES["ffn2_b"] = OPERAND_BYTES["ffn1_a"], OPERAND_BYTES["ffn1_b"]


def dispatches(directory, counter):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    f = max(files, key=os.path.getmtime)
    d = {}
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] == counter:
                d[int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]))
    return [d[k] for k in sorted(d)]


def by_role(seq):
    names = [n for n, _ in seq]
    vals = [v for _, v in seq]
    roles = defaultdict(list)
    i, n = 0, len(seq)
    G = "gemm_f16x2_tg_kernel<false, 4, 8>"
    while i < n:
        if "layernorm_split_kernel" in names[i] and i + 14 < n and G in names[i + 1] and G in names[i + 2] and "relpos_attention" in names[i + 5] and "dwconv" in names[i + 9]:
            idx = [i + 1, i + 2, i + 4, i + 6, i + 8, i + 10, i + 12, i + 13]
            if all(G in names[j] for j in idx):
                for k, j in enumerate(idx):
                    roles[ROLE[k]].append(vals[j])
                roles["attention"].append(vals[i + 5])
                roles["dwconv"].append(vals[i + 9])
                i += 14
                continue
        i += 1
    return roles


def main():
    out, fdir, wdir = sys.argv[1:4]
    fetch, write = by_role(dispatches(fdir, "FETCH_SIZE")), by_role(dispatches(wdir, "WRITE_SIZE"))
    res = {"_doc": __doc__.split("\n\n")[0].replace("\n", " "), "roles": {}, "groups": {}}
    for k in ROLE + ["attention", "dwconv"]:
        if fetch[k] and write[k]:
            f, w = sum(fetch[k]) / len(fetch[k]), sum(write[k]) / len(write[k])
            b = (2 * f + w) * 1024
            res["roles"][k] = {"launches": len(fetch[k]), "fetch_bytes": int(2 * f * 1024), "write_bytes": int(w * 1024), "hbm_bytes_per_launch": int(b)}
            if k in OPERAND_BYTES:
                res["roles"][k]["algorithmic_bytes"] = OPERAND_BYTES[k]
                res["roles"][k]["ratio"] = round(b / OPERAND_BYTES[k], 3)
    for g, members in GROUP.items():
        vals = [res["roles"][m]["hbm_bytes_per_launch"] for m in members if m in res["roles"]]
        if vals:
            res["groups"][g] = {"hbm_bytes_per_launch": int(sum(vals) / len(vals)), "roles": members}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res["roles"].items():
        print(f"{k:10s} {v['launches']:4d} launches  {v['hbm_bytes_per_launch'] / 1e9:7.3f} GB per launch" + (f"  ({v['ratio']:.2f} x the operand + output bytes)" if "ratio" in v else ""))


if __name__ == "__main__":
    main()
