"""Turn the per-kernel PMC summaries (tools/collect_pmc.py, one per workload) into (1) a derived table per workload — HBM bytes per launch with
the gfx950 FETCH_SIZE correction, clock, matrix-pipe and vector-unit busy fractions, LDS conflict share — and (2) profiles/<round>_traffic.json,
the `roofline.traffic` source of bench.py.

    python tools/pmc_report.py <round prefix, e.g. profiles/r02> acoustic=<pmc_summary.csv> semantic_m=<pmc_summary.csv> ...

Units (MI355X_MICROARCH.md): FETCH_SIZE / WRITE_SIZE are KiB and FETCH_SIZE reports half of a wide coalesced read stream -> bytes = (2 * FETCH +
WRITE) * 1024; GRBM_GUI_ACTIVE is summed over the 8 XCDs -> cycles = GUI / 8; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs ->
matrix-pipe busy fraction = MFMA_BUSY / (1024 * cycles); SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles per wave (ratios are
unit-free); SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = share of LDS-array cycles that are conflict replays.
"""
import csv
import json
import sys

GROUPS = {   # workload -> bench.py group -> kernel-name prefix (first match, most launches)
    "acoustic": {"stage0_fused": "at::seanet_stage0", "res1_down1": "at::seanet_res64down", "res1": "at::seanet_res64x3", "down1": "at::seanet_down64", "res2": "at::seanet_res128",
                 "lstm_rec": "at::lstm_seq", "rvq": "at::rvq_encode", "down3": "at::gemm_f16x2_tg_kernel<true", "lstm_ih": "at::gemm_f16x2_tg_kernel<false"},
    "semantic_m": {"ffn": "at::gemm_f16x2_tg_kernel<false", "attn_proj": "at::gemm_f16x2_tg_kernel<false", "conv_module": "at::gemm_f16x2_tg_kernel<false",
                   "attention": "at::relpos_attention", "layernorm": "at::layernorm_split_kernel"},
    "semantic_s": {"ffn": "at::gemm_f16x2_tg_kernel<false", "attn_proj": "at::gemm_f16x2_tg_kernel<false", "attention": "at::relpos_attention",
                   "feature_convs": "at::gemm_f16x2_tg_kernel<true", "positional_conv": "at::gemm_f32_kernel", "conv0": "at::hub_conv0"},
    "acoustic_decode": {"dec_tail": "at::seanet_dectail", "lstm_rec": "at::lstm_pipe", "dec_res1": "at::seanet_res128", "dec_res2": "at::seanet_res64",
                        "dec_up0": "at::gemm_f16x2_tg_kernel<true", "dec_res0": "at::gemm_f32_kernel"},
}


# bench.py kernel GROUPS that span several kernels: group -> (kernel-name prefixes, launches per encode, the kernel that runs once per encode). The
# figure is the group's HBM bytes per encode / launches per encode = bytes per launch AVERAGED over the group, the unit roofline_of() uses for `avg_launch_ms`.
GROUP_SUMS = {
    "semantic_s": {"feature_extractor": (("at::hub_wavstats", "at::hub_gn_coeff", "at::hub_conv0", "at::gemm_f16x2_tg_kernel<true"), 9, "at::hub_gn_coeff")},
}


def f(row, key):
    try:
        return float(row[key + "_per_launch"])
    except (KeyError, ValueError):
        return float("nan")


def main():
    prefix = sys.argv[1]
    traffic = {"_doc": "HBM bytes per launch from rocprofv3 PMC passes over `bench.py --workload <w> --steps 1 --warmup 1` (separate --pmc runs, kernel-trace "
                       "only): bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reports half of a wide coalesced read stream; WRITE_SIZE exact "
                       "for 16-B stores). A GEMM kernel's figure is the average over every launch of that kernel in a step (all layer shapes)."}
    for spec in sys.argv[2:]:
        workload, path = spec.split("=", 1)
        rows = list(csv.DictReader(open(path)))
        out_rows = []
        for r in rows:
            cyc = f(r, "GRBM_GUI_ACTIVE") / 8.0
            wc = f(r, "SQ_WAVE_CYCLES")
            fetch, write = f(r, "FETCH_SIZE"), f(r, "WRITE_SIZE")
            out_rows.append({
                "kernel": r["kernel"], "launches": int(r["launches"]),
                "hbm_bytes_per_launch": int((2 * fetch + write) * 1024) if fetch == fetch and write == write else None,
                "cycles_per_launch": round(cyc) if cyc == cyc else None,
                "mfma_pipe_busy_frac": round(f(r, "SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * cyc), 4) if cyc and cyc == cyc else None,
                "valu_active_per_wave_cycle": round(f(r, "SQ_ACTIVE_INST_VALU") / wc, 4) if wc == wc and wc else None,
                "wave_wait_any_frac": round(f(r, "SQ_WAIT_ANY") / wc, 4) if wc == wc and wc else None,
                "wave_wait_inst_frac": round(f(r, "SQ_WAIT_INST_ANY") / wc, 4) if wc == wc and wc else None,
                "lds_conflict_share": round(f(r, "SQ_LDS_BANK_CONFLICT") / max(f(r, "SQ_LDS_IDX_ACTIVE"), 1.0), 4),
                "lds_busy_frac": round(f(r, "SQ_LDS_IDX_ACTIVE") / 256.0 / cyc, 4) if cyc and cyc == cyc else None,
                "mfma_insts_per_launch": int(f(r, "SQ_INSTS_MFMA")) if f(r, "SQ_INSTS_MFMA") == f(r, "SQ_INSTS_MFMA") else None,
            })
        with open(f"{prefix}_{workload}_pmc_derived.csv", "w") as fh:
            cols = list(out_rows[0].keys())
            fh.write(",".join(cols) + "\n")
            for o in out_rows:
                fh.write(",".join(f'"{o[c]}"' if c == "kernel" else str(o[c]) for c in cols) + "\n")
        kernels = {}
        for group, pref in GROUPS.get(workload, {}).items():
            bare = pref.split("::")[-1].split("<")[0]   # rocprofv3 leaves some template instantiations mangled (_ZN2at26relpos_attention_w8_kernelILb1EEE...)
            cands = [o for o in out_rows if (o["kernel"].startswith(pref) or (o["kernel"].startswith("_Z") and bare in o["kernel"])) and o["hbm_bytes_per_launch"] is not None]
            if not cands:
                continue
            o = max(cands, key=lambda c: c["launches"])
            kernels[group] = {"kernel": o["kernel"], "traffic_bytes_per_launch": o["hbm_bytes_per_launch"], "mfma_pipe_busy_frac": o["mfma_pipe_busy_frac"],
                              "cycles_per_launch": o["cycles_per_launch"], "launches_in_pmc_run": o["launches"]}
        for group, (prefs, per_encode, once) in GROUP_SUMS.get(workload, {}).items():
            def match(o, pref):
                bare = pref.split("::")[-1].split("<")[0]
                return o["kernel"].replace("void ", "").startswith(pref) or (o["kernel"].startswith("_Z") and bare in o["kernel"])
            calls = sum(o["launches"] for o in out_rows if match(o, once))
            members = [o for o in out_rows if any(match(o, p) for p in prefs) and o["hbm_bytes_per_launch"] is not None]
            if calls and members:
                total = sum(o["launches"] * o["hbm_bytes_per_launch"] for o in members) / calls
                kernels[group] = {"kernel": " + ".join(sorted({o["kernel"].split("(")[0][:60] for o in members})), "traffic_bytes_per_encode": int(total),
                                  "traffic_bytes_per_launch": int(total / per_encode), "launches_per_encode": per_encode, "encodes_in_pmc_run": calls}
        if workload == "semantic_m":   # the one GEMM kernel symbol resolved by ROLE (tools/gemm_roles_pmc.py, per-dispatch counters of the same passes)
            import os
            rp = f"{prefix}_gemm_roles_traffic.json"
            if os.path.exists(rp):
                roles = json.load(open(rp))
                for g, v in roles.get("groups", {}).items():
                    if g in kernels:
                        kernels[g]["traffic_bytes_per_launch_all_roles_average"] = kernels[g]["traffic_bytes_per_launch"]
                        kernels[g]["traffic_bytes_per_launch"] = v["hbm_bytes_per_launch"]
                        kernels[g]["role_resolved"] = v["roles"]
                kernels["_gemm_roles"] = roles.get("roles", {})
        traffic[workload] = {"kernels": kernels}
        print(workload, json.dumps(kernels, indent=1)[:1500])
    with open(f"{prefix}_traffic.json", "w") as fh:
        json.dump(traffic, fh, indent=1)


if __name__ == "__main__":
    main()
