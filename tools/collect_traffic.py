"""Turn two rocprofv3 PMC passes (one with `--pmc FETCH_SIZE`, one with `--pmc WRITE_SIZE`, each over
`python3 bench.py --workload acoustic --steps 1 --warmup 1 --no-cpu-baseline`) into per-kernel summaries and the
`profiles/<round>_traffic.json` that bench.py reads for `roofline.traffic`.

    python tools/collect_traffic.py <fetch_dir> <write_dir> <out_prefix>     e.g.  ... gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01

Units and corrections follow MI355X_MICROARCH.md (section HBM): the counters are KiB; on gfx950 FETCH_SIZE reports half of
the bytes of a wide coalesced read stream, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024; WRITE_SIZE is exact for
16-B-per-lane stores (narrower stores are uncalibrated and over-count).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

GROUPS = {  # bench.py group -> kernel name prefix
    "lstm_rec": "at::lstm_seq",   # lstm_seq_kernel / lstm_seq_x3_kernel<..>
    "stage0_fused": "at::seanet_stage0",      # seanet_stage0_kernel / seanet_stage0x3_kernel, whichever ran
    "res1_down1": "at::seanet_res64down",
    "res1": "at::seanet_res64x3",
    "down1": "at::seanet_down64",
    "res2": "at::seanet_res128",
    "rvq": "at::rvq_encode",
}


def per_kernel(directory, counter):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {directory}")
    files = [max(files, key=os.path.getmtime)]   # gpurun merges successive runs into one directory: keep the newest
    tot, cnt = defaultdict(float), defaultdict(int)
    seen = set()
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = row["Kernel_Name"].split("(")[0]
                tot[name] += float(row["Counter_Value"])
                key = (f, row.get("Dispatch_Id"))
                if key not in seen:
                    seen.add(key)
                    cnt[name] += 1
    return tot, cnt


def main():
    fetch_dir, write_dir, prefix = sys.argv[1:4]
    out = {}
    for counter, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
        tot, cnt = per_kernel(d, counter)
        rows = sorted(tot, key=lambda k: -tot[k])
        with open(f"{prefix}_acoustic_pmc_{counter}.csv", "w") as fh:
            fh.write("kernel,launches,avg_KiB,total_KiB\n")
            for k in rows:
                fh.write(f"\"{k}\",{cnt[k]},{tot[k] / max(cnt[k], 1):.1f},{tot[k]:.1f}\n")
        out[counter] = {k: tot[k] / max(cnt[k], 1) for k in rows}
    kernels = {}
    for group, pref in GROUPS.items():
        f = next((v for k, v in out["FETCH_SIZE"].items() if k.replace("void ", "").startswith(pref)), None)
        w = next((v for k, v in out["WRITE_SIZE"].items() if k.replace("void ", "").startswith(pref)), None)
        if f is None or w is None:
            continue
        kernels[group] = {"kernel": pref, "FETCH_SIZE_KiB_per_launch": round(f, 1), "WRITE_SIZE_KiB_per_launch": round(w, 1),
                          "traffic_bytes_per_launch": int((2 * f + w) * 1024)}
    doc = ("HBM traffic per launch from rocprofv3 PMC (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over `bench.py --workload "
           "acoustic --steps 1 --warmup 1`, B=256 x 10 s). Units: counters are KiB. gfx950 correction (MI355X_MICROARCH.md section "
           "HBM): FETCH_SIZE reports half of a wide coalesced read stream, so bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; WRITE_SIZE is "
           "exact only for 16-B-per-lane stores (the LSTM's 4-B write-through h stores over-count).")
    with open(f"{prefix}_traffic.json", "w") as fh:
        json.dump({"_doc": doc, "kernels": kernels}, fh, indent=1)
    print(json.dumps(kernels, indent=1))


if __name__ == "__main__":
    main()
