"""Aggregate rocprofv3 PMC passes into one per-kernel table.

    python tools/collect_pmc.py <out.csv> <pass_dir> [<pass_dir> ...]

Every <pass_dir> is the -d directory of one `rocprofv3 --pmc ... --kernel-trace --output-format csv` run (counters are collected in
their own runs, never together with other trace domains). For every kernel name (template arguments kept, argument list
dropped) and every counter found, writes launches, the per-launch average and the total. Raw counter units are kept (FETCH_SIZE /
WRITE_SIZE: KiB; SQ_*_CYCLES: see MI355X_MICROARCH.md, cycle-constants table); derived columns are added by the caller.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def read_pass(directory):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        return {}
    f = max(files, key=os.path.getmtime)   # gpurun merges successive runs into one directory: keep the newest
    tot = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            tot[name][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[name].add(row.get("Dispatch_Id"))
    return {k: (len(disp[k]), dict(v)) for k, v in tot.items()}


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    table = defaultdict(dict)
    launches = {}
    counters = []
    for d in dirs:
        for name, (n, vals) in read_pass(d).items():
            launches[name] = n
            for c, v in vals.items():
                table[name][c] = v
                if c not in counters:
                    counters.append(c)
    rows = sorted(table, key=lambda k: -table[k].get("SQ_BUSY_CYCLES", table[k].get("FETCH_SIZE", 0.0)))
    with open(out, "w") as fh:
        fh.write("kernel,launches," + ",".join(f"{c}_per_launch" for c in counters) + "\n")
        for k in rows:
            fh.write(f"\"{k}\",{launches[k]}," + ",".join(f"{table[k].get(c, float('nan')) / max(launches[k], 1):.1f}" for c in counters) + "\n")
    print(open(out).read()[:6000])


if __name__ == "__main__":
    main()
