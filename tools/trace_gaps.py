"""How much of an encode is the GPU idle BETWEEN kernels? Reads a rocprofv3 kernel trace (CSV) of `bench.py --workload <w>`, takes the launches of the timed
encodes (between the first and the last launch of `marker`, a kernel that runs once per encode) and prints busy time, idle time between consecutive kernels and
the gap histogram. A dependent launch on one HIP stream costs a few microseconds of front-end time; with ~600 launches per semantic_m encode that is the upper bound of
what a captured graph / fewer launches could recover.      python tools/trace_gaps.py <kernel_trace.csv> [marker substring]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "vq_argmax_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if len(idx) < 2:
    print("marker kernel", marker, "seen", len(idx), "times: need two encodes"); sys.exit(0)
lo, hi = idx[0] + 1, idx[-1] + 1          # launches after the first encode's last kernel up to the last encode's last kernel = len(idx) - 1 whole encodes
seg = rows[lo:hi]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
gaps = [max(0, int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) for a, b in zip(seg[:-1], seg[1:])]
n_enc = len(idx) - 1
wall = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
print(f"{n_enc} encode(s) between markers '{marker}': {len(seg) / n_enc:.0f} launches per encode, wall {wall / n_enc / 1e6:.3f} ms, kernels busy {busy / n_enc / 1e6:.3f} ms, "
      f"idle between kernels {sum(gaps) / n_enc / 1e6:.3f} ms = {100.0 * sum(gaps) / wall:.2f} % (overlapping kernels count as no gap)")
edges = [1e3, 2e3, 4e3, 8e3, 16e3, 64e3, 1e6, 1e12]
hist = [0] * len(edges)
for g in gaps:
    for k, e in enumerate(edges):
        if g < e:
            hist[k] += 1
            break
small = [g for g in gaps if g < 100e3]     # gaps of 100 us and more are the harness between encodes (host synchronisation, taps, the next loop), not launch gaps
print(f"gaps below 100 us (launch gaps inside an encode): {len(small) / n_enc:.0f} per encode, {sum(small) / n_enc / 1e6:.3f} ms per encode = {100.0 * sum(small) / max(busy, 1):.2f} % of the kernels' busy time")
print("gap histogram (ns):", ", ".join(f"<{int(e)}: {h}" for e, h in zip(edges, hist)))
big = sorted(((g, seg[i]["Kernel_Name"][:50], seg[i + 1]["Kernel_Name"][:50]) for i, g in enumerate(gaps)), reverse=True)[:6]
for g, a, b in big:
    print(f"  {g / 1e3:8.1f} us between {a} -> {b}")
