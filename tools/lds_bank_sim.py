#!/usr/bin/env python3
"""LDS bank-conflict arithmetic for gfx950 (MI355X_MICROARCH.md, section LDS): given the byte address every lane of a wave64 presents to one DS instruction,
the LDS-array cycles it takes = sum over the instruction's lane groups of the largest number of DISTINCT addresses on one bank (identical addresses broadcast).
Used to find and fix the conflicting access patterns of the fused SEANet kernels (round 6; `python tools/lds_bank_sim.py` prints the table in
profiles/r06_lds_conflicts.txt)."""
from collections import defaultdict

G32 = [list(range(0, 32)), list(range(32, 64))]
G16 = [list(range(16 * i, 16 * i + 16)) for i in range(4)]
G8 = [list(range(8 * i, 8 * i + 8)) for i in range(8)]
B128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128 = B128 + [[l + 32 for l in g] for g in B128]
# instruction -> (lane groups, bank modulus in dwords, bytes per lane)
INSTR = {"ds_read_b32": (G32, 32, 4), "ds_read_b64": (G32, 64, 8), "ds_read_b128": (B128, 64, 16), "ds_read_b64_tr_b16": (G32, 64, 8),
         "ds_write_b32": (G32, 32, 4), "ds_write_b64": (G16, 32, 8), "ds_write_b128": (G8, 32, 16), "ds_write_b16": (G32, 32, 2)}


def cycles(instr, addr, active=None):
    """(LDS-array cycles, conflict-free cycles) of one wave-instruction; addr[lane] = byte address (None / inactive lanes take no part)."""
    groups, mod, nbytes = INSTR[instr]
    total = 0
    for g in groups:
        per_bank = defaultdict(set)
        for l in g:
            if addr[l] is None or (active is not None and not active[l]):
                continue
            for d in range(max(1, nbytes // 4)):
                dw = addr[l] // 4 + d
                per_bank[dw % mod].add(dw)
        total += max([len(v) for v in per_bank.values()], default=0)
    return total, len(groups)


def report(name, instr, addr_of_lane, count=1):
    c, ideal = cycles(instr, [addr_of_lane(l) for l in range(64)])
    print(f"{name:70s} {instr:14s} {c:3d} cycles (conflict-free {ideal}) x {count}")
    return c * count, ideal * count


if __name__ == "__main__":
    import sys
    sys.path.insert(0, __file__.rsplit("/", 1)[0])
    import lds_patterns
    lds_patterns.main()
