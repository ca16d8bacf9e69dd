"""The LDS access patterns of the fused SEANet kernels as address functions (fp16 element offsets x 2 = bytes), for tools/lds_bank_sim.py."""
from lds_bank_sim import report


def res128rs(tail_row, label):
    LDX, LDH = 144, 80
    tot = [0, 0]

    def add(r):
        tot[0] += r[0]; tot[1] += r[1]
    print(f"-- seanet_res128rs ({label}); per 32-row tile, all 8 waves")
    # conv3 fragment reads: lane (r16, q) reads 16 B at row (16 m + r16 + tap), chunk (ks & 3) * 4 + q; 12 ks x 2 m x 2 pieces per C wave, 4 C waves
    add(report("conv3 x fragment (rows consecutive)", "ds_read_b128", lambda l: 2 * (((l & 15) + 1) * LDX + ((l >> 4) << 3)), 12 * 2 * 2 * 4))
    for m in range(2):
        add(report(f"tail h fragment, m={m}", "ds_read_b128", lambda l: 2 * (tail_row(16 * m + (l & 15)) * LDH + ((l >> 4) << 3)), 2 * 2 * 4))
        add(report(f"tail x fragment, m={m}", "ds_read_b128", lambda l: 2 * ((tail_row(16 * m + (l & 15)) + 2) * LDX + ((l >> 4) << 3)), 4 * 2 * 4))
    # staging: chunk c = tid + 512 j -> row c >> 5, float4 c & 31 -> 8-byte stores into 4 planes
    add(report("stage x pieces", "ds_write_b64", lambda l: 2 * ((l >> 5) * LDX + ((l & 31) >> 1) * 8 + (l & 1) * 4), 4 * 17))   # 34 x 32 chunks / 64 lanes = 17 wave-stores x 4 planes
    # h epilogue: lane (r16, q) writes 8 B at row 16 m + r16, channels 16 w + 4 q
    add(report("h epilogue pieces", "ds_write_b64", lambda l: 2 * ((l & 15) * LDH + (((l >> 4) >> 1) << 3) + (((l >> 4) & 1) << 2)), 2 * 2 * 4))
    print(f"   total {tot[0]} LDS-array cycles per tile, conflict-free {tot[1]}: conflict share {(tot[0] - tot[1]) / tot[0]:.2f}")


def phase_order_r5(pos):
    pl = pos // 7 if pos < 14 else 2 + (pos - 14) // 6
    i = pos - 7 * pl if pos < 14 else pos - 14 - 6 * (pl - 2)
    return 5 * i + pl


RS_TAIL_ROWS = [2, 7, 12, 17, 0, 5, 10, 15, 20, 25, 30, 27, 22, 3, 8, 13,
                4, 9, 14, 19, 1, 6, 11, 16, 21, 26, 31, 28, 24, 29, 18, 23]


def main():
    res128rs(lambda p: p, "natural row order: the fp32-output path")
    res128rs(phase_order_r5, "round-5 phase order (rows sorted by row % 5)")
    assert sorted(RS_TAIL_ROWS) == list(range(32))
    res128rs(lambda p: RS_TAIL_ROWS[p], "round-6 phase order (bank-aware)")
