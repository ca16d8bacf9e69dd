"""The LDS access patterns of the fused SEANet kernels as address functions (fp16 element offsets x 2 = bytes), for tools/lds_bank_sim.py."""
from lds_bank_sim import report


def res128rs(tail_row, label, hoff=None):
    LDX, LDH = 144, 80
    if hoff is None:
        hoff = lambda row, chunk: row * LDH + chunk * 8     # round 5: rows 160 B apart
    tot = [0, 0]

    def add(r):
        tot[0] += r[0]; tot[1] += r[1]
    print(f"-- seanet_res128rs ({label}); per 32-row tile, all 8 waves")
    # conv3 fragment reads: lane (r16, q) reads 16 B at row (16 m + r16 + tap), chunk (ks & 3) * 4 + q; 12 ks x 2 m x 2 pieces per C wave, 4 C waves
    add(report("conv3 x fragment (rows consecutive)", "ds_read_b128", lambda l: 2 * (((l & 15) + 1) * LDX + ((l >> 4) << 3)), 12 * 2 * 2 * 4))
    for m in range(2):
        add(report(f"tail h fragment, m={m}", "ds_read_b128", lambda l: 2 * hoff(tail_row(16 * m + (l & 15)), l >> 4), 2 * 2 * 4))
        add(report(f"tail x fragment, m={m}", "ds_read_b128", lambda l: 2 * ((tail_row(16 * m + (l & 15)) + 2) * LDX + ((l >> 4) << 3)), 4 * 2 * 4))
    # staging: chunk c = tid + 512 j -> row c >> 5, float4 c & 31 -> 8-byte stores into 4 planes
    add(report("stage x pieces", "ds_write_b64", lambda l: 2 * ((l >> 5) * LDX + ((l & 31) >> 1) * 8 + (l & 1) * 4), 4 * 17))   # 34 x 32 chunks / 64 lanes = 17 wave-stores x 4 planes
    # h epilogue: lane (r16, q) writes 8 B at row 16 m + r16, channels 16 w + 4 q
    add(report("h epilogue pieces", "ds_write_b64", lambda l: 2 * (hoff(l & 15, (l >> 4) >> 1) + (((l >> 4) & 1) << 2)), 2 * 2 * 4))
    print(f"   total {tot[0]} LDS-array cycles per tile, conflict-free {tot[1]}: conflict share {(tot[0] - tot[1]) / tot[0]:.2f}")


def phase_order_r5(pos):
    pl = pos // 7 if pos < 14 else 2 + (pos - 14) // 6
    i = pos - 7 * pl if pos < 14 else pos - 14 - 6 * (pl - 2)
    return 5 * i + pl


RS_TAIL_ROWS = [2, 7, 12, 17, 0, 5, 10, 15, 20, 25, 30, 3, 22, 27, 8, 13,
                4, 9, 14, 19, 1, 6, 11, 16, 21, 26, 31, 28, 24, 29, 18, 23]


def main():
    res128rs(lambda p: p, "natural row order: the fp32-output path")
    res128rs(phase_order_r5, "round-5 phase order (rows sorted by row % 5)")
    assert sorted(RS_TAIL_ROWS) == list(range(32))
    res128rs(lambda p: RS_TAIL_ROWS[p], "round-6 phase order (bank-aware), h rows dense + chunk ^ (row & 7)", lambda row, chunk: row * 64 + ((chunk ^ (row & 7)) << 3))
    stage0x3()
    res64down()
    f = lambda row: (((row >> 2) & 1) * 2) ^ (((row >> 3) & 1) * 3)
    res64down(plane_pad=8, label="round 6: R planes + 16 B, h rows dense + swizzled",
              h_addr=lambda row, chunk: row * 32 + ((chunk ^ f(row)) << 3), h_waddr=lambda row, col: row * 32 + (((col >> 3) ^ f(row)) << 3) + (col & 7))


def stage0x3(LDX=48, LDH=24, LDR=48, RIDX=34, xbit=lambda r: r >> 2, rbit=lambda r: r ^ (r >> 3), swz=lambda col, bit: col ^ ((bit & 1) << 3), label="round 5"):
    """seanet_stage0x3_kernel<SchemeF16x2>: per 60-sample tile, 4 waves, 2 pieces."""
    tot = [0, 0]

    def add(r):
        tot[0] += r[0]; tot[1] += r[1]
    print(f"-- seanet_stage0x3 ({label}); per tile, all 4 waves")
    r16, q = (lambda l: l & 15), (lambda l: l >> 4)
    for mt in range(4):   # conv0 stores: unit (mt, nt): row i = 16 mt + r16, 8 B at column nt * 16 + 4 q
        for nt in range(2):
            add(report(f"conv0 store mt={mt} nt={nt}", "ds_write_b64", lambda l: 2 * ((16 * mt + r16(l)) * LDX + swz(nt * 16 + q(l) * 4, xbit(16 * mt + r16(l)))), 2))
    for w in range(4):
        for ks in range(3):
            add(report(f"conv3 x0 fragment wave={w} ks={ks}", "ds_read_b128", lambda l: 2 * ((16 * w + r16(l) + ks) * LDX + swz(q(l) * 8, xbit(16 * w + r16(l) + ks))), 2))
        add(report(f"h store wave={w}", "ds_write_b64", lambda l: 2 * ((16 * w + r16(l)) * LDH + q(l) * 4), 2))
        add(report(f"tail h fragment wave={w}", "ds_read_b128", lambda l: 2 * ((16 * w + r16(l)) * LDH + (q(l) & 1) * 8), 2))
        for nt in range(2):
            add(report(f"r store wave={w} nt={nt}", "ds_write_b64",
                       lambda l: 2 * ((((16 * w + r16(l)) & 1) * RIDX + ((16 * w + r16(l)) >> 1)) * LDR + swz(nt * 16 + q(l) * 4, rbit(16 * w + r16(l)))), 2))
    for ks in range(4):
        for om in range(2):
            add(report(f"down0 r fragment ks={ks} om={om}", "ds_read_b128",
                       lambda l: 2 * (((ks & 1) * RIDX + om * 16 + r16(l) + (ks >> 1)) * LDR + swz(q(l) * 8, rbit(2 * (om * 16 + r16(l) + (ks >> 1)) + (ks & 1)))), 2 * 4))
    print(f"   total {tot[0]} LDS-array cycles per tile, conflict-free {tot[1]}: conflict share {(tot[0] - tot[1]) / tot[0]:.2f}")
    return tot


def res64down(LDX=80, LDH=40, LDR=80, PL=17, plane_pad=0, label="round 5", h_addr=None, h_waddr=None):
    """seanet_res64down_kernel: per 64-row tile, 12 waves, 2 pieces."""
    tot = [0, 0]

    def add(r):
        tot[0] += r[0]; tot[1] += r[1]
    print(f"-- seanet_res64down ({label}); per 64-row tile, all 12 waves")
    r16, q = (lambda l: l & 15), (lambda l: l >> 4)
    PS = PL * LDR + plane_pad
    rd_off = lambda row: (row & 3) * PS + (row >> 2) * LDR
    if h_addr is None:
        h_addr = lambda row, chunk: row * LDH + chunk * 8            # fragment read: 8 fp16 at chunk
        h_waddr = lambda row, col: row * LDH + col                   # 4 fp16 at channel col
    # block role (4 waves): conv3 fragments, h stores, tail fragments, R stores
    add(report("conv3 x fragment", "ds_read_b128", lambda l: 2 * ((r16(l) + 1) * LDX + q(l) * 8), 6 * 2 * 2 * 4))
    for cn in range(2):
        add(report(f"h store cn={cn}", "ds_write_b64", lambda l: 2 * h_waddr(r16(l), cn * 16 + q(l) * 4), 2 * 2 * 2))
    add(report("tail h fragment", "ds_read_b128", lambda l: 2 * h_addr(r16(l), q(l)), 4 * 2 * 4))
    add(report("tail x fragment", "ds_read_b128", lambda l: 2 * ((r16(l) + 2) * LDX + q(l) * 8), 2 * 4 * 2 * 4))
    for rw in range(4):
        add(report(f"R store rw={rw}", "ds_write_b64", lambda l: 2 * (rd_off(16 + r16(l) + 4) + rw * 16 + q(l) * 4), 4 * 2))
    # conv role (8 waves): R fragments (16 K steps x 2 pieces), input staging (2 halves x 4 stores)
    add(report("down1 R fragment", "ds_read_b128", lambda l: 2 * (r16(l) * LDR + q(l) * 8 + rd_off(1)), 16 * 2 * 8))
    add(report("stage x pieces", "ds_write_b64", lambda l: 2 * ((l >> 4) * LDX + (l & 15) * 4), 2 * 4 * 8))
    print(f"   total {tot[0]} LDS-array cycles per tile, conflict-free {tot[1]}: conflict share {(tot[0] - tot[1]) / tot[0]:.2f}")
    return tot
