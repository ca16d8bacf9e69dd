"""Writes the FLAC fixtures of tests/test_flac_cpu.py: tests/golden/flac_{a,b,c}.flac and the PCM they encode (flac_pcm.npz).

There is no FLAC tool in this image (no flac / ffmpeg / soundfile), so the fixtures come from the small ENCODER below, written from the format's
specification (RFC 9639) independently of the decoder under test (audiotoken_amd/csrc/flac_decode.hip, C++): bit packing, Rice coding, the four fixed
predictors, LPC analysis (autocorrelation + Levinson-Durbin, quantised coefficients), wasted bits, the four stereo decorrelations, CRC-8 / CRC-16
and the STREAMINFO MD5 are implemented a second time here, in the opposite direction. What the test then pins: decode(encode(pcm)) == pcm for every
subframe type / channel assignment / sample size the encoder can be made to emit, the frame CRCs, and the MD5 of the decoded samples against the
one this script stored (hashlib). The fixtures are data (seeded synthetic PCM), not reference text.

    python tests/golden/make_flac.py        # rewrites the fixtures; deterministic
"""
import hashlib
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


class BitWriter:
    def __init__(self):
        self.acc = 0
        self.n = 0
        self.out = bytearray()

    def put(self, value, bits):
        if bits == 0:
            return
        self.acc = (self.acc << bits) | (int(value) & ((1 << bits) - 1))
        self.n += bits
        while self.n >= 8:
            self.n -= 8
            self.out.append((self.acc >> self.n) & 0xff)
        self.acc &= (1 << self.n) - 1

    def unary(self, zeros):
        while zeros >= 32:
            self.put(0, 32)
            zeros -= 32
        self.put(1, zeros + 1)

    def align(self):
        if self.n:
            self.put(0, 8 - self.n)

    def bytes(self):
        assert self.n == 0
        return bytes(self.out)


def crc8(data):
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xff if c & 0x80 else (c << 1) & 0xff
    return c


def crc16(data):
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xffff if c & 0x8000 else (c << 1) & 0xffff
    return c


def utf8_number(v):
    if v < 0x80:
        return bytes([v])
    out = []
    n = 1
    while v >= (1 << (6 * n + (6 - n))):
        n += 1
    for i in range(n):
        out.append(0x80 | ((v >> (6 * i)) & 0x3f))
    lead = ((0xff << (7 - n)) & 0xff) | (v >> (6 * n))
    return bytes([lead] + out[::-1])


def zigzag(r):
    return (r << 1) ^ (r >> 63) if r >= 0 else ((-r) << 1) - 1


def rice_bits(res, k):
    return sum((zigzag(int(r)) >> k) + 1 + k for r in res)


def write_residual(bw, res, blocksize, order, method, force_escape):
    pbits, esc = (4, 15) if method == 0 else (5, 31)
    kmax = esc - 1
    best = None
    for porder in range(0, 5):
        parts = 1 << porder
        if blocksize % parts or (blocksize >> porder) <= order:
            continue
        total, ks = 0, []
        pos = 0
        for p in range(parts):
            cnt = (blocksize >> porder) - (order if p == 0 else 0)
            seg = res[pos:pos + cnt]
            pos += cnt
            kbest = min(range(0, kmax + 1), key=lambda k: rice_bits(seg, k))
            ks.append(kbest)
            total += pbits + rice_bits(seg, kbest)
        if best is None or total < best[0]:
            best = (total, porder, ks)
    _, porder, ks = best
    bw.put(method, 2)
    bw.put(porder, 4)
    pos = 0
    for p in range(1 << porder):
        cnt = (blocksize >> porder) - (order if p == 0 else 0)
        seg = [int(r) for r in res[pos:pos + cnt]]
        pos += cnt
        if force_escape and p == 0:
            nb = max(1, max((abs(r) for r in seg), default=0).bit_length() + 1)
            bw.put(esc, pbits)
            bw.put(nb, 5)
            for r in seg:
                bw.put(r, nb)
            continue
        k = ks[p]
        bw.put(k, pbits)
        for r in seg:
            u = zigzag(r)
            bw.unary(u >> k)
            bw.put(u & ((1 << k) - 1), k)


FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def predict_residual(s, coefs, shift):
    order = len(coefs)
    s = [int(v) for v in s]
    res = []
    for i in range(order, len(s)):
        acc = sum(coefs[j] * s[i - 1 - j] for j in range(order))
        res.append(s[i] - (acc >> shift))
    return res


def lpc_coefficients(s, order, precision):
    x = np.asarray(s, dtype=np.float64)
    x = x * np.hanning(len(x))
    r = np.array([np.dot(x[:len(x) - l], x[l:]) for l in range(order + 1)])
    if r[0] <= 0:
        return None
    a = np.zeros(order + 1)
    a[0] = 1.0
    err = r[0]
    for i in range(1, order + 1):
        k = -(r[i] + np.dot(a[1:i], r[i - 1:0:-1])) / err
        a_new = a.copy()
        a_new[i] = k
        a_new[1:i] = a[1:i] + k * a[i - 1:0:-1]
        a = a_new
        err *= 1.0 - k * k
        if err <= 0:
            return None
    coefs = -a[1:]
    cmax = np.abs(coefs).max()
    if cmax == 0:
        return None
    shift = precision - 1 - int(np.floor(np.log2(cmax))) - 1
    shift = max(0, min(shift, 15))
    q = np.clip(np.round(coefs * (1 << shift)), -(1 << (precision - 1)), (1 << (precision - 1)) - 1).astype(np.int64)
    return [int(v) for v in q], shift


def write_subframe(bw, s, bps, kind, method=0, force_escape=False, lpc_order=8, lpc_precision=12):
    s = [int(v) for v in s]
    n = len(s)
    wasted = 0
    if any(s):
        while all((v >> wasted) & 1 == 0 for v in s) and wasted < bps - 1:
            wasted += 1
    if wasted:
        s = [v >> wasted for v in s]
    eff = bps - wasted
    if kind == "auto":
        if all(v == s[0] for v in s):
            kind = "constant"
        else:
            kind = "fixed"
    bw.put(0, 1)
    if kind == "constant":
        bw.put(0b000000, 6)
    elif kind == "verbatim":
        bw.put(0b000001, 6)
    elif kind == "fixed":
        order = min(range(0, 5), key=lambda o: sum(abs(r) for r in predict_residual(s, FIXED[o], 0)) if n > o else 1 << 62)
        bw.put(0b001000 | order, 6)
    elif kind == "lpc":
        lp = lpc_coefficients(s, lpc_order, lpc_precision)
        if lp is None:
            return write_subframe_plain(bw, s, eff, wasted, "verbatim")
        bw.put(0b100000 | (lpc_order - 1), 6)
    if wasted:
        bw.put(1, 1)
        bw.unary(wasted - 1)
    else:
        bw.put(0, 1)
    if kind == "constant":
        bw.put(s[0], eff)
    elif kind == "verbatim":
        for v in s:
            bw.put(v, eff)
    elif kind == "fixed":
        for v in s[:order]:
            bw.put(v, eff)
        write_residual(bw, predict_residual(s, FIXED[order], 0), n, order, method, force_escape)
    elif kind == "lpc":
        coefs, shift = lp
        for v in s[:lpc_order]:
            bw.put(v, eff)
        bw.put(lpc_precision - 1, 4)
        bw.put(shift, 5)
        for c in coefs:
            bw.put(c, lpc_precision)
        write_residual(bw, predict_residual(s, coefs, shift), n, lpc_order, method, force_escape)


def write_subframe_plain(bw, s, eff, wasted, kind):
    bw.put(0b000001, 6)
    if wasted:
        bw.put(1, 1)
        bw.unary(wasted - 1)
    else:
        bw.put(0, 1)
    for v in s:
        bw.put(v, eff)


BLOCK_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
RATE_CODES = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10, 96000: 11}
SIZE_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}


def encode(pcm, sample_rate, bps, blocksize, plan):
    """pcm int [C][N]; plan(frame_index) -> dict(kind, stereo, method, force_escape) per frame."""
    pcm = np.asarray(pcm, dtype=np.int64)
    C, N = pcm.shape
    frames = bytearray()
    min_f, max_f = 1 << 24, 0
    fi = 0
    for start in range(0, N, blocksize):
        blk = pcm[:, start:start + blocksize]
        n = blk.shape[1]
        p = plan(fi)
        stereo = p.get("stereo", "lr") if C == 2 else "lr"
        hdr = BitWriter()
        hdr.put(0b11111111111110, 14)
        hdr.put(0, 1)
        hdr.put(0, 1)                                   # fixed block size stream: the coded number is the frame index
        if n in BLOCK_CODES:
            hdr.put(BLOCK_CODES[n], 4)
        else:
            hdr.put(6 if n <= 256 else 7, 4)
        hdr.put(RATE_CODES.get(sample_rate, 0), 4)
        hdr.put({"lr": C - 1, "ls": 8, "sr": 9, "ms": 10}[stereo], 4)
        hdr.put(SIZE_CODES[bps] if p.get("explicit_size", True) else 0, 3)
        hdr.put(0, 1)
        for b in utf8_number(fi):
            hdr.put(b, 8)
        if n not in BLOCK_CODES:
            hdr.put(n - 1, 8 if n <= 256 else 16)
        hb = hdr.bytes()
        body = BitWriter()
        for b in hb:
            body.put(b, 8)
        body.put(crc8(hb), 8)
        if stereo == "lr":
            chans = [(blk[c], bps) for c in range(C)]
        else:
            left, right = blk[0], blk[1]
            side = left - right
            mid = (left + right) >> 1
            chans = {"ls": [(left, bps), (side, bps + 1)], "sr": [(side, bps + 1), (right, bps)], "ms": [(mid, bps), (side, bps + 1)]}[stereo]
        for s, b in chans:
            write_subframe(body, s, b, p.get("kind", "auto"), p.get("method", 0), p.get("force_escape", False))
        body.align()
        fb = body.bytes()
        fb += struct.pack(">H", crc16(fb))
        frames += fb
        min_f, max_f = min(min_f, len(fb)), max(max_f, len(fb))
        fi += 1
    nbytes = (bps + 7) // 8
    inter = pcm.T.reshape(-1)
    raw = b"".join(int(v).to_bytes(nbytes, "little", signed=True) for v in inter)
    md5 = hashlib.md5(raw).digest()
    si = BitWriter()
    si.put(blocksize, 16); si.put(blocksize, 16)
    si.put(min_f, 24); si.put(max_f, 24)
    si.put(sample_rate, 20); si.put(C - 1, 3); si.put(bps - 1, 5); si.put(N, 36)
    for b in md5:
        si.put(b, 8)
    sib = si.bytes()
    assert len(sib) == 34
    return b"fLaC" + bytes([0x80]) + len(sib).to_bytes(3, "big") + sib + bytes(frames), md5


def main():
    from audiotoken_amd import weights as W
    out = {}
    # a: mono 16-bit 16 kHz, 1.5 s, block 4096 + a short last block; frames cycle through fixed / lpc / verbatim
    a = np.round(W.synth_waveform(1, 24000, 16000, seed=901) * 20000).astype(np.int64)
    kinds = ["fixed", "lpc", "lpc", "verbatim", "fixed", "lpc"]
    fa, _ = encode(a, 16000, 16, 4096, lambda i: {"kind": kinds[i % len(kinds)], "method": i % 2})
    # b: stereo 16-bit 44.1 kHz, 0.4 s, block 1152: every channel assignment
    l = W.synth_waveform(1, 17640, 44100, seed=902)[0]
    r = 0.8 * l + 0.2 * W.synth_waveform(1, 17640, 44100, seed=903)[0]
    b = np.round(np.stack([l, r]) * 15000).astype(np.int64)
    st = ["lr", "ls", "sr", "ms"]
    fb, _ = encode(b, 44100, 16, 1152, lambda i: {"kind": "lpc" if i % 3 else "fixed", "stereo": st[i % 4]})
    # c: mono 24-bit 48 kHz, 0.25 s, block 2304: wasted bits (samples << 5), a silent block (constant subframe), escaped partitions, 5-bit Rice parameters
    c = np.round(W.synth_waveform(1, 12000, 48000, seed=904) * (1 << 17)).astype(np.int64) << 5
    c[:, 2304:4608] = 0
    c[:, 4608:6912] = 4096
    fc, _ = encode(c, 48000, 24, 2304, lambda i: {"kind": "auto" if i in (1, 2) else ("lpc" if i % 2 else "fixed"), "method": 1, "force_escape": i == 3,
                                                   "explicit_size": i != 4})
    for name, blob in (("flac_a.flac", fa), ("flac_b.flac", fb), ("flac_c.flac", fc)):
        with open(os.path.join(HERE, name), "wb") as f:
            f.write(blob)
        print(name, len(blob), "bytes")
    out = {"a": a.astype(np.int32), "b": b.astype(np.int32), "c": c.astype(np.int32)}
    np.savez_compressed(os.path.join(HERE, "flac_pcm.npz"), **out)


if __name__ == "__main__":
    main()
