// Host-side FLAC decoder (no device code): the input side of encode_batch_files. The reference decodes every container through
// torchaudio / ffmpeg (reference audiotoken/utils.py:71-101, StreamReader); neither exists here, and '.flac' is the lossless format of the
// reference's AUDIO_EXTS, so the format is restated from its published specification (RFC 9639): STREAMINFO, frame header (fixed / variable block
// size, every block-size / sample-rate / sample-size code, independent / left-side / side-right / mid-side channel assignments), CONSTANT / VERBATIM /
// FIXED (orders 0-4) / LPC (orders 1-32) subframes with wasted bits, Rice-coded residuals (4- and 5-bit parameters, escaped partitions), CRC-8 of every
// frame header and CRC-16 of every frame. The MD5 of the decoded samples (STREAMINFO) is returned for the caller to check (hashlib on the Python side).
// Plain C++ behind the C ABI (include/audiotoken_hip.h: at_flac_info / at_flac_decode): ctypes releases the GIL around it, so a thread pool decodes files in
// parallel. Sequential bit reading is what the format demands (every Rice code's position depends on the one before it): this is CPU work by nature.
#include "at_common.h"
#include "../../include/audiotoken_hip.h"

#include <cstring>
#include <vector>

namespace at {
namespace {

struct BitReader {
    const uint8_t* p;
    size_t n, pos = 0;          // byte position of the next byte to load
    uint64_t buf = 0;           // the `cnt` not yet consumed bits, right-aligned
    int cnt = 0;
    bool bad = false;
    BitReader(const uint8_t* d, size_t len) : p(d), n(len) {}
    inline void refill() {
        while (cnt <= 56 && pos < n) { buf = (buf << 8) | p[pos++]; cnt += 8; }
    }
    inline uint32_t bits(int k) {   // k <= 32
        if (k == 0) return 0;
        if (cnt < k) { refill(); if (cnt < k) { bad = true; return 0; } }
        cnt -= k;
        return (uint32_t)((buf >> cnt) & ((k == 32) ? 0xffffffffull : ((1ull << k) - 1)));
    }
    inline int32_t sbits(int k) {   // two's complement, k <= 32
        if (k == 0) return 0;
        const uint32_t u = bits(k);
        return k == 32 ? (int32_t)u : (int32_t)(u << (32 - k)) >> (32 - k);
    }
    inline int64_t sbits64(int k) {   // k <= 33 (the side channel of a 32-bit stream)
        if (k <= 32) return sbits(k);
        const uint64_t hi = bits(k - 32), lo = bits(32);
        const uint64_t u = (hi << 32) | lo;
        return (int64_t)(u << (64 - k)) >> (64 - k);
    }
    inline uint32_t unary() {       // zeros before the next 1 bit
        uint32_t z = 0;
        for (;;) {
            if (cnt == 0) { refill(); if (cnt == 0) { bad = true; return 0; } }
            const uint64_t window = buf & ((cnt == 64) ? ~0ull : ((1ull << cnt) - 1));
            if (window == 0) { z += (uint32_t)cnt; cnt = 0; continue; }
            const int lead = __builtin_clzll(window) - (64 - cnt);   // zeros in front of the first 1 inside the window
            z += (uint32_t)lead;
            cnt -= lead + 1;
            return z;
        }
    }
    inline void align() { cnt -= cnt & 7; }
    inline size_t byte_pos() const { return pos - (size_t)(cnt >> 3); }   // valid when aligned
};

uint8_t crc8(const uint8_t* d, size_t n) {
    uint8_t c = 0;
    for (size_t i = 0; i < n; ++i) {
        c ^= d[i];
        for (int b = 0; b < 8; ++b) c = (uint8_t)((c & 0x80) ? ((c << 1) ^ 0x07) : (c << 1));
    }
    return c;
}
struct Crc16Table {
    uint16_t t[256];
    Crc16Table() {
        for (int i = 0; i < 256; ++i) {
            uint16_t c = (uint16_t)(i << 8);
            for (int b = 0; b < 8; ++b) c = (uint16_t)((c & 0x8000) ? ((c << 1) ^ 0x8005) : (c << 1));
            t[i] = c;
        }
    }
};
uint16_t crc16(const uint8_t* d, size_t n) {
    static const Crc16Table tab;
    uint16_t c = 0;
    for (size_t i = 0; i < n; ++i) c = (uint16_t)((c << 8) ^ tab.t[((c >> 8) ^ d[i]) & 0xff]);
    return c;
}

struct StreamInfo { int sample_rate = 0, channels = 0, bits = 0; int64_t total = 0; uint8_t md5[16] = {}; size_t audio_off = 0; int max_block = 0; };

int parse_header(const uint8_t* d, size_t n, StreamInfo& si) {
    if (n < 42 || std::memcmp(d, "fLaC", 4) != 0) { set_error("flac: no fLaC marker"); return -1; }
    size_t off = 4;
    bool have = false;
    for (;;) {
        if (off + 4 > n) { set_error("flac: truncated metadata"); return -1; }
        const bool last = (d[off] & 0x80) != 0;
        const int type = d[off] & 0x7f;
        const size_t len = ((size_t)d[off + 1] << 16) | ((size_t)d[off + 2] << 8) | d[off + 3];
        off += 4;
        if (off + len > n) { set_error("flac: truncated metadata block"); return -1; }
        if (type == 0) {
            if (len < 34) { set_error("flac: short STREAMINFO"); return -1; }
            const uint8_t* s = d + off;
            si.max_block = (s[2] << 8) | s[3];
            si.sample_rate = (s[10] << 12) | (s[11] << 4) | (s[12] >> 4);
            si.channels = ((s[12] >> 1) & 7) + 1;
            si.bits = (((s[12] & 1) << 4) | (s[13] >> 4)) + 1;
            si.total = ((int64_t)(s[13] & 0x0f) << 32) | ((int64_t)s[14] << 24) | ((int64_t)s[15] << 16) | ((int64_t)s[16] << 8) | s[17];
            std::memcpy(si.md5, s + 18, 16);
            have = true;
        }
        off += len;
        if (last) break;
    }
    if (!have || si.sample_rate <= 0) { set_error("flac: no STREAMINFO"); return -1; }
    si.audio_off = off;
    return 0;
}

// residual of one subframe into out[order .. blocksize)
bool read_residual(BitReader& br, int64_t* out, int blocksize, int order) {
    const int method = (int)br.bits(2);
    if (method > 1) return false;
    const int pbits = method == 0 ? 4 : 5, esc = method == 0 ? 15 : 31;
    const int porder = (int)br.bits(4);
    const int parts = 1 << porder;
    if ((blocksize >> porder) << porder != blocksize && porder > 0) return false;
    int i = order;
    for (int p = 0; p < parts; ++p) {
        int cnt = (blocksize >> porder) - (p == 0 ? order : 0);
        if (cnt < 0) return false;
        const int k = (int)br.bits(pbits);
        if (k == esc) {
            const int nb = (int)br.bits(5);
            for (int j = 0; j < cnt; ++j) out[i++] = nb ? br.sbits(nb) : 0;
        } else {
            for (int j = 0; j < cnt; ++j) {
                const uint64_t q = br.unary();
                const uint64_t u = (q << k) | (k ? br.bits(k) : 0u);
                out[i++] = (int64_t)(u >> 1) ^ -(int64_t)(u & 1);
            }
        }
        if (br.bad) return false;
    }
    return i == blocksize;
}

bool read_subframe(BitReader& br, int64_t* s, int blocksize, int bps) {
    if (br.bits(1) != 0) return false;
    const int type = (int)br.bits(6);
    int wasted = 0;
    if (br.bits(1)) wasted = (int)br.unary() + 1;
    bps -= wasted;
    if (bps <= 0 || bps > 33) return false;
    if (type == 0) {
        const int64_t v = br.sbits64(bps);
        for (int i = 0; i < blocksize; ++i) s[i] = v;
    } else if (type == 1) {
        for (int i = 0; i < blocksize; ++i) s[i] = br.sbits64(bps);
    } else if (type >= 8 && type <= 12) {
        const int order = type - 8;
        if (order > blocksize) return false;
        for (int i = 0; i < order; ++i) s[i] = br.sbits64(bps);
        if (!read_residual(br, s, blocksize, order)) return false;
        switch (order) {
            case 0: break;
            case 1: for (int i = 1; i < blocksize; ++i) s[i] += s[i - 1]; break;
            case 2: for (int i = 2; i < blocksize; ++i) s[i] += 2 * s[i - 1] - s[i - 2]; break;
            case 3: for (int i = 3; i < blocksize; ++i) s[i] += 3 * s[i - 1] - 3 * s[i - 2] + s[i - 3]; break;
            case 4: for (int i = 4; i < blocksize; ++i) s[i] += 4 * s[i - 1] - 6 * s[i - 2] + 4 * s[i - 3] - s[i - 4]; break;
        }
    } else if (type >= 32) {
        const int order = (type & 31) + 1;
        if (order > blocksize) return false;
        for (int i = 0; i < order; ++i) s[i] = br.sbits64(bps);
        const int prec = (int)br.bits(4) + 1;
        if (prec == 16) return false;
        const int shift = br.sbits(5);
        if (shift < 0) return false;
        int32_t coef[32];
        for (int j = 0; j < order; ++j) coef[j] = br.sbits(prec);
        if (!read_residual(br, s, blocksize, order)) return false;
        for (int i = order; i < blocksize; ++i) {
            int64_t acc = 0;
            for (int j = 0; j < order; ++j) acc += (int64_t)coef[j] * s[i - 1 - j];
            s[i] += acc >> shift;
        }
    } else {
        return false;   // reserved subframe types
    }
    if (wasted)
        for (int i = 0; i < blocksize; ++i) s[i] <<= wasted;
    return !br.bad;
}

}  // namespace
}  // namespace at

extern "C" {

int at_flac_info(const uint8_t* data, size_t n, int* sample_rate, int* channels, int* bits_per_sample, int64_t* total_samples, uint8_t* md5_16) {
    using namespace at;
    AT_REQUIRE(data != nullptr, "at_flac_info: null data");
    StreamInfo si;
    if (int rc = parse_header(data, n, si)) return rc;
    if (sample_rate) *sample_rate = si.sample_rate;
    if (channels) *channels = si.channels;
    if (bits_per_sample) *bits_per_sample = si.bits;
    if (total_samples) *total_samples = si.total;
    if (md5_16) std::memcpy(md5_16, si.md5, 16);
    return 0;
}

int64_t at_flac_decode(const uint8_t* data, size_t n, int32_t* out, int64_t cap_samples_per_channel) {
    using namespace at;
    AT_REQUIRE(data != nullptr && out != nullptr, "at_flac_decode: null pointer");
    StreamInfo si;
    if (int rc = parse_header(data, n, si)) return rc;
    AT_REQUIRE(si.bits <= 32 && si.channels >= 1 && si.channels <= 8, "flac: unsupported sample size / channel count");
    static const int kBlock[16] = {0, 192, 576, 1152, 2304, 4608, -8, -16, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768};
    static const int kBits[8] = {0, 8, 12, -1, 16, 20, 24, 32};
    std::vector<int64_t> ch[8];
    size_t off = si.audio_off;
    int64_t done = 0;      // samples per channel written so far (planar output: out[c * cap + i])
    while (off + 2 <= n && (si.total == 0 || done < si.total)) {
        if (!(data[off] == 0xff && (data[off + 1] & 0xfe) == 0xf8)) { set_error("flac: lost frame sync"); return -1; }
        BitReader br(data + off, n - off);
        br.bits(14);
        if (br.bits(1)) { set_error("flac: reserved bit set in a frame header"); return -1; }
        br.bits(1);                                 // blocking strategy: only changes the meaning of the coded number
        const int bs_code = (int)br.bits(4), sr_code = (int)br.bits(4), ch_code = (int)br.bits(4), ss_code = (int)br.bits(3);
        if (br.bits(1)) { set_error("flac: reserved bit set in a frame header"); return -1; }
        {   // UTF-8-style coded frame / sample number: skip its continuation bytes
            const uint32_t b0 = br.bits(8);
            int extra = 0;
            if (b0 >= 0xfe) extra = 6; else if (b0 >= 0xfc) extra = 5; else if (b0 >= 0xf8) extra = 4; else if (b0 >= 0xf0) extra = 3;
            else if (b0 >= 0xe0) extra = 2; else if (b0 >= 0xc0) extra = 1; else if (b0 >= 0x80) { set_error("flac: bad coded number"); return -1; }
            for (int i = 0; i < extra; ++i) if ((br.bits(8) & 0xc0) != 0x80) { set_error("flac: bad coded number"); return -1; }
        }
        int blocksize = kBlock[bs_code];
        if (bs_code == 0) { set_error("flac: reserved block size"); return -1; }
        if (blocksize == -8) blocksize = (int)br.bits(8) + 1;
        else if (blocksize == -16) blocksize = (int)br.bits(16) + 1;
        if (sr_code == 12) br.bits(8); else if (sr_code == 13 || sr_code == 14) br.bits(16); else if (sr_code == 15) { set_error("flac: invalid sample rate code"); return -1; }
        const size_t hdr_len = br.byte_pos();
        const uint8_t want8 = (uint8_t)br.bits(8);
        if (br.bad || crc8(data + off, hdr_len) != want8) { set_error("flac: frame header CRC-8 mismatch"); return -1; }
        const int bps = ss_code == 0 ? si.bits : kBits[ss_code];
        if (bps <= 0) { set_error("flac: reserved sample size"); return -1; }
        const int nch = ch_code < 8 ? ch_code + 1 : 2;
        if (ch_code > 10 || nch != si.channels) { set_error("flac: channel assignment does not match STREAMINFO"); return -1; }
        for (int c = 0; c < nch; ++c) {
            ch[c].resize((size_t)blocksize);
            const bool side = (ch_code == 8 && c == 1) || (ch_code == 9 && c == 0) || (ch_code == 10 && c == 1);
            if (!read_subframe(br, ch[c].data(), blocksize, bps + (side ? 1 : 0))) { set_error("flac: damaged subframe"); return -1; }
        }
        br.align();
        const size_t body_end = br.byte_pos();
        const uint16_t want16 = (uint16_t)br.bits(16);
        if (br.bad || crc16(data + off, body_end) != want16) { set_error("flac: frame CRC-16 mismatch"); return -1; }
        if (ch_code == 8) { for (int i = 0; i < blocksize; ++i) ch[1][i] = ch[0][i] - ch[1][i]; }
        else if (ch_code == 9) { for (int i = 0; i < blocksize; ++i) ch[0][i] = ch[0][i] + ch[1][i]; }
        else if (ch_code == 10) {
            for (int i = 0; i < blocksize; ++i) {
                const int64_t side = ch[1][i], mid = (ch[0][i] << 1) | (side & 1);
                ch[0][i] = (mid + side) >> 1;
                ch[1][i] = (mid - side) >> 1;
            }
        }
        int64_t take = blocksize;
        if (si.total > 0 && done + take > si.total) take = si.total - done;
        if (done + take > cap_samples_per_channel) { set_error("flac: output buffer too small"); return -1; }
        for (int c = 0; c < nch; ++c)
            for (int64_t i = 0; i < take; ++i) out[c * cap_samples_per_channel + done + i] = (int32_t)ch[c][(size_t)i];
        done += take;
        off += body_end + 2;
    }
    if (si.total > 0 && done != si.total) { set_error("flac: stream ends before STREAMINFO's sample count"); return -1; }
    return done;
}

}  // extern "C"
