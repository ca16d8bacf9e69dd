// Device allocations of a finalized model in allocation order — and the replay of that order over ONE packed device blob
// (at_*_export_packed / at_*_import_packed, include/audiotoken_hip.h).
//
// finalize() of the two semantic models makes a few hundred device allocations (uploaded tensors, host-packed tensors, operand pieces split on
// the device). Their ORDER and SIZES depend only on the architecture (layer count, code book presence, arithmetic), not on the weight values. So a
// finalized handle can be shipped as the concatenation of its allocations (256-byte aligned) plus a small host-side record {sizes, max |w| of every
// uploaded tensor (the fp16 scheme's weight scales derive from it), layer count, flags}; the receiving side runs the SAME finalize code with the arena
// in import mode: every alloc() becomes the next slice of the blob (validated against the recorded size), host -> device uploads and the device-side
// split kernels are skipped (the bytes are already there), and no host tensor is needed. One rank reads / folds / uploads / splits the checkpoint;
// the others receive one RCCL broadcast of the blob (weights travel over xGMI once) — no D2H copy, no second host-side pass.
#pragma once
#include "at_common.h"

#include <cstdint>
#include <cstring>
#include <vector>

namespace at {

struct PackedHeader {
    uint32_t magic;      // 'ATPK'
    uint32_t version;
    uint32_t model;      // PACKED_MODEL_*
    uint32_t n_blocks;
    uint64_t blob_bytes;
    int32_t n_layers;
    int32_t flags;       // model-specific (code book present, which schemes are split)
    int32_t arith;
    int32_t reserved;
};
struct PackedBlock {
    uint64_t bytes;
    float wmax;          // max |w| of an uploaded fp32 tensor, 0 for derived data
    uint32_t reserved;
};
constexpr uint32_t kPackedMagic = 0x4b505441u, kPackedVersion = 1;
enum { PACKED_MODEL_W2VBERT = 1, PACKED_MODEL_HUBERT = 2 };

struct DeviceArena {
    struct Block { void* p; size_t bytes; float wmax; };
    std::vector<Block> blocks;          // allocation order
    bool importing = false;             // alloc() slices `blob` instead of calling hipMalloc
    char* blob = nullptr;               // import: the handle's single allocation
    size_t blob_bytes = 0, cur = 0;
    std::vector<PackedBlock> expect;    // import: the exporter's blocks, consumed one per alloc()
    size_t imported = 0;                // blocks that live in `blob` (never freed one by one)

    static size_t align(size_t n) { return (n + 255) / 256 * 256; }

    // the next allocation; nullptr (error set) on failure. Import mode: the next recorded block, which must have exactly this size
    void* alloc(size_t bytes) {
        if (importing) {
            const size_t idx = blocks.size();
            if (idx >= expect.size() || expect[idx].bytes != bytes || cur + align(bytes) > blob_bytes) {
                set_error("packed model does not match this build: block " + std::to_string(idx) + " wants " + std::to_string(bytes) + " bytes, the blob records " +
                          (idx < expect.size() ? std::to_string(expect[idx].bytes) : std::string("no such block")));
                return nullptr;
            }
            void* p = blob + cur;
            cur += align(bytes);
            blocks.push_back({p, bytes, expect[idx].wmax});
            imported = blocks.size();
            return p;
        }
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) { set_error("device allocation of " + std::to_string(bytes) + " bytes failed"); return nullptr; }
        blocks.push_back({p, bytes, 0.f});
        return p;
    }
    size_t packed_bytes() const {
        size_t n = 0;
        for (const Block& b : blocks) n += align(b.bytes);
        return n;
    }
    void free_all() {
        for (size_t i = imported; i < blocks.size(); ++i) (void)hipFree(blocks[i].p);
        if (blob) (void)hipFree(blob);
        blocks.clear(); blob = nullptr; imported = 0;
    }
};

// host record of a finalized arena: header + one PackedBlock per allocation. Returns its size; writes it when `cap` is large enough
inline int64_t packed_write_meta(const DeviceArena& a, uint32_t model, int n_layers, int flags, int arith, void* dst, int64_t cap) {
    const int64_t need = (int64_t)(sizeof(PackedHeader) + a.blocks.size() * sizeof(PackedBlock));
    if (!dst || cap < need) return need;
    PackedHeader hd{kPackedMagic, kPackedVersion, model, (uint32_t)a.blocks.size(), (uint64_t)a.packed_bytes(), n_layers, flags, arith, 0};
    std::memcpy(dst, &hd, sizeof(hd));
    PackedBlock* pb = reinterpret_cast<PackedBlock*>(static_cast<char*>(dst) + sizeof(hd));
    for (size_t i = 0; i < a.blocks.size(); ++i) pb[i] = PackedBlock{(uint64_t)a.blocks[i].bytes, a.blocks[i].wmax, 0};
    return need;
}
// concatenate the allocations into `dst` (device, >= packed_bytes()); asynchronous on `stream`
inline int packed_export(const DeviceArena& a, void* dst, int64_t bytes, hipStream_t stream) {
    AT_REQUIRE(dst && bytes >= (int64_t)a.packed_bytes(), "export_packed: destination smaller than packed_bytes()");
    size_t cur = 0;
    for (const DeviceArena::Block& b : a.blocks) {
        AT_CHECK_HIP(hipMemcpyAsync(static_cast<char*>(dst) + cur, b.p, b.bytes, hipMemcpyDeviceToDevice, stream));
        cur += DeviceArena::align(b.bytes);
    }
    return 0;
}
// put a FRESH arena into import mode: validate the record, allocate the handle's own blob and copy `src` (device) into it. The copy has completed when
// this returns (the finalize replay that follows is host code)
inline int packed_begin_import(DeviceArena& a, uint32_t model, const void* meta, int64_t meta_bytes, const void* src, int64_t bytes, hipStream_t stream, PackedHeader* out) {
    AT_REQUIRE(a.blocks.empty() && !a.blob, "import_packed needs a fresh handle");
    AT_REQUIRE(meta && meta_bytes >= (int64_t)sizeof(PackedHeader), "import_packed: metadata too short");
    PackedHeader hd;
    std::memcpy(&hd, meta, sizeof(hd));
    AT_REQUIRE(hd.magic == kPackedMagic && hd.version == kPackedVersion, "import_packed: not a packed-model record of this version");
    AT_REQUIRE(hd.model == model, "import_packed: the record belongs to a different model");
    AT_REQUIRE(meta_bytes >= (int64_t)(sizeof(PackedHeader) + (size_t)hd.n_blocks * sizeof(PackedBlock)), "import_packed: metadata truncated");
    AT_REQUIRE(src && bytes >= (int64_t)hd.blob_bytes, "import_packed: blob smaller than the record says");
    a.expect.resize(hd.n_blocks);
    std::memcpy(a.expect.data(), static_cast<const char*>(meta) + sizeof(hd), (size_t)hd.n_blocks * sizeof(PackedBlock));
    size_t total = 0;
    for (const PackedBlock& b : a.expect) total += DeviceArena::align((size_t)b.bytes);
    AT_REQUIRE(total == hd.blob_bytes, "import_packed: block sizes do not add up to the blob size");
    AT_CHECK_HIP(hipMalloc((void**)&a.blob, (size_t)hd.blob_bytes));
    a.blob_bytes = (size_t)hd.blob_bytes;
    a.cur = 0;
    AT_CHECK_HIP(hipMemcpyAsync(a.blob, src, (size_t)hd.blob_bytes, hipMemcpyDeviceToDevice, stream));
    AT_CHECK_HIP(hipStreamSynchronize(stream));
    a.importing = true;
    *out = hd;
    return 0;
}
// after the finalize replay: every recorded block must have been consumed
inline int packed_end_import(DeviceArena& a) {
    a.importing = false;
    AT_REQUIRE(a.blocks.size() == a.expect.size() && a.cur == a.blob_bytes, "import_packed: the blob holds more blocks than this build's finalize makes");
    a.expect.clear();
    return 0;
}

}  // namespace at
