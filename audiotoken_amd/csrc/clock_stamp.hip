// at_clock_stamp: which clock did the chip hold over a stretch of the stream? (under dense MFMA load an MI355X runs anywhere
// between ~1.5 and ~2.4 GHz, boxes differ by several per cent, and a benchmark line that does not record the clock cannot tell a better kernel from a
// faster box.) A 64-workgroup x 1-wave kernel; every wave writes {s_memtime, s_memrealtime} into the slot of ITS XCD (HW_REG_XCC_ID): s_memtime ticks
// with the shader clock, s_memrealtime at a constant 100 MHz (MI355X_MICROARCH.md, "DVFS give-back" (6)), so between two stamps on the same stream
//     held clock = (memtime_b - memtime_a) / (memrealtime_b - memrealtime_a) x 100 MHz,
// taken per XCD (the counters of different XCDs are not synchronised with each other) and reported as the median over the XCDs both stamps reached.
// The stamps bracket whole encodes from OUTSIDE (two ~5 us launches per bracket): no product kernel carries a stamp.
#include "at_common.h"
#include "../../include/audiotoken_hip.h"

namespace at {
__global__ __launch_bounds__(64) void clock_stamp_kernel(unsigned long long* __restrict__ slots) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    const unsigned long long r = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        unsigned long long* s = slots + 2 * (xcc & 15u);
        // (several workgroups of one XCD write the same slot within a microsecond of each other: any of them is the stamp; the pair is written by one lane,
        // and a torn pair — t from one wave, r from another — is off by that microsecond at most, against brackets of tens of milliseconds)
        __hip_atomic_store(s, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(s + 1, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
}  // namespace at

extern "C" int at_clock_stamp(uint64_t* slots_dev, at_stream_t stream) {
    using namespace at;
    AT_REQUIRE(slots_dev != nullptr, "at_clock_stamp: null slots");
    hipLaunchKernelGGL(clock_stamp_kernel, dim3(64), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<unsigned long long*>(slots_dev));
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}
