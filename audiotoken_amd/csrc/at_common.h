// Shared declarations for the audiotoken HIP library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include <cstdlib>

namespace at {

// ---- error plumbing (C-ABI returns codes; text via at_last_error) ---------------------------
void set_error(const std::string& msg);
#define AT_CHECK_HIP(expr)                                                                      \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            ::at::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                 \
            return -2;                                                                          \
        }                                                                                       \
    } while (0)
#define AT_REQUIRE(cond, msg)                                                                   \
    do {                                                                                        \
        if (!(cond)) {                                                                          \
            ::at::set_error(std::string("requirement failed: ") + #cond + " — " + (msg));       \
            return -1;                                                                          \
        }                                                                                       \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property of a kernel: a process that opens handles on several
// devices (the C ABI takes a device index) must set it on each. `flag` is a function-local static array indexed by device.
constexpr int kMaxDevices = 64;
struct LdsAttrFlags { bool set[kMaxDevices] = {}; };
template <typename K>
inline int set_max_dynamic_lds(LdsAttrFlags& f, K kernel, size_t bytes) {
    int dev = 0;
    AT_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices || !f.set[dev]) {
        AT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        if (dev >= 0 && dev < kMaxDevices) f.set[dev] = true;
    }
    return 0;
}
// compute units of the current device (per-device cache: a process may hold handles on several devices)
inline int device_cus() {
    static int cus[kMaxDevices] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 256;
    if (cus[dev] == 0) {
        int n = 0;
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cus[dev];
}
// $AUDIOTOKEN_HOST_ONLY_TEST=1: the sanitizer build's CPU test (tests/test_asan_cpu.py) runs tensor staging and the host-side packing of
// finalize() on a machine without a device; handles can then be created and the first real device call fails with an error code as usual
inline bool host_only_test() {
    const char* e = std::getenv("AUDIOTOKEN_HOST_ONLY_TEST");
    return e && e[0] == '1';
}
// RAII: make `device` current for the duration of a C-ABI call and restore the caller's device afterwards
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) { prev = -1; ok = host_only_test(); return; }
        if (prev != device && hipSetDevice(device) != hipSuccess) ok = false;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// Unsigned division by a run-time constant without the hardware's reciprocal sequence (Granlund-Montgomery): the host prepares (magic, sh1, sh2)
// from the divisor, the device computes n / d = (t + ((n - t) >> sh1)) >> sh2 with t = mulhi(n, magic), exact for every 32-bit n. Uniform operands
// stay on the scalar unit (s_mul_hi_u32); the compiler's own expansion of `/` goes through v_rcp_iflag_f32 on the VECTOR unit even for uniform values
// and hoists the reciprocal out of loops — two registers held across the whole two-group GEMM kernel, which has none to spare.
struct FastDivU {
    unsigned magic = 1, sh1 = 0, sh2 = 0, d = 1;
    FastDivU() = default;
    explicit FastDivU(unsigned div) : d(div ? div : 1) {
        unsigned l = 0;
        while ((1ull << l) < d) ++l;
        magic = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
        sh1 = l < 1 ? l : 1;
        sh2 = l > 1 ? l - 1 : 0;
    }
    __host__ __device__ __forceinline__ unsigned div(unsigned n) const {
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned t = __umulhi(n, magic);
#else
        const unsigned t = (unsigned)(((unsigned long long)n * magic) >> 32);
#endif
        return (t + ((n - t) >> sh1)) >> sh2;
    }
};

// ---- the one dense contraction every layer maps onto ----------------------------------------
// out[b][m][n] = epi( alpha * ( sum_kk A(b,m,kk) * W[n][kk] + bias[n] ) ) (+ R[b][m][n])
// where A(b,m,kk) = pro( X[b][ row(m, kk / Cin) ][ kk % Cin ] ),  row(m,tap) = m*stride + tap - pad_left,
// rows outside [0,Tin) are reflected (pad_mode 1) or read as zero (pad_mode 0).
// Activations are time-major / channels-last ([B][T][C], C contiguous), so a causal conv1d, a strided
// conv1d, a transposed conv1d (as s-phase GEMM) and a Linear layer are all this one contraction with a
// window of K = ktaps*Cin contiguous floats per output row — no im2col buffer ever exists.
enum Prologue { PRO_NONE = 0, PRO_ELU = 1, PRO_POWER = 2 };  // POWER: A = X[kk]^2 + X[kk + aux_off]^2 (|DFT|^2)
enum Epilogue {
    EPI_NONE = 0, EPI_SWISH = 1, EPI_ELU = 2, EPI_GELU = 3,
    EPI_LOGFLOOR = 4,  // log(max(v, floor)) with floor = alpha-free constant 1.192092955078125e-07 (mel floor)
    EPI_GLU = 5        // weight rows interleaved (a_c, b_c): out[m][n/2] = a * sigmoid(b); C has N/2 columns
};

struct GemmArgs {
    const float* X = nullptr;  // [batch][Tin][Cin]
    long long x_bstride = 0;   // floats between clips
    int Tin = 0, Cin = 0;
    int ldx = 0;               // floats between consecutive input rows (= Cin for a dense [T][Cin] clip)
    int ktaps = 1, stride = 1, pad_left = 0, pad_mode = 0;
    const float* X2 = nullptr;    // optional second A source for k >= K1 (ktaps == 1): same row mapping, no prologue
    long long x2_bstride = 0;
    int ld2 = 0, K1 = 0;
    const float* W = nullptr;     // [N][K], K = ktaps*Cin (+ width of X2), tap-major then channel
    const float* bias = nullptr;  // [N] or null
    float* C = nullptr;           // [batch][M][ldc]
    long long c_bstride = 0;
    int ldc = 0;
    const float* R = nullptr;  // residual, same indexing as C (may alias C)
    long long r_bstride = 0;
    int ldr = 0;
    int M = 0, N = 0, K = 0, batch = 1;
    int pro = PRO_NONE, epi = EPI_NONE;
    float alpha = 1.0f;
    int aux_off = 0;                  // PRO_POWER: offset (floats) of the imaginary half within an input row
    const float* row_mask = nullptr;  // optional [batch*M]: output rows whose mask is 0 are written as 0
};

int launch_gemm(const GemmArgs& a, hipStream_t stream);
int check_gemm_args(const GemmArgs& a);

// ---- optional per-group timing with HIP events on the launch stream (bench.py roofline) -------
struct Profiler {
    bool enabled = false;
    struct Span { int group; int launches; hipEvent_t a, b; };
    std::vector<std::string> names;
    std::vector<Span> spans;
    std::vector<hipEvent_t> pool;
    int open_ = -1;
    hipEvent_t get_event();
    void begin(const char* group, int launches, hipStream_t s);
    void end(hipStream_t s);
    void reset();
    // sums elapsed ms per group; synchronises on the recorded events
    int read(std::vector<float>& ms, std::vector<int>& launches);
    ~Profiler();
};

// ---- small helpers shared by kernels ---------------------------------------------------------
// ELU(alpha = 1) = x > 0 ? x : expm1(x) as v_mul, v_exp_f32, v_add, v_cmp, v_cndmask. Absolute error <= 1.2e-7 (one ulp
// of exp(x) <= 1): below the fp32 rounding of the activations it is added to. It is NOT relative-accurate near 0
// (ocml expm1f is, at ~4x the instruction count) — and instruction count is what matters here: on gfx950 the fp32
// MFMA and the VALU of a SIMD do not overlap (tools/coissue.hip: MFMA-only 1.90 ms + VALU-only 1.22 ms = 2.94 ms
// together), so every ELU instruction in a conv kernel is time taken from the matrix pipe.
__device__ __forceinline__ float elu1(float x) {
    const float e = __expf(x) - 1.0f;
    return x > 0.0f ? x : e;
}
// sigmoid on v_exp_f32 / v_rcp_f32 (~1 ulp each): GEMM epilogues run on the VALU the fp32 MFMA cannot overlap with
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// LSTM gate activations on the hardware transcendental units (v_exp_f32 / v_rcp_f32, ~1 ulp each) instead of the
// branchy ocml expf/tanhf: the cell update sits on the critical path of every one of the 750 dependent steps.
// Absolute error < 2e-7 (tanh switches to its odd Taylor polynomial below |x| = 1/8 where 1 - 2/(e^2x + 1) cancels).
__device__ __forceinline__ float lstm_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float lstm_tanh(float x) {
    const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f);
    const float x2 = x * x;
    float p = fmaf(x2, -5.3968254e-2f, 1.3333333e-1f);
    p = fmaf(x2, p, -3.3333333e-1f);
    p = fmaf(x2 * x, p, x);
    return fabsf(x) < 0.125f ? p : big;
}
// Exact-form GELU 0.5 y (1 + erf(y / sqrt 2)) with erfc from Abramowitz-Stegun 7.1.26 on v_rcp_f32 / v_exp_f32:
// |error| <= 4.3e-7 over [-10, 10] (about one ulp of the result where it is largest), ~20 VALU issue slots instead of
// ~45 for ocml erff — HuBERT evaluates 19 G of these per step and VALU time is not hidden behind the fp32 MFMA.
// half = 0.5 s: returns s * GELU(y) for a power of two s, bit for bit s times gelu_erf(y) (only the leading multiply's constant changes)
__device__ __forceinline__ float gelu_erf_scaled(float y, float half) {
    const float x = fabsf(y) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    const float e = (p * t) * __expf(-x * x);        // erfc(|y| / sqrt 2)
    return half * y * (y < 0.0f ? e : 2.0f - e);
}
__device__ __forceinline__ float gelu_erf(float y) { return gelu_erf_scaled(y, 0.5f); }
// two values at once on the packed fp32 instructions (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: two IEEE operations per issue slot; the reciprocal, the
// exponential and the sign select stay per element). The same operations in the same order as gelu_erf_scaled: bit-identical per element.
__device__ __forceinline__ auto gelu_erf_scaled2(float __attribute__((ext_vector_type(2))) y, float half) -> float __attribute__((ext_vector_type(2))) {
    typedef float f2_ __attribute__((ext_vector_type(2)));
    // x = y / sqrt 2 keeps its sign: |x| enters only the denominator below, as the abs MODIFIER of a plain v_fma_f32 (the packed fma has none: |y| first cost a
    // v_and_b32 per element), and -x * x is sign-blind; |y| c and |y c| are the same float, so every value is bit for bit that of gelu_erf_scaled
    const f2_ x = y * 0.70710678118654752440f;
    const f2_ t = {__builtin_amdgcn_rcpf(fmaf(0.3275911f, fabsf(x[0]), 1.0f)), __builtin_amdgcn_rcpf(fmaf(0.3275911f, fabsf(x[1]), 1.0f))};
    f2_ p = __builtin_elementwise_fma(t, f2_{1.061405429f, 1.061405429f}, f2_{-1.453152027f, -1.453152027f});
    p = __builtin_elementwise_fma(t, p, f2_{1.421413741f, 1.421413741f});
    p = __builtin_elementwise_fma(t, p, f2_{-0.284496736f, -0.284496736f});
    p = __builtin_elementwise_fma(t, p, f2_{0.254829592f, 0.254829592f});
    const f2_ a = -x * x;
    const f2_ e = (p * t) * f2_{__expf(a[0]), __expf(a[1])};
    const f2_ r = half * y;
    const f2_ two_minus = 2.0f - e;
    return r * f2_{y[0] < 0.0f ? e[0] : two_minus[0], y[1] < 0.0f ? e[1] : two_minus[1]};
}
// swish of the conformer conv module's depthwise-conv kernels: x * sigmoid(x) on v_exp_f32 / v_rcp_f32 (~1 ulp each) like the GEMM's swish and GLU
// epilogues (until late in round 3: ocml expf + a correctly rounded division, ~30 more vector instructions per value in a vector-bound kernel)
__device__ __forceinline__ float swishf_(float x) { return x * sigmoidf_(x); }

}  // namespace at
