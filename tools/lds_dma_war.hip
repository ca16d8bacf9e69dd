// Reduced reproducer of the "sporadically wrong tiles with two workgroups per CU" of round 2's shared-DMA variant of the two-group GEMM
// (csrc/gemm_f16x2_tg.hip; commit b1938cd, withdrawn in c45fa3b). Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/lds_dma_war.hip -o tools/lds_dma_war
//
// ROOT CAUSE (what this tool demonstrates): a write-after-read race INSIDE one barrier interval. In that variant a trailing wave issued the
// LDS-DMA of the weight chunks of K step kp + 2 into the ring pair of step kp "when every read of that pair has returned — the leaders' at
// the barrier, their own by lgkmcnt(0)". `s_waitcnt lgkmcnt(0)` retires only the ISSUING wave's ds_reads. The weight rows a trailing wave
// refills (rows 64 (w & 3) .. + 63 of the chunk) are read by ALL FOUR trailing waves in that same interval (they share wn = 1), and nothing
// orders a sibling's still-outstanding ds_read against the DMA write (MI355X_MICROARCH.md, "Nothing orders a ds_read behind a pending LDS-DMA
// except the issuing wave's covering vmcnt plus a barrier"; cdna_hip_programming.md, "WAR: restage a buffer >= 2 phases after its last
// ds_read, or 1 phase after when an lgkmcnt before the reading phase's first barrier retired those reads" — this was 0 phases after).
// With ONE workgroup per CU the four trailing waves sit on four different SIMDs next to leaders doing identical MFMA bursts, so their skew is a
// few hundred cycles and a DMA (issue -> landed 250-400 cycles from L2, ~2 k from HBM) never overtook a sibling's reads. With TWO workgroups
// per CU a sibling can be held back by the other workgroup's s_setprio(1) MFMA segment (~1 600 cycles) — longer than the DMA's flight.
//
// Protocols (argv[1]):
//   0  shipped in round 2: only the leading group issues DMA, one step ahead (safe: the refill starts one barrier after the trailers' reads);
//   1  the withdrawn variant: trailers refill W two steps ahead into the pair they are reading, after their OWN lgkmcnt(0) (racy);
//   2  the fix shipped in round 3: same issue split, but the weight ring has THREE pairs — the pair refilled in interval 2 kp + 1 is the one
//      read in steps kp - 1 (closed by two barriers), never the one being read.
// argv[2] = workgroups per CU (1: the launch asks for > 80 KB of LDS, 2: as the 128 x 128 product shape);
// argv[3] = wave to delay (-1 none), argv[4] = delay in s_sleep units of 64 cycles, applied before that wave's fragment reads of every step:
//           with protocol 1 a delayed TRAILING wave (4..7) turns the sporadic failure into a deterministic one at ANY occupancy, and
//           protocols 0 / 2 stay exact under the same delay.
// The data make every output element an exact small integer (A[m][k] = 1 + m % 3, W[n][k] = w(n, k / 16) in {-3..3}, lo pieces 0), so a
// single stale or early fragment shows up as a wrong integer; the tool prints wrong elements / tiles per launch over `reps` launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int TI = 2, TJ = 4, BM = 128;            // the 128 x 128 shape of the product kernel: 8 waves, 4 x 2, each 32 x 64
constexpr int PIECE = BM * 16;                     // fp16 elements of one (piece, k-block) chunk
constexpr int AB = 2 * PIECE;                      // one k-block of one operand: hi, lo

__host__ __device__ inline int wval(int n, int kb) { return (int)(((unsigned)n * 2654435761u + (unsigned)kb * 40503u) >> 7) % 7 - 3; }

// A pieces [2][K/16][M][16], W pieces [2][K/16][N][16] (the product's K-blocked layout). LDS: A ring 2 pairs x 2 k-blocks x AB, then the W ring
// (2 or 3 pairs) x 2 k-blocks x AB.
template <int PROTO>
__global__ __launch_bounds__(512, 4) void war_kernel(const _Float16* A, const _Float16* W, float* C, int M, int N, int K, int delay_wave, int delay) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    _Float16* ldsA = reinterpret_cast<_Float16*>(lds_raw);
    _Float16* ldsW = ldsA + 2 * 2 * AB;
    constexpr int WPAIRS = PROTO == 2 ? 3 : 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
    const int wm = wave & 3, wn = wave >> 2;
    const int ntn = N / BM;
    const int nt = blockIdx.x % ntn, mt = blockIdx.x / ntn;
    const int m0 = mt * BM, n0 = nt * BM;
    const int nk2 = K / 32;
    const long long psA = (long long)M * K, psW = (long long)N * K;
    const int srow = (wave & 3) * (BM / 4) + (lane >> 1), shalf = lane & 1;
    const _Float16* gA = A + ((long long)m0 + srow) * 16 + shalf * 8;
    const _Float16* gW = W + ((long long)n0 + srow) * 16 + shalf * 8;
    auto issue_A = [&](int kp) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            _Float16* s = ldsA + ((kp & 1) * 2 + h) * AB + (wave & 3) * (BM / 4) * 16;
#pragma unroll
            for (int p = 0; p < 2; ++p)
                __builtin_amdgcn_global_load_lds((glb_void*)(gA + p * psA + (long long)(2 * kp + h) * M * 16), (lds_void*)(s + p * PIECE), 16, 0, 0);
        }
    };
    auto issue_W = [&](int kp, int pair) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            _Float16* s = ldsW + (pair * 2 + h) * AB + (wave & 3) * (BM / 4) * 16;
#pragma unroll
            for (int p = 0; p < 2; ++p)
                __builtin_amdgcn_global_load_lds((glb_void*)(gW + p * psW + (long long)(2 * kp + h) * N * 16), (lds_void*)(s + p * PIECE), 16, 0, 0);
        }
    };
    const int fr = lane & 15, fq = lane >> 4;
    const int foff = (fq >> 1) * AB + fr * 16 + (fq & 1) * 8;
    f4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    if (grp == 0) { issue_A(0); issue_W(0, 0); }
    if (PROTO != 0 && grp == 1 && nk2 > 1) issue_W(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    int wpair = 0;                                    // W ring pair of step kp (kp % WPAIRS)
    for (int kp = 0; kp < nk2; ++kp) {
        if (wave == delay_wave)
            for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(1);
        const _Float16* sa = ldsA + (kp & 1) * 2 * AB + foff;
        const _Float16* sw = ldsW + wpair * 2 * AB + foff;
        f16x8 xa[2][TI], wb[2][TJ];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int i = 0; i < TI; ++i) xa[p][i] = *reinterpret_cast<const f16x8*>(sa + p * PIECE + (wm * TI * 16 + i * 16) * 16);
#pragma unroll
            for (int j = 0; j < TJ; ++j) wb[p][j] = *reinterpret_cast<const f16x8*>(sw + p * PIECE + (wn * TJ * 16 + j * 16) * 16);
        }
        const int wnext = wpair + 1 == WPAIRS ? 0 : wpair + 1;
        if (grp == 0 && kp + 1 < nk2) { issue_A(kp + 1); if (PROTO == 0) issue_W(kp + 1, wnext); }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (PROTO != 0 && grp == 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // W(kp + 1), issued one period ago, has landed
            // PROTO 1: into the pair THIS step reads (own reads retired, the siblings' not ordered); PROTO 2: into the third pair
            if (kp + 2 < nk2) issue_W(kp + 2, PROTO == 1 ? wpair : (wnext + 1 == WPAIRS ? 0 : wnext + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[t == 0 ? 1 : 0][j], xa[t == 1 ? 1 : 0][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        wpair = wnext;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int m = m0 + wm * TI * 16 + i * 16 + fr, n = n0 + wn * TJ * 16 + j * 16 + 4 * fq;
            *reinterpret_cast<f4*>(C + (long long)m * N + n) = acc[i][j];
        }
}

int main(int argc, char** argv) {
    const int proto = argc > 1 ? std::atoi(argv[1]) : 1;
    const int per_cu = argc > 2 ? std::atoi(argv[2]) : 2;
    const int delay_wave = argc > 3 ? std::atoi(argv[3]) : -1;
    const int delay = argc > 4 ? std::atoi(argv[4]) : 0;
    const int reps = argc > 5 ? std::atoi(argv[5]) : 50;
    const int M = 128 * 256, N = 512, K = 1024;       // 1024 tiles = 2 waves of 512 (two per CU) or 4 of 256
    std::vector<_Float16> hA((size_t)2 * M * K, (_Float16)0.f), hW((size_t)2 * N * K, (_Float16)0.f);
    for (int kb = 0; kb < K / 16; ++kb) {
        for (int m = 0; m < M; ++m) for (int c = 0; c < 16; ++c) hA[((size_t)kb * M + m) * 16 + c] = (_Float16)(float)(1 + m % 3);
        for (int n = 0; n < N; ++n) for (int c = 0; c < 16; ++c) hW[((size_t)kb * N + n) * 16 + c] = (_Float16)(float)wval(n, kb);
    }
    std::vector<float> colsum(N, 0.f);
    for (int n = 0; n < N; ++n) { int s = 0; for (int kb = 0; kb < K / 16; ++kb) s += wval(n, kb); colsum[n] = 16.f * s; }
    _Float16 *dA, *dW; float* dC;
    CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    const size_t need = (size_t)(2 * 2 * AB + (proto == 2 ? 3 : 2) * 2 * AB) * 2;
    const size_t lds = per_cu == 1 ? (size_t)100 * 1024 : need;
    auto launch = [&]() {
        const dim3 grid(M / BM * (N / BM));
        if (proto == 0) hipLaunchKernelGGL(war_kernel<0>, grid, dim3(512), lds, 0, dA, dW, dC, M, N, K, delay_wave, delay);
        else if (proto == 1) hipLaunchKernelGGL(war_kernel<1>, grid, dim3(512), lds, 0, dA, dW, dC, M, N, K, delay_wave, delay);
        else hipLaunchKernelGGL(war_kernel<2>, grid, dim3(512), lds, 0, dA, dW, dC, M, N, K, delay_wave, delay);
    };
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(war_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(war_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(war_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    std::vector<float> hC((size_t)M * N);
    long long bad_elems = 0, bad_tiles = 0, bad_launches = 0;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms_total = 0.f;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms_total += ms;
        CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
        long long be = 0, bt = 0;
        for (int mt = 0; mt < M / BM; ++mt)
            for (int nt = 0; nt < N / BM; ++nt) {
                long long e = 0;
                for (int m = mt * BM; m < (mt + 1) * BM; ++m)
                    for (int n = nt * BM; n < (nt + 1) * BM; ++n)
                        if (hC[(size_t)m * N + n] != (1 + m % 3) * colsum[n]) ++e;
                be += e; bt += e > 0;
            }
        bad_elems += be; bad_tiles += bt; bad_launches += be > 0;
    }
    std::printf("{\"tool\": \"lds_dma_war\", \"protocol\": %d, \"workgroups_per_cu\": %d, \"lds_bytes\": %zu, \"delay_wave\": %d, \"delay_x64cyc\": %d, \"launches\": %d, "
                "\"launches_with_wrong_tiles\": %lld, \"wrong_tiles\": %lld, \"wrong_elements\": %lld, \"avg_ms\": %.4f}\n",
                proto, per_cu, lds, delay_wave, delay, reps, bad_launches, bad_tiles, bad_elems, ms_total / reps);
    return 0;
}
