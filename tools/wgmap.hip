// Which workgroups of a 512-WG launch (2 per CU by LDS/registers) share a CU? Prints, for a few blocks, the XCC / CU they ran
// on and the partner block on the same CU. Build: hipcc -O3 --offload-arch=gfx950 tools/wgmap.hip -o tools/wgmap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void k(unsigned* out, int spin) {
    extern __shared__ float lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    lds[threadIdx.x] = (float)hw;
    // stay resident long enough for the whole grid to be placed
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; }
}
int main() {
    const int G = 512;
    unsigned* d; hipMalloc(&d, G * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipLaunchKernelGGL(k, dim3(G), dim3(256), 72 * 1024, 0, d, 200000);   // 2 ms at 100 MHz
    hipDeviceSynchronize();
    std::vector<unsigned> h(G * 2); hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;
    for (int i = 0; i < G; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        const unsigned cu_id = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cu[(xcc << 12) | (se << 8) | (sh << 4) | cu_id].push_back(i);
    }
    printf("distinct CUs used: %zu\n", cu.size());
    int shown = 0;
    for (auto& kv : cu) { if (shown++ < 12) { printf("xcc %u se %u sh %u cu %u:", kv.first >> 12, (kv.first >> 8) & 7, (kv.first >> 4) & 1, kv.first & 15); for (int b : kv.second) printf(" %d", b); printf("\n"); } }
    std::map<int, int> diffs;
    for (auto& kv : cu) if (kv.second.size() == 2) diffs[kv.second[1] - kv.second[0]]++;
    for (auto& d2 : diffs) printf("partner distance %d: %d CUs\n", d2.first, d2.second);
    return 0;
}
