// Where does seanet_down64x3_kernel's tile time go? Includes the product kernel with parts compiled out (-DDX_PROBE_*) and times
// it on the bench shape (256 clips x 120000 rows). Build one binary per variant: see tools/README.md.
#include "../audiotoken_amd/csrc/seanet_down64x3.hip"
#include <cstdio>
namespace at { void set_error(const std::string& m) { fprintf(stderr, "%s\n", m.c_str()); } }
int main() {
    const int B = 256, L = 120000;
    float *x, *out, *w, *b;
    hipMalloc(&x, (size_t)B * L * 64 * 4); hipMalloc(&out, (size_t)B * (L / 4) * 128 * 4); hipMalloc(&w, 128 * 512 * 4); hipMalloc(&b, 512);
    hipMemset(x, 0, (size_t)B * L * 64 * 4); hipMemset(w, 0, 128 * 512 * 4); hipMemset(b, 0, 512);
    at::Down64Args a; a.x = x; a.out = out; a.w = w; a.b = b; a.B = B; a.L = L;
    at::launch_seanet_down64x3(a, 0); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r) at::launch_seanet_down64x3(a, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%.3f ms\n", ms / 3);
    return 0;
}
