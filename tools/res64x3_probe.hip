// seanet_res64x3_kernel against seanet_res64_kernel on random data: max difference, and the kernel time on the bench shape.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on tools/res128x3_probe.hip -o tools/probe_res128x3
#include "../audiotoken_amd/csrc/seanet_res64.hip"
#include "../audiotoken_amd/csrc/seanet_res64x3.hip"
#include <cstdio>
#include <vector>
#include <cmath>
namespace at { void set_error(const std::string& m) { fprintf(stderr, "%s\n", m.c_str()); } }
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 2, L = argc > 2 ? atoi(argv[2]) : 200;
    std::vector<float> hx((size_t)B * L * 64), hw3(32 * 192), hb3(32), hwt(64 * 96), hbt(64);
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((s >> 11) * (1.0 / 9007199254740992.0)) * 2.f - 1.f; };
    for (auto& v : hx) v = rnd() * 2.f;
    for (auto& v : hw3) v = rnd() * 0.05f;
    for (auto& v : hwt) v = rnd() * 0.07f;
    for (auto& v : hb3) v = rnd() * 0.1f;
    for (auto& v : hbt) v = rnd() * 0.1f;
    float *x, *o0, *o1, *w3, *b3, *wt, *bt;
    hipMalloc(&x, hx.size() * 4); hipMalloc(&o0, hx.size() * 4); hipMalloc(&o1, hx.size() * 4);
    hipMalloc(&w3, hw3.size() * 4); hipMalloc(&b3, 128); hipMalloc(&wt, hwt.size() * 4); hipMalloc(&bt, 256);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(w3, hw3.data(), hw3.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(b3, hb3.data(), 128, hipMemcpyHostToDevice); hipMemcpy(wt, hwt.data(), hwt.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(bt, hbt.data(), 256, hipMemcpyHostToDevice);
    at::Res64Args a; a.x = x; a.w3 = w3; a.b3 = b3; a.wt = wt; a.bt = bt; a.B = B; a.L = L;
    a.out = o0; at::launch_seanet_res64(a, 0);
    a.out = o1; at::launch_seanet_res64x3(a, 0);
    hipDeviceSynchronize();
    std::vector<float> h0(hx.size()), h1(hx.size());
    hipMemcpy(h0.data(), o0, hx.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), o1, hx.size() * 4, hipMemcpyDeviceToHost);
    double md = 0, mr = 0; long long bad = 0, first = -1;
    for (size_t i = 0; i < h0.size(); ++i) {
        const double d = fabs((double)h0[i] - h1[i]);
        if (!(d <= 1e-4)) { ++bad; if (first < 0) first = (long long)i; }
        if (d > md) md = d;
        if (fabs(h0[i]) > mr) mr = fabs(h0[i]);
    }
    long long c6 = 0, c5 = 0, c4 = 0; double sum = 0;
    for (size_t i = 0; i < h0.size(); ++i) { const double d = fabs((double)h0[i] - h1[i]); sum += d; c6 += d > 1e-6; c5 += d > 1e-5; c4 += d > 1e-4; }
    printf("mean |diff| %.3e; > 1e-6: %lld, > 1e-5: %lld, > 1e-4: %lld of %zu\n", sum / h0.size(), c6, c5, c4, h0.size());
    printf("B %d L %d: max |diff| %.3e (max |ref| %.3f), %lld elements off by > 1e-4", B, L, md, mr, bad);
    if (first >= 0) printf(", first at clip %lld row %lld ch %lld: %g vs %g", first / ((long long)L * 64), first / 64 % L, first % 64, h0[first], h1[first]);
    printf("\n");
    if (argc > 3) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int v = 0; v < 2; ++v) {
            hipEventRecord(e0, 0);
            for (int r = 0; r < 3; ++r) v ? at::launch_seanet_res64x3(a, 0) : at::launch_seanet_res64(a, 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s %.3f ms\n", v ? "x3" : "fp32", ms / 3);
        }
    }
    return 0;
}
