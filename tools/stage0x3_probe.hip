// seanet_stage0x3_kernel against seanet_stage0_kernel on random data: max difference, and the kernel times on the bench shape.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on tools/stage0x3_probe.hip -o tools/probe_stage0x3
#include "../audiotoken_amd/csrc/seanet_stage0.hip"
#include "../audiotoken_amd/csrc/seanet_stage0x3.hip"
#include <cstdio>
#include <vector>
#include <cmath>
namespace at { void set_error(const std::string& m) { fprintf(stderr, "%s\n", m.c_str()); } }
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 2, N = argc > 2 ? atoi(argv[2]) : 2000;
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((s >> 11) * (1.0 / 9007199254740992.0)) * 2.f - 1.f; };
    std::vector<float> hw((size_t)B * N), w0(32 * 7), b0(32), w3(16 * 96), b3(16), wt(32 * 48), bt(32), wd(64 * 128), bd(64);
    for (auto& v : hw) v = rnd();
    for (auto& v : w0) v = rnd() * 0.5f;
    for (auto& v : w3) v = rnd() * 0.15f;
    for (auto& v : wt) v = rnd() * 0.2f;
    for (auto& v : wd) v = rnd() * 0.12f;
    for (auto* p : {&b0, &b3, &bt, &bd}) for (auto& v : *p) v = rnd() * 0.1f;
    auto up = [&](const std::vector<float>& h) { float* d; hipMalloc(&d, h.size() * 4); hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice); return d; };
    at::Stage0Args a;
    a.wav = up(hw); a.w0 = up(w0); a.b0 = up(b0); a.w3 = up(w3); a.b3 = up(b3); a.wt = up(wt); a.bt = up(bt); a.wd = up(wd); a.bd = up(bd);
    a.B = B; a.N = N;
    // the shortcut folded into conv0 (what encodec.hip's finalize computes): wsc0 = Wsc . W0, bsc0 = Wsc . b0 + bt
    std::vector<float> wsc0(32 * 7), bsc0(32);
    for (int c = 0; c < 32; ++c) {
        for (int j = 0; j < 7; ++j) { double acc = 0; for (int k = 0; k < 32; ++k) acc += (double)wt[c * 48 + 16 + k] * w0[k * 7 + j]; wsc0[c * 7 + j] = (float)acc; }
        double accb = bt[c]; for (int k = 0; k < 32; ++k) accb += (double)wt[c * 48 + 16 + k] * b0[k]; bsc0[c] = (float)accb;
    }
    a.wsc0 = up(wsc0); a.bsc0 = up(bsc0);
    const size_t no = (size_t)B * (N / 2) * 64;
    float *o0, *o1; hipMalloc(&o0, no * 4); hipMalloc(&o1, no * 4);
    hipMemset(o0, 0xff, no * 4); hipMemset(o1, 0xff, no * 4);
    a.x1 = o0; at::launch_seanet_stage0(a, 0);
    a.x1 = o1; at::launch_seanet_stage0x3(a, 0);
    hipDeviceSynchronize();
    std::vector<float> h0(no), h1(no);
    hipMemcpy(h0.data(), o0, no * 4, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), o1, no * 4, hipMemcpyDeviceToHost);
    double md = 0, mr = 0, sum = 0; long long bad = 0, first = -1, c6 = 0;
    for (size_t i = 0; i < no; ++i) {
        const double d = fabs((double)h0[i] - h1[i]);
        if (!(d <= 1e-4)) { ++bad; if (first < 0) first = (long long)i; }
        if (d > md) md = d;
        sum += d; c6 += d > 1e-6;
        if (fabs(h0[i]) > mr) mr = fabs(h0[i]);
    }
    printf("B %d N %d: mean |diff| %.3e max %.3e (max |ref| %.3f), > 1e-6: %lld, off by > 1e-4 (or NaN): %lld of %zu", B, N, sum / no, md, mr, c6, bad, no);
    if (first >= 0) printf(", first at clip %lld row %lld ch %lld: %g vs %g", first / ((long long)(N / 2) * 64), first / 64 % (N / 2), first % 64, h0[first], h1[first]);
    printf("\n");
    if (argc > 3) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int v = 0; v < 2; ++v) {
            hipEventRecord(e0, 0);
            for (int r = 0; r < 3; ++r) v ? at::launch_seanet_stage0x3(a, 0) : at::launch_seanet_stage0(a, 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s %.3f ms\n", v ? "x3" : "fp32", ms / 3);
        }
    }
    return 0;
}
