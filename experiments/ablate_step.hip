// Where does k_lif_step_c32 spend its time?  (diagnostic, not product)   ./ablate_step
#include "../snn_modulation_classification_amd/csrc/dcll_hip.hip"
#include <vector>
#include <algorithm>
template <int DBG>
static float run(int B, float *x, float *W, float *bias, float *tau, float *e0, float *e1, float *arp, float *s, float *pv, float *v)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_lif_step_c32<true, DBG>), dim3(B), dim3(256), 0, 0, x, dcll_wsrc{W, nullptr, nullptr}, bias, tau, tau + 8192, tau + 2 * 8192, tau + 3 * 8192, 1, e0, e1, arp, s, pv, v, 0.65f, 1.0f);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    return best * 1e3f;
}
// per-wave shader-clock stamps (DBG & 32): phases of the kernel, averaged over all waves
template <int DBG>
static void stamps(int B, float *x, float *W, float *bias, float *tau, float *e0, float *e1, float *arp, float *s, float *pv, float *v)
{
    run<DBG | 32>(B, x, W, bias, tau, e0, e1, arp, s, pv, v);
    std::vector<unsigned long long> h((size_t)B * 24);
    hipMemcpy(h.data(), arp + (size_t)B * 8192, h.size() * 8, hipMemcpyDeviceToHost);
    double pro = 0, loop = 0, epi = 0, clk = 0; unsigned long long t0 = ~0ull, t1 = 0;
    for (int i = 0; i < B * 4; ++i) {
        pro += h[6 * i + 1] - h[6 * i]; loop += h[6 * i + 2] - h[6 * i + 1]; epi += h[6 * i + 3] - h[6 * i + 2];
        clk += (double)(h[6 * i + 3] - h[6 * i]) / (double)(h[6 * i + 5] - h[6 * i + 4]) * 0.1;     // GHz
        if (h[6 * i + 4] < t0) t0 = h[6 * i + 4];
        if (h[6 * i + 5] > t1) t1 = h[6 * i + 5];
    }
    {   // spread over the launch: when do waves start / end, how long do they take (100 MHz constant clock)
        std::vector<double> st, du, en;
        for (int i = 0; i < B * 4; ++i) { st.push_back((h[6 * i + 4] - t0) * 0.01); du.push_back((h[6 * i + 5] - h[6 * i + 4]) * 0.01); en.push_back((h[6 * i + 5] - t0) * 0.01); }
        auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
        printf("   wave start (us after the first): p0 %.1f p10 %.1f p50 %.1f p90 %.1f p100 %.1f | duration: p0 %.1f p10 %.1f p50 %.1f p90 %.1f p100 %.1f | end: p0 %.1f p10 %.1f p50 %.1f p90 %.1f p100 %.1f\n",
               pct(st, 0), pct(st, .1), pct(st, .5), pct(st, .9), pct(st, 1), pct(du, 0), pct(du, .1), pct(du, .5), pct(du, .9), pct(du, 1), pct(en, 0), pct(en, .1), pct(en, .5), pct(en, .9), pct(en, 1));
        double s_lo = 0, s_hi = 0, d_lo = 0, d_hi = 0;
        for (int i = 0; i < B * 4; ++i) { if (i < B * 2) { s_lo += st[i]; d_lo += du[i]; } else { s_hi += st[i]; d_hi += du[i]; } }
        printf("   workgroups 0..B/2-1: mean start %.1f duration %.1f | B/2..B-1: mean start %.1f duration %.1f\n", s_lo / (B * 2), d_lo / (B * 2), s_hi / (B * 2), d_hi / (B * 2));
    }
    printf("   stamps DBG=%d B=%d: prologue %.0f  chunk loop %.0f (MFMA time of a wave: %d, of a SIMD's %d waves: %d)  epilogue %.0f  shader clock %.3f GHz  first entry -> last exit %.1f us\n",
           DBG, B, pro / (B * 4), loop / (B * 4), 1568 * 64, B >= 512 ? 2 : 1, (B >= 512 ? 2 : 1) * 1568 * 64, epi / (B * 4), clk / (B * 4), (t1 - t0) * 0.01);
}
int main()
{
    const int BM = 4096;
    size_t ns = (size_t)BM * 8192;
    float *x, *W, *bias, *tau, *e0, *e1, *arp, *s, *pv, *v;
    hipMalloc(&x, ns * 4); hipMalloc(&e0, ns * 4); hipMalloc(&e1, ns * 4); hipMalloc(&arp, ns * 4);
    hipMalloc(&s, ns * 4); hipMalloc(&pv, ns * 4); hipMalloc(&v, ns * 4);
    hipMemset(x, 0, ns * 4); hipMemset(e0, 0, ns * 4); hipMemset(e1, 0, ns * 4); hipMemset(arp, 0, ns * 4);
    hipMalloc(&W, 32 * 32 * 49 * 4); hipMalloc(&bias, 128); hipMalloc(&tau, 4 * 8192 * 4);
    std::vector<float> hw(32 * 32 * 49, 1e-6f), hb(32, 1e-4f), ht(4 * 8192, 0.9f);
    hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(bias, hb.data(), 128, hipMemcpyHostToDevice); hipMemcpy(tau, ht.data(), ht.size() * 4, hipMemcpyHostToDevice);
    printf("%6s %10s %10s %10s %10s %10s %10s %10s %10s  (us; ideal MFMA time at 157.3 TF)\n", "B", "full", "noMFMA", "noStores", "noState", "MFMAonly", "+noWstream", "+noBarrier", "ideal");
    for (int B : {64, 256, 512, 1024, 2048, 4096}) {
        float f = run<0>(B, x, W, bias, tau, e0, e1, arp, s, pv, v);
        float a = run<1>(B, x, W, bias, tau, e0, e1, arp, s, pv, v);
        float c = run<2>(B, x, W, bias, tau, e0, e1, arp, s, pv, v);
        float d = run<4>(B, x, W, bias, tau, e0, e1, arp, s, pv, v);
        float e = run<6>(B, x, W, bias, tau, e0, e1, arp, s, pv, v);
        float g = run<14>(B, x, W, bias, tau, e0, e1, arp, s, pv, v);
        float h = run<30>(B, x, W, bias, tau, e0, e1, arp, s, pv, v);
        printf("%6d %10.1f %10.1f %10.1f %10.1f %10.1f %10.1f %10.1f %10.1f\n", B, f, a, c, d, e, g, h, 2.0 * 32 * 1568 * 256 * (double)B / 157.3e12 * 1e6);
    }
    for (int B : {512, 4096})
        printf("B=%d: full %.1f  no weight streaming %.1f  fetch only (no LDS copy-in) %.1f  copy-in only (no fetch) %.1f us\n", B,
               run<0>(B, x, W, bias, tau, e0, e1, arp, s, pv, v), run<8>(B, x, W, bias, tau, e0, e1, arp, s, pv, v),
               run<64>(B, x, W, bias, tau, e0, e1, arp, s, pv, v), run<128>(B, x, W, bias, tau, e0, e1, arp, s, pv, v));
    for (int B : {256, 512}) {
        stamps<0>(B, x, W, bias, tau, e0, e1, arp, s, pv, v);
        stamps<30>(B, x, W, bias, tau, e0, e1, arp, s, pv, v);
    }
    return 0;
}
