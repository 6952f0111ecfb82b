// Ablation timing of k_lif_seq_c32 (diagnostic, not product): which part of a stage costs what.
#include "../snn_modulation_classification_amd/csrc/dcll_hip.hip"
#include <vector>
template <int AB, int PRIO = 5, int BASES = 1>
static float run(int B, int T, bool want_pv)
{
    size_t nin = (size_t)T * B * 32 * 8;
    uint32_t *spk_in, *spk_out; float *W, *bias, *tau4, *e0, *e1, *arp, *pv;
    hipMalloc(&spk_in, nin * 4); hipMalloc(&spk_out, nin * 4);
    hipMemset(spk_in, 0x11, nin * 4);
    hipMalloc(&W, 32 * 32 * 49 * 4); hipMalloc(&bias, 128); hipMalloc(&tau4, 512);
    std::vector<float> hw(32 * 32 * 49, 1e-6f), hb(32, 1e-4f), ht(128, 0.9f);
    hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(bias, hb.data(), 128, hipMemcpyHostToDevice);
    hipMemcpy(tau4, ht.data(), 512, hipMemcpyHostToDevice);
    size_t ns = (size_t)B * 32 * 256;
    hipMalloc(&e0, ns * 4); hipMalloc(&e1, ns * 4); hipMalloc(&arp, ns * 4);
    hipMemset(e0, 0, ns * 4); hipMemset(e1, 0, ns * 4); hipMemset(arp, 0, ns * 4);
    hipMalloc(&pv, (size_t)T * ns * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        if (want_pv) hipLaunchKernelGGL((k_lif_seq_c32<true, 1, 0, AB, PRIO, BASES>), dim3(B), dim3(512), 0, 0, spk_in, dcll_wsrc{W, nullptr, nullptr}, bias, tau4, e0, e1, arp, spk_out, pv, (float *)nullptr, (const float *)nullptr, (const float *)nullptr, (float *)nullptr, T, B, 0.65f, 1.0f);
        else hipLaunchKernelGGL((k_lif_seq_c32<true, 0, 0, AB, PRIO, BASES>), dim3(B), dim3(512), 0, 0, spk_in, dcll_wsrc{W, nullptr, nullptr}, bias, tau4, e0, e1, arp, spk_out, (float *)nullptr, (float *)nullptr, (const float *)nullptr, (const float *)nullptr, (float *)nullptr, T, B, 0.65f, 1.0f);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    hipFree(spk_in); hipFree(spk_out); hipFree(W); hipFree(bias); hipFree(tau4); hipFree(e0); hipFree(e1); hipFree(arp); hipFree(pv);
    return best;
}
int main()
{
    const int B = 1024, T = 128;
    double ideal = 2.0 * 32 * 1568 * 256 * (double)T * B / 157.3e12 * 1e3;
    printf("ideal at 157.3 TF: %.2f ms\n", ideal);
    printf("full                         %.2f ms\n", run<0>(B, T, true));
    printf("full, 2 hidden bases         %.2f ms\n", run<0, 5, 2>(B, T, true));
    printf("full, prio 0                 %.2f ms\n", run<0, 0>(B, T, true));
    printf("full (again)                 %.2f ms\n", run<0>(B, T, true));
    printf("full, no pv store            %.2f ms\n", run<0>(B, T, false));
    printf("no epilogue                  %.2f ms\n", run<1>(B, T, true));
    printf("no trace update              %.2f ms\n", run<2>(B, T, true));
    printf("no epilogue, no trace        %.2f ms\n", run<3>(B, T, true));
    printf("no hand-off                  %.2f ms\n", run<4>(B, T, true));
    printf("no epi/trace/hand-off        %.2f ms\n", run<7>(B, T, true));
    {   // per-wave time shares of workgroup 0 (diagnostic stamps, ABLATE bit 3)
        size_t nin = (size_t)T * B * 32 * 8;
        uint32_t *spk_in, *spk_out; float *W, *bias, *tau4, *e0, *e1, *arp, *pv; unsigned long long *dbg;
        hipMalloc(&spk_in, nin * 4); hipMalloc(&spk_out, nin * 4); hipMemset(spk_in, 0x11, nin * 4);
        hipMalloc(&W, 32 * 32 * 49 * 4); hipMalloc(&bias, 128); hipMalloc(&tau4, 512);
        std::vector<float> hw(32 * 32 * 49, 1e-6f), hb(32, 1e-4f), ht(128, 0.9f);
        hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(bias, hb.data(), 128, hipMemcpyHostToDevice); hipMemcpy(tau4, ht.data(), 512, hipMemcpyHostToDevice);
        size_t ns = (size_t)B * 32 * 256;
        hipMalloc(&e0, ns * 4); hipMalloc(&e1, ns * 4); hipMalloc(&arp, ns * 4);
        hipMemset(e0, 0, ns * 4); hipMemset(e1, 0, ns * 4); hipMemset(arp, 0, ns * 4);
        hipMalloc(&pv, (size_t)T * ns * 4); hipMalloc(&dbg, 4096); hipMemset(dbg, 0, 4096);
        hipLaunchKernelGGL((k_lif_seq_c32<true, 1, 0, 8>), dim3(B), dim3(512), 0, 0, spk_in, dcll_wsrc{W, nullptr, nullptr}, bias, tau4, e0, e1, arp, spk_out, pv, (float *)dbg, (const float *)nullptr, (const float *)nullptr, (float *)nullptr, T, B, 0.65f, 1.0f);
        unsigned long long h[64];
        hipMemcpy(h, dbg, 512, hipMemcpyDeviceToHost);
        printf("wave: total Mcyc | barrier-wait %% | epilogue %% | trace %% | chain+handoff %%   (s_memtime ticks, 100 MHz?)\n");
        for (int w = 0; w < 8; ++w)
            printf("  w%d: %8.3f | %5.1f | %5.1f | %5.1f | %5.1f\n", w, h[w * 8] / 1e6, 100.0 * h[w * 8 + 1] / h[w * 8],
                   100.0 * h[w * 8 + 2] / h[w * 8], 100.0 * h[w * 8 + 3] / h[w * 8], 100.0 * h[w * 8 + 4] / h[w * 8]);
    }
    return 0;
}
