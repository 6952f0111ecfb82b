// Stage-phase timing of k_lif_seq_c32d (diagnostic, not product).
//   ./ablate_c32d            the kernel alone
//   ./ablate_c32d co         with the co-resident readout (k_readout_direct, <= 64 VGPRs, no LDS) running on a second,
//                            lower-priority stream over the previous pv buffer — what test_sequence(overlap_readout=True)
//                            does: how much does a stage of the layer kernel stretch?
//   ./ablate_c32d presig     pv_presigmoid (round 3): the epilogue stores v instead of sigmoid(v) — launch time and the
//                            stamps of the non-MFMA phase with and without the four sigmoids per wave and stage
#include "../snn_modulation_classification_amd/csrc/dcll_hip.hip"
#include <vector>
int main(int argc, char **argv)
{
    const bool presig = argc > 1 && argv[1][0] == 'p';
    const bool co = argc > 1 && !presig;
    const int B = 1024, T = 128;
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
    float *dbg_v = nullptr;                     // a full-size v output whose head doubles as the stamp area (presig mode)
    if (presig) hipMalloc(&dbg_v, (size_t)T * ns * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    // co-resident readout: 24 rows over a second pv buffer (T*B rows of 8192), launched back to back on its own stream so
    // that it runs for the whole duration of the layer kernel
    float *pv2 = nullptr, *Wro = nullptr, *ro = nullptr;
    hipStream_t side = nullptr;
    int lo, hi;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStream_t hot;
    hipStreamCreateWithPriority(&hot, hipStreamNonBlocking, hi);
    if (co) {
        hipMalloc(&pv2, (size_t)T * ns * 4); hipMemset(pv2, 0, (size_t)T * ns * 4);
        hipMalloc(&Wro, 24 * 8192 * 4); hipMemset(Wro, 0, 24 * 8192 * 4); hipMalloc(&ro, (size_t)T * B * 24 * 4);
        hipStreamCreateWithPriority(&side, hipStreamNonBlocking, lo);
    }
    auto launch_side = [&](int n) {
        for (int i = 0; i < n; ++i) dcll_launch_readout_direct(pv2, Wro, nullptr, ro, (long)T * B, 8192, 24, side);
    };
    if (presig) {
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(a);
                if (mode == 0)
                    hipLaunchKernelGGL((k_lif_seq_c32d<true, 1, 0>), dim3(B), dim3(512), 0, 0, spk_in, dcll_wsrc{W, nullptr, nullptr}, bias, tau4, e0, e1, arp, spk_out, pv, (float *)nullptr, T, B, 0.65f, 1.0f);
                else
                    hipLaunchKernelGGL((k_lif_seq_c32d<true, 2, 0>), dim3(B), dim3(512), 0, 0, spk_in, dcll_wsrc{W, nullptr, nullptr}, bias, tau4, e0, e1, arp, spk_out, (float *)nullptr, pv, T, B, 0.65f, 1.0f);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            printf("k_lif_seq_c32d B=%d T=%d, buffer = %s: %.3f ms (best of 4)\n", B, T, mode ? "v (pv_presigmoid)" : "sigmoid(v)", best);
            // stamps: the debug area is the head of the v output (OUT = 3 / 2: both variants also write v there first)
            if (mode == 0)
                hipLaunchKernelGGL((k_lif_seq_c32d<true, 3, 1>), dim3(B), dim3(512), 0, 0, spk_in, dcll_wsrc{W, nullptr, nullptr}, bias, tau4, e0, e1, arp, spk_out, pv, (float *)dbg_v, T, B, 0.65f, 1.0f);
            else
                hipLaunchKernelGGL((k_lif_seq_c32d<true, 2, 1>), dim3(B), dim3(512), 0, 0, spk_in, dcll_wsrc{W, nullptr, nullptr}, bias, tau4, e0, e1, arp, spk_out, (float *)nullptr, (float *)dbg_v, T, B, 0.65f, 1.0f);
            hipDeviceSynchronize();
            unsigned long long h3[64];
            hipMemcpy(h3, dbg_v, 512, hipMemcpyDeviceToHost);
            const double nst3 = 4.0 * T + 8;
            printf("  cycles per stage | non-MFMA phase | barrier 2 | chains + slot write | barrier 1%s\n", mode ? "" : "   (this variant also stores v: OUT = 3)");
            for (int w = 0; w < 8; ++w)
                printf("  w%d: %8.0f | %7.0f | %7.0f | %7.0f | %7.0f\n", w, h3[w * 8] / nst3, h3[w * 8 + 1] / nst3, h3[w * 8 + 2] / nst3,
                       h3[w * 8 + 3] / nst3, h3[w * 8 + 4] / nst3);
        }
        return 0;
    }
    for (int rep = 0; rep < 3 && !co; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_lif_seq_c32d<true, 1, 0>), dim3(B), dim3(512), 0, 0, spk_in, dcll_wsrc{W, nullptr, nullptr}, bias, tau4, e0, e1, arp, spk_out, pv, (float *)nullptr, T, B, 0.65f, 1.0f);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("k_lif_seq_c32d B=%d T=%d: %.2f ms (ideal at 157.3 TF: %.2f)\n", B, T, ms, 2.0 * 32 * 1568 * 256 * (double)T * B / 157.3e12 * 1e3);
    }
    if (!co) {
        // what do the LDS bank conflicts of the B-fragment reads cost?  Same kernel with the tile's second image row read
        // 16 instead of 19 floats behind the first (conflict free, WRONG data — timing only), same stamps.
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL((k_lif_seq_c32d<true, 1, 2>), dim3(B), dim3(512), 0, 0, spk_in, dcll_wsrc{W, nullptr, nullptr}, bias, tau4, e0, e1, arp, spk_out, pv, (float *)nullptr, T, B, 0.65f, 1.0f);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("k_lif_seq_c32d, conflict-free B reads (wrong data): %.2f ms\n", ms);
        }
        hipLaunchKernelGGL((k_lif_seq_c32d<true, 1, 3>), dim3(B), dim3(512), 0, 0, spk_in, dcll_wsrc{W, nullptr, nullptr}, bias, tau4, e0, e1, arp, spk_out, pv, (float *)dbg, T, B, 0.65f, 1.0f);
        unsigned long long h2[64];
        hipMemcpy(h2, dbg, 512, hipMemcpyDeviceToHost);
        const double nst2 = 4.0 * T + 8;
        printf("conflict-free variant: cycles per stage | non-MFMA phase | barrier 2 | chains + slot write | barrier 1\n");
        for (int w = 0; w < 8; ++w)
            printf("  w%d: %8.0f | %7.0f | %7.0f | %7.0f | %7.0f\n", w, h2[w * 8] / nst2, h2[w * 8 + 1] / nst2, h2[w * 8 + 2] / nst2,
                   h2[w * 8 + 3] / nst2, h2[w * 8 + 4] / nst2);
        hipMemset(dbg, 0, 4096);
    }
    if (co) {
        hipDeviceSynchronize();
        launch_side(40);                            // far longer than the layer kernel
        hipEventRecord(a, hot);
    }
    hipLaunchKernelGGL((k_lif_seq_c32d<true, 1, 1>), dim3(B), dim3(512), 0, hot, spk_in, dcll_wsrc{W, nullptr, nullptr}, bias, tau4, e0, e1, arp, spk_out, pv, (float *)dbg, T, B, 0.65f, 1.0f);
    if (co) {
        hipEventRecord(b, hot); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("k_lif_seq_c32d B=%d T=%d with the co-resident readout running: %.2f ms\n", B, T, ms);
    }
    hipDeviceSynchronize();
    unsigned long long h[64];
    hipMemcpy(h, dbg, 512, hipMemcpyDeviceToHost);
    const double nst = 4.0 * T + 8;
    printf("wave: cycles per stage | non-MFMA phase | barrier 2 | chains + slot write | barrier 1   (s_memtime ticks per stage)\n");
    for (int w = 0; w < 8; ++w)
        printf("  w%d: %8.0f | %7.0f | %7.0f | %7.0f | %7.0f\n", w, h[w * 8] / nst, h[w * 8 + 1] / nst, h[w * 8 + 2] / nst,
               h[w * 8 + 3] / nst, h[w * 8 + 4] / nst);
    return 0;
}
