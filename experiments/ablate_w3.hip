// Launch time of k_lif_seq_w3<64> alone (diagnostic, not product)   ./ablate_w3 [B] [W]
// (Round 3: a build with s_memtime stamps per wave and phase — trace / chain A / epilogue A / chain B / epilogue B /
//  barrier — showed where a step went; the numbers are in DESIGN.md 4.1c.  Build with -DW3_EPG=n / -fno-slp-vectorize to
//  compare code-generation variants.)
#include "../snn_modulation_classification_amd/csrc/dcll_seq_w3.hip"
#include <vector>
#include <algorithm>
#include <stdlib.h>

char *dcll_err_buf(void) { static char b[512]; return b; }
void dcll_trace_note(const char *) {}

static float run(int B, int Wd, int T, const uint32_t *spk_in, const float *W, const float *bias, const float *tau4, float *e0,
                 float *e1, float *arp, uint32_t *spk_out, float *pv, int reps)
{
    const int HW = 16 * Wd, NTS = HW / 32;
    int logW = 0;
    while ((1 << logW) < Wd) ++logW;
    const long nwg = ((long)B * NTS + 7) / 8;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a);
        if (Wd >= 32)
            hipLaunchKernelGGL((k_lif_seq_w3<64, true, 5, 5, true>), dim3(nwg), dim3(512), 0, 0, spk_in,
                               dcll_wsrc{W, nullptr, nullptr}, bias, tau4, e0, e1, arp, spk_out, pv, (float *)nullptr, T, B, HW, logW, 0.65f, 1.0f);
        else {
#define W3_NARROW(LW_) hipLaunchKernelGGL((k_lif_seq_w3<64, true, 5, LW_, true>), dim3(nwg), dim3(512), 0, 0, spk_in, \
                               dcll_wsrc{W, nullptr, nullptr}, bias, tau4, e0, e1, arp, spk_out, pv, (float *)nullptr, T, B, HW, logW, 0.65f, 1.0f)
            if (logW == 4) W3_NARROW(4); else if (logW == 3) W3_NARROW(3); else if (logW == 2) W3_NARROW(2); else W3_NARROW(1);
#undef W3_NARROW
        }
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}

static void print_stamps(int T)
{
#ifdef W3_STAMPS
    // phases of workgroup 0, summed over the launches of run(): cycles per step and wave (s_memtime ticks at 100 MHz)
    unsigned long long st[8][8];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(w3_stamps), sizeof(st));
    const char *nm[8] = {"chain A", "epilogue A", "chain B", "epilogue B", "trace + spike words", "barrier", "barrier (narrow, 1st)", ""};
    for (int w = 0; w < 8; ++w) {
        printf("wave %d:", w);
        for (int p = 0; p < 7; ++p) printf("  %s %.0f", nm[p], st[w][p] / (4.0 * T));
        printf("   [memtime ticks per step]\n");
    }
#endif
}

// ./ablate_w3 B W first: the FIRST layer (c_in = 1, input = one cell index per sample and step; pooled pre-sigmoid map out)
static int run_first(int B, int Wd, int T)
{
    const int HW = 16 * Wd, NTS = HW / 32;
    int logW = 0;
    while ((1 << logW) < Wd) ++logW;
    std::vector<int32_t> hc((size_t)T * B);
    srand(1);
    for (auto &x : hc) x = rand() % HW;
    std::vector<float> hw(64 * 3), hb(64), ht(4, 0.9f);
    for (auto &x : hw) x = (rand() / (float)RAND_MAX - 0.5f) * 1e-3f;
    for (auto &x : hb) x = (rand() / (float)RAND_MAX - 0.5f) * 1e-3f;
    ht[1] = 20.f; ht[3] = 6.7f;
    int32_t *cells; uint32_t *spk_out; float *W, *bias, *tau4, *e0, *e1, *arp, *pv;
    const long nin = (long)B * HW, nout = (long)B * 64 * HW;
    hipMalloc(&cells, hc.size() * 4); hipMalloc(&spk_out, (long)T * B * 64 * (HW / 64) * 4 + 64);
    hipMalloc(&W, 1024); hipMalloc(&bias, 256); hipMalloc(&tau4, 64);
    hipMalloc(&e0, nin * 4); hipMalloc(&e1, nin * 4); hipMalloc(&arp, nout * 4);
    hipMalloc(&pv, (long)T * B * 64 * (HW / 2) * 4);
    hipMemcpy(cells, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(bias, hb.data(), 256, hipMemcpyHostToDevice); hipMemcpy(tau4, ht.data(), 16, hipMemcpyHostToDevice);
    hipMemset(e0, 0, nin * 4); hipMemset(e1, 0, nin * 4); hipMemset(arp, 0, nout * 4);
    const long nitems = (long)B * (HW / 128) * (64 / W3F_NCH), nblk = (nitems + W3F_WPB - 1) / W3F_WPB;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_lif_seq_w3f<true, 5, W3F_NCH>), dim3(nblk), dim3(64 * W3F_WPB), 0, 0, cells,
                           dcll_wsrc{W, nullptr, nullptr}, bias, tau4, e0, e1, arp, spk_out, pv, (float *)nullptr, T, B, HW, logW, nitems, 0.65f, 1.0f);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    const double bytes = (double)T * B * 64 * (HW / 2) * 4 * (1 + 1 / 32.0);
    printf("k_lif_seq_w3f (first layer) B=%d W=%d T=%d: %.2f ms = %.2f TB/s of pooled map + spikes\n", B, Wd, T, best, bytes / best / 1e9);
    print_stamps(T);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int main(int argc, char **argv)
{
    if (argc > 3) return run_first(atoi(argv[1]), atoi(argv[2]), 128);
    const int B = argc > 1 ? atoi(argv[1]) : 1024, Wd = argc > 2 ? atoi(argv[2]) : 64, T = 128;
    const long HW = 16L * Wd, nin = (long)B * 64 * HW, nsp = (long)T * B * 64 * (HW / 32);
    std::vector<uint32_t> hs(nsp);
    srand(1);
    // 8 % spike density; a 1 Mi-word pattern repeated (rand() per bit over 1e9 words took minutes of host time at B = 4096)
    const size_t npat = std::min<size_t>(hs.size(), 1u << 20);
    for (size_t k = 0; k < npat; ++k) { uint32_t v = 0; for (int i = 0; i < 32; ++i) v |= (uint32_t)((rand() % 100) < 8) << i; hs[k] = v; }
    for (size_t k = npat; k < hs.size(); ++k) hs[k] = hs[k - npat];
    std::vector<float> hw(64 * 64 * 3), hb(64), ht(4 * 64);
    for (auto &x : hw) x = (rand() / (float)RAND_MAX - 0.5f) * 1e-5f;
    for (auto &x : hb) x = (rand() / (float)RAND_MAX - 0.5f) * 1e-3f;
    for (int c = 0; c < 64; ++c) { ht[c] = 0.95f; ht[64 + c] = 20.f; ht[128 + c] = 0.85f; ht[192 + c] = 6.7f; }
    uint32_t *spk_in, *spk_out; float *W, *bias, *tau4, *e0, *e1, *arp, *pv;
    hipMalloc(&spk_in, nsp * 4); hipMalloc(&spk_out, (long)T * B * 64 * (HW / 64) * 4 + 64);
    hipMalloc(&W, hw.size() * 4); hipMalloc(&bias, 256); hipMalloc(&tau4, 1024);
    hipMalloc(&e0, nin * 4); hipMalloc(&e1, nin * 4); hipMalloc(&arp, nin * 4);
    hipMalloc(&pv, (long)T * B * 64 * (HW / 2) * 4);
    hipMemcpy(spk_in, hs.data(), nsp * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(bias, hb.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(tau4, ht.data(), 1024, hipMemcpyHostToDevice);
    hipMemset(e0, 0, nin * 4); hipMemset(e1, 0, nin * 4); hipMemset(arp, 0, nin * 4);
    const double ideal = 2.0 * 64 * 192 * HW * (double)T * B / 157.3e12 * 1e3;
    const float ms = run(B, Wd, T, spk_in, W, bias, tau4, e0, e1, arp, spk_out, pv, 4);
    printf("k_lif_seq_w3<64> B=%d W=%d T=%d: %.2f ms = %.1f %% of the fp32-MFMA peak (ideal at 157.3 TF: %.2f ms)\n", B, Wd, T, ms,
           100.0 * ideal / ms, ideal);
    print_stamps(T);
    return 0;
}
