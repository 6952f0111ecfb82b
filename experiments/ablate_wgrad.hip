// Where does k_bwd_wgrad_c32 (weight gradient of the 32 -> 32 layers, the other half of a learning timestep) spend its
// time?  (diagnostic, not product)   ./ablate_wgrad
#include "../snn_modulation_classification_amd/csrc/dcll_hip.hip"
#include <vector>
template <int DBG>
static float run(int B, int nwg, float *g, float *e1, float *part)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_bwd_wgrad_c32<ROWF, CHF, false, 1, DBG>), dim3(nwg), dim3(512), 0, 0, g, e1, part, B, 16, 16);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    return best * 1e3f;
}
template <int DBG>
static void stamps(int B, int nwg, float *g, float *e1, float *part)
{
    run<DBG | 1>(B, nwg, g, e1, part);
    std::vector<unsigned long long> h((size_t)nwg * 8 * 6);
    hipMemcpy(h.data(), part + (size_t)nwg * 32 * 1569, h.size() * 8, hipMemcpyDeviceToHost);
    double st = 0, bi = 0, mf = 0, en = 0, tot = 0; unsigned long long t0 = ~0ull, t1 = 0;
    for (int i = 0; i < nwg * 8; ++i) {
        st += h[6 * i]; bi += h[6 * i + 1]; mf += h[6 * i + 2]; en += h[6 * i + 3]; tot += h[6 * i + 4];
        if (h[6 * i + 5] < t0) t0 = h[6 * i + 5];
        if (h[6 * i + 5] + h[6 * i + 4] > t1) t1 = h[6 * i + 5] + h[6 * i + 4];
    }
    const int n = nwg * 8, jobs = (B + nwg - 1) / nwg;
    printf("   stamps DBG=%d B=%d (%d jobs per workgroup), cycles per wave: staging + barriers %.0f  bias sum %.0f  MFMA loop %.0f "
           "(MFMA time of a SIMD's 2 waves: %d)  stores + tile 48 %.0f  total %.0f; first entry -> last exit %llu\n",
           DBG, B, jobs, st / n, bi / n, mf / n, 2 * jobs * 784 * 64, en / n, tot / n, t1 - t0);
}
int main()
{
    const int BM = 4096, NWG = 256;
    size_t ns = (size_t)BM * 8192;
    float *g, *e1, *part;
    hipMalloc(&g, ns * 4); hipMalloc(&e1, ns * 4); hipMalloc(&part, ((size_t)NWG * 32 * 1569 + NWG * 8 * 12) * 4);
    hipMemset(g, 0, ns * 4); hipMemset(e1, 0, ns * 4);
    printf("%6s %10s %10s %10s %10s  (us; ideal MFMA time at 157.3 TF)\n", "B", "full", "noMFMA", "noStores", "ideal");
    for (int B : {256, 512, 1024, 4096}) {
        float f = run<0>(B, NWG, g, e1, part), a = run<2>(B, NWG, g, e1, part), c = run<4>(B, NWG, g, e1, part);
        printf("%6d %10.1f %10.1f %10.1f %10.1f\n", B, f, a, c, 2.0 * 32 * 1568 * 256 * (double)B / 157.3e12 * 1e6);
    }
    for (int B : {512, 4096}) {
        stamps<0>(B, NWG, g, e1, part);
        stamps<4>(B, NWG, g, e1, part);
    }
    return 0;
}
