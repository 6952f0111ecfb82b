// Does VALU work hide under a dependent fp32-MFMA chain on gfx950?  (diagnostic, not product)   ./mfma_valu_probe
//   same-wave:   every wave runs ONE dependent chain of v_mfma_f32_32x32x2_f32 (one accumulator, as k_lif_seq_w3's
//                chains) with NV independent VALU instructions (v_add_f32 / v_pk_add_f32 on registers of their own) placed
//                between two MFMAs, one wave per SIMD: cycles per MFMA against NV;
//   other-wave:  two waves per SIMD, one runs the bare chain, the other only VALU instructions: how far does each get?
// (the phase stamps of experiments/ablate_w3 -DW3_STAMPS show the second wave of a SIMD standing still — MFMA and VALU —
//  while the first runs its chain; this separates "the chain owns the issue port" from "fp32 MFMAs use the vector ALUs")
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NV, bool PK>
__device__ __forceinline__ void valu_block(float (&v)[8], f32x2 (&p)[4], float c)
{
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (PK) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k & 3]) : "v"(p[(k + 1) & 3]));
        else asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[k & 7]) : "v"(c));
    }
}

// MODE 0: same wave, 1 wave/SIMD (256 threads).  MODE 1: 512 threads, waves 0-3 chain only, waves 4-7 VALU only.
// MODE 2: 512 threads, every wave chain + NV VALU (two chains per SIMD).
template <int NV, bool PK, int MODE>
__global__ __launch_bounds__(512) void k_probe(unsigned long long *out, float *sink, int iters)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    float v[8];
    f32x2 p[4];
    for (int k = 0; k < 8; ++k) v[k] = lane * 1e-3f + k;
    for (int k = 0; k < 4; ++k) p[k] = f32x2{lane * 1e-3f + k, 1.0f};
    const float x = 1.0f + lane * 1e-6f, y = 0.5f, c = 1e-6f;
    const bool chain = MODE != 1 || w < 4, valu = MODE != 1 || w >= 4;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (MODE == 1 && !chain) {
        // VALU-only wave: the same number of VALU instructions per iteration as 32 x max(NV, 1)
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) valu_block<(NV > 0 ? NV : 1), PK>(v, p, c);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0);
                if (valu && MODE != 1) valu_block<NV, PK>(v, p, c);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int k = 0; k < 8; ++k) s += v[k];
    for (int k = 0; k < 4; ++k) s += p[k][0] + p[k][1];
    if (s == 12345.678f) sink[0] = s;
    if (lane == 0) out[blockIdx.x * 8 + w] = t1 - t0;
}

// What does the second wave of a SIMD get done WHILE the first runs its chain?  Waves 0-3: the bare chain, then a flag in
// LDS; waves 4-7: groups of instructions (KIND) until the flag is up, counting them.
//   KIND 0: 1 v_add_f32   1: v_add_f32 + s_nop 0   2: v_add_f32 + 2 s_add_u32   3: v_add_f32 + global_store_dword
//   KIND 4: 1 s_add_u32 only   5: v_cmp + v_cndmask (VALU -> VCC -> VALU)   6: 2 independent v_add_f32
template <int KIND, int NACC = 1>
__global__ __launch_bounds__(512) void k_share(unsigned long long *out, float *sink, float *dump, int iters)
{
    __shared__ volatile int flag[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x < 4) flag[threadIdx.x] = 0;
    f32x16 acc, acc2;
    for (int r = 0; r < 16; ++r) acc[r] = acc2[r] = 0.0f;
    float v[8];
    for (int k = 0; k < 8; ++k) v[k] = lane * 1e-3f + k;
    const float x = 1.0f + lane * 1e-6f, y = 0.5f, c = 1e-6f;
    int sc = 0;
    unsigned long long groups = 0;
    float *dp = dump + (blockIdx.x * 512 + threadIdx.x);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (w < 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                if (NACC % 10 == 2 && (u & 1)) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc2, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0);
            }
        }
        if (lane == 0) { flag[w] = 1; out[blockIdx.x * 8 + w] = __builtin_readcyclecounter() - t0; }
    } else {
        if (NACC >= 10) __builtin_amdgcn_s_setprio(3);      // NACC 11 / 12: as 1 / 2 with the second wave at the highest priority
        while (flag[w - 4] == 0) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[u & 7]) : "v"(c));
                if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1\n\ts_nop 0" : "+v"(v[u & 7]) : "v"(c));
                if (KIND == 2) asm volatile("v_add_f32 %0, %0, %2\n\ts_add_u32 %1, %1, 3\n\ts_add_u32 %1, %1, 5" : "+v"(v[u & 7]), "+s"(sc) : "v"(c));
                if (KIND == 3) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[u & 7]) : "v"(c)); __builtin_nontemporal_store(v[u & 7], dp); }
                if (KIND == 4) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sc));
                if (KIND == 5) asm volatile("v_cmp_lt_f32 vcc, 0, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[u & 7]) : "v"(c) : "vcc");
                if (KIND == 6) asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(v[u & 3]), "+v"(v[4 + (u & 3)]) : "v"(c));
            }
            groups += 16;
        }
    }
    float s = sc;
    for (int r = 0; r < 16; ++r) s += acc[r] + acc2[r];
    for (int k = 0; k < 8; ++k) s += v[k];
    if (s == 12345.678f) sink[0] = s;
    if (lane == 0 && w >= 4) out[blockIdx.x * 8 + w] = groups;
}

template <int KIND, int NACC = 1>
static void run_share(int iters, const char *what)
{
    const int nwg = 256;
    unsigned long long *out; float *sink, *dump;
    hipMalloc(&out, (size_t)nwg * 8 * 8); hipMalloc(&sink, 16); hipMalloc(&dump, (size_t)nwg * 512 * 4);
    hipMemset(out, 0, (size_t)nwg * 8 * 8);
    hipLaunchKernelGGL((k_share<KIND, NACC>), dim3(nwg), dim3(512), 0, 0, out, sink, dump, iters);
    if (hipDeviceSynchronize() != hipSuccess) printf("launch failed: %s\n", hipGetErrorString(hipGetLastError()));
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)nwg * 8);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double g = 0, cyc = 0;
    for (int b = 0; b < nwg; ++b)
        for (int w = 0; w < 8; ++w) (w < 4 ? cyc : g) += h[b * 8 + w];
    printf("%s chain; second wave issues [%s]: chain %.1f cycles per MFMA, second wave %.2f groups per MFMA\n",
           NACC == 2 ? "two interleaved accumulators" : NACC == 1 ? "one dependent" : NACC == 11 ? "one dependent (second wave s_setprio 3)" : "two interleaved accumulators (second wave s_setprio 3)", what, cyc / (nwg * 4.0) / (32.0 * iters), g / (nwg * 4.0) / (32.0 * iters));
    hipFree(out); hipFree(sink); hipFree(dump);
}

template <int NV, bool PK, int MODE>
static void run(int iters)
{
    const int nwg = 256, thr = MODE == 0 ? 256 : 512;
    unsigned long long *out; float *sink;
    hipMalloc(&out, (size_t)nwg * 8 * 8); hipMalloc(&sink, 16);
    hipMemset(out, 0, (size_t)nwg * 8 * 8);
    hipLaunchKernelGGL((k_probe<NV, PK, MODE>), dim3(nwg), dim3(thr), 0, 0, out, sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)nwg * 8);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double lo = 0, hi = 0;      // waves 0-3 / 4-7
    for (int g = 0; g < nwg; ++g)
        for (int w = 0; w < 8; ++w) (w < 4 ? lo : hi) += h[g * 8 + w];
    lo /= nwg * 4.0; hi /= nwg * 4.0;
    const double nm = 32.0 * iters;
    if (MODE == 0)
        printf("same wave, 1 wave/SIMD, %2d %s per MFMA: %.1f cycles per MFMA\n", NV, PK ? "v_pk_add_f32" : "v_add_f32", lo / nm);
    else if (MODE == 2)
        printf("two chain waves/SIMD,   %2d %s per MFMA: %.1f cycles per MFMA and wave (64 = one wave alone)\n", NV, PK ? "v_pk_add_f32" : "v_add_f32", lo / nm);
    else
        printf("chain wave + VALU wave on a SIMD (%s): chain wave %.1f cycles per MFMA; VALU wave %.2f cycles per instruction "
               "(its %d x %d instructions took %.2f x the chain's time)\n", PK ? "v_pk_add_f32" : "v_add_f32", lo / nm,
               hi / (nm * (NV > 0 ? NV : 1)), (int)nm, NV > 0 ? NV : 1, hi / lo);
    hipFree(out); hipFree(sink);
}

int main()
{
    const int it = 4000;
    run<0, false, 0>(it); run<2, false, 0>(it); run<4, false, 0>(it); run<8, false, 0>(it); run<12, false, 0>(it);
    run<16, false, 0>(it); run<24, false, 0>(it);
    run<4, true, 0>(it); run<8, true, 0>(it); run<16, true, 0>(it);
    run<0, false, 2>(it); run<4, false, 2>(it); run<8, false, 2>(it); run<16, false, 2>(it);
    run<1, false, 1>(it); run<4, false, 1>(it); run<16, false, 1>(it); run<4, true, 1>(it);
    run_share<0, 11>(it, "v_add_f32");
    run_share<0, 12>(it, "v_add_f32");
    run_share<5, 12>(it, "v_cmp_lt_f32 -> vcc -> v_cndmask_b32");
    run_share<4, 12>(it, "s_add_u32");
    run_share<0, 2>(it, "v_add_f32");
    run_share<6, 2>(it, "2 independent v_add_f32");
    run_share<2, 2>(it, "v_add_f32, 2 s_add_u32");
    run_share<3, 2>(it, "v_add_f32, global_store_dword");
    run_share<4, 2>(it, "s_add_u32");
    run_share<5, 2>(it, "v_cmp_lt_f32 -> vcc -> v_cndmask_b32");
    run_share<0>(it, "v_add_f32");
    run_share<6>(it, "2 independent v_add_f32");
    run_share<1>(it, "v_add_f32, s_nop 0");
    run_share<2>(it, "v_add_f32, 2 s_add_u32");
    run_share<3>(it, "v_add_f32, global_store_dword");
    run_share<4>(it, "s_add_u32");
    run_share<5>(it, "v_cmp_lt_f32 -> vcc -> v_cndmask_b32");
    return 0;
}
