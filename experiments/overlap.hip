// Does VALU work of one wave overlap with its SIMD partner's MFMA chain? (diagnostic, not product)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: all waves 98 MFMAs + barrier.  MODE 1: wave (it&7) first does NV dependent VALU ops.  MODE 2: same but the
// extra work is NV/8 global stores.  MODE 3: extra = LDS read-modify-write chain.
template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, float *sink, int iters, int NV)
{
    __shared__ float img[12000];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 12000; i += blockDim.x) img[i] = 0.001f * (i % 97);
    __syncthreads();
    float wf[98];
#pragma unroll
    for (int kk = 0; kk < 98; ++kk) wf[kk] = 0.01f * (kk + lane);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float *bp = img + (lane >> 5) * 361 + ((lane & 31) >> 4) * 19 + (lane & 15);
    float x = lane * 0.5f;
    for (int it = 0; it < iters; ++it) {
        if (MODE != 0 && (it & 7) == w) {
            if (MODE == 1) {
                for (int i = 0; i < NV; ++i) x = x * 1.000001f + 0.5f;     // dependent VALU chain, ~NV*2 instrs
            } else if (MODE == 4) {
                float y[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) y[u] = x + u;
                for (int i = 0; i < NV; ++i) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) y[u] = y[u] * 1.000001f + 0.5f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) x += y[u];
            } else if (MODE == 5) {
                for (int i = 0; i < NV; ++i) x = 1.0f / (1.0f + __expf(-x));
            } else if (MODE == 2) {
                for (int i = 0; i < NV / 8; ++i) sink[((size_t)blockIdx.x * 8 + w) * 4096 + (i & 63) * 64 + lane] = x + i;
            } else {
                for (int i = 0; i < NV / 8; ++i) { float t = img[6000 + lane + 64 * (i & 15)]; img[6000 + lane + 64 * (i & 15)] = t * 1.01f + x; }
            }
        }
#pragma unroll
        for (int kk = 0; kk < 98; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[kk], bp[(kk / 49) * 722 + ((kk % 49) / 7) * 19 + (kk % 7)], acc, 0, 0, 0);
        __syncthreads();
    }
    float s = x;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
static void run(const char *name, int NV)
{
    float *d, *sink;
    hipMalloc(&d, 256 * 512 * 4);
    hipMalloc(&sink, (size_t)256 * 8 * 4096 * 4);
    const int iters = 800;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 512>>>(d, sink, 10, NV);
    hipEventRecord(e0);
    k<MODE><<<256, 512>>>(d, sink, iters, NV);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s NV=%5d  %.3f ms  -> %.0f cycles per stage @2.4GHz (ideal 12544)\n", name, NV, ms, ms * 1e-3 / iters * 2.4e9);
    hipFree(d); hipFree(sink);
}
int main()
{
    run<0>("baseline", 0);
    run<0>("baseline", 0);
    run<1>("one wave: dependent VALU chain first", 25);
    run<1>("one wave: dependent VALU chain first", 50);
    run<1>("one wave: dependent VALU chain first", 100);
    run<1>("one wave: dependent VALU chain first", 200);
    run<4>("one wave: 8 independent VALU chains first", 25);
    run<4>("one wave: 8 independent VALU chains first", 50);
    run<4>("one wave: 8 independent VALU chains first", 100);
    run<4>("one wave: 8 independent VALU chains first", 200);
    run<5>("one wave: v_exp/v_rcp (sigmoid) x NV", 16);
    run<5>("one wave: v_exp/v_rcp (sigmoid) x NV", 64);
    run<5>("one wave: v_exp/v_rcp (sigmoid) x NV", 256);
    return 0;
}
