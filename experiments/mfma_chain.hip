// Microbenchmark (diagnostic, not product): dependent v_mfma_f32_32x32x2_f32 chains, 1 or 2 waves per SIMD,
// operands in registers vs B from LDS, with/without a workgroup barrier every 98 MFMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: regs only  1: B from LDS  2: B from LDS + barrier per 98  3: regs + barrier per 98
__global__ __launch_bounds__(512) void chain(float *out, int iters)
{
    __shared__ float img[12000];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 12000; i += blockDim.x) img[i] = 0.001f * (i % 97);
    __syncthreads();
    float wf[98];
#pragma unroll
    for (int k = 0; k < 98; ++k) wf[k] = 0.01f * (k + lane);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float *bp = img + (lane >> 5) * 361 + ((lane & 31) >> 4) * 19 + (lane & 15);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 98; ++k) {
            float b = (MODE == 1 || MODE == 2) ? bp[(k / 49) * 722 + ((k % 49) / 7) * 19 + (k % 7)] : wf[(k + 1) % 98];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[k], b, acc, 0, 0, 0);
        }
        if (MODE >= 2) __syncthreads();
    }
    float s = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(int threads, const char *name)
{
    float *d;
    hipMalloc(&d, 256 * 512 * 4 * 4);
    const int iters = 400, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    chain<MODE><<<blocks, threads>>>(d, 10);
    hipEventRecord(e0);
    chain<MODE><<<blocks, threads>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mfma_per_simd = (double)iters * 98 * (threads / 64) / 4.0;
    double tf = 2.0 * 32 * 32 * 2 * (double)iters * 98 * (threads / 64) * blocks / (ms * 1e-3) / 1e12;
    printf("%-34s threads=%d  %.3f ms  %.1f ns per MFMA-slot per SIMD (64 cyc @2.4GHz = 26.7 ns)  %.1f TFLOP/s\n", name, threads, ms,
           ms * 1e6 / mfma_per_simd, tf);
    hipFree(d);
}
int main()
{
    run<0>(256, "regs, 1 wave/SIMD");
    run<0>(512, "regs, 2 waves/SIMD");
    run<1>(256, "B from LDS, 1 wave/SIMD");
    run<1>(512, "B from LDS, 2 waves/SIMD");
    run<3>(512, "regs + barrier/98, 2 waves/SIMD");
    run<2>(512, "LDS + barrier/98, 2 waves/SIMD");
    run<2>(256, "LDS + barrier/98, 1 wave/SIMD");
    return 0;
}
