// Microbenchmark (diagnostic): one vs two independent accumulator chains per wave, 2 waves/SIMD, long run.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NCH>
__global__ __launch_bounds__(512) void chain(float *out, int iters)
{
    const int lane = threadIdx.x & 63;
    float wf[98];
#pragma unroll
    for (int k = 0; k < 98; ++k) wf[k] = 0.01f * (k + lane);
    f32x16 acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 98; ++k)
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[k], wf[(k + 1 + c) % 98], acc[c], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NCH>
static void run(int threads, int iters, const char *name)
{
    float *d;
    hipMalloc(&d, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    chain<NCH><<<256, threads>>>(d, 10);
    hipEventRecord(e0);
    chain<NCH><<<256, threads>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double tf = 2.0 * 32 * 32 * 2 * (double)iters * 98 * NCH * (threads / 64) * 256 / (ms * 1e-3) / 1e12;
    printf("%-30s threads=%d iters=%d  %.2f ms  %.1f TFLOP/s (%.1f%% of 157.3)\n", name, threads, iters, ms, tf, tf / 1.573);
    hipFree(d);
}
int main()
{
    run<1>(512, 400, "1 chain/wave, 2 waves/SIMD");
    run<1>(512, 8000, "1 chain/wave, 2 waves/SIMD");
    run<2>(512, 4000, "2 chains/wave, 2 waves/SIMD");
    run<2>(256, 8000, "2 chains/wave, 1 wave/SIMD");
    run<4>(256, 4000, "4 chains/wave, 1 wave/SIMD");
    run<1>(256, 16000, "1 chain/wave, 1 wave/SIMD");
    return 0;
}
