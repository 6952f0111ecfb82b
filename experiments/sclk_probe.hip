// What does the shader clock do under a full fp32-MFMA load?  (diagnostic, not product)   ./sclk_probe
// Every wave runs a dependent chain of v_mfma_f32_32x32x2_f32 (optionally with LDS operand reads, as the layer kernels
// do) and reads both timers before and after: s_memtime counts shader-clock cycles, s_memrealtime the constant 100 MHz
// reference.  Their ratio is the average shader clock DURING the kernel; MFMAs / s_memtime cycles is the issue rate in
// the hardware's own clock.  Peak figures quoted against 2.4 GHz assume the clock holds under this load.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int LDSOPS>
__global__ __launch_bounds__(256) void k_probe(unsigned long long *out, float *sink, int iters)
{
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 1e-3f * (i & 7);
    __syncthreads();
    f32x16 a0, a1;
    for (int r = 0; r < 16; ++r) a0[r] = a1[r] = 0.0f;
    const int lane = threadIdx.x & 63;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    float x = 1.0f + lane * 1e-6f, y = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (LDSOPS == 1) {                 // scattered addresses (conflicts), computed per read
                x = lds[(lane + 67 * u + it) & 8191];
                y = lds[(lane * 2 + 129 * u + it) & 8191];
            } else if (LDSOPS == 2) {          // conflict-free, immediate offsets from one lane base
                const float *pl = lds + lane + (it & 7) * 16;
                x = pl[64 * u];
                y = pl[64 * u + 2048];
            } else if (LDSOPS == 3) {          // the layer kernels' pattern: A fragment conflict-free, B fragment from the
                const float *pa = lds + lane + (it & 7) * 16;                  // 19-float-row image (2-way on 3 banks)
                const float *pb = lds + 4096 + (lane >> 5) * 361 + ((lane & 31) >> 4) * 19 + (lane & 15) + (it & 7) * 19;
                x = pa[64 * u];
                y = pb[u];
            }
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
    if (s == 12345.678f) sink[0] = s;
    if (lane == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        out[2 * w] = t1 - t0;
        out[2 * w + 1] = r1 - r0;
    }
}
template <int LDSOPS>
static void run(int nwg, int iters, const char *what)
{
    unsigned long long *out; float *sink;
    hipMalloc(&out, (size_t)nwg * 4 * 16); hipMalloc(&sink, 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k_probe<LDSOPS>, dim3(nwg), dim3(256), 0, 0, out, sink, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        std::vector<unsigned long long> h((size_t)nwg * 8);
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        double st = 0, rt = 0;
        for (int w = 0; w < nwg * 4; ++w) { st += h[2 * w]; rt += h[2 * w + 1]; }
        const double mfma = 32.0 * iters, waves = nwg * 4.0;
        printf("%-28s %5d WGs: %.2f ms | shader clock %.0f MHz | %.1f shader cycles per MFMA per wave | %.1f TFLOP/s = %.1f %% of 157.3\n", what, nwg, ms,
               st / rt * 100.0, st / waves / mfma, waves * mfma * 4096 / (ms * 1e-3) / 1e12, waves * mfma * 4096 / (ms * 1e-3) / 157.3e12 * 100);
    }
    hipFree(out); hipFree(sink);
}
int main()
{
    run<0>(256, 20000, "registers only, 1 wave/SIMD");
    run<0>(512, 20000, "registers only, 2 waves/SIMD");
    run<1>(256, 20000, "LDS scattered, 1 wave/SIMD");
    run<1>(512, 20000, "LDS scattered, 2 waves/SIMD");
    run<2>(256, 20000, "LDS conflict-free, 1 w/SIMD");
    run<2>(512, 20000, "LDS conflict-free, 2 w/SIMD");
    run<3>(256, 20000, "LDS layer pattern, 1 w/SIMD");
    run<3>(512, 20000, "LDS layer pattern, 2 w/SIMD");
    run<0>(64, 20000, "registers only, 64 CUs");
    return 0;
}
