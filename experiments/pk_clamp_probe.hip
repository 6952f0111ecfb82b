// Probe: is clamp(clamp(v * 2^127) * 2^127) with v_pk_mul_f32 ... clamp EXACTLY (v > 0 ? 1.0f : 0.0f) for every fp32 class
// (zeros, denormals, normals, infinities), in the default HIP denormal mode?  It would replace 2 v_cmp + 2 v_cndmask per
// register pair of the refractory update (s * wrp) by 2 packed instructions.   hipcc --offload-arch=gfx950 -O3 -o pk_clamp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 spike01(f32x2 v)
{
    const f32x2 big = {0x1p127f, 0x1p127f};
    f32x2 r;
    asm volatile("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(v), "v"(big));
    asm volatile("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(r), "v"(big));
    return r;
}

__global__ void k(const float *in, float *out, float *out1, int n)
{
    const int i = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (i + 1 >= n + 1) return;
    f32x2 v = {in[i], in[i + 1]};
    f32x2 r = spike01(v);
    out[i] = r[0], out[i + 1] = r[1];
    const f32x2 big = {0x1p127f, 0x1p127f};
    f32x2 r1;
    asm volatile("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r1) : "v"(v), "v"(big));      // single clamp, for the record
    out1[i] = r1[0], out1[i + 1] = r1[1];
}

int main()
{
    unsigned bits[] = {0x00000000u, 0x80000000u, 0x00000001u, 0x80000001u, 0x007fffffu, 0x807fffffu, 0x00800000u, 0x80800000u,
                       0x3f800000u, 0xbf800000u, 0x7f7fffffu, 0xff7fffffu, 0x7f800000u, 0xff800000u, 0x00000100u, 0x33800000u};
    const int n = sizeof(bits) / sizeof(bits[0]);
    float h[n], o[n], o1[n];
    memcpy(h, bits, sizeof(bits));
    float *d, *e, *f;
    hipMalloc(&d, sizeof(h)); hipMalloc(&e, sizeof(h)); hipMalloc(&f, sizeof(h));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e, f, n);
    hipMemcpy(o, e, sizeof(h), hipMemcpyDeviceToHost);
    hipMemcpy(o1, f, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        const float want = h[i] > 0.0f ? 1.0f : 0.0f;
        unsigned ob, o1b;
        memcpy(&ob, &o[i], 4); memcpy(&o1b, &o1[i], 4);
        printf("in %08x (%g) -> double clamp %08x single clamp %08x want %g %s\n", bits[i], h[i], ob, o1b, want,
               (o[i] == want && !std::signbit(o[i])) ? "ok" : "MISMATCH");
        bad += !(o[i] == want);
    }
    printf("%s\n", bad ? "FAILED" : "all exact");
    return bad;
}
