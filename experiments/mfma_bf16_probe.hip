// Probe (diagnostic): how does v_mfma_f32_32x32x16_bf16 add the 16 products of one instruction and the accumulator?
// Every row of A carries the same 16-value pattern, B is all ones, so every D element = C + sum_k pattern[k]; patterns
// are chosen so that an exact (single-rounding) sum, a sequential fp32 sum and common tree orders give different answers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(const float *pat, float c0, float *out)
{
    const int lane = threadIdx.x, half = lane >> 5;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)pat[8 * half + e]; b[e] = (__bf16)1.0f; }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = c0;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    if (lane == 0) out[0] = c[0];
    if (lane == 37) out[1] = c[5];
}
static float run(const float *p, float c0)
{
    float *dp, *dout, h[2];
    hipMalloc(&dp, 64); hipMalloc(&dout, 8);
    hipMemcpy(dp, p, 64, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dp, c0, dout);
    hipMemcpy(h, dout, 8, hipMemcpyDeviceToHost);
    hipFree(dp); hipFree(dout);
    if (h[0] != h[1]) printf("  (elements differ: %g vs %g)\n", h[0], h[1]);
    return h[0];
}
static float seq(const float *p, float c0) { float s = c0; for (int k = 0; k < 16; ++k) s += p[k]; return s; }
static double exact(const float *p, float c0) { double s = c0; for (int k = 0; k < 16; ++k) s += p[k]; return s; }
int main()
{
    const float B30 = 1073741824.0f, B24 = 16777216.0f;
    struct { const char *name; float c0; float p[16]; } T[] = {
        {"big, 1, -big (same half)", 0, {B30, 1, -B30}},
        {"1, big, -big (same half)", 0, {1, B30, -B30}},
        {"big k=0, -big k=8, 1 k=1 (across halves)", 0, {B30, 1, 0, 0, 0, 0, 0, 0, -B30}},
        {"2^24 + fifteen ones", 0, {B24, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}},
        {"fifteen ones + 2^24 last", 0, {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, B24}},
        {"C = 2^24, sixteen ones", B24, {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}},
        {"C = 2^30, one, -? (C + 1 then -2^30 impossible): C=2^30, p = {-2^30, 1}", B30, {-B30, 1}},
        {"C = 1, big, -big", 1, {B30, -B30}},
        {"pairs: (big,-big) in k=0,1 ; ones elsewhere", 0, {B30, -B30, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}},
        {"2^24, 1 | 1, 1 ... by fours", 0, {B24, 1, 1, 1, 1, 1, 1, 1, 1}},
    };
    for (auto &t : T) {
        float got = run(t.p, t.c0);
        printf("%-60s MFMA %-14.1f sequential fp32 %-14.1f exact %-14.1f\n", t.name, got, seq(t.p, t.c0), exact(t.p, t.c0));
    }
    return 0;
}
