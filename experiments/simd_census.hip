// Which SIMD does each wave of a 512-thread / 112 KB-LDS workgroup land on?  (diagnostic, not product)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ __launch_bounds__(512) void census(unsigned *out, int spin)
{
    __shared__ float big[28000];
    unsigned hwid = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
    big[threadIdx.x] = hwid;
    float a = big[(threadIdx.x * 7) % 28000];
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 1.0f;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hwid;
    if (a == 12345.f) out[0] = 0;
}
int main()
{
    const int nb = 2048;
    unsigned *d, *h = (unsigned *)malloc(nb * 8 * 4);
    hipMalloc(&d, nb * 8 * 4);
    census<<<nb, 512>>>(d, 20000);
    hipMemcpy(h, d, nb * 8 * 4, hipMemcpyDeviceToHost);
    int hist[5] = {0}, pat[4][4] = {{0}};
    for (int b = 0; b < nb; ++b) {
        int cnt[4] = {0};
        for (int w = 0; w < 8; ++w) cnt[(h[b * 8 + w] >> 4) & 3]++;
        int mx = 0;
        for (int s = 0; s < 4; ++s) mx = cnt[s] > mx ? cnt[s] : mx;
        hist[mx]++;
        if (b < 6) {
            printf("block %d: ", b);
            for (int w = 0; w < 8; ++w) printf("w%d:simd%d(cu%d,wave%d) ", w, (h[b * 8 + w] >> 4) & 3, (h[b * 8 + w] >> 8) & 15, h[b * 8 + w] & 15);
            printf("\n");
        }
    }
    printf("max waves on one SIMD per workgroup: 2:%d 3:%d 4:%d\n", hist[2], hist[3], hist[4]);
    return 0;
}
