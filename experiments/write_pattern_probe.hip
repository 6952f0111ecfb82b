// Probe: HBM write bandwidth as a function of the ADDRESS PATTERN, at the first layer of BASELINE config 5's volume.
// k_lif_seq_w3<1> writes, per workgroup and timestep, 64 pieces of 512 bytes at a stride of 4 KB (one piece per channel plane of
// the pooled map (T,B,64,16,64)), the next timestep 1 GB further on: 16 384 half-kilobyte pieces in flight over the chip.
// Here: `nwg` workgroups of 512 threads; per step each writes nseg segments of seglen bytes at stride segstride (16 bytes per
// lane and store), then jumps stepstride bytes; workgroup w starts at w * wgstride.  Same total bytes for every pattern.
//   hipcc --offload-arch=gfx950 -O3 -o write_pattern_probe write_pattern_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(float *base, int wgper, long wgstride, int nseg, long seglen, long segstride, long stepstride, int T)
{
    // workgroup blockIdx.x = (sample blockIdx.x / wgper, part blockIdx.x % wgper): sample base 256 KB apart, parts wgstride apart
    char *p = (char *)base + (long)(blockIdx.x / wgper) * (256L * 1024) + (long)(blockIdx.x % wgper) * wgstride;
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    const long per = seglen / 16;                               // 16-byte pieces per segment
    const long total = (long)nseg * per;
    for (int t = 0; t < T; ++t) {
        for (long i = threadIdx.x; i < total; i += 512) {
            const long s = i / per, o = i % per;
            *(f32x4 *)(p + s * segstride + o * 16) = v;
        }
        p += stepstride;
    }
}

int main()
{
    const long GB = 1L << 30;
    const long bytes = 33 * GB;                                  // buffer (32 GB of pattern + slack; every pattern is bounds-checked below)
    float *buf;
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    struct P { const char *name; int nwg; int wgper; long wgstride; int nseg; long seglen, segstride, stepstride; int T; };
    // 32 GB = 128 steps x 1024 "samples" x 256 KB.  (a) the w3<1> pattern: 8 workgroups per sample, each 64 x 512 B at 4 KB
    // stride; (b) 2 workgroups per sample, each 32 planes x 4 KB = 128 KB contiguous; (c) 1 workgroup per sample, 256 KB contiguous
    const P pats[] = {
        {"w3<1>: 8 WG/sample, 64 x 512 B @ 4 KB", 8192, 8, 512, 64, 512, 4096, 256L * 1024 * 1024, 128},
        {"4 WG/sample, 64 x 1 KB @ 4 KB", 4096, 4, 1024, 64, 1024, 4096, 256L * 1024 * 1024, 128},
        {"2 WG/sample, 128 KB contiguous", 2048, 2, 128 * 1024, 1, 128 * 1024, 0, 256L * 1024 * 1024, 128},
        {"1 WG/sample, 256 KB contiguous", 1024, 1, 256 * 1024, 1, 256 * 1024, 0, 256L * 1024 * 1024, 128},
        {"8 WG/sample, 32 KB contiguous each", 8192, 8, 32 * 1024, 1, 32 * 1024, 0, 256L * 1024 * 1024, 128},
        {"8 WG/sample, 64 x 512 B @ 4 KB, 4-byte stores emulated by 128 x 256 B", 8192, 8, 512, 128, 256, 2048, 256L * 1024 * 1024, 128},
    };
    for (const P &q : pats) {
        // last byte any workgroup of this pattern touches — checked on the host before anything is launched
        const long last = (long)((q.nwg - 1) / q.wgper) * (256L * 1024) + (long)((q.nwg - 1) % q.wgper) * q.wgstride +
                          (long)(q.T - 1) * q.stepstride + (long)(q.nseg - 1) * q.segstride + q.seglen;
        if (last > bytes || q.seglen % 16 != 0 || q.wgstride % 16 != 0 || q.segstride % 16 != 0) {
            printf("%-42s SKIPPED: pattern leaves the buffer (%ld > %ld) or is not 16-byte aligned\n", q.name, last, bytes);
            continue;
        }
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(q.nwg), dim3(512), 0, 0, buf, q.wgper, q.wgstride, q.nseg, q.seglen, q.segstride, q.stepstride, q.T);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double tot = (double)q.nwg * q.nseg * q.seglen * q.T;
        printf("%-42s %6.1f GB in %7.2f ms = %.2f TB/s\n", q.name, tot / 1e9, best, tot / best / 1e9);
    }
    return 0;
}
