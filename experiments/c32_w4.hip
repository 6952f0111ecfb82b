// Experiment (not product): k_lif_seq_c32 with FOUR waves (one per SIMD, up to 512 registers each) instead of eight.
// Each wave owns 8 input channels (196 weight fragments); every wave does, in EVERY stage, one epilogue quad, one
// trace share (one channel) and 196 chained MFMAs — perfectly uniform stages, no two-wave arbitration on the SIMD,
// half the accumulator hand-offs.  Question: does it beat the 8-wave kernel (101.4 ms at B=4096)?
#include "../snn_modulation_classification_amd/csrc/dcll_hip.hip"
#include <vector>

template <bool REFRACTORY>
__global__ __launch_bounds__(256) void k_lif_seq_c32_w4(const uint32_t *__restrict__ spk_in, const float *__restrict__ W,
                                                         const float *__restrict__ bias, const float *__restrict__ tau4,
                                                         float *__restrict__ eps0_g, float *__restrict__ eps1_g,
                                                         float *__restrict__ arp_g, uint32_t *__restrict__ spk_out,
                                                         float *__restrict__ pv_out, int T, int B, float alpharp, float wrp)
{
    constexpr int NW = 4, CP = 8;
    __shared__ __attribute__((aligned(16))) float lds[2 * IMG_FLOATS + NW * 2 * SLOT_FLOATS + 32];
    float *slots = lds + 2 * IMG_FLOATS;
    float *sbias = slots + NW * 2 * SLOT_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, j = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long b = blockIdx.x;
    for (int i = tid; i < 2 * IMG_FLOATS; i += 256) lds[i] = 0.0f;
    if (tid < 32) sbias[tid] = bias[tid];
    float wf[4][49];
#pragma unroll
    for (int cp = 0; cp < 4; ++cp)
#pragma unroll
        for (int k = 0; k < 49; ++k) wf[cp][k] = W[((long)j * 32 + CP * w + 2 * cp + h) * 49 + k];
    float e0[32];
    const int ioff = (CP * w) * CHF + ((lane >> 4) + 3) * ROWF + (lane & 15) + 3;
    const unsigned long long *in_wave = (const unsigned long long *)(spk_in + (b * 32 + CP * w) * 8);
    const long in_step = (long)B * 32 * 4;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CP; ++c) {
        const float ta = tau4[0 * 32 + CP * w + c], tm = tau4[1 * 32 + CP * w + c];
        const float tas = tau4[2 * 32 + CP * w + c], ts = tau4[3 * 32 + CP * w + c];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const long gidx = (b * 32 + CP * w + c) * 256 + ii * 64 + lane;
            e0[c * 4 + ii] = eps0_g[gidx];
            float e1 = eps1_g[gidx];
            float xin = (float)((in_wave[c * 4 + ii] >> lane) & 1ull);
            trace_update(xin, ta, tm, tas, ts, e0[c * 4 + ii], e1);
            lds[ioff + c * CHF + ii * 4 * ROWF] = e1;
        }
    }
    float arp[8][4];      // tile m, quad w: channel rr + 8w + 4h, pixel 32m + j
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) arp[m][rr] = REFRACTORY ? arp_g[(b * 32 + rr + 8 * w + 4 * h) * 256 + 32 * m + j] : 0.0f;
    const int bbase = (CP * w + h) * CHF + (j >> 4) * ROWF + (j & 15);
    unsigned long long pw0 = 0, pw1 = 0, pw2 = 0, pw3 = 0;
    if (T > 1) {
        const unsigned long long *ip = in_wave + in_step;
        pw0 = ip[0]; pw1 = ip[1]; pw2 = ip[2]; pw3 = ip[3];
    }
    auto trace_elem = [&](unsigned long long mask, float &e0r, float e1, float *dst, float ta, float tm, float tas, float ts) {
        float a;
        asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(a) : "v"(ts), "s"(mask));
        float bb = tas * e0r;
        e0r = a + bb;
        float cc = ta * e1;
        float dd = e0r * tm;
        *dst = cc + dd;
    };
    __syncthreads();
    const int nstage = 8 * T + NW + 1;
    for (int g = 0; g < nstage; ++g) {
        // epilogue: quad w of tile qe = g - NW (wave NW-1 finished it in stage g-1)
        const int qe = g - NW;
        if (qe >= 0 && qe < 8 * T) {
            const int te = qe >> 3, me = qe & 7;
            const f32x4 v4 = *((const f32x4 *)(slots + ((NW - 1) * 2 + ((g - 1) & 1)) * SLOT_FLOATS) + w * 64 + lane);
            const long obase = ((long)te * B + b) * 32 + 8 * w + 4 * h;
            float *pvp = pv_out + obase * 256 + 32 * me + j;
            auto quad = [&](float (&ar)[4]) {
                uint32_t myword = 0;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    float v = v4[rr];
                    bool s;
                    if (REFRACTORY) v = refractory(v4[rr], ar[rr], alpharp, wrp, s);
                    else s = v > 0.0f;
                    unsigned long long mk = __ballot(s);
                    uint32_t mine = h ? (uint32_t)(mk >> 32) : (uint32_t)mk;
                    myword = (j == rr) ? mine : myword;
                    pvp[rr * 256] = sigmoidf_dev(v);
                }
                if (spk_out && j < 4) spk_out[(obase + j) * 8 + me] = myword;
            };
            switch (me) {
            case 0: quad(arp[0]); break; case 1: quad(arp[1]); break; case 2: quad(arp[2]); break; case 3: quad(arp[3]); break;
            case 4: quad(arp[4]); break; case 5: quad(arp[5]); break; case 6: quad(arp[6]); break; default: quad(arp[7]); break;
            }
        }
        const int q = g - w;
        if (q >= 0 && q < 8 * T) {
            const int m = q & 7, t = q >> 3;
            float *img = lds + (t & 1) * IMG_FLOATS;
            if (t + 1 < T) {      // trace share: channel c = m of step t+1
                const int c = m;
                const float ta = tau4[0 * 32 + CP * w + c], tm = tau4[1 * 32 + CP * w + c];
                const float tas = tau4[2 * 32 + CP * w + c], ts = tau4[3 * 32 + CP * w + c];
                const unsigned long long w0 = pw0, w1 = pw1, w2 = pw2, w3 = pw3;
                {
                    const int tn = (c < 7) ? t + 1 : t + 2, cn = (c + 1) & 7;
                    if (tn < T) {
                        const unsigned long long *ip = in_wave + (long)tn * in_step + cn * 4;
                        pw0 = ip[0]; pw1 = ip[1]; pw2 = ip[2]; pw3 = ip[3];
                    }
                }
                const float *src = img + ioff + c * CHF;
                float *dst = lds + ((t + 1) & 1) * IMG_FLOATS + ioff + c * CHF;
                const float s0 = src[0], s1 = src[4 * ROWF], s2 = src[8 * ROWF], s3 = src[12 * ROWF];
                switch (c) {
#define TC(C) case C: trace_elem(w0, e0[C * 4 + 0], s0, dst, ta, tm, tas, ts); trace_elem(w1, e0[C * 4 + 1], s1, dst + 4 * ROWF, ta, tm, tas, ts); \
                      trace_elem(w2, e0[C * 4 + 2], s2, dst + 8 * ROWF, ta, tm, tas, ts); trace_elem(w3, e0[C * 4 + 3], s3, dst + 12 * ROWF, ta, tm, tas, ts); break;
                    TC(0) TC(1) TC(2) TC(3) TC(4) TC(5) TC(6) TC(7)
#undef TC
                }
            }
            f32x16 acc;
            if (w == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = sbias[(r & 3) + 8 * (r >> 2) + 4 * h];
            } else {
                const f32x4 *sp = (const f32x4 *)(slots + ((w - 1) * 2 + ((g - 1) & 1)) * SLOT_FLOATS) + lane;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    f32x4 v4 = sp[c * 64];
                    acc[4 * c + 0] = v4[0]; acc[4 * c + 1] = v4[1]; acc[4 * c + 2] = v4[2]; acc[4 * c + 3] = v4[3];
                }
            }
            const float *bp = img + bbase + m * 2 * ROWF;
            float bq[2][7];
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) bq[0][kx] = bp[kx];
#pragma unroll
            for (int r = 0; r < 28; ++r) {
                if (r + 1 < 28) {
                    const int cpn = (r + 1) / 7, kyn = (r + 1) % 7;
#pragma unroll
                    for (int kx = 0; kx < 7; ++kx) bq[(r + 1) & 1][kx] = bp[cpn * 2 * CHF + kyn * ROWF + kx];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kx = 0; kx < 7; ++kx)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[r / 7][(r % 7) * 7 + kx], bq[r & 1][kx], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            f32x4 *dp = (f32x4 *)(slots + (w * 2 + (g & 1)) * SLOT_FLOATS) + lane;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f32x4 v4 = {acc[4 * c + 0], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]};
                dp[c * 64] = v4;
            }
        }
        __syncthreads();
    }
    const float *fin = lds + ((T - 1) & 1) * IMG_FLOATS;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        long gidx = (b * 32 + CP * w + (i >> 2)) * 256 + (i & 3) * 64 + lane;
        eps0_g[gidx] = e0[i];
        eps1_g[gidx] = fin[ioff + (i >> 2) * CHF + (i & 3) * 4 * ROWF];
    }
    if (REFRACTORY) {
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) arp_g[(b * 32 + rr + 8 * w + 4 * h) * 256 + 32 * m + j] = arp[m][rr];
    }
}

int main()
{
    const int B = 1024, T = 128;
    size_t nin = (size_t)T * B * 32 * 8;
    uint32_t *spk_in, *spk_a, *spk_b; float *W, *bias, *tau4, *st[6], *pv;
    hipMalloc(&spk_in, nin * 4); hipMalloc(&spk_a, nin * 4); hipMalloc(&spk_b, nin * 4);
    std::vector<uint32_t> hin(nin);
    unsigned x = 12345;
    for (auto &v : hin) { x = x * 1664525u + 1013904223u; v = (x >> 7) & (x >> 13) & (x >> 3) & (x >> 21); }   // ~6 % ones
    hipMemcpy(spk_in, hin.data(), nin * 4, hipMemcpyHostToDevice);
    hipMalloc(&W, 32 * 32 * 49 * 4); hipMalloc(&bias, 128); hipMalloc(&tau4, 512);
    std::vector<float> hw(32 * 32 * 49), hb(32), ht(128);
    for (size_t i = 0; i < hw.size(); ++i) { x = x * 1664525u + 1013904223u; hw[i] = ((int)(x >> 8) % 2001 - 1000) * 6e-9f; }
    for (int i = 0; i < 32; ++i) { x = x * 1664525u + 1013904223u; hb[i] = ((int)(x >> 8) % 2001 - 1000) * 1e-7f; }
    for (int i = 0; i < 32; ++i) { ht[i] = 0.9f + 0.002f * i; ht[32 + i] = 1.0f / (1.0f - ht[i]); ht[64 + i] = 0.8f + 0.003f * i; ht[96 + i] = 1.0f / (1.0f - ht[64 + i]); }
    hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(bias, hb.data(), 128, hipMemcpyHostToDevice); hipMemcpy(tau4, ht.data(), 512, hipMemcpyHostToDevice);
    size_t ns = (size_t)B * 32 * 256;
    for (int i = 0; i < 6; ++i) { hipMalloc(&st[i], ns * 4); hipMemset(st[i], 0, ns * 4); }
    hipMalloc(&pv, (size_t)T * ns * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms8 = 1e9, ms4 = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 6; ++i) hipMemset(st[i], 0, ns * 4);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_lif_seq_c32<true, 1, 0>), dim3(B), dim3(512), 0, 0, spk_in, dcll_wsrc{W, nullptr, nullptr}, bias, tau4, st[0], st[1], st[2], spk_a, pv,
                           (float *)nullptr, (const float *)nullptr, (const float *)nullptr, (float *)nullptr, T, B, 0.65f, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms8 = ms < ms8 ? ms : ms8;
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_lif_seq_c32_w4<true>), dim3(B), dim3(256), 0, 0, spk_in, W, bias, tau4, st[3], st[4], st[5], spk_b, pv, T, B, 0.65f, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); ms4 = ms < ms4 ? ms : ms4;
    }
    std::vector<uint32_t> ha(nin), hb2(nin);
    hipMemcpy(ha.data(), spk_a, nin * 4, hipMemcpyDeviceToHost); hipMemcpy(hb2.data(), spk_b, nin * 4, hipMemcpyDeviceToHost);
    size_t diff = 0, ones = 0;
    for (size_t i = 0; i < nin; ++i) { diff += ha[i] != hb2[i]; ones += __builtin_popcount(ha[i]); }
    double ideal = 2.0 * 32 * 1568 * 256 * (double)T * B / 157.3e12 * 1e3;
    printf("ideal %.2f ms | 8 waves (product) %.2f ms (%.1f%%) | 4 waves %.2f ms (%.1f%%) | spike words differing %zu of %zu, ones %.3f%%\n", ideal, ms8,
           100 * ideal / ms8, ms4, 100 * ideal / ms4, diff, nin, 100.0 * ones / (nin * 32.0));
    return 0;
}
