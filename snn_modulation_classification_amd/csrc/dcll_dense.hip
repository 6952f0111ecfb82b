// dcll_dense.hip — the dense twins: DenseDCLLlayer.forward (dcll/pytorch_libdcll.py:250-255) over CLLDenseModule.forward
// (:131-148) / CLLDenseRRPModule.forward (:171-195) — sixth translation unit of libdcll_hip.so.
//
//   v[b, o] = bias[o] + sum_k eps1[b, k] * W[o, k]      one fmaf chain per output, k ascending (include/dcll_hip.h)
//
// k_dense_lif_mfma — ONE step, any shape, as an fp32-MFMA GEMM whose K loop is never split: v_mfma_f32_32x32x2_f32 with its
//   two k lanes on the feature pair (2s, 2s + 1) is bit for bit fmaf(a1, b1, fmaf(a0, b0, acc)), so a tile's accumulator walks
//   the pinned chain when the k-steps are issued in order.  A = eps1 (row = batch sample), B = W (column = output neuron):
//   the accumulator has the output neuron on the lane — 32 consecutive floats per store.  Workgroup = 128 samples x 64
//   neurons (4 waves x (32 samples x 2 neuron tiles): one A fragment feeds two MFMAs), K in chunks of 32 staged through LDS
//   (row stride 34 floats: 8-byte aligned rows, and 34 i + k covers the 64 banks once over a fragment's 64 lanes), the next
//   chunk fetched into registers while the MFMAs of this one run.  Features beyond `in` are staged as zeros in BOTH operands:
//   fmaf(0, 0, acc) == acc.  (Round 1/2: one thread per output with a serial, uncoalesced K loop.)
//
// k_dense_lif_seq — ALL T steps of a small dense layer in one launch with the neuron state ON CHIP: a workgroup owns 32
//   samples for the whole sequence, eps0 in registers, eps1 in LDS in the A-fragment layout (updated in place, read by the
//   MFMAs of the same step), arp in registers; W streams from L2 per step in chunks through LDS.  Serves in_features <= 1024
//   and out_features <= 128 (what fits: 32 x 1024 eps1 floats = 128 KB of the 160 KB LDS; 4 waves x one neuron tile);
//   larger layers run step by step through k_trace + k_dense_lif_mfma with the state in HBM (dcll_dense_lif_sequence in
//   dcll_hip.hip decides) — at in_features = 8192 a single sample's traces are 64 KB: they cannot stay on chip.
#include "dcll_internal.h"

constexpr int DN_BT = 128, DN_OT = 64, DN_KC = 32, DN_LD = 34;

// 4 consecutive floats of row `row` (nrows rows of n floats) from column k0, zeros outside; float4 when it is aligned
__device__ __forceinline__ f32x4 dn_load4(const float *__restrict__ m, long row, long nrows, int n, int k0, bool vec)
{
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < nrows) {
        const float *p = m + row * n + k0;
        if (vec && k0 + 3 < n) {
            v = *(const f32x4 *)p;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k0 + e < n) v[e] = p[e];
        }
    }
    return v;
}

template <bool REFRACTORY>
__global__ __launch_bounds__(256) void k_dense_lif_mfma(int in, int out, const float *__restrict__ eps1,
                                                         const float *__restrict__ W, const float *__restrict__ bias,
                                                         float *__restrict__ arp, float *__restrict__ s_out,
                                                         float *__restrict__ pv_out, float *__restrict__ v_out, int B,
                                                         float alpharp, float wrp)
{
    __shared__ __attribute__((aligned(16))) float sE[DN_BT * DN_LD];
    __shared__ __attribute__((aligned(16))) float sW[DN_OT * DN_LD];
    const int tid = threadIdx.x, lane = tid & 63, jj = lane & 31, kk = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long b0 = (long)blockIdx.y * DN_BT;
    const int o0 = blockIdx.x * DN_OT;
    const int r = tid >> 3, c4 = (tid & 7) * 4;             // staging: 8 threads x float4 = one 32-float chunk of a row
    const bool vec = (in % 4 == 0) && ((((uintptr_t)eps1 | (uintptr_t)W) & 15) == 0);
    f32x4 re[4], rw[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) re[q] = dn_load4(eps1, b0 + r + 32 * q, B, in, k0 + c4, vec);
#pragma unroll
        for (int q = 0; q < 2; ++q) rw[q] = dn_load4(W, o0 + r + 32 * q, out, in, k0 + c4, vec);
    };
    f32x16 acc0, acc1;                                      // chains start from the bias (lane = output neuron)
    {
        const float bz0 = (bias && o0 + jj < out) ? bias[o0 + jj] : 0.0f, bz1 = (bias && o0 + 32 + jj < out) ? bias[o0 + 32 + jj] : 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc0[q] = bz0; acc1[q] = bz1; }
    }
    fetch(0);
    for (int k0 = 0; k0 < in; k0 += DN_KC) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float *dst = sE + (r + 32 * q) * DN_LD + c4;
            *(f32x2 *)dst = f32x2{re[q][0], re[q][1]};
            *(f32x2 *)(dst + 2) = f32x2{re[q][2], re[q][3]};
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float *dst = sW + (r + 32 * q) * DN_LD + c4;
            *(f32x2 *)dst = f32x2{rw[q][0], rw[q][1]};
            *(f32x2 *)(dst + 2) = f32x2{rw[q][2], rw[q][3]};
        }
        __syncthreads();
        if (k0 + DN_KC < in) fetch(k0 + DN_KC);
        const float *ea = sE + (32 * w + jj) * DN_LD + kk, *wb0 = sW + jj * DN_LD + kk, *wb1 = sW + (32 + jj) * DN_LD + kk;
#pragma unroll
        for (int s = 0; s < DN_KC / 2; ++s) {
            const float a = ea[2 * s];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb0[2 * s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb1[2 * s], acc1, 0, 0, 0);
        }
        __syncthreads();
    }
    // D layout: register q of lane (jj, kk) = sample (q & 3) + 8 (q >> 2) + 4 kk of the wave's 32, neuron jj of the tile
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
        const int o = o0 + 32 * tl + jj;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const long b = b0 + 32 * w + (q & 3) + 8 * (q >> 2) + 4 * kk;
            if (b < B && o < out) {
                const long i = b * out + o;
                const float pvm = tl ? acc1[q] : acc0[q];
                float v = pvm;
                bool s;
                if (REFRACTORY) {
                    float a = arp[i];
                    v = refractory(pvm, a, alpharp, wrp, s);
                    arp[i] = a;
                } else {
                    s = v > 0.0f;
                }
                if (v_out) v_out[i] = v;
                if (s_out) s_out[i] = s ? 1.0f : 0.0f;
                if (pv_out) pv_out[i] = sigmoidf_dev(v);
            }
        }
    }
}

int dcll_launch_dense_mfma(const dcll_dense_desc *d, const float *eps1, const float *W, const float *b, float *arp,
                           float *out_s, float *out_pv, float *out_v, int32_t B, hipStream_t st)
{
    const dim3 g((unsigned)((d->out_features + DN_OT - 1) / DN_OT), (unsigned)((B + DN_BT - 1) / DN_BT));
    if (g.y > 65535) return fail(DCLL_ERR_UNSUPPORTED, "dense layer: batch above 8 M samples per call");
    if (d->refractory)
        hipLaunchKernelGGL(k_dense_lif_mfma<true>, g, dim3(256), 0, st, d->in_features, d->out_features, eps1, W, b, arp, out_s,
                           out_pv, out_v, B, d->alpharp, d->wrp);
    else
        hipLaunchKernelGGL(k_dense_lif_mfma<false>, g, dim3(256), 0, st, d->in_features, d->out_features, eps1, W, b, arp, out_s,
                           out_pv, out_v, B, d->alpharp, d->wrp);
    HIP_CHECK_LAUNCH("k_dense_lif_mfma");
    return DCLL_OK;
}

// ------------------------------------------------------------------------------------------------------------
// k_dense_lif_seq: all T steps, 32 samples per workgroup, state on chip (see the file header)
// ------------------------------------------------------------------------------------------------------------
constexpr int DS_MAXIN = 1024, DS_MAXOUT = 128, DS_KC = 32;
constexpr int DS_NC = DS_MAXIN / 64;            // feature columns per lane at the largest in_features

template <bool REFRACTORY>
__global__ __launch_bounds__(256) void k_dense_lif_seq(int in, int out, const float *__restrict__ x, const float *__restrict__ W,
                                                        const float *__restrict__ bias, const float *__restrict__ alpha,
                                                        const float *__restrict__ tau_m, const float *__restrict__ alphas,
                                                        const float *__restrict__ tau_s, int tau_is_tensor,
                                                        float *__restrict__ eps0_g, float *__restrict__ eps1_g,
                                                        float *__restrict__ arp_g, float *__restrict__ s_out,
                                                        float *__restrict__ pv_out, float *__restrict__ v_out, int T, int B,
                                                        float alpharp, float wrp)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // eps1 of my 32 samples in A-fragment layout: sample i, feature k at i * LDE + k.  LDE = (in rounded up to the chunk) + 2:
    // rows 8-byte aligned and LDE / 2 odd, so the 64 lanes (i, k parity) of a fragment read cover the 64 banks once; the
    // features between `in` and the end of the last chunk stay zero (never written)
    const int LDE = ((in + DS_KC - 1) / DS_KC) * DS_KC + 2;
    float *sE = lds, *sW = lds + 32 * LDE;                  // sW: 128 neurons x DN_LD, one K-chunk of W
    const int tid = threadIdx.x, lane = tid & 63, jj = lane & 31, kk = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long b0 = (long)blockIdx.x * 32;
    // trace ownership: thread (wave w, lane) owns samples w + 4 m (m < 8), features lane + 64 c (c < in / 64): a wave's
    // access is 64 consecutive features of one sample — coalesced x / state traffic, conflict-free LDS; eps0 in registers
    float e0[8][DS_NC];
    for (int i = tid; i < 32 * LDE; i += 256) sE[i] = 0.0f;
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int c = 0; c < DS_NC; ++c) {
            const int i = w + 4 * m, k = lane + 64 * c;
            e0[m][c] = 0.0f;
            if (64 * c < in && k < in && b0 + i < B) {
                e0[m][c] = eps0_g[(b0 + i) * in + k];
                sE[i * LDE + k] = eps1_g[(b0 + i) * in + k];
            }
        }
    float arp[16];
    const int o = 32 * w + jj;                              // my neuron (wave w = neuron tile w)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const long b = b0 + (q & 3) + 8 * (q >> 2) + 4 * kk;
        arp[q] = (REFRACTORY && b < B && o < out) ? arp_g[b * out + o] : 0.0f;
    }
    const float bz = (bias && o < out) ? bias[o] : 0.0f;
    const int r = tid >> 3, c4 = (tid & 7) * 4;             // W staging: 8 threads x float4 per row chunk, rows r + 32 q
    const bool vec = (in % 4 == 0) && (((uintptr_t)W & 15) == 0);
    const bool active = 32 * w < out;                       // wave-uniform: my neuron tile exists
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        // ---- traces of step t (dcll/pytorch_libdcll.py:139-140, every op rounded separately) ----
        const float *xt = x + (long)t * B * in;
#pragma unroll
        for (int c = 0; c < DS_NC; ++c) {
            if (64 * c < in) {                              // wave-uniform
                const int k = lane + 64 * c, q = tau_is_tensor ? min(k, in - 1) : 0;
                const float ta = alpha[q], tm = tau_m[q], tas = alphas[q], ts = tau_s[q];
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int i = w + 4 * m;
                    if (k < in && b0 + i < B) {
                        float e1 = sE[i * LDE + k];
                        trace_update(xt[(b0 + i) * in + k], ta, tm, tas, ts, e0[m][c], e1);
                        sE[i * LDE + k] = e1;
                    }
                }
            }
        }
        // ---- the chain of my 32 x 32 tile over all of K, W in chunks of 32 through LDS ----
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = bz;
        f32x4 rw[4];
        auto fetch = [&](int k0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) rw[q] = dn_load4(W, r + 32 * q, out, in, k0 + c4, vec);
        };
        fetch(0);
        for (int k0 = 0; k0 < in; k0 += DS_KC) {
            __syncthreads();                                // traces of this step written / previous chunk's reads done
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float *dst = sW + (r + 32 * q) * DN_LD + c4;
                *(f32x2 *)dst = f32x2{rw[q][0], rw[q][1]};
                *(f32x2 *)(dst + 2) = f32x2{rw[q][2], rw[q][3]};
            }
            __syncthreads();
            if (k0 + DS_KC < in) fetch(k0 + DS_KC);
            if (active) {
                const float *ea = sE + jj * LDE + k0 + kk, *wb = sW + (32 * w + jj) * DN_LD + kk;
#pragma unroll
                for (int s = 0; s < DS_KC / 2; ++s)        // (features >= in: zeros in sE — never written — and in sW)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ea[2 * s], wb[2 * s], acc, 0, 0, 0);
            }
        }
        // ---- epilogue: register q of lane (jj, kk) = sample (q & 3) + 8 (q >> 2) + 4 kk, neuron 32 w + jj ----
        if (active && o < out) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const long b = b0 + (q & 3) + 8 * (q >> 2) + 4 * kk;
                float v = acc[q];
                bool s;
                if (REFRACTORY) v = refractory(acc[q], arp[q], alpharp, wrp, s);
                else s = v > 0.0f;
                if (b < B) {
                    const long i = ((long)t * B + b) * out + o;
                    if (v_out) v_out[i] = v;
                    if (s_out) s_out[i] = s ? 1.0f : 0.0f;
                    if (pv_out) pv_out[i] = sigmoidf_dev(v);
                }
            }
        }
        __syncthreads();                                    // every chain has read sE before the next step's traces
    }
    // ---- state back to HBM ----
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int c = 0; c < DS_NC; ++c) {
            const int i = w + 4 * m, k = lane + 64 * c;
            if (64 * c < in && k < in && b0 + i < B) {
                eps0_g[(b0 + i) * in + k] = e0[m][c];
                eps1_g[(b0 + i) * in + k] = sE[i * LDE + k];
            }
        }
    if (REFRACTORY && active && o < out) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const long b = b0 + (q & 3) + 8 * (q >> 2) + 4 * kk;
            if (b < B) arp_g[b * out + o] = arp[q];
        }
    }
}

bool dcll_dense_seq_fits(const dcll_dense_desc *d) { return d->in_features <= DS_MAXIN && d->out_features <= DS_MAXOUT; }

int dcll_launch_dense_seq(const dcll_dense_desc *d, const float *x, const float *W, const float *b, const float *alpha,
                          const float *tau_m, const float *alphas, const float *tau_s, float *eps0, float *eps1, float *arp,
                          float *out_s, float *out_pv, float *out_v, int32_t T, int32_t B, hipStream_t st)
{
    const int LDE = ((d->in_features + DS_KC - 1) / DS_KC) * DS_KC + 2;
    const size_t lds_bytes = (size_t)(32 * LDE + 128 * DN_LD) * sizeof(float);
    const unsigned grid = (unsigned)((B + 31) / 32);
#define DCLL_DENSE_SEQ(R_)                                                                                              \
    do {                                                                                                                \
        if (hipFuncSetAttribute((const void *)k_dense_lif_seq<R_>, hipFuncAttributeMaxDynamicSharedMemorySize,          \
                                (int)lds_bytes) != hipSuccess) {                                                        \
            (void)hipGetLastError();                                                                                    \
            return fail(DCLL_ERR_LAUNCH, "k_dense_lif_seq: cannot reserve its LDS");                                    \
        }                                                                                                               \
        hipLaunchKernelGGL(k_dense_lif_seq<R_>, dim3(grid), dim3(256), lds_bytes, st, d->in_features, d->out_features,  \
                           x, W, b, alpha, tau_m, alphas, tau_s, d->tau_is_tensor, eps0, eps1, arp, out_s, out_pv,     \
                           out_v, T, B, d->alpharp, d->wrp);                                                            \
    } while (0)
    if (d->refractory) DCLL_DENSE_SEQ(true);
    else DCLL_DENSE_SEQ(false);
#undef DCLL_DENSE_SEQ
    HIP_CHECK_LAUNCH("k_dense_lif_seq");
    return DCLL_OK;
}
