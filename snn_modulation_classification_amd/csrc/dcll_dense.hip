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

constexpr int DN_BT = 128, DN_KC = 32, DN_LD = 34;

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

// NTW = neuron tiles per wave: 2 (one A fragment feeds two MFMAs; workgroup = 128 samples x 64 neurons) or 1 (128 x 32: twice
// the workgroups — chosen when the launch would otherwise put fewer than two workgroups = two waves per SIMD on a CU: with one
// wave per SIMD nothing covers its barrier and its LDS round trip, 58 % of the MFMA peak at in = 8192, out = 512, B = 4096)
template <bool REFRACTORY, int NTW>
__global__ __launch_bounds__(256) void k_dense_lif_mfma(int in, int out, const float *__restrict__ eps1,
                                                         const float *__restrict__ W, const float *__restrict__ bias,
                                                         float *__restrict__ arp, float *__restrict__ s_out,
                                                         float *__restrict__ pv_out, float *__restrict__ v_out, int B,
                                                         float alpharp, float wrp)
{
    // two LDS buffers and two register sets (round 4): chunk c is computed from buffer c & 1 while chunk c + 1 is written into
    // the other buffer and chunk c + 3 is requested — one barrier per chunk, and a request has two chunks of MFMAs (4k cycles)
    // to land.  (At in = 8192, out = 512, B = 4096 the launch is ONE workgroup per CU: no other workgroup hides the latency.)
    constexpr int DN_OT = 32 * NTW;
    __shared__ __attribute__((aligned(16))) float sE[2][DN_BT * DN_LD];
    __shared__ __attribute__((aligned(16))) float sW[2][DN_OT * DN_LD];
    const int tid = threadIdx.x, lane = tid & 63, jj = lane & 31, kk = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware tile order (round 4).  Workgroup L of a launch goes to XCD L % 8 (each XCD has its own 4 MB L2).  With the
    // plain order (neuron tile fastest) an XCD gets ONE neuron tile of every sample block: every block of eps1 (the large
    // operand: 128 x in floats) is fetched by all eight XCDs — 8 x the state through the fabric, 1.2 GB per step at in = 8192,
    // B = 4096.  Here the workgroups of one XCD walk ALL neuron tiles of a sample block before the next block: eps1 crosses
    // once (its other seven readers hit that XCD's L2), only W (small) is read by every XCD.
    const int gx = (out + DN_OT - 1) / DN_OT, gy = (B + DN_BT - 1) / DN_BT;
    const int L = blockIdx.x;
    int tx, ty;
    if (gy % 8 == 0) {
        const int xcd = L & 7, slot = L >> 3;
        tx = slot % gx;
        ty = (slot / gx) * 8 + xcd;
    } else {
        tx = L % gx;
        ty = L / gx;
    }
    const long b0 = (long)ty * DN_BT;
    const int o0 = tx * DN_OT;
    const int r = tid >> 3, c4 = (tid & 7) * 4;             // staging: 8 threads x float4 = one 32-float chunk of a row
    const bool vec = (in % 4 == 0) && ((((uintptr_t)eps1 | (uintptr_t)W) & 15) == 0);
    f32x4 re[2][4], rw[2][NTW];
    // Whole 32-feature chunks of 16-byte aligned rows are fetched with BUFFER loads (round 4): descriptor = this workgroup's
    // 128 samples of eps1 (resp. its 64 rows of W), record count = the rows that exist (a row past the end reads as zeros),
    // lane offset = (row, 4 floats), scalar offset = the chunk — no vector instruction, no branch.  (The general fetch
    // below it is three exec-mask branches and ~20 address / select instructions per float4: ~120 vector instructions and
    // ~40 branches per chunk of 32 MFMAs, on the pipe the MFMAs execute on — it now serves only a ragged last chunk and
    // unaligned operands.)
    const int kfast = (vec && in < (1 << 22)) ? (in & ~(DN_KC - 1)) : 0;
    const long nb = B - b0 < DN_BT ? B - b0 : DN_BT;
    const int no = out - o0 < DN_OT ? out - o0 : DN_OT;
    const auto ers = __builtin_amdgcn_make_buffer_rsrc((void *)(eps1 + b0 * in), 0, (int)(nb * in * 4), 0x00020000);
    const auto wrs = __builtin_amdgcn_make_buffer_rsrc((void *)(W + (long)o0 * in), 0, no * in * 4, 0x00020000);
    unsigned evo[4], wvo[NTW];
#pragma unroll
    for (int q = 0; q < 4; ++q) evo[q] = 4u * (unsigned)((r + 32 * q) * in + c4);
#pragma unroll
    for (int q = 0; q < NTW; ++q) wvo[q] = 4u * (unsigned)((r + 32 * q) * in + c4);
    auto fetch = [&](auto setc, int k0) {
        constexpr int S = decltype(setc)::value;
        if (k0 >= in) return;
        if (k0 < kfast) {
#pragma unroll
            for (int q = 0; q < 4; ++q) re[S][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ers, evo[q], 4u * k0, 0));
#pragma unroll
            for (int q = 0; q < NTW; ++q) rw[S][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo[q], 4u * k0, 0));
            return;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) re[S][q] = dn_load4(eps1, b0 + r + 32 * q, B, in, k0 + c4, vec);
#pragma unroll
        for (int q = 0; q < NTW; ++q) rw[S][q] = dn_load4(W, o0 + r + 32 * q, out, in, k0 + c4, vec);
    };
    auto store = [&](auto setc) {                           // register set S -> LDS buffer S
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float *dst = sE[S] + (r + 32 * q) * DN_LD + c4;
            *(f32x2 *)dst = f32x2{re[S][q][0], re[S][q][1]};
            *(f32x2 *)(dst + 2) = f32x2{re[S][q][2], re[S][q][3]};
        }
#pragma unroll
        for (int q = 0; q < NTW; ++q) {
            float *dst = sW[S] + (r + 32 * q) * DN_LD + c4;
            *(f32x2 *)dst = f32x2{rw[S][q][0], rw[S][q][1]};
            *(f32x2 *)(dst + 2) = f32x2{rw[S][q][2], rw[S][q][3]};
        }
    };
    f32x16 acc0, acc1;                                      // chains start from the bias (lane = output neuron)
    {
        const float bz0 = (bias && o0 + jj < out) ? bias[o0 + jj] : 0.0f;
        const float bz1 = (NTW == 2 && bias && o0 + 32 + jj < out) ? bias[o0 + 32 + jj] : 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc0[q] = bz0; acc1[q] = bz1; }
    }
    auto compute = [&](auto bufc) {
        constexpr int S = decltype(bufc)::value;
        const float *ea = sE[S] + (32 * w + jj) * DN_LD + kk, *wb0 = sW[S] + jj * DN_LD + kk, *wb1 = sW[S] + (32 * (NTW - 1) + jj) * DN_LD + kk;
#pragma unroll
        for (int s = 0; s < DN_KC / 2; ++s) {
            const float a = ea[2 * s];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb0[2 * s], acc0, 0, 0, 0);
            if (NTW == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb1[2 * s], acc1, 0, 0, 0);
        }
    };
    const std::integral_constant<int, 0> S0;
    const std::integral_constant<int, 1> S1;
    fetch(S0, 0);
    fetch(S1, DN_KC);
    store(S0);
    fetch(S0, 2 * DN_KC);
    __syncthreads();
    for (int k0 = 0; k0 < in; k0 += 2 * DN_KC) {
        // chunk k0 from buffer 0; chunk k0 + 32 (register set 1, requested two chunks ago) -> buffer 1; request chunk k0 + 96
        if (k0 + DN_KC < in) store(S1);
        fetch(S1, k0 + 3 * DN_KC);
        compute(S0);
        __syncthreads();
        if (k0 + DN_KC >= in) break;
        if (k0 + 2 * DN_KC < in) store(S0);
        fetch(S0, k0 + 4 * DN_KC);
        compute(S1);
        __syncthreads();
    }
    // D layout: register q of lane (jj, kk) = sample (q & 3) + 8 (q >> 2) + 4 kk of the wave's 32, neuron jj of the tile
#pragma unroll
    for (int tl = 0; tl < NTW; ++tl) {
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
    const long gy = (B + DN_BT - 1) / DN_BT;
    const bool narrow = ((d->out_features + 63) / 64) * gy < 512;       // fewer than two 128 x 64 workgroups per CU
    const long ntile = (long)((d->out_features + (narrow ? 31 : 63)) / (narrow ? 32 : 64)) * gy;
    if (ntile > 0x7fffffffL) return fail(DCLL_ERR_UNSUPPORTED, "dense layer: more than 2^31 output tiles per call");
    const dim3 g((unsigned)ntile);
#define DCLL_DENSE(R_, N_) hipLaunchKernelGGL((k_dense_lif_mfma<R_, N_>), g, dim3(256), 0, st, d->in_features, d->out_features, eps1, W,  \
                                              b, arp, out_s, out_pv, out_v, B, d->alpharp, d->wrp)
    if (d->refractory) { if (narrow) DCLL_DENSE(true, 1); else DCLL_DENSE(true, 2); }
    else { if (narrow) DCLL_DENSE(false, 1); else DCLL_DENSE(false, 2); }
#undef DCLL_DENSE
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
