// dcll_internal.h — what the translation units of libdcll_hip.so share: vector typedefs, the geometry constants of
// the LDS-resident layouts, the error plumbing of the C ABI and the device helpers that define the pinned arithmetic
// (include/dcll_hip.h).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/dcll_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// geometry of the LDS-resident 32-channel 16x16 eps1 image shared by k_lif_seq_c32 and k_bwd_wgrad_c32
constexpr int NWAVE = 8, CPW = 4, ROWF = 19, CHF = 361;
constexpr int IMG_FLOATS = ((32 * CHF + 3 * ROWF + 3 + 61) + 3) & ~3;     // 11676 >= offset of (ci=31, y=18, x=18) + 1
constexpr int SLOT_FLOATS = 16 * 64;

// ------------------------------------------------------------------------------------------------------------
// error plumbing (the message buffer lives in dcll_hip.hip; one per thread)
// ------------------------------------------------------------------------------------------------------------
__attribute__((visibility("hidden"))) char *dcll_err_buf(void);
constexpr size_t DCLL_ERR_LEN = 512;

static inline int fail(int code, const char *msg, const char *who = nullptr)
{
    if (who) snprintf(dcll_err_buf(), DCLL_ERR_LEN, "%s: %s", who, msg);
    else snprintf(dcll_err_buf(), DCLL_ERR_LEN, "%s", msg);
    return code;
}

// dcll_kernel_trace (ABI 5): while a thread records, every launch check notes the kernel's name (dcll_hip.hip)
__attribute__((visibility("hidden"))) void dcll_trace_note(const char *name);

#define HIP_CHECK_LAUNCH(name)                                                              \
    do {                                                                                    \
        dcll_trace_note(name);                                                              \
        hipError_t e_ = hipGetLastError();                                                  \
        if (e_ != hipSuccess) {                                                             \
            snprintf(dcll_err_buf(), DCLL_ERR_LEN, "%s: %s", name, hipGetErrorString(e_));  \
            return DCLL_ERR_LAUNCH;                                                         \
        }                                                                                   \
    } while (0)

// ------------------------------------------------------------------------------------------------------------
// conv weights as a kernel sees them: fp32, or int8 + one fp32 scale per output channel (dcll_layer_opts, ABI v3).
// at(idx, co) is the ONE place a weight is converted: (float)q * scale[co], a single rounded multiply — bit for bit what
// the dequantised fp32 tensor holds, so every chain downstream is unchanged.  The branch is wave-uniform.
// ------------------------------------------------------------------------------------------------------------
// thresholds of sample b of the IQ quantiser: the scalar-path table for the samples the mask marks (dcll_iq_tail)
__device__ __forceinline__ void iq_tables(const float *thr_i, const float *thr_q, const dcll_iq_tail &tail, long b,
                                          const float *&ti, const float *&tq)
{
    const bool tl = tail.tail_mask && tail.tail_mask[b];
    ti = tl ? tail.thr_i_tail : thr_i;
    tq = tl ? tail.thr_q_tail : thr_q;
}
static inline dcll_iq_tail make_iq_tail(const dcll_iq_tail *t) { return t ? *t : dcll_iq_tail{nullptr, nullptr, nullptr}; }

struct dcll_wsrc {
    const float *f;
    const int8_t *q;
    const float *scale;
    __device__ __forceinline__ float at(long idx, int co) const { return q ? (float)q[idx] * scale[co] : f[idx]; }
};
// the 2 x 49 stationary A fragments of a 32 -> 32 7x7 sequence kernel: W[j][4 w + 2 cp + h][tap]; ONE wave-uniform branch
// on the weight format around the whole load (not one per weight)
__device__ __forceinline__ void load_wf_c32(const dcll_wsrc &W, int j, int w, int h, float (&wf)[2][49])
{
    if (W.q) {
        const float sc = W.scale[j];
#pragma unroll
        for (int cp = 0; cp < 2; ++cp)
#pragma unroll
            for (int k = 0; k < 49; ++k) wf[cp][k] = (float)W.q[((long)j * 32 + 4 * w + 2 * cp + h) * 49 + k] * sc;
    } else {
#pragma unroll
        for (int cp = 0; cp < 2; ++cp)
#pragma unroll
            for (int k = 0; k < 49; ++k) wf[cp][k] = W.f[((long)j * 32 + 4 * w + 2 * cp + h) * 49 + k];
    }
}

static inline dcll_wsrc make_wsrc(const float *W, const dcll_layer_opts *o)
{
    return (o && o->w_q8) ? dcll_wsrc{nullptr, o->w_q8, o->w_scale} : dcll_wsrc{W, nullptr, nullptr};
}
// opts of a call checked once: -> DCLL_OK, or the failure code with the message set
static inline int check_opts(const float *W, const dcll_layer_opts *o, bool allow_presig, const char *who)
{
    if (!o) return W ? DCLL_OK : fail(DCLL_ERR_INVALID, "null weight pointer", who);
    if (o->reserved != 0) return fail(DCLL_ERR_INVALID, "dcll_layer_opts.reserved must be 0", who);
    if (o->w_q8 ? !o->w_scale : !W) return fail(DCLL_ERR_INVALID, "weights: need W (fp32) or w_q8 + w_scale (int8)", who);
    if (o->pv_presigmoid && !allow_presig) return fail(DCLL_ERR_INVALID, "pv_presigmoid is an option of the sequence calls", who);
    return DCLL_OK;
}

// k_lif_seq_c32t (dcll_seq_tiled.hip): the 32 -> 32 channel sequence layer on planes with h % 8 == 0, w % 32 == 0
__attribute__((visibility("hidden")))
int dcll_launch_seq_c32t(const dcll_conv_desc *d, const uint32_t *spk_in, dcll_wsrc W, const float *b,
                         const float *tau4, float *eps0, float *eps1, float *arp, uint32_t *spk_out, float *pv_out,
                         float *v_out, float *state_scratch, int32_t T, int32_t B, hipStream_t st);
// k_lif_seq_c1t: the first layer (c_in 1) on such planes, input as cell indices or raw IQ
__attribute__((visibility("hidden")))
int dcll_launch_seq_c1t(const dcll_conv_desc *d, const int32_t *cells, const float *iq, const float *thr_i,
                        const float *thr_q, dcll_iq_tail tail, int L, int t0, dcll_wsrc W, const float *b, const float *tau4,
                        float *eps0, float *eps1, float *arp, uint32_t *spk_out, float *pv_out, float *v_out,
                        float *state_scratch, int T, int B, hipStream_t st, bool presig = false);

// k_lif_seq_w3 (dcll_seq_w3.hip): the (1,3)-kernel / 64-channel / (1,2)-pool layers of radio_ml_conv_ref.yaml, all T steps
__attribute__((visibility("hidden"))) bool dcll_seq_w3_geometry(const dcll_conv_desc *d);
__attribute__((visibility("hidden")))
int dcll_launch_seq_w3(const dcll_conv_desc *d, const uint32_t *spk_in, const int32_t *cells, dcll_wsrc W, const float *b,
                       const float *tau4, float *eps0, float *eps1, float *arp, uint32_t *spk_out, float *pv_out,
                       float *v_out, bool presigmoid, int32_t T, int32_t B, hipStream_t st);

// dense twins (dcll_dense.hip): one step as an fp32-MFMA GEMM on the updated traces; all T steps with the state on chip
__attribute__((visibility("hidden")))
int dcll_launch_dense_mfma(const dcll_dense_desc *d, const float *eps1, const float *W, const float *b, float *arp,
                           float *out_s, float *out_pv, float *out_v, int32_t B, hipStream_t st);
__attribute__((visibility("hidden"))) bool dcll_dense_seq_fits(const dcll_dense_desc *d);
__attribute__((visibility("hidden")))
int dcll_launch_dense_seq(const dcll_dense_desc *d, const float *x, const float *W, const float *b, const float *alpha,
                          const float *tau_m, const float *alphas, const float *tau_s, float *eps0, float *eps1, float *arp,
                          float *out_s, float *out_pv, float *out_v, int32_t T, int32_t B, hipStream_t st);

// k_readout_direct (dcll_readout.hip): LDS-free 16x16x4 readout GEMM in <= 64 VGPRs (fits beside a sequence kernel)
__attribute__((visibility("hidden")))
int dcll_launch_readout_direct(const float *pv, const float *Wt, const float *bias, float *out, long rows, int K, int N,
                               hipStream_t st);

// k_readout_t16 (dcll_readout.hip): LDS-staged 16x16x4 readout GEMM, 128 rows x 16 NT readout rows per workgroup
__attribute__((visibility("hidden")))
int dcll_launch_readout_t16(const float *pv, const float *Wt, const float *bias, float *out, long rows, int K, int N,
                            int kslice, hipStream_t st, int act = DCLL_ACT_NONE);

// the split-K passes of several per-step readouts in one launch (k_readout_t16m)
__attribute__((visibility("hidden")))
int dcll_launch_readout_t16_multi(const float *const *pv, const float *const *Wt, float *const *out, const long *rows,
                                  const int *K, const int *N, const int *kslice, int n, hipStream_t st);

// ------------------------------------------------------------------------------------------------------------
// shared device helpers
// ------------------------------------------------------------------------------------------------------------
// pv = 1/(1+exp(-v)): v_exp_f32 + v_rcp_f32 (each ~1 ulp); pv is not bit-pinned (include/dcll_hip.h), |err| ~1e-7.
__device__ __forceinline__ float sigmoidf_dev(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// dcll/pytorch_libdcll.py:493-494 — (x*tau_s) + (alphas*eps0) ; (alpha*eps1) + (eps0'*tau_m); every op rounded.
__device__ __forceinline__ void trace_update(float x, float alpha, float tau_m, float alphas, float tau_s,
                                             float &e0, float &e1)
{
    float a = x * tau_s;
    float b = alphas * e0;
    e0 = a + b;
    float c = alpha * e1;
    float d = e0 * tau_m;
    e1 = c + d;
}

// :497-503 — returns v, updates arp, sets s.
__device__ __forceinline__ float refractory(float pvmem, float &arp, float alpharp, float wrp, bool &s)
{
    float a = alpharp * arp;
    float v = pvmem + a;
    s = v > 0.0f;
    float sw = s ? wrp : 0.0f;      // s*wrp, exact
    arp = a - sw;
    return v;
}

// (v > 0 ? 1.0f : 0.0f) on a register pair in TWO packed instructions: clamp(clamp(v * 2^127) * 2^127), clamp = the VOP3P
// clamp to [0, 1].  Exact for every fp32 value a membrane can take — a positive denormal reaches 2^-22 after the first
// multiplication and 1 after the second (fp32 denormals are on in this build), zeros and negatives clamp to +0
// (experiments/pk_clamp_probe.hip checks every class on the device).  Replaces 2 v_cmp + 2 v_cndmask in the refractory
// update s * wrp of the epilogues that share their vector pipe with fp32 MFMAs.
__device__ __forceinline__ f32x2 spike01_pk(f32x2 v)
{
    const f32x2 big = {0x1p127f, 0x1p127f};          // (an SGPR pair: the kernels that use this have no vector register to spare)
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(v), "s"(big));
    asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(r), "s"(big));
    return r;
}

// Buffer-store addressing for the epilogues of the sequence kernels: a 128-bit descriptor (base = a wave-uniform pointer, in
// SGPRs) + 32-bit scalar or immediate offset + 32-bit lane byte offset — no 64-bit address per store and no address
// register pairs to keep (or spill) across the time loop.  Offsets must stay below 2^31.
__device__ __forceinline__ long uniform_long(long x)
{
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)x), hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long)x >> 32));
    return (long)(((unsigned long)hi << 32) | lo);
}
__device__ __forceinline__ auto tile_rsrc(const void *base)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)uniform_long((long)base), 0, 0x7fffffff, 0x00020000);
}

// compile-time loop: f(std::integral_constant<int, R0>{}), ..., f(std::integral_constant<int, R1 - 1>{})
template <int R0, int R1, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (R0 < R1) {
        f(std::integral_constant<int, R0>{});
        static_for<R0 + 1, R1>(f);
    }
}

// A 32-bit LDS address the compiler cannot fold into static offsets (after `asm volatile("" : "+v"(p))`): reads through it are
// base + immediate, where a visible static offset beyond the instruction's range makes the compiler rebuild a base per read.
typedef __attribute__((address_space(3))) const float lds_cfloat;

// Workgroup barrier that orders LDS traffic only: waits for this wave's outstanding LDS (and scalar) operations, then
// s_barrier — without the s_waitcnt vmcnt(0) that __syncthreads() adds for its global-memory fence, so a wave does not
// stall on its own pv / spike stores.  The builtin keeps the barrier convergent for the compiler.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// one DPP step of a wave-wide sum: v + (v moved by CTRL); lanes without a source add 0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v)
{
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
// sum over the 64 lanes; the total is valid in lane 63
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    v = dpp_add<0xB1, 0xF>(v);       // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);       // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xF>(v);      // row_half_mirror
    v = dpp_add<0x140, 0xF>(v);      // row_mirror      -> every lane holds its 16-lane row sum
    v = dpp_add<0x142, 0xA>(v);      // row_bcast15     -> rows 1,3 += previous row
    v = dpp_add<0x143, 0xC>(v);      // row_bcast31     -> rows 2,3 += row 1
    return v;
}

