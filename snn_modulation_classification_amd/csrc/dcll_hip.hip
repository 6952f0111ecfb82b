// dcll_hip.hip — hand-written gfx950 (MI355X, CDNA4) kernels + the C ABI of include/dcll_hip.h.
//
// Replaces, for the DCLL hot path of ohjay/snn-modulation-classification:
//   Conv2dDCLLlayer.forward          dcll/pytorch_libdcll.py:599-608
//   ContinuousConv2D.forward         dcll/pytorch_libdcll.py:407-426
//   ContinuousRelativeRefractoryConv2D.forward   dcll/pytorch_libdcll.py:485-509
//   DenseDCLLlayer.forward / CLLDense*Module.forward   :250-255, :131-148, :171-195
//   DCLLClassification.forward (argmax per step) + get_predictions_by_vote   :722-729, :44-56
//   iq2spiketrain cell quantisation  data/utils.py:60-82
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see csrc/Makefile).
// -ffp-contract=off is part of the arithmetic contract: the trace lines are three separately rounded ops.
//
// Kernels
//   k_lif_seq_c32d     the hot kernel (T >= 8): k_lif_seq_c32 with two pixel tiles per wave and stage — see its header.
//   k_lif_seq_c32      one 32->32 7x7 layer, ALL T timesteps, one sample per workgroup (short sequences, per-step calls).
//                      8 waves; wave w owns input channels 4w..4w+3 (a K-slice of the implicit GEMM):
//                      their eps0/eps1 traces (registers + a zero-padded LDS image) and the 2x49 weight
//                      fragments of v_mfma_f32_32x32x2_f32 (weight-stationary in VGPRs).  The fp32
//                      accumulator of a 32-channel x 32-pixel tile is handed from wave w to wave w+1 through LDS,
//                      so the result is ONE fmaf chain in the pinned order of include/dcll_hip.h; waves run
//                      skewed by one tile (a systolic chain), one s_barrier per tile-stage.
//   k_lif_seq_c1       first layer (c_in = 1, one input spike per step as a cell index or raw IQ): fp32 MFMA with the
//                      49 taps in 25 k-pairs (only tap 49 is a zero-weight pad), weight-stationary, one sample per workgroup.
//   k_lif_step_c32     ONE step of a 32->32 7x7 layer on the 16x16 plane (per-step drop-in, learning forward): no
//                      systolic hand-off, every wave runs two whole chains, weights stream through LDS; every global
//                      access a buffer access (descriptor + lane offset + scalar / immediate offset).
//   k_trace / k_conv_lif_tiled / k_conv_lif / k_pool   generic per-step path (any geometry, state in HBM) — the exact
//                      drop-in for `.forward`; same pinned order.
//   (dense twins: k_dense_lif_mfma / k_dense_lif_seq in dcll_dense.hip.)
//   k_bwd_dv[_nopool], k_bwd_wgrad_c32 (MFMA, 16x16 plane or 16x16 tiles with halo), k_bwd_wgrad (any geometry, row
//                      bands), k_bwd_reduce[4], k_bwd_outgrad[_part/_reduce]   backward of one layer step (local learning).
//   k_readout_v4 / k_readout_ks / k_readout_rows / k_readout (+ k_readout_sum)   fp32-MFMA GEMMs for i2o / output_:
//                      many rows, long rows (also split over K with caller scratch), few rows, any shape.
//   k_argmax, k_vote   per-step argmax and vote.
//   k_iq_encode, k_pack, k_unpack, k_permute_readout     glue.
// The kernels for planes larger than 16x16 (k_lif_seq_c32t, k_lif_seq_c1t) live in dcll_seq_tiled.hip.
#include "dcll_internal.h"

static thread_local char g_err[DCLL_ERR_LEN] = "";
char *dcll_err_buf(void) { return g_err; }

extern "C" int dcll_version(void) { return DCLL_ABI_VERSION; }
extern "C" const char *dcll_last_error(void) { return g_err; }

// dcll_kernel_trace: which kernels did this thread's calls dispatch?  Host-side bookkeeping only (a name per launch
// check, newline separated, while recording); the log stops growing at DCLL_TRACE_LEN and says so ("...").
constexpr size_t DCLL_TRACE_LEN = 1 << 16;
static thread_local bool g_trace_on = false;
static thread_local size_t g_trace_len = 0;
static thread_local char *g_trace = nullptr;
void dcll_trace_note(const char *name)
{
    if (!g_trace_on || !g_trace) return;
    const size_t n = strlen(name);
    if (g_trace_len + n + 5 >= DCLL_TRACE_LEN) {
        if (g_trace_len + 4 < DCLL_TRACE_LEN && (g_trace_len < 4 || memcmp(g_trace + g_trace_len - 4, "...\n", 4) != 0)) {
            memcpy(g_trace + g_trace_len, "...\n", 4);
            g_trace_len += 4;
        }
        return;
    }
    memcpy(g_trace + g_trace_len, name, n);
    g_trace[g_trace_len + n] = '\n';
    g_trace_len += n + 1;
}
extern "C" int dcll_kernel_trace(int32_t enable)
{
    if (enable) {
        if (!g_trace) g_trace = (char *)malloc(DCLL_TRACE_LEN);
        if (!g_trace) return fail(DCLL_ERR_INVALID, "dcll_kernel_trace: out of host memory");
        g_trace_len = 0;
    }
    g_trace_on = enable != 0;
    return DCLL_OK;
}
extern "C" int64_t dcll_kernel_trace_read(char *buf, int64_t cap)
{
    if (buf && cap > 0) {
        const size_t n = g_trace_len < (size_t)(cap - 1) ? g_trace_len : (size_t)(cap - 1);
        if (n) memcpy(buf, g_trace, n);
        buf[n] = 0;
    }
    return (int64_t)g_trace_len + 1;
}

// dcll_conv_lif_step on the 16x16 plane: batches up to this many samples run two workgroups per sample (8-row tiles of
// k_lif_step_c32t), larger ones k_lif_step_c32.  DCLL_SPLIT16_MAX_BATCH moves the switch (tests: both forms on one batch).
static inline int split16_max_batch(void)
{
    const char *e = getenv("DCLL_SPLIT16_MAX_BATCH");
    return e && *e ? atoi(e) : 256;
}

static inline void conv_shape(const dcll_conv_desc *d, int *ch, int *cw, int *ph, int *pw)
{
    // (reference get_output_shape, dcll/pytorch_libdcll.py:368-375; stride = dilation = 1: h + 2 pad - kh + 1)
    *ch = (d->h + 2 * d->pad_h - d->dilation * (d->kh - 1) - 1) / d->stride + 1;
    *cw = (d->w + 2 * d->pad_w - d->dilation * (d->kw - 1) - 1) / d->stride + 1;
    *ph = (*ch + 2 * ((d->pool_h - 1) / 2) - d->pool_h) / d->pool_h + 1;
    *pw = (*cw + 2 * ((d->pool_w - 1) / 2) - d->pool_w) / d->pool_w + 1;
}

// the specialised kernels (MFMA step / sequence / weight-gradient kernels, the tiled VALU kernel) are plain convolutions
static inline bool plain_conv(const dcll_conv_desc *d) { return d->stride == 1 && d->dilation == 1 && d->groups == 1; }

static int check_desc(const dcll_conv_desc *d)
{
    if (!d) return fail(DCLL_ERR_INVALID, "null descriptor");
    if (d->c_in < 1 || d->c_out < 1 || d->h < 1 || d->w < 1 || d->kh < 1 || d->kw < 1 || d->pad_h < 0 ||
        d->pad_w < 0 || d->pool_h < 1 || d->pool_w < 1 || d->target < 0)
        return fail(DCLL_ERR_INVALID, "descriptor has a non-positive dimension");
    // stride / dilation / groups other than 1 (F.conv2d's, reference :417 / :495): served by the generic per-step kernels
    // (k_conv_lif, k_bwd_wgrad) only — ConvNetwork never builds such a layer
    if (d->stride < 1 || d->dilation < 1 || d->groups < 1 || d->c_in % d->groups != 0 || d->c_out % d->groups != 0)
        return fail(DCLL_ERR_INVALID, "stride / dilation / groups must be >= 1 and groups must divide c_in and c_out");
    if (d->h + 2 * d->pad_h < d->dilation * (d->kh - 1) + 1 || d->w + 2 * d->pad_w < d->dilation * (d->kw - 1) + 1)
        return fail(DCLL_ERR_INVALID, "empty conv/pool output");
    int ch, cw, ph, pw;
    conv_shape(d, &ch, &cw, &ph, &pw);
    if (ch < 1 || cw < 1 || ph < 1 || pw < 1) return fail(DCLL_ERR_INVALID, "empty conv/pool output");
    return DCLL_OK;
}

extern "C" int dcll_conv_out_shape(const dcll_conv_desc *d, int32_t *ch, int32_t *cw, int32_t *ph, int32_t *pw)
{
    int rc = check_desc(d);
    if (rc) return rc;
    int a, b, c, e;
    conv_shape(d, &a, &b, &c, &e);
    if (ch) *ch = a;
    if (cw) *cw = b;
    if (ph) *ph = c;
    if (pw) *pw = e;
    return DCLL_OK;
}

// ------------------------------------------------------------------------------------------------------------
// generic per-step path (state in HBM, any geometry)
// ------------------------------------------------------------------------------------------------------------
__global__ void k_trace(const float *__restrict__ x, const float *__restrict__ alpha, const float *__restrict__ tau_m,
                        const float *__restrict__ alphas, const float *__restrict__ tau_s, float *__restrict__ eps0,
                        float *__restrict__ eps1, long n, long per_sample, int tau_is_tensor)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        long q = tau_is_tensor ? (i % per_sample) : 0;
        float e0 = eps0[i], e1 = eps1[i];
        trace_update(x[i], alpha[q], tau_m[q], alphas[q], tau_s[q], e0, e1);
        eps0[i] = e0;
        eps1[i] = e1;
    }
}

// one thread per conv output element (b, co, y, x); pinned fmaf chain (cp, ky, kx, h).
__global__ void k_conv_lif(dcll_conv_desc d, int ch, int cw, const float *__restrict__ eps1,
                           const dcll_wsrc W, const float *__restrict__ bias, float *__restrict__ arp,
                           float *__restrict__ s_full, float *__restrict__ pv_full, float *__restrict__ v_out, long n)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int xx = (int)(i % cw);
    long r = i / cw;
    int y = (int)(r % ch);
    r /= ch;
    int co = (int)(r % d.c_out);
    long b = r / d.c_out;
    // groups (F.conv2d): output channel co belongs to group co / (c_out / groups) and sees that group's cig = c_in / groups
    // input channels; the pinned chain runs over the group's channel pairs.  stride / dilation: tap (ky, kx) of output (y, x)
    // reads input (y * stride + ky * dilation - pad, ...).  With all three at 1 this is the plain convolution, bit for bit.
    const int cig = d.c_in / d.groups, grp = co / (d.c_out / d.groups);
    const float *e = eps1 + (b * d.c_in + (long)grp * cig) * d.h * d.w;
    const long wbase = (long)co * cig * d.kh * d.kw;
    float acc = bias ? bias[co] : 0.0f;
    const int npair = (cig + 1) >> 1;
    for (int cp = 0; cp < npair; ++cp)
        for (int ky = 0; ky < d.kh; ++ky) {
            int yy = y * d.stride + ky * d.dilation - d.pad_h;
            bool yin = yy >= 0 && yy < d.h;
            for (int kx = 0; kx < d.kw; ++kx) {
                int xq = xx * d.stride + kx * d.dilation - d.pad_w;
                bool in = yin && xq >= 0 && xq < d.w;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    int ci = 2 * cp + hh;
                    if (ci < cig) {
                        float ev = in ? e[((long)ci * d.h + yy) * d.w + xq] : 0.0f;
                        acc = __builtin_fmaf(ev, W.at(wbase + ((long)ci * d.kh + ky) * d.kw + kx, co), acc);
                    }
                }
            }
        }
    float v = acc;
    bool s;
    if (d.refractory) {
        float a = arp[i];
        v = refractory(acc, a, d.alpharp, d.wrp, s);
        arp[i] = a;
    } else {
        s = v > 0.0f;
    }
    if (v_out) v_out[i] = v;
    s_full[i] = s ? 1.0f : 0.0f;
    pv_full[i] = sigmoidf_dev(v);
}

// Tiled version of k_conv_lif for the common kernel sizes: a workgroup computes a 16x16 tile of output pixels for COG
// output channels.  Per input-channel pair the (16+KH-1)x(16+KW-1) input tiles and the COG x KH*KW x 2 weights are
// staged in LDS; a thread keeps the two KH x KW windows of its pixel in registers and runs COG independent fmaf
// chains in the pinned order (cp, ky, kx, h) — bit-identical to k_conv_lif, ~100x faster (weights come as LDS
// broadcasts, inputs from registers).  Out-of-image taps and channels beyond c_in contribute fmaf(0, w, acc).
template <int KH, int KW, int COG>
__global__ __launch_bounds__(256) void k_conv_lif_tiled(dcll_conv_desc d, int ch, int cw, const float *__restrict__ eps1,
                                                         const dcll_wsrc W, const float *__restrict__ bias,
                                                         float *__restrict__ arp, float *__restrict__ s_full,
                                                         float *__restrict__ pv_full, float *__restrict__ v_out)
{
    constexpr int IH = 16 + KH - 1, IW = 16 + KW - 1, KK = KH * KW;
    __shared__ float in[2][IH * IW];
    __shared__ __attribute__((aligned(16))) float wl[COG * KK * 2];
    const int ntx = (cw + 15) >> 4;
    const int tx0 = (blockIdx.x % ntx) * 16, ty0 = (blockIdx.x / ntx) * 16;
    const int co0 = blockIdx.y * COG;
    const long b = blockIdx.z;
    const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4;
    const int oy = ty0 + ly, ox = tx0 + lx;
    float acc[COG];
#pragma unroll
    for (int c = 0; c < COG; ++c) acc[c] = (bias && co0 + c < d.c_out) ? bias[co0 + c] : 0.0f;
    const float *eb = eps1 + b * d.c_in * d.h * d.w;
    const int npair = (d.c_in + 1) >> 1;
    for (int cp = 0; cp < npair; ++cp) {
        __syncthreads();
        for (int i = tid; i < 2 * IH * IW; i += 256) {
            const int hh = i / (IH * IW), r = i % (IH * IW);
            const int yy = ty0 + r / IW - d.pad_h, xx = tx0 + r % IW - d.pad_w, ci = 2 * cp + hh;
            const bool ok = ci < d.c_in && yy >= 0 && yy < d.h && xx >= 0 && xx < d.w;
            in[hh][r] = ok ? eb[((long)ci * d.h + yy) * d.w + xx] : 0.0f;
        }
        for (int i = tid; i < COG * KK * 2; i += 256) {
            const int hh = i & 1, tap = (i >> 1) % KK, c = (i >> 1) / KK, ci = 2 * cp + hh;
            wl[i] = (co0 + c < d.c_out && ci < d.c_in) ? W.at(((long)(co0 + c) * d.c_in + ci) * KK + tap, co0 + c) : 0.0f;
        }
        __syncthreads();
        float win[2][KK];
#pragma unroll
        for (int ky = 0; ky < KH; ++ky)
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) {
                win[0][ky * KW + kx] = in[0][(ly + ky) * IW + lx + kx];
                win[1][ky * KW + kx] = in[1][(ly + ky) * IW + lx + kx];
            }
#pragma unroll
        for (int c = 0; c < COG; ++c) {
            const f32x2 *wp = (const f32x2 *)(wl + c * KK * 2);
#pragma unroll
            for (int tap = 0; tap < KK; ++tap) {
                const f32x2 w2 = wp[tap];
                acc[c] = __builtin_fmaf(win[0][tap], w2[0], acc[c]);
                acc[c] = __builtin_fmaf(win[1][tap], w2[1], acc[c]);
            }
        }
    }
    if (oy < ch && ox < cw) {
#pragma unroll
        for (int c = 0; c < COG; ++c) {
            if (co0 + c < d.c_out) {
                const long i = ((b * d.c_out + co0 + c) * ch + oy) * cw + ox;
                float v = acc[c];
                bool s;
                if (d.refractory) {
                    float a = arp[i];
                    v = refractory(acc[c], a, d.alpharp, d.wrp, s);
                    arp[i] = a;
                } else {
                    s = v > 0.0f;
                }
                if (v_out) v_out[i] = v;
                s_full[i] = s ? 1.0f : 0.0f;
                pv_full[i] = sigmoidf_dev(v);
            }
        }
    }
}

// MaxPool2d(kernel=stride=pool, padding=(pool-1)/2), one thread per pooled element.
__global__ void k_pool(dcll_conv_desc d, int ch, int cw, int ph, int pw, const float *__restrict__ s_full,
                       const float *__restrict__ pv_full, float *__restrict__ s_out, float *__restrict__ pv_out, long n)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int px = (int)(i % pw);
    long r = i / pw;
    int py = (int)(r % ph);
    long bc = r / ph;
    const float *sp = s_full + bc * ch * cw, *pp = pv_full + bc * ch * cw;
    float ms = -INFINITY, mp = -INFINITY;
    int y0 = py * d.pool_h - (d.pool_h - 1) / 2, x0 = px * d.pool_w - (d.pool_w - 1) / 2;
    for (int dy = 0; dy < d.pool_h; ++dy)
        for (int dx = 0; dx < d.pool_w; ++dx) {
            int yy = y0 + dy, xq = x0 + dx;
            if (yy < 0 || yy >= ch || xq < 0 || xq >= cw) continue;
            ms = fmaxf(ms, sp[yy * cw + xq]);
            mp = fmaxf(mp, pp[yy * cw + xq]);
        }
    s_out[i] = ms;
    pv_out[i] = mp;
}

// ------------------------------------------------------------------------------------------------------------
// backward of one Conv2dDCLLlayer step for local learning (DCLLBase.train_dcll, dcll/pytorch_libdcll.py:690-718).
// Only i2h.weight / i2h.bias (through pv -> i2o -> loss, and optionally through pv / pvmem directly) and
// output_.weight / output_.bias receive gradients: i2o is frozen (:570-571), the neuron state is detached (:504-507),
// spikes come from `>` and output_ sees flatten.detach() (:606).  Gradients are not bit-pinned (fp32 sums).
// ------------------------------------------------------------------------------------------------------------
// g_v_full[b,co,y,x] = [this element is its pool window's (first) maximum of pv] * (g_pv[b,co,py,px] +
//                      sum_n g_p[b,n] * i2o_W[n, flat(co,py,px)]) * pv*(1-pv)  +  g_v[b,co,y,x]
__global__ void k_bwd_dv(dcll_conv_desc d, int ch, int cw, int ph, int pw, const float *__restrict__ v,
                         const float *__restrict__ g_p, const float *__restrict__ g_pv, const float *__restrict__ g_v,
                         const float *__restrict__ i2o_W, float *__restrict__ gvf, long n)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int x = (int)(i % cw);
    long r = i / cw;
    int y = (int)(r % ch);
    r /= ch;
    int co = (int)(r % d.c_out);
    long b = r / d.c_out;
    const int pph = (d.pool_h - 1) / 2, ppw = (d.pool_w - 1) / 2;
    const int py = (y + pph) / d.pool_h, px = (x + ppw) / d.pool_w;
    float g = 0.0f;
    const float pv = sigmoidf_dev(v[i]);
    if (py < ph && px < pw) {
        // am I the first maximum of my window (row-major scan, strict >, like MaxPool2d's forward)?
        const float *vp = v + (b * d.c_out + co) * (long)ch * cw;
        const int y0 = py * d.pool_h - pph, x0 = px * d.pool_w - ppw;
        bool is_max = true;
        for (int dy = 0; dy < d.pool_h && is_max; ++dy)
            for (int dx = 0; dx < d.pool_w; ++dx) {
                int yy = y0 + dy, xx = x0 + dx;
                if (yy < 0 || yy >= ch || xx < 0 || xx >= cw || (yy == y && xx == x)) continue;
                float q = sigmoidf_dev(vp[yy * cw + xx]);
                bool before = (yy < y) || (yy == y && xx < x);
                if (q > pv || (before && q == pv)) { is_max = false; break; }
            }
        if (is_max) {
            const long pidx = ((b * d.c_out + co) * ph + py) * pw + px;
            if (g_pv) g = g_pv[pidx];
            if (g_p) {
                const int K = d.c_out * ph * pw;
                const int k = (co * ph + py) * pw + px;
                float acc = 0.0f;
                for (int nn = 0; nn < d.target; ++nn) acc = __builtin_fmaf(g_p[b * d.target + nn], i2o_W[(long)nn * K + k], acc);
                g += acc;
            }
        }
    }
    float out = g * pv * (1.0f - pv);
    if (g_v) out += g_v[i];
    gvf[i] = out;
}

// dW[co,ci,ky,kx] = sum_{b,y,x} gvf[b,co,y,x] * eps1[b,ci,y+ky-pad,x+kx-pad];  db[co] = sum gvf[b,co,:,:]
// Generic kernel: workgroup (co*ci, chunk) sums the samples b = chunk, chunk + nchunk, ... into the partial row
// part[chunk][co][ci*ntap + tap] (row length c_in*ntap + 1, the last entry = bias gradient); k_bwd_reduce adds the
// chunks in a fixed order.  thread = conv output position (strided), per-thread tap accumulators, LDS tree at the end.
constexpr int WG_MAXTAPS = 64;
__global__ __launch_bounds__(256) void k_bwd_wgrad(dcll_conv_desc d, int ch, int cw, const float *__restrict__ gvf,
                                                    const float *__restrict__ eps1, float *__restrict__ part, int B,
                                                    int RB)
{
    extern __shared__ float sm[];
    // (groups: output channel co sees the cig = c_in / groups input channels of its group; row length cig * ntap + 1)
    const int cig = d.c_in / d.groups;
    const int co = blockIdx.x / cig, cil = blockIdx.x % cig, ci = (co / (d.c_out / d.groups)) * cig + cil;
    const int WP = d.w + 2 * d.pad_w, ntap = d.kh * d.kw;
    // zero-padded band of the eps1 plane of (b, ci): the input rows RB output rows read = (RB - 1) stride + (kh - 1) dilation + 1
    const int band = (RB - 1) * d.stride + (d.kh - 1) * d.dilation + 1;
    float *e = sm;
    float *red = sm + band * WP;                  // 4 x (WG_MAXTAPS + 1) wave totals
    float acc[WG_MAXTAPS];
#pragma unroll
    for (int t = 0; t < WG_MAXTAPS; ++t) acc[t] = 0.0f;
    float accb = 0.0f;
    const long rowlen = (long)cig * ntap + 1;
    float *prow = part + ((long)blockIdx.y * d.c_out + co) * rowlen;
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        const float *ep = eps1 + ((long)b * d.c_in + ci) * d.h * d.w;
        const float *gp = gvf + ((long)b * d.c_out + co) * ch * cw;
        for (int y0 = 0; y0 < ch; y0 += RB) {         // bands of RB output rows (the whole plane when it fits LDS)
            const int rows = min(RB, ch - y0);
            __syncthreads();
            for (int i = threadIdx.x; i < ((rows - 1) * d.stride + (d.kh - 1) * d.dilation + 1) * WP; i += 256) {
                const int iy = y0 * d.stride + i / WP - d.pad_h, ix = i % WP - d.pad_w;
                e[i] = ((unsigned)iy < (unsigned)d.h && (unsigned)ix < (unsigned)d.w) ? ep[iy * d.w + ix] : 0.0f;
            }
            __syncthreads();
            for (int pos = threadIdx.x; pos < rows * cw; pos += 256) {
                const int yy = pos / cw, xx = pos % cw;
                const float g = gp[(y0 + yy) * cw + xx];
                accb += g;
                const float *eb = e + yy * d.stride * WP + xx * d.stride;
#pragma unroll
                for (int t = 0; t < WG_MAXTAPS; ++t)
                    if (t < ntap) acc[t] = __builtin_fmaf(g, eb[(t / d.kw) * d.dilation * WP + (t % d.kw) * d.dilation], acc[t]);
            }
        }
    }
    // reduce every tap (and the bias gradient) over the 256 threads: DPP tree inside each wave (fixed order), the
    // four wave totals through LDS — two barriers instead of nine per tap
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < WG_MAXTAPS; ++t) {
        if (t < ntap) {
            const float tot = wave_sum_to_lane63(acc[t]);
            if (lane == 63) red[wave * (WG_MAXTAPS + 1) + t] = tot;
        }
    }
    {
        const float tot = wave_sum_to_lane63(accb);
        if (lane == 63) red[wave * (WG_MAXTAPS + 1) + WG_MAXTAPS] = tot;
    }
    __syncthreads();
    if (threadIdx.x <= ntap) {
        const int t = threadIdx.x < ntap ? threadIdx.x : WG_MAXTAPS;
        const float tot = ((red[t] + red[(WG_MAXTAPS + 1) + t]) + red[2 * (WG_MAXTAPS + 1) + t]) + red[3 * (WG_MAXTAPS + 1) + t];
        if (threadIdx.x < ntap) prow[(long)cil * ntap + threadIdx.x] = tot;
        else if (cil == 0) prow[rowlen - 1] = tot;
    }
}

// Weight gradient of the 32 -> 32, 7x7, 16x16 layers as fp32 MFMA: dW[co][n] = sum_{b,pix} g[b,co,pix] * E[b][n][pix],
// n = ci*49 + tap (1568 columns = 49 tiles of 32), E = zero-padded eps1 (same compact LDS image as k_lif_seq_c32).
// One workgroup takes samples b = blockIdx.x, + gridDim.x, ...; per sample g (32 x 256, row stride 257 against bank
// conflicts) and the image are staged in LDS; wave w accumulates column tiles w, w+8, ... (7 x 16 accumulator
// registers) over the 128 pixel pairs: A[co][pixel pair], B[pixel pair][column] = one ds_read at an immediate offset
// from the column's lane base.  Partial sums go to part[workgroup][co][1568 + 1] (k_bwd_reduce adds them in order).
constexpr int WG32_GLD = 257;
#ifndef WG32_SPLIT2_MAX_BATCH
#define WG32_SPLIT2_MAX_BATCH 1024
#endif
// TILED = false: the 16x16 plane, one sample per job, image rows / channels share their zero padding (RF 19, CF 361).
// TILED = true: planes with h % 16 == 0 and w % 16 == 0, one 16x16 tile of a sample per job; the tile's 22x22 eps1
// region with its real halo (zero outside the plane) is staged per channel (RF 22, CF 484).
// SPLIT: workgroups per batch chunk (blockIdx.y), each with 6 / SPLIT of every wave's column tiles — small batches have
// fewer samples than the GPU has CUs, so a sample's 49 column tiles are spread over several workgroups (which all
// stage the same g and image; the columns they write are disjoint, the fixed-order reduce is unchanged).
// DBG (experiments/ablate_wgrad.hip only; 0 in the product): 1 = per-wave shader-clock totals of the phases (staging incl.
// barriers / bias sum / MFMA loop / partial-sum stores) written behind the partial sums; 2 = no MFMAs; 4 = no stores
template <int RF, int CF, bool TILED, int SPLIT = 1, int DBG = 0>
__global__ __launch_bounds__(512) void k_bwd_wgrad_c32(const float *__restrict__ gvf, const float *__restrict__ eps1,
                                                        float *__restrict__ part, int B, int H, int Wd)
{
    unsigned long long tA = 0, tStage = 0, tBias = 0, tMfma = 0, tEnd = 0, tEntry = 0;
    if (DBG & 1) tEntry = tA = __builtin_amdgcn_s_memtime();
    auto lap = [&](unsigned long long &acc) {
        if (DBG & 1) {
            const unsigned long long n = __builtin_amdgcn_s_memtime();
            acc += n - tA;
            tA = n;
        }
    };
    constexpr int IMG = TILED ? 32 * CF : IMG_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[IMG + 32 * WG32_GLD];
    float *img = lds, *gl = lds + IMG;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, j = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 49 column tiles over 8 waves: wave w owns tiles w, w + 8, ..., w + 40 (six) and ONE EIGHTH of tile 48 — the pixel
    // pairs 16w .. 16w+15 of every sample, summed over the waves in wave order at the end.  (Tile 48 whole on wave 0 made
    // its SIMD carry 13 tiles against 12: the slowest SIMD sets the time, +6 %.)
    constexpr int NQ = 6 / SPLIT;                             // whole tiles per wave in this workgroup
    static_assert(6 % SPLIT == 0, "SPLIT divides the six whole tiles of a wave");
    const int q0 = blockIdx.y * NQ;                           // my tiles: w + 8 (q0 + q), q < NQ
    const bool last48 = blockIdx.y == 0;                      // tile 48 belongs to the first workgroup of the chunk
    int bbase[NQ + 1];
#pragma unroll
    for (int q = 0; q < NQ + 1; ++q) {
        const int n = (q < NQ ? w + 8 * (q0 + q) : 48) * 32 + j;     // my column in tile q
        const int ci = n / 49, tap = n % 49;
        bbase[q] = ci * CF + (tap / 7) * RF + (tap % 7) + h;
    }
    f32x16 acc[NQ + 1];
#pragma unroll
    for (int q = 0; q < NQ + 1; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;
    float bsum = 0.0f;                                        // wave w, lanes: co = 4w + (lane>>4), 16 pixels per pass
    const int tpr = Wd >> 4, tps = (H >> 4) * tpr;            // 16x16 tiles per row / per sample
    const long njob = TILED ? (long)B * tps : B;
    const long HW = (long)H * Wd;
    // the next job's g and eps1 (16 + 16 floats per thread on the 16x16 plane, 16 + 32 for a tile with its halo: thread =
    // one position of the 22x22 region for all 32 channels) are fetched into registers while the MFMAs of this one run;
    // only the LDS copy and two barriers stay between the jobs of a workgroup
    constexpr int NPE = TILED ? 32 : 16;
    float pg[16], pe[NPE];
    auto fetch = [&](long job) {
        if (TILED) {
            const long b = job / tps;
            const int tile = (int)(job % tps), y0 = (tile / tpr) * 16, x0 = (tile % tpr) * 16;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int i = tid + 512 * k, c = i >> 8, p = i & 255;
                pg[k] = gvf[(b * 32 + c) * HW + (long)(y0 + (p >> 4)) * Wd + x0 + (p & 15)];
            }
            const int gy = y0 + tid / RF - 3, gx = x0 + tid % RF - 3;
            const bool in = tid < CF && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)Wd;
            const float *src = eps1 + b * 32 * HW + (long)gy * Wd + gx;
#pragma unroll
            for (int c = 0; c < NPE; ++c) pe[c] = in ? src[c * HW] : 0.0f;
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                pg[k] = gvf[job * 8192 + tid + 512 * k];
                pe[k] = eps1[job * 8192 + tid + 512 * k];
            }
        }
    };
    if ((long)blockIdx.x < njob) fetch(blockIdx.x);
    for (int i = tid; i < IMG; i += 512) img[i] = 0.0f;       // (under the first job's requests)
    for (long job = blockIdx.x; job < njob; job += gridDim.x) {
        __syncthreads();
        if (TILED) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int i = tid + 512 * k;
                gl[(i >> 8) * WG32_GLD + (i & 255)] = pg[k];
            }
            if (tid < CF) {
#pragma unroll
                for (int c = 0; c < NPE; ++c) img[c * CF + tid] = pe[c];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int i = tid + 512 * k, c = i >> 8, p = i & 255;
                gl[c * WG32_GLD + p] = pg[k];
                img[c * CF + ((p >> 4) + 3) * RF + (p & 15) + 3] = pe[k];
            }
        }
        __syncthreads();
        if (job + gridDim.x < njob) fetch(job + gridDim.x);
        lap(tStage);
        {   // bias gradient: co = 4w + lane/16, pixels lane%16 + 16*k
            const float *gr = gl + (4 * w + (lane >> 4)) * WG32_GLD + (lane & 15);
#pragma unroll
            for (int k = 0; k < 16; ++k) bsum += gr[16 * k];
        }
        lap(tBias);
        const float *ga = gl + j * WG32_GLD + h;              // A: co = j, pixel p + h
        // my six whole tiles over all 128 pixel pairs: branch-free (a scalar branch between two groups of MFMAs holds the MFMA
        // issue — the per-pair test for the tile-48 share that used to sit here cost the loop 14 % — and an if / else around
        // two copies of the loop makes the compiler merge the 112 accumulator registers with v_movs: 128 -> 175 us)
        // FULLY unrolled: every operand address is a per-lane base + an immediate.  With a runtime pixel-pair index the
        // compiler rebuilt the seven addresses of every pair with 12-14 vector adds — on the pipe the fp32 MFMAs execute
        // on, ~4.8 cycles each: 16-23k of a job's 128k cycles (experiments/ablate_wgrad.hip, round 4)
#pragma unroll
        for (int pp = 0; pp < 128; ++pp) {
            const int p = 2 * pp;
            const float a = ga[p];
            const int poff = (p >> 4) * RF + (p & 15);
            if (DBG & 2) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q][0] += a * img[bbase[q] + poff];
                continue;
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, img[bbase[q] + poff], acc[q], 0, 0, 0);
        }
        // my eighth of tile 48: pixel pairs 16w .. 16w+15 (one more chain of 16; all operands fetched up front)
        if (last48 && !(DBG & 2)) {
            float a48[16], b48[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int p = 2 * (16 * w + k);
                a48[k] = ga[p];
                b48[k] = img[bbase[NQ] + (p >> 4) * RF + (p & 15)];
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[NQ] = __builtin_amdgcn_mfma_f32_32x32x2f32(a48[k], b48[k], acc[NQ], 0, 0, 0);
        }
        if (DBG & 1) asm volatile("" ::"v"(acc[0][0]), "v"(acc[NQ - 1][15]));
        lap(tMfma);
    }
    if ((DBG & 4) && acc[0][0] != 12345.678f) {
        if ((DBG & 1) && lane == 0) {
            unsigned long long *dst = (unsigned long long *)(part + (long)gridDim.x * 32 * 1569) + ((long)blockIdx.x * 8 + w) * 6;
            dst[0] = tStage; dst[1] = tBias; dst[2] = tMfma; dst[3] = 0; dst[4] = __builtin_amdgcn_s_memtime() - tEntry; dst[5] = tEntry;
        }
        return;
    }
    float *pw = part + (long)blockIdx.x * 32 * 1569;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int n = (w + 8 * (q0 + q)) * 32 + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) pw[(long)((r & 3) + 8 * (r >> 2) + 4 * h) * 1569 + n] = acc[q][r];
    }
    if (!last48) return;                                      // (workgroup-uniform)
    // tile 48: the eight partial tiles through LDS (the staging area is free), added in wave order
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) lds[(w * 16 + r) * 64 + lane] = acc[NQ][r];
    __syncthreads();
    for (int e = tid; e < 16 * 64; e += 512) {
        const int l = e & 63, r = e >> 6;
        float tot = lds[r * 64 + l];
#pragma unroll
        for (int ww = 1; ww < 8; ++ww) tot += lds[(ww * 16 + r) * 64 + l];
        pw[(long)((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 1569 + 48 * 32 + (l & 31)] = tot;
    }
    // bias partial: reduce the 16 lanes of each co group
    bsum += __shfl_xor(bsum, 1); bsum += __shfl_xor(bsum, 2); bsum += __shfl_xor(bsum, 4); bsum += __shfl_xor(bsum, 8);
    if ((lane & 15) == 0) pw[(long)(4 * w + (lane >> 4)) * 1569 + 1568] = bsum;
    if ((DBG & 1) && lane == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        lap(tEnd);
        unsigned long long *dst = (unsigned long long *)(part + (long)gridDim.x * 32 * 1569) + ((long)blockIdx.x * 8 + w) * 6;
        dst[0] = tStage; dst[1] = tBias; dst[2] = tMfma; dst[3] = tEnd; dst[4] = tA - tEntry; dst[5] = tEntry;
    }
}

// Weight gradient of the FIRST layer (c_in = 1 -> 32 channels, 7x7, 16x16 plane) as fp32 MFMA, the one-channel sibling
// of k_bwd_wgrad_c32: dW[co][tap] = sum_{b,pix} g[b,co,pix] * E[b][pix + tap] — 49 columns = two 32-column tiles, one per
// wave of a 128-thread workgroup; per sample g (32 x 256, row stride 257) and the zero-padded plane (row stride ROWF) are
// staged in LDS.  Partial sums go to part[workgroup][co][49 + 1] (k_bwd_reduce adds them in order).  (The generic
// k_bwd_wgrad needs 78 us for this layer at B = 512: 0.4 GFLOP spread over 2048 workgroups with a 49-value tree each.)
// TILED: planes of several 16x16 tiles (h % 16 == 0, w % 16 == 0) — one tile of a sample per job, its 22x22 eps1 region
// with the real halo (zero outside the plane) staged at row stride 22 (the generic k_bwd_wgrad needs 637 us for this
// layer on the 128x128 plane at B = 64).
template <int RF, bool TILED>
__global__ __launch_bounds__(128) void k_bwd_wgrad_c1(const float *__restrict__ gvf, const float *__restrict__ eps1,
                                                       float *__restrict__ part, int B, int H, int Wd)
{
    constexpr int IMG = (21 * RF + 21 + 2 + 3) & ~3;            // >= offset of (row 21, column 21) + 1 + the k lane
    __shared__ __attribute__((aligned(16))) float lds[IMG + 32 * WG32_GLD];
    float *img = lds, *gl = lds + IMG;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, j = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < IMG; i += 128) img[i] = 0.0f;
    const int n = w * 32 + j;                                   // my column = tap index (valid below 49)
    const int tap = n < 49 ? n : 0;
    const int bbase = (tap / 7) * RF + (tap % 7) + h;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    float bsum = 0.0f;                                          // wave w: co = 16 w + lane / 4, pixels lane % 4 + 4 k
    const int tpr = Wd >> 4, tps = (H >> 4) * tpr;
    const long njob = TILED ? (long)B * tps : B, HW = (long)H * Wd;
    for (long job = blockIdx.x; job < njob; job += gridDim.x) {
        __syncthreads();
        if (TILED) {
            const long b = job / tps;
            const int tile = (int)(job % tps), y0 = (tile / tpr) * 16, x0 = (tile % tpr) * 16;
            for (int i = tid; i < 32 * 256; i += 128) {
                const int c = i >> 8, p = i & 255;
                gl[c * WG32_GLD + p] = gvf[(b * 32 + c) * HW + (long)(y0 + (p >> 4)) * Wd + x0 + (p & 15)];
            }
            for (int i = tid; i < 22 * 22; i += 128) {
                const int ry = i / 22, rx = i % 22, gy = y0 + ry - 3, gx = x0 + rx - 3;
                img[ry * RF + rx] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)Wd)
                                        ? eps1[b * HW + (long)gy * Wd + gx] : 0.0f;
            }
        } else {
            for (int i = tid; i < 32 * 256; i += 128) gl[(i >> 8) * WG32_GLD + (i & 255)] = gvf[job * 8192 + i];
            for (int p = tid; p < 256; p += 128) img[((p >> 4) + 3) * RF + (p & 15) + 3] = eps1[job * 256 + p];
        }
        __syncthreads();
        {
            const float *gr = gl + (16 * w + (lane >> 2)) * WG32_GLD + (lane & 3);
#pragma unroll 8
            for (int k = 0; k < 64; ++k) bsum += gr[4 * k];
        }
        const float *ga = gl + j * WG32_GLD + h;                // A: co = j, pixel p + h
#pragma unroll 8
        for (int pp = 0; pp < 128; ++pp) {
            const int p = 2 * pp;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[p], img[bbase + (p >> 4) * RF + (p & 15)], acc, 0, 0, 0);
        }
    }
    float *pw = part + (long)blockIdx.x * 32 * 50;
    if (n < 49) {
#pragma unroll
        for (int r = 0; r < 16; ++r) pw[(long)((r & 3) + 8 * (r >> 2) + 4 * h) * 50 + n] = acc[r];
    }
    bsum += __shfl_xor(bsum, 1);
    bsum += __shfl_xor(bsum, 2);
    if ((lane & 3) == 0) pw[(long)(16 * w + (lane >> 2)) * 50 + 49] = bsum;
}

// dW / db = fixed-order sum of the partial rows part[chunk][co][rowlen] (rowlen = c_in*ntap + 1, last = bias gradient)
// k_bwd_dv for layers without pooling: thread = one position k = (co, y, x) of the map, the (<= 32) readout weights of
// that position in registers, a chunk of samples looped over (g_p is wave-uniform -> scalar loads): i2o_W is read once
// per chunk instead of once per sample (786 KB x B through L2 before).
constexpr int DV_MAXPB = 16;            // samples per workgroup of k_bwd_dv_nopool (its per_block argument), at most
// NP: N rounded up to a multiple of 8 — the sum over the readout weights runs over NP terms without a test per term (the
// padding terms are 0 * 0); with the test the loop was a chain of branch, LDS read, wait, FMA (24 us at B = 512)
// FROM_PV: `v` holds the layer's pv = sigmoid(v) as its forward wrote it (a layer without pooling: same shape, and the same
// bits this kernel would recompute): the learning forward then does not have to store the membrane map at all
template <int NP, bool FROM_PV>
__device__ __forceinline__ void bwd_dv_nopool_body(const int K, const int N, const float *__restrict__ v,
                                                    const float *__restrict__ g_p, const float *__restrict__ g_pv,
                                                    const float *__restrict__ g_v, const float *__restrict__ i2o_W,
                                                    float *__restrict__ gvf, const int B, const int per_block,
                                                    const unsigned bx, const unsigned by)
{
    // the chunk's g_p rows through LDS (read back as broadcasts): as scalar loads from global memory they were a chain
    // of ~400 dependent s_load latencies per thread
    __shared__ __attribute__((aligned(16))) float gp[DV_MAXPB][32];
    const int b0 = (int)by * per_block, b1 = min(B, b0 + per_block);
    for (int e = threadIdx.x; e < per_block * 32; e += 256) {
        const int bb = b0 + (e >> 5), n = e & 31;
        gp[e >> 5][n] = (g_p && n < N && bb < b1) ? g_p[(long)bb * N + n] : 0.0f;
    }
    __syncthreads();
    const int k = (int)bx * 256 + threadIdx.x;
    if (k >= K) return;
    float wk[NP];
#pragma unroll
    for (int n = 0; n < NP; ++n) wk[n] = (g_p && n < N) ? i2o_W[(long)n * K + k] : 0.0f;
    // samples in groups of DV_G: all loads of a group are issued before the first is used (one sample at a time the loop
    // is a chain of HBM latencies: 29 us for 16 samples per thread at B = 512, 1.1 TB/s)
    constexpr int DV_G = 8;
    for (int bg = b0; bg < b1; bg += DV_G) {
        float vv[DV_G], gg[DV_G], ga[DV_G];
#pragma unroll
        for (int q = 0; q < DV_G; ++q) {
            const int b = min(bg + q, b1 - 1);
            const long i = (long)b * K + k;
            vv[q] = v[i];
            gg[q] = g_pv ? g_pv[i] : 0.0f;
            ga[q] = g_v ? g_v[i] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < DV_G; ++q) {
            const int b = min(bg + q, b1 - 1);
            float g = gg[q];
            if (g_p) {
                float acc = 0.0f;
#pragma unroll
                for (int n4 = 0; n4 < NP; n4 += 4) {
                    const f32x4 gq = *(const f32x4 *)&gp[b - b0][n4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = __builtin_fmaf(gq[u], wk[n4 + u], acc);
                }
                g += acc;
            }
            const float pv = FROM_PV ? vv[q] : sigmoidf_dev(vv[q]);
            float out = g * pv * (1.0f - pv);
            if (g_v) out += ga[q];
            if (bg + q < b1) gvf[(long)b * K + k] = out;
        }
    }
}

template <int NP, bool FROM_PV>
__global__ __launch_bounds__(256) void k_bwd_dv_nopool(int K, int N, const float *__restrict__ v,
                                                        const float *__restrict__ g_p, const float *__restrict__ g_pv,
                                                        const float *__restrict__ g_v, const float *__restrict__ i2o_W,
                                                        float *__restrict__ gvf, int B, int per_block)
{
    bwd_dv_nopool_body<NP, FROM_PV>(K, N, v, g_p, g_pv, g_v, i2o_W, gvf, B, per_block, blockIdx.x, blockIdx.y);
}
// the dv launches of several layers (the slices of one learning timestep) in one: blockIdx.z selects the item
// (dcll_conv_lif_backward_open_multi)
constexpr int BWD_MULTI_MAX = 8;
struct bwd_dv_items {
    const float *v[BWD_MULTI_MAX], *g_p[BWD_MULTI_MAX], *g_pv[BWD_MULTI_MAX], *g_v[BWD_MULTI_MAX], *i2o_W[BWD_MULTI_MAX];
    float *gvf[BWD_MULTI_MAX];
    int K[BWD_MULTI_MAX], N[BWD_MULTI_MAX], B[BWD_MULTI_MAX];
};
template <int NP, bool FROM_PV>
__global__ __launch_bounds__(256) void k_bwd_dv_nopool_m(const bwd_dv_items it, int per_block)
{
    const int z = blockIdx.z;
    if ((int)blockIdx.x * 256 >= it.K[z] || (int)blockIdx.y * per_block >= it.B[z]) return;     // (whole workgroups)
    bwd_dv_nopool_body<NP, FROM_PV>(it.K[z], it.N[z], it.v[z], it.g_p[z], it.g_pv[z], it.g_v[z], it.i2o_W[z], it.gvf[z], it.B[z],
                                    per_block, blockIdx.x, blockIdx.y);
}

// The same sum with four threads per output (64 outputs x 4 groups of partial rows per workgroup), the four group sums
// combined in fixed order through LDS: a quarter of the dependent-load latency of k_bwd_reduce (54 -> ~15 us for 256
// partial rows of 50 208 floats), still deterministic.
template <int GR>
__global__ __launch_bounds__(64 * GR) void k_bwd_reduce4(const float *__restrict__ part, float *__restrict__ dW,
                                                           float *__restrict__ db, int nchunk, int c_out, long rowlen)
{
    __shared__ float red[GR][64];
    const long i = (long)blockIdx.x * 64 + (threadIdx.x & 63);
    const int grp = threadIdx.x >> 6;
    const long total = c_out * rowlen;
    float acc = 0.0f;
    if (i < total) {
        const int per = (nchunk + GR - 1) / GR, c0 = grp * per, c1 = min(nchunk, c0 + per);
        const float *src = part + i;
        int c = c0;
        for (; c + 8 <= c1; c += 8) {                     // eight loads in flight, added in chunk order
            float t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = src[(long)(c + q) * total];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += t[q];
        }
        for (; c < c1; ++c) acc += src[(long)c * total];
    }
    red[grp][threadIdx.x & 63] = acc;
    __syncthreads();
    if (grp == 0 && i < total) {
        const int l = threadIdx.x & 63;
        float tot = red[0][l];
#pragma unroll
        for (int q = 1; q < GR; ++q) tot += red[q][l];
        const int co = (int)(i / rowlen);
        const long n = i % rowlen;
        if (n < rowlen - 1) dW[(long)co * (rowlen - 1) + n] = tot;
        else if (db) db[co] = tot;
    }
}

__global__ void k_bwd_reduce(const float *__restrict__ part, float *__restrict__ dW, float *__restrict__ db, int nchunk,
                             int c_out, long rowlen)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c_out * rowlen) return;
    const int co = (int)(i / rowlen);
    const long n = i % rowlen;
    float acc = 0.0f;
    for (int c = 0; c < nchunk; ++c) acc += part[((long)c * c_out + co) * rowlen + n];
    if (n < rowlen - 1) dW[(long)co * (rowlen - 1) + n] = acc;
    else if (db) db[co] = acc;
}

// d_outW[n,k] = sum_b g_o[b,n] * pvp[b,k];  d_outb[n] = sum_b g_o[b,n]      (output_ sees pv.detach())
__global__ void k_bwd_outgrad(const float *__restrict__ g_o, const float *__restrict__ pvp, float *__restrict__ dW,
                              float *__restrict__ db, int B, int N, int K)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * (K + 1)) return;
    int n = (int)(i / (K + 1)), k = (int)(i % (K + 1));
    float acc = 0.0f;
    if (k < K) {
        for (int b = 0; b < B; ++b) acc = __builtin_fmaf(g_o[(long)b * N + n], pvp[(long)b * K + k], acc);
        dW[(long)n * K + k] = acc;
    } else {
        for (int b = 0; b < B; ++b) acc += g_o[(long)b * N + n];
        db[n] = acc;
    }
}

// ------------------------------------------------------------------------------------------------------------
// readout GEMM: out[r,n] = sum_k pv[r,k] * Wt[n,k] + bias[n]      (i2o / output_, :602-606), fp32 MFMA
//   workgroup = 4 waves = 128 rows x 32 columns; K in chunks of 32 staged through LDS (row stride 33: conflict-free
//   column reads of the v_mfma_f32_32x32x2_f32 fragments).
// ------------------------------------------------------------------------------------------------------------

constexpr int RO_ROWS = 128, RO_KC = 32, RO_LD = 33;

__global__ __launch_bounds__(256) void k_readout(const float *__restrict__ pv, const float *__restrict__ Wt,
                                                  const float *__restrict__ bias, float *__restrict__ out,
                                                  long rows, int K, int N)
{
    __shared__ float sA[RO_ROWS * RO_LD];
    __shared__ float sB[32 * RO_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long row0 = (long)blockIdx.x * RO_ROWS;
    const int n0 = blockIdx.y * 32;
    const int col = tid & 31, rsub = tid >> 5;     // 8 row groups of 32 consecutive k
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (int k0 = 0; k0 < K; k0 += RO_KC) {
        const int k = k0 + col;
        const bool kin = k < K;
#pragma unroll
        for (int i = 0; i < RO_ROWS / 8; ++i) {
            int rr = rsub + 8 * i;
            long gr = row0 + rr;
            sA[rr * RO_LD + col] = (kin && gr < rows) ? pv[gr * K + k] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int nn = rsub + 8 * i;
            sB[nn * RO_LD + col] = (kin && n0 + nn < N) ? Wt[(long)(n0 + nn) * K + k] : 0.0f;
        }
        __syncthreads();
        const float *a = sA + (wave * 32 + (lane & 31)) * RO_LD + (lane >> 5);
        const float *b = sB + (lane & 31) * RO_LD + (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < RO_KC / 2; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * kk], b[2 * kk], acc, 0, 0, 0);
        __syncthreads();
    }
    const int n = n0 + (lane & 31);
    if (n < N) {
        const float bn = bias ? bias[n] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            long gr = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (gr < rows) out[gr * N + n] = acc[r] + bn;
        }
    }
}

// Fast path of the readout GEMM for K % 32 == 0 and 16-byte aligned rows: 128 rows x (32*NT) columns per workgroup
// (NT = 2 serves i2o and output_ of the output layer in ONE pass over pv), float4 global loads, the next K-chunk is
// fetched into registers while the MFMAs of the current one run (register double buffer).
template <int NT>
__global__ __launch_bounds__(256) void k_readout_v4(const float *__restrict__ pv, const float *__restrict__ Wt,
                                                     const float *__restrict__ bias, float *__restrict__ out,
                                                     long rows, int K, int N)
{
    __shared__ float sA[RO_ROWS * RO_LD];
    __shared__ float sB[NT * 32 * RO_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long row0 = (long)blockIdx.x * RO_ROWS;
    const int kq = (tid & 7) * 4, rsub = tid >> 3;          // 8 threads x float4 = one 32-float K-chunk of a row
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    f32x4 ra[4], rb[NT];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long gr = row0 + rsub + 32 * i;
            ra[i] = gr < rows ? *(const f32x4 *)(pv + gr * K + k0 + kq) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int nn = rsub + 32 * t;
            rb[t] = nn < N ? *(const f32x4 *)(Wt + (long)nn * K + k0 + kq) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += RO_KC) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) sA[(rsub + 32 * i) * RO_LD + kq + e] = ra[i][e];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) sB[(rsub + 32 * t) * RO_LD + kq + e] = rb[t][e];
        __syncthreads();
        if (k0 + RO_KC < K) fetch(k0 + RO_KC);
        const float *a = sA + (wave * 32 + (lane & 31)) * RO_LD + (lane >> 5);
        const float *bb = sB + (lane & 31) * RO_LD + (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < RO_KC / 2; ++kk) {
            const float av = a[2 * kk];
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bb[t * 32 * RO_LD + 2 * kk], acc[t], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = t * 32 + (lane & 31);
        if (n < N) {
            const float bn = bias ? bias[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                long gr = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (gr < rows) out[gr * N + n] = acc[t][r] + bn;
            }
        }
    }
}

// Long rows, moderately many of them (rows = T*B of a 128x128 plane: 8192 rows of K = 524288): 128-row tiles would be
// only 64 workgroups, so here a workgroup takes 32 rows and its 4 waves split every 256-float K-chunk between them
// (wave w: floats 64w..64w+63); the four partial tiles are combined in LDS in fixed order (deterministic).
constexpr int RK_ROWS = 32, RK_KC = 256, RK_LD = 257;
// kslice > 0 (dcll_readout_splitk): workgroup (x, y) handles only columns [y * kslice, (y+1) * kslice) of K and writes
// its partial tile, without the bias, to out + y * rows * N; k_readout_sum adds the slices in order.
template <int NT>
__global__ __launch_bounds__(256) void k_readout_ks(const float *__restrict__ pv, const float *__restrict__ Wt,
                                                     const float *__restrict__ bias, float *__restrict__ out,
                                                     long rows, int K, int N, int kslice)
{
    __shared__ float sm[(RK_ROWS + NT * 32) * RK_LD];
    float *sA = sm, *sB = sm + RK_ROWS * RK_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long row0 = (long)blockIdx.x * RK_ROWS;
    const int kq = (tid & 63) * 4, rsub = tid >> 6;         // 64 threads x float4 = one 256-float K-chunk of a row
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    f32x4 ra[8], rb[NT * 8];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            long gr = row0 + rsub + 4 * i;
            ra[i] = gr < rows ? *(const f32x4 *)(pv + gr * K + k0 + kq) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < NT * 8; ++i) {
            int nn = rsub + 4 * i;
            rb[i] = nn < N ? *(const f32x4 *)(Wt + (long)nn * K + k0 + kq) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    const int kbeg = kslice > 0 ? blockIdx.y * kslice : 0, kend = kslice > 0 ? kbeg + kslice : K;
    if (kslice > 0) { out += (long)blockIdx.y * rows * N; bias = nullptr; }
    fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += RK_KC) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) sA[(rsub + 4 * i) * RK_LD + kq + e] = ra[i][e];
#pragma unroll
        for (int i = 0; i < NT * 8; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) sB[(rsub + 4 * i) * RK_LD + kq + e] = rb[i][e];
        __syncthreads();
        if (k0 + RK_KC < kend) fetch(k0 + RK_KC);
        const float *a = sA + (lane & 31) * RK_LD + 64 * wave + (lane >> 5);
        const float *bb = sB + (lane & 31) * RK_LD + 64 * wave + (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            const float av = a[2 * kk];
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bb[t * 32 * RK_LD + 2 * kk], acc[t], 0, 0, 0);
        }
        __syncthreads();
    }
    // partial tiles -> LDS [wave][t][r][lane], then 256 threads add the four in wave order
    float *red = sm;
    static_assert(4 * NT * 1024 <= (RK_ROWS + NT * 32) * RK_LD, "reduction buffer must fit the tile buffers");
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * NT + t) * 16 + r) * 64 + lane] = acc[t][r];
    __syncthreads();
    for (int e = tid; e < NT * 1024; e += 256) {
        const int l = e & 63, r = (e >> 6) & 15, t = e >> 10;
        const int n = t * 32 + (l & 31);
        const long gr = row0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        if (n < N && gr < rows) {
            float tot = red[e];
#pragma unroll
            for (int wv = 1; wv < 4; ++wv) tot += red[wv * NT * 1024 + e];
            out[gr * N + n] = tot + (bias ? bias[n] : 0.0f);
        }
    }
}

// Few rows (per-step calls: rows = batch): the 128-row tiles above would leave most CUs idle, so here one workgroup
// takes RS_RB rows and RS_NG readout rows of Wt (grid = rows/RS_RB x ceil(N / RS_NG): enough workgroups even when K is huge,
// e.g. 32*128*128 on the 128x128 plane), its 256 threads stride over K, and a fixed-order LDS tree combines them
// (deterministic, no atomics).
constexpr int RS_NG = 4, RS_RB = 4;
__global__ __launch_bounds__(256) void k_readout_rows(const float *__restrict__ pv, const float *__restrict__ Wt,
                                                       const float *__restrict__ bias, float *__restrict__ out,
                                                       long rows, int K, int N)
{
    __shared__ float red[RS_RB * RS_NG][256];
    const long row0 = (long)blockIdx.x * RS_RB;
    const int n0 = blockIdx.y * RS_NG;
    float acc[RS_RB][RS_NG];
#pragma unroll
    for (int r = 0; r < RS_RB; ++r)
#pragma unroll
        for (int u = 0; u < RS_NG; ++u) acc[r][u] = 0.0f;
    for (int k = threadIdx.x; k < K; k += 256) {
        float a[RS_RB], w[RS_NG];
#pragma unroll
        for (int r = 0; r < RS_RB; ++r) a[r] = row0 + r < rows ? pv[(row0 + r) * K + k] : 0.0f;
#pragma unroll
        for (int u = 0; u < RS_NG; ++u) w[u] = n0 + u < N ? Wt[(long)(n0 + u) * K + k] : 0.0f;
#pragma unroll
        for (int r = 0; r < RS_RB; ++r)
#pragma unroll
            for (int u = 0; u < RS_NG; ++u) acc[r][u] = __builtin_fmaf(a[r], w[u], acc[r][u]);
    }
#pragma unroll
    for (int r = 0; r < RS_RB; ++r)
#pragma unroll
        for (int u = 0; u < RS_NG; ++u) red[r * RS_NG + u][threadIdx.x] = acc[r][u];
    __syncthreads();
    for (int sft = 128; sft > 0; sft >>= 1) {
        if (threadIdx.x < sft) {
#pragma unroll
            for (int q = 0; q < RS_RB * RS_NG; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + sft];
        }
        __syncthreads();
    }
    if (threadIdx.x < RS_RB * RS_NG) {
        const int r = threadIdx.x / RS_NG, u = threadIdx.x % RS_NG;
        if (row0 + r < rows && n0 + u < N) out[(row0 + r) * N + n0 + u] = red[threadIdx.x][0] + (bias ? bias[n0 + u] : 0.0f);
    }
}

// ------------------------------------------------------------------------------------------------------------
// argmax per (t,b) (first maximum, like torch.argmax) and vote per b (Counter.most_common(1): ties -> first seen)
// ------------------------------------------------------------------------------------------------------------
__global__ void k_argmax(const float *__restrict__ logits, int32_t *__restrict__ clout, long rows, int N)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const float *l = logits + i * N;
    int best = 0;
    float bv = l[0];
    if (N % 4 == 0 && ((uintptr_t)logits & 15) == 0) {
        // rows of 16-byte multiples (N = 24): N / 4 loads of 16 bytes per thread instead of N of 4 — a wave's rows are 4 N
        // bytes apart, so every scalar load touched every cache line of the wave's block (68 us for 50 MB at T x B = 524 288)
        for (int n4 = 0; n4 < N; n4 += 4) {
            const f32x4 q = *(const f32x4 *)(l + n4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (q[e] > bv) { bv = q[e]; best = n4 + e; }
        }
    } else {
        for (int n = 1; n < N; ++n) {
            float v = l[n];
            if (v > bv) { bv = v; best = n; }
        }
    }
    clout[i] = best;
}

constexpr int VOTE_MAXN = 64;
__global__ __launch_bounds__(64) void k_vote(const int32_t *__restrict__ clout, int32_t *__restrict__ vote, int T,
                                              int B, int N, int t_begin)
{
    __shared__ int cnt[VOTE_MAXN * 64];
    __shared__ int first[VOTE_MAXN * 64];
    const int b = blockIdx.x * 64 + threadIdx.x;
    for (int n = 0; n < N; ++n) { cnt[n * 64 + threadIdx.x] = 0; first[n * 64 + threadIdx.x] = T; }
    if (b >= B) return;
    for (int t0 = t_begin; t0 < T; t0 += 16) {        // sixteen recorded steps requested at once: the loop is one thread per
        int c[16];                                    // sample, 64 workgroups of one wave — nothing else hides a round trip
#pragma unroll
        for (int k = 0; k < 16; ++k) c[k] = t0 + k < T ? clout[(long)(t0 + k) * B + b] : -1;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (c[k] >= 0 && cnt[c[k] * 64 + threadIdx.x]++ == 0) first[c[k] * 64 + threadIdx.x] = t0 + k;
    }
    int best = -1, bc = 0, bf = 0;
    for (int n = 0; n < N; ++n) {
        int c = cnt[n * 64 + threadIdx.x], f = first[n * 64 + threadIdx.x];
        if (c == 0) continue;
        if (best < 0 || c > bc || (c == bc && f < bf)) { best = n; bc = c; bf = f; }
    }
    vote[b] = best;
}

// ------------------------------------------------------------------------------------------------------------
// glue: IQ -> cell index, pack / unpack
// ------------------------------------------------------------------------------------------------------------
__global__ void k_iq_encode(const float *__restrict__ iq, const float *__restrict__ thr_i,
                            const float *__restrict__ thr_q, const dcll_iq_tail tail, int32_t *__restrict__ cells, int B, int L, int t0, int T,
                            int w, int h)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)T * B) return;
    int b = (int)(i % B), t = (int)(i / B);
    float vi = iq[((long)b * 2 + 0) * L + t0 + t];
    float vq = iq[((long)b * 2 + 1) * L + t0 + t];
    int ci = 0, cq = 0;
    const float *ti, *tq;
    iq_tables(thr_i, thr_q, tail, b, ti, tq);
    for (int j = 0; j < w - 1; ++j) ci += vi >= ti[j];
    for (int j = 0; j < h - 1; ++j) cq += vq >= tq[j];
    cells[i] = cq * w + ci;
}

__global__ void k_unpack(const uint32_t *__restrict__ packed, float *__restrict__ dense, long nwords)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per output float
    if (i >= nwords * 32) return;
    dense[i] = (float)((packed[i >> 5] >> (i & 31)) & 1u);
}

__global__ void k_pack(const float *__restrict__ dense, uint32_t *__restrict__ packed, long nwords)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per input float; wave = 2 words
    bool bit = (i < nwords * 32) && dense[i] != 0.0f;
    unsigned long long m = __ballot(bit);
    int lane = threadIdx.x & 63;
    if (lane == 0 && (i >> 5) < nwords) packed[i >> 5] = (uint32_t)m;
    if (lane == 32 && (i >> 5) < nwords) packed[i >> 5] = (uint32_t)(m >> 32);
}

// ------------------------------------------------------------------------------------------------------------
// first layer sequence kernel: c_in == 1, 16x16 plane, 7x7 pad 3, pool 1, c_out <= 32.
//   one workgroup (256 threads = one per pixel) per sample, all T steps; eps0/eps1 of the pixel in registers, eps1
//   mirrored into a zero-padded 22x22 LDS plane; each thread gathers its 49 taps once per step and runs the fmaf
//   chain (tap order = the pinned order for c_in == 1) for every output channel with wave-uniform weights.
// ------------------------------------------------------------------------------------------------------------
constexpr int C1_MAXT = 4096;       // longest window the fused IQ encoder of k_lif_seq_c1 keeps in LDS

// Input either as cell indices (IQ = false) or as raw IQ (IQ = true): then the quantisation of iq2spiketrain
// (data/utils.py:60-82, threshold form as in k_iq_encode) is fused here: thread t quantises sample t0+t (coalesced
// loads of the I and the Q row of this window) and parks the cell index of step t in LDS.
//
// Conv as fp32 MFMA, weight-stationary: the K = 49 taps in their pinned linear order (ky, kx) are paired two by two over
// the MFMA's k lanes — pair p = taps (2p, 2p+1), 25 MFMAs per tile, only tap 49 is padding (ZERO weight:
// fmaf(x, 0, acc) == acc, so every channel still sees the chain bias, tap 0, tap 1, ... tap 48).  A = weights (lane: co =
// lane&31, tap parity = lane>>5; 25 VGPRs for the whole sequence), B = eps1 from a zero-padded 22x24 LDS plane: lane h = 1
// reads one float behind lane h = 0, except for the three pairs whose second tap starts the next kernel row (p = 3, 10,
// 17: PS - 6 floats behind) — two per-lane bases + immediates.  (Round 1 padded every kernel row to 4 pairs: 28 MFMAs.)
// D[co][pixel]; 4 waves = 8 pixel tiles of 32 (wave w: image rows 4w..4w+3); a thread also owns pixel `tid` of the traces.
// Spike words: the ballot of accumulator register r IS the two spike words of channels (r, h = 0 / 1) of this tile; they
// are parked in lanes r and 32 + r of one VGPR with v_writelane (immediate lane, SGPR source: two VALU instructions per
// value, no select masks — the select chain of round 1 kept 16 64-bit masks and spilled ~100 SGPRs).
// FAST: c_out == 32 and exactly the outputs spk_out + pv_out — the benchmark's configuration: no per-value channel /
// pointer guards, and stores as uniform base + 32-bit lane offset.  FAST == 2: pv_out receives v instead of sigmoid(v)
// (dcll_layer_opts pv_presigmoid: the readout applies the sigmoid) — 4 of the ~10 VALU instructions per value, two of them
// quarter-rate transcendentals, leave the pipe this kernel shares with its MFMAs.
template <bool REFRACTORY, int FAST, bool IQ>
__global__ __launch_bounds__(256, 3) void k_lif_seq_c1(int c_out, const int32_t *__restrict__ cells,
                                                    const float *__restrict__ iq, const float *__restrict__ thr_i,
                                                    const float *__restrict__ thr_q, const dcll_iq_tail tail, int L, int t0,
                                                    const dcll_wsrc W, const float *__restrict__ bias,
                                                    const float *__restrict__ tau4, float *__restrict__ eps0_g,
                                                    float *__restrict__ eps1_g, float *__restrict__ arp_g,
                                                    uint32_t *__restrict__ spk_out, float *__restrict__ pv_out,
                                                    float *__restrict__ v_out, int T, int B, float alpharp, float wrp)
{
    constexpr int PS = 24;                      // plane row stride: 16 + 2*3 padding + the pad tap's column
    constexpr int PLANE = 22 * PS + 8;
    // two planes, by step parity: a step's MFMAs read plane[t & 1] while the next step's trace update already writes the
    // other one — ONE barrier per step (round 3; a single plane needed a second barrier behind the MFMAs)
    __shared__ float plane[2 * PLANE];
    __shared__ float sbias[32];
    __shared__ int scell[IQ ? C1_MAXT : 1];
    const int b = blockIdx.x, pix = threadIdx.x, y = pix >> 4, x = pix & 15, lane = pix & 63;
    const int w = __builtin_amdgcn_readfirstlane(pix >> 6);
    const int h = lane >> 5, j = lane & 31;
    const float alpha = tau4[0], tau_m = tau4[1], alphas = tau4[2], tau_s = tau4[3];
    for (int i = pix; i < 2 * PLANE; i += 256) plane[i] = 0.0f;
    if (IQ) {
        const float *ti, *tq;
        iq_tables(thr_i, thr_q, tail, b, ti, tq);
        for (int t = pix; t < T; t += 256) {
            const float vi = iq[((long)b * 2 + 0) * L + t0 + t], vq = iq[((long)b * 2 + 1) * L + t0 + t];
            int ci = 0, cq = 0;
            for (int k = 0; k < 15; ++k) { ci += vi >= ti[k]; cq += vq >= tq[k]; }
            scell[t] = cq * 16 + ci;
        }
    }
    float e0 = eps0_g[(long)b * 256 + pix], e1 = eps1_g[(long)b * 256 + pix];
    // weight fragments: pair p: lane (co = j, tap = 2p + h); the pad tap and channels >= c_out carry 0
    float wf[25];
#pragma unroll
    for (int p = 0; p < 25; ++p) {
        const int tap = 2 * p + h;
        wf[p] = (tap < 49 && j < c_out) ? W.at(j * 49 + tap, j) : 0.0f;
    }
    // refractory trace of my two tiles: arp[tl][r] <-> channel (r&3)+8(r>>2)+4h, pixel 32*(2w+tl) + j
    float arp[2][16];
    if (pix < 32) sbias[pix] = pix < c_out ? bias[pix] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
            arp[tl][r] = (REFRACTORY && co < c_out) ? arp_g[((long)b * c_out + co) * 256 + 32 * (2 * w + tl) + j] : 0.0f;
    }
    __syncthreads();
    int cell = IQ ? scell[0] : cells[b];
    for (int t = 0; t < T; ++t) {
        // next step's cell index: requested now, needed after the barrier at the end of this step
        const int tn = t + 1 < T ? t + 1 : t;
        const int cell_next = IQ ? scell[tn] : cells[(long)tn * B + b];
        trace_update(cell == pix ? 1.0f : 0.0f, alpha, tau_m, alphas, tau_s, e0, e1);
        float *pl = plane + (t & 1) * PLANE;
        pl[(y + 3) * PS + x + 3] = e1;
        // LDS-only barrier (does not wait for this step's pv stores): every wave's reads of plane[t & 1] from step t - 2 were
        // issued before it passed the barrier of step t - 1, and that barrier waits for the wave's outstanding LDS operations
        lds_barrier();
        const long obase = ((long)t * B + b) * c_out;
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const int m = 2 * w + tl;
            const float *bn = pl + (2 * m + (j >> 4)) * PS + (j & 15) + h;               // second tap: + 1
            const float *bx = pl + (2 * m + (j >> 4)) * PS + (j & 15) + h * (PS - 6);    // ... or the next row's first
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = sbias[(r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
            for (int p = 0; p < 25; ++p) {
                const int tap = 2 * p, off = (tap / 7) * PS + tap % 7;
                const bool cross = (tap % 7 == 6) && p < 24;        // p = 24: the partner is the pad tap (weight 0)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[p], cross ? bx[off] : bn[off], acc, 0, 0, 0);
            }
            int myword = 0;
            float *pvb = pv_out + obase * 256;                      // wave-uniform base of this step's pv planes
            const unsigned loff = 4 * h * 256 + 32 * m + j;         // + ((r&3) + 8(r>>2)) * 256 per value
            static_for<0, 16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const int co = (r & 3) + 8 * (r >> 2) + 4 * h;
                float v = acc[r];
                bool s;
                if (REFRACTORY) v = refractory(acc[r], arp[tl][r], alpharp, wrp, s);
                else s = v > 0.0f;
                const unsigned long long mk = __ballot(s);
                // wait states as the compiler places them around its own v_writelane (SGPR spills): two after the VALU
                // write of the source SGPRs (v_cmp), two between writes of the same VGPR.  (Without them lane r received
                // the stale content of the SGPR: measured, wrong spike words with a bit-exact v.)
                asm("s_nop 1\n\tv_writelane_b32 %0, %1, %3\n\ts_nop 1\n\tv_writelane_b32 %0, %2, %4"
                    : "+v"(myword) : "s"((uint32_t)mk), "s"((uint32_t)(mk >> 32)), "n"(r), "n"(32 + r));
                if (FAST) {
                    (pvb + ((r & 3) + 8 * (r >> 2)) * 256)[loff] = FAST == 2 ? v : sigmoidf_dev(v);
                } else if (co < c_out) {
                    if (pv_out) pv_out[(obase + co) * 256 + 32 * m + j] = sigmoidf_dev(v);
                    if (v_out) v_out[(obase + co) * 256 + 32 * m + j] = v;
                }
            });
            const int cow = (j & 3) + 8 * (j >> 2) + 4 * h;
            if (FAST) {
                if (j < 16) (spk_out + obase * 8)[cow * 8 + m] = (uint32_t)myword;
            } else if (spk_out && j < 16 && cow < c_out) {
                spk_out[(obase + cow) * 8 + m] = (uint32_t)myword;
            }
        }
        cell = cell_next;
    }
    eps0_g[(long)b * 256 + pix] = e0;
    eps1_g[(long)b * 256 + pix] = e1;
    if (REFRACTORY) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
            for (int tl = 0; tl < 2; ++tl)
                if (co < c_out) arp_g[((long)b * c_out + co) * 256 + 32 * (2 * w + tl) + j] = arp[tl][r];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// THE HOT KERNEL: 32 -> 32 channels, 7x7 pad 3, 16x16 plane, pool 1, all T steps, one sample per workgroup.
//
// Implicit GEMM per step:  V[co, pix] = bias[co] + sum_k Wm[co, k] * E[k, pix],  k = (ci, ky, kx), K = 1568.
//   MFMA v_mfma_f32_32x32x2_f32: A = Wm fragment (lane: co = lane&31, k = lane>>5), B = E fragment (lane: pixel =
//   lane&31, k = lane>>5), D[co][pix] with pixel on the lane and 16 channels in registers.  The two k of one
//   instruction are the input-channel pair (2cp, 2cp+1) at the same tap, so both halves of the wave read LDS at
//   one immediate offset from a per-lane base.  The instruction is, bit for bit, acc = fmaf(a1,b1,fmaf(a0,b0,acc)).
//
// LDS image of eps1 (zero padding shared between rows and between channels), DOUBLE-buffered by step parity:
//   element (ci, y, x), y,x in [-3,18]  at float offset  ci*361 + (y+3)*19 + (x+3);   only 0<=y,x<16 is ever written.
//   (x=16..18 of row y aliases x=-3..-1 of row y+1; rows 16..18 of channel c alias rows -3..-1 of channel c+1.)
//
// Stage g (one s_barrier per stage): wave w computes tile q = g - w (step t = q>>3, pixel tile m = q&7 = image rows
//   2m, 2m+1) from image[t&1]: takes the accumulator of wave w-1 from slot[w-1][(g-1)&1] (wave 0: bias), adds its 98
//   MFMAs, puts it in slot[w][g&1]: a systolic chain, so every output is ONE fmaf chain in the pinned order.
//
// Measured on MI355X (experiments/overlap.hip): v_mfma_f32_32x32x2_f32 runs on the SIMD's fp32 vector ALUs, so every
//   VALU instruction of either wave of a SIMD adds its cycles to the MFMA time — nothing hides under the partner's
//   MFMAs — and all waves meet at the stage barrier.  The non-MFMA work is therefore spread evenly over the four SIMDs
//   (waves w and w+4 share a SIMD) in every stage:
//   - trace update: wave w advances ONE of its 4 input channels per stage, for step t+1, in every second stage of
//     step t (reads image[t&1], writes image[(t+1)&1]; eps0 lives in registers) — the stages in which it also has an
//     epilogue share, so that its SIMD partner has no extra work in that stage;
//   - epilogue of the tile handed over by wave 7 (refractory trace, threshold, sigmoid, ballot-packed spikes): split
//     by accumulator register quad over the 4 waves with (w>>2) == (m&1); wave w owns quad w&3 (channels
//     rr + 8*(w&3) + 4*(lane>>5)) of the tiles of its parity and keeps their 16 arp values in registers.
// ------------------------------------------------------------------------------------------------------------

// ABLATE is a diagnostic knob for experiments/ablate_c32.hip only (bit0: no epilogue, bit1: no trace update,
// bit2: no accumulator hand-off); every product launch uses ABLATE = 0.
// NRO > 0 fuses the local readout(s) into the epilogue: logits[t][b][n] = sum_{co,pix} pv * Wro[n][co][pix] + b[n]
// (i2o, and output_ stacked behind it on the last layer: dcll/pytorch_libdcll.py:602-606) with Wro pre-permuted to
// the epilogue's register layout (dcll_permute_readout); pv then never travels through HBM.
// PRIO: s_setprio of the wave that carries the stage's non-MFMA work (bits 0-1: level, bit 2: keep it through the chain).
// Measured (experiments/ablate_c32.hip, B=1024): 0 -> 25.56 ms, 1 -> 25.18 ms, 5 -> 25.13 ms.
template <bool REFRACTORY, int OUT = 3, int NRO = 0, int ABLATE = 0, int PRIO = 5, int BASES = 1>  // OUT bit0: pv, bit1: v
__global__ __launch_bounds__(512) void k_lif_seq_c32(const uint32_t *__restrict__ spk_in, const dcll_wsrc W,
                                                      const float *__restrict__ bias, const float *__restrict__ tau4,
                                                      float *__restrict__ eps0_g, float *__restrict__ eps1_g,
                                                      float *__restrict__ arp_g, uint32_t *__restrict__ spk_out,
                                                      float *__restrict__ pv_out, float *__restrict__ v_out,
                                                      const float *__restrict__ ro_Wp, const float *__restrict__ ro_b,
                                                      float *__restrict__ ro_out, int T, int B, float alpharp,
                                                      float wrp)
{
    constexpr int NROA = NRO > 0 ? NRO : 1;
    __shared__ __attribute__((aligned(16))) float lds[2 * IMG_FLOATS + NWAVE * 2 * SLOT_FLOATS + 32 + 2 * NWAVE * NRO];
    float *slots = lds + 2 * IMG_FLOATS;
    float *sbias = slots + NWAVE * 2 * SLOT_FLOATS;
    float *ropart = sbias + 32;                 // [step parity][wave][NRO] lane-reduced partial logits
    float racc[NROA];
#pragma unroll
    for (int n = 0; n < NROA; ++n) racc[n] = 0.0f;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, j = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave index, kept scalar
    const int wq = w & 3, wpar = w >> 2;                          // epilogue quad / tile parity owned by this wave
    // my trace shares fall in the SAME stages as my epilogue shares (stage parity g&1 == wpar  <=>  tile parity
    // m&1 == tpar): per SIMD and stage ONE wave carries all the non-MFMA work and its partner starts its MFMAs at
    // once.  (Measured: with the two jobs on different waves of a SIMD both begin the stage waiting on LDS and that
    // SIMD reaches the barrier last.)
    const int tpar = wpar ^ (w & 1);
    const long b = blockIdx.x;

    for (int i = tid; i < 2 * IMG_FLOATS; i += 512) lds[i] = 0.0f;
    if (tid < 32) sbias[tid] = bias[tid];

    // weight fragments: A[co = j][k = h] of MFMA (cp, tap) = W[j][4w + 2cp + h][tap]
    float wf[2][49];
    load_wf_c32(W, j, w, h, wf);

    // trace state of my 4 channels: element (c, ii): channel 4w+c, pixel ii*64 + lane.  eps0 in registers, eps1 in
    // the LDS images at float offset ioff + c*CHF + ii*4*ROWF.
    float e0[16];
    const int ioff = (4 * w) * CHF + ((lane >> 4) + 3) * ROWF + (lane & 15) + 3;
    // input spikes of step t for my channels: channel c = 8 consecutive uint32 = 4 x 64-bit masks; mask ii covers
    // pixels ii*64 .. ii*64+63, i.e. bit l of mask ii is the input of (pixel ii*64 + lane l): the mask IS the lane mask
    // of a v_cndmask.  Wave-uniform addresses -> scalar loads (lgkmcnt; no vmcnt traffic inside the stage loop).
    const unsigned long long *in_wave = (const unsigned long long *)(spk_in + (b * 32 + 4 * w) * 8);
    const long in_step = (long)B * 32 * 4;      // in 64-bit units
    __syncthreads();        // images zeroed

    // one trace element update: x*tau_s by lane mask -> eps0 (register) and eps1 (src image -> dst image);
    // dcll/pytorch_libdcll.py:493-494, every op rounded separately.
    auto trace_elem = [&](unsigned long long mask, float &e0r, float e1, float *dst, float ta, float tm, float tas,
                          float ts) {
        float a;                                    // x * tau_s with x in {0,1}: exact select
        asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(a) : "v"(ts), "s"(mask));
        float bb = tas * e0r;
        e0r = a + bb;
        float cc = ta * e1;
        float dd = e0r * tm;
        *dst = cc + dd;
    };

    // prologue: state from HBM, advanced to step 0 with the input bits of step 0 -> image[0]
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const float ta = tau4[0 * 32 + 4 * w + c], tm = tau4[1 * 32 + 4 * w + c];
        const float tas = tau4[2 * 32 + 4 * w + c], ts = tau4[3 * 32 + 4 * w + c];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const long gidx = (b * 32 + 4 * w + c) * 256 + ii * 64 + lane;
            e0[c * 4 + ii] = eps0_g[gidx];
            float e1 = eps1_g[gidx];
            if (!(ABLATE & 2)) {
                float xin = (float)((in_wave[c * 4 + ii] >> lane) & 1ull);
                trace_update(xin, ta, tm, tas, ts, e0[c * 4 + ii], e1);
            }
            lds[ioff + c * CHF + ii * 4 * ROWF] = e1;
        }
    }
    // refractory trace of my epilogue share: tiles m = 2k + wpar (k = 0..3), quad wq:
    //   arp[k][rr] <-> channel rr + 8*wq + 4h, pixel 32*(2k+wpar) + j
    float arp[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
            arp[k][rr] = REFRACTORY ? arp_g[(b * 32 + rr + 8 * wq + 4 * h) * 256 + 32 * (2 * k + wpar) + j] : 0.0f;

    // per-lane base of the B-fragment reads: channel 4w+h, pixel row (j>>4), col (j&15); tile m adds 2 rows
    const int bbase = (4 * w + h) * CHF + (j >> 4) * ROWF + (j & 15);
    // input masks of my NEXT trace share (step 1, channel 0 first), fetched one share ahead into SGPRs
    unsigned long long pw0 = 0, pw1 = 0, pw2 = 0, pw3 = 0;
    if (T > 1) {
        const unsigned long long *ip = in_wave + in_step;
        pw0 = ip[0]; pw1 = ip[1]; pw2 = ip[2]; pw3 = ip[3];
    }
    // every prologue load (weights, state, arp) has landed before the stage loop: keeps vmcnt waits out of the loop
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) asm volatile("" ::"v"(arp[k][rr]));
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(e0[i]));
#pragma unroll
    for (int cp = 0; cp < 2; ++cp)
#pragma unroll
        for (int k = 0; k < 49; ++k) asm volatile("" ::"v"(wf[cp][k]));
    __syncthreads();

    const int nstage = 8 * T + (NRO > 0 ? 17 : 9);
    unsigned long long dbg_wait = 0, dbg_t0 = 0, dbg_epi = 0, dbg_tr = 0, dbg_mf = 0;   // ABLATE & 8 only
    if (ABLATE & 8) dbg_t0 = __builtin_amdgcn_s_memtime();
    for (int g = 0; g < nstage; ++g) {
        unsigned long long dbg_a = 0;
        if (ABLATE & 8) dbg_a = __builtin_amdgcn_s_memtime();
        // ---- (1) epilogue share: quad wq of tile qe = g - 8 (finished by wave 7 in the previous stage) ----
        const int qe = g - 8;
        const bool loaded = ((g & 1) == wpar);          // this wave carries the stage's non-MFMA work of its SIMD
        if (loaded) __builtin_amdgcn_s_setprio(PRIO & 3);
        if (!(ABLATE & 1) && qe >= 0 && qe < 8 * T && ((qe & 1) == wpar)) {
            const int te = qe >> 3, me = qe & 7;
            const f32x4 v4 = *((const f32x4 *)(slots + (7 * 2 + ((g - 1) & 1)) * SLOT_FLOATS) + wq * 64 + lane);
            const long obase = ((long)te * B + b) * 32 + 8 * wq + 4 * h;      // + rr = channel
            const long oelem = obase * 256 + 32 * me + j;                     // + rr*256
            float *pvp = pv_out + oelem, *vp = v_out + oelem;
            float pvq[4];
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
                    if ((OUT & 1) || NRO > 0) pvq[rr] = sigmoidf_dev(v);
                    if (OUT & 1) pvp[rr * 256] = pvq[rr];
                    if (OUT & 2) vp[rr * 256] = v;
                }
                if (spk_out && j < 4) spk_out[(obase + j) * 8 + me] = myword;
            };
            switch (me >> 1) {          // wave-uniform: keeps arp[][] statically indexed (registers)
            case 0: quad(arp[0]); break;
            case 1: quad(arp[1]); break;
            case 2: quad(arp[2]); break;
            default: quad(arp[3]); break;
            }
            if (NRO > 0) {
                // my 4 pv values against the matching 4 readout weights of every class (16 B per lane per class,
                // 1 KB contiguous per wave-load, L2 resident); partial sums stay in registers over my 4 tiles of a
                // step.  Loads go out in batches of RO_CH before their FMAs (order pinned) so that one L2 round trip
                // is paid per batch, not per class.
                constexpr int RO_CH = (NRO > 24) ? 6 : 12;
                const f32x4 *wp = (const f32x4 *)ro_Wp + (long)((me * 4 + wq) * NRO) * 64 + lane;
#pragma unroll
                for (int n0 = 0; n0 < NRO; n0 += RO_CH) {
                    f32x4 wb[RO_CH];
#pragma unroll
                    for (int k = 0; k < RO_CH; ++k) wb[k] = wp[(n0 + k) * 64];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < RO_CH; ++k)
                        racc[n0 + k] = __builtin_fmaf(pvq[0], wb[k][0], __builtin_fmaf(pvq[1], wb[k][1],
                                       __builtin_fmaf(pvq[2], wb[k][2], __builtin_fmaf(pvq[3], wb[k][3], racc[n0 + k]))));
                    __builtin_amdgcn_sched_barrier(0);
                }
                if ((me >> 1) == 3) {       // my last tile of step te: reduce over lanes, park in LDS, restart
                    float *dst = ropart + ((te & 1) * NWAVE + w) * NRO;
#pragma unroll
                    for (int n = 0; n < NRO; ++n) {
                        const float tot = wave_sum_to_lane63(racc[n]);
                        if (lane == 63) dst[n] = tot;
                        racc[n] = 0.0f;
                    }
                }
            }
        }
        if (NRO > 0) {
            // all 8 partials of step tc were parked by the end of stage 8*tc + 15: combine + bias -> logits
            const int gc = g - 16;
            if (gc >= 0 && (gc & 7) == 0 && (gc >> 3) < T && w == ((gc >> 3) & 7) && lane < NRO) {
                const int tc = gc >> 3;
                const float *src = ropart + ((tc & 1) * NWAVE) * NRO + lane;
                float tot = ro_b[lane];
#pragma unroll
                for (int k = 0; k < NWAVE; ++k) tot += src[k * NRO];
                ro_out[((long)tc * B + b) * NRO + lane] = tot;
            }
        }
        if (ABLATE & 8) { unsigned long long x_ = __builtin_amdgcn_s_memtime(); dbg_epi += x_ - dbg_a; dbg_a = x_; }
        const int q = g - w;
        const bool active = q >= 0 && q < 8 * T;
        const bool tracing = active && !(ABLATE & 2) && (((q & 7) & 1) == tpar) && (q >> 3) + 1 < T;
        if (loaded && !tracing && !(PRIO & 4)) __builtin_amdgcn_s_setprio(0);
        if (active) {
            const int m = q & 7, t = q >> 3;
            float *img = lds + (t & 1) * IMG_FLOATS;
            // ---- (2) trace share: channel c of step t+1 in stage m = 2c + wpar ----
            if (tracing) {
                const int c = m >> 1;
                const float ta = tau4[0 * 32 + 4 * w + c], tm = tau4[1 * 32 + 4 * w + c];
                const float tas = tau4[2 * 32 + 4 * w + c], ts = tau4[3 * 32 + 4 * w + c];
                const unsigned long long w0 = pw0, w1 = pw1, w2 = pw2, w3 = pw3;
                {   // prefetch the masks of my next share: (t+1, c+1) or (t+2, 0)
                    const int tn = (c < 3) ? t + 1 : t + 2, cn = (c + 1) & 3;
                    if (tn < T) {
                        const unsigned long long *ip = in_wave + (long)tn * in_step + cn * 4;
                        pw0 = ip[0]; pw1 = ip[1]; pw2 = ip[2]; pw3 = ip[3];
                    }
                }
                const float *src = img + ioff + c * CHF;
                float *dst = lds + ((t + 1) & 1) * IMG_FLOATS + ioff + c * CHF;
                const float s0 = src[0], s1 = src[4 * ROWF], s2 = src[8 * ROWF], s3 = src[12 * ROWF];
                switch (c) {            // wave-uniform: keeps e0[] statically indexed (registers)
#define DCLL_TRACE_CASE(C)                                                                \
    case C:                                                                               \
        trace_elem(w0, e0[C * 4 + 0], s0, dst + 0 * 4 * ROWF, ta, tm, tas, ts);           \
        trace_elem(w1, e0[C * 4 + 1], s1, dst + 1 * 4 * ROWF, ta, tm, tas, ts);           \
        trace_elem(w2, e0[C * 4 + 2], s2, dst + 2 * 4 * ROWF, ta, tm, tas, ts);           \
        trace_elem(w3, e0[C * 4 + 3], s3, dst + 3 * 4 * ROWF, ta, tm, tas, ts);           \
        break;
                    DCLL_TRACE_CASE(0)
                    DCLL_TRACE_CASE(1)
                    DCLL_TRACE_CASE(2)
                    DCLL_TRACE_CASE(3)
#undef DCLL_TRACE_CASE
                }
            }
            if (ABLATE & 8) { unsigned long long x_ = __builtin_amdgcn_s_memtime(); dbg_tr += x_ - dbg_a; dbg_a = x_; }
            if (loaded && tracing && !(PRIO & 4)) __builtin_amdgcn_s_setprio(0);
            // ---- (3) my K-slice of the chain ----
            f32x16 acc;
            if (w == 0 || (ABLATE & 4)) {
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
            // 14 rows of 7 taps (row = (cp, ky)); the B fragments of row r+1 are fetched from LDS before the MFMAs of
            // row r are issued (explicit double buffer, order pinned with sched_barrier).
            // 14 rows of 7 taps (row = (cp, ky)); the B fragments of row r+1 are fetched from LDS before the MFMAs of
            // row r are issued (explicit double buffer, order pinned with sched_barrier).
            // (Tried: four hand-made bases 256 dwords apart to spare the ~22 v_add per wave-stage that ds_read2_b32's
            //  8-bit offsets cost — measured 20 % slower, the compiler's own pairing is better.)
            // LDS index of tap (cp,ky,kx) = i0 + cp*722 + ky*19 + kx.  ds_read2_b32 reaches 255 dwords from its base
            // register; the second channel pair gets its own base (an integer hidden from the optimiser, so the
            // accesses stay LDS-typed) instead of the ~22 per-row v_add the compiler would otherwise re-derive.
            const int i0 = (t & 1) * IMG_FLOATS + bbase + m * 2 * ROWF;
            int i1 = i0 + 2 * CHF;
            if (BASES == 2) asm volatile("" : "+v"(i1));
            auto tapval = [&](int cp, int off) -> float { return (BASES == 2 && cp) ? lds[i1 + off] : lds[i0 + cp * 2 * CHF + off]; };
            float bq[2][7];
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) bq[0][kx] = tapval(0, kx);
#pragma unroll
            for (int r = 0; r < 14; ++r) {
                if (r + 1 < 14) {
                    const int cpn = (r + 1) / 7, kyn = (r + 1) % 7;
#pragma unroll
                    for (int kx = 0; kx < 7; ++kx) bq[(r + 1) & 1][kx] = tapval(cpn, kyn * ROWF + kx);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kx = 0; kx < 7; ++kx)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[r / 7][(r % 7) * 7 + kx], bq[r & 1][kx], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            f32x4 *dp = (f32x4 *)(slots + (w * 2 + (g & 1)) * SLOT_FLOATS) + lane;
            if (ABLATE & 4) {
                if (acc[0] + acc[5] + acc[10] + acc[15] == 12345.678f) dp[0] = f32x4{acc[0], acc[1], acc[2], acc[3]};
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    f32x4 v4 = {acc[4 * c + 0], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]};
                    dp[c * 64] = v4;
                }
            }
        }
        if (loaded && (PRIO & 4)) __builtin_amdgcn_s_setprio(0);
        if (ABLATE & 8) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            unsigned long long x_ = __builtin_amdgcn_s_memtime();
            dbg_mf += x_ - dbg_a;
            __syncthreads();
            dbg_wait += __builtin_amdgcn_s_memtime() - x_;
        } else {
            __syncthreads();
        }
    }
    if ((ABLATE & 8) && lane == 0 && b == 0) {
        unsigned long long tot = __builtin_amdgcn_s_memtime() - dbg_t0;
        unsigned long long *dp = (unsigned long long *)v_out + w * 8;       // v_out doubles as the debug buffer
        dp[0] = tot; dp[1] = dbg_wait; dp[2] = dbg_epi; dp[3] = dbg_tr; dp[4] = dbg_mf;
    }

    // state back to HBM: eps1 of the last step lives in image[(T-1)&1]
    const float *fin = lds + ((T - 1) & 1) * IMG_FLOATS;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        long gidx = (b * 32 + 4 * w + (i >> 2)) * 256 + (i & 3) * 64 + lane;
        eps0_g[gidx] = e0[i];
        eps1_g[gidx] = fin[ioff + (i >> 2) * CHF + (i & 3) * 4 * ROWF];
    }
    if (REFRACTORY) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                arp_g[(b * 32 + rr + 8 * wq + 4 * h) * 256 + 32 * (2 * k + wpar) + j] = arp[k][rr];
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_lif_seq_c32d — k_lif_seq_c32 with TWO pixel tiles per wave and stage (the long-sequence variant).
//
// Same data layout, same weight-stationary systolic chain, same pinned order.  Per stage a wave runs the K-slices of
// the tile pair (2p, 2p+1) — image rows 4p..4p+3 — as two INDEPENDENT accumulator chains:
//   - back-to-back MFMAs of one wave no longer depend on each other (microbenchmark: 98.3 % vs 96.2 % of peak);
//   - the B fragment of tap row ky of the second tile IS the fragment of tap row ky+2 of the first: 9 LDS rows per
//     channel pair feed both tiles (126 ds_read dwords per stage instead of 196);
//   - half as many barriers / hand-off latencies per MFMA;
//   - every wave has the same non-MFMA work in every stage (one epilogue quad-share of the pair finished by wave 7, one
//     channel of the trace update), done by ALL waves at the start of the stage, before any MFMA of the stage is in
//     flight.  Measured at B=4096 (ms per launch): this arrangement 100.7; the same work on one wave of a SIMD while
//     its partner runs its chains (the arrangement of k_lif_seq_c32) 102.2; cut into pieces and interleaved into the
//     wave's own MFMA rows 103.4 — a VALU / LDS instruction issued against an MFMA stream, another wave's or the
//     wave's own, waits for the MFMAs in flight, so the non-MFMA work is cheapest when the matrix pipe is empty anyway.
//   - the stage loop is unrolled by four: the pair index of the epilogue share (g & 3) and the register group of the
//     trace share are then compile-time constants (the traces of channel c of wave w live in register group
//     (c + w) & 3), so the non-MFMA phase is straight-line code whose two halves overlap their latencies.
// The hand-off slots hold two tiles per wave and are therefore single-buffered: a wave reads its inputs at the start
// of a stage and a second barrier orders those reads before this stage's slot writes.  Pipeline fill is 8
// pair-stages, so k_lif_seq_c32 stays the kernel for short sequences.
// ------------------------------------------------------------------------------------------------------------
// DBG is a diagnostic knob for experiments/ablate_c32d.hip only (s_memtime stamps of workgroup 0 into v_out).
template <bool REFRACTORY, int OUT, int DBG = 0>     // OUT bit0: pv, bit1: v
__global__ __launch_bounds__(512) void k_lif_seq_c32d(const uint32_t *__restrict__ spk_in, const dcll_wsrc W,
                                                       const float *__restrict__ bias, const float *__restrict__ tau4,
                                                       float *__restrict__ eps0_g, float *__restrict__ eps1_g,
                                                       float *__restrict__ arp_g, uint32_t *__restrict__ spk_out,
                                                       float *__restrict__ pv_out, float *__restrict__ v_out, int T,
                                                       int B, float alpharp, float wrp)
{
    __shared__ __attribute__((aligned(16))) float lds[2 * IMG_FLOATS + (NWAVE * 2 + 1) * SLOT_FLOATS];
    float *slots = lds + 2 * IMG_FLOATS;        // [wave][tile of the pair][16 x 64]
    float *sbias = slots + NWAVE * 2 * SLOT_FLOATS;     // the bias as a slot-shaped tile: wave 0's chain input
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, j = lane & 31;
    // chain position w of hardware wave i: 0 1 4 5 2 3 6 7 — the two waves of a SIMD (i, i + 4) are then TWO positions apart
    // and work, in every stage, on the tile pairs p and p + 2: one border pair (whose all-padding tap rows are skipped,
    // below) and one inner pair on every SIMD, instead of the same pair on both
    const int wi = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = (DBG & 4) ? wi : ((wi & 1) | ((wi & 4) >> 1) | ((wi & 2) << 1));
    const int wq = w & 3, wpar = w >> 2;        // my epilogue share: quad wq of the pair's tile wpar
    const long b = blockIdx.x;
    // Highest wave priority for the whole kernel: all eight waves carry it, so nothing changes among them, but the waves
    // of ANOTHER kernel resident on the same SIMDs (the co-resident readout of test_sequence(overlap_readout=True),
    // priority 0) only get the issue slots these waves leave idle.
    __builtin_amdgcn_s_setprio(3);

    for (int i = tid; i < 2 * IMG_FLOATS; i += 512) lds[i] = 0.0f;
    // slot layout: float4 c of lane l = accumulator registers 4c..4c+3 = channels (r&3) + 8c + 4(l>>5)
    for (int i = tid; i < SLOT_FLOATS; i += 512) sbias[i] = bias[(i & 3) + 8 * (i >> 8) + 4 * ((i >> 7) & 1)];

    float wf[2][49];
    load_wf_c32(W, j, w, h, wf);

    // eps0 of my 4 channels: register group grp holds channel (grp - w) & 3 (see the header), element ii = pixel
    // ii*64 + lane; eps1 lives in the LDS images at float offset ioff + c*CHF + ii*4*ROWF
    float e0[4][4];
    const int ioff = (4 * w) * CHF + ((lane >> 4) + 3) * ROWF + (lane & 15) + 3;
    const unsigned long long *in_wave = (const unsigned long long *)(spk_in + (b * 32 + 4 * w) * 8);
    const long in_step = (long)B * 32 * 4;      // in 64-bit units
    __syncthreads();        // images zeroed

    // prologue: state from HBM, advanced to step 0 with the input bits of step 0 -> image[0]
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
        const int c = (grp - w) & 3;
        const float ta = tau4[0 * 32 + 4 * w + c], tm = tau4[1 * 32 + 4 * w + c];
        const float tas = tau4[2 * 32 + 4 * w + c], ts = tau4[3 * 32 + 4 * w + c];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const long gidx = (b * 32 + 4 * w + c) * 256 + ii * 64 + lane;
            e0[grp][ii] = eps0_g[gidx];
            float e1 = eps1_g[gidx];
            float xin = (float)((in_wave[c * 4 + ii] >> lane) & 1ull);
            trace_update(xin, ta, tm, tas, ts, e0[grp][ii], e1);
            lds[ioff + c * CHF + ii * 4 * ROWF] = e1;
        }
    }
    // refractory trace of my epilogue share: tiles m = 2k + wpar (k = pair index), quad wq
    float arp[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
            arp[k][rr] = REFRACTORY ? arp_g[(b * 32 + rr + 8 * wq + 4 * h) * 256 + 32 * (2 * k + wpar) + j] : 0.0f;

    // DBG & 2 (experiments/ablate_c32d.hip only, WRONG results): the second image row of a tile is read 16 instead of
    // ROWF = 19 floats behind the first, which makes the B-fragment reads bank-conflict free — isolates what the 2-way
    // conflicts on three banks of the real layout cost.
    const int bbase = (4 * w + h) * CHF + (j >> 4) * ((DBG & 2) ? 16 : ROWF) + (j & 15);
    // epilogue stores: lane byte offset inside the (8 channels x 256 pixels) block of quad wq — channel rr + 4 h, pixel
    // 32 (2 U + wpar) + j; the (rr, U) part is an immediate of each store
    const unsigned eoff = 4u * (4 * h * 256 + 32 * wpar + j);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) asm volatile("" ::"v"(arp[k][rr]));
#pragma unroll
    for (int grp = 0; grp < 4; ++grp)
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) asm volatile("" ::"v"(e0[grp][ii]));
#pragma unroll
    for (int cp = 0; cp < 2; ++cp)
#pragma unroll
        for (int k = 0; k < 49; ++k) asm volatile("" ::"v"(wf[cp][k]));
    __syncthreads();

    unsigned long long dbg[4] = {0, 0, 0, 0}, dbg_t0 = 0;       // DBG only: non-MFMA phase, barrier 2, chains, barrier 1
    if (DBG & 1) dbg_t0 = __builtin_amdgcn_s_memtime();
    // inputs of a wave's trace share of stage g (wave-uniform scalars + 4 eps1 values), fetched one stage ahead
    float sv[4] = {0.f, 0.f, 0.f, 0.f}, ta = 0.f, tm = 0.f, tas = 0.f, ts = 0.f;
    unsigned long long wm[4] = {0, 0, 0, 0};
    auto fetch_trace_inputs = [&](const int g) {
        const int q = g - w;
        if (q >= 0 && q < 4 * T && (q >> 2) + 1 < T) {
            const int p = q & 3, t = q >> 2;
            const float *src = lds + (t & 1) * IMG_FLOATS + ioff + p * CHF;
#pragma unroll
            for (int i = 0; i < 4; ++i) sv[i] = src[i * 4 * ROWF];
            const unsigned long long *ip = in_wave + (long)(t + 1) * in_step + p * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) wm[i] = ip[i];
            ta = tau4[0 * 32 + 4 * w + p]; tm = tau4[1 * 32 + 4 * w + p];
            tas = tau4[2 * 32 + 4 * w + p]; ts = tau4[3 * 32 + 4 * w + p];
        }
    };
    fetch_trace_inputs(0);
    // one stage; U = g & 3 at compile time
    auto stage = [&](const int g, auto UC) {
        constexpr int U = decltype(UC)::value;
        unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0;
        if (DBG & 1) st0 = __builtin_amdgcn_s_memtime();
        const int q = g - w;
        const bool active = q >= 0 && q < 4 * T;
        const int p = q & 3, t = q >> 2;
        // ---- (0) everything this stage reads from LDS / SMEM goes out first ----
        //   chain inputs out of the slots (written in the previous stage)
        f32x16 accA, accB;
        if (active) {
            // wave 0 starts both chains from the bias tile, wave w > 0 from the two tiles wave w-1 left: one code path
            // (a branch here costs 32 v_mov per stage to merge the accumulator tuples)
            const float *inA = (w == 0) ? sbias : slots + ((w - 1) * 2) * SLOT_FLOATS;
            const float *inB = (w == 0) ? sbias : slots + ((w - 1) * 2 + 1) * SLOT_FLOATS;
            const f32x4 *spa = (const f32x4 *)inA + lane, *spb = (const f32x4 *)inB + lane;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f32x4 va = spa[c * 64], vb = spb[c * 64];
                accA[4 * c + 0] = va[0]; accA[4 * c + 1] = va[1]; accA[4 * c + 2] = va[2]; accA[4 * c + 3] = va[3];
                accB[4 * c + 0] = vb[0]; accB[4 * c + 1] = vb[1]; accB[4 * c + 2] = vb[2]; accB[4 * c + 3] = vb[3];
            }
        }
        //   epilogue share: quad wq of tile wpar of the pair qe = g - 8 (pair index U) that wave 7 finished last stage
        const int qe = g - 8;
        const bool epi = qe >= 0 && qe < 4 * T;
        f32x4 v4 = {0.f, 0.f, 0.f, 0.f};
        if (epi) v4 = *((const f32x4 *)(slots + (7 * 2 + wpar) * SLOT_FLOATS) + wq * 64 + lane);
        //   trace share: channel p of step t+1 (register group U), reads image[t&1], writes image[(t+1)&1]; its inputs
        //   (eps1 values sv, input masks wm, time constants) were fetched during the previous stage's chains
        const bool tr = active && t + 1 < T;
        //   first B-fragment row of the chains
        const int i0 = (t & 1) * IMG_FLOATS + bbase + p * 4 * ROWF;
        // tap rows that lie in the zero padding for BOTH image rows of a tile are skipped — fmaf(w, 0, acc) == acc: the pair
        // p = 0 (image rows 0..3) has none of tile A's tap rows ky = 0, 1 (LDS rows rho = 0, 1), the pair p = 3 none of tile
        // B's ky = 5, 6 (rho = 7, 8): 168 instead of 196 MFMAs for those two of the four pairs
        const bool skip = !(DBG & 4);
        const int rl = (skip && p == 0) ? 2 : 0;        // first LDS row of my chains
        // B-fragment bases of my two channel pairs as opaque 32-bit LDS addresses: every read of the chains is base +
        // immediate (9 rows x 19 floats + 7 taps = 158 dwords, inside ds_read2_b32's 8-bit offsets).  Left visible, the
        // pair stride (722 dwords) does not fit and the compiler rebuilds a base for 27 of the reads — vector instructions
        // on the pipe the MFMAs execute on (experiments/isa_blocks.py)
        lds_cfloat *ib0 = (lds_cfloat *)(lds + i0), *ib1 = (lds_cfloat *)(lds + i0 + 2 * CHF);
        asm volatile("" : "+v"(ib0), "+v"(ib1));
        float bq[2][7];
        if (active) {
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) bq[0][kx] = ib0[rl * ROWF + kx];
        }
        // ---- (2) trace share (dcll/pytorch_libdcll.py:493-494, every op rounded separately) ----
        if (tr) {
            float *dst = lds + ((t + 1) & 1) * IMG_FLOATS + ioff + p * CHF;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float a;                                    // x * tau_s with x in {0,1}: exact select
                asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(a) : "v"(ts), "s"(wm[i]));
                const float bb = tas * e0[U][i];
                e0[U][i] = a + bb;
                const float cc = ta * sv[i];
                const float dd = e0[U][i] * tm;
                dst[i * 4 * ROWF] = cc + dd;
            }
        }
        // ---- (1) epilogue share ----
        if (epi) {
            const int te = qe >> 2, me = 2 * U + wpar;
            // buffer stores: descriptor = channel 8 wq of this step and sample (wave-uniform, SGPRs) + the loop-invariant
            // lane byte offset + an IMMEDIATE per value and pair (rr * 256 + 64 U floats): no address registers at all
            // (flat stores kept a 64-bit lane pointer per U alive across the loop — with the three chain variants below
            // they were spilled, and their reload sat in this phase)
            const long ubase = (((long)te * B + b) * 32 + 8 * wq) * 256;
            const auto prs = tile_rsrc(pv_out + ubase), vrs = tile_rsrc(v_out + ubase);
            uint32_t myword = 0;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                float v = v4[rr];
                bool s;
                if (REFRACTORY) v = refractory(v4[rr], arp[U][rr], alpharp, wrp, s);
                else s = v > 0.0f;
                const unsigned long long mk = __ballot(s);
                const uint32_t mine = h ? (uint32_t)(mk >> 32) : (uint32_t)mk;
                myword = (j == rr) ? mine : myword;
                if (OUT & 1) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sigmoidf_dev(v)), prs, eoff + 4u * (rr * 256 + 64 * U), 0, 0);
                if (OUT & 2) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), vrs, eoff + 4u * (rr * 256 + 64 * U), 0, 0);
            }
            if (spk_out && j < 4)
                __builtin_amdgcn_raw_buffer_store_b32(myword, tile_rsrc(spk_out + (ubase >> 5)), 4u * ((4 * h + j) * 8 + wpar) + 8u * U, 0, 0);
        }
        if (DBG & 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); st1 = __builtin_amdgcn_s_memtime(); }
        // every slot read of this stage has completed before any wave writes its slots again
        lds_barrier();
        if (DBG & 1) st2 = __builtin_amdgcn_s_memtime();
        // inputs of the NEXT stage's trace share: they land while the chains run (the eps1 values it reads were
        // written by my own trace share four stages ago; nobody else touches my channels).  Issued before the first
        // MFMA: in the middle of the chains the same loads cost 1.5 % (24.65 vs 24.27 ms at B=1024).
        fetch_trace_inputs(g + 1);
        // ---- (3) my K-slice of both chains ----
        if (active) {
            // LDS rows RL..RH-1 below the pair's first image row, per channel pair: row rho is tap row ky = rho of tile A
            // (rho <= 6) and tap row ky = rho - 2 of tile B (rho >= 2); next row fetched before the MFMAs of this one.
            auto chains = [&](auto RLC, auto RHC) {
                constexpr int RL = decltype(RLC)::value, RH = decltype(RHC)::value, NRH = RH - RL, NR = 2 * NRH;
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int cp = r / NRH, rho = RL + r % NRH;
                    if (r + 1 < NR) {
                        const int cpn = (r + 1) / NRH, rhon = RL + (r + 1) % NRH;
#pragma unroll
                        for (int kx = 0; kx < 7; ++kx) bq[(r + 1) & 1][kx] = (cpn ? ib1 : ib0)[rhon * ROWF + kx];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kx = 0; kx < 7; ++kx) {
                        if (rho <= 6)
                            accA = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[cp][rho * 7 + kx], bq[r & 1][kx], accA, 0, 0, 0);
                        if (rho >= 2)
                            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[cp][(rho - 2) * 7 + kx], bq[r & 1][kx], accB, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (cp == 1 && rho == 6 && r + 1 < NR) {     // tile A is complete (its last tap row) while B still runs
                        f32x4 *dpa = (f32x4 *)(slots + (w * 2) * SLOT_FLOATS) + lane;
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            dpa[c * 64] = f32x4{accA[4 * c + 0], accA[4 * c + 1], accA[4 * c + 2], accA[4 * c + 3]};
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (RH == 7) {          // (last pair: both tiles end with the same row)
                    f32x4 *dpa = (f32x4 *)(slots + (w * 2) * SLOT_FLOATS) + lane;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        dpa[c * 64] = f32x4{accA[4 * c + 0], accA[4 * c + 1], accA[4 * c + 2], accA[4 * c + 3]};
                }
            };
            if (skip && p == 0) chains(std::integral_constant<int, 2>{}, std::integral_constant<int, 9>{});
            else if (skip && p == 3) chains(std::integral_constant<int, 0>{}, std::integral_constant<int, 7>{});
            else chains(std::integral_constant<int, 0>{}, std::integral_constant<int, 9>{});
            f32x4 *dp = (f32x4 *)(slots + (w * 2 + 1) * SLOT_FLOATS) + lane;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                dp[c * 64] = f32x4{accB[4 * c + 0], accB[4 * c + 1], accB[4 * c + 2], accB[4 * c + 3]};
        }
        if (DBG & 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); st3 = __builtin_amdgcn_s_memtime(); }
        // stage barrier: only the LDS traffic has to be complete, not the pv / spike stores of the epilogue
        lds_barrier();
        if (DBG & 1) {
            const unsigned long long st4 = __builtin_amdgcn_s_memtime();
            dbg[0] += st1 - st0; dbg[1] += st2 - st1; dbg[2] += st3 - st2; dbg[3] += st4 - st3;
        }
    };

    const int nstage = 4 * T + 8;       // a multiple of 4
    for (int g = 0; g < nstage; g += 4) {
        stage(g + 0, std::integral_constant<int, 0>{});
        stage(g + 1, std::integral_constant<int, 1>{});
        stage(g + 2, std::integral_constant<int, 2>{});
        stage(g + 3, std::integral_constant<int, 3>{});
    }

    if ((DBG & 1) && lane == 0 && b == 0) {
        unsigned long long *dp = (unsigned long long *)v_out + w * 8;       // v_out doubles as the debug buffer
        dp[0] = __builtin_amdgcn_s_memtime() - dbg_t0;
        dp[1] = dbg[0]; dp[2] = dbg[1]; dp[3] = dbg[2]; dp[4] = dbg[3];
    }
    // state back to HBM: eps1 of the last step lives in image[(T-1)&1]
    const float *fin = lds + ((T - 1) & 1) * IMG_FLOATS;
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
        const int c = (grp - w) & 3;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const long gidx = (b * 32 + 4 * w + c) * 256 + ii * 64 + lane;
            eps0_g[gidx] = e0[grp][ii];
            eps1_g[gidx] = fin[ioff + c * CHF + ii * 4 * ROWF];
        }
    }
    if (REFRACTORY) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                arp_g[(b * 32 + rr + 8 * wq + 4 * h) * 256 + 32 * (2 * k + wpar) + j] = arp[k][rr];
    }
}

// readout weights (N, 32*256) [n][co][pix]  ->  epilogue layout [me][wq][n][lane][rr]:
//   co = rr + 8*wq + 4*(lane>>5), pix = 32*me + (lane&31)
__global__ void k_permute_readout(const float *__restrict__ Wt, float *__restrict__ Wp, int N)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * 8192) return;
    int rr = i & 3, lane = (i >> 2) & 63;
    int r = i >> 8;
    int n = r % N;
    r /= N;
    int wq = r & 3, me = r >> 2;
    int co = rr + 8 * wq + 4 * (lane >> 5), pix = 32 * me + (lane & 31);
    Wp[i] = Wt[(long)n * 8192 + co * 256 + pix];
}

// ------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------
static inline unsigned nblk(long n, int bs) { return (unsigned)((n + bs - 1) / bs); }

// mode: DCLL_READOUT_AUTO, _CORESIDENT (k_readout_direct in its <= 64 VGPR, LDS-free form whenever the shape allows), _LDS
// (never k_readout_direct: the LDS-staged 32x32x2 kernels; kept for measurements)
static int launch_readout(const float *pv, const float *Wt, const float *bias, float *out, long rows, int K, int N,
                          hipStream_t st, int mode = DCLL_READOUT_AUTO)
{
    if (rows == 0 || N == 0) return DCLL_OK;
    const bool fast = (K % RO_KC == 0) && N <= 64 && (((uintptr_t)pv | (uintptr_t)Wt) & 15) == 0 &&
                      K < (1 << 22);     // (32-bit buffer offsets inside a workgroup's 128 rows)
    const bool direct_ok = fast && K % 64 == 0 && N <= 48 && K <= 16384 && mode != DCLL_READOUT_LDS;
    // (standalone the LDS-free kernel is slower than the LDS-staged ones — 4.6 vs 3.5 ms for 24 rows, 6.9 vs 5.3 ms for 48
    //  at B = 4096: its fragment-layout loads give the texture addresser one 16-byte piece per lane — so it only serves
    //  the co-resident mode)
    if (direct_ok && mode == DCLL_READOUT_CORESIDENT)
        return dcll_launch_readout_direct(pv, Wt, bias, out, rows, K, N, st);
    if (fast && mode == DCLL_READOUT_T16) return dcll_launch_readout_t16(pv, Wt, bias, out, rows, K, N, 0, st);     // any row count
    if (rows <= 2048) {
        hipLaunchKernelGGL(k_readout_rows, dim3((unsigned)((rows + RS_RB - 1) / RS_RB), (N + RS_NG - 1) / RS_NG), dim3(256),
                           0, st, pv, Wt, bias, out, rows, K, N);
    } else if (fast && K % RK_KC == 0 && K >= 65536 && rows < 256L * RO_ROWS) {
        // long rows (large planes) and fewer 128-row tiles than CUs: 32-row tiles with the K-chunk split over the
        // waves.  (K of the 16x16 plane, 8192, always takes k_readout_v4: its logits then do not depend on how a
        // run is split into launches.)
        if (N <= 32)
            hipLaunchKernelGGL(k_readout_ks<1>, dim3(nblk(rows, RK_ROWS)), dim3(256), 0, st, pv, Wt, bias, out, rows, K, N, 0);
        else
            hipLaunchKernelGGL(k_readout_ks<2>, dim3(nblk(rows, RK_ROWS)), dim3(256), 0, st, pv, Wt, bias, out, rows, K, N, 0);
    } else if (fast && mode != DCLL_READOUT_LDS) {
        // 16x16x4 tiles: 24 rows = 2 tiles, 48 stacked rows = 3 tiles without padding (2.9 / 3.9 ms at B = 4096 vs 3.5 / 5.3 ms
        // for the 32-column tiles of k_readout_v4 below, which stay for DCLL_READOUT_LDS)
        return dcll_launch_readout_t16(pv, Wt, bias, out, rows, K, N, 0, st);
    } else if (fast && N <= 32) {
        hipLaunchKernelGGL(k_readout_v4<1>, dim3(nblk(rows, RO_ROWS)), dim3(256), 0, st, pv, Wt, bias, out, rows, K, N);
    } else if (fast) {
        hipLaunchKernelGGL(k_readout_v4<2>, dim3(nblk(rows, RO_ROWS)), dim3(256), 0, st, pv, Wt, bias, out, rows, K, N);
    } else {
        dim3 grid(nblk(rows, RO_ROWS), (N + 31) / 32);
        hipLaunchKernelGGL(k_readout, grid, dim3(256), 0, st, pv, Wt, bias, out, rows, K, N);
    }
    HIP_CHECK_LAUNCH("k_readout");
    return DCLL_OK;
}

extern "C" int dcll_readout(const float *pv, const float *Wt, const float *bias, float *out, int64_t rows, int32_t K,
                            int32_t N, void *stream)
{
    if (rows == 0) return DCLL_OK;
    if (!pv || !Wt || !out || rows < 0 || K < 1 || N < 1) return fail(DCLL_ERR_INVALID, "dcll_readout: bad argument");
    return launch_readout(pv, Wt, bias, out, rows, K, N, (hipStream_t)stream);
}

extern "C" int dcll_readout_mode(const float *pv, const float *Wt, const float *bias, float *out, int64_t rows,
                                 int32_t K, int32_t N, int32_t mode, void *stream)
{
    if (rows == 0) return DCLL_OK;
    if (!pv || !Wt || !out || rows < 0 || K < 1 || N < 1 || mode < DCLL_READOUT_AUTO || mode > DCLL_READOUT_T16)
        return fail(DCLL_ERR_INVALID, "dcll_readout_mode: bad argument");
    return launch_readout(pv, Wt, bias, out, rows, K, N, (hipStream_t)stream, mode);
}

// out[i] = bias + the slices' partial values added in slice order.  Four threads per output (each its quarter of the
// slices, the four sub-sums combined in fixed order through LDS): a quarter of the dependent-load latency.
__global__ __launch_bounds__(256) void k_readout_sum(const float *__restrict__ part, const float *__restrict__ bias,
                                                      float *__restrict__ out, long n_out, int N, int nslice)
{
    __shared__ float red[4][64];
    const long i = (long)blockIdx.x * 64 + (threadIdx.x & 63);
    const int grp = threadIdx.x >> 6;
    float acc = 0.0f;
    if (i < n_out) {
        const int per = (nslice + 3) / 4, s0 = grp * per, s1 = min(nslice, s0 + per);
        for (int sl = s0; sl < s1; ++sl) acc += part[(long)sl * n_out + i];
    }
    red[grp][threadIdx.x & 63] = acc;
    __syncthreads();
    if (grp == 0 && i < n_out) {
        const int l = threadIdx.x & 63;
        out[i] = (bias ? bias[i % N] : 0.0f) + (((red[0][l] + red[1][l]) + red[2][l]) + red[3][l]);
    }
}

// K slice of a split-K readout: 4096 for very long rows (large planes: k_readout_ks), 256 for the per-step calls on the
// 16x16 plane (rows = batch <= 2048, K = 8192: 32 slices x rows/128 workgroups of k_readout_t16 instead of rows/4 x N/4
// workgroups of k_readout_rows that re-read every pv row N/4 times); 0 = shape not supported
static int splitk_slice(int64_t rows, int32_t K, int32_t N)
{
    if (rows < 1 || N < 1 || N > 64) return 0;
    if (K >= 65536 && K % 4096 == 0) {
        // very long rows (large planes).  Few of them (per-step calls): 4096-column slices through k_readout_ks.  Many
        // (the sequence path: rows = T*B, e.g. 8192 x 524288): 8 slices through k_readout_t16 — 8 x rows/128 workgroups with
        // b128 LDS staging instead of rows/32 workgroups of k_readout_ks (3.2 TB/s there).  The slice count does not depend on
        // the row count, so a row's logits do not depend on how a batch is chunked.
        return rows <= 2048 ? 4096 : K / 8;
    }
    // (rows <= 512: 64 slices of 128 — 32 slices x rows/128 row tiles would leave half of the 256 CUs without a workgroup)
    if (rows <= 2048 && K >= 2048 && K < 65536 && K % 256 == 0) return (rows <= 512 && N <= 32) ? 128 : 256;
    return 0;
}

extern "C" int64_t dcll_readout_splitk_scratch(int64_t rows, int32_t K, int32_t N)
{
    const int ks = splitk_slice(rows, K, N);
    return ks ? (int64_t)(K / ks) * rows * N : 0;
}

extern "C" int dcll_readout_splitk(const float *pv, const float *Wt, const float *bias, float *out, float *scratch,
                                   int64_t scratch_floats, int64_t rows, int32_t K, int32_t N, void *stream)
{
    if (rows == 0 || N == 0) return DCLL_OK;
    if (!pv || !Wt || !out || !scratch || rows < 0 || K < 1 || N < 1)
        return fail(DCLL_ERR_INVALID, "dcll_readout_splitk: bad argument");
    const int ks = splitk_slice(rows, K, N);
    if (ks == 0 || ((((uintptr_t)pv | (uintptr_t)Wt)) & 15) != 0)
        return fail(DCLL_ERR_UNSUPPORTED, "dcll_readout_splitk: needs N <= 64, 16-byte aligned operands and K >= 65536 with "
                                          "K % 4096 == 0, or rows <= 2048 with 2048 <= K < 65536, K % 256 == 0");
    if (scratch_floats < dcll_readout_splitk_scratch(rows, K, N))
        return fail(DCLL_ERR_INVALID, "dcll_readout_splitk: scratch too small (dcll_readout_splitk_scratch)");
    hipStream_t st = (hipStream_t)stream;
    const int nslice = K / ks;
    if (ks != 4096) {
        int rc = dcll_launch_readout_t16(pv, Wt, nullptr, scratch, rows, K, N, ks, st);
        if (rc) return rc;
    } else {
        dim3 grid(nblk(rows, RK_ROWS), nslice);
        if (N <= 32) hipLaunchKernelGGL(k_readout_ks<1>, grid, dim3(256), 0, st, pv, Wt, bias, scratch, rows, K, N, 4096);
        else hipLaunchKernelGGL(k_readout_ks<2>, grid, dim3(256), 0, st, pv, Wt, bias, scratch, rows, K, N, 4096);
        HIP_CHECK_LAUNCH("k_readout_ks (split K)");
    }
    hipLaunchKernelGGL(k_readout_sum, dim3(nblk(rows * N, 64)), dim3(256), 0, st, scratch, bias, out, rows * N, N, nslice);
    HIP_CHECK_LAUNCH("k_readout_sum");
    return DCLL_OK;
}

// The readout of the whole-sequence path (rows = T x chunk of the batch): the LDS-staged 16x16x4 kernel, whole for K < 65536,
// in 8 ... 64 K-slices + k_readout_sum for longer rows (large planes) — chosen by K alone, never by the row count, so a row's
// logits do not depend on how a batch is chunked (dcll_readout / dcll_readout_splitk pick by row count: per-step calls).
// act = DCLL_ACT_SIGMOID: pv holds v (dcll_layer_opts pv_presigmoid), the sigmoid is applied to the staged values.
// K-slices of dcll_readout_act: 0 (unsplit) below 65536 columns, else 8 ... 64 slices of >= 8192 columns — by K alone.  (Round
// 4: always eight left the 128x128 plane, K = 524288 and rows = T x 64, with 512 workgroups of 65536 columns each — two per
// CU, 3.9 TB/s; 64 slices are 4096 workgroups.)
static int act_nslice(int32_t K)
{
    if (K < 65536 || K % 256 != 0) return 0;
    int n = 8;
    while (n < 64 && K / (2 * n) >= 8192 && K % (2 * n * RO_KC) == 0) n *= 2;
    return n;
}
extern "C" int64_t dcll_readout_act_scratch(int64_t rows, int32_t K, int32_t N)
{
    return (rows > 0 && N > 0) ? (int64_t)act_nslice(K) * rows * N : 0;
}

extern "C" int dcll_readout_act(const float *pv, const float *Wt, const float *bias, float *out, float *scratch,
                                int64_t scratch_floats, int64_t rows, int32_t K, int32_t N, int32_t act, void *stream)
{
    if (rows == 0 || N == 0) return DCLL_OK;
    if (!pv || !Wt || !out || rows < 0 || K < 1 || N < 1 || (act != DCLL_ACT_NONE && act != DCLL_ACT_SIGMOID))
        return fail(DCLL_ERR_INVALID, "dcll_readout_act: bad argument");
    if (!(K % RO_KC == 0 && K < (1 << 22) && N <= 64 && (((uintptr_t)pv | (uintptr_t)Wt) & 15) == 0))
        return fail(DCLL_ERR_UNSUPPORTED, "dcll_readout_act: needs K % 32 == 0, K < 2^22, N <= 64, 16-byte aligned pv / Wt");
    hipStream_t st = (hipStream_t)stream;
    const int64_t need = dcll_readout_act_scratch(rows, K, N);
    if (need == 0) return dcll_launch_readout_t16(pv, Wt, bias, out, rows, K, N, 0, st, act);
    if (!scratch || scratch_floats < need)
        return fail(DCLL_ERR_INVALID, "dcll_readout_act: K >= 65536 is split over K and needs scratch (dcll_readout_act_scratch)");
    const int ns = act_nslice(K);
    int rc = dcll_launch_readout_t16(pv, Wt, nullptr, scratch, rows, K, N, K / ns, st, act);
    if (rc) return rc;
    hipLaunchKernelGGL(k_readout_sum, dim3(nblk(rows * N, 64)), dim3(256), 0, st, scratch, bias, out, rows * N, N, ns);
    HIP_CHECK_LAUNCH("k_readout_sum");
    return DCLL_OK;
}

// ------------------------------------------------------------------------------------------------------------
// k_lif_step_c32 — ONE timestep of a 32 -> 32 channel 7x7 layer on the 16x16 plane (the per-step drop-in and the
// forward of a local-learning step), one sample per 256-thread workgroup, state through HBM.
//
// The sequence kernels hand a tile's accumulator from wave to wave; for a single step that systolic pipeline would
// spend 9 of its 17 stages filling and draining (k_lif_seq_c32 at T = 1: 174 us at B = 512, 48 % of the MFMA time).
// Here every wave keeps a PAIR of pixel tiles (image rows 4w..4w+3) and runs both whole K = 1568 chains itself, in
// the pinned order, as two independent accumulator chains; the weights stream through LDS in 16 chunks of one
// input-channel pair (49 taps x 64 lanes, double-buffered, next chunk fetched into registers during the MFMAs of this
// one).  A row of 7 weight fragments is read once and serves tile A at tap row ky and tile B two rows later; a row of
// B fragments serves both tiles: 112 ds_read dwords per 98 MFMAs.  Input is the dense fp32 map x (any values, not
// only {0,1}), outputs are the dense s / pv / v maps of dcll_conv_lif_step: no packing, no separate trace kernel.
// ------------------------------------------------------------------------------------------------------------
// Weight chunk (A fragments of one input-channel pair) in LDS, LANE-major (round 4): the 49 taps of fragment lane
// l = hh*32 + co are consecutive, the lanes STEP_WLS = 49 floats apart (an odd stride: the 64 lanes of a fragment read hit
// 64 different banks; the copy-in writes consecutive taps of one lane — consecutive addresses).  Every tap of a lane is
// then within ds_read2_b32's 8-bit offset range of ONE per-lane base.  (Round 2-3 layout: tap-major, the taps 65 floats
// apart — 49 taps span 3185 dwords, so the compiler rebuilt a base register for nearly every ds_read2: 56 v_add_u32 per
// 98 MFMAs in the chunk loop, on the pipe the fp32 MFMAs execute on.)
// wch[l * STEP_WLS + tap] = W[co][2cp + hh][tap].  In global memory the 98 floats of (co, channel pair cp) are
// contiguous: thread t of the 256 moves elements t, t+256, ... of the 32 x 98 block through registers (fetched while the
// MFMAs of the previous chunk run); the index arithmetic is done once per kernel.
constexpr int STEP_WLS = 49, STEP_WCH = 64 * STEP_WLS + 4;
// Every global access of the per-step kernels is a BUFFER access (round 4): 128-bit descriptor of a wave-uniform base in
// SGPRs + a loop-invariant 32-bit lane offset + a scalar / immediate offset that walks with the chunk — no 64-bit vector
// address is formed inside the chunk loop (flat accesses: v_add_co / v_addc per load and a v_cndmask + exec-mask branch per
// conditional one, 43 vector instructions and ~10 scalar branches per chunk on the pipe the fp32 MFMAs execute on).  The one
// ragged element set (tid < 64 of i = 12: 3136 = 12 * 256 + 64) loads element 0 and parks it in the chunk's pad slot.
__device__ __forceinline__ float buf_ldf(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}
__device__ __forceinline__ void buf_stf(float v, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, voff, soff, 0);
}
constexpr unsigned BUF_OOB = 0x80000000u;      // beyond tile_rsrc's 2^31 - 1 records: the load returns 0, no access is made
template <bool Q8>             // int8 weights (dcll_layer_opts): a template parameter — as a runtime test the compiler
struct step_wchunk {            // speculates the 26 conversion instructions of the int8 branch into the fp32 path's MFMA block
    static constexpr int NW = 13;
    int raw[NW];                // fetched weights: fp32 bit patterns, or sign-extended int8 values
    float qs[NW];               // int8: the scale of my elements' output channels
    unsigned goff[NW];          // offset of my element i inside a channel pair's slice: bytes (fp32) or elements (int8)
    int loff[NW];
    __amdgpu_buffer_rsrc_t rs;
    __device__ __forceinline__ void init(int tid, const dcll_wsrc &W)
    {
        constexpr bool q8 = Q8;
        rs = q8 ? tile_rsrc(W.q) : tile_rsrc(W.f);
        // (co, r) of element tid + 256 i, stepped from i to i + 1 (256 = 2 * 98 + 60) instead of thirteen divisions by 98
        // and 49: the index arithmetic sits in front of the kernel's first weight request
        int co = tid >= 196 ? 2 : tid >= 98 ? 1 : 0, r = tid - 98 * co;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const bool valid = i < NW - 1 || tid < 32 * 98 - 256 * (NW - 1);
            const int cv = valid ? co : 0, rv = valid ? r : 0, hh = rv >= 49 ? 1 : 0;
            goff[i] = (unsigned)(cv * 1568 + rv) * (q8 ? 1u : 4u);
            loff[i] = valid ? (hh * 32 + cv) * STEP_WLS + (rv - 49 * hh) : 64 * STEP_WLS;
            qs[i] = q8 ? W.scale[cv] : 0.0f;
            r += 60;
            co += 2;
            if (r >= 98) { r -= 98; co += 1; }
        }
    }
    __device__ __forceinline__ void fetch(int cp)          // requests only: nothing here waits for the data
    {
        if (Q8) {
#pragma unroll
            for (int i = 0; i < NW; ++i) raw[i] = (int)(int8_t)__builtin_amdgcn_raw_buffer_load_b8(rs, goff[i], (unsigned)cp * 98u, 0);
            return;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) raw[i] = (int)__builtin_amdgcn_raw_buffer_load_b32(rs, goff[i], (unsigned)cp * 392u, 0);
    }
    __device__ __forceinline__ void store(float *wch, int buf) const
    {
        if (Q8) {                               // int8 weights (dcll_layer_opts): one rounded multiply per weight, here
#pragma unroll
            for (int i = 0; i < NW; ++i) wch[buf * STEP_WCH + loff[i]] = (float)raw[i] * qs[i];
            return;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) wch[buf * STEP_WCH + loff[i]] = __int_as_float(raw[i]);
    }
};

// DBG (experiments/ablate_step.hip only; 0 in the product): 1 no MFMAs, 2 no epilogue stores, 4 no state traffic,
// 8 no weight streaming, 16 no barrier per chunk (8, 16: wrong results, timing only)
template <bool REFRACTORY, int DBG = 0, bool Q8 = false>
__global__ __launch_bounds__(256) void k_lif_step_c32(const float *__restrict__ x, const dcll_wsrc W,
                                                       const float *__restrict__ bias, const float *__restrict__ alpha,
                                                       const float *__restrict__ tau_m, const float *__restrict__ alphas,
                                                       const float *__restrict__ tau_s, int tau_is_tensor,
                                                       float *__restrict__ eps0_g, float *__restrict__ eps1_g,
                                                       float *__restrict__ arp_g, float *__restrict__ out_s,
                                                       float *__restrict__ out_pv, float *__restrict__ out_v,
                                                       float alpharp, float wrp)
{
    __shared__ __attribute__((aligned(16))) float lds[IMG_FLOATS + 2 * STEP_WCH + 32];
    float *img = lds, *wch = lds + IMG_FLOATS, *sbias = wch + 2 * STEP_WCH;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, j = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // my tile pair: image rows 4w..4w+3
    const long b = blockIdx.x;
    // DBG & 32 (experiments/ablate_step.hip): shader-clock stamps of every wave — kernel entry, first MFMA row, end of the
    // chunk loop, end of the epilogue — written behind the arp plane of the launch (timing only)
    unsigned long long stamp0 = 0, stamp1 = 0, stamp2 = 0, real0 = 0;
    if (DBG & 32) {
        stamp0 = __builtin_amdgcn_s_memtime();
        real0 = __builtin_amdgcn_s_memrealtime();           // constant 100 MHz: the shader clock under THIS load
    }

    if (tid < 32) sbias[tid] = bias[tid];
    step_wchunk<Q8> wc;
    auto fetch_w = [&](int cp) { wc.fetch(cp); };
    auto store_w = [&](int buf) { wc.store(wch, buf); };
    // traces of this step (dcll/pytorch_libdcll.py:493-494), state updated in HBM, eps1 -> image — one channel PAIR at a
    // time (thread t owns pixel t of both channels): pair cp + 1 is fetched while the MFMAs of pair cp run and finished
    // (trace arithmetic, state stores, image write) behind them, so that the 160 KB of state traffic per sample is spread
    // over the MFMA phase instead of sitting in front of it (all workgroups of a launch start together: a separate
    // prologue is an HBM-bound phase during which no matrix core works).  Buffer accesses: descriptor of the sample's 32
    // planes + lane offset 4 tid (+ 1 KB for the pair's second channel, an immediate) + scalar offset 2 KB x cp.
    const auto xrs = tile_rsrc(x + b * 8192), e0rs = tile_rsrc(eps0_g + b * 8192), e1rs = tile_rsrc(eps1_g + b * 8192);
    const auto ars = tile_rsrc(alpha), tmrs = tile_rsrc(tau_m), asrs = tile_rsrc(alphas), tsrs = tile_rsrc(tau_s);
    const unsigned tvo = 4u * tid;
    const unsigned tcv0 = tau_is_tensor ? tvo : 0u, tcv1 = tau_is_tensor ? tvo + 1024u : 0u;   // time constants: (C,H,W) or (1)
    const unsigned tcs = tau_is_tensor ? 2048u : 0u;
    float tx[2], te0[2], te1[2], ta[2], ttm[2], tas[2], tts[2];
    auto fetch_t = [&](int cp) {
        const unsigned so = 2048u * cp, tso = tcs * cp;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (DBG & 4) {
                tx[i] = te0[i] = te1[i] = (float)tid;
            } else {
                tx[i] = buf_ldf(xrs, tvo + 1024u * i, so);
                te0[i] = buf_ldf(e0rs, tvo + 1024u * i, so);
                te1[i] = buf_ldf(e1rs, tvo + 1024u * i, so);
            }
            const unsigned tv = i ? tcv1 : tcv0;
            ta[i] = buf_ldf(ars, tv, tso);
            ttm[i] = buf_ldf(tmrs, tv, tso);
            tas[i] = buf_ldf(asrs, tv, tso);
            tts[i] = buf_ldf(tsrs, tv, tso);
        }
    };
    const int ipix = ((tid >> 4) + 3) * ROWF + (tid & 15) + 3;
    auto finish_t = [&](int cp) {
        const unsigned so = 2048u * cp;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            trace_update(tx[i], ta[i], ttm[i], tas[i], tts[i], te0[i], te1[i]);
            if (!(DBG & 4) || te0[i] == 12345.678f) {
                buf_stf(te0[i], e0rs, tvo + 1024u * i, so);
                buf_stf(te1[i], e1rs, tvo + 1024u * i, so);
            }
            img[(2 * cp + i) * CHF + ipix] = te1[i];
        }
    };
    fetch_t(0);             // (the state request first: it needs no index arithmetic — the weight chunk's does, wc.init, and
    wc.init(tid, W);        //  runs while it is in flight; the refractory trace, 32 KB per sample that nothing needs before the
    fetch_w(0);             //  epilogue, is requested BEHIND the first chunk's operands)
    // the refractory trace of my 2 x 16 outputs is requested NOW and lands under the chunk loop: the epilogue of a launch
    // (all workgroups reach it together) is an HBM burst — arp in, s / pv / v / arp out, 160 KB per sample at ~5.6 TB/s —
    // and these 32 KB per sample are the part of it that does not depend on the MFMAs.  Output addressing (also of the
    // epilogue below): descriptor of the sample's 32 planes + lane offset (channel 4h, pixel j) + immediate (r & 3 channels,
    // tile of the pair) + scalar offset (8 (r >> 2) channels, my tile pair)
    const unsigned ovo = 4096u * h + 4u * j, oso = 256u * w;
    const auto aprs = tile_rsrc(arp_g + b * 8192);
    float arp_pre[2][16];
    if (REFRACTORY && !(DBG & 2)) {
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                arp_pre[tl][r] = buf_ldf(aprs, ovo + 1024u * (r & 3) + 128u * tl, oso + 8192u * (r >> 2));
    }
    // (the image is zeroed while those requests are in flight: all workgroups of a launch start together, and the first
    //  chunk cannot begin before the slowest of them has its first operands)
    {
        constexpr int NZ4 = IMG_FLOATS / 4, NZF = NZ4 / 256;      // float4 stores: NZF per thread + a ragged one (immediates)
        f32x4 *z = (f32x4 *)img + tid;
#pragma unroll
        for (int k = 0; k < NZF; ++k) z[256 * k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (tid < NZ4 - 256 * NZF) z[256 * NZF] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    __syncthreads();        // image zeroed
    finish_t(0);
    store_w(0);
    __syncthreads();        // channel pair 0 of the image, bias and chunk 0 in place
    f32x16 accA, accB;
#pragma unroll
    for (int r = 0; r < 16; ++r) accA[r] = accB[r] = sbias[(r & 3) + 8 * (r >> 2) + 4 * h];
    const int bbase = h * CHF + ((j >> 4) + 4 * w) * ROWF + (j & 15);
    if (DBG & 32) stamp1 = __builtin_amdgcn_s_memtime();
    // one chunk = one input-channel pair; the loop runs two chunks per iteration so that the chunk buffer (cp & 1) is a
    // compile-time constant: its LDS offset folds into the immediates of the 13 copy-in writes (13 v_add_u32 per chunk else)
    auto chunk = [&](int cp, auto parity) {
        constexpr int par = decltype(parity)::value;
        if (cp + 1 < 16) {                                 // land during the MFMAs below
            if (!(DBG & 8) && !(DBG & 128)) fetch_w(cp + 1);
            fetch_t(cp + 1);
        }
        // per-lane bases as opaque 32-bit LDS addresses: every operand read below is base + immediate (ds_read2's 8-bit
        // dword offsets reach all 49 taps / all 9 rows).  Left visible, the arrays' static LDS offsets (the chunk buffers
        // sit 47 KB into the allocation) do not fit the immediates and the compiler rebuilds a base per read.
        lds_cfloat *wa = (lds_cfloat *)(wch + par * STEP_WCH + lane * STEP_WLS);
        lds_cfloat *ib = (lds_cfloat *)(img + bbase + cp * 2 * CHF);
        asm volatile("" : "+v"(wa), "+v"(ib));
        // LDS rows rho = 0..8 below the pair's first image row: row rho is tap row ky = rho of tile A (rho <= 6) and
        // tap row ky = rho - 2 of tile B (rho >= 2); the weight fragments of tap row ky are read once (at rho = ky)
        // and kept for tile B two rows later
        float wr[3][7];
#pragma unroll
        for (int rho = 0; rho < 9; ++rho) {
            float bq[7];
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) bq[kx] = ib[rho * ROWF + kx];
            if (rho <= 6) {
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) wr[rho % 3][kx] = wa[rho * 7 + kx];
            }
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                if (DBG & 1) {
                    if (rho <= 6) accA[kx] += wr[rho % 3][kx] * bq[kx];
                    if (rho >= 2) accB[kx] += wr[(rho - 2) % 3][kx] * bq[kx];
                    continue;
                }
                if (rho <= 6) accA = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[rho % 3][kx], bq[kx], accA, 0, 0, 0);
                if (rho >= 2) accB = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[(rho - 2) % 3][kx], bq[kx], accB, 0, 0, 0);
            }
        }
        if (cp + 1 < 16) {
            if (!(DBG & 8) && !(DBG & 64)) store_w(par ^ 1);      // the other buffer: nobody reads it in this iteration
            finish_t(cp + 1);                              // image channels nobody reads in this iteration
        }
        // LDS-only barrier: it orders the image / chunk writes above against the next chunk's reads; the state stores of
        // finish_t need not have landed (__syncthreads() also waits for their acknowledgement: s_waitcnt vmcnt(0))
        if (!(DBG & 16)) lds_barrier();
    };
    for (int cp = 0; cp < 16; cp += 2) {
        chunk(cp, std::integral_constant<int, 0>{});
        chunk(cp + 1, std::integral_constant<int, 1>{});
    }
    if (DBG & 32) {
        asm volatile("" ::"v"(accA[0]), "v"(accB[0]));
        stamp2 = __builtin_amdgcn_s_memtime();
    }
    // epilogue of my two tiles: channel (r&3) + 8(r>>2) + 4h, pixel 32(2w + tl) + j
    const auto srs = tile_rsrc(out_s + b * 8192), pvrs = tile_rsrc(out_pv + b * 8192);
    const auto vrs = tile_rsrc((out_v ? out_v : out_pv) + b * 8192);
    auto epilogue = [&](auto want_v) {          // (one wave-uniform test of out_v, not one per value)
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned vo = ovo + 1024u * (r & 3) + 128u * tl, so = oso + 8192u * (r >> 2);
                const float pvm = tl ? accB[r] : accA[r];
                float v = pvm;
                bool s;
                if (REFRACTORY) {
                    float ar = (DBG & 2) ? 0.0f : arp_pre[tl][r];
                    v = refractory(pvm, ar, alpharp, wrp, s);
                    if (!(DBG & 2) || ar == 12345.678f) buf_stf(ar, aprs, vo, so);
                } else {
                    s = v > 0.0f;
                }
                if ((DBG & 2) && v != 12345.678f) continue;
                buf_stf(s ? 1.0f : 0.0f, srs, vo, so);
                buf_stf(sigmoidf_dev(v), pvrs, vo, so);
                if (decltype(want_v)::value) buf_stf(v, vrs, vo, so);
            }
    };
    if (out_v) epilogue(std::true_type{});
    else epilogue(std::false_type{});
    if ((DBG & 32) && lane == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long stamp3 = __builtin_amdgcn_s_memtime(), real3 = __builtin_amdgcn_s_memrealtime();
        unsigned long long *dst = (unsigned long long *)(arp_g + (long)gridDim.x * 8192) + (b * 4 + w) * 6;
        dst[0] = stamp0; dst[1] = stamp1; dst[2] = stamp2; dst[3] = stamp3; dst[4] = real0; dst[5] = real3;
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_lif_step_c32t — the conv + neuron part of one step of a 32 -> 32 layer on planes larger than 16x16 (h % 16 == 0,
// w % 16 == 0; the reference's argparse default is 128x128): one workgroup per 16x16 tile of a sample, the same pair of
// MFMA tiles per wave and the same weight chunks through LDS as k_lif_step_c32.  The traces are advanced by a separate
// elementwise pass BEFORE this kernel (k_trace4, in place): a tile needs the new eps1 of its 3-pixel halo, which belongs to
// neighbouring workgroups — fused into this kernel that meant a snapshot of the state (2 x 134 MB copied per layer step
// at B = 64, 128x128), seven loads per region element (x, eps0, eps1, four time constants) and a 1.9x redundant trace
// update; measured 1.19 ms + 0.21 ms of copies per layer step against 0.17 + 0.8 ms for the two-pass form.  Here the eps1
// image is staged per channel PAIR (the 22x22 region of the tile, zero outside the plane; double-buffered with the
// weight chunk: 7.7 + 25.5 KB of LDS): one load per region element.
// TH = tile height: 16 (a wave = two MFMA tiles, image rows 4w..4w+3 of the tile) or 8 (a wave = one MFMA tile, rows 2w,
// 2w+1): with 8-row tiles a 16x16 plane is TWO workgroups per sample — used when the batch has fewer samples than half
// the CUs (one workgroup per sample runs 69 us however small the batch is).
// ------------------------------------------------------------------------------------------------------------
constexpr int ST_RF = 22;                                          // region row stride
template <bool REFRACTORY, int TH, bool Q8 = false>
__global__ __launch_bounds__(256) void k_lif_step_c32t(const dcll_wsrc W, const float *__restrict__ bias,
                                                        const float *__restrict__ eps1_g, float *__restrict__ arp_g,
                                                        float *__restrict__ out_s, float *__restrict__ out_pv,
                                                        float *__restrict__ out_v, int H, int Wd, float alpharp, float wrp)
{
    constexpr int ST_CF = (TH + 6) * ST_RF, ST_PAIR = 2 * ST_CF;   // floats per channel / per channel pair of the region
    __shared__ __attribute__((aligned(16))) float lds[2 * ST_PAIR + 2 * STEP_WCH + 32];
    float *img = lds, *wch = lds + 2 * ST_PAIR, *sbias = wch + 2 * STEP_WCH;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, j = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // my MFMA tile(s): tile rows (TH/4) w ...
    const int tpr = Wd >> 4, tps = (H / TH) * tpr;               // TH x 16 tiles per row / per sample
    const long b = blockIdx.x / tps;
    const int tile = blockIdx.x % tps, y0 = (tile / tpr) * TH, x0 = (tile % tpr) * 16;
    const long HW = (long)H * Wd;
    if (tid < 32) sbias[tid] = bias[tid];
    step_wchunk<Q8> wc;                                          // weight chunks exactly as in k_lif_step_c32
    wc.init(tid, W);
    auto fetch_w = [&](int cp) { wc.fetch(cp); };
    auto store_w = [&](int buf) { wc.store(wch, buf); };
    // eps1 of one channel pair over the tile's (TH+6) x 22 region: element e = tid + 256 i of the 2 x ST_CF; addresses:
    // wave-uniform base of the sample + a 32-bit offset inside its 32 planes
    constexpr int NT = (ST_PAIR + 255) / 256;
    unsigned toff[NT];          // byte offset inside the channel pair's two planes; outside the plane (zero) or no element:
                                // BUF_OOB — the buffer load returns 0 without a branch or an access
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int e = tid + 256 * i, r = e % ST_CF, ry = r / ST_RF, rx = r % ST_RF;
        const int gy = y0 + ry - 3, gx = x0 + rx - 3;
        const bool in = e < ST_PAIR && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)Wd;
        toff[i] = in ? 4u * (unsigned)((e >= ST_CF ? (int)HW : 0) + gy * Wd + gx) : BUF_OOB;
    }
    const auto e1rs = tile_rsrc(eps1_g + b * 32 * HW);
    const unsigned pair32 = 8u * (unsigned)HW;         // bytes per channel pair
    float te1[NT];
    auto fetch_t = [&](int cp) {
#pragma unroll
        for (int i = 0; i < NT; ++i) te1[i] = buf_ldf(e1rs, toff[i], (unsigned)cp * pair32);
    };
    auto store_t = [&](int buf) {
        float *dst = img + buf * ST_PAIR;
#pragma unroll
        for (int i = 0; i < NT; ++i)
            if (tid + 256 * i < ST_PAIR) dst[tid + 256 * i] = te1[i];
    };
    fetch_w(0);
    fetch_t(0);
    store_t(0);
    store_w(0);
    __syncthreads();        // channel pair 0 of the image, bias and chunk 0 in place
    f32x16 accA, accB;
#pragma unroll
    for (int r = 0; r < 16; ++r) accA[r] = accB[r] = sbias[(r & 3) + 8 * (r >> 2) + 4 * h];
    const int bbase = h * ST_CF + ((j >> 4) + (TH / 4) * w) * ST_RF + (j & 15);
    // the refractory trace of my outputs, requested now: it lands under the chunk loop instead of being waited for, load
    // by load, in the epilogue (k_lif_step_c32: epilogue 37k -> 12k cycles)
    // output addressing (here and in the epilogue): descriptor of the sample's 32 planes + lane offset (channel 4h, my
    // pixel of MFMA tile tq) + scalar offset (channel (r & 3) + 8 (r >> 2)): no 64-bit address per value
    const auto aprs = tile_rsrc(arp_g + b * 32 * HW);
    unsigned ovo[TH / 8];
#pragma unroll
    for (int tq = 0; tq < TH / 8; ++tq) {
        const int p = 32 * ((TH / 8) * w + tq) + j;
        ovo[tq] = 4u * (unsigned)(4 * h * (int)HW + (y0 + (p >> 4)) * Wd + x0 + (p & 15));
    }
    const unsigned plane4 = 4u * (unsigned)HW;
    float arp_pre[TH / 8][16];
    if (REFRACTORY) {
#pragma unroll
        for (int tq = 0; tq < TH / 8; ++tq)
#pragma unroll
            for (int r = 0; r < 16; ++r) arp_pre[tq][r] = buf_ldf(aprs, ovo[tq], plane4 * ((r & 3) + 8 * (r >> 2)));
    }
    auto chunk = [&](int cp, auto parity) {     // two chunks per loop iteration: the buffer index is a constant (k_lif_step_c32)
        constexpr int par = decltype(parity)::value;
        if (cp + 1 < 16) {                                 // land during the MFMAs below
            fetch_w(cp + 1);
            fetch_t(cp + 1);
        }
        lds_cfloat *wa = (lds_cfloat *)(wch + par * STEP_WCH + lane * STEP_WLS);      // opaque bases: see k_lif_step_c32
        lds_cfloat *ib = (lds_cfloat *)(img + par * ST_PAIR + bbase);
        asm volatile("" : "+v"(wa), "+v"(ib));
        float wr[3][7];
#pragma unroll
        for (int rho = 0; rho < (TH == 16 ? 9 : 7); ++rho) {
            float bq[7];
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) bq[kx] = ib[rho * ST_RF + kx];
            if (rho <= 6) {
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) wr[rho % 3][kx] = wa[rho * 7 + kx];
            }
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                if (rho <= 6) accA = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[rho % 3][kx], bq[kx], accA, 0, 0, 0);
                if (TH == 16 && rho >= 2)
                    accB = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[(rho - 2) % 3][kx], bq[kx], accB, 0, 0, 0);
            }
        }
        if (cp + 1 < 16) {
            store_w(par ^ 1);                              // the other buffers: nobody reads them in this iteration
            store_t(par ^ 1);
        }
        __syncthreads();
    };
    for (int cp = 0; cp < 16; cp += 2) {
        chunk(cp, std::integral_constant<int, 0>{});
        chunk(cp + 1, std::integral_constant<int, 1>{});
    }
    // epilogue of my MFMA tile(s): channel (r&3) + 8(r>>2) + 4h, tile pixel 32((TH/8) w + tq) + j
    const auto srs = tile_rsrc(out_s + b * 32 * HW), pvrs = tile_rsrc(out_pv + b * 32 * HW);
    const auto vrs = tile_rsrc((out_v ? out_v : out_pv) + b * 32 * HW);
    auto epilogue = [&](auto want_v) {
#pragma unroll
        for (int tq = 0; tq < TH / 8; ++tq)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned vo = ovo[tq], so = plane4 * ((r & 3) + 8 * (r >> 2));
                const float pvm = tq ? accB[r] : accA[r];
                float v = pvm;
                bool s;
                if (REFRACTORY) {
                    float ar = arp_pre[tq][r];
                    v = refractory(pvm, ar, alpharp, wrp, s);
                    buf_stf(ar, aprs, vo, so);
                } else {
                    s = v > 0.0f;
                }
                buf_stf(s ? 1.0f : 0.0f, srs, vo, so);
                buf_stf(sigmoidf_dev(v), pvrs, vo, so);
                if (decltype(want_v)::value) buf_stf(v, vrs, vo, so);
            }
    };
    if (out_v) epilogue(std::true_type{});
    else epilogue(std::false_type{});
}

// The trace update of a whole state tensor, four elements per thread (n % 4 == 0, per_sample % 4 == 0): the pass in
// front of k_lif_step_c32t.
__global__ __launch_bounds__(256) void k_trace4(const f32x4 *__restrict__ x, const float *__restrict__ alpha,
                                                 const float *__restrict__ tau_m, const float *__restrict__ alphas,
                                                 const float *__restrict__ tau_s, f32x4 *__restrict__ eps0,
                                                 f32x4 *__restrict__ eps1, long n4, unsigned per_sample4, int tau_is_tensor)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 a, tm, as, ts;
    if (tau_is_tensor) {
        const unsigned q = (unsigned)(i % per_sample4);
        a = ((const f32x4 *)alpha)[q];
        tm = ((const f32x4 *)tau_m)[q];
        as = ((const f32x4 *)alphas)[q];
        ts = ((const f32x4 *)tau_s)[q];
    } else {
        a = tm = as = ts = (f32x4){0.f, 0.f, 0.f, 0.f};
        a += alpha[0];
        tm += tau_m[0];
        as += alphas[0];
        ts += tau_s[0];
    }
    const f32x4 xv = x[i];
    f32x4 e0 = eps0[i], e1 = eps1[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float p0 = e0[k], p1 = e1[k];
        trace_update(xv[k], a[k], tm[k], as[k], ts[k], p0, p1);
        e0[k] = p0;
        e1[k] = p1;
    }
    eps0[i] = e0;
    eps1[i] = e1;
}

// ------------------------------------------------------------------------------------------------------------
// k_lif_step_c1 — ONE timestep of the first layer (c_in = 1 -> c_out <= 32 channels, 7x7 pad 3, 16x16 plane, pool 1)
// for the per-step drop-in and the forward of a learning step: the MFMA form of k_lif_seq_c1 (25 MFMAs per 32-pixel tile,
// taps paired over the k lanes in their pinned linear order, the pad tap with weight 0) on a dense fp32 input map and
// dense s / pv / v outputs, state through HBM.  One 256-thread workgroup per sample (thread = pixel of the traces, wave w
// = image rows 4w..4w+3 = two tiles).  The kernel is bound by its maps (160 KB per sample); the generic pair
// k_trace + k_conv_lif_tiled it replaces ran 98 VALU FMAs per output with LDS weight broadcasts (41 us at B = 512).
// ------------------------------------------------------------------------------------------------------------
// TILED: planes of several 16x16 tiles (h % 16 == 0, w % 16 == 0): one workgroup per tile of a sample; the traces have
// been advanced by k_trace4 before (the tile reads the new eps1 of its 3-pixel halo, x / eps0 are not touched here).
template <bool REFRACTORY, bool TILED>
__global__ __launch_bounds__(256) void k_lif_step_c1(int c_out, const float *__restrict__ x, const dcll_wsrc W,
                                                      const float *__restrict__ bias, const float *__restrict__ alpha,
                                                      const float *__restrict__ tau_m, const float *__restrict__ alphas,
                                                      const float *__restrict__ tau_s, int tau_is_tensor,
                                                      float *__restrict__ eps0_g, float *__restrict__ eps1_g,
                                                      float *__restrict__ arp_g, float *__restrict__ out_s,
                                                      float *__restrict__ out_pv, float *__restrict__ out_v,
                                                      float alpharp, float wrp, int H, int Wd)
{
    constexpr int PS = 24;                      // plane row stride: 16 + 2*3 padding + the pad tap's column
    __shared__ float plane[22 * PS + 8];
    __shared__ float sbias[32];
    const int pix = threadIdx.x, y = pix >> 4, xx = pix & 15, lane = pix & 63;
    const int w = __builtin_amdgcn_readfirstlane(pix >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int tpr = TILED ? Wd >> 4 : 1, tps = TILED ? (H >> 4) * tpr : 1;
    const long b = blockIdx.x / tps;
    const int tile = blockIdx.x % tps, y0 = (tile / tpr) * 16, x0 = (tile % tpr) * 16;
    const long HW = TILED ? (long)H * Wd : 256;
    if (pix < 32) sbias[pix] = pix < c_out ? bias[pix] : 0.0f;
    // the refractory trace of my 2 x 16 outputs, requested at kernel entry (not load by load in the epilogue)
    float arp_pre[2][16];
    if (REFRACTORY) {
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (r & 3) + 8 * (r >> 2) + 4 * h, p = 32 * (2 * w + tl) + j;
                arp_pre[tl][r] = co < c_out ? arp_g[(b * c_out + co) * HW + (TILED ? (long)(y0 + (p >> 4)) * Wd + x0 + (p & 15) : (long)p)]
                                            : 0.0f;
            }
    }
    // weight fragments: pair p: lane (co = j, tap = 2p + h); the pad tap and channels >= c_out carry 0
    float wf[25];
#pragma unroll
    for (int p = 0; p < 25; ++p) {
        const int tap = 2 * p + h;
        wf[p] = (tap < 49 && j < c_out) ? W.at(j * 49 + tap, j) : 0.0f;
    }
    if (TILED) {
        for (int i = pix; i < 22 * PS + 8; i += 256) {
            const int ry = i / PS, rx = i % PS, gy = y0 + ry - 3, gx = x0 + rx - 3;
            plane[i] = (ry < 22 && rx < 22 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)Wd)
                           ? eps1_g[b * HW + (long)gy * Wd + gx] : 0.0f;
        }
    } else {
        for (int i = pix; i < 22 * PS + 8; i += 256) plane[i] = 0.0f;
        const int ti = tau_is_tensor ? pix : 0;
        float e0 = eps0_g[b * 256 + pix], e1 = eps1_g[b * 256 + pix];
        const float xin = x[b * 256 + pix];
        const float al = alpha[ti], tm = tau_m[ti], as = alphas[ti], ts = tau_s[ti];
        __syncthreads();                            // plane zeroed
        trace_update(xin, al, tm, as, ts, e0, e1);  // dcll/pytorch_libdcll.py:493-494
        eps0_g[b * 256 + pix] = e0;
        eps1_g[b * 256 + pix] = e1;
        plane[(y + 3) * PS + xx + 3] = e1;
    }
    __syncthreads();
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
        const int m = 2 * w + tl;
        const float *bn = plane + (2 * m + (j >> 4)) * PS + (j & 15) + h;               // second tap: + 1
        const float *bx = plane + (2 * m + (j >> 4)) * PS + (j & 15) + h * (PS - 6);    // ... or the next row's first
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = sbias[(r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
        for (int p = 0; p < 25; ++p) {
            const int tap = 2 * p, off = (tap / 7) * PS + tap % 7;
            const bool cross = (tap % 7 == 6) && p < 24;        // p = 24: the partner is the pad tap (weight 0)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[p], cross ? bx[off] : bn[off], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (r & 3) + 8 * (r >> 2) + 4 * h;
            if (co >= c_out) continue;
            const int p = 32 * m + j;
            const long o = (b * c_out + co) * HW + (TILED ? (long)(y0 + (p >> 4)) * Wd + x0 + (p & 15) : (long)p);
            float v = acc[r];
            bool s;
            if (REFRACTORY) {
                float ar = arp_pre[tl][r];
                v = refractory(acc[r], ar, alpharp, wrp, s);
                arp_g[o] = ar;
            } else {
                s = v > 0.0f;
            }
            out_s[o] = s ? 1.0f : 0.0f;
            out_pv[o] = sigmoidf_dev(v);
            if (out_v) out_v[o] = v;
        }
    }
}

extern "C" int dcll_conv_lif_step(const dcll_conv_desc *d, const float *x, const float *Wf, const float *b,
                                  const float *alpha, const float *tau_m, const float *alphas, const float *tau_s,
                                  float *eps0, float *eps1, float *arp, const float *i2o_W, const float *i2o_b,
                                  const float *out_W, const float *out_b, float *out_s, float *out_p, float *out_o,
                                  float *out_pv, float *out_v, float *scratch, const dcll_layer_opts *opts, int32_t B,
                                  void *stream)
{
    int rc = check_desc(d);
    if (rc) return rc;
    if (B == 0) return DCLL_OK;
    rc = check_opts(Wf, opts, false, "dcll_conv_lif_step");
    if (rc) return rc;
    const dcll_wsrc W = make_wsrc(Wf, opts);
    if (!x || !alpha || !tau_m || !alphas || !tau_s || !eps0 || !eps1 || !out_s || !out_pv)
        return fail(DCLL_ERR_INVALID, "dcll_conv_lif_step: null pointer");
    if (d->refractory && !arp) return fail(DCLL_ERR_INVALID, "dcll_conv_lif_step: refractory layer needs arp");
    if (B < 0) return fail(DCLL_ERR_INVALID, "dcll_conv_lif_step: negative batch");
    if (B == 0) return DCLL_OK;
    hipStream_t st = (hipStream_t)stream;
    int ch, cw, ph, pw;
    conv_shape(d, &ch, &cw, &ph, &pw);
    const long per = (long)d->c_in * d->h * d->w, nin = per * B;
    const long nconv = (long)B * d->c_out * ch * cw, npool = (long)B * d->c_out * ph * pw;
    const int K = d->c_out * ph * pw;
    const bool k7 = d->kh == 7 && d->kw == 7 && d->pad_h == 3 && d->pad_w == 3 && d->pool_h == 1 && d->pool_w == 1 && b && plain_conv(d);
    const bool plane16 = d->h == 16 && d->w == 16 && k7;
    // (16x16 plane: two workgroups per sample — 8-row tiles — when the batch alone would leave half the CUs idle)
    const bool split16 = plane16 && B <= split16_max_batch();
    const bool ptr16 = ((((uintptr_t)x | (uintptr_t)eps0 | (uintptr_t)eps1) & 15) == 0) &&
                       (!d->tau_is_tensor || (((uintptr_t)alpha | (uintptr_t)tau_m | (uintptr_t)alphas | (uintptr_t)tau_s) & 15) == 0);
    const bool c1t = d->c_in == 1 && d->c_out <= 32 && k7 && !plane16;
    if (((d->c_in == 32 && d->c_out == 32 && k7 && (!plane16 || split16)) || c1t) && d->h % 16 == 0 && d->w % 16 == 0 &&
        ptr16 && per < (1L << 31)) {                            // (32-bit offsets inside a sample's planes)
        // larger planes: the traces in an elementwise pass, then one MFMA workgroup per 16x16 tile
        hipLaunchKernelGGL(k_trace4, dim3(nblk(nin / 4, 256)), dim3(256), 0, st, (const f32x4 *)x, alpha, tau_m, alphas, tau_s,
                           (f32x4 *)eps0, (f32x4 *)eps1, nin / 4, (unsigned)(per / 4), d->tau_is_tensor);
        HIP_CHECK_LAUNCH("k_trace4");
        const long njob = (long)B * (d->h / (split16 ? 8 : 16)) * (d->w / 16);
        if (njob > 0x7fffffffL) return fail(DCLL_ERR_UNSUPPORTED, "dcll_conv_lif_step: more than 2^31 tiles");
#define DCLL_STEP_TQ(R_, TH_, Q_) hipLaunchKernelGGL((k_lif_step_c32t<R_, TH_, Q_>), dim3((unsigned)njob), dim3(256), 0, st, W, b, \
                                                     eps1, arp, out_s, out_pv, out_v, d->h, d->w, d->alpharp, d->wrp)
#define DCLL_STEP_T(R_, TH_) do { if (W.q) DCLL_STEP_TQ(R_, TH_, true); else DCLL_STEP_TQ(R_, TH_, false); } while (0)
        if (c1t) {
            if (d->refractory)
                hipLaunchKernelGGL((k_lif_step_c1<true, true>), dim3((unsigned)njob), dim3(256), 0, st, d->c_out, x, W, b, alpha,
                                   tau_m, alphas, tau_s, d->tau_is_tensor, eps0, eps1, arp, out_s, out_pv, out_v, d->alpharp,
                                   d->wrp, d->h, d->w);
            else
                hipLaunchKernelGGL((k_lif_step_c1<false, true>), dim3((unsigned)njob), dim3(256), 0, st, d->c_out, x, W, b, alpha,
                                   tau_m, alphas, tau_s, d->tau_is_tensor, eps0, eps1, arp, out_s, out_pv, out_v, d->alpharp,
                                   d->wrp, d->h, d->w);
        } else if (split16) {
            if (d->refractory) DCLL_STEP_T(true, 8); else DCLL_STEP_T(false, 8);
        } else {
            if (d->refractory) DCLL_STEP_T(true, 16); else DCLL_STEP_T(false, 16);
        }
#undef DCLL_STEP_T
#undef DCLL_STEP_TQ
        HIP_CHECK_LAUNCH(c1t ? "k_lif_step_c1 (tiled)" : split16 ? "k_lif_step_c32t (8-row tiles)" : "k_lif_step_c32t");
        if (i2o_W && out_p) {
            rc = launch_readout(out_pv, i2o_W, i2o_b, out_p, B, K, d->target, st);
            if (rc) return rc;
        }
        if (d->output_layer) {
            if (!out_W || !out_o) return fail(DCLL_ERR_INVALID, "dcll_conv_lif_step: output layer needs out_W and out_o");
            rc = launch_readout(out_pv, out_W, out_b, out_o, B, K, d->target, st);
            if (rc) return rc;
        }
        return DCLL_OK;
    }
    if ((d->c_in == 32 && d->c_out == 32 && plane16) || (d->c_in == 1 && d->c_out <= 32 && plane16)) {
        // the whole layer step in one MFMA kernel (traces, conv in the pinned order, refractory, threshold, sigmoid)
        if (d->c_in == 1) {
            if (d->refractory)
                hipLaunchKernelGGL((k_lif_step_c1<true, false>), dim3(B), dim3(256), 0, st, d->c_out, x, W, b, alpha, tau_m,
                                   alphas, tau_s, d->tau_is_tensor, eps0, eps1, arp, out_s, out_pv, out_v, d->alpharp, d->wrp,
                                   16, 16);
            else
                hipLaunchKernelGGL((k_lif_step_c1<false, false>), dim3(B), dim3(256), 0, st, d->c_out, x, W, b, alpha, tau_m,
                                   alphas, tau_s, d->tau_is_tensor, eps0, eps1, arp, out_s, out_pv, out_v, d->alpharp, d->wrp,
                                   16, 16);
        } else {
#define DCLL_STEP_16(R_, Q_) hipLaunchKernelGGL((k_lif_step_c32<R_, 0, Q_>), dim3(B), dim3(256), 0, st, x, W, b, alpha, tau_m, alphas, \
                                                tau_s, d->tau_is_tensor, eps0, eps1, arp, out_s, out_pv, out_v, d->alpharp, d->wrp)
            if (d->refractory) { if (W.q) DCLL_STEP_16(true, true); else DCLL_STEP_16(true, false); }
            else { if (W.q) DCLL_STEP_16(false, true); else DCLL_STEP_16(false, false); }
#undef DCLL_STEP_16
        }
        HIP_CHECK_LAUNCH(d->c_in == 1 ? "k_lif_step_c1" : "k_lif_step_c32");
        if (i2o_W && out_p) {
            rc = launch_readout(out_pv, i2o_W, i2o_b, out_p, B, K, d->target, st);
            if (rc) return rc;
        }
        if (d->output_layer) {
            if (!out_W || !out_o) return fail(DCLL_ERR_INVALID, "dcll_conv_lif_step: output layer needs out_W and out_o");
            rc = launch_readout(out_pv, out_W, out_b, out_o, B, K, d->target, st);
            if (rc) return rc;
        }
        return DCLL_OK;
    }
    hipLaunchKernelGGL(k_trace, dim3(nblk(nin, 256) > 4096 ? 4096 : nblk(nin, 256)), dim3(256), 0, st, x, alpha, tau_m,
                       alphas, tau_s, eps0, eps1, nin, per, d->tau_is_tensor);
    HIP_CHECK_LAUNCH("k_trace");
    const bool pooled = !(d->pool_h == 1 && d->pool_w == 1);
    // without pooling the conv kernel writes s / pv straight into the outputs; with pooling the un-pooled maps go to
    // the caller's scratch (2 * B*c_out*ch*cw floats) and k_pool produces the outputs.
    float *s_full = out_s, *pv_full = out_pv;
    if (pooled) {
        if (!scratch) return fail(DCLL_ERR_INVALID, "dcll_conv_lif_step: pooling layer needs scratch");
        s_full = scratch;
        pv_full = scratch + nconv;
    }
    {
        constexpr int COG = 8;
        const dim3 tg(((cw + 15) / 16) * ((ch + 15) / 16), (d->c_out + COG - 1) / COG, B);
        const bool tile_ok = ch >= 8 && cw >= 8 && B <= 65535 && plain_conv(d);
#define DCLL_TILED(KH_, KW_)                                                                                          \
    hipLaunchKernelGGL((k_conv_lif_tiled<KH_, KW_, COG>), tg, dim3(256), 0, st, *d, ch, cw, eps1, W, b, arp, s_full,    \
                       pv_full, out_v)
        if (tile_ok && d->kh == 7 && d->kw == 7) DCLL_TILED(7, 7);
        else if (tile_ok && d->kh == 5 && d->kw == 5) DCLL_TILED(5, 5);
        else if (tile_ok && d->kh == 3 && d->kw == 3) DCLL_TILED(3, 3);
        else if (tile_ok && d->kh == 1 && d->kw == 3) DCLL_TILED(1, 3);       // radio_ml_conv_ref.yaml
        else
            hipLaunchKernelGGL(k_conv_lif, dim3(nblk(nconv, 256)), dim3(256), 0, st, *d, ch, cw, eps1, W, b, arp, s_full,
                               pv_full, out_v, nconv);
#undef DCLL_TILED
    }
    HIP_CHECK_LAUNCH("k_conv_lif");
    if (pooled) {
        hipLaunchKernelGGL(k_pool, dim3(nblk(npool, 256)), dim3(256), 0, st, *d, ch, cw, ph, pw, s_full, pv_full, out_s,
                           out_pv, npool);
        HIP_CHECK_LAUNCH("k_pool");
    }
    if (i2o_W && out_p) {
        rc = launch_readout(out_pv, i2o_W, i2o_b, out_p, B, K, d->target, st);
        if (rc) return rc;
    }
    if (d->output_layer) {
        if (!out_W || !out_o) return fail(DCLL_ERR_INVALID, "dcll_conv_lif_step: output layer needs out_W and out_o");
        rc = launch_readout(out_pv, out_W, out_b, out_o, B, K, d->target, st);
        if (rc) return rc;
    }
    return DCLL_OK;
}

extern "C" int dcll_dense_lif_step(const dcll_dense_desc *d, const float *x, const float *W, const float *b,
                                   const float *alpha, const float *tau_m, const float *alphas, const float *tau_s,
                                   float *eps0, float *eps1, float *arp, const float *i2o_W, const float *i2o_b,
                                   float *out_s, float *out_p, float *out_pv, float *out_v, int32_t B, void *stream)
{
    if (!d || d->in_features < 1 || d->out_features < 1) return fail(DCLL_ERR_INVALID, "dcll_dense_lif_step: bad descriptor");
    if (B == 0) return DCLL_OK;
    if (!x || !W || !alpha || !tau_m || !alphas || !tau_s || !eps0 || !eps1 || !out_pv)
        return fail(DCLL_ERR_INVALID, "dcll_dense_lif_step: null pointer");
    if (d->refractory && !arp) return fail(DCLL_ERR_INVALID, "dcll_dense_lif_step: refractory layer needs arp");
    if (B < 0) return fail(DCLL_ERR_INVALID, "dcll_dense_lif_step: negative batch");
    if (B == 0) return DCLL_OK;
    hipStream_t st = (hipStream_t)stream;
    const long nin = (long)B * d->in_features;
    hipLaunchKernelGGL(k_trace, dim3(nblk(nin, 256) > 4096 ? 4096 : nblk(nin, 256)), dim3(256), 0, st, x, alpha, tau_m,
                       alphas, tau_s, eps0, eps1, nin, (long)d->in_features, d->tau_is_tensor);
    HIP_CHECK_LAUNCH("k_trace");
    int rc = dcll_launch_dense_mfma(d, eps1, W, b, arp, out_s, out_pv, out_v, B, st);       // pinned chain as an MFMA GEMM
    if (rc) return rc;
    if (i2o_W && out_p) return launch_readout(out_pv, i2o_W, i2o_b, out_p, B, d->out_features, d->target, st);
    return DCLL_OK;
}

// All T steps of a dense layer in one call: `for t: DenseDCLLlayer.forward(x[t])` (dcll/pytorch_libdcll.py:250-255).
// Small layers (dcll_dense_seq_fits: in_features <= 1024, out_features <= 128) run k_dense_lif_seq — one launch, neuron
// state on chip for the whole sequence; larger ones advance step by step (k_trace + k_dense_lif_mfma, state in HBM: the
// traces of ONE sample at in_features = 8192 are 64 KB).  Either way the local readout runs once over all T x B rows.
extern "C" int dcll_dense_lif_sequence(const dcll_dense_desc *d, const float *x, const float *W, const float *b,
                                       const float *alpha, const float *tau_m, const float *alphas, const float *tau_s,
                                       float *eps0, float *eps1, float *arp, const float *i2o_W, const float *i2o_b,
                                       float *out_s, float *out_p, float *out_pv, float *out_v, int32_t T, int32_t B,
                                       void *stream)
{
    const char *who = "dcll_dense_lif_sequence";
    if (!d || d->in_features < 1 || d->out_features < 1) return fail(DCLL_ERR_INVALID, "bad descriptor", who);
    if (T == 0 || B == 0) return DCLL_OK;
    if (!x || !W || !alpha || !tau_m || !alphas || !tau_s || !eps0 || !eps1) return fail(DCLL_ERR_INVALID, "null pointer", who);
    if (d->refractory && !arp) return fail(DCLL_ERR_INVALID, "refractory layer needs arp", who);
    if (T < 0 || B < 0) return fail(DCLL_ERR_INVALID, "negative size", who);
    if (i2o_W && out_p && !out_pv) return fail(DCLL_ERR_INVALID, "the local readout needs out_pv", who);
    hipStream_t st = (hipStream_t)stream;
    const long nin = (long)B * d->in_features, nout = (long)B * d->out_features;
    int rc;
    if (dcll_dense_seq_fits(d)) {
        rc = dcll_launch_dense_seq(d, x, W, b, alpha, tau_m, alphas, tau_s, eps0, eps1, arp, out_s, out_pv, out_v, T, B, st);
        if (rc) return rc;
    } else {
        for (int t = 0; t < T; ++t) {
            hipLaunchKernelGGL(k_trace, dim3(nblk(nin, 256) > 4096 ? 4096 : nblk(nin, 256)), dim3(256), 0, st, x + t * nin, alpha,
                               tau_m, alphas, tau_s, eps0, eps1, nin, (long)d->in_features, d->tau_is_tensor);
            HIP_CHECK_LAUNCH("k_trace");
            rc = dcll_launch_dense_mfma(d, eps1, W, b, arp, out_s ? out_s + t * nout : nullptr,
                                        out_pv ? out_pv + t * nout : nullptr, out_v ? out_v + t * nout : nullptr, B, st);
            if (rc) return rc;
        }
    }
    if (i2o_W && out_p) return launch_readout(out_pv, i2o_W, i2o_b, out_p, (long)T * B, d->out_features, d->target, st);
    return DCLL_OK;
}

// The same gradient with every pv row read ONCE: thread = one column k of pv, up to 32 readout rows accumulated in
// registers (g_o is wave-uniform -> scalar loads), the batch split into gridDim.y chunks whose partial sums
// part[chunk][n][K+1] are added in chunk order by k_bwd_outgrad_reduce (deterministic, no atomics).
__global__ __launch_bounds__(256) void k_bwd_outgrad_part(const float *__restrict__ g_o, const float *__restrict__ pvp,
                                                           float *__restrict__ part, int B, int N, int K)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int per = (B + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = min(B, b0 + per);
    float acc[32];
#pragma unroll
    for (int n = 0; n < 32; ++n) acc[n] = 0.0f;
    if (k < K) {
        for (int b = b0; b < b1; ++b) {
            const float p = pvp[(long)b * K + k];
#pragma unroll
            for (int n = 0; n < 32; ++n)
                if (n < N) acc[n] = __builtin_fmaf(g_o[(long)b * N + n], p, acc[n]);
        }
    } else if (k - K < N) {             // the bias gradient of readout row k - K, in acc[0]
        for (int b = b0; b < b1; ++b) acc[0] += g_o[(long)b * N + (k - K)];
    }
    float *pp = part + (long)blockIdx.y * N * (K + 1);
    if (k < K) {
#pragma unroll
        for (int n = 0; n < 32; ++n)
            if (n < N) pp[(long)n * (K + 1) + k] = acc[n];
    } else if (k - K < N) {
        pp[(long)(k - K) * (K + 1) + K] = acc[0];
    }
}

__global__ void k_bwd_outgrad_reduce(const float *__restrict__ part, float *__restrict__ dW, float *__restrict__ db,
                                     int nchunk, int N, int K)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * (K + 1)) return;
    float tot = 0.0f;
    for (int c = 0; c < nchunk; ++c) tot += part[(long)c * N * (K + 1) + i];
    const int n = (int)(i / (K + 1)), k = (int)(i % (K + 1));
    if (k < K) dW[(long)n * K + k] = tot;
    else db[n] = tot;
}

// The output_ gradient as fp32 MFMA for N <= 32 readout rows and K % 32 == 0: d_outW[n][k] = sum_b g_o[b][n] * pv[b][k] is
// a (N x B) . (B x K) GEMM; a workgroup takes 32 columns of K, its four waves a quarter of the batch each (the MFMA's two k
// lanes = two consecutive samples; the B operand is a coalesced 128-byte piece of a pv row), the four partial tiles are
// added in wave order through LDS.  pv is read once, nothing is written but the result (the partial-sum version above
// moves 12 MB of partials per call: 46 + 6 us at B = 512 against ~8 us).  Workgroup 0 also sums the bias gradient.
__global__ __launch_bounds__(256) void k_bwd_outgrad_mfma(const float *__restrict__ g_o, const float *__restrict__ pvp,
                                                           float *__restrict__ dW, float *__restrict__ db, int B, int N,
                                                           int K)
{
    __shared__ float red[4][16][64];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, j = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k0 = blockIdx.x * 32;
    const int per = (((B + 3) / 4) + 1) & ~1;                   // samples per wave, even
    const int b0 = w * per, b1 = min(B, b0 + per);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    int b = b0;
    // sixteen sample pairs per group, register double buffer: the loads of group g + 1 are in flight under the MFMAs of
    // group g (round 4: one group at a time meant eight exposed HBM round trips per wave at B = 512 — 21 us for a kernel
    // whose traffic is 17 MB)
    constexpr int GP = 16;
    float av[2][GP], bvv[2][GP];
    auto fetch = [&](int buf, int bs) {
#pragma unroll
        for (int q = 0; q < GP; ++q) {
            const int bb = bs + 2 * q + h;
            av[buf][q] = j < N ? g_o[(long)bb * N + j] : 0.0f;
            bvv[buf][q] = pvp[(long)bb * K + k0 + j];
        }
    };
    auto mfmas = [&](int buf) {
#pragma unroll
        for (int q = 0; q < GP; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][q], bvv[buf][q], acc, 0, 0, 0);
    };
    if (b + 2 * GP <= b1) fetch(0, b);
    for (; b + 4 * GP <= b1; b += 4 * GP) {                    // two groups per trip: static buffer indices
        fetch(1, b + 2 * GP);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0);
        __builtin_amdgcn_sched_barrier(0);
        if (b + 6 * GP <= b1) fetch(0, b + 4 * GP);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (b + 2 * GP <= b1) {                                    // a last single group (its loads were issued above)
        mfmas(0);
        b += 2 * GP;
    }
    for (; b < b1; b += 2) {
        const int bb = b + h;
        const bool ok = bb < b1;
        const float a = (ok && j < N) ? g_o[(long)bb * N + j] : 0.0f;
        const float bv = ok ? pvp[(long)bb * K + k0 + j] : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[w][r][lane] = acc[r];
    __syncthreads();
    for (int e = tid; e < 16 * 64; e += 256) {
        const int l = e & 63, r = e >> 6;
        const int n = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        if (n < N) dW[(long)n * K + k0 + (l & 31)] = ((red[0][r][l] + red[1][r][l]) + red[2][r][l]) + red[3][r][l];
    }
    if (blockIdx.x == 0) {                                      // d_outb[n] = sum_b g_o[b][n]: 8 strided parts per row
        __syncthreads();
        float *rb = &red[0][0][0];
        const int n = tid & 31, part = tid >> 5;
        float sum = 0.0f;
        if (n < N)
            for (int b = part; b < B; b += 8) sum += g_o[(long)b * N + n];
        rb[part * 32 + n] = sum;
        __syncthreads();
        if (tid < N) {
            float tot = 0.0f;
#pragma unroll
            for (int q = 0; q < 8; ++q) tot += rb[q * 32 + tid];
            db[tid] = tot;
        }
    }
}

// open_part != nullptr: the weight gradient's partial rows are left in scratch (*open_part, *open_nchunk) for
// dcll_grad_reduce_adam; dW / db are not written
static int conv_lif_backward_impl(const dcll_conv_desc *d, const float *eps1, const float *v, const float *pv_pooled,
                                  const float *g_p, const float *g_o, const float *g_pv, const float *g_v,
                                  const float *i2o_W, float *dW, float *db, float *d_outW, float *d_outb,
                                  float *scratch, int64_t scratch_floats, int32_t B, void *stream,
                                  const float **open_part, int32_t *open_nchunk, bool dv_done = false)
{
    int rc = check_desc(d);
    if (rc) return rc;
    const bool nopool = d->pool_h == 1 && d->pool_w == 1 && d->target <= 32;
    if (!eps1 || (!dW && !open_part) || !scratch) return fail(DCLL_ERR_INVALID, "dcll_conv_lif_backward: null pointer");
    if (!v && !(nopool && pv_pooled))
        return fail(DCLL_ERR_INVALID, "dcll_conv_lif_backward: v may be NULL only for a layer without pooling whose pv is given");
    if (g_p && !i2o_W) return fail(DCLL_ERR_INVALID, "dcll_conv_lif_backward: g_p needs i2o_W");
    if (g_o && (!pv_pooled || !d_outW || !d_outb)) return fail(DCLL_ERR_INVALID, "dcll_conv_lif_backward: g_o needs pv_pooled, d_outW, d_outb");
    if (d->kh * d->kw > WG_MAXTAPS) return fail(DCLL_ERR_UNSUPPORTED, "dcll_conv_lif_backward: kernels up to 64 taps");
    if (B < 1) return fail(DCLL_ERR_INVALID, "dcll_conv_lif_backward: empty batch");
    hipStream_t st = (hipStream_t)stream;
    int ch, cw, ph, pw;
    conv_shape(d, &ch, &cw, &ph, &pw);
    const long nconv = (long)B * d->c_out * ch * cw;
    if (dv_done) {
        // (dcll_conv_lif_backward_open_multi ran this layer's dv with the other layers': the gradient map is in scratch)
    } else if (nopool) {
        const int Kmap = d->c_out * ch * cw, per_block = 16;
        const dim3 grid(nblk(Kmap, 256), nblk(B, per_block));
#define DCLL_DV(NP_)                                                                                                    \
    do {                                                                                                                \
        if (v)                                                                                                          \
            hipLaunchKernelGGL((k_bwd_dv_nopool<NP_, false>), grid, dim3(256), 0, st, Kmap, d->target, v, g_p, g_pv, g_v,   \
                               i2o_W, scratch, B, per_block);                                                           \
        else        /* sigmoid' from the stored pv (bit-identical: the forward's own sigmoid of the same v) */           \
            hipLaunchKernelGGL((k_bwd_dv_nopool<NP_, true>), grid, dim3(256), 0, st, Kmap, d->target, pv_pooled, g_p, g_pv, \
                               g_v, i2o_W, scratch, B, per_block);                                                      \
    } while (0)
        if (d->target <= 8) DCLL_DV(8);
        else if (d->target <= 16) DCLL_DV(16);
        else if (d->target <= 24) DCLL_DV(24);
        else DCLL_DV(32);
#undef DCLL_DV
    } else {
        hipLaunchKernelGGL(k_bwd_dv, dim3(nblk(nconv, 256)), dim3(256), 0, st, *d, ch, cw, ph, pw, v, g_p, g_pv, g_v, i2o_W,
                           scratch, nconv);
    }
    if (!dv_done) HIP_CHECK_LAUNCH("k_bwd_dv");
    // weight gradient: partial sums over batch chunks (after the g_v_full plane in scratch), then a fixed-order reduce
    const long rowlen = (long)(d->c_in / d->groups) * d->kh * d->kw + 1;      // (a weight row: the c_in / groups channels of co's group)
    const long per_chunk = (long)d->c_out * rowlen;
    float *part = scratch + nconv;
    long nchunk = (scratch_floats - nconv) / per_chunk;
    if (nchunk < 1) return fail(DCLL_ERR_INVALID, "dcll_conv_lif_backward: scratch too small (need B*c_out*ch*cw + k*(c_out*(c_in*kh*kw+1)), k >= 1)");
    const bool c32 = d->c_in == 32 && d->c_out == 32 && d->kh == 7 && d->kw == 7 && d->pad_h == 3 && d->pad_w == 3 && plain_conv(d);
    if (c32 && d->h == 16 && d->w == 16) {
        if (nchunk > 256) nchunk = 256;
        if (nchunk > B) nchunk = B;
        // round 6: up to WG32_SPLIT2_MAX_BATCH samples 128 batch chunks x 2 column halves instead of 256 x 1 — half the partial
        // rows (25.7 instead of 51.4 MB), which every workgroup stores at the END of the launch, all at once, and the reduction
        // reads back: at B = 512 k_bwd_wgrad_c32 103.4 -> 100.6 us, k_grad_reduce_adam 25.0 -> 18.1 us; B = 384 (1.5 jobs per
        // chunk before: two against one) 116 -> 93 us for the closed call; neutral at 2048, a loss at 4096 (the per-job staging
        // doubles against the MFMAs) — experiments/wgrad_split_sweep.py, HISTORY.md "Round 6"
        if (B > 160 && B <= WG32_SPLIT2_MAX_BATCH && nchunk > 128) nchunk = 128;
        // fewer samples than CUs: a sample's column tiles over 2, 3 or 6 workgroups
        if (nchunk <= 48)
            hipLaunchKernelGGL((k_bwd_wgrad_c32<ROWF, CHF, false, 6>), dim3((unsigned)nchunk, 6), dim3(512), 0, st, scratch,
                               eps1, part, B, 16, 16);
        else if (nchunk <= 96)
            hipLaunchKernelGGL((k_bwd_wgrad_c32<ROWF, CHF, false, 3>), dim3((unsigned)nchunk, 3), dim3(512), 0, st, scratch,
                               eps1, part, B, 16, 16);
        else if (nchunk <= 160)
            hipLaunchKernelGGL((k_bwd_wgrad_c32<ROWF, CHF, false, 2>), dim3((unsigned)nchunk, 2), dim3(512), 0, st, scratch,
                               eps1, part, B, 16, 16);
        else
            hipLaunchKernelGGL((k_bwd_wgrad_c32<ROWF, CHF, false>), dim3((unsigned)nchunk), dim3(512), 0, st, scratch, eps1,
                               part, B, 16, 16);
        HIP_CHECK_LAUNCH("k_bwd_wgrad_c32");
    } else if (plain_conv(d) && d->c_in == 1 && d->c_out == 32 && d->kh == 7 && d->kw == 7 && d->pad_h == 3 && d->pad_w == 3 && d->h == 16 &&
               d->w == 16) {                                   // first layer of radio_ml_conv.yaml: MFMA, two column tiles
        if (nchunk > 512) nchunk = 512;
        if (nchunk > B) nchunk = B;
        hipLaunchKernelGGL((k_bwd_wgrad_c1<ROWF, false>), dim3((unsigned)nchunk), dim3(128), 0, st, scratch, eps1, part, B,
                           16, 16);
        HIP_CHECK_LAUNCH("k_bwd_wgrad_c1");
    } else if (plain_conv(d) && d->c_in == 1 && d->c_out == 32 && d->kh == 7 && d->kw == 7 && d->pad_h == 3 && d->pad_w == 3 &&
               d->h % 16 == 0 && d->w % 16 == 0) {             // first layer on large planes: one 16x16 tile per job
        const long njob = (long)B * (d->h / 16) * (d->w / 16);
        if (nchunk > 1024) nchunk = 1024;
        if (nchunk > njob) nchunk = njob;
        hipLaunchKernelGGL((k_bwd_wgrad_c1<22, true>), dim3((unsigned)nchunk), dim3(128), 0, st, scratch, eps1, part, B,
                           d->h, d->w);
        HIP_CHECK_LAUNCH("k_bwd_wgrad_c1 (tiled)");
    } else if (c32 && d->h % 16 == 0 && d->w % 16 == 0) {     // large planes: one 16x16 tile of a sample per job
        const long njob = (long)B * (d->h / 16) * (d->w / 16);
        if (nchunk > 256) nchunk = 256;
        if (nchunk > njob) nchunk = njob;
        hipLaunchKernelGGL((k_bwd_wgrad_c32<22, 484, true>), dim3((unsigned)nchunk), dim3(512), 0, st, scratch, eps1,
                           part, B, d->h, d->w);
        HIP_CHECK_LAUNCH("k_bwd_wgrad_c32 (tiled)");
    } else {
        // the eps1 plane is staged in LDS in bands of RB output rows (+ kh - 1 halo rows), at most 48 KB
        // (RB output rows read (RB - 1) stride + (kh - 1) dilation + 1 input rows)
        const int WP = d->w + 2 * d->pad_w;
        const int rows_fit = (48 * 1024 / 4 - 4 * (WG_MAXTAPS + 1)) / WP;
        int RB = rows_fit < (d->kh - 1) * d->dilation + 1 ? 0 : (rows_fit - (d->kh - 1) * d->dilation - 1) / d->stride + 1;
        if (RB < 1) return fail(DCLL_ERR_UNSUPPORTED, "dcll_conv_lif_backward: input rows too wide for the LDS-staged weight-gradient kernel");
        if (RB > ch) RB = ch;
        const size_t lds = ((size_t)((RB - 1) * d->stride + (d->kh - 1) * d->dilation + 1) * WP + 4 * (WG_MAXTAPS + 1)) * sizeof(float);
        if (nchunk > 64) nchunk = 64;
        if (nchunk > B) nchunk = B;
        hipLaunchKernelGGL(k_bwd_wgrad, dim3(d->c_out * (d->c_in / d->groups), (unsigned)nchunk), dim3(256), lds, st, *d, ch, cw,
                           scratch, eps1, part, B, RB);
        HIP_CHECK_LAUNCH("k_bwd_wgrad");
    }
    if (open_part) {
        *open_part = part;
        *open_nchunk = (int32_t)nchunk;
    } else {
        if (nchunk >= 64)
            hipLaunchKernelGGL(k_bwd_reduce4<16>, dim3(nblk(per_chunk, 64)), dim3(1024), 0, st, part, dW, db, (int)nchunk, d->c_out, rowlen);
        else if (nchunk >= 16)
            hipLaunchKernelGGL(k_bwd_reduce4<4>, dim3(nblk(per_chunk, 64)), dim3(256), 0, st, part, dW, db, (int)nchunk, d->c_out, rowlen);
        else
            hipLaunchKernelGGL(k_bwd_reduce, dim3(nblk(per_chunk, 256)), dim3(256), 0, st, part, dW, db, (int)nchunk, d->c_out, rowlen);
        HIP_CHECK_LAUNCH("k_bwd_reduce");
    }
    if (g_o) {
        const int K = d->c_out * ph * pw, N = d->target;
        // closed form: the partial-sum area of the weight gradient is free again (stream order) and holds the batch chunks.
        // Open form: the partial rows are still wanted, the batch chunks go BEHIND the nchunk rows in use.  With scratch for
        // min(B, 16) chunks in either place (what ops.conv_lif_backward provides) both forms split the batch alike, so
        // d_outW / d_outb are the same bits from dcll_conv_lif_backward and dcll_conv_lif_backward_open.
        float *opart = open_part ? part + nchunk * per_chunk : part;
        long nsplit = (scratch_floats - (opart - scratch)) / ((long)N * (K + 1));
        if (nsplit > 16) nsplit = 16;
        if (nsplit > B) nsplit = B;
        if (N <= 32 && K % 32 == 0) {
            hipLaunchKernelGGL(k_bwd_outgrad_mfma, dim3(K / 32), dim3(256), 0, st, g_o, pv_pooled, d_outW, d_outb, B, N, K);
            HIP_CHECK_LAUNCH("k_bwd_outgrad_mfma");
        } else if (N <= 32 && nsplit >= 1) {
            hipLaunchKernelGGL(k_bwd_outgrad_part, dim3(nblk((long)K + N, 256), (unsigned)nsplit), dim3(256), 0, st, g_o,
                               pv_pooled, opart, B, N, K);
            HIP_CHECK_LAUNCH("k_bwd_outgrad_part");
            hipLaunchKernelGGL(k_bwd_outgrad_reduce, dim3(nblk((long)N * (K + 1), 256)), dim3(256), 0, st, opart, d_outW,
                               d_outb, (int)nsplit, N, K);
            HIP_CHECK_LAUNCH("k_bwd_outgrad_reduce");
        } else {
            hipLaunchKernelGGL(k_bwd_outgrad, dim3(nblk((long)N * (K + 1), 256)), dim3(256), 0, st, g_o, pv_pooled,
                               d_outW, d_outb, B, N, K);
            HIP_CHECK_LAUNCH("k_bwd_outgrad");
        }
    }
    return DCLL_OK;
}

extern "C" int dcll_conv_lif_backward(const dcll_conv_desc *d, const float *eps1, const float *v, const float *pv_pooled,
                                      const float *g_p, const float *g_o, const float *g_pv, const float *g_v,
                                      const float *i2o_W, float *dW, float *db, float *d_outW, float *d_outb,
                                      float *scratch, int64_t scratch_floats, int32_t B, void *stream)
{
    return conv_lif_backward_impl(d, eps1, v, pv_pooled, g_p, g_o, g_pv, g_v, i2o_W, dW, db, d_outW, d_outb, scratch,
                                  scratch_floats, B, stream, nullptr, nullptr);
}

extern "C" int dcll_conv_lif_backward_open(const dcll_conv_desc *d, const float *eps1, const float *v, const float *pv_pooled,
                                           const float *g_p, const float *g_o, const float *g_pv, const float *g_v,
                                           const float *i2o_W, float *d_outW, float *d_outb, float *scratch,
                                           int64_t scratch_floats, int32_t B, const float **part, int32_t *nchunk, void *stream)
{
    if (!part || !nchunk) return fail(DCLL_ERR_INVALID, "dcll_conv_lif_backward_open: null part / nchunk");
    return conv_lif_backward_impl(d, eps1, v, pv_pooled, g_p, g_o, g_pv, g_v, i2o_W, nullptr, nullptr, d_outW, d_outb, scratch,
                                  scratch_floats, B, stream, part, nchunk);
}

// dcll_conv_lif_backward_open for n layers — the slices of one learning timestep — with their dv launches as ONE launch
// (k_bwd_dv_nopool_m) where all of them are layers without pooling of one readout-width class; the weight / output_ gradient
// kernels follow layer by layer as in the single call.  Results per item = dcll_conv_lif_backward_open on it.
extern "C" int dcll_conv_lif_backward_open_multi(dcll_bwd_item *items, int32_t n, void *stream)
{
    const char *who = "dcll_conv_lif_backward_open_multi";
    if (n == 0) return DCLL_OK;
    if (!items || n < 0 || n > BWD_MULTI_MAX) return fail(DCLL_ERR_INVALID, "1 .. 8 items", who);
    bool joint = n > 1;
    int npc = -1, frompv = -1;
    bwd_dv_items it;
    memset(&it, 0, sizeof(it));
    unsigned gx = 1, gy = 1;
    for (int i = 0; i < n; ++i) {
        const dcll_bwd_item &a = items[i];
        if (a.reserved != 0) return fail(DCLL_ERR_INVALID, "dcll_bwd_item.reserved must be 0", who);
        if (!a.d || !a.eps1 || !a.scratch || a.B < 1) return fail(DCLL_ERR_INVALID, "null pointer / empty batch", who);
        int rc = check_desc(a.d);
        if (rc) return rc;
        const bool nopool = a.d->pool_h == 1 && a.d->pool_w == 1 && a.d->target <= 32;
        if (!a.v && !(nopool && a.pv_pooled))
            return fail(DCLL_ERR_INVALID, "v may be NULL only for a layer without pooling whose pv is given", who);
        if (a.g_p && !a.i2o_W) return fail(DCLL_ERR_INVALID, "g_p needs i2o_W", who);
        int ch, cw, ph, pw;
        conv_shape(a.d, &ch, &cw, &ph, &pw);
        const long Kmap = (long)a.d->c_out * ch * cw;
        if ((long)a.B * Kmap > a.scratch_floats) return fail(DCLL_ERR_INVALID, "scratch too small", who);
        const int cls = a.d->target <= 8 ? 8 : a.d->target <= 16 ? 16 : a.d->target <= 24 ? 24 : 32, fp = a.v ? 0 : 1;
        if (!nopool || Kmap >= (1L << 31) || (npc >= 0 && (cls != npc || fp != frompv))) joint = false;
        npc = cls, frompv = fp;
        if (joint) {
            it.v[i] = a.v ? a.v : a.pv_pooled, it.g_p[i] = a.g_p, it.g_pv[i] = a.g_pv, it.g_v[i] = a.g_v, it.i2o_W[i] = a.i2o_W;
            it.gvf[i] = a.scratch, it.K[i] = (int)Kmap, it.N[i] = a.d->target, it.B[i] = a.B;
            gx = max(gx, (unsigned)nblk(Kmap, 256)), gy = max(gy, (unsigned)nblk(a.B, 16));
        }
    }
    hipStream_t st = (hipStream_t)stream;
    if (joint) {
        const dim3 grid(gx, gy, (unsigned)n);
#define DCLL_DVM(NP_)                                                                                                   \
    do {                                                                                                                \
        if (frompv) hipLaunchKernelGGL((k_bwd_dv_nopool_m<NP_, true>), grid, dim3(256), 0, st, it, 16);                  \
        else hipLaunchKernelGGL((k_bwd_dv_nopool_m<NP_, false>), grid, dim3(256), 0, st, it, 16);                        \
    } while (0)
        if (npc == 8) DCLL_DVM(8);
        else if (npc == 16) DCLL_DVM(16);
        else if (npc == 24) DCLL_DVM(24);
        else DCLL_DVM(32);
#undef DCLL_DVM
        HIP_CHECK_LAUNCH("k_bwd_dv_nopool_m");
    }
    for (int i = 0; i < n; ++i) {
        dcll_bwd_item &a = items[i];
        int rc = conv_lif_backward_impl(a.d, a.eps1, a.v, a.pv_pooled, a.g_p, a.g_o, a.g_pv, a.g_v, a.i2o_W, nullptr, nullptr,
                                        a.d_outW, a.d_outb, a.scratch, a.scratch_floats, a.B, stream, &a.part, &a.nchunk, joint);
        if (rc) return rc;
    }
    return DCLL_OK;
}

// ------------------------------------------------------------------------------------------------------------
// pv activity statistics (DCLLBase.forward, dcll/pytorch_libdcll.py:658-661 + write_stats :678-688): on every step whose
// 1-based iteration count is a multiple of 20 the reference histograms pv on the host (np.histogram, 19 bins over
// [0,1]) and reports only the first and the last bin.  Here: two counters per sampled step, pv < e1 and pv >= e18 with
// the edges of np.linspace(0, 1, 20) — evaluated exactly like numpy's float64 comparison by comparing against the
// smallest float32 >= edge — counted in one pass over the sampled steps' pv planes (wave ballots + popcount, one
// atomic pair per workgroup).  counts[k][0..1], k = index of the sampled step inside the call.
// ------------------------------------------------------------------------------------------------------------
// SIG: the buffer holds v (dcll_layer_opts pv_presigmoid): the counted value is sigmoid(v), the same device function the
// layer kernels would have applied — identical counters by construction (this pass reads 6 of 128 steps: the
// transcendentals are free here).
template <bool SIG>
__global__ __launch_bounds__(256) void k_pv_lowhigh(const float *__restrict__ pv, long per_step, int iter0,
                                                     float thr_low, float thr_high,
                                                     unsigned long long *__restrict__ counts)
{
    __shared__ unsigned red[2][4];
    const int k = blockIdx.y;
    const long t = ((long)(iter0 / 20) + k + 1) * 20 - iter0 - 1;          // 0-based step of the k-th sampled iteration
    const float *p = pv + t * per_step;
    unsigned lo = 0, hi = 0;                                                // wave-uniform
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if ((((uintptr_t)p) & 15) == 0) {
        const long n4 = per_step >> 2;
        for (; i < n4 + stride - 1 - (n4 + stride - 1) % stride; i += stride) {     // uniform trip count (ballots)
            f32x4 v = {.5f, .5f, .5f, .5f};
            const bool in = i < n4;
            if (in) v = ((const f32x4 *)p)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float q = SIG ? sigmoidf_dev(v[e]) : v[e];
                lo += __popcll(__ballot(in && q < thr_low));
                hi += __popcll(__ballot(in && q >= thr_high));
            }
        }
        i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x;                  // scalar tail (per_step % 4 floats)
    }
    for (; i < per_step + stride - 1 - (per_step + stride - 1) % stride; i += stride) {
        const bool in = i < per_step;
        const float v = in ? (SIG ? sigmoidf_dev(p[i]) : p[i]) : 0.5f;
        lo += __popcll(__ballot(in && v < thr_low));
        hi += __popcll(__ballot(in && v >= thr_high));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = lo; red[1][wave] = hi; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const unsigned tot = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        if (tot) atomicAdd(counts + 2 * k + threadIdx.x, (unsigned long long)tot);
    }
}

// smallest float32 >= e: for a float32 x, (x < e in float64) <=> x < f32_ceil(e), and likewise for >=
static float f32_ceil(double e)
{
    float f = (float)e;
    if ((double)f < e) f = nextafterf(f, INFINITY);
    return f;
}

static inline int n_sampled_steps(int iter0, int T) { return (iter0 + T) / 20 - iter0 / 20; }

static int launch_pv_lowhigh(const float *pv, long per_step, int T, int iter0, unsigned long long *counts,
                             hipStream_t st, const char *who, bool presig = false)
{
    if (iter0 < 0) return fail(DCLL_ERR_INVALID, "negative iteration count", who);
    const int ns = n_sampled_steps(iter0, T);
    if (ns == 0) return DCLL_OK;
    if (!pv) return fail(DCLL_ERR_INVALID, "pv statistics need the pv output of the call", who);
    if (hipMemsetAsync(counts, 0, (size_t)ns * 2 * sizeof(unsigned long long), st) != hipSuccess) {
        (void)hipGetLastError();
        return fail(DCLL_ERR_LAUNCH, "hipMemsetAsync of the pv counters failed", who);
    }
    const double e1 = 1.0 / 19.0, e18 = 18.0 * (1.0 / 19.0);               // np.linspace(0, 1, 20)[1], [18]
    long nb = (per_step / 4 + 255) / 256;
    if (nb < 1) nb = 1;
    if (nb > 1024) nb = 1024;
    if (presig)
        hipLaunchKernelGGL(k_pv_lowhigh<true>, dim3((unsigned)nb, (unsigned)ns), dim3(256), 0, st, pv, per_step, iter0,
                           f32_ceil(e1), f32_ceil(e18), counts);
    else
        hipLaunchKernelGGL(k_pv_lowhigh<false>, dim3((unsigned)nb, (unsigned)ns), dim3(256), 0, st, pv, per_step, iter0,
                           f32_ceil(e1), f32_ceil(e18), counts);
    HIP_CHECK_LAUNCH("k_pv_lowhigh");
    return DCLL_OK;
}

extern "C" int dcll_pv_lowhigh_act(const float *pv, int64_t per_step, int32_t T, int32_t iter0, uint64_t *counts,
                                   int32_t act, void *stream)
{
    if (T < 0 || per_step < 0 || iter0 < 0 || (act != DCLL_ACT_NONE && act != DCLL_ACT_SIGMOID))
        return fail(DCLL_ERR_INVALID, "dcll_pv_lowhigh: bad argument");
    if (T == 0 || per_step == 0 || n_sampled_steps(iter0, T) == 0) return DCLL_OK;     // no histogram step: counts may be NULL
    if (!counts) return fail(DCLL_ERR_INVALID, "dcll_pv_lowhigh: null counters");
    return launch_pv_lowhigh(pv, per_step, T, iter0, (unsigned long long *)counts, (hipStream_t)stream, "dcll_pv_lowhigh",
                             act == DCLL_ACT_SIGMOID);
}

extern "C" int dcll_pv_lowhigh(const float *pv, int64_t per_step, int32_t T, int32_t iter0, uint64_t *counts,
                               void *stream)
{
    return dcll_pv_lowhigh_act(pv, per_step, T, iter0, counts, DCLL_ACT_NONE, stream);
}

extern "C" int32_t dcll_pv_lowhigh_steps(int32_t iter0, int32_t T) { return (iter0 < 0 || T < 0) ? 0 : n_sampled_steps(iter0, T); }

static int check_seq_geometry(const dcll_conv_desc *d, int c_in, const char *who)
{
    int rc = check_desc(d);
    if (rc) return rc;
    const bool small = d->h == 16 && d->w == 16;
    const bool tiled = d->h >= 8 && d->h % 8 == 0 && d->w >= 32 && d->w % 32 == 0;   // k_lif_seq_c1t / k_lif_seq_c32t
    if (!plain_conv(d) || d->c_in != c_in || d->c_out > 32 || (c_in == 32 && d->c_out != 32) || !(small || tiled) || d->kh != 7 ||
        d->kw != 7 || d->pad_h != 3 || d->pad_w != 3 || d->pool_h != 1 || d->pool_w != 1)
        return fail(DCLL_ERR_UNSUPPORTED,
                    "sequence kernel supports 7x7 pad 3, pool 1, c_out<=32 (==32 for c_in 32) on a 16x16 plane or a plane "
                    "with h % 8 == 0, w % 32 == 0", who);
    return DCLL_OK;
}

extern "C" int dcll_permute_readout(const float *Wt, float *Wp, int32_t N, void *stream)
{
    if (!Wt || !Wp || N < 1) return fail(DCLL_ERR_INVALID, "dcll_permute_readout: bad argument");
    hipLaunchKernelGGL(k_permute_readout, dim3(nblk((long)N * 8192, 256)), dim3(256), 0, (hipStream_t)stream, Wt, Wp, N);
    HIP_CHECK_LAUNCH("k_permute_readout");
    return DCLL_OK;
}

constexpr int DCLL_C32D_MIN_T = 8;      // shorter sequences: k_lif_seq_c32 (half the pipeline fill)

// In presigmoid mode a 7x7 kernel writes v through its v output: (pv_out, v_out) as the kernel gets them.  Both wanted
// (tests): the kernel writes v_out, a stream-ordered copy fills pv_out afterwards.
struct seq_outs { float *pv, *v, *copy_dst; };
static inline seq_outs presig_outs(float *pv_out, float *v_out, bool presig)
{
    if (!presig || !pv_out) return {pv_out, v_out, nullptr};
    if (!v_out) return {nullptr, pv_out, nullptr};
    return {nullptr, v_out, pv_out};
}
static int presig_copy(const seq_outs &o, long n, hipStream_t st, const char *who)
{
    if (!o.copy_dst) return DCLL_OK;
    if (hipMemcpyAsync(o.copy_dst, o.v, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
        (void)hipGetLastError();
        return fail(DCLL_ERR_LAUNCH, "copy of v into pv_out failed", who);
    }
    return DCLL_OK;
}

template <bool R, int NRO>
static void launch_c32(int out, int B, hipStream_t st, const uint32_t *spk_in, dcll_wsrc W, const float *b,
                       const float *tau4, float *eps0, float *eps1, float *arp, uint32_t *spk_out, float *pv_out,
                       float *v_out, const float *ro_Wp, const float *ro_b, float *ro_out, int T, float alpharp,
                       float wrp)
{
#define DCLL_LAUNCH_C32(O)                                                                                             \
    hipLaunchKernelGGL((k_lif_seq_c32<R, O, NRO>), dim3(B), dim3(512), 0, st, spk_in, W, b, tau4, eps0, eps1, arp,     \
                       spk_out, pv_out, v_out, ro_Wp, ro_b, ro_out, T, B, alpharp, wrp)
    switch (out) {
    case 0: DCLL_LAUNCH_C32(0); break;
    case 1: DCLL_LAUNCH_C32(1); break;
    case 2: DCLL_LAUNCH_C32(2); break;
    default: DCLL_LAUNCH_C32(3); break;
    }
#undef DCLL_LAUNCH_C32
}

static int dcll_conv_lif_sequence_run(const dcll_conv_desc *d, const uint32_t *spk_in, dcll_wsrc W, const float *b,
                                      const float *tau4, float *eps0, float *eps1, float *arp, uint32_t *spk_out,
                                      float *pv_out, float *v_out, const float *ro_Wp, const float *ro_b,
                                      float *ro_out, int32_t n_ro, float *state_scratch, int32_t T, int32_t B,
                                      hipStream_t st)
{
    if (n_ro != 0 && n_ro != 24 && n_ro != 48)
        return fail(DCLL_ERR_UNSUPPORTED, "dcll_conv_lif_sequence: fused readout supports 24 or 48 rows (target 24)");
    if (n_ro && (!ro_Wp || !ro_b || !ro_out)) return fail(DCLL_ERR_INVALID, "dcll_conv_lif_sequence: fused readout needs ro_Wp, ro_b, ro_out");
    const int out = (pv_out ? 1 : 0) | (v_out ? 2 : 0);
    if (d->h != 16 || d->w != 16) {         // large plane: k_lif_seq_c32t, one workgroup per (sample, 8 x 32 tile)
        if (n_ro) return fail(DCLL_ERR_UNSUPPORTED, "dcll_conv_lif_sequence: fused readout only on the 16x16 plane");
        return dcll_launch_seq_c32t(d, spk_in, W, b, tau4, eps0, eps1, arp, spk_out, pv_out, v_out, state_scratch, T, B, st);
    }
    if (n_ro == 0 && T >= DCLL_C32D_MIN_T) {      // long sequence: two tiles per wave and stage
#define DCLL_LAUNCH_C32D(R, O)                                                                                          \
    hipLaunchKernelGGL((k_lif_seq_c32d<R, O>), dim3(B), dim3(512), 0, st, spk_in, W, b, tau4, eps0, eps1, arp, spk_out, \
                       pv_out, v_out, T, B, d->alpharp, d->wrp)
        if (d->refractory) {
            switch (out) {
            case 0: DCLL_LAUNCH_C32D(true, 0); break;
            case 1: DCLL_LAUNCH_C32D(true, 1); break;
            case 2: DCLL_LAUNCH_C32D(true, 2); break;
            default: DCLL_LAUNCH_C32D(true, 3); break;
            }
        } else {
            switch (out) {
            case 0: DCLL_LAUNCH_C32D(false, 0); break;
            case 1: DCLL_LAUNCH_C32D(false, 1); break;
            case 2: DCLL_LAUNCH_C32D(false, 2); break;
            default: DCLL_LAUNCH_C32D(false, 3); break;
            }
        }
#undef DCLL_LAUNCH_C32D
        HIP_CHECK_LAUNCH("k_lif_seq_c32d");
        return DCLL_OK;
    }
#define DCLL_ARGS out, B, st, spk_in, W, b, tau4, eps0, eps1, arp, spk_out, pv_out, v_out, ro_Wp, ro_b, ro_out, T, d->alpharp, d->wrp
    if (d->refractory) {
        if (n_ro == 0) launch_c32<true, 0>(DCLL_ARGS);
        else if (n_ro == 24) launch_c32<true, 24>(DCLL_ARGS);
        else launch_c32<true, 48>(DCLL_ARGS);
    } else {
        if (n_ro == 0) launch_c32<false, 0>(DCLL_ARGS);
        else if (n_ro == 24) launch_c32<false, 24>(DCLL_ARGS);
        else launch_c32<false, 48>(DCLL_ARGS);
    }
#undef DCLL_ARGS
    HIP_CHECK_LAUNCH("k_lif_seq_c32");
    return DCLL_OK;
}

extern "C" int dcll_conv_lif_sequence(const dcll_conv_desc *d, const uint32_t *spk_in, const float *Wf, const float *b,
                                      const float *tau4, float *eps0, float *eps1, float *arp, uint32_t *spk_out,
                                      float *pv_out, float *v_out, const float *ro_Wp, const float *ro_b,
                                      float *ro_out, int32_t n_ro, float *state_scratch, uint64_t *pv_lowhigh,
                                      int32_t iter0, const dcll_layer_opts *opts, int32_t T, int32_t B, void *stream)
{
    const char *who = "dcll_conv_lif_sequence";
    int rc = check_desc(d);
    if (rc) return rc;
    const bool w3 = dcll_seq_w3_geometry(d) && d->c_in == 64;
    if (!w3) {
        rc = check_seq_geometry(d, 32, who);
        if (rc) return rc;
    }
    if (T == 0 || B == 0) return DCLL_OK;      // empty input: nothing to do (its pointers may be NULL)
    rc = check_opts(Wf, opts, true, who);
    if (rc) return rc;
    if (!spk_in || !b || !tau4 || !eps0 || !eps1) return fail(DCLL_ERR_INVALID, "null pointer", who);
    if (d->refractory && !arp) return fail(DCLL_ERR_INVALID, "refractory layer needs arp", who);
    if (T < 0 || B < 0) return fail(DCLL_ERR_INVALID, "negative size", who);
    const bool presig = opts && opts->pv_presigmoid;
    if (presig && n_ro) return fail(DCLL_ERR_INVALID, "pv_presigmoid cannot be combined with the fused readout", who);
    const dcll_wsrc W = make_wsrc(Wf, opts);
    hipStream_t st = (hipStream_t)stream;
    long per_step;
    if (w3) {       // radio_ml_conv_ref.yaml geometry: pooled outputs (dcll_seq_w3.hip)
        if (n_ro) return fail(DCLL_ERR_UNSUPPORTED, "fused readout only for the 7x7 layers on the 16x16 plane", who);
        rc = dcll_launch_seq_w3(d, spk_in, nullptr, W, b, tau4, eps0, eps1, arp, spk_out, pv_out, v_out, presig, T, B, st);
        per_step = (long)B * d->c_out * d->h * (d->w / 2);
    } else {
        const seq_outs o = presig_outs(pv_out, v_out, presig);
        rc = dcll_conv_lif_sequence_run(d, spk_in, W, b, tau4, eps0, eps1, arp, spk_out, o.pv, o.v, ro_Wp, ro_b, ro_out,
                                        n_ro, state_scratch, T, B, st);
        per_step = (long)B * d->c_out * d->h * d->w;
        if (!rc) rc = presig_copy(o, per_step * T, st, who);
    }
    if (rc || !pv_lowhigh) return rc;
    return launch_pv_lowhigh(pv_out, per_step, T, iter0, (unsigned long long *)pv_lowhigh, st, who, presig);
}

static int launch_c1(const dcll_conv_desc *d, const int32_t *cells, const float *iq, const float *thr_i,
                     const float *thr_q, dcll_iq_tail tail, int L, int t0, dcll_wsrc W, const float *b, const float *tau4, float *eps0,
                     float *eps1, float *arp, uint32_t *spk_out, float *pv_out, float *v_out, float *state_scratch,
                     uint64_t *pv_lowhigh, int iter0, bool presig, int T, int B, hipStream_t st)
{
    const char *who = "dcll_conv_lif_sequence_cells/_iq";
    const long per_step = (long)B * d->c_out * d->h * d->w;
    if (pv_lowhigh) {       // statistics pass over the sampled steps' pv planes after the layer kernel
        int rc = launch_c1(d, cells, iq, thr_i, thr_q, tail, L, t0, W, b, tau4, eps0, eps1, arp, spk_out, pv_out, v_out,
                           state_scratch, nullptr, 0, presig, T, B, st);
        if (rc) return rc;
        return launch_pv_lowhigh(pv_out, per_step, T, iter0, (unsigned long long *)pv_lowhigh, st, who, presig);
    }
    // both the presigmoid pv_out and v_out wanted (tests): v_out is written, pv_out is a copy of it
    if (presig && pv_out && v_out) {
        int rc = launch_c1(d, cells, iq, thr_i, thr_q, tail, L, t0, W, b, tau4, eps0, eps1, arp, spk_out, nullptr, v_out,
                           state_scratch, nullptr, 0, false, T, B, st);
        if (rc) return rc;
        return presig_copy(seq_outs{nullptr, v_out, pv_out}, per_step * T, st, who);
    }
    if (d->h != 16 || d->w != 16)       // large plane: k_lif_seq_c1t, one workgroup per (sample, 8 x 32 tile)
        return dcll_launch_seq_c1t(d, cells, iq, thr_i, thr_q, tail, L, t0, W, b, tau4, eps0, eps1, arp, spk_out, pv_out, v_out,
                                   state_scratch, T, B, st, presig);
    const bool fastpath = d->c_out == 32 && spk_out && pv_out && !v_out;
    if (presig && !fastpath) { v_out = pv_out; pv_out = nullptr; }
#define DCLL_LAUNCH_C1(R, F, Q)                                                                                         \
    hipLaunchKernelGGL((k_lif_seq_c1<R, F, Q>), dim3(B), dim3(256), 0, st, d->c_out, cells, iq, thr_i, thr_q, tail, L, t0, W,  \
                       b, tau4, eps0, eps1, arp, spk_out, pv_out, v_out, T, B, d->alpharp, d->wrp)
#define DCLL_LAUNCH_C1Q(R, F)                                                                                           \
    do { if (iq) DCLL_LAUNCH_C1(R, F, true); else DCLL_LAUNCH_C1(R, F, false); } while (0)
    if (d->refractory) {
        if (fastpath && presig) DCLL_LAUNCH_C1Q(true, 2);
        else if (fastpath) DCLL_LAUNCH_C1Q(true, 1);
        else DCLL_LAUNCH_C1Q(true, 0);
    } else {
        if (fastpath && presig) DCLL_LAUNCH_C1Q(false, 2);
        else if (fastpath) DCLL_LAUNCH_C1Q(false, 1);
        else DCLL_LAUNCH_C1Q(false, 0);
    }
#undef DCLL_LAUNCH_C1Q
#undef DCLL_LAUNCH_C1
    HIP_CHECK_LAUNCH("k_lif_seq_c1");
    return DCLL_OK;
}

extern "C" int dcll_conv_lif_sequence_cells(const dcll_conv_desc *d, const int32_t *cells, const float *Wf, const float *b,
                                            const float *tau4, float *eps0, float *eps1, float *arp, uint32_t *spk_out,
                                            float *pv_out, float *v_out, float *state_scratch, uint64_t *pv_lowhigh,
                                            int32_t iter0, const dcll_layer_opts *opts, int32_t T, int32_t B, void *stream)
{
    const char *who = "dcll_conv_lif_sequence_cells";
    int rc = check_desc(d);
    if (rc) return rc;
    const bool w3 = dcll_seq_w3_geometry(d) && d->c_in == 1;
    if (!w3) {
        rc = check_seq_geometry(d, 1, who);
        if (rc) return rc;
    }
    if (T == 0 || B == 0) return DCLL_OK;      // empty input: nothing to do (its pointers may be NULL)
    rc = check_opts(Wf, opts, true, who);
    if (rc) return rc;
    if (!cells || !b || !tau4 || !eps0 || !eps1) return fail(DCLL_ERR_INVALID, "null pointer", who);
    if (d->refractory && !arp) return fail(DCLL_ERR_INVALID, "refractory layer needs arp", who);
    if (T < 0 || B < 0) return fail(DCLL_ERR_INVALID, "negative size", who);
    const bool presig = opts && opts->pv_presigmoid;
    const dcll_wsrc W = make_wsrc(Wf, opts);
    if (w3) {       // first layer of radio_ml_conv_ref.yaml: pooled outputs (dcll_seq_w3.hip)
        rc = dcll_launch_seq_w3(d, nullptr, cells, W, b, tau4, eps0, eps1, arp, spk_out, pv_out, v_out, presig, T, B,
                                (hipStream_t)stream);
        if (rc || !pv_lowhigh) return rc;
        return launch_pv_lowhigh(pv_out, (long)B * d->c_out * d->h * (d->w / 2), T, iter0, (unsigned long long *)pv_lowhigh,
                                 (hipStream_t)stream, who, presig);
    }
    return launch_c1(d, cells, nullptr, nullptr, nullptr, make_iq_tail(nullptr), 0, 0, W, b, tau4, eps0, eps1, arp, spk_out, pv_out, v_out,
                     state_scratch, pv_lowhigh, iter0, presig, T, B, (hipStream_t)stream);
}

extern "C" int dcll_conv_lif_sequence_iq(const dcll_conv_desc *d, const float *iq, const float *thr_i, const float *thr_q,
                                         const dcll_iq_tail *tail, int32_t L, int32_t t0, const float *Wf, const float *b, const float *tau4,
                                         float *eps0, float *eps1, float *arp, uint32_t *spk_out, float *pv_out,
                                         float *v_out, float *state_scratch, uint64_t *pv_lowhigh, int32_t iter0,
                                         const dcll_layer_opts *opts, int32_t T, int32_t B, void *stream)
{
    const char *who = "dcll_conv_lif_sequence_iq";
    int rc = check_seq_geometry(d, 1, who);
    if (rc) return rc;
    if (T == 0 || B == 0) return DCLL_OK;      // empty input: nothing to do (its pointers may be NULL)
    rc = check_opts(Wf, opts, true, who);
    if (rc) return rc;
    if (!iq || !thr_i || !thr_q || !b || !tau4 || !eps0 || !eps1) return fail(DCLL_ERR_INVALID, "null pointer", who);
    if (tail && tail->tail_mask && (!tail->thr_i_tail || !tail->thr_q_tail)) return fail(DCLL_ERR_INVALID, "tail mask without tail tables", who);
    if (d->refractory && !arp) return fail(DCLL_ERR_INVALID, "refractory layer needs arp", who);
    if (T < 0 || B < 0 || t0 < 0 || t0 + T > L) return fail(DCLL_ERR_INVALID, "window [t0, t0+T) outside the IQ row", who);
    if (T > C1_MAXT) return fail(DCLL_ERR_UNSUPPORTED, "at most 4096 timesteps per launch", who);
    return launch_c1(d, nullptr, iq, thr_i, thr_q, make_iq_tail(tail), L, t0, make_wsrc(Wf, opts), b, tau4, eps0, eps1, arp, spk_out, pv_out,
                     v_out, state_scratch, pv_lowhigh, iter0, opts && opts->pv_presigmoid, T, B, (hipStream_t)stream);
}

extern "C" int dcll_argmax_vote(const float *logits, int32_t *clout, int32_t *vote, int32_t T, int32_t B, int32_t N,
                                int32_t t_begin, void *stream)
{
    if (T == 0 || B == 0) return DCLL_OK;
    if (!logits || !clout || T < 0 || B < 0 || N < 1) return fail(DCLL_ERR_INVALID, "dcll_argmax_vote: bad argument");
    if (vote && N > VOTE_MAXN) return fail(DCLL_ERR_UNSUPPORTED, "dcll_argmax_vote: vote supports at most 64 classes");
    if (T == 0 || B == 0) return DCLL_OK;
    hipStream_t st = (hipStream_t)stream;
    const long rows = (long)T * B;
    hipLaunchKernelGGL(k_argmax, dim3(nblk(rows, 256)), dim3(256), 0, st, logits, clout, rows, N);
    HIP_CHECK_LAUNCH("k_argmax");
    if (vote) {
        hipLaunchKernelGGL(k_vote, dim3(nblk(B, 64)), dim3(64), 0, st, clout, vote, T, B, N, t_begin);
        HIP_CHECK_LAUNCH("k_vote");
    }
    return DCLL_OK;
}

extern "C" int dcll_iq_encode(const float *iq, const float *thr_i, const float *thr_q, const dcll_iq_tail *tail,
                              int32_t *cells, int32_t B,
                              int32_t L, int32_t t0, int32_t T, int32_t w, int32_t h, void *stream)
{
    if (!iq || !thr_i || !thr_q || !cells || B < 0 || T < 0 || t0 < 0 || t0 + T > L || w < 1 || h < 1)
        return fail(DCLL_ERR_INVALID, "dcll_iq_encode: bad argument");
    if (tail && tail->tail_mask && (!tail->thr_i_tail || !tail->thr_q_tail))
        return fail(DCLL_ERR_INVALID, "dcll_iq_encode: tail mask without tail tables");
    if (T == 0 || B == 0) return DCLL_OK;
    hipLaunchKernelGGL(k_iq_encode, dim3(nblk((long)T * B, 256)), dim3(256), 0, (hipStream_t)stream, iq, thr_i, thr_q,
                       make_iq_tail(tail), cells, B, L, t0, T, w, h);
    HIP_CHECK_LAUNCH("k_iq_encode");
    return DCLL_OK;
}

extern "C" int dcll_unpack_spikes(const uint32_t *packed, float *dense, int64_t nwords, void *stream)
{
    if (nwords == 0) return DCLL_OK;
    if (!packed || !dense || nwords < 0) return fail(DCLL_ERR_INVALID, "dcll_unpack_spikes: bad argument");
    if (nwords == 0) return DCLL_OK;
    hipLaunchKernelGGL(k_unpack, dim3(nblk(nwords * 32, 256)), dim3(256), 0, (hipStream_t)stream, packed, dense, (long)nwords);
    HIP_CHECK_LAUNCH("k_unpack");
    return DCLL_OK;
}

extern "C" int dcll_pack_spikes(const float *dense, uint32_t *packed, int64_t nwords, void *stream)
{
    if (nwords == 0) return DCLL_OK;
    if (!packed || !dense || nwords < 0) return fail(DCLL_ERR_INVALID, "dcll_pack_spikes: bad argument");
    if (nwords == 0) return DCLL_OK;
    hipLaunchKernelGGL(k_pack, dim3(nblk(nwords * 32, 256)), dim3(256), 0, (hipStream_t)stream, dense, packed, (long)nwords);
    HIP_CHECK_LAUNCH("k_pack");
    return DCLL_OK;
}
