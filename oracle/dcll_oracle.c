/*
 * ORACLE (test infrastructure, not product): plain-C restatement of the reference's DCLL layer step with a
 * PINNED summation order.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * What it restates (paths relative to the reference checkout):
 *   traces + conv + refractory + threshold     dcll/pytorch_libdcll.py:493-503 (refractory), :415-420 (plain)
 *   max-pool of spikes and pv, i2o / output_   dcll/pytorch_libdcll.py:601-606
 *   dense twin                                 dcll/pytorch_libdcll.py:139-148, :179-195, :250-255
 *   per-step argmax + vote                     dcll/pytorch_libdcll.py:44-56, :724-728
 *
 * Why a second oracle beside oracle/torch_ref.py: the reference's F.conv2d runs in oneDNN, whose accumulation
 * order is unspecified, so "bit-exact spikes" can only be asserted against an implementation whose order is
 * pinned.  The order pinned here is the contract of include/dcll_hip.h:
 *     acc = bias[co];  for cp in ci-pairs: for ky: for kx: for h in {0,1}: ci = 2cp+h (skipped if ci >= c_in)
 *         acc = fmaf(eps1_zero_padded[ci][y+ky-pad][x+kx-pad], W[co][ci][ky][kx], acc)
 * which is what v_mfma_f32_32x32x2_f32 computes (a k-ordered fmaf chain, one rounding per product) when the two
 * k-lanes of the instruction carry an input-channel pair.  Traces are three separately rounded ops per line
 * (build with -ffp-contract=off).  This oracle is pinned against the reference by tests/golden (traces
 * bit-exact; spikes teacher-forced with flips only inside the rounding band; logits 1e-4).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/dcll_hip.h"

static void out_shape(const dcll_conv_desc *d, int *ch, int *cw, int *ph, int *pw)
{
    /* get_output_shape, pytorch_libdcll.py:368-375 (stride = dilation = 1: h + 2 pad - kh + 1) */
    *ch = (d->h + 2 * d->pad_h - d->dilation * (d->kh - 1) - 1) / d->stride + 1;
    *cw = (d->w + 2 * d->pad_w - d->dilation * (d->kw - 1) - 1) / d->stride + 1;
    /* MaxPool2d(kernel=stride=pool, padding=(pool-1)/2), floor mode */
    *ph = (*ch + 2 * ((d->pool_h - 1) / 2) - d->pool_h) / d->pool_h + 1;
    *pw = (*cw + 2 * ((d->pool_w - 1) / 2) - d->pool_w) / d->pool_w + 1;
}

int dcll_oracle_conv_out_shape(const dcll_conv_desc *d, int32_t *ch, int32_t *cw, int32_t *ph, int32_t *pw)
{
    int a, b, c, e;
    out_shape(d, &a, &b, &c, &e);
    *ch = a; *cw = b; *ph = c; *pw = e;
    return 0;
}

static inline float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

/* pytorch_libdcll.py:493-494 — each product and each sum rounded separately. */
static inline void trace_update(float x, float alpha, float tau_m, float alphas, float tau_s, float *e0, float *e1)
{
    volatile float a = x * tau_s;
    volatile float b = alphas * (*e0);
    float n0 = a + b;
    volatile float c = alpha * (*e1);
    volatile float dd = n0 * tau_m;
    *e0 = n0;
    *e1 = c + dd;
}

int dcll_oracle_conv_lif_step(const dcll_conv_desc *d, const float *x, const float *W, const float *b,
                              const float *alpha, const float *tau_m, const float *alphas, const float *tau_s,
                              float *eps0, float *eps1, float *arp,
                              const float *i2o_W, const float *i2o_b, const float *out_W, const float *out_b,
                              float *out_s, float *out_p, float *out_o, float *out_pv, float *out_v, int32_t B)
{
    if (d->stride < 1 || d->dilation < 1 || d->groups < 1 || d->c_in % d->groups || d->c_out % d->groups) return DCLL_ERR_INVALID;
    int ch, cw, ph, pw;
    out_shape(d, &ch, &cw, &ph, &pw);
    const int C = d->c_in, O = d->c_out, H = d->h, Wd = d->w, KH = d->kh, KW = d->kw;
    const int HP = H + 2 * d->pad_h, WP = Wd + 2 * d->pad_w;
    const int plane = H * Wd, cplane = ch * cw, pplane = ph * pw;
    const int pool_ph = (d->pool_h - 1) / 2, pool_pw = (d->pool_w - 1) / 2;
    /* Three phases, each parallel over independent work items (the arithmetic of every output is the chain above, in the
     * order above, whatever the loop schedule): (1) traces per sample, (2) the conv chains per (sample, output channel)
     * with the chains of one output ROW advanced together — `omp simd` over x: a lane is one output pixel's own chain, every
     * step an exactly rounded fmaf, so vector and scalar execution give the same bits —, (3) pooling + readouts per sample. */
    float *pad = (float *)calloc((size_t)B * C * HP * WP, sizeof(float));
    float *sfull = (float *)malloc((size_t)B * O * cplane * sizeof(float));
    float *pvfull = (float *)malloc((size_t)B * O * cplane * sizeof(float));
    if (!pad || !sfull || !pvfull) { free(pad); free(sfull); free(pvfull); return DCLL_ERR_INVALID; }
    /* (1) traces (elementwise on the layer INPUT) */
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < B; ++bi)
        for (int ci = 0; ci < C; ++ci)
            for (int i = 0; i < plane; ++i) {
                size_t g = ((size_t)bi * C + ci) * plane + i;
                size_t tq = d->tau_is_tensor ? (size_t)ci * plane + i : 0;
                trace_update(x[g], alpha[tq], tau_m[tq], alphas[tq], tau_s[tq], &eps0[g], &eps1[g]);
                pad[(((size_t)bi * C + ci) * HP + (i / Wd + d->pad_h)) * WP + (i % Wd + d->pad_w)] = eps1[g];
            }
    /* (2) conv as the pinned fmaf chain + refractory + threshold */
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < B; ++bi)
        for (int co = 0; co < O; ++co) {
            float accrow[cw];
            const float *padb = pad + (size_t)bi * C * HP * WP;
            for (int y = 0; y < ch; ++y) {
                for (int xx = 0; xx < cw; ++xx) accrow[xx] = b ? b[co] : 0.0f;
                /* F.conv2d's groups / stride / dilation (:417, :495): co sees the cig channels of its group; tap (ky, kx) of
                 * output (y, x) reads padded input (y stride + ky dilation, x stride + kx dilation); the chain runs over the
                 * group's channel pairs.  All three at 1: the plain convolution. */
                const int cig = C / d->groups, grp = co / (O / d->groups), st = d->stride, dl = d->dilation;
                for (int cp = 0; cp < (cig + 1) / 2; ++cp)
                    for (int ky = 0; ky < KH; ++ky)
                        for (int kx = 0; kx < KW; ++kx)
                            for (int hh = 0; hh < 2; ++hh) {
                                int ci = 2 * cp + hh;
                                if (ci >= cig) continue;
                                const float *row = padb + ((size_t)(grp * cig + ci) * HP + (y * st + ky * dl)) * WP + kx * dl;
                                const float w = W[(((size_t)co * cig + ci) * KH + ky) * KW + kx];
#pragma omp simd
                                for (int xx = 0; xx < cw; ++xx) accrow[xx] = fmaf(row[(size_t)xx * st], w, accrow[xx]);
                            }
                for (int xx = 0; xx < cw; ++xx) {
                    const float acc = accrow[xx];
                    size_t og = ((size_t)bi * O + co) * cplane + (size_t)y * cw + xx;
                    float v = acc, s;
                    if (d->refractory) {
                        volatile float a = d->alpharp * arp[og];      /* :497 */
                        v = acc + a;                                  /* :498 */
                        s = v > 0.0f ? 1.0f : 0.0f;                   /* :499 */
                        volatile float sw = s * d->wrp;
                        arp[og] = a - sw;                             /* :503 */
                    } else {
                        s = v > 0.0f ? 1.0f : 0.0f;                   /* :420 */
                    }
                    if (out_v) out_v[og] = v;
                    sfull[og] = s;
                    pvfull[og] = sigmoidf_(v);                        /* :500 / :419 */
                }
            }
        }
    int rc = 0;
    /* (3) max-pool + readouts */
#pragma omp parallel for schedule(static)
    for (int bi = 0; bi < B; ++bi) {
        float *pvp = (float *)malloc((size_t)O * pplane * sizeof(float));
        if (!pvp) { rc = DCLL_ERR_INVALID; continue; }
        const float *sfb = sfull + (size_t)bi * O * cplane, *pvb = pvfull + (size_t)bi * O * cplane;
        /* max-pool (kernel=stride=pool, pad (pool-1)/2 with -inf) :601 */
        for (int co = 0; co < O; ++co)
            for (int py = 0; py < ph; ++py)
                for (int px = 0; px < pw; ++px) {
                    float ms = -INFINITY, mp = -INFINITY;
                    for (int dy = 0; dy < d->pool_h; ++dy)
                        for (int dx = 0; dx < d->pool_w; ++dx) {
                            int yy = py * d->pool_h - pool_ph + dy, xq = px * d->pool_w - pool_pw + dx;
                            if (yy < 0 || yy >= ch || xq < 0 || xq >= cw) continue;
                            float a = sfb[(size_t)co * cplane + yy * cw + xq];
                            float q = pvb[(size_t)co * cplane + yy * cw + xq];
                            if (a > ms) ms = a;
                            if (q > mp) mp = q;
                        }
                    size_t pg = ((size_t)bi * O + co) * pplane + (size_t)py * pw + px;
                    if (out_s) out_s[pg] = ms;
                    if (out_pv) out_pv[pg] = mp;
                    pvp[(size_t)co * pplane + py * pw + px] = mp;
                }
        /* readouts :602-606 (not order-pinned: double accumulation, one final rounding) */
        {
            const int K = O * pplane;
            for (int n = 0; n < d->target; ++n) {
                double acc = i2o_b ? i2o_b[n] : 0.0;
                for (int k = 0; k < K; ++k) acc += (double)pvp[k] * (double)i2o_W[(size_t)n * K + k];
                if (out_p) out_p[(size_t)bi * d->target + n] = (float)acc;
                if (d->output_layer && out_o) {
                    double a2 = out_b ? out_b[n] : 0.0;
                    for (int k = 0; k < K; ++k) a2 += (double)pvp[k] * (double)out_W[(size_t)n * K + k];
                    out_o[(size_t)bi * d->target + n] = (float)a2;
                }
            }
        }
        free(pvp);
    }
    free(pad); free(sfull); free(pvfull);
    return rc;
}

int dcll_oracle_dense_lif_step(const dcll_dense_desc *d, const float *x, const float *W, const float *b,
                               const float *alpha, const float *tau_m, const float *alphas, const float *tau_s,
                               float *eps0, float *eps1, float *arp, const float *i2o_W, const float *i2o_b,
                               float *out_s, float *out_p, float *out_pv, float *out_v, int32_t B)
{
    const int I = d->in_features, O = d->out_features;
#pragma omp parallel for schedule(static)
    for (int bi = 0; bi < B; ++bi) {
        float *pv = (float *)malloc((size_t)O * sizeof(float));
        for (int i = 0; i < I; ++i) {
            size_t g = (size_t)bi * I + i;
            size_t tq = d->tau_is_tensor ? (size_t)i : 0;
            trace_update(x[g], alpha[tq], tau_m[tq], alphas[tq], tau_s[tq], &eps0[g], &eps1[g]);
        }
        for (int o = 0; o < O; ++o) {
            /* same pinned order with "ci" = input feature, no spatial taps: pairs (2cp, 2cp+1) in sequence */
            float acc = b ? b[o] : 0.0f;
            for (int i = 0; i < I; ++i) acc = fmaf(eps1[(size_t)bi * I + i], W[(size_t)o * I + i], acc);
            size_t og = (size_t)bi * O + o;
            float v = acc, s;
            if (d->refractory) {
                volatile float a = d->alpharp * arp[og];
                v = acc + a;
                s = v > 0.0f ? 1.0f : 0.0f;
                volatile float sw = s * d->wrp;
                arp[og] = a - sw;
            } else {
                s = v > 0.0f ? 1.0f : 0.0f;
            }
            if (out_v) out_v[og] = v;
            if (out_s) out_s[og] = s;
            pv[o] = sigmoidf_(v);
            if (out_pv) out_pv[og] = pv[o];
        }
        for (int n = 0; n < d->target; ++n) {
            double acc = i2o_b ? i2o_b[n] : 0.0;
            for (int o = 0; o < O; ++o) acc += (double)pv[o] * (double)i2o_W[(size_t)n * O + o];
            if (out_p) out_p[(size_t)bi * d->target + n] = (float)acc;
        }
        free(pv);
    }
    return 0;
}

/* torch.argmax semantics (first maximum) + Counter.most_common(1) semantics (highest count, first seen wins). */
int dcll_oracle_argmax_vote(const float *logits, int32_t *clout, int32_t *vote, int32_t T, int32_t B, int32_t N,
                            int32_t t_begin)
{
    for (int t = 0; t < T; ++t)
        for (int bi = 0; bi < B; ++bi) {
            const float *l = logits + ((size_t)t * B + bi) * N;
            int best = 0;
            for (int n = 1; n < N; ++n)
                if (l[n] > l[best]) best = n;
            clout[(size_t)t * B + bi] = best;
        }
    if (vote) {
        int *cnt = (int *)malloc((size_t)N * sizeof(int));
        int *first = (int *)malloc((size_t)N * sizeof(int));
        for (int bi = 0; bi < B; ++bi) {
            for (int n = 0; n < N; ++n) { cnt[n] = 0; first[n] = T; }
            for (int t = t_begin; t < T; ++t) {
                int c = clout[(size_t)t * B + bi];
                if (cnt[c]++ == 0) first[c] = t;
            }
            int best = -1;
            for (int n = 0; n < N; ++n) {
                if (cnt[n] == 0) continue;
                if (best < 0 || cnt[n] > cnt[best] || (cnt[n] == cnt[best] && first[n] < first[best])) best = n;
            }
            vote[bi] = best;
        }
        free(cnt); free(first);
    }
    return 0;
}
