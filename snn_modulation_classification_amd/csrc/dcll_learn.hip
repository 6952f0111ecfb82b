// dcll_learn.hip — the pieces of a local-learning step (DCLLBase.train_dcll, dcll/pytorch_libdcll.py:690-718) that sit
// between the layer forward (dcll_conv_lif_step) and its backward (dcll_conv_lif_backward), so that no torch op is left
// in the per-timestep loop (fourth translation unit of libdcll_hip.so):
//   k_loss_grad        gradient (and value) of the local losses crit(pvoutput, target) [+ output_crit(output, target)]
//                      for SmoothL1Loss / MSELoss with mean reduction — what loss.backward() hands to the readouts;
//   k_adam_multi       torch.optim.Adam's update (L2 weight decay, bias correction; train.py:164-168 builds it with
//                      betas = (0, beta), weight_decay = 10) over several parameter tensors in ONE launch;
//   k_cells_to_planes  iq2spiketrain's dense spike planes (data/utils.py:81-82) from device-side cell indices: the
//                      per-step input of the first layer without the host loop and the (T,B,1,H,W) upload.
#include "dcll_internal.h"

// local loss of one logit (value l, derivative g with respect to the logit)
__device__ __forceinline__ void loss_elem(float d, int kind, float &l, float &g)
{
    if (kind == DCLL_LOSS_MSE) {
        l = d * d;
        g = 2.0f * d;
    } else {                                // SmoothL1Loss, beta = 1
        const float a = fabsf(d);
        l = a < 1.0f ? 0.5f * d * d : a - 0.5f;
        g = a < 1.0f ? d : (d > 0.0f ? 1.0f : -1.0f);
    }
}

// gradients: one thread per logit over the whole grid.  The loss VALUE (only wanted by callers that look at it — the
// network's learning loop discards it) is summed by workgroup 0 alone in a fixed order: per-thread strided partial sums,
// DPP tree per wave, the four wave totals added in wave order.
// clout (optional): the per-sample argmax of the logits that DCLLClassification.forward records (:724-728: of o on the
// output layer, else of p; first maximum like torch.argmax) — the rows are being read here anyway.
__global__ __launch_bounds__(256) void k_loss_grad(const float *__restrict__ p, const float *__restrict__ o,
                                                    const float *__restrict__ target, float *__restrict__ g_p,
                                                    float *__restrict__ g_o, float *__restrict__ loss,
                                                    int32_t *__restrict__ clout, int n, int N, int kind)
{
    __shared__ float red[4];
    const float inv = 1.0f / (float)n;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (clout && i < n / N) {
        const float *l = (o ? o : p) + (long)i * N;
        int best = 0;
        float bv = l[0];
        for (int k = 1; k < N; ++k) {
            const float v = l[k];
            if (v > bv) { bv = v; best = k; }
        }
        clout[i] = best;
    }
    if (i < n) {
        const float t = target[i];
        float l, g;
        loss_elem(p[i] - t, kind, l, g);
        g_p[i] = g * inv;
        if (o) {
            loss_elem(o[i] - t, kind, l, g);
            g_o[i] = g * inv;
        }
    }
    if (!loss || blockIdx.x != 0) return;
    float acc = 0.0f;
    for (int k = threadIdx.x; k < n; k += 256) {
        const float t = target[k];
        float l, g;
        loss_elem(p[k] - t, kind, l, g);
        acc += l;
        if (o) {
            loss_elem(o[k] - t, kind, l, g);
            acc += l;
        }
    }
    acc = wave_sum_to_lane63(acc);
    if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *loss = (((red[0] + red[1]) + red[2]) + red[3]) * inv;
}

extern "C" int dcll_local_loss_grad(const float *p, const float *o, const float *target, float *g_p, float *g_o,
                                    float *loss, int32_t *clout, int32_t B, int32_t N, int32_t kind, void *stream)
{
    if (B == 0 || N == 0) return DCLL_OK;
    if (!p || !target || !g_p || B < 0 || N < 0 || (o && !g_o))
        return fail(DCLL_ERR_INVALID, "dcll_local_loss_grad: bad argument");
    if (kind != DCLL_LOSS_SMOOTH_L1 && kind != DCLL_LOSS_MSE)
        return fail(DCLL_ERR_UNSUPPORTED, "dcll_local_loss_grad: SmoothL1Loss (beta 1) and MSELoss, mean reduction");
    if ((long)B * N > (1L << 24)) return fail(DCLL_ERR_UNSUPPORTED, "dcll_local_loss_grad: more than 2^24 logits");
    hipLaunchKernelGGL(k_loss_grad, dim3((B * N + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, o, target, g_p, g_o,
                       loss, clout, B * N, N, kind);
    HIP_CHECK_LAUNCH("k_loss_grad");
    return DCLL_OK;
}

// ------------------------------------------------------------------------------------------------------------
// k_step_readout_finish — everything behind the split-K readout GEMM of ONE layer step in one launch (round 4: the
// per-step paths spent 63 of 340 us per timestep in eight 5-11 us readout launches and three argmax / loss launches):
// one WAVE per batch row, lane n = stacked readout row n (i2o's N1 rows, then output_'s N2 on the output layer: both
// share ONE pass over pv).  The lane adds its slices' partial values in k_readout_sum's order (four quarter sub-sums in
// slice order, ((q0 + q1) + q2) + q3, then the bias: bit-identical to dcll_readout_splitk), writes p / o as separate
// contiguous (rows, N1) / (rows, N2) arrays, the row's argmax as DCLLClassification.forward records it (:724-728: of o
// on the output layer, else of p; first maximum like torch.argmax) and — LEARN — the local-loss gradients of
// k_loss_grad (mean reduction over rows * N1 logits each).
// ------------------------------------------------------------------------------------------------------------
template <bool LEARN>
__device__ __forceinline__ void step_readout_finish_row(const long r, const float *__restrict__ part, const float *__restrict__ bias,
                                                         const long rows, const int N1, const int N2, const int nslice,
                                                         float *__restrict__ p, float *__restrict__ o,
                                                         int32_t *__restrict__ clout, const float *__restrict__ target,
                                                         float *__restrict__ g_p, float *__restrict__ g_o, const int kind)
{
    // one workgroup per batch row: wave g adds quarter g of the slices for column `lane` (eight loads in flight, added in
    // slice order), wave 0 combines the four quarters in k_readout_sum's order and does the row's tail
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int N = N1 + N2;
    const long n_out = rows * N, i = r * N + lane;
    float acc = 0.0f;
    if (lane < N) {
        const int per = (nslice + 3) / 4, s0 = grp * per, s1 = min(nslice, s0 + per);
        for (int sl = s0; sl < s1; sl += 8) {
            float t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = sl + k < s1 ? part[(long)(sl + k) * n_out + i] : 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (sl + k < s1) acc += t[k];
        }
    }
    red[grp][lane] = acc;
    __syncthreads();
    if (grp != 0) return;
    float val = -INFINITY;
    if (lane < N) {
        val = (bias ? bias[lane] : 0.0f) + (((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane]);
        if (lane < N1) p[r * N1 + lane] = val;
        else o[r * N2 + (lane - N1)] = val;
        if (LEARN) {
            const int n = lane < N1 ? lane : lane - N1;
            const float inv = 1.0f / (float)(rows * N1);
            float l, g;
            loss_elem(val - target[r * N1 + n], kind, l, g);
            if (lane < N1) g_p[r * N1 + n] = g * inv;
            else g_o[r * N2 + n] = g * inv;
        }
    }
    if (!clout) return;
    // argmax over the lanes of the recorded output: (value, index) butterfly; the larger value wins, among equal values the
    // smaller index (= torch.argmax's first maximum); lanes outside the output carry -inf
    const int lo = N2 > 0 ? N1 : 0, hi = N2 > 0 ? N : N1;
    const bool mine = lane >= lo && lane < hi;
    float bv = mine ? val : -INFINITY;
    int bi = mine ? lane - lo : 0x7fffffff;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        const bool take = ov > bv || (ov == bv && oi < bi);
        bv = take ? ov : bv;
        bi = take ? oi : bi;
    }
    if (lane == 0) clout[r] = bi == 0x7fffffff ? 0 : bi;
}

template <bool LEARN>
__global__ __launch_bounds__(256) void k_step_readout_finish(const float *__restrict__ part, const float *__restrict__ bias,
                                                              long rows, int N1, int N2, int nslice,
                                                              float *__restrict__ p, float *__restrict__ o,
                                                              int32_t *__restrict__ clout, const float *__restrict__ target,
                                                              float *__restrict__ g_p, float *__restrict__ g_o, int kind)
{
    step_readout_finish_row<LEARN>(blockIdx.x, part, bias, rows, N1, N2, nslice, p, o, clout, target, g_p, g_o, kind);
}
// the finishing launches of several layer steps in one (dcll_step_readouts_multi): blockIdx.y selects the item
constexpr int STEP_RO_MAX = 8;
struct step_ro_items {
    const float *part[STEP_RO_MAX], *bias[STEP_RO_MAX], *target[STEP_RO_MAX];
    float *p[STEP_RO_MAX], *o[STEP_RO_MAX], *g_p[STEP_RO_MAX], *g_o[STEP_RO_MAX];
    int32_t *clout[STEP_RO_MAX];
    long rows[STEP_RO_MAX];
    int N1[STEP_RO_MAX], N2[STEP_RO_MAX], nslice[STEP_RO_MAX], kind[STEP_RO_MAX];
};
template <bool LEARN>
__global__ __launch_bounds__(256) void k_step_readout_finish_m(const step_ro_items it)
{
    const int z = blockIdx.y;
    if ((long)blockIdx.x >= it.rows[z]) return;             // (whole workgroups: before the barrier)
    step_readout_finish_row<LEARN>(blockIdx.x, it.part[z], it.bias[z], it.rows[z], it.N1[z], it.N2[z], it.nslice[z], it.p[z],
                                   it.o[z], it.clout[z], it.target[z], it.g_p[z], it.g_o[z], it.kind[z]);
}

extern "C" int64_t dcll_step_readouts_scratch(int64_t rows, int32_t K, int32_t N1, int32_t N2)
{
    if (K >= 65536 || N1 + N2 > 64 || N1 < 1 || N2 < 0) return 0;          // (long rows: dcll_readout_splitk's 4096-column form)
    return dcll_readout_splitk_scratch(rows, K, N1 + N2);
}

extern "C" int dcll_step_readouts(const float *pv, const float *Wt, const float *bias, float *scratch, int64_t scratch_floats,
                                  int64_t rows, int32_t K, int32_t N1, int32_t N2, float *p, float *o, int32_t *clout,
                                  const float *target, float *g_p, float *g_o, int32_t kind, void *stream)
{
    if (rows == 0) return DCLL_OK;
    const int N = N1 + N2;
    if (!pv || !Wt || !scratch || !p || rows < 0 || K < 1 || N1 < 1 || N2 < 0 || (N2 > 0 && (!o || N2 != N1)) ||
        (target && (!g_p || (N2 > 0 && !g_o))))
        return fail(DCLL_ERR_INVALID, "dcll_step_readouts: bad argument");
    if (target && kind != DCLL_LOSS_SMOOTH_L1 && kind != DCLL_LOSS_MSE)
        return fail(DCLL_ERR_UNSUPPORTED, "dcll_step_readouts: SmoothL1Loss (beta 1) and MSELoss, mean reduction");
    const int64_t need = dcll_step_readouts_scratch(rows, K, N1, N2);
    if (need == 0 || ((((uintptr_t)pv | (uintptr_t)Wt)) & 15) != 0)
        return fail(DCLL_ERR_UNSUPPORTED, "dcll_step_readouts: needs rows <= 2048, 2048 <= K < 65536, K % 256 == 0, N1 + N2 <= 64 "
                                          "and 16-byte aligned operands (else dcll_readout + dcll_argmax_vote / dcll_local_loss_grad)");
    if (scratch_floats < need) return fail(DCLL_ERR_INVALID, "dcll_step_readouts: scratch too small (dcll_step_readouts_scratch)");
    hipStream_t st = (hipStream_t)stream;
    const int ks = (int)((int64_t)K * rows * N / need);        // the slice width dcll_readout_splitk uses for this shape
    int rc = dcll_launch_readout_t16(pv, Wt, nullptr, scratch, rows, K, N, ks, st);
    if (rc) return rc;
    const dim3 grid((unsigned)rows);
    if (target)
        hipLaunchKernelGGL(k_step_readout_finish<true>, grid, dim3(256), 0, st, scratch, bias, (long)rows, N1, N2, K / ks, p, o,
                           clout, target, g_p, g_o, kind);
    else
        hipLaunchKernelGGL(k_step_readout_finish<false>, grid, dim3(256), 0, st, scratch, bias, (long)rows, N1, N2, K / ks, p, o,
                           clout, target, g_p, g_o, kind);
    HIP_CHECK_LAUNCH("k_step_readout_finish");
    return DCLL_OK;
}

// The readout tails of n layer steps (the slices of one network timestep) in TWO launches instead of 2 n: one split-K pass over
// all pv maps (k_readout_t16m), one finishing launch (k_step_readout_finish_m).  Results per item = dcll_step_readouts on it,
// bit for bit (same slice widths, same summation order).
extern "C" int dcll_step_readouts_multi(const dcll_step_ro *items, int32_t n, void *stream)
{
    if (n == 0) return DCLL_OK;
    if (!items || n < 0 || n > STEP_RO_MAX) return fail(DCLL_ERR_INVALID, "dcll_step_readouts_multi: 1 .. 8 items");
    const float *pv[STEP_RO_MAX], *Wt[STEP_RO_MAX];
    float *out[STEP_RO_MAX];
    long rows[STEP_RO_MAX];
    int K[STEP_RO_MAX], N[STEP_RO_MAX], ks[STEP_RO_MAX];
    step_ro_items it;
    memset(&it, 0, sizeof(it));
    long maxrows = 0;
    const bool learn = items[0].target != nullptr;
    for (int i = 0; i < n; ++i) {
        const dcll_step_ro &a = items[i];
        if (a.reserved != 0) return fail(DCLL_ERR_INVALID, "dcll_step_readouts_multi: dcll_step_ro.reserved must be 0");
        if (a.rows < 1) return fail(DCLL_ERR_INVALID, "dcll_step_readouts_multi: empty item (call dcll_step_readouts for it)");
        const int Nn = a.N1 + a.N2;
        if (!a.pv || !a.Wt || !a.scratch || !a.p || a.K < 1 || a.N1 < 1 || a.N2 < 0 || (a.N2 > 0 && (!a.o || a.N2 != a.N1)) ||
            (a.target && (!a.g_p || (a.N2 > 0 && !a.g_o))))
            return fail(DCLL_ERR_INVALID, "dcll_step_readouts_multi: bad argument");
        if ((a.target != nullptr) != learn)
            return fail(DCLL_ERR_INVALID, "dcll_step_readouts_multi: either every item has a target or none");
        if (a.target && a.kind != DCLL_LOSS_SMOOTH_L1 && a.kind != DCLL_LOSS_MSE)
            return fail(DCLL_ERR_UNSUPPORTED, "dcll_step_readouts_multi: SmoothL1Loss (beta 1) and MSELoss, mean reduction");
        const int64_t need = dcll_step_readouts_scratch(a.rows, a.K, a.N1, a.N2);
        if (need == 0 || ((((uintptr_t)a.pv | (uintptr_t)a.Wt)) & 15) != 0)
            return fail(DCLL_ERR_UNSUPPORTED, "dcll_step_readouts_multi: an item dcll_step_readouts does not serve");
        if (a.scratch_floats < need) return fail(DCLL_ERR_INVALID, "dcll_step_readouts_multi: scratch too small (dcll_step_readouts_scratch)");
        for (int j = 0; j < i; ++j)
            if (items[j].scratch == a.scratch) return fail(DCLL_ERR_INVALID, "dcll_step_readouts_multi: items must not share scratch");
        pv[i] = a.pv, Wt[i] = a.Wt, out[i] = a.scratch, rows[i] = a.rows, K[i] = a.K, N[i] = Nn;
        ks[i] = (int)((int64_t)a.K * a.rows * Nn / need);       // the slice width dcll_readout_splitk uses for this shape
        it.part[i] = a.scratch, it.bias[i] = a.bias, it.target[i] = a.target, it.p[i] = a.p, it.o[i] = a.o, it.g_p[i] = a.g_p;
        it.g_o[i] = a.g_o, it.clout[i] = a.clout, it.rows[i] = a.rows, it.N1[i] = a.N1, it.N2[i] = a.N2, it.nslice[i] = a.K / ks[i];
        it.kind[i] = a.kind;
        maxrows = max(maxrows, (long)a.rows);
    }
    hipStream_t st = (hipStream_t)stream;
    int rc = dcll_launch_readout_t16_multi(pv, Wt, out, rows, K, N, ks, n, st);
    if (rc) return rc;
    const dim3 grid((unsigned)maxrows, (unsigned)n);
    if (learn) hipLaunchKernelGGL(k_step_readout_finish_m<true>, grid, dim3(256), 0, st, it);
    else hipLaunchKernelGGL(k_step_readout_finish_m<false>, grid, dim3(256), 0, st, it);
    HIP_CHECK_LAUNCH("k_step_readout_finish_m");
    return DCLL_OK;
}

// torch.optim.Adam (amsgrad = False, maximize = False), one thread per parameter element of the concatenated tensors:
//   g = grad + weight_decay * p ; m = lerp(m, g, 1 - beta1) ; v = beta2 * v + (1 - beta2) * g * g
//   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)         bc1 = 1 - beta1^step, bc2 = 1 - beta2^step (host, float64)
// hyper-parameters per tensor: the tensors of several optimizers go into one launch
struct adam_args {
    dcll_adam_tensor t[DCLL_ADAM_MAX_TENSORS];
    long first[DCLL_ADAM_MAX_TENSORS + 1];      // prefix sums of n, in 256-element blocks
    float inv_bc1[DCLL_ADAM_MAX_TENSORS], inv_sqrt_bc2[DCLL_ADAM_MAX_TENSORS];
    const float *dyn;                           // optional, DEVICE: per tensor (lr, 1/bc1, 1/sqrt(bc2)) — see dcll_adam_step_dyn
    int n_tensors;
};

// the update of element i of tensor k, its gradient given (the ONE place the arithmetic lives: k_adam_multi and
// k_grad_reduce_adam give the same bits for the same gradient)
__device__ __forceinline__ void adam_update(const adam_args &a, int k, long i, float grad, float p, float m, float v);
__device__ __forceinline__ void adam_element(const adam_args &a, int k, long i, float grad)
{
    const dcll_adam_tensor &t = a.t[k];
    adam_update(a, k, i, grad, t.param[i], t.exp_avg[i], t.exp_avg_sq[i]);
}
// (parameter and moments already in registers: k_grad_reduce_adam requests them before it adds the partial rows)
__device__ __forceinline__ void adam_update(const adam_args &a, int k, long i, float grad, float p, float m, float v)
{
    const dcll_adam_tensor &t = a.t[k];
    const float g = grad + t.weight_decay * p;
    const float w = 1.0f - t.beta1;
    m = w < 0.5f ? m + w * (g - m) : g - (g - m) * (1.0f - w);                 // torch's lerp
    v = v * t.beta2 + ((1.0f - t.beta2) * g) * g;
    const float lr = a.dyn ? a.dyn[3 * k] : t.lr;
    const float ibc1 = a.dyn ? a.dyn[3 * k + 1] : a.inv_bc1[k], isbc2 = a.dyn ? a.dyn[3 * k + 2] : a.inv_sqrt_bc2[k];
    const float denom = sqrtf(v) * isbc2 + t.eps;
    p = p - (lr * ibc1) * (m / denom);
    t.exp_avg[i] = m;
    t.exp_avg_sq[i] = v;
    t.param[i] = p;
}

__global__ __launch_bounds__(256) void k_adam_multi(adam_args a)
{
    int k = 0;
    while (k + 1 < a.n_tensors && (long)blockIdx.x >= a.first[k + 1]) ++k;      // wave-uniform
    const long i = ((long)blockIdx.x - a.first[k]) * 256 + threadIdx.x;
    if (i >= a.t[k].n) return;
    adam_element(a, k, i, a.t[k].grad[i]);
}

static int adam_launch(const dcll_adam_tensor *tensors, int32_t n_tensors, const float *dyn, void *stream);

extern "C" int dcll_adam_step(const dcll_adam_tensor *tensors, int32_t n_tensors, void *stream)
{
    return adam_launch(tensors, n_tensors, nullptr, stream);
}

extern "C" int dcll_adam_step_dyn(const dcll_adam_tensor *tensors, int32_t n_tensors, const float *dyn, void *stream)
{
    if (n_tensors > 0 && !dyn) return fail(DCLL_ERR_INVALID, "dcll_adam_step_dyn: null dyn");
    return adam_launch(tensors, n_tensors, dyn, stream);
}

// tensors -> adam_args with `per_block` elements per workgroup; tensors with skip[k] set get no blocks of their own
static int adam_fill(adam_args &a, const dcll_adam_tensor *tensors, int32_t n_tensors, const float *dyn, int per_block,
                     const bool *skip, long *blocks_out, const char *who)
{
    if (!tensors || n_tensors < 0 || n_tensors > DCLL_ADAM_MAX_TENSORS)
        return fail(DCLL_ERR_INVALID, "bad argument (1..8 tensors)", who);
    a.dyn = dyn;
    long blocks = 0;
    for (int k = 0; k < n_tensors; ++k) {
        if (!tensors[k].param || !tensors[k].grad || !tensors[k].exp_avg || !tensors[k].exp_avg_sq || tensors[k].n < 0 ||
            tensors[k].step < 1)
            return fail(DCLL_ERR_INVALID, "null tensor or step < 1", who);
        a.t[k] = tensors[k];
        a.first[k] = blocks;
        if (!(skip && skip[k])) blocks += (tensors[k].n + per_block - 1) / per_block;
        a.inv_bc1[k] = (float)(1.0 / (1.0 - pow((double)tensors[k].beta1, (double)tensors[k].step)));
        a.inv_sqrt_bc2[k] = (float)(1.0 / sqrt(1.0 - pow((double)tensors[k].beta2, (double)tensors[k].step)));
    }
    a.first[n_tensors] = blocks;
    a.n_tensors = n_tensors;
    *blocks_out = blocks;
    return DCLL_OK;
}

static int adam_launch(const dcll_adam_tensor *tensors, int32_t n_tensors, const float *dyn, void *stream)
{
    if (n_tensors == 0) return DCLL_OK;
    adam_args a;
    long blocks = 0;
    int rc = adam_fill(a, tensors, n_tensors, dyn, 256, nullptr, &blocks, "dcll_adam_step");
    if (rc) return rc;
    if (blocks == 0) return DCLL_OK;
    hipLaunchKernelGGL(k_adam_multi, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    HIP_CHECK_LAUNCH("k_adam_multi");
    return DCLL_OK;
}

// ------------------------------------------------------------------------------------------------------------
// k_grad_reduce_adam — the end of a learning timestep in ONE launch (round 5; before: a k_bwd_reduce4 per layer + k_adam_multi,
// four dependent launches of 5-8 us around 50 k-element arrays): the weight-gradient partial rows that
// dcll_conv_lif_backward_open left in the layers' scratch are added in the fixed order of k_bwd_reduce4 / k_bwd_reduce
// (same grouping by row count: bit-identical gradients), written to dW / db, and the thread that holds a finished
// gradient element applies torch.optim.Adam's update to its parameter element right there (adam_element).  Tensors no
// layer refers to (output_.weight / output_.bias, whose gradients k_bwd_outgrad_mfma wrote) get the plain elementwise
// update in further workgroups of the same launch.
// ------------------------------------------------------------------------------------------------------------
struct reduce_adam_args {
    adam_args a;
    dcll_grad_parts L[DCLL_REDUCE_MAX_LAYERS];
    long lfirst[DCLL_REDUCE_MAX_LAYERS + 1];     // prefix sums of the layers' workgroups (64 gradient elements each)
    int groups[DCLL_REDUCE_MAX_LAYERS];          // partial-row groups: 16 / 4 / 1, as the standalone reduce kernels choose
    int n_layers;
};

__global__ __launch_bounds__(1024) void k_grad_reduce_adam(reduce_adam_args ra)
{
    __shared__ float red[16][64];
    const long blk = blockIdx.x;
    if (blk >= ra.lfirst[ra.n_layers]) {                        // plain Adam workgroups: 1024 elements each
        const long b2 = blk - ra.lfirst[ra.n_layers];
        int k = 0;
        while (k + 1 < ra.a.n_tensors && b2 >= ra.a.first[k + 1]) ++k;
        const long i = (b2 - ra.a.first[k]) * 1024 + threadIdx.x;
        if (i < ra.a.t[k].n) adam_element(ra.a, k, i, ra.a.t[k].grad[i]);
        return;
    }
    int l = 0;
    while (l + 1 < ra.n_layers && blk >= ra.lfirst[l + 1]) ++l;     // wave-uniform
    const dcll_grad_parts &P = ra.L[l];
    const int GR = ra.groups[l];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long i = (blk - ra.lfirst[l]) * 64 + lane;
    const long total = (long)P.c_out * P.rowlen;
    // the element this lane finishes (group 0 only) and its optimizer operands: requested NOW, so that they land under the
    // partial-row loads (behind the reduction they were a second exposed memory round trip per workgroup: 40 instead of
    // 29 us for the timestep's tail, against the four launches this kernel replaces)
    const int co = (int)(i / P.rowlen);
    const long n = i % P.rowlen;
    const bool isw = n < P.rowlen - 1;
    const int kt = isw ? P.adam_w : P.adam_b;
    const long e = isw ? (long)co * (P.rowlen - 1) + n : co;
    float p0 = 0.0f, m0 = 0.0f, v0 = 0.0f;
    if (grp == 0 && i < total && kt >= 0) {
        p0 = ra.a.t[kt].param[e];
        m0 = ra.a.t[kt].exp_avg[e];
        v0 = ra.a.t[kt].exp_avg_sq[e];
    }
    float acc = 0.0f;
    if (i < total && grp < GR) {
        const int per = (P.nchunk + GR - 1) / GR, c0 = grp * per, c1 = min(P.nchunk, c0 + per);
        const float *src = P.part + i;
        int c = c0;
        if (GR > 1) {
            for (; c + 8 <= c1; c += 8) {                     // eight loads in flight, added in chunk order
                float t[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) t[q] = src[(long)(c + q) * total];
#pragma unroll
                for (int q = 0; q < 8; ++q) acc += t[q];
            }
        }
        for (; c < c1; ++c) acc += src[(long)c * total];
    }
    red[grp][lane] = acc;
    __syncthreads();
    if (grp != 0 || i >= total) return;
    float tot = red[0][lane];
    for (int q = 1; q < GR; ++q) tot += red[q][lane];
    if (isw) P.dW[e] = tot;
    else if (P.db) P.db[co] = tot;
    if (kt >= 0) adam_update(ra.a, kt, e, tot, p0, m0, v0);
}

extern "C" int dcll_grad_reduce_adam(const dcll_grad_parts *layers, int32_t n_layers, const dcll_adam_tensor *tensors,
                                     int32_t n_tensors, const float *dyn, void *stream)
{
    if (n_layers == 0 && n_tensors == 0) return DCLL_OK;
    if (n_layers < 0 || n_layers > DCLL_REDUCE_MAX_LAYERS || (n_layers > 0 && !layers))
        return fail(DCLL_ERR_INVALID, "dcll_grad_reduce_adam: bad argument (0..4 layers)");
    // (the layer loop below reads tensors[adam_w / adam_b]: checked here, before adam_fill's own check is reached)
    if (n_tensors < 0 || n_tensors > DCLL_ADAM_MAX_TENSORS || (n_tensors > 0 && !tensors))
        return fail(DCLL_ERR_INVALID, "dcll_grad_reduce_adam: bad argument (0..8 tensors, non-NULL when n_tensors > 0)");
    reduce_adam_args ra;
    bool taken[DCLL_ADAM_MAX_TENSORS] = {false};
    long blocks = 0;
    for (int l = 0; l < n_layers; ++l) {
        const dcll_grad_parts &P = layers[l];
        if (!P.part || !P.dW || P.nchunk < 1 || P.c_out < 1 || P.rowlen < 2)
            return fail(DCLL_ERR_INVALID, "dcll_grad_reduce_adam: bad layer entry");
        for (int idx : {P.adam_w, P.adam_b}) {
            if (idx < -1 || idx >= n_tensors) return fail(DCLL_ERR_INVALID, "dcll_grad_reduce_adam: tensor index out of range");
            if (idx >= 0) {
                if (taken[idx]) return fail(DCLL_ERR_INVALID, "dcll_grad_reduce_adam: a tensor is referred to twice");
                taken[idx] = true;
            }
        }
        if (P.adam_w >= 0 && tensors[P.adam_w].n != (int64_t)P.c_out * (P.rowlen - 1))
            return fail(DCLL_ERR_INVALID, "dcll_grad_reduce_adam: weight tensor size != c_out * (rowlen - 1)");
        if (P.adam_b >= 0 && tensors[P.adam_b].n != P.c_out)
            return fail(DCLL_ERR_INVALID, "dcll_grad_reduce_adam: bias tensor size != c_out");
        ra.L[l] = P;
        ra.groups[l] = P.nchunk >= 64 ? 16 : P.nchunk >= 16 ? 4 : 1;
        ra.lfirst[l] = blocks;
        blocks += ((long)P.c_out * P.rowlen + 63) / 64;
    }
    ra.lfirst[n_layers] = blocks;
    ra.n_layers = n_layers;
    long ablocks = 0;
    if (n_tensors > 0) {
        int rc = adam_fill(ra.a, tensors, n_tensors, dyn, 1024, taken, &ablocks, "dcll_grad_reduce_adam");
        if (rc) return rc;
    } else {
        ra.a.n_tensors = 0;
        ra.a.dyn = nullptr;
        ra.a.first[0] = 0;
    }
    if (blocks + ablocks == 0) return DCLL_OK;
    if (blocks + ablocks > 0x7fffffffL) return fail(DCLL_ERR_INVALID, "dcll_grad_reduce_adam: too many elements");
    hipLaunchKernelGGL(k_grad_reduce_adam, dim3((unsigned)(blocks + ablocks)), dim3(1024), 0, (hipStream_t)stream, ra);
    HIP_CHECK_LAUNCH("k_grad_reduce_adam");
    return DCLL_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Local learning on a DENSE slice (round 6): the backward of one DenseDCLLlayer step (dcll/pytorch_libdcll.py:250-255 under
// DCLLBase.train_dcll :690-718 — the optimizer is built from dclllayer.i2h.parameters() whatever the layer type, :634-635).
//   dv[b,o] = (sum_n g_p[b,n] * i2o_W[n,o] + g_pv[b,o]) * pv[b,o] * (1 - pv[b,o]) + g_v[b,o]     pv = sigmoid(v): sigmoid'
//   dW[o,k] = sum_b dv[b,o] * eps1[b,k]        db[o] = sum_b dv[b,o]
// (i2o is frozen, the spikes and the neuron state are detached: nothing else carries a gradient.)
// k_dense_bwd_dv: one thread per (b, o), the readout rows ascending (fmaf).
// k_dense_bwd_wgrad: an fp32-MFMA GEMM dv^T x [eps1 | 1] over a batch chunk per workgroup: v_mfma_f32_32x32x2_f32 with the
//   two k lanes on a SAMPLE pair, A = dv (lane = output neuron), B = eps1 (lane = input feature; the column `in` is the
//   constant 1: the bias gradient is one more column of the same product) — both operands are 32 consecutive floats of a
//   row per half-wave, straight from global memory (L2-resident: a chunk's dv rows are re-read by every feature tile).
//   Partial rows part[chunk][o][in + 1] in dcll_grad_parts' layout: dcll_grad_reduce_adam finishes them (fixed chunk order)
//   together with the optimizer step, like the conv layers' — or alone, for the closed form.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dense_bwd_dv(int out, int target, const float *__restrict__ pv,
                                                       const float *__restrict__ g_p, const float *__restrict__ g_pv,
                                                       const float *__restrict__ g_v, const float *__restrict__ i2o_W,
                                                       float *__restrict__ dv, long n)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long b = i / out;
    const int o = (int)(i - b * out);
    float g = g_pv ? g_pv[i] : 0.0f;
    if (g_p)
        for (int r = 0; r < target; ++r) g = __builtin_fmaf(g_p[b * target + r], i2o_W[(long)r * out + o], g);
    const float s = pv[i];
    float d = g * (s * (1.0f - s));
    if (g_v) d += g_v[i];
    dv[i] = d;
}

constexpr int DW_NT = 4;        // feature tiles (32 columns each) per wave: one A fragment feeds four MFMAs
__global__ __launch_bounds__(256) void k_dense_bwd_wgrad(int in, int out, const float *__restrict__ dv,
                                                          const float *__restrict__ eps1, float *__restrict__ part, int B,
                                                          int per_chunk)
{
    const int lane = threadIdx.x & 63, jj = lane & 31, kk = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rowlen = in + 1;
    const int o0 = blockIdx.y * 32;
    const int c0 = (blockIdx.x * 4 + w) * (32 * DW_NT);                 // first column of this wave's tiles
    if (c0 >= rowlen) return;                                           // (whole waves; the kernel has no barrier)
    const int chunk = blockIdx.z;
    const int b0 = chunk * per_chunk, b1 = min(B, b0 + per_chunk);
    f32x16 acc[DW_NT];
#pragma unroll
    for (int t = 0; t < DW_NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const bool ook = o0 + jj < out;
    int col[DW_NT];
#pragma unroll
    for (int t = 0; t < DW_NT; ++t) col[t] = c0 + 32 * t + jj;
    for (int b = b0; b < b1; b += 2) {
        const int bb = b + kk;
        const bool bok = bb < b1;
        const float a = (bok && ook) ? dv[(long)bb * out + o0 + jj] : 0.0f;
        float e[DW_NT];
#pragma unroll
        for (int t = 0; t < DW_NT; ++t)
            e[t] = !bok ? 0.0f : col[t] < in ? eps1[(long)bb * in + col[t]] : col[t] == in ? 1.0f : 0.0f;
#pragma unroll
        for (int t = 0; t < DW_NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, e[t], acc[t], 0, 0, 0);
    }
    // accumulator register r of lane l: row (r & 3) + 8 (r >> 2) + 4 (l >> 5) = output neuron, column l & 31 = feature
    float *prow = part + (long)chunk * out * rowlen;
#pragma unroll
    for (int t = 0; t < DW_NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
            if (o < out && col[t] < rowlen) prow[(long)o * rowlen + col[t]] = acc[t][r];
        }
}

static int dense_backward_impl(const dcll_dense_desc *d, const float *eps1, const float *pv, const float *g_p,
                               const float *g_pv, const float *g_v, const float *i2o_W, float *scratch,
                               int64_t scratch_floats, int32_t B, hipStream_t st, dcll_grad_parts *P, const char *who)
{
    if (!d || d->in_features < 1 || d->out_features < 1) return fail(DCLL_ERR_INVALID, "bad descriptor", who);
    if (!eps1 || !pv || !scratch) return fail(DCLL_ERR_INVALID, "null pointer", who);
    if (g_p && (!i2o_W || d->target < 1)) return fail(DCLL_ERR_INVALID, "g_p needs i2o_W and target >= 1", who);
    if (B < 1) return fail(DCLL_ERR_INVALID, "empty batch", who);
    const int in = d->in_features, out = d->out_features;
    const long ndv = (long)B * out, per_row = (long)out * (in + 1);
    long nchunk = (scratch_floats - ndv) / per_row;
    if (nchunk < 1) return fail(DCLL_ERR_INVALID, "scratch too small (need B*out + k*out*(in+1) floats, k >= 1)", who);
    // chunks of an even number of samples (an MFMA k-step is a sample pair); enough of them to fill the chip with the tile grid
    const long tiles = (long)((in + 1 + 4 * 32 * DW_NT - 1) / (4 * 32 * DW_NT)) * ((out + 31) / 32);
    long want = (1024 + tiles - 1) / tiles;
    if (want > 64) want = 64;
    if (nchunk > want) nchunk = want;
    long per = (B + nchunk - 1) / nchunk;
    per += per & 1;
    nchunk = (B + per - 1) / per;
    hipLaunchKernelGGL(k_dense_bwd_dv, dim3((unsigned)((ndv + 255) / 256)), dim3(256), 0, st, out, g_p ? d->target : 0, pv, g_p, g_pv, g_v, i2o_W,
                       scratch, ndv);
    HIP_CHECK_LAUNCH("k_dense_bwd_dv");
    float *part = scratch + ndv;
    hipLaunchKernelGGL(k_dense_bwd_wgrad, dim3((unsigned)((in + 1 + 4 * 32 * DW_NT - 1) / (4 * 32 * DW_NT)), (unsigned)((out + 31) / 32),
                                               (unsigned)nchunk), dim3(256), 0, st, in, out, scratch, eps1, part, B, (int)per);
    HIP_CHECK_LAUNCH("k_dense_bwd_wgrad");
    P->part = part;
    P->rowlen = in + 1;
    P->nchunk = (int32_t)nchunk;
    P->c_out = out;
    return DCLL_OK;
}

extern "C" int dcll_dense_lif_backward(const dcll_dense_desc *d, const float *eps1, const float *pv, const float *g_p,
                                       const float *g_pv, const float *g_v, const float *i2o_W, float *dW, float *db,
                                       float *scratch, int64_t scratch_floats, int32_t B, void *stream)
{
    const char *who = "dcll_dense_lif_backward";
    if (!dW) return fail(DCLL_ERR_INVALID, "null dW", who);
    dcll_grad_parts P;
    memset(&P, 0, sizeof(P));
    int rc = dense_backward_impl(d, eps1, pv, g_p, g_pv, g_v, i2o_W, scratch, scratch_floats, B, (hipStream_t)stream, &P, who);
    if (rc) return rc;
    P.dW = dW;
    P.db = db;
    P.adam_w = P.adam_b = -1;
    return dcll_grad_reduce_adam(&P, 1, nullptr, 0, nullptr, stream);          // the closed form: reduce only
}

extern "C" int dcll_dense_lif_backward_open(const dcll_dense_desc *d, const float *eps1, const float *pv, const float *g_p,
                                            const float *g_pv, const float *g_v, const float *i2o_W, float *scratch,
                                            int64_t scratch_floats, int32_t B, const float **part, int32_t *nchunk,
                                            void *stream)
{
    const char *who = "dcll_dense_lif_backward_open";
    if (!part || !nchunk) return fail(DCLL_ERR_INVALID, "null part / nchunk", who);
    dcll_grad_parts P;
    memset(&P, 0, sizeof(P));
    int rc = dense_backward_impl(d, eps1, pv, g_p, g_pv, g_v, i2o_W, scratch, scratch_floats, B, (hipStream_t)stream, &P, who);
    if (rc) return rc;
    *part = P.part;
    *nchunk = P.nchunk;
    return DCLL_OK;
}

// planes[t][b][:] = 0 except planes[t][b][cells[t][b]] = 1      (one thread per output float4; hw % 4 == 0)
__global__ __launch_bounds__(256) void k_cells_to_planes(const int32_t *__restrict__ cells, float *__restrict__ planes,
                                                          long n_samples, int hw)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;           // float4 index
    const int q = hw >> 2;
    if (i >= n_samples * q) return;
    const long s = i / q;
    const int off = (int)(i % q) * 4;
    const int c = cells[s] - off;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)c < 4u) v[c] = 1.0f;
    ((f32x4 *)planes)[i] = v;
}

extern "C" int dcll_cells_to_planes(const int32_t *cells, float *planes, int64_t n_samples, int32_t hw, void *stream)
{
    if (n_samples == 0) return DCLL_OK;
    if (!cells || !planes || n_samples < 0 || hw < 4 || hw % 4 != 0)
        return fail(DCLL_ERR_INVALID, "dcll_cells_to_planes: bad argument (h*w must be a multiple of 4)");
    const long n4 = n_samples * (hw / 4);
    hipLaunchKernelGGL(k_cells_to_planes, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cells,
                       planes, (long)n_samples, hw);
    HIP_CHECK_LAUNCH("k_cells_to_planes");
    return DCLL_OK;
}

// ------------------------------------------------------------------------------------------------------------
// k_vote_tally — the per-class tallies of an evaluated batch on the device (what the reference computes on the host from
// clout: accuracy_by_vote / confusion_matrix, dcll/pytorch_libdcll.py:44-61, :740-749; here the summable form the ranks
// all-reduce, parallel.py): per layer the confusion matrix [pred][label] (n x n), the number of correct votes and the
// number of votes.  One workgroup per layer: LDS histogram (integer atomics: the result does not depend on their order),
// every output element written — no zero fill, no partial sums.  (Round 3 built this from nine torch kernels per layer.)
// ------------------------------------------------------------------------------------------------------------
constexpr int TALLY_MAXL = 16;
struct tally_args { const int32_t *votes[TALLY_MAXL]; };

__global__ __launch_bounds__(1024) void k_vote_tally(tally_args a, const int64_t *__restrict__ labels, int64_t *__restrict__ out,
                                                      int B, int n)
{
    extern __shared__ unsigned hist[];                          // n * n bins + [correct]
    const int nn = n * n;
    for (int i = threadIdx.x; i <= nn; i += 1024) hist[i] = 0u;
    __syncthreads();
    const int32_t *v = a.votes[blockIdx.x];
    unsigned ok = 0;
    for (int b = threadIdx.x; b < B; b += 1024) {
        const int pred = v[b];
        const long lab = labels[b];
        if ((unsigned)pred < (unsigned)n && (unsigned long)lab < (unsigned long)n) atomicAdd(&hist[pred * n + (int)lab], 1u);
        ok += (long)pred == lab;
    }
    if (ok) atomicAdd(&hist[nn], ok);
    __syncthreads();
    int64_t *row = out + (long)blockIdx.x * (nn + 2);
    for (int i = threadIdx.x; i <= nn; i += 1024) row[i] = (int64_t)hist[i];
    if (threadIdx.x == 0) row[nn + 1] = B;
}

extern "C" int dcll_vote_tallies(const int32_t *const *votes, int32_t n_layers, const int64_t *labels, int64_t *out, int32_t B,
                                 int32_t n_classes, void *stream)
{
    if (n_layers == 0) return DCLL_OK;
    if (!votes || !labels || !out || n_layers < 0 || B < 0 || n_classes < 1)
        return fail(DCLL_ERR_INVALID, "dcll_vote_tallies: bad argument");
    if (n_layers > TALLY_MAXL || n_classes > 96)
        return fail(DCLL_ERR_UNSUPPORTED, "dcll_vote_tallies: at most 16 layers and 96 classes");
    tally_args a;
    for (int l = 0; l < n_layers; ++l) {
        if (!votes[l]) return fail(DCLL_ERR_INVALID, "dcll_vote_tallies: null vote tensor");
        a.votes[l] = votes[l];
    }
    const size_t lds = (size_t)(n_classes * n_classes + 1) * sizeof(unsigned);
    hipLaunchKernelGGL(k_vote_tally, dim3(n_layers), dim3(1024), lds, (hipStream_t)stream, a, labels, out, B, n_classes);
    HIP_CHECK_LAUNCH("k_vote_tally");
    return DCLL_OK;
}
