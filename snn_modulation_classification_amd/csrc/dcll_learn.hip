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

__global__ __launch_bounds__(256) void k_adam_multi(adam_args a)
{
    int k = 0;
    while (k + 1 < a.n_tensors && (long)blockIdx.x >= a.first[k + 1]) ++k;      // wave-uniform
    const dcll_adam_tensor t = a.t[k];
    const long i = ((long)blockIdx.x - a.first[k]) * 256 + threadIdx.x;
    if (i >= t.n) return;
    float p = t.param[i];
    const float g = t.grad[i] + t.weight_decay * p;
    float m = t.exp_avg[i], v = t.exp_avg_sq[i];
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

static int adam_launch(const dcll_adam_tensor *tensors, int32_t n_tensors, const float *dyn, void *stream)
{
    if (n_tensors == 0) return DCLL_OK;
    if (!tensors || n_tensors < 0 || n_tensors > DCLL_ADAM_MAX_TENSORS)
        return fail(DCLL_ERR_INVALID, "dcll_adam_step: bad argument (1..8 tensors)");
    adam_args a;
    a.dyn = dyn;
    long blocks = 0;
    for (int k = 0; k < n_tensors; ++k) {
        if (!tensors[k].param || !tensors[k].grad || !tensors[k].exp_avg || !tensors[k].exp_avg_sq || tensors[k].n < 0 ||
            tensors[k].step < 1)
            return fail(DCLL_ERR_INVALID, "dcll_adam_step: null tensor or step < 1");
        a.t[k] = tensors[k];
        a.first[k] = blocks;
        blocks += (tensors[k].n + 255) / 256;
        a.inv_bc1[k] = (float)(1.0 / (1.0 - pow((double)tensors[k].beta1, (double)tensors[k].step)));
        a.inv_sqrt_bc2[k] = (float)(1.0 / sqrt(1.0 - pow((double)tensors[k].beta2, (double)tensors[k].step)));
    }
    a.first[n_tensors] = blocks;
    a.n_tensors = n_tensors;
    if (blocks == 0) return DCLL_OK;
    hipLaunchKernelGGL(k_adam_multi, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    HIP_CHECK_LAUNCH("k_adam_multi");
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
