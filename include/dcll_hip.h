/*
 * dcll_hip.h — C ABI of libdcll_hip.so: the MI355X (gfx950) replacement for the per-timestep DCLL layer
 * forward of ohjay/snn-modulation-classification.
 *
 * The reference exposes a Python object protocol, not an FFI; the seam this ABI replaces is
 *     o, p, pv, pvmem = self.dclllayer.forward(input)          dcll/pytorch_libdcll.py:657
 * i.e. Conv2dDCLLlayer.forward (:599-608) -> ContinuousConv2D.forward (:407-426) /
 * ContinuousRelativeRefractoryConv2D.forward (:485-509), and DenseDCLLlayer.forward (:250-255) ->
 * CLLDenseModule.forward (:131-148) / CLLDenseRRPModule.forward (:171-195), plus the per-step vote collection of
 * DCLLClassification.forward (:722-729) and the T-loop of test_radio_ml.py:144-145.
 *
 * Conventions
 *   - plain C: pointers + sizes only, no torch types. All pointers are DEVICE pointers (HBM) unless marked host.
 *   - every buffer is allocated and owned by the caller (PyTorch-ROCm tensors); the library never allocates
 *     persistent device memory and never frees anything.
 *   - tensors are contiguous fp32, NCHW, unless a packed layout is documented at the parameter.
 *   - calls are asynchronous on `stream` (a hipStream_t passed as void*; NULL = the null stream) and never
 *     synchronise internally; thread-safe for distinct (stream, state) pairs.
 *   - return value: DCLL_OK (0) or a negative DCLL_ERR_* code; never throws across the ABI.
 *     dcll_last_error() returns a thread-local, human-readable message for the last failing call.
 *   - arithmetic contract (bit-level, see DESIGN.md "Pinned arithmetic"):
 *       eps0' = x*tau_s + alphas*eps0         three separately rounded fp32 ops (no FMA contraction)
 *       eps1' = alpha*eps1 + eps0'*tau_m      idem
 *       pvmem[co,y,x] = fmaf-chain starting from bias[co], over k = (ci-pair cp, ky, kx, ci = 2cp+h), in that
 *                       nesting order, zero padding contributing fmaf(0, w, acc); one rounding per product
 *       arp' = alpharp*arp ; v = pvmem + arp' ; s = v > 0 ; arp'' = arp' - s*wrp     (refractory variant)
 *       pv = 1/(1+exp(-v)) (not bit-pinned) ; readouts p,o fp32 (not bit-pinned; |err| <= 1e-4)
 */
#ifndef DCLL_HIP_H
#define DCLL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCLL_ABI_VERSION 7

enum {
    DCLL_OK = 0,
    DCLL_ERR_INVALID = -1,      /* bad descriptor / null pointer / size mismatch            */
    DCLL_ERR_UNSUPPORTED = -2,  /* legal in the reference but not implemented by this build */
    DCLL_ERR_LAUNCH = -3        /* HIP runtime reported an error (message has the string)   */
};

/* Geometry + constants of one Conv2dDCLLlayer (ctor: dcll/pytorch_libdcll.py:513-581). */
typedef struct dcll_conv_desc {
    int32_t c_in, c_out;        /* i2h.weight is (c_out, c_in, kh, kw)                                   */
    int32_t h, w;               /* input plane = im_dims                                                 */
    int32_t kh, kw;             /* kernel_size                                                           */
    int32_t pad_h, pad_w;       /* padding                                                               */
    int32_t stride, dilation, groups; /* F.conv2d's (:417, :495); ConvNetwork builds 1,1,1 (networks/__init__.py:132-145).
                                       * Other values (>= 1, groups dividing c_in and c_out; W is then (c_out, c_in/groups,
                                       * kh, kw), the chain runs over the group's channel pairs) are served by the per-step
                                       * calls dcll_conv_lif_step / dcll_conv_lif_backward[_open] on their generic kernels;
                                       * the sequence calls return DCLL_ERR_UNSUPPORTED for them                          */
    int32_t pool_h, pool_w;     /* MaxPool2d(kernel=stride=pool, padding=(pool-1)/2)   :542-549          */
    int32_t target;             /* target_size: rows of i2o.weight (and output_.weight)                  */
    int32_t output_layer;       /* !=0: also o = output_(flatten(pv))  :605-606                          */
    int32_t tau_is_tensor;      /* 0: alpha,tau_m,alphas,tau_s hold 1 float; 1: (c_in,h,w) floats :391-405 */
    int32_t refractory;         /* !=0: wrp > 0 variant (:485-509), arp state used                       */
    float alpharp;              /* :497                                                                  */
    float wrp;                  /* :503                                                                  */
} dcll_conv_desc;

/* Geometry + constants of one DenseDCLLlayer (ctor: dcll/pytorch_libdcll.py:199-242). */
typedef struct dcll_dense_desc {
    int32_t in_features, out_features;  /* i2h.weight is (out, in)                                       */
    int32_t target;                     /* i2o is Linear(out, target)                                    */
    int32_t tau_is_tensor;              /* 0: 1 float each; 1: (in_features) floats :119-129             */
    int32_t refractory;
    float alpharp, wrp;
} dcll_dense_desc;

/*
 * Optional per-call extras of the layer calls (ABI v3).  NULL = all defaults (fp32 weights, pv = sigmoid(v)).
 *
 *   w_q8 / w_scale   BASELINE config 5 ("int8 weights"; no reference code — the build's definition): when w_q8 != NULL the
 *                    conv weight is read as int8 (c_out,c_in,kh,kw) with one fp32 scale per OUTPUT channel, w_scale
 *                    (c_out), and the fp32 `W` argument of the call is ignored (may be NULL).  Every kernel converts a
 *                    weight exactly once, where it would have loaded the fp32 value: w = (float)q * w_scale[co] — ONE
 *                    rounded fp32 multiply — and then runs the pinned fmaf chain on it, so the results are bit-identical to
 *                    the same call on the dequantised fp32 tensor.  The weight-stationary sequence kernels convert in their
 *                    prologue (once per launch); the per-step kernels where they stage the weights (1 byte per weight
 *                    through L2 instead of 4).  Not accepted by dcll_conv_lif_backward (learning updates fp32 weights).
 *   pv_presigmoid    sequence calls only.  != 0: pv_out receives v = pvmem + arp' — what the threshold saw (:498-499),
 *                    max-pooled where the layer pools — INSTEAD of pv = sigmoid(v).  The sigmoid then runs in the consumer:
 *                    dcll_readout_act(..., DCLL_ACT_SIGMOID) applies it to every value it stages (the readout is HBM-bound
 *                    and has the vector slots; the layer kernels share their vector pipe with the fp32 MFMAs), and the pv
 *                    statistics (pv_lowhigh) are counted on sigmoid(v) as before.  sigmoid is monotone, so
 *                    max-pool(sigmoid(v)) == sigmoid(max-pool(v)) up to the last ulp of the (not bit-pinned) sigmoid.
 *                    Not combined with the fused readout (n_ro > 0).
 */
/*
 * Second threshold table of the IQ quantiser (dcll_iq_encode, dcll_conv_lif_sequence_iq; NULL = one table for all samples).
 * The reference quantises one time sample of the whole batch at a time with torch CPU ops (data/utils.py:60-79); torch's
 * float `pow` runs full groups of 16 / 32 batch positions through its vector implementation and the remaining positions
 * (batch size not a multiple of the group, chunk ends of its thread pool) through scalar libm, and the two differ in the
 * last ulp at some cell boundaries — so which cell a boundary value lands in depends on the sample's POSITION in the batch.
 * thr_*_tail are the thresholds of the scalar path, tail_mask[b] != 0 marks the samples the reference would have sent
 * through it (the host derives all three from torch itself, data/utils.py IQEncoder): device cells == reference cells for
 * every batch size.
 */
typedef struct dcll_iq_tail {
    const float *thr_i_tail;    /* (w-1) */
    const float *thr_q_tail;    /* (h-1) */
    const uint8_t *tail_mask;   /* (B)   */
} dcll_iq_tail;

typedef struct dcll_layer_opts {
    const int8_t *w_q8;         /* int8 conv weights (c_out,c_in,kh,kw), or NULL: the call's fp32 W is used */
    const float *w_scale;       /* (c_out) fp32, required with w_q8                                          */
    int32_t pv_presigmoid;      /* sequence calls: pv_out = v (pooled by max) instead of sigmoid(v)          */
    int32_t reserved;           /* must be 0                                                                 */
} dcll_layer_opts;

int dcll_version(void);                 /* DCLL_ABI_VERSION of the loaded library                        */
const char *dcll_last_error(void);      /* thread-local message of the last failing call ("" if none)    */

/*
 * Which kernels did a call dispatch?  (ABI 5; diagnostics, host-side only.)  The reference has one code path per layer
 * (F.conv2d, dcll/pytorch_libdcll.py:417,495); this library picks a kernel by geometry and batch, and a parity test is
 * only worth its name if it ran the kernel it claims to cover.  dcll_kernel_trace(1) clears the calling thread's log
 * and starts recording one line per kernel launch of that thread's calls; dcll_kernel_trace(0) stops.
 * dcll_kernel_trace_read copies the log (newline-separated kernel names, NUL-terminated, truncated to cap) and returns
 * the bytes needed for all of it; the log itself stops growing at 64 KiB (last line "...").
 */
int dcll_kernel_trace(int32_t enable);
int64_t dcll_kernel_trace_read(char *buf, int64_t cap);

/* Output spatial sizes of the conv and of the pooled map (get_output_shape :368-375, :593-597). */
int dcll_conv_out_shape(const dcll_conv_desc *d, int32_t *conv_h, int32_t *conv_w, int32_t *pool_h, int32_t *pool_w);

/*
 * One timestep of Conv2dDCLLlayer.forward — exact drop-in for dcll/pytorch_libdcll.py:599-608.
 *   x        (B,c_in,h,w)   input spikes (any fp32 values are accepted, the reference does not check)
 *   W,b      i2h.weight (c_out,c_in/groups,kh,kw), i2h.bias (c_out) — b may be NULL (bias=False, :323-326: the chains start at 0)
 *   alpha,tau_m,alphas,tau_s   i2h.alpha, i2h.tau_m__dt, i2h.alphas, i2h.tau_s__dt
 *   eps0,eps1 (B,c_in,h,w)  neuron state, updated IN PLACE          arp (B,c_out,ch,cw) idem (may be NULL if !refractory)
 *   i2o_W (target, c_out*ph*pw), i2o_b (target)        out_W,out_b: output_ layer, NULL unless output_layer
 *   out_s  (B,c_out,ph,pw) pooled spikes          out_p (B,target) local readout      out_o (B,target) or NULL
 *   out_pv (B,c_out,ph,pw) pooled sigmoid         out_v (B,c_out,ch,cw) = pvmem (+arp) before pooling, may be NULL
 *   scratch 2*B*c_out*ch*cw floats, required iff pooling != 1 (un-pooled spikes and sigmoid), else may be NULL
 *   i2o_W / out_p may be NULL to skip the local readout (then the call is ContinuousConv2D.forward + pool).
 *   opts     NULL, or int8 weights (dcll_layer_opts; pv_presigmoid must be 0 here: the drop-in returns pv).
 */
int dcll_conv_lif_step(const dcll_conv_desc *d, const float *x, const float *W, const float *b,
                       const float *alpha, const float *tau_m, const float *alphas, const float *tau_s,
                       float *eps0, float *eps1, float *arp,
                       const float *i2o_W, const float *i2o_b, const float *out_W, const float *out_b,
                       float *out_s, float *out_p, float *out_o, float *out_pv, float *out_v, float *scratch,
                       const dcll_layer_opts *opts, int32_t B, void *stream);

/*
 * Backward of one Conv2dDCLLlayer step for local learning — what loss.backward() reaches in DCLLBase.train_dcll
 * (dcll/pytorch_libdcll.py:690-704).  Call after dcll_conv_lif_step of the same step, before the next one.
 *   eps1 (B,c_in,h,w) state after the step ; v = out_v of the step ; pv_pooled = out_pv of the step.  v may be NULL for a
 *   layer WITHOUT pooling and target <= 32 whose pv_pooled is given: sigmoid'(v) = pv (1 - pv) is then taken from the stored pv —
 *   the bits the kernel would recompute from v (what torch's sigmoid backward does, too) — and the step need not write out_v
 *   incoming gradients (each may be NULL): g_p (B,target) of pvoutput, g_o (B,target) of the output_ logits,
 *   g_pv (B,c_out,ph,pw) of pv, g_v (B,c_out,ch,cw) of pvmem
 *   results: dW (c_out,c_in,kh,kw), db (c_out) [may be NULL]; d_outW (target, c_out*ph*pw), d_outb (target) iff g_o
 *   scratch: scratch_floats >= B*c_out*ch*cw + k*c_out*(c_in*kh*kw + 1) floats with k >= 1 batch chunks for the weight
 *   gradient's partial sums (more chunks = more parallelism; up to 256 are used).  An output_ layer with target <= 32 whose
 *   K = c_out*ph*pw is not a multiple of 32 sums its gradient over min(B, 16, m) batch chunks of target*(K+1) floats, m = what
 *   fits into the partial-sum area (closed form: the area is reused) or behind the rows in use (dcll_conv_lif_backward_open):
 *   with room for min(B, 16) chunks in either place both forms give the same bits.
 *   i2o is frozen, neuron state detached, output_ sees pv.detach() (:570,:504,:606).
 */
int dcll_conv_lif_backward(const dcll_conv_desc *d, const float *eps1, const float *v, const float *pv_pooled,
                           const float *g_p, const float *g_o, const float *g_pv, const float *g_v,
                           const float *i2o_W, float *dW, float *db, float *d_outW, float *d_outb,
                           float *scratch, int64_t scratch_floats, int32_t B, void *stream);

/*
 * The rest of a local-learning step (DCLLBase.train_dcll :690-718) around dcll_conv_lif_step / dcll_conv_lif_backward, so
 * that the per-timestep loop needs no torch op:
 *
 * dcll_local_loss_grad — gradient and value of the local losses with mean reduction (what loss.backward() hands to the
 *   readouts): loss = crit(p, target) [+ crit(o, target) when o != NULL], g_p = d loss / d p, g_o = d loss / d o.
 *   p, o, target, g_p, g_o (B,N) fp32; loss: 1 float, may be NULL; clout (B) int32, may be NULL: the per-sample argmax of
 *   o (of p when o == NULL) that DCLLClassification.forward records (:724-728).  kind: DCLL_LOSS_SMOOTH_L1 (torch.nn.SmoothL1Loss,
 *   beta 1 — train.py's default --loss_type) or DCLL_LOSS_MSE (torch.nn.MSELoss).
 *
 * dcll_adam_step — torch.optim.Adam's update (amsgrad False; L2 weight decay added to the gradient; bias correction by
 *   `step`) over up to DCLL_ADAM_MAX_TENSORS parameter tensors — of one or of several optimizers — in one launch.  train.py
 *   (:164-168) builds the optimizers with betas (0, beta) and weight_decay 10; exp_avg / exp_avg_sq are the optimizer's
 *   own state tensors, so optimizer.state_dict() stays what torch would have produced.  `tensors` is a HOST array.
 *
 * dcll_cells_to_planes — iq2spiketrain's dense spike planes (data/utils.py:81-82) from cell indices on the device:
 *   planes (n_samples, hw) fp32 = one-hot of cells (n_samples) int32; hw % 4 == 0.
 */
enum { DCLL_LOSS_SMOOTH_L1 = 0, DCLL_LOSS_MSE = 1 };
int dcll_local_loss_grad(const float *p, const float *o, const float *target, float *g_p, float *g_o, float *loss,
                         int32_t *clout, int32_t B, int32_t N, int32_t kind, void *stream);

#define DCLL_ADAM_MAX_TENSORS 8
typedef struct dcll_adam_tensor {
    float *param;               /* updated in place                                                       */
    const float *grad;
    float *exp_avg, *exp_avg_sq; /* optimizer state, updated in place                                     */
    int64_t n;                  /* elements                                                               */
    int64_t step;               /* 1-based count of this update (bias correction)                         */
    float lr, weight_decay, beta1, beta2, eps;  /* of the tensor's param group (the slices' optimizers differ: */
                                /* optimizer2 of the output layer runs torch's default betas, :637-638)  */
} dcll_adam_tensor;
int dcll_adam_step(const dcll_adam_tensor *tensors, int32_t n_tensors, void *stream);
/* The same update with the quantities that change from step to step read from DEVICE memory at execution time: dyn
 * (n_tensors x 3 floats) = per tensor (lr, 1 / (1 - beta1^step), 1 / sqrt(1 - beta2^step)); `lr` and `step` of the structs
 * are ignored.  This is the form a captured hipGraph of a learning timestep replays (the host refreshes `dyn` with a
 * stream-ordered copy before each replay). */
int dcll_adam_step_dyn(const dcll_adam_tensor *tensors, int32_t n_tensors, const float *dyn, void *stream);

/*
 * The end of a learning timestep in one launch (ABI 5).  dcll_conv_lif_backward_open is dcll_conv_lif_backward with the last
 * step of the weight gradient left open: the partial rows stay in `scratch` — *part points at *nchunk rows of
 * c_out * (c_in*kh*kw + 1) floats, valid until the next call that uses this scratch — and dW / db are not written.
 * dcll_grad_reduce_adam then finishes up to DCLL_REDUCE_MAX_LAYERS layers at once: the rows are added in dcll_conv_lif_backward's
 * own fixed order (bit-identical gradients), written to dW / db, and the Adam update of `tensors[adam_w]` / `tensors[adam_b]`
 * (index, or -1: no update) is applied by the thread that holds the finished gradient element; the tensors no layer refers
 * to get dcll_adam_step's elementwise update in the same launch.  dyn: NULL, or as dcll_adam_step_dyn (n_tensors x 3).
 * Replaces, per timestep of train_dcll (:690-718), one reduce launch per layer + the optimizer launch.
 */
#define DCLL_REDUCE_MAX_LAYERS 4
typedef struct dcll_grad_parts {
    const float *part;          /* nchunk partial rows (dcll_conv_lif_backward_open)                       */
    float *dW, *db;             /* (c_out, c_in*kh*kw) and (c_out): the reduced gradients; db may be NULL  */
    int64_t rowlen;             /* c_in*kh*kw + 1                                                          */
    int32_t nchunk, c_out;
    int32_t adam_w, adam_b;     /* entries of `tensors` to update with dW / db, or -1                      */
} dcll_grad_parts;
int dcll_conv_lif_backward_open(const dcll_conv_desc *d, const float *eps1, const float *v, const float *pv_pooled,
                                const float *g_p, const float *g_o, const float *g_pv, const float *g_v,
                                const float *i2o_W, float *d_outW, float *d_outb, float *scratch, int64_t scratch_floats,
                                int32_t B, const float **part, int32_t *nchunk, void *stream);
int dcll_grad_reduce_adam(const dcll_grad_parts *layers, int32_t n_layers, const dcll_adam_tensor *tensors,
                          int32_t n_tensors, const float *dyn, void *stream);

/*
 * dcll_conv_lif_backward_open for SEVERAL layers — the slices of one learning timestep — in one call (ABI 6): where all of
 * them are layers without pooling with the same readout-width class their dv launches (max-pool routing / i2o^T / sigmoid')
 * run as ONE launch, then the weight / output_ gradient kernels layer by layer.  Each field is the argument of that name of
 * dcll_conv_lif_backward_open; `part` / `nchunk` are written (its outputs).  Per item the results are those of the single
 * call, bit for bit.  1 <= n <= 8, reserved must be 0.
 */
typedef struct dcll_bwd_item {
    const dcll_conv_desc *d;
    const float *eps1, *v, *pv_pooled, *g_p, *g_o, *g_pv, *g_v, *i2o_W;
    float *d_outW, *d_outb, *scratch;
    int64_t scratch_floats;
    int32_t B, reserved;
    const float *part;          /* out */
    int32_t nchunk, reserved2;  /* out; 0 */
} dcll_bwd_item;
int dcll_conv_lif_backward_open_multi(dcll_bwd_item *items, int32_t n, void *stream);

int dcll_cells_to_planes(const int32_t *cells, float *planes, int64_t n_samples, int32_t hw, void *stream);

/* One timestep of DenseDCLLlayer.forward — drop-in for dcll/pytorch_libdcll.py:250-255 (dropout = identity). */
int dcll_dense_lif_step(const dcll_dense_desc *d, const float *x, const float *W, const float *b,
                        const float *alpha, const float *tau_m, const float *alphas, const float *tau_s,
                        float *eps0, float *eps1, float *arp, const float *i2o_W, const float *i2o_b,
                        float *out_s, float *out_p, float *out_pv, float *out_v, int32_t B, void *stream);

/*
 * Backward of one DenseDCLLlayer step for local learning (ABI 7) — DCLLBase.train_dcll (:690-718) is layer-agnostic: it builds
 * the optimizer from dclllayer.i2h.parameters() (:634-635) and DenseDCLLlayer.forward keeps pvoutput in the graph (:250-255).
 *   dv = (g_p . i2o_W + g_pv) * pv * (1 - pv) + g_v ;  dW (out,in) = dv^T . eps1 ;  db (out) = sum_b dv
 * eps1 (B,in): the layer's eps1 state AFTER the step; pv (B,out) = sigmoid(v) as the step returned it; g_p (B,target), g_pv,
 * g_v (B,out): gradients of the loss w.r.t. pvoutput / pv / pvmem, each may be NULL (g_p needs i2o_W (target,out)).
 * scratch: scratch_floats >= B*out + k*out*(in+1), k >= 1 batch chunks (up to 64 are used).  The _open form leaves the last
 * reduction to dcll_grad_reduce_adam (rowlen = in + 1, c_out = out): *part / *nchunk as dcll_conv_lif_backward_open.
 * Sums run in a fixed order (an fp32-MFMA GEMM over sample pairs per chunk, chunks ascending): deterministic, and the closed
 * form is the open form + dcll_grad_reduce_adam without tensors — the same bits.
 */
int dcll_dense_lif_backward(const dcll_dense_desc *d, const float *eps1, const float *pv, const float *g_p,
                            const float *g_pv, const float *g_v, const float *i2o_W, float *dW, float *db,
                            float *scratch, int64_t scratch_floats, int32_t B, void *stream);
int dcll_dense_lif_backward_open(const dcll_dense_desc *d, const float *eps1, const float *pv, const float *g_p,
                                 const float *g_pv, const float *g_v, const float *i2o_W, float *scratch,
                                 int64_t scratch_floats, int32_t B, const float **part, int32_t *nchunk, void *stream);

/*
 * All T timesteps of a DenseDCLLlayer in one call — `for t: layer.forward(x[t])` (:250-255) — the dense twin of
 * dcll_conv_lif_sequence.  x (T,B,in) fp32; neuron state in/out as in dcll_dense_lif_step (read at t = 0, written after
 * t = T-1); out_s, out_pv, out_v (T,B,out), out_p (T,B,target), each may be NULL (out_p needs out_pv).
 * in_features <= 1024 and out_features <= 128: ONE launch with the state on chip for all T (32 samples per workgroup, eps0
 * in registers, eps1 in LDS, W streamed from L2); larger layers advance step by step inside the call with the state in
 * HBM.  The local readout runs once over all T x B rows.  Same pinned chain as the per-step call: bit-identical v / s / state.
 */
int dcll_dense_lif_sequence(const dcll_dense_desc *d, const float *x, const float *W, const float *b,
                            const float *alpha, const float *tau_m, const float *alphas, const float *tau_s,
                            float *eps0, float *eps1, float *arp, const float *i2o_W, const float *i2o_b,
                            float *out_s, float *out_p, float *out_pv, float *out_v, int32_t T, int32_t B, void *stream);

/*
 * Whole-sequence fast path behind ConvNetwork.test (networks/__init__.py:182-185) for one layer: all T timesteps
 * of dcll/pytorch_libdcll.py:485-509 / :407-426 in ONE launch with the neuron state held on-chip.
 * Supported geometry (else DCLL_ERR_UNSUPPORTED): c_in==32, c_out==32, 7x7, pad 3, pool 1, time constants constant
 * over (h,w) per input channel (what randomize_tau produces, :391-405), on the 16x16 plane of the reference's scripts
 * (one sample per workgroup) or on any plane with h % 8 == 0 and w % 32 == 0 — e.g. the 128x128 default of
 * test_radio_ml.py:52 — (one workgroup per 8x32 tile; no fused readout there).
 * Also served: the layers of networks/radio_ml_conv_ref.yaml — c_in==64 (first layer: dcll_conv_lif_sequence_cells, c_in==1),
 * c_out==64, kernel (1,3), padding (0,1), pooling (1,2), w a power of two <= 256, h*w % 32 == 0 (k_lif_seq_w3; the first
 * layer, k_lif_seq_w3f: h*w % 128 == 0, state and output pointers 8-byte aligned).  For this
 * POOLING geometry the outputs are the pooled maps: spk_out (T,B,64,h*(w/2)/32) [needs h*w % 64 == 0], pv_out
 * (T,B,64,h,w/2); v_out stays un-pooled (T,B,64,h,w); state_scratch is not used.
 *   spk_in   (T,B,c_in,h*w/32) uint32   packed input spikes: bit (y*w+x)%32 of word (y*w+x)/32
 *   tau4     (4,c_in) fp32              rows: alpha, tau_m, alphas, tau_s per input channel
 *   eps0,eps1,arp                       neuron state in/out as in dcll_conv_lif_step (read at t=0, written after t=T-1)
 *   spk_out  (T,B,c_out,h*w/32) uint32  packed output spikes (same layout, feeds the next layer); may be NULL
 *   pv_out   (T,B,c_out,h,w) fp32       sigmoid(v) for the readout GEMM; may be NULL
 *   v_out    (T,B,c_out,h,w) fp32       debugging / parity only; may be NULL
 *   fused local readout (n_ro = 24 or 48; 0 = none): ro_out (T,B,n_ro) = flatten(pv) . Wro^T + ro_b, i.e. i2o and,
 *   stacked behind it on the output layer, output_ (:602-606), computed in the epilogue so that pv never leaves the
 *   chip.  ro_Wp = the (n_ro, c_out*h*w) weight matrix re-laid-out by dcll_permute_readout; ro_b (n_ro).
 *   state_scratch  2*B*c_in*h*w floats, REQUIRED on planes other than 16x16 (may be NULL on 16x16): the tiled kernels
 *                  read a tile's initial traces plus a halo owned by neighbouring tiles, so the launch first snapshots
 *                  eps0 / eps1 there (stream-ordered copies), reads only the snapshot and writes only eps0 / eps1.
 *   pv_lowhigh     NULL = off.  Else uint64 [n][2], n = dcll_pv_lowhigh_steps(iter0, T): for every step t of this call
 *                  with (iter0 + t + 1) % 20 == 0 — DCLLBase.forward's histogram steps, :658-661 — the number of pv
 *                  values in the first and in the last of the 19 bins of np.linspace(0, 1, 20) (what write_stats
 *                  reports, :678-688).  iter0 = the slice's iteration count before the call.  Needs pv_out.
 *   opts           NULL, or int8 weights / pv_presigmoid (dcll_layer_opts above).
 */
int dcll_conv_lif_sequence(const dcll_conv_desc *d, const uint32_t *spk_in, const float *W, const float *b,
                           const float *tau4, float *eps0, float *eps1, float *arp,
                           uint32_t *spk_out, float *pv_out, float *v_out,
                           const float *ro_Wp, const float *ro_b, float *ro_out, int32_t n_ro,
                           float *state_scratch, uint64_t *pv_lowhigh, int32_t iter0,
                           const dcll_layer_opts *opts, int32_t T, int32_t B, void *stream);

/* Re-lay-out a readout matrix Wt (N, 32*16*16) [n][co][pix] for the fused epilogue of dcll_conv_lif_sequence:
 * Wp[me][wq][n][lane][rr] with co = rr + 8*wq + 4*(lane>>5), pix = 32*me + (lane&31).  Wp has N*8192 floats. */
int dcll_permute_readout(const float *Wt, float *Wp, int32_t N, void *stream);

/*
 * First-layer sequence kernel (c_in==1): the input is exactly one spike per sample per step (iq2spiketrain,
 * data/utils.py:43-87), given as its cell index q*w+i.  Geometry: c_in==1, c_out<=32, 7x7, pad 3, pool 1, plane 16x16
 * or h % 8 == 0 and w % 32 == 0.
 *   cells (T,B) int32 ; tau4 (4,1) ; other arguments as dcll_conv_lif_sequence (state_scratch: 2*B*h*w floats).
 */
int dcll_conv_lif_sequence_cells(const dcll_conv_desc *d, const int32_t *cells, const float *W, const float *b,
                                 const float *tau4, float *eps0, float *eps1, float *arp,
                                 uint32_t *spk_out, float *pv_out, float *v_out,
                                 float *state_scratch, uint64_t *pv_lowhigh, int32_t iter0,
                                 const dcll_layer_opts *opts, int32_t T, int32_t B, void *stream);

/*
 * Same kernel with iq2spiketrain's quantisation (data/utils.py:60-82) fused in: the input is the raw IQ window
 * iq (B,2,L) fp32, samples t0..t0+T-1, quantised with the thresholds of dcll_iq_encode (thr_i: w-1 floats, thr_q:
 * h-1 floats).  T <= 4096.
 */
int dcll_conv_lif_sequence_iq(const dcll_conv_desc *d, const float *iq, const float *thr_i, const float *thr_q,
                              const dcll_iq_tail *tail, int32_t L, int32_t t0, const float *W, const float *b, const float *tau4,
                              float *eps0, float *eps1, float *arp, uint32_t *spk_out, float *pv_out, float *v_out,
                              float *state_scratch, uint64_t *pv_lowhigh, int32_t iter0,
                              const dcll_layer_opts *opts, int32_t T, int32_t B, void *stream);

/*
 * The pv statistics of DCLLBase.forward (:658-661) as a call of its own (the per-step path uses it with T = 1):
 * pv (T, per_step) fp32 = the pv outputs of T consecutive steps, per_step = B*c_out*ph*pw values each; counts as
 * pv_lowhigh above, uint64 [dcll_pv_lowhigh_steps(iter0, T)][2].  Steps that are not histogram steps are not read.
 */
int dcll_pv_lowhigh(const float *pv, int64_t per_step, int32_t T, int32_t iter0, uint64_t *counts, void *stream);
/* The same counters for a buffer written with pv_presigmoid (act = DCLL_ACT_SIGMOID: sigmoid(v) is what is counted). */
enum { DCLL_ACT_NONE = 0, DCLL_ACT_SIGMOID = 1 };
int dcll_pv_lowhigh_act(const float *pv, int64_t per_step, int32_t T, int32_t iter0, uint64_t *counts, int32_t act,
                        void *stream);
int32_t dcll_pv_lowhigh_steps(int32_t iter0, int32_t T);    /* (iter0 + T) / 20 - iter0 / 20 */

/*
 * Local readout for many rows at once: out[r, n] = sum_k pv[r,k]*Wt[n,k] + bias[n]   (i2o / output_, :602-606),
 * fp32 MFMA.  rows = T*B, K = c_out*ph*pw, N = rows of Wt (e.g. i2o and output_ stacked: 48).
 */
int dcll_readout(const float *pv, const float *Wt, const float *bias, float *out,
                 int64_t rows, int32_t K, int32_t N, void *stream);
/*
 * The readout of the whole-sequence path (rows = T x a chunk of the batch), optionally on a pv_presigmoid buffer:
 *     out[r, n] = sum_k act(pv[r,k])*Wt[n,k] + bias[n]
 * act = DCLL_ACT_SIGMOID applies the layer kernels' sigmoid (v_exp_f32 + v_rcp_f32) to every staged value: the logits equal
 * those of the pv = sigmoid(v) form up to the readout's summation order.  Always the LDS-staged 16x16x4 kernel: whole for
 * K < 65536, in 8 ... 64 K-slices of >= 8192 columns (partials in caller scratch, summed in slice order) for longer rows —
 * chosen by K alone, so
 * a row's logits do not depend on the row count, i.e. on how the caller chunks a batch (dcll_readout / dcll_readout_splitk
 * choose by row count: they serve the per-step calls).  Needs K % 32 == 0 (K % 256 == 0 when split), K < 2^22 (32-bit
 * offsets inside a workgroup's 128 rows), N <= 64, 16-byte aligned pv / Wt — else DCLL_ERR_UNSUPPORTED (the caller keeps pv = sigmoid(v) and dcll_readout for such shapes);
 * scratch_floats >= dcll_readout_act_scratch(rows, K, N) (0: scratch may be NULL).
 */
int64_t dcll_readout_act_scratch(int64_t rows, int32_t K, int32_t N);
int dcll_readout_act(const float *pv, const float *Wt, const float *bias, float *out, float *scratch,
                     int64_t scratch_floats, int64_t rows, int32_t K, int32_t N, int32_t act, void *stream);

/*
 * The same readout with the kernel form chosen by the caller:
 *   DCLL_READOUT_AUTO        what dcll_readout does;
 *   DCLL_READOUT_CORESIDENT  the LDS-free, <= 64-VGPR form of the 16x16x4 kernel (K % 64 == 0, K <= 16384, N <= 48,
 *                            16-byte aligned rows; other shapes fall back to AUTO): a workgroup of it fits on a CU beside a
 *                            resident sequence-kernel workgroup, so a readout launched on a second stream runs in the gaps
 *                            of the next layer's kernel instead of after it;
 *   DCLL_READOUT_LDS         the LDS-staged 32x32x2 kernels only (measurements);
 *   DCLL_READOUT_T16         the LDS-staged 16x16x4 kernel (K % 32 == 0, N <= 64, aligned rows; else AUTO).
 * Logits of different modes differ by summation order only (within the 1e-4 contract).
 */
enum { DCLL_READOUT_AUTO = 0, DCLL_READOUT_CORESIDENT = 1, DCLL_READOUT_LDS = 2, DCLL_READOUT_T16 = 3 };
int dcll_readout_mode(const float *pv, const float *Wt, const float *bias, float *out,
                      int64_t rows, int32_t K, int32_t N, int32_t mode, void *stream);

/*
 * The same readout for FEW rows (per-step calls: rows = batch): K is split into slices over the workgroups, the partial
 * tiles go to caller-provided scratch and are added in slice order (deterministic).  Served, with N <= 64 and 16-byte
 * aligned pv / Wt: K >= 65536, K % 4096 == 0 (large planes, K = c_out*128*128: slices of 4096 for rows <= 2048, else 8
 * slices — any row count), or rows <= 2048 with
 * 2048 <= K < 65536, K % 256 == 0 (the 16x16 plane, K = 8192: slices of 256 — of 128 for rows <= 512 and N <= 32);
 * scratch_floats >= dcll_readout_splitk_scratch(rows, K, N) (0 = this shape is not supported, use dcll_readout).
 */
int64_t dcll_readout_splitk_scratch(int64_t rows, int32_t K, int32_t N);
int dcll_readout_splitk(const float *pv, const float *Wt, const float *bias, float *out, float *scratch,
                        int64_t scratch_floats, int64_t rows, int32_t K, int32_t N, void *stream);

/*
 * ABI 4 — the readouts of ONE layer step and everything behind them in two launches (per-step calls: rows = batch):
 *   Conv2dDCLLlayer.forward :602-606     p = i2o(flatten(pv)) and, on the output layer, o = output_(flatten(pv)): ONE
 *                                        split-K pass over pv against the N1 + N2 STACKED rows Wt / bias (N2 = 0 or N1);
 *   DCLLClassification.forward :724-728  clout (rows) int32 = argmax of o (of p when N2 == 0), first maximum; NULL = off;
 *   DCLLBase.train_dcll :692-704         target != NULL: g_p, g_o = gradients of the mean local losses of kind `kind`
 *                                        (dcll_local_loss_grad without the loss value; NULL = inference step).
 * p (rows, N1) and o (rows, N2) come out as separate contiguous arrays; per column bit-identical to a dcll_readout_splitk call
 * that splits K into slices of the same width (the slice width follows rows and N1 + N2: for rows <= 512 a call with
 * N1 <= 32 < N1 + N2 here uses 256-column slices where a call on the N1 columns alone would use 128).
 * Shapes: rows <= 2048, 2048 <= K < 65536, K % 256 == 0, N1 + N2 <= 64, 16-byte aligned pv / Wt — else
 * DCLL_ERR_UNSUPPORTED (callers fall back to dcll_readout + dcll_argmax_vote / dcll_local_loss_grad);
 * scratch_floats >= dcll_step_readouts_scratch(rows, K, N1, N2).
 */
int64_t dcll_step_readouts_scratch(int64_t rows, int32_t K, int32_t N1, int32_t N2);
int dcll_step_readouts(const float *pv, const float *Wt, const float *bias, float *scratch, int64_t scratch_floats,
                       int64_t rows, int32_t K, int32_t N1, int32_t N2, float *p, float *o, int32_t *clout,
                       const float *target, float *g_p, float *g_o, int32_t kind, void *stream);

/*
 * The readout tails of SEVERAL layer steps — the slices of one network timestep (ConvNetwork.test / .learn,
 * networks/__init__.py:175-185: slice l+1 consumes slice l's spikes, nobody's readouts) — in TWO launches instead of two per
 * layer (ABI 6): one split-K pass over all items' pv, one finishing launch.  Each field is the argument of that name of
 * dcll_step_readouts; per item the results are those of dcll_step_readouts, bit for bit.  1 <= n <= 8; every item must be one
 * dcll_step_readouts serves, with rows >= 1 and a scratch area of its own; either every item has a target (learning step) or
 * none.  reserved must be 0.
 */
typedef struct dcll_step_ro {
    const float *pv, *Wt, *bias;
    float *scratch;
    int64_t scratch_floats, rows;
    int32_t K, N1, N2, kind;
    float *p, *o;
    int32_t *clout;
    const float *target;
    float *g_p, *g_o;
    int64_t reserved;
} dcll_step_ro;
int dcll_step_readouts_multi(const dcll_step_ro *items, int32_t n, void *stream);

/*
 * Per-step argmax + vote (DCLLClassification.forward :724-728, get_predictions_by_vote :44-56):
 *   logits (T,B,N) -> clout (T,B) int32 (first maximum wins, like torch.argmax) and, if vote != NULL,
 *   vote (B) int32 = mode over t in [t_begin,T) of clout, ties broken by first occurrence in time.
 */
int dcll_argmax_vote(const float *logits, int32_t *clout, int32_t *vote, int32_t T, int32_t B, int32_t N,
                     int32_t t_begin, void *stream);

/*
 * The per-class tallies of an evaluated batch (accuracy_by_vote / confusion_matrix, dcll/pytorch_libdcll.py:44-61, :740-749, in
 * the summable form the ranks all-reduce): out (n_layers, n_classes*n_classes + 2) int64 = per layer the confusion matrix
 * [pred][label] flattened, the number of votes equal to their label, the number of votes.  votes: HOST array of n_layers
 * device pointers to (B) int32 votes (dcll_argmax_vote); labels (B) int64.  Every element of `out` is written.
 */
int dcll_vote_tallies(const int32_t *const *votes, int32_t n_layers, const int64_t *labels, int64_t *out, int32_t B,
                      int32_t n_classes, void *stream);

/*
 * iq2spiketrain on device (data/utils.py:60-82) as a threshold search: cell = #{j : x >= thr[j]} for the
 * monotone map x -> int(clamp(gamma(x),0,1)*(R-1)); thr_i (w-1) and thr_q (h-1) are produced on the host from
 * the host encoder so that the result is bit-identical to it.
 *   iq (B,2,L) fp32 ; cells (T,B) int32 = q*w + i for samples t0..t0+T-1
 */
int dcll_iq_encode(const float *iq, const float *thr_i, const float *thr_q, const dcll_iq_tail *tail, int32_t *cells,
                   int32_t B, int32_t L, int32_t t0, int32_t T, int32_t w, int32_t h, void *stream);

/* Unpack (T*B, C, HW/32) packed spikes to fp32 (T*B, C, HW) and back — glue for the tensor-level API. */
int dcll_unpack_spikes(const uint32_t *packed, float *dense, int64_t nwords, void *stream);
int dcll_pack_spikes(const float *dense, uint32_t *packed, int64_t nwords, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DCLL_HIP_H */
