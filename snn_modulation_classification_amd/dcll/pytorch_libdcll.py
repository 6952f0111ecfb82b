"""DCLL layers and slices with the reference's public surface, computed by libdcll_hip.so on an MI355X.

Mirrors the names, constructor keywords, attributes, state-dict keys, return tuples and exception types of the
reference module dcll/pytorch_libdcll.py so `networks.ConvNetwork`, `train.py` and `test_radio_ml.py` keep working:

    ContinuousConv2D / ContinuousRelativeRefractoryConv2D     reference :296-429 / :432-509
    Conv2dDCLLlayer                                           reference :512-612
    CLLDenseModule / CLLDenseRRPModule / DenseDCLLlayer       reference :72-148 / :151-195 / :198-266
    DCLLBase / DCLLClassification                             reference :615-718 / :721-749
    get_predictions_by_vote / accuracy_by_vote                reference :44-61

What is different by design: `.forward` does not run torch ops — it hands device pointers to the HIP kernels
through the C ABI (include/dcll_hip.h).  There is no CPU path: a tensor that is not on a GPU raises DCLLHipError.
Layers also expose `forward_sequence(...)`, the whole-T fast path (state on chip for all T steps).
Local learning (`train_dcll`, reference :690-718) runs the same forward kernels inside an autograd node whose
backward is dcll_conv_lif_backward; torch supplies only the loss module and the optimizer.
"""
import logging
import math
import os
from collections import Counter, namedtuple

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

from .. import ops
from .._lib import DenseDesc

logger = logging.getLogger(__name__)

device = 'cuda'      # same module-level switch as the reference (:34); 'cuda' is the MI355X under ROCm
_UNSET = object()


# ---------------------------------------------------------------------------------------------------------------
# vote helpers (host side, reference :44-61)
# ---------------------------------------------------------------------------------------------------------------
def _mode_first_seen(row):
    """Most common entry; among equally common ones the first that appears (Counter.most_common(1))."""
    return Counter(row).most_common(1)[0][0]


def _clout_to_numpy(pvoutput):
    """clout as a (T,B) integer array; entries may be numpy arrays or (device) tensors appended without a sync."""
    if len(pvoutput) and all(isinstance(c, torch.Tensor) for c in pvoutput):
        return torch.stack(list(pvoutput)).cpu().numpy()
    return np.asarray([c.cpu().numpy() if isinstance(c, torch.Tensor) else c for c in pvoutput])


def _modes_first_seen(rows):
    """_mode_first_seen of every row of a (B,T) integer array, without a Python loop over B (the reference's loop is
    0.3 s per call at B = 4096): per row and value the count and the first position; the winner has the largest count,
    among equal counts the smallest first position."""
    rows = np.asarray(rows)
    B, T = rows.shape
    lo = int(rows.min()) if rows.size else 0
    V = (int(rows.max()) - lo + 1) if rows.size else 0
    if B == 0 or T == 0 or rows.dtype.kind not in 'iu' or V > 4096:       # (class indices: a few dozen values)
        return np.array([_mode_first_seen(r) for r in rows], dtype=np.float64)
    vals = np.arange(lo, lo + V)
    flat = np.arange(B, dtype=np.int64)[:, None] * V + (rows.astype(np.int64) - lo)
    counts = np.bincount(flat.ravel(), minlength=B * V).reshape(B, V)
    first = np.full(B * V, T, dtype=np.int64)
    # positions written in descending order: for a repeated index the last assignment (the smallest position) stays
    first[flat[:, ::-1].ravel()] = np.broadcast_to(np.arange(T - 1, -1, -1, dtype=np.int64), (B, T)).ravel()
    key = counts * (T + 1) + (T - first.reshape(B, V))
    return vals[key.argmax(axis=1)]


def get_predictions_by_vote(pvoutput, labels):
    """pvoutput: T arrays of per-sample argmax; labels: (T,B,C) one-hot tensor.  -> (pred (B), label (B))."""
    votes = _clout_to_numpy(pvoutput).T
    pred = np.asarray(_modes_first_seen(votes), dtype=np.float64)
    return pred, _label_votes(labels)


def _label_votes(labels):
    """Mode over T of the labels' argmax ((T,B,C) one-hot tensor) -> (B) float64."""
    if isinstance(labels, torch.Tensor) and labels.dim() == 3 and labels.shape[0] > 0 and labels.stride(0) == 0:
        # one label row per sample expanded over T (iq2spiketrain repeats the labels): the mode over T is that row's
        labv = labels[0].detach().cpu().numpy().argmax(axis=1).astype(np.float64)
    else:
        lab = labels.detach().cpu().numpy().argmax(axis=2).T
        labv = np.asarray(_modes_first_seen(lab), dtype=np.float64)
    return labv


def accuracy_by_vote(pvoutput, labels):
    pred, labv = get_predictions_by_vote(pvoutput, labels)
    return float(np.mean(pred == labv))


# ---------------------------------------------------------------------------------------------------------------
# conv LIF dynamics
# ---------------------------------------------------------------------------------------------------------------
def _as_pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _is_sigmoid(act):
    """The fused kernels compute pv = sigmoid(v) (and the backward its derivative).  A layer built with another activation —
    or with spiking=False, bias=False, stride / dilation / groups other than 1: the constructor options of the reference
    (:299-313, :75-104) that ConvNetwork never uses — takes the GENERAL step (`_general()`): the HIP step for the traces, the
    convolution in the pinned order (the generic kernels take stride / dilation / groups and a NULL bias), the refractory trace
    and the threshold; the activation, the pooling and the readouts as torch ops on the device; gradients through an autograd
    node whose backward is the HIP weight-gradient kernel (`_ConvLIFVFn` / `_DenseLIFVFn`).  Pinned by fixture G1x."""
    return isinstance(act, nn.Sigmoid)


def _max_pool(x, pooling):
    """nn.MaxPool2d(kernel_size=pooling, stride=pooling, padding=(pooling - 1) // 2) (reference :542-549)."""
    ph, pw = pooling
    if (ph, pw) == (1, 1):
        return x
    return torch.nn.functional.max_pool2d(x, (ph, pw), (ph, pw), ((ph - 1) // 2, (pw - 1) // 2))


class ContinuousConv2D(nn.Module):
    """Two leaky traces on the input + conv + threshold / sigmoid (reference :296-429)."""
    NeuronState = namedtuple('NeuronState', ('eps0', 'eps1'))

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=2, dilation=1, groups=1, bias=True,
                 alpha=.95, alphas=.9, act=nn.Sigmoid(), random_tau=False, spiking=True, **kwargs):
        super().__init__()
        if in_channels % groups != 0:
            raise ValueError('in_channels must be divisible by groups')
        if out_channels % groups != 0:
            raise ValueError('out_channels must be divisible by groups')
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _as_pair(kernel_size)
        self.padding = _as_pair(padding)
        self.stride, self.dilation, self.groups = stride, dilation, groups
        self.random_tau = random_tau
        self.act = act
        self.spiking = spiking
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()
        self._set_tau(torch.Tensor([alpha]), torch.Tensor([alphas]))
        self.wrp = 0.
        self.alpharp = .65
        self.state = None
        self._q8 = None

    # -- int8 weights (BASELINE config 5; the build's definition, quant.py) -----------------------------------------
    def set_int8_weights(self, q, scale):
        """Keep the conv weight's int8 form (q int8 like `weight`, scale fp32 (c_out,)) beside the fp32 Parameter, which
        must hold exactly the dequantised values q * scale (quant.apply_int8_weights writes both).  From now on the
        kernels read the int8 tensor through the C ABI (dcll_layer_opts) — bit-identical results, a quarter of the
        weight bytes.  Dropped as soon as `weight` is modified: an in-place torch op or load_state_dict (seen through the
        tensor's version counter), reset_parameters, or a native learning step — dcll_adam_step writes through the raw
        pointer, which no version counter sees, so the learning paths call Conv2dDCLLlayer.weights_written()."""
        if q is None:
            self._q8 = None
            return
        if tuple(q.shape) != tuple(self.weight.shape) or q.dtype != torch.int8 or scale.shape != (self.out_channels,):
            raise ValueError('int8 weights must be int8 %s with a (%d,) fp32 scale' % (tuple(self.weight.shape),
                                                                                     self.out_channels))
        dev = self.weight.device
        self._q8 = (q.to(dev).contiguous(), scale.to(dev, torch.float32).contiguous(), self.weight._version,
                    self.weight.data_ptr())

    def int8_weights(self):
        """(q, scale) while they still describe `weight`, else None."""
        q8 = self._q8
        if q8 is None or os.environ.get('DCLL_INT8_ABI', '1') == '0':
            return None
        if q8[2] != self.weight._version or q8[3] != self.weight.data_ptr() or q8[0].device != self.weight.device:
            logger.warning('conv weight changed after quantisation: int8 form dropped, kernels read the fp32 weight')
            self._q8 = None
            return None
        return q8[0], q8[1]

    # -- parameters ---------------------------------------------------------------------------------------------
    def _set_tau(self, alpha, alphas):
        """alpha -> tau/dt = 1/(1-alpha) in fp32, as Parameters without grad (reference :349-356, :398-405)."""
        old = [getattr(self, n, None) for n in ('alpha', 'tau_m__dt', 'alphas', 'tau_s__dt')]
        if all(isinstance(t, nn.Parameter) for t in old) and old[0].shape == alpha.shape and \
                old[2].shape == alphas.shape and old[0].device == alpha.device and old[2].device == alphas.device:
            # same geometry (the refractory variant re-draws on every init_state): overwrite in place, so that the device
            # addresses a captured learning step has baked in stay valid
            with torch.no_grad():
                self.alpha.copy_(alpha)
                self.tau_m__dt.copy_(1. / (1 - self.alpha))
                self.alphas.copy_(alphas)
                self.tau_s__dt.copy_(1. / (1 - self.alphas))
            return
        self.alpha = nn.Parameter(alpha, requires_grad=False)
        self.tau_m__dt = nn.Parameter(1. / (1 - self.alpha), requires_grad=False)
        self.alphas = nn.Parameter(alphas, requires_grad=False)
        self.tau_s__dt = nn.Parameter(1. / (1 - self.alphas), requires_grad=False)

    def reset_parameters(self):
        """W ~ U(+-1e-2/(250 sqrt n)), b ~ U(+-1/(250 sqrt n)), n = C_in*kH*kW (reference :359-366)."""
        n = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        stdv = 1. / math.sqrt(n) / 250
        self.weight.data.uniform_(-stdv * 1e-2, stdv * 1e-2)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)
        self._q8 = None                     # (.data writes leave the version counter alone)

    def randomize_tau(self, im_dims, low=[5, 5], high=[10, 35]):
        """tau_m ~ U(5,35) ms, tau_s ~ U(5,10) ms per input channel, stored as (C,H,W) (reference :391-405).
        Draw order (membrane first) and float64 -> float32 conversion follow the reference."""
        taum = np.random.uniform(low[1], high[1], size=[self.in_channels]) * 1e-3
        taus = np.random.uniform(low[0], high[0], size=[self.in_channels]) * 1e-3
        shape = (self.in_channels, im_dims[0], im_dims[1])
        expand = lambda t: np.ascontiguousarray(np.broadcast_to(t[:, None, None], shape))
        dev = self.weight.device
        self._set_tau(torch.Tensor(1 - 1e-3 / expand(taum)).to(dev), torch.Tensor(1 - 1e-3 / expand(taus)).to(dev))

    def get_output_shape(self, im_dims):
        h = (im_dims[0] + 2 * self.padding[0] - self.dilation * (self.kernel_size[0] - 1) - 1) // self.stride + 1
        w = (im_dims[1] + 2 * self.padding[1] - self.dilation * (self.kernel_size[1] - 1) - 1) // self.stride + 1
        return h, w

    # -- state --------------------------------------------------------------------------------------------------
    def _state_tensors(self, shapes, init_values):
        """Neuron state tensors of the given shapes filled with init_values: the existing ones re-filled in place when
        the geometry is unchanged (the per-batch reset of train.py; keeps the addresses a captured learning step has
        baked in), new ones otherwise."""
        dev = self.weight.device
        old = getattr(self, 'state', None)
        if old is not None and len(old) == len(shapes) and all(
                tuple(t.shape) == tuple(sh) and t.device == dev and t.dtype == torch.float32
                for t, sh in zip(old, shapes)):
            for t, val in zip(old, init_values):
                t.fill_(val)
            return list(old)
        return [torch.zeros(tuple(sh), device=dev) + val for sh, val in zip(shapes, init_values)]

    def _alloc_state(self, batch_size, im_dims, init_value):
        shape = (batch_size, self.in_channels, im_dims[0], im_dims[1])
        return self._state_tensors([shape, shape], [init_value, init_value])

    def init_state(self, batch_size, im_dims, init_value=0):
        self.state = self.NeuronState(*self._alloc_state(batch_size, im_dims, init_value))
        if self.random_tau:
            self.randomize_tau(im_dims)
            self.random_tau = False          # the plain variant randomises once (reference :385-387)
        return self.state

    def _check_batch(self, input):
        if self.state is None or not (input.shape[0] == self.state.eps0.shape[0] == self.state.eps1.shape[0]):
            old = -1 if self.state is None else self.state.eps0.shape[0]
            logger.warning("Batch size changed from {} to {} since last iteration. Reallocating states."
                           .format(old, input.shape[0]))
            self.init_state(input.shape[0], input.shape[2:4])

    # -- HIP descriptors ----------------------------------------------------------------------------------------
    def make_desc(self, im_dims, pooling=(1, 1), target=0, output_layer=False):
        return ops.make_conv_desc(self.in_channels, self.out_channels, im_dims, self.kernel_size, self.padding,
                                  pooling, target, output_layer, self.alpha.numel() > 1, self.wrp, self.alpharp,
                                  self.stride, self.dilation, self.groups)

    def tau_per_channel(self):
        """(4, C_in) tensor [alpha, tau_m, alphas, tau_s] if the time constants do not vary over (H,W), else None.
        Cached until one of the four tensors is replaced or modified (the check costs a device sync)."""
        key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in (self.alpha, self.tau_m__dt, self.alphas,
                                                                        self.tau_s__dt))
        cache = getattr(self, '_tau4_cache', None)
        if cache is not None and cache[0] == key:
            return cache[1]
        tau4 = self._tau_per_channel_uncached()
        self._tau4_cache = (key, tau4)
        return tau4

    def _tau_per_channel_uncached(self):
        rows = []
        for t in (self.alpha, self.tau_m__dt, self.alphas, self.tau_s__dt):
            if t.numel() == 1:
                rows.append(t.detach().reshape(1).expand(self.in_channels))
            else:
                flat = t.detach().reshape(self.in_channels, -1)
                if not bool((flat == flat[:, :1]).all()):
                    return None
                rows.append(flat[:, 0])
        return torch.stack(rows).contiguous()

    def _step(self, input, pooling=(1, 1), i2o=None, output_=None, out=None, stacked=None, finish=None, want_v=True,
              defer_ro=False):
        """Run one step through dcll_conv_lif_step; returns (s_pooled, p, o, pv_pooled, v).  `out`: optional dict of
        reusable output buffers; `stacked` / `finish` / `defer_ro`: the fused readout tail of the step (ops.conv_lif_step)."""
        self._check_batch(input)
        desc = self.make_desc(input.shape[2:4], pooling, 0 if i2o is None else i2o.weight.shape[0],
                              output_ is not None)
        st = self.state
        arp = st.arp if len(st) > 2 else None
        with torch.no_grad():
            return ops.conv_lif_step(
                desc, input, self.weight, self._bias_or_zero(), self.alpha, self.tau_m__dt, self.alphas, self.tau_s__dt,
                st.eps0, st.eps1, arp,
                None if i2o is None else i2o.weight, None if i2o is None else i2o.bias,
                None if output_ is None else output_.weight, None if output_ is None else output_.bias, out=out,
                q8=self.int8_weights(), stacked=stacked, finish=finish, want_v=want_v, defer_ro=defer_ro)

    def _general(self):
        """True when the layer was built with an option outside the fused step (see _is_sigmoid)."""
        return not (_is_sigmoid(self.act) and self.spiking and self.bias is not None and self.stride == 1 and
                    self.dilation == 1 and self.groups == 1)

    def _bias_or_zero(self):
        """bias=False: a zero vector — the pinned chain then starts at +0.0, which IS the bias-free convolution (and lets the
        specialised kernels serve the layer)."""
        if self.bias is not None:
            return self.bias
        z = self.__dict__.get('_zero_bias')
        if z is None or z.device != self.weight.device:
            z = self.__dict__['_zero_bias'] = torch.zeros(self.out_channels, device=self.weight.device)
        return z

    def _outputs(self, s, pv, v):
        """(output, pv) of the reference's forward from the HIP step's (spikes, sigmoid(v), v): output_act (:336-339) and
        act (:419 / :500) as the layer was built."""
        return (s if self.spiking else v), (pv if _is_sigmoid(self.act) else self.act(v))

    def forward(self, input):
        """-> (output, pv, pvmem), un-pooled (reference :407-426)."""
        s, _, _, pv, v = self._step(input)
        o, pv = self._outputs(s, pv, v)
        return o, pv, v


class ContinuousRelativeRefractoryConv2D(ContinuousConv2D):
    """Adds the relative-refractory trace on the output (reference :432-509)."""
    NeuronState = namedtuple('NeuronState', ('eps0', 'eps1', 'arp'))

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=2, dilation=1, groups=1, bias=True,
                 alpha=.95, alphas=.9, alpharp=.65, wrp=1, act=nn.Sigmoid(), random_tau=False, **kwargs):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, alpha,
                         alphas, act)
        self.wrp = wrp
        self.alpharp = alpharp
        self.tau_rp__dt = 1. / (1 - self.alpharp)
        self.random_tau = random_tau

    def init_state(self, batch_size, im_dims, init_value=0):
        oh, ow = self.get_output_shape(im_dims)
        shape = (batch_size, self.in_channels, im_dims[0], im_dims[1])
        st = self._state_tensors([shape, shape, (batch_size, self.out_channels, oh, ow)], [init_value, init_value, 0])
        self.state = self.NeuronState(*st)
        if self.random_tau:
            # observable quirk: this variant never clears the flag, so EVERY init_state re-draws the
            # time constants (reference :479-481) — kept so that seeded runs match the reference.
            self.randomize_tau(im_dims)
        return self.state

    def forward(self, input):
        """-> (output spikes, pv, pvmem + arp), un-pooled (reference :485-509)."""
        if not self.spiking:
            raise Exception('Refractory not allowed in non-spiking mode')
        s, _, _, pv, v = self._step(input)
        o, pv = self._outputs(s, pv, v)
        return o, pv, v


class _ConvLIFStepFn(torch.autograd.Function):
    """One Conv2dDCLLlayer step as an autograd node: forward = dcll_conv_lif_step, backward =
    dcll_conv_lif_backward.  Gradients reach i2h.weight / i2h.bias (via pvoutput, pv, pvmem) and output_.weight /
    output_.bias (via the output logits; output_ sees pv.detach(), reference :606); i2o is frozen, spikes and the
    neuron state carry no gradient — exactly the graph the reference builds with eager ops."""

    @staticmethod
    def forward(ctx, layer, x, W, b, out_W, out_b):
        i2h = layer.i2h
        s, p, o, pv, v = i2h._step(x, layer.pooling, layer.i2o, layer.output_ if layer.output_layer else None)
        ctx.layer = layer
        ctx.desc = i2h.make_desc(x.shape[2:4], layer.pooling, layer.i2o.weight.shape[0], layer.output_layer)
        # the state buffers are updated in place by the next step: keep this step's eps1
        ctx.eps1 = i2h.state.eps1.clone()
        ctx.v, ctx.pv = v, pv
        ctx.mark_non_differentiable(s)
        if o is None:
            return s, p, pv, v
        return s, p, pv, v, o

    @staticmethod
    def backward(ctx, gs, gp, gpv, gv, go=None):
        layer = ctx.layer
        dW, db, d_outW, d_outb = ops.conv_lif_backward(ctx.desc, ctx.eps1, ctx.v, ctx.pv, gp, go, gpv, gv,
                                                       layer.i2o.weight, want_out=layer.output_layer and go is not None)
        return None, None, dW, db, d_outW, d_outb


class _ConvLIFVFn(torch.autograd.Function):
    """The conv + neuron part of a GENERAL layer step (_is_sigmoid) as an autograd node: forward = dcll_conv_lif_step without
    readouts -> (spikes, v), un-pooled; backward = dcll_conv_lif_backward fed with dL/dv alone (dW = dv (*) eps1, db = sum dv).
    Activation, pooling and readouts are torch ops on top of v, so whatever the layer's `act` is, torch differentiates it."""

    @staticmethod
    def forward(ctx, i2h, x, W, b):
        s, _, _, _, v = i2h._step(x)
        ctx.i2h = i2h
        ctx.desc = i2h.make_desc(x.shape[2:4])
        ctx.eps1 = i2h.state.eps1.clone()       # (the state buffers are updated in place by the next step)
        ctx.v = v
        ctx.mark_non_differentiable(s)
        return s, v

    @staticmethod
    def backward(ctx, gs, gv):
        dW, db, _, _ = ops.conv_lif_backward(ctx.desc, ctx.eps1, ctx.v, None, None, None, None, gv.contiguous(), None,
                                             want_out=False)
        return None, None, dW, (db if ctx.i2h.bias is not None else None)


class Conv2dDCLLlayer(nn.Module):
    """LIF conv + max-pool + frozen local readout (+ trainable output readout on the last layer), reference :512-612."""

    def __init__(self, in_channels, out_channels, kernel_size=5, im_dims=(28, 28), target_size=10, pooling=None,
                 stride=1, dilation=1, padding=2, alpha=.95, alphas=.9, alpharp=.65, wrp=0, act=nn.Sigmoid(),
                 lc_dropout=False, lc_ampl=.5, spiking=True, random_tau=False, output_layer=False):
        super().__init__()
        self.im_dims = tuple(im_dims)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lc_ampl = lc_ampl
        self.output_layer = output_layer
        if pooling is not None:
            self.pooling = _as_pair(pooling) if not hasattr(pooling, '__len__') else tuple(pooling)
        else:
            self.pooling = (1, 1)
        self.kernel_size = kernel_size
        self.target_size = target_size
        if wrp > 0:
            if not spiking:
                raise Exception('Non-spiking not allowed with refractory neurons')
            self.i2h = ContinuousRelativeRefractoryConv2D(
                in_channels, out_channels, kernel_size, padding=padding, dilation=dilation, stride=stride,
                alpha=alpha, alphas=alphas, alpharp=alpharp, wrp=wrp, act=act, random_tau=random_tau)
        else:
            self.i2h = ContinuousConv2D(
                in_channels, out_channels, kernel_size, padding=padding, dilation=dilation, stride=stride,
                alpha=alpha, alphas=alphas, act=act, spiking=spiking, random_tau=random_tau)
        conv_shape = self.i2h.get_output_shape(self.im_dims)
        ph_, pw_ = self.pooling
        # what nn.MaxPool2d(kernel=stride=pooling, padding=(pooling-1)//2) yields (reference :545-549, :567)
        pooled = ((conv_shape[0] + 2 * ((ph_ - 1) // 2) - ph_) // ph_ + 1,
                  (conv_shape[1] + 2 * ((pw_ - 1) // 2) - pw_) // pw_ + 1)
        self.output_shape = torch.Size(pooled)
        flat = self.get_flat_size()
        if flat != out_channels * pooled[0] * pooled[1]:
            # the reference sizes i2o from conv_shape // pooling (:593-597) and would fail in F.linear here
            raise RuntimeError('pooled map %s does not match conv_shape // pooling %s'
                               % (tuple(pooled), self.get_output_shape()))
        self.i2o = nn.Linear(flat, target_size, bias=True)
        self.i2o.weight.requires_grad = False
        self.i2o.bias.requires_grad = False
        # dropout on the local readout (reference :572-575, :603): an nn.Dropout module (follows .train() / .eval()) applied to
        # pvoutput behind the HIP step; while it is active the learning step takes the autograd path (its mask scales g_p)
        self.lc_dropout = lc_dropout
        self.dropout = nn.Dropout(p=lc_dropout) if lc_dropout is not False else None
        if output_layer:
            self.output_ = nn.Linear(flat, target_size, bias=True)
        self.reset_lc_parameters()
        if output_layer:
            self.stacked_readout()              # aliasing established HERE (and in _apply), not lazily inside a forward
            self._register_state_dict_hook(Conv2dDCLLlayer._unalias_state_dict)

    def reset_lc_parameters(self):
        stdv = self.lc_ampl / math.sqrt(self.i2o.weight.size(1))
        self.i2o.weight.data.uniform_(-stdv, stdv)
        if self.i2o.bias is not None:
            self.i2o.bias.data.uniform_(-stdv, stdv)
        self.weights_written()

    def weights_written(self):
        """Drop everything derived from this layer's weights: the int8 form of the conv weight and the permuted readout
        matrix of the fused sequence epilogue.  Their caches are keyed on the tensors' version counters, which in-place torch ops and
        load_state_dict advance — but NOT a write through `.data` or through the raw device pointer (dcll_adam_step, the
        native learning step): whoever writes that way calls this."""
        if getattr(self, 'i2h', None) is not None:
            self.i2h._q8 = None
        self._ro_cache = None                   # (the stacked readout aliases the parameters: nothing to drop)

    def get_output_shape(self):
        conv_shape = self.i2h.get_output_shape(self.im_dims)
        return conv_shape[0] // self.pooling[0], conv_shape[1] // self.pooling[1]

    def get_flat_size(self):
        h, w = self.get_output_shape()
        return int(h * w * self.out_channels)

    def forward(self, input):
        """-> (output, pvoutput, pv, pvmem): next-layer spikes (or output_ logits on the last layer), local
        readout logits, pooled sigmoid, un-pooled membrane (reference :599-608) — one C-ABI call."""
        if self.i2h._general():
            return self._forward_general(input)
        if getattr(self, 'build_graph', False) and torch.is_grad_enabled():
            # local-learning step: same kernels, wrapped in an autograd node (see _ConvLIFStepFn)
            out = _ConvLIFStepFn.apply(self, input, self.i2h.weight, self.i2h.bias,
                                       self.output_.weight if self.output_layer else None,
                                       self.output_.bias if self.output_layer else None)
            if self.output_layer:
                s, p, pv, v, o = out
                return o, self._drop(p), pv, v
            s, p, pv, v = out
            return s, self._drop(p), pv, v
        # (`_finish`: a slice asks for the step's fused tail — the recorded argmax — DCLLClassification.forward;
        #  `_skip_vmem`: ConvNetwork.test discards the tuple, so the un-pooled membrane map is not written: pvmem = None)
        #  `_defer_sink` (with `_finish`): ConvNetwork.test launches the readout tails of all slices together, later)
        #  `_step_bufs`: ConvNetwork.test keeps one set of output maps per layer instead of allocating five tensors per step —
        #  it discards the tuple; a direct caller of forward() gets fresh tensors, like the reference's)
        fin = self.__dict__.get('_finish')
        s, p, o, pv, v = self.i2h._step(input, self.pooling, self.i2o, self.output_ if self.output_layer else None,
                                        stacked=self.stacked_readout() if self.output_layer else None,
                                        finish=fin, want_v=not self.__dict__.get('_skip_vmem', False),
                                        defer_ro=fin is not None and self.__dict__.get('_defer_sink') is not None,
                                        out=self.__dict__.get('_step_bufs'))
        return (o if self.output_layer else s), self._drop(p), pv, v

    def _forward_general(self, input):
        """forward() of a layer built with an option outside the fused step (_is_sigmoid): the HIP step yields spikes and v
        (reference :407-426 / :485-509 up to the threshold), the rest of :599-608 — act, the two poolings, i2o, output_ — are
        torch ops on the device; under train_dcll's graph v is differentiable through _ConvLIFVFn."""
        i2h = self.i2h
        if getattr(self, 'build_graph', False) and torch.is_grad_enabled():
            s, v = _ConvLIFVFn.apply(i2h, input, i2h.weight, i2h.bias)
        else:
            s, _, _, _, v = i2h._step(input)
        pv = i2h.act(v)
        output = _max_pool(s if i2h.spiking else v, self.pooling)
        pv = _max_pool(pv, self.pooling)
        flat = pv.reshape(pv.shape[0], -1)
        pvoutput = self._drop(torch.nn.functional.linear(flat, self.i2o.weight, self.i2o.bias))
        if self.output_layer:
            output = torch.nn.functional.linear(flat.detach(), self.output_.weight, self.output_.bias)
        return output, pvoutput, pv, v

    def _drop(self, p):
        return p if self.dropout is None else self.dropout(p)

    def dropout_active(self):
        """True while pvoutput is really masked: lc_dropout > 0 and the module in training mode."""
        return self.dropout is not None and self.training and self.dropout.p > 0

    def init_hiddens(self, batch_size, init_value=0):
        self.i2h.init_state(batch_size, self.im_dims, init_value=init_value)
        return self

    # -- whole-sequence fast path ---------------------------------------------------------------------------------
    def sequence_kind(self):
        """'cells' / 'packed' if a fused all-T kernel exists for this geometry (include/dcll_hip.h), else None:
        the 7x7 / pad 3 / pool 1 layers of radio_ml_conv.yaml (1 -> <=32 and 32 -> 32 channels; 16x16 plane or H % 8 == 0,
        W % 32 == 0), or the (1,3) / pad (0,1) / pool (1,2) layers of radio_ml_conv_ref.yaml (1 -> 64 and 64 -> 64
        channels; W a power of two <= 256; k_lif_seq_w3)."""
        i = self.i2h
        H, W = self.im_dims
        if i._general() or i.tau_per_channel() is None or self.dropout_active():   # (a masked readout is drawn per step)
            return None
        if i.kernel_size == (7, 7) and i.padding == (3, 3) and self.pooling == (1, 1) and i.out_channels <= 32 and \
                ((H, W) == (16, 16) or (H % 8 == 0 and W % 32 == 0)):         # k_lif_seq_c1/c32 or the tiled c1t/c32t
            if i.in_channels == 1:
                return 'cells'
            if i.in_channels == 32 and i.out_channels == 32:
                return 'packed'
        if i.kernel_size == (1, 3) and i.padding == (0, 1) and self.pooling == (1, 2) and i.out_channels == 64 and \
                2 <= W <= 256 and (W & (W - 1)) == 0 and (H * W) % 32 == 0:
            if i.in_channels == 1 and (H * W) % 128 == 0:
                return 'cells'
            if i.in_channels == 64:
                return 'packed'
        return None

    def _apply(self, fn, *args, **kwargs):
        """module.to() / .cuda() / .cpu() re-bind every parameter's .data: the readouts' shared storage is re-established
        right behind it — a defined point, never inside a (possibly captured) forward."""
        out = super()._apply(fn, *args, **kwargs)
        if self.output_layer and 'output_' in self._modules:
            self.__dict__.pop('_stacked', None)
            self.stacked_readout()
        return out

    @staticmethod
    def _unalias_state_dict(module, state_dict, prefix, local_metadata):
        """state_dict() hands out the readouts' tensors as tensors of their own (contiguous clones), not as offset views of
        the stacked storage: a checkpoint then holds four independent tensors, like the reference's."""
        for name in ('i2o.weight', 'i2o.bias', 'output_.weight', 'output_.bias'):
            t = state_dict.get(prefix + name)
            if t is not None:
                state_dict[prefix + name] = t.detach().clone()
        return state_dict

    def stacked_readout(self):
        """(weight (24|48, K), bias) of i2o, with output_ stacked behind it on the output layer — one readout GEMM
        then serves both (reference :602-606).  Free of copies: the two Linear modules' parameters are made VIEWS of one
        stacked storage (their `.data` re-bound, values kept), so every later write to them — an in-place torch op,
        load_state_dict's copy_, dcll_adam_step through the raw pointer — is a write to the stacked matrix.  The aliasing is
        established at the end of __init__ and behind every _apply (to / cuda / cpu re-bind `.data`); should somebody
        re-bind `.data` by hand, the mismatch is detected here by address and repaired — but never during a stream
        capture, where the new storage would land in the graph's private pool (round-4 advisor)."""
        if not self.output_layer:
            return self.i2o.weight, self.i2o.bias
        W1, W2, b1, b2 = self.i2o.weight, self.output_.weight, self.i2o.bias, self.output_.bias
        n, K = W1.shape
        st = self.__dict__.get('_stacked')
        if not (st is not None and st[0].device == W1.device and W1.data_ptr() == st[0].data_ptr() and
                W2.data_ptr() == st[0].data_ptr() + 4 * n * K and b1.data_ptr() == st[1].data_ptr() and
                b2.data_ptr() == st[1].data_ptr() + 4 * n and W2.shape == W1.shape):
            if W1.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError('Conv2dDCLLlayer: the readout parameters were re-bound (.data) since the last step; their '
                                   'shared storage cannot be re-established inside a stream capture — run one eager step first')
            with torch.no_grad():
                Wt = torch.cat([W1.detach(), W2.detach()], 0).contiguous()
                bias = torch.cat([b1.detach(), b2.detach()], 0).contiguous()
                W1.data, W2.data, b1.data, b2.data = Wt[:n], Wt[n:], bias[:n], bias[n:]
            st = self._stacked = (Wt, bias)
        return st

    def fused_readout_weights(self):
        """(permuted weights, bias) of the readout(s) for the fused epilogue of the 'packed' sequence kernel:
        i2o, with output_ stacked behind it on the output layer.  Cached until a weight tensor changes."""
        mods = [self.i2o] + ([self.output_] if self.output_layer else [])
        key = tuple((m.weight.data_ptr(), m.weight._version, m.bias.data_ptr(), m.bias._version) for m in mods)
        cache = getattr(self, '_ro_cache', None)
        if cache is None or cache[0] != key:
            with torch.no_grad():
                Wt = torch.cat([m.weight.detach() for m in mods], 0)
                bias = torch.cat([m.bias.detach() for m in mods], 0).contiguous()
                cache = (key, ops.permute_readout(Wt), bias)
            self._ro_cache = cache
        return cache[1], cache[2]

    def forward_sequence(self, inp, T, B, kind, want_spikes=True, buffers=None, fuse_readout=False, batch_slice=None,
                         want_pv=True, lowhigh_iter0=None, presigmoid=False):
        """All T steps in one launch.  inp: cells (T,B) int32 ('cells') or packed spikes (T,B,32,H*W/32) int32 ('packed').
        Neuron state is read from / written back to self.i2h.state (rows batch_slice .. batch_slice+B of it when
        `batch_slice` is given: a chunk of a larger batch).
        -> (packed spikes, pv (T,B,C,H,W) or None, logits (T,B,24|48) or None).  With fuse_readout ('packed' only)
        the readout(s) are computed in the kernel's epilogue and pv is not materialised.
        `lowhigh_iter0` (the slice's iteration count before the sequence): also count pv's first / last histogram bin
        on the reference's histogram steps (:658-661); the (n,2) int64 counters are left in buffers['lowhigh'].
        `presigmoid`: the returned pv buffer holds v = pvmem + arp (max-pooled where the layer pools) instead of
        sigmoid(v); the caller's readout applies the sigmoid (ops.readout_act) — the statistics are unchanged."""
        i2h = self.i2h
        buffers = {} if buffers is None else buffers
        buffers.pop('lowhigh', None)
        if not want_pv or fuse_readout:
            lowhigh_iter0 = None                  # no pv is materialised: no statistics (documented in test_sequence)
        if batch_slice is None:
            if i2h.state is None or i2h.state.eps0.shape[0] != B:
                i2h.init_state(B, self.im_dims)
            st = i2h.state
        else:
            # a chunk of the batch: B samples starting at batch_slice of the layer's (larger) state tensors
            st = type(i2h.state)(*[t[batch_slice:batch_slice + B] for t in i2h.state])
        desc = i2h.make_desc(self.im_dims, self.pooling, self.target_size, self.output_layer)
        tau4 = i2h.tau_per_channel()
        arp = st.arp if len(st) > 2 else None
        q8 = i2h.int8_weights()
        presigmoid = bool(presigmoid and want_pv and not fuse_readout)
        with torch.no_grad():
            if kind == 'cells':
                spk, pv, _ = ops.conv_lif_sequence_cells(desc, inp, i2h.weight, i2h.bias, tau4, st.eps0, st.eps1, arp,
                                                         T, B, want_spikes=want_spikes, want_pv=want_pv, out=buffers,
                                                         lowhigh_iter0=lowhigh_iter0, q8=q8, presigmoid=presigmoid)
                return spk, pv, None
            if kind == 'iq':        # inp = (iq (B,2,L), thr_i, thr_q, t0[, tail]): encoder fused into the layer kernel
                iq, thr_i, thr_q, t0 = inp[:4]
                tail = inp[4] if len(inp) > 4 else None
                spk, pv, _ = ops.conv_lif_sequence_iq(desc, iq, thr_i, thr_q, t0, i2h.weight, i2h.bias, tau4, st.eps0,
                                                      st.eps1, arp, T, B, want_spikes=want_spikes, want_pv=want_pv,
                                                      out=buffers, lowhigh_iter0=lowhigh_iter0, q8=q8,
                                                      presigmoid=presigmoid, tail=tail)
                return spk, pv, None
            if fuse_readout:
                Wp, rb = self.fused_readout_weights()
                spk, _, _, logits = ops.conv_lif_sequence(desc, inp, i2h.weight, i2h.bias, tau4, st.eps0, st.eps1, arp,
                                                          T, B, want_spikes=want_spikes, want_pv=False, out=buffers,
                                                          ro_Wp=Wp, ro_b=rb, q8=q8)
                return spk, None, logits
            spk, pv, _ = ops.conv_lif_sequence(desc, inp, i2h.weight, i2h.bias, tau4, st.eps0, st.eps1, arp, T, B,
                                               want_spikes=want_spikes, want_pv=want_pv, out=buffers,
                                               lowhigh_iter0=lowhigh_iter0, q8=q8, presigmoid=presigmoid)
        return spk, pv, None


# ---------------------------------------------------------------------------------------------------------------
# dense LIF dynamics (reference :72-266)
# ---------------------------------------------------------------------------------------------------------------
class CLLDenseModule(nn.Module):
    NeuronState = namedtuple('NeuronState', ['eps0', 'eps1'])

    def __init__(self, in_channels, out_channels, bias=True, alpha=.9, alphas=.85, act=nn.Sigmoid(), spiking=True,
                 random_tau=False):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()
        self.act = act
        self.random_tau = random_tau
        self._set_tau(torch.Tensor([alpha]), torch.Tensor([alphas]))
        self.spiking = spiking
        self.wrp, self.alpharp = 0., .65
        self.state = None

    _set_tau = ContinuousConv2D._set_tau
    _outputs = ContinuousConv2D._outputs

    def _general(self):
        """True when the layer was built with an option outside the fused step (see _is_sigmoid; the dense kernels take a
        NULL bias as they are)."""
        return not (_is_sigmoid(self.act) and self.spiking and self.bias is not None)

    def reset_parameters(self):
        stdv = 1. / math.sqrt(self.weight.size(1))
        self.weight.data.uniform_(-stdv * 1e-2, stdv * 1e-2)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    def _alloc_state(self, batch_size, init_value):
        dev = self.weight.device
        return [torch.zeros(batch_size, self.in_channels, device=dev) + init_value,
                torch.zeros(batch_size, self.in_channels, device=dev) + init_value]

    def init_state(self, batch_size, init_value=0):
        self.state = self.NeuronState(*self._alloc_state(batch_size, init_value))
        if self.random_tau:
            self.randomize_tau()
        return self.state

    def randomize_tau(self, low=[5, 5], high=[10, 35]):
        taum = np.random.uniform(low[1], high[1], size=[self.in_channels]) * 1e-3
        taus = np.random.uniform(low[0], high[0], size=[self.in_channels]) * 1e-3
        dev = self.weight.device
        self._set_tau(torch.Tensor(1 - 1e-3 / taum).to(dev), torch.Tensor(1 - 1e-3 / taus).to(dev))

    def make_desc(self, i2o=None):
        return DenseDesc(self.in_channels, self.out_channels, 0 if i2o is None else i2o.weight.shape[0],
                         int(self.alpha.numel() > 1), int(self.wrp > 0), float(self.alpharp), float(self.wrp))

    def _step(self, input, i2o=None):
        if self.state is None or not (input.shape[0] == self.state.eps0.shape[0] == self.state.eps1.shape[0]):
            old = -1 if self.state is None else self.state.eps0.shape[0]
            logger.warning("Batch size changed from {} to {} since last iteration. Reallocating states."
                           .format(old, input.shape[0]))
            self.init_state(input.shape[0])
        desc = self.make_desc(i2o)
        st = self.state
        with torch.no_grad():
            return ops.dense_lif_step(desc, input, self.weight, self.bias, self.alpha, self.tau_m__dt, self.alphas,
                                      self.tau_s__dt, st.eps0, st.eps1, st.arp if len(st) > 2 else None,
                                      None if i2o is None else i2o.weight, None if i2o is None else i2o.bias)

    def forward(self, input):
        s, _, pv, v = self._step(input)
        o, pv = self._outputs(s, pv, v)
        return o, pv, v


class CLLDenseRRPModule(CLLDenseModule):
    NeuronState = namedtuple('NeuronState', ('eps0', 'eps1', 'arp'))

    def __init__(self, in_channels, out_channels, bias=True, alpha=.95, alphas=.9, alpharp=.65, wrp=100,
                 act=nn.Sigmoid(), spiking=True, random_tau=False):
        super().__init__(in_channels, out_channels, bias, alpha, alphas, act, spiking=spiking, random_tau=random_tau)
        self.wrp = wrp
        self.alpharp = alpharp

    def init_state(self, batch_size, init_value=0):
        st = self._alloc_state(batch_size, init_value)
        st.append(torch.zeros(batch_size, self.out_channels, device=self.weight.device) + init_value)
        self.state = self.NeuronState(*st)
        return self.state

    def forward(self, input):
        if not self.spiking:
            raise Exception('Refractory not allowed in non-spiking mode')
        return super().forward(input)


class _DenseLIFStepFn(torch.autograd.Function):
    """One DenseDCLLlayer step as an autograd node: forward = dcll_dense_lif_step, backward = dcll_dense_lif_backward.
    Gradients reach i2h.weight / i2h.bias via pvoutput, pv and pvmem; i2o is frozen, spikes and the neuron state carry no
    gradient — the graph the reference builds with eager ops (:131-148 / :171-195, :250-255)."""

    @staticmethod
    def forward(ctx, layer, x, W, b):
        i2h = layer.i2h
        s, p, pv, v = i2h._step(x, layer.i2o)
        ctx.layer = layer
        ctx.desc = i2h.make_desc(layer.i2o)
        ctx.eps1 = i2h.state.eps1.clone()          # (the state buffers are updated in place by the next step)
        ctx.pv = pv
        ctx.mark_non_differentiable(s)
        return s, p, pv, v

    @staticmethod
    def backward(ctx, gs, gp, gpv, gv):
        dW, db = ops.dense_lif_backward(ctx.desc, ctx.eps1, ctx.pv, gp, gpv, gv, ctx.layer.i2o.weight)
        return None, None, dW, (db if ctx.layer.i2h.bias is not None else None)


class _DenseLIFVFn(torch.autograd.Function):
    """The neuron part of a GENERAL dense layer step (_is_sigmoid) as an autograd node: forward = dcll_dense_lif_step without a
    readout -> (spikes, v); backward = dcll_dense_lif_backward fed with dL/dv alone (dW = dv^T . eps1, db = sum dv)."""

    @staticmethod
    def forward(ctx, i2h, x, W, b):
        s, _, pv, v = i2h._step(x)
        ctx.i2h = i2h
        ctx.desc = i2h.make_desc()
        ctx.eps1 = i2h.state.eps1.clone()
        ctx.pv = pv
        ctx.mark_non_differentiable(s)
        return s, v

    @staticmethod
    def backward(ctx, gs, gv):
        dW, db = ops.dense_lif_backward(ctx.desc, ctx.eps1, ctx.pv, None, None, gv.contiguous(), None)
        return None, None, dW, (db if ctx.i2h.bias is not None else None)


class DenseDCLLlayer(nn.Module):
    def __init__(self, in_channels, out_channels, target_size=None, bias=True, alpha=.9, alphas=.85, alpharp=.65,
                 wrp=0., act=nn.Sigmoid(), lc_dropout=False, lc_ampl=.5, spiking=True, random_tau=False,
                 output_layer=False):
        if target_size is None:
            target_size = out_channels
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lc_ampl = lc_ampl
        self.target_size = target_size
        self.output_layer = False          # the reference forces this (:222)
        if wrp > 0:
            self.i2h = CLLDenseRRPModule(in_channels, out_channels, alpha=alpha, alphas=alphas, alpharp=alpharp,
                                         wrp=wrp, bias=bias, act=act, spiking=spiking, random_tau=random_tau)
        else:
            self.i2h = CLLDenseModule(in_channels, out_channels, alpha=alpha, alphas=alphas, bias=bias, act=act,
                                      spiking=spiking, random_tau=random_tau)
        self.i2o = nn.Linear(out_channels, target_size, bias=bias)
        self.i2o.weight.requires_grad = False
        if bias:
            self.i2o.bias.requires_grad = False
        self.input_size = self.out_channels
        self.reset_lc_parameters()
        self.lc_dropout = lc_dropout
        self.dropout = nn.Dropout(p=lc_dropout) if lc_dropout is not False else None       # (reference :238-241, :253)

    reset_lc_parameters = Conv2dDCLLlayer.reset_lc_parameters

    def weights_written(self):
        """(nothing is derived from a dense layer's weights)"""

    def forward(self, input):
        """-> (output spikes, pvoutput, pv, pvmem)  (reference :250-255)."""
        x = input.reshape(-1, self.in_channels)
        i2h = self.i2h
        if i2h._general():
            # an option outside the fused step (_is_sigmoid): spikes and v from the HIP step, act and i2o as torch ops; under
            # train_dcll's graph v is differentiable through _DenseLIFVFn (reference :131-148 / :171-195, :250-255)
            if getattr(self, 'build_graph', False) and torch.is_grad_enabled():
                s, v = _DenseLIFVFn.apply(i2h, x, i2h.weight, i2h.bias)
            else:
                s, _, _, v = i2h._step(x)
            pv = i2h.act(v)
            p = torch.nn.functional.linear(pv, self.i2o.weight, self.i2o.bias)
            return ((s if i2h.spiking else v).detach(), (p if self.dropout is None else self.dropout(p)), pv, v)
        if getattr(self, 'build_graph', False) and torch.is_grad_enabled():
            # local-learning step: same kernel, wrapped in an autograd node (see _DenseLIFStepFn)
            s, p, pv, v = _DenseLIFStepFn.apply(self, x, self.i2h.weight, self.i2h.bias)
        else:
            s, p, pv, v = self.i2h._step(x, self.i2o)
        return s, (p if self.dropout is None else self.dropout(p)), pv, v

    def forward_sequence(self, x_seq, want_v=False):
        """`for t: self.forward(x_seq[t])` in one C-ABI call (dcll_dense_lif_sequence): x_seq (T,B,...) -> (output spikes
        (T,B,out), pvoutput (T,B,target), pv (T,B,out), pvmem (T,B,out) or None).  Small layers keep their neuron state on
        chip for all T; the results are bit-identical to T calls of forward()."""
        i2h = self.i2h
        T, B = x_seq.shape[0], x_seq.shape[1]
        x_seq = x_seq.reshape(T, B, self.in_channels)
        if i2h._general():        # (an option outside the fused kernels: step by step, stacked)
            outs = [self.forward(x_seq[t]) for t in range(T)]
            return tuple(torch.stack([o_[k] for o_ in outs]) if (k < 3 or want_v) else None for k in range(4))
        if i2h.state is None or i2h.state.eps0.shape[0] != B:
            i2h.init_state(B)
        desc = DenseDesc(i2h.in_channels, i2h.out_channels, self.i2o.weight.shape[0], int(i2h.alpha.numel() > 1),
                         int(i2h.wrp > 0), float(i2h.alpharp), float(i2h.wrp))
        st = i2h.state
        with torch.no_grad():
            return ops.dense_lif_sequence(desc, x_seq, i2h.weight, i2h.bias, i2h.alpha, i2h.tau_m__dt, i2h.alphas,
                                          i2h.tau_s__dt, st.eps0, st.eps1, st.arp if len(st) > 2 else None,
                                          self.i2o.weight, self.i2o.bias, want_v=want_v)

    def init_hiddens(self, batch_size, init_value=0):
        self.i2h.init_state(batch_size, init_value=init_value)
        return self


# ---------------------------------------------------------------------------------------------------------------
# slices: per-step driver + vote collection (reference :615-749)
# ---------------------------------------------------------------------------------------------------------------
class DCLLBase(nn.Module):
    num_instances = 0

    def __init__(self, dclllayer, name='DCLLbase', batch_size=48, loss=torch.nn.MSELoss, optimizer=optim.SGD,
                 kwargs_optimizer={'lr': 5e-5}, burnin=200, collect_stats=False):
        super().__init__()
        self.dclllayer = dclllayer
        if loss is not None:
            self.crit = loss().to(device)
            self.output_crit = loss().to(device)
        if optimizer is not None:
            self.optimizer = self._make_optimizer(optimizer, dclllayer.i2h.parameters(), dict(kwargs_optimizer))
            if self.dclllayer.output_layer:
                self.optimizer2 = self._make_optimizer(optimizer, dclllayer.output_.parameters(), dict(lr=1e-4))
        self.burnin = burnin
        self.batch_size = batch_size
        self.collect_stats = collect_stats
        self.init(self.batch_size)
        self.stats_bins = np.linspace(0, 1, 20)
        self.name = name
        self.slice_id = DCLLBase.num_instances
        DCLLBase.num_instances += 1

    @property
    def clout(self):
        """Per-step argmax of the slice as the reference keeps it: a list of T numpy arrays (B,).  Internally the
        entries stay on the device until somebody looks (one transfer instead of one sync per timestep)."""
        if any(isinstance(c, torch.Tensor) for c in self._clout):
            # device entries (appended without a sync, possibly after an earlier look converted the older ones):
            # one stacked transfer for all of them
            idx = [k for k, c in enumerate(self._clout) if isinstance(c, torch.Tensor)]
            host = torch.stack([self._clout[k].to(torch.int64) for k in idx]).cpu().numpy()
            for k, row in zip(idx, host):
                self._clout[k] = row
        return self._clout

    @clout.setter
    def clout(self, value):
        self._clout = value
        self._seq_vote = None           # (a stashed device-side vote belongs to the list it was computed from)

    def init(self, batch_size, init_states=True):
        self.clout = []
        self.activity_hist = []
        self.iter = 0
        if init_states:
            self.dclllayer.init_hiddens(batch_size, init_value=0)

    def forward(self, input):
        self.iter += 1
        o, p, pv, pvmem = self.dclllayer.forward(input)
        if self.collect_stats and (self.iter % 20) == 0:
            # the reference histograms pv on the host (np.histogram, 19 bins over [0,1]) and reports only the first and
            # the last bin (:678-688): those two are counted on the device (dcll_pv_lowhigh, numpy's bin edges), 16
            # bytes cross PCIe, and only when write_stats asks for them
            self.activity_hist.append((ops.pv_lowhigh(pv.detach(), 1, self.iter - 1)[0], pv.numel()))
        return o, p, pv, pvmem

    def _activity_rows(self):
        """activity_hist as the (n, 19) array the reference keeps: bins 0 and 18 are exact, the other 17 bins are not
        tracked (their total sits in bin 9) — write_stats (:678-688) reads only pd[0] and pd[-1]."""
        rows = []
        for counts, numel in self.activity_hist:
            lo, hi = (int(c) for c in counts.cpu())
            h = np.zeros(19)
            h[0], h[-1], h[9] = lo, hi, numel - lo - hi
            rows.append(h)
        return np.asarray(rows)

    def write_stats(self, writer, label, epoch):
        writer.add_histogram(self.name + '/weight', self.dclllayer.i2h.weight.flatten(), epoch)
        if self.dclllayer.i2h.bias is not None:
            writer.add_histogram(self.name + '/bias', self.dclllayer.i2h.bias.flatten(), epoch)
        if self.collect_stats and len(self.activity_hist):
            pd = np.mean(self._activity_rows(), axis=0)
            pd = pd / pd.sum()
            writer.add_scalar(self.name + '/low_pv/' + label, pd[0], epoch)
            writer.add_scalar(self.name + '/high_pv/' + label, pd[-1], epoch)
            print(self.name + ' low:{0:1.3} high:{1:1.3}'.format(pd[0], pd[-1]))

    @staticmethod
    def _make_optimizer(optimizer, params, kwargs):
        """optimizer(params, **kwargs) as in the reference (:634-638).  With DCLL_NATIVE_LEARNING=0 (the autograd
        fallback) torch's Adam family is asked for its fused multi-tensor form; the native learning step below applies
        Adam itself (dcll_adam_step) on the state tensors of this very optimizer object."""
        params = list(params)
        if (os.environ.get('DCLL_NATIVE_LEARNING', '1') == '0' and os.environ.get('DCLL_FUSED_OPTIMIZER', '1') != '0' and
                optimizer in (optim.Adam, optim.AdamW) and 'fused' not in kwargs and 'foreach' not in kwargs and params
                and all(p.is_cuda for p in params)):
            try:
                return optimizer(params, fused=True, **kwargs)
            except (TypeError, RuntimeError, ValueError):
                pass
        return optimizer(params, **kwargs)

    # -- native local-learning step: HIP kernels only, no autograd graph, no torch op per timestep ---------------------
    def _native_learning(self):
        """The loss kind (ops.LOSS_KINDS) if this slice's learning step runs without torch ops — forward, loss gradient,
        backward and Adam as C-ABI calls — else None (then train_dcll builds the autograd graph around the same HIP
        forward / backward).  Served: Conv2dDCLLlayer; crit = SmoothL1Loss (beta 1) or MSELoss with mean reduction;
        optimizer(s) = torch.optim.Adam without amsgrad / maximize / capturable / fused; DCLL_NATIVE_LEARNING != 0.
        A DenseDCLLlayer slice (with a bias) is served the same way (_learn_dense)."""
        cached = getattr(self, '_native_kind', _UNSET)      # (0 is a loss kind: SmoothL1Loss)
        if cached is not _UNSET:
            return cached
        kind = None
        crit, opt = getattr(self, 'crit', None), getattr(self, 'optimizer', None)
        ok = (os.environ.get('DCLL_NATIVE_LEARNING', '1') != '0' and
              isinstance(self.dclllayer, (Conv2dDCLLlayer, DenseDCLLlayer)) and
              not self.dclllayer.i2h._general() and            # (act / spiking / bias / stride ...: the autograd path)
              crit is not None and opt is not None and getattr(crit, 'reduction', None) == 'mean' and
              self.dclllayer.dropout is None)               # (lc_dropout: torch's mask and its gradient, autograd path)
        if ok and type(crit) is nn.SmoothL1Loss and getattr(crit, 'beta', 1.0) == 1.0:
            kind = ops.LOSS_KINDS['SmoothL1Loss']
        elif ok and type(crit) is nn.MSELoss:
            kind = ops.LOSS_KINDS['MSELoss']
        if kind is not None and isinstance(self.dclllayer, DenseDCLLlayer) and self.dclllayer.i2h.bias is None:
            kind = None
        if kind is not None and self.dclllayer.output_layer:
            if type(self.output_crit) is not type(crit) or getattr(self.output_crit, 'reduction', None) != 'mean':
                kind = None
        opts = [opt] + ([getattr(self, 'optimizer2', None)] if (kind is not None and self.dclllayer.output_layer) else [])
        for o in opts:
            if kind is None:
                break
            if type(o) is not optim.Adam:
                kind = None
                break
            for g in o.param_groups:
                if g.get('amsgrad') or g.get('maximize') or g.get('capturable') or g.get('fused') or \
                        g.get('differentiable') or isinstance(g['lr'], torch.Tensor):
                    kind = None
        self._native_kind = kind
        return kind

    def _learn_forward_backward(self, input, target, want_loss=True, clout_out=None):
        """Forward of one step plus — once iter >= burnin (reference :691) — the gradients of the local loss(es) in the
        .grad of i2h.weight / i2h.bias (and output_.weight / output_.bias): dcll_conv_lif_step -> dcll_local_loss_grad
        -> dcll_conv_lif_backward, all on preallocated buffers.  No optimizer step.
        clout_out: write the recorded argmax there instead of appending a fresh tensor to clout (graph capture: the
        caller copies it out after each replay).
        -> (output, pvoutput, pv, pvmem, loss (1,) device tensor or None, learned)"""
        ctx = self._learn_forward(input, target, want_loss=want_loss, clout_out=clout_out)
        loss = self._learn_tail(ctx)
        return ctx['out'] + (loss, ctx['learned'])

    def _learn_forward(self, input, target, want_loss=True, clout_out=None, defer=False, want_v=True):
        """The layer kernel of one learning step (+ its readout tail unless `defer`): -> ctx for _learn_tail.  With `defer`
        the readouts, loss gradients and the backward are ALL left to _learn_tail: ConvNetwork.learn launches the layer
        kernels of every slice first — slice l+1 needs slice l's spikes, not its readouts — and the tails behind them.
        `want_v=False` (the caller discards the returned tuple): a layer without pooling does not store its membrane map —
        the backward takes sigmoid' from the stored pv, the same bits (dcll_conv_lif_backward with v == NULL)."""
        L = self.dclllayer
        i2h = L.i2h
        bufs = self.__dict__.setdefault('_learn_bufs', {})
        self.iter += 1
        with torch.no_grad():
            learned = self.iter >= self.burnin
            # (DCLLClassification records the per-step argmax once the burn-in is over, :724-728)
            rec = isinstance(self, DCLLClassification)
            fin = None
            if learned and not want_loss:
                # the readouts' finishing launch also yields the local-loss gradients and the recorded argmax
                fin = dict(clout=(clout_out if clout_out is not None else True) if rec else None, target=target,
                           kind=self._native_learning())
            elif defer:
                fin = dict(clout=None)
            s, p, o, pv, v = i2h._step(input, L.pooling, L.i2o, L.output_ if L.output_layer else None, out=bufs,
                                       stacked=L.stacked_readout() if L.output_layer else None, finish=fin,
                                       defer_ro=defer, want_v=want_v or not self._backward_from_pv())
            if self.collect_stats and (self.iter % 20) == 0:
                self.activity_hist.append((ops.pv_lowhigh(pv, 1, self.iter - 1)[0], pv.numel()))
        return dict(out=((o if L.output_layer else s), p, pv, v), fin=fin, learned=learned, rec=rec, input=input,
                    target=target, want_loss=want_loss, clout_out=clout_out, p=p, o=o, pv=pv, v=v)

    def _learn_dense(self, input, target):
        """_learn_forward_backward for a DenseDCLLlayer slice: dcll_dense_lif_step -> dcll_local_loss_grad ->
        dcll_dense_lif_backward into the .grad of i2h.weight / i2h.bias.  No optimizer step.
        -> (output, pvoutput, pv, pvmem, loss (1,) device tensor or None, learned)"""
        L = self.dclllayer
        i2h = L.i2h
        bufs = self.__dict__.setdefault('_learn_bufs', {})
        self.iter += 1
        with torch.no_grad():
            learned = self.iter >= self.burnin
            rec = isinstance(self, DCLLClassification)
            s, p, pv, v = i2h._step(input.reshape(-1, L.in_channels), L.i2o)
            if self.collect_stats and (self.iter % 20) == 0:
                self.activity_hist.append((ops.pv_lowhigh(pv, 1, self.iter - 1)[0], pv.numel()))
            if not learned:
                return s, p, pv, v, None, False
            res = ops.local_loss_grad(p, None, target, self._native_learning(), out=bufs, want_loss=True, want_clout=rec)
            if rec:
                self._clout.append(res[3])
            for q in (i2h.weight, i2h.bias):
                if q.grad is None:
                    q.grad = torch.zeros_like(q)
            gb = bufs.setdefault('grads', {})
            gb.update(dW=i2h.weight.grad, db=i2h.bias.grad)
            ops.dense_lif_backward(i2h.make_desc(L.i2o), i2h.state.eps1, pv, res[0], None, None, L.i2o.weight, out=gb)
        return s, p, pv, v, res[2], True

    def _backward_from_pv(self):
        """True if this slice's backward can take sigmoid' from pv: no pooling, <= 32 readout rows (k_bwd_dv_nopool)."""
        L = self.dclllayer
        return tuple(L.pooling) == (1, 1) and L.i2o.weight.shape[0] <= 32

    def _learn_tail(self, ctx, open_reduce=False, defer_backward=None):
        """What follows the layer kernel of a learning step: the (deferred) readout tail, then — once iter >= burnin — the
        local-loss gradients (from the readouts' finishing launch where it served them, else dcll_local_loss_grad) and
        dcll_conv_lif_backward into the parameters' .grad.  `open_reduce`: the weight gradient's last reduction is left to
        the caller's ops.grad_reduce_adam (self._learn_bufs['grads']['parts']); `defer_backward` (a list, with open_reduce): the
        backward is not launched here but appended for ops.conv_lif_backward_open_multi (all slices' dv in one launch).
        -> loss (1,) device tensor or None"""
        L = self.dclllayer
        i2h = L.i2h
        fin, rec, clout_out = ctx['fin'], ctx['rec'], ctx['clout_out']
        bufs = self._learn_bufs
        with torch.no_grad():
            if fin is not None and 'run_readouts' in fin:
                fin.pop('run_readouts')()
            if not ctx['learned']:
                return None
            p, o, pv, v, target = ctx['p'], ctx['o'], ctx['pv'], ctx['v'], ctx['target']
            if fin is not None and fin.get('done') and fin.get('g_p') is not None:
                g_p, g_o, loss = fin['g_p'], fin['g_o'], None
                if rec and clout_out is None:
                    self._clout.append(fin['clout'])
            else:
                res = ops.local_loss_grad(p, o if L.output_layer else None, target, self._native_learning(), out=bufs,
                                          want_loss=ctx['want_loss'], want_clout=rec, clout_out=clout_out)
                g_p, g_o, loss = res[:3]
                if rec and clout_out is None:
                    self._clout.append(res[3])
            prm = [i2h.weight, i2h.bias] + ([L.output_.weight, L.output_.bias] if L.output_layer else [])
            for q in prm:
                if q.grad is None:
                    q.grad = torch.zeros_like(q)
            desc = i2h.make_desc(ctx['input'].shape[2:4], L.pooling, L.i2o.weight.shape[0], L.output_layer)
            gb = bufs.setdefault('grads', {})
            gb.update(dW=prm[0].grad, db=prm[1].grad)
            if L.output_layer:
                gb.update(d_outW=prm[2].grad, d_outb=prm[3].grad)
            ops.conv_lif_backward(desc, i2h.state.eps1, v, pv, g_p, g_o, None, None, L.i2o.weight,
                                  want_out=L.output_layer, out=gb, open_reduce=open_reduce,
                                  defer=defer_backward if open_reduce else None)
        return loss

    def _grads_into_slab(self):
        """Make the .grad tensors of this slice's trainable parameters views of ONE flat buffer (+ one count element): the
        backward kernels then write straight into what the per-slice all-reduce sends — no gather / scatter copies around
        the collective (parallel.allreduce_slab_begin).  -> the slab."""
        L = self.dclllayer
        prm = [L.i2h.weight, L.i2h.bias] + ([L.output_.weight, L.output_.bias] if L.output_layer else [])
        slab = getattr(self, '_grad_slab', None)
        n = sum(q.numel() for q in prm)
        ok = slab is not None and slab.numel() == n + 1 and slab.device == prm[0].device
        off = 0
        for q in prm:
            ok = ok and q.grad is not None and q.grad.data_ptr() == (slab.data_ptr() + 4 * off if slab is not None else -1)
            off += q.numel()
        if ok:
            return slab
        slab = torch.zeros(n + 1, device=prm[0].device, dtype=torch.float32)
        off = 0
        for q in prm:
            view = slab[off:off + q.numel()].view_as(q)
            if q.grad is not None:
                view.copy_(q.grad)
            q.grad = view
            off += q.numel()
        self._grad_slab = slab
        return slab

    def _grad_tensors(self):
        L = self.dclllayer
        prm = [L.i2h.weight, L.i2h.bias] + ([L.output_.weight, L.output_.bias] if L.output_layer else [])
        return [q.grad for q in prm if q is not None]

    def _adam_tensors(self, advance=True):
        """This slice's parameters as dcll_adam_step entries, on the state tensors of its torch optimizer objects (created
        here, in torch's own layout, on first use): optimizer.state_dict() stays loadable by torch.optim.Adam.
        advance: count this call as an update (state['step'] += 1); False only describes the tensors."""
        out = []
        opts = [self.optimizer] + ([self.optimizer2] if self.dclllayer.output_layer else [])
        for opt in opts:
            for g in opt.param_groups:
                for q in g['params']:
                    if q.grad is None:
                        continue
                    st = opt.state[q]
                    if len(st) == 0:
                        st['step'] = torch.tensor(0.0)
                        st['exp_avg'] = torch.zeros_like(q, memory_format=torch.preserve_format)
                        st['exp_avg_sq'] = torch.zeros_like(q, memory_format=torch.preserve_format)
                    if advance:
                        st['step'] += 1
                    out.append(dict(param=q.data, grad=q.grad, exp_avg=st['exp_avg'], exp_avg_sq=st['exp_avg_sq'],
                                    lr=g['lr'], weight_decay=g['weight_decay'], beta1=g['betas'][0], beta2=g['betas'][1],
                                    eps=g['eps'], step=max(1, int(st['step']))))
        return out

    def train_dcll(self, input, target, do_train=True, regularize=0.05):
        """One local-learning step (reference :690-718): forward; after burn-in the local loss on pvoutput (+ the loss
        on the output_ logits for the output layer), its gradients through the HIP backward kernels, optimizer step(s).
        With torch.distributed initialised the gradients are averaged over the ranks (RCCL) before the step.
        Without regularisers and with SmoothL1 / MSE losses and Adam (what train.py runs) the whole step is C-ABI calls
        (`_native_learning`); otherwise the same HIP forward / backward run inside an autograd node and torch supplies the
        loss module and the optimizer.

        NOTE (native path): output, pvoutput, pv, pvmem and the loss are VIEWS of this slice's preallocated step buffers —
        valid until the slice's next step, which overwrites them (the reference returns fresh tensors each step).  A caller
        that collects them over timesteps must .clone() them; ConvNetwork.learn / train.py only chain `output` into the
        next slice within the same step."""
        from .. import parallel
        if not regularize and self._native_learning() is not None:
            if isinstance(self.dclllayer, DenseDCLLlayer):
                output, pvoutput, pv, pvmem, loss, learned = self._learn_dense(input, target)
            else:
                output, pvoutput, pv, pvmem, loss, learned = self._learn_forward_backward(input, target)
            if learned:
                parallel.allreduce_mean_tensors(self._grad_tensors(), local_n=input.shape[0])
                if do_train:
                    ops.adam_step(self._adam_tensors())
                    self.dclllayer.weights_written()
            # (the loss value lives in a reused one-element device buffer: valid until the next step of this slice)
            return output, pvoutput, pv, pvmem, (loss.reshape(()) if learned else torch.Tensor([0]))
        learn_now = (self.iter + 1) >= self.burnin
        self.dclllayer.build_graph = learn_now
        try:
            output, pvoutput, pv, pvmem = self.forward(input)
        finally:
            self.dclllayer.build_graph = False
        if learn_now:
            self.dclllayer.zero_grad()
            tgt_loss = self.crit(pvoutput, target)
            if self.dclllayer.output_layer:
                tgt_loss = tgt_loss + self.output_crit(output, target)
            if regularize > 0:
                reg_loss = 20.0 * regularize * torch.mean(torch.relu(pvmem + 0.01))
                reg2_loss = 0.1 * regularize * (torch.relu(0.1 - torch.mean(pv)))
                loss = tgt_loss + reg_loss + reg2_loss
            else:
                loss = tgt_loss
            loss.backward()
            parallel.allreduce_mean_grads([p for p in self.dclllayer.parameters() if p.grad is not None],
                                          local_n=input.shape[0])
            if do_train:
                self.optimizer.step()
                if self.dclllayer.output_layer:
                    self.optimizer2.step()
        else:
            tgt_loss = torch.Tensor([0])
        return output, pvoutput, pv, pvmem, tgt_loss.detach()


class DCLLClassification(DCLLBase):
    def forward(self, input, ignore_burnin=False):
        L = self.dclllayer
        record = ignore_burnin or (self.iter + 1) >= self.burnin
        fused = (record and isinstance(L, Conv2dDCLLlayer) and not L.i2h._general() and       # (torch readouts there)
                 not L.dropout_active() and                                                    # (a masked p: argmax after the mask)
                 not (getattr(L, 'build_graph', False) and torch.is_grad_enabled()))      # (not the autograd node)
        if fused:
            L.__dict__['_finish'] = {'clout': True}     # the step's finishing launch also writes the argmax recorded below
        try:
            o, p, pv, pvmem = super().forward(input)
        finally:
            fin = L.__dict__.pop('_finish', None) if fused else None
        if record:
            # kept on the device (no sync per step, unlike the reference's .cpu() at :726-728); converted on demand
            if fin is not None and 'run_readouts' in fin:
                # deferred readout tail (ConvNetwork.test): the caller runs it with the other slices' and fills this entry in
                self._clout.append(None)
                L.__dict__['_defer_sink'].append((self, fin, len(self._clout) - 1))
            elif fin is not None and fin.get('done') and fin.get('clout') is not None:
                self._clout.append(fin['clout'])
            else:
                logits = o if L.output_layer else p
                self._clout.append(ops.argmax(logits.detach()))      # k_argmax: first maximum wins, like torch.argmax
        return o, p, pv, pvmem

    def set_sequence_result(self, clout_dev, n_steps, lowhigh=None, numel=0, vote=None):
        """Install the results of a whole-sequence run as n_steps calls of forward() would have left them: the per-step
        argmax ((T,B) int32 on device) appended to `clout`, the iteration count advanced, and — `lowhigh` (n,2) int64
        device counters of the histogram steps, `numel` pv values per step — the pv statistics appended to
        `activity_hist` (reference :658-661).  `vote` ((B) int32 on device): the mode over exactly these steps as the
        vote kernel computed it (ties -> first seen, dcll_argmax_vote); accuracy() / confusion_matrix() use it as long
        as `clout` holds these steps and nothing else, instead of reading all T x B entries back and voting on the host."""
        self.iter += n_steps
        fresh = len(self._clout) == 0
        if clout_dev is not None:       # (None: steps whose argmax is not recorded — the burn-in of a learning sequence)
            self._clout.extend(clout_dev.to(torch.int64))   # like n_steps calls of forward(): appended, not replaced
        self._seq_vote = (vote, len(self._clout)) if (vote is not None and clout_dev is not None and fresh) else None
        if lowhigh is not None and self.collect_stats:
            self.activity_hist.extend((row, numel) for row in lowhigh)

    def write_stats(self, writer, label, epoch):
        super().write_stats(writer, label, epoch)
        writer.add_scalar(self.name + '/acc/' + label, self.acc, epoch)

    def _predictions(self, targets):
        """get_predictions_by_vote(self.clout, targets[-len(clout):]) (reference :735-749); the prediction comes from
        the device-side vote of a whole-sequence run when `clout` is exactly that run (set_sequence_result)."""
        sv = getattr(self, '_seq_vote', None)
        begin = len(self._clout)
        if sv is not None and sv[1] == begin and begin > 0:
            pred = sv[0].detach().cpu().numpy().astype(np.float64)
            return pred, _label_votes(targets[-begin:])
        return get_predictions_by_vote(self.clout, targets[-begin:])

    def accuracy(self, targets):
        pred, labv = self._predictions(targets)
        self.acc = float(np.mean(pred == labv))
        return self.acc

    def confusion_matrix(self, targets):
        pred, labv = self._predictions(targets)
        n = self.dclllayer.target_size
        cm = np.zeros((n, n), dtype=int)
        np.add.at(cm, (pred.astype(np.int64), labv.astype(np.int64)), 1)
        return cm
