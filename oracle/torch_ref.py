"""ORACLE (test infrastructure, not product): torch-CPU restatement of the reference's DCLL path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It issues the *same eager op sequence* as the reference so that, with the same torch build, results are
bit-identical to the imported reference (pinned by tests/golden/*.npz, see tests/test_oracle_torch.py).
It is also the "reference CPU PyTorch path" timed as `cpu_baseline` (kind "port") on the GPU node.

Reference lines restated (paths relative to the reference checkout):
  conv LIF dynamics, refractory      dcll/pytorch_libdcll.py:485-509
  conv LIF dynamics, plain           dcll/pytorch_libdcll.py:407-426
  dense LIF dynamics                 dcll/pytorch_libdcll.py:131-148, :171-195
  layer = dynamics + pool + readout  dcll/pytorch_libdcll.py:599-608, :250-255
  vote helpers                       dcll/pytorch_libdcll.py:44-61
  network chaining                   networks/__init__.py:182-185
"""
from collections import Counter

import numpy as np
import torch
import torch.nn.functional as F


def _pair(v):
    return tuple(v) if hasattr(v, "__len__") else (v, v)


def conv_lif_step(x, weight, bias, alpha, tau_m, alphas, tau_s, state, alpharp=.65, wrp=0.0,
                  stride=1, padding=0, dilation=1, groups=1, act=None, spiking=True):
    """One timestep of ContinuousConv2D / ContinuousRelativeRefractoryConv2D.

    state = (eps0, eps1) or (eps0, eps1, arp).  Returns (spikes, pv, v, new_state).
    Each line is one separately rounded fp32 op, exactly as pytorch_libdcll.py:493-503 / :415-420.
    act: the layer's activation (:419 / :500; None = nn.Sigmoid(), the default); spiking=False (plain variant only, :336-339):
    the output is pvmem itself instead of (pvmem > 0).
    """
    act = torch.sigmoid if act is None else act
    eps0 = x * tau_s + alphas * state[0]                       # :493 / :415
    eps1 = alpha * state[1] + eps0 * tau_m                     # :494 / :416
    pvmem = F.conv2d(eps1, weight, bias, stride, padding, dilation, groups)   # :495 / :417
    if wrp > 0:
        arp = alpharp * state[2]                               # :497
        v = pvmem + arp                                        # :498
        s = (v > 0).float()                                    # :499
        pv = act(v)                                            # :500
        arp = arp - s * wrp                                    # :503 (in-place there)
        return s, pv, v, (eps0, eps1, arp)
    pv = act(pvmem)                                            # :419
    s = (pvmem > 0).float() if spiking else pvmem              # :420 (output_act :336-339)
    return s, pv, pvmem, (eps0, eps1)


def dense_lif_step(x, weight, bias, alpha, tau_m, alphas, tau_s, state, alpharp=.65, wrp=0.0, act=None, spiking=True):
    """One timestep of CLLDenseModule / CLLDenseRRPModule (pytorch_libdcll.py:139-148, :179-195)."""
    act = torch.sigmoid if act is None else act
    eps0 = x * tau_s + alphas * state[0]
    eps1 = alpha * state[1] + eps0 * tau_m
    pvmem = F.linear(eps1, weight, bias)
    if wrp > 0:
        arp = alpharp * state[2]
        v = pvmem + arp
        s = (v > 0).float()
        pv = act(v)
        arp = arp - s * wrp
        return s, pv, v, (eps0, eps1, arp)
    return ((pvmem > 0).float() if spiking else pvmem), act(pvmem), pvmem, (eps0, eps1)


def max_pool(x, pooling):
    """nn.MaxPool2d(kernel=stride=pooling, padding=(pooling-1)//2)  (pytorch_libdcll.py:542-549)."""
    ph, pw = _pair(pooling)
    return F.max_pool2d(x, (ph, pw), (ph, pw), ((ph - 1) // 2, (pw - 1) // 2))


class RefConvLayer:
    """Functional twin of Conv2dDCLLlayer built from a state-dict-like mapping of tensors."""

    def __init__(self, sd, padding, pooling, wrp, alpharp=.65, output_layer=False, stride=1, dilation=1, groups=1, act=None,
                 spiking=True):
        self.w = sd["i2h.weight"]
        self.b = sd.get("i2h.bias")         # (bias=False: no such entry)
        self.stride, self.dilation, self.groups, self.act, self.spiking = stride, dilation, groups, act, spiking
        self.alpha, self.tau_m = sd["i2h.alpha"], sd["i2h.tau_m__dt"]
        self.alphas, self.tau_s = sd["i2h.alphas"], sd["i2h.tau_s__dt"]
        self.i2o_w, self.i2o_b = sd["i2o.weight"], sd["i2o.bias"]
        self.output_layer = output_layer
        if output_layer:
            self.out_w, self.out_b = sd["output_.weight"], sd["output_.bias"]
        self.padding = _pair(padding)
        self.pooling = _pair(pooling)
        self.wrp = float(wrp)
        self.alpharp = float(alpharp)
        self.state = None

    def out_hw(self, hw):
        kh, kw = self.w.shape[2:]
        return ((hw[0] + 2 * self.padding[0] - self.dilation * (kh - 1) - 1) // self.stride + 1,
                (hw[1] + 2 * self.padding[1] - self.dilation * (kw - 1) - 1) // self.stride + 1)

    def init_state(self, batch, hw):
        cin, cout = self.w.shape[1] * self.groups, self.w.shape[0]
        z = torch.zeros(batch, cin, *hw)
        st = [z, z.clone()]
        if self.wrp > 0:
            st.append(torch.zeros(batch, cout, *self.out_hw(hw)))
        self.state = tuple(st)

    def forward(self, x):
        """(output, pvoutput, pv, pvmem) like Conv2dDCLLlayer.forward (pytorch_libdcll.py:599-608)."""
        if self.state is None or self.state[0].shape[0] != x.shape[0]:
            self.init_state(x.shape[0], x.shape[2:4])
        s, pv, v, self.state = conv_lif_step(x, self.w, self.b, self.alpha, self.tau_m, self.alphas, self.tau_s,
                                             self.state, self.alpharp, self.wrp, self.stride, self.padding, self.dilation,
                                             self.groups, self.act, self.spiking)
        s, pv = max_pool(s, self.pooling), max_pool(pv, self.pooling)
        flat = pv.reshape(pv.shape[0], -1)
        p = F.linear(flat, self.i2o_w, self.i2o_b)
        o = F.linear(flat, self.out_w, self.out_b) if self.output_layer else s
        return o, p, pv, v


class RefDenseLayer:
    """Functional twin of DenseDCLLlayer (pytorch_libdcll.py:198-255)."""

    def __init__(self, sd, wrp, alpharp=.65, act=None, spiking=True):
        self.w, self.b = sd["i2h.weight"], sd.get("i2h.bias")
        self.alpha, self.tau_m = sd["i2h.alpha"], sd["i2h.tau_m__dt"]
        self.alphas, self.tau_s = sd["i2h.alphas"], sd["i2h.tau_s__dt"]
        self.i2o_w, self.i2o_b = sd["i2o.weight"], sd.get("i2o.bias")     # (bias=False strips both, :229)
        self.wrp, self.alpharp, self.act, self.spiking = float(wrp), float(alpharp), act, spiking
        self.state = None

    def forward(self, x):
        x = x.reshape(-1, self.w.shape[1])
        if self.state is None or self.state[0].shape[0] != x.shape[0]:
            z = torch.zeros(x.shape[0], self.w.shape[1])
            st = [z, z.clone()]
            if self.wrp > 0:
                st.append(torch.zeros(x.shape[0], self.w.shape[0]))
            self.state = tuple(st)
        s, pv, v, self.state = dense_lif_step(x, self.w, self.b, self.alpha, self.tau_m, self.alphas, self.tau_s,
                                              self.state, self.alpharp, self.wrp, self.act, self.spiking)
        return s, F.linear(pv, self.i2o_w, self.i2o_b), pv, v


class RefConvNetwork:
    """Chain of RefConvLayer + per-step argmax collection (ConvNetwork.test, DCLLClassification.forward)."""

    def __init__(self, layer_sds, convs, wrp, alpharp=.65):
        n = len(convs)
        self.layers = [RefConvLayer(sd, c["padding"], c["pooling"], wrp, alpharp, output_layer=(i == n - 1))
                       for i, (sd, c) in enumerate(zip(layer_sds, convs))]
        self.clout = [[] for _ in self.layers]

    def reset(self, init_states=False):
        self.clout = [[] for _ in self.layers]
        if init_states:
            for l in self.layers:
                l.state = None

    def test(self, x):
        cur = x
        outs = []
        for i, l in enumerate(self.layers):
            o, p, pv, v = l.forward(cur)
            self.clout[i].append((o if l.output_layer else p).argmax(1).numpy())
            outs.append((o, p, pv, v))
            cur = o
        return outs

    def votes(self):
        return [predictions_by_vote(c) for c in self.clout]


def predictions_by_vote(clout):
    """Mode over T of per-step argmax; ties -> first seen (Counter.most_common), pytorch_libdcll.py:44-56."""
    arr = np.array(clout).T
    return np.array([Counter(row).most_common(1)[0][0] for row in arr], dtype=np.int64)


def accuracy_by_vote(clout, labels_1h):
    """labels_1h: (T,B,C) one-hot; mode over T of its argmax (pytorch_libdcll.py:49-61)."""
    pred = predictions_by_vote(clout)
    lab = predictions_by_vote(list(np.asarray(labels_1h).argmax(axis=2)))
    return float(np.mean(pred == lab))
