"""Tensor-level wrappers over the C ABI (include/dcll_hip.h): torch tensors in HBM in, torch tensors out.

PyTorch is only the owner of device memory and streams here; all arithmetic happens in libdcll_hip.so.
"""
import ctypes
import os
import math

import numpy as np
import torch

from . import _lib
from ._lib import ACT_NONE, ACT_SIGMOID, ConvDesc, DenseDesc, IQTail, LayerOpts, check, ptr, stream_ptr


def _pair(v):
    return (int(v[0]), int(v[1])) if hasattr(v, "__len__") else (int(v), int(v))


def make_conv_desc(c_in, c_out, hw, kernel_size, padding, pooling, target, output_layer, tau_is_tensor, wrp,
                   alpharp=.65, stride=1, dilation=1, groups=1):
    (kh, kw), (pah, paw) = _pair(kernel_size), _pair(padding)
    poh, pow_ = _pair(pooling) if pooling is not None else (1, 1)
    return ConvDesc(int(c_in), int(c_out), int(hw[0]), int(hw[1]), kh, kw, pah, paw, int(stride), int(dilation),
                    int(groups), poh, pow_, int(target), int(bool(output_layer)), int(bool(tau_is_tensor)),
                    int(wrp > 0), float(alpharp), float(wrp))


def conv_lif_backward_open_multi(deferred):
    """The open backward of several layers (conv_lif_backward(..., defer=list)) in one dcll_conv_lif_backward_open_multi call:
    their dv launches as ONE launch where the layers allow it, then weight / output_ gradients layer by layer.  Fills each
    layer's out['parts'] for grad_reduce_adam.  Per layer bit-identical to conv_lif_backward(open_reduce=True)."""
    if not deferred:
        return
    if len(deferred) > _lib.BWD_MULTI_MAX:
        raise ValueError("conv_lif_backward_open_multi: at most %d layers" % _lib.BWD_MULTI_MAX)
    arr = (_lib.BwdItem * len(deferred))(*[e['item'] for e in deferred])
    check(_lib.get().dcll_conv_lif_backward_open_multi(arr, len(deferred), stream_ptr()), "dcll_conv_lif_backward_open_multi")
    for e, a in zip(deferred, arr):
        d = e['desc']
        e['out']['parts'] = dict(part=a.part, nchunk=a.nchunk, c_out=d.c_out, rowlen=d.c_in * d.kh * d.kw + 1, dW=e['dW'],
                                 db=e['db'], keep=e['scratch'])


_OUT_SHAPES = {}


def conv_out_shape(desc):
    """(conv_h, conv_w, pool_h, pool_w) of a descriptor, as the library computes them (dcll_conv_out_shape; remembered per
    geometry: the per-step paths ask several times per layer step)."""
    key = (desc.c_in, desc.c_out, desc.h, desc.w, desc.kh, desc.kw, desc.pad_h, desc.pad_w, desc.stride, desc.dilation,
           desc.groups, desc.pool_h, desc.pool_w, desc.target)      # (everything the library's descriptor check looks at)
    got = _OUT_SHAPES.get(key)
    if got is None:
        v = [ctypes.c_int32() for _ in range(4)]
        check(_lib.get().dcll_conv_out_shape(ctypes.byref(desc), *[ctypes.byref(i) for i in v]), "dcll_conv_out_shape")
        got = _OUT_SHAPES[key] = tuple(i.value for i in v)
    return got


def _f32(t, name):
    if t is not None and t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))
    return t


def _expect(t, name, dtype, shape=None, numel=None):
    """Host-side operand check before a raw pointer crosses the ABI: a kernel indexes exactly the extents its
    descriptor implies, so a wrong dtype / shape here would be an out-of-bounds access on the GPU."""
    if t is None:
        return
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError("%s must have shape %s, got %s" % (name, tuple(shape), tuple(t.shape)))
    if numel is not None and t.numel() != numel:
        raise ValueError("%s must have %d elements, got %d" % (name, numel, t.numel()))


def layer_opts(desc, q8=None, presigmoid=False):
    """dcll_layer_opts of a call as a ctypes reference (None = defaults).  q8 = (q int8 (c_out,c_in,kh,kw), scale fp32
    (c_out)): the kernels read the conv weight as int8 and convert it once, (float)q * scale[co] (include/dcll_hip.h);
    presigmoid: the sequence call writes v instead of sigmoid(v) into pv_out.  The caller keeps the tensors alive."""
    if q8 is None and not presigmoid:
        return None
    o = LayerOpts()
    if q8 is not None:
        q, scale = q8
        _expect(q, "w_q8", torch.int8, (desc.c_out, desc.c_in // desc.groups, desc.kh, desc.kw))
        _expect(scale, "w_scale", torch.float32, (desc.c_out,))
        o.w_q8, o.w_scale = ptr(q).value, ptr(scale).value
    o.pv_presigmoid = int(bool(presigmoid))
    return ctypes.byref(o)


def _check_layer_operands(desc, W, b, eps0, eps1, arp, B, tau=None, tau4=None):
    ch, cw, _, _ = conv_out_shape(desc)
    _expect(W, "W", torch.float32, (desc.c_out, desc.c_in // desc.groups, desc.kh, desc.kw))
    _expect(b, "b", torch.float32, (desc.c_out,))       # (None = bias=False: the generic kernels start their chains at 0)
    _expect(eps0, "eps0", torch.float32, (B, desc.c_in, desc.h, desc.w))
    _expect(eps1, "eps1", torch.float32, (B, desc.c_in, desc.h, desc.w))
    _expect(arp, "arp", torch.float32, (B, desc.c_out, ch, cw))
    if desc.refractory and arp is None:
        raise ValueError("refractory layer needs an arp state tensor")
    if tau is not None:
        n = desc.c_in * desc.h * desc.w if desc.tau_is_tensor else 1
        for k, t in enumerate(tau):
            _expect(t, "time constant %d" % k, torch.float32, numel=n)
    if tau4 is not None:
        _expect(tau4, "tau4", torch.float32, (4, desc.c_in))


def conv_lif_step(desc, x, W, b, alpha, tau_m, alphas, tau_s, eps0, eps1, arp, i2o_W=None, i2o_b=None,
                  out_W=None, out_b=None, want_v=True, out=None, q8=None, stacked=None, finish=None, defer_ro=False):
    """One Conv2dDCLLlayer.forward step (dcll/pytorch_libdcll.py:599-608); state tensors are updated in place.

    Returns (s_pooled, p, o, pv_pooled, v) — p / o are None when the corresponding weights are None.
    `out`: optional dict of preallocated outputs ('s', 'pv', 'v', 'p', 'o', 'scratch'; filled in when absent) — the
    learning loop reuses one set per layer instead of allocating five tensors per step.
    `q8`: (int8 weights, per-output-channel scale) — the kernels then read the conv weight as int8 (W may be None).
    `stacked`: (Wt, bias) = i2o's rows with output_'s stacked behind them (Conv2dDCLLlayer.stacked_readout): on the output
    layer both readouts then share ONE pass over pv (dcll_step_readouts).
    `finish`: dict asking dcll_step_readouts for what follows the readouts of this step — 'clout': True or an int32 (B)
    tensor to write the recorded argmax to; 'target' (B, target) + 'kind': the local-loss gradients — left in `finish` as
    'clout', 'g_p', 'g_o' (only where the fused path serves the shape: finish['done'] says so).
    `defer_ro` (needs `finish`): where the fused readout tail serves the shape, only the layer kernel is launched now; the
    readout tail is left in finish['run_readouts'] for the caller to launch later (p / o are unwritten until then) — a
    learning timestep launches all layer kernels first (layer l+1 needs layer l's spikes, not its readouts).
    """
    out = {} if out is None else out
    B = x.shape[0]
    ch, cw, ph, pw = conv_out_shape(desc)
    dev = x.device
    x = _f32(x, "x").contiguous()
    _expect(x, "x", torch.float32, (B, desc.c_in, desc.h, desc.w))
    _check_layer_operands(desc, W, b, eps0, eps1, arp, B, tau=(alpha, tau_m, alphas, tau_s))
    opts = layer_opts(desc, q8)
    K_ro = desc.c_out * ph * pw
    if i2o_W is not None:
        _expect(i2o_W, "i2o_W", torch.float32, (desc.target, K_ro))
        _expect(i2o_b, "i2o_b", torch.float32, (desc.target,))
    if desc.output_layer:
        _expect(out_W, "out_W", torch.float32, (desc.target, K_ro))
        _expect(out_b, "out_b", torch.float32, (desc.target,))
    def buf(key, shape, want=True):
        if not want:
            return None
        t = out.get(key)
        if t is None or tuple(t.shape) != tuple(shape):
            t = out[key] = torch.empty(shape, device=dev, dtype=torch.float32)
        return t
    pooled = not (desc.pool_h == 1 and desc.pool_w == 1)
    s = buf('s', (B, desc.c_out, ph, pw))
    pv = buf('pv', (B, desc.c_out, ph, pw))
    v = buf('v', (B, desc.c_out, ch, cw), want_v)
    p = buf('p', (B, desc.target), i2o_W is not None)
    o = buf('o', (B, desc.target), bool(desc.output_layer))
    scratch = buf('scratch', (2, B, desc.c_out, ch, cw), pooled)
    if finish is not None:
        finish['done'] = False
    n2 = desc.target if desc.output_layer else 0
    if (i2o_W is not None and B <= 2048 and 2 * desc.target <= 64 and
            _lib.get().dcll_step_readouts_scratch(B, K_ro, desc.target, n2) > 0 and pv.data_ptr() % 16 == 0 and
            (stacked is not None or not desc.output_layer) and
            (i2o_W if stacked is None else stacked[0]).data_ptr() % 16 == 0):
        # few rows (rows = batch): conv + neuron kernel, ONE split-K readout pass over pv for i2o (+ output_), ONE finishing
        # launch (slice sums, p / o, the recorded argmax, the local-loss gradients)
        d2 = ConvDesc.from_buffer_copy(desc)
        d2.output_layer = 0
        rc = _lib.get().dcll_conv_lif_step(
            ctypes.byref(d2), ptr(x), ptr(W), ptr(b), ptr(alpha), ptr(tau_m), ptr(alphas), ptr(tau_s),
            ptr(eps0), ptr(eps1), ptr(arp), None, None, None, None,
            ptr(s), None, None, ptr(pv), ptr(v), ptr(scratch), opts, B, stream_ptr())
        check(rc, "dcll_conv_lif_step")
        Wt, bias = (i2o_W, i2o_b) if stacked is None else stacked
        if defer_ro and finish is not None:
            finish['run_readouts'] = lambda: step_readouts(pv.reshape(B, -1), Wt, bias, desc.target, n2, p, o, scratch=out,
                                                           finish=finish)
            finish['ro_call'] = (pv.reshape(B, -1), Wt, bias, desc.target, n2, p, o, out)      # (for run_deferred_readouts)
            return s, p, o, pv, v
        step_readouts(pv.reshape(B, -1), Wt, bias, desc.target, n2, p, o, scratch=out, finish=finish)
        return s, p, o, pv, v
    if (i2o_W is not None and B <= 2048 and _lib.get().dcll_readout_splitk_scratch(B, K_ro, desc.target) > 0 and
            pv.data_ptr() % 16 == 0 and i2o_W.data_ptr() % 16 == 0 and
            (not desc.output_layer or out_W.data_ptr() % 16 == 0)):
        # few rows (rows = batch): the readouts go through readout() below, which splits K over the chip
        # (dcll_readout_splitk: 32 x B/128 MFMA workgroups on the 16x16 plane) — inside the step call they would run on
        # rows/4 x N/4 workgroups that re-read every pv row N/4 times
        d2 = ConvDesc.from_buffer_copy(desc)
        d2.output_layer = 0
        rc = _lib.get().dcll_conv_lif_step(
            ctypes.byref(d2), ptr(x), ptr(W), ptr(b), ptr(alpha), ptr(tau_m), ptr(alphas), ptr(tau_s),
            ptr(eps0), ptr(eps1), ptr(arp), None, None, None, None,
            ptr(s), None, None, ptr(pv), ptr(v), ptr(scratch), opts, B, stream_ptr())
        check(rc, "dcll_conv_lif_step")
        readout(pv.reshape(B, -1), i2o_W, i2o_b, out=p, scratch=out)
        if desc.output_layer:
            readout(pv.reshape(B, -1), out_W, out_b, out=o, scratch=out)
        return s, p, o, pv, v
    rc = _lib.get().dcll_conv_lif_step(
        ctypes.byref(desc), ptr(x), ptr(W), ptr(b), ptr(alpha), ptr(tau_m), ptr(alphas), ptr(tau_s),
        ptr(eps0), ptr(eps1), ptr(arp), ptr(i2o_W), ptr(i2o_b), ptr(out_W), ptr(out_b),
        ptr(s), ptr(p), ptr(o), ptr(pv), ptr(v), ptr(scratch), opts, B, stream_ptr())
    check(rc, "dcll_conv_lif_step")
    return s, p, o, pv, v


def conv_lif_backward(desc, eps1, v, pv_pooled, g_p, g_o, g_pv, g_v, i2o_W, want_out, out=None, open_reduce=False, defer=None):
    """Gradients of one layer step (dcll_conv_lif_backward) -> (dW, db, d_outW, d_outb).  `out`: optional dict with
    preallocated 'dW', 'db', 'd_outW', 'd_outb', 'bwd_scratch' (the learning loop writes into the parameters' .grad).
    `open_reduce`: dcll_conv_lif_backward_open — dW / db are NOT written yet; the partial rows of the weight gradient stay
    in out['bwd_scratch'] and out['parts'] describes them for grad_reduce_adam, which finishes several layers in one launch.
    `defer` (a list; open form): nothing is launched — the prepared call is appended for conv_lif_backward_open_multi."""
    B = eps1.shape[0]
    dev = eps1.device
    out = {} if out is None else out
    ch, cw, ph, pw = conv_out_shape(desc)
    K = desc.c_out * ph * pw

    def buf(key, shape, want=True):
        if not want:
            return None
        t = out.get(key)
        if t is None:
            t = out[key] = torch.empty(shape, device=dev, dtype=torch.float32)
        _expect(t, key, torch.float32, shape)
        return t
    dW = buf('dW', (desc.c_out, desc.c_in // desc.groups, desc.kh, desc.kw))
    db = buf('db', (desc.c_out,))
    d_outW = buf('d_outW', (desc.target, K), want_out)
    d_outb = buf('d_outb', (desc.target,), want_out)
    # partial-sum rows for the weight gradient: one per workgroup, up to 256 (a sample of a large plane is many 16x16
    # tile jobs, so small batches still fill the chip); the same area then holds the batch chunks of the output_ gradient
    jobs = B * max(1, (desc.h // 16) * (desc.w // 16))
    per_chunk = desc.c_out * ((desc.c_in // desc.groups) * desc.kh * desc.kw + 1)
    nchunk = min(jobs, 1024 if desc.c_in == 1 else 256)      # (first layer: 128-thread workgroups, 6 KB partial rows)
    part = nchunk * per_chunk
    if want_out and desc.target <= 32 and K % 32 != 0:
        # the output_ gradient's batch chunks (k_bwd_outgrad_part; K % 32 == 0 runs the MFMA form without them): BEHIND the
        # weight gradient's rows — the open form keeps those — and for min(B, 16) chunks, the split both forms then take
        part += min(B, 16) * desc.target * (K + 1)
    n_scratch = B * desc.c_out * ch * cw + part
    scratch = out.get('bwd_scratch')
    if scratch is None or scratch.numel() != n_scratch:
        scratch = out['bwd_scratch'] = torch.empty((n_scratch,), device=dev, dtype=torch.float32)
    c = lambda t: None if t is None else _f32(t, "grad").contiguous()
    if defer is not None:       # (open form, launched later with other layers': conv_lif_backward_open_multi)
        gp_, go_, gpv_, gv_ = c(g_p), (c(g_o) if want_out else None), c(g_pv), c(g_v)
        item = _lib.BwdItem(ctypes.pointer(desc), ptr(eps1), ptr(v), ptr(pv_pooled), ptr(gp_), ptr(go_), ptr(gpv_), ptr(gv_),
                            ptr(i2o_W), ptr(d_outW), ptr(d_outb), ptr(scratch), n_scratch, B, 0, None, 0, 0)
        defer.append(dict(item=item, out=out, desc=desc, dW=dW, db=db, scratch=scratch,
                          keep=(eps1, v, pv_pooled, gp_, go_, gpv_, gv_, i2o_W, d_outW, d_outb)))
        return dW, db, d_outW, d_outb
    if open_reduce:
        part, nchunk = ctypes.c_void_p(), ctypes.c_int32()
        rc = _lib.get().dcll_conv_lif_backward_open(
            ctypes.byref(desc), ptr(eps1), ptr(v), ptr(pv_pooled), ptr(c(g_p)), ptr(c(g_o) if want_out else None),
            ptr(c(g_pv)), ptr(c(g_v)), ptr(i2o_W), ptr(d_outW), ptr(d_outb), ptr(scratch), n_scratch, B,
            ctypes.byref(part), ctypes.byref(nchunk), stream_ptr())
        check(rc, "dcll_conv_lif_backward_open")
        out['parts'] = dict(part=part.value, nchunk=nchunk.value, c_out=desc.c_out,
                            rowlen=(desc.c_in // desc.groups) * desc.kh * desc.kw + 1, dW=dW, db=db, keep=scratch)
        return dW, db, d_outW, d_outb
    rc = _lib.get().dcll_conv_lif_backward(
        ctypes.byref(desc), ptr(eps1), ptr(v), ptr(pv_pooled), ptr(c(g_p)), ptr(c(g_o) if want_out else None),
        ptr(c(g_pv)), ptr(c(g_v)), ptr(i2o_W), ptr(dW), ptr(db), ptr(d_outW), ptr(d_outb), ptr(scratch), n_scratch, B,
        stream_ptr())
    check(rc, "dcll_conv_lif_backward")
    return dW, db, d_outW, d_outb


def dense_lif_step(desc, x, W, b, alpha, tau_m, alphas, tau_s, eps0, eps1, arp, i2o_W=None, i2o_b=None):
    """One DenseDCLLlayer.forward step (dcll/pytorch_libdcll.py:250-255)."""
    B = x.shape[0]
    dev = x.device
    x = _f32(x, "x").contiguous()
    s = torch.empty((B, desc.out_features), device=dev, dtype=torch.float32)
    pv, v = torch.empty_like(s), torch.empty_like(s)
    p = torch.empty((B, desc.target), device=dev, dtype=torch.float32) if i2o_W is not None else None
    rc = _lib.get().dcll_dense_lif_step(
        ctypes.byref(desc), ptr(x), ptr(W), ptr(b), ptr(alpha), ptr(tau_m), ptr(alphas), ptr(tau_s),
        ptr(eps0), ptr(eps1), ptr(arp), ptr(i2o_W), ptr(i2o_b), ptr(s), ptr(p), ptr(pv), ptr(v), B, stream_ptr())
    check(rc, "dcll_dense_lif_step")
    return s, p, pv, v


def dense_lif_sequence(desc, x_seq, W, b, alpha, tau_m, alphas, tau_s, eps0, eps1, arp, i2o_W=None, i2o_b=None,
                       want_s=True, want_v=False):
    """All T steps of a DenseDCLLlayer in one call (dcll_dense_lif_sequence): x_seq (T,B,in) fp32 -> (s (T,B,out) or None,
    p (T,B,target) or None, pv (T,B,out), v (T,B,out) or None); the state tensors are updated in place."""
    T, B = x_seq.shape[0], x_seq.shape[1]
    dev = x_seq.device
    x_seq = _f32(x_seq, "x").contiguous()
    _expect(x_seq, "x", torch.float32, (T, B, desc.in_features))
    _expect(W, "W", torch.float32, (desc.out_features, desc.in_features))
    _expect(eps0, "eps0", torch.float32, (B, desc.in_features))
    _expect(eps1, "eps1", torch.float32, (B, desc.in_features))
    _expect(arp, "arp", torch.float32, (B, desc.out_features))
    n = desc.in_features if desc.tau_is_tensor else 1
    for k, t in enumerate((alpha, tau_m, alphas, tau_s)):
        _expect(t, "time constant %d" % k, torch.float32, numel=n)
    if i2o_W is not None:
        _expect(i2o_W, "i2o_W", torch.float32, (desc.target, desc.out_features))
    new = lambda want, last: torch.empty((T, B, last), device=dev, dtype=torch.float32) if want else None
    s, pv, v = new(want_s, desc.out_features), new(True, desc.out_features), new(want_v, desc.out_features)
    p = new(i2o_W is not None, desc.target)
    rc = _lib.get().dcll_dense_lif_sequence(
        ctypes.byref(desc), ptr(x_seq), ptr(W), ptr(b), ptr(alpha), ptr(tau_m), ptr(alphas), ptr(tau_s), ptr(eps0),
        ptr(eps1), ptr(arp), ptr(i2o_W), ptr(i2o_b), ptr(s), ptr(p), ptr(pv), ptr(v), T, B, stream_ptr())
    check(rc, "dcll_dense_lif_sequence")
    return s, p, pv, v


def dense_lif_backward(desc, eps1, pv, g_p, g_pv, g_v, i2o_W, out=None, open_reduce=False):
    """Gradients of one DenseDCLLlayer step (dcll_dense_lif_backward) -> (dW (out,in), db (out)).  `out`: optional dict with
    preallocated 'dW', 'db', 'bwd_scratch' (the learning loop writes into the parameters' .grad).  `open_reduce`:
    dcll_dense_lif_backward_open — dW / db are not written yet, out['parts'] describes the partial rows for grad_reduce_adam
    (the same record a conv layer's open backward leaves)."""
    B = eps1.shape[0]
    dev = eps1.device
    out = {} if out is None else out
    nin, nout = desc.in_features, desc.out_features
    _expect(eps1, "eps1", torch.float32, (B, nin))
    _expect(pv, "pv", torch.float32, (B, nout))
    if g_p is not None:
        _expect(g_p, "g_p", torch.float32, (B, desc.target))
        _expect(i2o_W, "i2o_W", torch.float32, (desc.target, nout))
    for name, t in (("g_pv", g_pv), ("g_v", g_v)):
        if t is not None:
            _expect(t, name, torch.float32, (B, nout))

    def buf(key, shape):
        t = out.get(key)
        if t is None:
            t = out[key] = torch.empty(shape, device=dev, dtype=torch.float32)
        _expect(t, key, torch.float32, shape)
        return t
    dW, db = buf('dW', (nout, nin)), buf('db', (nout,))
    n_scratch = B * nout + min(64, (B + 1) // 2) * nout * (nin + 1)
    scratch = out.get('bwd_scratch')
    if scratch is None or scratch.numel() != n_scratch:
        scratch = out['bwd_scratch'] = torch.empty((n_scratch,), device=dev, dtype=torch.float32)
    c = lambda t: None if t is None else _f32(t, "grad").contiguous()
    if open_reduce:
        part, nchunk = ctypes.c_void_p(), ctypes.c_int32()
        rc = _lib.get().dcll_dense_lif_backward_open(
            ctypes.byref(desc), ptr(eps1), ptr(pv.contiguous()), ptr(c(g_p)), ptr(c(g_pv)), ptr(c(g_v)), ptr(i2o_W), ptr(scratch),
            n_scratch, B, ctypes.byref(part), ctypes.byref(nchunk), stream_ptr())
        check(rc, "dcll_dense_lif_backward_open")
        out['parts'] = dict(part=part.value, nchunk=nchunk.value, c_out=nout, rowlen=nin + 1, dW=dW, db=db, keep=scratch)
        return dW, db
    rc = _lib.get().dcll_dense_lif_backward(
        ctypes.byref(desc), ptr(eps1), ptr(pv.contiguous()), ptr(c(g_p)), ptr(c(g_pv)), ptr(c(g_v)), ptr(i2o_W), ptr(dW), ptr(db),
        ptr(scratch), n_scratch, B, stream_ptr())
    check(rc, "dcll_dense_lif_backward")
    return dW, db


def permute_readout(Wt):
    """(N, 8192) readout matrix -> the fused epilogue's layout (dcll_permute_readout)."""
    Wt = Wt.contiguous()
    N, K = Wt.shape
    if K != 8192:
        raise ValueError("permute_readout: expected (N, 32*16*16) weights")
    Wp = torch.empty(N * K, device=Wt.device, dtype=torch.float32)
    check(_lib.get().dcll_permute_readout(ptr(Wt), ptr(Wp), N, stream_ptr()), "dcll_permute_readout")
    return Wp


def _seq_extras(desc, B, T, dev, out, lowhigh_iter0):
    """(state_scratch, pv_lowhigh counters, iter0) of a sequence call.  Planes other than 16x16 run the tiled kernels,
    which read a snapshot of the initial state (halo reads must not see a neighbour tile's final state): the scratch
    comes from `out['state_scratch']` or is allocated here.  lowhigh_iter0 = the slice's iteration count before the
    call turns the pv statistics on (None = off)."""
    scratch = None
    if (desc.h, desc.w) != (16, 16) and (desc.kh, desc.kw) == (7, 7):
        n = 2 * B * desc.c_in * desc.h * desc.w
        scratch = out.get("state_scratch")
        if scratch is None or scratch.numel() < n:
            scratch = torch.empty((n,), device=dev, dtype=torch.float32)
        _expect(scratch, "state_scratch", torch.float32)
    counts, iter0 = None, 0
    if lowhigh_iter0 is not None:
        iter0 = int(lowhigh_iter0)
        counts = torch.zeros((pv_lowhigh_steps(iter0, T), 2), device=dev, dtype=torch.int64)
    return scratch, counts, iter0


def pv_lowhigh_steps(iter0, T):
    """Number of histogram steps (1-based iteration count % 20 == 0, reference :658-661) among T steps after iter0."""
    return int(_lib.get().dcll_pv_lowhigh_steps(int(iter0), int(T)))


def pv_lowhigh(pv, T, iter0, presigmoid=False):
    """pv (T, ...) fp32 -> int64 (n, 2): per histogram step the counts of pv in the first / last of the reference's 19
    bins over [0, 1] (dcll_pv_lowhigh).  presigmoid: the buffer holds v, sigmoid(v) is what is counted."""
    _expect(pv, "pv", torch.float32)
    pv = pv.contiguous()
    counts = torch.zeros((pv_lowhigh_steps(iter0, T), 2), device=pv.device, dtype=torch.int64)
    check(_lib.get().dcll_pv_lowhigh_act(ptr(pv), pv.numel() // max(T, 1), T, int(iter0), ptr(counts),
                                         ACT_SIGMOID if presigmoid else ACT_NONE, stream_ptr()), "dcll_pv_lowhigh")
    return counts


def conv_lif_sequence(desc, spk_in, W, b, tau4, eps0, eps1, arp, T, B, want_spikes=True, want_pv=True,
                      want_v=False, out=None, ro_Wp=None, ro_b=None, lowhigh_iter0=None, q8=None, presigmoid=False):
    """All T steps of one 32->32 layer in one launch (k_lif_seq_c32). spk_in: (T,B,32,8) int32 packed.
    With ro_Wp / ro_b the local readout(s) are fused: returns (spk, pv, v, logits (T,B,n_ro)).
    With lowhigh_iter0 the pv statistics of the histogram steps are returned in out['lowhigh'] ((n,2) int64).
    q8 / presigmoid: dcll_layer_opts (int8 weights; pv holds v, the readout applies the sigmoid: readout_act)."""
    dev = b.device
    out = {} if out is None else out
    words = desc.h * desc.w // 32
    ch, cw, ph, pw = conv_out_shape(desc)          # pooling layers (the (1,3) / pool (1,2) geometry) emit POOLED maps
    owords = ph * pw // 32
    _expect(spk_in, "spk_in", torch.int32, (T, B, desc.c_in, words))
    _check_layer_operands(desc, W, b, eps0, eps1, arp, B, tau4=tau4)
    spk = out.get("spk") if want_spikes else None
    if want_spikes and spk is None:
        spk = torch.empty((T, B, desc.c_out, owords), device=dev, dtype=torch.int32)
    pv = out.get("pv") if want_pv else None
    if want_pv and pv is None:
        pv = torch.empty((T, B, desc.c_out, ph, pw), device=dev, dtype=torch.float32)
    _expect(spk, "spk_out", torch.int32, (T, B, desc.c_out, owords))
    _expect(pv, "pv_out", torch.float32, (T, B, desc.c_out, ph, pw))
    v = torch.empty((T, B, desc.c_out, ch, cw), device=dev, dtype=torch.float32) if want_v else None
    n_ro, logits = 0, None
    if ro_Wp is not None:
        n_ro = ro_b.numel()
        _expect(ro_Wp, "ro_Wp", torch.float32, numel=n_ro * desc.c_out * desc.h * desc.w)
        logits = out.get("ro")
        if logits is None:
            logits = torch.empty((T, B, n_ro), device=dev, dtype=torch.float32)
        _expect(logits, "ro_out", torch.float32, (T, B, n_ro))
    scratch, counts, iter0 = _seq_extras(desc, B, T, dev, out, lowhigh_iter0)
    rc = _lib.get().dcll_conv_lif_sequence(ctypes.byref(desc), ptr(spk_in), ptr(W), ptr(b), ptr(tau4), ptr(eps0),
                                           ptr(eps1), ptr(arp), ptr(spk), ptr(pv), ptr(v), ptr(ro_Wp), ptr(ro_b),
                                           ptr(logits), n_ro, ptr(scratch), ptr(counts), iter0,
                                           layer_opts(desc, q8, presigmoid), T, B, stream_ptr())
    check(rc, "dcll_conv_lif_sequence")
    if counts is not None:
        out["lowhigh"] = counts
    if ro_Wp is not None:
        return spk, pv, v, logits
    return spk, pv, v


def conv_lif_sequence_cells(desc, cells, W, b, tau4, eps0, eps1, arp, T, B, want_spikes=True, want_pv=True,
                            want_v=False, out=None, lowhigh_iter0=None, q8=None, presigmoid=False):
    """All T steps of the first layer (c_in = 1) from cell indices (T,B) int32 (k_lif_seq_c1)."""
    dev = b.device
    out = {} if out is None else out
    ch, cw, ph, pw = conv_out_shape(desc)          # pooling layers emit POOLED maps
    owords = ph * pw // 32
    _expect(cells, "cells", torch.int32, (T, B))
    _check_layer_operands(desc, W, b, eps0, eps1, arp, B, tau4=tau4)
    spk = out.get("spk") if want_spikes else None
    if want_spikes and spk is None:
        spk = torch.empty((T, B, desc.c_out, owords), device=dev, dtype=torch.int32)
    pv = out.get("pv") if want_pv else None
    if want_pv and pv is None:
        pv = torch.empty((T, B, desc.c_out, ph, pw), device=dev, dtype=torch.float32)
    _expect(spk, "spk_out", torch.int32, (T, B, desc.c_out, owords))
    _expect(pv, "pv_out", torch.float32, (T, B, desc.c_out, ph, pw))
    v = torch.empty((T, B, desc.c_out, ch, cw), device=dev, dtype=torch.float32) if want_v else None
    scratch, counts, iter0 = _seq_extras(desc, B, T, dev, out, lowhigh_iter0)
    rc = _lib.get().dcll_conv_lif_sequence_cells(ctypes.byref(desc), ptr(cells), ptr(W), ptr(b), ptr(tau4),
                                                 ptr(eps0), ptr(eps1), ptr(arp), ptr(spk), ptr(pv), ptr(v),
                                                 ptr(scratch), ptr(counts), iter0, layer_opts(desc, q8, presigmoid),
                                                 T, B, stream_ptr())
    check(rc, "dcll_conv_lif_sequence_cells")
    if counts is not None:
        out["lowhigh"] = counts
    return spk, pv, v


def iq_tail(tail, w, h, B):
    """dcll_iq_tail of a call as a ctypes reference (None = one threshold table).  tail = (thr_i_tail (w-1), thr_q_tail
    (h-1), mask (B) uint8 on the device): the samples the mask marks are quantised with the scalar-path thresholds
    (include/dcll_hip.h; data/utils.py IQEncoder).  The caller keeps the tensors alive."""
    if tail is None:
        return None
    ti, tq, mask = tail
    _expect(ti, "thr_i_tail", torch.float32, (w - 1,))
    _expect(tq, "thr_q_tail", torch.float32, (h - 1,))
    _expect(mask, "tail_mask", torch.uint8, (B,))
    t = IQTail()
    t.thr_i_tail, t.thr_q_tail, t.tail_mask = ptr(ti).value, ptr(tq).value, ptr(mask).value
    return ctypes.byref(t)


def conv_lif_sequence_iq(desc, iq, thr_i, thr_q, t0, W, b, tau4, eps0, eps1, arp, T, B, want_spikes=True,
                         want_pv=True, want_v=False, out=None, lowhigh_iter0=None, q8=None, presigmoid=False, tail=None):
    """First layer from the raw IQ window (B,2,L): iq2spiketrain's quantisation fused into k_lif_seq_c1."""
    dev = b.device
    out = {} if out is None else out
    iq = iq.reshape(B, 2, -1).contiguous()
    L = iq.shape[-1]
    words = desc.h * desc.w // 32
    _expect(iq, "iq", torch.float32)
    _expect(thr_i, "thr_i", torch.float32, (desc.w - 1,))
    _expect(thr_q, "thr_q", torch.float32, (desc.h - 1,))
    _check_layer_operands(desc, W, b, eps0, eps1, arp, B, tau4=tau4)
    spk = out.get("spk") if want_spikes else None
    if want_spikes and spk is None:
        spk = torch.empty((T, B, desc.c_out, words), device=dev, dtype=torch.int32)
    pv = out.get("pv") if want_pv else None
    if want_pv and pv is None:
        pv = torch.empty((T, B, desc.c_out, desc.h, desc.w), device=dev, dtype=torch.float32)
    v = torch.empty((T, B, desc.c_out, desc.h, desc.w), device=dev, dtype=torch.float32) if want_v else None
    scratch, counts, iter0 = _seq_extras(desc, B, T, dev, out, lowhigh_iter0)
    rc = _lib.get().dcll_conv_lif_sequence_iq(ctypes.byref(desc), ptr(iq), ptr(thr_i), ptr(thr_q),
                                              iq_tail(tail, desc.w, desc.h, B), L, t0, ptr(W), ptr(b),
                                              ptr(tau4), ptr(eps0), ptr(eps1), ptr(arp), ptr(spk), ptr(pv), ptr(v),
                                              ptr(scratch), ptr(counts), iter0, layer_opts(desc, q8, presigmoid), T, B,
                                              stream_ptr())
    check(rc, "dcll_conv_lif_sequence_iq")
    if counts is not None:
        out["lowhigh"] = counts
    return spk, pv, v


READOUT_AUTO, READOUT_CORESIDENT, READOUT_LDS, READOUT_T16 = 0, 1, 2, 3       # dcll_readout_mode (include/dcll_hip.h)


def _step_readouts_prepare(pv2d, Wt, bias, N1, N2, p, o, scratch=None, finish=None):
    """Argument checks and buffers of one dcll_step_readouts call -> (dcll_step_ro item, tensors it points to, what `finish`
    learns once the call is enqueued)."""
    rows, K = pv2d.shape
    _expect(Wt, "Wt", torch.float32, (N1 + N2, K))
    _expect(bias, "bias", torch.float32, (N1 + N2,))
    _expect(p, "p", torch.float32, (rows, N1))
    if N2:
        _expect(o, "o", torch.float32, (rows, N2))
    lib = _lib.get()
    scratch = {} if scratch is None else scratch
    need = lib.dcll_step_readouts_scratch(rows, K, N1, N2)
    area = scratch.get('step_ro')
    if area is None or area.numel() < need or area.device != pv2d.device:
        area = scratch['step_ro'] = torch.empty((need,), device=pv2d.device, dtype=torch.float32)
    clout = target = g_p = g_o = None
    kind = 0
    if finish is not None:
        cl = finish.get('clout')
        if cl is True:
            cl = torch.empty((rows,), device=pv2d.device, dtype=torch.int32)
        if cl is not None and cl is not False:
            _expect(cl, "clout", torch.int32, (rows,))
            clout = cl
        target = finish.get('target')
        if target is not None:
            _expect(target, "target", torch.float32, (rows, N1))
            target = target.contiguous()
            kind = int(finish['kind'])

            def gbuf(key, n):
                t = scratch.get(key)
                if t is None or tuple(t.shape) != (rows, n) or t.device != pv2d.device:
                    t = scratch[key] = torch.empty((rows, n), device=pv2d.device, dtype=torch.float32)
                return t
            g_p = gbuf('g_p', N1)
            g_o = gbuf('g_o', N2) if N2 else None
    item = _lib.StepRo(ptr(pv2d), ptr(Wt), ptr(bias), ptr(area), need, rows, K, N1, N2, kind, ptr(p), ptr(o), ptr(clout),
                       ptr(target), ptr(g_p), ptr(g_o), 0)
    return item, (pv2d, Wt, bias, area, p, o, clout, target, g_p, g_o), dict(done=True, clout=clout, g_p=g_p, g_o=g_o)


def step_readouts(pv2d, Wt, bias, N1, N2, p, o, scratch=None, finish=None):
    """dcll_step_readouts: the readouts of one layer step (rows = batch) against the N1 + N2 stacked rows Wt / bias in one
    split-K pass + one finishing launch that writes p (rows, N1), o (rows, N2) and — `finish` (see conv_lif_step) — the
    recorded argmax and the local-loss gradients.  `scratch`: dict keeping the partial-sum area ('step_ro') and the
    gradient buffers between calls."""
    it, _keep, res = _step_readouts_prepare(pv2d, Wt, bias, N1, N2, p, o, scratch, finish)
    check(_lib.get().dcll_step_readouts(it.pv, it.Wt, it.bias, it.scratch, it.scratch_floats, it.rows, it.K, it.N1, it.N2, it.p,
                                        it.o, it.clout, it.target, it.g_p, it.g_o, it.kind, stream_ptr()), "dcll_step_readouts")
    if finish is not None:
        finish.update(res)
    return p, o


def run_deferred_readouts(finishes):
    """The deferred readout tails of several layer steps (conv_lif_step(defer_ro=True) left them in finish['run_readouts'] /
    finish['ro_call']): ONE dcll_step_readouts_multi call — two launches for the whole timestep — where there are several of one
    kind (all with a target = learning steps, or none), else one dcll_step_readouts each.  Results are bit-identical."""
    pend = [f for f in finishes if f is not None and 'run_readouts' in f]
    if not pend:
        return
    learn = [f.get('target') is not None for f in pend]
    if (len(pend) < 2 or len(pend) > _lib.STEP_RO_MAX or any('ro_call' not in f for f in pend) or
            any(l != learn[0] for l in learn) or os.environ.get('DCLL_STEP_RO_MULTI', '1') == '0'):      # ('0': the control)
        for f in pend:
            f.pop('ro_call', None)
            f.pop('run_readouts')()
        return
    prepared = []
    for f in pend:
        f.pop('run_readouts')
        prepared.append(_step_readouts_prepare(*f.pop('ro_call'), finish=f))
    arr = (_lib.StepRo * len(prepared))(*[it for it, _, _ in prepared])
    check(_lib.get().dcll_step_readouts_multi(arr, len(prepared), stream_ptr()), "dcll_step_readouts_multi")
    for f, (_, _keep, res) in zip(pend, prepared):
        f.update(res)


def readout(pv2d, Wt, bias, out=None, mode=READOUT_AUTO, scratch=None):
    """out[r,n] = sum_k pv2d[r,k] Wt[n,k] + bias[n]  (i2o / output_), fp32 MFMA.  `mode`: kernel form, see
    dcll_readout_mode — READOUT_CORESIDENT is the LDS-free <= 64-VGPR form that fits beside a resident sequence kernel.
    `scratch`: optional dict in which the split-K partial-sum area of few-row calls is kept between calls."""
    rows, K = pv2d.shape
    N = Wt.shape[0]
    if Wt.shape[1] != K:
        raise ValueError("readout: K mismatch %d vs %d" % (Wt.shape[1], K))
    _expect(pv2d, "pv", torch.float32)
    _expect(Wt, "Wt", torch.float32)
    _expect(bias, "bias", torch.float32, (N,))
    if out is None:
        out = torch.empty((rows, N), device=pv2d.device, dtype=torch.float32)
    _expect(out, "out", torch.float32, (rows, N))
    lib = _lib.get()
    need = lib.dcll_readout_splitk_scratch(rows, K, N) if ((rows <= 2048 or K >= 65536) and mode == READOUT_AUTO) else 0
    if need > 0 and pv2d.data_ptr() % 16 == 0 and Wt.data_ptr() % 16 == 0:
        # few rows of a long K (per-step calls on a large plane): K split over the workgroups, partials in scratch
        area = None if scratch is None else scratch.get('splitk')
        if area is None or area.numel() < need:
            area = torch.empty((need,), device=pv2d.device, dtype=torch.float32)
            if scratch is not None:
                scratch['splitk'] = area
        check(lib.dcll_readout_splitk(ptr(pv2d), ptr(Wt), ptr(bias), ptr(out), ptr(area), need, rows, K, N,
                                      stream_ptr()), "dcll_readout_splitk")
        return out
    if mode != READOUT_AUTO:
        check(lib.dcll_readout_mode(ptr(pv2d), ptr(Wt), ptr(bias), ptr(out), rows, K, N, int(mode), stream_ptr()),
              "dcll_readout_mode")
        return out
    check(lib.dcll_readout(ptr(pv2d), ptr(Wt), ptr(bias), ptr(out), rows, K, N, stream_ptr()), "dcll_readout")
    return out


def readout_act_supported(pv2d, Wt):
    """True if dcll_readout_act serves this shape (the LDS-staged 16x16x4 kernel: K % 32 == 0 — K % 256 == 0 when it is
    split over K —, N <= 64, 16-byte aligned operands)."""
    K, N = pv2d.shape[1], Wt.shape[0]
    return (K % 32 == 0 and (K < 65536 or K % 256 == 0) and N <= 64 and pv2d.data_ptr() % 16 == 0 and
            Wt.data_ptr() % 16 == 0 and pv2d.is_contiguous() and Wt.is_contiguous())


def readout_act(pv2d, Wt, bias, out=None, presigmoid=False, scratch=None):
    """The readout of the whole-sequence path (dcll_readout_act): out[r,n] = sum_k act(pv2d[r,k]) Wt[n,k] + bias[n] with
    act = sigmoid when `presigmoid` (the layer kernel wrote v).  The kernel form depends on K only — a row's logits do
    not depend on the row count, i.e. on how a batch is chunked.  `scratch`: dict keeping the split-K area ('act_splitk')."""
    rows, K = pv2d.shape
    N = Wt.shape[0]
    if Wt.shape[1] != K:
        raise ValueError("readout: K mismatch %d vs %d" % (Wt.shape[1], K))
    _expect(pv2d, "pv", torch.float32)
    _expect(Wt, "Wt", torch.float32)
    _expect(bias, "bias", torch.float32, (N,))
    if out is None:
        out = torch.empty((rows, N), device=pv2d.device, dtype=torch.float32)
    _expect(out, "out", torch.float32, (rows, N))
    lib = _lib.get()
    need = lib.dcll_readout_act_scratch(rows, K, N)
    area = None
    if need > 0:
        area = None if scratch is None else scratch.get('act_splitk')
        if area is None or area.numel() < need:
            area = torch.empty((need,), device=pv2d.device, dtype=torch.float32)
            if scratch is not None:
                scratch['act_splitk'] = area
    check(lib.dcll_readout_act(ptr(pv2d), ptr(Wt), ptr(bias), ptr(out), ptr(area), need, rows, K, N,
                               ACT_SIGMOID if presigmoid else ACT_NONE, stream_ptr()), "dcll_readout_act")
    return out


def vote_tallies(votes, labels, n_classes):
    """dcll_vote_tallies: per layer the confusion matrix [pred][label], the correct count and the vote count of a batch as one
    (L, n*n + 2) int64 tensor (the form parallel.py all-reduces) — one launch.  votes: list of (B) int32 device tensors;
    labels (B) int64 on the same device."""
    L, B = len(votes), labels.shape[0]
    for v in votes:
        _expect(v, "vote", torch.int32, (B,))
    _expect(labels, "labels", torch.int64, (B,))
    out = torch.empty((L, n_classes * n_classes + 2), device=labels.device, dtype=torch.int64)
    arr = (ctypes.c_void_p * L)(*[v.data_ptr() for v in votes])
    check(_lib.get().dcll_vote_tallies(arr, L, ptr(labels), ptr(out), B, n_classes, stream_ptr()), "dcll_vote_tallies")
    return out


def argmax_vote(logits, t_begin=0, want_vote=True):
    """logits (T,B,N) -> clout (T,B) int32, vote (B) int32."""
    T, B, N = logits.shape
    _expect(logits, "logits", torch.float32)
    clout = torch.empty((T, B), device=logits.device, dtype=torch.int32)
    vote = torch.empty((B,), device=logits.device, dtype=torch.int32) if want_vote else None
    check(_lib.get().dcll_argmax_vote(ptr(logits), ptr(clout), ptr(vote), T, B, N, t_begin, stream_ptr()),
          "dcll_argmax_vote")
    return clout, vote


def argmax(logits):
    """logits (rows, N) fp32 -> (rows,) int32, first maximum wins like torch.argmax (k_argmax; DCLLClassification.forward
    :724-728 per step)."""
    rows, N = logits.shape
    _expect(logits, "logits", torch.float32)
    logits = logits.contiguous()
    out = torch.empty((rows,), device=logits.device, dtype=torch.int32)
    check(_lib.get().dcll_argmax_vote(ptr(logits), ptr(out), None, 1, rows, N, 0, stream_ptr()), "dcll_argmax_vote")
    return out


def iq_encode(iq, thr_i, thr_q, t0, T, w, h, tail=None):
    """iq (B,2,L) fp32 -> cells (T,B) int32 = q*w + i.  tail: see iq_tail."""
    B, two, L = iq.shape
    assert two == 2
    _expect(iq, "iq", torch.float32)
    _expect(thr_i, "thr_i", torch.float32, (w - 1,))
    _expect(thr_q, "thr_q", torch.float32, (h - 1,))
    cells = torch.empty((T, B), device=iq.device, dtype=torch.int32)
    check(_lib.get().dcll_iq_encode(ptr(iq), ptr(thr_i), ptr(thr_q), iq_tail(tail, w, h, B), ptr(cells), B, L, t0, T, w, h,
                                    stream_ptr()), "dcll_iq_encode")
    return cells


def unpack_spikes(packed):
    """(..., nw) int32 -> (..., nw*32) float32"""
    packed = packed.contiguous()
    dense = torch.empty(packed.shape[:-1] + (packed.shape[-1] * 32,), device=packed.device, dtype=torch.float32)
    check(_lib.get().dcll_unpack_spikes(ptr(packed), ptr(dense), packed.numel(), stream_ptr()), "dcll_unpack_spikes")
    return dense


def pack_spikes(dense):
    """(..., n) float32 with n % 32 == 0 -> (..., n/32) int32"""
    dense = dense.contiguous()
    if dense.shape[-1] % 32:
        raise ValueError("pack_spikes: last dimension must be a multiple of 32")
    packed = torch.empty(dense.shape[:-1] + (dense.shape[-1] // 32,), device=dense.device, dtype=torch.int32)
    check(_lib.get().dcll_pack_spikes(ptr(dense), ptr(packed), packed.numel(), stream_ptr()), "dcll_pack_spikes")
    return packed


LOSS_KINDS = {"SmoothL1Loss": _lib.LOSS_SMOOTH_L1, "MSELoss": _lib.LOSS_MSE}


def local_loss_grad(p, o, target, kind, out=None, want_loss=True, want_clout=False, clout_out=None):
    """Gradient and value of the local losses with mean reduction (dcll_local_loss_grad): crit(p, target) [+
    crit(o, target)] -> (g_p, g_o or None, loss (1,) or None[, clout (B) int32 = argmax of o, or of p without o]).
    kind: LOSS_KINDS[...].  clout_out: where to write the argmax (else a fresh tensor, which the caller keeps)."""
    out = {} if out is None else out
    B, N = p.shape
    _expect(p, "p", torch.float32)
    _expect(target, "target", torch.float32, (B, N))
    _expect(o, "o", torch.float32, (B, N))

    def buf(key, shape, want=True):
        if not want:
            return None
        t = out.get(key)
        if t is None or tuple(t.shape) != tuple(shape):
            t = out[key] = torch.empty(shape, device=p.device, dtype=torch.float32)
        return t
    g_p, g_o, loss = buf('g_p', (B, N)), buf('g_o', (B, N), o is not None), buf('loss', (1,), want_loss)
    clout = None
    if want_clout:
        clout = torch.empty((B,), device=p.device, dtype=torch.int32) if clout_out is None else clout_out
        _expect(clout, "clout", torch.int32, (B,))
    check(_lib.get().dcll_local_loss_grad(ptr(p.contiguous()), ptr(None if o is None else o.contiguous()),
                                          ptr(target.contiguous()), ptr(g_p), ptr(g_o), ptr(loss), ptr(clout), B, N,
                                          int(kind), stream_ptr()), "dcll_local_loss_grad")
    if want_clout:
        return g_p, g_o, loss, clout
    return g_p, g_o, loss


def adam_dyn_values(tensors):
    """Per tensor (lr, 1 / (1 - beta1^step), 1 / sqrt(1 - beta2^step)) as a flat float list — the `dyn` of adam_step,
    computed exactly as dcll_adam_step does on the host (float64 arithmetic on the float32 betas of the struct)."""
    out = []
    for t in tensors:
        b1, b2, step = float(np.float32(t["beta1"])), float(np.float32(t["beta2"])), int(t["step"])
        out += [float(t["lr"]), 1.0 / (1.0 - math.pow(b1, step)), 1.0 / math.sqrt(1.0 - math.pow(b2, step))]
    return out


def adam_step(tensors, dyn=None):
    """torch.optim.Adam's update over several tensors — of one or several optimizers — in one launch (dcll_adam_step).
    tensors: list of dicts with param, grad, exp_avg, exp_avg_sq (tensors) and lr, weight_decay, beta1, beta2, eps, step
    (1-based count of this update).  More than 8 tensors are split into several launches.
    dyn: optional device tensor (3 floats per tensor, adam_dyn_values): lr and the bias corrections are then read on the
    device at execution time (dcll_adam_step_dyn) — the form that can be captured in a graph and replayed."""
    lib = _lib.get()
    if dyn is not None:
        _expect(dyn, "dyn", torch.float32, numel=3 * len(tensors))
    for k0 in range(0, len(tensors), _lib.ADAM_MAX_TENSORS):
        part = tensors[k0:k0 + _lib.ADAM_MAX_TENSORS]
        arr = (_lib.AdamTensor * len(part))()
        for a, t in zip(arr, part):
            prm = t["param"]
            if not prm.is_cuda or not prm.is_contiguous():
                raise _lib.DCLLHipError("adam_step: parameters must be contiguous device tensors")
            for key in ("param", "grad", "exp_avg", "exp_avg_sq"):
                _expect(t[key], key, torch.float32, numel=prm.numel())
            a.param, a.grad = prm.data_ptr(), ptr(t["grad"]).value
            a.exp_avg, a.exp_avg_sq = ptr(t["exp_avg"]).value, ptr(t["exp_avg_sq"]).value
            a.n, a.step = prm.numel(), int(t["step"])
            a.lr, a.weight_decay = float(t["lr"]), float(t["weight_decay"])
            a.beta1, a.beta2, a.eps = float(t["beta1"]), float(t["beta2"]), float(t["eps"])
        if dyn is not None:
            check(lib.dcll_adam_step_dyn(arr, len(part), ptr(dyn[3 * k0:3 * (k0 + len(part))]), stream_ptr()),
                  "dcll_adam_step_dyn")
        else:
            check(lib.dcll_adam_step(arr, len(part), stream_ptr()), "dcll_adam_step")


def _adam_array(tensors):
    arr = (_lib.AdamTensor * max(1, len(tensors)))()
    for a, t in zip(arr, tensors):
        prm = t["param"]
        if not prm.is_cuda or not prm.is_contiguous():
            raise _lib.DCLLHipError("adam: parameters must be contiguous device tensors")
        for key in ("param", "grad", "exp_avg", "exp_avg_sq"):
            _expect(t[key], key, torch.float32, numel=prm.numel())
        a.param, a.grad = prm.data_ptr(), ptr(t["grad"]).value
        a.exp_avg, a.exp_avg_sq = ptr(t["exp_avg"]).value, ptr(t["exp_avg_sq"]).value
        a.n, a.step = prm.numel(), int(t["step"])
        a.lr, a.weight_decay = float(t["lr"]), float(t["weight_decay"])
        a.beta1, a.beta2, a.eps = float(t["beta1"]), float(t["beta2"]), float(t["eps"])
    return arr


def grad_reduce_adam(layers, tensors, dyn=None):
    """The end of a learning timestep in one launch (dcll_grad_reduce_adam): `layers` = the out['parts'] of
    conv_lif_backward(open_reduce=True), each with 'adam_w' / 'adam_b' = index into `tensors` (adam_step's dicts) of the
    Adam entry of its weight / bias (-1: reduce only); the other tensors get the plain update.  dyn as in adam_step."""
    if len(layers) > _lib.REDUCE_MAX_LAYERS or len(tensors) > _lib.ADAM_MAX_TENSORS:
        raise ValueError("grad_reduce_adam: at most %d layers and %d tensors" % (_lib.REDUCE_MAX_LAYERS, _lib.ADAM_MAX_TENSORS))
    if dyn is not None:
        _expect(dyn, "dyn", torch.float32, numel=3 * len(tensors))
    larr = (_lib.GradParts * max(1, len(layers)))()
    for a, L in zip(larr, layers):
        _expect(L["dW"], "dW", torch.float32, numel=L["c_out"] * (L["rowlen"] - 1))
        _expect(L["db"], "db", torch.float32, numel=L["c_out"])
        a.part, a.dW, a.db = L["part"], ptr(L["dW"]).value, ptr(L["db"]).value
        a.rowlen, a.nchunk, a.c_out = L["rowlen"], L["nchunk"], L["c_out"]
        a.adam_w, a.adam_b = int(L.get("adam_w", -1)), int(L.get("adam_b", -1))
        for idx, key in ((a.adam_w, "dW"), (a.adam_b, "db")):
            if idx >= 0 and tensors[idx]["grad"].data_ptr() != L[key].data_ptr():
                raise ValueError("grad_reduce_adam: tensors[%d]['grad'] is not the layer's %s" % (idx, key))
    check(_lib.get().dcll_grad_reduce_adam(larr, len(layers), _adam_array(tensors), len(tensors), ptr(dyn), stream_ptr()),
          "dcll_grad_reduce_adam")


def cells_to_planes(cells, hw):
    """cells (...) int32 on device -> one-hot planes (..., hw) fp32 (dcll_cells_to_planes): the dense per-step input that
    iq2spiketrain builds on the host (data/utils.py:81-82)."""
    _expect(cells, "cells", torch.int32)
    cells = cells.contiguous()
    planes = torch.empty(tuple(cells.shape) + (hw,), device=cells.device, dtype=torch.float32)
    check(_lib.get().dcll_cells_to_planes(ptr(cells), ptr(planes), cells.numel(), int(hw), stream_ptr()),
          "dcll_cells_to_planes")
    return planes


class kernel_trace:
    """`with ops.kernel_trace() as tr: ...; tr.names` — the kernels the calls inside dispatched, in launch order
    (dcll_kernel_trace, ABI 5; this thread's calls only).  Diagnostics for tests and profiles: the library picks a kernel
    by geometry and batch, and this says which one a call took."""

    def __enter__(self):
        self.names = []
        _lib.check(_lib.get().dcll_kernel_trace(1), "dcll_kernel_trace")
        return self

    def __exit__(self, *exc):
        import ctypes
        lib = _lib.get()
        lib.dcll_kernel_trace(0)
        need = int(lib.dcll_kernel_trace_read(None, 0))
        buf = ctypes.create_string_buffer(need)
        lib.dcll_kernel_trace_read(buf, need)
        self.names = [n for n in buf.value.decode("ascii", "replace").split("\n") if n]
        return False

    def count(self, name):
        return sum(1 for n in self.names if n == name)
