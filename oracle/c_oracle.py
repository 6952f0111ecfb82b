"""ctypes binding of oracle/dcll_oracle.c (TEST INFRASTRUCTURE, see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdcll_oracle.so")


class ConvDesc(ctypes.Structure):
    """Mirror of dcll_conv_desc (include/dcll_hip.h)."""
    _fields_ = [(n, ctypes.c_int32) for n in
                ("c_in", "c_out", "h", "w", "kh", "kw", "pad_h", "pad_w", "stride", "dilation", "groups",
                 "pool_h", "pool_w", "target", "output_layer", "tau_is_tensor", "refractory")] + \
               [("alpharp", ctypes.c_float), ("wrp", ctypes.c_float)]


class DenseDesc(ctypes.Structure):
    """Mirror of dcll_dense_desc (include/dcll_hip.h)."""
    _fields_ = [(n, ctypes.c_int32) for n in ("in_features", "out_features", "target", "tau_is_tensor",
                                               "refractory")] + \
               [("alpharp", ctypes.c_float), ("wrp", ctypes.c_float)]


def build(force=False):
    src = os.path.join(_HERE, "dcll_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-B", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _p(a):
    if a is None:
        return None
    assert a.dtype in (np.float32, np.int32) and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(ctypes.c_void_p)


def _pair(v):
    return tuple(v) if hasattr(v, "__len__") else (v, v)


def conv_desc(c_in, c_out, hw, kernel, padding, pooling, target, output_layer, tau_is_tensor, wrp, alpharp=.65,
              stride=1, dilation=1, groups=1):
    (kh, kw), (pah, paw), (poh, pow_) = _pair(kernel), _pair(padding), _pair(pooling)
    return ConvDesc(c_in, c_out, hw[0], hw[1], kh, kw, pah, paw, stride, dilation, groups, poh, pow_, target, int(output_layer),
                    int(tau_is_tensor), int(wrp > 0), alpharp, wrp)


def conv_out_shape(d):
    v = [ctypes.c_int32() for _ in range(4)]
    lib().dcll_oracle_conv_out_shape(ctypes.byref(d), *[ctypes.byref(i) for i in v])
    return tuple(i.value for i in v)


class OracleConvLayer:
    """Stateful wrapper: numpy float32 in/out, neuron state updated in place like the C ABI does."""

    def __init__(self, sd, hw, padding, pooling, wrp, alpharp=.65, output_layer=False, stride=1, dilation=1, groups=1):
        f = lambda k: np.ascontiguousarray(np.asarray(sd[k], dtype=np.float32))
        self.W, self.b = f("i2h.weight"), (f("i2h.bias") if "i2h.bias" in sd else None)      # (bias=False: chains start at 0)
        self.tau = [f("i2h.alpha"), f("i2h.tau_m__dt"), f("i2h.alphas"), f("i2h.tau_s__dt")]
        self.i2o_W, self.i2o_b = f("i2o.weight"), f("i2o.bias")
        self.out_W = f("output_.weight") if output_layer else None
        self.out_b = f("output_.bias") if output_layer else None
        c_out, cig, kh, kw = self.W.shape
        self.d = conv_desc(cig * groups, c_out, hw, (kh, kw), padding, pooling, self.i2o_W.shape[0], output_layer,
                           self.tau[0].size > 1, wrp, alpharp, stride, dilation, groups)
        self.ch, self.cw, self.ph, self.pw = conv_out_shape(self.d)
        self.state = None

    def init_state(self, B):
        d = self.d
        self.state = [np.zeros((B, d.c_in, d.h, d.w), np.float32), np.zeros((B, d.c_in, d.h, d.w), np.float32),
                      np.zeros((B, d.c_out, self.ch, self.cw), np.float32)]

    def forward(self, x, want_v=True):
        x = np.ascontiguousarray(x, dtype=np.float32)
        B, d = x.shape[0], self.d
        if self.state is None or self.state[0].shape[0] != B:
            self.init_state(B)
        s = np.empty((B, d.c_out, self.ph, self.pw), np.float32)
        pv = np.empty_like(s)
        v = np.empty((B, d.c_out, self.ch, self.cw), np.float32) if want_v else None
        p = np.empty((B, d.target), np.float32)
        o = np.empty((B, d.target), np.float32) if d.output_layer else None
        rc = lib().dcll_oracle_conv_lif_step(
            ctypes.byref(d), _p(x), _p(self.W), _p(self.b), *[_p(t) for t in self.tau],
            _p(self.state[0]), _p(self.state[1]), _p(self.state[2]), _p(self.i2o_W), _p(self.i2o_b),
            _p(self.out_W), _p(self.out_b), _p(s), _p(p), _p(o), _p(pv), _p(v), ctypes.c_int32(B))
        assert rc == 0, rc
        return (o if d.output_layer else s), p, pv, v, s


class OracleDenseLayer:
    def __init__(self, sd, wrp, alpharp=.65):
        f = lambda k: np.ascontiguousarray(np.asarray(sd[k], dtype=np.float32))
        self.W, self.b = f("i2h.weight"), f("i2h.bias")
        self.tau = [f("i2h.alpha"), f("i2h.tau_m__dt"), f("i2h.alphas"), f("i2h.tau_s__dt")]
        self.i2o_W, self.i2o_b = f("i2o.weight"), f("i2o.bias")
        self.d = DenseDesc(self.W.shape[1], self.W.shape[0], self.i2o_W.shape[0], int(self.tau[0].size > 1),
                           int(wrp > 0), alpharp, wrp)
        self.state = None

    def forward(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, self.d.in_features)
        B, d = x.shape[0], self.d
        if self.state is None or self.state[0].shape[0] != B:
            self.state = [np.zeros((B, d.in_features), np.float32), np.zeros((B, d.in_features), np.float32),
                          np.zeros((B, d.out_features), np.float32)]
        s = np.empty((B, d.out_features), np.float32)
        pv, v = np.empty_like(s), np.empty_like(s)
        p = np.empty((B, d.target), np.float32)
        rc = lib().dcll_oracle_dense_lif_step(
            ctypes.byref(d), _p(x), _p(self.W), _p(self.b), *[_p(t) for t in self.tau],
            _p(self.state[0]), _p(self.state[1]), _p(self.state[2]), _p(self.i2o_W), _p(self.i2o_b),
            _p(s), _p(p), _p(pv), _p(v), ctypes.c_int32(B))
        assert rc == 0, rc
        return s, p, pv, v


def argmax_vote(logits, t_begin=0):
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    T, B, N = logits.shape
    clout = np.empty((T, B), np.int32)
    vote = np.empty((B,), np.int32)
    lib().dcll_oracle_argmax_vote(_p(logits), _p(clout), _p(vote), T, B, N, t_begin)
    return clout, vote


class OracleConvNetwork:
    """Chain of OracleConvLayer (ConvNetwork.test, networks/__init__.py:182-185)."""

    def __init__(self, layer_sds, convs, im_hw, wrp, alpharp=.65):
        self.layers = []
        hw = tuple(im_hw)
        n = len(convs)
        for i, (sd, c) in enumerate(zip(layer_sds, convs)):
            l = OracleConvLayer(sd, hw, c["padding"], c["pooling"], wrp, alpharp, output_layer=(i == n - 1))
            self.layers.append(l)
            hw = (l.ph, l.pw)

    def step(self, x, want_v=False):
        outs = []
        cur = x
        for l in self.layers:
            o, p, pv, v, s = l.forward(cur, want_v=want_v)
            outs.append(dict(o=o, p=p, pv=pv, v=v, s=s))
            cur = s
        return outs
