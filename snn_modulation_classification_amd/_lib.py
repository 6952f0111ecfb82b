"""ctypes binding of libdcll_hip.so (include/dcll_hip.h).

There is NO fallback: if the library is missing or a call fails, a DCLLHipError is raised.  The product never
imports oracle/.
"""
import ctypes
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
# (DCLL_HIP_SO: another build of the same sources, for A/B measurements of compile-time switches — experiments/)
SO_PATH = os.environ.get("DCLL_HIP_SO") or os.path.join(_PKG, "libdcll_hip.so")
CSRC = os.path.join(_PKG, "csrc")

DCLL_OK, DCLL_ERR_INVALID, DCLL_ERR_UNSUPPORTED, DCLL_ERR_LAUNCH = 0, -1, -2, -3
ABI_VERSION = 7


class DCLLHipError(RuntimeError):
    pass


class DCLLUnsupported(NotImplementedError):
    pass


class ConvDesc(ctypes.Structure):
    """dcll_conv_desc"""
    _fields_ = [(n, ctypes.c_int32) for n in
                ("c_in", "c_out", "h", "w", "kh", "kw", "pad_h", "pad_w", "stride", "dilation", "groups",
                 "pool_h", "pool_w", "target", "output_layer", "tau_is_tensor", "refractory")] + \
               [("alpharp", ctypes.c_float), ("wrp", ctypes.c_float)]


class DenseDesc(ctypes.Structure):
    """dcll_dense_desc"""
    _fields_ = [(n, ctypes.c_int32) for n in ("in_features", "out_features", "target", "tau_is_tensor",
                                               "refractory")] + \
               [("alpharp", ctypes.c_float), ("wrp", ctypes.c_float)]


class AdamTensor(ctypes.Structure):
    """dcll_adam_tensor"""
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
                ("exp_avg_sq", ctypes.c_void_p), ("n", ctypes.c_int64), ("step", ctypes.c_int64),
                ("lr", ctypes.c_float), ("weight_decay", ctypes.c_float), ("beta1", ctypes.c_float),
                ("beta2", ctypes.c_float), ("eps", ctypes.c_float)]


class GradParts(ctypes.Structure):
    """dcll_grad_parts"""
    _fields_ = [("part", ctypes.c_void_p), ("dW", ctypes.c_void_p), ("db", ctypes.c_void_p), ("rowlen", ctypes.c_int64),
                ("nchunk", ctypes.c_int32), ("c_out", ctypes.c_int32), ("adam_w", ctypes.c_int32), ("adam_b", ctypes.c_int32)]


class StepRo(ctypes.Structure):
    """dcll_step_ro (ABI 6): the arguments of one dcll_step_readouts call, as an item of dcll_step_readouts_multi"""
    _fields_ = [("pv", ctypes.c_void_p), ("Wt", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("scratch", ctypes.c_void_p),
                ("scratch_floats", ctypes.c_int64), ("rows", ctypes.c_int64), ("K", ctypes.c_int32), ("N1", ctypes.c_int32),
                ("N2", ctypes.c_int32), ("kind", ctypes.c_int32), ("p", ctypes.c_void_p), ("o", ctypes.c_void_p),
                ("clout", ctypes.c_void_p), ("target", ctypes.c_void_p), ("g_p", ctypes.c_void_p), ("g_o", ctypes.c_void_p),
                ("reserved", ctypes.c_int64)]


STEP_RO_MAX = 8


class BwdItem(ctypes.Structure):
    """dcll_bwd_item (ABI 6): the arguments of one dcll_conv_lif_backward_open call, as an item of ..._open_multi"""
    _fields_ = [("d", ctypes.POINTER(ConvDesc))] + \
               [(n, ctypes.c_void_p) for n in ("eps1", "v", "pv_pooled", "g_p", "g_o", "g_pv", "g_v", "i2o_W", "d_outW", "d_outb",
                                               "scratch")] + \
               [("scratch_floats", ctypes.c_int64), ("B", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("part", ctypes.c_void_p), ("nchunk", ctypes.c_int32), ("reserved2", ctypes.c_int32)]


BWD_MULTI_MAX = 8


class LayerOpts(ctypes.Structure):
    """dcll_layer_opts (ABI v3): int8 conv weights + per-output-channel scale, pv written before the sigmoid"""
    _fields_ = [("w_q8", ctypes.c_void_p), ("w_scale", ctypes.c_void_p), ("pv_presigmoid", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class IQTail(ctypes.Structure):
    """dcll_iq_tail: thresholds of torch's scalar pow path + the batch positions that take it"""
    _fields_ = [("thr_i_tail", ctypes.c_void_p), ("thr_q_tail", ctypes.c_void_p), ("tail_mask", ctypes.c_void_p)]


ACT_NONE, ACT_SIGMOID = 0, 1
ADAM_MAX_TENSORS = 8
REDUCE_MAX_LAYERS = 4
LOSS_SMOOTH_L1, LOSS_MSE = 0, 1

_P, _I32, _I64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
_F32 = ctypes.c_float
_DP = ctypes.POINTER(ConvDesc)
_DDP = ctypes.POINTER(DenseDesc)
_IP = ctypes.POINTER(ctypes.c_int32)
_OP = ctypes.POINTER(LayerOpts)

# every symbol include/dcll_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "dcll_version": (_I32, []),
    "dcll_last_error": (ctypes.c_char_p, []),
    "dcll_kernel_trace": (_I32, [_I32]),
    "dcll_kernel_trace_read": (_I64, [ctypes.c_char_p, _I64]),
    "dcll_conv_out_shape": (_I32, [_DP, _IP, _IP, _IP, _IP]),
    "dcll_conv_lif_step": (_I32, [_DP] + [_P] * 20 + [_OP, _I32, _P]),
    "dcll_conv_lif_backward": (_I32, [_DP] + [_P] * 13 + [_I64, _I32, _P]),
    "dcll_local_loss_grad": (_I32, [_P] * 7 + [_I32, _I32, _I32, _P]),
    "dcll_adam_step": (_I32, [ctypes.POINTER(AdamTensor), _I32, _P]),
    "dcll_adam_step_dyn": (_I32, [ctypes.POINTER(AdamTensor), _I32, _P, _P]),
    "dcll_conv_lif_backward_open": (_I32, [_DP] + [_P] * 11 + [_I64, _I32, ctypes.POINTER(ctypes.c_void_p), _IP, _P]),
    "dcll_grad_reduce_adam": (_I32, [ctypes.POINTER(GradParts), _I32, ctypes.POINTER(AdamTensor), _I32, _P, _P]),
    "dcll_conv_lif_backward_open_multi": (_I32, [ctypes.POINTER(BwdItem), _I32, _P]),
    "dcll_cells_to_planes": (_I32, [_P, _P, _I64, _I32, _P]),
    "dcll_dense_lif_step": (_I32, [_DDP] + [_P] * 16 + [_I32, _P]),
    "dcll_dense_lif_sequence": (_I32, [_DDP] + [_P] * 16 + [_I32, _I32, _P]),
    "dcll_dense_lif_backward": (_I32, [_DDP] + [_P] * 9 + [_I64, _I32, _P]),
    "dcll_dense_lif_backward_open": (_I32, [_DDP] + [_P] * 7 + [_I64, _I32, ctypes.POINTER(ctypes.c_void_p), _IP, _P]),
    "dcll_conv_lif_sequence": (_I32, [_DP] + [_P] * 13 + [_I32, _P, _P, _I32, _OP, _I32, _I32, _P]),
    "dcll_permute_readout": (_I32, [_P, _P, _I32, _P]),
    "dcll_conv_lif_sequence_cells": (_I32, [_DP] + [_P] * 10 + [_P, _P, _I32, _OP, _I32, _I32, _P]),
    "dcll_conv_lif_sequence_iq": (_I32, [_DP, _P, _P, _P, _P, _I32, _I32] + [_P] * 9 + [_P, _P, _I32, _OP, _I32, _I32, _P]),
    "dcll_pv_lowhigh": (_I32, [_P, _I64, _I32, _I32, _P, _P]),
    "dcll_pv_lowhigh_act": (_I32, [_P, _I64, _I32, _I32, _P, _I32, _P]),
    "dcll_readout_act_scratch": (_I64, [_I64, _I32, _I32]),
    "dcll_readout_act": (_I32, [_P] * 5 + [_I64, _I64, _I32, _I32, _I32, _P]),
    "dcll_pv_lowhigh_steps": (_I32, [_I32, _I32]),
    "dcll_readout": (_I32, [_P, _P, _P, _P, _I64, _I32, _I32, _P]),
    "dcll_readout_mode": (_I32, [_P, _P, _P, _P, _I64, _I32, _I32, _I32, _P]),
    "dcll_readout_splitk_scratch": (_I64, [_I64, _I32, _I32]),
    "dcll_readout_splitk": (_I32, [_P, _P, _P, _P, _P, _I64, _I64, _I32, _I32, _P]),
    "dcll_vote_tallies": (_I32, [_P, _I32, _P, _P, _I32, _I32, _P]),
    "dcll_step_readouts_scratch": (_I64, [_I64, _I32, _I32, _I32]),
    "dcll_step_readouts": (_I32, [_P, _P, _P, _P, _I64, _I64, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _I32, _P]),
    "dcll_step_readouts_multi": (_I32, [ctypes.POINTER(StepRo), _I32, _P]),
    "dcll_argmax_vote": (_I32, [_P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "dcll_iq_encode": (_I32, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "dcll_unpack_spikes": (_I32, [_P, _P, _I64, _P]),
    "dcll_pack_spikes": (_I32, [_P, _P, _I64, _P]),
}


def build(force=False):
    """Compile csrc/*.hip for gfx950 in-tree (hipcc cross-compiles without a GPU).  `make` decides what is stale from
    its dependency rules (every source, dcll_internal.h, the ABI header, the Makefile itself); force=True rebuilds all."""
    subprocess.check_call(["make", "-s", "-j4", "-C", CSRC] + (["-B"] if force else []))
    return SO_PATH


_lib = None


def get():
    """The loaded library with typed entry points; raises DCLLHipError if it cannot be loaded."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise DCLLHipError(
                "libdcll_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C %s`. There is no CPU fallback for the DCLL layer forward." % (SO_PATH, CSRC))
        # torch first: PyTorch-ROCm bundles its own libamdhip64.so.7; whichever HIP runtime is mapped first serves
        # BOTH torch and this library (same SONAME), and the kernels must run in the runtime that owns torch's
        # streams and allocations.
        import torch  # noqa: F401
        try:
            lib = ctypes.CDLL(SO_PATH)
        except OSError as e:
            raise DCLLHipError("cannot load %s: %s" % (SO_PATH, e))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)        # AttributeError here = header/library mismatch: loud by design
            fn.restype = res
            fn.argtypes = args
        if lib.dcll_version() != ABI_VERSION:
            raise DCLLHipError("libdcll_hip.so ABI %d != binding ABI %d" % (lib.dcll_version(), ABI_VERSION))
        _lib = lib
    return _lib


def check(rc, what):
    if rc == DCLL_OK:
        return
    msg = get().dcll_last_error().decode("utf-8", "replace")
    if rc == DCLL_ERR_UNSUPPORTED:
        raise DCLLUnsupported("%s: %s" % (what, msg))
    if rc == DCLL_ERR_INVALID:
        raise ValueError("%s: %s" % (what, msg))
    raise DCLLHipError("%s failed (%d): %s" % (what, rc, msg))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL).  The tensor must be contiguous and live on a GPU."""
    if t is None:
        return None
    if not t.is_cuda:
        raise DCLLHipError("DCLL HIP kernels need tensors on an MI355X ('cuda') device, got %s — there is no CPU "
                           "fallback" % t.device)
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    """The current stream of the current device as a hipStream_t (queried per call: a graph capture runs on a side stream).
    torch's raw accessor where it exists — torch.cuda.current_stream() builds a Stream object, ~6 us, four times per timestep."""
    import torch
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        return ctypes.c_void_p(raw(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
