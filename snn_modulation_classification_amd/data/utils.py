"""Input encoders — host restatements of the reference's data/utils.py plus the on-device IQ encoder.

  to_one_hot         reference data/utils.py:10-12
  image2spiketrain   reference data/utils.py:15-40  (frozen Poisson trains, numpy RNG driven)
  iq2spiketrain      reference data/utils.py:43-87  (one spike per I/Q sample in the I/Q plane)
  iq2cells           same quantisation, returning the cell index q*W + i per (t, b) instead of a dense plane
  IQEncoder          the same map on the MI355X (dcll_iq_encode): thresholds found on the host by bisection over the
                     host encoder, so device cells are bit-identical to the host's
"""
import numpy as np
import torch


def add_gaussian(x, gs_stdev):
    return x + torch.empty_like(x).normal_(0, gs_stdev)


def to_one_hot(t, width):
    onehot = torch.zeros(*t.shape + (width,))
    return onehot.scatter_(1, t.unsqueeze(-1), 1)


def _repeat_over_time(y, n):
    """(B,C) labels -> (n,B,C); a torch tensor stays a torch tensor (what np.repeat yields in the reference)."""
    if isinstance(y, torch.Tensor):
        return y.unsqueeze(0).repeat(n, 1, 1)
    return np.repeat(y[np.newaxis, :, :], n, axis=0)


def image2spiketrain(x, y, input_shape, gain=50, min_duration=None, max_duration=500):
    """Frozen Poisson spike trains from pixel intensities; same numpy RNG call order as the reference."""
    if min_duration is None:
        min_duration = max_duration - 1
    batch = x.shape[0]
    n_in = int(np.prod(input_shape))
    keep_silent_p = (1000.0 - np.array(gain * x.reshape(batch, -1))) / 1000
    durations = np.random.randint(min_duration, max_duration, batch)
    trains = np.zeros((max_duration, batch, n_in))
    for b in range(batch):
        silent = np.random.uniform(size=(durations[b], n_in)) < keep_silent_p[b]
        trains[:durations[b], b, :] = np.where(silent, 0.0, 1.0)
    trains = trains.reshape(max_duration, batch, *input_shape)
    return trains, _repeat_over_time(y, max_duration)


def _quantise(v, lo, hi, n_cells, do_gamma):
    """float32 torch ops in the reference's order: normalise, (gamma 1/1.2 on [-1,1]), clamp, scale, truncate."""
    c = (v - lo) / (hi - lo)
    if do_gamma:
        c = c * 2.0 - 1.0
        c = c.sign() * (c.abs() ** (1.0 / 1.2))
        c = (c + 1.0) * 0.5
    return (c.clamp(0, 1) * (n_cells - 1)).int()


def iq2cells(x, out_w=28, out_h=28, min_I=-1, max_I=1, min_Q=-1, max_Q=1, max_duration=500, do_gamma=True,
             gs_stdev=0):
    """(B,2,1,L) or (B,2,L) IQ -> int32 cells (max_duration, B), cell = q*out_w + i.  Draws the random crop start
    exactly like the reference (np.random.randint, even when L == max_duration) and, like it, quantises one time
    sample (a length-B vector) at a time: torch's float pow takes its vector path for full 16/32-element groups and the
    scalar libm path for the tail, and the two can differ in the last ulp, so the slicing is part of the result."""
    x = x.squeeze()
    if gs_stdev > 0:
        x = add_gaussian(x, gs_stdev)
    n_t = x.shape[-1]
    assert max_duration <= n_t
    t0 = np.random.randint(0, n_t - max_duration + 1)
    cells = torch.empty((max_duration, x.shape[0]), dtype=torch.int32)
    for i, t in enumerate(range(t0, t0 + max_duration)):
        ci = _quantise(x[:, 0, t], min_I, max_I, out_w, do_gamma)
        cq = _quantise(x[:, 1, t], min_Q, max_Q, out_h, do_gamma)
        cells[i] = cq * out_w + ci
    return cells, t0


def iq2spiketrain(x, y, out_w=28, out_h=28, min_I=-1, max_I=1, min_Q=-1, max_Q=1, max_duration=500, do_gamma=True,
                  gs_stdev=0):
    """Dense float64 (T,B,1,out_h,out_w) spike planes + labels repeated over T, as the reference returns."""
    cells, _ = iq2cells(x, out_w, out_h, min_I, max_I, min_Q, max_Q, max_duration, do_gamma, gs_stdev)
    T, B = cells.shape
    trains = np.zeros((T, B, out_h * out_w))
    np.put_along_axis(trains, cells.numpy()[:, :, None].astype(np.int64), 1.0, axis=2)
    trains = trains.reshape(T, B, 1, out_h, out_w)
    return trains, _repeat_over_time(y, max_duration)


# ---------------------------------------------------------------------------------------------------------------
# on-device encoder
# ---------------------------------------------------------------------------------------------------------------
def _f32_to_ordered(u):
    """Map float32 bit patterns to integers that sort like the floats."""
    u = u.astype(np.int64)
    return np.where(u & 0x80000000, 0x80000000 - (u & 0x7FFFFFFF), u + 0x80000000)


def _ordered_to_f32(k):
    k = np.asarray(k, dtype=np.int64)
    u = np.where(k >= 0x80000000, k - 0x80000000, (0x80000000 - k) | 0x80000000).astype(np.uint32)
    return u.view(np.float32)


def cell_thresholds(lo, hi, n_cells, do_gamma=True, window=2048):
    """thr[j] = smallest float32 x with quantise(x) >= j+1, j = 0..n_cells-2, by bisection over the float32 line.

    The host map is evaluated through torch's VECTOR path (the candidate replicated to 64 lanes: torch's float loops
    run 2 x Vec::size() = 16 (AVX2) or 32 (AVX-512) elements per vector iteration and hand the tail to scalar libm),
    which is what every sample of a batch whose size is a multiple of 32 goes through in the reference.  `pow` need not be
    monotone to the last ulp, so every threshold is verified on `window` consecutive floats on either side and a
    ValueError is raised if the map is not a clean step there (never observed)."""
    def f(vals):
        vals = np.asarray(vals, dtype=np.float32)
        pad = (-len(vals)) % 64
        full = np.concatenate([vals, np.repeat(vals[-1:], pad)]) if pad else vals
        return _quantise(torch.from_numpy(full), lo, hi, n_cells, do_gamma).numpy()[:len(vals)]
    lo_k = int(_f32_to_ordered(np.array([np.float32(lo - 1.0)]).view(np.uint32))[0])
    hi_k = int(_f32_to_ordered(np.array([np.float32(hi + 1.0)]).view(np.uint32))[0])
    thr = np.empty(n_cells - 1, dtype=np.float32)
    for j in range(n_cells - 1):
        a, b = lo_k, hi_k               # f(a) < j+1 <= f(b)
        while b - a > 1:
            mid = (a + b) // 2
            if f(_ordered_to_f32([mid] * 64))[0] >= j + 1:
                b = mid
            else:
                a = mid
        thr[j] = _ordered_to_f32([b])[0]
        ks = np.arange(b - window, b + window)
        got = f(_ordered_to_f32(ks))
        if not (np.all(got[:window] <= j) and np.all(got[window:] >= j + 1)):
            raise ValueError('host IQ quantiser is not monotone around cell boundary %d' % (j + 1))
    return thr


class IQEncoder:
    """iq2spiketrain's quantisation on the GPU: raw IQ (B,2,L) fp32 in HBM -> cells (T,B) int32."""

    def __init__(self, out_w, out_h, I_bounds=(-1, 1), Q_bounds=(-1, 1), do_gamma=True, device='cuda'):
        self.w, self.h = out_w, out_h
        self.thr_i = torch.from_numpy(cell_thresholds(I_bounds[0], I_bounds[1], out_w, do_gamma)).to(device)
        self.thr_q = torch.from_numpy(cell_thresholds(Q_bounds[0], Q_bounds[1], out_h, do_gamma)).to(device)

    def __call__(self, iq, max_duration, t0=None):
        from .. import ops
        iq = iq.reshape(iq.shape[0], 2, -1).contiguous()
        L = iq.shape[-1]
        if t0 is None:
            t0 = np.random.randint(0, L - max_duration + 1)       # same draw as the host encoder
        return ops.iq_encode(iq, self.thr_i, self.thr_q, t0, max_duration, self.w, self.h)
