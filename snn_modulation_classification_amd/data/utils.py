"""Input encoders — host restatements of the reference's data/utils.py plus the on-device IQ encoder.

  to_one_hot         reference data/utils.py:10-12
  image2spiketrain   reference data/utils.py:15-40  (frozen Poisson trains, numpy RNG driven)
  iq2spiketrain      reference data/utils.py:43-87  (one spike per I/Q sample in the I/Q plane)
  iq2cells           same quantisation, returning the cell index q*W + i per (t, b) instead of a dense plane
  IQEncoder          the same map on the MI355X (dcll_iq_encode): thresholds found on the host by bisection over the
                     host encoder — one table per code path of torch's float pow (vector groups / scalar tail) plus the
                     batch positions that take the scalar one — so device cells are bit-identical to the host's for
                     every batch size
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


def cell_thresholds(lo, hi, n_cells, do_gamma=True, window=2048, path='vector'):
    """thr[j] = smallest float32 x with quantise(x) >= j+1, j = 0..n_cells-2, by bisection over the float32 line.

    path='vector': the host map is evaluated through torch's VECTOR path (the candidates padded to a multiple of 64
    lanes: torch's float loops run 2 x Vec::size() = 16 (AVX2) or 32 (AVX-512) elements per vector iteration and hand
    the tail to scalar libm), which is what the samples in full groups of a batch go through in the reference.
    path='scalar': through the scalar tail (tensors of 15 elements — shorter than any vector iteration), which is what the
    last B mod 16 / 32 samples of a batch (and the samples at the chunk ends of torch's thread pool) go through.  The two
    differ in the last ulp at some cell boundaries (on an AVX-512 host: 4 of the 15 boundaries of a 16-cell axis).
    `pow` need not be monotone to the last ulp, so every threshold is verified on `window` consecutive floats on either
    side and a ValueError is raised if the map is not a clean step there (never observed)."""
    def f(vals):
        vals = np.asarray(vals, dtype=np.float32)
        if path == 'scalar':
            out = [_quantise(torch.from_numpy(vals[k:k + 15].copy()), lo, hi, n_cells, do_gamma).numpy()
                   for k in range(0, len(vals), 15)]
            return np.concatenate(out)
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
            if f(_ordered_to_f32([mid] * (1 if path == 'scalar' else 64)))[0] >= j + 1:
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
    """iq2spiketrain's quantisation on the GPU: raw IQ (B,2,L) fp32 in HBM -> cells (T,B) int32.

    Exact for every batch size: the reference quantises a length-B vector per time sample with torch CPU ops, and which
    implementation of `pow` an element meets — vector or scalar — depends on its position in that vector.  The encoder
    holds the thresholds of both paths and asks torch itself which positions of a B-vector take the scalar one (a witness
    value on which the two paths disagree, pushed through the host quantiser once per batch size): `tail(B)` is what the
    kernels get as dcll_iq_tail.  `exact_tail=False` keeps the single vector-path table (exact for batches that are
    multiples of 32)."""

    def __init__(self, out_w, out_h, I_bounds=(-1, 1), Q_bounds=(-1, 1), do_gamma=True, device='cuda', exact_tail=True):
        self.w, self.h, self.device = out_w, out_h, device
        self._bounds, self._gamma = (tuple(I_bounds), tuple(Q_bounds)), do_gamma
        ti = cell_thresholds(I_bounds[0], I_bounds[1], out_w, do_gamma)
        tq = cell_thresholds(Q_bounds[0], Q_bounds[1], out_h, do_gamma)
        self.thr_i, self.thr_q = torch.from_numpy(ti).to(device), torch.from_numpy(tq).to(device)
        self.thr_i_tail = self.thr_q_tail = None
        self._witness, self._masks, self._dev_masks = None, {}, {}
        if exact_tail:
            si = cell_thresholds(I_bounds[0], I_bounds[1], out_w, do_gamma, path='scalar')
            sq = cell_thresholds(Q_bounds[0], Q_bounds[1], out_h, do_gamma, path='scalar')
            for axis, (v_, s_) in enumerate(((ti, si), (tq, sq))):
                d = np.nonzero(v_ != s_)[0]
                if len(d) and self._witness is None:
                    # at min(thr) the path with the lower threshold already says cell j + 1, the other still j
                    j = int(d[0])
                    x = min(v_[j], s_[j])
                    self._witness = (axis, float(x), int(j + 1) if s_[j] < v_[j] else int(j), out_w if axis == 0 else out_h)
            if self._witness is not None:       # (else: the two paths agree on this host — one table serves all)
                self.thr_i_tail, self.thr_q_tail = torch.from_numpy(si).to(device), torch.from_numpy(sq).to(device)

    def tail_mask_host(self, B):
        """uint8 (B): 1 where torch's quantiser sends position b of a length-B vector through its scalar pow path.
        Above torch's parallel grain (32 768 elements) the vector is cut into one chunk per thread and every chunk END takes
        the scalar path, so the positions also depend on the thread count in force: it is part of the cache key."""
        key = (B, torch.get_num_threads())
        m = self._masks.get(key)
        if m is None:
            axis, x, scalar_cell, n_cells = self._witness
            lo, hi = self._bounds[axis]
            got = _quantise(torch.full((B,), x, dtype=torch.float32), lo, hi, n_cells, self._gamma).numpy()
            m = self._masks[key] = (got == scalar_cell).astype(np.uint8)
        return m

    def tail(self, B, start=0, stop=None, total=None):
        """dcll_iq_tail operands (thr_i_tail, thr_q_tail, mask) for samples start..stop of a batch of `total` (default: the
        B samples are the whole batch), or None when one table serves all."""
        if self._witness is None:
            return None
        total = B if total is None else total
        stop = start + B if stop is None else stop
        key = (total, start, stop, torch.get_num_threads())
        if key not in self._dev_masks:              # one upload per (batch, shard), not one per call of the hot path
            mask = self.tail_mask_host(total)[start:stop]
            if len(self._dev_masks) > 64:
                self._dev_masks.clear()
            self._dev_masks[key] = torch.from_numpy(np.ascontiguousarray(mask)).to(self.device) if mask.any() else None
        dm = self._dev_masks[key]
        return None if dm is None else (self.thr_i_tail, self.thr_q_tail, dm)

    def __call__(self, iq, max_duration, t0=None, start=0, total=None):
        from .. import ops
        iq = iq.reshape(iq.shape[0], 2, -1).contiguous()
        L = iq.shape[-1]
        if t0 is None:
            t0 = np.random.randint(0, L - max_duration + 1)       # same draw as the host encoder
        return ops.iq_encode(iq, self.thr_i, self.thr_q, t0, max_duration, self.w, self.h,
                             tail=self.tail(iq.shape[0], start=start, total=total))

