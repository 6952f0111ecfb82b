"""ORACLE-side checker (test infrastructure, not product): spike flips of a device run against the reference's CPU path.

north_star says "spike trains must match the reference bit-exactly".  The reference's conv runs in oneDNN, whose summation
order is unspecified, so two correct fp32 implementations can disagree on `v > 0` where |v| is inside the
rounding band of the sum; a flipped spike then changes arp by wrp and the trains stay apart (reference
dcll/pytorch_libdcll.py:495-503).  SURVEY 8(c) therefore asks for the mismatch COUNT, and for every first mismatch to sit
inside the band 8*eps*sum|w*eps1| (+ one rounding of v).  This module counts, free-running over all T:

  flips per layer, samples with any flip, the first flip's step, and — for the FIRST flip of each sample (earliest step,
  lowest layer: up to there both runs saw bit-identical inputs and traces) — whether |v_ref| is inside the band.

Only tests/ and bench.py's cpu_baseline leg import it."""
import numpy as np
import torch
import torch.nn.functional as F

EPS = float(np.finfo(np.float32).eps)


def unpack_words(words, hw):
    """(T,B,C,HW/32) int32 packed spikes (bit pix%32 of word pix/32) -> (T,B,C,H,W) bool"""
    w = np.ascontiguousarray(words).view(np.uint32)
    bits = (w[..., None] >> np.arange(32, dtype=np.uint32)) & 1
    return bits.reshape(w.shape[:-1] + tuple(hw)).astype(bool)


def spike_flips(ref, x, dev_spikes):
    """ref: oracle.torch_ref.RefConvNetwork (state freshly reset); x: (T,B,1,H,W) float tensor of input planes;
    dev_spikes: per layer (T,B,C,H,W) bool arrays of the device run on the same input.
    Runs the reference path free over all T and compares.  -> dict of counts (see module docstring)."""
    T, B = x.shape[0], x.shape[1]
    L = len(ref.layers)
    flips = np.zeros(L, dtype=np.int64)
    total = np.zeros(L, dtype=np.int64)
    first_step = np.full(B, -1, dtype=np.int64)
    first_layer = np.full(B, -1, dtype=np.int64)
    in_band = outside = 0
    worst_ratio = 0.0
    with torch.no_grad():
        for t in range(T):
            outs = ref.test(x[t])
            for l, (o, p, pv, v) in enumerate(outs):
                s_ref = (v > 0).numpy()
                mism = s_ref != dev_spikes[l][t]
                total[l] += mism.size
                if not mism.any():
                    continue
                flips[l] += int(mism.sum())
                per_sample = mism.reshape(B, -1).any(axis=1)
                for b in np.nonzero(per_sample & (first_step < 0))[0]:
                    # first flip of this sample: lower layers are clean up to and including step t (they were visited
                    # first), so this layer's traces are bit-identical in both runs and only the conv's order differs
                    first_step[b], first_layer[b] = t, l
                    lay = ref.layers[l]
                    e1 = lay.state[1][b:b + 1].abs()
                    bnd = 8 * EPS * F.conv2d(e1, lay.w.abs(), lay.b.abs(), 1, lay.padding)[0] + EPS * v[b].abs()
                    m = torch.from_numpy(mism[b])
                    ratio = (v[b].abs()[m] / bnd[m]).max().item()
                    worst_ratio = max(worst_ratio, ratio)
                    if ratio <= 1.0:
                        in_band += 1
                    else:
                        outside += 1
    flipped = first_step >= 0
    return {"windows": int(B), "steps": int(T), "spikes_compared_per_layer": [int(n) for n in total],
            "flips_per_layer": [int(n) for n in flips], "windows_with_a_flip": int(flipped.sum()),
            "first_flip_step_min": int(first_step[flipped].min()) if flipped.any() else None,
            "first_flip_layers": [int((first_layer == l).sum()) for l in range(L)],
            "first_flips_inside_rounding_band": int(in_band), "first_flips_outside_rounding_band": int(outside),
            "worst_first_flip_v_over_band": float(worst_ratio)}


def merge(a, b):
    """Sum two spike_flips() results (batches of the same network)."""
    if a is None:
        return b
    out = dict(a)
    out["windows"] = a["windows"] + b["windows"]
    for k in ("spikes_compared_per_layer", "flips_per_layer", "first_flip_layers"):
        out[k] = [x + y for x, y in zip(a[k], b[k])]
    for k in ("windows_with_a_flip", "first_flips_inside_rounding_band", "first_flips_outside_rounding_band"):
        out[k] = a[k] + b[k]
    mins = [m for m in (a["first_flip_step_min"], b["first_flip_step_min"]) if m is not None]
    out["first_flip_step_min"] = min(mins) if mins else None
    out["worst_first_flip_v_over_band"] = max(a["worst_first_flip_v_over_band"], b["worst_first_flip_v_over_band"])
    return out
