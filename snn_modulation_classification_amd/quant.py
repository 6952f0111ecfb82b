"""Config 5 of BASELINE.json — int8 conv weights + 1-bit packed spikes — as this build defines it.

The reference has no quantisation code (its README only says "(quantized)"), so there is nothing to be bit-exact
against: PARITY UNPINNED.  Definition used here:

* weights: per-output-channel symmetric int8, scale[co] = max|W[co]| / 127, q = round-half-even(W / scale) in
  [-127, 127]; the kernels READ THE INT8 TENSOR through the C ABI (`dcll_layer_opts.w_q8 / w_scale`, ABI v3) and
  convert every weight exactly once, w = (float)q * scale[co] — one rounded fp32 multiply, the same value
  `dequantize` produces — before the pinned fmaf chain runs on it: bit-identical to the run on the dequantised fp32
  tensor (tested), a quarter of the weight bytes.  `i2h.weight` keeps the dequantised values (state-dict, per-step
  autograd path);
* spikes: between layers as 1-bit packed words (`ops.pack_spikes` / the sequence kernels' native format) — lossless.

The parity statement that can be made and is tested: the HIP path on the dequantised weights equals the C oracle on the
same dequantised weights bit for bit, and packing the inter-layer spikes changes nothing.
"""
import torch


def quantize_int8_per_channel(w):
    """(c_out, ...) fp32 -> (q int8 same shape, scale (c_out,) fp32)."""
    flat = w.detach().reshape(w.shape[0], -1)
    scale = flat.abs().amax(dim=1) / 127.0
    scale = torch.where(scale > 0, scale, torch.ones_like(scale))
    q = torch.clamp(torch.round(flat / scale[:, None]), -127, 127).to(torch.int8)
    return q.reshape(w.shape), scale


def dequantize(q, scale):
    return (q.to(torch.float32).reshape(q.shape[0], -1) * scale[:, None]).reshape(q.shape)


def apply_int8_weights(net):
    """Quantise every slice's conv weight: i2h.weight <- the dequantised values, and the int8 tensor + scales are kept on
    the layer (`i2h.set_int8_weights`) — from here on every kernel call of the layer hands THOSE across the C ABI.
    Returns [(q, scale)] per slice."""
    out = []
    with torch.no_grad():
        for s in net.dcll_slices:
            i2h = s.dclllayer.i2h
            q, scale = quantize_int8_per_channel(i2h.weight)
            i2h.weight.copy_(dequantize(q, scale))
            i2h.set_int8_weights(q, scale)
            out.append((q, scale))
    return out
