"""Config 5 of BASELINE.json — int8 conv weights + 1-bit packed spikes — as this build defines it.

The reference has no quantisation code (its README only says "(quantized)"), so there is nothing to be bit-exact
against: PARITY UNPINNED.  Definition used here:

* weights: per-output-channel symmetric int8, scale[co] = max|W[co]| / 127, q = round-half-even(W / scale) in
  [-127, 127]; the kernels run on the DEQUANTISED fp32 weights q * scale (exactly representable products of an
  integer and one fp32 scale are what an int8 MAC followed by one fp32 multiply per channel yields up to the
  accumulation order, which the pinned fmaf chain fixes);
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
    """Quantise every slice's conv weight in place (i2h.weight <- dequantised int8); returns [(q, scale)] per slice so
    that a caller can store the 8-bit form (1/4 of the bytes)."""
    out = []
    with torch.no_grad():
        for s in net.dcll_slices:
            w = s.dclllayer.i2h.weight
            q, scale = quantize_int8_per_channel(w)
            w.copy_(dequantize(q, scale))
            out.append((q, scale))
    return out
