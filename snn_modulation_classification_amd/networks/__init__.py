"""YAML network builder — the drop-in surface of the reference's networks/__init__.py (:10-18, :116-199).

`load_network_spec(yaml)` and `ConvNetwork(args, im_dims, batch_size, convs, target_size, act, loss, opt, opt_param,
learning_rates, DCLLSlice, burnin)` with `.learn/.test/.reset/.accuracy/.confusion_matrix/.write_stats` keep the
reference's signatures.  `.test(x_t)` is the per-step path (one C-ABI call per layer per step);
`.test_sequence(...)` is the MI355X fast path: one fused launch per layer for all T steps, bit-packed spikes between
layers, fp32-MFMA readout over all (t, b) rows, per-step argmax and vote on device.
"""
from ast import literal_eval

import numpy as np
import torch
import yaml

from .. import ops
from ..dcll import pytorch_libdcll as _libdcll
from ..dcll.pytorch_libdcll import Conv2dDCLLlayer, DenseDCLLlayer, DCLLClassification  # noqa: F401


def load_network_spec(yaml_path):
    """YAML -> list of layer dicts; non-int values such as "(1, 3)" become tuples (reference :10-18).
    Uses SafeLoader (the reference's bare yaml.load fails on PyYAML >= 6)."""
    with open(yaml_path, 'r') as f:
        spec = yaml.load(f, Loader=yaml.SafeLoader)
    convs = spec['conv_layers']
    for layer in convs:
        for key, val in layer.items():
            if type(val) != int:
                layer[key] = literal_eval(val)
    return convs


class ConvNetwork(torch.nn.Module):
    def __init__(self, args, im_dims, batch_size, convs, target_size, act, loss, opt, opt_param, learning_rates,
                 DCLLSlice=DCLLClassification, burnin=50):
        super().__init__()
        self.batch_size = batch_size
        self.target_size = target_size
        dev = _libdcll.device
        shape = tuple(im_dims)                    # (C, H, W)
        self.num_layers = len(convs)
        self.dcll_slices = torch.nn.ModuleList()
        for i, conf in enumerate(convs):
            last = (i == self.num_layers - 1)
            layer = Conv2dDCLLlayer(in_channels=shape[0],
                                    out_channels=int(conf['out_channels'] * args.netscale),
                                    kernel_size=conf['kernel_size'], padding=conf['padding'],
                                    pooling=conf['pooling'], im_dims=shape[1:3], target_size=target_size,
                                    alpha=args.alpha, alphas=args.alphas, alpharp=args.alpharp, wrp=args.arp,
                                    act=act, lc_ampl=args.lc_ampl, random_tau=args.random_tau, spiking=True,
                                    lc_dropout=False, output_layer=last).to(dev).init_hiddens(batch_size)
            # layers fed by another layer only ever see binary spike maps: lets their per-step forward use the
            # bit-packed MFMA kernel (the first layer takes whatever the caller passes and stays on the generic path)
            layer.i2h.binary_input = (i > 0)
            shape = (layer.out_channels,) + tuple(layer.output_shape)
            layer_opt = dict(opt_param)
            if learning_rates is not None:
                layer_opt['lr'] = learning_rates[min(i, len(learning_rates) - 1)]
            self.dcll_slices.append(DCLLSlice(dclllayer=layer, name='conv%d' % i, batch_size=batch_size, loss=loss,
                                              optimizer=opt, kwargs_optimizer=layer_opt, collect_stats=True,
                                              burnin=burnin))
        self._seq_buffers = {}

    # -- reference protocol (per step) ----------------------------------------------------------------------------
    def learn(self, x, labels):
        spikes = x
        for s in self.dcll_slices:
            spikes, _, _, _, _ = s.train_dcll(spikes, labels, regularize=False)

    def test(self, x):
        spikes = x
        for s in self.dcll_slices:
            spikes, _, _, _ = s.forward(spikes, ignore_burnin=True)

    def reset(self, init_states=False):
        for s in self.dcll_slices:
            s.init(self.batch_size, init_states=init_states)

    def write_stats(self, writer, epoch, comment=''):
        for s in self.dcll_slices:
            s.write_stats(writer, label='test' + comment, epoch=epoch)

    def accuracy(self, labels):
        return [s.accuracy(labels) for s in self.dcll_slices]

    def confusion_matrix(self, labels):
        return self.dcll_slices[-1].confusion_matrix(labels)

    # -- whole-sequence fast path -----------------------------------------------------------------------------------
    def sequence_supported(self):
        """True if every layer has a fused all-T kernel: radio_ml_conv.yaml on the 16x16 I/Q plane of the reference's
        scripts, or on a plane with H % 8 == 0 and W % 32 == 0 (the 128x128 argparse default; there the pv buffer of one
        layer is T*B*32*H*W*4 bytes — 17 GB at T=128, B=64 — so size the batch for it)."""
        kinds = [s.dclllayer.sequence_kind() for s in self.dcll_slices]
        return kinds[0] == 'cells' and all(k == 'packed' for k in kinds[1:])

    def _sequence_buffers(self, T, B, dev):
        key = (T, B, str(dev))
        if key not in self._seq_buffers:
            self._seq_buffers.clear()
            L = self.dcll_slices[0].dclllayer
            C, (H, W) = L.out_channels, L.output_shape
            self._seq_buffers[key] = dict(
                spk=[torch.empty((T, B, C, H * W // 32), device=dev, dtype=torch.int32) for _ in range(2)],
                pv=torch.empty((T, B, C, H, W), device=dev, dtype=torch.float32),
                ro=[torch.empty((T, B, self.target_size * (2 if i == self.num_layers - 1 else 1)), device=dev,
                                dtype=torch.float32) for i in range(self.num_layers)])
        return self._seq_buffers[key]

    def zero_states(self):
        """Zero every layer's neuron state in place (time constants untouched — unlike reset(True), quirk Q4)."""
        for s in self.dcll_slices:
            for t in s.dclllayer.i2h.state:
                t.zero_()

    @torch.no_grad()
    def test_sequence(self, cells=None, collect=True, profile=None, fuse_readout=False, iq=None, encoder=None,
                      T=None, t0=None):
        """Equivalent of `for t in range(T): net.test(x[t])` for input given as cell indices (T,B) int32 on device
        (one input spike per sample per step, what iq2spiketrain produces), or as the raw IQ batch `iq` (B,2,L) with an
        `IQEncoder` — then the quantisation runs inside the first layer's kernel (T steps from sample t0; t0 drawn
        like the host encoder when None).  Fills every slice's `clout`.

        `fuse_readout`: compute the readouts in the layer kernels' epilogue instead of materialising pv + a GEMM.
        Measured slower on MI355X (re-streaming the 786 KB readout matrix per sample-step through L2 costs more than
        the pv round trip through HBM: 124 vs 113 ms per layer launch at B=4096), hence off by default.
        `profile`: optional dict; (start, end) torch.cuda.Event pairs of every launch are appended under
        'lif_c1' / 'lif_c32' / 'readout' / 'vote' (events live on the current stream = the launch stream).

        Returns a dict with device tensors: 'logits' (per layer, (T,B,target); last entry = output_ layer),
        'clout' (per layer (T,B) int32) and 'vote' (per layer (B) int32)."""
        if not self.sequence_supported():
            raise ops._lib.DCLLUnsupported('no fused sequence kernel for this network geometry; use net.test(x[t])')
        if iq is not None:
            iq = iq.reshape(iq.shape[0], 2, -1)
            B = iq.shape[0]
            if t0 is None:
                t0 = np.random.randint(0, iq.shape[-1] - T + 1)       # same draw as iq2spiketrain
            first_input, first_kind, dev = (iq, encoder.thr_i, encoder.thr_q, int(t0)), 'iq', iq.device
        else:
            T, B = cells.shape
            first_input, first_kind, dev = cells.contiguous(), 'cells', cells.device
        buf = self._sequence_buffers(T, B, dev)

        def timed(key, fn, *a, **kw):
            if profile is None:
                return fn(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            profile.setdefault(key, []).append((e0, e1))
            return out

        cur = first_input
        res = dict(logits=[], clout=[], vote=[])
        for i, s in enumerate(self.dcll_slices):
            L = s.dclllayer
            last = (i == self.num_layers - 1)
            fused = fuse_readout and i > 0
            spk, pv, ro = timed('lif_c1' if i == 0 else 'lif_c32', L.forward_sequence, cur, T, B,
                                first_kind if i == 0 else 'packed', want_spikes=not last,
                                buffers=dict(spk=buf['spk'][i & 1], pv=buf['pv'], ro=buf['ro'][i]),
                                fuse_readout=fused)
            if fused:
                # readout(s) came out of the layer kernel's epilogue: (T,B,24) or, on the output layer, (T,B,48)
                p = ro[..., :self.target_size]
                logits = p
                if last:
                    logits = ro[..., self.target_size:]
                    res['o'] = logits
            else:
                # readout GEMM over all (t, b) rows; on the output layer i2o and output_ share ONE pass over pv
                pv2d = pv.reshape(T * B, -1)
                Wt, bias = L.stacked_readout()
                ro = timed('readout', ops.readout, pv2d, Wt, bias, out=buf['ro'][i].reshape(T * B, -1))
                ro = ro.reshape(T, B, -1)
                p = ro[..., :self.target_size]
                logits = p
                if last:
                    logits = ro[..., self.target_size:]
                    res['o'] = logits
            clout, vote = timed('vote', ops.argmax_vote, logits.contiguous())
            res['logits'].append(p)
            res['clout'].append(clout)
            res['vote'].append(vote)
            if collect:
                s.set_sequence_result(clout, T)
            cur = spk
        return res
