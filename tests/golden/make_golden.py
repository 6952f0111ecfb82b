#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ from the *imported* reference.

Runs ONLY in the build container (needs /root/reference); the outputs (.npz / .json, data only)
are committed, this script is committed, the reference's source is never copied.

Shims: an empty in-memory `apex`/`apex.amp`; `device='cpu'` in both reference
modules; `yaml.load` given SafeLoader (PyYAML >= 6).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz, meta.json
    python tests/golden/make_golden.py --only-r32 # only the 32x32-plane rollout (added for the tiled kernels)
    python tests/golden/make_golden.py --only-g9  # only the checkpoint fixture (added in round 2)
    python tests/golden/make_golden.py --only-g6b # only the production-geometry learning fixture (added in round 5)
    python tests/golden/make_golden.py --only-t1024 # only the T = 1024 rollout (added in round 5)
    python tests/golden/make_golden.py --only-refyaml # only the radio_ml_conv_ref.yaml rollout (added in round 5)
    python tests/golden/make_golden.py --only-r128 # only the 128x128-plane rollout (added in round 5)
    python tests/golden/make_golden.py --only-g6r # only the regularised / MSELoss learning steps (added in round 5)
    python tests/golden/make_golden.py --only-variants # only the arp 0 / scalar-tau rollouts at 16x16 (added in round 5)
    python tests/golden/make_golden.py --only-g7b # only the dense sequences (added in round 5)
    python tests/golden/make_golden.py --only-carry # only the two-batch state carry-over run (added in round 5)
    python tests/golden/make_golden.py --only-refyaml-variants # only the arp 0 / scalar-tau runs of radio_ml_conv_ref.yaml (round 5)
    python tests/golden/make_golden.py --only-g6d # only the dense-slice learning steps (added in round 6)
    python tests/golden/make_golden.py --only-g1x # only the layer-option cases: stride / dilation / groups / bias / act / spiking (round 6)

Fixture list: G1 single layer-steps, G2 three-layer rollouts, G3 iq2spiketrain,
G4 vote helpers, G5 load_network_spec, G6 train_dcll steps (reduced net), G6b train_dcll steps at the production geometry, G7 dense layer steps,
G8 image2spiketrain (seeded), G9 a reference-written .pth checkpoint and the reference's run after restoring it.
"""
import json
import os
import sys
import types
from argparse import Namespace

import numpy as np
import torch
import yaml

REF = os.environ.get("DCLL_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    apex = types.ModuleType("apex")
    apex.amp = types.ModuleType("apex.amp")
    sys.modules["apex"] = apex
    sys.modules["apex.amp"] = apex.amp
    _load = yaml.load
    yaml.load = lambda stream, Loader=None: _load(stream, Loader=yaml.SafeLoader)
    sys.path.insert(0, REF)
    import dcll.pytorch_libdcll as lib
    import networks as nets
    import data.utils as du
    lib.device = "cpu"
    nets.device = "cpu"
    return lib, nets, du


def seed(s=1):
    torch.manual_seed(s)
    np.random.seed(s)


def npy(t):
    return t.detach().cpu().numpy().copy()


def state_dict_np(module, prefix="sd/"):
    return {prefix + k: npy(v) for k, v in module.state_dict().items()}


def pack_bits(x):
    """(…, N) {0,1} float -> uint8 little-endian bit packing along the last axis."""
    return np.packbits(x.astype(np.uint8), axis=-1, bitorder="little")


# ---------------------------------------------------------------------------------------------
def g1_layer_steps(lib):
    """Single Conv2dDCLLlayer, 3 consecutive steps from zero state (state in(t) = state out(t-1)); several variants."""
    cases = {
        # name: (Cin, Cout, k, pad, pool, im_dims, wrp, random_tau, output_layer, B)
        "radio_l0": (1, 32, 7, 3, 1, (16, 16), 1.0, True, False, 3),
        "radio_l1": (32, 32, 7, 3, 1, (16, 16), 1.0, True, False, 1),
        "radio_l2_out": (32, 32, 7, 3, 1, (16, 16), 1.0, True, True, 1),
        "radio_norp": (32, 32, 7, 3, 1, (16, 16), 0.0, True, False, 1),
        "scalar_tau": (4, 8, 5, 2, 1, (12, 10), 1.0, False, False, 3),
        "mnist_l0": (1, 16, 7, 2, 2, (28, 28), 0.0, False, False, 2),
        "mnist_l2": (24, 32, 7, 2, 2, (11, 11), 0.0, True, True, 2),
        "ref_tuple": (2, 64, (1, 3), (0, 1), (1, 2), (1, 128), 1.0, True, False, 2),
        "pool3": (3, 5, 3, 1, 3, (9, 9), 0.5, False, False, 2),
    }
    out = {}
    meta = {}
    for name, (cin, cout, k, pad, pool, im, wrp, rtau, outl, B) in cases.items():
        seed(7)
        layer = lib.Conv2dDCLLlayer(cin, cout, kernel_size=k, padding=pad, pooling=pool, im_dims=im,
                                    target_size=24, alpha=.92, alphas=.85, alpharp=.65, wrp=wrp,
                                    act=torch.nn.Sigmoid(), lc_ampl=.5, random_tau=rtau, spiking=True,
                                    lc_dropout=False, output_layer=outl).init_hiddens(B)
        # make the conv drive strong enough that both spike values occur after step 0
        with torch.no_grad():
            layer.i2h.weight.mul_(30.0)
            layer.i2h.bias.mul_(0.05)
        pre = "g1/%s/" % name
        out.update(state_dict_np(layer, pre + "sd/"))
        for t in range(3):
            x = (torch.rand(B, cin, *im) < 0.15).float()
            o, p, pv, v = layer.forward(x)
            st_out = [npy(s) for s in layer.i2h.state]
            out[pre + "x%d" % t] = npy(x)
            for i, nm in enumerate(layer.i2h.state._fields):
                out[pre + "out_%s%d" % (nm, t)] = st_out[i]
            out[pre + "o%d" % t] = npy(o)
            out[pre + "p%d" % t] = npy(p)
            out[pre + "pv%d" % t] = npy(pv)
            out[pre + "v%d" % t] = npy(v)
        meta[name] = dict(cin=cin, cout=cout, k=k, pad=pad, pool=pool, im=im, wrp=wrp, random_tau=rtau,
                          output_layer=outl, B=B, alpharp=.65)
    np.savez_compressed(os.path.join(OUT, "g1_layer_steps.npz"), **out)
    return meta


def make_args(**kw):
    a = dict(netscale=1.0, alpha=.92, alphas=.85, alpharp=.65, arp=1.0, lc_ampl=.5, random_tau=True)
    a.update(kw)
    return Namespace(**a)


def synth_iq(B, L=128, s=1):
    g = torch.Generator().manual_seed(s)
    return 0.4 * torch.randn(B, 2, 1, L, generator=g)


def rollout(lib, nets, du, yaml_name, R, T, B, args, full_traces, weight_gain=1.0, store_readouts=True, L=128, iq_seed=1,
            hw=None, store_final=True):
    convs = nets.load_network_spec(os.path.join(REF, "networks", yaml_name))
    seed(1)
    H, Wd = (R, R) if hw is None else hw
    net = nets.ConvNetwork(args, (1, H, Wd), B, convs, 24, act=torch.nn.Sigmoid(), loss=None, opt=None,
                           opt_param={}, learning_rates=None, burnin=20)
    net.reset(True)
    if weight_gain != 1.0:
        with torch.no_grad():
            for s in net.dcll_slices:
                s.dclllayer.i2h.weight.mul_(weight_gain)
    x = synth_iq(B, L, iq_seed)
    labels = torch.randint(0, 24, (B,), generator=torch.Generator().manual_seed(2))
    y1h = du.to_one_hot(labels, 24)
    np.random.seed(3)
    spikes, targets = du.iq2spiketrain(x, y1h, out_w=Wd, out_h=H, max_duration=T)
    xin = torch.Tensor(spikes)
    out = {}
    for i, s in enumerate(net.dcll_slices):
        sd = state_dict_np(s.dclllayer, "sd/%d/" % i)
        if not store_readouts:
            # the frozen readout matrices of a large plane are megabytes of uniform noise that the seeded constructor
            # reproduces exactly (tests/test_host_logic.py): keep a float64 checksum instead of the matrix
            for k in list(sd):
                if k.split("/")[-1].startswith(("i2o.weight", "output_.weight")):
                    w = sd.pop(k).astype(np.float64)
                    out["sdsum/" + k[3:]] = np.array([w.sum(), np.abs(w).sum(), w.reshape(-1)[::4097].sum()])
        out.update(sd)
    out["iq"] = npy(x)
    out["labels"] = labels.numpy()
    cells = np.zeros((T, B), dtype=np.int32)
    for t in range(T):
        for b in range(B):
            cells[t, b] = int(np.flatnonzero(spikes[t, b, 0].reshape(-1))[0])
    out["cells"] = cells
    nl = len(net.dcll_slices)
    spk = [[] for _ in range(nl)]
    pl = [[] for _ in range(nl)]
    minabs = np.zeros((nl, T), dtype=np.float32)
    ol = []
    traces = {}
    net.reset()
    net.eval()
    for t in range(T):
        cur = xin[t]
        for i, s in enumerate(net.dcll_slices):
            if full_traces:
                for j, nm in enumerate(s.dclllayer.i2h.state._fields):
                    traces.setdefault("tr/%d/in_%s" % (i, nm), []).append(npy(s.dclllayer.i2h.state[j]))
            o, p, pv, v = s.forward(cur, ignore_burnin=True)
            pooled_s = s.dclllayer.pool((v > 0).float())
            spk[i].append(pack_bits(npy(pooled_s).reshape(B, -1)))
            pl[i].append(npy(p))
            minabs[i, t] = float(v.detach().abs().min())
            if full_traces:
                traces.setdefault("tr/%d/v" % i, []).append(npy(v))
                traces.setdefault("tr/%d/pv" % i, []).append(npy(pv))
            if s.dclllayer.output_layer:
                ol.append(npy(o))
            cur = o
    t1h = torch.Tensor(targets)
    accs = net.accuracy(t1h)
    conf = net.confusion_matrix(t1h)
    for i in range(nl):
        out["spikes/%d" % i] = np.stack(spk[i])          # (T,B,ceil(CHW/8)) uint8, C-major flatten
        out["p/%d" % i] = np.stack(pl[i])                 # (T,B,24)
        out["clout/%d" % i] = np.array(net.dcll_slices[i].clout)  # (T,B)
        pred, lab = lib.get_predictions_by_vote(net.dcll_slices[i].clout, t1h)
        out["vote/%d" % i] = pred.astype(np.int64)
    out["o_last"] = np.stack(ol)
    out["minabs_v"] = minabs
    out["acc"] = np.array(accs, dtype=np.float64)
    out["confusion"] = conf
    for k, vlist in traces.items():
        out[k] = np.stack(vlist)
    # final neuron state (for the state-carry-over quirk Q3); large planes: float64 checksums instead
    for i, s in enumerate(net.dcll_slices):
        for j, nm in enumerate(s.dclllayer.i2h.state._fields):
            st = npy(s.dclllayer.i2h.state[j])
            if store_final:
                out["final/%d/%s" % (i, nm)] = st
            else:
                w = st.astype(np.float64)
                out["finalsum/%d/%s" % (i, nm)] = np.array([w.sum(), np.abs(w).sum(), w.reshape(-1)[::997].sum()])
    return out


def g2_rollouts(lib, nets, du):
    full = rollout(lib, nets, du, "radio_ml_conv.yaml", R=16, T=128, B=2, args=make_args(), full_traces=False)
    np.savez_compressed(os.path.join(OUT, "g2_radio_r16_t128_b2.npz"), **full)
    small = rollout(lib, nets, du, "radio_ml_conv.yaml", R=8, T=32, B=3, args=make_args(netscale=0.25),
                    full_traces=True)
    np.savez_compressed(os.path.join(OUT, "g2_radio_r8_t32_b3_traces.npz"), **small)
    norp = rollout(lib, nets, du, "radio_ml_conv.yaml", R=8, T=24, B=2,
                   args=make_args(netscale=0.25, arp=0.0, random_tau=False), full_traces=True)
    np.savez_compressed(os.path.join(OUT, "g2_radio_r8_t24_b2_norp_traces.npz"), **norp)
    return {"full": dict(R=16, T=128, B=2), "small": dict(R=8, T=32, B=3, netscale=0.25),
            "norp": dict(R=8, T=24, B=2, netscale=0.25, arp=0.0, random_tau=False)}


def g2_r32(lib, nets, du):
    """radio_ml_conv.yaml on a 32x32 plane (4 tiles of the large-plane kernels, every tile touching the plane edge on
    three sides): T=40, B=2, free-running reference spikes / logits / argmax / votes / final state."""
    r32 = rollout(lib, nets, du, "radio_ml_conv.yaml", R=32, T=40, B=2, args=make_args(), full_traces=False,
                  store_readouts=False)
    np.savez_compressed(os.path.join(OUT, "g2_radio_r32_t40_b2.npz"), **r32)
    return dict(R=32, T=40, B=2)


def g2_t1024(lib, nets, du):
    """The reference's OWN sequence length (n_iters_test = 1024: train.py:63-66, scripts/test_radio_ml.sh:17-18; RadioML-2018
    windows are 1024 samples, data/utils.py:56-59): radio_ml_conv.yaml, 16x16, B = 2 windows of 1024 samples, all 1024 steps
    free-running — packed spikes of all three layers, readouts, per-step argmax, votes, final state.  IQ seed 852: the first
    of 2 500 seeds for which the reference's spike trains and those of the pinned summation order (oracle/dcll_oracle.c) agree
    on every one of the 3 x 2 x 8192 x 1024 neuron-steps (a band-internal tie-break of the two fp32 orders occurs in ~97 % of
    the windows within 1024 steps — SURVEY 7 H1 —, so a band-free fixture has to be looked for; the search compared the two
    CPU implementations only)."""
    r = rollout(lib, nets, du, "radio_ml_conv.yaml", R=16, T=1024, B=2, args=make_args(), full_traces=False,
                store_readouts=False, L=1024, iq_seed=852)
    r.pop("minabs_v")
    np.savez_compressed(os.path.join(OUT, "g2_radio_r16_t1024_b2.npz"), **r)
    return dict(R=16, T=1024, B=2, L=1024, iq_seed=852)


def g2_ref_yaml(lib, nets, du):
    """networks/radio_ml_conv_ref.yaml THROUGH THE DCLL BUILDER (ConvNetwork; the network of BASELINE config 5 before this
    build's int8 quantisation): seven layers of 64 channels, kernels (1,3), padding (0,1), max-pool (1,2), on a Q = 16 x I = 128
    plane — the reference can build it there (at 16x16 the width reaches 0 after four poolings).  fp32 weights as the
    reference initialises them, B = 2, T = 64, free-running: pooled packed spikes of all seven layers, readouts, argmax, votes.
    Pins the (1,3) / pooling kernels (k_lif_seq_w3, per-step path) against the reference itself; the int8 form of config 5 is
    then tied to it by "int8 through the ABI == the dequantised fp32 run" (no reference counterpart: parity unpinned)."""
    r = rollout(lib, nets, du, "radio_ml_conv_ref.yaml", R=None, T=64, B=2, args=make_args(), full_traces=False,
                store_readouts=False, hw=(16, 128), store_final=False)
    # the time constants are per-input-channel values broadcast over the plane (:398-405): stored per channel
    for k in list(r):
        if k.startswith("sd/") and k.split(".")[-1] in ("alpha", "tau_m__dt", "alphas", "tau_s__dt"):
            a = r[k]
            assert np.array_equal(a, np.broadcast_to(a[:, :1, :1], a.shape))
            r[k] = np.ascontiguousarray(a[:, 0, 0])
    np.savez_compressed(os.path.join(OUT, "g2_ref_yaml_h16_w128_t64_b2.npz"), **r)
    return dict(H=16, W=128, T=64, B=2, layers=7)


def g2_ref_yaml_variants(lib, nets, du, seeds=(1, 1)):
    """networks/radio_ml_conv_ref.yaml through the DCLL builder (g2_ref_yaml's network and plane) in the two settings the scripts
    do not use: `--arp 0` (the NON-refractory ContinuousConv2D, :407-426) with random_tau, and `random_tau=False` (scalar time
    constants, :349-356) with arp 1.  B = 2, T = 32.  Pins the REFRACTORY = false instantiations of the (1,3) kernels — the
    streaming first layer k_lif_seq_w3f included — and their scalar-tau use against the reference itself."""
    out = {}
    for tag, kw, sd_ in (("norp", dict(arp=0.0), seeds[0]), ("scalar_tau", dict(random_tau=False), seeds[1])):
        r = rollout(lib, nets, du, "radio_ml_conv_ref.yaml", R=None, T=32, B=2, args=make_args(**kw), full_traces=False,
                    store_readouts=False, hw=(16, 128), store_final=False, iq_seed=sd_)
        r.pop("minabs_v")
        for k in list(r):       # time constants broadcast over the plane (:398-405): stored per channel
            if k.startswith("sd/") and k.split(".")[-1] in ("alpha", "tau_m__dt", "alphas", "tau_s__dt") and r[k].ndim == 3:
                a = r[k]
                assert np.array_equal(a, np.broadcast_to(a[:, :1, :1], a.shape))
                r[k] = np.ascontiguousarray(a[:, 0, 0])
        out.update({tag + "/" + k: v for k, v in r.items()})
    np.savez_compressed(os.path.join(OUT, "g2_ref_yaml_h16_w128_t32_b2_variants.npz"), **out)
    return dict(H=16, W=128, T=32, B=2, layers=7, iq_seeds=list(seeds),
                variants=["norp (arp 0, random_tau)", "scalar_tau (arp 1, random_tau False)"])


def g2_r16_variants(lib, nets, du):
    """radio_ml_conv.yaml at the production geometry (32 channels, 16x16, B = 2, T = 64) in the two settings the scripts do
    not use but the argparse surface offers: `--arp 0` (train.py:84-85 default: the NON-refractory ContinuousConv2D,
    :407-426) with random_tau, and `random_tau=False` (scalar time constants, :349-356) with arp 1.  Band-free IQ seeds (1 / 2)."""
    a = rollout(lib, nets, du, "radio_ml_conv.yaml", R=16, T=64, B=2, args=make_args(arp=0.0), full_traces=False,
                store_readouts=False, store_final=False, iq_seed=1)
    b = rollout(lib, nets, du, "radio_ml_conv.yaml", R=16, T=64, B=2, args=make_args(random_tau=False), full_traces=False,
                store_readouts=False, store_final=False, iq_seed=2)
    out = {}
    for tag, r in (("norp", a), ("scalar_tau", b)):
        r.pop("minabs_v")
        out.update({tag + "/" + k: v for k, v in r.items()})
    np.savez_compressed(os.path.join(OUT, "g2_radio_r16_t64_b2_variants.npz"), **out)
    return dict(R=16, T=64, B=2, variants=["norp (arp 0, random_tau)", "scalar_tau (arp 1, random_tau False)"])


def g2_carry(lib, nets, du):
    """Quirk Q3: `net.reset()` between batches does NOT zero the neuron state (networks/__init__.py:187-189 passes
    init_states=False; dcll/pytorch_libdcll.py:648-653) — the second batch starts from the first batch's final state.  Two
    consecutive batches of B = 2 windows, T = 24 each, production geometry, the reference's evaluation protocol
    (test_radio_ml.py:142-146: reset, T x test, accuracy): spikes / readouts / argmax of the SECOND batch and its votes."""
    convs = nets.load_network_spec(os.path.join(REF, "networks", "radio_ml_conv.yaml"))
    seed(1)
    B, R, T = 2, 16, 24
    net = nets.ConvNetwork(make_args(), (1, R, R), B, convs, 24, act=torch.nn.Sigmoid(), loss=None, opt=None,
                           opt_param={}, learning_rates=None, burnin=20)
    net.reset(True)
    out = {}
    for i, s in enumerate(net.dcll_slices):
        sd = state_dict_np(s.dclllayer, "sd/%d/" % i)
        for k in list(sd):
            if k.split("/")[-1].startswith(("i2o.weight", "output_.weight")):
                w = sd.pop(k).astype(np.float64)
                out["sdsum/" + k[3:]] = np.array([w.sum(), np.abs(w).sum(), w.reshape(-1)[::4097].sum()])
        out.update(sd)
    for k in range(2):
        x = synth_iq(B, 128, 21 + k)
        labels = torch.randint(0, 24, (B,), generator=torch.Generator().manual_seed(5 + k))
        np.random.seed(30 + k)
        spikes, targets = du.iq2spiketrain(x, du.to_one_hot(labels, 24), out_w=R, out_h=R, max_duration=T)
        xin = torch.Tensor(spikes)
        out["cells/%d" % k] = spikes.reshape(T, B, -1).argmax(-1).astype(np.int32)
        out["labels/%d" % k] = labels.numpy()
        net.reset()                                              # (init_states=False: state carried over)
        spk = [[] for _ in net.dcll_slices]
        pl = [[] for _ in net.dcll_slices]
        for t in range(T):
            cur = xin[t]
            for i, s in enumerate(net.dcll_slices):
                o, p, pv, v = s.forward(cur, ignore_burnin=True)
                spk[i].append(pack_bits(npy((v > 0).float()).reshape(B, -1)))
                pl[i].append(npy(p))
                cur = o
        for i in range(3):
            out["spikes/%d/%d" % (k, i)] = np.stack(spk[i])
            out["p/%d/%d" % (k, i)] = np.stack(pl[i])
            out["clout/%d/%d" % (k, i)] = np.array(net.dcll_slices[i].clout)
        out["acc/%d" % k] = np.array(net.accuracy(torch.Tensor(targets)))
    np.savez_compressed(os.path.join(OUT, "g2_radio_r16_carry_over.npz"), **out)


def g2_r128(lib, nets, du):
    """radio_ml_conv.yaml on the ARGPARSE-DEFAULT 128x128 plane (train.py:37-40, test_radio_ml.py:52-53): B = 2, T = 12,
    free-running reference spikes / readouts / argmax / votes (the 32x32 fixture covers every tile-edge combination of the
    tiled kernels; this one the default geometry itself: 64 tiles per layer, interior tiles included).  Readout matrices
    (3 x 50 MB) and final state as float64 checksums."""
    r = rollout(lib, nets, du, "radio_ml_conv.yaml", R=128, T=12, B=2, args=make_args(), full_traces=False,
                store_readouts=False, store_final=False)
    for k in list(r):
        if k.startswith("sd/") and k.split(".")[-1] in ("alpha", "tau_m__dt", "alphas", "tau_s__dt"):
            a = r[k]
            assert np.array_equal(a, np.broadcast_to(a[:, :1, :1], a.shape))
            r[k] = np.ascontiguousarray(a[:, 0, 0])
    r.pop("minabs_v")
    np.savez_compressed(os.path.join(OUT, "g2_radio_r128_t12_b2.npz"), **r)
    return dict(R=128, T=12, B=2)


def g2_mnist(lib, nets, du):
    """BASELINE config 1 plumbing: mnist_conv.yaml, synthetic 28x28, T=50, B=4, arp=0."""
    convs = nets.load_network_spec(os.path.join(REF, "networks", "mnist_conv.yaml"))
    seed(1)
    B, T = 4, 50
    args = make_args(arp=0.0)
    net = nets.ConvNetwork(args, (1, 28, 28), B, convs, 10, act=torch.nn.Sigmoid(), loss=None, opt=None,
                           opt_param={}, learning_rates=None, burnin=20)
    net.reset(True)
    img = torch.rand(B, 28, 28, generator=torch.Generator().manual_seed(5))
    labels = torch.randint(0, 10, (B,), generator=torch.Generator().manual_seed(6))
    np.random.seed(4)
    spikes, targets = du.image2spiketrain(img.numpy(), du.to_one_hot(labels, 10).numpy(), (1, 28, 28),
                                          gain=100, min_duration=T - 1, max_duration=T)
    xin = torch.Tensor(spikes)
    out = {"img": img.numpy(), "labels": labels.numpy(), "x": pack_bits(spikes.reshape(T, B, -1))}
    for i, s in enumerate(net.dcll_slices):
        out.update(state_dict_np(s.dclllayer, "sd/%d/" % i))
    nl = len(net.dcll_slices)
    spk = [[] for _ in range(nl)]
    pl = [[] for _ in range(nl)]
    ol = []
    net.reset()
    for t in range(T):
        cur = xin[t]
        for i, s in enumerate(net.dcll_slices):
            o, p, pv, v = s.forward(cur, ignore_burnin=True)
            spk[i].append(pack_bits(npy(s.dclllayer.pool((v > 0).float())).reshape(B, -1)))
            pl[i].append(npy(p))
            if s.dclllayer.output_layer:
                ol.append(npy(o))
            cur = o
    for i in range(nl):
        out["spikes/%d" % i] = np.stack(spk[i])
        out["p/%d" % i] = np.stack(pl[i])
        out["clout/%d" % i] = np.array(net.dcll_slices[i].clout)
    out["o_last"] = np.stack(ol)
    out["acc"] = np.array(net.accuracy(torch.Tensor(targets)))
    np.savez_compressed(os.path.join(OUT, "g2_mnist_t50_b4.npz"), **out)
    return dict(B=B, T=T)


def g3_iq(du):
    out = {}
    seed(1)
    # boundary / out-of-range / exact values
    special = np.array([-2.0, -1.0, -0.999999, -0.5, -1e-3, -0.0, 0.0, 1e-7, 1e-3, 0.25, 0.5, 0.7071, 0.999999,
                        1.0, 1.5, 3.0], dtype=np.float32)
    for R in (16, 28, 128):
        B, L = 8, 64
        x = 0.5 * torch.randn(B, 2, 1, L)
        x[0, 0, 0, :16] = torch.from_numpy(special)
        x[0, 1, 0, :16] = torch.from_numpy(special[::-1].copy())
        # values sitting near cell boundaries of the gamma curve
        cells = np.arange(1, R, dtype=np.float64) / (R - 1)
        bnd = np.sign(2 * cells - 1) * np.abs(2 * cells - 1) ** 1.2
        n = min(L, len(bnd))
        x[1, 0, 0, :n] = torch.from_numpy(bnd[:n].astype(np.float32))
        x[1, 1, 0, :n] = torch.from_numpy(np.nextafter(bnd[:n].astype(np.float32), np.float32(-2)))
        y = du.to_one_hot(torch.arange(B) % 24, 24)
        for T in (L, L // 2):
            np.random.seed(11)
            st, tg = du.iq2spiketrain(x, y, out_w=R, out_h=R, max_duration=T)
            np.random.seed(11)
            t0 = np.random.randint(0, L - T + 1)
            assert st.sum() == T * B
            idx = st.reshape(T, B, -1).argmax(-1)
            out["R%d_T%d/x" % (R, T)] = npy(x)
            out["R%d_T%d/cell" % (R, T)] = idx.astype(np.int32)      # q*R + i
            out["R%d_T%d/t0" % (R, T)] = np.array(t0)
            out["R%d_T%d/target" % (R, T)] = np.asarray(tg, dtype=np.float32)
    # non-square plane, non-default bounds, no gamma
    x = 0.7 * torch.randn(5, 2, 1, 32)
    y = du.to_one_hot(torch.arange(5) % 24, 24)
    np.random.seed(12)
    st, _ = du.iq2spiketrain(x, y, out_w=20, out_h=12, min_I=-2, max_I=1.5, min_Q=-0.5, max_Q=0.75,
                             max_duration=32, do_gamma=False)
    out["rect/x"] = npy(x)
    out["rect/cell"] = st.reshape(32, 5, -1).argmax(-1).astype(np.int32)
    np.savez_compressed(os.path.join(OUT, "g3_iq2spiketrain.npz"), **out)


def g4_votes(lib):
    rng = np.random.RandomState(5)
    T, B, C = 9, 64, 5
    clout = [rng.randint(0, C, size=B) for _ in range(T)]
    # force ties: sample 0 alternates, sample 1 has a late majority tie
    for t in range(T):
        clout[t][0] = [3, 1, 3, 1, 2, 2, 4, 4, 0][t]
        clout[t][1] = [2, 2, 0, 0, 1, 1, 3, 3, 4][t]
    lab = rng.randint(0, C, size=B)
    y = torch.zeros(T, B, C)
    y[:, np.arange(B), lab] = 1
    pred, labv = lib.get_predictions_by_vote(clout, y)
    acc = lib.accuracy_by_vote(clout, y)
    np.savez_compressed(os.path.join(OUT, "g4_votes.npz"), clout=np.array(clout), labels=lab,
                        pred=pred.astype(np.int64), labv=labv.astype(np.int64), acc=np.array(acc))


def g5_specs(nets):
    res = {}
    for n in ("radio_ml_conv.yaml", "mnist_conv.yaml", "radio_ml_conv_ref.yaml"):
        res[n] = nets.load_network_spec(os.path.join(REF, "networks", n))
    return res


def g6_train_step(lib, nets, du):
    """Three post-burn-in train_dcll steps on the reduced radio net (for SURVEY 8(f)-2)."""
    convs = nets.load_network_spec(os.path.join(REF, "networks", "radio_ml_conv.yaml"))
    seed(1)
    B, R, T = 3, 8, 6
    args = make_args(netscale=0.25)
    opt_param = {"betas": [0.0, .95], "weight_decay": 10.0}
    net = nets.ConvNetwork(args, (1, R, R), B, convs, 24, act=torch.nn.Sigmoid(), loss=torch.nn.SmoothL1Loss,
                           opt=torch.optim.Adam, opt_param=opt_param, learning_rates=[1e-6], burnin=3)
    net.reset(True)
    x = synth_iq(B, 128, 4)
    labels = torch.randint(0, 24, (B,), generator=torch.Generator().manual_seed(2))
    np.random.seed(3)
    spikes, targets = du.iq2spiketrain(x, du.to_one_hot(labels, 24), out_w=R, out_h=R, max_duration=T)
    xin, tg = torch.Tensor(spikes), torch.Tensor(targets)
    out = {"x": spikes.astype(np.float32), "targets": np.asarray(targets, dtype=np.float32)}
    for i, s in enumerate(net.dcll_slices):
        out.update(state_dict_np(s.dclllayer, "sd0/%d/" % i))
    net.reset()
    net.train()
    for t in range(T):
        net.learn(xin[t], tg[t])
        for i, s in enumerate(net.dcll_slices):
            if s.iter >= s.burnin:
                out["grad/%d/%d/w" % (t, i)] = npy(s.dclllayer.i2h.weight.grad)
                out["grad/%d/%d/b" % (t, i)] = npy(s.dclllayer.i2h.bias.grad)
    for i, s in enumerate(net.dcll_slices):
        out.update(state_dict_np(s.dclllayer, "sd1/%d/" % i))
    np.savez_compressed(os.path.join(OUT, "g6_train_steps.npz"), **out)


def g6r_train_variants(lib, nets, du):
    """Two more train_dcll variants on the reduced radio net of G6 (netscale 0.25, 8x8, B = 3, T = 6, burn-in 3, Adam betas
    (0, .95), weight_decay 10, lr 1e-6):
      reg/  SmoothL1Loss with train_dcll's DEFAULT regularize = 0.05 (dcll/pytorch_libdcll.py:690, :697-701: the two regulariser
            terms reach pvmem and pv directly) — the slices called one after the other as ConvNetwork.learn does, but with the
            default argument;
      mse/  MSELoss (train.py --loss_type MSELoss), regularize = False as ConvNetwork.learn passes it.
    Per variant: initial state dicts, the gradients of every post-burn-in step, the losses train_dcll returns, final state
    dicts (inputs and targets are G6's)."""
    convs = nets.load_network_spec(os.path.join(REF, "networks", "radio_ml_conv.yaml"))
    B, R, T = 3, 8, 6
    out = {}
    for tag, loss, reg in (("reg", torch.nn.SmoothL1Loss, 0.05), ("mse", torch.nn.MSELoss, False)):
        seed(1)
        net = nets.ConvNetwork(make_args(netscale=0.25), (1, R, R), B, convs, 24, act=torch.nn.Sigmoid(), loss=loss,
                               opt=torch.optim.Adam, opt_param={"betas": [0.0, .95], "weight_decay": 10.0},
                               learning_rates=[1e-6], burnin=3)
        net.reset(True)
        x = synth_iq(B, 128, 4)
        labels = torch.randint(0, 24, (B,), generator=torch.Generator().manual_seed(2))
        np.random.seed(3)
        spikes, targets = du.iq2spiketrain(x, du.to_one_hot(labels, 24), out_w=R, out_h=R, max_duration=T)
        xin, tg = torch.Tensor(spikes), torch.Tensor(targets)
        if tag == "reg":
            out["x"] = spikes.astype(np.float32)
            out["targets"] = np.asarray(targets, dtype=np.float32)
        for i, s in enumerate(net.dcll_slices):
            out.update(state_dict_np(s.dclllayer, "%s/sd0/%d/" % (tag, i)))
        net.reset()
        net.train()
        for t in range(T):
            cur = xin[t]
            for i, s in enumerate(net.dcll_slices):
                cur, _, _, _, l = s.train_dcll(cur, tg[t], regularize=reg)
                out["%s/loss/%d/%d" % (tag, t, i)] = npy(l).reshape(-1)
                if s.iter >= s.burnin:
                    out["%s/grad/%d/%d/w" % (tag, t, i)] = npy(s.dclllayer.i2h.weight.grad)
                    out["%s/grad/%d/%d/b" % (tag, t, i)] = npy(s.dclllayer.i2h.bias.grad)
        for i, s in enumerate(net.dcll_slices):
            out.update(state_dict_np(s.dclllayer, "%s/sd1/%d/" % (tag, i)))
    np.savez_compressed(os.path.join(OUT, "g6r_train_variants.npz"), **out)


def g6b_train_production(lib, nets, du):
    """Eight consecutive train_dcll steps of the reference at the PRODUCTION geometry (round-4 verdict, weak #1):
    radio_ml_conv.yaml, netscale 1 (32 channels), 16x16 plane, arp 1.0, random_tau, SmoothL1 + Adam(betas (0,.95),
    weight_decay 10, lr 1e-6 = train.py's default) + the output layer's optimizer2 (Adam, lr 1e-4), B = 8, burn-in 20,
    T = 27 -> iter 20..27 learn with the neuron state and the Adam moments carried from step to step
    (dcll/pytorch_libdcll.py:690-718, networks/__init__.py:176-180, train.py:249-251).
    Stored: input cells, labels, the initial conv tensors + time constants (the i2o tensors and the initial output_.weight
    as SHA-256: 3.1 MB of seeded uniforms that the seeded construction reproduces bit for bit), per step the packed spikes of
    the two hidden layers and every readout, the gradients of the first / a middle / the last learning step, the
    final trainable tensors."""
    import hashlib
    convs = nets.load_network_spec(os.path.join(REF, "networks", "radio_ml_conv.yaml"))
    seed(1)
    B, R, T, burnin = 8, 16, 27, 20
    args = make_args()
    opt_param = {"betas": [0.0, .95], "weight_decay": 10.0}
    net = nets.ConvNetwork(args, (1, R, R), B, convs, 24, act=torch.nn.Sigmoid(), loss=torch.nn.SmoothL1Loss,
                           opt=torch.optim.Adam, opt_param=opt_param, learning_rates=[1e-6], burnin=burnin)
    net.reset(True)
    x = synth_iq(B, 128, 11)
    labels = torch.randint(0, 24, (B,), generator=torch.Generator().manual_seed(12))
    np.random.seed(13)
    spikes, targets = du.iq2spiketrain(x, du.to_one_hot(labels, 24), out_w=R, out_h=R, max_duration=T)
    xin, tg = torch.Tensor(spikes), torch.Tensor(targets)
    assert spikes.reshape(T, B, -1).sum(-1).min() == 1 == spikes.reshape(T, B, -1).sum(-1).max()
    out = {"cells": spikes.reshape(T, B, -1).argmax(-1).astype(np.int32), "labels": labels.numpy().astype(np.int64)}
    sha = {}
    for i, s in enumerate(net.dcll_slices):
        for k, v in state_dict_np(s.dclllayer, "sd0/%d/" % i).items():
            if "/i2o." in k or k.endswith("/output_.weight"):
                sha[k] = hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()
            else:
                out[k] = v
    net.reset()
    net.train()
    learn_steps = [t for t in range(T) if t + 1 >= burnin]
    keep = {learn_steps[0], learn_steps[len(learn_steps) // 2], learn_steps[-1]}
    nl = len(net.dcll_slices)
    spk = [[] for _ in range(nl - 1)]
    pl = [[] for _ in range(nl)]
    ol = []
    for t in range(T):
        cur = xin[t]
        for i, s in enumerate(net.dcll_slices):       # ConvNetwork.learn, unrolled to see every slice's outputs
            cur, p, _, _, _ = s.train_dcll(cur, tg[t], regularize=False)
            pl[i].append(npy(p))
            if s.dclllayer.output_layer:
                ol.append(npy(cur))
            else:
                spk[i].append(pack_bits(npy(cur).reshape(B, -1)))
            if t in keep:
                out["grad/%d/%d/w" % (t, i)] = npy(s.dclllayer.i2h.weight.grad)
                out["grad/%d/%d/b" % (t, i)] = npy(s.dclllayer.i2h.bias.grad)
                if s.dclllayer.output_layer:
                    out["grad/%d/%d/ow" % (t, i)] = npy(s.dclllayer.output_.weight.grad)
                    out["grad/%d/%d/ob" % (t, i)] = npy(s.dclllayer.output_.bias.grad)
    for i in range(nl):
        out["p/%d" % i] = np.stack(pl[i])
        out["clout/%d" % i] = np.array(net.dcll_slices[i].clout)
        if i < nl - 1:
            out["spikes/%d" % i] = np.stack(spk[i])
    out["o_last"] = np.stack(ol)
    for i, s in enumerate(net.dcll_slices):
        for k, v in state_dict_np(s.dclllayer, "sd1/%d/" % i).items():
            if "/i2h.weight" in k or "/i2h.bias" in k or "/output_." in k:
                out[k] = v
            elif "/i2o." in k:                      # frozen: must still hash to the initial tensor
                assert hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() == sha[k.replace("sd1/", "sd0/")]
        arp = npy(s.dclllayer.i2h.state.arp)
        out["final_arp_sum/%d" % i] = np.array([np.abs(arp.astype(np.float64)).sum()])
    np.savez_compressed(os.path.join(OUT, "g6b_train_production.npz"), **out)
    return dict(B=B, R=R, T=T, burnin=burnin, lr=1e-6, learn_steps=learn_steps, grad_steps=sorted(keep),
                i2o_sha256=sha)


def g7_dense(lib):
    out = {}
    for name, wrp, rtau in (("rrp", 1.0, False), ("plain", 0.0, False), ("plain_rtau", 0.0, True)):
        seed(9)
        layer = lib.DenseDCLLlayer(40, 24, target_size=10, alpha=.9, alphas=.85, alpharp=.65, wrp=wrp,
                                   random_tau=rtau).init_hiddens(5)
        with torch.no_grad():
            layer.i2h.weight.mul_(200.0)
            layer.i2h.bias.mul_(0.02)
        pre = "g7/%s/" % name
        out.update(state_dict_np(layer, pre + "sd/"))
        for t in range(3):
            x = (torch.rand(5, 40) < 0.2).float()
            o, p, pv, v = layer.forward(x)
            out[pre + "x%d" % t] = npy(x)
            out[pre + "o%d" % t] = npy(o)
            out[pre + "p%d" % t] = npy(p)
            out[pre + "pv%d" % t] = npy(pv)
            out[pre + "v%d" % t] = npy(v)
            for i, nm in enumerate(layer.i2h.state._fields):
                out[pre + "out_%s%d" % (nm, t)] = npy(layer.i2h.state[i])
    np.savez_compressed(os.path.join(OUT, "g7_dense.npz"), **out)


def g7b_dense_sequence(lib, x_seeds=(11, 12)):
    """DenseDCLLlayer over T = 24 steps at sizes the two dense kernel forms serve (the all-T on-chip kernel: in <= 1024,
    out <= 128; the per-step fp32-MFMA GEMM beyond): (512 -> 128, refractory, random_tau) and (600 -> 160, plain, scalar
    tau; odd tile counts), B = 5, target 10 — input spikes, output spikes, readouts per step, final state.  x_seeds: input
    seeds for which the reference's and the pinned-order spike trains agree on every neuron-step (tests/test_oracle_c.py)."""
    out = {}
    for (name, cin, cout, wrp, rtau), xs in zip((("rrp_512_128", 512, 128, 1.0, True), ("plain_600_160", 600, 160, 0.0, False)), x_seeds):
        seed(9)
        layer = lib.DenseDCLLlayer(cin, cout, target_size=10, alpha=.9, alphas=.85, alpharp=.65, wrp=wrp,
                                   random_tau=rtau).init_hiddens(5)
        with torch.no_grad():
            layer.i2h.weight.mul_(60.0)
            layer.i2h.bias.mul_(0.02)
        pre = "g7b/%s/" % name
        out.update(state_dict_np(layer, pre + "sd/"))
        g = torch.Generator().manual_seed(xs)
        T, B = 24, 5
        xs_, ss, ps = [], [], []
        for t in range(T):
            x = (torch.rand(B, cin, generator=g) < 0.2).float()
            o, p, pv, v = layer.forward(x)
            xs_.append(pack_bits(npy(x)))
            ss.append(pack_bits(npy(o)))
            ps.append(npy(p))
        out[pre + "x"] = np.stack(xs_)
        out[pre + "s"] = np.stack(ss)
        out[pre + "p"] = np.stack(ps)
        for i, nm in enumerate(layer.i2h.state._fields):
            out[pre + "final_%s" % nm] = npy(layer.i2h.state[i])
    np.savez_compressed(os.path.join(OUT, "g7b_dense_sequence.npz"), **out)


def g6d_dense_learning(lib):
    """DCLLBase.train_dcll on a DENSE slice (round-5 verdict, missing #2): DCLLClassification(DenseDCLLlayer(512, 128, target 24)),
    SmoothL1Loss, Adam(betas (0, .95), weight_decay 10, lr 1e-5), B = 8, burn-in 4, T = 9 -> iter 4..9 learn (six steps, the
    neuron state and the Adam moments carried along), regularize = False as ConvNetwork.learn passes it
    (dcll/pytorch_libdcll.py:198-255, :634-635, :690-718).  Variants: rrp (wrp 1, random_tau=True — which the refractory dense
    module never applies: its init_state :160-169 does not call randomize_tau), plain (wrp 0, scalar tau), plain_rtau (wrp 0,
    random_tau: per-feature time constants), reg (rrp with train_dcll's default regularize = 0.05: the regularisers reach
    pvmem and pv directly).  Per variant: initial state dict, packed input / output spikes, pvoutput and loss per step,
    gradients of i2h.weight / i2h.bias at the first and the last learning step, final state dict and neuron state."""
    out = {}
    B, T, target, burnin = 8, 9, 24, 4
    # (rrp / reg at 512 -> 128, the two non-refractory variants at 256 -> 64 with odd tile counts left to G7b: fixture size)
    for name, cin, cout, wrp, rtau, reg in (("rrp", 512, 128, 1.0, True, False), ("plain", 256, 64, 0.0, False, False),
                                            ("plain_rtau", 200, 72, 0.0, True, False), ("reg", 512, 128, 1.0, False, 0.05)):
        seed(9)
        layer = lib.DenseDCLLlayer(cin, cout, target_size=target, alpha=.9, alphas=.85, alpharp=.65, wrp=wrp, random_tau=rtau)
        with torch.no_grad():
            layer.i2h.weight.mul_(60.0)
            layer.i2h.bias.mul_(0.02)
        sl = lib.DCLLClassification(dclllayer=layer, name="dense", batch_size=B, loss=torch.nn.SmoothL1Loss,
                                    optimizer=torch.optim.Adam,
                                    kwargs_optimizer={"lr": 1e-5, "betas": [0.0, .95], "weight_decay": 10.0}, burnin=burnin)
        pre = "g6d/%s/" % name
        out.update(state_dict_np(layer, pre + "sd0/"))
        g = torch.Generator().manual_seed(21)
        labels = torch.randint(0, target, (B,), generator=g)
        tgt = torch.zeros(B, target)
        tgt[torch.arange(B), labels] = 1
        out[pre + "target"] = npy(tgt)
        xs, ss, ps, losses = [], [], [], []
        sl.train()
        for t in range(T):
            x = (torch.rand(B, cin, generator=g) < 0.2).float()
            o, p, pv, v, l = sl.train_dcll(x, tgt, regularize=reg)
            xs.append(pack_bits(npy(x)))
            ss.append(pack_bits(npy(o)))
            ps.append(npy(p))
            losses.append(npy(l).reshape(-1)[0])
            if sl.iter == sl.burnin or t == T - 1:          # the first and the last learning step
                out[pre + "grad/%d/w" % t] = npy(layer.i2h.weight.grad)
                out[pre + "grad/%d/b" % t] = npy(layer.i2h.bias.grad)
        out[pre + "x"], out[pre + "s"], out[pre + "p"] = np.stack(xs), np.stack(ss), np.stack(ps)
        out[pre + "loss"] = np.asarray(losses, dtype=np.float32)
        out[pre + "clout"] = np.stack(sl.clout)
        out.update(state_dict_np(layer, pre + "sd1/"))
        for i, nm in enumerate(layer.i2h.state._fields):
            out[pre + "final_%s" % nm] = npy(layer.i2h.state[i])
    np.savez_compressed(os.path.join(OUT, "g6d_dense_learning.npz"), **out)


# ---------------------------------------------------------------------------------------------
# G1x (round 6): the constructor options ConvNetwork never uses — stride / dilation / groups other than 1, bias=False, an
# activation other than nn.Sigmoid(), spiking=False (dcll/pytorch_libdcll.py:299-313, :407-426, :485-509, :75-148, :599-608).
G1X_ACTS = {0: torch.nn.Sigmoid, 1: torch.nn.Tanh, 2: torch.nn.ReLU}
# cfg row (int32): kind (0 Conv2dDCLLlayer, 1 ContinuousConv2D / ...RefractoryConv2D alone, 2 DenseDCLLlayer), cin, cout, kh, kw,
#                  pad_h, pad_w, pool_h, pool_w, H, W, stride, dilation, groups, bias, spiking, act id, wrp * 100, random_tau,
#                  output_layer, B, learn (1: two train_dcll steps instead of three forward steps)
G1X_CASES = {
    "stride2_rrp":      (0, 4, 6, 3, 3, 1, 1, 1, 1, 11, 12, 2, 1, 1, 1, 1, 0, 100, 1, 0, 3, 0),
    "dil2_pool2":       (0, 3, 5, 3, 3, 2, 2, 2, 2, 10, 12, 1, 2, 1, 1, 1, 0, 0, 0, 0, 2, 0),
    "stride2_dil2_out": (0, 4, 8, 3, 3, 2, 2, 1, 1, 13, 13, 2, 2, 1, 1, 1, 0, 50, 1, 1, 2, 0),
    "stride3_k5":       (0, 2, 4, 5, 5, 2, 2, 1, 1, 17, 16, 3, 1, 1, 1, 1, 0, 0, 1, 0, 2, 0),
    "tanh_plain":       (0, 4, 6, 3, 3, 1, 1, 2, 2, 12, 12, 1, 1, 1, 1, 1, 1, 0, 1, 0, 3, 0),
    "tanh_rrp_out":     (0, 4, 6, 3, 3, 1, 1, 1, 1, 8, 8, 1, 1, 1, 1, 1, 1, 100, 0, 1, 2, 0),
    "relu_stride2":     (0, 3, 6, 3, 3, 1, 1, 1, 1, 12, 10, 2, 1, 1, 1, 1, 2, 0, 0, 0, 2, 0),
    "nonspiking":       (0, 4, 6, 3, 3, 1, 1, 2, 2, 12, 12, 1, 1, 1, 1, 0, 0, 0, 1, 0, 3, 0),
    "nonspiking_out":   (0, 4, 6, 5, 5, 2, 2, 1, 1, 9, 9, 1, 1, 1, 1, 0, 0, 0, 0, 1, 2, 0),
    "i2h_groups2":      (1, 4, 6, 3, 3, 1, 1, 1, 1, 10, 10, 1, 1, 2, 1, 1, 0, 0, 1, 0, 3, 0),
    "i2h_groups3_s2_rrp": (1, 6, 9, 3, 3, 1, 1, 1, 1, 11, 9, 2, 1, 3, 1, 1, 0, 100, 1, 0, 2, 0),
    "i2h_nobias":       (1, 4, 6, 3, 3, 1, 1, 1, 1, 10, 10, 1, 1, 1, 0, 1, 0, 0, 0, 0, 3, 0),
    "i2h_nobias_groups2_dil2_tanh": (1, 4, 4, 3, 3, 2, 2, 1, 1, 10, 10, 1, 2, 2, 0, 1, 1, 100, 1, 0, 2, 0),
    "i2h_depthwise":    (1, 5, 5, 3, 3, 1, 1, 1, 1, 9, 9, 1, 1, 5, 1, 1, 0, 0, 0, 0, 2, 0),
    "dense_nobias":     (2, 40, 24, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 0, 1, 0, 0, 0, 0, 5, 0),
    "dense_tanh_rrp":   (2, 40, 24, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 100, 0, 0, 5, 0),
    "dense_nonspiking": (2, 40, 24, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 1, 0, 5, 0),
    "learn_stride2_tanh": (0, 4, 6, 3, 3, 1, 1, 1, 1, 11, 12, 2, 1, 1, 1, 1, 1, 100, 1, 1, 4, 1),
    "learn_dil2_nonspiking": (0, 3, 5, 3, 3, 2, 2, 2, 2, 10, 12, 1, 2, 1, 1, 0, 0, 0, 0, 0, 4, 1),
    "learn_dense_nobias_tanh": (2, 40, 24, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 0, 1, 1, 0, 0, 0, 5, 1),
}


def g1x_layer_options(lib):
    """Three consecutive steps from zero state (or, `learn` cases, two train_dcll steps with burn-in 1: the second one learns)
    of layers built with the options above; per case: cfg, the initial state dict, inputs, (output, pvoutput, pv, pvmem) and the
    neuron state per step; learn cases: target, gradients of i2h.weight (/ i2h.bias, output_.*) of the learning step, loss."""
    out = {}
    for name, cfg in G1X_CASES.items():
        (kind, cin, cout, kh, kw, pah, paw, poh, pow_, H, W, stride, dil, groups, bias, spiking, act_id, wrp100, rtau, outl, B,
         learn) = cfg
        seed(11)
        act, wrp = G1X_ACTS[act_id](), wrp100 / 100.0
        pre = "g1x/%s/" % name
        out[pre + "cfg"] = np.asarray(cfg, dtype=np.int32)
        if kind == 0:
            layer = lib.Conv2dDCLLlayer(cin, cout, kernel_size=(kh, kw), padding=(pah, paw), pooling=(poh, pow_), im_dims=(H, W),
                                        target_size=7, stride=stride, dilation=dil, alpha=.92, alphas=.85, alpharp=.65, wrp=wrp,
                                        act=act, lc_ampl=.5, random_tau=bool(rtau), spiking=bool(spiking), lc_dropout=False,
                                        output_layer=bool(outl)).init_hiddens(B)
            mod, i2h, shape = layer, layer.i2h, (cin, H, W)
        elif kind == 1:
            kw_ = dict(stride=stride, padding=(pah, paw), dilation=dil, groups=groups, bias=bool(bias), alpha=.92, alphas=.85,
                       act=act, random_tau=bool(rtau))
            i2h = (lib.ContinuousRelativeRefractoryConv2D(cin, cout, (kh, kw), alpharp=.65, wrp=wrp, **kw_) if wrp > 0 else
                   lib.ContinuousConv2D(cin, cout, (kh, kw), spiking=bool(spiking), **kw_))
            i2h.init_state(B, (H, W))
            mod, shape = i2h, (cin, H, W)
        else:
            layer = lib.DenseDCLLlayer(cin, cout, target_size=7, bias=bool(bias), alpha=.9, alphas=.85, alpharp=.65, wrp=wrp, act=act,
                                       spiking=bool(spiking), random_tau=bool(rtau)).init_hiddens(B)
            mod, i2h, shape = layer, layer.i2h, (cin,)
        with torch.no_grad():               # a drive strong enough that both spike values occur
            i2h.weight.mul_(200.0 if kind == 2 else 40.0)
            if i2h.bias is not None:
                i2h.bias.mul_(0.02 if kind == 2 else 0.05)
        out.update(state_dict_np(mod, pre + "sd/"))
        g = torch.Generator().manual_seed(5)
        if learn:
            sl = lib.DCLLClassification(dclllayer=layer, name="g1x", batch_size=B, loss=torch.nn.SmoothL1Loss,
                                        optimizer=torch.optim.Adam,
                                        kwargs_optimizer={"lr": 1e-6, "betas": [0.0, .95], "weight_decay": 10.0}, burnin=2)
            out.update(state_dict_np(mod, pre + "sd/"))     # (DCLLBase.init re-ran init_hiddens: random_tau draws again)
            labels = torch.randint(0, 7, (B,), generator=g)
            tgt = torch.zeros(B, 7)
            tgt[torch.arange(B), labels] = 1
            out[pre + "target"] = npy(tgt)
            sl.train()
        for t in range(2 if learn else 3):
            x = (torch.rand(B, *shape, generator=g) < 0.2).float()
            out[pre + "x%d" % t] = npy(x)
            if learn:
                o, p, pv, v, l = sl.train_dcll(x, tgt, regularize=False)
                out[pre + "loss%d" % t] = npy(l).reshape(-1)[:1]
            elif kind == 1:
                o, pv, v = mod.forward(x)
                p = None
            else:
                o, p, pv, v = mod.forward(x)
            for nm, val in (("o", o), ("p", p), ("pv", pv), ("v", v)):
                if val is not None:
                    out[pre + "%s%d" % (nm, t)] = npy(val)
            for i, nm in enumerate(i2h.state._fields):
                out[pre + "out_%s%d" % (nm, t)] = npy(i2h.state[i])
        if learn:
            for pn, prm in mod.named_parameters():
                if prm.grad is not None:
                    out[pre + "grad/" + pn] = npy(prm.grad)
            out.update(state_dict_np(mod, pre + "sd1/"))
    np.savez_compressed(os.path.join(OUT, "g1x_layer_options.npz"), **out)


def g8_image(du):
    rng = np.random.RandomState(3)
    x = rng.rand(3, 6, 6).astype(np.float32)
    y = np.eye(10, dtype=np.float32)[[1, 4, 7]]
    np.random.seed(21)
    a, tg = du.image2spiketrain(x, y, (1, 6, 6), gain=100, min_duration=19, max_duration=20)
    np.savez_compressed(os.path.join(OUT, "g8_image2spiketrain.npz"), x=x, y=y,
                        spikes=pack_bits(a.reshape(20, 3, -1)), target=np.asarray(tg))


def g9_checkpoint(lib, nets, du):
    """A checkpoint written by the reference's own protocol (train.py:300-303: torch.save(net.cpu().state_dict(), path))
    and what the reference computes after restoring it the way test_radio_ml.py does (:104-110: load_state_dict, then
    net.reset(True) — which re-draws the time constants, quirk Q4).  radio_ml_conv.yaml, netscale 0.25, 8x8 plane: the
    .pth is data (parameter tensors), 0.2 MB."""
    convs = nets.load_network_spec(os.path.join(REF, "networks", "radio_ml_conv.yaml"))
    args = make_args(netscale=0.25)
    B, R, T = 3, 8, 16
    seed(21)
    trained = nets.ConvNetwork(args, (1, R, R), B, convs, 24, act=torch.nn.Sigmoid(), loss=None, opt=None,
                               opt_param={}, learning_rates=None, burnin=2)
    trained.reset(True)
    path = os.path.join(OUT, "g9_reference_parameters.pth")
    torch.save(trained.cpu().state_dict(), path)
    seed(99)                                            # a differently initialised network, then restore
    net = nets.ConvNetwork(args, (1, R, R), B, convs, 24, act=torch.nn.Sigmoid(), loss=None, opt=None,
                           opt_param={}, learning_rates=None, burnin=2)
    net.load_state_dict(torch.load(path))
    np.random.seed(7)
    net.reset(True)
    x = synth_iq(B, 128, 4)
    labels = torch.randint(0, 24, (B,), generator=torch.Generator().manual_seed(5))
    y1h = du.to_one_hot(labels, 24)
    np.random.seed(8)
    spikes, targets = du.iq2spiketrain(x, y1h, out_w=R, out_h=R, max_duration=T)
    xin = torch.Tensor(spikes)
    out = {"x": pack_bits(spikes.reshape(T, B, -1)), "labels": labels.numpy()}
    pl = [[] for _ in net.dcll_slices]
    ol = []
    net.reset()
    for t in range(T):
        cur = xin[t]
        for i, s in enumerate(net.dcll_slices):
            o, p, pv, v = s.forward(cur, ignore_burnin=True)
            pl[i].append(npy(p))
            if s.dclllayer.output_layer:
                ol.append(npy(o))
            cur = o
    for i in range(len(net.dcll_slices)):
        out["p/%d" % i] = np.stack(pl[i])
        out["clout/%d" % i] = np.array(net.dcll_slices[i].clout)
        out.update(state_dict_np(net.dcll_slices[i].dclllayer, "sd_after_reset/%d/" % i))
    out["o_last"] = np.stack(ol)
    out["acc"] = np.array(net.accuracy(torch.Tensor(targets)))
    np.savez_compressed(os.path.join(OUT, "g9_restored_run.npz"), **out)
    return dict(B=B, R=R, T=T, netscale=0.25, burnin=2, np_seed_before_reset=7)


def main():
    lib, nets, du = import_reference()
    torch.set_num_threads(1)        # pin the oneDNN reduction schedule used for the fixtures
    if "--only-g9" in sys.argv:
        with open(os.path.join(OUT, "meta.json")) as f:
            meta = json.load(f)
        meta["g9"] = g9_checkpoint(lib, nets, du)
        with open(os.path.join(OUT, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, default=lambda o: list(o))
        return
    if "--only-carry" in sys.argv:
        g2_carry(lib, nets, du)
        return
    if "--only-g7b" in sys.argv:
        g7b_dense_sequence(lib)
        return
    if "--only-g6d" in sys.argv:
        g6d_dense_learning(lib)
        return
    if "--only-g1x" in sys.argv:
        g1x_layer_options(lib)
        return
    if "--only-variants" in sys.argv:
        with open(os.path.join(OUT, "meta.json")) as f:
            meta = json.load(f)
        meta["g2_r16_variants"] = g2_r16_variants(lib, nets, du)
        with open(os.path.join(OUT, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, default=lambda o: list(o))
        return
    if "--only-g6r" in sys.argv:
        g6r_train_variants(lib, nets, du)
        return
    if "--only-r128" in sys.argv:
        with open(os.path.join(OUT, "meta.json")) as f:
            meta = json.load(f)
        meta["g2_r128"] = g2_r128(lib, nets, du)
        with open(os.path.join(OUT, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, default=lambda o: list(o))
        return
    if "--only-refyaml-variants" in sys.argv:
        with open(os.path.join(OUT, "meta.json")) as f:
            meta = json.load(f)
        meta["g2_ref_yaml_variants"] = g2_ref_yaml_variants(lib, nets, du)
        with open(os.path.join(OUT, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, default=lambda o: list(o))
        return
    if "--only-refyaml" in sys.argv:
        with open(os.path.join(OUT, "meta.json")) as f:
            meta = json.load(f)
        meta["g2_ref_yaml"] = g2_ref_yaml(lib, nets, du)
        with open(os.path.join(OUT, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, default=lambda o: list(o))
        return
    if "--only-t1024" in sys.argv:
        with open(os.path.join(OUT, "meta.json")) as f:
            meta = json.load(f)
        meta["g2_t1024"] = g2_t1024(lib, nets, du)
        with open(os.path.join(OUT, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, default=lambda o: list(o))
        return
    if "--only-g6b" in sys.argv:
        with open(os.path.join(OUT, "meta.json")) as f:
            meta = json.load(f)
        meta["g6b"] = g6b_train_production(lib, nets, du)
        with open(os.path.join(OUT, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, default=lambda o: list(o))
        return
    if "--only-r32" in sys.argv:
        with open(os.path.join(OUT, "meta.json")) as f:
            meta = json.load(f)
        meta["g2_r32"] = g2_r32(lib, nets, du)
        with open(os.path.join(OUT, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, default=lambda o: list(o))
        return
    meta = {
        "torch": torch.__version__,
        "numpy": np.__version__,
        "torch_num_threads": torch.get_num_threads(),
        "torch_config": torch.__config__.show(),
        "g1": g1_layer_steps(lib),
        "g2": g2_rollouts(lib, nets, du),
        "g2_r32": g2_r32(lib, nets, du),
        "g2_mnist": g2_mnist(lib, nets, du),
        "g2_t1024": g2_t1024(lib, nets, du),
        "g2_ref_yaml": g2_ref_yaml(lib, nets, du),
        "g2_r128": g2_r128(lib, nets, du),
        "g2_r16_variants": g2_r16_variants(lib, nets, du),
        "g2_ref_yaml_variants": g2_ref_yaml_variants(lib, nets, du),
        "g5": {k: [{kk: (list(vv) if isinstance(vv, tuple) else vv) for kk, vv in d.items()} for d in v]
               for k, v in g5_specs(nets).items()},
    }
    g3_iq(du)
    g4_votes(lib)
    g6_train_step(lib, nets, du)
    g6r_train_variants(lib, nets, du)
    meta["g6b"] = g6b_train_production(lib, nets, du)
    g7_dense(lib)
    g7b_dense_sequence(lib)
    g6d_dense_learning(lib)
    g1x_layer_options(lib)
    g2_carry(lib, nets, du)
    g8_image(du)
    meta["g9"] = g9_checkpoint(lib, nets, du)
    with open(os.path.join(OUT, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, default=lambda o: list(o))
    for fn in sorted(os.listdir(OUT)):
        print("%-40s %8d" % (fn, os.path.getsize(os.path.join(OUT, fn))))


if __name__ == "__main__":
    main()
