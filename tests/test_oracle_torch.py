"""oracle/torch_ref.py (the torch-CPU port timed as cpu_baseline) must be BIT-identical to the imported
reference on the golden vectors (same torch build => same oneDNN kernels)."""
import numpy as np
import pytest
import torch

from oracle import torch_ref as R
from conftest import unpack_bits

torch.set_num_threads(1)


def t(a):
    return torch.from_numpy(np.asarray(a))


def sd_t(d):
    return {k: t(v) for k, v in d.items()}


G1_CASES = ["radio_l0", "radio_l1", "radio_l2_out", "radio_norp", "scalar_tau", "mnist_l0", "mnist_l2",
            "ref_tuple", "pool3"]


@pytest.mark.parametrize("case", G1_CASES)
def test_g1_layer_steps_bit_exact(golden, golden_meta, case):
    g = golden("g1_layer_steps.npz")
    m = golden_meta["g1"][case]
    sd = sd_t(g.sub("g1/%s/sd/" % case))
    layer = R.RefConvLayer(sd, m["pad"], m["pool"], m["wrp"], m["alpharp"], m["output_layer"])
    for step in range(3):
        x = t(g["g1/%s/x%d" % (case, step)])
        o, p, pv, v = layer.forward(x)
        for name, got in (("o", o), ("p", p), ("pv", pv), ("v", v)):
            exp = g["g1/%s/%s%d" % (case, name, step)]
            assert np.array_equal(got.numpy(), exp), (case, step, name, np.abs(got.numpy() - exp).max())
        names = ("eps0", "eps1", "arp")
        for i, st in enumerate(layer.state):
            assert np.array_equal(st.numpy(), g["g1/%s/out_%s%d" % (case, names[i], step)])


@pytest.mark.parametrize("name,R_,wrp", [("g2_radio_r8_t32_b3_traces.npz", 8, 1.0),
                                          ("g2_radio_r8_t24_b2_norp_traces.npz", 8, 0.0),
                                          ("g2_radio_r16_t128_b2.npz", 16, 1.0)])
def test_g2_rollout_bit_exact(golden, name, R_, wrp):
    g = golden(name)
    convs = [dict(padding=3, pooling=1)] * 3
    net = R.RefConvNetwork([sd_t(g.sub("sd/%d/" % i)) for i in range(3)], convs, wrp)
    cells = g["cells"]
    T, B = cells.shape
    for step in range(T):
        x = torch.zeros(B, 1, R_ * R_)
        x[torch.arange(B), 0, t(cells[step]).long()] = 1.0
        outs = net.test(x.reshape(B, 1, R_, R_))
        for i, (o, p, pv, v) in enumerate(outs):
            s = (v > 0).float().reshape(B, -1).numpy()
            assert np.array_equal(s, unpack_bits(g["spikes/%d" % i][step], s.shape[1])), (step, i)
            assert np.array_equal(p.numpy(), g["p/%d" % i][step]), (step, i)
        assert np.array_equal(outs[-1][0].numpy(), g["o_last"][step])
    for i in range(3):
        assert np.array_equal(np.array(net.clout[i]), g["clout/%d" % i])
        assert np.array_equal(net.votes()[i], g["vote/%d" % i])


def test_g2_mnist_plumbing_bit_exact(golden):
    """BASELINE config 1: mnist_conv.yaml, T=50, CPU only."""
    g = golden("g2_mnist_t50_b4.npz")
    convs = [dict(padding=2, pooling=2), dict(padding=2, pooling=1), dict(padding=2, pooling=2)]
    net = R.RefConvNetwork([sd_t(g.sub("sd/%d/" % i)) for i in range(3)], convs, 0.0)
    xs = unpack_bits(g["x"], 28 * 28)
    T, B = xs.shape[:2]
    for step in range(T):
        outs = net.test(t(xs[step]).reshape(B, 1, 28, 28))
        for i, (o, p, pv, v) in enumerate(outs):
            assert np.array_equal(p.numpy(), g["p/%d" % i][step])
        assert np.array_equal(outs[-1][0].numpy(), g["o_last"][step])
    for i in range(3):
        assert np.array_equal(np.array(net.clout[i]), g["clout/%d" % i])


@pytest.mark.parametrize("case,wrp", [("rrp", 1.0), ("plain", 0.0), ("plain_rtau", 0.0)])
def test_g7_dense_bit_exact(golden, case, wrp):
    g = golden("g7_dense.npz")
    layer = R.RefDenseLayer(sd_t(g.sub("g7/%s/sd/" % case)), wrp)
    for step in range(3):
        s, p, pv, v = layer.forward(t(g["g7/%s/x%d" % (case, step)]))
        for name, got in (("o", s), ("p", p), ("pv", pv), ("v", v)):
            assert np.array_equal(got.numpy(), g["g7/%s/%s%d" % (case, name, step)]), (case, step, name)


def test_g4_votes(golden):
    g = golden("g4_votes.npz")
    clout = list(g["clout"])
    assert np.array_equal(R.predictions_by_vote(clout), g["pred"])
    T, B = g["clout"].shape
    y = np.zeros((T, B, 5), np.float32)
    y[:, np.arange(B), g["labels"]] = 1
    assert R.accuracy_by_vote(clout, y) == float(g["acc"])


@pytest.mark.parametrize("case,cin,cout,wrp", [("rrp_512_128", 512, 128, 1.0), ("plain_600_160", 600, 160, 0.0)])
def test_g7b_dense_sequence_bit_exact(golden, case, cin, cout, wrp):
    g = golden("g7b_dense_sequence.npz")
    pre = "g7b/%s/" % case
    layer = R.RefDenseLayer(sd_t(g.sub(pre + "sd/")), wrp)
    torch.set_num_threads(1)
    for step in range(g[pre + "x"].shape[0]):
        x = t(np.unpackbits(g[pre + "x"][step], axis=-1, bitorder="little")[:, :cin].astype(np.float32))
        s, p, pv, v = layer.forward(x)
        assert np.array_equal(s.numpy(), np.unpackbits(g[pre + "s"][step], axis=-1, bitorder="little")[:, :cout])
        assert np.array_equal(p.numpy(), g[pre + "p"][step])
    for i, nm in enumerate(("eps0", "eps1", "arp")[:3 if wrp > 0 else 2]):
        assert np.array_equal(layer.state[i].numpy(), g[pre + "final_" + nm])


G6D = [("rrp", 1.0, False), ("plain", 0.0, False), ("plain_rtau", 0.0, False), ("reg", 1.0, 0.05)]


@pytest.mark.parametrize("name,wrp,reg", G6D)
def test_g6d_dense_learning_steps_bit_exact(golden, name, wrp, reg):
    """Fixture G6d (DCLLBase.train_dcll on a DenseDCLLlayer slice, generated by importing the reference): the oracle's
    dense step under torch autograd with the reference's loss / regulariser expressions (:694-703) and torch.optim.Adam
    reproduces every spike, readout, loss, gradient and the trained weights bit for bit (same torch build)."""
    g = golden("g6d_dense_learning.npz")
    pre = "g6d/%s/" % name
    sd = sd_t(g.sub(pre + "sd0/"))
    W = sd["i2h.weight"].clone().requires_grad_(True)
    b = sd["i2h.bias"].clone().requires_grad_(True)
    opt = torch.optim.Adam([W, b], lr=1e-5, betas=[0.0, .95], weight_decay=10.0)
    crit = torch.nn.SmoothL1Loss()
    cin, cout = W.shape[1], W.shape[0]
    x_all = unpack_bits(g[pre + "x"], cin)
    tgt = t(g[pre + "target"])
    T, B = x_all.shape[:2]
    st = [torch.zeros(B, cin), torch.zeros(B, cin)] + ([torch.zeros(B, cout)] if wrp > 0 else [])
    burnin = 4
    for step in range(T):
        s, pv, v, new = R.dense_lif_step(t(x_all[step]), W, b, sd["i2h.alpha"], sd["i2h.tau_m__dt"], sd["i2h.alphas"],
                                         sd["i2h.tau_s__dt"], st, .65, wrp)
        st = [q.detach() for q in new]
        p = torch.nn.functional.linear(pv, sd["i2o.weight"], sd["i2o.bias"])
        assert np.array_equal(s.numpy(), unpack_bits(g[pre + "s"][step], cout)), step
        assert np.array_equal(p.detach().numpy(), g[pre + "p"][step]), step
        if step + 1 >= burnin:
            opt.zero_grad()
            loss = crit(p, tgt)
            if reg:
                loss = loss + 20.0 * reg * torch.mean(torch.relu(v + 0.01)) + 0.1 * reg * torch.relu(0.1 - torch.mean(pv))
            loss.backward()
            if pre + "grad/%d/w" % step in g.keys():
                assert np.array_equal(W.grad.numpy(), g[pre + "grad/%d/w" % step])
                assert np.array_equal(b.grad.numpy(), g[pre + "grad/%d/b" % step])
            opt.step()
    assert np.array_equal(W.detach().numpy(), g[pre + "sd1/i2h.weight"])
    assert np.array_equal(b.detach().numpy(), g[pre + "sd1/i2h.bias"])
    for i, nm in enumerate(("eps0", "eps1", "arp")[:len(st)]):
        assert np.array_equal(st[i].numpy(), g[pre + "final_" + nm]), nm


@pytest.mark.parametrize("case", __import__("conftest").G1X_CASES)
def test_g1x_layer_options_torch_port_is_bit_identical(golden, case):
    """Fixture G1x (round 6; make_golden.py --only-g1x): layers built with stride / dilation / groups other than 1, bias=False,
    an activation other than nn.Sigmoid(), spiking=False (dcll/pytorch_libdcll.py:299-313, :75-104) — the torch restatement
    with the same options reproduces the reference's three steps bit for bit (output, pvoutput, pv, pvmem, neuron state)."""
    from conftest import g1x_cfg
    g = golden("g1x_layer_options.npz")
    c = g1x_cfg(g, case)
    pre = "g1x/%s/" % case
    sd = {k: torch.from_numpy(v) for k, v in g.sub(pre + "sd/").items()}
    if c.kind == 2:
        layer = R.RefDenseLayer(sd, c.wrp, .65, act=c.act_module(), spiking=bool(c.spiking))
    else:
        if c.kind == 1:         # (the bare i2h module: no readout tensors in its state dict)
            sd = dict({"i2h." + k: v for k, v in sd.items()}, **{"i2o.weight": torch.zeros(1, 1), "i2o.bias": torch.zeros(1)})
        layer = R.RefConvLayer(sd, (c.pad_h, c.pad_w), (c.pool_h, c.pool_w), c.wrp, .65, bool(c.output_layer), c.stride, c.dilation,
                               c.groups, c.act_module(), bool(c.spiking))
    for t in range(3):
        x = torch.from_numpy(g[pre + "x%d" % t])
        if c.kind == 1:
            if layer.state is None:
                layer.init_state(c.B, (c.H, c.W))
            o, pv, v, layer.state = R.conv_lif_step(x, layer.w, layer.b, layer.alpha, layer.tau_m, layer.alphas, layer.tau_s,
                                                    layer.state, .65, c.wrp, c.stride, layer.padding, c.dilation, c.groups,
                                                    layer.act, layer.spiking)
            got = {"o": o, "pv": pv, "v": v}
        else:
            o, p, pv, v = layer.forward(x)
            got = {"o": o, "p": p, "pv": pv, "v": v}
        for nm, val in got.items():
            assert np.array_equal(val.numpy(), g[pre + "%s%d" % (nm, t)]), (case, t, nm)
        for i, nm in enumerate(("eps0", "eps1", "arp")[:len(layer.state)]):
            assert np.array_equal(layer.state[i].numpy(), g[pre + "out_%s%d" % (nm, t)]), (case, t, nm)
