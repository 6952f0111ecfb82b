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
