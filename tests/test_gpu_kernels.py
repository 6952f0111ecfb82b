"""GPU parity tests proper: every HIP entry point of include/dcll_hip.h, called through the C ABI
(snn_modulation_classification_amd.ops -> ctypes -> libdcll_hip.so), against the pinned-order C oracle
(bit-exact for traces, v, spikes, arp) and against the golden vectors from the reference."""
import numpy as np
import pytest
import torch

from conftest import unpack_bits

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-4      # BASELINE.json north_star: class logits within 1e-4 fp32
PV_TOL = 2e-6         # sigmoid: v_exp_f32/v_rcp_f32 vs libm expf, a few ulp of values in (0,1)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def cu(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def bits_equal(a, b):
    return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


G1_CASES = ["radio_l0", "radio_l1", "radio_l2_out", "radio_norp", "scalar_tau", "mnist_l0", "mnist_l2",
            "ref_tuple", "pool3"]


@pytest.mark.parametrize("case", G1_CASES)
def test_step_vs_oracle_and_golden(golden, golden_meta, dev, case):
    """dcll_conv_lif_step free-running for 3 steps == C oracle bit for bit; == reference within the band."""
    from snn_modulation_classification_amd import ops
    from oracle import c_oracle as C
    g = golden("g1_layer_steps.npz")
    m = golden_meta["g1"][case]
    sd = g.sub("g1/%s/sd/" % case)
    orc = C.OracleConvLayer(sd, m["im"], m["pad"], m["pool"], m["wrp"], m["alpharp"], m["output_layer"])
    d = ops.make_conv_desc(m["cin"], m["cout"], m["im"], m["k"], m["pad"], m["pool"], 24, m["output_layer"],
                           sd["i2h.alpha"].size > 1, m["wrp"], m["alpharp"])
    ch, cw, ph, pw = ops.conv_out_shape(d)
    assert (ch, cw, ph, pw) == (orc.ch, orc.cw, orc.ph, orc.pw)
    B = m["B"]
    t = {k: cu(v, dev) for k, v in sd.items()}
    eps0 = torch.zeros((B, m["cin"]) + tuple(m["im"]), device=dev)
    eps1 = torch.zeros_like(eps0)
    arp = torch.zeros((B, m["cout"], ch, cw), device=dev)
    for step in range(3):
        x = g["g1/%s/x%d" % (case, step)]
        s, p, o, pv, v = ops.conv_lif_step(d, cu(x, dev), t["i2h.weight"], t["i2h.bias"], t["i2h.alpha"],
                                           t["i2h.tau_m__dt"], t["i2h.alphas"], t["i2h.tau_s__dt"], eps0, eps1, arp,
                                           t["i2o.weight"], t["i2o.bias"], t.get("output_.weight"),
                                           t.get("output_.bias"))
        oo, op, opv, ov, os_ = orc.forward(x)
        assert bits_equal(eps0.cpu().numpy(), orc.state[0])
        assert bits_equal(eps1.cpu().numpy(), orc.state[1])
        assert bits_equal(v.cpu().numpy(), ov), np.abs(v.cpu().numpy() - ov).max()
        assert np.array_equal(s.cpu().numpy(), os_)
        if m["wrp"] > 0:
            assert bits_equal(arp.cpu().numpy(), orc.state[2])
        np.testing.assert_allclose(pv.cpu().numpy(), opv, atol=PV_TOL, rtol=0)
        np.testing.assert_allclose(p.cpu().numpy(), op, atol=LOGIT_TOL, rtol=0)
        if m["output_layer"]:
            np.testing.assert_allclose(o.cpu().numpy(), oo, atol=LOGIT_TOL, rtol=0)
        # against the reference itself (the oracle showed zero flips on these fixtures)
        e = lambda n: g["g1/%s/%s%d" % (case, n, step)]
        assert bits_equal(eps1.cpu().numpy(), e("out_eps1"))
        assert np.array_equal(v.cpu().numpy() > 0, e("v") > 0)
        np.testing.assert_allclose(p.cpu().numpy(), e("p"), atol=LOGIT_TOL, rtol=0)


def _rand_layer(rng, cin, cout, k=7, gain=1.0):
    n = cin * k * k
    stdv = 1.0 / np.sqrt(n) / 250
    W = rng.uniform(-stdv * 1e-2, stdv * 1e-2, size=(cout, cin, k, k)).astype(np.float32) * gain
    b = rng.uniform(-stdv, stdv, size=(cout,)).astype(np.float32)
    taum = rng.uniform(5, 35, size=cin) * 1e-3
    taus = rng.uniform(5, 10, size=cin) * 1e-3
    alpha = (1 - 1e-3 / taum).astype(np.float32)
    alphas = (1 - 1e-3 / taus).astype(np.float32)
    tau_m = (np.float32(1) / (np.float32(1) - alpha)).astype(np.float32)
    tau_s = (np.float32(1) / (np.float32(1) - alphas)).astype(np.float32)
    return W, b, alpha, tau_m, alphas, tau_s


def _sd_from(W, b, alpha, tau_m, alphas, tau_s, hw, target=24, rng=None):
    bc = lambda a: np.ascontiguousarray(np.broadcast_to(a[:, None, None], (a.shape[0],) + hw)).astype(np.float32)
    K = W.shape[0] * hw[0] * hw[1]
    return {"i2h.weight": W, "i2h.bias": b, "i2h.alpha": bc(alpha), "i2h.tau_m__dt": bc(tau_m),
            "i2h.alphas": bc(alphas), "i2h.tau_s__dt": bc(tau_s),
            "i2o.weight": rng.uniform(-.0055, .0055, size=(target, K)).astype(np.float32),
            "i2o.bias": rng.uniform(-.0055, .0055, size=(target,)).astype(np.float32)}


@pytest.mark.parametrize("wrp,T,B,zero_state", [(1.0, 24, 3, True), (0.0, 9, 2, True), (1.0, 10, 5, False),
                                                 (0.0, 26, 2, False), (1.0, 31, 2, False), (1.0, 5, 3, False),
                                                 (0.0, 7, 2, True), (1.0, 1, 4, False)])
def test_sequence_c32_vs_oracle(dev, wrp, T, B, zero_state):
    """k_lif_seq_c32 (T < 8) and k_lif_seq_c32d (T >= 8; two tiles per wave and stage) — MFMA systolic chain, state
    on chip — == C oracle stepping, bit for bit, incl. final state; both variants (refractory or not), zero and
    non-zero initial state, T not a multiple of anything convenient."""
    from snn_modulation_classification_amd import ops
    from oracle import c_oracle as C
    rng = np.random.RandomState(11)
    W, b, alpha, tau_m, alphas, tau_s = _rand_layer(rng, 32, 32, gain=3.0)
    sd = _sd_from(W, b, alpha, tau_m, alphas, tau_s, (16, 16), rng=rng)
    orc = C.OracleConvLayer(sd, (16, 16), 3, 1, wrp)
    orc.init_state(B)
    if not zero_state:
        orc.state[0][...] = rng.uniform(0, 5, size=orc.state[0].shape)
        orc.state[1][...] = rng.uniform(0, 50, size=orc.state[1].shape)
        orc.state[2][...] = -rng.uniform(0, 2, size=orc.state[2].shape)
    eps0, eps1, arp = [cu(s.copy(), dev) for s in orc.state]
    x = (rng.uniform(size=(T, B, 32, 256)) < 0.08).astype(np.float32)
    x[0] = (rng.uniform(size=(B, 32, 256)) < 0.5)          # first-step burst
    d = ops.make_conv_desc(32, 32, (16, 16), 7, 3, 1, 24, False, True, wrp)
    spk_in = ops.pack_spikes(cu(x, dev))
    assert spk_in.shape == (T, B, 32, 8)
    tau4 = cu(np.stack([alpha, tau_m, alphas, tau_s]), dev)
    spk, pv, v = ops.conv_lif_sequence(d, spk_in, cu(W, dev), cu(b, dev), tau4, eps0, eps1, arp, T, B, want_v=True)
    torch.cuda.synchronize()
    spk_d = ops.unpack_spikes(spk).cpu().numpy().reshape(T, B, 32, 16, 16)
    v, pv = v.cpu().numpy(), pv.cpu().numpy()
    for t in range(T):
        oo, op, opv, ov, os_ = orc.forward(x[t].reshape(B, 32, 16, 16))
        assert bits_equal(v[t], ov), (t, np.abs(v[t] - ov).max())
        assert np.array_equal(spk_d[t], os_), t
        np.testing.assert_allclose(pv[t], opv, atol=PV_TOL, rtol=0)
    assert bits_equal(eps0.cpu().numpy(), orc.state[0])
    assert bits_equal(eps1.cpu().numpy(), orc.state[1])
    if wrp > 0:
        assert bits_equal(arp.cpu().numpy(), orc.state[2])
    assert (0.01 if T > 1 else 0.001) < (spk_d[1:] if T > 1 else spk_d).mean() < 0.9, "degenerate test: spikes all equal"


@pytest.mark.parametrize("hw,T", [((16, 16), 12), ((16, 16), 3), ((32, 32), 6)])
@pytest.mark.parametrize("wrp", [1.0, 0.0])
def test_sequence_output_variants_agree(dev, hw, T, wrp):
    """Every compiled output variant of the 32 -> 32 sequence kernels (pv and / or v written or not, spikes written or
    not: the OUT template parameter of k_lif_seq_c32 / c32d / c32t) leaves the same state and, where produced, the same
    spikes, pv and v."""
    from snn_modulation_classification_amd import ops
    rng = np.random.RandomState(23)
    H, Wd = hw
    B = 2
    W, b, alpha, tau_m, alphas, tau_s = _rand_layer(rng, 32, 32, gain=3.0)
    x = (rng.uniform(size=(T, B, 32, H * Wd)) < 0.1).astype(np.float32)
    d = ops.make_conv_desc(32, 32, hw, 7, 3, 1, 24, False, wrp > 0, wrp)
    spk_in = ops.pack_spikes(cu(x, dev))
    tau4 = cu(np.stack([alpha, tau_m, alphas, tau_s]), dev)
    ref = None
    for want_pv in (True, False):
        for want_v in (True, False):
            for want_spikes in (True, False):
                st = [torch.zeros((B, 32, H, Wd), device=dev) for _ in range(3)]
                spk, pv, v = ops.conv_lif_sequence(d, spk_in, cu(W, dev), cu(b, dev), tau4, st[0], st[1],
                                                   st[2] if wrp > 0 else None, T, B, want_spikes=want_spikes,
                                                   want_pv=want_pv, want_v=want_v)
                got = dict(spk=spk, pv=pv, v=v, eps0=st[0], eps1=st[1], arp=st[2])
                if ref is None:
                    ref = got
                    assert float(ref["spk"].ne(0).float().mean()) > 0
                for k, val in got.items():
                    if val is not None:
                        assert torch.equal(val, ref[k]), (k, want_pv, want_v, want_spikes)


@pytest.mark.parametrize("hw,wrp,T,B,zero_state", [((32, 32), 1.0, 11, 2, True), ((16, 64), 0.0, 10, 3, False),
                                                    ((24, 96), 1.0, 9, 2, False), ((128, 128), 1.0, 3, 1, True),
                                                    ((8, 64), 1.0, 9, 2, False)])
def test_sequence_c32_tiled_vs_oracle(dev, hw, wrp, T, B, zero_state):
    """k_lif_seq_c32t (large planes: one workgroup per 8 x 32 tile with a recomputed 3-pixel trace halo) == C oracle
    stepping, bit for bit, incl. the final state; planes with 1, 2, 3 and 4 tiles per row / column exercise every
    border combination (tile touching both, one or no plane edge)."""
    from snn_modulation_classification_amd import ops
    from oracle import c_oracle as C
    rng = np.random.RandomState(13)
    H, Wd = hw
    W, b, alpha, tau_m, alphas, tau_s = _rand_layer(rng, 32, 32, gain=3.0)
    sd = _sd_from(W, b, alpha, tau_m, alphas, tau_s, hw, rng=rng)
    orc = C.OracleConvLayer(sd, hw, 3, 1, wrp)
    orc.init_state(B)
    if not zero_state:
        orc.state[0][...] = rng.uniform(0, 5, size=orc.state[0].shape)
        orc.state[1][...] = rng.uniform(0, 50, size=orc.state[1].shape)
        if wrp > 0:
            orc.state[2][...] = -rng.uniform(0, 2, size=orc.state[2].shape)
    state = [cu(s.copy(), dev) for s in orc.state]
    eps0, eps1 = state[0], state[1]
    arp = state[2] if wrp > 0 else None
    x = (rng.uniform(size=(T, B, 32, H * Wd)) < 0.08).astype(np.float32)
    x[0] = (rng.uniform(size=(B, 32, H * Wd)) < 0.5)
    d = ops.make_conv_desc(32, 32, hw, 7, 3, 1, 24, False, wrp > 0, wrp)
    spk_in = ops.pack_spikes(cu(x, dev))
    assert spk_in.shape == (T, B, 32, H * Wd // 32)
    tau4 = cu(np.stack([alpha, tau_m, alphas, tau_s]), dev)
    spk, pv, v = ops.conv_lif_sequence(d, spk_in, cu(W, dev), cu(b, dev), tau4, eps0, eps1, arp, T, B, want_v=True)
    torch.cuda.synchronize()
    spk_d = ops.unpack_spikes(spk).cpu().numpy().reshape(T, B, 32, H, Wd)
    v, pv = v.cpu().numpy(), pv.cpu().numpy()
    for t in range(T):
        oo, op, opv, ov, os_ = orc.forward(x[t].reshape(B, 32, H, Wd))
        assert bits_equal(v[t], ov), (t, np.abs(v[t] - ov).max(), np.argwhere(v[t] != ov)[:5])
        assert np.array_equal(spk_d[t], os_), t
        np.testing.assert_allclose(pv[t], opv, atol=PV_TOL, rtol=0)
    assert bits_equal(eps0.cpu().numpy(), orc.state[0])
    assert bits_equal(eps1.cpu().numpy(), orc.state[1])
    if wrp > 0:
        assert bits_equal(arp.cpu().numpy(), orc.state[2])
    assert 0.01 < spk_d[1:].mean() < 0.9, "degenerate test: spikes all equal"


@pytest.mark.parametrize("wrp,hw,scalar_tau,B", [(1.0, (32, 48), False, 3), (0.0, (16, 32), True, 2),
                                                 (1.0, (128, 128), False, 2), (0.0, (48, 16), False, 5),
                                                 (1.0, (16, 16), False, 7), (0.0, (16, 16), True, 128)])
def test_tiled_step_mfma_vs_oracle(dev, wrp, hw, scalar_tau, B):
    """k_trace4 + k_lif_step_c32t (per-step forward of a 32 -> 32 layer on planes of several 16x16 tiles, incl. the argparse
    default 128x128; behind dcll_conv_lif_step) == C oracle stepping bit for bit over several steps from a non-zero state: every
    tile reads the new eps1 of its halo, advanced by the elementwise pass in front (k_trace4) — state, v and spikes must
    agree everywhere, in particular along the tile borders; arbitrary fp32 input.  The 16x16 cases: batches <= 256 run two
    workgroups per sample on 8-row tiles."""
    from snn_modulation_classification_amd import ops
    from oracle import c_oracle as C
    rng = np.random.RandomState(29)
    W, b, alpha, tau_m, alphas, tau_s = _rand_layer(rng, 32, 32, gain=3.0)
    sd = _sd_from(W, b, alpha, tau_m, alphas, tau_s, hw, rng=rng)
    if scalar_tau:
        for k, a in (("i2h.alpha", alpha), ("i2h.tau_m__dt", tau_m), ("i2h.alphas", alphas), ("i2h.tau_s__dt", tau_s)):
            sd[k] = a[:1].copy()
    orc = C.OracleConvLayer(sd, hw, 3, 1, wrp)
    orc.init_state(B)
    orc.state[0][...] = rng.uniform(0, 5, size=orc.state[0].shape) * (rng.uniform(size=orc.state[0].shape) < 0.3)
    orc.state[1][...] = rng.uniform(0, 50, size=orc.state[1].shape) * (rng.uniform(size=orc.state[1].shape) < 0.3)
    if wrp > 0:
        orc.state[2][...] = -rng.uniform(0, 2, size=orc.state[2].shape)
    d = ops.make_conv_desc(32, 32, hw, 7, 3, 1, 24, False, not scalar_tau, wrp)
    t = {k: cu(v, dev) for k, v in sd.items()}
    eps0, eps1 = cu(orc.state[0].copy(), dev), cu(orc.state[1].copy(), dev)
    arp = cu(orc.state[2].copy(), dev) if wrp > 0 else torch.zeros((B, 32) + hw, device=dev)
    n_spk = 0
    for step in range(3):
        x = ((rng.uniform(size=(B, 32) + hw) < 0.15) * rng.choice([1.0, 1.0, 0.5], size=(B, 32) + hw)).astype(np.float32)
        s, p, o, pv, v = ops.conv_lif_step(d, cu(x, dev), t["i2h.weight"], t["i2h.bias"], t["i2h.alpha"],
                                           t["i2h.tau_m__dt"], t["i2h.alphas"], t["i2h.tau_s__dt"], eps0, eps1, arp,
                                           t["i2o.weight"], t["i2o.bias"])
        oo, op, opv, ov, os_ = orc.forward(x)
        assert bits_equal(eps0.cpu().numpy(), orc.state[0]) and bits_equal(eps1.cpu().numpy(), orc.state[1])
        assert bits_equal(v.cpu().numpy(), ov), (step, np.abs(v.cpu().numpy() - ov).max())
        assert np.array_equal(s.cpu().numpy(), os_)
        if wrp > 0:
            assert bits_equal(arp.cpu().numpy(), orc.state[2])
        np.testing.assert_allclose(pv.cpu().numpy(), opv, atol=PV_TOL, rtol=0)
        np.testing.assert_allclose(p.cpu().numpy(), op, atol=LOGIT_TOL, rtol=0)
        n_spk += int(os_.sum())
    assert n_spk > 0, "degenerate test: no spikes"


@pytest.mark.parametrize("wrp,cout,scalar_tau,B,hw", [(1.0, 32, False, 5, (16, 16)), (0.0, 32, True, 3, (16, 16)),
                                                      (1.0, 8, True, 4, (16, 16)), (0.0, 20, False, 2, (16, 16)),
                                                      (1.0, 32, False, 300, (16, 16)), (1.0, 32, False, 3, (32, 48)),
                                                      (0.0, 8, True, 2, (128, 128)), (1.0, 20, False, 2, (16, 64))])
def test_first_layer_step_mfma_vs_oracle(dev, wrp, cout, scalar_tau, B, hw):
    """k_lif_step_c1 (the per-step forward of a 1 -> c_out <= 32 layer on the 16x16 plane, behind dcll_conv_lif_step) ==
    C oracle stepping bit for bit over several steps from a non-zero state — arbitrary fp32 input maps (not only one-hot
    planes), (1,H,W) tensor or scalar time constants, fewer than 32 channels, both variants, logits through the readout;
    larger planes: the tiled form (k_trace4 + one workgroup per 16x16 tile)."""
    from snn_modulation_classification_amd import ops
    from oracle import c_oracle as C
    rng = np.random.RandomState(23)
    W, b, alpha, tau_m, alphas, tau_s = _rand_layer(rng, 1, cout, gain=3.0)
    sd = _sd_from(W, b, alpha, tau_m, alphas, tau_s, hw, rng=rng)
    if scalar_tau:
        for k, a in (("i2h.alpha", alpha), ("i2h.tau_m__dt", tau_m), ("i2h.alphas", alphas), ("i2h.tau_s__dt", tau_s)):
            sd[k] = a[:1].copy()
    orc = C.OracleConvLayer(sd, hw, 3, 1, wrp)
    orc.init_state(B)
    orc.state[0][...] = rng.uniform(0, 5, size=orc.state[0].shape) * (rng.uniform(size=orc.state[0].shape) < 0.3)
    orc.state[1][...] = rng.uniform(0, 50, size=orc.state[1].shape) * (rng.uniform(size=orc.state[1].shape) < 0.3)
    if wrp > 0:
        orc.state[2][...] = -rng.uniform(0, 2, size=orc.state[2].shape)
    d = ops.make_conv_desc(1, cout, hw, 7, 3, 1, 24, False, not scalar_tau, wrp)
    t = {k: cu(v, dev) for k, v in sd.items()}
    eps0, eps1 = cu(orc.state[0].copy(), dev), cu(orc.state[1].copy(), dev)
    arp = cu(orc.state[2].copy(), dev) if wrp > 0 else torch.zeros((B, cout) + hw, device=dev)
    n_spk = 0
    for step in range(4):
        x = (rng.uniform(0, 2, size=(B, 1) + hw) * (rng.uniform(size=(B, 1) + hw) < 0.2)).astype(np.float32)
        if step == 0:
            x[0, 0, 0, 0], x[0, 0, hw[0] - 1, hw[1] - 1] = 1.0, 1.0
        s, p, o, pv, v = ops.conv_lif_step(d, cu(x, dev), t["i2h.weight"], t["i2h.bias"], t["i2h.alpha"],
                                           t["i2h.tau_m__dt"], t["i2h.alphas"], t["i2h.tau_s__dt"], eps0, eps1, arp,
                                           t["i2o.weight"], t["i2o.bias"])
        oo, op, opv, ov, os_ = orc.forward(x)
        assert bits_equal(eps0.cpu().numpy(), orc.state[0]) and bits_equal(eps1.cpu().numpy(), orc.state[1])
        assert bits_equal(v.cpu().numpy(), ov), (step, np.abs(v.cpu().numpy() - ov).max())
        assert np.array_equal(s.cpu().numpy(), os_)
        if wrp > 0:
            assert bits_equal(arp.cpu().numpy(), orc.state[2])
        np.testing.assert_allclose(pv.cpu().numpy(), opv, atol=PV_TOL, rtol=0)
        np.testing.assert_allclose(p.cpu().numpy(), op, atol=LOGIT_TOL, rtol=0)
        n_spk += int(os_.sum())
    assert n_spk > 0, "degenerate test: no spikes"


@pytest.mark.parametrize("wrp,cout,hw", [(1.0, 32, (16, 16)), (0.0, 32, (16, 16)), (1.0, 8, (16, 16)),
                                         (1.0, 32, (32, 32)), (0.0, 8, (24, 64)), (1.0, 32, (128, 128))])
def test_sequence_c1_vs_oracle(dev, wrp, cout, hw):
    """First-layer sequence kernels (k_lif_seq_c1 on 16x16, the tiled k_lif_seq_c1t on larger planes) == C oracle
    stepping, bit for bit, from a non-zero initial state, incl. the final state."""
    from snn_modulation_classification_amd import ops
    from oracle import c_oracle as C
    rng = np.random.RandomState(5)
    H, Wd = hw
    T, B = (17, 4) if H * Wd <= 4096 else (5, 2)
    W, b, alpha, tau_m, alphas, tau_s = _rand_layer(rng, 1, cout, gain=3.0)
    sd = _sd_from(W, b, alpha, tau_m, alphas, tau_s, hw, rng=rng)
    orc = C.OracleConvLayer(sd, hw, 3, 1, wrp)
    orc.init_state(B)
    orc.state[0][...] = rng.uniform(0, 5, size=orc.state[0].shape) * (rng.uniform(size=orc.state[0].shape) < 0.3)
    orc.state[1][...] = rng.uniform(0, 50, size=orc.state[1].shape) * (rng.uniform(size=orc.state[1].shape) < 0.3)
    if wrp > 0:
        orc.state[2][...] = -rng.uniform(0, 2, size=orc.state[2].shape)
    cells = rng.randint(0, H * Wd, size=(T, B)).astype(np.int32)
    cells[0, 0], cells[1, 0] = 0, H * Wd - 1                    # plane corners
    d = ops.make_conv_desc(1, cout, hw, 7, 3, 1, 24, False, True, wrp)
    eps0, eps1 = cu(orc.state[0].copy(), dev), cu(orc.state[1].copy(), dev)
    arp = cu(orc.state[2].copy(), dev) if wrp > 0 else torch.zeros((B, cout, H, Wd), device=dev)
    tau4 = cu(np.stack([alpha, tau_m, alphas, tau_s]), dev)
    spk, pv, v = ops.conv_lif_sequence_cells(d, cu(cells, dev), cu(W, dev), cu(b, dev), tau4, eps0, eps1, arp, T, B,
                                             want_v=True)
    spk_d = ops.unpack_spikes(spk).cpu().numpy().reshape(T, B, cout, H, Wd)
    v, pv = v.cpu().numpy(), pv.cpu().numpy()
    for t in range(T):
        x = np.zeros((B, 1, H * Wd), np.float32)
        x[np.arange(B), 0, cells[t]] = 1
        oo, op, opv, ov, os_ = orc.forward(x.reshape(B, 1, H, Wd))
        assert bits_equal(v[t], ov), (t, np.abs(v[t] - ov).max())
        assert np.array_equal(spk_d[t], os_)
        np.testing.assert_allclose(pv[t], opv, atol=PV_TOL, rtol=0)
    assert bits_equal(eps0.cpu().numpy(), orc.state[0])
    assert bits_equal(eps1.cpu().numpy(), orc.state[1])
    if wrp > 0:
        assert bits_equal(arp.cpu().numpy(), orc.state[2])


def test_full_rollout_vs_reference_golden(golden, dev):
    """radio_ml_conv.yaml, R=16, T=128, B=2: cells -> k_lif_seq_c1 -> k_lif_seq_c32 x2 -> readout -> argmax/vote,
    free-running, against the REFERENCE's own spikes (bit-exact), logits (1e-4), per-step argmax and votes."""
    from snn_modulation_classification_amd import ops
    g = golden("g2_radio_r16_t128_b2.npz")
    cells = g["cells"]
    T, B = cells.shape
    cur = None
    for i in range(3):
        sd = {k: cu(v, dev) for k, v in g.sub("sd/%d/" % i).items()}
        cin = sd["i2h.weight"].shape[1]
        d = ops.make_conv_desc(cin, 32, (16, 16), 7, 3, 1, 24, i == 2, True, 1.0)
        tau4 = torch.stack([sd[k][:, 0, 0] for k in ("i2h.alpha", "i2h.tau_m__dt", "i2h.alphas", "i2h.tau_s__dt")])
        tau4 = tau4.contiguous()
        eps0 = torch.zeros((B, cin, 16, 16), device=dev)
        eps1 = torch.zeros_like(eps0)
        arp = torch.zeros((B, 32, 16, 16), device=dev)
        if i == 0:
            spk, pv, _ = ops.conv_lif_sequence_cells(d, cu(cells, dev), sd["i2h.weight"], sd["i2h.bias"], tau4, eps0,
                                                     eps1, arp, T, B)
        else:
            spk, pv, _ = ops.conv_lif_sequence(d, cur, sd["i2h.weight"], sd["i2h.bias"], tau4, eps0, eps1, arp, T, B)
        ref = g["spikes/%d" % i].view(np.int32).reshape(T, B, 32, 8)
        assert np.array_equal(spk.cpu().numpy(), ref), "layer %d spike trains differ from the reference" % i
        p = ops.readout(pv.reshape(T * B, -1), sd["i2o.weight"], sd["i2o.bias"]).reshape(T, B, 24)
        np.testing.assert_allclose(p.cpu().numpy(), g["p/%d" % i], atol=LOGIT_TOL, rtol=0)
        logits = p
        if i == 2:
            o = ops.readout(pv.reshape(T * B, -1), sd["output_.weight"], sd["output_.bias"]).reshape(T, B, 24)
            np.testing.assert_allclose(o.cpu().numpy(), g["o_last"], atol=LOGIT_TOL, rtol=0)
            logits = o
        clout, vote = ops.argmax_vote(logits)
        # argmax may differ only where the reference's top two logits are closer than the tolerance
        ref_logits = g["o_last"] if i == 2 else g["p/%d" % i]
        top2 = np.sort(ref_logits, axis=-1)[..., -2:]
        close = (top2[..., 1] - top2[..., 0]) < 2 * LOGIT_TOL
        mism = clout.cpu().numpy() != g["clout/%d" % i]
        assert not (mism & ~close).any()
        if not mism.any():
            assert np.array_equal(vote.cpu().numpy(), g["vote/%d" % i])
        cur = spk
        for nm, st in (("eps0", eps0), ("eps1", eps1), ("arp", arp)):
            exp = g["final/%d/%s" % (i, nm)]
            if nm == "arp":
                np.testing.assert_allclose(st.cpu().numpy(), exp, atol=1e-6, rtol=0)
            else:
                assert bits_equal(st.cpu().numpy(), exp), (i, nm)


@pytest.mark.parametrize("rows,K,N", [(1, 8192, 24), (300, 8192, 24), (37, 2704, 10), (129, 100, 32), (5, 24, 10)])
def test_readout_gemm(dev, rows, K, N):
    from snn_modulation_classification_amd import ops
    rng = np.random.RandomState(0)
    pv = rng.uniform(0, 1, size=(rows, K)).astype(np.float32)
    Wt = rng.uniform(-.01, .01, size=(N, K)).astype(np.float32)
    b = rng.uniform(-.01, .01, size=(N,)).astype(np.float32)
    out = ops.readout(cu(pv, dev), cu(Wt, dev), cu(b, dev)).cpu().numpy()
    ref = (pv.astype(np.float64) @ Wt.astype(np.float64).T + b).astype(np.float32)
    np.testing.assert_allclose(out, ref, atol=2e-5, rtol=0)


def test_argmax_vote_vs_oracle(dev):
    from snn_modulation_classification_amd import ops
    from oracle import c_oracle as C
    rng = np.random.RandomState(3)
    T, B, N = 40, 200, 24
    logits = rng.randn(T, B, N).astype(np.float32)
    logits[:, :50] = np.round(logits[:, :50])          # many exact ties -> first maximum must win
    logits[:, 50:60, :] = 0.0                            # all equal -> class 0
    for tb in (0, 7):
        c_ref, v_ref = C.argmax_vote(logits, t_begin=tb)
        c, v = ops.argmax_vote(cu(logits, dev), t_begin=tb)
        assert np.array_equal(c.cpu().numpy(), c_ref)
        assert np.array_equal(v.cpu().numpy(), v_ref)


def test_pack_unpack_roundtrip(dev):
    from snn_modulation_classification_amd import ops
    rng = np.random.RandomState(1)
    x = (rng.uniform(size=(7, 3, 32, 256)) < 0.3).astype(np.float32)
    p = ops.pack_spikes(cu(x, dev))
    assert np.array_equal(p.cpu().numpy().view(np.uint8).reshape(7, 3, -1),
                          np.packbits(x.reshape(7, 3, -1).astype(np.uint8), axis=-1, bitorder="little"))
    assert np.array_equal(ops.unpack_spikes(p).cpu().numpy().reshape(x.shape), x)


def test_dense_step_vs_oracle(golden, dev):
    from snn_modulation_classification_amd import ops
    from snn_modulation_classification_amd._lib import DenseDesc
    from oracle import c_oracle as C
    g = golden("g7_dense.npz")
    for case, wrp in (("rrp", 1.0), ("plain", 0.0), ("plain_rtau", 0.0)):
        sd = g.sub("g7/%s/sd/" % case)
        orc = C.OracleDenseLayer(sd, wrp)
        t = {k: cu(v, dev) for k, v in sd.items()}
        d = DenseDesc(40, 24, 10, int(sd["i2h.alpha"].size > 1), int(wrp > 0), .65, wrp)
        eps0 = torch.zeros((5, 40), device=dev)
        eps1 = torch.zeros_like(eps0)
        arp = torch.zeros((5, 24), device=dev)
        for step in range(3):
            x = g["g7/%s/x%d" % (case, step)]
            s, p, pv, v = ops.dense_lif_step(d, cu(x, dev), t["i2h.weight"], t["i2h.bias"], t["i2h.alpha"],
                                             t["i2h.tau_m__dt"], t["i2h.alphas"], t["i2h.tau_s__dt"], eps0, eps1,
                                             arp, t["i2o.weight"], t["i2o.bias"])
            os_, op, opv, ov = orc.forward(x)
            assert bits_equal(v.cpu().numpy(), ov)
            assert np.array_equal(s.cpu().numpy(), os_)
            assert bits_equal(eps1.cpu().numpy(), orc.state[1])
            np.testing.assert_allclose(p.cpu().numpy(), op, atol=LOGIT_TOL, rtol=0)
            np.testing.assert_allclose(p.cpu().numpy(), g["g7/%s/p%d" % (case, step)], atol=LOGIT_TOL, rtol=0)


def test_error_conventions(dev):
    """Bad geometry -> DCLLUnsupported (NotImplementedError); nulls -> ValueError; never a crash."""
    from snn_modulation_classification_amd import ops, _lib
    d = ops.make_conv_desc(32, 32, (20, 20), 7, 3, 1, 24, False, True, 1.0)
    st = lambda: torch.zeros((1, 32, 20, 20), device=dev)
    with pytest.raises(NotImplementedError):      # well-formed operands, but a plane the fused kernel does not cover
        ops.conv_lif_sequence(d, torch.zeros((1, 1, 32, 400 // 32), device=dev, dtype=torch.int32),
                              torch.zeros((32, 32, 7, 7), device=dev), torch.zeros(32, device=dev),
                              torch.ones((4, 32), device=dev), st(), st(), st(), 1, 1)
    d2 = ops.make_conv_desc(1, 4, (8, 8), 3, 1, 1, 10, False, False, 0.0, stride=2)
    assert ops.conv_out_shape(d2) == (4, 4, 4, 4)               # (round 6: stride / dilation / groups run on the generic kernels)
    with pytest.raises(ValueError):                             # groups must divide both channel counts
        ops.conv_out_shape(ops.make_conv_desc(3, 4, (8, 8), 3, 1, 1, 10, False, False, 0.0, groups=2))
    with pytest.raises(NotImplementedError):                    # ... but no fused sequence kernel takes them
        d3 = ops.make_conv_desc(32, 32, (16, 16), 7, 3, 1, 24, False, True, 1.0, stride=2)
        ops.conv_lif_sequence(d3, torch.zeros((1, 1, 32, 8), device=dev, dtype=torch.int32),
                              torch.zeros((32, 32, 7, 7), device=dev), torch.zeros(32, device=dev),
                              torch.ones((4, 32), device=dev), torch.zeros((1, 32, 16, 16), device=dev),
                              torch.zeros((1, 32, 16, 16), device=dev), torch.zeros((1, 32, 8, 8), device=dev), 1, 1)
    with pytest.raises(_lib.DCLLHipError):
        ops.readout(torch.zeros(2, 4), torch.zeros(3, 4), None)      # CPU tensors: no CPU fallback


@pytest.mark.parametrize("n_ro,wrp", [(24, 1.0), (48, 1.0), (24, 0.0)])
def test_sequence_c32_fused_readout(dev, n_ro, wrp):
    """Readout fused into the epilogue (pv never leaves the chip) == readout GEMM over the materialised pv, and the
    spikes / state are unchanged by the fusion."""
    from snn_modulation_classification_amd import ops
    rng = np.random.RandomState(21)
    T, B = 19, 3
    W, b, alpha, tau_m, alphas, tau_s = _rand_layer(rng, 32, 32, gain=3.0)
    Wro = rng.uniform(-.0055, .0055, size=(n_ro, 8192)).astype(np.float32)
    bro = rng.uniform(-.0055, .0055, size=(n_ro,)).astype(np.float32)
    x = (rng.uniform(size=(T, B, 32, 256)) < 0.1).astype(np.float32)
    d = ops.make_conv_desc(32, 32, (16, 16), 7, 3, 1, 24, n_ro == 48, True, wrp)
    spk_in = ops.pack_spikes(cu(x, dev))
    tau4 = cu(np.stack([alpha, tau_m, alphas, tau_s]), dev)
    st = lambda: [torch.zeros((B, 32, 16, 16), device=dev) for _ in range(3)]
    sa, sb = st(), st()
    spk_a, pv_a, _ = ops.conv_lif_sequence(d, spk_in, cu(W, dev), cu(b, dev), tau4, *sa, T, B)
    ref = ops.readout(pv_a.reshape(T * B, -1), cu(Wro, dev), cu(bro, dev)).reshape(T, B, n_ro)
    Wp = ops.permute_readout(cu(Wro, dev))
    spk_b, pv_b, _, logits = ops.conv_lif_sequence(d, spk_in, cu(W, dev), cu(b, dev), tau4, *sb, T, B,
                                                   want_pv=False, ro_Wp=Wp, ro_b=cu(bro, dev))
    assert pv_b is None and logits.shape == (T, B, n_ro)
    assert torch.equal(spk_a, spk_b)
    for u, v in zip(sa, sb):
        assert torch.equal(u, v)
    np.testing.assert_allclose(logits.cpu().numpy(), ref.cpu().numpy(), atol=1e-5, rtol=0)
    exact = (pv_a.reshape(T, B, -1).cpu().numpy().astype(np.float64) @ Wro.astype(np.float64).T + bro)
    np.testing.assert_allclose(logits.cpu().numpy(), exact, atol=LOGIT_TOL / 10, rtol=0)


@pytest.mark.parametrize("rows,K,N", [(300, 8192, 48), (129, 64, 33), (1000, 8192, 24), (7, 32, 64)])
def test_readout_gemm_fast_path(dev, rows, K, N):
    """k_readout_v4 (K % 32 == 0): float4 loads, register double buffer, up to 64 stacked readout rows."""
    from snn_modulation_classification_amd import ops
    rng = np.random.RandomState(1)
    pv = rng.uniform(0, 1, size=(rows, K)).astype(np.float32)
    Wt = rng.uniform(-.01, .01, size=(N, K)).astype(np.float32)
    b = rng.uniform(-.01, .01, size=(N,)).astype(np.float32)
    out = ops.readout(cu(pv, dev), cu(Wt, dev), cu(b, dev)).cpu().numpy()
    ref = (pv.astype(np.float64) @ Wt.astype(np.float64).T + b).astype(np.float32)
    np.testing.assert_allclose(out, ref, atol=2e-5, rtol=0)


@pytest.mark.parametrize("rows,K,N", [(2100, 65536, 24), (2049, 131072, 48), (2500, 65792, 33)])
def test_readout_gemm_long_rows(dev, rows, K, N):
    """k_readout_ks: more than 2048 long rows (K >= 65536: T*B rows of a large plane) but fewer 128-row tiles than CUs — 32-row tiles,
    the K-chunk split over the 4 waves, partials combined in fixed order (run-to-run identical)."""
    from snn_modulation_classification_amd import ops
    rng = np.random.RandomState(2)
    pv = rng.uniform(0, 1, size=(rows, K)).astype(np.float32)
    Wt = rng.uniform(-.01, .01, size=(N, K)).astype(np.float32)
    b = rng.uniform(-.01, .01, size=(N,)).astype(np.float32)
    a1 = ops.readout(cu(pv, dev), cu(Wt, dev), cu(b, dev))
    a2 = ops.readout(cu(pv, dev), cu(Wt, dev), cu(b, dev))
    assert torch.equal(a1, a2)
    ref = (pv.astype(np.float64) @ Wt.astype(np.float64).T + b).astype(np.float32)
    np.testing.assert_allclose(a1.cpu().numpy(), ref, atol=1e-4, rtol=0)


@pytest.mark.parametrize("rows,K,N", [(64, 524288, 24), (5, 65536, 48), (700, 131072, 10), (512, 8192, 24), (37, 2048, 48),
                                      (2048, 8192, 33), (2300, 131072, 24)])
def test_readout_few_rows_long_k(dev, rows, K, N):
    """dcll_readout_splitk (through ops.readout): few rows — the per-step readouts, rows = batch — with K split over the
    chip: 4096-column slices of a very long K (128x128 plane), 256-column slices on the 16x16 plane (K = 8192; 128-column
    slices up to 512 rows of <= 32 readout rows: 256 instead of 128 workgroups); partial tiles summed in slice order
    (run-to-run identical)."""
    from snn_modulation_classification_amd import ops, _lib
    assert _lib.get().dcll_readout_splitk_scratch(rows, K, N) == \
        ((8 if rows > 2048 else K // 4096) if K >= 65536 else K // (128 if (rows <= 512 and N <= 32) else 256)) * rows * N
    assert _lib.get().dcll_readout_splitk_scratch(rows, 8192 + 32, N) == 0
    assert _lib.get().dcll_readout_splitk_scratch(4096, 8192, N) == 0          # many rows: the plain GEMM
    rng = np.random.RandomState(4)
    pv = rng.uniform(0, 1, size=(rows, K)).astype(np.float32)
    Wt = rng.uniform(-.002, .002, size=(N, K)).astype(np.float32)
    b = rng.uniform(-.01, .01, size=(N,)).astype(np.float32)
    a1 = ops.readout(cu(pv, dev), cu(Wt, dev), cu(b, dev))
    a2 = ops.readout(cu(pv, dev), cu(Wt, dev), cu(b, dev))
    assert torch.equal(a1, a2)
    ref = (pv.astype(np.float64) @ Wt.astype(np.float64).T + b).astype(np.float32)
    np.testing.assert_allclose(a1.cpu().numpy(), ref, atol=1e-4, rtol=0)


@pytest.mark.parametrize("case", ["mnist_l0", "mnist_l2", "pool3", "scalar_tau", "radio_l2_out", "ref_tuple"])
def test_backward_vs_torch_autograd(golden, golden_meta, dev, case):
    """dcll_conv_lif_backward (all four incoming gradients, pooling incl. ties routing, output layer) against torch
    autograd through the CPU oracle ops on the same step."""
    from snn_modulation_classification_amd import ops
    from oracle import torch_ref as R
    g = golden("g1_layer_steps.npz")
    m = golden_meta["g1"][case]
    sd = {k: torch.from_numpy(v) for k, v in g.sub("g1/%s/sd/" % case).items()}
    x = torch.from_numpy(g["g1/%s/x0" % case])
    B = x.shape[0]
    rng = np.random.RandomState(3)
    # --- CPU: autograd through the reference op sequence
    W = sd["i2h.weight"].clone().requires_grad_(True)
    b = sd["i2h.bias"].clone().requires_grad_(True)
    layer = R.RefConvLayer(dict(sd, **{"i2h.weight": W, "i2h.bias": b}), m["pad"], m["pool"], m["wrp"], m["alpharp"],
                           m["output_layer"])
    if m["output_layer"]:
        layer.out_w = sd["output_.weight"].clone().requires_grad_(True)
        layer.out_b = sd["output_.bias"].clone().requires_grad_(True)
    layer.init_state(B, x.shape[2:4])
    # emulate flatten.detach() of the reference for output_
    s_, pv_, v_, st = R.conv_lif_step(x, W, b, layer.alpha, layer.tau_m, layer.alphas, layer.tau_s, layer.state,
                                      layer.alpharp, layer.wrp, 1, layer.padding)
    pvp = R.max_pool(pv_, layer.pooling)
    flat = pvp.reshape(B, -1)
    p_ = torch.nn.functional.linear(flat, sd["i2o.weight"], sd["i2o.bias"])
    r_p = torch.from_numpy(rng.randn(*p_.shape).astype(np.float32))
    r_pv = torch.from_numpy(rng.randn(*pvp.shape).astype(np.float32)) * 0.1
    r_v = torch.from_numpy(rng.randn(*v_.shape).astype(np.float32)) * 0.01
    loss = (p_ * r_p).sum() + (pvp * r_pv).sum() + (v_ * r_v).sum()
    r_o = None
    if m["output_layer"]:
        o_ = torch.nn.functional.linear(flat.detach(), layer.out_w, layer.out_b)
        r_o = torch.from_numpy(rng.randn(*o_.shape).astype(np.float32))
        loss = loss + (o_ * r_o).sum()
    loss.backward()
    # --- GPU
    t = {k: cu(v.numpy(), dev) for k, v in sd.items()}
    d = ops.make_conv_desc(m["cin"], m["cout"], m["im"], m["k"], m["pad"], m["pool"], 24, m["output_layer"],
                           sd["i2h.alpha"].numel() > 1, m["wrp"], m["alpharp"])
    ch, cw, ph, pw = ops.conv_out_shape(d)
    eps0 = torch.zeros((B, m["cin"]) + tuple(m["im"]), device=dev)
    eps1 = torch.zeros_like(eps0)
    arp = torch.zeros((B, m["cout"], ch, cw), device=dev)
    s, p, o, pv, v = ops.conv_lif_step(d, x.to(dev), t["i2h.weight"], t["i2h.bias"], t["i2h.alpha"],
                                       t["i2h.tau_m__dt"], t["i2h.alphas"], t["i2h.tau_s__dt"], eps0, eps1, arp,
                                       t["i2o.weight"], t["i2o.bias"], t.get("output_.weight"), t.get("output_.bias"))
    dW, db, doW, dob = ops.conv_lif_backward(d, eps1, v, pv, r_p.to(dev), None if r_o is None else r_o.to(dev),
                                             r_pv.to(dev), r_v.to(dev), t["i2o.weight"], want_out=m["output_layer"])
    tol = lambda ref: dict(rtol=2e-3, atol=2e-5 * float(ref.abs().max()) + 1e-12)
    np.testing.assert_allclose(dW.cpu().numpy(), W.grad.numpy(), **tol(W.grad))
    np.testing.assert_allclose(db.cpu().numpy(), b.grad.numpy(), **tol(b.grad))
    if m["output_layer"]:
        np.testing.assert_allclose(doW.cpu().numpy(), layer.out_w.grad.numpy(), **tol(layer.out_w.grad))
        np.testing.assert_allclose(dob.cpu().numpy(), layer.out_b.grad.numpy(), **tol(layer.out_b.grad))


@pytest.mark.parametrize("cin,cout,out_layer,hw,B", [(1, 8, False, (128, 128), 2), (3, 4, True, (128, 128), 2),
                                                     (32, 32, False, (32, 48), 2), (32, 32, True, (128, 128), 2),
                                                     (1, 32, False, (16, 16), 300), (32, 32, True, (16, 16), 5),
                                                     (32, 32, False, (16, 16), 70), (32, 32, False, (16, 16), 130),
                                                     (32, 32, True, (16, 16), 300), (32, 32, False, (16, 16), 1030),
                                                     # K = 3 * 60 = 180, not a multiple of 32: the output_ gradient over batch
                                                     # chunks (k_bwd_outgrad_part) — open and closed form must split alike
                                                     (2, 3, True, (6, 10), 20)])
def test_backward_on_large_planes(dev, cin, cout, out_layer, hw, B):
    """dcll_conv_lif_backward on large planes incl. the argparse default 128x128 (train.py:40-41): the generic
    weight-gradient kernel stages the eps1 plane in LDS in row bands (two bands at 128 rows), the 32 -> 32 layers use the
    tiled MFMA kernel, the output_ gradient runs over K = c_out*h*w columns — against torch autograd through the CPU
    oracle ops."""
    from snn_modulation_classification_amd import ops
    from oracle import torch_ref as R
    rng = np.random.RandomState(17)
    # 32 -> 32 layers: the MFMA weight-gradient kernel over 16x16 tiles with their real halo; the two 16x16 cases: the
    # first layer's two-tile MFMA kernel k_bwd_wgrad_c1 (300 samples over 256 workgroups) and k_bwd_wgrad_c32 with a
    # sample's column tiles over 6 / 3 / 2 / 2 / 1 workgroups (B = 5 / 70 / 130 / 300 / 1030: since round 6 every batch from
    # 161 to 1024 samples runs 128 chunks x 2 column halves — 300 samples = ragged chunks of two and three jobs —, larger
    # ones 256 chunks x 1)
    Wn, bn, alpha, tau_m, alphas, tau_s = _rand_layer(rng, cin, cout, gain=3.0)
    sdn = _sd_from(Wn, bn, alpha, tau_m, alphas, tau_s, hw, rng=rng)
    K = cout * hw[0] * hw[1]
    if out_layer:
        sdn["output_.weight"] = rng.uniform(-.002, .002, size=(24, K)).astype(np.float32)
        sdn["output_.bias"] = rng.uniform(-.002, .002, size=(24,)).astype(np.float32)
    sd = {k: torch.from_numpy(v) for k, v in sdn.items()}
    x = torch.from_numpy((rng.uniform(size=(B, cin) + hw) < 0.2).astype(np.float32))
    W = sd["i2h.weight"].clone().requires_grad_(True)
    b = sd["i2h.bias"].clone().requires_grad_(True)
    layer = R.RefConvLayer(dict(sd, **{"i2h.weight": W, "i2h.bias": b}), 3, 1, 1.0, 0.65, out_layer)
    if out_layer:
        layer.out_w = sd["output_.weight"].clone().requires_grad_(True)
        layer.out_b = sd["output_.bias"].clone().requires_grad_(True)
    layer.init_state(B, hw)
    s_, pv_, v_, st = R.conv_lif_step(x, W, b, layer.alpha, layer.tau_m, layer.alphas, layer.tau_s, layer.state,
                                      layer.alpharp, layer.wrp, 1, layer.padding)
    flat = pv_.reshape(B, -1)
    p_ = torch.nn.functional.linear(flat, sd["i2o.weight"], sd["i2o.bias"])
    r_p = torch.from_numpy(rng.randn(*p_.shape).astype(np.float32))
    loss = (p_ * r_p).sum()
    r_o = None
    if out_layer:
        o_ = torch.nn.functional.linear(flat.detach(), layer.out_w, layer.out_b)
        r_o = torch.from_numpy(rng.randn(*o_.shape).astype(np.float32))
        loss = loss + (o_ * r_o).sum()
    loss.backward()
    t = {k: cu(v.numpy(), dev) for k, v in sd.items()}
    d = ops.make_conv_desc(cin, cout, hw, 7, 3, 1, 24, out_layer, True, 1.0, 0.65)
    eps0 = torch.zeros((B, cin) + hw, device=dev)
    eps1 = torch.zeros_like(eps0)
    arp = torch.zeros((B, cout) + hw, device=dev)
    s, p, o, pv, v = ops.conv_lif_step(d, x.to(dev), t["i2h.weight"], t["i2h.bias"], t["i2h.alpha"],
                                       t["i2h.tau_m__dt"], t["i2h.alphas"], t["i2h.tau_s__dt"], eps0, eps1, arp,
                                       t["i2o.weight"], t["i2o.bias"], t.get("output_.weight"), t.get("output_.bias"))
    dW, db, doW, dob = ops.conv_lif_backward(d, eps1, v, pv, r_p.to(dev), None if r_o is None else r_o.to(dev),
                                             None, None, t["i2o.weight"], want_out=out_layer)
    tol = lambda ref: dict(rtol=2e-3, atol=5e-5 * float(ref.abs().max()) + 1e-12)
    np.testing.assert_allclose(dW.cpu().numpy(), W.grad.numpy(), **tol(W.grad))
    np.testing.assert_allclose(db.cpu().numpy(), b.grad.numpy(), **tol(b.grad))
    if out_layer:
        np.testing.assert_allclose(doW.cpu().numpy(), layer.out_w.grad.numpy(), **tol(layer.out_w.grad))
        np.testing.assert_allclose(dob.cpu().numpy(), layer.out_b.grad.numpy(), **tol(layer.out_b.grad))
    # The open form (dcll_conv_lif_backward_open + dcll_grad_reduce_adam: the last reduction of the weight gradient and the
    # optimizer step in ONE launch, what ConvNetwork.learn runs): gradients bit-identical to the closed form for every
    # partial-row count (serial / 4 / 16 groups), parameters and Adam moments bit-identical to dcll_adam_step on them.
    hp = dict(lr=1e-6, weight_decay=10.0, beta1=0.0, beta2=.95, eps=1e-8)
    hp2 = dict(lr=1e-4, weight_decay=0.0, beta1=.9, beta2=.999, eps=1e-8)

    def adam_entries(params, grads, hps, step):
        return [dict(param=q, grad=g_, exp_avg=torch.zeros_like(q), exp_avg_sq=torch.zeros_like(q), step=step, **h)
                for q, g_, h in zip(params, grads, hps)]
    names = ["i2h.weight", "i2h.bias"] + (["output_.weight", "output_.bias"] if out_layer else [])
    hps = [hp, hp] + ([hp2, hp2] if out_layer else [])
    closed_p = [t[n].clone() for n in names]
    closed_t = adam_entries(closed_p, [dW, db] + ([doW, dob] if out_layer else []), hps, 3)
    ops.adam_step(closed_t)
    out2 = {}
    dW2, db2, doW2, dob2 = ops.conv_lif_backward(d, eps1, v, pv, r_p.to(dev), None if r_o is None else r_o.to(dev), None, None,
                                                 t["i2o.weight"], want_out=out_layer, out=out2, open_reduce=True)
    open_p = [t[n].clone() for n in names]
    open_t = adam_entries(open_p, [dW2, db2] + ([doW2, dob2] if out_layer else []), hps, 3)
    ops.grad_reduce_adam([dict(out2['parts'], adam_w=0, adam_b=1)], open_t)
    assert torch.equal(dW2, dW) and torch.equal(db2, db)
    if out_layer:
        assert torch.equal(doW2, doW) and torch.equal(dob2, dob)
    for a, b_ in zip(closed_t, open_t):
        for key in ("param", "exp_avg", "exp_avg_sq"):
            assert torch.equal(a[key], b_[key]), key
    assert not torch.equal(open_p[0], t["i2h.weight"])
    # reduce only (no optimizer entries): the gradients alone
    out3 = {}
    dW3, db3, _, _ = ops.conv_lif_backward(d, eps1, v, pv, r_p.to(dev), None, None, None, t["i2o.weight"], want_out=False,
                                           out=out3, open_reduce=True)
    dW3.fill_(7.0)
    ops.grad_reduce_adam([dict(out3['parts'])], [])
    assert torch.equal(dW3, dW) and torch.equal(db3, db)
    # v == NULL (a layer without pooling): sigmoid' from the stored pv — the bits the kernel recomputes from v, so the
    # learning forward need not write the membrane map at all (ConvNetwork.learn: want_v=False)
    if d.pool_h == 1 and d.pool_w == 1 and d.target <= 32:
        dW4, db4, doW4, dob4 = ops.conv_lif_backward(d, eps1, None, pv, r_p.to(dev), None if r_o is None else r_o.to(dev), None,
                                                     None, t["i2o.weight"], want_out=out_layer)
        assert torch.equal(dW4, dW) and torch.equal(db4, db)
        if out_layer:
            assert torch.equal(doW4, doW) and torch.equal(dob4, dob)
    else:
        with pytest.raises(Exception):
            ops.conv_lif_backward(d, eps1, None, pv, r_p.to(dev), None, None, None, t["i2o.weight"], want_out=False)


@pytest.mark.parametrize("B,from_pv", [(70, True), (300, False), (5, True)])
def test_backward_open_multi_equals_the_per_layer_calls(dev, B, from_pv):
    """dcll_conv_lif_backward_open_multi (ABI 6): the open backward of the three slices of radio_ml_conv.yaml in one call — their
    dv launches as ONE launch (k_bwd_dv_nopool_m; launch log) — == dcll_conv_lif_backward_open per layer, bit for bit: the
    reduced weight / bias gradients and the output_ gradients; with and without the membrane map (v == NULL: sigmoid' from pv).
    A pooling layer among the items: the call falls back to one dv launch per layer, same results."""
    from snn_modulation_classification_amd import ops
    rng = np.random.RandomState(B)

    def layer(cin, out_layer, pool=(1, 1)):
        d = ops.make_conv_desc(cin, 32, (16, 16), (7, 7), (3, 3), pool, 24, out_layer, True, 1.0)
        ch, cw, ph, pw = ops.conv_out_shape(d)
        K = 32 * ph * pw
        v = cu(rng.randn(B, 32, ch, cw).astype(np.float32), dev)
        pv = torch.sigmoid(v) if pool == (1, 1) else cu(rng.uniform(0, 1, size=(B, 32, ph, pw)).astype(np.float32), dev)
        return dict(d=d, eps1=cu(rng.uniform(0, 3, size=(B, cin, 16, 16)).astype(np.float32), dev), v=v, pv=pv,
                    g_p=cu(rng.randn(B, 24).astype(np.float32) * 1e-3, dev),
                    g_o=cu(rng.randn(B, 24).astype(np.float32) * 1e-3, dev) if out_layer else None,
                    W=cu(rng.uniform(-.005, .005, size=(24, K)).astype(np.float32), dev), out_layer=out_layer)

    def run(layers, multi):
        outs, deferred = [], ([] if multi else None)
        for L in layers:
            out = {}
            v = None if (from_pv and L['d'].pool_h == 1) else L['v']
            ops.conv_lif_backward(L['d'], L['eps1'], v, L['pv'], L['g_p'], L['g_o'], None, None, L['W'], want_out=L['out_layer'],
                                  out=out, open_reduce=True, defer=deferred)
            outs.append(out)
        if multi:
            ops.conv_lif_backward_open_multi(deferred)
        for out in outs:
            out['dW'].fill_(7.0)
            ops.grad_reduce_adam([dict(out['parts'])], [])
        return outs
    layers = [layer(1, False), layer(32, False), layer(32, True)]
    single = run(layers, False)
    with ops.kernel_trace() as tr:
        multi = run(layers, True)
    assert tr.count("k_bwd_dv_nopool_m") == 1 and tr.count("k_bwd_dv") == 0, tr.names
    assert tr.count("k_bwd_wgrad_c32") == 2 and tr.count("k_bwd_wgrad_c1") == 1 and tr.count("k_bwd_outgrad_mfma") == 1
    for a, b in zip(single, multi):
        assert torch.equal(a['dW'], b['dW']) and torch.equal(a['db'], b['db'])
    assert torch.equal(single[2]['d_outW'], multi[2]['d_outW']) and torch.equal(single[2]['d_outb'], multi[2]['d_outb'])
    # a pooling layer among the items: no joint dv launch, same results
    mixed = [layer(32, False), layer(32, False, pool=(2, 2))]
    single = run(mixed, False)
    with ops.kernel_trace() as tr:
        multi = run(mixed, True)
    assert tr.count("k_bwd_dv_nopool_m") == 0 and tr.count("k_bwd_dv") == 2, tr.names
    for a, b in zip(single, multi):
        assert torch.equal(a['dW'], b['dW']) and torch.equal(a['db'], b['db'])


def test_edge_cases_empty_and_single(dev):
    """Empty batch / zero timesteps are no-ops, B = 1 and T = 1 work (the reference itself breaks at B = 1 in
    iq2spiketrain's squeeze), invalid windows are rejected with ValueError."""
    from snn_modulation_classification_amd import ops
    from oracle import c_oracle as C
    rng = np.random.RandomState(2)
    W, b, alpha, tau_m, alphas, tau_s = _rand_layer(rng, 32, 32, gain=3.0)
    d = ops.make_conv_desc(32, 32, (16, 16), 7, 3, 1, 24, False, True, 1.0)
    tau4 = cu(np.stack([alpha, tau_m, alphas, tau_s]), dev)
    z = lambda B: [torch.zeros((B, 32, 16, 16), device=dev) for _ in range(3)]
    # T = 0 and B = 0: nothing happens, nothing crashes
    st = z(2)
    spk, pv, _ = ops.conv_lif_sequence(d, torch.zeros((0, 2, 32, 8), device=dev, dtype=torch.int32), cu(W, dev), cu(b, dev),
                                       tau4, *st, 0, 2)
    assert spk.shape == (0, 2, 32, 8) and all(float(s.abs().sum()) == 0 for s in st)
    spk, pv, _ = ops.conv_lif_sequence(d, torch.zeros((3, 0, 32, 8), device=dev, dtype=torch.int32), cu(W, dev), cu(b, dev),
                                       tau4, *z(0), 3, 0)
    assert spk.shape == (3, 0, 32, 8)
    assert ops.readout(torch.zeros((0, 64), device=dev), torch.zeros((5, 64), device=dev), None).shape == (0, 5)
    assert ops.pack_spikes(torch.zeros((0, 32), device=dev)).shape == (0, 1)
    # B = 1, T = 1 against the oracle
    sd = _sd_from(W, b, alpha, tau_m, alphas, tau_s, (16, 16), rng=rng)
    orc = C.OracleConvLayer(sd, (16, 16), 3, 1, 1.0)
    x = (rng.uniform(size=(1, 1, 32, 256)) < 0.3).astype(np.float32)
    st = z(1)
    spk, pv, v = ops.conv_lif_sequence(d, ops.pack_spikes(cu(x, dev)), cu(W, dev), cu(b, dev), tau4, *st, 1, 1, want_v=True)
    oo, op, opv, ov, os_ = orc.forward(x[0].reshape(1, 32, 16, 16))
    assert bits_equal(v[0].cpu().numpy(), ov)
    assert np.array_equal(ops.unpack_spikes(spk).cpu().numpy().reshape(1, 32, 16, 16), os_)
    # invalid IQ window
    with pytest.raises(ValueError):
        ops.iq_encode(torch.zeros((2, 2, 16), device=dev), torch.zeros(15, device=dev), torch.zeros(15, device=dev),
                      10, 8, 16, 16)
    # vote over an empty window (t_begin == T): no class has a vote -> -1
    c, vt = ops.argmax_vote(torch.zeros((4, 3, 5), device=dev), t_begin=4)
    assert vt.cpu().tolist() == [-1, -1, -1] and c.shape == (4, 3)


def test_operand_validation_before_the_abi(dev):
    """Wrong dtype / shape is rejected on the host (TypeError / ValueError) instead of becoming an OOB access."""
    from snn_modulation_classification_amd import ops
    d = ops.make_conv_desc(1, 32, (16, 16), 7, 3, 1, 24, False, True, 1.0)
    W, b = torch.zeros((32, 1, 7, 7), device=dev), torch.zeros(32, device=dev)
    tau4 = torch.ones((4, 1), device=dev)
    st = lambda B, c: torch.zeros((B, c, 16, 16), device=dev)
    with pytest.raises(TypeError):        # int64 cells
        ops.conv_lif_sequence_cells(d, torch.zeros((3, 2), device=dev, dtype=torch.int64), W, b, tau4, st(2, 1), st(2, 1),
                                    st(2, 32), 3, 2)
    with pytest.raises(ValueError):       # state for a different batch size
        ops.conv_lif_sequence_cells(d, torch.zeros((3, 2), device=dev, dtype=torch.int32), W, b, tau4, st(5, 1), st(2, 1),
                                    st(2, 32), 3, 2)
    with pytest.raises(ValueError):       # weights of another geometry
        ops.conv_lif_sequence_cells(d, torch.zeros((3, 2), device=dev, dtype=torch.int32), torch.zeros((32, 1, 5, 5), device=dev),
                                    b, tau4, st(2, 1), st(2, 1), st(2, 32), 3, 2)
    d32 = ops.make_conv_desc(32, 32, (16, 16), 7, 3, 1, 24, False, True, 1.0)
    with pytest.raises(ValueError):       # packed input with the wrong word count
        ops.conv_lif_sequence(d32, torch.zeros((3, 2, 32, 4), device=dev, dtype=torch.int32),
                              torch.zeros((32, 32, 7, 7), device=dev), b, torch.ones((4, 32), device=dev), st(2, 32),
                              st(2, 32), st(2, 32), 3, 2)
    with pytest.raises(ValueError):       # readout bias of the wrong length
        ops.readout(torch.zeros((4, 64), device=dev), torch.zeros((5, 64), device=dev), torch.zeros(4, device=dev))


def test_tiled_sequence_state_update_is_race_free(dev):
    """The tiled kernels read a tile's initial traces plus a 3-pixel halo that neighbouring tiles own and write their
    interior back at the end; nothing orders the workgroups of a grid.  With far more workgroups than CUs (40 samples
    x 16 tiles = 640 > 256) and a non-zero initial state, every sample's spikes and final state must still equal the
    C oracle bit for bit: the launch snapshots the initial state (state_scratch) and reads only the snapshot."""
    from snn_modulation_classification_amd import ops
    from oracle import c_oracle as C
    rng = np.random.RandomState(21)
    hw, (H, Wd), T, B, wrp = (64, 64), (64, 64), 2, 40, 1.0
    # 32 -> 32 layer (k_lif_seq_c32t)
    W, b, alpha, tau_m, alphas, tau_s = _rand_layer(rng, 32, 32, gain=3.0)
    sd = _sd_from(W, b, alpha, tau_m, alphas, tau_s, hw, rng=rng)
    orc = C.OracleConvLayer(sd, hw, 3, 1, wrp)
    orc.init_state(B)
    orc.state[0][...] = rng.uniform(0, 5, size=orc.state[0].shape)
    orc.state[1][...] = rng.uniform(0, 50, size=orc.state[1].shape)
    orc.state[2][...] = -rng.uniform(0, 2, size=orc.state[2].shape)
    eps0, eps1, arp = [cu(s.copy(), dev) for s in orc.state]
    x = (rng.uniform(size=(T, B, 32, H * Wd)) < 0.08).astype(np.float32)
    d = ops.make_conv_desc(32, 32, hw, 7, 3, 1, 24, False, True, wrp)
    tau4 = cu(np.stack([alpha, tau_m, alphas, tau_s]), dev)
    spk, pv, _ = ops.conv_lif_sequence(d, ops.pack_spikes(cu(x, dev)), cu(W, dev), cu(b, dev), tau4, eps0, eps1, arp, T, B)
    spk_d = ops.unpack_spikes(spk).cpu().numpy().reshape(T, B, 32, H, Wd)
    for t in range(T):
        os_ = orc.forward(x[t].reshape(B, 32, H, Wd), want_v=False)[4]
        assert np.array_equal(spk_d[t], os_), (t, np.argwhere(spk_d[t] != os_)[:5])
    for got, want in zip((eps0, eps1, arp), orc.state):
        assert bits_equal(got.cpu().numpy(), want)
    # first layer (k_lif_seq_c1t), 8 samples x 16 tiles x ... : same protocol
    W, b, alpha, tau_m, alphas, tau_s = _rand_layer(rng, 1, 32, gain=3.0)
    sd = _sd_from(W, b, alpha, tau_m, alphas, tau_s, hw, rng=rng)
    B1, T1 = 96, 3
    orc = C.OracleConvLayer(sd, hw, 3, 1, wrp)
    orc.init_state(B1)
    orc.state[0][...] = rng.uniform(0, 5, size=orc.state[0].shape)
    orc.state[1][...] = rng.uniform(0, 50, size=orc.state[1].shape)
    eps0, eps1, arp = [cu(s.copy(), dev) for s in orc.state]
    cells = rng.randint(0, H * Wd, size=(T1, B1)).astype(np.int32)
    d = ops.make_conv_desc(1, 32, hw, 7, 3, 1, 24, False, True, wrp)
    tau4 = cu(np.stack([alpha, tau_m, alphas, tau_s]), dev)
    spk, pv, _ = ops.conv_lif_sequence_cells(d, cu(cells, dev), cu(W, dev), cu(b, dev), tau4, eps0, eps1, arp, T1, B1)
    spk_d = ops.unpack_spikes(spk).cpu().numpy().reshape(T1, B1, 32, H, Wd)
    for t in range(T1):
        xx = np.zeros((B1, 1, H * Wd), np.float32)
        xx[np.arange(B1), 0, cells[t]] = 1
        os_ = orc.forward(xx.reshape(B1, 1, H, Wd), want_v=False)[4]
        assert np.array_equal(spk_d[t], os_), t
    for got, want in zip((eps0, eps1, arp), orc.state):
        assert bits_equal(got.cpu().numpy(), want)
    # the scratch is part of the ABI contract on these planes: a NULL pointer is refused, not dereferenced
    from snn_modulation_classification_amd import _lib
    import ctypes
    rc = _lib.get().dcll_conv_lif_sequence_cells(ctypes.byref(d), _lib.ptr(cu(cells, dev)), _lib.ptr(cu(W, dev)),
                                                 _lib.ptr(cu(b, dev)), _lib.ptr(tau4), _lib.ptr(eps0), _lib.ptr(eps1),
                                                 _lib.ptr(arp), None, None, None, None, None, 0, None, T1, B1, None)
    assert rc == _lib.DCLL_ERR_INVALID and b"state_scratch" in _lib.get().dcll_last_error()


@pytest.mark.parametrize("per_step,T,iter0", [(32 * 256 * 3, 45, 0), (1001, 61, 17), (8, 20, 0), (4096, 19, 0),
                                              (12345, 1, 39), (64, 1, 38)])
def test_pv_lowhigh_counts_match_numpy_histogram(dev, per_step, T, iter0):
    """dcll_pv_lowhigh == bins 0 and 18 of np.histogram(pv, np.linspace(0, 1, 20)) on the steps whose 1-based iteration
    count is a multiple of 20 (DCLLBase.forward, reference :658-661), incl. values exactly on / next to the bin edges."""
    from snn_modulation_classification_amd import ops
    rng = np.random.RandomState(3)
    pv = rng.uniform(0, 1, size=(T, per_step)).astype(np.float32)
    edges = np.linspace(0, 1, 20)
    special = np.array([0.0, 1.0, edges[1], edges[18], np.float32(edges[1]), np.float32(edges[18]),
                        np.nextafter(np.float32(edges[1]), np.float32(0)), np.nextafter(np.float32(edges[1]), np.float32(1)),
                        np.nextafter(np.float32(edges[18]), np.float32(0)), np.nextafter(np.float32(edges[18]), np.float32(1))],
                       dtype=np.float32)
    pv[:, :min(per_step, special.size)] = special[:min(per_step, special.size)]
    got = ops.pv_lowhigh(cu(pv, dev), T, iter0).cpu().numpy()
    steps = [t for t in range(T) if (iter0 + t + 1) % 20 == 0]
    assert got.shape == (len(steps), 2) and ops.pv_lowhigh_steps(iter0, T) == len(steps)
    for k, t in enumerate(steps):
        h = np.histogram(pv[t], bins=edges)[0]
        assert (int(got[k, 0]), int(got[k, 1])) == (int(h[0]), int(h[-1])), (k, t)


@pytest.mark.parametrize("rows,K,N", [(2049, 8192, 24), (5000, 8192, 48), (2111, 8192, 10), (4096, 2048, 33),
                                      (2300, 64, 48), (130, 8192, 24)])
def test_readout_direct_forms(dev, rows, K, N):
    """k_readout_direct (LDS-free 16x16x4 MFMA tiles, operands straight from global memory): its standalone form
    (what dcll_readout picks for rows > 2048, K % 64 == 0, N <= 48) and its co-resident <= 64-VGPR form, against
    float64 and against the LDS-staged kernels; ragged row / column tiles; a row's result does not depend on how many
    rows the launch has."""
    from snn_modulation_classification_amd import ops
    rng = np.random.RandomState(7)
    pv = rng.uniform(0, 1, size=(rows, K)).astype(np.float32)
    W = rng.uniform(-.0055, .0055, size=(N, K)).astype(np.float32)
    b = rng.uniform(-.0055, .0055, size=(N,)).astype(np.float32)
    ref = pv.astype(np.float64) @ W.astype(np.float64).T + b
    dpv, dW, db = cu(pv, dev), cu(W, dev), cu(b, dev)
    outs = {}
    for mode in (ops.READOUT_AUTO, ops.READOUT_CORESIDENT, ops.READOUT_LDS, ops.READOUT_T16):
        guard = torch.full((rows + 8, N), 7.0, device=dev)          # rows behind the output must stay untouched
        out = ops.readout(dpv, dW, db, out=guard[:rows], mode=mode)
        assert float(guard[rows:].min()) == 7.0 and float(guard[rows:].max()) == 7.0
        outs[mode] = out.cpu().numpy()
        np.testing.assert_allclose(outs[mode], ref, atol=2e-5, rtol=0)
    if rows > 2100:
        part = ops.readout(dpv[:2100].contiguous(), dW, db, mode=ops.READOUT_CORESIDENT).cpu().numpy()
        assert np.array_equal(part, outs[ops.READOUT_CORESIDENT][:2100])
        part = ops.readout(dpv[:2100].contiguous(), dW, db).cpu().numpy()
        assert np.array_equal(part, outs[ops.READOUT_AUTO][:2100])


@pytest.mark.parametrize("cin,hw,wrp,T,B,zero_state", [(64, (16, 64), 1.0, 7, 2, False), (64, (16, 16), 1.0, 6, 3, False),
                                                       (64, (16, 4), 0.0, 5, 3, True), (64, (16, 2), 1.0, 6, 5, False),
                                                       (64, (4, 32), 1.0, 5, 3, False), (64, (16, 8), 1.0, 7, 5, False), (64, (16, 16), 0.0, 9, 2, False),
                                                       (64, (16, 8), 1.0, 5, 4, False), (64, (16, 4), 1.0, 5, 4, False), (64, (16, 2), 1.0, 5, 8, False),
                                                       (1, (16, 128), 1.0, 9, 2, False),
                                                       (1, (2, 128), 0.0, 6, 3, True),
                                                       (1, (1, 128), 1.0, 7, 5, False), (1, (8, 32), 1.0, 8, 3, False),
                                                       (1, (4, 256), 1.0, 6, 3, False), (1, (64, 2), 0.0, 6, 2, False)])
def test_sequence_w3_vs_oracle(dev, cin, hw, wrp, T, B, zero_state):
    """k_lif_seq_w3 — the fused all-T kernel of the radio_ml_conv_ref.yaml geometry (64 channels, (1,3) kernel, pad
    (0,1), max-pool (1,2); pixel tiles over the flattened plane, pooling pairs in lanes j / j+16) — == C oracle
    stepping, bit for bit: un-pooled v, POOLED spikes (packed) and the final state; pooled pv within the sigmoid
    tolerance.  Widths from 64 down to 2 (a tile = 1/2 ... 16 rows), ragged workgroups (B x tiles not a multiple of 8)
    and full ones (the kernel's FULL specialisation) at every narrow width, first layer from cell indices."""
    from snn_modulation_classification_amd import ops
    from oracle import c_oracle as C
    rng = np.random.RandomState(31)
    H, Wd = hw
    cout, n = 64, cin * 3
    stdv = 1.0 / np.sqrt(n) / 250
    W = (rng.uniform(-stdv * 1e-2, stdv * 1e-2, size=(cout, cin, 1, 3)) * 3.0).astype(np.float32)
    b = rng.uniform(-stdv, stdv, size=(cout,)).astype(np.float32)
    taum, taus = rng.uniform(5, 35, size=cin) * 1e-3, rng.uniform(5, 10, size=cin) * 1e-3
    alpha, alphas = (1 - 1e-3 / taum).astype(np.float32), (1 - 1e-3 / taus).astype(np.float32)
    tau_m = (np.float32(1) / (np.float32(1) - alpha)).astype(np.float32)
    tau_s = (np.float32(1) / (np.float32(1) - alphas)).astype(np.float32)
    bc = lambda a: np.ascontiguousarray(np.broadcast_to(a[:, None, None], (cin,) + hw)).astype(np.float32)
    K = cout * H * (Wd // 2)
    sd = {"i2h.weight": W, "i2h.bias": b, "i2h.alpha": bc(alpha), "i2h.tau_m__dt": bc(tau_m), "i2h.alphas": bc(alphas),
          "i2h.tau_s__dt": bc(tau_s), "i2o.weight": rng.uniform(-.005, .005, size=(24, K)).astype(np.float32),
          "i2o.bias": np.zeros(24, np.float32)}
    orc = C.OracleConvLayer(sd, hw, (0, 1), (1, 2), wrp)
    orc.init_state(B)
    if not zero_state:
        orc.state[0][...] = rng.uniform(0, 5, size=orc.state[0].shape)
        orc.state[1][...] = rng.uniform(0, 50, size=orc.state[1].shape)
        if wrp > 0:
            orc.state[2][...] = -rng.uniform(0, 2, size=orc.state[2].shape)
    state = [cu(s_.copy(), dev) for s_ in orc.state]
    eps0, eps1 = state[0], state[1]
    arp = state[2] if wrp > 0 else None
    d = ops.make_conv_desc(cin, cout, hw, (1, 3), (0, 1), (1, 2), 24, False, True, wrp)
    assert ops.conv_out_shape(d) == (H, Wd, H, Wd // 2)
    tau4 = cu(np.stack([alpha, tau_m, alphas, tau_s]), dev)
    want_spk = (H * Wd) % 64 == 0
    if cin == 1:
        cells = rng.randint(0, H * Wd, size=(T, B)).astype(np.int32)
        cells[0, 0], cells[1, 0] = 0, H * Wd - 1
        x = np.zeros((T, B, 1, H * Wd), np.float32)
        x[np.arange(T)[:, None], np.arange(B)[None, :], 0, cells] = 1
        spk, pv, v = ops.conv_lif_sequence_cells(d, cu(cells, dev), cu(W, dev), cu(b, dev), tau4, eps0, eps1, arp, T, B,
                                                 want_spikes=want_spk, want_v=True)
    else:
        x = (rng.uniform(size=(T, B, cin, H * Wd)) < 0.1).astype(np.float32)
        x[0] = rng.uniform(size=(B, cin, H * Wd)) < 0.5
        spk, pv, v = ops.conv_lif_sequence(d, ops.pack_spikes(cu(x, dev)), cu(W, dev), cu(b, dev), tau4, eps0, eps1, arp,
                                           T, B, want_spikes=want_spk, want_v=True)
    torch.cuda.synchronize()
    assert v.shape == (T, B, cout, H, Wd) and pv.shape == (T, B, cout, H, Wd // 2)
    v, pv = v.cpu().numpy(), pv.cpu().numpy()
    if want_spk:
        assert spk.shape == (T, B, cout, H * Wd // 64)
        spk_d = ops.unpack_spikes(spk).cpu().numpy().reshape(T, B, cout, H, Wd // 2)
    nspk = 0
    for t in range(T):
        oo, op, opv, ov, os_ = orc.forward(x[t].reshape(B, cin, H, Wd))
        assert bits_equal(v[t], ov), (t, np.abs(v[t] - ov).max(), np.argwhere(v[t] != ov)[:5])
        if want_spk:
            assert np.array_equal(spk_d[t], os_), (t, np.argwhere(spk_d[t] != os_)[:5])
        np.testing.assert_allclose(pv[t], opv, atol=PV_TOL, rtol=0)
        nspk += os_.sum()
    assert bits_equal(eps0.cpu().numpy(), orc.state[0])
    assert bits_equal(eps1.cpu().numpy(), orc.state[1])
    if wrp > 0:
        assert bits_equal(arp.cpu().numpy(), orc.state[2])
    assert 0.005 < nspk / (T * B * cout * H * Wd / 2) < 0.95, "degenerate test"



def test_sequence_w3_first_layer_grid_beyond_residency_with_carried_state(dev):
    """k_lif_seq_w3f reads the input traces of a sample from 8 channel-group waves per segment (and from the neighbouring
    segments' halo lanes); until round 6 the channel-group-0 wave wrote the advanced traces back IN PLACE, so on a grid
    larger than residency a wave that started after that writer had retired took end-of-sequence traces as its initial
    state (round-5 advisor, high).  Measured with experiments/w3f_race_check.py on the round-5 library: 128 of 2 048 samples
    wrong at T = 32 (none at B = 384, T = 6 — hence this size: 262 144 waves against ~5 000 resident).  The traces are now
    advanced by k_w3f_traces_advance behind the kernel.  Non-zero initial state; EVERY sample of the batch against the
    same sample run in a co-resident pair (membrane map and final state, bit for bit), samples spread over the batch
    against the C oracle (v, pooled spikes, state), the whole batch's final traces against the pinned recurrence in
    numpy fp32, and the launch log."""
    from snn_modulation_classification_amd import ops
    from oracle import c_oracle as C
    rng = np.random.RandomState(77)
    H, Wd, cin, cout, wrp, T, B = 16, 128, 1, 64, 1.0, 32, 2048
    hw = (H, Wd)
    stdv = 1.0 / np.sqrt(3) / 250
    W = (rng.uniform(-stdv * 1e-2, stdv * 1e-2, size=(cout, cin, 1, 3)) * 3.0).astype(np.float32)
    b = rng.uniform(-stdv, stdv, size=(cout,)).astype(np.float32)
    alpha, alphas = np.float32([1 - 1e-3 / 20e-3]), np.float32([1 - 1e-3 / 7e-3])
    tau_m = (np.float32(1) / (np.float32(1) - alpha)).astype(np.float32)
    tau_s = (np.float32(1) / (np.float32(1) - alphas)).astype(np.float32)
    bc = lambda a: np.ascontiguousarray(np.broadcast_to(a[:, None, None], (cin,) + hw)).astype(np.float32)
    K = cout * H * (Wd // 2)
    sd = {"i2h.weight": W, "i2h.bias": b, "i2h.alpha": bc(alpha), "i2h.tau_m__dt": bc(tau_m), "i2h.alphas": bc(alphas),
          "i2h.tau_s__dt": bc(tau_s), "i2o.weight": np.zeros((24, K), np.float32), "i2o.bias": np.zeros(24, np.float32)}
    e0 = rng.uniform(0, 5, size=(B, 1, H, Wd)).astype(np.float32)
    e1 = rng.uniform(0, 50, size=(B, 1, H, Wd)).astype(np.float32)
    ar = -rng.uniform(0, 2, size=(B, cout, H, Wd)).astype(np.float32)
    cells = rng.randint(0, H * Wd, size=(T, B)).astype(np.int32)
    eps0, eps1, arp = cu(e0, dev), cu(e1, dev), cu(ar, dev)
    d = ops.make_conv_desc(cin, cout, hw, (1, 3), (0, 1), (1, 2), 24, False, True, wrp)
    tau4 = cu(np.stack([alpha, tau_m, alphas, tau_s]), dev)
    dW, db, dcells = cu(W, dev), cu(b, dev), cu(cells, dev)
    with ops.kernel_trace() as launched:
        spk, pv, v = ops.conv_lif_sequence_cells(d, dcells, dW, db, tau4, eps0, eps1, arp, T, B, want_spikes=True, want_v=True)
    torch.cuda.synchronize()
    assert launched.names == ["k_lif_seq_w3f", "k_w3f_traces_advance"], launched.names
    # every sample == the same sample in a pair of its own (all waves co-resident: what the small oracle tests cover)
    de0, de1, dar = cu(e0, dev), cu(e1, dev), cu(ar, dev)
    bad = []
    for k in range(0, B, 2):
        q0, q1, q2 = de0[k:k + 2].clone(), de1[k:k + 2].clone(), dar[k:k + 2].clone()
        _, _, vk = ops.conv_lif_sequence_cells(d, dcells[:, k:k + 2].contiguous(), dW, db, tau4, q0, q1, q2, T, 2,
                                               want_spikes=True, want_v=True)
        if not (torch.equal(vk, v[:, k:k + 2]) and torch.equal(q0, eps0[k:k + 2]) and torch.equal(q1, eps1[k:k + 2]) and
                torch.equal(q2, arp[k:k + 2])):
            bad.append(k)
    assert not bad, "%d sample pairs differ from their co-resident run, first %s" % (len(bad), bad[:8])
    pick = np.array([0, 1, 255, 256, 511, 1023, 1024, 1500, 2046, 2047])
    orc = C.OracleConvLayer(sd, hw, (0, 1), (1, 2), wrp)
    orc.init_state(len(pick))
    orc.state[0][...], orc.state[1][...], orc.state[2][...] = e0[pick], e1[pick], ar[pick]
    idx = torch.from_numpy(pick).to(dev)
    v_p = v[:, idx].cpu().numpy()
    spk_p = ops.unpack_spikes(spk[:, idx].contiguous()).cpu().numpy().reshape(T, len(pick), cout, H, Wd // 2)
    for t in range(T):
        x = np.zeros((len(pick), 1, H * Wd), np.float32)
        x[np.arange(len(pick)), 0, cells[t, pick]] = 1
        oo, op, opv, ov, os_ = orc.forward(x.reshape(len(pick), 1, H, Wd))
        assert bits_equal(v_p[t], ov), (t, np.argwhere(v_p[t] != ov)[:5])
        assert np.array_equal(spk_p[t], os_), (t, np.argwhere(spk_p[t] != os_)[:5])
    g0, g1 = eps0.cpu().numpy(), eps1.cpu().numpy()
    assert bits_equal(g0[pick], orc.state[0]) and bits_equal(g1[pick], orc.state[1])
    assert bits_equal(arp[idx].cpu().numpy(), orc.state[2])
    # every sample's final traces: the pinned recurrence, each operation rounded to fp32 (dcll/pytorch_libdcll.py:493-494)
    w0, w1 = e0.reshape(B, -1).copy(), e1.reshape(B, -1).copy()
    for t in range(T):
        x = np.zeros_like(w0)
        x[np.arange(B), cells[t]] = 1
        w0 = x * tau_s[0] + alphas[0] * w0
        w1 = alpha[0] * w1 + w0 * tau_m[0]
    assert w0.dtype == np.float32 and bits_equal(g0.reshape(B, -1), w0) and bits_equal(g1.reshape(B, -1), w1)


@pytest.mark.parametrize("case", ["radio_l1", "radio_l2_out", "radio_norp", "scalar_tau", "mnist_l2", "ref_tuple", "pool3"])
def test_integration_md_binding_executes_and_matches_oracle(golden, golden_meta, dev, case):
    """INTEGRATION.md section B documents the binding a maintainer of the reference would add to route
    Conv2dDCLLlayer.forward (reference dcll/pytorch_libdcll.py:599-608, called at :657) through dcll_conv_lif_step.  This
    test EXECUTES that documentation: the fenced block with `def hip_forward` is extracted from INTEGRATION.md, exec'd
    (only the library's file name is replaced by its in-tree path), bound as `forward` onto a minimal object that carries
    the attribute set of the REFERENCE's layer classes (i2h.{in_channels, out_channels, kernel_size, padding, weight, bias,
    alpha, tau_m__dt, alphas, tau_s__dt, state, init_state, get_output_shape[, wrp, alpharp]}, i2o, output_, im_dims,
    pooling, target_size, output_layer, output_shape — none of this build's classes), and three free-running steps of a
    G1 golden case are checked against the C oracle: v / spikes / state bit for bit, logits within 1e-4."""
    import os
    import re
    import types
    from collections import namedtuple
    from conftest import ROOT
    from oracle import c_oracle as C
    from snn_modulation_classification_amd import _lib
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = [b for b in re.findall(r"```python\n(.*?)```", text, flags=re.S) if "def hip_forward" in b]
    assert len(blocks) == 1, "INTEGRATION.md must hold exactly one fenced python block defining hip_forward"
    src = blocks[0]
    assert src.count('ctypes.CDLL("libdcll_hip.so")') == 1
    _lib.get()                                     # (built, and torch's HIP runtime is the one in the process)
    src = src.replace('ctypes.CDLL("libdcll_hip.so")', 'ctypes.CDLL(%r)' % _lib.SO_PATH)
    ns = {}
    exec(compile(src, "INTEGRATION.md:hip_forward", "exec"), ns)

    g = golden("g1_layer_steps.npz")
    m = golden_meta["g1"][case]
    sd = g.sub("g1/%s/sd/" % case)
    t = {k: cu(v, dev) for k, v in sd.items()}
    B, cin, cout = m["B"], m["cin"], m["cout"]
    pair = lambda v: tuple(v) if hasattr(v, "__len__") else (v, v)
    kh, kw = pair(m["k"])
    pad, pool, im = pair(m["pad"]), pair(m["pool"]), tuple(m["im"])
    refractory = m["wrp"] > 0
    State = namedtuple("NeuronState", ("eps0", "eps1", "arp") if refractory else ("eps0", "eps1"))

    i2h = types.SimpleNamespace(in_channels=cin, out_channels=cout, kernel_size=(kh, kw), padding=pad,
                                weight=t["i2h.weight"], bias=t["i2h.bias"], alpha=t["i2h.alpha"],
                                tau_m__dt=t["i2h.tau_m__dt"], alphas=t["i2h.alphas"], tau_s__dt=t["i2h.tau_s__dt"])
    if refractory:                                 # (the reference's plain variant has neither attribute)
        i2h.wrp, i2h.alpharp = m["wrp"], m["alpharp"]
    i2h.get_output_shape = lambda dims: (dims[0] + 2 * pad[0] - kh + 1, dims[1] + 2 * pad[1] - kw + 1)   # reference :368-375

    def init_state(batch, dims, init_value=0):     # reference :377-389 / :468-483
        ch_, cw_ = i2h.get_output_shape(dims)
        st = [torch.zeros(batch, cin, *dims, device=dev), torch.zeros(batch, cin, *dims, device=dev)]
        if refractory:
            st.append(torch.zeros(batch, cout, ch_, cw_, device=dev))
        i2h.state = State(*st)
    i2h.init_state = init_state
    init_state(B + 1, im)                          # wrong batch on purpose: the stub must re-allocate like :488-491
    ch, cw = i2h.get_output_shape(im)
    oh, ow = (ch + 2 * ((pool[0] - 1) // 2) - pool[0]) // pool[0] + 1, (cw + 2 * ((pool[1] - 1) // 2) - pool[1]) // pool[1] + 1
    layer = types.SimpleNamespace(i2h=i2h, im_dims=im, pooling=pool, target_size=24, output_layer=bool(m["output_layer"]),
                                  output_shape=(oh, ow),
                                  i2o=types.SimpleNamespace(weight=t["i2o.weight"], bias=t["i2o.bias"]))
    if m["output_layer"]:
        layer.output_ = types.SimpleNamespace(weight=t["output_.weight"], bias=t["output_.bias"])
    layer.forward = types.MethodType(ns["hip_forward"], layer)

    orc = C.OracleConvLayer(sd, m["im"], m["pad"], m["pool"], m["wrp"], m["alpharp"], m["output_layer"])
    for step in range(3):
        x = g["g1/%s/x%d" % (case, step)]
        out, p, pv, v = layer.forward(cu(x, dev))
        torch.cuda.synchronize()
        oo, op, opv, ov, os_ = orc.forward(x)
        assert i2h.state.eps0.shape[0] == B
        assert bits_equal(i2h.state.eps0.cpu().numpy(), orc.state[0])
        assert bits_equal(i2h.state.eps1.cpu().numpy(), orc.state[1])
        assert bits_equal(v.cpu().numpy(), ov)
        if refractory:
            assert bits_equal(i2h.state.arp.cpu().numpy(), orc.state[2])
        np.testing.assert_allclose(pv.cpu().numpy(), opv, atol=PV_TOL, rtol=0)
        np.testing.assert_allclose(p.cpu().numpy(), op, atol=LOGIT_TOL, rtol=0)
        if m["output_layer"]:
            np.testing.assert_allclose(out.cpu().numpy(), oo, atol=LOGIT_TOL, rtol=0)
        else:
            assert np.array_equal(out.cpu().numpy(), os_)
    # the documented error convention: a negative status + dcll_last_error() -> RuntimeError, nothing thrown across the ABI
    bad = types.SimpleNamespace(**vars(layer))
    bad.i2h = types.SimpleNamespace(**vars(i2h))
    bad.i2h.kernel_size = (0, kw)
    bad.forward = types.MethodType(ns["hip_forward"], bad)
    with pytest.raises(RuntimeError):
        bad.forward(cu(g["g1/%s/x0" % case], dev))


@pytest.mark.parametrize("rows,K,N2,learn", [(512, 8192, 0, False), (512, 8192, 24, True), (37, 8192, 24, False),
                                              (1024, 2048, 0, True), (5, 4096, 24, True)])
def test_step_readouts_fused_tail_equals_the_separate_calls(dev, rows, K, N2, learn):
    """dcll_step_readouts (ABI 4): ONE split-K pass over pv against the stacked i2o / output_ rows + ONE finishing launch ==
    dcll_readout_splitk per readout (bit for bit per column wherever both use the same slice width), k_argmax's first maximum
    (ties included) and dcll_local_loss_grad's gradients — the launches it replaces in every per-step call."""
    from snn_modulation_classification_amd import ops
    rng = np.random.RandomState(rows + K + N2)
    N1 = 24
    pv = cu(rng.uniform(0, 1, size=(rows, K)).astype(np.float32), dev)
    Wt = cu(rng.uniform(-.0055, .0055, size=(N1 + N2, K)).astype(np.float32), dev)
    bias = cu(rng.uniform(-.0055, .0055, size=(N1 + N2,)).astype(np.float32), dev)
    if rows >= 8:                                   # exact ties between readout rows: the first one must win
        Wt[N2 + 3] = Wt[N2 + 11]
        bias[N2 + 3] = bias[N2 + 11]
    target = torch.zeros(rows, N1, device=dev)
    target[torch.arange(rows), torch.from_numpy(rng.randint(0, N1, rows))] = 1
    p = torch.empty(rows, N1, device=dev)
    o = torch.empty(rows, N2, device=dev) if N2 else None
    fin = dict(clout=True)
    if learn:
        fin.update(target=target, kind=ops.LOSS_KINDS['SmoothL1Loss'])
    ops.step_readouts(pv, Wt, bias, N1, N2, p, o, finish=fin)
    assert fin['done']
    # the slice width of the split-K pass depends on (rows, N): up to 512 rows the stacked 48-row pass runs 32 slices, a
    # 24-row pass 64 — there the separate calls add the same products in another order (logit tolerance), else bit-equal
    same = not (rows <= 512 and N2 > 0)
    eq = (lambda a, b: torch.equal(a, b)) if same else (lambda a, b: bool((a - b).abs().max() <= 1e-5))
    ref_p = ops.readout(pv, Wt[:N1].contiguous(), bias[:N1].contiguous())
    assert eq(p, ref_p)
    logits = p
    if N2:
        ref_o = ops.readout(pv, Wt[N1:].contiguous(), bias[N1:].contiguous())
        assert eq(o, ref_o)
        logits = o
    assert torch.equal(fin['clout'], ops.argmax(logits))
    assert torch.equal(fin['clout'].long(), logits.argmax(1))
    if learn:
        g_p, g_o, _, cl = ops.local_loss_grad(p, o if N2 else None, target, ops.LOSS_KINDS['SmoothL1Loss'],
                                              want_loss=False, want_clout=True)
        assert torch.equal(fin['g_p'], g_p) and torch.equal(fin['clout'], cl)
        if N2:
            assert torch.equal(fin['g_o'], g_o)
    else:
        assert fin['g_p'] is None


@pytest.mark.parametrize("learn", [False, True])
@pytest.mark.parametrize("shapes", [[(512, 8192, 0), (512, 8192, 0), (512, 8192, 24)],         # the slices of radio_ml_conv.yaml
                                    [(37, 8192, 24), (300, 2048, 0)], [(5, 4096, 24), (1024, 8192, 0), (70, 8192, 0), (512, 2048, 24)]])
def test_step_readouts_multi_equals_the_per_layer_calls(dev, shapes, learn):
    """dcll_step_readouts_multi (ABI 6): the readout tails of several layer steps in TWO launches (k_readout_t16m +
    k_step_readout_finish_m; asserted from the launch log) == dcll_step_readouts per item, bit for bit: p, o, the recorded
    argmax, the local-loss gradients — items of different row counts, K and stacked widths in one call.  Argument checks:
    a refused call launches nothing."""
    import ctypes
    from snn_modulation_classification_amd import ops, _lib
    rng = np.random.RandomState(len(shapes) + 7 * learn)
    N1 = 24
    single, multi, calls = [], [], []
    for rows, K, N2 in shapes:
        pv = cu(rng.uniform(0, 1, size=(rows, K)).astype(np.float32), dev)
        Wt = cu(rng.uniform(-.0055, .0055, size=(N1 + N2, K)).astype(np.float32), dev)
        bias = cu(rng.uniform(-.0055, .0055, size=(N1 + N2,)).astype(np.float32), dev)
        target = torch.zeros(rows, N1, device=dev)
        target[torch.arange(rows), torch.from_numpy(rng.randint(0, N1, rows))] = 1

        def fin():
            f = dict(clout=True)
            if learn:
                f.update(target=target, kind=ops.LOSS_KINDS['SmoothL1Loss'])
            return f
        mk = lambda: (torch.full((rows, N1), 7.0, device=dev), torch.full((rows, N2), 7.0, device=dev) if N2 else None)
        p1, o1 = mk()
        f1 = fin()
        ops.step_readouts(pv, Wt, bias, N1, N2, p1, o1, scratch={}, finish=f1)
        single.append((p1, o1, f1))
        p2, o2 = mk()
        f2 = fin()
        f2['run_readouts'] = lambda: (_ for _ in ()).throw(AssertionError("the per-layer call ran"))
        f2['ro_call'] = (pv, Wt, bias, N1, N2, p2, o2, {})
        multi.append((p2, o2, f2))
    with ops.kernel_trace() as tr:
        ops.run_deferred_readouts([f for _, _, f in multi])
    assert tr.names == ["k_readout_t16m", "k_step_readout_finish_m"], tr.names
    for (p1, o1, f1), (p2, o2, f2) in zip(single, multi):
        assert f2['done'] and 'run_readouts' not in f2 and 'ro_call' not in f2
        assert torch.equal(p1, p2) and (o1 is None or torch.equal(o1, o2))
        assert torch.equal(f1['clout'], f2['clout'])
        if learn:
            assert torch.equal(f1['g_p'], f2['g_p']) and (o1 is None or torch.equal(f1['g_o'], f2['g_o']))
        else:
            assert f2['g_p'] is None
    # refused: more than 8 items, a non-zero reserved field, items with and without a target mixed, shared scratch
    lib = _lib.get()
    it, keep, _ = ops._step_readouts_prepare(cu(rng.uniform(0, 1, size=(8, 2048)).astype(np.float32), dev),
                                             cu(np.zeros((N1, 2048), np.float32), dev), cu(np.zeros(N1, np.float32), dev), N1, 0,
                                             torch.empty(8, N1, device=dev), None, {}, finish=None)
    arr = (_lib.StepRo * 9)(*[it] * 9)
    assert lib.dcll_step_readouts_multi(arr, 9, None) == -1 and b"1 .. 8" in lib.dcll_last_error()
    assert lib.dcll_step_readouts_multi(arr, 2, None) == -1 and b"share scratch" in lib.dcll_last_error()
    bad = _lib.StepRo.from_buffer_copy(it)
    bad.reserved = 1
    assert lib.dcll_step_readouts_multi((_lib.StepRo * 1)(bad), 1, None) == -1 and b"reserved" in lib.dcll_last_error()
    assert lib.dcll_step_readouts_multi(None, 0, None) == 0


@pytest.mark.parametrize("B,L,n", [(4096, 3, 24), (37, 1, 24), (1, 2, 10), (5000, 7, 24)])
def test_vote_tallies_equal_the_torch_construction(dev, B, L, n):
    """dcll_vote_tallies (one launch; parallel.tallies on device tensors) == the torch construction it replaced: per layer
    the confusion matrix [pred][label] via bincount, the correct count, the vote count — the form the ranks all-reduce."""
    from snn_modulation_classification_amd import ops, parallel
    g = torch.Generator().manual_seed(B + L)
    votes = [torch.randint(0, n, (B,), generator=g, dtype=torch.int32).to(dev) for _ in range(L)]
    labels = torch.randint(0, n, (B,), generator=g).to(dev)
    got = parallel.tallies(votes, labels, n)
    assert got.dtype == torch.int64 and tuple(got.shape) == (L, n * n + 2)
    want = parallel.tallies([v.cpu() for v in votes], labels.cpu(), n)           # CPU tensors: the torch construction
    assert torch.equal(got.cpu(), want)
    cm, acc = parallel.split_tallies(got, n)
    assert int(cm.sum()) == L * B and all(int(got[l, -1]) == B for l in range(L))
    assert torch.equal(ops.vote_tallies(votes, labels, n), got)


@pytest.mark.parametrize("B", [161, 255, 257, 511, 513, 1023, 1024, 1025])
def test_weight_gradient_chunking_on_ragged_batches(dev, B):
    """k_bwd_wgrad_c32 around the batch sizes where its chunking changes (round 6: 161 .. 1024 samples run 128 batch chunks x 2
    column halves, ragged when 128 does not divide B; above, 256 chunks x 1): dW / db of a 32 -> 32 layer step from a given dL/dv
    against a float64 reference on the device (unfold + matmul) — a mis-indexed or dropped job would be an error of order 1/B."""
    from snn_modulation_classification_amd import ops
    g = torch.Generator(device="cpu").manual_seed(B)
    eps1 = torch.rand(B, 32, 16, 16, generator=g).to(dev)
    gv = (torch.randn(B, 32, 16, 16, generator=g) / B).to(dev)
    d = ops.make_conv_desc(32, 32, (16, 16), 7, 3, 1, 0, False, True, 1.0)
    dW, db, _, _ = ops.conv_lif_backward(d, eps1, torch.zeros_like(gv), None, None, None, None, gv, None, want_out=False)
    cols = torch.nn.functional.unfold(eps1.double(), 7, padding=3)                       # (B, 32*49, 256)
    ref = torch.einsum("bop,bkp->ok", gv.double().reshape(B, 32, 256), cols).reshape(32, 32, 7, 7)
    np.testing.assert_allclose(dW.cpu().numpy(), ref.cpu().numpy(), rtol=2e-4, atol=2e-5 * float(ref.abs().max()))
    np.testing.assert_allclose(db.cpu().numpy(), gv.double().sum((0, 2, 3)).cpu().numpy(), rtol=2e-4, atol=1e-6)
