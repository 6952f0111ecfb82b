"""Layers built with the constructor options ConvNetwork never uses — stride / dilation / groups other than 1, bias=False, an
activation other than nn.Sigmoid(), spiking=False (reference dcll/pytorch_libdcll.py:299-313, :75-104, :407-426, :485-509,
:599-608) — against fixture G1x, generated from the imported reference (tests/golden/make_golden.py --only-g1x).  The product
layers run their GENERAL step (dcll/pytorch_libdcll.py: _is_sigmoid): dcll_conv_lif_step / dcll_dense_lif_step for the traces,
the convolution in the pinned order (generic kernels: stride / dilation / groups, NULL bias), the refractory trace and the
threshold; act, pooling and readouts as torch ops on the device; gradients through dcll_conv_lif_backward /
dcll_dense_lif_backward fed with dL/dv."""
import numpy as np
import pytest
import torch

from conftest import G1X_CASES, G1X_LEARN_CASES, g1x_cfg

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-4      # BASELINE.json north_star: class logits within 1e-4 fp32


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def bits_equal(a, b):
    return np.array_equal(np.asarray(a, dtype=np.float32).view(np.uint32), np.asarray(b, dtype=np.float32).view(np.uint32))


def _build(c, dev):
    """The product's layer for a G1x cfg (time constants scalar here: the fixture's tensors are installed by _install)."""
    from snn_modulation_classification_amd.dcll import pytorch_libdcll as L
    act = c.act_module()
    if c.kind == 0:
        layer = L.Conv2dDCLLlayer(c.cin, c.cout, kernel_size=(c.kh, c.kw), padding=(c.pad_h, c.pad_w), pooling=(c.pool_h, c.pool_w),
                                  im_dims=(c.H, c.W), target_size=7, stride=c.stride, dilation=c.dilation, alpha=.92, alphas=.85,
                                  alpharp=.65, wrp=c.wrp, act=act, lc_ampl=.5, random_tau=False, spiking=bool(c.spiking),
                                  lc_dropout=False, output_layer=bool(c.output_layer))
        return layer.to(dev), layer.i2h
    if c.kind == 1:
        kw = dict(stride=c.stride, padding=(c.pad_h, c.pad_w), dilation=c.dilation, groups=c.groups, bias=bool(c.bias), alpha=.92,
                  alphas=.85, act=act, random_tau=False)
        i2h = (L.ContinuousRelativeRefractoryConv2D(c.cin, c.cout, (c.kh, c.kw), alpharp=.65, wrp=c.wrp, **kw) if c.wrp > 0 else
               L.ContinuousConv2D(c.cin, c.cout, (c.kh, c.kw), spiking=bool(c.spiking), **kw))
        return i2h.to(dev), i2h
    layer = L.DenseDCLLlayer(c.cin, c.cout, target_size=7, bias=bool(c.bias), alpha=.9, alphas=.85, alpharp=.65, wrp=c.wrp, act=act,
                             spiking=bool(c.spiking), random_tau=False)
    return layer.to(dev), layer.i2h


def _install(mod, sd, dev):
    """The fixture's state dict into the module, tensor by tensor (the time constants of a random_tau layer are (C,H,W) / (C)
    tensors where a freshly built layer holds scalars: those Parameters are replaced, like the reference's randomize_tau does)."""
    own = dict(mod.named_parameters())
    assert set(own) == set(sd), (sorted(own), sorted(sd))
    with torch.no_grad():
        for name, arr in sd.items():
            t = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
            p = own[name]
            if tuple(p.shape) == tuple(t.shape):
                p.copy_(t)
            else:
                owner = mod
                *path, leaf = name.split(".")
                for a in path:
                    owner = getattr(owner, a)
                setattr(owner, leaf, torch.nn.Parameter(t, requires_grad=p.requires_grad))


@pytest.mark.parametrize("case", G1X_CASES)
def test_layer_options_reproduce_the_reference(golden, dev, case):
    """Three consecutive steps of every G1x case: traces bit for bit, the reference's spikes (the CPU suite shows these
    fixtures free of band-internal flips: tests/test_oracle_c.py), v inside the rounding band of the two summation orders,
    pv / pvoutput / output within the logit tolerance (non-spiking layers: `output` IS the membrane value) — and v, spikes,
    refractory state bit for bit against the pinned-order C oracle with the same options."""
    from oracle import c_oracle as C
    g = golden("g1x_layer_options.npz")
    c = g1x_cfg(g, case)
    pre = "g1x/%s/" % case
    sd = g.sub(pre + "sd/")
    mod, i2h = _build(c, dev)
    _install(mod, sd, dev)
    assert i2h._general()
    if c.kind == 1:
        i2h.init_state(c.B, (c.H, c.W))
    else:
        mod.init_hiddens(c.B)
    orc = None
    if c.kind != 2:
        osd = dict(sd) if c.kind == 0 else {"i2h." + k: v for k, v in sd.items()}
        ch = (c.H + 2 * c.pad_h - c.dilation * (c.kh - 1) - 1) // c.stride + 1
        cw = (c.W + 2 * c.pad_w - c.dilation * (c.kw - 1) - 1) // c.stride + 1
        # (the oracle's own readouts are not what is checked here: a one-row dummy over the un-pooled map)
        osd["i2o.weight"], osd["i2o.bias"] = np.zeros((1, c.cout * ch * cw), np.float32), np.zeros((1,), np.float32)
        orc = C.OracleConvLayer(osd, (c.H, c.W), (c.pad_h, c.pad_w), (1, 1), c.wrp, .65, False, c.stride, c.dilation, c.groups)
        orc.init_state(c.B)
    for t in range(3):
        x = g[pre + "x%d" % t]
        with torch.no_grad():
            out = mod.forward(torch.from_numpy(x).to(dev))
        names = ("o", "pv", "v") if c.kind == 1 else ("o", "p", "pv", "v")
        got = {n: o_.detach().cpu().numpy() for n, o_ in zip(names, out)}
        e = lambda n: g[pre + "%s%d" % (n, t)]
        for i, nm in enumerate(i2h.state._fields[:2]):
            assert bits_equal(i2h.state[i].cpu().numpy(), e("out_" + nm)), (case, t, nm)
        v, vr = got["v"], e("v")
        assert v.shape == vr.shape
        scale = float(np.abs(vr).max()) + 1e-30
        assert np.abs(v - vr).max() <= 2e-5 * scale, (case, t, np.abs(v - vr).max(), scale)
        assert np.array_equal(v > 0, vr > 0), (case, t, "spike pattern")
        if c.spiking and not (c.kind == 0 and c.output_layer):
            assert np.array_equal(got["o"], e("o")), (case, t, "output spikes")
        else:
            np.testing.assert_allclose(got["o"], e("o"), atol=max(LOGIT_TOL, 2e-5 * scale), rtol=0)
        np.testing.assert_allclose(got["pv"], e("pv"), atol=2e-5 * max(1.0, scale), rtol=1e-5)
        if "p" in got:
            np.testing.assert_allclose(got["p"], e("p"), atol=LOGIT_TOL, rtol=0)
        if c.wrp > 0:
            np.testing.assert_allclose(i2h.state.arp.cpu().numpy(), e("out_arp"), atol=2e-5 * scale, rtol=0)
        if orc is not None:
            _, _, _, ov, os_ = orc.forward(x)
            assert bits_equal(v, ov), (case, t, "v vs the pinned-order oracle", np.abs(v - ov).max())
            assert bits_equal(i2h.state.eps1.cpu().numpy(), orc.state[1])
            if c.wrp > 0:
                assert bits_equal(i2h.state.arp.cpu().numpy(), orc.state[2])


@pytest.mark.parametrize("case", G1X_LEARN_CASES)
def test_learning_with_layer_options_matches_reference_train_steps(golden, dev, case):
    """DCLLBase.train_dcll (reference :690-718) on slices whose layer takes the general step: two steps with burn-in 2 — the
    second one learns — SmoothL1Loss + Adam(betas (0, .95), weight_decay 10) [+ optimizer2 on an output layer]: outputs of both
    steps, the loss value, the gradients of every trainable tensor and the parameters after the step against the reference's."""
    from snn_modulation_classification_amd.dcll import pytorch_libdcll as L
    g = golden("g1x_layer_options.npz")
    c = g1x_cfg(g, case)
    pre = "g1x/%s/" % case
    layer, i2h = _build(c, dev)
    _install(layer, g.sub(pre + "sd/"), dev)
    sl = L.DCLLClassification(dclllayer=layer, name="g1x", batch_size=c.B, loss=torch.nn.SmoothL1Loss, optimizer=torch.optim.Adam,
                              kwargs_optimizer={"lr": 1e-6, "betas": [0.0, .95], "weight_decay": 10.0}, burnin=2)
    _install(layer, g.sub(pre + "sd/"), dev)        # (DCLLBase.init re-ran init_hiddens)
    assert sl._native_learning() is None            # the autograd path around the HIP forward / backward
    tgt = torch.from_numpy(g[pre + "target"]).to(dev)
    sl.train()
    for t in range(2):
        x = torch.from_numpy(g[pre + "x%d" % t]).to(dev)
        o, p, pv, v, loss = sl.train_dcll(x, tgt, regularize=False)
        vr = g[pre + "v%d" % t]
        scale = float(np.abs(vr).max()) + 1e-30
        assert np.abs(v.detach().cpu().numpy() - vr).max() <= 2e-5 * scale and np.array_equal(v.detach().cpu().numpy() > 0, vr > 0)
        np.testing.assert_allclose(p.detach().cpu().numpy(), g[pre + "p%d" % t], atol=LOGIT_TOL, rtol=0)
        np.testing.assert_allclose(o.detach().cpu().numpy(), g[pre + "o%d" % t], atol=max(LOGIT_TOL, 2e-5 * scale), rtol=0)
        np.testing.assert_allclose(float(loss.reshape(-1)[0]), float(g[pre + "loss%d" % t][0]), rtol=1e-4, atol=1e-7)
    grads = g.sub(pre + "grad/")
    own = dict(layer.named_parameters())
    assert set(grads) == {n for n, q in own.items() if q.grad is not None}, (sorted(grads), [n for n, q in own.items() if q.grad is not None])
    for name, ref in grads.items():
        got = own[name].grad.cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-5 * float(np.abs(ref).max()) + 1e-12, err_msg=name)
    for name, ref in g.sub(pre + "sd1/").items():
        np.testing.assert_allclose(own[name].detach().cpu().numpy(), ref, rtol=2e-3, atol=2e-3 * float(np.abs(ref).max()) + 1e-12,
                                   err_msg=name)


def test_options_through_the_c_abi_vs_oracle(dev):
    """dcll_conv_lif_step / dcll_conv_lif_backward with stride, dilation and groups other than 1 and a NULL bias on random
    layers (odd sizes, pooling, refractory or not): v / spikes / state bit for bit against the pinned-order C oracle, the
    weight gradient against torch autograd through F.conv2d."""
    from snn_modulation_classification_amd import ops
    from oracle import c_oracle as C
    rng = np.random.RandomState(3)
    for (cin, cout, k, pad, stride, dil, groups, hw, pool, wrp, bias, B) in [
            (6, 9, 3, 1, 2, 1, 3, (13, 11), 1, 1.0, True, 3), (4, 4, 5, 4, 1, 2, 2, (12, 15), 2, 0.0, False, 2),
            (8, 8, 3, 2, 3, 2, 8, (17, 17), 1, 0.5, True, 2), (2, 6, (1, 3), (0, 1), 2, 1, 1, (4, 33), (1, 2), 0.0, False, 5)]:
        kh, kw = (k, k) if isinstance(k, int) else k
        cig = cin // groups
        W = (rng.randn(cout, cig, kh, kw) * 0.3).astype(np.float32)
        b = (rng.randn(cout) * 0.1).astype(np.float32) if bias else None
        alpha = rng.uniform(.8, .97, size=(cin,) + hw).astype(np.float32)
        alphas = rng.uniform(.8, .9, size=(cin,) + hw).astype(np.float32)
        tau_m = (np.float32(1) / (np.float32(1) - alpha)).astype(np.float32)
        tau_s = (np.float32(1) / (np.float32(1) - alphas)).astype(np.float32)
        d = ops.make_conv_desc(cin, cout, hw, (kh, kw), pad, pool, 0, False, True, wrp, .65, stride, dil, groups)
        ch, cw, ph, pw = ops.conv_out_shape(d)
        sd = {"i2h.weight": W, "i2h.alpha": alpha, "i2h.tau_m__dt": tau_m, "i2h.alphas": alphas, "i2h.tau_s__dt": tau_s,
              "i2o.weight": np.zeros((1, cout * ph * pw), np.float32), "i2o.bias": np.zeros((1,), np.float32)}
        if bias:
            sd["i2h.bias"] = b
        orc = C.OracleConvLayer(sd, hw, pad, pool, wrp, .65, False, stride, dil, groups)
        assert (orc.ch, orc.cw, orc.ph, orc.pw) == (ch, cw, ph, pw)
        orc.init_state(B)
        t_ = lambda a: None if a is None else torch.from_numpy(a).to(dev)
        eps0 = torch.zeros((B, cin) + hw, device=dev)
        eps1 = torch.zeros_like(eps0)
        arp = torch.zeros((B, cout, ch, cw), device=dev)
        for step in range(3):
            x = (rng.uniform(size=(B, cin) + hw) < 0.3).astype(np.float32)
            s, _, _, pv, v = ops.conv_lif_step(d, t_(x), t_(W), t_(b), t_(alpha), t_(tau_m), t_(alphas), t_(tau_s), eps0, eps1,
                                               arp if wrp > 0 else None)
            _, _, opv, ov, os_ = orc.forward(x)
            assert bits_equal(v.cpu().numpy(), ov), (cin, cout, stride, dil, groups, step)
            assert np.array_equal(s.cpu().numpy(), os_)
            assert bits_equal(eps1.cpu().numpy(), orc.state[1])
            if wrp > 0:
                assert bits_equal(arp.cpu().numpy(), orc.state[2])
            np.testing.assert_allclose(pv.cpu().numpy(), opv, atol=2e-6, rtol=0)
        # weight gradient of dL/dv = r through the generic kernel vs torch autograd (float64 reference)
        r = rng.randn(B, cout, ch, cw).astype(np.float32)
        Wt = torch.from_numpy(W).double().requires_grad_(True)
        bt = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
        vv = torch.nn.functional.conv2d(eps1.cpu().double(), Wt, bt, stride, (pad, pad) if isinstance(pad, int) else pad, dil, groups)
        (vv * torch.from_numpy(r).double()).sum().backward()
        dW, db, _, _ = ops.conv_lif_backward(d, eps1, v, None, None, None, None, t_(r), None, want_out=False)
        np.testing.assert_allclose(dW.cpu().numpy(), Wt.grad.numpy(), rtol=1e-4, atol=1e-5 * float(Wt.grad.abs().max()))
        np.testing.assert_allclose(db.cpu().numpy(), bt.grad.numpy(), rtol=1e-4, atol=1e-5 * float(bt.grad.abs().max()))


def test_int8_weights_with_groups_and_stride_equal_the_dequantised_run(dev):
    """dcll_layer_opts.w_q8 (int8 conv weights + per-output-channel scale) on a grouped, strided, dilated layer — the generic kernel
    converts a weight once, (float)q * scale[co]: v, spikes and state bit-identical to the call on the dequantised fp32 tensor."""
    from snn_modulation_classification_amd import ops, quant
    rng = np.random.RandomState(8)
    cin, cout, groups, hw, B = 6, 9, 3, (12, 10), 3
    W = torch.from_numpy((rng.randn(cout, cin // groups, 3, 3) * 0.3).astype(np.float32)).to(dev)
    q, scale = quant.quantize_int8_per_channel(W)
    Wd = quant.dequantize(q, scale).contiguous()
    b = torch.from_numpy((rng.randn(cout) * 0.1).astype(np.float32)).to(dev)
    tau = [torch.full((1,), v, device=dev) for v in (.92, 1. / (1 - .92), .85, 1. / (1 - .85))]
    d = ops.make_conv_desc(cin, cout, hw, 3, 1, 1, 0, False, False, 1.0, .65, 2, 2, groups)
    ch, cw, _, _ = ops.conv_out_shape(d)
    runs = []
    for q8 in (None, (q, scale)):
        eps0 = torch.zeros((B, cin) + hw, device=dev)
        eps1, arp = torch.zeros_like(eps0), torch.zeros((B, cout, ch, cw), device=dev)
        outs = []
        r2 = np.random.RandomState(9)
        for _ in range(3):
            x = torch.from_numpy((r2.uniform(size=(B, cin) + hw) < 0.3).astype(np.float32)).to(dev)
            s, _, _, pv, v = ops.conv_lif_step(d, x, Wd, b, *tau, eps0, eps1, arp, q8=q8)
            outs.append((s.clone(), v.clone()))
        runs.append((outs, eps1.clone(), arp.clone()))
    for (s0, v0), (s1, v1) in zip(runs[0][0], runs[1][0]):
        assert torch.equal(s0, s1) and bits_equal(v0.cpu().numpy(), v1.cpu().numpy())
    assert torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])
    assert float(runs[0][0][-1][0].mean()) not in (0.0, 1.0)       # (both spike values occur: not vacuous)


def test_dense_forward_sequence_of_a_general_layer_equals_the_step_loop(dev):
    """DenseDCLLlayer.forward_sequence on a layer outside the fused kernels (act = Tanh, bias=False): the step loop, stacked."""
    from snn_modulation_classification_amd.dcll import pytorch_libdcll as L
    torch.manual_seed(3)
    mk = lambda: L.DenseDCLLlayer(40, 24, target_size=7, bias=False, alpha=.9, alphas=.85, wrp=1.0, act=torch.nn.Tanh()).to(dev)
    a, b = mk(), mk()
    b.load_state_dict(a.state_dict())
    with torch.no_grad():
        a.i2h.weight.mul_(200.0)
        b.i2h.weight.mul_(200.0)
    x = (torch.rand(6, 5, 40, device=dev) < 0.2).float()
    a.init_hiddens(5)
    b.init_hiddens(5)
    with torch.no_grad():
        seq = a.forward_sequence(x, want_v=True)
        steps = [b.forward(x[t]) for t in range(6)]
    for k in range(4):
        assert torch.equal(seq[k], torch.stack([s_[k] for s_ in steps])), k
    assert 0.0 < float(seq[0].mean()) < 1.0
