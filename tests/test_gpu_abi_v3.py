"""ABI v3 (`dcll_layer_opts`, include/dcll_hip.h): int8 conv weights handed across the C ABI, and pv written before the
sigmoid with the sigmoid applied by the readout GEMM.

BASELINE config 5 ("int8 weights + 1-bit packed spikes") has no reference code (parity unpinned); what can
be pinned, and is here: a call that reads the int8 tensor + per-channel scales is bit-identical — membrane v, spikes,
final state — to the same call on the dequantised fp32 tensor, for EVERY kernel family behind the layer calls, and both
equal the C oracle on the dequantised weights.  pv_presigmoid changes where a (not bit-pinned) sigmoid runs, nothing
else: v in the buffer is the bit-exact v, the logits stay inside the 1e-4 contract, the pv statistics are unchanged."""
import numpy as np
import pytest
import torch

from test_gpu_kernels import LOGIT_TOL, PV_TOL, _rand_layer, _sd_from, bits_equal, cu, dev  # noqa: F401

pytestmark = pytest.mark.gpu


def _quant(W):
    """per-output-channel symmetric int8, as quant.py defines it -> (q int8, scale fp32, dequantised fp32)"""
    from snn_modulation_classification_amd import quant
    q, scale = quant.quantize_int8_per_channel(torch.from_numpy(W))
    return q.numpy(), scale.numpy(), quant.dequantize(q, scale).numpy()


def _state(orc, rng, zero):
    if not zero:
        orc.state[0][...] = rng.uniform(0, 5, size=orc.state[0].shape)
        orc.state[1][...] = rng.uniform(0, 50, size=orc.state[1].shape)
        if len(orc.state) > 2 and orc.state[2] is not None:
            orc.state[2][...] = -rng.uniform(0, 2, size=orc.state[2].shape)


# (name, c_in, c_out, plane, kernel, padding, pooling, T, B): one case per kernel family behind the sequence calls
SEQ_CASES = [
    ("c32d", 32, 32, (16, 16), 7, 3, 1, 11, 3),          # k_lif_seq_c32d (T >= 8)
    ("c32", 32, 32, (16, 16), 7, 3, 1, 5, 2),            # k_lif_seq_c32 (short sequence)
    ("c32t", 32, 32, (16, 64), 7, 3, 1, 6, 2),           # k_lif_seq_c32t (tiled planes)
    ("c1", 1, 32, (16, 16), 7, 3, 1, 9, 3),              # k_lif_seq_c1
    ("c1_narrow", 1, 8, (16, 16), 7, 3, 1, 6, 2),        # k_lif_seq_c1, generic (guarded) epilogue
    ("c1t", 1, 32, (32, 32), 7, 3, 1, 5, 2),             # k_lif_seq_c1t
    ("w3_wide", 64, 64, (16, 64), (1, 3), (0, 1), (1, 2), 6, 2),     # k_lif_seq_w3<64, WIDE>
    ("w3_mid", 64, 64, (16, 8), (1, 3), (0, 1), (1, 2), 6, 5),       # k_lif_seq_w3<64, !WIDE>, 4 rows per tile
    ("w3_narrow", 64, 64, (16, 4), (1, 3), (0, 1), (1, 2), 5, 3),    # k_lif_seq_w3<64, !WIDE>
    ("w3_first", 1, 64, (16, 128), (1, 3), (0, 1), (1, 2), 7, 2),    # k_lif_seq_w3<1>
]


@pytest.mark.parametrize("wrp", [1.0, 0.0])
@pytest.mark.parametrize("case", SEQ_CASES, ids=[c[0] for c in SEQ_CASES])
def test_int8_abi_sequence_kernels_equal_dequantised_fp32_and_oracle(dev, case, wrp):
    """Every sequence kernel with int8 weights + scales through dcll_layer_opts == the same call on the dequantised fp32
    tensor == the C oracle on the dequantised weights: v, packed spikes and the final state, bit for bit."""
    from snn_modulation_classification_amd import ops
    from oracle import c_oracle as C
    name, cin, cout, hw, ks, pad, pool, T, B = case
    rng = np.random.RandomState(5 + len(name))
    H, Wd = hw
    kh, kw = (ks, ks) if isinstance(ks, int) else ks
    n = cin * kh * kw
    stdv = 1.0 / np.sqrt(n) / 250
    W = (rng.uniform(-stdv * 1e-2, stdv * 1e-2, size=(cout, cin, kh, kw)) * 3.0).astype(np.float32)
    b = rng.uniform(-stdv, stdv, size=(cout,)).astype(np.float32)
    _, _, alpha, tau_m, alphas, tau_s = _rand_layer(rng, cin, cout)
    q, scale, Wd_ = _quant(W)
    assert np.abs(Wd_ - W).max() <= scale.max() * 0.5 * 1.0001 and not np.array_equal(Wd_, W)
    ph, pw = (H, Wd) if pool == 1 else (H, Wd // 2)
    sd = _sd_from(Wd_, b, alpha, tau_m, alphas, tau_s, hw, rng=rng)
    sd["i2o.weight"] = rng.uniform(-.005, .005, size=(24, cout * ph * pw)).astype(np.float32)
    orc = C.OracleConvLayer(sd, hw, pad, pool, wrp)
    orc.init_state(B)
    _state(orc, rng, zero=False)
    init = [None if s_ is None else s_.copy() for s_ in orc.state]
    d = ops.make_conv_desc(cin, cout, hw, ks, pad, pool, 24, False, True, wrp)
    tau4 = cu(np.stack([alpha, tau_m, alphas, tau_s]), dev)
    if cin == 1:
        cells = rng.randint(0, H * Wd, size=(T, B)).astype(np.int32)
        x = np.zeros((T, B, 1, H * Wd), np.float32)
        x[np.arange(T)[:, None], np.arange(B)[None, :], 0, cells] = 1
    else:
        x = (rng.uniform(size=(T, B, cin, H * Wd)) < 0.1).astype(np.float32)
        x[0] = rng.uniform(size=(B, cin, H * Wd)) < 0.5
    want_spk = pool == 1 or (H * Wd) % 64 == 0

    def run(Wt, q8):
        st = [None if s_ is None else cu(s_.copy(), dev) for s_ in init]
        arp = st[2] if wrp > 0 else None
        if cin == 1:
            spk, pv, v = ops.conv_lif_sequence_cells(d, cu(cells, dev), Wt, cu(b, dev), tau4, st[0], st[1], arp, T, B,
                                                     want_spikes=want_spk, want_v=True, q8=q8)
        else:
            spk, pv, v = ops.conv_lif_sequence(d, ops.pack_spikes(cu(x, dev)), Wt, cu(b, dev), tau4, st[0], st[1], arp,
                                               T, B, want_spikes=want_spk, want_v=True, q8=q8)
        torch.cuda.synchronize()
        return (None if spk is None else spk.cpu().numpy(), pv.cpu().numpy(), v.cpu().numpy(),
                [None if s_ is None else s_.cpu().numpy() for s_ in st])

    f_spk, f_pv, f_v, f_st = run(cu(Wd_, dev), None)                             # fp32 call on the dequantised tensor
    q_spk, q_pv, q_v, q_st = run(None, (cu(q, dev), cu(scale, dev)))             # int8 tensor across the ABI, W = NULL
    assert bits_equal(q_v, f_v), np.argwhere(q_v != f_v)[:5]
    assert bits_equal(q_pv, f_pv)
    if want_spk:
        assert np.array_equal(q_spk, f_spk)
    for a, c in zip(q_st, f_st):
        if a is not None:
            assert bits_equal(a, c)
    nspk = 0
    for t in range(T):
        _, _, opv, ov, os_ = orc.forward(x[t].reshape(B, cin, H, Wd))
        assert bits_equal(q_v[t], ov), (t, np.abs(q_v[t] - ov).max())
        np.testing.assert_allclose(q_pv[t], opv, atol=PV_TOL, rtol=0)
        nspk += os_.sum()
    for a, c in zip(q_st, orc.state):
        if a is not None and c is not None:
            assert bits_equal(a, c)
    assert 0.002 < nspk / (T * B * cout * ph * pw) < 0.95, "degenerate test"
    # the fp32 ORIGINAL weights give a different membrane: the comparison above is not vacuous
    o_v = run(cu(W, dev), None)[2]
    assert not bits_equal(o_v, q_v)


# (name, c_in, c_out, plane, kernel, padding, pooling, B): one case per kernel family behind dcll_conv_lif_step
STEP_CASES = [
    ("step_c32", 32, 32, (16, 16), 7, 3, 1, 300),        # k_lif_step_c32 (B > 256: one workgroup per sample)
    ("step_c32_split", 32, 32, (16, 16), 7, 3, 1, 5),    # k_trace4 + k_lif_step_c32t<8>
    ("step_c32t", 32, 32, (32, 48), 7, 3, 1, 2),         # k_lif_step_c32t<16>
    ("step_c1", 1, 32, (16, 16), 7, 3, 1, 4),            # k_lif_step_c1
    ("step_c1t", 1, 16, (32, 32), 7, 3, 1, 2),           # k_lif_step_c1 (tiled)
    ("step_tiled_1x3", 64, 64, (16, 32), (1, 3), (0, 1), (1, 2), 2),     # k_conv_lif_tiled<1,3> + k_pool
    ("step_tiled_5x5", 3, 12, (28, 28), 5, 2, 2, 2),     # k_conv_lif_tiled<5,5> (mnist geometry)
    ("step_generic", 3, 5, (9, 11), (2, 4), (1, 2), 1, 2),               # k_conv_lif (any geometry)
]


@pytest.mark.parametrize("wrp", [1.0, 0.0])
@pytest.mark.parametrize("case", STEP_CASES, ids=[c[0] for c in STEP_CASES])
def test_int8_abi_step_kernels_equal_dequantised_fp32_and_oracle(dev, case, wrp):
    """dcll_conv_lif_step with int8 weights through dcll_layer_opts, three free-running steps from a non-zero state ==
    the fp32 call on the dequantised tensor == the C oracle: s, v and the state bit for bit."""
    from snn_modulation_classification_amd import ops
    from oracle import c_oracle as C
    name, cin, cout, hw, ks, pad, pool, B = case
    rng = np.random.RandomState(17 + len(name))
    H, Wd = hw
    kh, kw = (ks, ks) if isinstance(ks, int) else ks
    n = cin * kh * kw
    stdv = 1.0 / np.sqrt(n) / 250
    W = (rng.uniform(-stdv * 1e-2, stdv * 1e-2, size=(cout, cin, kh, kw)) * 3.0).astype(np.float32)
    b = rng.uniform(-stdv, stdv, size=(cout,)).astype(np.float32)
    _, _, alpha, tau_m, alphas, tau_s = _rand_layer(rng, cin, cout)
    q, scale, Wd_ = _quant(W)
    d = ops.make_conv_desc(cin, cout, hw, ks, pad, pool, 24, False, True, wrp)
    ch, cw, ph, pw = ops.conv_out_shape(d)
    sd = _sd_from(Wd_, b, alpha, tau_m, alphas, tau_s, hw, rng=rng)
    sd["i2o.weight"] = rng.uniform(-.005, .005, size=(24, cout * ph * pw)).astype(np.float32)
    orc = C.OracleConvLayer(sd, hw, pad, pool, wrp)
    orc.init_state(B)
    _state(orc, rng, zero=False)
    init = [None if s_ is None else s_.copy() for s_ in orc.state]
    taus = [cu(sd[k], dev) for k in ("i2h.alpha", "i2h.tau_m__dt", "i2h.alphas", "i2h.tau_s__dt")]
    xs = [(rng.uniform(size=(B, cin, H, Wd)) < 0.15).astype(np.float32) for _ in range(3)]
    i2o_W, i2o_b = cu(sd["i2o.weight"], dev), cu(sd["i2o.bias"], dev)

    def run(Wt, q8):
        st = [None if s_ is None else cu(s_.copy(), dev) for s_ in init]
        outs = []
        for x in xs:
            s, p, o, pv, v = ops.conv_lif_step(d, cu(x, dev), Wt, cu(b, dev), *taus, st[0], st[1],
                                               st[2] if wrp > 0 else None, i2o_W, i2o_b, q8=q8)
            outs.append([t.cpu().numpy() for t in (s, p, pv, v)])
        return outs, [None if s_ is None else s_.cpu().numpy() for s_ in st]

    f_out, f_st = run(cu(Wd_, dev), None)
    q_out, q_st = run(None, (cu(q, dev), cu(scale, dev)))
    for k, x in enumerate(xs):
        for a, c in zip(q_out[k], f_out[k]):
            assert bits_equal(a, c), (k, np.abs(a - c).max())
        oo, op, opv, ov, os_ = orc.forward(x)
        assert bits_equal(q_out[k][3], ov), (k, np.abs(q_out[k][3] - ov).max())
        assert np.array_equal(q_out[k][0], os_)
        np.testing.assert_allclose(q_out[k][1], op, atol=LOGIT_TOL, rtol=0)
    for a, c, o_ in zip(q_st, f_st, orc.state):
        if a is not None:
            assert bits_equal(a, c) and bits_equal(a, o_)


def test_layer_opts_validation(dev):
    """dcll_layer_opts is checked before any launch: int8 weights without scales, a reserved field that is not 0,
    pv_presigmoid on the per-step drop-in (which returns pv by contract), and neither weight form."""
    import ctypes
    from snn_modulation_classification_amd import _lib, ops
    lib = _lib.get()
    d = ops.make_conv_desc(32, 32, (16, 16), 7, 3, 1, 24, False, True, 1.0)
    B, T = 2, 8
    z = lambda *s, dt=torch.float32: torch.zeros(*s, device=dev, dtype=dt)
    spk_in, W, b, tau4 = z(T, B, 32, 8, dt=torch.int32), z(32, 32, 7, 7), z(32), z(4, 32)
    e0, e1, arp, pv = z(B, 32, 16, 16), z(B, 32, 16, 16), z(B, 32, 16, 16), z(T, B, 32, 16, 16)
    q = z(32, 32, 7, 7, dt=torch.int8)

    def seq(opts, Wt=W):
        return lib.dcll_conv_lif_sequence(ctypes.byref(d), _lib.ptr(spk_in), _lib.ptr(Wt), _lib.ptr(b), _lib.ptr(tau4),
                                          _lib.ptr(e0), _lib.ptr(e1), _lib.ptr(arp), None, _lib.ptr(pv), None, None, None,
                                          None, 0, None, None, 0, opts, T, B, None)
    o = _lib.LayerOpts()
    assert seq(ctypes.byref(o)) == _lib.DCLL_OK
    o.w_q8 = q.data_ptr()
    assert seq(ctypes.byref(o)) == _lib.DCLL_ERR_INVALID and b"w_scale" in lib.dcll_last_error()
    o = _lib.LayerOpts()
    o.reserved = 1
    assert seq(ctypes.byref(o)) == _lib.DCLL_ERR_INVALID and b"reserved" in lib.dcll_last_error()
    assert seq(None, Wt=None) == _lib.DCLL_ERR_INVALID
    o = _lib.LayerOpts()
    o.pv_presigmoid = 1
    x, s = z(B, 32, 16, 16), z(B, 32, 16, 16)
    one = z(1)
    rc = lib.dcll_conv_lif_step(ctypes.byref(d), _lib.ptr(x), _lib.ptr(W), _lib.ptr(b), _lib.ptr(one), _lib.ptr(one),
                                _lib.ptr(one), _lib.ptr(one), _lib.ptr(e0), _lib.ptr(e1), _lib.ptr(arp), None, None, None,
                                None, _lib.ptr(s), None, None, _lib.ptr(pv[0]), None, None, ctypes.byref(o), B, None)
    assert rc == _lib.DCLL_ERR_INVALID and b"pv_presigmoid" in lib.dcll_last_error()
    torch.cuda.synchronize()


def test_step_readouts_argument_checks(dev):
    """dcll_step_readouts (ABI 4) refuses what it does not serve with a status and a message — nothing is launched, nothing
    throws across the ABI: shapes outside the split-K form (callers then fall back to dcll_readout + dcll_argmax_vote),
    null / inconsistent pointers, scratch that is too small, an unknown loss kind."""
    from snn_modulation_classification_amd import _lib
    lib = _lib.get()
    z = lambda *s, dt=torch.float32: torch.zeros(*s, device=dev, dtype=dt)
    rows, K, N = 64, 8192, 24
    pv, Wt, bias, p, o = z(rows, K), z(2 * N, K), z(2 * N), z(rows, N), z(rows, N)
    need = lib.dcll_step_readouts_scratch(rows, K, N, N)
    assert need == lib.dcll_readout_splitk_scratch(rows, K, 2 * N) > 0
    assert lib.dcll_step_readouts_scratch(rows, 8192 + 32, N, 0) == 0          # K % 256 != 0
    assert lib.dcll_step_readouts_scratch(rows, 131072, N, 0) == 0             # long rows: the 4096-column split form
    assert lib.dcll_step_readouts_scratch(4096, K, N, 0) == 0                  # many rows: the plain GEMM
    assert lib.dcll_step_readouts_scratch(rows, K, 40, 40) == 0                # more than 64 stacked rows
    sc = z(need)
    cl, tg, gp, go = z(rows, dt=torch.int32), z(rows, N), z(rows, N), z(rows, N)
    P = _lib.ptr

    def call(pv_=pv, K_=K, N2=N, o_=o, sc_=sc, need_=need, tg_=None, gp_=None, go_=None, kind=0):
        return lib.dcll_step_readouts(P(pv_), P(Wt), P(bias), P(sc_), need_, rows, K_, N, N2, P(p), P(o_), P(cl), P(tg_), P(gp_),
                                      P(go_), kind, None)
    assert call() == _lib.DCLL_OK
    assert call(tg_=tg, gp_=gp, go_=go, kind=1) == _lib.DCLL_OK
    assert call(o_=None) == _lib.DCLL_ERR_INVALID                              # output layer without o
    assert call(N2=7) == _lib.DCLL_ERR_INVALID                                 # output_ rows != i2o rows
    assert call(tg_=tg) == _lib.DCLL_ERR_INVALID                               # loss gradients asked for, no place to put them
    assert call(tg_=tg, gp_=gp, go_=go, kind=5) == _lib.DCLL_ERR_UNSUPPORTED and b"SmoothL1" in lib.dcll_last_error()
    assert call(need_=need - 1) == _lib.DCLL_ERR_INVALID and b"scratch" in lib.dcll_last_error()
    assert call(K_=8192 + 32) == _lib.DCLL_ERR_UNSUPPORTED
    assert call(pv_=pv.reshape(-1)[1:1 + rows * (K - 1)]) == _lib.DCLL_ERR_UNSUPPORTED      # pv not 16-byte aligned
    assert lib.dcll_step_readouts(P(pv), P(Wt), P(bias), P(sc), need, 0, K, N, N, P(p), P(o), None, None, None, None, 0,
                                  None) == _lib.DCLL_OK                       # empty batch: nothing to do
    torch.cuda.synchronize()


@pytest.mark.parametrize("case", [c for c in SEQ_CASES if c[0] in ("c32d", "c32", "c32t", "c1", "c1_narrow", "c1t",
                                                                      "w3_wide", "w3_mid", "w3_narrow", "w3_first")],
                         ids=lambda c: c[0])
def test_presigmoid_buffer_holds_v_and_statistics_are_unchanged(dev, case):
    """pv_presigmoid: pv_out receives the bit-exact v (max-pooled where the layer pools) and the pv statistics counted on
    it (sigmoid applied inside the counting pass) equal those of the pv = sigmoid(v) run; spikes, state and v_out do not
    change.  T covers two histogram steps (iterations 20 and 40)."""
    from snn_modulation_classification_amd import ops
    name, cin, cout, hw, ks, pad, pool, _, B = case
    T = 41
    rng = np.random.RandomState(29 + len(name))
    H, Wd = hw
    kh, kw = (ks, ks) if isinstance(ks, int) else ks
    stdv = 1.0 / np.sqrt(cin * kh * kw) / 250
    W = (rng.uniform(-stdv * 1e-2, stdv * 1e-2, size=(cout, cin, kh, kw)) * 3.0).astype(np.float32)
    # biases large enough that pv leaves the middle bins on some neurons: the first / last bins are populated
    b = (rng.uniform(-1, 1, size=(cout,)) * 4.0).astype(np.float32)
    _, _, alpha, tau_m, alphas, tau_s = _rand_layer(rng, cin, cout)
    d = ops.make_conv_desc(cin, cout, hw, ks, pad, pool, 24, False, True, 1.0)
    ch, cw, ph, pw = ops.conv_out_shape(d)
    tau4 = cu(np.stack([alpha, tau_m, alphas, tau_s]), dev)
    if cin == 1:
        inp = cu(rng.randint(0, H * Wd, size=(T, B)).astype(np.int32), dev)
    else:
        inp = ops.pack_spikes(cu((rng.uniform(size=(T, B, cin, H * Wd)) < 0.1).astype(np.float32), dev))
    want_spk = pool == 1 or (H * Wd) % 64 == 0
    fn = ops.conv_lif_sequence_cells if cin == 1 else ops.conv_lif_sequence

    def run(presig, want_v):
        st = [torch.zeros((B, cin, H, Wd), device=dev) for _ in range(2)] + [torch.zeros((B, cout, ch, cw), device=dev)]
        out = {}
        spk, pv, v = fn(d, inp, cu(W, dev), cu(b, dev), tau4, st[0], st[1], st[2], T, B, want_spikes=want_spk,
                        want_v=want_v, out=out, lowhigh_iter0=0, presigmoid=presig)
        torch.cuda.synchronize()
        return spk, pv, v, st, out["lowhigh"].cpu().numpy()

    spk0, pv0, v0, st0, lh0 = run(False, True)
    for want_v in (False, True):         # (fast epilogues write only pv_out; with v_out the guarded ones run)
        spk1, pv1, v1, st1, lh1 = run(True, want_v)
        if want_spk:
            assert torch.equal(spk0, spk1)
        for a, c in zip(st0, st1):
            assert torch.equal(a, c)
        if pool == 1:
            assert torch.equal(pv1, v0)                                     # the buffer IS v, bit for bit
        else:
            pooled = torch.maximum(v0[..., 0::2], v0[..., 1::2])             # (1,2) max-pool of the un-pooled v
            assert torch.equal(pv1, pooled)
        if want_v:
            assert torch.equal(v1, v0)
        np.testing.assert_allclose(torch.sigmoid(pv1).cpu().numpy(), pv0.cpu().numpy(), atol=PV_TOL, rtol=0)
        assert lh0.shape == (2, 2) and np.array_equal(lh0, lh1), (lh0, lh1)
    assert lh0.sum() > 0, "degenerate test: no pv value in the first or last bin"
    # the counting pass on a presigmoid buffer as a call of its own
    assert np.array_equal(ops.pv_lowhigh(pv1.reshape(T, -1), T, 0, presigmoid=True).cpu().numpy(), lh0)


@pytest.mark.parametrize("rows,K,N", [(300, 8192, 24), (2500, 8192, 48), (129, 64, 33), (5, 32, 64), (2100, 65536, 24),
                                      (40, 131072, 48), (1000, 1024, 24)])
def test_readout_act(dev, rows, K, N):
    """dcll_readout_act: with DCLL_ACT_SIGMOID == float64 sigmoid(v) . W^T + b inside the logit tolerance; with
    DCLL_ACT_NONE the plain readout; and in both forms a row's logits do not depend on how many rows the call has (the
    sequence path chunks batches) — incl. K >= 65536, which is split over K by K alone."""
    from snn_modulation_classification_amd import ops
    rng = np.random.RandomState(9)
    v = rng.normal(0, 2.5, size=(rows, K)).astype(np.float32)
    W = rng.uniform(-.0055, .0055, size=(N, K)).astype(np.float32) * np.float32(np.sqrt(8192.0 / K))
    b = rng.uniform(-.0055, .0055, size=(N,)).astype(np.float32)
    sig = 1.0 / (1.0 + np.exp(-v.astype(np.float64)))
    dv, dW, db = cu(v, dev), cu(W, dev), cu(b, dev)
    assert ops.readout_act_supported(dv, dW)
    guard = torch.full((rows + 8, N), 7.0, device=dev)
    out = ops.readout_act(dv, dW, db, out=guard[:rows], presigmoid=True)
    assert float(guard[rows:].min()) == 7.0 and float(guard[rows:].max()) == 7.0
    np.testing.assert_allclose(out.cpu().numpy(), sig @ W.astype(np.float64).T + b, atol=3e-5, rtol=0)
    plain = ops.readout_act(dv, dW, db, presigmoid=False)
    np.testing.assert_allclose(plain.cpu().numpy(), v.astype(np.float64) @ W.astype(np.float64).T + b, atol=2e-4, rtol=0)
    # sigmoid applied by the kernel == the kernel on a buffer that already holds the device sigmoid of the same values
    # (summation order identical) up to the ulp the two sigmoid instruction sequences may differ by
    pre = ops.readout_act(torch.sigmoid(dv), dW, db, presigmoid=False)
    np.testing.assert_allclose(out.cpu().numpy(), pre.cpu().numpy(), atol=2e-6, rtol=0)
    if rows > 140:
        part = ops.readout_act(dv[:133].contiguous(), dW, db, presigmoid=True)
        assert torch.equal(part, out[:133])
        part = ops.readout_act(dv[:133].contiguous(), dW, db, presigmoid=False)
        assert torch.equal(part, plain[:133])
    # shapes the 16x16x4 kernel does not serve are refused, not mis-read
    from snn_modulation_classification_amd import _lib
    bad = torch.zeros((4, 100), device=dev)
    assert not ops.readout_act_supported(bad, torch.zeros((3, 100), device=dev))
    with pytest.raises(_lib.DCLLUnsupported):
        ops.readout_act(bad, torch.zeros((3, 100), device=dev), torch.zeros(3, device=dev))


def _net(yaml_name, im, B, int8):
    import os
    from argparse import Namespace
    from conftest import ROOT
    from snn_modulation_classification_amd import quant
    from snn_modulation_classification_amd.networks import ConvNetwork, load_network_spec
    convs = load_network_spec(os.path.join(ROOT, "snn_modulation_classification_amd", "networks", yaml_name))
    args = Namespace(netscale=1.0, alpha=.92, alphas=.85, alpharp=.65, arp=1.0, lc_ampl=.5, random_tau=True)
    torch.manual_seed(2)
    np.random.seed(2)
    net = ConvNetwork(args, im, B, convs, 24, act=torch.nn.Sigmoid(), loss=None, opt=None, opt_param={},
                      learning_rates=None, burnin=2)
    net.reset(True)
    if int8:
        quant.apply_int8_weights(net)
    return net


@pytest.mark.parametrize("yaml_name,im,B,T,int8", [("radio_ml_conv.yaml", (1, 16, 16), 21, 45, False),
                                                   ("radio_ml_conv.yaml", (1, 32, 32), 3, 21, True),
                                                   ("radio_ml_conv_ref.yaml", (1, 16, 128), 5, 23, True)])
def test_network_int8_abi_and_presigmoid_do_not_change_the_run(dev, monkeypatch, yaml_name, im, B, T, int8):
    """ConvNetwork.test_sequence with the int8 tensors handed across the ABI and v in the pv buffer (the defaults) == the
    same run with DCLL_INT8_ABI=0 (kernels read the dequantised fp32 Parameter) and DCLL_PRESIGMOID=0 (pv = sigmoid(v) in
    the buffer): neuron state of every layer bit for bit, logits within the sigmoid's ulp, identical pv statistics, same
    per-step argmax up to logit ties."""
    from snn_modulation_classification_amd.data.utils import IQEncoder
    H, W = im[1], im[2]
    torch.manual_seed(5)
    iq = (0.4 * torch.randn(B, 2, 128)).to(dev)
    enc = IQEncoder(W, H, device=dev)
    monkeypatch.setenv("DCLL_PRESIGMOID", "1")             # (the default, 'auto', turns it on for the pooling layers only)
    a = _net(yaml_name, im, B, int8)
    assert a.presigmoid == '1' and all((s.dclllayer.i2h.int8_weights() is not None) == int8 for s in a.dcll_slices)
    a.reset()
    ra = a.test_sequence(iq=iq, encoder=enc, T=T, t0=2)
    monkeypatch.setenv("DCLL_INT8_ABI", "0")
    monkeypatch.setenv("DCLL_PRESIGMOID", "0")
    b = _net(yaml_name, im, B, int8)
    assert b.presigmoid == '0' and all(s.dclllayer.i2h.int8_weights() is None for s in b.dcll_slices)
    b.reset()
    rb = b.test_sequence(iq=iq, encoder=enc, T=T, t0=2)
    torch.cuda.synchronize()
    for i, (sa, sb) in enumerate(zip(a.dcll_slices, b.dcll_slices)):
        for x, y in zip(sa.dclllayer.i2h.state, sb.dclllayer.i2h.state):
            assert torch.equal(x, y), i
        np.testing.assert_allclose(ra["logits"][i].cpu().numpy(), rb["logits"][i].cpu().numpy(), atol=2e-5, rtol=0)
        assert torch.equal(ra["lowhigh"][i], rb["lowhigh"][i]) and ra["lowhigh"][i].shape[0] == T // 20
        lg = (ra["o"] if i == len(a.dcll_slices) - 1 else ra["logits"][i]).cpu().numpy()
        top2 = np.sort(lg, axis=-1)[..., -2:]
        tie = (top2[..., 1] - top2[..., 0]) <= 1e-4
        ca, cb = ra["clout"][i].cpu().numpy(), rb["clout"][i].cpu().numpy()
        assert np.array_equal(ca[~tie], cb[~tie]), i
    np.testing.assert_allclose(ra["o"].cpu().numpy(), rb["o"].cpu().numpy(), atol=2e-5, rtol=0)


def test_int8_form_is_dropped_when_the_weight_changes(dev):
    """The int8 tensor describes i2h.weight only until somebody writes the Parameter (a learning step, load_state_dict):
    then the layer goes back to the fp32 weight instead of silently running on stale int8 values."""
    net = _net("radio_ml_conv.yaml", (1, 16, 16), 2, True)
    i2h = net.dcll_slices[1].dclllayer.i2h
    assert i2h.int8_weights() is not None
    with torch.no_grad():
        i2h.weight.mul_(1.5)
    assert i2h.int8_weights() is None
    sd = net.dcll_slices[2].dclllayer.state_dict()
    net.dcll_slices[2].dclllayer.load_state_dict(sd)
    assert net.dcll_slices[2].dclllayer.i2h.int8_weights() is None
    assert net.dcll_slices[0].dclllayer.i2h.int8_weights() is not None


@pytest.mark.parametrize("graph", [False, True])
def test_native_learning_step_drops_int8_form_and_readout_caches(dev, graph, monkeypatch):
    """dcll_adam_step writes the parameters through raw device pointers — no tensor version counter moves — so the
    learning paths (train_dcll, ConvNetwork.learn eager and as a hipGraph replay) must invalidate what was derived from
    the weights themselves: after quant.apply_int8_weights + native learning steps the layers read the (updated) fp32
    weight again, and the stacked output-layer readout equals the live output_ weight."""
    import os
    from argparse import Namespace
    from conftest import ROOT
    from snn_modulation_classification_amd import quant
    from snn_modulation_classification_amd.networks import ConvNetwork, load_network_spec
    monkeypatch.setenv("DCLL_GRAPH_LEARN", "1" if graph else "0")
    convs = load_network_spec(os.path.join(ROOT, "snn_modulation_classification_amd", "networks", "radio_ml_conv.yaml"))
    args = Namespace(netscale=1.0, alpha=.92, alphas=.85, alpharp=.65, arp=1.0, lc_ampl=.5, random_tau=True)
    torch.manual_seed(2)
    np.random.seed(2)
    B = 8
    net = ConvNetwork(args, (1, 16, 16), B, convs, 24, act=torch.nn.Sigmoid(), loss=torch.nn.SmoothL1Loss,
                      opt=torch.optim.Adam, opt_param={"betas": [0.0, .95], "weight_decay": 10.0},
                      learning_rates=[1e-4], burnin=2)
    net.reset(True)
    assert all(s._native_learning() is not None for s in net.dcll_slices)
    quant.apply_int8_weights(net)
    last = net.dcll_slices[-1].dclllayer
    Wt0, _ = last.stacked_readout()
    Wt0 = Wt0.clone()
    w_before = [s.dclllayer.i2h.weight.detach().clone() for s in net.dcll_slices]
    rng = np.random.RandomState(0)
    y = torch.zeros(B, 24, device=dev)
    y[torch.arange(B), torch.from_numpy(rng.randint(0, 24, B))] = 1
    n_steps = 8 if graph else 3               # (the capture is taken after two eager learning steps)
    for t in range(n_steps):
        x = torch.zeros(B, 256, device=dev)
        x[torch.arange(B), torch.from_numpy(rng.randint(0, 256, B))] = 1
        net.learn(x.reshape(B, 1, 16, 16), y)
    torch.cuda.synchronize()
    if graph:
        assert net._learn_graphs, "the learning step was not replayed from a captured graph"
    for s, w0 in zip(net.dcll_slices, w_before):
        assert not torch.equal(s.dclllayer.i2h.weight, w0), "the step did not train"
        assert s.dclllayer.i2h.int8_weights() is None
    Wt, bias = last.stacked_readout()
    assert torch.equal(Wt[24:], last.output_.weight) and torch.equal(bias[24:], last.output_.bias)
    assert not torch.equal(Wt, Wt0)
    Wp, _ = last.fused_readout_weights()
    from snn_modulation_classification_amd import ops
    assert torch.equal(Wp, ops.permute_readout(torch.cat([last.i2o.weight, last.output_.weight], 0)))


# (in_features, out_features, B, T, tensor tau, wrp)
DENSE_CASES = [(40, 24, 5, 4, False, 1.0), (40, 24, 5, 4, True, 0.0), (257, 130, 133, 3, True, 1.0), (1024, 128, 70, 6, True, 1.0),
               (1000, 96, 33, 5, False, 0.0), (7, 3, 1, 5, True, 1.0), (2050, 64, 40, 3, True, 1.0), (96, 200, 64, 4, False, 1.0),
               # launches large enough for the 128 x 64 workgroups of k_dense_lif_mfma (>= 512 of them; smaller launches take the
               # 128 x 32 form): ragged in both directions with the plain tile order, and a multiple of eight sample blocks =
               # the XCD-aware tile order; the same order on the 128 x 32 form
               (36, 520, 7300, 2, True, 1.0), (64, 512, 8192, 2, False, 0.0), (48, 96, 1024, 2, True, 1.0)]


@pytest.mark.parametrize("cin,cout,B,T,ttau,wrp", DENSE_CASES)
def test_dense_twins_vs_oracle(dev, cin, cout, B, T, ttau, wrp):
    """The dense twins (reference dcll/pytorch_libdcll.py:131-148, :171-195, :250-255): dcll_dense_lif_step — the fp32-MFMA
    GEMM with an unsplit, in-order K loop — and dcll_dense_lif_sequence — all T steps in one call, state on chip when the
    layer is small (in <= 1024, out <= 128), step by step inside the call otherwise — == the C oracle stepping: v, s, eps0 /
    eps1 / arp bit for bit, from a non-zero state; (C_in,)-tensor and scalar time constants, refractory or not, odd K, ragged
    tiles, out above the on-chip limit, in above it."""
    from snn_modulation_classification_amd import ops
    from snn_modulation_classification_amd._lib import DenseDesc
    from oracle import c_oracle as C
    rng = np.random.RandomState(cin + cout)
    stdv = 1.0 / np.sqrt(cin)
    W = (rng.uniform(-stdv * 1e-2, stdv * 1e-2, size=(cout, cin)) * 3.0).astype(np.float32)
    b = (rng.uniform(-stdv, stdv, size=(cout,)) * 0.02).astype(np.float32)
    n = cin if ttau else 1
    taum, taus = rng.uniform(5, 35, size=n) * 1e-3, rng.uniform(5, 10, size=n) * 1e-3
    alpha, alphas = (1 - 1e-3 / taum).astype(np.float32), (1 - 1e-3 / taus).astype(np.float32)
    tau_m = (np.float32(1) / (np.float32(1) - alpha)).astype(np.float32)
    tau_s = (np.float32(1) / (np.float32(1) - alphas)).astype(np.float32)
    sd = {"i2h.weight": W, "i2h.bias": b, "i2h.alpha": alpha, "i2h.tau_m__dt": tau_m, "i2h.alphas": alphas,
          "i2h.tau_s__dt": tau_s, "i2o.weight": rng.uniform(-.05, .05, size=(10, cout)).astype(np.float32),
          "i2o.bias": rng.uniform(-.05, .05, size=(10,)).astype(np.float32)}
    orc = C.OracleDenseLayer(sd, wrp)
    x = (rng.uniform(size=(T, B, cin)) < 0.2).astype(np.float32)
    x[:, :, ::5] *= rng.uniform(0.5, 2.0, size=x[:, :, ::5].shape).astype(np.float32)      # any fp32 input is legal
    orc.forward(np.zeros((B, cin), np.float32))            # allocates the state
    init = [rng.uniform(0, 5, size=(B, cin)).astype(np.float32), rng.uniform(0, 50, size=(B, cin)).astype(np.float32),
            -rng.uniform(0, 2, size=(B, cout)).astype(np.float32)]
    for s_, i_ in zip(orc.state, init):
        s_[...] = i_
    t = {k: cu(v, dev) for k, v in sd.items()}
    d = DenseDesc(cin, cout, 10, int(ttau), int(wrp > 0), .65, wrp)
    args = (t["i2h.weight"], t["i2h.bias"], t["i2h.alpha"], t["i2h.tau_m__dt"], t["i2h.alphas"], t["i2h.tau_s__dt"])
    st_a = [cu(i_.copy(), dev) for i_ in init]             # per-step twin
    st_b = [cu(i_.copy(), dev) for i_ in init]             # sequence twin
    s_seq, p_seq, pv_seq, v_seq = ops.dense_lif_sequence(d, cu(x, dev), *args, st_b[0], st_b[1],
                                                         st_b[2] if wrp > 0 else None, t["i2o.weight"], t["i2o.bias"],
                                                         want_v=True)
    nspk = 0
    for k in range(T):
        s, p, pv, v = ops.dense_lif_step(d, cu(x[k], dev), *args, st_a[0], st_a[1], st_a[2] if wrp > 0 else None,
                                         t["i2o.weight"], t["i2o.bias"])
        os_, op, opv, ov = orc.forward(x[k])
        assert bits_equal(v.cpu().numpy(), ov), (k, np.abs(v.cpu().numpy() - ov).max())
        assert np.array_equal(s.cpu().numpy(), os_)
        assert bits_equal(v_seq[k].cpu().numpy(), ov), (k, np.abs(v_seq[k].cpu().numpy() - ov).max())
        assert np.array_equal(s_seq[k].cpu().numpy(), os_)
        np.testing.assert_allclose(pv_seq[k].cpu().numpy(), opv, atol=PV_TOL, rtol=0)
        np.testing.assert_allclose(p.cpu().numpy(), op, atol=LOGIT_TOL, rtol=0)
        np.testing.assert_allclose(p_seq[k].cpu().numpy(), op, atol=LOGIT_TOL, rtol=0)
        nspk += os_.sum()
    for j in range(3 if wrp > 0 else 2):
        assert bits_equal(st_a[j].cpu().numpy(), orc.state[j]), j
        assert bits_equal(st_b[j].cpu().numpy(), orc.state[j]), j
    assert 0.01 < nspk / (T * B * cout) < 0.99, "degenerate test"


@pytest.mark.parametrize("case,cin,cout,wrp,rtau", [("rrp_512_128", 512, 128, 1.0, True), ("plain_600_160", 600, 160, 0.0, False)])
def test_dense_layer_reproduces_the_reference_sequence(golden, dev, case, cin, cout, wrp, rtau):
    """Fixture g7b (generated by importing the reference): DenseDCLLlayer over 24 steps — 512 -> 128 refractory with
    per-feature time constants (the all-T on-chip kernel k_dense_lif_seq) and 600 -> 160 plain with scalar ones (the per-step
    fp32-MFMA GEMM k_dense_lif_mfma, odd tile counts).  `.forward` step by step AND `.forward_sequence` give the reference's
    output spikes bit for bit, its readouts within 1e-4, and its final state (traces bit for bit)."""
    from snn_modulation_classification_amd.dcll.pytorch_libdcll import DenseDCLLlayer
    g = golden("g7b_dense_sequence.npz")
    pre = "g7b/%s/" % case
    T, B = g[pre + "x"].shape[0], 5
    x = torch.from_numpy(np.unpackbits(g[pre + "x"], axis=-1, bitorder="little")[..., :cin].astype(np.float32)).to(dev)
    want_s = np.unpackbits(g[pre + "s"], axis=-1, bitorder="little")[..., :cout].astype(np.float32)

    def make():
        L = DenseDCLLlayer(cin, cout, target_size=10, alpha=.9, alphas=.85, alpharp=.65, wrp=wrp, random_tau=False)
        sd = {k: torch.from_numpy(v) for k, v in g.sub(pre + "sd/").items()}
        for nm in ("alpha", "tau_m__dt", "alphas", "tau_s__dt"):       # (per-feature tensors when the reference drew them)
            getattr(L.i2h, nm).data = sd["i2h." + nm].clone()
        L.load_state_dict(sd)
        return L.to(dev).init_hiddens(B)
    a, b = make(), make()
    for t_ in range(T):
        s, p, pv, v = a.forward(x[t_])
        assert np.array_equal(s.cpu().numpy(), want_s[t_]), (t_, int((s.cpu().numpy() != want_s[t_]).sum()))
        np.testing.assert_allclose(p.cpu().numpy(), g[pre + "p"][t_], atol=LOGIT_TOL, rtol=0)
    s_seq, p_seq, pv_seq, _ = b.forward_sequence(x)
    assert np.array_equal(s_seq.cpu().numpy(), want_s)
    np.testing.assert_allclose(p_seq.cpu().numpy(), g[pre + "p"], atol=LOGIT_TOL, rtol=0)
    for L in (a, b):
        for i, nm in enumerate(("eps0", "eps1", "arp")[:3 if wrp > 0 else 2]):
            assert bits_equal(getattr(L.i2h.state, nm).cpu().numpy(), g[pre + "final_" + nm]), (nm,)


@pytest.mark.parametrize("native", [True, False])
@pytest.mark.parametrize("name,wrp,reg", [("rrp", 1.0, False), ("plain", 0.0, False), ("plain_rtau", 0.0, False), ("reg", 1.0, 0.05)])
def test_dense_slice_local_learning_reproduces_the_reference(golden, dev, name, wrp, reg, native, monkeypatch):
    """Fixture G6d (generated by importing the reference): DCLLClassification(DenseDCLLlayer) under DCLLBase.train_dcll
    (dcll/pytorch_libdcll.py:690-718 — layer-agnostic: the optimizer is built from dclllayer.i2h.parameters(), :634-635),
    SmoothL1Loss + Adam(betas (0, .95), weight_decay 10), burn-in 4, six learning steps with the neuron state and the Adam
    moments carried along; 512 -> 128 refractory, 256 -> 64 / 200 -> 72 plain with scalar / per-feature time constants, and
    the refractory layer with train_dcll's default regularisers (which reach pvmem and pv directly: autograd path whatever
    `native` says).  Until round 6 train_dcll raised NotImplementedError on a dense slice (round-5 verdict, missing #2).
    native: dcll_dense_lif_step -> dcll_local_loss_grad -> dcll_dense_lif_backward -> dcll_adam_step, no torch op per step;
    else the same HIP forward / backward inside an autograd node with torch's loss module and optimizer.
    Output spikes bit for bit, readouts within 1e-4, losses, recorded argmax, gradients at the first and last learning step
    (fp32 sums in another order: rtol 2e-3), the weight CHANGE over the six steps within 2 % of its largest element."""
    from snn_modulation_classification_amd.dcll.pytorch_libdcll import DenseDCLLlayer, DCLLClassification
    from snn_modulation_classification_amd import ops
    monkeypatch.setenv("DCLL_NATIVE_LEARNING", "1" if native else "0")
    g = golden("g6d_dense_learning.npz")
    pre = "g6d/%s/" % name
    sd0 = {k: torch.from_numpy(v) for k, v in g.sub(pre + "sd0/").items()}
    cout, cin = sd0["i2h.weight"].shape
    T, B = g[pre + "x"].shape[0], 8
    x = torch.from_numpy(np.unpackbits(g[pre + "x"], axis=-1, bitorder="little")[..., :cin].astype(np.float32)).to(dev)
    want_s = np.unpackbits(g[pre + "s"], axis=-1, bitorder="little")[..., :cout].astype(np.float32)
    tgt = torch.from_numpy(g[pre + "target"]).to(dev)
    L = DenseDCLLlayer(cin, cout, target_size=24, alpha=.9, alphas=.85, alpharp=.65, wrp=wrp, random_tau=False)
    for nm in ("alpha", "tau_m__dt", "alphas", "tau_s__dt"):           # (per-feature tensors when the reference drew them)
        getattr(L.i2h, nm).data = sd0["i2h." + nm].clone()
    L.load_state_dict(sd0)
    L = L.to(dev)
    sl = DCLLClassification(dclllayer=L, name="dense", batch_size=B, loss=torch.nn.SmoothL1Loss, optimizer=torch.optim.Adam,
                            kwargs_optimizer={"lr": 1e-5, "betas": [0.0, .95], "weight_decay": 10.0}, burnin=4)
    assert (sl._native_learning() is not None) == native
    sl.train()
    with ops.kernel_trace() as launched:
        for t_ in range(T):
            o, p, pv, v, loss = sl.train_dcll(x[t_], tgt, regularize=reg)
            assert np.array_equal(o.cpu().numpy(), want_s[t_]), (t_, int((o.cpu().numpy() != want_s[t_]).sum()))
            np.testing.assert_allclose(p.detach().cpu().numpy(), g[pre + "p"][t_], atol=LOGIT_TOL, rtol=0)
            np.testing.assert_allclose(float(loss), g[pre + "loss"][t_], rtol=1e-4, atol=1e-7)
            key = pre + "grad/%d/w" % t_
            if key in g.keys():
                gw, gb = L.i2h.weight.grad.cpu().numpy(), L.i2h.bias.grad.cpu().numpy()
                np.testing.assert_allclose(gw, g[key], rtol=2e-3, atol=1e-6 * np.abs(g[key]).max())
                np.testing.assert_allclose(gb, g[pre + "grad/%d/b" % t_], rtol=2e-3, atol=1e-6 * np.abs(g[pre + "grad/%d/b" % t_]).max())
    if native and not reg:      # (the autograd engine runs a node's backward on its own thread: the per-thread launch log of this
        #                          thread sees the forward only)
        assert launched.count("k_dense_bwd_wgrad") == 6 and launched.count("k_dense_bwd_dv") == 6, launched.names
        assert launched.count("k_adam_multi") == 6, launched.names
    else:
        assert launched.count("k_adam_multi") == 0, launched.names                                # torch's optimizer
    assert np.array_equal(np.asarray(sl.clout), g[pre + "clout"])
    for nm in ("weight", "bias"):
        w0, w1 = g[pre + "sd0/i2h." + nm], g[pre + "sd1/i2h." + nm]
        mine = getattr(L.i2h, nm).detach().cpu().numpy()
        np.testing.assert_allclose(mine - w0, w1 - w0, rtol=0, atol=2e-2 * np.abs(w1 - w0).max(), err_msg=nm)
        assert np.abs(w1 - w0).max() > 0
    for i, nm in enumerate(("eps0", "eps1", "arp")[:3 if wrp > 0 else 2]):
        assert bits_equal(getattr(L.i2h.state, nm).cpu().numpy(), g[pre + "final_" + nm]), nm
    for k in ("i2o.weight", "i2o.bias"):
        assert np.array_equal(L.state_dict()[k].cpu().numpy(), g[pre + "sd1/" + k])          # frozen


@pytest.mark.parametrize("cin,cout,target,B", [(40, 24, 10, 5), (777, 130, 24, 67), (512, 128, 24, 512)])
def test_dense_backward_closed_open_and_torch(dev, cin, cout, target, B):
    """dcll_dense_lif_backward against torch autograd through the same expressions in float64 (dv = (g_p . i2o + g_pv) * pv *
    (1 - pv) + g_v ; dW = dv^T eps1 ; db = sum dv), odd sizes (ragged MFMA tiles, an odd batch: the last sample pair is half
    empty); the open form + dcll_grad_reduce_adam gives the closed form's bits and dcll_adam_step's parameters."""
    from snn_modulation_classification_amd import ops, _lib
    rng = np.random.RandomState(5)
    eps1 = rng.uniform(0, 40, size=(B, cin)).astype(np.float32)
    pv = rng.uniform(0.01, 0.99, size=(B, cout)).astype(np.float32)
    g_p = rng.randn(B, target).astype(np.float32)
    g_pv = (rng.randn(B, cout) * 0.1).astype(np.float32)
    g_v = (rng.randn(B, cout) * 0.1).astype(np.float32)
    i2o = rng.uniform(-.1, .1, size=(target, cout)).astype(np.float32)
    d = _lib.DenseDesc(cin, cout, target, 0, 1, .65, 1.0)
    dv = (g_p.astype(np.float64) @ i2o + g_pv) * (pv.astype(np.float64) * (1 - pv)) + g_v
    ref_W, ref_b = dv.T @ eps1.astype(np.float64), dv.sum(0)
    dW, db = ops.dense_lif_backward(d, cu(eps1, dev), cu(pv, dev), cu(g_p, dev), cu(g_pv, dev), cu(g_v, dev), cu(i2o, dev))
    np.testing.assert_allclose(dW.cpu().numpy(), ref_W, rtol=1e-4, atol=1e-5 * np.abs(ref_W).max())
    np.testing.assert_allclose(db.cpu().numpy(), ref_b, rtol=1e-4, atol=1e-5 * np.abs(ref_b).max())
    # g_p alone (what the native learning step passes)
    dv2 = (g_p.astype(np.float64) @ i2o) * (pv.astype(np.float64) * (1 - pv))
    dW2, db2 = ops.dense_lif_backward(d, cu(eps1, dev), cu(pv, dev), cu(g_p, dev), None, None, cu(i2o, dev))
    np.testing.assert_allclose(dW2.cpu().numpy(), dv2.T @ eps1.astype(np.float64), rtol=1e-4, atol=1e-5 * np.abs(ref_W).max())
    # open form + reduce + Adam in one launch == closed form + dcll_adam_step
    hp = dict(lr=1e-5, weight_decay=10.0, beta1=0.0, beta2=.95, eps=1e-8)
    W0, b0 = cu(rng.randn(cout, cin).astype(np.float32) * 0.02, dev), cu(rng.randn(cout).astype(np.float32) * 0.02, dev)
    ent = lambda P, G: [dict(param=q, grad=g_, exp_avg=torch.zeros_like(q), exp_avg_sq=torch.zeros_like(q), step=2, **hp)
                        for q, g_ in zip(P, G)]
    closed = ent([W0.clone(), b0.clone()], [dW, db])
    ops.adam_step(closed)
    out = {}
    dW3, db3 = ops.dense_lif_backward(d, cu(eps1, dev), cu(pv, dev), cu(g_p, dev), cu(g_pv, dev), cu(g_v, dev), cu(i2o, dev),
                                      out=out, open_reduce=True)
    opened = ent([W0.clone(), b0.clone()], [dW3, db3])
    ops.grad_reduce_adam([dict(out["parts"], adam_w=0, adam_b=1)], opened)
    assert torch.equal(dW3, dW) and torch.equal(db3, db)
    for a, b_ in zip(closed, opened):
        for key in ("param", "exp_avg", "exp_avg_sq"):
            assert torch.equal(a[key], b_[key]), key
    assert not torch.equal(opened[0]["param"], W0)
    with pytest.raises(ValueError):
        import ctypes
        _lib.check(_lib.get().dcll_dense_lif_backward(ctypes.byref(d), _lib.ptr(cu(eps1, dev)), _lib.ptr(cu(pv, dev)), None, None,
                                                      None, None, _lib.ptr(dW), _lib.ptr(db), _lib.ptr(out["bwd_scratch"]), 7, B,
                                                      None), "too little scratch")


def test_dense_layer_forward_sequence_equals_forward(dev):
    """DenseDCLLlayer.forward_sequence == T calls of .forward (the reference's protocol, :250-255), bit for bit."""
    from snn_modulation_classification_amd.dcll.pytorch_libdcll import DenseDCLLlayer
    T, B = 9, 37
    x = (torch.rand(T, B, 300, device=dev) < 0.3).float()
    outs = []
    for _ in range(2):
        torch.manual_seed(4)
        np.random.seed(4)
        outs.append(DenseDCLLlayer(300, 50, target_size=10, wrp=1.0, random_tau=True).to(dev).init_hiddens(B))
    a, b = outs
    sa, pa, pva, va = a.forward_sequence(x, want_v=True)
    for t in range(T):
        s, p, pv, v = b.forward(x[t])
        assert torch.equal(sa[t], s) and torch.equal(va[t], v) and torch.equal(pva[t], pv)
        np.testing.assert_allclose(pa[t].cpu().numpy(), p.cpu().numpy(), atol=2e-5, rtol=0)
    for u, w_ in zip(a.i2h.state, b.i2h.state):
        assert torch.equal(u, w_)


@pytest.mark.parametrize("B,R_", [(37, 16), (4099, 16), (45, 32)])
def test_device_iq_encoder_is_exact_for_ragged_batches(dev, B, R_):
    """Device cells == host iq2cells (= the reference's per-time-sample torch quantiser, data/utils.py:60-79) for batch
    sizes that are NOT multiples of 32, on inputs placed on every cell boundary and one ulp either side: torch's float pow
    sends the last B mod 32 (16 on AVX2) positions of a batch vector through scalar libm, whose boundaries differ from the
    vector path's in the last ulp; the kernels get both threshold tables and the positions (dcll_iq_tail).  Checked for the
    stand-alone encoder kernel and for the quantiser fused into the first layer's sequence kernel (16x16: k_lif_seq_c1,
    32x32: k_lif_seq_c1t), whole and chunked batches."""
    from test_host_logic import _boundary_iq
    from snn_modulation_classification_amd import ops
    from snn_modulation_classification_amd.data.utils import IQEncoder, iq2cells
    enc = IQEncoder(R_, R_, device=dev)
    tabs = [enc.thr_i.cpu().numpy(), enc.thr_q.cpu().numpy()]
    if enc.thr_i_tail is not None:
        tabs += [enc.thr_i_tail.cpu().numpy(), enc.thr_q_tail.cpu().numpy()]
    L, T = 40, 33
    iq = _boundary_iq(B, L, tabs, seed=B + R_)
    np.random.seed(3)
    want, t0 = iq2cells(torch.from_numpy(iq), R_, R_, max_duration=T)
    got = enc(cu(iq, dev), T, t0=t0)
    assert torch.equal(got.cpu(), want), np.argwhere(got.cpu().numpy() != want.numpy())[:5]
    if enc.thr_i_tail is not None:       # not vacuous: the single-table encoder misplaces boundary samples of the tail
        plain = ops.iq_encode(cu(iq, dev), enc.thr_i, enc.thr_q, t0, T, R_, R_)
        assert not torch.equal(plain.cpu(), want)
    # fused into the first layer's kernel == the cells path on the host's cells, whole batch and chunks of 16
    d = ops.make_conv_desc(1, 32, (R_, R_), 7, 3, 1, 24, False, True, 1.0)
    rng = np.random.RandomState(1)
    W, b, alpha, tau_m, alphas, tau_s = _rand_layer(rng, 1, 32, gain=3.0)
    tau4 = cu(np.stack([alpha, tau_m, alphas, tau_s]), dev)

    def state(n):
        return [torch.zeros((n, 1, R_, R_), device=dev), torch.zeros((n, 1, R_, R_), device=dev),
                torch.zeros((n, 32, R_, R_), device=dev)]
    st = state(B)
    ref_spk, _, _ = ops.conv_lif_sequence_cells(d, cu(want.numpy(), dev), cu(W, dev), cu(b, dev), tau4, *st, T, B,
                                                want_pv=False)
    st = state(B)
    spk, _, _ = ops.conv_lif_sequence_iq(d, cu(iq, dev), enc.thr_i, enc.thr_q, t0, cu(W, dev), cu(b, dev), tau4, *st, T, B,
                                         want_pv=False, tail=enc.tail(B))
    assert torch.equal(spk, ref_spk)
    for b0 in range(0, B, 16):
        b1 = min(B, b0 + 16)
        st = state(b1 - b0)
        part, _, _ = ops.conv_lif_sequence_iq(d, cu(iq[b0:b1], dev), enc.thr_i, enc.thr_q, t0, cu(W, dev), cu(b, dev), tau4,
                                              *st, T, b1 - b0, want_pv=False, tail=enc.tail(b1 - b0, b0, b1, B))
        assert torch.equal(part, ref_spk[:, b0:b1]), b0


def test_one_layer_network_runs_the_sequence_path(dev):
    """A ConvNetwork with a single layer (sequence_supported() accepts it): its spike view must not be sized from "all
    layers but the last" (round-2 advisor finding: an empty max) — test_sequence runs and equals the per-step path."""
    import os
    from argparse import Namespace
    from conftest import ROOT
    from snn_modulation_classification_amd import ops
    from snn_modulation_classification_amd.networks import ConvNetwork, load_network_spec
    convs = load_network_spec(os.path.join(ROOT, "snn_modulation_classification_amd", "networks", "radio_ml_conv.yaml"))[:1]
    args = Namespace(netscale=1.0, alpha=.92, alphas=.85, alpharp=.65, arp=1.0, lc_ampl=.5, random_tau=True)
    B, T = 5, 9
    nets = []
    for _ in range(2):
        torch.manual_seed(3)
        np.random.seed(3)
        net = ConvNetwork(args, (1, 16, 16), B, convs, 24, act=torch.nn.Sigmoid(), loss=None, opt=None, opt_param={},
                          learning_rates=None, burnin=2)
        net.reset(True)
        nets.append(net)
    a, b = nets
    assert a.sequence_supported() and a.num_layers == 1
    cells = torch.randint(0, 256, (T, B), device=dev, dtype=torch.int32)
    a.reset()
    res = a.test_sequence(cells, keep_spikes=True)
    assert res["spikes"][0].shape == (T, B, 32, 8) and res["o"].shape == (T, B, 24)
    planes = ops.cells_to_planes(cells, 256)
    b.reset()
    for t in range(T):
        b.test(planes[t].reshape(B, 1, 16, 16))
    for x, y in zip(a.dcll_slices[0].dclllayer.i2h.state, b.dcll_slices[0].dclllayer.i2h.state):
        assert torch.equal(x, y)
    assert a.dcll_slices[0].iter == b.dcll_slices[0].iter == T


@pytest.mark.timeout(900)
def test_config5_properties_at_batch_1024_t128(dev):
    """BASELINE config 5 at a size the oracle cannot walk (radio_ml_conv_ref.yaml, int8 weights through the ABI, Q=16 x
    I=128 plane, T=128, batch 1024: 34 GB of pooled maps): size-independent properties — the run is deterministic, a run in
    chunks of 384 windows (pv budget) equals the unchunked run sample for sample (state, votes, per-step argmax; logits
    bit for bit: the sequence readout's summation order does not depend on the row count), and two samples are walked by
    the C oracle for the first steps (logits within 1e-4)."""
    from snn_modulation_classification_amd.data.utils import IQEncoder
    from snn_modulation_classification_amd.networks import load_network_spec
    from oracle import c_oracle as C
    import os
    from conftest import ROOT
    B, T, H, W = 1024, 128, 16, 128
    torch.manual_seed(8)
    iq = (0.4 * torch.randn(B, 2, 128)).to(dev)
    enc = IQEncoder(W, H, device=dev)

    def run(budget_gb):
        net = _net("radio_ml_conv_ref.yaml", (1, H, W), B, True)
        net.pv_budget_bytes = budget_gb * 2 ** 30
        net.reset()
        res = net.test_sequence(iq=iq, encoder=enc, T=T, t0=0)
        torch.cuda.synchronize()
        return net, res
    n1, r1 = run(150)
    n2, r2 = run(150)
    n3, r3 = run(12.6)          # 33.5 MB of pooled first-layer map per window: chunks of 385 windows
    for i in range(7):
        for other_n, other_r in ((n2, r2), (n3, r3)):
            for x, y in zip(n1.dcll_slices[i].dclllayer.i2h.state, other_n.dcll_slices[i].dclllayer.i2h.state):
                assert torch.equal(x, y), i
            assert torch.equal(r1["logits"][i], other_r["logits"][i]), i
            assert torch.equal(r1["clout"][i], other_r["clout"][i]) and torch.equal(r1["vote"][i], other_r["vote"][i])
            assert torch.equal(r1["lowhigh"][i], other_r["lowhigh"][i])
    assert torch.equal(r1["o"], r3["o"])
    # oracle on samples 0 and 1023, first 3 steps
    convs = load_network_spec(os.path.join(ROOT, "snn_modulation_classification_amd", "networks", "radio_ml_conv_ref.yaml"))
    sds = [{k: v.detach().cpu().numpy() for k, v in s.dclllayer.state_dict().items()} for s in n1.dcll_slices]
    orc = C.OracleConvNetwork(sds, convs, (H, W), 1.0)
    pick = [0, B - 1]
    cells = enc(iq, T, t0=0).cpu().numpy()
    for t in range(3):
        x = np.zeros((2, 1, H * W), np.float32)
        x[np.arange(2), 0, cells[t, pick]] = 1
        outs = orc.step(x.reshape(2, 1, H, W))
        for i in range(7):
            np.testing.assert_allclose(r1["logits"][i][t][pick].cpu().numpy(), outs[i]["p"], atol=LOGIT_TOL, rtol=0)


def test_sharded_evaluation_quantises_like_the_whole_batch(dev):
    """test_radio_ml.py under ranks: every rank runs test_sequence(iq=its shard, shard=(start, total)); the fused encoder
    must put each sample in the cell the reference's whole-batch quantiser would (positions 32..36 of a 37-window batch
    take torch's scalar pow path, whichever rank holds them): shard results == whole-batch results, bit for bit, on
    boundary-valued inputs."""
    from test_host_logic import _boundary_iq
    from snn_modulation_classification_amd.data.utils import IQEncoder
    B, T = 37, 21
    enc = IQEncoder(16, 16, device=dev)
    tabs = [enc.thr_i.cpu().numpy(), enc.thr_q.cpu().numpy()]
    if enc.thr_i_tail is not None:
        tabs += [enc.thr_i_tail.cpu().numpy(), enc.thr_q_tail.cpu().numpy()]
    iq = cu(_boundary_iq(B, 32, tabs, seed=5), dev)
    whole = _net("radio_ml_conv.yaml", (1, 16, 16), B, False)
    whole.reset()
    rw = whole.test_sequence(iq=iq, encoder=enc, T=T, t0=4)
    for lo, hi in ((0, 20), (20, 37)):
        part = _net("radio_ml_conv.yaml", (1, 16, 16), hi - lo, False)
        part.reset()
        rp = part.test_sequence(iq=iq[lo:hi].contiguous(), encoder=enc, T=T, t0=4, shard=(lo, B))
        for i in range(3):
            assert torch.equal(rp["clout"][i], rw["clout"][i][:, lo:hi]) and torch.equal(rp["vote"][i], rw["vote"][i][lo:hi])
            for x, y in zip(part.dcll_slices[i].dclllayer.i2h.state, whole.dcll_slices[i].dclllayer.i2h.state):
                assert torch.equal(x, y[lo:hi]), (lo, i)
