#!/usr/bin/env python
"""Headline benchmark: IQ windows/sec of the DCLL LIF timestep loop (radio_ml_conv.yaml, 16x16 I/Q plane, T=128).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B_per_gpu] [--plane R]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch of synthetic IQ windows already resident in HBM:
net.zero_states() + net.reset() -> all T steps of all three layers (fused sequence kernels; the IQ -> spike encoding
is fused into the first layer's kernel) -> readouts -> per-step argmax + vote -> per-class tallies (all-reduced).
That is the span of the reference's test_radio_ml.py:142-146 plus its input encoding (:133-135).
`--plane 128` runs the same path on the reference's argparse-default 128x128 I/Q plane (test_radio_ml.py:52; tiled
kernels k_lif_seq_c1t / k_lif_seq_c32t, default batch 64) — a secondary configuration, not the headline number.
Prints ONE JSON line on rank 0 (contract in the task statement) incl. `roofline` for the dominant kernel
(k_lif_seq_c32d, fp32 MFMA bound; `traffic` = its HBM bytes per launch measured in the run by two `rocprofv3 --pmc` child
passes) and `cpu_baseline` (the torch-CPU port of the reference timed on this host).
"""
import argparse
import json
import os
import sys
import time
from argparse import Namespace

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from snn_modulation_classification_amd import parallel  # noqa: E402
from snn_modulation_classification_amd.data.utils import IQEncoder  # noqa: E402
from snn_modulation_classification_amd.networks import ConvNetwork, load_network_spec  # noqa: E402

T_STEPS, L_IQ, N_CLASSES = 128, 128, 24
R = 16          # I/Q plane resolution; --plane overrides (16 = the reference scripts' setting = the headline config)
# algorithmic FLOPs of one k_lif_seq_c32 launch per sample per step per pixel: 2 * c_out * (c_in*7*7)
FLOP_C32_PER_SAMPLE_STEP_PIXEL = 2 * 32 * (32 * 49)
PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = fp32 vector rate
PEAK_HBM_GBS = 8000.0


def executed_frac(plane):
    """MFMA k-steps the 32->32 layer kernel issues / the algorithmic count.  16x16 (k_lif_seq_c32d, tiles of two image
    rows): the tap rows that lie in the zero padding for both rows of a border tile are not multiplied — 4 of the 56
    (tile, tap row) combinations of a sample.  Larger planes (k_lif_seq_c32t): everything is multiplied; skipping the 6 of
    56 padded (image row, tap row) combinations of the first / last tile row was built twice in round 4 and measured
    slower both times (experiments/rejected/c32t_padding_skip_variants.patch.txt)."""
    return 13.0 / 14.0 if plane == 16 else 1.0


def build_net(batch, device):
    convs = load_network_spec(os.path.join(ROOT, "snn_modulation_classification_amd", "networks",
                                           "radio_ml_conv.yaml"))
    args = Namespace(netscale=1.0, alpha=.92, alphas=.85, alpharp=.65, arp=1.0, lc_ampl=.5, random_tau=True)
    torch.manual_seed(1)
    np.random.seed(1)
    net = ConvNetwork(args, (1, R, R), batch, convs, N_CLASSES, act=torch.nn.Sigmoid(), loss=None, opt=None,
                      opt_param={}, learning_rates=None, burnin=20)
    net.reset(True)
    return net, convs


class _BusyBlocker:
    """Keeps the stream busy for a given time with matrix products (so that the host can enqueue a whole loop behind it).
    Not torch.cuda._sleep: while the device idles in a spin kernel its clocks come down, and the launches timed right
    behind it read 10-15 % slow (experiments/step_data_dependence.py)."""

    def __init__(self, dev):
        self.a = torch.randn(4096, 4096, device=dev)
        self.b = torch.randn(4096, 4096, device=dev)
        self.c = torch.empty(4096, 4096, device=dev)
        for _ in range(3):
            torch.mm(self.a, self.b, out=self.c)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            torch.mm(self.a, self.b, out=self.c)
        e1.record()
        torch.cuda.synchronize(dev)
        self.ms_per_mm = max(e0.elapsed_time(e1) / 10, 1e-3)

    def block(self, ms):
        for _ in range(int(ms / self.ms_per_mm) + 1):
            torch.mm(self.a, self.b, out=self.c)


def per_step_paths(dev, batch=512, steps=48, reps=3):
    """The reference's per-timestep protocol on the same network, reported beside the headline (not part of `value`):
    `net.test(x[t])` (test_radio_ml.py:142-146) and `net.learn(x[t], labels)` (train.py:249-251: SmoothL1Loss, Adam
    betas (0, .95), weight_decay 10) at the reference scripts' batch 512, per timestep over `steps` steps after the
    burn-in / warm-up, best of `reps` repetitions.  Two clocks per path, so that the record can tell a slow HOST from a
    slow DEVICE:
      wall_ms    host wall clock around the loop (+ a final synchronize): what a user of the loop sees;
      device_ms  HIP-event time of the same launches executing BACK TO BACK: the stream is first kept busy (matrix products,
                 _BusyBlocker) for about the wall time of the loop, the host enqueues all `steps` timesteps behind them, and
                 the events around them then bracket pure device work (no launch gaps).
    wall_ms / device_ms ~ 1: the device sets the pace; >> 1: the host's launch path does (then hipGraph replays help —
    ConvNetwork decides that by measurement per geometry, `graph_decision`)."""
    convs = load_network_spec(os.path.join(ROOT, "snn_modulation_classification_amd", "networks", "radio_ml_conv.yaml"))
    args = Namespace(netscale=1.0, alpha=.92, alphas=.85, alpharp=.65, arp=1.0, lc_ampl=.5, random_tau=True)
    torch.manual_seed(1)
    np.random.seed(1)
    net = ConvNetwork(args, (1, R, R), batch, convs, N_CLASSES, act=torch.nn.Sigmoid(), loss=torch.nn.SmoothL1Loss,
                      opt=torch.optim.Adam, opt_param={"betas": [0.0, .95], "weight_decay": 10.0},
                      learning_rates=[2.5e-10], burnin=2)
    net.reset(True)
    warm = 24                                  # burn-in, eager steps, the graph-vs-eager measurement of ConvNetwork
    n = steps + warm
    x = torch.zeros(n, batch, 1, R * R, device=dev)
    x.scatter_(3, torch.randint(0, R * R, (n, batch), device=dev)[:, :, None, None], 1.0)
    x = x.reshape(n, batch, 1, R, R)
    y = torch.zeros(batch, N_CLASSES, device=dev)
    y[torch.arange(batch), torch.randint(0, N_CLASSES, (batch,))] = 1
    blocker = _BusyBlocker(dev)
    out = {"batch": batch, "timesteps_timed": steps, "repetitions": reps}
    # A full pass of CPython's cyclic collector over this process's heap (~265 k tracked objects, most of them torch's
    # import-time ones) takes ~100 ms: landing in a 48-timestep window it reads as +2 ms per timestep (round-3 driver run:
    # learn 1.99 instead of 0.73 ms; experiments/per_step_outliers.py shows the collector's own clock).  The entry
    # points freeze the start-up heap (parallel.freeze_startup_heap): later full passes only walk what the loop allocated.
    parallel.freeze_startup_heap()
    for name, fn in (("test", lambda t: net.test(x[t])), ("learn", lambda t: net.learn(x[t], y))):
        net.reset()
        for t in range(warm):
            fn(t)
        torch.cuda.synchronize(dev)
        walls, devs = [], []
        for _ in range(reps):
            t0 = time.perf_counter()
            for t in range(warm, n):
                fn(t)
            torch.cuda.synchronize(dev)
            walls.append((time.perf_counter() - t0) / steps)
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            blocker.block(1.15 * min(walls) * steps * 1e3)                  # the host gets ahead of the device
            e0.record()
            for t in range(warm, n):
                fn(t)
            e1.record()
            torch.cuda.synchronize(dev)
            devs.append(e0.elapsed_time(e1) / steps)
        dt = min(walls)
        out[name + "_ms_per_timestep"] = 1e3 * dt
        out[name + "_wall_ms_per_timestep_all"] = [1e3 * w for w in walls]
        out[name + "_device_ms_per_timestep"] = min(devs)
        out[name + "_device_ms_per_timestep_all"] = devs
        out[name + "_wall_over_device"] = 1e3 * dt / max(min(devs), 1e-9)
        out[name + "_windows_per_s_at_T128"] = batch / (dt * 128)
    out["graph_decision"] = net.graph_decisions()
    # the MFMA kernel the per-step forward spends its time in, alone: `.forward` of a 32 -> 32 layer without its readouts =
    # one k_lif_step_c32 launch per call, launched back to back (106 us of device work per launch against ~10 us of host
    # work: the HIP events bracket device time); frac = algorithmic FLOPs of the layer step / time / fp32-MFMA peak
    i2h = net.dcll_slices[1].dclllayer.i2h
    spikes = (torch.rand(batch, 32, R, R, device=dev) < 0.05).float()
    obuf = {}                                   # reusable output maps: no allocation per call
    best = float("inf")
    for rep in range(4):                        # (the first repetitions also bring the clocks up: best of four)
        for _ in range(20):
            i2h._step(spikes, want_v=False, out=obuf)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(60):
            i2h._step(spikes, want_v=False, out=obuf)
        e1.record()
        torch.cuda.synchronize(dev)
        best = min(best, 1e3 * e0.elapsed_time(e1) / 60)
    us = best
    flop = FLOP_C32_PER_SAMPLE_STEP_PIXEL * R * R * batch
    out["k_lif_step_c32"] = {"us_per_launch": us, "batch": batch, "TFLOPs": flop / us / 1e6,
                             "frac_of_fp32_mfma_peak": flop / us / 1e6 / PEAK_FP32_MFMA_TFLOPS,
                             "what": "one layer step of a 32->32 7x7 layer on the 16x16 plane (dcll_conv_lif_step without "
                                     "readouts, s and pv written, state in HBM), 60 launches back to back, HIP events, best of 4"}
    return out


def run_ref_network(dev, B, steps, warmup, rank=0, world=1, validate=True):
    """BASELINE config 5 as this build defines it (quant.py; the reference has no quantisation code, so
    parity is unpinned): radio_ml_conv_ref.yaml — 7 x (64 channels, (1,3) kernels, (1,2) pooling) — on a Q=16 x I=128 I/Q
    plane, per-output-channel int8 conv weights handed to the kernels AS INT8 through the C ABI (dcll_layer_opts), 1-bit
    packed spikes between the layers, T=128, `B` windows per GPU.  Fused sequence path (k_lif_seq_w3 per layer; v in the pv
    buffer, the sigmoid in the readout GEMMs) — or, for a geometry without one, the per-step path.
    -> the fields of a bench line (value, ms_per_step, roofline of k_lif_seq_w3<64>, first-layer stream, ...)."""
    from snn_modulation_classification_amd import ops, quant
    H, W = 16, 128
    convs = load_network_spec(os.path.join(ROOT, "snn_modulation_classification_amd", "networks",
                                           "radio_ml_conv_ref.yaml"))
    args = Namespace(netscale=1.0, alpha=.92, alphas=.85, alpharp=.65, arp=1.0, lc_ampl=.5, random_tau=True)
    torch.manual_seed(1)
    np.random.seed(1)
    net = ConvNetwork(args, (1, H, W), B, convs, N_CLASSES, act=torch.nn.Sigmoid(), loss=None, opt=None, opt_param={},
                      learning_rates=None, burnin=20)
    net.reset(True)
    quant.apply_int8_weights(net)
    int8_abi = all(s.dclllayer.i2h.int8_weights() is not None for s in net.dcll_slices)
    # one pv buffer for the whole batch (33.5 MB per window: 137 GB at batch 4096 of the 288 GB): the narrow late layers
    # have only B / 8 ... B workgroups, chunks of the default 24 GB budget (767 windows) would leave the chip half empty
    net.pv_budget_bytes = float(os.environ.get("DCLL_PV_BUDGET_GB", "150")) * 2 ** 30
    enc = IQEncoder(W, H, device=dev)
    g = torch.Generator().manual_seed(1 + rank)
    iq = (0.4 * torch.randn(B, 2, L_IQ, generator=g)).to(dev)
    labels = torch.randint(0, N_CLASSES, (B,), generator=g).to(dev)
    fused = net.sequence_supported()

    prof, last = {}, {}

    def make_net(batch):
        torch.manual_seed(1)
        np.random.seed(1)
        n_ = ConvNetwork(args, (1, H, W), batch, convs, N_CLASSES, act=torch.nn.Sigmoid(), loss=None, opt=None, opt_param={},
                         learning_rates=None, burnin=20)
        n_.reset(True)
        quant.apply_int8_weights(n_)
        n_.pv_budget_bytes = net.pv_budget_bytes
        return n_

    def step():
        net.zero_states()
        net.reset()
        if fused:
            res = net.test_sequence(iq=iq, encoder=enc, T=T_STEPS, t0=0, collect=False, profile=prof)
            last["res"] = res
            return parallel.allreduce_tallies(parallel.tallies(res["vote"], labels, N_CLASSES))
        cells = enc(iq, T_STEPS, t0=0)
        planes = ops.cells_to_planes(cells, H * W)
        for t in range(T_STEPS):
            net.test(planes[t].reshape(B, 1, H, W))
        return None

    def fence():
        torch.cuda.synchronize()
        parallel.barrier()
        torch.cuda.synchronize()

    if parallel.is_distributed():
        parallel.all_reduce_(torch.zeros(1, device=dev))          # communicator up before the timed region
    for _ in range(warmup):
        step()
    fence()
    prof.clear()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if parallel.is_distributed():
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        parallel.all_reduce_(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    # the timed outputs are checked (untimed): 16 windows spread over the batch, run again on a network of the same seeds at
    # batch 16 — every wave of a sample co-resident, the regime the oracle-checked tests cover — must give the same votes in
    # all seven layers and the same output logits.  (Round 5's first-layer kernel passed every small test and was wrong for
    # 6 % of the samples at this batch: an in-place write-back raced with late waves — tests/test_gpu_kernels.py
    # test_sequence_w3_first_layer_grid_beyond_residency_with_carried_state.)
    validation = None
    if validate and fused and rank == 0 and "res" in last:
        try:
            pick = torch.unique(torch.linspace(0, B - 1, 16).round().long()).to(dev)
            small = make_net(len(pick))
            cells = enc(iq, T_STEPS, t0=0)
            small.zero_states()
            small.reset()
            r2 = small.test_sequence(cells[:, pick].contiguous(), collect=False)
            big = last["res"]
            validation = {
                "windows_checked": int(len(pick)),
                "votes_equal_per_layer": [bool(torch.equal(big["vote"][i][pick], r2["vote"][i])) for i in range(len(r2["vote"]))],
                "output_logits_max_abs_diff": float((big["o"][:, pick] - r2["o"]).abs().max()),
                "what": "windows spread over the batch re-run at batch %d on a same-seed network (fused path, all waves "
                        "co-resident) vs their results inside the timed batch of %d" % (len(pick), B)}
            del small, r2
        except Exception as e:                  # noqa: BLE001
            validation = {"error": "%s: %s" % (type(e).__name__, e)}
    flop = sum(2 * 64 * (1 if i == 0 else 64) * 3 * H * (W >> i) for i in range(7)) * T_STEPS * B
    # dominant kernel: k_lif_seq_w3<64> (layers 1..6), HIP-event time of its launches on the launch stream
    w3_ms = [s_.elapsed_time(e_) for s_, e_ in prof.get("lif_c32", [])]
    flop_w3 = sum(2 * 64 * 64 * 3 * H * (W >> i) for i in range(1, 7)) * T_STEPS * B * steps
    roof = None
    if w3_ms:
        ach = flop_w3 / (sum(w3_ms) / 1e3) / 1e12
        roof = {"kernel": "k_lif_seq_w3<64> (six 64->64 layers; sum over their launches)", "bound": "mfma", "achieved": ach,
                "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                "launch_ms_per_step": sum(w3_ms) / steps, "launches_per_step": len(w3_ms) / steps,
                "algorithmic_flop_per_step": flop_w3 / steps}
    if roof:
        # HBM traffic of the six launches: PMC counters cannot be read from inside this process — the summary of the separate
        # `rocprofv3 --pmc` passes of this command (profiles/collect_r05.sh ref) is used when it is for this batch
        for rnd in (6, 5, 4, 3):
            name = "r%02d_pmc_ref_b%d.json" % (rnd, B)
            try:
                with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)) as f:
                    ker = json.load(f)["kernels"]
                w3 = {k: v for k, v in ker.items() if k.startswith("k_lif_seq_w3<64")}
                if sum(v["launches"] for v in w3.values()) == 6:
                    roof["traffic"] = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in w3.values())
                    roof["traffic_unit"] = "HBM bytes per step over the six launches (2*FETCH_SIZE+WRITE_SIZE, rocprofv3 PMC)"
                    roof["traffic_source"] = ("profiles/%s (builder-side rocprofv3 --pmc passes of this command; not measured "
                                              "in this run)" % name)
                    break
            except (OSError, KeyError, ValueError):
                continue
    kernel_ms = {k: float(np.sum([s_.elapsed_time(e_) for s_, e_ in v])) / steps for k, v in prof.items()}
    hbm = {}
    if kernel_ms.get("lif_c1"):
        # first layer (k_lif_seq_w3f): an HBM write stream — pooled pv / v (T*B*64*16*64 floats) + packed pooled spikes
        byt = T_STEPS * B * 64 * H * (W // 2) * 4 * (1 + 1 / 32.0)
        hbm["first_layer (k_lif_seq_w3f + statistics pass)"] = {
            "bytes_per_launch": byt, "ms": kernel_ms["lif_c1"], "GBps": byt / kernel_ms["lif_c1"] / 1e6,
            "frac_of_peak": byt / kernel_ms["lif_c1"] / 1e6 / PEAK_HBM_GBS}
    if kernel_ms.get("readout"):
        byt = sum(T_STEPS * B * 64 * H * (W >> (i + 1)) * 4 for i in range(7))
        hbm["readouts (seven k_readout_t16 launches over the pooled maps)"] = {
            "bytes_per_step": byt, "ms": kernel_ms["readout"], "GBps": byt / kernel_ms["readout"] / 1e6,
            "frac_of_peak": byt / kernel_ms["readout"] / 1e6 / PEAK_HBM_GBS}
    return {
        "value": world * B * steps / dt, "unit": "IQ windows/s", "steps": steps, "warmup": warmup,
        "ms_per_step": 1e3 * dt / steps, "dtype": "f32 arithmetic on int8 weights (one fp32 scale per output channel)",
        "config": {"workload": "radio_ml_conv_ref.yaml (7 x 64 ch, (1,3) kernels, (1,2) pooling), Q=16 x I=128 I/Q plane, "
                               "T=128, int8 per-channel conv weights %s + 1-bit packed spikes, batch %d, %s; "
                               "parity unpinned (no reference quantisation code)" %
                               ("read as int8 by the kernels (dcll_layer_opts)" if int8_abi else "(dequantised)", B,
                                "fused sequence kernels" if fused else "per-step path (7 layer calls per timestep)"),
                   "batch_per_gpu": B, "global_batch": B * world, "T": T_STEPS, "plane": [H, W],
                   "path": "sequence" if fused else "per-step", "int8_weights_through_abi": int8_abi,
                   "pv_presigmoid": net.presigmoid != '0',
                   "parallelism": "batch shards, %d rank(s), tally all-reduce only" % world},
        "roofline": roof, "kernel_ms_per_step": kernel_ms, "hbm_bound_kernels": hbm, "validation": validation,
        "conv_tflops_per_gpu": flop * steps / dt / 1e12}


def bench_ref_network(a):
    """`bench.py --network ref [--gpus N]`: config 5 (run_ref_network) as a bench line of its own; batch shards like the
    headline benchmark.  (No CPU leg: the torch-CPU port needs ~1 s per window here.)"""
    rank, local_rank, world = parallel.init_process_group()
    assert world == a.gpus, "torchrun --nproc-per-node must equal --gpus (WORLD_SIZE=%d, --gpus %d)" % (world, a.gpus)
    dev = torch.device("cuda", parallel.local_device(local_rank))
    torch.cuda.set_device(dev)
    rec = run_ref_network(dev, a.batch or 1024, a.steps, a.warmup, rank, world, validate=bool(a.validate))
    if parallel.is_distributed():
        parallel.barrier()
    if rank == 0:
        out = {"metric": "IQ windows/sec (RadioML 2x128, T=128)", "n_gpus": world, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "data": "synthetic"}
        out.update(rec)
        print(json.dumps(out))
    if parallel.is_distributed():
        dist.destroy_process_group()


def sweep_point(dev, B, steps=3, warmup=1, T=None, cpu_windows=0):
    """One more batch size of the headline workload (BASELINE configs 2 and 3: batch 512 / 8192 on one MI355X) — or, with
    T, another sequence length (T = 1024 = the reference's n_iters default and script setting, train.py:63-66,
    scripts/test_radio_ml.sh:17-18; windows of max(128, T) samples): the same step as main()'s on a network of its own —
    value, ms per step and the roofline fraction of the hot kernel from the HIP events of its launches.  A batch above
    the pv budget runs in chunks (8192 = 6144 + 2048)."""
    T = T_STEPS if T is None else T
    L = max(L_IQ, T)
    net, _ = build_net(B, dev)
    enc = IQEncoder(R, R, device=dev)
    g = torch.Generator().manual_seed(11)
    iq = (0.4 * torch.randn(B, 2, L, generator=g)).to(dev)
    labels = torch.randint(0, N_CLASSES, (B,), generator=g).to(dev)
    prof = {}

    last = {}

    def step(profile=None):
        net.zero_states()
        net.reset()
        res = net.test_sequence(iq=iq, encoder=enc, T=T, t0=0, collect=False, profile=profile)
        last["vote"] = res["vote"][-1]
        return parallel.tallies(res["vote"], labels, N_CLASSES)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step(profile=prof)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    c32_ms = [s_.elapsed_time(e_) for s_, e_ in prof.get("lif_c32", [])]
    flop = FLOP_C32_PER_SAMPLE_STEP_PIXEL * R * R * T * B * 2 * steps          # both 32->32 layers, all steps
    ach = flop / (sum(c32_ms) / 1e3) / 1e12 if c32_ms else float("nan")
    rec = {"batch": B, "T": T, "value": B * steps / dt, "unit": "IQ windows/s", "steps": steps, "warmup": warmup,
           "ms_per_step": 1e3 * dt / steps, "timesteps_per_s": B * T * steps / dt,
           "roofline": {"kernel": "k_lif_seq_c32d" if R == 16 else "k_lif_seq_c32t", "bound": "mfma", "achieved": ach,
                        "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_MFMA_TFLOPS,
                        "launches": len(c32_ms), "launch_ms_total_per_step": sum(c32_ms) / steps},
           "kernel_ms_per_step": {k: float(np.sum([s_.elapsed_time(e_) for s_, e_ in v])) / steps for k, v in prof.items()}}
    if cpu_windows > 0:
        # the reference's CPU path on the first windows of this batch (one pass: seconds per window on planes this size),
        # same weights, same cells: its rate beside the GPU's, and whether the votes agree
        from oracle import torch_ref
        nw = min(cpu_windows, B)
        convs = load_network_spec(os.path.join(ROOT, "snn_modulation_classification_amd", "networks", "radio_ml_conv.yaml"))
        sds = [{k: v.detach().cpu() for k, v in s_.dclllayer.state_dict().items()} for s_ in net.dcll_slices]
        ref = torch_ref.RefConvNetwork(sds, convs, wrp=1.0)
        cells = enc(iq, T, t0=0)[:, :nw].cpu().long()
        x = torch.zeros(T, nw, R * R).scatter_(2, cells.unsqueeze(-1), 1.0).reshape(T, nw, 1, R, R)
        cores = min(16, usable_cores())
        torch.set_num_threads(cores)
        with torch.no_grad():
            ref.reset(True)
            t0 = time.perf_counter()
            for t in range(T):
                ref.test(x[t])
            votes = ref.votes()
            dtc = time.perf_counter() - t0
        rec["cpu_baseline"] = {"value": nw / dtc, "unit": "IQ windows/s", "cores": cores, "kind": "port",
                               "sample": "%d windows x T=%d, %dx%d plane, one pass of reset -> T x test(x[t]) -> votes (%.1f s), "
                                         "torch %s CPU" % (nw, T, R, R, dtc, torch.__version__),
                               "vote_agreement_with_gpu": float(np.mean(votes[-1] == last["vote"][:nw].cpu().numpy()))}
        rec["speedup_vs_cpu_baseline"] = rec["value"] / rec["cpu_baseline"]["value"]
    del net
    torch.cuda.empty_cache()
    return rec


def plane128_point(dev, batch=64, steps=2, cpu_windows=4):
    """The reference's ARGPARSE-DEFAULT plane (train.py:37-40, test_radio_ml.py:25-28: I_resolution = Q_resolution = 128;
    BASELINE.md section 3 plans it beside the scripts' 16x16): the headline workload on a 128x128 plane at batch 64 — the tiled
    kernels k_lif_seq_c1t / k_lif_seq_c32t (spatial tiles with a 3-pixel halo, state snapshot) — with its roofline fraction
    and a few windows of the CPU reference path."""
    global R
    saved, R = R, 128
    try:
        rec = sweep_point(dev, batch, steps=steps, warmup=1, cpu_windows=cpu_windows)
        rec["plane"] = [128, 128]
        rec["roofline"]["executed_over_algorithmic_flop"] = executed_frac(128)
        return rec
    finally:
        R = saved


def trained_top1(dev, n_batches=4, batch=512, train_steps=25):
    """north_star's "top-1 within 0.1 % of reference" on a network that has learned (oracle/trained_parity.py): train
    radio_ml_conv.yaml with train.py on the synthetic modulation set, restore the checkpoint the reference's way, evaluate
    the same held-out windows on the fused HIP path and on the CPU reference path (oracle/torch_ref.py)."""
    import tempfile
    from oracle import trained_parity
    torch.set_num_threads(max(1, min(16, usable_cores())))        # (the CPU leg: an over-subscribed pool is far slower)
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.perf_counter()
        ckpt = trained_parity.train_checkpoint(tmp, steps=train_steps, batch=batch)
        t_train = time.perf_counter() - t0
        net, ref, _, enc = trained_parity.restore_pair(ckpt, batch, device=dev)
    rep = trained_parity.evaluate(net, ref, enc, trained_parity.held_out_batches(n_batches, batch), count_flips=True, log=log)
    rep["training"] = ("train.py --synthetic: %d batches of %d windows, T=128, 16x16, arp 1.0, burn-in 20, SmoothL1 + Adam "
                       "(lr 1e-6; output_ 1e-4), %.1f s incl. the two evaluations train.py runs" % (train_steps, batch, t_train))
    rep["held_out"] = "%d x %d synthetic modulation windows, SNR cycling 6 .. 30 dB, zero neuron state per batch" % (n_batches, batch)
    return rep


def live_hbm_traffic(extra_args, kernel_prefix, timeout=300):
    """HBM bytes per launch of the dominant kernel, MEASURED IN THIS RUN: two child passes of this very command (1 step) under
    `rocprofv3 --pmc` — FETCH_SIZE and WRITE_SIZE in separate passes, no trace domain beside them, the program itself behind
    `--` (MI355X_MICROARCH.md, HBM / rocprofv3 section) — and the guide's gfx950 correction: bytes = (2 * FETCH_SIZE +
    WRITE_SIZE) KiB (wide coalesced reads are tallied at half).  The children are ordinary child processes (no exec from a
    process that has touched the GPU).  -> (bytes per launch or None, how it was obtained)"""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="dcll_pmc_", dir="/tmp")
    vals = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = [exe, "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                   os.path.abspath(__file__)] + list(extra_args) + [
                   "--steps", "1", "--warmup", "0", "--cpu-windows", "0", "--per-step", "0", "--config5", "0", "--batch-sweep", "0",
                   "--trained", "0", "--live-traffic", "0", "--t1024", "0", "--plane128", "0", "--validate", "0"]
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE",
                                                                      "MASTER_ADDR", "MASTER_PORT", "DCLL_FORCE_DIST")}
            env["TMPDIR"] = "/tmp"
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout)
            if r.returncode != 0:
                return None, "rocprofv3 --pmc %s pass failed (rc %d): %s" % (ctr, r.returncode, r.stderr.decode("utf-8", "replace")[-200:])
            tot, disp = 0.0, set()
            for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(fn)):
                    if row["Counter_Name"] == ctr and row["Kernel_Name"].replace("void ", "").startswith(kernel_prefix):
                        tot += float(row["Counter_Value"])
                        disp.add(row["Dispatch_Id"])
            if not disp:
                return None, "no %s dispatch in the %s pass" % (kernel_prefix, ctr)
            vals[ctr] = (tot / len(disp), len(disp))
    except (OSError, subprocess.TimeoutExpired, KeyError, ValueError) as e:
        return None, "%s: %s" % (type(e).__name__, e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    byt = (2.0 * vals["FETCH_SIZE"][0] + vals["WRITE_SIZE"][0]) * 1024.0
    return byt, ("measured in this run: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE as two child passes of this command with --steps 1 "
                 "(%d / %d %s launches averaged), HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB" %
                 (vals["FETCH_SIZE"][1], vals["WRITE_SIZE"][1], kernel_prefix))


def log(msg):
    print("[bench %7.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def usable_cores():
    """CPU threads this process may really use: affinity mask, capped by a cgroup-v2 quota if one is set."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(net, convs, cells_cpu, gpu_votes, n_windows, gpu_spikes=None):
    """The reference's CPU PyTorch path (oracle/torch_ref.py: same eager op sequence, fixture-verified against the
    imported reference) on this host's cores: reset -> T-loop of test(x[t]) -> votes, on a bounded sample.
    gpu_spikes: packed spike trains of all layers of the GPU run on the same windows -> `spike_flips_vs_cpu` (flip counts,
    first flips inside the rounding band; oracle/flip_count.py)."""
    from oracle import flip_count, torch_ref
    sds = [{k: v.detach().cpu() for k, v in s.dclllayer.state_dict().items()} for s in net.dcll_slices]
    ref = torch_ref.RefConvNetwork(sds, convs, wrp=1.0)
    cells = cells_cpu[:, :n_windows].long()
    T = cells.shape[0]
    x = torch.zeros(T, n_windows, R * R)
    x.scatter_(2, cells.unsqueeze(-1), 1.0)
    x = x.reshape(T, n_windows, 1, R, R)
    with torch.no_grad():
        # pick the thread count that is fastest on this host (an over-subscribed OpenMP pool is far slower)
        cand = sorted({c for c in (8, 16, 32, 64, usable_cores()) if c <= usable_cores()} or {1})
        best = None
        for c in cand:
            torch.set_num_threads(c)
            ref.reset(True)
            ref.test(x[0])                  # warm up oneDNN primitives
            t0 = time.perf_counter()
            for t in range(1, 4):
                ref.test(x[t])
            dtc = time.perf_counter() - t0
            log("cpu baseline calibration: %d threads -> %.3f s / 3 steps" % (c, dtc))
            if best is None or dtc < best[1]:
                best = (c, dtc)
        cores = best[0]
        torch.set_num_threads(cores)
        dts = []
        for rep in range(3):                # best of 3 (best of >= 2), about 10 s of CPU work
            ref.reset(True)                 # start from zero state again
            t0 = time.perf_counter()
            for t in range(T):
                ref.test(x[t])
            votes = ref.votes()
            dts.append(time.perf_counter() - t0)
            log("cpu baseline rep %d: %.2f s for %d windows" % (rep, dts[-1], n_windows))
        dt = min(dts)
        flips = None
        if gpu_spikes is not None:          # one more (untimed) pass: every spike of the three layers, CPU vs GPU
            ref.reset(True)
            flips = flip_count.spike_flips(ref, x, [flip_count.unpack_words(s_[:, :n_windows], (R, R)) for s_ in gpu_spikes])
            log("spike flips vs the CPU path on %d windows: %s" % (n_windows, flips["flips_per_layer"]))
    agree = float(np.mean(votes[-1] == gpu_votes[:n_windows]))
    return {"value": n_windows / dt, "unit": "IQ windows/s", "cores": cores, "kind": "port", "spike_flips_vs_cpu": flips,
            "sample": "batch of %d windows x T=%d, %dx%d plane, reset -> T x test(x[t]) -> votes, best of 3 (%.1f s "
                      "each), torch %s CPU, %d threads (fastest of the calibrated counts <= cgroup quota)" %
                      (n_windows, T, R, R, dt, torch.__version__, torch.get_num_threads()),
            "vote_agreement_with_gpu": agree}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=None,
                    help="IQ windows per GPU per step (weak scaling); default 4096 (16x16 plane) / 64 (larger planes)")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="STRONG scaling: IQ windows per step over ALL GPUs (north_star: 'batch 4096 ... at 1/2/4/8 GPUs'); "
                         "every rank runs its contiguous shard of G / N windows and the JSON line says \"scaling\": "
                         "\"strong\".  Default: weak scaling with --batch windows per GPU")
    ap.add_argument("--plane", type=int, default=16, help="I/Q plane resolution R (R x R cells): 16 or a multiple of 32")
    ap.add_argument("--steps-T", type=int, default=128, dest="steps_T",
                    help="timesteps per window T (IQ windows of max(128, T) samples).  128 = BASELINE's metric; 1024 = the "
                         "reference's n_iters default and script setting (train.py:63-66, scripts/test_radio_ml.sh:17-18)")
    ap.add_argument("--t1024", type=int, default=1,
                    help="1 (default, N=1 headline run only): also run T = 1024 at batch 512 (2 steps) and report it as `t1024`")
    ap.add_argument("--batch-sweep", type=int, default=1,
                    help="1 (default, N=1 headline run only): also run BASELINE configs 2 and 3 (batch 512 and 8192, 3 steps "
                         "each) and report them as `batch_sweep`")
    ap.add_argument("--live-traffic", type=int, default=1,
                    help="1 (default, N=1 only): measure roofline.traffic in this run — two child passes of this command under "
                         "rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE; ~30 s); 0 or on failure: the committed builder-side summary")
    ap.add_argument("--trained", type=int, default=1,
                    help="1 (default, N=1 headline run only): also train the network on the synthetic modulation set (train.py, "
                         "25 batches of 512) and compare top-1 of the fused path with the CPU reference path on 2048 held-out "
                         "windows: `trained_top1`")
    ap.add_argument("--cpu-windows", type=int, default=None,
                    help="CPU baseline batch (0 = skip); default 512 = batch_size_test of the reference's scripts "
                         "(4 on planes larger than 16x16, where the CPU path needs seconds per window)")
    ap.add_argument("--fuse-readout", type=int, default=0, help="1: readouts in the layer kernel's epilogue")
    ap.add_argument("--output-only", type=int, default=0,
                    help="1: serving mode — hidden layers skip pv and their local readouts (NOT the headline workload)")
    ap.add_argument("--overlap-readout", type=int, default=None,
                    help="1: readouts / statistics / votes on a second stream under the next layer's kernel "
                         "(default: DCLL_OVERLAP_READOUT)")
    ap.add_argument("--per-step", type=int, default=1,
                    help="1 (default, N=1 only): also time the per-timestep protocol net.test / net.learn at batch 512 and "
                         "report it as `per_step_paths` (a second or two; not part of `value`)")
    ap.add_argument("--config5", type=int, default=1,
                    help="1 (default, N=1 headline run only): also run BASELINE config 5 (radio_ml_conv_ref.yaml, int8 weights "
                         "through the ABI, batch 4096, 3 steps) and report it as the `config5` sub-record")
    ap.add_argument("--plane128", type=int, default=1,
                    help="1 (default, N=1 headline run only): also run the reference's argparse-default 128x128 plane at batch 64 "
                         "(2 steps + 4 windows of the CPU path) and report it as the `plane128` sub-record")
    ap.add_argument("--validate", type=int, default=1,
                    help="1 (default): config 5 re-runs 16 windows of its timed batch at batch 16 and compares (0: the PMC child "
                         "passes — their per-launch averages must see the timed launches only)")
    ap.add_argument("--network", default="radio", choices=["radio", "ref"],
                    help="radio = radio_ml_conv.yaml (headline); ref = radio_ml_conv_ref.yaml with int8 weights (config 5)")
    a = ap.parse_args()
    if a.gpus > 1 and not parallel.under_launcher():
        # started plainly (`python bench.py --gpus N`): this process becomes the launcher of N fresh rank processes and
        # never touches the GPU itself; rank 0's JSON line goes straight to our stdout.  Under torchrun the ranks
        # arrive here with RANK / WORLD_SIZE set and fall through.
        sys.exit(parallel.spawn_local_ranks(a.gpus))
    if a.network == "ref":
        return bench_ref_network(a)
    global R, T_STEPS, L_IQ
    R = a.plane
    T_STEPS, L_IQ = a.steps_T, max(128, a.steps_T)
    if a.batch is None:
        a.batch = 4096 if R == 16 else 64
    if a.cpu_windows is None:
        a.cpu_windows = 512 if R == 16 else 4
    hot_kernel = "k_lif_seq_c32d" if R == 16 else "k_lif_seq_c32t"

    rank, local_rank, world = parallel.init_process_group()
    assert world == a.gpus, "torchrun --nproc-per-node must equal --gpus (WORLD_SIZE=%d, --gpus %d)" % (world, a.gpus)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    dev = torch.device("cuda", parallel.local_device(local_rank))
    torch.cuda.set_device(dev)
    B = a.batch
    strong = a.global_batch is not None
    if strong:
        lo, hi = parallel.shard_range(a.global_batch, rank, world)
        B = hi - lo
        assert B > 0, "--global-batch %d leaves rank %d of %d without windows" % (a.global_batch, rank, world)
    total_windows = a.global_batch if strong else world * B

    net, convs = build_net(B, dev)
    enc = IQEncoder(R, R, device=dev)
    shard = None
    if strong:
        # a FIXED global batch: the same windows whatever the rank count, every rank takes its contiguous slice — the
        # all-reduced tallies of an N-rank run then equal the single-process run's (tests/test_gpu_multirank.py)
        g = torch.Generator().manual_seed(1)
        iq_all = 0.4 * torch.randn(a.global_batch, 2, L_IQ, generator=g)
        labels_all = torch.randint(0, N_CLASSES, (a.global_batch,), generator=g)
        iq, labels = iq_all[lo:hi].contiguous().to(dev), labels_all[lo:hi].contiguous().to(dev)
        shard = (lo, a.global_batch)        # (the quantiser treats a sample as the reference would at ITS batch position)
        del iq_all, labels_all
    else:
        g = torch.Generator().manual_seed(1 + rank)          # weak scaling: every rank its own synthetic batch
        iq = (0.4 * torch.randn(B, 2, L_IQ, generator=g)).to(dev)
        labels = torch.randint(0, N_CLASSES, (B,), generator=g).to(dev)
    prof = {}

    def step(profile=None):
        # raw IQ in HBM -> (encoder fused into the first layer's kernel) -> three layers -> readouts -> votes -> tallies
        net.zero_states()
        net.reset()
        res = net.test_sequence(iq=iq, encoder=enc, T=T_STEPS, t0=0, collect=False, profile=profile,
                                fuse_readout=bool(a.fuse_readout), output_only=bool(a.output_only),
                                overlap_readout=None if a.overlap_readout is None else bool(a.overlap_readout), shard=shard)
        votes = [v if v is not None else res["vote"][-1] for v in res["vote"]]      # output_only: hidden layers have none
        tal = parallel.tallies(votes, labels, N_CLASSES)
        if profile is not None and parallel.is_distributed():
            # the step's only collective, bracketed by events on the launch stream (the backend's own stream is joined to it
            # before the call returns): a sub-6x scaling result can then name the collective — or rule it out
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            tal = parallel.allreduce_tallies(tal)
            e1.record()
            profile.setdefault("allreduce", []).append((e0, e1))
        else:
            tal = parallel.allreduce_tallies(tal)
        return res, tal

    def fence():
        torch.cuda.synchronize()
        parallel.barrier()
        torch.cuda.synchronize()

    net._sequence_buffers(T_STEPS, min(B, max(1, int(net.pv_budget_bytes // (4 * T_STEPS * 32 * R * R)))), dev)  # allocate once, outside the timed region
    if parallel.is_distributed():
        # create the RCCL communicator now (lazy otherwise: it would land in the first step, timed when --warmup 0)
        parallel.all_reduce_(torch.zeros(1, device=dev))
        torch.cuda.synchronize()
    log("rank %d/%d: network built, B=%d per GPU" % (rank, world, B))
    for _ in range(a.warmup):
        step()
    fence()
    log("warmup done")
    t0 = time.perf_counter()
    for _ in range(a.steps):
        res, tal = step(profile=prof)
    fence()
    dt = time.perf_counter() - t0
    log("timed region done: %.3f s for %d steps" % (dt, a.steps))
    # the same K steps with the raw IQ handed over as a HOST buffer each step ("with and without
    # encode+upload", reference test_radio_ml.py:133-135 — there a dense T*B*R*R fp32 spike tensor, here 1 KB of raw IQ per
    # window from pinned memory; the encoding runs inside the first layer's kernel either way).  Never `value`.
    iq_host = iq.cpu().pin_memory()

    def step_upload():
        iq.copy_(iq_host, non_blocking=True)
        return step()

    step_upload()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step_upload()
    fence()
    dt_up = time.perf_counter() - t0
    multi = None
    if parallel.is_distributed():
        # max over ranks = the job's time (contract); min / max of the ranks' own clocks, of the device time of their layer
        # kernels and of the collective say WHERE a scaling loss comes from: a straggler rank (per_rank spread), the
        # collective (allreduce_ms_per_step), or the host (wall >> device_busy on every rank)
        ar_ms = float(np.sum([s_.elapsed_time(e_) for s_, e_ in prof.get("allreduce", [])])) / max(1, a.steps)
        busy_ms = float(np.sum([s_.elapsed_time(e_) for k_, v_ in prof.items() if k_ != "allreduce" for s_, e_ in v_])) / max(1, a.steps)
        dt, dt_up, multi = parallel.job_timing(dt, dt_up, ar_ms, busy_ms, a.steps, dev, "%s/cuda:%d/%s" % (
            os.uname().nodename, dev.index, parallel.device_identity(dev.index)))
    log("upload-inclusive region done: %.3f s for %d steps" % (dt_up, a.steps))

    # dominant kernel: HIP-event time of every k_lif_seq_c32d launch of the timed region (same stream as the launch)
    c32_ms = [s.elapsed_time(e) for s, e in prof.get("lif_c32", [])]
    avg_c32_s = float(np.mean(c32_ms)) / 1e3 if c32_ms else float("nan")
    # two 32->32 layers per step; a batch above the pv budget runs in chunks (more, smaller launches)
    flop_per_launch = FLOP_C32_PER_SAMPLE_STEP_PIXEL * R * R * T_STEPS * B * 2 * a.steps / max(1, len(c32_ms))
    achieved = flop_per_launch / avg_c32_s / 1e12
    # HBM bytes of the dominant kernel: PMC counters cannot be read from inside this process; the committed summary of
    # the separate `rocprofv3 --pmc` passes of this same command (profiles/r04_pmc_b4096.json; collect_r04.sh) is used
    # when the batch matches, else null.
    traffic, traffic_src = None, None
    for name in ([] if T_STEPS != 128 else ["r%02d_pmc_b%d.json" % (r_, B) for r_ in (6, 5, 4, 3, 2, 1)] if R == 16 else
                 ["r%02d_pmc_plane%d_b%d.json" % (r_, R, B) for r_ in (6, 5, 4, 3, 2, 1)]):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                pmc = json.load(f)
            if R == 16:
                if pmc.get("k_lif_seq_c32_batch") != B:
                    continue
                traffic = pmc["k_lif_seq_c32_traffic_bytes_per_launch"]
            else:
                traffic = [v["hbm_bytes_per_launch"] for k, v in pmc["kernels"].items() if k.startswith("k_lif_seq_c32t")][0]
            traffic_src = "profiles/%s (builder-side rocprofv3 --pmc passes of this command; not measured in this run)" % name
            break
        except (OSError, ValueError, KeyError, IndexError):
            continue
    if rank == 0 and world == 1 and a.live_traffic:
        t_pm = time.perf_counter()
        live, how = live_hbm_traffic(["--batch", str(B), "--plane", str(R), "--steps-T", str(T_STEPS)], hot_kernel)
        log("live HBM traffic of %s: %s (%s; %.0f s)" % (hot_kernel, live, how[:60], time.perf_counter() - t_pm))
        if live is not None:
            if traffic is not None:
                how += "; builder-side summary %s: %.4g" % (traffic_src.split(" ")[0], traffic)
            traffic, traffic_src = live, how
        elif traffic_src is not None:
            traffic_src += " [live measurement unavailable: %s]" % how
    kernel_ms = {k: float(np.mean([s.elapsed_time(e) for s, e in v])) for k, v in prof.items()}
    if len(c32_ms) >= 2:        # the two 32->32 layers of a step (the output layer carries a second readout)
        kernel_ms["lif_c32_layer1"] = float(np.mean(c32_ms[0::2]))
        kernel_ms["lif_c32_layer2"] = float(np.mean(c32_ms[1::2]))
    cm, acc = parallel.split_tallies(tal, N_CLASSES)

    if rank == 0:
        out = {
            "metric": "IQ windows/sec (RadioML 2x%d, T=%d)" % (L_IQ, T_STEPS), "value": total_windows * a.steps / dt,
            "unit": "IQ windows/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "radio_ml_conv.yaml, %dx%d I/Q plane, T=%d, arp=1.0, random_tau, %s%s, "
                                   "synthetic IQ 0.4*randn(B,2,%d), seeded init" %
                                   (R, R, T_STEPS, ("global batch %d sharded over %d GPU(s) (strong scaling)" % (total_windows, world))
                                    if strong else ("batch %d per GPU" % B),
                                    (" (north_star headline batch)" if ((total_windows if strong else B) == 4096 and R == 16)
                                     else "") +
                                    (", OUTPUT-ONLY serving mode (hidden-layer readouts skipped)" if a.output_only else ""), L_IQ),
                       "batch_per_gpu": B, "global_batch": total_windows, "T": T_STEPS, "plane": [R, R],
                       "parallelism": "batch shards, %d rank(s) on %d GPU(s)%s, tally all-reduce only (backend %s)" %
                                      (world, min(world, torch.cuda.device_count()),
                                       " — REHEARSAL: ranks share a device" if world > torch.cuda.device_count() else "",
                                       dist.get_backend() if parallel.is_distributed() else "none")},
            "roofline": {"kernel": hot_kernel, "bound": "mfma", "achieved": achieved,
                         "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_unit": "HBM bytes per launch (2*FETCH_SIZE+WRITE_SIZE, rocprofv3 PMC)",
                         "avg_launch_ms": avg_c32_s * 1e3, "launches": len(c32_ms),
                         "algorithmic_flop_per_launch": flop_per_launch,
                         # the kernels skip tap rows that lie wholly in the zero padding (fmaf(w, 0, acc) == acc;
                         # executed_frac): what the matrix pipe executes is less than the algorithmic count, so `frac`
                         # (algorithmic) may exceed what the pipe alone allows — the pipe's own busy fraction
                         # is frac * executed / algorithmic
                         "frac_definition": "algorithmic FLOPs of SURVEY 8(d) per launch / HIP-event launch time / peak (the task's "
                                            "definition of `achieved`); the kernel ISSUES executed_over_algorithmic_flop of them — "
                                            "matrix_pipe_frac = frac * that ratio is how busy the matrix pipe really is",
                         "executed_over_algorithmic_flop": executed_frac(R),
                         "matrix_pipe_frac": achieved / PEAK_FP32_MFMA_TFLOPS * executed_frac(R),
                         "hbm": {"achieved_GBps": (traffic / avg_c32_s / 1e9) if traffic else None,
                                 "peak_GBps": PEAK_HBM_GBS,
                                 "frac": (traffic / avg_c32_s / 1e9 / PEAK_HBM_GBS) if traffic else None}},
            "kernel_ms_per_launch": kernel_ms,
            "vote_accuracy_vs_random_labels": [float(x) for x in acc.cpu()],
            # the all-reduced tallies of the last step: (correct, total) per layer and a digest of the confusion matrices
            "tallies": {"correct_total": [[int(v) for v in row] for row in tal[:, -2:].cpu()],
                        "confusion_sha1": __import__("hashlib").sha1(cm.cpu().numpy().astype(np.int64).tobytes()).hexdigest()},
            "value_incl_upload": total_windows * a.steps / dt_up,
            "ms_per_step_incl_upload": 1e3 * dt_up / a.steps,
            "upload": "raw IQ (B,2,%d) fp32 = %d bytes per step per GPU, pinned host memory -> HBM on the launch stream, in "
                      "front of every step; `value` is the device-resident form" % (L_IQ, B * 2 * L_IQ * 4),
        }
        # the HBM-bound kernels around the hot one (north_star: achieved HBM GB/s of the LIF-update kernel against the
        # roofline): bytes they must move by this design (pv written once by the first layer's kernel, read once by a
        # readout GEMM; + packed spikes / logits), over their HIP-event time of this run
        chunk_b = B * a.steps / max(1, len(prof.get("lif_c1", [])))         # windows per launch (chunked batches)
        pv_bytes = T_STEPS * chunk_b * 32 * R * R * 4
        hb = {}
        if "lif_c1" in kernel_ms:
            byt = pv_bytes * (1 + 1 / 32.0)                                   # pv + packed spikes out
            hb["lif_c1 (k_lif_seq_c1 + pv statistics pass)"] = {
                "bytes_per_launch": byt, "ms": kernel_ms["lif_c1"], "GBps": byt / kernel_ms["lif_c1"] / 1e6,
                "frac_of_peak": byt / kernel_ms["lif_c1"] / 1e6 / PEAK_HBM_GBS,
                "note": "bound by the shared matrix / vector pipe, not by this stream (DESIGN.md 4.2)"}
        if "readout" in kernel_ms:
            hb["readout (k_readout_t16, mean of the 24- and 48-row launches)"] = {
                "bytes_per_launch": pv_bytes, "ms": kernel_ms["readout"], "GBps": pv_bytes / kernel_ms["readout"] / 1e6,
                "frac_of_peak": pv_bytes / kernel_ms["readout"] / 1e6 / PEAK_HBM_GBS}
        out["hbm_bound_kernels"] = hb
        if multi is not None:
            out["multi_gpu"] = multi
        if world == 1 and R == 16 and a.per_step and T_STEPS == 128:
            try:                                    # an extra, never at the price of the headline line
                out["per_step_paths"] = per_step_paths(dev)
            except Exception as e:                  # noqa: BLE001
                out["per_step_paths"] = {"error": "%s: %s" % (type(e).__name__, e)}
        head = world == 1 and R == 16 and B == 4096 and T_STEPS == 128        # the headline run: extras beside it
        if head and a.t1024:
            # the reference's own sequence length (n_iters = n_iters_test = 1024) at its scripts' batch 512
            try:
                net._seq_buffers.clear()
                torch.cuda.empty_cache()
                out["t1024"] = sweep_point(dev, 512, steps=2, warmup=1, T=1024)
                log("T = 1024 at batch 512: %.0f windows/s, frac %.3f" % (out["t1024"]["value"], out["t1024"]["roofline"]["frac"]))
            except Exception as e:                  # noqa: BLE001
                out["t1024"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if head and a.batch_sweep:
            # BASELINE configs 2 and 3 beside the headline (never at its price)
            try:
                net._seq_buffers.clear()
                torch.cuda.empty_cache()
                out["batch_sweep"] = [sweep_point(dev, b_, steps=8 if b_ <= 512 else 3, warmup=4 if b_ <= 512 else 1) for b_ in (512, 8192)]
                log("batch sweep done: %s" % [(r_["batch"], round(r_["value"])) for r_ in out["batch_sweep"]])
            except Exception as e:                  # noqa: BLE001
                out["batch_sweep"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if head and a.plane128:
            try:
                net._seq_buffers.clear()
                torch.cuda.empty_cache()
                t_pl = time.perf_counter()
                out["plane128"] = plane128_point(dev, cpu_windows=4 if a.cpu_windows > 0 else 0)
                log("128x128 plane at batch 64: %.0f windows/s, %s frac %.3f (%.0f s incl. the CPU leg)" % (
                    out["plane128"]["value"], out["plane128"]["roofline"]["kernel"], out["plane128"]["roofline"]["frac"],
                    time.perf_counter() - t_pl))
            except Exception as e:                  # noqa: BLE001
                out["plane128"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if head and a.config5:
            # BASELINE config 5 beside the headline (never at its price): its own network, 3 steps at batch 4096
            try:
                net._seq_buffers.clear()            # the headline's pv / spike buffers: config 5 needs 137 GB of pv
                torch.cuda.empty_cache()
                out["config5"] = run_ref_network(dev, 4096, 3, 1)
            except Exception as e:                  # noqa: BLE001
                out["config5"] = {"error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.empty_cache()
            roof5 = out["config5"].get("roofline") if isinstance(out["config5"], dict) else None
            if roof5 and a.live_traffic:            # its HBM traffic measured in this run as well (children: 137 GB of pv each)
                live, how = live_hbm_traffic(["--network", "ref", "--batch", "4096"], "k_lif_seq_w3<64")
                log("live HBM traffic of the six k_lif_seq_w3<64> launches: %s (%s)" % (live, how[:50]))
                if live is not None:
                    roof5["traffic"] = live * roof5["launches_per_step"]
                    roof5["traffic_source"] = how + "; x %g launches per step" % roof5["launches_per_step"]
        if head and a.trained and a.cpu_windows > 0:
            try:
                out["trained_top1"] = trained_top1(dev)
            except Exception as e:                  # noqa: BLE001
                out["trained_top1"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and a.cpu_windows > 0:
            cells = enc(iq, T_STEPS, t0=0)          # the same quantisation as a separate kernel, for the CPU leg
            nw = min(a.cpu_windows, B)
            spikes = None
            if R == 16:                             # the GPU's spike trains of the same windows, for the flip count
                # (one more untimed pass over the WHOLE batch: a different batch size would re-allocate the state, and the
                #  refractory variant re-draws its time constants whenever it does — reference quirk Q4)
                net.zero_states()
                net.reset()
                sub = net.test_sequence(iq=iq, encoder=enc, T=T_STEPS, t0=0, collect=False, keep_spikes=True)
                spikes = [s_[:, :nw].cpu().numpy() for s_ in sub["spikes"]]
                del sub
            out["cpu_baseline"] = cpu_baseline(net, convs, cells.cpu(), res["vote"][-1].cpu().numpy(), nw, spikes)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if parallel.is_distributed():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
