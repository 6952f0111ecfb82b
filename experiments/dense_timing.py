"""Timing of the dense twins: dcll_dense_lif_step at in = 8192, out = 512, B = 4096 (the shape the round-2
verdict names) and dcll_dense_lif_sequence on a small layer with the state on chip.   python experiments/dense_timing.py"""


def main():
    import os, sys, time
    sys.path.insert(0, os.getcwd())
    import numpy as np, torch
    from snn_modulation_classification_amd import ops
    from snn_modulation_classification_amd._lib import DenseDesc

    dev = torch.device("cuda", 0)


    def layer(cin, cout, B):
        g = torch.Generator().manual_seed(1)
        W = ((torch.rand(cout, cin, generator=g) - 0.5) * 2e-2 / np.sqrt(cin)).to(dev)
        b = ((torch.rand(cout, generator=g) - 0.5) * 2 / np.sqrt(cin) * 0.02).to(dev)
        tau = [torch.full((cin,), v, device=dev) for v in (0.95, 20.0, 0.85, 6.7)]
        st = [torch.zeros(B, cin, device=dev), torch.zeros(B, cin, device=dev), torch.zeros(B, cout, device=dev)]
        i2o = ((torch.rand(24, cout, generator=g) - 0.5) * 0.1).to(dev)
        return W, b, tau, st, i2o, torch.zeros(24, device=dev)


    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps


    cin, cout, B = 8192, 512, 4096
    W, b, tau, st, i2o, i2ob = layer(cin, cout, B)
    x = (torch.rand(B, cin, device=dev) < 0.1).float()
    d = DenseDesc(cin, cout, 24, 1, 1, .65, 1.0)
    ms = timed(lambda: ops.dense_lif_step(d, x, W, b, *tau, *st, i2o, i2ob), 10)
    flop = 2.0 * B * cout * cin
    print("dcll_dense_lif_step in=%d out=%d B=%d: %.3f ms per step (trace pass + MFMA GEMM + readout) = %.1f TFLOP/s = %.1f %% of the "
          "fp32-MFMA peak; state traffic %.0f MB per step" % (cin, cout, B, ms, flop / ms / 1e9, 100 * flop / ms / 1e9 / 157.3,
                                                             5 * B * cin * 4 / 1e6))
    if len(sys.argv) > 1 and sys.argv[1] == 'big':       # (under rocprofv3: only the large shape, so that the kernel table is its own)
        sys.exit(0)
    cin, cout, B, T = 512, 128, 8192, 64
    W, b, tau, st, i2o, i2ob = layer(cin, cout, B)
    xs = (torch.rand(T, B, cin, device=dev) < 0.1).float()
    d = DenseDesc(cin, cout, 24, 1, 1, .65, 1.0)
    ms = timed(lambda: ops.dense_lif_sequence(d, xs, W, b, *tau, *st, i2o, i2ob), 5)
    ms_step = timed(lambda: [ops.dense_lif_step(d, xs[t], W, b, *tau, *st, i2o, i2ob) for t in range(T)], 3)
    print("dcll_dense_lif_sequence in=%d out=%d B=%d T=%d (state on chip): %.3f ms per sequence = %.1f us per step; the same as T "
          "per-step calls: %.3f ms" % (cin, cout, B, T, ms, 1e3 * ms / T, ms_step))


if __name__ == "__main__":
    main()
