"""Randomised check of the readout GEMM entry points against float64 (not part of the test suite): row counts that are not
multiples of the 128-row workgroup tile (the rows past the end are out-of-range buffer reads = zeros), N = 1 .. 64 (every
column-tile count, ragged last tile), K = multiples of 32 up to 16384, with and without the sigmoid on the staged values, the
auto / 16x16x4 / LDS modes of dcll_readout and the split-K per-step form.   python experiments/readout_random.py [seed] [trials]"""


def main():
    import os, sys
    import numpy as np, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from snn_modulation_classification_amd import ops

    rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    dev = torch.device("cuda", 0)
    worst = 0.0
    for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 100):
        rows = int(rng.choice([1, 3, 127, 128, 129, 500, 1000, 2049, 5000, int(rng.randint(1, 9000))]))
        K = 32 * int(rng.choice([1, 2, 8, 64, 256, 512, int(rng.randint(1, 300))]))
        N = int(rng.choice([1, 10, 16, 17, 24, 32, 33, 48, 49, 64, int(rng.randint(1, 65))]))
        pv = torch.rand(rows, K, device=dev) * (4 if trial % 3 == 0 else 1) - (2 if trial % 3 == 0 else 0)
        W = (torch.rand(N, K, device=dev) - 0.5) * (2.0 / np.sqrt(K))
        b = (torch.rand(N, device=dev) - 0.5) * 0.1
        want = (pv.double() @ W.double().t() + b.double()).cpu().numpy()
        want_sig = (torch.sigmoid(pv.double()) @ W.double().t() + b.double()).cpu().numpy()
        for name, mode in (("auto", ops.READOUT_AUTO), ("t16", ops.READOUT_T16), ("lds", ops.READOUT_LDS)):
            got = ops.readout(pv, W, b, mode=mode).cpu().numpy()
            err = np.abs(got - want).max()
            worst = max(worst, err)
            assert err < 2e-5, (trial, name, rows, K, N, err)
        for act, ref in ((0, want), (1, want_sig)):
            try:
                got = ops.readout_act(pv, W, b, presigmoid=bool(act)).cpu().numpy()
            except NotImplementedError:
                continue
            err = np.abs(got - ref).max()
            worst = max(worst, err)
            assert err < 2e-5, (trial, "act", act, rows, K, N, err)
        print("trial %d ok: rows=%d K=%d N=%d" % (trial, rows, K, N))
    print("all trials agree; worst |err| vs float64 %.2e" % worst)


if __name__ == "__main__":
    main()
