"""k_lif_seq_c32p (persistent, pipeline carried across samples) against k_lif_seq_c32d (one workgroup per sample) on the
headline shape.   python experiments/c32p_timing.py [B]"""


def main():
    import os, sys
    sys.path.insert(0, os.getcwd())
    import numpy as np, torch
    from snn_modulation_classification_amd import ops
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    T, dev = 128, torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    W = ((torch.rand(32, 32, 7, 7, generator=g) - 0.5) * 2e-6).to(dev)
    b = ((torch.rand(32, generator=g) - 0.5) * 2e-4).to(dev)
    tau4 = torch.stack([torch.full((32,), v) for v in (0.95, 20.0, 0.85, 6.7)]).to(dev)
    spk_in = torch.randint(-2 ** 31, 2 ** 31 - 1, (T, B, 32, 8), generator=g, dtype=torch.int64).to(torch.int32).to(dev) & \
        torch.randint(-2 ** 31, 2 ** 31 - 1, (T, B, 32, 8), generator=g, dtype=torch.int64).to(torch.int32).to(dev) & 0x11111111
    d = ops.make_conv_desc(32, 32, (16, 16), 7, 3, 1, 24, False, True, 1.0)
    out = {"spk": torch.empty((T, B, 32, 8), device=dev, dtype=torch.int32), "pv": torch.empty((T, B, 32, 16, 16), device=dev)}
    res = {}
    for mode in ("0", "1", "0", "1"):
        os.environ["DCLL_C32_PERSISTENT"] = mode
        st = [torch.zeros((B, 32, 16, 16), device=dev) for _ in range(3)]
        ops.conv_lif_sequence(d, spk_in, W, b, tau4, *st, T, B, out=out)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            for s_ in st:
                s_.zero_()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.conv_lif_sequence(d, spk_in, W, b, tau4, *st, T, B, out=out)
            e.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(e))
        res.setdefault(mode, []).append(best)
        res["state" + mode] = [s_.clone() for s_ in st] + [out["spk"].clone()]
    ideal = 2.0 * 32 * 1568 * 256 * T * B / 157.3e12 * 1e3
    for mode, name in (("0", "k_lif_seq_c32d (one workgroup per sample)"), ("1", "k_lif_seq_c32p (persistent)")):
        ms = min(res[mode])
        print("%-44s B=%d T=%d: %.2f ms = %.1f %% of the fp32-MFMA peak (ideal %.2f ms)" % (name, B, T, ms, 100 * ideal / ms, ideal))
    print("identical results:", all(torch.equal(x, y) for x, y in zip(res["state0"], res["state1"])))


if __name__ == "__main__":
    main()
