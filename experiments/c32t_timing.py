"""Time k_lif_seq_c32t (large-plane fused layer) on one MI355X: python experiments/c32t_timing.py [H W B T]."""


def main():
    import sys
    import numpy as np
    import torch
    sys.path.insert(0, ".")
    from snn_modulation_classification_amd import ops

    H, Wd, B, T = (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (128, 128, 64, 16)))
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    W = ((torch.rand((32, 32, 7, 7), generator=g) - .5) * 2e-4).to(dev)
    b = ((torch.rand(32, generator=g) - .5) * 1e-3).to(dev)
    tau4 = torch.stack([torch.full((32,), .95), torch.full((32,), 20.), torch.full((32,), .9), torch.full((32,), 10.)]).to(dev)
    spk_in = torch.randint(-2**31, 2**31 - 1, (T, B, 32, H * Wd // 32), generator=g, dtype=torch.int64).to(torch.int32)
    spk_in = (spk_in & torch.randint(-2**31, 2**31 - 1, spk_in.shape, generator=g, dtype=torch.int64).to(torch.int32)).to(dev)
    d = ops.make_conv_desc(32, 32, (H, Wd), 7, 3, 1, 24, False, True, 1.0)
    st = [torch.zeros((B, 32, H, Wd), device=dev) for _ in range(3)]
    out = dict(spk=torch.empty((T, B, 32, H * Wd // 32), device=dev, dtype=torch.int32),
               pv=torch.empty((T, B, 32, H, Wd), device=dev))
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.conv_lif_sequence(d, spk_in, W, b, tau4, *st, T, B, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        fl = 2.0 * T * B * H * Wd * 32 * 1568
        print("plane %dx%d B=%d T=%d: %.2f ms  %.1f TFLOP/s (%.1f%% of 157.3)" % (H, Wd, B, T, ms, fl / ms / 1e9, fl / ms / 1e9 / 1.573))


if __name__ == "__main__":
    main()
