

def main():
    import sys, torch
    sys.path.insert(0, "/root/repo")
    sys.path.insert(0, ".")
    from snn_modulation_classification_amd import ops
    dev = torch.device("cuda:0")
    B = 1024
    g = torch.Generator().manual_seed(0)
    W = ((torch.rand((32, 32, 7, 7), generator=g) - .5) * 2e-4).to(dev)
    b = ((torch.rand(32, generator=g) - .5) * 1e-3).to(dev)
    tau4 = torch.stack([torch.full((32,), .95), torch.full((32,), 20.), torch.full((32,), .9), torch.full((32,), 10.)]).to(dev)
    d = ops.make_conv_desc(32, 32, (16, 16), 7, 3, 1, 24, False, True, 1.0)
    for T in (8, 12, 16, 20, 23, 24, 32, 48):
        spk = torch.randint(-2**31, 2**31 - 1, (T, B, 32, 8), generator=g, dtype=torch.int64).to(torch.int32).to(dev)
        st = [torch.zeros((B, 32, 16, 16), device=dev) for _ in range(3)]
        out = dict(spk=torch.empty((T, B, 32, 8), device=dev, dtype=torch.int32), pv=torch.empty((T, B, 32, 16, 16), device=dev))
        best = 1e9
        for rep in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.conv_lif_sequence(d, spk, W, b, tau4, *st, T, B, out=out); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print("T=%3d: %.3f ms  (%.1f us per step)" % (T, best, best * 1e3 / T))


if __name__ == "__main__":
    main()
