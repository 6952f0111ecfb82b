"""dcll_conv_lif_backward (dv + k_bwd_wgrad_c32 + fixed-order reduce) of a 32 -> 32, 7x7, 16x16 layer over a batch sweep —
run once per library build (DCLL_HIP_SO=experiments/_variants/libdcll_hip_<name>.so) to place WG32_SPLIT2_MAX_BATCH:
    python experiments/wgrad_split_sweep.py [B ...]"""


def main():
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from snn_modulation_classification_amd import ops
    dev = torch.device("cuda", 0)
    Bs = [int(a) for a in sys.argv[1:]] or [192, 256, 384, 512, 768, 1024, 2048, 4096]
    desc = ops.make_conv_desc(32, 32, (16, 16), 7, 3, 1, 24, False, True, 1.0, 0.65)
    torch.manual_seed(0)
    i2o = torch.randn(24, 8192, device=dev) * 0.01
    res = []
    for B in Bs:
        eps1 = torch.rand(B, 32, 16, 16, device=dev)
        pv = torch.rand(B, 32, 16, 16, device=dev)
        g_p = torch.randn(B, 24, device=dev) / B
        out = {}
        for _ in range(5):
            ops.conv_lif_backward(desc, eps1, None, pv, g_p, None, None, None, i2o, False, out=out)
        best = 1e9
        for rep in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(40):
                ops.conv_lif_backward(desc, eps1, None, pv, g_p, None, None, None, i2o, False, out=out)
            e.record()
            torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / 40 * 1e3)
        res.append((B, best))
    print(os.environ.get("DCLL_HIP_SO", "product").split("/")[-1], " ".join("B=%d: %.1f us" % r for r in res))


if __name__ == "__main__":
    main()
