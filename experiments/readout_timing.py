"""Readout GEMM forms on the headline shape (rows = T*B = 128*B, K = 8192, N = 24 / 48): LDS-staged 32x32x2 kernels
(k_readout_v4, round 1) vs the LDS-free 16x16x4 kernel k_readout_direct in its standalone and co-resident forms.
    python experiments/readout_timing.py [B]
"""


def main():
    import os, sys
    import torch
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    from snn_modulation_classification_amd import ops

    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    rows, K = 128 * B, 8192
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    pv = torch.rand(rows, K, device=dev)
    for N in (24, 48):
        W = (torch.rand(N, K, device=dev) - 0.5) * 0.011
        b = (torch.rand(N, device=dev) - 0.5) * 0.011
        outs = {}
        for name, mode in (("lds_32x32x2", ops.READOUT_LDS), ("lds_16x16x4", ops.READOUT_T16), ("auto", ops.READOUT_AUTO),
                           ("direct_coresident", ops.READOUT_CORESIDENT)):
            out = torch.empty(rows, N, device=dev)
            for _ in range(2):
                ops.readout(pv, W, b, out=out, mode=mode)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.readout(pv, W, b, out=out, mode=mode)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            outs[name] = out
            print("N=%d %-18s %.3f ms  %.2f TB/s of pv  %.1f TFLOP/s (N as given)" %
                  (N, name, ms, rows * K * 4 / ms / 1e9, 2.0 * rows * K * N / ms / 1e9), flush=True)
        ref = (pv[:4096].double() @ W.double().T + b.double()).float()
        for name, out in outs.items():
            print("   %-18s max |err| vs float64 on 4096 rows: %.2e   vs lds form: %.2e" %
                  (name, float((out[:4096] - ref).abs().max()), float((out - outs["lds_32x32x2"]).abs().max())))


if __name__ == "__main__":
    main()
