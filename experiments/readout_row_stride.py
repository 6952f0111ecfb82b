

def main():
    import os, sys, torch
    sys.path.insert(0, os.getcwd())
    from snn_modulation_classification_amd import ops
    dev = torch.device("cuda", 0)
    for K, rows in ((8192, 524288), (65536, 131072), (65536 + 64, 131072), (65536 + 2048, 131072), (32768, 131072), (32768 + 64, 131072)):
        pv = torch.rand(rows, K, device=dev)
        W = (torch.rand(24, K, device=dev) - 0.5) * 0.01
        b = torch.zeros(24, device=dev)
        out = torch.empty(rows, 24, device=dev)
        for _ in range(2): ops.readout(pv, W, b, out=out, mode=ops.READOUT_T16)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): ops.readout(pv, W, b, out=out, mode=ops.READOUT_T16)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print("K=%6d rows=%7d: %.3f ms  %.2f TB/s" % (K, rows, ms, rows * K * 4 / ms / 1e9), flush=True)
        del pv


if __name__ == "__main__":
    main()
