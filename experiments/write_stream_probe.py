#!/usr/bin/env python
"""What a pure HBM WRITE stream reaches on this MI355X, beside a pure read stream and a copy — the yardstick for the first
layer of BASELINE config 5 (k_lif_seq_w3<1>: 137 GB of pooled v written per 4096-window step, round-4 verdict #4 (i)).
torch kernels on a 16 GiB fp32 tensor (far beyond the 256 MiB Infinity Cache), HIP events, best of 5."""


def main():
    import json
    import torch

    dev = torch.device("cuda")
    n = 4 * 2 ** 30                      # 4 Gi floats = 16 GiB
    x = torch.empty(n, device=dev)
    y = torch.empty(n, device=dev)


    def best(fn, reps=5):
        t = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            t.append(e0.elapsed_time(e1))
        return min(t)


    out = {}
    ms = best(lambda: x.fill_(1.0))
    out["fill (write only)"] = {"ms": ms, "TBps": 4 * n / ms / 1e9}
    ms = best(lambda: x.zero_())
    out["zero_ (memset)"] = {"ms": ms, "TBps": 4 * n / ms / 1e9}
    ms = best(lambda: torch.sum(x))
    out["sum (read only)"] = {"ms": ms, "TBps": 4 * n / ms / 1e9}
    ms = best(lambda: y.copy_(x))
    out["copy (read + write)"] = {"ms": ms, "TBps_total": 8 * n / ms / 1e9}
    ms = best(lambda: torch.mul(x, 2.0, out=y))
    out["mul out= (read + write)"] = {"ms": ms, "TBps_total": 8 * n / ms / 1e9}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
