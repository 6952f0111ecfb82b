#!/usr/bin/env python
"""What does a kernel that merely moves N bytes cost on this MI355X when it runs in a stream of dependent launches — the floor
under the small kernels of the per-timestep paths (readout pass 16.8 MB in 8.7 us, dv 33.6 MB in 11.3 us, ...)?
torch elementwise kernels (copy = read + write, fill = write, mul(out=) = read + write), 200 back-to-back launches, HIP events."""


def main():
    import json
    import torch

    dev = torch.device("cuda")
    out = {}
    for mb in (1, 4, 16.8, 33.6, 67, 134, 268):
        n = int(mb * 1e6 / 4)
        x = torch.randn(n, device=dev)
        y = torch.empty_like(x)
        rec = {}
        for name, fn, moved in (("fill (write N)", lambda: y.fill_(1.0), 1), ("copy (read N + write N)", lambda: y.copy_(x), 2),
                                ("mul out= (read N + write N)", lambda: torch.mul(x, 2.0, out=y), 2)):
            for _ in range(20):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 200
            rec[name] = {"us_per_launch": round(us, 2), "TBps": round(moved * 4 * n / us / 1e6, 2)}
        out["%g MB" % mb] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
