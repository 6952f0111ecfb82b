"""dcll_step_readouts alone (per-step readout tail: split-K pass + finishing launch) at rows = batch — diagnostic.
usage: python experiments/step_readout_timing.py

Round 4: 13.9 us (N = 24) / 18.6-19.2 us (N = 48) per call at 128 AND at 512 rows - the pair of launches is bound by its fixed
launch / drain latency, not by its 17 MB: a register ring of four K-chunks in flight per thread (instead of one) changed
nothing (13.93 vs 13.73, 19.15 vs 18.55 us; 2048 rows: 25.4 vs 23.9) and was not kept."""


def main():
    import os, sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from snn_modulation_classification_amd import ops
    dev = 'cuda'
    for rows in (128, 512, 2048):
        for N2 in (0, 24):
            K, N1 = 8192, 24
            pv = torch.rand(rows, K, device=dev)
            Wt = (torch.rand(N1 + N2, K, device=dev) - .5) * .01
            bias = torch.zeros(N1 + N2, device=dev)
            p = torch.empty(rows, N1, device=dev)
            o = torch.empty(rows, N2, device=dev) if N2 else None
            sc = {}
            fin = dict(clout=torch.empty(rows, device=dev, dtype=torch.int32))
            for _ in range(5):
                ops.step_readouts(pv, Wt, bias, N1, N2, p, o, scratch=sc, finish=dict(fin))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 200
            e0.record()
            for _ in range(n):
                ops.step_readouts(pv, Wt, bias, N1, N2, p, o, scratch=sc, finish=dict(fin))
            e1.record()
            torch.cuda.synchronize()
            print("rows %5d N %2d: %.2f us per call (two launches)" % (rows, N1 + N2, 1e3 * e0.elapsed_time(e1) / n))


if __name__ == "__main__":
    main()
