"""k_lif_seq_w3f on a grid beyond residency with non-zero initial traces: the whole batch in one call against the same
samples in co-resident pairs.  With the round-5 library (in-place trace write-back by the channel-group-0 wave) samples
differ; with the out-of-place advance (k_w3f_traces_advance) none do.  usage: python experiments/w3f_race_check.py [B] [T]"""
import os
import sys


def main():
    sys.path.insert(0, os.getcwd())
    import numpy as np
    import torch
    from snn_modulation_classification_amd import ops
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 384
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    H, Wd, cout = 16, 128, 64
    rng = np.random.RandomState(77)
    dev = torch.device("cuda:0")
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    stdv = 1.0 / np.sqrt(3) / 250
    W = cu((rng.uniform(-stdv * 1e-2, stdv * 1e-2, size=(cout, 1, 1, 3)) * 3.0).astype(np.float32))
    b = cu(rng.uniform(-stdv, stdv, size=(cout,)).astype(np.float32))
    tau4 = cu(np.float32([[.95], [20.0], [1 - 1 / 7.0], [7.0]]))
    e0 = rng.uniform(0, 5, size=(B, 1, H, Wd)).astype(np.float32)
    e1 = rng.uniform(0, 50, size=(B, 1, H, Wd)).astype(np.float32)
    ar = -rng.uniform(0, 2, size=(B, cout, H, Wd)).astype(np.float32)
    cells = rng.randint(0, H * Wd, size=(T, B)).astype(np.int32)
    d = ops.make_conv_desc(1, cout, (H, Wd), (1, 3), (0, 1), (1, 2), 24, False, True, 1.0)
    s0, s1, s2 = cu(e0), cu(e1), cu(ar)
    spk, pv, v = ops.conv_lif_sequence_cells(d, cu(cells), W, b, tau4, s0, s1, s2, T, B, want_spikes=True, want_v=True)
    torch.cuda.synchronize()
    bad_v = bad_state = 0
    for k in range(0, B, 2):
        q0, q1, q2 = cu(e0[k:k + 2]), cu(e1[k:k + 2]), cu(ar[k:k + 2])
        _, _, vk = ops.conv_lif_sequence_cells(d, cu(cells[:, k:k + 2]), W, b, tau4, q0, q1, q2, T, 2, want_spikes=True, want_v=True)
        for j in range(2):
            bad_v += int(not torch.equal(vk[:, j], v[:, k + j]))
            bad_state += int(not (torch.equal(q0[j], s0[k + j]) and torch.equal(q1[j], s1[k + j]) and torch.equal(q2[j], s2[k + j])))
    print("B=%d T=%d: %d samples with a different membrane trace, %d with a different final state (of %d)" % (B, T, bad_v, bad_state, B))


if __name__ == "__main__":
    main()
