"""Randomised cross-check of independent code paths on the GPU (not part of the test suite): the fused all-T path
(k_lif_seq_c1 / c32d or the tiled kernels) against the per-step path (k_conv_lif_tiled / k_lif_step_c32 or the T = 1
tiled kernel) for random batch sizes, sequence lengths and planes; state, per-step argmax and logits must agree."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
from snn_modulation_classification_amd.networks import ConvNetwork, load_network_spec
from snn_modulation_classification_amd.data.utils import IQEncoder
PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "snn_modulation_classification_amd")
convs = load_network_spec(os.path.join(PKG, "networks", "radio_ml_conv.yaml"))
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_ok = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    R = int(rng.choice([16, 16, 16, 32, 64]))
    B = int(rng.randint(1, 40 if R == 16 else 6))
    T = int(rng.randint(1, 70 if R == 16 else 12))
    arp = float(rng.choice([1.0, 0.0]))
    args = Namespace(netscale=1.0, alpha=.92, alphas=.85, alpharp=.65, arp=arp, lc_ampl=.5, random_tau=True)
    nets = []
    for _ in range(2):
        torch.manual_seed(trial); np.random.seed(trial)
        n = ConvNetwork(args, (1, R, R), B, convs, 24, act=torch.nn.Sigmoid(), loss=None, opt=None, opt_param={},
                        learning_rates=None, burnin=0)
        n.reset(True)
        nets.append(n)
    a, b = nets
    enc = IQEncoder(R, R, device="cuda")
    for rnd in range(2):                       # two consecutive batches: state carry-over
        iq = (0.45 * torch.randn(B, 2, 128)).cuda()
        cells = enc(iq, T, t0=0)
        a.reset(); b.reset()
        ra = a.test_sequence(cells)
        logits = [[] for _ in range(3)]
        for t in range(T):
            x = torch.zeros(B, R * R, device="cuda")
            x[torch.arange(B), cells[t].long()] = 1
            cur = x.reshape(B, 1, R, R)
            for i, s in enumerate(b.dcll_slices):
                o, p, pv, v = s.forward(cur, ignore_burnin=True)
                logits[i].append(p)
                cur = o
        for i in range(3):
            for name in ("eps0", "eps1") + (("arp",) if arp > 0 else ()):
                sa = getattr(a.dcll_slices[i].dclllayer.i2h.state, name)
                sb = getattr(b.dcll_slices[i].dclllayer.i2h.state, name)
                assert torch.equal(sa, sb), (trial, rnd, R, B, T, arp, i, name)
            ref = torch.stack(logits[i])
            err = float((ra["logits"][i] - ref).abs().max())
            assert err < 1e-4, (trial, rnd, R, B, T, arp, i, err)
    n_ok += 1
    print("trial %2d ok: plane %dx%d, B=%d, T=%d, arp=%g" % (trial, R, R, B, T, arp), flush=True)
print("all %d trials agree" % n_ok)
