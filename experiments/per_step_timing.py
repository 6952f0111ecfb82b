"""Timing of the per-step paths (net.test / net.learn) — diagnostic."""


def main():
    import os, sys, time
    import numpy as np, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from argparse import Namespace
    from snn_modulation_classification_amd.networks import ConvNetwork, load_network_spec
    PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "snn_modulation_classification_amd")
    B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 70
    convs = load_network_spec(os.path.join(PKG, "networks", "radio_ml_conv.yaml"))
    args = Namespace(netscale=1.0, alpha=.92, alphas=.85, alpharp=.65, arp=1.0, lc_ampl=.5, random_tau=True)
    torch.manual_seed(1); np.random.seed(1)
    net = ConvNetwork(args, (1, 16, 16), B, convs, 24, act=torch.nn.Sigmoid(), loss=torch.nn.SmoothL1Loss,
                      opt=torch.optim.Adam, opt_param={"betas": [0.0, .95], "weight_decay": 10.0}, learning_rates=[1e-9], burnin=2)
    net.reset(True)
    x = torch.zeros(T, B, 1, 256, device='cuda')
    idx = torch.randint(0, 256, (T, B), device='cuda')
    x.scatter_(3, idx[:, :, None, None], 1.0)
    x = x.reshape(T, B, 1, 16, 16)
    y = torch.zeros(B, 24, device='cuda'); y[torch.arange(B), torch.randint(0, 24, (B,))] = 1
    for name, fn in (("test (per step)", lambda t: net.test(x[t])), ("learn (per step)", lambda t: net.learn(x[t], y))):
        net.reset()
        for t in range(6):                     # (learn: burn-in, two eager learning steps, the graph capture)
            fn(t)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(6, T):
            fn(t)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (T - 6)
        print("%-18s graph=%s B=%d" % (name, os.environ.get("DCLL_GRAPH_LEARN", "1"), B) + ": %.2f ms per timestep -> %.0f windows/s at T=128" % (dt * 1e3, B / (dt * 128)))
        continue
        print("%-18s B=%d: %.2f ms per timestep -> %.0f windows/s at T=128" % (name, B, dt * 1e3, B / (dt * 128)))


if __name__ == "__main__":
    main()
