"""Host time of one graph-replayed learning timestep (cProfile) — diagnostic."""


def main():
    import cProfile, os, pstats, sys, time
    import numpy as np, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from argparse import Namespace
    from snn_modulation_classification_amd.networks import ConvNetwork, load_network_spec
    PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "snn_modulation_classification_amd")
    B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 400
    convs = load_network_spec(os.path.join(PKG, "networks", "radio_ml_conv.yaml"))
    args = Namespace(netscale=1.0, alpha=.92, alphas=.85, alpharp=.65, arp=1.0, lc_ampl=.5, random_tau=True)
    torch.manual_seed(1); np.random.seed(1)
    net = ConvNetwork(args, (1, 16, 16), B, convs, 24, act=torch.nn.Sigmoid(), loss=torch.nn.SmoothL1Loss,
                      opt=torch.optim.Adam, opt_param={"betas": [0.0, .95], "weight_decay": 10.0}, learning_rates=[1e-9], burnin=2)
    for s in net.dcll_slices:
        s.collect_stats = False
    net.reset(True)
    x = torch.zeros(8, B, 1, 256, device='cuda')
    x.scatter_(3, torch.randint(0, 256, (8, B), device='cuda')[:, :, None, None], 1.0)
    x = x.reshape(8, B, 1, 16, 16)
    y = torch.zeros(B, 24, device='cuda'); y[torch.arange(B), torch.randint(0, 24, (B,))] = 1
    for t in range(8):
        net.learn(x[t % 8], y)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for t in range(T):
        net.learn(x[t % 8], y)
    pr.disable()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("B=%d: host %.1f us per step (profiled), wall incl. GPU drain %.1f us per step" % (B, t_host / T * 1e6, t_all / T * 1e6))
    pstats.Stats(pr).sort_stats("cumtime").print_stats(18)


if __name__ == "__main__":
    main()
