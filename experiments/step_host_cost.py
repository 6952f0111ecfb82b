"""Host time of an eager net.test(x[t]) timestep at B = 512 (cProfile): where does the launch path spend its ~0.25 ms?"""


def main():
    import os, sys, cProfile, pstats, time
    sys.path.insert(0, os.getcwd())
    import numpy as np, torch
    from argparse import Namespace
    from snn_modulation_classification_amd.networks import ConvNetwork, load_network_spec
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    convs = load_network_spec("snn_modulation_classification_amd/networks/radio_ml_conv.yaml")
    args = Namespace(netscale=1.0, alpha=.92, alphas=.85, alpharp=.65, arp=1.0, lc_ampl=.5, random_tau=True)
    torch.manual_seed(1); np.random.seed(1)
    net = ConvNetwork(args, (1, 16, 16), B, convs, 24, act=torch.nn.Sigmoid(), loss=None, opt=None, opt_param={}, learning_rates=None, burnin=2)
    net.graph_learn = False
    net.reset(True)
    x = torch.zeros(B, 1, 16, 16, device="cuda"); x[:, 0, 3, 5] = 1
    for t in range(30): net.test(x)
    torch.cuda.synchronize()
    # host-only pace: the device is far behind nothing (the queue never fills at this depth)
    t0 = time.perf_counter()
    for t in range(19 * 10):
        if (net.dcll_slices[0].iter + 1) % 20 == 0: net.test(x); continue
        net.test(x)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("host enqueue %.1f us per timestep, + %.1f us drain per timestep" % (1e6 * (t1 - t0) / 190, 1e6 * (t2 - t1) / 190))
    pr = cProfile.Profile(); pr.enable()
    for t in range(200): net.test(x)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(34)


if __name__ == "__main__":
    main()
