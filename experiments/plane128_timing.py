"""Per-step path at the argparse-default 128x128 I/Q plane (stress configuration) — diagnostic timing."""


def main():
    import os, sys, time
    import numpy as np, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from argparse import Namespace
    from snn_modulation_classification_amd.networks import ConvNetwork, load_network_spec
    PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "snn_modulation_classification_amd")
    B, T, R = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 8, 128
    convs = load_network_spec(os.path.join(PKG, "networks", "radio_ml_conv.yaml"))
    args = Namespace(netscale=1.0, alpha=.92, alphas=.85, alpharp=.65, arp=1.0, lc_ampl=.5, random_tau=True)
    torch.manual_seed(1); np.random.seed(1)
    net = ConvNetwork(args, (1, R, R), B, convs, 24, act=torch.nn.Sigmoid(), loss=None, opt=None, opt_param={},
                      learning_rates=None, burnin=2)
    net.reset(True)
    x = torch.zeros(T, B, 1, R * R, device='cuda')
    x.scatter_(3, torch.randint(0, R * R, (T, B), device='cuda')[:, :, None, None], 1.0)
    x = x.reshape(T, B, 1, R, R)
    net.reset(); net.test(x[0]); net.test(x[1])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(2, T):
        net.test(x[t])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (T - 2)
    flop = 2 * 32 * 49 * R * R * (1 + 32 + 32) * B
    print("128x128 plane, B=%d: %.1f ms per timestep (%.2f TFLOP/s) -> %.2f windows/s at T=128" % (B, dt * 1e3, flop / dt / 1e12, B / (dt * 128)))

    # fused all-T path (k_lif_seq_c1t / k_lif_seq_c32t): python experiments/plane128_timing.py B seq [T]
    if len(sys.argv) > 2 and sys.argv[2] == "seq":
        from snn_modulation_classification_amd.data.utils import IQEncoder
        Ts = int(sys.argv[3]) if len(sys.argv) > 3 else 128
        enc = IQEncoder(R, R, device='cuda')
        iq = (0.4 * torch.randn(B, 2, 128)).cuda()
        for rep in range(3):
            prof = {}
            net.zero_states(); net.reset()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            net.test_sequence(iq=iq, encoder=enc, T=Ts, t0=0, profile=prof)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            parts = {k: [round(a.elapsed_time(b), 2) for a, b in v] for k, v in prof.items()}
            print("fused, B=%d T=%d: %.1f ms -> %.1f windows/s  %s" % (B, Ts, dt * 1e3, B / dt, parts))

    # local-learning step on the 128x128 plane: python experiments/plane128_timing.py B learn
    if len(sys.argv) > 2 and sys.argv[2] == "learn":
        torch.manual_seed(1); np.random.seed(1)
        lnet = ConvNetwork(args, (1, R, R), B, convs, 24, act=torch.nn.Sigmoid(), loss=torch.nn.SmoothL1Loss,
                           opt=torch.optim.Adam, opt_param={"betas": [0.0, .95], "weight_decay": 10.0}, learning_rates=[1e-9],
                           burnin=2)
        lnet.reset(True)
        y = torch.zeros(B, 24, device='cuda'); y[torch.arange(B), torch.randint(0, 24, (B,))] = 1
        lnet.reset()
        for t in range(3):
            lnet.learn(x[t], y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(3, T):
            lnet.learn(x[t], y)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (T - 3)
        print("learn, 128x128 plane, B=%d: %.1f ms per timestep -> %.2f windows/s at T=128" % (B, dt * 1e3, B / (dt * 128)))


if __name__ == "__main__":
    main()
