"""Debug: graph-replayed learning steps vs eager ones, step by step."""


def main():
    import os, sys
    import numpy as np, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from argparse import Namespace
    from snn_modulation_classification_amd.networks import ConvNetwork, load_network_spec
    PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "snn_modulation_classification_amd")
    B, T = 8, 30
    convs = load_network_spec(os.path.join(PKG, "networks", "radio_ml_conv.yaml"))
    args = Namespace(netscale=1.0, alpha=.92, alphas=.85, alpharp=.65, arp=1.0, lc_ampl=.5, random_tau=True)
    def make(graph):
        torch.manual_seed(1); np.random.seed(1)
        net = ConvNetwork(args, (1, 16, 16), B, convs, 24, act=torch.nn.Sigmoid(), loss=torch.nn.SmoothL1Loss,
                          opt=torch.optim.Adam, opt_param={"betas": [0.0, .95], "weight_decay": 10.0}, learning_rates=[1e-6], burnin=2)
        net.graph_learn = graph
        net.reset(True)
        return net
    a, b = make(True), make(False)
    x = torch.zeros(T, B, 1, 256, device='cuda')
    idx = torch.randint(0, 256, (T, B), device='cuda')
    x.scatter_(3, idx[:, :, None, None], 1.0)
    x = x.reshape(T, B, 1, 16, 16)
    y = torch.zeros(B, 24, device='cuda'); y[torch.arange(B), torch.randint(0, 24, (B,))] = 1
    for t in range(T):
        a.learn(x[t], y); b.learn(x[t], y)
        torch.cuda.synchronize()
        sa, sb = a.state_dict(), b.state_dict()
        bad = [k for k in sa if not torch.equal(sa[k], sb[k])]
        st = []
        for i, (s1, s2) in enumerate(zip(a.dcll_slices, b.dcll_slices)):
            for n, (t1, t2) in enumerate(zip(s1.dclllayer.i2h.state, s2.dclllayer.i2h.state)):
                if not torch.equal(t1, t2):
                    st.append((i, n))
            for k in ('g_p', 'g_o'):
                if k in s1._learn_bufs and not torch.equal(s1._learn_bufs[k], s2._learn_bufs[k]):
                    st.append((i, k))
            for q1, q2 in zip(s1._grad_tensors() if s1.dclllayer.i2h.weight.grad is not None else [], s2._grad_tensors() if s2.dclllayer.i2h.weight.grad is not None else []):
                if not torch.equal(q1, q2):
                    st.append((i, 'grad', tuple(q1.shape), float((q1 - q2).abs().max()), float(q1.abs().max())))
        print(t, "graphs", {k: g['n'] for k, g in a._learn_graphs.items()}, "bad", bad[:4], st[:6])
        if bad:
            for k in bad[:3]:
                print("   ", k, float((sa[k] - sb[k]).abs().max()), float(sa[k].abs().max()))
            break


if __name__ == "__main__":
    main()
