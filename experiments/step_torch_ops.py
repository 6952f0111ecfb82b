"""Which torch operators still launch kernels inside one timed headline step, and from which line?  (diagnostic, not product)
torch.profiler with stacks around ONE step of bench.py's protocol (zero_states, reset, test_sequence, tallies) at batch 4096:
every non-library kernel with its duration and the innermost frame of this repo that issued it.
    python experiments/step_torch_ops.py"""


def main():
    import os, sys, collections
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from torch.profiler import profile, ProfilerActivity
    import bench
    from snn_modulation_classification_amd.data.utils import IQEncoder
    from snn_modulation_classification_amd import parallel

    dev = torch.device("cuda", 0)
    B = 4096
    net, _ = bench.build_net(B, dev)
    enc = IQEncoder(bench.R, bench.R, device=dev)
    g = torch.Generator().manual_seed(11)
    iq = (0.4 * torch.randn(B, 2, bench.L_IQ, generator=g)).to(dev)
    labels = torch.randint(0, bench.N_CLASSES, (B,), generator=g).to(dev)


    def step():
        net.zero_states()
        net.reset()
        res = net.test_sequence(iq=iq, encoder=enc, T=bench.T_STEPS, t0=0, collect=False)
        return parallel.tallies(res["vote"], labels, bench.N_CLASSES)


    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if ev.device_type is None or "cuda" not in str(ev.device_type).lower():
            continue
    for ev in prof.key_averages(group_by_stack_n=12):
        dt = getattr(ev, "device_time_total", None) or getattr(ev, "cuda_time_total", 0)
        if not dt or ev.key.startswith(("k_", "void k_")):
            continue
        where = next((f for f in ev.stack if root in f and "experiments" not in f), ev.stack[0] if ev.stack else "?")
        agg[(ev.key[:48], where.replace(root + "/", "")[:90])][0] += ev.count
        agg[(ev.key[:48], where.replace(root + "/", "")[:90])][1] += dt
    for (k, w), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
        print("%8.1f us  x%-3d %-48s %s" % (t, n, k, w))


if __name__ == "__main__":
    main()
