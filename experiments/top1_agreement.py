"""Top-1 (vote) agreement between the MI355X fused path and the torch-CPU port of the reference on >= 10k synthetic
windows ("within 0.1 %" measured as prediction agreement).  Writes profiles/r01_top1_agreement.json.
Uses oracle/ as the checker only (this is a verification script, not product)."""


def main():
    import json, os, sys, time
    import numpy as np, torch
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import bench
    from oracle import torch_ref
    from snn_modulation_classification_amd.data.utils import IQEncoder

    NB, B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 20, 512, bench.T_STEPS
    dev = torch.device("cuda", 0)
    net, convs = bench.build_net(B, dev)
    enc = IQEncoder(bench.R, bench.R, device=dev)
    sds = [{k: v.detach().cpu() for k, v in s.dclllayer.state_dict().items()} for s in net.dcll_slices]
    ref = torch_ref.RefConvNetwork(sds, convs, wrp=1.0)
    torch.set_num_threads(bench.usable_cores())
    agree = [0, 0, 0]
    clout_agree = 0.0
    t0 = time.time()
    for i in range(NB):
        g = torch.Generator().manual_seed(100 + i)
        iq = (0.4 * torch.randn(B, 2, bench.L_IQ, generator=g)).to(dev)
        net.zero_states(); net.reset()
        res = net.test_sequence(iq=iq, encoder=enc, T=T, t0=0, collect=False)
        cells = enc(iq, T, t0=0).cpu().long()
        x = torch.zeros(T, B, bench.R * bench.R).scatter_(2, cells.unsqueeze(-1), 1.0).reshape(T, B, 1, bench.R, bench.R)
        with torch.no_grad():
            ref.reset(True)
            for t in range(T):
                ref.test(x[t])
        votes = ref.votes()
        for l in range(3):
            agree[l] += int((votes[l] == res["vote"][l].cpu().numpy()).sum())
        clout_agree += float((np.array(ref.clout[2]) == res["clout"][2].cpu().numpy()).mean())
        print("batch %d/%d: cumulative top-1 agreement (output layer) %.5f  [%.0f s]" % (i + 1, NB, agree[2] / ((i + 1) * B), time.time() - t0), flush=True)
    out = {"windows": NB * B, "T": T, "vote_agreement_per_layer": [a / (NB * B) for a in agree],
           "per_step_argmax_agreement_output_layer": clout_agree / NB,
           "setup": "radio_ml_conv.yaml 16x16, arp=1, random_tau, seeded init, synthetic IQ 0.4*randn; GPU = fused sequence path, "
                    "CPU = oracle/torch_ref.py (op-for-op port of the reference, bit-identical to it on the golden vectors)"}
    with open(os.path.join(ROOT, "profiles", "r01_top1_agreement.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
