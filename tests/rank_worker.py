"""One rank of the multi-rank rehearsal (tests/test_gpu_multirank.py): NOT a test module.

    RANK=r WORLD_SIZE=n MASTER_ADDR=127.0.0.1 MASTER_PORT=p [DCLL_DIST_BACKEND=gloo] python tests/rank_worker.py OUT B T

Runs the real network on this rank's contiguous shard of a seeded B-window batch (fused sequence path, T steps) and a
short local-learning run on its shard of a second small batch, exactly as bench.py / test_radio_ml.py / train.py do
under several ranks, and writes what it computed to OUT/rank_<r>.npz.  With WORLD_SIZE unset it is the single-process
run on the full batch that the ranks are compared with.
"""
import os
import sys
from argparse import Namespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from snn_modulation_classification_amd import parallel  # noqa: E402
from snn_modulation_classification_amd.data.utils import IQEncoder  # noqa: E402
from snn_modulation_classification_amd.networks import ConvNetwork, load_network_spec  # noqa: E402

PKG = os.path.join(ROOT, "snn_modulation_classification_amd")
N_CLASSES = 24
LEARN_B, LEARN_T, LEARN_BURNIN = 48, 7, 3


def radio_net(B, R_, dev, learn=False):
    convs = load_network_spec(os.path.join(PKG, "networks", "radio_ml_conv.yaml"))
    args = Namespace(netscale=1.0, alpha=.92, alphas=.85, alpharp=.65, arp=1.0, lc_ampl=.5, random_tau=True)
    torch.manual_seed(1)
    np.random.seed(1)
    kw = dict(loss=None, opt=None, opt_param={}, learning_rates=None, burnin=20)
    if learn:
        kw = dict(loss=torch.nn.SmoothL1Loss, opt=torch.optim.Adam,
                  opt_param={"betas": [0.0, .95], "weight_decay": 10.0}, learning_rates=[1e-7], burnin=LEARN_BURNIN)
    net = ConvNetwork(args, (1, R_, R_), B, convs, N_CLASSES, act=torch.nn.Sigmoid(), **kw)
    net.reset(True)
    return net


def main():
    out_dir, B, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rank, local_rank, world = parallel.init_process_group()
    from snn_modulation_classification_amd.dcll import pytorch_libdcll
    idx = parallel.local_device(local_rank)
    pytorch_libdcll.device = "cuda:%d" % idx
    torch.cuda.set_device(idx)
    dev = torch.device("cuda", idx)
    lo, hi = parallel.shard_range(B, rank, world)

    # ---- evaluation: fused sequence path on my shard, tallies all-reduced (bench.py / test_radio_ml.py) ----
    g = torch.Generator().manual_seed(11)
    iq = 0.4 * torch.randn(B, 2, 128, generator=g)
    labels = torch.randint(0, N_CLASSES, (B,), generator=g)
    net = radio_net(hi - lo, 16, dev)
    enc = IQEncoder(16, 16, device=dev)
    net.zero_states()
    net.reset()
    res = net.test_sequence(iq=iq[lo:hi].to(dev), encoder=enc, T=T, t0=0, collect=True)
    tal = parallel.allreduce_tallies(parallel.tallies(res["vote"], labels[lo:hi].to(dev), N_CLASSES))
    out = {"lo": lo, "hi": hi, "tallies": tal.cpu().numpy(), "o": res["o"].cpu().numpy()}
    for i in range(3):
        out["clout%d" % i] = res["clout"][i].cpu().numpy()
        out["vote%d" % i] = res["vote"][i].cpu().numpy()
        out["host_clout%d" % i] = np.asarray(net.dcll_slices[i].clout)

    # ---- local learning: net.learn on my shard, gradients averaged over the ranks every timestep (train.py) ----
    llo, lhi = parallel.shard_range(LEARN_B, rank, world)
    rng = np.random.RandomState(5)
    cells = rng.randint(0, 256, size=(LEARN_T, LEARN_B))
    x = np.zeros((LEARN_T, LEARN_B, 256), np.float32)
    x[np.arange(LEARN_T)[:, None], np.arange(LEARN_B)[None, :], cells] = 1
    x = torch.from_numpy(x.reshape(LEARN_T, LEARN_B, 1, 16, 16))[:, llo:lhi].contiguous().to(dev)
    lab = rng.randint(0, N_CLASSES, size=LEARN_B)
    y = torch.zeros(LEARN_T, LEARN_B, N_CLASSES)
    y[:, np.arange(LEARN_B), lab] = 1
    y = y[:, llo:lhi].contiguous().to(dev)
    lnet = radio_net(lhi - llo, 16, dev, learn=True)
    lnet.reset()
    lnet.train()
    lnet.global_batch = LEARN_B
    for t in range(LEARN_T):
        lnet.learn(x[t], y[t])
        if t == LEARN_BURNIN - 1:         # first learning step: gradients of identical weights
            for i, s in enumerate(lnet.dcll_slices):
                out["g0_w%d" % i] = s.dclllayer.i2h.weight.grad.cpu().numpy()
                out["g0_b%d" % i] = s.dclllayer.i2h.bias.grad.cpu().numpy()
            out["g0_ow"] = lnet.dcll_slices[-1].dclllayer.output_.weight.grad.cpu().numpy()
    for i, s in enumerate(lnet.dcll_slices):
        out["w%d" % i] = s.dclllayer.i2h.weight.detach().cpu().numpy()
        out["b%d" % i] = s.dclllayer.i2h.bias.detach().cpu().numpy()
    out["ow"] = lnet.dcll_slices[-1].dclllayer.output_.weight.detach().cpu().numpy()
    out["ob"] = lnet.dcll_slices[-1].dclllayer.output_.bias.detach().cpu().numpy()
    torch.cuda.synchronize()
    out["backend"] = np.array(torch.distributed.get_backend() if parallel.is_distributed() else "none")
    np.savez(os.path.join(out_dir, "rank_%d_of_%d.npz" % (rank, world)), **out)
    if parallel.is_distributed():
        parallel.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
