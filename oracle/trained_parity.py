"""ORACLE-side checker (test infrastructure, not product): top-1 of the fused MI355X path against the reference's CPU path
on a network that has LEARNED something.

The reference's protocol is train (train.py:239-254: per timestep net.learn, checkpoint parameters_{step}.pth :297-303) ->
restore (test_radio_ml.py:97-110: load_state_dict, net.reset(True)) -> evaluate (:142-146: T x net.test, accuracy_by_vote).
north_star asks for "top-1 accuracy within 0.1 % of reference"; on freshly initialised weights that comparison happens at
chance level (biases dominate, ~5 % activity), so this module

  1. trains radio_ml_conv.yaml with the build's own train.py on the seeded synthetic modulation set (PSK / APSK rings +
     AWGN; test_radio_ml.synthetic_modulation_batches) — 16x16 plane, arp 1.0, batch 512, T = 128, burn-in 20, SmoothL1 +
     Adam as in the reference's scripts — and takes the checkpoint train.py wrote;
  2. restores it the reference's way into the HIP network (the product) and hands the SAME tensors (state_dict of the
     restored network, after reset(True) — quirk Q4 re-draws the time constants there, in both) to oracle/torch_ref.py;
  3. evaluates the same held-out windows on both and reports per-layer top-1 (accuracy_by_vote), their difference, the
     vote agreement, the per-step argmax agreement of the output layer and the spike flips (oracle/flip_count.py).

Only tests/ and bench.py's cpu_baseline leg import it."""
import contextlib
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

R, T_STEPS, N_CLASSES = 16, 128, 24
COMMON = ['--I_resolution', str(R), '--Q_resolution', str(R), '--arp', '1.0', '--burnin', '20', '--seed', '1']


def train_checkpoint(out_dir, steps=25, batch=512, lr=1e-6, quiet=True):
    """train.py --synthetic for `steps` batches of `batch` windows -> path of the last parameters_{step}.pth it wrote.
    (stdout of the entry point goes to stderr when `quiet`: bench.py's stdout is one JSON line.)"""
    import train
    interval = max(1, steps - 1)
    argv = COMMON + ['--batch_size', str(batch), '--batch_size_test', str(batch), '--n_test_samples', str(batch),
                     '--synthetic', str(batch), '--n_steps', str(steps), '--n_iters', str(T_STEPS), '--n_iters_test',
                     str(T_STEPS), '--n_test_interval', str(interval), '--learning_rates', repr(lr), '--output', out_dir]
    with (contextlib.redirect_stdout(sys.stderr) if quiet else contextlib.nullcontext()):
        run_dir = train.main(argv)
    ckpts = sorted(glob.glob(os.path.join(run_dir, 'parameters_*.pth')),
                   key=lambda p: int(os.path.basename(p)[len('parameters_'):-len('.pth')]))
    assert ckpts, 'train.py wrote no checkpoint into %s' % run_dir
    return ckpts[-1]


def restore_pair(ckpt, batch, device='cuda'):
    """The reference's restore (test_radio_ml.py:93-110) into the HIP network, and oracle/torch_ref.py on the same tensors.
    -> (net, ref, convs, encoder)"""
    from argparse import Namespace
    from oracle import torch_ref
    from snn_modulation_classification_amd.data.utils import IQEncoder
    from snn_modulation_classification_amd.networks import ConvNetwork, load_network_spec
    convs = load_network_spec(os.path.join(ROOT, 'snn_modulation_classification_amd', 'networks', 'radio_ml_conv.yaml'))
    args = Namespace(netscale=1.0, alpha=.92, alphas=.85, alpharp=.65, arp=1.0, lc_ampl=.5, random_tau=True)
    torch.manual_seed(1)
    np.random.seed(1)
    net = ConvNetwork(args, (1, R, R), batch, convs, N_CLASSES, act=torch.nn.Sigmoid(), loss=None, opt=None,
                      opt_param={}, learning_rates=None, burnin=20)
    net.load_state_dict(torch.load(ckpt))
    net = net.to(device)
    net.reset(True)
    sds = [{k: v.detach().cpu() for k, v in s.dclllayer.state_dict().items()} for s in net.dcll_slices]
    ref = torch_ref.RefConvNetwork(sds, convs, wrp=1.0)
    return net, ref, convs, IQEncoder(R, R, device=device)


def held_out_batches(n_batches, batch, seed=4242, snrs=tuple(range(6, 32, 2)), length=T_STEPS):
    """`n_batches` batches of the synthetic modulation set the training run has not seen (its batches are seeded
    seed + 7919 * (step + 1), these 4242 + ...), cycling through the SNRs test_radio_ml.py evaluates (:114)."""
    from test_radio_ml import synthetic_modulation_batches
    out = []
    for i in range(n_batches):
        snr = int(snrs[i % len(snrs)])
        out.append(synthetic_modulation_batches(batch, batch, snr, length, seed + 31 * i)[0] + (snr,))
    return out


def evaluate(net, ref, enc, batches, count_flips=True, log=None, T=None):
    """Both paths over the same windows, zero neuron state per batch (as bench.py's step; the flip classification needs
    it); T timesteps per window (default 128; the windows must be at least that long).  -> report dict"""
    from oracle import flip_count
    T_STEPS = globals()['T_STEPS'] if T is None else int(T)
    L = len(net.dcll_slices)
    n = 0
    correct_gpu, correct_cpu, agree = np.zeros(L), np.zeros(L), np.zeros(L)
    step_agree = 0.0
    flips = None
    for i, (iq, labels, snr) in enumerate(batches):
        B = iq.shape[0]
        iq_d = iq.reshape(B, 2, -1).to(enc.device)
        net.zero_states()
        net.reset()
        res = net.test_sequence(iq=iq_d, encoder=enc, T=T_STEPS, t0=0, collect=False, keep_spikes=count_flips)
        cells = enc(iq_d, T_STEPS, t0=0).cpu().long()
        x = torch.zeros(T_STEPS, B, R * R).scatter_(2, cells.unsqueeze(-1), 1.0).reshape(T_STEPS, B, 1, R, R)
        ref.reset(True)
        with torch.no_grad():
            if count_flips:
                dev_spikes = [flip_count.unpack_words(s_.cpu().numpy(), (R, R)) for s_ in res['spikes']]
                flips = flip_count.merge(flips, flip_count.spike_flips(ref, x, dev_spikes))
            else:
                for t in range(T_STEPS):
                    ref.test(x[t])
        votes = ref.votes()
        lab = labels.numpy()
        for l in range(L):
            vg = res['vote'][l].cpu().numpy()
            correct_gpu[l] += int((vg == lab).sum())
            correct_cpu[l] += int((votes[l] == lab).sum())
            agree[l] += int((vg == votes[l]).sum())
        step_agree += float((np.array(ref.clout[L - 1]) == res['clout'][L - 1].cpu().numpy()).mean()) * B
        n += B
        if log is not None:
            log('trained-weights parity: batch %d/%d (SNR %d dB): top-1 so far GPU %s CPU %s'
                % (i + 1, len(batches), snr, np.round(correct_gpu / n, 4).tolist(), np.round(correct_cpu / n, 4).tolist()))
    acc_g, acc_c = correct_gpu / n, correct_cpu / n
    return {'windows': int(n), 'top1_gpu': [float(a) for a in acc_g], 'top1_cpu_reference_path': [float(a) for a in acc_c],
            'top1_abs_diff': [float(abs(a - b)) for a, b in zip(acc_g, acc_c)],
            'vote_agreement_per_layer': [float(a) for a in agree / n],
            'output_layer_per_step_argmax_agreement': float(step_agree / n),
            'chance': 1.0 / N_CLASSES, 'spike_flips': flips}
