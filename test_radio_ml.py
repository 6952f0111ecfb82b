#!/usr/bin/env python
"""Per-SNR evaluation of a DCLL network on RadioML IQ windows — MI355X build.

Keeps the command line, the printed lines and the output files of the reference's test_radio_ml.py (flags :17-61;
snr_evaluation.txt, confusion_matrix_snr_%d.npy, snr_evaluation_accs.npy :71,156-169) while the inner loop
(:142-146) runs on the HIP kernels: the fused whole-sequence path when the network geometry has one
(radio_ml_conv.yaml on a 16x16 I/Q plane), otherwise one C-ABI call per layer per step.

Data: `--radio_ml_data_dir` holds the per-(class, SNR) blocks of RadioML 2018.01A (the reference's .hdf5 split — needs
h5py — or the same blocks as .npy) or the RadioML 2016.10a pickle (data/load_radio_ml.py).  `--synthetic N` evaluates
N seeded synthetic windows per SNR instead (the build's own generator: PSK/QAM constellations + AWGN), which is also
what the tests use.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

from snn_modulation_classification_amd import parallel
from snn_modulation_classification_amd.data.utils import IQEncoder, iq2spiketrain, to_one_hot
from snn_modulation_classification_amd.dcll import pytorch_libdcll
from snn_modulation_classification_amd.networks import ConvNetwork, load_network_spec

TARGET_SIZE = 24      # hard-coded in the reference (test_radio_ml.py:78)


def str2bool_like_reference(v):
    """argparse `type=bool` of the reference: any non-empty string is True."""
    return bool(v)


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--radio_ml_data_dir', type=str, default='2018.01', help='folder with the RadioML HDF5 file(s)')
    p.add_argument('--per_h5_frac', type=float, default=0.5, metavar='N', help='fraction of each HDF5 file to use')
    p.add_argument('--train_frac', type=float, default=0.9, metavar='N', help='train split (rest is test)')
    p.add_argument('--network_spec', type=str, default=os.path.join(
        os.path.dirname(os.path.abspath(__file__)), 'snn_modulation_classification_amd', 'networks',
        'radio_ml_conv.yaml'), metavar='S', help='YAML file describing the architecture')
    p.add_argument('--I_resolution', type=int, default=128, metavar='N', help='I size of the I/Q plane image')
    p.add_argument('--Q_resolution', type=int, default=128, metavar='N', help='Q size of the I/Q plane image')
    p.add_argument('--I_bounds', type=float, default=(-1, 1), nargs=2, help='value range of the I axis')
    p.add_argument('--Q_bounds', type=float, default=(-1, 1), nargs=2, help='value range of the Q axis')
    p.add_argument('--restore_path', type=str, metavar='S', help='.pth state-dict to restore')
    p.add_argument('--burnin', type=int, default=50, metavar='N', help='burnin')
    p.add_argument('--batch_size_test', type=int, default=64, metavar='N', help='test batch size')
    p.add_argument('--seed', type=int, default=1, metavar='S', help='random seed')
    p.add_argument('--n_test_samples', type=int, default=128, metavar='N', help='test samples per SNR')
    p.add_argument('--n_iters_test', type=int, default=1024, metavar='N', help='timesteps per sample')
    p.add_argument('--alpha', type=float, default=.92, metavar='N', help='membrane time constant')
    p.add_argument('--alphas', type=float, default=.85, metavar='N', help='synapse time constant')
    p.add_argument('--alpharp', type=float, default=.65, metavar='N', help='refractory time constant')
    p.add_argument('--arp', type=float, default=0, metavar='N', help='refractory weight (wrp)')
    p.add_argument('--random_tau', type=str2bool_like_reference, default=True, help='randomize time constants')
    p.add_argument('--beta', type=float, default=.95, metavar='N', help='Adam beta2 (unused at test time)')
    p.add_argument('--lc_ampl', type=float, default=.5, metavar='N', help='local classifier init magnitude')
    p.add_argument('--netscale', type=float, default=1., metavar='N', help='scale network width')
    p.add_argument('--print_all_confusion_matrices', action='store_true')
    # additions of this build
    p.add_argument('--synthetic', type=int, default=0, metavar='N',
                   help='evaluate N seeded synthetic IQ windows per SNR instead of reading HDF5')
    p.add_argument('--out_dir', type=str, default=None, help='where to write results (default: dir of restore_path)')
    p.add_argument('--no_sequence_path', action='store_true', help='force the per-step path (net.test per timestep)')
    p.add_argument('--min_snr', type=int, default=6, metavar='N', help='first SNR evaluated (reference: fixed 6)')
    p.add_argument('--max_snr', type=int, default=30, metavar='N', help='last SNR evaluated (reference: fixed 30)')
    p.add_argument('--gpus', type=int, default=1, metavar='N',
                   help='ranks (one process per GPU): every test batch is sharded over them, tallies are all-reduced')
    return p.parse_args(argv)


def synthetic_modulation_batches(n, batch, snr_db, length, seed):
    """Seeded stand-in for the RadioML loader: 24 constellation 'classes', AWGN at snr_db, unit power * 0.5."""
    rng = np.random.RandomState(seed + 1000 * (snr_db + 50))
    out = []
    for _ in range(int(np.ceil(n / batch))):
        labels = rng.randint(0, TARGET_SIZE, size=batch)
        order = 2 + labels % 6                                  # points per ring
        rings = 1 + labels // 6                                 # number of amplitude rings
        k = rng.randint(0, 1 << 16, size=(batch, length))
        phase = 2 * np.pi * (k % order[:, None]) / order[:, None] + 0.1 * labels[:, None]
        amp = (1 + (k // 7) % rings[:, None]) / rings[:, None]
        sym = amp * np.exp(1j * phase)
        sym /= np.sqrt(np.mean(np.abs(sym) ** 2, axis=1, keepdims=True))
        noise = (rng.randn(batch, length) + 1j * rng.randn(batch, length)) * np.sqrt(0.5 * 10 ** (-snr_db / 10))
        x = 0.5 * (sym + noise)
        iq = np.stack([x.real, x.imag], axis=1)[:, :, None, :].astype(np.float32)
        out.append((torch.from_numpy(iq), torch.from_numpy(labels)))
    return out


def load_batches(args, snr, n_batches):
    if args.synthetic:
        return synthetic_modulation_batches(args.synthetic, args.batch_size_test, snr, max(args.n_iters_test, 128),
                                            args.seed)[:n_batches]
    from snn_modulation_classification_amd.data.load_radio_ml import get_radio_ml_loader
    try:
        loader = get_radio_ml_loader(args.batch_size_test, train=False, data_dir=args.radio_ml_data_dir, min_snr=snr,
                                     max_snr=snr, per_h5_frac=args.per_h5_frac, train_frac=args.train_frac)
    except (FileNotFoundError, RuntimeError) as e:
        sys.exit('Cannot read RadioML data from `%s`: %s\nPass --synthetic N to evaluate synthetic IQ windows.'
                 % (args.radio_ml_data_dir, e))
    it = iter(loader)
    return [next(it) for _ in range(n_batches)]


def evaluate_batch(net, args, samples, labels1h, encoder, use_sequence):
    """One batch through all timesteps; returns (per-layer accuracy list, confusion matrix of the last layer).
    Under several ranks every rank evaluates its contiguous shard of the batch (no data-path collective: samples are
    independent) and the per-layer correct counts + the confusion matrix are summed over the ranks."""
    T = args.n_iters_test
    rank, _, world = parallel.env_ranks()
    shard = None
    if world > 1:
        lo, hi = parallel.shard_range(samples.shape[0], rank, world)
        shard = (lo, samples.shape[0])          # my samples' positions in the whole batch (exact quantisation)
        samples, labels1h = samples[lo:hi], labels1h[lo:hi]
        net.batch_size = hi - lo
        if hi == lo:                                    # more ranks than samples: this rank only joins the reduction
            tal = torch.zeros(len(net.dcll_slices) + 1 + TARGET_SIZE * TARGET_SIZE, dtype=torch.int64)
            parallel.all_reduce_(tal)
            return _split_eval_tally(tal, len(net.dcll_slices))
    if use_sequence:
        targets = labels1h.unsqueeze(0).expand(T, -1, -1)        # iq2spiketrain repeats the labels over t
        net.reset()
        net.eval()
        # raw IQ to the GPU; quantisation to I/Q-plane cells happens inside the first layer's kernel
        net.test_sequence(iq=samples.to(pytorch_libdcll.device), encoder=encoder, T=T, shard=shard)
    else:
        spikes, targets = iq2spiketrain(samples, labels1h, out_w=args.I_resolution, out_h=args.Q_resolution,
                                        min_I=args.I_bounds[0], max_I=args.I_bounds[1], min_Q=args.Q_bounds[0],
                                        max_Q=args.Q_bounds[1], max_duration=T)
        try:
            test_input = torch.Tensor(spikes).to(pytorch_libdcll.device)
        except RuntimeError as e:
            print('Exception: ' + str(e) + '. Try to decrease your batch_size_test with the --batch_size_test argument.')
            raise
        net.reset()
        net.eval()
        for t in range(T):
            net.test(x=test_input[t])
    if not isinstance(targets, torch.Tensor):
        targets = torch.as_tensor(np.asarray(targets), dtype=torch.float32)
    acc, cm = net.accuracy(targets), net.confusion_matrix(targets)
    if world > 1:
        n = samples.shape[0]
        tal = torch.tensor([int(round(a * n)) for a in acc] + [n] + [int(v) for v in np.asarray(cm).reshape(-1)],
                           dtype=torch.int64)
        parallel.all_reduce_(tal)
        return _split_eval_tally(tal, len(acc))
    return acc, cm


def _split_eval_tally(tal, n_layers):
    """[correct per layer..., total, confusion matrix...] summed over the ranks -> (accuracies, confusion matrix)"""
    total = max(int(tal[n_layers]), 1)
    acc = [float(tal[i]) / total for i in range(n_layers)]
    cm = tal[n_layers + 1:].reshape(TARGET_SIZE, TARGET_SIZE).numpy().astype(int)
    return acc, cm


def main(argv=None):
    args = parse_args(argv)
    if args.gpus > 1 and not parallel.under_launcher():
        # plain start: become the launcher of one fresh process per rank (never touches the GPU itself)
        return sys.exit(parallel.spawn_local_ranks(args.gpus, argv=[os.path.abspath(__file__)] +
                                                   (sys.argv[1:] if argv is None else list(argv))))
    rank, local_rank, world = parallel.init_process_group()
    if world > 1 and torch.cuda.is_available():
        pytorch_libdcll.device = 'cuda:%d' % parallel.local_device(local_rank)
        torch.cuda.set_device(parallel.local_device(local_rank))
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    out_dir = args.out_dir or (os.path.dirname(args.restore_path) if args.restore_path else '.')
    os.makedirs(out_dir or '.', exist_ok=True)
    # every rank evaluates its shard; rank 0 alone reports and writes the result files
    with open(os.path.join(out_dir, 'snr_evaluation.txt') if rank == 0 else os.devnull, 'a+') as logfile:
        def say(text):
            if rank != 0:
                return
            print(text)
            logfile.write(text + '\n')

        im_dims = (1, args.Q_resolution, args.I_resolution)
        n_test = int(np.ceil(float(args.n_test_samples) / args.batch_size_test))
        convs = load_network_spec(args.network_spec)
        # the network is sized for this rank's shard of a test batch (identical weights / time constants on every
        # rank: their RNG draws do not depend on the batch size)
        lo, hi = parallel.shard_range(args.batch_size_test, rank, world)
        net = ConvNetwork(args, im_dims, max(hi - lo, 1), convs, TARGET_SIZE, act=torch.nn.Sigmoid(), loss=None,
                          opt=None, opt_param={}, learning_rates=None, burnin=args.burnin)
        if args.restore_path:
            say('-' * 80)
            if not os.path.isfile(args.restore_path):
                say('ERROR: Cannot load `%s`.' % args.restore_path)
                say('File does not exist! Aborting...')
                sys.exit(0)
            net.load_state_dict(torch.load(args.restore_path))
            say('Loaded the SNN model from `%s`.' % args.restore_path)
            say('-' * 80)
        net = net.to(pytorch_libdcll.device)
        net.reset(True)
        parallel.freeze_startup_heap()      # (a full GC pass over the start-up heap costs ~100 ms inside the T-loop)
        use_sequence = net.sequence_supported() and not args.no_sequence_path
        encoder = IQEncoder(args.I_resolution, args.Q_resolution, args.I_bounds, args.Q_bounds,
                            device=pytorch_libdcll.device) if use_sequence else None

        accs = []
        snrs = np.array(range(args.min_snr, args.max_snr + 2, 2))        # reference :114: range(6, 32, 2)
        total_cm = np.zeros((TARGET_SIZE, TARGET_SIZE), dtype=int)
        for snr in snrs:
            t_start = time.time()
            batches = load_batches(args, int(snr), n_test)
            acc_test = np.zeros([len(batches), len(net.dcll_slices)])
            cm = np.zeros((TARGET_SIZE, TARGET_SIZE), dtype=int)
            for i, (samples, labels) in enumerate(batches):
                acc_test[i, :], cm_i = evaluate_batch(net, args, samples, to_one_hot(labels, TARGET_SIZE), encoder,
                                                      use_sequence)
                cm += cm_i
            acc = np.mean(acc_test, axis=0)
            say('SNR {} \t Accuracy {} \t Time Elapsed {}'.format(str(snr).zfill(2), acc,
                                                                   '%.2f s' % (time.time() - t_start)))
            if args.print_all_confusion_matrices:
                say('Confusion matrix:')
                say(np.array2string(cm, max_line_width=300))
            if rank == 0:
                np.save(os.path.join(out_dir, 'confusion_matrix_snr_%d.npy' % snr), cm)
            accs.append(acc)
            total_cm += cm
        say('---\nTotal confusion matrix:')
        say(np.array2string(total_cm, max_line_width=300))
        npy_out = os.path.join(out_dir, 'snr_evaluation_accs.npy')
        if rank == 0:
            np.save(npy_out, accs)
        say('Wrote `%s`.' % npy_out)
        try:                                  # plots are optional extras of the reference (:170-188)
            if rank != 0:
                raise RuntimeError('rank 0 writes the plots')
            import matplotlib
            matplotlib.use('Agg')
            import matplotlib.pyplot as plt
            plt.imsave(os.path.join(out_dir, 'total_confusion_matrix.png'), total_cm, cmap='gray')
        except Exception:
            pass
    if world > 1:
        parallel.barrier()
        torch.distributed.destroy_process_group()
    return accs


if __name__ == '__main__':
    main()
