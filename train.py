#!/usr/bin/env python
"""Training entry point — flag surface of the reference's train.py (:19-94) on the MI355X build.

Runs the DCLL half of the reference's loop on the HIP kernels: per training step a batch is encoded to spike planes,
`net.learn(x[t], labels[t])` is called for every timestep (train.py:249-251: forward + local loss + backward + Adam
step per layer per timestep after burn-in, dcll/pytorch_libdcll.py:690-718), learning rates are halved every 1000
steps (:222-229), and every `n_test_interval` steps the test batches are evaluated and `acc_test.npy` /
`parameters_{step}.pth` (reference state-dict keys) are written (:263-303).  Not run: the non-spiking baseline CNN
(ReferenceConvNetwork, networks/__init__.py:21-113 — outside the DCLL path) and tensorboard dumps.
Data: RadioML HDF5 needs h5py (absent here); `--synthetic N` uses the build's seeded constellation generator.
"""
import argparse
import datetime
import os
import pickle
import sys

import numpy as np
import torch

from snn_modulation_classification_amd import parallel
from snn_modulation_classification_amd.data.utils import to_one_hot
from snn_modulation_classification_amd.dcll import pytorch_libdcll
from snn_modulation_classification_amd.networks import ConvNetwork, load_network_spec
import test_radio_ml as evaluation

_NETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'snn_modulation_classification_amd', 'networks')


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='DCLL')
    p.add_argument('--data', type=str, default='RadioML', choices=['MNIST', 'RadioML'], help='which data to use')
    p.add_argument('--radio_ml_data_dir', type=str, default='2018.01', help='folder with the RadioML HDF5 file(s)')
    p.add_argument('--mnist_data_dir', type=str, default='./data', help='folder with the MNIST IDX files (--data MNIST)')
    p.add_argument('--min_snr', type=int, default=6, metavar='N', help='minimum SNR (inclusive)')
    p.add_argument('--max_snr', type=int, default=30, metavar='N', help='maximum SNR (inclusive)')
    p.add_argument('--per_h5_frac', type=float, default=0.5, metavar='N', help='fraction of each HDF5 file to use')
    p.add_argument('--train_frac', type=float, default=0.9, metavar='N', help='train split')
    p.add_argument('--network_spec', type=str, default=os.path.join(_NETS, 'radio_ml_conv.yaml'), metavar='S')
    p.add_argument('--ref_network_spec', type=str, default=os.path.join(_NETS, 'radio_ml_conv_ref.yaml'), metavar='S')
    p.add_argument('--just_ref', action='store_true', help='train only the non-spiking reference network')
    p.add_argument('--I_resolution', type=int, default=128, metavar='N')
    p.add_argument('--Q_resolution', type=int, default=128, metavar='N')
    p.add_argument('--I_bounds', type=float, default=(-1, 1), nargs=2)
    p.add_argument('--Q_bounds', type=float, default=(-1, 1), nargs=2)
    p.add_argument('--restore_path', type=str, metavar='S', help='.pth state-dict to restore')
    p.add_argument('--burnin', type=int, default=50, metavar='N')
    p.add_argument('--batch_size', type=int, default=64, metavar='N')
    p.add_argument('--batch_size_test', type=int, default=64, metavar='N')
    p.add_argument('--n_steps', type=int, default=10000, metavar='N', help='number of steps to train')
    p.add_argument('--no_save', type=evaluation.str2bool_like_reference, default=False, metavar='N')
    p.add_argument('--seed', type=int, default=1, metavar='S')
    p.add_argument('--n_test_interval', type=int, default=20, metavar='N')
    p.add_argument('--n_test_samples', type=int, default=128, metavar='N')
    p.add_argument('--n_iters', type=int, default=1024, metavar='N')
    p.add_argument('--n_iters_test', type=int, default=1024, metavar='N')
    p.add_argument('--optim_type', type=str, default='Adam', metavar='S')
    p.add_argument('--loss_type', type=str, default='SmoothL1Loss', metavar='S')
    p.add_argument('--learning_rates', type=float, default=[1e-6], nargs='+', metavar='N')
    p.add_argument('--ref_lr', type=float, default=1e-3, metavar='N')
    p.add_argument('--alpha', type=float, default=.92, metavar='N')
    p.add_argument('--alphas', type=float, default=.85, metavar='N')
    p.add_argument('--alpharp', type=float, default=.65, metavar='N')
    p.add_argument('--arp', type=float, default=0, metavar='N')
    p.add_argument('--random_tau', type=evaluation.str2bool_like_reference, default=True)
    p.add_argument('--beta', type=float, default=.95, metavar='N')
    p.add_argument('--lc_ampl', type=float, default=0.5, metavar='N')
    p.add_argument('--netscale', type=float, default=1., metavar='N')
    p.add_argument('--comment', type=str, default='')
    p.add_argument('--output', type=str, default='results')
    # additions of this build
    p.add_argument('--synthetic', type=int, default=0, metavar='N', help='use N synthetic test windows')
    p.add_argument('--eval_only', action='store_true', help='run the periodic evaluation once and save parameters')
    p.add_argument('--host_encoding', action='store_true',
                   help='encode the training batches with the host iq2spiketrain loop (the reference\'s way) instead of on '
                        'the device')
    p.add_argument('--gpus', type=int, default=1, metavar='N',
                   help='ranks (one process per GPU): every batch is sharded over them, the local-learning gradients are '
                        'averaged over the ranks every timestep (one bucketed all-reduce)')
    return p.parse_args(argv)


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
    if args.data == 'MNIST':
        return main_mnist(args)
    if args.just_ref:
        sys.exit('ReferenceConvNetwork (plain CNN baseline) is outside the DCLL hot path and not part of this build.')
    stamp = datetime.datetime.now().strftime('%b%d_%H-%M-%S')
    out_dir = os.path.join(args.output, args.data, stamp)
    if world > 1:                                  # one results directory for the job: rank 0 names it
        names = [out_dir]
        torch.distributed.broadcast_object_list(names, src=0)
        out_dir = names[0]
    os.makedirs(out_dir, exist_ok=world > 1)
    if rank == 0:
        print('out dir: {}'.format(out_dir))
    if rank != 0:
        args.no_save = True                        # every rank trains its shard; rank 0 alone writes files

    im_dims = (1, args.Q_resolution, args.I_resolution)
    target_size = evaluation.TARGET_SIZE
    opt = getattr(torch.optim, args.optim_type)
    opt_param = {'betas': [0.0, args.beta], 'weight_decay': 10.0}
    loss = getattr(torch.nn, args.loss_type)
    convs = load_network_spec(args.network_spec)
    # sized for this rank's shard of a batch (identical initial weights / time constants on every rank: their RNG draws
    # do not depend on the batch size)
    lo, hi = parallel.shard_range(args.batch_size_test if args.eval_only else args.batch_size, rank, world)
    net = ConvNetwork(args, im_dims, max(hi - lo, 1), convs, target_size,
                      act=torch.nn.Sigmoid(), loss=loss, opt=opt, opt_param=opt_param,
                      learning_rates=args.learning_rates, burnin=args.burnin)
    if args.restore_path:
        print('-' * 80)
        if not os.path.isfile(args.restore_path):
            print('ERROR: Cannot load `%s`.' % args.restore_path)
            print('File does not exist! Aborting load...')
        else:
            net.load_state_dict(torch.load(args.restore_path))
            print('Loaded the SNN model from `%s`.' % args.restore_path)
        print('-' * 80)
    net = net.to(pytorch_libdcll.device)
    net.reset(True)
    parallel.freeze_startup_heap()          # (a full GC pass over the start-up heap costs ~100 ms inside the T-loop)

    if not args.no_save:
        with open(os.path.join(out_dir, 'args.txt'), 'w') as f:
            f.write(str(args))
        with open(os.path.join(out_dir, 'args.pkl'), 'wb') as f:
            pickle.dump(vars(args), f)

    from snn_modulation_classification_amd.data.utils import IQEncoder, iq2spiketrain
    n_test = int(np.ceil(float(args.n_test_samples) / args.batch_size_test))
    gen_train = train_data = None
    if args.synthetic:
        test_batches = evaluation.synthetic_modulation_batches(args.synthetic, args.batch_size_test, args.max_snr,
                                                               max(args.n_iters_test, 128), args.seed)[:n_test]
    else:
        # the reference's data path (train.py:134-141, :196-207): per-(class, SNR) RadioML blocks, interleaved;
        # n_test fixed test batches drawn once, training batches from a shuffling loader that restarts when exhausted
        from snn_modulation_classification_amd.data.load_radio_ml import get_radio_ml_loader
        kw = dict(data_dir=args.radio_ml_data_dir, min_snr=args.min_snr, max_snr=args.max_snr,
                  per_h5_frac=args.per_h5_frac, train_frac=args.train_frac)
        train_data = get_radio_ml_loader(args.batch_size, train=True, **kw)
        gen_train = iter(train_data)
        gen_test = iter(get_radio_ml_loader(args.batch_size_test, train=False, **kw))
        test_batches = [next(gen_test) for _ in range(n_test)]
        n_test = len(test_batches)
    use_sequence = net.sequence_supported()
    device_encoding = not args.host_encoding
    encoder = IQEncoder(args.I_resolution, args.Q_resolution, args.I_bounds, args.Q_bounds,
                        device=pytorch_libdcll.device) if (use_sequence or device_encoding) else None
    import time
    t_train, n_train = 0.0, 0

    def run_tests(step):
        net.batch_size = max(1, np.diff(parallel.shard_range(args.batch_size_test, rank, world))[0])
        acc = np.empty([len(test_batches), len(net.dcll_slices)])
        for i, (samples, labels) in enumerate(test_batches):
            acc[i, :], _ = evaluation.evaluate_batch(net, args, samples, to_one_hot(labels, target_size), encoder,
                                                     use_sequence)
        net.batch_size = max(1, np.diff(parallel.shard_range(args.batch_size, rank, world))[0])
        if rank == 0:
            print('[TEST]  Step {} \t Accuracy {} \t Ref {}'.format(str(step).zfill(5), np.mean(acc, axis=0), 'N/A'))
        return acc

    if args.eval_only:
        acc_test = run_tests(0)[None]
    else:
        n_tests_total = int(np.ceil(float(args.n_steps) / args.n_test_interval))
        acc_test = np.empty([n_tests_total, n_test, len(net.dcll_slices)])
        st_kw = dict(out_w=args.I_resolution, out_h=args.Q_resolution, min_I=args.I_bounds[0], max_I=args.I_bounds[1],
                     min_Q=args.Q_bounds[0], max_Q=args.Q_bounds[1], max_duration=args.n_iters, gs_stdev=0)
        for step in range(args.n_steps):
            if ((step + 1) % 1000) == 0:                      # reference :222-229
                for sl in net.dcll_slices:
                    sl.optimizer.param_groups[-1]['lr'] /= 2
                net.dcll_slices[-1].optimizer2.param_groups[-1]['lr'] /= 2
                print('Adjusting learning rates')
            if gen_train is None:
                snr = int(np.random.randint(args.min_snr // 2, args.max_snr // 2 + 1) * 2)
                samples, labels = evaluation.synthetic_modulation_batches(
                    args.batch_size, args.batch_size, snr, max(args.n_iters, 128), args.seed + 7919 * (step + 1))[0]
            else:
                try:
                    samples, labels = next(gen_train)
                except StopIteration:                         # reference :231-235
                    gen_train = iter(train_data)
                    samples, labels = next(gen_train)
                if samples.shape[0] != args.batch_size:       # ragged last batch: the state is sized for batch_size
                    gen_train = iter(train_data)
                    samples, labels = next(gen_train)
            net.global_batch = samples.shape[0]               # (the gradient all-reduce weighs the shards by it)
            enc_pos = dict(start=0, total=samples.shape[0])   # my samples' positions in the whole batch (exact quantisation)
            if world > 1:                                     # my contiguous shard of the batch
                a, b = parallel.shard_range(samples.shape[0], rank, world)
                samples, labels = samples[a:b], labels[a:b]
                enc_pos['start'] = a
            t_step = time.perf_counter()
            labels1h = to_one_hot(labels, target_size)
            net.reset()
            net.train()
            if device_encoding:
                # raw IQ to the GPU, iq2spiketrain's quantisation as a kernel (same random crop draw), burn-in steps on
                # the fused sequence kernels, learning steps on device-built planes
                dev = pytorch_libdcll.device
                cells = encoder(samples.to(dev), args.n_iters, **enc_pos)
                y = torch.as_tensor(np.asarray(labels1h), dtype=torch.float32).to(dev)
                net.learn_sequence(cells, y)
                labels_spikes = y.unsqueeze(0).expand(args.n_iters, -1, -1)
            else:
                spikes, targets = iq2spiketrain(samples, labels1h, **st_kw)
                input_spikes = torch.Tensor(spikes).to(pytorch_libdcll.device)
                labels_spikes = torch.as_tensor(np.asarray(targets), dtype=torch.float32).to(pytorch_libdcll.device)
                for t in range(args.n_iters):
                    net.learn(x=input_spikes[t], labels=labels_spikes[t])
            acc_train = net.accuracy(labels_spikes)           # (reads the per-step argmax back: ends the step's GPU work)
            t_train += time.perf_counter() - t_step
            n_train += samples.shape[0] * world
            if rank == 0:
                print('[TRAIN] Step {} \t Accuracy {} \t {:.0f} windows/s incl. encoding'.format(
                    str(step).zfill(5), acc_train, n_train / t_train))
            if (step % args.n_test_interval) == 0:
                acc_test[step // args.n_test_interval] = run_tests(step)
                if not args.no_save:
                    np.save(os.path.join(out_dir, 'acc_test.npy'), acc_test)
                    save_path = os.path.join(out_dir, 'parameters_{}.pth'.format(step))
                    torch.save(net.cpu().state_dict(), save_path)
                    net.to(pytorch_libdcll.device)
                    print('-' * 80)
                    print('Saved network parameters to `%s`.' % save_path)
                    print('-' * 80)
        if world > 1:
            parallel.barrier()
            torch.distributed.destroy_process_group()
        return out_dir

    # --eval_only: the periodic evaluation block of the reference (train.py:263-303) once, then save
    if not args.no_save:
        np.save(os.path.join(out_dir, 'acc_test.npy'), acc_test)
        save_path = os.path.join(out_dir, 'parameters_{}.pth'.format(0))
        torch.save(net.cpu().state_dict(), save_path)
        net.to(pytorch_libdcll.device)
        print('-' * 80)
        print('Saved network parameters to `%s`.' % save_path)
        print('-' * 80)
    return out_dir


def main_mnist(args):
    """--data MNIST (reference train.py:118-131): 28x28 images as frozen Poisson spike trains (image2spiketrain,
    gain 100), 10 classes, any conv spec that fits 28x28 (networks/mnist_conv.yaml), per-step protocol for learning
    and for the periodic test.  Images from the IDX files under --mnist_data_dir, or `--synthetic N` random images."""
    from snn_modulation_classification_amd.data.utils import image2spiketrain
    if args.just_ref:
        sys.exit('ReferenceConvNetwork (plain CNN baseline) is outside the DCLL hot path and not part of this build.')
    stamp = datetime.datetime.now().strftime('%b%d_%H-%M-%S')
    out_dir = os.path.join(args.output, args.data, stamp)
    os.makedirs(out_dir)
    print('out dir: {}'.format(out_dir))
    im_dims, target_size = (1, 28, 28), 10
    opt = getattr(torch.optim, args.optim_type)
    opt_param = {'betas': [0.0, args.beta], 'weight_decay': 10.0}
    loss = getattr(torch.nn, args.loss_type)
    convs = load_network_spec(args.network_spec)
    net = ConvNetwork(args, im_dims, args.batch_size, convs, target_size, act=torch.nn.Sigmoid(), loss=loss, opt=opt,
                      opt_param=opt_param, learning_rates=args.learning_rates, burnin=args.burnin)
    if args.restore_path and os.path.isfile(args.restore_path):
        net.load_state_dict(torch.load(args.restore_path))
        print('Loaded the SNN model from `%s`.' % args.restore_path)
    net = net.to(pytorch_libdcll.device)
    net.reset(True)
    parallel.freeze_startup_heap()          # (a full GC pass over the start-up heap costs ~100 ms inside the T-loop)
    n_test = int(np.ceil(float(args.n_test_samples) / args.batch_size_test))
    if args.synthetic:
        g = torch.Generator().manual_seed(args.seed)
        def batches(n, bs):
            return [(torch.rand(bs, 28, 28, generator=g), torch.randint(0, 10, (bs,), generator=g)) for _ in range(n)]
        test_batches = batches(n_test, args.batch_size_test)
        train_data = batches(max(1, args.synthetic // args.batch_size), args.batch_size)
    else:
        from snn_modulation_classification_amd.data.load_mnist import get_mnist_loader
        train_data = get_mnist_loader(args.batch_size, train=True, data_dir=args.mnist_data_dir)
        gen_test = iter(get_mnist_loader(args.batch_size_test, train=False, data_dir=args.mnist_data_dir))
        test_batches = [next(gen_test) for _ in range(n_test)]
    gen_train = iter(train_data)
    st_train = dict(input_shape=im_dims, gain=100, min_duration=args.n_iters - 1, max_duration=args.n_iters)
    st_test = dict(input_shape=im_dims, gain=100, min_duration=args.n_iters_test - 1, max_duration=args.n_iters_test)
    dev = pytorch_libdcll.device

    def spikes_of(samples, labels, kw):
        sp, tg = image2spiketrain(samples.numpy(), to_one_hot(labels, target_size), **kw)
        return (torch.Tensor(sp).to(dev), torch.as_tensor(np.asarray(tg), dtype=torch.float32).to(dev))

    n_tests_total = int(np.ceil(float(args.n_steps) / args.n_test_interval))
    acc_test = np.empty([n_tests_total, n_test, len(net.dcll_slices)])
    for step in range(args.n_steps):
        if ((step + 1) % 1000) == 0:
            for sl in net.dcll_slices:
                sl.optimizer.param_groups[-1]['lr'] /= 2
            net.dcll_slices[-1].optimizer2.param_groups[-1]['lr'] /= 2
            print('Adjusting learning rates')
        try:
            samples, labels = next(gen_train)
        except StopIteration:
            gen_train = iter(train_data)
            samples, labels = next(gen_train)
        if samples.shape[0] != args.batch_size:
            gen_train = iter(train_data)
            samples, labels = next(gen_train)
        x, y = spikes_of(samples, labels, st_train)
        net.batch_size = args.batch_size
        net.reset()
        net.train()
        for t in range(args.n_iters):
            net.learn(x=x[t], labels=y[t])
        print('[TRAIN] Step {} \t Accuracy {}'.format(str(step).zfill(5), net.accuracy(y)))
        if (step % args.n_test_interval) == 0:
            net.batch_size = args.batch_size_test
            for i, (ts, tl) in enumerate(test_batches):
                tx, ty = spikes_of(ts, tl, st_test)
                net.reset()
                net.eval()
                for t in range(args.n_iters_test):
                    net.test(x=tx[t])
                acc_test[step // args.n_test_interval, i, :] = net.accuracy(ty)
            print('[TEST]  Step {} \t Accuracy {} \t Ref N/A'.format(
                str(step).zfill(5), np.mean(acc_test[step // args.n_test_interval], axis=0)))
            if not args.no_save:
                np.save(os.path.join(out_dir, 'acc_test.npy'), acc_test)
                save_path = os.path.join(out_dir, 'parameters_{}.pth'.format(step))
                torch.save(net.cpu().state_dict(), save_path)
                net.to(dev)
                print('Saved network parameters to `%s`.' % save_path)
    return out_dir


if __name__ == '__main__':
    main()
