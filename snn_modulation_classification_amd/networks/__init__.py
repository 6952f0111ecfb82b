"""YAML network builder — the drop-in surface of the reference's networks/__init__.py (:10-18, :116-199).

`load_network_spec(yaml)` and `ConvNetwork(args, im_dims, batch_size, convs, target_size, act, loss, opt, opt_param,
learning_rates, DCLLSlice, burnin)` with `.learn/.test/.reset/.accuracy/.confusion_matrix/.write_stats` keep the
reference's signatures.  `.test(x_t)` is the per-step path (one C-ABI call per layer per step);
`.test_sequence(...)` is the MI355X fast path: one fused launch per layer for all T steps, bit-packed spikes between
layers, fp32-MFMA readout over all (t, b) rows, per-step argmax and vote on device.
"""
from ast import literal_eval

import os

import numpy as np
import torch
import yaml

from .. import ops
from ..dcll import pytorch_libdcll as _libdcll
from ..dcll.pytorch_libdcll import Conv2dDCLLlayer, DenseDCLLlayer, DCLLClassification  # noqa: F401


def load_network_spec(yaml_path):
    """YAML -> list of layer dicts; non-int values such as "(1, 3)" become tuples (reference :10-18).
    Uses SafeLoader (the reference's bare yaml.load fails on PyYAML >= 6)."""
    with open(yaml_path, 'r') as f:
        spec = yaml.load(f, Loader=yaml.SafeLoader)
    convs = spec['conv_layers']
    for layer in convs:
        for key, val in layer.items():
            if type(val) != int:
                layer[key] = literal_eval(val)
    return convs


class ConvNetwork(torch.nn.Module):
    def __init__(self, args, im_dims, batch_size, convs, target_size, act, loss, opt, opt_param, learning_rates,
                 DCLLSlice=DCLLClassification, burnin=50):
        super().__init__()
        self.batch_size = batch_size
        self.target_size = target_size
        dev = _libdcll.device
        shape = tuple(im_dims)                    # (C, H, W)
        self.num_layers = len(convs)
        self.dcll_slices = torch.nn.ModuleList()
        for i, conf in enumerate(convs):
            last = (i == self.num_layers - 1)
            layer = Conv2dDCLLlayer(in_channels=shape[0],
                                    out_channels=int(conf['out_channels'] * args.netscale),
                                    kernel_size=conf['kernel_size'], padding=conf['padding'],
                                    pooling=conf['pooling'], im_dims=shape[1:3], target_size=target_size,
                                    alpha=args.alpha, alphas=args.alphas, alpharp=args.alpharp, wrp=args.arp,
                                    act=act, lc_ampl=args.lc_ampl, random_tau=args.random_tau, spiking=True,
                                    lc_dropout=False, output_layer=last).to(dev).init_hiddens(batch_size)
            shape = (layer.out_channels,) + tuple(layer.output_shape)
            layer_opt = dict(opt_param)
            if learning_rates is not None:
                layer_opt['lr'] = learning_rates[min(i, len(learning_rates) - 1)]
            self.dcll_slices.append(DCLLSlice(dclllayer=layer, name='conv%d' % i, batch_size=batch_size, loss=loss,
                                              optimizer=opt, kwargs_optimizer=layer_opt, collect_stats=True,
                                              burnin=burnin))
        self._seq_buffers = {}
        self._seq_scratch = {}
        # samples of the GLOBAL batch of the current learning step when the batch is sharded over ranks (the entry points
        # set it next to shard_range): lets the gradient all-reduce weigh the shards exactly; None = derive it in the
        # collective (a count element travels with every slab)
        self.global_batch = None
        # captured learning timesteps (hipGraph), per input shape; see _learn_graphed
        self.graph_learn = os.environ.get('DCLL_GRAPH_LEARN', '1') != '0'
        self._learn_graphs, self._learn_eager_steps, self._learn_last_key = {}, {}, None
        self._test_graphs, self._test_eager_steps = {}, {}
        # graph-or-eager decisions taken by measurement, per (path, input geometry): see _graph_tuned
        self.graph_autotune = os.environ.get('DCLL_GRAPH_AUTOTUNE', '1') != '0'
        self._graph_tune = {}
        # largest pv buffer (one layer, all T steps) the sequence path allocates; bigger batches run in chunks
        self.pv_budget_bytes = float(os.environ.get('DCLL_PV_BUDGET_GB', '24')) * 2 ** 30
        # sequence path: the layer kernels can write v and let the readout GEMM apply the sigmoid (dcll_layer_opts
        # pv_presigmoid + dcll_readout_act) — the transcendentals then leave the kernels that share their vector pipe with
        # the fp32 MFMAs.  DCLL_PRESIGMOID: 'auto' (default) = where it was measured to pay — the pooling (1,3) layers of
        # radio_ml_conv_ref.yaml (first layer 39 -> 33 ms at batch 4096); on the 7x7 layers of radio_ml_conv.yaml it is a
        # net LOSS (k_lif_seq_c1 5.47 -> 5.05 ms, k_lif_seq_c32d 24.25 -> 24.21 ms per 1024 windows, but the three
        # readouts 9.7 -> 10.8 ms: DESIGN.md 8) and stays off; '1' / '0' force it on / off everywhere.
        self.presigmoid = os.environ.get('DCLL_PRESIGMOID', 'auto')

    # -- reference protocol (per step) ----------------------------------------------------------------------------
    def learn(self, x, labels, global_batch=None):
        """One timestep of local learning in every slice (reference :175-180).  When every slice's step is native
        (DCLLBase._native_learning) the three phases are batched over the slices: all forwards + backwards (a slice's
        weight update only matters from the NEXT timestep on, and slice l+1 consumes slice l's spikes, not its
        weights), then ONE bucketed all-reduce of all gradients (multi-rank), then ONE Adam launch for all tensors.
        `global_batch` (ranks only): samples of the whole batch this shard belongs to, for this call; default: the
        attribute `self.global_batch` the entry points set per batch (None = the collective derives it from a count
        element).  A shard larger than the batch it claims to be part of is refused — a stale value would weigh the
        gradients wrongly without any other symptom."""
        if global_batch is None:
            global_batch = self.global_batch
        if global_batch is not None and x.shape[0] > global_batch:
            raise ValueError('shard of %d samples of a global batch of %d: set net.global_batch (or pass global_batch=) '
                             'for THIS batch' % (x.shape[0], global_batch))
        if not all(s._native_learning() is not None for s in self.dcll_slices):
            spikes = x
            for s in self.dcll_slices:
                spikes, _, _, _, _ = s.train_dcll(spikes, labels, regularize=False)
            return
        key = (tuple(x.shape), tuple(labels.shape))
        if key != self._learn_last_key:                # another geometry: the slices' buffers are reallocated
            self._learn_last_key, self._learn_eager_steps[key] = key, 0
        from .. import parallel
        ranks = parallel.is_distributed()
        if ranks:
            for s in self.dcll_slices:
                s._grads_into_slab()
        if self._graph_learn_ok(x, labels, key):
            if self._learn_graphed(x, labels, key, global_batch):
                return
            self._graph_interrupted('learn', key)          # (capture dropped or refused: this step runs eagerly)
        # Under ranks every slice's gradients are ONE slab whose all-reduce starts as soon as its backward is enqueued and
        # runs under the next slice's forward / backward (slice l+1 needs slice l's spikes, not its gradients); the
        # single Adam launch waits for all of them.
        learned, pending = self._learn_tails(self._learn_forwards(x, labels), ranks, x.shape[0], global_batch)
        for h in pending:
            parallel.allreduce_slab_end(h)
        if learned:
            if ranks:
                ops.adam_step([t for s in learned for t in s._adam_tensors()])
            else:
                self._finish_learning(learned)
            for s in learned:
                s.dclllayer.weights_written()          # (raw-pointer write: no version counter sees it)
            if len(learned) == len(self.dcll_slices):
                self._learn_eager_steps[key] = self._learn_eager_steps.get(key, 0) + 1

    def _learn_forwards(self, x, labels, clout_rows=None):
        """The layer kernels of one learning timestep, slice after slice: the chain the timestep cannot shorten (slice l+1
        consumes slice l's spikes).  Everything else of the step is left to _learn_tails.  -> per-slice contexts"""
        spikes, ctxs = x, []
        for i, s in enumerate(self.dcll_slices):
            ctx = s._learn_forward(spikes, labels, want_loss=False, defer=True,              # (nobody reads the loss value)
                                   clout_out=None if clout_rows is None else clout_rows[i],
                                   want_v=False)                                     # (nor the returned tuple's pvmem)
            spikes = ctx['out'][0]
            ctxs.append(ctx)
        return ctxs

    def _learn_tails(self, ctxs, ranks=False, local_n=0, global_batch=None):
        """Readouts, local-loss gradients and backward of every slice, behind all layer kernels of the timestep.  Under ranks
        a slice's gradient slab starts its all-reduce as soon as its backward is enqueued and travels under the next slice's
        tail.  Single rank: the weight gradients' last reduction stays open — _finish_learning does it for all slices
        together with the optimizer step.  -> (slices that learned, pending slab handles)"""
        from .. import parallel
        learned, pending = [], []
        ops.run_deferred_readouts([ctx['fin'] for ctx in ctxs])        # all slices' readout tails: two launches
        # single rank, small batches: the slices' backward as one call, their dv launches as one launch — where the timestep is
        # launch-bound (captured timesteps at B = 64: +5 %); at B = 512 the three gradient maps written together (50 MB) and read
        # back later cost more than the two launches (0.556 -> 0.56-0.58 ms per timestep), so each slice's dv stays in front of
        # its weight gradient there.  DCLL_BWD_MULTI_MAX_BATCH (default 128; 0: never) moves the limit.
        batch = ctxs[0]['input'].shape[0] if ctxs else 0
        bwd = [] if (not ranks and batch <= int(os.environ.get('DCLL_BWD_MULTI_MAX_BATCH', '128'))) else None
        for s, ctx in zip(self.dcll_slices, ctxs):
            s._learn_tail(ctx, open_reduce=not ranks, defer_backward=bwd)
            if ctx['learned']:
                learned.append(s)
                if ranks:
                    pending.append(parallel.allreduce_slab_begin(s._grad_slab, local_n, global_batch))
        if bwd:
            ops.conv_lif_backward_open_multi(bwd)
        return learned, pending

    @staticmethod
    def _finish_learning(learned, advance=True, dyn=None):
        """Single rank: ONE launch reduces the open weight gradients of the slices that learned (fixed order: the gradients
        in .grad are bit-identical to the per-slice reduction) and applies torch.optim.Adam's update to all their tensors
        (ops.grad_reduce_adam; output_.* get the plain elementwise update in the same launch)."""
        from .. import _lib
        tensors, layers, done = [], [], 0

        def flush():
            nonlocal tensors, layers, done
            if layers:
                ops.grad_reduce_adam(layers, tensors, dyn=None if dyn is None else dyn[3 * done:3 * (done + len(tensors))])
            done += len(tensors)
            tensors, layers = [], []
        # the ABI takes DCLL_REDUCE_MAX_LAYERS layers / DCLL_ADAM_MAX_TENSORS tensors per launch: the default three slices
        # are one launch, a deeper spec (radio_ml_conv_ref.yaml: 7 slices, 16 tensors) goes out in groups, slice order kept
        # (dyn holds 3 floats per tensor in that same order)
        for s in learned:
            mine = s._adam_tensors(advance=advance)
            if len(layers) + 1 > _lib.REDUCE_MAX_LAYERS or len(tensors) + len(mine) > _lib.ADAM_MAX_TENSORS:
                flush()
            parts = dict(s._learn_bufs['grads']['parts'])
            parts.update(adam_w=len(tensors), adam_b=len(tensors) + 1)     # (_adam_tensors: i2h.weight, i2h.bias first)
            tensors += mine
            layers.append(parts)
        flush()

    # -- the learning timestep as a captured hipGraph ------------------------------------------------------------------
    # At the reference's small batches (argparse default 64) a learning timestep is ~25 kernel launches of a few
    # microseconds each, and the host's launch path sets the pace (0.43 ms per timestep at B = 64 against 0.31 ms of GPU
    # work; at B = 512 the GPU is the limit and a graph changes nothing).  The launches of a step in which EVERY slice
    # learns are the same from step to step except for Adam's step-dependent scalars, so they are captured once
    # (torch.cuda.CUDAGraph = hipGraph) on static input buffers and replayed: the step-dependent scalars are read on the
    # device (dcll_adam_step_dyn), refreshed from the host before each replay.
    _DYN_RING = 32
    # Where a replay beats the eager loop depends on the HOST: on the pool's usual boxes the eager loop is host-bound up to
    # 128 samples of a 16x16 plane (learn: 0.46 -> 0.34 ms at B = 128, but 0.48 -> 0.49 at B = 256 and 0.73 -> 0.76 at
    # B = 512), while a slower host launches the same ~25 kernels of a B = 512 timestep in 2 ms against 0.73 ms of device
    # work (round-3 driver run).  So: replays unconditionally up to GRAPH_MAX_PIXELS, and between that and
    # GRAPH_TUNE_MAX_PIXELS the choice is MEASURED on the running job (_graph_tuned): TUNE_STEPS eager timesteps against
    # TUNE_STEPS replays, wall clock between two synchronisations each — four host syncs per geometry, once.  Larger
    # planes are device-bound on any host.
    GRAPH_MAX_PIXELS = 128 * 256
    GRAPH_TUNE_MAX_PIXELS = 2048 * 256
    TUNE_STEPS = 6
    TUNE_WINDOWS = 2

    def _graph_small(self, x):
        return x.dim() == 4 and x.shape[0] * x.shape[2] * x.shape[3] <= self.GRAPH_MAX_PIXELS

    def _graph_tuned(self, path, x, key):
        """Whether this step of `path` ('learn' / 'test') on geometry `key` should be a graph replay.  Small workloads: yes.
        Mid-size ones: the running job is timed — TUNE_WINDOWS windows of TUNE_STEPS eligible steps eagerly, one replay to
        take the capture, the same number of windows of replays — and the faster form (best window of each) is kept; ties:
        eager, it has no static-buffer copies.  A step that is not eligible (burn-in, pv statistics) restarts the window it
        falls into (graph_interrupted).  When eager wins the capture, its static buffers and its pinned ring are dropped;
        the decision is logged."""
        if self._graph_small(x):
            return True
        if not (self.graph_autotune and x.dim() == 4 and
                x.shape[0] * x.shape[2] * x.shape[3] <= self.GRAPH_TUNE_MAX_PIXELS):
            return False
        import time
        tk = (path, key)
        st = self._graph_tune.get(tk)
        if st is None:
            st = self._graph_tune[tk] = dict(phase='eager', n=0, w=0, t0=0.0, eager_ms=None, graph_ms=None, use_graph=None)
        if st['use_graph'] is not None:
            return st['use_graph']
        if st['phase'] in ('eager', 'graph') and st['n'] == self.TUNE_STEPS:
            torch.cuda.synchronize(x.device)
            ms = 1e3 * (time.perf_counter() - st['t0']) / self.TUNE_STEPS
            which = st['phase'] + '_ms'
            st[which] = ms if st[which] is None else min(st[which], ms)
            st['n'], st['w'] = 0, st['w'] + 1
            if st['w'] == self.TUNE_WINDOWS:
                if st['phase'] == 'eager':
                    st.update(phase='capture', w=0)
                else:
                    st['use_graph'] = bool(st['graph_ms'] < 0.97 * st['eager_ms'])
                    import logging
                    logging.getLogger(__name__).info(
                        'net.%s %s: eager %.3f ms, hipGraph replay %.3f ms per timestep (best of %d windows of %d) -> %s',
                        path, key, st['eager_ms'], st['graph_ms'], self.TUNE_WINDOWS, self.TUNE_STEPS,
                        'replay' if st['use_graph'] else 'eager')
                    if not st['use_graph']:             # the capture is not going to be replayed again: free it
                        (self._learn_graphs if path == 'learn' else self._test_graphs).pop(key, None)
                    return st['use_graph']
        if st['phase'] == 'capture':               # one untimed replay takes the capture
            st.update(phase='graph', n=0, w=0)
            return True
        if st['n'] == 0:
            torch.cuda.synchronize(x.device)
            st['t0'] = time.perf_counter()
        st['n'] += 1
        return st['phase'] == 'graph'

    def _graph_interrupted(self, path, key):
        """An ineligible step (burn-in, statistics, a dropped capture) fell into a measurement window: start it again."""
        st = self._graph_tune.get((path, key))
        if st is not None and st['use_graph'] is None and st['phase'] in ('eager', 'graph'):
            st['n'] = 0

    def graph_decisions(self):
        """{path: {geometry: {'eager_ms', 'graph_ms', 'use_graph'}}} of the measured graph-or-eager choices so far."""
        out = {}
        for (path, key), st in self._graph_tune.items():
            out.setdefault(path, {})[str(key)] = {k: st[k] for k in ('eager_ms', 'graph_ms', 'use_graph')}
        return out

    def _graph_learn_ok(self, x, labels, key):
        if not (self.graph_learn and x.is_cuda and x.dtype == torch.float32 and labels.dtype == torch.float32):
            return False
        if self._learn_eager_steps.get(key, 0) < 2:            # buffers, .grad and Adam state exist after eager steps
            self._graph_interrupted('learn', key)
            return False
        for s in self.dcll_slices:
            it = s.iter + 1
            if it < s.burnin or (s.collect_stats and it % 20 == 0):      # burn-in and histogram steps run eagerly
                self._graph_interrupted('learn', key)
                return False
        return self._graph_tuned('learn', x, key)

    def _graph_signature(self):
        """Everything a captured step has baked in: addresses of state, parameters, gradients and optimizer state, and
        the hyper-parameters that are kernel arguments.  (lr and the step count are read on the device.)"""
        sig = []

        def walk(d):
            for v in d.values():
                if isinstance(v, dict):
                    walk(v)
                elif isinstance(v, torch.Tensor):
                    sig.append(v.data_ptr())
        for s in self.dcll_slices:
            L = s.dclllayer
            sig += [t.data_ptr() for t in L.i2h.state]
            sig += [t.data_ptr() for t in (L.i2h.alpha, L.i2h.tau_m__dt, L.i2h.alphas, L.i2h.tau_s__dt, L.i2o.weight,
                                           L.i2o.bias)]
            walk(s.__dict__.get('_learn_bufs', {}))
            for t in s._adam_tensors(advance=False):
                sig += [t['param'].data_ptr(), t['grad'].data_ptr(), t['exp_avg'].data_ptr(),
                        t['exp_avg_sq'].data_ptr(), t['weight_decay'], t['beta1'], t['beta2'], t['eps']]
        return tuple(sig)

    @torch.no_grad()
    def _learn_graphed(self, x, labels, key, global_batch=None):
        """One learning timestep by replaying its captured graph -> True; False (nothing done) when the capture on
        record no longer matches the tensors in use — the caller then runs eager steps, after which a new one is taken."""
        sig = self._graph_signature()
        g = self._learn_graphs.get(key)
        if g is not None and g['sig'] != sig:
            del self._learn_graphs[key]
            self._learn_eager_steps[key] = 0
            return False
        if g is None:
            try:
                g = self._learn_graphs[key] = self._capture_learn(x, labels, sig)
            except RuntimeError as e:           # a capture that the runtime refuses must not stop training: eager from now on
                import logging
                logging.getLogger(__name__).warning('hipGraph capture of the learning step failed (%s): running the '
                                                    'learning steps eagerly', e)
                self.graph_learn = False
                return False
        g['x'].copy_(x)
        g['y'].copy_(labels)
        tensors = [t for s in self.dcll_slices for t in s._adam_tensors()]            # counts the update
        slot = g['n'] % self._DYN_RING
        g['n'] += 1
        if g['events'][slot] is not None:
            g['events'][slot].synchronize()                     # the copy that last read this pinned row is done
        else:
            g['events'][slot] = torch.cuda.Event()
        g['dyn_host_np'][slot, :] = ops.adam_dyn_values(tensors)
        g['dyn'].copy_(g['dyn_host'][slot], non_blocking=True)
        g['events'][slot].record()
        if g['segments'] is None:
            g['graph'].replay()
        else:
            # under ranks: one captured segment per slice, its gradient slab's all-reduce started (eagerly — collectives
            # stay outside the graphs) right behind it so that it runs under the next segment, Adam as a last segment
            from .. import parallel
            pending = []
            for seg, s in zip(g['segments'], self.dcll_slices):
                seg.replay()
                pending.append(parallel.allreduce_slab_begin(s._grad_slab, x.shape[0], global_batch))
            for h in pending:
                parallel.allreduce_slab_end(h)
            g['graph'].replay()
        rec = g['clout'].clone()
        for i, s in enumerate(self.dcll_slices):
            s.iter += 1
            s.dclllayer.weights_written()              # (the replayed Adam launch wrote through raw pointers)
            if g['records'][i]:
                s._clout.append(rec[i])
        return True

    def _capture_learn(self, x, labels, sig):
        from ..dcll.pytorch_libdcll import DCLLClassification
        n_t = sum(len(s._adam_tensors(advance=False)) for s in self.dcll_slices)
        dev = x.device
        g = dict(sig=sig, n=0, x=torch.empty_like(x), y=torch.empty_like(labels),
                 clout=torch.zeros((self.num_layers, x.shape[0]), device=dev, dtype=torch.int32),
                 dyn=torch.zeros(3 * n_t, device=dev), dyn_host=torch.zeros((self._DYN_RING, 3 * n_t)).pin_memory(),
                 events=[None] * self._DYN_RING,
                 records=[isinstance(s, DCLLClassification) for s in self.dcll_slices])
        g['dyn_host_np'] = g['dyn_host'].numpy()
        iters = [s.iter for s in self.dcll_slices]
        from .. import parallel
        ranks = parallel.is_distributed()
        graph = torch.cuda.CUDAGraph()
        segments = [torch.cuda.CUDAGraph() for _ in self.dcll_slices] if ranks else None
        torch.cuda.synchronize(dev)
        try:
            # (thread_local: API calls of other host threads — a data loader pinning memory — do not break the capture)
            if not ranks:
                with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                    learned, _ = self._learn_tails(self._learn_forwards(g['x'], g['y'], clout_rows=g['clout']))
                    assert len(learned) == len(self.dcll_slices)
                    self._finish_learning(learned, advance=False, dyn=g['dyn'])
            else:
                # one graph per slice (the slabs' collectives run between them, outside any capture) + one for Adam,
                # all in one memory pool: a segment's outputs are the next one's inputs
                spikes, pool = g['x'], None
                for i, s in enumerate(self.dcll_slices):
                    with torch.cuda.graph(segments[i], pool=pool, capture_error_mode='thread_local'):
                        spikes, _, _, _, _, l = s._learn_forward_backward(spikes, g['y'], want_loss=False,
                                                                         clout_out=g['clout'][i])
                        assert l
                    pool = segments[i].pool() if pool is None else pool
                with torch.cuda.graph(graph, pool=pool, capture_error_mode='thread_local'):
                    ops.adam_step([t for s in self.dcll_slices for t in s._adam_tensors(advance=False)], dyn=g['dyn'])
        finally:
            for s, it in zip(self.dcll_slices, iters):         # capturing records the launches, it runs nothing
                s.iter = it
        g['graph'], g['segments'] = graph, segments
        return g

    @torch.no_grad()
    def learn_sequence(self, cells, labels):
        """train.py's inner loop `for t in range(T): net.learn(x[t], labels[t])` (reference train.py:249-251) for input
        given as cell indices (T,B) int32 ON THE DEVICE (IQEncoder: iq2spiketrain's quantisation as a kernel) and one
        one-hot label row per sample (B, target) — iq2spiketrain repeats the labels over t.  No host spike encoding, no
        dense (T,B,1,H,W) upload:
          - the burn-in steps (no slice learns before its iter reaches burnin, :691, so the weights are frozen) run on
            the fused all-T sequence kernels when the geometry has them; their argmax is not recorded, exactly like
            DCLLClassification.forward without ignore_burnin (:724);
          - the learning steps run per timestep (the weights change every step) on one-hot planes built on the device,
            a block of timesteps at a time."""
        T, B = cells.shape
        L0 = self.dcll_slices[0].dclllayer
        H, W = L0.im_dims
        # steps during which NO slice learns yet: slice s learns in the step that takes its iter to >= burnin
        nb = max(0, min(T, min(s.burnin - 1 - s.iter for s in self.dcll_slices)))
        if nb > 0 and self.sequence_supported():
            res = self.test_sequence(cells[:nb].contiguous(), collect=False)
            for i, s in enumerate(self.dcll_slices):
                Li = s.dclllayer
                s.set_sequence_result(None, nb, lowhigh=res['lowhigh'][i],
                                      numel=B * Li.out_channels * int(np.prod(Li.output_shape)))
        else:
            nb = 0
        block = max(1, int((1 << 28) // max(1, B * H * W)))        # <= 1 GiB of planes at a time
        for t0 in range(nb, T, block):
            planes = ops.cells_to_planes(cells[t0:t0 + block].contiguous(), H * W)
            for k in range(planes.shape[0]):
                self.learn(planes[k].reshape(B, 1, H, W), labels)

    def _no_vmem(self):
        """Context in which the layers' per-step forward does not write pvmem (the 4th element of its tuple is None):
        net.test discards every layer's tuple except the spikes it chains (reference :182-185), and the un-pooled membrane
        map is 32 KB per sample and layer step of pure HBM write traffic."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            layers = [s.dclllayer for s in self.dcll_slices if isinstance(s.dclllayer, Conv2dDCLLlayer)]
            for L in layers:
                L.__dict__['_skip_vmem'] = True         # (plain attribute: not through nn.Module.__setattr__)
            try:
                yield
            finally:
                for L in layers:
                    L.__dict__.pop('_skip_vmem', None)
        return ctx()

    def test(self, x):
        """One inference timestep in every slice (reference :182-185).  At batches where the host's launch path sets
        the pace the step is replayed from a captured hipGraph (see _test_graphed)."""
        if self._graph_test_ok(x):
            if self._test_graphed(x):
                return
            self._graph_interrupted('test', tuple(x.shape))
        self._test_slices(x)
        if isinstance(x, torch.Tensor):
            key = tuple(x.shape)
            self._test_eager_steps[key] = self._test_eager_steps.get(key, 0) + 1

    def _test_slices(self, x):
        """The layer steps of one inference timestep, slice after slice, with the readout tails DEFERRED: slice l+1 consumes
        slice l's spikes, nobody's readouts, so the split-K passes and finishing launches of all slices run behind the layer
        kernels as ONE dcll_step_readouts_multi call (two launches for the timestep instead of two per slice; results bit
        for bit those of the per-slice calls).  The recorded argmax of every slice is appended to its `clout` afterwards."""
        spikes, pend = x, []
        layers = [s.dclllayer for s in self.dcll_slices]
        with self._no_vmem():
            try:
                bufs = self.__dict__.setdefault('_test_step_bufs', {})
                for i, L in enumerate(layers):
                    L.__dict__['_defer_sink'] = pend
                    if isinstance(L, Conv2dDCLLlayer):      # (one set of output maps per layer, reused from step to step)
                        L.__dict__['_step_bufs'] = bufs.setdefault((i, tuple(x.shape)), {})   # (per geometry: a capture bakes them in)
                for s in self.dcll_slices:
                    spikes, _, _, _ = s.forward(spikes, ignore_burnin=True)
                ops.run_deferred_readouts([fin for _, fin, _ in pend])
                for s, fin, pos in pend:
                    s._clout[pos] = fin['clout']
            except BaseException:
                for s, _, pos in reversed(pend):           # (no placeholder of an unfinished step stays behind)
                    if pos < len(s._clout) and s._clout[pos] is None:
                        del s._clout[pos]
                raise
            finally:
                for L in layers:
                    L.__dict__.pop('_defer_sink', None)
                    L.__dict__.pop('_step_bufs', None)
        return spikes

    # -- the inference timestep as a captured hipGraph ------------------------------------------------------------------
    # Same idea as _learn_graphed, without an optimizer: the launches of `net.test(x[t])` (three layer steps, readouts,
    # the per-step argmax) are captured once per input geometry on a static input buffer; a replay is followed by one
    # copy of the three argmax rows.  Only used where the eager loop is host-bound (_graph_small: measured
    # 0.25 -> 0.21 ms per timestep at B = 128, but 0.24 -> 0.25 at B = 256, profiles/r02_per_step_small_batches.txt); the
    # every-20th-step pv statistics run eagerly.

    def _graph_test_ok(self, x):
        if not (self.graph_learn and isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32):
            return False
        if self._test_eager_steps.get(tuple(x.shape), 0) < 2:      # state and lazily built caches exist after eager steps
            self._graph_interrupted('test', tuple(x.shape))
            return False
        for s in self.dcll_slices:
            if not isinstance(s, DCLLClassification) or not isinstance(s.dclllayer, Conv2dDCLLlayer):
                return False
            if s.collect_stats and (s.iter + 1) % 20 == 0:
                self._graph_interrupted('test', tuple(x.shape))
                return False
        return self._graph_tuned('test', x, tuple(x.shape))

    def _test_signature(self):
        sig = []
        for s in self.dcll_slices:
            L = s.dclllayer
            sig += [t.data_ptr() for t in L.i2h.state]
            sig += [t.data_ptr() for t in (L.i2h.weight, L.i2h.bias, L.i2h.alpha, L.i2h.tau_m__dt, L.i2h.alphas,
                                           L.i2h.tau_s__dt, L.i2o.weight, L.i2o.bias)]
            if L.output_layer:
                sig += [L.output_.weight.data_ptr(), L.output_.bias.data_ptr()]
            q8 = L.i2h.int8_weights()            # (a capture taken on the int8 form must not outlive it)
            sig.append(None if q8 is None else (q8[0].data_ptr(), q8[1].data_ptr()))
        return tuple(sig)

    @torch.no_grad()
    def _test_graphed(self, x):
        key = tuple(x.shape)
        sig = self._test_signature()
        g = self._test_graphs.get(key)
        if g is not None and g['sig'] != sig:
            del self._test_graphs[key]
            self._test_eager_steps[key] = 0
            return False
        if g is None:
            g = dict(sig=sig, x=torch.empty_like(x), n=0)
            g['x'].copy_(x)
            iters = [s.iter for s in self.dcll_slices]
            hist = [len(s.activity_hist) for s in self.dcll_slices]
            lens = [len(s._clout) for s in self.dcll_slices]
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize(x.device)
            try:
                with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                    self._test_slices(g['x'])
                    g['clout'] = torch.stack([s._clout[-1] for s in self.dcll_slices])
            except RuntimeError as e:
                import logging
                logging.getLogger(__name__).warning('hipGraph capture of the inference step failed (%s): running the '
                                                    'steps eagerly', e)
                self.graph_learn = False
                return False
            finally:                                   # capturing records the launches, it runs nothing
                for s, it, nh, nc in zip(self.dcll_slices, iters, hist, lens):
                    s.iter = it
                    del s.activity_hist[nh:]
                    del s._clout[nc:]
            g['graph'] = graph
            self._test_graphs[key] = g
        g['x'].copy_(x)
        g['graph'].replay()
        g['n'] += 1
        rec = g['clout'].clone()
        for i, s in enumerate(self.dcll_slices):
            s.iter += 1
            s._clout.append(rec[i])
        return True

    def reset(self, init_states=False):
        for s in self.dcll_slices:
            s.init(self.batch_size, init_states=init_states)

    def write_stats(self, writer, epoch, comment=''):
        for s in self.dcll_slices:
            s.write_stats(writer, label='test' + comment, epoch=epoch)

    def accuracy(self, labels):
        return [s.accuracy(labels) for s in self.dcll_slices]

    def confusion_matrix(self, labels):
        return self.dcll_slices[-1].confusion_matrix(labels)

    # -- whole-sequence fast path -----------------------------------------------------------------------------------
    def sequence_supported(self):
        """True if every layer has a fused all-T kernel: radio_ml_conv.yaml on the 16x16 I/Q plane of the reference's
        scripts, or on a plane with H % 8 == 0 and W % 32 == 0 (the 128x128 argparse default; there the pv buffer of one
        layer is T*B*32*H*W*4 bytes — 17 GB at T=128, B=64 — so size the batch for it)."""
        kinds = [s.dclllayer.sequence_kind() for s in self.dcll_slices]
        return kinds[0] == 'cells' and all(k == 'packed' for k in kinds[1:])

    def _sequence_buffers(self, T, B, dev, n_pv=1):
        """Inter-layer spike, pv and logit buffers of one chunk, cached: flat allocations sized for the largest layer
        and the largest batch seen so far at this T, handed out as per-layer (T, B, ...) views (a smaller last chunk
        reuses them).  n_pv = number of pv buffers: 1 (every layer's readout runs before the next layer's kernel
        overwrites it) or one per layer (the readouts run on a second stream under the next layer's kernel)."""
        key = (T, str(dev))
        layers = [s.dclllayer for s in self.dcll_slices]
        npix = [int(np.prod(L.output_shape)) for L in layers]               # (pooled) pixels of a layer's output map
        chans = [L.out_channels for L in layers]
        spk_max = max(max(c * (n // 32), 1) for c, n in zip(chans, npix))     # (every layer gets a view, also the last)
        pv_max = max(c * n for c, n in zip(chans, npix))
        H, W = layers[0].im_dims
        cache = self._seq_buffers.get(key)
        if cache is not None and cache['cap'] >= B:
            while len(cache['pv']) < n_pv:
                cache['pv'].append(torch.empty(T * cache['cap'] * pv_max, device=dev, dtype=torch.float32))
        if cache is None or cache['cap'] < B:
            self._seq_buffers.clear()
            n_ro = [self.target_size * (2 if i == self.num_layers - 1 else 1) for i in range(self.num_layers)]
            cmax = max(L.in_channels for L in layers)
            tiled = (H, W) != (16, 16) and layers[0].i2h.kernel_size == (7, 7)
            cache = dict(cap=B,
                         spk=[torch.empty(T * B * spk_max, device=dev, dtype=torch.int32) for _ in range(2)],
                         pv=[torch.empty(T * B * pv_max, device=dev, dtype=torch.float32) for _ in range(n_pv)],
                         ro=[torch.empty(T * B * n, device=dev, dtype=torch.float32) for n in n_ro],
                         # snapshot area of the tiled 7x7 kernels (planes other than 16x16): initial eps0 / eps1 of a layer
                         state_scratch=(torch.empty(2 * B * cmax * H * W, device=dev, dtype=torch.float32) if tiled else None))
            self._seq_buffers[key] = cache
        return dict(spk=[[t[:T * B * c * (n // 32)].view(T, B, c, n // 32) for t in cache['spk']]
                         for c, n in zip(chans, npix)],
                    pv=[[t[:T * B * L.out_channels * n].view(T, B, L.out_channels, *L.output_shape) for t in cache['pv']]
                        for L, n in zip(layers, npix)],
                    ro=[t[:T * B * (t.numel() // (T * cache['cap']))].view(T, B, -1) for t in cache['ro']],
                    state_scratch=cache['state_scratch'])

    def zero_states(self):
        """Zero every layer's neuron state in place (time constants untouched — unlike reset(True), quirk Q4)."""
        for s in self.dcll_slices:
            for t in s.dclllayer.i2h.state:
                t.zero_()

    @torch.no_grad()
    def test_sequence(self, cells=None, collect=True, profile=None, fuse_readout=False, iq=None, encoder=None,
                      T=None, t0=None, output_only=False, overlap_readout=None, keep_spikes=False, shard=None):
        """Equivalent of `for t in range(T): net.test(x[t])` for input given as cell indices (T,B) int32 on device
        (one input spike per sample per step, what iq2spiketrain produces), or as the raw IQ batch `iq` (B,2,L) with an
        `IQEncoder` — then the quantisation runs inside the first layer's kernel (T steps from sample t0; t0 drawn
        like the host encoder when None).  Fills every slice's `clout`.

        `fuse_readout`: compute the readouts in the layer kernels' epilogue instead of materialising pv + a GEMM.
        Measured slower on MI355X (re-streaming the 786 KB readout matrix per sample-step through L2 costs more than
        the pv round trip through HBM: 124 vs 113 ms per layer launch at B=4096), hence off by default.
        `profile`: optional dict; (start, end) torch.cuda.Event pairs of every launch are appended under
        'lif_c1' / 'lif_c32' / 'readout' / 'vote' (events live on the current stream = the launch stream).

        `output_only` (serving mode, no reference counterpart): only the output layer's prediction is wanted, so the
        hidden layers neither materialise pv nor run their local readouts (their entries in the result are None and
        their `clout` is left untouched) — the spike trains and the output layer's logits / votes are unchanged.

        `overlap_readout` (default: DCLL_OVERLAP_READOUT, off): the layer kernels run on a high-priority stream and
        the HBM-bound work behind each of them — pv statistics, readout GEMM (in its LDS-free <= 64-VGPR form), argmax /
        vote — on the caller's stream, so that it executes in the gaps of the NEXT layer's matrix-bound kernel instead
        of after it; one pv buffer per layer instead of one.  Results are identical up to the readout's summation order.

        `shard` = (start, total): `iq` holds samples start.. of a batch of `total` that is sharded over ranks — the encoder
        then quantises every sample as the reference would at ITS position in the whole batch (torch's vector / scalar pow
        paths, data/utils.py IQEncoder); default: `iq` is the whole batch.

        `keep_spikes`: also return every layer's packed output spike train, 'spikes' (per layer (T,B,C,HW/32) int32,
        bit pix%32 of word pix/32; copies — the working buffers are reused) — for parity checks against the reference.

        Returns a dict with device tensors: 'logits' (per layer, (T,B,target); last entry = output_ layer),
        'clout' (per layer (T,B) int32) and 'vote' (per layer (B) int32)."""
        if overlap_readout is None:
            overlap_readout = os.environ.get('DCLL_OVERLAP_READOUT', '0') != '0'
        if not self.sequence_supported():
            raise ops._lib.DCLLUnsupported('no fused sequence kernel for this network geometry; use net.test(x[t])')
        if iq is not None:
            iq = iq.reshape(iq.shape[0], 2, -1)
            B = iq.shape[0]
            if t0 is None:
                t0 = np.random.randint(0, iq.shape[-1] - T + 1)       # same draw as iq2spiketrain
            dev = iq.device
            if self.dcll_slices[0].dclllayer.i2h.kernel_size != (7, 7):
                # only the 7x7 first-layer kernels have the quantisation fused in: encode with its own (cheap) kernel
                cells, iq = encoder(iq, T, t0=int(t0)), None
                T, B = cells.shape
        if iq is None:
            T, B = cells.shape
            cells = cells.contiguous()
            dev = cells.device
        # Samples are independent, so a batch whose pv buffer (T*B*C*H*W floats per layer) would exceed the budget is
        # run in chunks — on the 128x128 plane T=128 x 512 windows would otherwise need 137 GB for pv alone.
        s0, stot = (0, B) if shard is None else (int(shard[0]), int(shard[1]))
        per_sample = 4 * T * max(s.dclllayer.out_channels * int(np.prod(s.dclllayer.output_shape)) for s in self.dcll_slices)
        chunk = max(1, min(B, int(self.pv_budget_bytes // max(per_sample, 1))))
        if chunk < B and dev.type == 'cuda' and os.environ.get('DCLL_CHUNK_ROUND', '1') != '0':
            # the 32 -> 32 layer kernels run one workgroup per sample and one workgroup per CU: a chunk that is not a multiple
            # of the CU count ends in a partly filled generation.  (The default 24 GiB lands on multiples by itself — 6144
            # windows at T = 128, 768 at T = 1024 — other budgets need not: 20 GB at T = 1024 = 596 = 256 + 256 + 84.)
            ncu = torch.cuda.get_device_properties(dev).multi_processor_count
            if chunk >= ncu:
                chunk -= chunk % ncu
        if chunk < B:
            for s in self.dcll_slices:
                if s.dclllayer.i2h.state is None or s.dclllayer.i2h.state.eps0.shape[0] != B:
                    s.dclllayer.i2h.init_state(B, s.dclllayer.im_dims)
            parts = []
            for b0 in range(0, B, chunk):
                b1 = min(B, b0 + chunk)
                parts.append(self._sequence_chunk(
                    (iq[b0:b1].contiguous(), encoder.thr_i, encoder.thr_q, int(t0),
                     encoder.tail(b1 - b0, s0 + b0, s0 + b1, stot)) if iq is not None
                    else cells[:, b0:b1].contiguous(),
                    'iq' if iq is not None else 'cells', T, b1 - b0, dev, profile, fuse_readout, batch_slice=b0,
                    output_only=output_only, overlap=overlap_readout, keep_spikes=keep_spikes))
            cat = lambda key, dim: [None if parts[0][key][i] is None else torch.cat([p[key][i] for p in parts], dim)
                                    for i in range(self.num_layers)]
            res = dict(logits=cat('logits', 1), clout=cat('clout', 1), vote=cat('vote', 0),
                       lowhigh=[None if parts[0]['lowhigh'][i] is None else sum(p['lowhigh'][i] for p in parts)
                                for i in range(self.num_layers)])
            if 'o' in parts[0]:
                res['o'] = torch.cat([p['o'] for p in parts], 1)
            if keep_spikes:
                res['spikes'] = cat('spikes', 1)
        else:
            res = self._sequence_chunk((iq, encoder.thr_i, encoder.thr_q, int(t0), encoder.tail(B, s0, s0 + B, stot))
                                       if iq is not None else cells,
                                       'iq' if iq is not None else 'cells', T, B, dev, profile, fuse_readout,
                                       output_only=output_only, overlap=overlap_readout, keep_spikes=keep_spikes)
        if collect:
            for i, s in enumerate(self.dcll_slices):
                if res['clout'][i] is not None:
                    L = s.dclllayer
                    s.set_sequence_result(res['clout'][i], T, lowhigh=res['lowhigh'][i],
                                          numel=B * L.out_channels * int(np.prod(L.output_shape)), vote=res['vote'][i])
        return res

    def _sequence_chunk(self, first_input, first_kind, T, B, dev, profile, fuse_readout, batch_slice=None,
                        output_only=False, overlap=False, keep_spikes=False):
        """All layers over all T steps for B samples (the whole batch, or rows batch_slice.. of every layer's state)."""
        for s in self.dcll_slices:
            i2h = s.dclllayer.i2h
            if batch_slice is None and (i2h.state is None or i2h.state.eps0.shape[0] != B):
                i2h.init_state(B, s.dclllayer.im_dims)
        overlap = overlap and not fuse_readout
        buf = self._sequence_buffers(T, B, dev, n_pv=self.num_layers if overlap else 1)
        main = torch.cuda.current_stream(dev)
        hot = None
        if overlap:
            # layer kernels on a high-priority stream of their own; the caller's stream keeps everything behind them
            if getattr(self, '_hot_stream', None) is None or self._hot_stream.device != torch.device(dev):
                self._hot_stream = torch.cuda.Stream(device=dev, priority=-1)
                self._ro_done = {}
            hot = self._hot_stream
            hot.wait_stream(main)                   # inputs and neuron state were prepared on the caller's stream

        def timed(key, fn, *a, **kw):
            if profile is None:
                return fn(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            profile.setdefault(key, []).append((e0, e1))
            return out

        cur = first_input
        res = dict(logits=[], clout=[], vote=[], lowhigh=[])
        if keep_spikes:
            res['spikes'] = []
        for i, s in enumerate(self.dcll_slices):
            L = s.dclllayer
            last = (i == self.num_layers - 1)
            fused = fuse_readout and i > 0
            hidden_skip = output_only and not last
            lbuf = dict(spk=buf['spk'][i][i & 1], pv=buf['pv'][i][i if overlap else 0], ro=buf['ro'][i],
                        state_scratch=buf['state_scratch'])
            # pv statistics of the reference's histogram steps (:658-661) for slices that collect them; counted from
            # the slice's iteration count as T calls of forward() would
            stats_iter0 = s.iter if (s.collect_stats and not hidden_skip) else None
            kind = first_kind if i == 0 else 'packed'
            # pv written before the sigmoid when this layer's readout is the act-capable GEMM (not fused, not co-resident)
            Wt, bias = L.stacked_readout()
            pv_view = lbuf['pv'].reshape(T * B, -1)
            seq_ro = (not overlap and not fused and not hidden_skip and ops.readout_act_supported(pv_view, Wt))
            presig = seq_ro and (self.presigmoid == '1' or (self.presigmoid == 'auto' and L.pooling != (1, 1)))
            if overlap:
                if i in self._ro_done:
                    hot.wait_event(self._ro_done[i])    # the previous readout of this layer's pv buffer has finished
                with torch.cuda.stream(hot):
                    spk, pv, ro = timed('lif_c1' if i == 0 else 'lif_c32', L.forward_sequence, cur, T, B, kind,
                                        want_spikes=(not last) or keep_spikes, buffers=lbuf, fuse_readout=False,
                                        batch_slice=batch_slice, want_pv=not hidden_skip, lowhigh_iter0=None)
                    done = torch.cuda.Event()
                    done.record(hot)
                main.wait_event(done)
                if stats_iter0 is not None:
                    lbuf['lowhigh'] = ops.pv_lowhigh(pv.reshape(T, -1), T, stats_iter0)
            else:
                spk, pv, ro = timed('lif_c1' if i == 0 else 'lif_c32', L.forward_sequence, cur, T, B, kind,
                                    want_spikes=(not last) or keep_spikes, buffers=lbuf,
                                    fuse_readout=fused and not hidden_skip,
                                    batch_slice=batch_slice, want_pv=not hidden_skip, lowhigh_iter0=stats_iter0,
                                    presigmoid=presig)
            res['lowhigh'].append(lbuf.get('lowhigh'))
            if keep_spikes:
                res['spikes'].append(spk.clone())
            if hidden_skip:
                res['logits'].append(None)
                res['clout'].append(None)
                res['vote'].append(None)
                cur = spk
                continue
            if fused:
                # readout(s) came out of the layer kernel's epilogue: (T,B,24) or, on the output layer, (T,B,48)
                p = ro[..., :self.target_size]
                logits = p
                if last:
                    logits = ro[..., self.target_size:]
                    res['o'] = logits
            else:
                # readout GEMM over all (t, b) rows; on the output layer i2o and output_ share ONE pass over pv
                pv2d = pv.reshape(T * B, -1)
                if seq_ro:      # kernel form chosen by K alone: a row's logits do not depend on the chunking
                    ro = timed('readout', ops.readout_act, pv2d, Wt, bias, out=buf['ro'][i].reshape(T * B, -1),
                               presigmoid=presig, scratch=self._seq_scratch)
                else:
                    ro = timed('readout', ops.readout, pv2d, Wt, bias, out=buf['ro'][i].reshape(T * B, -1),
                               mode=ops.READOUT_CORESIDENT if overlap else ops.READOUT_AUTO)
                if overlap:
                    self._ro_done[i] = torch.cuda.Event()
                    self._ro_done[i].record(main)
                ro = ro.reshape(T, B, -1)
                p = ro[..., :self.target_size]
                logits = p
                if last:
                    logits = ro[..., self.target_size:]
                    res['o'] = logits
            clout, vote = timed('vote', ops.argmax_vote, logits.contiguous())
            # a chunk's logits live in the shared per-chunk buffers: copy them out before the next chunk reuses them
            res['logits'].append(p.clone() if batch_slice is not None else p)
            res['clout'].append(clout)
            res['vote'].append(vote)
            cur = spk
        if overlap:
            main.wait_stream(hot)                   # the neuron state written back by the last layer kernel
        if batch_slice is not None and 'o' in res:
            res['o'] = res['o'].clone()
        return res
