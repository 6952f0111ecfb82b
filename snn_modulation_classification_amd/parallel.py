"""Batch sharding over the GPUs of one node.

The DCLL forward path has no cross-sample operation, so every rank (one process per GPU) runs the full network on
a contiguous shard of the batch with replicated weights and NO data-path collective.  The only exchanges are
  - evaluation: one small all-reduce of per-class tallies (confusion matrix [pred, label] + (correct, total) per layer);
  - local learning: one all-reduce per timestep of the gradients of all slices in ONE flat bucket.
RCCL over xGMI on GPUs (backend "nccl"), gloo in the CPU tests and in rehearsals where several ranks share one GPU
(DCLL_DIST_BACKEND=gloo: device tensors are staged through the host, RCCL refuses two ranks on one device).

DCLL_FORCE_DIST=1 makes a ONE-rank job (WORLD_SIZE=1) form its process group and route every exchange through the
backend anyway: on a 1-GPU box that executes the whole RCCL code path — communicator creation with `device_id`, the int64
tally all-reduce, the fp32 gradient bucket, the fp64 MAX of the timings, barrier, destroy — on device tensors, with
results that must equal the plain single-process run bit for bit (tests/test_gpu_multirank.py).

`spawn_local_ranks` is the launcher of the entry points (`bench.py --gpus N`, `test_radio_ml.py --gpus N`,
`train.py --gpus N`) when they are started plainly instead of under torchrun: the parent never touches the GPU.
"""
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist


def env_ranks():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when launched plainly."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


def under_launcher():
    """True if this process was started as one rank of a job (torchrun or spawn_local_ranks)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def local_device(local_rank):
    """GPU index of this rank: LOCAL_RANK, folded onto the GPUs that exist (only matters for rehearsals in which
    several ranks share one GPU; on an 8-GPU node it is the identity)."""
    n = torch.cuda.device_count()
    return local_rank % n if n > 0 else 0


def device_identity(index):
    """Something that names the PHYSICAL device behind cuda:<index> of this process (two processes that see the same GPU
    through different *_VISIBLE_DEVICES masks get the same string): the device's uuid, else its PCI address, else None."""
    if os.environ.get("DCLL_FAKE_DEVICE_ID"):          # (tests of the shared-device refusal on a box without GPUs)
        return os.environ["DCLL_FAKE_DEVICE_ID"]
    try:
        p = torch.cuda.get_device_properties(index)
    except Exception:                                   # noqa: BLE001
        return None
    u = getattr(p, "uuid", None)
    if u is not None:
        return str(u)
    pci = tuple(getattr(p, n, None) for n in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    return None if any(v is None for v in pci) else "pci-%04x:%02x:%02x" % pci


def ranks_sharing_a_device(identities):
    """identities: per rank (hostname, device identity or None).  -> sorted ranks that share a physical device with another
    rank (identities of None are unknown and never counted)."""
    seen = {}
    for r, ident in enumerate(identities):
        if ident is not None and ident[1] is not None:
            seen.setdefault(tuple(ident), []).append(r)
    return sorted(r for rs in seen.values() if len(rs) > 1 for r in rs)


def _refuse_shared_devices(rank, world, device_index):
    """RCCL hangs in its rendezvous when two ranks sit on one physical device.  A *_VISIBLE_DEVICES mask hides that from
    device_count(): a per-rank mask (every rank sees ONE device, each a different one) is fine, a job-wide mask such as
    HIP_VISIBLE_DEVICES=0 under torchrun maps every rank to the same GPU.  The ranks therefore compare what is behind
    their device in a short gloo pre-rendezvous (host name + device uuid / PCI address) before RCCL is touched; any two
    ranks on one device end the job with a message instead of a hang."""
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        ids = [None] * world
        dist.all_gather_object(ids, (socket.gethostname(), device_identity(device_index)))
    finally:
        dist.destroy_process_group()
    shared = ranks_sharing_a_device(ids)
    if shared:
        sys.exit("rank %d: ranks %s share a physical GPU (%s) — RCCL cannot run several ranks on one device; give every "
                 "rank its own GPU, or set DCLL_DIST_BACKEND=gloo for a rehearsal" % (rank, shared, ids[shared[0]][1]))


def freeze_startup_heap():
    """gc.collect() + gc.freeze(): move everything allocated so far (torch's import-time heap, the network) into the
    permanent generation, so that a full pass of CPython's cyclic collector inside a per-timestep loop walks only what
    the loop itself allocated.  Measured on the MI355X box (experiments/per_step_outliers.py): a generation-2 pass over
    the unfrozen heap of a bench / train process takes ~100 ms (265 k tracked objects) and, falling into a window of 48
    timesteps, reads as +2 ms per timestep; frozen, the same pass takes < 20 ms.  The T-loops of train.py /
    test_radio_ml.py / bench.py call it once after set-up.  Idempotent; DCLL_GC_FREEZE=0 turns it off."""
    if os.environ.get("DCLL_GC_FREEZE", "1") == "0":
        return
    import gc
    gc.collect()
    gc.freeze()


def forced():
    """DCLL_FORCE_DIST=1: a one-rank job still forms its group and reduces through the backend (module docstring)."""
    return os.environ.get("DCLL_FORCE_DIST", "0") == "1"


def init_process_group(backend=None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.  nccl == RCCL under ROCm."""
    rank, local_rank, world = env_ranks()
    if (world > 1 or forced()) and not dist.is_initialized():
        if backend is None:
            # DCLL_DIST_BACKEND=gloo lets the multi-rank path be rehearsed where RCCL cannot run (CPU container, or
            # several ranks sharing one GPU on a 1-GPU box)
            backend = os.environ.get("DCLL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            # RCCL refuses two ranks on one device (and hangs the others in the rendezvous): say so before joining
            # (a launcher that hands every rank its own GPU through *_VISIBLE_DEVICES leaves every rank ONE device, each a
            #  different one — which is checked, not assumed: _refuse_shared_devices)
            n_local = int(os.environ.get("LOCAL_WORLD_SIZE", world))
            masked = any(os.environ.get(v) is not None for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES",
                                                                 "CUDA_VISIBLE_DEVICES"))
            if torch.cuda.device_count() < n_local and not masked:
                sys.exit("rank %d: %d local rank(s) over RCCL but this process sees %d GPU(s) — use --gpus <= %d, or "
                         "DCLL_DIST_BACKEND=gloo for a rehearsal in which ranks share devices"
                         % (rank, n_local, torch.cuda.device_count(), max(torch.cuda.device_count(), 1)))
            if torch.cuda.device_count() < n_local and world > 1:
                # fewer visible devices than local ranks under a mask: per-rank masks, or one mask for the whole job?
                _refuse_shared_devices(rank, world, local_device(local_rank))
            torch.cuda.set_device(local_device(local_rank))
            kw["device_id"] = torch.device("cuda", local_device(local_rank))
        # Both backends announce themselves with a printf on STDOUT while the group / the communicator forms (gloo its
        # connections, RCCL a five-line version banner — seen on the 1-GPU box with the one-rank RCCL group); stdout is
        # reserved for the job's report (rank 0's one JSON line), so fd 1 points at stderr until the first collective
        # has run (RCCL creates its communicator lazily when no device_id is given).
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
            if backend == "nccl":
                warm = torch.zeros(1, device=kw["device_id"])
                dist.all_reduce(warm)
                torch.cuda.synchronize(kw["device_id"])
            dist.barrier()
        finally:
            sys.stdout.flush()
            try:                                # the banner sits in libc's stdout buffer when fd 1 is a pipe: push it
                import ctypes                   # out while fd 1 still points at stderr
                ctypes.CDLL(None).fflush(None)
            except OSError:
                pass
            os.dup2(saved, 1)
            os.close(saved)
    return rank, local_rank, world


def is_distributed():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or forced())


def all_reduce_(t, op=None):
    """In-place all-reduce of `t` over the ranks (no-op for a single process).  Under the gloo rehearsal backend a
    device tensor is staged through the host (gloo's device support is not relied upon); under RCCL it is reduced
    where it lives."""
    if not is_distributed():
        return t
    op = dist.ReduceOp.SUM if op is None else op
    if t.is_cuda and dist.get_backend() == "gloo":
        host = t.detach().cpu()
        dist.all_reduce(host, op=op)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=op)
    return t


def barrier():
    if is_distributed():
        dist.barrier()


def job_timing(dt, dt_up, allreduce_ms, busy_ms, steps, device, device_label):
    """The timing part of a multi-rank bench line (bench.py; contract: the job's time is the MAX over the ranks' clocks).
    dt / dt_up: this rank's wall time of the K timed steps (device-resident / upload-inclusive), allreduce_ms / busy_ms:
    its per-step HIP-event time of the tally all-reduce / of its kernels.  -> (dt, dt_up, report): the maxima over the ranks
    and a dict with the min / max of every rank's own figures — a scaling loss can then be put on a straggler rank (per-rank
    spread), on the collective, or on the host (wall >> device busy on every rank) — the ranks seen, the backend, and what is
    behind every rank's device.  ONE fp64 MAX all-reduce (minima travel negated) + one all_gather_object; `device` is where
    the reduced vector lives (the rank's GPU under RCCL; the CPU in the gloo rehearsals of tests/test_parallel_gloo.py)."""
    own = 1e3 * dt / max(1, steps)
    vec = torch.tensor([dt, dt_up, -own, own, -allreduce_ms, allreduce_ms, -busy_ms, busy_ms], device=device,
                       dtype=torch.float64)
    all_reduce_(vec, op=dist.ReduceOp.MAX)
    vec = vec.cpu()
    rep = {"per_rank_ms_per_step": {"min": -float(vec[2]), "max": float(vec[3]), "this_rank0": own},
           "allreduce_ms_per_step": {"min": -float(vec[4]), "max": float(vec[5]),
                                     "what": "HIP events around the tally all-reduce (the step's only collective), per step"},
           "device_busy_ms_per_step": {"min": -float(vec[6]), "max": float(vec[7]),
                                       "what": "sum of the HIP-event times of a rank's layer kernels, readouts and votes per step"},
           "ranks_seen": dist.get_world_size(), "backend": dist.get_backend()}
    try:
        ids = [None] * dist.get_world_size()
        dist.all_gather_object(ids, device_label)
        rep["rank_devices"] = ids
        rep["distinct_devices"] = len(set(ids))
    except Exception as e:                      # noqa: BLE001  (diagnostics must never cost the line)
        rep["rank_devices"] = "%s: %s" % (type(e).__name__, e)
    return float(vec[0]), float(vec[1]), rep


def shard_range(total, rank, world):
    """Contiguous, balanced [start, stop) of `total` samples for `rank` (first total % world ranks get one more)."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def tallies(votes, labels, n_classes):
    """votes: list (per layer) of (B) int tensors; labels (B) int.  -> int64 tensor (L, n*n + 2):
    flattened confusion matrix [pred, label] followed by (correct, total)."""
    labels = labels.to(torch.int64)
    if labels.is_cuda and len(votes) <= 16 and n_classes <= 96 and all(v.is_cuda and v.dtype == torch.int32 and
                                                                      v.is_contiguous() for v in votes):
        from . import ops                           # one HIP launch instead of nine torch kernels per layer
        return ops.vote_tallies(list(votes), labels.contiguous(), n_classes)
    rows = []
    for v in votes:
        v = v.to(torch.int64)
        cm = torch.bincount(v * n_classes + labels, minlength=n_classes * n_classes)
        rows.append(torch.cat([cm, torch.stack([(v == labels).sum(), torch.tensor(v.numel(), device=v.device)])]))
    return torch.stack(rows)


def allreduce_tallies(t):
    """Sum the tallies of all shards (no-op for a single process)."""
    return all_reduce_(t)


def split_tallies(t, n_classes):
    """-> (confusion (L,n,n), accuracy (L) float)"""
    cm = t[:, :n_classes * n_classes].reshape(-1, n_classes, n_classes)
    acc = t[:, -2].double() / t[:, -1].clamp(min=1).double()
    return cm, acc


def allreduce_mean_tensors(tensors, local_n=None):
    """Replace every tensor (the gradients of local mean losses) by its global-batch value: ONE flat bucket, one
    all-reduce.  With `local_n` (samples of this rank's shard) the shards are weighted by their size, so ragged shards
    give exactly the mean over the global batch; without it the shards are taken to be equal.  No-op for one process."""
    tensors = [t for t in tensors if t is not None]
    if not is_distributed() or not tensors:
        return
    # (one rank: its shard IS the global batch, weight 1 — the bucket still travels through the backend)
    w = float(local_n) if (local_n is not None and dist.get_world_size() > 1) else 1.0
    flat = torch.cat([t.reshape(-1) for t in tensors] + [torch.ones(1, device=tensors[0].device, dtype=tensors[0].dtype)])
    flat *= w
    all_reduce_(flat)
    flat /= flat[-1].clone()
    off = 0
    for t in tensors:
        t.copy_(flat[off:off + t.numel()].view_as(t))
        off += t.numel()


def allreduce_slab_begin(slab, local_n, global_n=None):
    """Start the all-reduce of ONE slice's gradient slab — a flat fp32 buffer whose views are the slice's .grad tensors,
    followed by one count element — and return a handle for allreduce_slab_end; the caller goes on enqueueing the next
    slice's kernels, which run under the collective (RCCL works on its own stream and orders itself behind the kernels
    already enqueued on the current one).  The slab holds the gradient of the LOCAL mean loss: it is weighted by the
    shard's share of the global batch first — local_n / global_n when the caller knows the global batch (exact for
    power-of-two rank counts, no division afterwards), else local_n with the count element carrying the weights' sum.
    One rank: nothing to weigh, the slab still travels through the backend.  None when not distributed."""
    if not is_distributed():
        return None
    world = dist.get_world_size()
    counted = False
    if world > 1:
        if global_n:
            slab.mul_(float(local_n) / float(global_n))
        else:
            slab[-1] = 1.0
            slab.mul_(float(local_n))
            counted = True
    if slab.is_cuda and dist.get_backend() == "gloo":      # rehearsal backend: staged through the host, synchronous
        host = slab.detach().cpu()
        dist.all_reduce(host)
        slab.copy_(host)
        return (None, slab, counted)
    return (dist.all_reduce(slab, async_op=True), slab, counted)


def allreduce_slab_end(handle):
    """Make the current stream wait for a slab's collective (no host sync under RCCL) and finish its weighting."""
    if handle is None:
        return
    work, slab, counted = handle
    if work is not None:
        work.wait()
    if counted:
        slab.div_(slab[-1].clone())


def allreduce_mean_grads(params, local_n=None):
    """Average .grad over the ranks (the local losses are means over the LOCAL batch, so the global-batch gradient is
    the shard-size-weighted mean of the shard gradients).  One flat all-reduce per call; no-op for a single process."""
    allreduce_mean_tensors([p.grad for p in params], local_n)


# ---------------------------------------------------------------------------------------------------------------------
# launcher
# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _is_amd_compute_node(name):
    """A DRM render node counts as a GPU of this job only if its PCI vendor is AMD (0x1002) — an iGPU or display adapter of
    another vendor, or a virtual render node, is no place for an RCCL rank.  Where sysfs does not say (no `device/vendor`
    entry: some containers mount /dev/dri without /sys/class/drm) the node is kept."""
    try:
        with open(os.path.join("/sys/class/drm", name, "device", "vendor")) as f:
            return int(f.read().strip(), 16) == 0x1002
    except (OSError, ValueError):
        return True


def _kfd_gpu_count():
    """GPU agents the compute driver (amdkfd) exposes: topology nodes with SIMDs (CPU nodes have simd_count 0), or None
    where the topology cannot be read."""
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for d in os.listdir(root):
            with open(os.path.join(root, d, "properties")) as f:
                for line in f:
                    if line.startswith("simd_count"):
                        n += int(line.split()[1]) > 0
                        break
        return n
    except (OSError, ValueError, IndexError):
        return None


def visible_gpu_count():
    """GPUs this job may use, counted WITHOUT the HIP runtime: the AMD DRM render nodes this process can open (a container
    that was given one GPU of an 8-GPU host has one), not more than the compute driver's topology lists, narrowed by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES.  Only where there is no /dev/dri torch's own
    device query is asked — the launcher must not bring a runtime up.  (An over-count would start RCCL ranks on fewer
    compute GPUs than ranks; every rank therefore re-checks against its runtime: local_device.)"""
    try:
        nodes = [d for d in os.listdir("/dev/dri") if d.startswith("renderD")]
    except OSError:
        return torch.cuda.device_count()
    n = sum(1 for d in nodes if os.access(os.path.join("/dev/dri", d), os.R_OK | os.W_OK) and _is_amd_compute_node(d))
    kfd = _kfd_gpu_count()
    if kfd:                                       # (0 / None: topology hidden from this container — no information)
        n = min(n, kfd)
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def spawn_local_ranks(n, argv=None, timeout=None):
    """Start `n` fresh processes of the running script (one rank per GPU of this node) with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, wait for them and return the exit code (0 iff every rank exited 0).
    Rank 0 inherits stdout (its one JSON line / report is the job's output); the other ranks' stdout goes to stderr.
    The CALLER must not have initialised the GPU: it only launches and waits (never re-execs).  If the node has fewer
    GPUs than ranks the job is a rehearsal: ranks share GPUs and the backend falls to gloo unless DCLL_DIST_BACKEND
    says otherwise."""
    argv = list(sys.argv if argv is None else argv)
    env = dict(os.environ)
    env.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    n_dev = visible_gpu_count()
    if n_dev < n and "DCLL_DIST_BACKEND" not in env:
        env["DCLL_DIST_BACKEND"] = "gloo"
        print("[launcher] %d ranks on %d GPU(s): rehearsal, ranks share devices, backend gloo" % (n, n_dev),
              file=sys.stderr, flush=True)
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable] + argv, env=e, stdout=None if r == 0 else sys.stderr))
    t0 = time.time()
    rc = 0
    live = set(range(n))
    try:
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print("[launcher] rank %d exited with %d; stopping the other ranks" % (r, code), file=sys.stderr,
                          flush=True)
            if rc != 0 or (timeout is not None and time.time() - t0 > timeout):
                if rc == 0:
                    rc = 124
                break
            if live:
                time.sleep(0.1)
    finally:
        for r in live:                          # exact PIDs of the children this call started
            if procs[r].poll() is None:
                procs[r].terminate()
        for r in live:
            try:
                procs[r].wait(timeout=20)
            except subprocess.TimeoutExpired:
                procs[r].kill()
                procs[r].wait()
    return rc
