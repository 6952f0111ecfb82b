"""RadioML loaders for the DCLL entry points.

`get_radio_ml_loader(batch_size, train, data_dir=..., min_snr=6, max_snr=30, per_h5_frac=0.5, train_frac=0.9)` keeps
the call signature and the sample ordering of the reference's data/load_radio_ml.py (:10-132): per (class, SNR) block
the first `per_h5_frac` of the examples is used, split into train / test at `train_frac`, and the blocks are
INTERLEAVED (sample k of block j sits at index j + k * n_blocks), samples shaped (2, 1, L) float32, labels int64.

Block sources, tried in this order inside `data_dir`:
  class{c}_snr{s}.hdf5   the reference's per-(class, SNR) split of RadioML 2018.01A ('X': (n, 1024, 2)), read with h5py
                         when it is installed, else with the build's own minimal reader (data/mini_hdf5.py: contiguous
                         datasets of old-style files, which is what h5py's create_dataset(data=...) writes)
  class{c}_snr{s}.npy    the same blocks as plain numpy files (n, L, 2) — build-specific, h5py-free
  GOLD_XYZ_OSC.0001_1024.hdf5   the monolithic 2018.01A file ('X' (N,1024,2) f32, 'Y' (N,24) one-hot, 'Z' (N,1) SNR): split
                         into per-(class, SNR) blocks on first use like the reference does (:23-50) — as .npy blocks
  RML2016.10a_dict.pkl   RadioML 2016.10a: pickled dict {(modulation, snr): (n, 2, 128)} (11 classes) — the dataset
                         BASELINE.json names; the reference itself cannot read it
"""
import os
import pickle

import numpy as np
import torch
from torch.utils.data import DataLoader, TensorDataset

RML2016_FILES = ("RML2016.10a_dict.pkl", "RML2016.10a_dict.dat")
GOLD_2018 = "GOLD_XYZ_OSC.0001_1024.hdf5"
SPLIT_DONE_BLOCK = (23, 30)             # the reference takes class23_snr30 for "the split has been done" (:23)


def open_hdf5(path):
    """A read-only mapping name -> array-like of the datasets in the root group of `path`: h5py.File when h5py is
    installed, else data/mini_hdf5.MiniHdf5 (numpy memmaps).  The caller slices what it needs."""
    try:
        import h5py
    except ImportError:
        from .mini_hdf5 import MiniHdf5
        return MiniHdf5(path)
    return h5py.File(path, "r")


def split_gold_file(data_dir, log=print):
    """The reference's first-use split of the monolithic RadioML 2018.01A file into per-(class, SNR) blocks
    (data/load_radio_ml.py:23-50: labels = argmax of the one-hot 'Y', SNR = 'Z'[:, 0], one block per class and SNR, examples
    in file order), written as class{c}_snr{s}.npy (the h5py-free block format of this build).  X is sliced class by class
    through a memmap (the file is 20 GB; the data is ordered by class, :31-34)."""
    f = open_hdf5(os.path.join(data_dir, GOLD_2018))
    Y = np.argmax(np.asarray(f["Y"][:]), axis=1)
    Z = np.asarray(f["Z"][:])[:, 0]
    X = f["X"]

    def write(c, snr, arr):
        # written under a temporary name and moved into place: a reader never sees a half-written block, and an interrupted
        # split leaves no file that a later run would take for complete
        path = os.path.join(data_dir, "class%d_snr%d.npy" % (c, int(snr)))
        tmp = "%s.tmp%d" % (path, os.getpid())
        with open(tmp, "wb") as fh:
            np.save(fh, arr)
        os.replace(tmp, path)
        log("Wrote (SNR {z}, class {cl}) data to `{path}`.".format(z=int(snr), cl=c, path=path))
    last = None                                           # the block whose presence says "split done" (:23) goes last
    for c in range(int(Y.max()) + 1):
        idx = np.nonzero(Y == c)[0]
        if len(idx) == 0:
            continue
        lo, hi = int(idx[0]), int(idx[-1]) + 1            # one class = one contiguous stretch of the file
        class_x = np.asarray(X[lo:hi])[idx - lo]
        class_z = Z[idx]
        for snr in np.unique(class_z):
            arr = np.ascontiguousarray(class_x[class_z == snr], dtype=np.float32)
            if (c, int(snr)) == SPLIT_DONE_BLOCK:
                last = arr
            else:
                write(c, snr, arr)
    if last is not None:
        write(SPLIT_DONE_BLOCK[0], SPLIT_DONE_BLOCK[1], last)


def _split_done(data_dir):
    return any(os.path.exists(os.path.join(data_dir, "class%d_snr%d%s" % (SPLIT_DONE_BLOCK + (ext,))))
               for ext in (".hdf5", ".npy"))


def ensure_split(data_dir):
    """The reference's trigger (:23): the monolithic file is there and no class23_snr30 block yet -> split it.  ONE process
    per directory does it: every process — plain ones that share the directory, and every rank under `--gpus N` — takes
    the directory's lock file and looks again once it holds it.  No collective is involved: a multi-minute pass over the
    20 GB file cannot run into the backend's collective timeout, node-local data directories each get their split (from
    whichever local rank arrives first), and a failing split raises on the rank that ran it while the next holder of the
    lock tries — and fails — itself instead of waiting at a barrier nobody reaches.  (Unguarded, N ranks made N passes over
    the file at once, each rewriting the same blocks under the others' np.load.)"""
    if not os.path.exists(os.path.join(data_dir, GOLD_2018)) or _split_done(data_dir):
        return
    import fcntl
    with open(os.path.join(data_dir, ".split.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)                  # (released when the file is closed)
        if not _split_done(data_dir):
            split_gold_file(data_dir)


def _block_2018(data_dir, class_idx, snr):
    """One (class, SNR) block as (n, L, 2) float32, or None if the directory holds no such block."""
    stem = os.path.join(data_dir, "class%d_snr%d" % (class_idx, snr))
    if os.path.exists(stem + ".npy"):
        return np.load(stem + ".npy").astype(np.float32, copy=False)
    if os.path.exists(stem + ".hdf5"):
        return np.asarray(open_hdf5(stem + ".hdf5")["X"][:]).astype(np.float32, copy=False)
    return None


def _blocks_2016(path, min_snr, max_snr):
    """2016.10a pickle -> (class names sorted, {(class_idx, snr): (n, L, 2)})."""
    with open(path, "rb") as f:
        d = pickle.load(f, encoding="latin1")
    mods = sorted({k[0] for k in d})
    blocks = {}
    for (mod, snr), x in d.items():
        if min_snr <= snr <= max_snr:
            blocks[(mods.index(mod), int(snr))] = np.transpose(np.asarray(x, dtype=np.float32), (0, 2, 1))
    return mods, blocks


def load_split(data_dir, train, min_snr=6, max_snr=30, per_h5_frac=0.5, train_frac=0.9, block_size=None):
    """-> (X (N, 2, 1, L) float32, Y (N) int64, n_classes) in the reference's interleaved order."""
    snrs = list(range(min_snr, max_snr + 2, 2))
    pkl = next((os.path.join(data_dir, n) for n in RML2016_FILES if os.path.exists(os.path.join(data_dir, n))), None)
    if pkl is not None:
        mods, table = _blocks_2016(pkl, min_snr, max_snr)
        n_classes = len(mods)
        get = lambda c, s: table.get((c, s))
    else:
        n_classes = 24
        ensure_split(data_dir)                        # the reference's trigger: no class23_snr30 block yet (:23)
        get = lambda c, s: _block_2018(data_dir, c, s)
    blocks = []
    for c in range(n_classes):
        for s in snrs:
            x = get(c, s)
            if x is None:
                raise FileNotFoundError("no RadioML block for class %d, SNR %d under %s" % (c, s, data_dir))
            blocks.append((c, x))
    full = block_size if block_size is not None else min(len(x) for _, x in blocks)
    use = int(per_h5_frac * full)
    n_train = int(train_frac * use)
    lo, hi = (0, n_train) if train else (n_train, use)
    n_blocks, per = len(blocks), hi - lo
    L = blocks[0][1].shape[1]
    X = np.zeros((n_blocks * per, L, 2), dtype=np.float32)
    Y = np.zeros(n_blocks * per, dtype=np.int64)
    for j, (c, x) in enumerate(blocks):
        X[j::n_blocks] = x[lo:hi]
        Y[j::n_blocks] = c
    return np.ascontiguousarray(X.transpose(0, 2, 1))[:, :, None, :], Y, n_classes


def get_radio_ml_loader(batch_size, train, **kwargs):
    X, Y, _ = load_split(kwargs['data_dir'], train, kwargs.get('min_snr', 6), kwargs.get('max_snr', 30),
                         kwargs.get('per_h5_frac', 0.5), kwargs.get('train_frac', 0.9), kwargs.get('block_size'))
    name = 'train' if train else 'test'
    print('[%s] dataset size: %d' % (name, len(X)))
    loader = DataLoader(TensorDataset(torch.from_numpy(X), torch.from_numpy(Y)), batch_size=batch_size, shuffle=train)
    loader.name = 'RadioML_{}'.format(name)
    return loader
