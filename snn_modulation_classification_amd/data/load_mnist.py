"""MNIST loader for the DCLL entry points — counterpart of the reference's data/load_mnist.py:5-23, without
torchvision: reads the four IDX files (optionally .gz) from `data_dir` (or its torchvision-style `MNIST/raw`
sub-directory).  Samples are (28, 28) float32 in [0, 1] (what ToTensor + Normalize(0, 1) + view([28, 28]) yield),
labels int64; the loader shuffles for `train=True` and carries the reference's `taskid` / `name` / `short_name`."""
import gzip
import os
import struct

import numpy as np
import torch
from torch.utils.data import DataLoader, TensorDataset

FILES = {True: ("train-images-idx3-ubyte", "train-labels-idx1-ubyte"),
         False: ("t10k-images-idx3-ubyte", "t10k-labels-idx1-ubyte")}


def _open(data_dir, stem):
    for d in (data_dir, os.path.join(data_dir, "MNIST", "raw")):
        for name, op in ((stem, open), (stem + ".gz", gzip.open)):
            path = os.path.join(d, name)
            if os.path.exists(path):
                return op(path, "rb")
    raise FileNotFoundError("MNIST file %s[.gz] not found under %s (or its MNIST/raw); there is no network here to "
                            "download it — place the four IDX files there" % (stem, data_dir))


def read_idx(data_dir, stem):
    """One IDX file -> numpy array (uint8)."""
    with _open(data_dir, stem) as f:
        magic, = struct.unpack(">I", f.read(4))
        ndim = magic & 0xff
        if (magic >> 8) != 0x08:
            raise ValueError("%s: not an unsigned-byte IDX file (magic %#x)" % (stem, magic))
        shape = struct.unpack(">" + "I" * ndim, f.read(4 * ndim))
        data = np.frombuffer(f.read(), dtype=np.uint8)
    if data.size != int(np.prod(shape)):
        raise ValueError("%s: truncated IDX file" % stem)
    return data.reshape(shape)


def get_mnist_loader(batch_size, train, taskid=0, data_dir="./data", **kwargs):
    images, labels = (read_idx(data_dir, stem) for stem in FILES[bool(train)])
    x = torch.from_numpy(images.astype(np.float32) / 255.0)
    y = torch.from_numpy(labels.astype(np.int64))
    loader = DataLoader(TensorDataset(x, y), batch_size=batch_size, shuffle=bool(train))
    loader.taskid = taskid
    loader.name = "MNIST_{}".format(taskid)
    loader.short_name = "MNIST"
    return loader
