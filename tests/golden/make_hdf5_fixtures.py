"""Generates tests/golden/radioml2018/*.hdf5 — small HDF5 files written by the REAL library (h5py 3.3.0 / HDF5 1.10.6)
exactly the way the reference writes and expects them (data/load_radio_ml.py:23-50), so that the build's h5py-free reader
(snn_modulation_classification_amd/data/mini_hdf5.py) is pinned against files it did not produce itself.

Run with an interpreter that has h5py — in this image:  /opt/conda/bin/python3.9 tests/golden/make_hdf5_fixtures.py
(the product's interpreter, /usr/bin/python3, has no h5py: that is why the reader exists).

  gold_mini/GOLD_XYZ_OSC.0001_1024.hdf5   the monolithic layout: 'X' (N,L,2) float32, 'Y' (N,24) int64 one-hot, 'Z' (N,1)
                                          int64; ordered by class, SNRs 28 / 30 interleaved within a class; N = 24*2*3, L = 8
  blocks/class{c}_snr{s}.hdf5             per-(class, SNR) files as `h5f.create_dataset('X', data=...)` writes them (:44-50):
                                          24 classes x SNR 28, 30; 5 examples of L = 8
  late_meta/behind_data.hdf5              metadata BEHIND a data block (round-4 advisor): 'X' (40000, 2) float32 = 320 KB is
                                          written first, then eleven small datasets — their object headers, the second
                                          symbol-table node and the grown heap are allocated after X's data (> 64 KiB into
                                          the file), which a reader that only looks at the head of the file never sees
  expected.npz                            the arrays that went in
"""
import os

import h5py
import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "radioml2018")
rng = np.random.RandomState(2018)
L, PER = 8, 3
os.makedirs(os.path.join(OUT, "gold_mini"), exist_ok=True)
os.makedirs(os.path.join(OUT, "blocks"), exist_ok=True)

# monolithic file
N = 24 * 2 * PER
X = rng.randn(N, L, 2).astype(np.float32)
labels = np.repeat(np.arange(24), 2 * PER)
Y = np.zeros((N, 24), dtype=np.int64)
Y[np.arange(N), labels] = 1
Z = np.tile(np.array([28, 30, 30, 28, 28, 30]), 24)[:, None].astype(np.int64)
with h5py.File(os.path.join(OUT, "gold_mini", "GOLD_XYZ_OSC.0001_1024.hdf5"), "w") as f:
    f.create_dataset("X", data=X)
    f.create_dataset("Y", data=Y)
    f.create_dataset("Z", data=Z)

# per-(class, SNR) blocks
blocks = {}
for c in range(24):
    for snr in (28, 30):
        x = rng.randn(5, L, 2).astype(np.float32)
        blocks["class%d_snr%d" % (c, snr)] = x
        h5f = h5py.File(os.path.join(OUT, "blocks", "class%d_snr%d.hdf5" % (c, snr)), "w")
        h5f.create_dataset("X", data=x)
        h5f.close()
# metadata behind data
os.makedirs(os.path.join(OUT, "late_meta"), exist_ok=True)
late = {"late_X": (np.arange(80000, dtype=np.float32) * 0.25).reshape(40000, 2)}
with h5py.File(os.path.join(OUT, "late_meta", "behind_data.hdf5"), "w") as f:
    f.create_dataset("X", data=late["late_X"])
    for k in range(11):
        late["late_d%d" % k] = (rng.randint(-5, 5, size=(3, k + 1))).astype(np.int64)
        f.create_dataset("d%d" % k, data=late["late_d%d" % k])
np.savez_compressed(os.path.join(OUT, "expected.npz"), X=X, Y=Y, Z=Z, **blocks, **late)
print("h5py %s, HDF5 %s" % (h5py.__version__, h5py.version.hdf5_version))
