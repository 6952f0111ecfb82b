import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


class Golden:
    """Lazy view of one tests/golden/*.npz with '/'-separated keys."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name))

    def __getitem__(self, k):
        return self.z[k]

    def sub(self, prefix):
        n = len(prefix)
        return {k[n:]: self.z[k] for k in self.z.files if k.startswith(prefix)}

    def keys(self):
        return self.z.files


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return get


@pytest.fixture(scope="session")
def golden_meta():
    import json
    with open(os.path.join(GOLDEN, "meta.json")) as f:
        return json.load(f)


def unpack_bits(packed, n):
    """Inverse of make_golden.pack_bits: (..., ceil(n/8)) uint8 -> (..., n) float32."""
    return np.unpackbits(packed, axis=-1, bitorder="little")[..., :n].astype(np.float32)


@pytest.fixture(scope="session")
def trained_checkpoint(tmp_path_factory):
    """A parameters_{step}.pth written by the build's train.py on the GPU: radio_ml_conv.yaml trained for 40 batches of
    512 synthetic modulation windows (oracle/trained_parity.py) — the trained-weights parity tests share it."""
    from oracle import trained_parity
    return trained_parity.train_checkpoint(str(tmp_path_factory.mktemp("trained")), steps=40)
