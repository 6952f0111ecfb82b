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


# -- fixture G1x (round 6): layers built with the options ConvNetwork never uses -----------------------------------------------
G1X_FIELDS = ("kind", "cin", "cout", "kh", "kw", "pad_h", "pad_w", "pool_h", "pool_w", "H", "W", "stride", "dilation", "groups",
              "bias", "spiking", "act", "wrp100", "random_tau", "output_layer", "B", "learn")
G1X_CASES = ["stride2_rrp", "dil2_pool2", "stride2_dil2_out", "stride3_k5", "tanh_plain", "tanh_rrp_out", "relu_stride2",
             "nonspiking", "nonspiking_out", "i2h_groups2", "i2h_groups3_s2_rrp", "i2h_nobias", "i2h_nobias_groups2_dil2_tanh",
             "i2h_depthwise", "dense_nobias", "dense_tanh_rrp", "dense_nonspiking"]
G1X_LEARN_CASES = ["learn_stride2_tanh", "learn_dil2_nonspiking", "learn_dense_nobias_tanh"]


def g1x_cfg(g, case):
    """The cfg row of a G1x case (tests/golden/make_golden.py: G1X_CASES) as a namespace; .act_module() builds the activation."""
    import types
    import torch
    c = types.SimpleNamespace(**{k: int(v) for k, v in zip(G1X_FIELDS, g["g1x/%s/cfg" % case])})
    c.wrp = c.wrp100 / 100.0
    c.act_module = lambda: {0: torch.nn.Sigmoid, 1: torch.nn.Tanh, 2: torch.nn.ReLU}[c.act]()
    return c
