import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Load tests/golden/<name>.npz; `*_bf16bits` arrays come back as fp32 tensors holding
    exactly-bf16-representable values under the key without the suffix."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    for k in z.files:
        a = z[k]
        if k.endswith("_bf16bits"):
            t = torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16).float()
            out[k[: -len("_bf16bits")]] = t
        elif a.dtype.kind in "fiu" and a.ndim > 0:
            out[k] = torch.from_numpy(a.copy())
        else:
            out[k] = a
    return out


def rms_rel(ref: torch.Tensor, got: torch.Tensor) -> float:
    """fla's parity metric (fla:ops/utils/testing.py:12-16): RMS(err)/RMS(ref)."""
    ref, got = ref.double().flatten(), got.double().flatten()
    return float((ref - got).square().mean().sqrt() / (ref.square().mean().sqrt() + 1e-12))


@pytest.fixture(scope="session")
def golden():
    return load_golden
