"""pytest configuration: registers the `gpu` marker and shared fixture helpers."""
import glob
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
    """liblamp_hip.so is a build artefact (git-ignored): compile it for gfx950 if this checkout has none, or if
    a source is newer.  hipcc cross-compiles without a GPU, so this also works in the CPU-only container."""
    from lamp_amd import build as _build
    if _build.needs_build():
        _build.build()


def load_golden(name):
    """Load one committed fixture -> (dict of torch tensors / python scalars, state_dict)."""
    z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    data, sd = {}, {}
    for k in z.files:
        a = z[k]
        if k.startswith('sd__'):
            sd[k[4:]] = torch.from_numpy(a)
        elif a.dtype.kind in 'US':
            data[k] = str(a)
        elif a.ndim == 0:
            data[k] = a.item()
        else:
            data[k] = torch.from_numpy(a)
    return data, sd


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + '*.npz')))


def max_abs_diff(a, b):
    """max |a-b| treating NaN==NaN as equal; inf if the NaN patterns differ."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    na, nb = torch.isnan(a), torch.isnan(b)
    if not torch.equal(na, nb):
        return float('inf')
    d = (a - b).abs()
    d[na] = 0
    return d.max().item() if d.numel() else 0.0


@pytest.fixture(scope='session')
def has_reference():
    return os.path.isdir('/root/reference/lamp')
