import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
    """The C-ABI library and the C oracle are build products (git-ignored): a fresh checkout builds them once before the
    first test (hipcc cross-compiles gfx950 without a GPU; `__graft_entry__.build()` does the same)."""
    import shutil
    from neurite_amd import build as nbuild
    if not os.path.exists(nbuild.LIB) and shutil.which(os.environ.get('HIPCC', 'hipcc')):
        nbuild.build()
    from oracle import build as obuild
    if not os.path.exists(obuild.LIB) and shutil.which('gcc'):
        obuild.build()


def load_golden(name):
    """Fixtures produced by tests/golden/make_golden.py from the reference's own source."""
    with np.load(os.path.join(GOLDEN_DIR, name + '.npz'), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def golden_cases(g):
    """Group 'tag__field' keys of an interpn-style fixture into {tag: {field: array}}."""
    cases = {}
    for k, v in g.items():
        if '__' in k:
            tag, field = k.split('__', 1)
            cases.setdefault(tag, {})[field] = v
    return cases


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no ROCm device')
    return torch.device('cuda:0')


def bits_equal(a, b):
    """Bit-for-bit equality of two float arrays, treating every NaN as equal to every NaN."""
    a = np.asarray(a)
    b = np.asarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    if np.issubdtype(a.dtype, np.floating):
        both_nan = np.isnan(a) & np.isnan(b)
        return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | both_nan)) if a.dtype == np.float32 \
            else bool(np.all((a == b) | both_nan))
    return bool(np.array_equal(a, b))
