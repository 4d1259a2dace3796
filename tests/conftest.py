import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# stated tolerances: max|delta| <= tol * max|ref|
TOL_ACT = 1e-5      # y, grad_x
TOL_PARAM = 1e-4    # grad_weight_real / grad_weight_imag / grad_bias


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names(kind="layer"):
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "G*.npz")))
    if kind == "layer":
        return [n for n in names if "wirtinger" not in n]
    return [n for n in names if "wirtinger" in n]


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def rel_err(a, ref):
    cplx = np.iscomplexobj(a) or np.iscomplexobj(ref)
    a = np.asarray(a, np.complex128 if cplx else np.float64)
    ref = np.asarray(ref, np.complex128 if cplx else np.float64)
    assert a.shape == ref.shape, (a.shape, ref.shape)
    m = np.abs(ref).max() if ref.size else 0.0
    d = np.abs(a - ref).max() if ref.size else 0.0
    return float(d / m) if m > 0 else float(d)


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    # SMX_TEST_OPTS="nsplit=2;placement=0" reruns the whole GPU suite under non-default plans
    opts = os.environ.get("SMX_TEST_OPTS", "")
    if opts:
        from tensor_cuda_fft_amd import _lib
        for o in filter(None, opts.split(";")):
            k, v = o.split("=")
            _lib.set_option(k, int(v))
    return torch.device("cuda:0")
