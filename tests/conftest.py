import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            with np.load(os.path.join(GOLDEN, f"{name}.npz")) as z:  # a dict of arrays: an NpzFile decompresses a member on EVERY access
                cache[name] = {k: z[k] for k in z.files}
        return cache[name]

    return load


@pytest.fixture(autouse=True)
def _poisoned_lds(request):
    """Every GPU test starts with the LDS of all CUs full of quiet NaNs (tma_debug_poison_lds): a kernel that reads LDS words it never
    wrote then fails its comparison instead of passing on whatever finite bytes the previous kernel left there."""
    if request.node.get_closest_marker("gpu") is not None:
        import torch

        if torch.cuda.is_available():
            from three_mlagents_amd import _lib

            _lib.check(_lib.lib().tma_debug_poison_lds(0, _lib.stream_ptr()))
    yield
