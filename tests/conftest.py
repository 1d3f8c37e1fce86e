import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Collection order on the GPU box: kernel-level parity first, then the VAE, the text encoder, the model / session level and
# the two-process runs last - a slow box must never hide the kernel evidence behind end-to-end tests (GPUTEST_r02).
_FILE_ORDER = ["test_kernels_gpu.py", "test_vae_gpu.py", "test_text_encoder_gpu.py", "test_dit_gpu.py",
               "test_depth_gpu.py", "test_context_parallel_gpu.py"]
PER_TEST_TIMEOUT_S = 180


def pytest_collection_modifyitems(config, items):
    def rank(item):
        name = os.path.basename(str(item.fspath))
        return _FILE_ORDER.index(name) if name in _FILE_ORDER else -1     # CPU files keep their place in front
    items.sort(key=rank)                                                   # stable: order inside a file is kept
    if config.pluginmanager.hasplugin("timeout"):
        for item in items:
            if "gpu" in item.keywords and item.get_closest_marker("timeout") is None:
                item.add_marker(pytest.mark.timeout(PER_TEST_TIMEOUT_S))
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _torch_reference_convs_without_miopen():
    """The torch REFERENCE convolutions of the GPU tests (F.conv3d / conv2d in fp32, the oracle graphs evaluated on the
    device) run on torch's native vol2col + GEMM path: on a fresh box MIOpen compiles a kernel per new shape (190 s for
    the full-resolution VAE oracle graph in the r03 dry run).  The product path never calls a torch convolution."""
    with torch.backends.cudnn.flags(enabled=False):
        yield


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def _load(name):
        if name not in cache:
            cache[name] = torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=False)
        return cache[name]

    return _load


def rel_l2(a, b):
    a, b = a.double(), b.double().to(a.device)
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_abs(a, b):
    return float((a.double() - b.double().to(a.device)).abs().max())
