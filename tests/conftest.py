import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "wgrad_independent: a test of a dual-mode module that computes no parameter gradient (or sets both "
                                       "weight-gradient modes itself): it runs once")


def pytest_collection_modifyitems(config, items):
    """GPU tests must never silently pass on a box without a GPU."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# Weight-gradient arithmetic (round 4): the 256x256 weight-gradient blocks run with two bf16 terms per operand by default
# (I2SDF_OPT_WGRAD_BF16X2, i2sdf_amd/config.py); every GPU test module that computes parameter gradients runs twice, once per mode, under
# the same bars -- the engine reads I2SDF_WGRAD_BF16X2 when it is constructed, and worker processes inherit it.
WGRAD_MODE_MODULES = {"test_gpu_backward", "test_gpu_baseline_sizes", "test_gpu_determinism", "test_gpu_network", "test_gpu_training_parity",
                      "test_gpu_training_curve_full", "test_gpu_psnr_ensemble", "test_gpu_dp_equivalence", "test_gpu_loss", "test_gpu_edge_cases", "test_gpu_optim",
                      "test_gpu_eikonal_outputs", "test_gpu_rccl"}


def pytest_generate_tests(metafunc):
    if metafunc.definition.get_closest_marker("wgrad_independent"):
        return
    if metafunc.module.__name__.split(".")[-1] in WGRAD_MODE_MODULES and "wgrad_mode" in metafunc.fixturenames:
        metafunc.parametrize("wgrad_mode", ["wgrad-bf16x2", "wgrad-bf16x3"], indirect=True)


@pytest.fixture(autouse=True)
def wgrad_mode(request, monkeypatch):
    mode = getattr(request, "param", None)
    if mode is not None:
        monkeypatch.setenv("I2SDF_WGRAD_BF16X2", "1" if mode == "wgrad-bf16x2" else "0")
    return mode


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load
