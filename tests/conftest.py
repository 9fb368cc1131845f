import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

WEIGHTS_DIR = os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dsm():
    import dsm_amd
    return dsm_amd


@pytest.fixture(scope="session")
def lib(dsm):
    """The HIP library, built in-tree if missing (hipcc cross-compiles without a GPU)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "dsm_build", os.path.join(ROOT, "delayed-streams-modeling_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build()
    return dsm.load_library()


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def tiny_weights(dsm):
    from dsm_amd import synth
    cfg = dsm.config_tiny()
    return synth.make_synth_weights(cfg, WEIGHTS_DIR, tag="tiny")


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    if not has_gpu():
        pytest.fail("this test is marked gpu but no HIP device is visible")
    return True
