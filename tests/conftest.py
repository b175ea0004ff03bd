import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def model_dir(tmp_path_factory):
    """Synthetic models-DF2K directory (x4.param + fp16-tagged x4.bin, seed 42), shared per user."""
    from realsr_ncnn_vulkan_amd import synth
    root = os.environ.get("RSR_MODELS", "/tmp/rsr_models")
    return synth.make_model_dir(root, "models-DF2K", 42)


@pytest.fixture(scope="session")
def oracle_net(model_dir):
    import oracle
    return oracle.OracleNet(os.path.join(model_dir, "x4.param"), os.path.join(model_dir, "x4.bin"))


@pytest.fixture(scope="session")
def weights():
    from realsr_ncnn_vulkan_amd import synth
    return synth.make_weights(42)
