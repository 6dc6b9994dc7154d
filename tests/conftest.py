import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


import geoformer_amd  # noqa: E402

geoformer_amd.configure_runtime()  # GPU_MAX_HW_QUEUES, before the test process's first HIP call


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Collection order for `pytest -x`: the cheap operator-level comparisons with the oracle first, the reference-generated
# goldens next, the 150k / 550k-point files last, so that a failure late in the run cannot hide the operator evidence.
_FILE_ORDER = [
    "test_oracle_kats", "test_oracle_spconv", "test_oracle_golden", "test_host_logic", "test_model_host_logic",
    "test_gpu_pointops", "test_gpu_spconv", "test_gpu_geodesic", "test_gpu_heads", "test_gpu_dormant_ops",
    "test_gpu_dropin", "test_gpu_bn_train", "test_gpu_voxel_transformer_train", "test_gpu_decoder_train", "test_gpu_unet_exec", "test_gpu_model",
    "test_criterion_golden",
    "test_training_golden", "test_gpu_feeder", "test_gpu_serving", "test_route_a", "test_parallel_gloo",
    "test_training_step", "test_gpu_fullsize",
]


def pytest_collection_modifyitems(config, items):
    rank = {name: i for i, name in enumerate(_FILE_ORDER)}

    def key(item):
        stem = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return rank.get(stem, len(_FILE_ORDER) - 2)

    items.sort(key=key)  # stable: the order inside a file is kept


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc

    orc.build()
    return orc


@pytest.fixture(scope="session")
def hip():
    """The HIP library must be the thing that runs on a GPU box: fail loudly if it is missing."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from geoformer_amd import _lib

    return _lib.load()
