import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "muzero-general_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the driver)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    from oracle import ref_shim

    if not ref_shim.available():
        skip = pytest.mark.skip(reason="reference tree not present (GPU box)")
        for item in items:
            if "reference" in item.keywords:
                item.add_marker(skip)
    # `pytest tests` on a box without a GPU (or without the built library): skip the gpu-marked tests
    # instead of erroring out of every one of them (`-m gpu` on the GPU box is unaffected)
    import torch

    lib = os.path.join(ROOT, "muzero-general_amd", "mzx", "libmzx.so")
    if not torch.cuda.is_available() or not os.path.isfile(lib):
        why = "no GPU visible" if os.path.isfile(lib) else "libmzx.so not built"
        skip_gpu = pytest.mark.skip(reason=f"gpu test: {why}")
        for item in items:
            if "gpu" in item.keywords:
                item.add_marker(skip_gpu)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
