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


@pytest.fixture(autouse=True)
def _tuning_defaults():
    """Every test starts from, and leaves behind, the library's default tuning table (include/mzx.h "Tuning")."""
    yield
    lib_path = os.path.join(ROOT, "muzero-general_amd", "mzx", "libmzx.so")
    try:
        with open("/proc/self/maps") as f:
            loaded = "libmzx.so" in f.read()
    except OSError:
        loaded = False
    if not loaded or not os.path.isfile(lib_path):
        return      # (no test of this process has loaded the product library: nothing to restore)
    import ctypes

    cdll = ctypes.CDLL(lib_path)      # the handle of the library already mapped: the process-wide table
    cdll.mzx_tuning_name.restype = ctypes.c_char_p
    cdll.mzx_tuning_name.argtypes = [ctypes.c_int32]
    cdll.mzx_tuning_get.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)]
    cdll.mzx_tuning_set.argtypes = [ctypes.c_char_p, ctypes.c_int32]
    i = 0
    while True:
        name = cdll.mzx_tuning_name(i)
        if name is None:
            break
        dflt = ctypes.c_int32()
        assert cdll.mzx_tuning_get(name, None, ctypes.byref(dflt)) == 0
        assert cdll.mzx_tuning_set(name, dflt.value) == 0
        i += 1
