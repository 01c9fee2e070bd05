"""
The drop-in boundary: libmzx.so loads without a GPU and exports every symbol
include/mzx.h declares (no compute calls here); the host package refuses to
compute without a GPU instead of falling back to anything.
"""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT
from mzx import _lib, configs, models


def header_symbols():
    text = open(os.path.join(ROOT, "include", "mzx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mzx_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(_lib.PROTOTYPES)


def test_library_exports_every_declared_symbol():
    import importlib.util
    spec = importlib.util.spec_from_file_location("mzx_build", os.path.join(ROOT, "muzero-general_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    path = mod.build(verbose=False)
    cdll = ctypes.CDLL(path)
    for name in header_symbols():
        assert hasattr(cdll, name), name
    lib = _lib.Library(path)
    assert lib.mzx_abi_version() == _lib.ABI_VERSION and lib.mzx_is_device_build() == 1
    # struct sizes the C side was compiled with must match the ctypes mirrors
    assert ctypes.sizeof(_lib.NetConfig) == 4 * (7 + 1 + 5 * 9 + 6 + 3 * 9)
    # host-only entry points work without a GPU: configuration validation + weight table
    cfg = models.net_config_from(configs.connect4())
    h = ctypes.c_void_p()
    lib.check(lib.mzx_net_create(ctypes.byref(cfg), ctypes.byref(h)))
    assert lib.mzx_net_num_params(h) == 730681 + 2560  # reference parameters + BatchNorm running stats
    assert lib.mzx_net_hidden_size(h) == 64 * 6 * 7
    lib.mzx_net_destroy(h)
    bad = models.net_config_from(configs.cartpole())
    bad.network = 7
    assert lib.mzx_net_create(ctypes.byref(bad), ctypes.byref(h)) != 0
    assert b"fullyconnected" in lib.mzx_last_error()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_silent_cpu_fallback():
    with pytest.raises(_lib.MzxError):
        models.MuZeroNetwork(configs.cartpole())


def test_config_errors_match_reference():
    with pytest.raises(NotImplementedError):
        models.net_config_from(configs.cartpole(network="transformer"))
    assert models.net_config_from(configs.breakout(downsample="CNN")).downsample == 2
    with pytest.raises(NotImplementedError):        # models.py:327
        models.net_config_from(configs.breakout(downsample="pool"))
