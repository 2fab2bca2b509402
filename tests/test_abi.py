"""The C-ABI library builds for gfx950, loads, and exports every symbol include/nbp_hip.h declares."""
import ctypes
import os
import re

from nextbestpath_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "nbp_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nbp_[a-z0-9_]+)\s*\(", txt)))


def test_library_exists_and_loads():
    assert os.path.exists(_lib.LIB_PATH), "run `python -m nextbestpath_amd.build` (or __graft_entry__.build())"
    ctypes.CDLL(_lib.LIB_PATH)


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared()
    assert len(names) >= 20
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/nbp_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in nextbestpath_amd/_lib.py"
    for n in _lib.SIGNATURES:
        assert n in names, f"{n} bound in _lib.py but not declared in the header"


def test_host_only_queries():
    L = _lib.lib()
    assert L.nbp_abi_version() == 1
    assert L.nbp_packed_weights_bytes() > 49_000_000 * 4
    assert L.nbp_forward_workspace_bytes(1, 256) > 0
    assert L.nbp_forward_workspace_bytes(1, 250) == 0          # S % 16 != 0
    # 2*91.206 GMAC at 256^2
    assert abs(L.nbp_forward_flops(1, 256) / 1e9 - 182.41) < 0.05
    assert abs(L.nbp_forward_flops(1, 128) / 1e9 - 45.60) < 0.02


def test_no_cpu_fallback():
    import pytest
    import torch
    from nextbestpath_amd.networks.nbp_model import NBP
    from nextbestpath_amd.utility import utils
    with torch.device("meta"):
        net = NBP()
    net.eval()
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 5, 32, 32))
    with pytest.raises(RuntimeError):
        utils.map_points_to_n_imgs(torch.zeros(1, 4, 2), (8, 8), (-1, 1))
