"""CPU suite: the C-ABI boundary and the repo layout rules."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "cosa_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b((?:cosa_[a-z0-9_]+|bilateralfilter(?:_batch)?))\s*\(", txt)))


@pytest.fixture(scope="module")
def built_lib():
    from cosa_amd import build
    return build.build_all()


def test_header_declares_the_path():
    syms = _header_symbols()
    for s in ("cosa_cam2mask", "cosa_par_forward", "cosa_cam_minmax_norm", "cosa_bilateralfilter_batch_dev",
              "cosa_dense_energy_forward", "bilateralfilter", "bilateralfilter_batch"):
        assert s in syms


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib)          # loading needs no GPU; nothing is launched
    for s in _header_symbols():
        assert hasattr(lib, s), f"{s} declared in include/cosa_hip.h but not exported"
    assert lib.cosa_abi_version() >= 1


def test_python_binding_matches_header(built_lib):
    from cosa_amd import _C
    assert sorted(_C.declared_symbols()) == _header_symbols()
    _C.lib()                               # resolves every symbol with its signature


def test_product_never_touches_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "cosa_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b|oracle/|liboracle|/root/reference", txt, flags=re.M):
                    bad.append(os.path.join(base, f))
    assert not bad, f"product code references oracle/ or the reference tree: {bad}"


def test_ops_fail_loudly_without_device():
    import torch
    from cosa_amd import _C
    from cosa_amd.utils import seg_helper
    if torch.cuda.is_available():
        pytest.skip("device present")
    with pytest.raises(_C.CosaError):
        seg_helper.cam2mask(torch.zeros(1, 3, 8, 8), torch.tensor([[0, 8, 0, 8]]), torch.zeros(1, 2, 8, 8),
                            torch.ones(1, 2), 0.7, 0.25)
