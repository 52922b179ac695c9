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


def test_checkpoint_format_round_trip(tmp_path):
    """SURVEY f-3: files in the reference's checkpoint layout (utils/torch_helper.py:101-117) load strictly (main.py:410-412),
    with the reference's parameter names"""
    import types
    import torch
    from cosa_amd.models import build_model
    from cosa_amd.utils import torch_helper as th
    args = types.SimpleNamespace(model='vit', backbone='vit_base_patch16_224', decoder='LargeFOV', pretrained=False, aux_layer=-3,
                                 num_classes=21, compute_dtype=torch.float32)
    torch.manual_seed(0)
    m1 = build_model(args)
    p = th.save_best(tmp_path, m1, 123, {"miou": 1.0}, {"lr": 6e-5}, 't', comment='seg')
    assert p.endswith("best_seg.pth")
    ck = torch.load(p, map_location="cpu", weights_only=False)
    assert set(ck) == {'s_or_t', 'model', 'epoch', 'args', 'result'} and ck['epoch'] == 123 and ck['s_or_t'] == 't'
    for k in ("encoder.cls_token", "encoder.pos_embed", "encoder.blocks.0.attn.qkv.weight", "encoder.blocks.11.mlp.fc2.bias",
              "encoder.norm.weight", "classifier.weight", "aux_classifier.weight", "decoder.conv8.weight"):
        assert k in ck['model'], k
    torch.manual_seed(1)
    m2 = build_model(args)
    th.load_best(m2, p)
    for (k1, v1), (k2, v2) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
