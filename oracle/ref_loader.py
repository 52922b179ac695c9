"""Load the REFERENCE's Python files by path (authoring container only; /root/reference does not
exist on the GPU box).  TEST INFRASTRUCTURE ONLY -- used by oracle/gen_golden.py to produce the
golden vectors under tests/golden/ and by a few `-m "not gpu"` tests that skip when the
reference tree is absent.

Recipe: never import the reference's packages (their __init__ pull timm/mmcv);
load single files with importlib and satisfy *import-time-only* dependencies that are absent
from this image with empty module objects.  None of the stand-ins takes part in any arithmetic
that a golden vector records: cv2 / pydensecrf are used only by visualisation / CRF code that is
never called, timm contributes initialisers and a registry decorator, and the `bilateralfilter`
module is the reference's own C++ compiled into oracle/_ref (c_oracle.ref_bilateralfilter_batch).
"""
import importlib.util
import os
import sys
import types

import torch

REF = os.environ.get("COSA_REFERENCE", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REF, "models", "PAR.py"))


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _stub(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_cache = {}


def par_module():
    if "par" not in _cache:
        _cache["par"] = _load("_cosa_ref_PAR", "models/PAR.py")
    return _cache["par"]


def seg_helper():
    if "seg" in _cache:
        return _cache["seg"]
    from . import c_oracle
    _stub("cv2")
    dcrf = _stub("pydensecrf")
    dcrf.densecrf = _stub("pydensecrf.densecrf")
    dcrf.utils = _stub("pydensecrf.utils", unary_from_softmax=None)
    _stub("bilateralfilter", bilateralfilter=None, bilateralfilter_batch=c_oracle.ref_bilateralfilter_batch)
    # the reference hard-codes .cuda(); on this CPU-only box it is the identity
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self
    _cache["seg"] = _load("_cosa_ref_seg_helper", "utils/seg_helper.py")
    return _cache["seg"]


def vit_module():
    if "vit" in _cache:
        return _cache["vit"]
    import torch.nn as nn
    _stub("timm")
    _stub("timm.data", IMAGENET_DEFAULT_MEAN=(0.485, 0.456, 0.406), IMAGENET_DEFAULT_STD=(0.229, 0.224, 0.225))
    _stub("timm.models", resnet26d=None, resnet50d=None)
    _stub("timm.models.helpers", load_pretrained=lambda *a, **k: None)
    _stub("timm.models.layers", DropPath=nn.Identity, to_2tuple=lambda x: (x, x) if not isinstance(x, tuple) else x,
          trunc_normal_=nn.init.trunc_normal_)
    _stub("timm.models.registry", register_model=lambda f: f)
    _cache["vit"] = _load("_cosa_ref_vit", "models/vit/vit.py")
    return _cache["vit"]


def conv_head_module():
    if "head" not in _cache:
        _cache["head"] = _load("_cosa_ref_conv_head", "models/decoder/conv_head.py")
    return _cache["head"]


def torch_helper_fns():
    """denormalize_img / PolyWarmupAdamW live in utils/torch_helper.py which imports sklearn+texttable."""
    if "th" not in _cache:
        _stub("texttable", Texttable=object)
        # the file does `from . import misc` (distributed helpers, unused here): give it an empty parent package
        pkg = _stub("_cosa_ref_utils")
        pkg.__path__ = []
        pkg.misc = _stub("_cosa_ref_utils.misc")
        _cache["th"] = _load("_cosa_ref_utils.torch_helper", "utils/torch_helper.py")
    return _cache["th"]


def evaluation_module():
    """utils/evaluation.py (numpy + sklearn.metrics only)."""
    if "eval" not in _cache:
        _cache["eval"] = _load("_cosa_ref_evaluation", "utils/evaluation.py")
    return _cache["eval"]


def dataloader_modules():
    """(transforms, randaug) of the reference's dataloaders/.  mmcv is absent from this image: transforms.py uses it only in
    PhotoMetricDistortion (never called here); randaug.py calls mmcv.solarize in ONE op, RandSolarize -- that single function is
    supplied by the oracle's restatement of mmcv's published one-liner, and the golden cases that went through it are marked."""
    if "dl" not in _cache:
        from . import aug_oracle
        import numpy as np
        _stub("mmcv", solarize=lambda img, thr=128: np.where(img < thr, img, 255 - img).astype(img.dtype))
        _cache["dl"] = (_load("_cosa_ref_transforms", "dataloaders/transforms.py"), _load("_cosa_ref_randaug", "dataloaders/randaug.py"))
    return _cache["dl"]
