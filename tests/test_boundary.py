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


def test_call_surface_matches_the_reference_signatures():
    """SURVEY b-1: every function / constructor of the path keeps the reference's parameters -- same names, same order, same defaults
    (tests/golden/ref_signatures.json, written from the reference's own files by oracle/gen_golden.py:gen_signatures).  cosa_amd may
    append keyword parameters of its own (underscore-prefixed switches, `device=`), never rename or reorder the reference's."""
    import importlib
    import inspect
    import json
    table = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_signatures.json")))
    for key, ref in table.items():
        parts = key.split(".")
        for cut in (2, 3):
            try:
                mod = importlib.import_module("cosa_amd." + ".".join(parts[:cut]))
                break
            except ModuleNotFoundError:
                continue
        obj = mod
        for a in parts[cut:]:
            obj = getattr(obj, a)
        ours = [(n, None if p.default is inspect.Parameter.empty else repr(p.default), p.kind.name)
                for n, p in inspect.signature(obj).parameters.items() if n != "self"]
        assert len(ours) >= len(ref), key
        for (rn, rd, rk), (on, od, ok) in zip(ref, ours):
            assert rn == on, f"{key}: parameter {on!r} where the reference has {rn!r}"
            if rk != "VAR_KEYWORD":
                assert rd == od or (rd is not None and od is not None and eval(rd) == eval(od)), f"{key}.{rn}: default {od} != {rd}"
        for n, d, kind in ours[len(ref):]:
            assert d is not None or kind in ("VAR_KEYWORD", "VAR_POSITIONAL"), f"{key}: extra parameter {n!r} without a default"


def test_dropin_aliases_resolve_the_reference_import_lines():
    """dropin/: the reference's import lines (main.py:13-21) resolve to cosa_amd without a patch"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, 'dropin'); "
            "import utils.misc as utils; from args import get_parser as a, handle_defaults as b; from args_coco import get_parser as c; "
            "from dataloaders import build_dataloader; from evaluation_engine import evaluate; from models import build_model; "
            "from utils import seg_helper, torch_helper; import models.PAR; from utils.rrm_utils import DenseEnergyLoss; "
            "import cosa_amd; assert seg_helper.cam2mask is cosa_amd.utils.seg_helper.cam2mask and models.PAR.PAR is cosa_amd.models.PAR; print('ok')")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr[-2000:]


def test_launcher_flag_table_matches_run_scripts():
    """cosa_amd.args: run_voc.sh / run_coco.sh command lines parse to the reference's effective configuration"""
    from cosa_amd.args import parse
    a, changed = parse(["EXP_VOC", "--work_dir", "/tmp/x", "--dataset", "VOC12", "--voc12_root", "/data/VOC2012", "--max_iters", "32000",
                        "--aux_layer", "-4"])
    assert (a.max_iters, a.aux_layer, a.num_classes, a.batch_size, a.high_thre, a.warmup_iters, a.eval_iters) == (32000, -4, 21, 2, 0.7, 6000, 2000)
    assert a.pretrained is True and a.usepar is False and a.pseudo_scales == [1.0, 0.5, 1.5] and set(changed) == {
        "work_dir", "dataset", "voc12_root", "max_iters", "aux_layer"}
    c, _ = parse(["EXP_COCO", "--work_dir", "/tmp/x", "--dataset", "COCO", "--coco_root", "/data/coco/"])
    assert (c.max_iters, c.aux_layer, c.num_classes, c.batch_size, c.high_thre, c.warmup_iters, c.eval_iters) == (60000, -3, 81, 4, 0.65, 10000, 6000)


def test_launcher_rejects_flags_it_does_not_honour():
    """--camloss_version v2 / v3 (and raw-CAM dumps) parse like in the reference but are not built: the launcher must fail
    loudly (the reference dispatches cam_lossv2 / cam_lossv3_wrap or raises NotImplementedError, main.py:216-224), not train v1 silently"""
    import pytest
    from cosa_amd import args as cosa_args, main as launcher
    ok, _ = cosa_args.parse(["exp"])
    launcher.check_supported(ok)
    for extra in (["--camloss_version", "v2"], ["--camloss_version", "v3", "--segconf_thre", "0.3"], ["--turnon_rawcam"]):
        bad, _ = cosa_args.parse(["exp"] + extra)
        with pytest.raises(NotImplementedError):
            launcher.check_supported(bad)
    thr, changed = cosa_args.parse(["exp", "--eval_threshold_filters", "0.11", "0.25"])
    assert thr.eval_threshold_filters == [0.11, 0.25] and "eval_threshold_filters" in changed


def test_teacher_precision_mode_strings():
    """`set_nograd_precision`: "base", "base-n" (blocks from n on plain fp16) and "base-nmk" (their MLP halves from block k on); the launcher's
    and the trainer's default is the cheapest map with no failing draw on record (profiles/r05_accuracy_teacher.txt); "base-xn[mk]": the
    blocks below n (their MLP halves below k) on bf16x3 operands"""
    import torch
    from cosa_amd.models import build_model
    from cosa_amd.train_step import default_args
    from cosa_amd import args as launcher_args
    from cosa_amd.train_step import resolve_teacher_precision
    a = default_args("VOC12", crop_size=64)
    assert a.teacher_precision == "auto"
    assert [resolve_teacher_precision("auto", c) for c in (224, 448, 512, 640)] == ["fp16x3"] * 4          # (round 6; rounds 5: fp16c8-x2)
    assert resolve_teacher_precision("bf16", 640) == "bf16" and resolve_teacher_precision("auto", 448, usepar=True) == "fp16x3"
    net = build_model(a)
    for mode, hdt in (("bf16x3", torch.bfloat16), ("fp16x3", torch.float16)):          # the three-term path: bf16 halves / fp16 halves (round 6)
        net.set_nograd_precision(mode)
        assert net.encoder.precision == "bf16x3" and net.encoder.x3_dtype == hdt and net.encoder.compute_dtype == hdt, mode
    for mode, prec, dt, plain in (("bf16", None, torch.bfloat16, (12, 12)), ("fp16c8", "fp16c8", torch.float16, (12, 12)),
                                  ("fp16c8-9", "fp16c8", torch.float16, (9, 9)), ("fp16c4-8", "fp16c4", torch.float16, (8, 8)),
                                  ("fp16c4-12m8", "fp16c4", torch.float16, (12, 8)), ("fp16c4-9m7", "fp16c4", torch.float16, (9, 7))):
        net.set_nograd_precision(mode)
        assert net.encoder.precision == prec and net.encoder.compute_dtype == dt and net.encoder._plain_from() == plain, mode
        assert net.encoder._x3_until() == (0, 0)
    for mode, prec, x3 in (("fp16c8-x2", "fp16c8", (2, 2)), ("fp16c8-x6m4", "fp16c8", (6, 4)), ("fp16c4-x0m3", "fp16c4", (0, 3))):      # round 5: early blocks on bf16x3
        net.set_nograd_precision(mode)
        assert net.encoder.precision == prec and net.encoder._x3_until() == x3 and net.encoder._plain_from() == (12, 12), mode
        assert net.encoder.c4_from is None and [net.encoder._is_c4(i) for i in (0, 11)] == [prec == "fp16c4"] * 2
    for mode, x3, c4 in (("fp16c8-x2c6", (2, 2), 6), ("fp16c8-c8", (0, 0), 8)):      # ... and the late blocks' qkv / fc1 / fc2 on fp16c4 operands (mixed maps)
        net.set_nograd_precision(mode)
        assert net.encoder.precision == "fp16c8" and net.encoder._x3_until() == x3 and net.encoder.c4_from == c4
        assert [net.encoder._is_c4(i) for i in range(12)] == [i >= c4 for i in range(12)]
    for bad in ("fp16c4-", "fp16c4-m8", "bf16-3", "fp8", "fp16c4-9m", "fp16c8-x", "bf16x3-x2", "fp16x3-2", "fp16c4-c6", "fp16c8-c", "fp16c8-9c6"):
        with pytest.raises(AssertionError):
            net.set_nograd_precision(bad)


def test_auto_teacher_precision_is_backed_by_the_committed_accuracy_record():
    """`resolve_teacher_precision("auto", ...)` may only name a mode whose every line in the newest committed accuracy record
    (profiles/rNN_accuracy_teacher.txt: fused HIP teacher vs the fp32 CPU oracle, written on the GPU by tests/test_precision_gpu.py) keeps
    BASELINE.json's bars; bench.py reads `tolerance_met` from the same record with the same parser and ties it to the kernel sources"""
    import importlib.util
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from cosa_amd.train_step import resolve_teacher_precision
    src = open(os.path.join(root, "tests", "test_precision_gpu.py")).read()
    names = re.search(r"TEACHER_CSRC = \(([^)]*)\)", src).group(1)
    assert tuple(re.findall(r'"([^"]+)"', names)) == bench.TEACHER_CSRC            # one definition of "the teacher's kernels" on both sides
    assert tuple(re.findall(r'"([^"]+)"', re.search(r"TEACHER_HOST = \(([^)]*)\)", src).group(1))) == bench.TEACHER_HOST
    # the pre-registered criterion: ONE set of constants (the test file's), bench.py reads the record by the same numbers
    for name in ("CAM_BAR", "COND_MAX", "FP64_FACTOR", "AGREE_BAR", "MIOU_BAR", "MIN_DRAWS"):
        assert float(re.search(rf"^{name} = (\S+)", src, re.M).group(1)) == float(getattr(bench, name)), name
    for crop in (224, 448, 640):
        c = bench.conformance(resolve_teacher_precision("auto", crop), crop)
        assert c.get("lines", 0) >= 8 and c["draws_all_crops"] >= bench.MIN_DRAWS, (crop, c)
        assert c["planes_failed"] == 0 and c["label_agreement_min"] >= bench.AGREE_BAR and min(c["mask_miou_pooled"].values()) >= bench.MIOU_BAR, (crop, c)
        assert c["tolerance_met"] or "another tree" in c.get("note", ""), (crop, c)      # (a source change after the record was taken is bench.py's business: it then reports false)
    c = bench.conformance(resolve_teacher_precision("auto", 448), 448)
    assert c["draws"] >= bench.MIN_DRAWS and c["seeds"] >= 32 and 16 in c["batch_sizes"], c          # the wide sweep and the bench's own batch size
    # held out in the strict sense: seeds 300-323 were drawn after the criterion was fixed, 400-439 after the default mode was chosen
    rows, _ = bench.parse_accuracy_record(bench.newest_profile("accuracy_teacher.txt"))
    mode = resolve_teacher_precision("auto", 448)
    late = {r["seed"] for r in rows if r["mode"] == mode and r["S"] == 448 and r["seed"] >= 300}
    assert len(late) >= bench.MIN_DRAWS and not any(r["fail"] for r in rows if r["mode"] == mode and r["seed"] >= 300), len(late)


def test_no_inline_asm_valu_on_mfma_accumulators():
    """round 5: on gfx950 the wait states between an MFMA and a VALU read of its result are inserted by the compiler's hazard recognizer, which
    does not look into asm statements.  `max3f` of the attention forward was `asm("v_max3_f32 ...")` on score accumulators: correct only while
    the schedule happened to keep it far behind the MFMAs, run-to-run non-deterministic in every build that moved it
    (profiles/r05_attn_variants.txt).  Rule since: no VALU instruction in inline asm in the files whose VALU code reads MFMA results, except
    the listed address adds / loads / waits that touch no accumulator."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    allowed = {"attn_kernels.hip": (), "gemm_kernels.hip": ("v_add_u32",), "vit_kernels.hip": (), "split_kernels.hip": ("v_mul_f32",)}   # (split_kernels.hip has no MFMA)
    for f, ok in allowed.items():
        src = open(os.path.join(root, "cosa_amd", "csrc", f)).read()
        src = re.sub(r"//[^\n]*", "", src)          # (comments may quote the old statement)
        ops = set(re.findall(r'asm(?:\s+volatile)?\s*\(\s*"\s*(v_[a-z0-9_]+)', src))
        assert ops <= set(ok), (f, ops - set(ok))
