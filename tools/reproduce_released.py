"""Score a released CoSA checkpoint with the MI355X-native evaluation path and print the reference's score table (SURVEY f-3).

    python tools/reproduce_released.py voc_weights.pth  $HOME/data/VOCdevkit/VOC2012            # README.md:131-132: 76.2 val mIoU
    python tools/reproduce_released.py coco_weights.pth $HOME/data/coco --dataset COCO          #                    51.0

What it does is the reference's `finaleval` (main.py:401-433): build the network with the run script's flags (`--aux_layer -4` for VOC,
run_voc.sh:11), load `ckpt["model"]` STRICT (the state-dict key names are the on-disk contract: encoder.* / decoder.conv6-8.weight /
classifier.weight / aux_classifier.weight) -- or the file itself when it is a bare state dict -- and run `evaluate(..., isfinal=True,
getcrf=True)` on the validation split: the rows CAM / CAM_aux / Seg / Seg_crf of the table the reference prints.  The released `.pth` files
are not in the build environment (no network); tests/test_launcher_gpu.py runs this script on a synthetic checkpoint written in the
reference's dict format (utils/torch_helper.py:101-117) over a tiny VOC-shaped tree, so the path itself is exercised."""
import argparse
import os
import sys
from pathlib import Path

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("checkpoint")
    ap.add_argument("data_root", help="VOC2012 directory (JPEGImages/, SegmentationClassAug/) or the COCO root")
    ap.add_argument("--dataset", default="VOC12", choices=["VOC12", "COCO"])
    ap.add_argument("--name_list_dir", default=None, help="split lists (default: the package's own copy of the reference's lists)")
    ap.add_argument("--work_dir", default="work_dirs")
    ap.add_argument("--crf", default="true", choices=["true", "false"], help="dense-CRF row (Seg_crf) as finaleval prints it")
    ap.add_argument("--extra", nargs=argparse.REMAINDER, default=[], help="further launcher flags, passed through (e.g. --aux_layer -3)")
    opt = ap.parse_args()
    from cosa_amd import args as cosa_args
    from cosa_amd import main as launcher
    from cosa_amd.dataloaders import build_test_loader
    from cosa_amd.evaluation_engine import evaluate
    from cosa_amd.models import build_model
    argv = ["REPRODUCE", "--work_dir", opt.work_dir, "--dataset", opt.dataset, "--pretrained", "false"]
    argv += ["--voc12_root" if opt.dataset == "VOC12" else "--coco_root", opt.data_root]
    if opt.dataset == "VOC12":
        argv += ["--aux_layer", "-4"]                              # run_voc.sh:11 (the released VOC weights were trained with it)
    if opt.name_list_dir:
        argv += ["--name_list_dir", opt.name_list_dir]
    args, _ = cosa_args.parse(argv + opt.extra)
    launcher.check_supported(args)
    args.rank, args.world_size, args.gpu, args.distributed = 0, 1, 0, False
    out_dir = Path(args.work_dir) / args.name
    out_dir.mkdir(parents=True, exist_ok=True)
    args.output_dir = out_dir
    torch.cuda.set_device(0)
    ckpt = torch.load(opt.checkpoint, map_location="cpu", weights_only=False)
    state = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt and isinstance(ckpt["model"], dict) else ckpt
    state = {k[len("module."):] if k.startswith("module.") else k: v for k, v in state.items()}      # (a DDP-wrapped save)
    model = build_model(launcher._trainer_args(args))
    model.load_state_dict(state, strict=True)                      # main.py:412
    model = model.to(torch.device("cuda", 0))
    with torch.no_grad():
        res = evaluate(model, build_test_loader(args), args, df=None, epoch="released", isfinal=True, getcrf=opt.crf == "true",
                       threshold_filters=None)
    meta = {k: ckpt[k] for k in ("s_or_t", "epoch") if isinstance(ckpt, dict) and k in ckpt}
    print(f"checkpoint {opt.checkpoint} {meta or ''}\nFinal Model Result:\n{res[0]}", flush=True)
    with (out_dir / "log_val.txt").open("a") as f:
        f.write("------------" * 3 + f"\nReleased checkpoint {opt.checkpoint}:\n" + "------------" * 3 + "\n" + res[0] + "\n")


if __name__ == "__main__":
    main()
