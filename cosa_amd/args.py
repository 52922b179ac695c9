"""Command-line surface of the reference launcher (args.py:82-165 / args_coco.py, `handle_defaults` :168-180) as ONE table.

Every flag of the reference is accepted with the same name, type and default (VOC12 column; the COCO column lists the seven defaults
args_coco.py changes); flags left unset on the command line take the table's default and the ones that were set are reported as
"changed", as the reference does.  `--usepar`, inert in the reference (args.py:67 is never read by main.py), is live here: it selects
`refine_model=PAR(num_iter=10, dilations=[1,2,4,8,12,24])` for both cam2mask calls."""
import argparse


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    if v.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('Boolean value expected.')


_LISTF = ("floats", )          # marker: nargs='+' of float
# (flag, type, VOC12 default)            type None = store_true flag
FLAGS = [
    # model
    ("model", str, 'vit'), ("backbone", str, 'vit_base_patch16_224'), ("decoder", str, 'LargeFOV'), ("pretrained", str2bool, True),
    ("freeze_norm", None, False), ("aux_layer", int, -3), ("isgap", str2bool, False),
    # misc
    ("finalval", str2bool, True), ("seed", int, 0), ("random_seed", None, False), ("work_dir", str, ''), ("output_dir", str, None),
    ("device", str, 'cuda'), ("save_per_eval", int, 10), ("eval_iters", int, 2000), ("turnon_rawcam", None, False), ("fasteval", None, False),
    ("valfull", None, False), ("eval_threshold_filters", _LISTF, None),
    # data
    ("dataset", str, 'VOC12'), ("coco_root", str, ''), ("voc12_root", str, ''), ("crop_size", int, 448), ("scales", tuple, (0.5, 2)),
    ("ignore_index", int, 255), ("num_classes", int, 21), ("batch_size", int, 2), ("num_workers", int, 4),
    # train
    ("max_iters", int, 40000), ("warmup_iters", int, 6000), ("lr", float, 6e-5), ("lrscale", float, 10.), ("min_mult", float, 0.),
    ("wt_dec", float, 1e-2), ("wt_dec_mult", float, 1.), ("cam_weight", float, 0.05), ("camloss_version", str, 'v1'),
    ("seg_weight", float, 0.1), ("segfg_alpha", float, 0.5), ("reg_weight", float, 0.05), ("momentum", float, 0.9994),
    ("pseudo_scales", _LISTF, [1.0, 0.5, 1.5]), ("high_thre", float, 0.7), ("high_thre_aux", float, 0.7), ("bkg_thre", float, 0.5),
    ("low_thre", float, 0.25), ("low_thre_aux", float, 0.25), ("usegmm", str2bool, False), ("usegmmaux", str2bool, False),
    ("gmmscale", int, 16), ("gmmfilter_thre", float, 0.05), ("gmmemadecay", float, 0.99), ("queue_update_ratio", int, 100),
    ("camweight_beta", float, 1.0), ("par_downscale", int, 2), ("usepar", str2bool, False), ("aux_cam2seg", str2bool, True),
    ("aux_cam2seg_traditional", str2bool, True), ("aux_cam2seg_alpha", float, 0.5), ("aux_seg2cam", str2bool, False),
    ("aux_seg2cam_alpha", float, 0.5), ("seg_softmaxtemp", float, 0.01), ("segconf_thre", float, 0.25), ("after_softmax", str2bool, False),
    ("detach", str, 'none'), ("use_cammix", str2bool, False), ("oracle_camloss_version", str, 'v1'),
    ("oracle_camloss_detach", str2bool, False), ("oracle_camloss_bgmax", str2bool, True), ("find_unused", str2bool, True),
]
COCO_DEFAULTS = dict(eval_iters=6000, dataset='COCO', num_classes=81, batch_size=4, max_iters=60000, warmup_iters=10000, high_thre=0.65)
# what this build adds (none of them changes the reference's defaults)
EXTRA = [
    ("pretrained_path", str, None),          # local timm ViT-B/16 checkpoint for --pretrained true (no network here)
    ("name_list_dir", str, None),            # split lists / cls_labels_onehot.npy (default: ./dataloaders/<dataset>/ as the reference)
    ("teacher_precision", str, "auto"), # auto (fp16x3: train_step.resolve_teacher_precision) | bf16 | fp16 | bf16x3 | fp16x3 | fp16c8[-n[mk]|-xn[mk]] | fp16c4[...]: operand precision of the teacher's no-grad passes;
                                             # the default meets the 1e-3 / IoU 0.999 tolerance against the fp32 reference, bf16 (faster) does not
    ("log_iters", int, 20),
]


def get_parser():
    p = argparse.ArgumentParser('End to end weakly supervised segmentation model (cosa_amd launcher)', add_help=True)
    p.add_argument('name', type=str)
    for flag, typ, _default in FLAGS + EXTRA:
        if typ is None:
            p.add_argument('--' + flag, action='store_true', default=None)
        elif typ is _LISTF:
            p.add_argument('--' + flag, type=float, metavar='N', nargs='+', default=None)
        elif typ is tuple:
            p.add_argument('--' + flag, type=float, metavar='N', nargs=2, default=None)
        else:
            p.add_argument('--' + flag, type=typ, default=None)
    return p


def handle_defaults(args):
    """args.py:168-180: unset flags take their default (COCO's when --dataset COCO); returns (args, {flag: value given on the command line})"""
    table = {f: d for f, _t, d in FLAGS + EXTRA}
    dataset = args.dataset if args.dataset is not None else table["dataset"]
    if dataset == 'COCO':
        table.update(COCO_DEFAULTS)
    elif dataset != 'VOC12':
        raise NotImplementedError(dataset)
    changed = {}
    for k, v in table.items():
        cur = getattr(args, k)
        if cur is None:
            setattr(args, k, v)
        else:
            if k == "scales":
                cur = tuple(cur)
                setattr(args, k, cur)
            changed[k] = cur
    return args, changed


def parse(argv=None):
    return handle_defaults(get_parser().parse_args(argv))
