"""Launcher of the MI355X-native trainer -- the reference's `main.py` (set-up :24-104, loop :106-252, logging :254-311, evaluation cadence
:313-383, finaleval :401-433) on cosa_amd's own loop:

    torchrun --master_port $PORT --nproc_per_node=8 -m cosa_amd.main EXP_VOC --work_dir $DIR --dataset VOC12 \
        --voc12_root $HOME/data/VOCdevkit/VOC2012 --max_iters 32000 --aux_layer -4            (run_voc.sh:7-11)

Same flags and defaults (cosa_amd/args.py), same artefacts in <work_dir>/<name>/ (best_seg.pth, best_cam.pth, log_val.txt,
loss_dataframe.pt).  What differs is where the work runs: one process per GPU over RCCL (`backend="nccl"` is RCCL on ROCm), the device
input pipeline, the fused training step (CoSATrainer.step: no per-iteration host sync -- the losses and the classification AP of an
iteration stay on the device and are read back once per `log_iters`), device-resident evaluation.  `--usepar true` is live."""
import datetime
import os
import random
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

from . import args as cosa_args
from .dataloaders import build_test_loader, build_train_loader, build_val_loader
from .evaluation_engine import evaluate
from .models import build_model
from .train_step import CoSATrainer, default_args
from .utils import torch_helper


def init_distributed_mode(args):
    """utils/misc.py:405-445: env:// rendezvous from torchrun's RANK / WORLD_SIZE / LOCAL_RANK (or SLURM_PROCID); one process per GPU.
    A plain `python -m cosa_amd.main` runs as a world of one (the reference raises NotImplementedError there)."""
    if 'RANK' in os.environ and 'WORLD_SIZE' in os.environ:
        args.rank, args.world_size, args.gpu = int(os.environ["RANK"]), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
    elif 'SLURM_PROCID' in os.environ:
        args.rank = int(os.environ['SLURM_PROCID'])
        args.world_size = int(os.environ.get('SLURM_NTASKS', '1'))
        args.gpu = args.rank % max(torch.cuda.device_count(), 1)
    else:
        args.rank, args.world_size, args.gpu, args.distributed = 0, 1, 0, False
        torch.cuda.set_device(0)
        return
    args.distributed = True
    args.gpu = args.gpu % max(torch.cuda.device_count(), 1)          # (single-card rehearsals put several ranks on one card)
    torch.cuda.set_device(args.gpu)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("COSA_DIST_BACKEND", "nccl")             # "nccl" == RCCL; gloo only for single-card rehearsals
    print(f'| distributed init (rank {args.rank}): env://, backend {backend}', flush=True)
    dist.init_process_group(backend, **({"device_id": torch.device("cuda", args.gpu)} if backend == "nccl" else {}))
    dist.barrier()


def _trainer_args(args):
    """the parsed flags + the build's own switches (compute dtype, fused paths) in the form CoSATrainer reads"""
    return default_args(args.dataset, **{k: v for k, v in vars(args).items() if k != "dataset"})


def check_supported(args):
    """Flags this build parses but does not honour must not be accepted silently (the reference dispatches on them, main.py:216-224,338):
    a run that asks for another objective or for dumps fails here instead of training / evaluating something else."""
    if args.camloss_version != 'v1':
        raise NotImplementedError(f"--camloss_version {args.camloss_version}: only cam_loss v1 is built (the reference dispatches "
                                  f"cam_lossv2 / cam_lossv3_wrap with --segconf_thre, main.py:216-224)")
    if args.turnon_rawcam:
        raise NotImplementedError("--turnon_rawcam: raw-CAM dumps (evaluate(save_rawcam=True)) are not part of the device evaluation path")
    if args.model != 'vit' or args.decoder != 'LargeFOV':
        raise NotImplementedError("only --model vit --decoder LargeFOV (the run scripts' configuration) is built")
    notes = []
    if not args.find_unused:
        notes.append("--find_unused false: no effect (DDP runs without find_unused_parameters; the unused ImageNet head is frozen)")
    if args.segconf_thre != 0.25:
        notes.append("--segconf_thre: only read by cam_loss v3, which is not built")
    for n in notes:
        print("note:", n, flush=True)


def main(args):
    check_supported(args)
    output_dir = Path(args.output_dir) if args.output_dir else Path(args.work_dir) / args.name
    output_dir.mkdir(parents=True, exist_ok=True)
    args.output_dir = output_dir
    init_distributed_mode(args)
    device = torch.device("cuda", args.gpu)
    if args.random_seed:
        args.seed = random.randint(1, 10000)
    is_main = args.rank == 0
    log = (lambda *a, **k: print(*a, **k, flush=True)) if is_main else (lambda *a, **k: None)
    log("{}".format(args).replace(', ', ',\n'))

    targs = _trainer_args(args)
    trainer = CoSATrainer(targs, device, ddp=args.distributed, seed=args.seed)       # seeds, both networks, DDP, PolyWarmupAdamW, EMA, PAR hook
    train_loader = build_train_loader(args, device=device, num_workers=args.num_workers)
    val_loader = build_val_loader(args)
    log(f"train items: {len(train_loader.dataset)}, val items: {len(val_loader.dataset)}")
    n_parameters = sum(p.numel() for p in trainer.student.parameters() if p.requires_grad)
    log('Number of trainable params for Network: {}M'.format(n_parameters // 1000000))

    def new_iter():
        if getattr(train_loader, "sampler", None) is not None and hasattr(train_loader.sampler, "set_epoch"):
            train_loader.sampler.set_epoch(np.random.randint(args.max_iters))       # main.py:74,111
        return iter(train_loader)

    it = new_iter()
    log("Start training")
    start_time, time0, tick = time.time(), datetime.datetime.now().replace(microsecond=0), time.time()
    keys = ('overall_loss', 'cls_loss', 'cls_acc', 'cls_aux_loss', 'cls_aux_acc', 'seg_loss', 'cam_loss', 'reg_loss')
    loss_df = {k: [] for k in keys + ('iters',)}
    acc = torch.zeros(len(keys), device=device, dtype=torch.float64)           # running sums of the interval, on the device
    best_seg = best_cam = -1
    df = None
    for n_iter in range(args.max_iters):
        try:
            img_name, wimg, simg, cls_label, img_box = next(it)
        except StopIteration:
            it = new_iter()
            img_name, wimg, simg, cls_label, img_box = next(it)
        cls_label = cls_label.to(device, non_blocking=True)
        logs = trainer.step(wimg, simg, cls_label, img_box, n_iter)
        with torch.no_grad():                                                     # main.py:257-268, without the per-iteration .item() syncs
            ap, ok = torch_helper.average_precision(cls_label, torch.sigmoid(logs["cls_logits"].float()))
            apa, oka = torch_helper.average_precision(cls_label, torch.sigmoid(logs["cls_aux_logits"].float()))
            acc += torch.stack([t.reshape(()).double() for t in (
                logs['overall_loss'], logs['cls_loss'], (ap * ok).sum() / ok.sum().clamp_min(1), logs['cls_aux_loss'],
                (apa * oka).sum() / oka.sum().clamp_min(1), logs['seg_loss'], logs['cam_loss'], logs['reg_loss'])])
        if (n_iter + 1) % args.log_iters == 0:
            vals = (acc / args.log_iters).tolist()                                # the one host sync of the interval
            acc.zero_()
            now = time.time()
            itertime, tick = (now - tick) / args.log_iters, now
            delta = datetime.datetime.now().replace(microsecond=0) - time0
            eta = delta * (args.max_iters - n_iter - 1) / (n_iter + 1)
            if is_main:
                for k, v in zip(keys, vals):
                    loss_df[k].append(v)
                loss_df['iters'].append(n_iter + 1)
                log("Iter: %d; Elasped: %s; ETA: %s; Itertime: %.3f; LR: %.3e; \n overall_loss: %.4f, cls_loss: %.4f, cls_acc: %.3f,  "
                    "cls_aux_loss: %.4f, cls_aux_acc: %.3f, seg_loss: %.4f, cam_loss: %.4f, reg_loss: %.4f ..."
                    % ((n_iter + 1, delta, str(eta).split('.')[0], itertime, trainer.optimizer.param_groups[0]['lr']) + tuple(vals)))
        if (n_iter + 1) % args.eval_iters == 0:                                   # main.py:313-383
            res_o = evaluate(trainer.student, val_loader, args, df=df, epoch=n_iter + 1, s_or_t='s', get_camiou=True,
                             threshold_filters=args.eval_threshold_filters)
            if is_main:
                tab, segvd, camiou, df, aps = res_o
                log(f'ON Model Classification: cls:{aps[0]}, clsaux: {aps[1]}')
                log(tab)
            res_a = evaluate(trainer.model_AN, val_loader, args, df=df, epoch=n_iter + 1, s_or_t='t', get_camiou=True,
                             threshold_filters=args.eval_threshold_filters)
            if is_main:
                tab_a, segvd_a, camiou_a, df, aps_a = res_a
                log(f'AN: cls:{aps_a[0]}, clsaux: {aps_a[1]}')
                log(tab_a)
                for kind, cands, best in (("seg", [round(segvd, 2), round(segvd_a, 2)], best_seg), ("cam", [round(camiou, 2), round(camiou_a, 2)], best_cam)):
                    cmp_list = cands + [best]
                    idx = max(range(3), key=cmp_list.__getitem__)
                    if idx != 2:
                        torch_helper.save_best(output_dir, trainer.student if idx == 0 else trainer.model_AN, finish_epoch=n_iter + 1,
                                               result=max(cmp_list), args=args, s_or_t='s' if idx == 0 else 't', comment=kind)
                    if kind == "seg":
                        best_seg = max(cmp_list)
                    else:
                        best_cam = max(cmp_list)
                with (output_dir / "log_val.txt").open("a") as f:
                    f.write(f'iters:{n_iter}\n')
                    f.write(f'ON model: cls:{aps[0]}, clsaux: {aps[1]}\n{tab}\n')
                    f.write(f'AN model: cls:{aps_a[0]}, clsaux: {aps_a[1]}\n{tab_a}\n')
    torch.cuda.synchronize()
    if is_main:
        total = str(datetime.timedelta(seconds=int(time.time() - start_time)))
        log('Training time {}'.format(total), 'Best val Seg mIoU: %.2f' % best_seg, 'Best val CAM mIoU: %.2f' % best_cam)
        torch.save(loss_df, output_dir / 'loss_dataframe.pt')                    # (a plain dict of lists: the reference wraps it in a DataFrame)
    if args.distributed:
        dist.barrier()
    if args.finalval and (output_dir / 'best_seg.pth').exists():
        args.bestseg_path = output_dir / 'best_seg.pth'
        log('Perform final validation on best model')
        finaleval(args)
    if args.distributed:
        dist.destroy_process_group()


@torch.no_grad()
def finaleval(args):
    """main.py:401-433: reload best_seg.pth (strict) into a fresh network and evaluate it on the test split with dense-CRF post-processing
    (getcrf=True: rows Seg_vd and Seg_crf; the CRF is seg_helper.DenseCRF on the device lattice kernels)."""
    output_dir = Path(args.output_dir) if args.output_dir else Path(args.work_dir) / args.name
    device = torch.device("cuda", getattr(args, "gpu", 0))
    model = build_model(_trainer_args(args))
    torch_helper.load_best(model, args.bestseg_path, strict=True)
    model = model.to(device)
    res = evaluate(model, build_test_loader(args), args, df=None, epoch='best1', isfinal=True, getcrf=True,   # main.py:414-425
                   threshold_filters=None)
    if getattr(args, "rank", 0) == 0:
        print('Final Model Result:\n' + res[0], flush=True)
        with (output_dir / "log_val.txt").open("a") as f:
            f.write('------------' * 3 + "\nFinal Model Result:\n" + '------------' * 3 + "\n" + res[0] + "\n")
    return res


if __name__ == "__main__":
    parsed, changed = cosa_args.parse()
    print(f'runnning on {parsed.dataset}')
    print("Changed arguments:")
    print(changed)
    main(parsed)
