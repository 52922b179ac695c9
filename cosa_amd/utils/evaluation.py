"""utils.evaluation -- segmentation scores from a confusion matrix (reference: utils/evaluation.py).

`scores` / `pseudo_scores` keep the reference's call surface (lists of label maps in, dict of pAcc / mAcc / miou / iou out) but count
on the GPU: the maps go through `cosa_confusion_hist` (LDS-privatised counters) into one [nc, nc] int64 matrix.  `ConfusionMeter`
is the streaming form the evaluation engine uses: device-resident accumulation over the whole validation set, one all-reduce across
ranks at the end (the reference gathers every prediction map on rank 0 through temp files, evaluation_engine.py:203-216).
"""
import numpy as np
import torch

from .. import _C


def _as_u8_cuda(x, device):
    if not torch.is_tensor(x):
        x = torch.from_numpy(np.ascontiguousarray(x))
    if x.dtype != torch.uint8:
        x = x.to(torch.uint8)                       # the reference stores every map as uint8 (evaluation_engine.py:198-200)
    return x.to(device, non_blocking=True).contiguous()


class ConfusionMeter:
    """hist[t, p] over all pixels with truth t < num_classes (255 = ignore).  pseudo=True applies pseudo_scores' relabelling
    (utils/evaluation.py:43-46): pixels whose PREDICTION is 255 are dropped."""

    def __init__(self, num_classes, device=None, pseudo=False):
        device = torch.device(device if device is not None else "cuda")
        if device.type != "cuda":
            raise RuntimeError("ConfusionMeter counts on the GPU (cosa_confusion_hist); no CPU path")
        self.num_classes, self.pseudo = int(num_classes), bool(pseudo)
        self.hist = torch.zeros((self.num_classes, self.num_classes), device=device, dtype=torch.int64)

    def update(self, label_true, label_pred):
        gt, pr = _as_u8_cuda(label_true, self.hist.device), _as_u8_cuda(label_pred, self.hist.device)
        if gt.numel() != pr.numel():
            raise ValueError("ConfusionMeter.update: truth and prediction differ in size")
        _C.check(_C.lib().cosa_confusion_hist(_C.ptr(gt), _C.ptr(pr), gt.numel(), self.num_classes, int(self.pseudo), _C.ptr(self.hist),
                                              _C.stream_ptr()), "cosa_confusion_hist")

    def all_reduce(self):
        """sum over ranks (replaces the temp-file gather of evaluation_engine.py:203-216)"""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.hist)
        return self

    def scores(self):
        return scores_from_hist(self.hist.cpu().numpy())


def scores_from_hist(hist):
    """utils/evaluation.py:21-35 on an accumulated confusion matrix (row = truth)."""
    hist = np.asarray(hist, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        acc = np.diag(hist).sum() / hist.sum()
        acc_cls = np.nanmean(np.diag(hist) / hist.sum(axis=1))
        iu = np.diag(hist) / (hist.sum(axis=1) + hist.sum(axis=0) - np.diag(hist))
    valid = hist.sum(axis=1) > 0
    return {"pAcc": acc, "mAcc": acc_cls, "miou": np.nanmean(iu[valid]), "iou": dict(zip(range(hist.shape[0]), iu))}


def _scores(label_trues, label_preds, num_classes, pseudo):
    meter = ConfusionMeter(num_classes, pseudo=pseudo)
    for lt, lp in zip(label_trues, label_preds):
        meter.update(lt, lp)
    return meter.scores()


def scores(label_trues, label_preds, num_classes):
    """utils/evaluation.py:17-35"""
    return _scores(label_trues, label_preds, num_classes, False)


def pseudo_scores(label_trues, label_preds, num_classes):
    """utils/evaluation.py:37-70 (the inputs are NOT modified in place, unlike the reference's `lt[lp==255] = 255`)"""
    return _scores(label_trues, label_preds, num_classes, True)
