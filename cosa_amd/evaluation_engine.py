"""evaluation_engine -- the validation pass of the reference (evaluation_engine.py:11-297), device-resident.

Same call surface and return values as the reference's `evaluate`.  What changes is where the work happens:

* per image: five scales x two flips through the network (`seg_helper.multi_scale_camsegv3`), then ONE kernel turns the (S,S) CAMs and
  logits into the three uint8 label maps at the ground truth's resolution (`seg_helper.eval_label_maps` = F.interpolate x3 +
  cam_to_label x2 + seg_validation + argmax x2 of the reference) -- nothing of size [1,C,H,W] is materialised;
* the maps never leave the GPU: four confusion matrices are accumulated on the device (`evaluation.ConfusionMeter`) and summed over
  ranks with one all-reduce, instead of storing every map in host lists and shipping them to rank 0 through temp files (:203-216);
* per-image average precision is computed on the device and accumulated without a host sync.

`threshold_filters` (evaluation_engine.py:43-50,132-152,252-262): for every threshold t the pseudo-label maps cam2mask(valid CAM,
high = 1 - t, low = t) of the main and the auxiliary CAMs at the ground truth's resolution, scored with `pseudo_scores` (pixels labelled
255 are dropped) into rows `cam_<t>` / `camaux_<t>` of the table.

`getcrf` (evaluation_engine.py:204-211,247-250; `finaleval` passes it): per image the validated logits at the ground truth's resolution ->
softmax -> one mean-field step of the dense CRF (`seg_helper.crf_inference_infv2`: two permutohedral-lattice filters on the device) ->
argmax, scored as the row `Seg_crf`.  Parity of that row with pydensecrf is unpinned (see seg_helper.DenseCRF).

Not built: image / CAM dumps (`save_result`, `save_rawcam`); asking for them raises NotImplementedError.
"""
import torch
import torch.distributed as dist
import torch.nn.functional as F

from .utils import evaluation, seg_helper, torch_helper

EVAL_SCALES = [1.0, 0.5, 1.5, 0.75, 1.25]          # evaluation_engine.py:84


class _GraphedCamSeg:
    """multi_scale_camsegv3 at a fixed input shape (evaluation resizes every image to crop_size x crop_size, batch 1) as one hipGraph:
    ten encoder passes at batch 1 are ~1000 small launches, i.e. launch-bound when issued one by one.  Two eager calls first (library
    kernel selection), then capture; any other input shape falls back to eager."""

    def __init__(self, model, scales, enabled=True):
        self.model, self.scales, self.enabled = model, scales, enabled
        self.calls, self.graph, self.static_in, self.outs = 0, None, None, None

    def __call__(self, inputs):
        eager = lambda x: seg_helper.multi_scale_camsegv3(self.model, x, self.scales, getcls=True, _per_image_cls=True)
        if not self.enabled or (self.static_in is not None and inputs.shape != self.static_in.shape):
            return eager(inputs)
        if self.graph is None:
            self.calls += 1
            if self.calls <= 2:
                return eager(inputs)
            self.static_in = inputs.clone()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self.outs = eager(self.static_in)
            self.graph = g
        self.static_in.copy_(inputs)
        self.graph.replay()
        return self.outs


def evaluate(model, data_loader, args, df=None, save_result=False, save_rawcam=False, epoch=None, threshold_filters=None, getcrf=False,
             s_or_t='t', get_camiou=False, isfinal=False, class_list=None, use_graph=True, eval_group=4):
    if save_result or save_rawcam:
        raise NotImplementedError("evaluate: save_result / save_rawcam (image dumps) are not part of the device path")
    threshold_filters = list(threshold_filters) if threshold_filters else []
    assert s_or_t in ['s', 't']
    distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if distributed else 0
    device = next(model.parameters()).device
    if device.type != "cuda":
        raise RuntimeError("evaluate runs on the GPU (HIP kernels); no CPU path")
    nc = args.num_classes
    meters = {k: evaluation.ConfusionMeter(nc, device) for k in ("cam", "cam_aux", "seg_ps", "seg_vd")}
    if getcrf:
        meters["seg_crf"] = evaluation.ConfusionMeter(nc, device)
    for thre in threshold_filters:                    # pseudo_scores' relabelling: a prediction of 255 drops the pixel (utils/evaluation.py:43-46)
        meters[f"cam_{thre}"] = evaluation.ConfusionMeter(nc, device, pseudo=True)
        meters[f"camaux_{thre}"] = evaluation.ConfusionMeter(nc, device, pseudo=True)
    ap_sum = torch.zeros(2, device=device, dtype=torch.float64)          # sums of per-batch mean AP (cls, cls_aux) ...
    ap_cnt = 0                                                           # ... over the batches (AverageMeter semantics, :86-92)
    was_training = model.training
    model.eval()
    net = model.module if hasattr(model, "module") else model
    heads_before = getattr(net, "batch_invariant_heads", None)
    if heads_before is not None:                       # grouped items must reproduce the one-at-a-time bits: narrow heads on own kernel
        net.batch_invariant_heads = True
        net.decoder.batch_invariant = True
    camseg = _GraphedCamSeg(model, EVAL_SCALES, enabled=use_graph and getattr(model, "can_forward_multi", None) is not None)
    # Every image is resized to crop_size x crop_size before the network (:82), so `eval_group` loader items share one multi-scale pass
    # (GEMMs with several times the rows instead of ten batch-1 encoder passes per image: 137 -> 242 img/s at 4); label maps, histograms
    # and AP stay per image.  Every kernel on the CAM / seg path (patch projection, encoder, convs, narrow heads, CAM tail) gives a token
    # the same bits whatever else is in the batch, so the score table and the AP are those of the one-image-at-a-time loop exactly
    # (tested).  eval_group=1 is the reference's loop literally.
    def flush(group):
        if not group:
            return
        nonlocal ap_cnt
        inputs = torch.cat([g[0] for g in group], dim=0)
        cams, cams_aux, seg_ps, cls_final, cls_aux = camseg(inputs)
        for i, (_, labels, cls_label, img_org) in enumerate(group):
            # classification AP of this item (:86-92; cls_* are sums over scales and flips, compared with every label row)
            for j, logit in enumerate((cls_final[i:i + 1], cls_aux[i:i + 1])):
                ap, valid = torch_helper.average_precision(cls_label, torch.sigmoid(logit.float()).expand_as(cls_label))
                ap_sum[j] += (ap * valid).sum() / valid.sum().clamp_min(1)
            ap_cnt += 1
            size = labels.shape[1:]
            cam_label, pred_ps, pred_vd = seg_helper.eval_label_maps(cams[i:i + 1], seg_ps[i:i + 1], cls_label, size, args.bkg_thre)
            cam_aux_label, _, _ = seg_helper.eval_label_maps(cams_aux[i:i + 1], None, cls_label, size, args.bkg_thre)
            gt = labels.to(torch.uint8)
            meters["cam"].update(gt, cam_label)
            meters["cam_aux"].update(gt, cam_aux_label)
            meters["seg_ps"].update(gt, pred_ps)
            meters["seg_vd"].update(gt, pred_vd)
            if getcrf:
                # evaluation_engine.py:204-211: softmax of the validated logits at the ground truth's size, the de-normalised uint8 image,
                # one mean-field step, argmax
                rs = F.interpolate(seg_ps[i:i + 1], size=size, mode='bilinear', align_corners=False)
                vd = seg_helper.seg_validation(rs, cls_label).softmax(dim=1)[0]
                ori = torch_helper.denormalize_img_(img_org)[0].permute(1, 2, 0)
                q = seg_helper.crf_inference_infv2(ori, vd.contiguous())
                meters["seg_crf"].update(gt, q.argmax(dim=0, keepdim=True).to(torch.uint8))
            if threshold_filters:
                # evaluation_engine.py:132-152: the CAMs resized to the ground truth's size, validated, then cam2mask with the box
                # [0, -1, 0, -1] (Python slice semantics: the last row and column stay `ignore`, as in the reference) and no refine model
                shape_only = torch.zeros(1, device=device).expand(1, 3, *size)
                for thre in threshold_filters:
                    for key, c in ((f"cam_{thre}", cams), (f"camaux_{thre}", cams_aux)):
                        rc = F.interpolate(c[i:i + 1], size=size, mode='bilinear', align_corners=False)
                        m = seg_helper.cam2mask(images=shape_only, img_boxes=[[0, -1, 0, -1]], cams=seg_helper.cam_validation(rc, cls_label),
                                                cls_labels=cls_label, threshold_high=1 - thre, threshold_low=thre)
                        meters[key].update(gt, m.to(torch.uint8))

    with torch.no_grad():
        group = []
        for data in data_loader:
            name, img_org, labels, cls_label = data
            labels = labels.to(device, non_blocking=True)
            cls_label = cls_label.to(device, non_blocking=True).float()
            img_org = img_org.to(device, non_blocking=True)
            inputs = F.interpolate(img_org, size=[args.crop_size, args.crop_size], mode='bilinear', align_corners=False)
            if inputs.shape[0] != 1:                      # a loader with its own batching: one pass per item, as before
                flush(group)
                group = []
                for i in range(inputs.shape[0]):
                    flush([(inputs[i:i + 1], labels[i:i + 1], cls_label[i:i + 1], img_org[i:i + 1])])
                continue
            group.append((inputs, labels, cls_label, img_org if getcrf else None))
            if len(group) >= max(1, int(eval_group)):
                flush(group)
                group = []
        flush(group)
    for m in meters.values():
        m.all_reduce()
    if heads_before is not None:
        net.batch_invariant_heads = heads_before
        net.decoder.batch_invariant = heads_before
    if was_training:
        model.train()
    if rank != 0:
        return (None,) * (5 if get_camiou else 4)

    cam_score, cam_aux_score, seg_vd_score = meters["cam"].scores(), meters["cam_aux"].scores(), meters["seg_vd"].scores()
    metrics, names = [cam_score, cam_aux_score, seg_vd_score], ["CAM", "aux_CAM", "Seg_vd"]
    if isfinal:
        metrics, names = [seg_vd_score], ["Seg_vd"]
    if getcrf:                                           # evaluation_engine.py:247-250
        metrics, names = metrics + [meters["seg_crf"].scores()], names + ["Seg_crf"]
    if threshold_filters:                                # evaluation_engine.py:252-262: inserted after the first three rows
        tk = [f"cam_{t}" for t in threshold_filters] + [f"camaux_{t}" for t in threshold_filters]
        metrics = metrics[:3] + [meters[k].scores() for k in tk] + metrics[3:]
        names = names[:3] + tk + names[3:]
    cls_aps = [float(v) / max(ap_cnt, 1) for v in ap_sum.tolist()]
    if class_list is None:
        class_list = [str(i) for i in range(nc)]
    tab_results, _, mioulist = torch_helper.format_tabs(scores=metrics, name_list=names, cat_list=class_list)
    if not df:
        df = {'Iterations': [], 'mIoU': [], 'Metrics': [], 'ST': []}
    df['Iterations'].extend([epoch] * len(names))
    df['mIoU'].extend(mioulist)
    df['Metrics'].extend(names)
    df['ST'].extend([s_or_t] * len(names))
    # (as the reference, evaluation_engine.py:289: the last row, or the one before it with getcrf -- with threshold_filters that is a
    # `cam_<t>` / `camaux_<t>` row, not Seg_vd)
    # evaluation_engine.py:287-290 of the reference: the "seg" score handed back is the LAST row of the table (second to last with the CRF row).
    # With `threshold_filters` that row is a pseudo-label sweep row (camaux_<t>), not Seg_vd -- the reference's behaviour, kept so that
    # best-checkpoint selection in main.py follows the reference run for run (README: "--eval_threshold_filters").
    seg_vd_miou, cam_miou = (mioulist[-1] if not getcrf else mioulist[-2]), mioulist[0]
    if get_camiou:
        return tab_results, seg_vd_miou, cam_miou, df, cls_aps
    return tab_results, seg_vd_miou, df, cls_aps
