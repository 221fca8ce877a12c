"""mAP@0.5 (README.md:50) — the evaluation step after decode -> score -> NMS.
VOC protocol, SPEC.md §7: per class, detections of all images sorted by descending score (ties: image, then
box index); a detection is a true positive if its best-IoU ground truth of the same image and class has
IoU >= 0.5 and is not yet claimed, otherwise a false positive; AP = area under the monotone precision envelope
(all-point interpolation); mAP = mean over the classes that have at least one ground truth.
IoUs come from the GPU pairwise kernel (reference op order, `iou_xywh_torch`); the matching is host code."""
import numpy as np
import torch

from .boxes import pairwise_iou


def average_precision(tp, n_gt):
    """tp: bool array in score order; n_gt > 0."""
    tp = np.asarray(tp, dtype=np.float64)
    ctp = np.cumsum(tp)
    cfp = np.cumsum(1.0 - tp)
    rec = ctp / n_gt
    prec = ctp / np.maximum(ctp + cfp, 1e-12)
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    for i in range(mpre.size - 2, -1, -1):
        mpre[i] = max(mpre[i], mpre[i + 1])
    idx = np.nonzero(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[idx + 1] - mrec[idx]) * mpre[idx + 1]))


@torch.no_grad()
def mean_average_precision(boxes, score, cls, keep, count, gt, iou_thr=0.5, num_classes=None):
    """boxes [N,A,4] xywh, score [N,A], cls [N,A], keep [N,A] / count [N] as returned by ``nms``;
    gt [N,G,5] (cx,cy,w,h,class; w<=0 rows are padding).  All CUDA tensors.  -> (mAP, {class: AP})."""
    N = boxes.shape[0]
    gt_h = gt.detach().cpu().numpy()
    cnt = count.cpu().numpy()
    keep_h = keep.cpu().numpy()
    score_h = score.detach().cpu().numpy()
    cls_h = cls.detach().cpu().numpy()
    dets = {}      # class -> list of (-score, image, box index, iou row over the image's gts of that class)
    n_gt = {}
    for n in range(N):
        valid = np.nonzero((gt_h[n, :, 2] > 0) & (gt_h[n, :, 3] > 0))[0]
        gcls = gt_h[n, valid, 4].astype(np.int64)
        for c in gcls:
            n_gt[int(c)] = n_gt.get(int(c), 0) + 1
        k = keep_h[n, :cnt[n]]
        if k.size == 0:
            continue
        iou = None
        if valid.size:
            kb = boxes[n][torch.from_numpy(k.astype(np.int64)).to(boxes.device)]
            iou = pairwise_iou(kb, gt[n, torch.from_numpy(valid).to(gt.device), :4].contiguous()).cpu().numpy()
        for r, b in enumerate(k):
            c = int(cls_h[n, b])
            row = iou[r] if iou is not None else np.zeros(0, np.float32)
            dets.setdefault(c, []).append((-float(score_h[n, b]), n, int(b), np.where(gcls == c, row, -1.0), valid))
    classes = sorted(n_gt) if num_classes is None else [c for c in range(num_classes) if c in n_gt]
    aps = {}
    for c in classes:
        claimed = set()
        tps = []
        for negs, n, b, row, valid in sorted(dets.get(c, []), key=lambda d: (d[0], d[1], d[2])):
            j = int(np.argmax(row)) if row.size else -1
            if j >= 0 and row[j] >= iou_thr and (n, int(valid[j])) not in claimed:
                claimed.add((n, int(valid[j])))
                tps.append(True)
            else:
                tps.append(False)
        aps[c] = average_precision(tps, n_gt[c]) if tps else 0.0
    m = float(np.mean(list(aps.values()))) if aps else 0.0
    return m, aps
