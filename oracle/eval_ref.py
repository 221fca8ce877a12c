"""CPU oracle (test infrastructure only) for the two SURVEY.md §8(f) rank-3 pieces the reference names but does
not implement (README.md:50, :56): k-means anchors with 1 - IoU distance and mAP@0.5.  Written from SPEC.md §6-§7
as plain numpy / Python loops; parity with the reference is unpinned (no reference code or fixture exists); the
k=1 precursor `estimate_size_priors` (practice/yolo_toy_regressor_3_torch.py:278-286) is a median and is mirrored
as such in the package."""
import numpy as np


def shape_iou(w, h, aw, ah):
    w, h, aw, ah = np.float32(w), np.float32(h), np.float32(aw), np.float32(ah)
    inter = np.float32(min(w, aw)) * np.float32(min(h, ah))
    uni = np.float32(np.float32(w * h) + np.float32(aw * ah)) - inter
    return np.float32(inter / uni) if uni > 0 else np.float32(0.0)


def anchor_assign(wh, anchors):
    wh = np.asarray(wh, np.float32)
    anchors = np.asarray(anchors, np.float32)
    a = np.zeros(len(wh), np.int32)
    best = np.zeros(len(wh), np.float32)
    for i, (w, h) in enumerate(wh):
        bi, bj = np.float32(-1.0), 0
        for j, (aw, ah) in enumerate(anchors):
            v = shape_iou(w, h, aw, ah)
            if v > bi:
                bi, bj = v, j
        a[i], best[i] = bj, bi
    return a, best


def kmeans_anchors(wh, k=5, iters=100):
    wh32 = np.asarray(wh, np.float32)
    host = wh32.astype(np.float64)
    n = len(host)
    order = np.argsort(host[:, 0] * host[:, 1], kind="stable")
    cent = host[order[((np.arange(k) + 0.5) * n / k).astype(np.int64)]].copy()
    prev = None
    for _ in range(iters):
        a, _ = anchor_assign(wh32, cent.astype(np.float32))
        if prev is not None and np.array_equal(a, prev):
            break
        prev = a
        for j in range(k):
            if (a == j).any():
                cent[j] = host[a == j].mean(axis=0)
    cent = cent[np.argsort(cent[:, 0] * cent[:, 1], kind="stable")].astype(np.float32)
    _, best = anchor_assign(wh32, cent)
    return cent, float(best.astype(np.float64).mean())


def iou_xywh(a, b):
    """Reference op order (yolo_toy_regressor_torch.py:159-182) in fp32."""
    f = np.float32
    ax1, ay1, ax2, ay2 = f(a[0] - f(0.5) * a[2]), f(a[1] - f(0.5) * a[3]), f(a[0] + f(0.5) * a[2]), f(a[1] + f(0.5) * a[3])
    bx1, by1, bx2, by2 = f(b[0] - f(0.5) * b[2]), f(b[1] - f(0.5) * b[3]), f(b[0] + f(0.5) * b[2]), f(b[1] + f(0.5) * b[3])
    iw = max(f(min(ax2, bx2) - max(ax1, bx1)), f(0))
    ih = max(f(min(ay2, by2) - max(ay1, by1)), f(0))
    inter = f(iw * ih)
    uni = f(f(f((ax2 - ax1) * (ay2 - ay1)) + f((bx2 - bx1) * (by2 - by1))) - inter)
    return f(inter / uni) if uni > 0 else f(0)


def mean_average_precision(boxes, score, cls, keep, count, gt, iou_thr=0.5):
    boxes, score, gt = np.asarray(boxes, np.float32), np.asarray(score, np.float32), np.asarray(gt, np.float32)
    N = boxes.shape[0]
    n_gt, dets = {}, []
    for n in range(N):
        for g in gt[n]:
            if g[2] > 0 and g[3] > 0:
                n_gt[int(g[4])] = n_gt.get(int(g[4]), 0) + 1
        for b in keep[n][:count[n]]:
            dets.append((int(cls[n][b]), -float(score[n][b]), n, int(b)))
    aps = {}
    for c in sorted(n_gt):
        claimed, tp = set(), []
        for _, _, n, b in sorted(d for d in dets if d[0] == c):
            best, bj = -1.0, -1
            for j, g in enumerate(gt[n]):
                if g[2] > 0 and g[3] > 0 and int(g[4]) == c:
                    v = float(iou_xywh(boxes[n][b], g[:4]))
                    if v > best:
                        best, bj = v, j
            ok = bj >= 0 and best >= iou_thr and (n, bj) not in claimed
            if ok:
                claimed.add((n, bj))
            tp.append(ok)
        if not tp:
            aps[c] = 0.0
            continue
        ctp = np.cumsum(np.asarray(tp, np.float64))
        cfp = np.cumsum(1.0 - np.asarray(tp, np.float64))
        rec, prec = ctp / n_gt[c], ctp / (ctp + cfp)
        ap, prev_r = 0.0, 0.0
        for i in range(len(tp)):                       # area under the monotone envelope, recall step by recall step
            if rec[i] != prev_r:
                ap += (rec[i] - prev_r) * prec[i:].max()
                prev_r = rec[i]
        aps[c] = float(ap)
    return (float(np.mean(list(aps.values()))) if aps else 0.0), aps
