"""Anchor priors: the reference's k=1 size prior and the k-means clustering with
1 - IoU distance that README.md:56 names as the step before YOLOv2 target assignment.  SPEC.md §6."""
import numpy as np
import torch

from . import _lib as L


def estimate_size_priors(loader):
    """practice/yolo_toy_regressor_3_torch.py:278-286: median w and h over a loader of (x, y[..., 4]) batches,
    clamped to [1e-3, 1 - 1e-3].  Plain host code, as in the reference."""
    ws, hs = [], []
    for _, y in loader:
        ws.append(y[..., 2].reshape(-1))
        hs.append(y[..., 3].reshape(-1))
    w = torch.cat(ws).median().item()
    h = torch.cat(hs).median().item()
    eps = 1e-3
    return (min(max(w, eps), 1 - eps), min(max(h, eps), 1 - eps))


def init_head_biases(model, head_attr="fc", prior_xy=(0.5, 0.5), prior_wh=(0.25, 0.35)):
    """practice/yolo_toy_regressor_3_torch.py:263-276: Xavier-uniform weights and prior logits in the first four
    biases of the head's last Linear layer.  In-place writes: a FlatSGD created earlier notices them through the
    tensor versions (train.bf16_shadow)."""
    import math
    import torch.nn as nn
    head = getattr(model, head_attr)
    last = None
    for m in reversed(list(head.modules())):
        if isinstance(m, nn.Linear):
            last = m
            break
    if last is None:
        raise L.YfsError("init_head_biases: no Linear layer in the head")
    nn.init.xavier_uniform_(last.weight)
    with torch.no_grad():
        last.bias.zero_()
        for i, p in enumerate((prior_xy[0], prior_xy[1], prior_wh[0], prior_wh[1])):
            last.bias[i] = math.log(p / (1 - p))
    return last


@torch.no_grad()
def anchor_assign(wh, anchors):
    """wh [n,2], anchors [k,2] (CUDA) -> (assign int32 [n], best_iou fp32 [n]); argmax shape IoU, ties -> lowest."""
    L.require_cuda(wh, anchors)
    wh = wh.detach().to(torch.float32).contiguous()
    anchors = anchors.detach().to(torch.float32).contiguous()
    n, k = wh.shape[0], anchors.shape[0]
    assign = torch.empty(n, device=wh.device, dtype=torch.int32)
    best = torch.empty(n, device=wh.device, dtype=torch.float32)
    if n:
        L.check(L.lib().yfs_anchor_assign_f32(L.ptr(wh), L.i64(n), L.ptr(anchors), L.c_int(k), L.ptr(assign), L.ptr(best),
                                              L.stream()), "yfs_anchor_assign_f32")
    return assign, best


@torch.no_grad()
def kmeans_anchors(wh, k=5, iters=100):
    """k-means over box sizes with distance 1 - IoU (YOLOv2).  wh [n,2] CUDA tensor, any positive units.
    Deterministic: centroids start at the boxes found at the (i + 0.5)/k quantiles of the area-sorted list; the
    assignment step runs on the GPU, the centroid update (mean of the members, fp64, empty cluster keeps its
    centroid) on the host; stops when no assignment changes.  Returns (anchors [k,2] fp32 sorted by area, mean IoU)."""
    L.require_cuda(wh)
    wh = wh.detach().to(torch.float32).contiguous()
    n = wh.shape[0]
    if n < k:
        raise L.YfsError(f"kmeans_anchors: {n} boxes for k={k}")
    host = wh.cpu().numpy().astype(np.float64)
    order = np.argsort(host[:, 0] * host[:, 1], kind="stable")
    cent = host[order[((np.arange(k) + 0.5) * n / k).astype(np.int64)]].copy()
    prev = None
    for _ in range(iters):
        a, _ = anchor_assign(wh, torch.from_numpy(cent.astype(np.float32)).to(wh.device))
        a = a.cpu().numpy()
        if prev is not None and np.array_equal(a, prev):
            break
        prev = a
        for j in range(k):
            m = a == j
            if m.any():
                cent[j] = host[m].mean(axis=0)
    cent = cent[np.argsort(cent[:, 0] * cent[:, 1], kind="stable")].astype(np.float32)
    _, best = anchor_assign(wh, torch.from_numpy(cent).to(wh.device))
    return torch.from_numpy(cent), float(best.double().mean())
