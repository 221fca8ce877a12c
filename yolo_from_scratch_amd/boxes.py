"""Box math, decode and NMS on the MI355X — host side of kernel set 3 (csrc/box_kernels.hip).

Mirrors the reference's call surface: ``iou_xywh_torch`` (yolo_toy_regressor_torch.py:159-182),
``decode_head`` (practice/yolo_toy_regressor_3_torch.py:290-295), ``decode_xywh``
(practice/yolo_toy_regressor_torch_20251015.py:411-421), ``box_iou_xyxy`` (..._3_torch.py:112-119).
``decode_grid``/``class_scores``/``nms``/``decode_nms`` have no reference code (README.md:44-47).
"""
import torch

from . import _lib as L


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


@torch.no_grad()
def iou_xywh_rows(pred, gt):
    """Per-pair IoU [n] in the op order of iou_xywh_torch."""
    L.require_cuda(pred, gt)
    pred, gt = _f32c(pred), _f32c(gt)
    n = pred.shape[0]
    out = torch.empty(n, device=pred.device, dtype=torch.float32)
    if n == 0:
        return out
    L.check(L.lib().yfs_iou_xywh_f32(L.ptr(pred), L.ptr(gt), L.i64(n), L.ptr(out), L.ptr(None), L.ptr(None),
                                     L.c_size(0), L.stream()), "yfs_iou_xywh_f32")
    return out


@torch.no_grad()
def iou_xywh_torch(pred, gt):
    """Drop-in for iou_xywh_torch(pred[B,4], gt[B,4]) -> python float (batch mean)."""
    L.require_cuda(pred, gt)
    pred, gt = _f32c(pred), _f32c(gt)
    n = pred.shape[0]
    ws_bytes = L.lib().yfs_iou_xywh_workspace_bytes(L.i64(n))
    ws = torch.empty(max(ws_bytes // 4, 1) + 1, device=pred.device, dtype=torch.float32)
    mean = ws[-1:]
    L.check(L.lib().yfs_iou_xywh_f32(L.ptr(pred), L.ptr(gt), L.i64(n), L.ptr(None), L.ptr(mean), L.ptr(ws),
                                     L.c_size(ws_bytes), L.stream()), "yfs_iou_xywh_f32")
    return mean.item()


@torch.no_grad()
def iou_variant(pred, gt, variant):
    """variant 1: box_iou_xyxy(xywh_to_xyxy(.)); variant 2: the older clamped iou_xywh_torch."""
    L.require_cuda(pred, gt)
    pred, gt = _f32c(pred), _f32c(gt)
    out = torch.empty(pred.shape[0], device=pred.device, dtype=torch.float32)
    L.check(L.lib().yfs_iou_variant_f32(L.ptr(pred), L.ptr(gt), L.i64(pred.shape[0]), L.c_int(variant), L.ptr(out),
                                        L.stream()), "yfs_iou_variant_f32")
    return out


@torch.no_grad()
def pairwise_iou(a, b):
    """[Na,4] x [Nb,4] xywh -> [Na,Nb] IoU."""
    L.require_cuda(a, b)
    a, b = _f32c(a), _f32c(b)
    out = torch.empty(a.shape[0], b.shape[0], device=a.device, dtype=torch.float32)
    if out.numel() == 0:
        return out
    L.check(L.lib().yfs_pairwise_iou_f32(L.ptr(a), L.i64(a.shape[0]), L.ptr(b), L.i64(b.shape[0]), L.ptr(out),
                                         L.stream()), "yfs_pairwise_iou_f32")
    return out


PRIOR_W, PRIOR_H = 0.25, 0.35   # practice/yolo_toy_regressor_3_torch.py:288


@torch.no_grad()
def decode_head(t, prior_w=None, prior_h=None):
    L.require_cuda(t)
    shape = t.shape
    t = _f32c(t).view(-1, 4)
    out = torch.empty_like(t)
    L.check(L.lib().yfs_decode_head_f32(L.ptr(t), L.i64(t.shape[0]), L.c_float(PRIOR_W if prior_w is None else prior_w),
                                        L.c_float(PRIOR_H if prior_h is None else prior_h), L.ptr(out), L.stream()),
            "yfs_decode_head_f32")
    return out.view(shape)


@torch.no_grad()
def decode_xywh(raw):
    L.require_cuda(raw)
    shape = raw.shape
    t = _f32c(raw).view(-1, 4)
    out = torch.empty_like(t)
    L.check(L.lib().yfs_decode_xywh_f32(L.ptr(t), L.i64(t.shape[0]), L.ptr(out), L.stream()), "yfs_decode_xywh_f32")
    return out.view(shape)


def _grid_shapes(raw, anchors):
    if raw.dim() != 5 or raw.shape[1] != raw.shape[2]:
        raise L.YfsError(f"decode_grid expects raw [N,S,S,A,5+C], got {tuple(raw.shape)}")
    N, S, _, A, D = raw.shape
    if tuple(anchors.shape) != (A, 2):
        raise L.YfsError(f"anchors must be [{A},2], got {tuple(anchors.shape)}")
    return N, S, A, D - 5


@torch.no_grad()
def decode_grid(raw, anchors, all_scores=False):
    """raw [N,S,S,A,5+C] -> boxes [N,S*S*A,4] xywh, score [N,S*S*A], cls int32 [N,S*S*A] (+ all class scores)."""
    L.require_cuda(raw, anchors)
    N, S, A, C = _grid_shapes(raw, anchors)
    raw, anchors = _f32c(raw), _f32c(anchors)
    nb = S * S * A
    dev = raw.device
    boxes = torch.empty(N, nb, 4, device=dev, dtype=torch.float32)
    score = torch.empty(N, nb, device=dev, dtype=torch.float32)
    cls = torch.empty(N, nb, device=dev, dtype=torch.int32)
    alls = torch.empty(N, nb, C, device=dev, dtype=torch.float32) if all_scores else None
    L.check(L.lib().yfs_decode_grid_f32(L.ptr(raw), L.i64(N), L.c_int(S), L.c_int(A), L.c_int(C), L.ptr(anchors),
                                        L.ptr(boxes), L.ptr(score), L.ptr(cls), L.ptr(alls), L.stream()),
            "yfs_decode_grid_f32")
    return (boxes, score, cls, alls) if all_scores else (boxes, score, cls)


@torch.no_grad()
def class_scores(raw, anchors):
    """confidence = objectness x class probability for every class: [N,S*S*A,C] (README.md:46)."""
    return decode_grid(raw, anchors, all_scores=True)[3]


@torch.no_grad()
def decode_v1(pred, B=2, C=20):
    """YOLOv1 head [N,S,S,5B+C] -> boxes [N,S*S*B,4], score, cls."""
    L.require_cuda(pred)
    pred = _f32c(pred)
    N, S, S2, D = pred.shape
    if S != S2 or D != 5 * B + C:
        raise L.YfsError(f"decode_v1 expects [N,S,S,{5 * B + C}], got {tuple(pred.shape)}")
    nb = S * S * B
    boxes = torch.empty(N, nb, 4, device=pred.device, dtype=torch.float32)
    score = torch.empty(N, nb, device=pred.device, dtype=torch.float32)
    cls = torch.empty(N, nb, device=pred.device, dtype=torch.int32)
    L.check(L.lib().yfs_decode_v1_f32(L.ptr(pred), L.i64(N), L.c_int(S), L.c_int(B), L.c_int(C), L.ptr(boxes),
                                      L.ptr(score), L.ptr(cls), L.stream()), "yfs_decode_v1_f32")
    return boxes, score, cls


@torch.no_grad()
def nms(boxes, score, cls, score_thr=0.05, iou_thr=0.5):
    """Class-wise greedy NMS per image.  -> keep [N,A] int32 (score order, -1 padded), count [N] int32."""
    L.require_cuda(boxes, score, cls)
    boxes, score = _f32c(boxes), _f32c(score)
    cls = cls.detach().to(torch.int32).contiguous()
    N, A = score.shape
    keep = torch.empty(N, A, device=boxes.device, dtype=torch.int32)
    count = torch.empty(N, device=boxes.device, dtype=torch.int32)
    if N == 0 or A == 0:
        return keep, count
    L.check(L.lib().yfs_nms_classwise_f32(L.ptr(boxes), L.ptr(score), L.ptr(cls), L.i64(N), L.c_int(A),
                                          L.c_float(score_thr), L.c_float(iou_thr), L.ptr(keep), L.ptr(count),
                                          L.stream()), "yfs_nms_classwise_f32")
    return keep, count


class DecodeNms:
    """decode_grid + NMS in one launch; output buffers are allocated once and reused (serving loop)."""

    def __init__(self, N, S, A, C, device, score_thr=0.05, iou_thr=0.5):
        nb = S * S * A
        self.shape = (N, S, A, C)
        self.score_thr, self.iou_thr = score_thr, iou_thr
        self.boxes = torch.empty(N, nb, 4, device=device, dtype=torch.float32)
        self.score = torch.empty(N, nb, device=device, dtype=torch.float32)
        self.cls = torch.empty(N, nb, device=device, dtype=torch.int32)
        self.keep = torch.empty(N, nb, device=device, dtype=torch.int32)
        self.count = torch.empty(N, device=device, dtype=torch.int32)

    @torch.no_grad()
    def __call__(self, raw, anchors):
        N, S, A, C = self.shape
        if _grid_shapes(raw, anchors) != (N, S, A, C) or raw.dtype != torch.float32 or not raw.is_contiguous():
            raise L.YfsError("DecodeNms: raw must be contiguous fp32 of the shape given at construction")
        L.check(L.lib().yfs_decode_nms_f32(L.ptr(raw), L.i64(N), L.c_int(S), L.c_int(A), L.c_int(C), L.ptr(anchors),
                                           L.c_float(self.score_thr), L.c_float(self.iou_thr), L.ptr(self.boxes),
                                           L.ptr(self.score), L.ptr(self.cls), L.ptr(self.keep), L.ptr(self.count),
                                           L.stream()), "yfs_decode_nms_f32")
        return self.boxes, self.score, self.cls, self.keep, self.count


@torch.no_grad()
def decode_nms(raw, anchors, score_thr=0.05, iou_thr=0.5):
    L.require_cuda(raw, anchors)
    N, S, A, C = _grid_shapes(raw, anchors)
    op = DecodeNms(N, S, A, C, raw.device, score_thr, iou_thr)
    return op(_f32c(raw), _f32c(anchors))
