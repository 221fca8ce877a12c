"""YOLOv2 anchor target assignment + ignore-IoU mask — host side of kernel set 4
(csrc/assign_kernels.hip).  No reference code (README.md:55-59); SPEC.md §4."""
import torch

from . import _lib as L

# VOC YOLOv2 priors in grid units ("VOC defaults", README.md:56)
VOC_ANCHORS = ((1.3221, 1.73145), (3.19275, 4.00944), (5.05587, 8.09892), (9.47112, 4.84053), (11.2364, 10.0071))


@torch.no_grad()
def assign_targets_v2(gt, anchors, S, pred_boxes=None, ignore_thr=0.6):
    """gt [N,G,5] (cx,cy,w,h,class; w<=0 rows are padding), anchors [A,2] grid units,
    pred_boxes [N,S,S,A,4] or [N,S*S*A,4] xywh (optional) ->
    obj_mask uint8 [N,S,S,A], ignore_mask uint8 [N,S,S,A], targets fp32 [N,S,S,A,5]."""
    L.require_cuda(gt, anchors, pred_boxes)
    gt = gt.detach().to(torch.float32).contiguous()
    anchors = anchors.detach().to(torch.float32).contiguous()
    if gt.dim() != 3 or gt.shape[2] != 5:
        raise L.YfsError(f"assign_targets_v2 expects gt [N,G,5], got {tuple(gt.shape)}")
    N, G, _ = gt.shape
    A = anchors.shape[0]
    dev = gt.device
    if pred_boxes is not None:
        pred_boxes = pred_boxes.detach().to(torch.float32).contiguous()
        if pred_boxes.numel() != N * S * S * A * 4:
            raise L.YfsError(f"pred_boxes has {pred_boxes.numel()} elements, expected {N * S * S * A * 4}")
    obj = torch.empty(N, S, S, A, device=dev, dtype=torch.uint8)
    ign = torch.empty(N, S, S, A, device=dev, dtype=torch.uint8)
    tgt = torch.empty(N, S, S, A, 5, device=dev, dtype=torch.float32)
    if N == 0:
        return obj, ign, tgt
    L.check(L.lib().yfs_assign_targets_v2_f32(L.ptr(gt), L.i64(N), L.c_int(G), L.c_int(S), L.c_int(A), L.ptr(anchors),
                                              L.ptr(pred_boxes), L.c_float(ignore_thr), L.ptr(obj), L.ptr(ign),
                                              L.ptr(tgt), L.stream()), "yfs_assign_targets_v2_f32")
    return obj, ign, tgt
