"""The reference's data path after decoding: segmentation mask -> boxes -> normalised
[xc, yc, w, h] -> YOLOv1 grid targets, on the GPU.

Mirrors ``yolo_toy_regressor_torch.py``: ``mask_bytes_to_boxes`` (:98-111, without the PIL decode: takes the mask
array; the reference labels with skimage's default full connectivity — here ``csrc/mask_kernels.hip``, a union-find
over the pixels of the whole batch), ``box_to_xywh_norm`` (:113-119), the largest-box rule of
``get_samples_from_dataset`` (:194-199).  The VOC parquet files, PIL decoding and resizing stay with the caller (no
dataset ships with the reference).  There is no host implementation: masks given as numpy arrays are uploaded."""
import numpy as np
import torch

from . import _lib as L

_ws = {}


def masks_to_boxes(masks, max_boxes=64, dims=None):
    """masks: [N,H,W] or [N,H,W,CH], any integer / bool / float dtype, CUDA tensor (or numpy array: uploaded), non-zero = object.
    This batched call is the fast path; ``mask_to_boxes`` / ``mask_to_target_box`` below are per-sample conveniences with
    the reference's signatures (one upload, seven launches and one host read each: not for a DataLoader's inner loop).
    -> (boxes int32 [N,max_boxes,4] as (x_min, y_min, x_max, y_max) in label order, rows past count undefined;
        count int32 [N]; best_box int32 [N,4] (-1 for an empty mask); best_xywh fp32 [N,4]) as :98-119 and :195-196.
    ``dims`` [N,2] = (H_i, W_i): the original sizes of masks that were zero-padded to a common H x W."""
    if not torch.is_tensor(masks):
        masks = torch.from_numpy(np.ascontiguousarray(masks)).cuda()
    L.require_cuda(masks)
    if masks.dim() not in (3, 4):
        raise L.YfsError(f"masks_to_boxes: expected [N,H,W] or [N,H,W,CH], got {tuple(masks.shape)}")
    if masks.dtype != torch.uint8:
        # any dtype counts as "non-zero = object", as the reference's np.array(mask) != 0 does (:103); one pass on the device
        masks = (masks != 0).to(torch.uint8)
    masks = masks.contiguous()
    n, h, w = masks.shape[:3]
    ch = masks.shape[3] if masks.dim() == 4 else 1
    dev = masks.device
    if dims is not None:
        dims = torch.as_tensor(dims, dtype=torch.int32).to(dev).contiguous()
        if tuple(dims.shape) != (n, 2):
            raise L.YfsError(f"masks_to_boxes: dims must be [N, 2], got {tuple(dims.shape)}")
    boxes = torch.empty(n, max(max_boxes, 1), 4, device=dev, dtype=torch.int32)
    count = torch.empty(n, device=dev, dtype=torch.int32)
    best_box = torch.empty(n, 4, device=dev, dtype=torch.int32)
    best_xywh = torch.empty(n, 4, device=dev, dtype=torch.float32)
    nbytes = L.lib().yfs_mask_to_boxes_workspace_bytes(L.i64(n), L.c_int(h), L.c_int(w))
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _ws.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _ws[key] = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    L.check(L.lib().yfs_mask_to_boxes_u8(L.ptr(masks), L.i64(n), L.c_int(h), L.c_int(w), L.c_int(ch), L.ptr(dims), L.ptr(boxes),
                                         L.c_int(max_boxes), L.ptr(count), L.ptr(best_box), L.ptr(best_xywh), L.ptr(ws),
                                         L.c_size(nbytes), L.stream()), "yfs_mask_to_boxes_u8")
    return boxes[:, :max_boxes], count, best_box, best_xywh


def mask_to_boxes(mask):
    """One mask ([H,W] or [H,W,ch], numpy or CUDA tensor) -> list of (x_min, y_min, x_max, y_max), inclusive pixel
    coordinates, one per 8-connected component, in label (raster first-pixel) order — the return value of :98-111."""
    m = mask if torch.is_tensor(mask) else np.asarray(mask)
    cap = 64
    while True:
        boxes, count, _, _ = masks_to_boxes(m[None], max_boxes=cap)
        k = int(count[0])
        if k <= cap:
            return [tuple(b) for b in boxes[0, :k].cpu().tolist()]
        cap = k


def box_to_xywh_norm(box, H, W):
    """:113-119, same arithmetic (float64 intermediates, float32 result)."""
    x_min, y_min, x_max, y_max = box
    w = x_max - x_min + 1
    h = y_max - y_min + 1
    xc = x_min + (w - 1) / 2.0
    yc = y_min + (h - 1) / 2.0
    return np.array([xc / W, yc / H, w / W, h / H], dtype=np.float32)


def largest_box(boxes):
    """:196-197: the box with the largest inclusive pixel area (first one on ties, as np.argmax)."""
    areas = [(b[2] - b[0] + 1) * (b[3] - b[1] + 1) for b in boxes]
    return boxes[int(np.argmax(areas))]


def mask_to_target_box(mask):
    """mask -> [xc, yc, w, h] float32 of its largest component, or None when the mask is empty (:190-199)."""
    m = mask if torch.is_tensor(mask) else np.asarray(mask)
    _, count, _, xywh = masks_to_boxes(m[None], max_boxes=0)
    if int(count[0]) == 0:
        return None
    return xywh[0].cpu().numpy()


def encode_targets_v1(gt, S=7, B=2, C=20):
    """gt [N,G,5] fp32 CUDA (xc, yc, w, h, class; w <= 0 = padding) -> target [N,S,S,5B+C] for YoloV1Loss."""
    L.require_cuda(gt)
    if gt.dim() != 3 or gt.shape[2] != 5:
        raise L.YfsError("encode_targets_v1: gt must be [N, G, 5]")
    gt = gt.float().contiguous()
    n, g, _ = gt.shape
    out = torch.empty(n, S, S, 5 * B + C, device=gt.device, dtype=torch.float32)
    L.check(L.lib().yfs_encode_targets_v1_f32(L.ptr(gt), L.i64(n), L.c_int(g), L.c_int(S), L.c_int(B), L.c_int(C), L.ptr(out),
                                              L.stream()), "yfs_encode_targets_v1_f32")
    return out
