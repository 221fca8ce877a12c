"""Host side of the reference's data path after decoding: segmentation mask -> boxes ->
normalised [xc, yc, w, h] -> YOLOv1 grid targets (the last step on the GPU).

Mirrors ``yolo_toy_regressor_torch.py``: ``mask_bytes_to_boxes`` (:98-111, without the PIL decode: takes the mask
array; the reference labels with skimage's default full connectivity, done here with scipy.ndimage and a 3x3
structure), ``box_to_xywh_norm`` (:113-119), the largest-box rule of ``get_samples_from_dataset`` (:194-199).
The VOC parquet files, PIL decoding and resizing stay with the caller (no dataset ships with the reference)."""
import numpy as np
import torch

from . import _lib as L


def mask_to_boxes(mask):
    """mask: [H,W] or [H,W,ch] array, non-zero = object.  -> list of (x_min, y_min, x_max, y_max), inclusive pixel
    coordinates, one per 8-connected component, in label (raster first-pixel) order — as :98-111."""
    from scipy import ndimage
    arr = np.asarray(mask)
    mask_bin = np.any(arr != 0, axis=2) if arr.ndim == 3 else (arr != 0)
    lab, n = ndimage.label(mask_bin, structure=np.ones((3, 3), np.int32))
    boxes = []
    for sl in ndimage.find_objects(lab):
        (r0, r1), (c0, c1) = (sl[0].start, sl[0].stop), (sl[1].start, sl[1].stop)
        boxes.append((c0, r0, c1 - 1, r1 - 1))
    return boxes


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
    boxes = mask_to_boxes(mask)
    if not boxes:
        return None
    H, W = np.asarray(mask).shape[:2]
    return box_to_xywh_norm(largest_box(boxes), H, W)


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
