"""yolo_from_scratch_amd — the MI355X-native hot path of yolo-from-scratch behind the reference's
own Python call surface.  Everything here runs hand-written gfx950 HIP kernels from libyfs.so
(C ABI in include/yfs.h); there is no CPU fallback (the CPU oracle in oracle/ is test-only)."""
from ._lib import YfsError, lib, LIB_PATH  # noqa: F401
from .boxes import (iou_xywh_torch, iou_xywh_rows, iou_variant, pairwise_iou, decode_head, decode_xywh,  # noqa: F401
                    decode_grid, class_scores, decode_v1, nms, decode_nms, DecodeNms)
from .loss import YoloV1Loss, MSELoss, ciou_loss, yolo_single_box_loss  # noqa: F401
from .assign import assign_targets_v2, VOC_ANCHORS  # noqa: F401
from .anchors import estimate_size_priors, kmeans_anchors, anchor_assign, init_head_biases  # noqa: F401
from .metrics import mean_average_precision  # noqa: F401
from .data import mask_to_boxes, box_to_xywh_norm, largest_box, mask_to_target_box, encode_targets_v1  # noqa: F401
from .conv_ops import release_workspaces  # noqa: F401,E402
