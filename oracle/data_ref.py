"""CPU oracle (test infrastructure only) for yolo_from_scratch_amd/data.py: plain-Python restatements.
label_boxes follows yolo_toy_regressor_torch.py:98-111 (8-connected components as skimage.measure.label's default,
bbox per component in label order) with a flood fill instead of skimage (absent here: "parity unpinned" against
skimage itself; the box arithmetic :113-119 is pinned by the known answers in tests/).  encode_targets_v1 restates
SPEC.md's grid encoding (README.md:90-93 of the reference) as loops."""
import numpy as np


def label_boxes(mask_bin):
    m = np.asarray(mask_bin) != 0
    H, W = m.shape
    seen = np.zeros((H, W), bool)
    boxes = []
    for r in range(H):
        for c in range(W):
            if not m[r, c] or seen[r, c]:
                continue
            stack = [(r, c)]
            seen[r, c] = True
            r0 = r1 = r
            c0 = c1 = c
            while stack:
                y, x = stack.pop()
                r0, r1, c0, c1 = min(r0, y), max(r1, y), min(c0, x), max(c1, x)
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        yy, xx = y + dy, x + dx
                        if 0 <= yy < H and 0 <= xx < W and m[yy, xx] and not seen[yy, xx]:
                            seen[yy, xx] = True
                            stack.append((yy, xx))
            boxes.append((c0, r0, c1, r1))
    return boxes


def target_box(mask_bin, H=None, W=None):
    """:192-199 for one mask: boxes -> the largest inclusive area (np.argmax: first on ties) -> box_to_xywh_norm
    (:113-119, fp64 arithmetic, one rounding to fp32).  -> (box, xywh) or (None, None) for an empty mask."""
    boxes = label_boxes(mask_bin)
    if not boxes:
        return None, None
    areas = [(b[2] - b[0] + 1) * (b[3] - b[1] + 1) for b in boxes]
    x_min, y_min, x_max, y_max = best = boxes[int(np.argmax(areas))]
    h0, w0 = np.asarray(mask_bin).shape[:2]
    H, W = (h0 if H is None else H), (w0 if W is None else W)
    w = x_max - x_min + 1
    h = y_max - y_min + 1
    xc = x_min + (w - 1) / 2.0
    yc = y_min + (h - 1) / 2.0
    return best, np.array([xc / W, yc / H, w / W, h / H], dtype=np.float32)


def encode_targets_v1(gt, S=7, B=2, C=20):
    gt = np.asarray(gt, np.float32)
    N, G, _ = gt.shape
    t = np.zeros((N, S, S, 5 * B + C), np.float32)
    for n in range(N):
        for b in gt[n]:
            if not b[2] > 0:
                continue
            col = max(min(int(np.floor(np.float32(b[0] * np.float32(S)))), S - 1), 0)
            row = max(min(int(np.floor(np.float32(b[1] * np.float32(S)))), S - 1), 0)
            if t[n, row, col, 4] > 0:
                continue
            t[n, row, col, 0] = np.float32(b[0] * np.float32(S)) - np.float32(col)
            t[n, row, col, 1] = np.float32(b[1] * np.float32(S)) - np.float32(row)
            t[n, row, col, 2:4] = b[2:4]
            t[n, row, col, 4] = 1.0
            c = int(b[4])
            if 0 <= c < C:
                t[n, row, col, 5 * B + c] = 1.0
    return t
