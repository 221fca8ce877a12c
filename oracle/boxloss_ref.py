"""CPU oracle (test infrastructure only): numpy restatements of the reference's two single-box losses,
``ciou_loss`` (practice/yolo_toy_regressor_3_torch.py:106-138) and ``yolo_single_box_loss``
(practice/yolo_toy_regressor_torch_20251015.py:368-383), in the reference's op order.  Pinned by
tests/golden/g6_boxloss.npz, which gen_golden.py produced by running the reference's own functions (values and
autograd gradients).  Gradients here are central differences in float64 — independent of the hand-written backward
in the HIP kernels."""
import numpy as np


def ciou_loss(pred, gt, dtype=np.float32, alpha_const=None):
    p_, g_ = np.asarray(pred, dtype), np.asarray(gt, dtype)
    half = dtype(2.0)

    def xyxy(b):
        return np.stack([b[..., 0] - b[..., 2] / half, b[..., 1] - b[..., 3] / half, b[..., 0] + b[..., 2] / half,
                         b[..., 1] + b[..., 3] / half], -1)
    p, g = xyxy(p_), xyxy(g_)
    tl, br = np.maximum(p[..., :2], g[..., :2]), np.minimum(p[..., 2:], g[..., 2:])
    inter = np.prod(np.clip(br - tl, 0, None), -1)
    area_a = np.clip(p[..., 2] - p[..., 0], 0, None) * np.clip(p[..., 3] - p[..., 1], 0, None)
    area_b = np.clip(g[..., 2] - g[..., 0], 0, None) * np.clip(g[..., 3] - g[..., 1], 0, None)
    iou = inter / (area_a + area_b - inter + dtype(1e-9))
    pc, gc = (p[..., :2] + p[..., 2:]) / half, (g[..., :2] + g[..., 2:]) / half
    cd = ((pc - gc) ** 2).sum(-1)
    c2 = ((np.maximum(p[..., 2:], g[..., 2:]) - np.minimum(p[..., :2], g[..., :2])) ** 2).sum(-1) + dtype(1e-9)
    pw, ph = np.clip(p[..., 2] - p[..., 0], 1e-9, None), np.clip(p[..., 3] - p[..., 1], 1e-9, None)
    gw, gh = np.clip(g[..., 2] - g[..., 0], 1e-9, None), np.clip(g[..., 3] - g[..., 1], 1e-9, None)
    v = dtype(4.0 / np.pi ** 2) * (np.arctan(gw / gh) - np.arctan(pw / ph)) ** 2
    alpha = v / (dtype(1.0) - iou + v + dtype(1e-9)) if alpha_const is None else alpha_const
    ciou = iou - cd / c2 - alpha * v
    return np.clip(dtype(1.0) - ciou, 0, None).astype(dtype), iou.astype(dtype), ciou.astype(dtype), alpha


def ciou_grad_fd(pred, gt, h=1e-6):
    """d sum(loss) / d pred by central differences in float64, alpha frozen at its value at ``pred`` (no_grad)."""
    pred = np.asarray(pred, np.float64)
    alpha = ciou_loss(pred, gt, np.float64)[3]
    out = np.zeros_like(pred)
    for k in range(4):
        e = np.zeros(4)
        e[k] = h
        lp = ciou_loss(pred + e, gt, np.float64, alpha)[0]
        lm = ciou_loss(pred - e, gt, np.float64, alpha)[0]
        out[..., k] = (lp - lm) / (2 * h)
    return out


def _sl1(d):
    a = np.abs(d)
    return np.where(a < 1.0, 0.5 * d * d, a - 0.5)


def yolo_single_box_loss(raw, targets, dtype=np.float32):
    r, t = np.asarray(raw, dtype), np.asarray(targets, dtype)
    x, y = 1.0 / (1.0 + np.exp(-r[..., 0])), 1.0 / (1.0 + np.exp(-r[..., 1]))
    eps = dtype(1e-6)
    lxy = _sl1(x - t[..., 0]).mean() + _sl1(y - t[..., 1]).mean()
    lwh = _sl1(r[..., 2] - np.log(t[..., 2] + eps)).mean() + _sl1(r[..., 3] - np.log(t[..., 3] + eps)).mean()
    return dtype(2.0 * lxy + 4.0 * lwh)


def single_box_grad_fd(raw, targets, h=1e-6):
    raw = np.asarray(raw, np.float64)
    out = np.zeros_like(raw)
    it = np.nditer(raw, flags=["multi_index"])
    for _ in it:
        i = it.multi_index
        rp, rm = raw.copy(), raw.copy()
        rp[i] += h
        rm[i] -= h
        out[i] = (yolo_single_box_loss(rp, targets, np.float64) - yolo_single_box_loss(rm, targets, np.float64)) / (2 * h)
    return out
