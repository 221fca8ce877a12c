"""Synthetic inputs of the BASELINE.json configs.  numpy only; seeds fixed."""
import numpy as np

VOC_ANCHORS = np.array([[1.3221, 1.73145], [3.19275, 4.00944], [5.05587, 8.09892], [9.47112, 4.84053],
                        [11.2364, 10.0071]], dtype=np.float32)


def gt_boxes(N, G=32, seed=1, kmin=1, kmax=3, C=20):
    """[N,G,5] (cx,cy,w,h,class); per image k~U{kmin..kmax} boxes, cx,cy~U(0,1), w,h~U(.05,.6); zero padded."""
    rng = np.random.default_rng(seed)
    gt = np.zeros((N, G, 5), np.float32)
    for n in range(N):
        k = int(rng.integers(kmin, kmax + 1))
        gt[n, :k, 0:2] = rng.uniform(0.0, 1.0, (k, 2))
        gt[n, :k, 2:4] = rng.uniform(0.05, 0.6, (k, 2))
        gt[n, :k, 4] = rng.integers(0, C, k)
    return gt


def v1_targets(N, S=7, B=2, C=20, seed=1):
    """YOLOv1 target grid [N,S,S,5B+C] from gt_boxes: slot 0 = (x,y offsets, w, h, 1), one-hot classes."""
    gt = gt_boxes(N, G=4, seed=seed, C=C)
    t = np.zeros((N, S, S, 5 * B + C), np.float32)
    for n in range(N):
        for b in gt[n]:
            if b[2] <= 0:
                continue
            col = min(int(np.floor(b[0] * S)), S - 1)
            row = min(int(np.floor(b[1] * S)), S - 1)
            if t[n, row, col, 4] > 0:
                continue
            t[n, row, col, 0] = b[0] * S - col
            t[n, row, col, 1] = b[1] * S - row
            t[n, row, col, 2:4] = b[2:4]
            t[n, row, col, 4] = 1.0
            t[n, row, col, 5 * B + int(b[4])] = 1.0
    return t


def raw_head_v2(N, S=13, A=5, C=20, seed=1234, obj_shift=-1.0):
    """Config 4 raw head [N,S,S,A,5+C] ~ N(0,1), objectness channel shifted by -1."""
    rng = np.random.default_rng(seed)
    raw = rng.standard_normal((N, S, S, A, 5 + C), dtype=np.float32)
    raw[..., 4] += obj_shift
    return raw
