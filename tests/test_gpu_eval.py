"""k-means anchors and mAP@0.5 against the numpy oracle (oracle/eval_ref.py) and
hand-computed answers.  No reference code exists for either (README.md:50, :56): parity unpinned, SPEC.md §6-§7."""
import numpy as np
import pytest
import torch

from oracle import eval_ref
import oracle
import yolo_from_scratch_amd as y
from yolo_from_scratch_amd import anchors as ya, metrics as ymx, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _wh(n, seed):
    rng = np.random.default_rng(seed)
    modes = np.array([[0.08, 0.12], [0.2, 0.45], [0.5, 0.3], [0.8, 0.85]], np.float32)
    wh = modes[rng.integers(0, 4, n)] * rng.uniform(0.7, 1.3, (n, 2)).astype(np.float32)
    return wh.astype(np.float32)


@pytest.mark.parametrize("n,k", [(500, 5), (64, 1), (2000, 9), (5, 5)])
def test_anchor_assign_and_kmeans_vs_oracle(n, k):
    wh = _wh(n, seed=n + k)
    anc = _wh(k, seed=7)
    a, best = ya.anchor_assign(torch.from_numpy(wh).to(DEV), torch.from_numpy(anc).to(DEV))
    a_o, best_o = eval_ref.anchor_assign(wh, anc)
    np.testing.assert_array_equal(a.cpu().numpy(), a_o)                   # index work: bit exact
    np.testing.assert_array_equal(best.cpu().numpy(), best_o)
    cent, miou = ya.kmeans_anchors(torch.from_numpy(wh).to(DEV), k=k)
    cent_o, miou_o = eval_ref.kmeans_anchors(wh, k=k)
    np.testing.assert_allclose(cent.numpy(), cent_o, rtol=1e-6)
    assert abs(miou - miou_o) < 1e-6
    areas = cent.numpy()[:, 0] * cent.numpy()[:, 1]
    assert np.all(np.diff(areas) >= 0)


def test_kmeans_recovers_separated_modes_and_size_prior():
    wh = _wh(4000, seed=1)
    cent, miou = ya.kmeans_anchors(torch.from_numpy(wh).to(DEV), k=4)
    want = np.array([[0.08, 0.12], [0.2, 0.45], [0.5, 0.3], [0.8, 0.85]], np.float32)
    np.testing.assert_allclose(cent.numpy(), want, rtol=0.06)
    assert miou > 0.7                  # +-30 % size jitter around four well separated modes
    # k=1 precursor of the reference: median w, h (practice/yolo_toy_regressor_3_torch.py:278-286)
    ys = torch.rand(3, 8, 4, generator=torch.Generator().manual_seed(0))
    pw, ph = ya.estimate_size_priors([(None, ys[i]) for i in range(3)])
    assert pw == pytest.approx(float(ys[..., 2].reshape(-1).median())) and ph == pytest.approx(float(ys[..., 3].reshape(-1).median()))


def test_map_known_answers():
    # one image, one class, 2 gts; detections by score: TP, FP, TP  ->  AP = 0.5*1 + 0.5*(2/3)
    gt = np.zeros((1, 4, 5), np.float32)
    gt[0, 0] = [0.3, 0.3, 0.2, 0.2, 1]
    gt[0, 1] = [0.7, 0.7, 0.2, 0.2, 1]
    boxes = np.array([[[0.3, 0.3, 0.2, 0.2], [0.5, 0.1, 0.1, 0.1], [0.7, 0.7, 0.2, 0.22], [0.3, 0.3, 0.2, 0.2]]], np.float32)
    score = np.array([[0.9, 0.8, 0.7, 0.6]], np.float32)             # the last one re-detects gt 0: duplicate -> FP
    cls = np.ones((1, 4), np.int32)
    keep = np.array([[0, 1, 2, 3]], np.int32)
    count = np.array([4], np.int32)
    t = lambda a: torch.from_numpy(a).to(DEV)
    m, aps = ymx.mean_average_precision(t(boxes), t(score), t(cls), t(keep), t(count), t(gt))
    assert aps == {1: pytest.approx(0.5 + 0.5 * 2 / 3)} and m == pytest.approx(0.5 + 0.5 * 2 / 3)
    # perfect detections -> 1.0; no detections -> 0.0
    m1, _ = ymx.mean_average_precision(t(gt[:, :2, :4].copy()), t(np.array([[0.9, 0.8]], np.float32)),
                                       t(np.ones((1, 2), np.int32)), t(np.array([[0, 1]], np.int32)),
                                       t(np.array([2], np.int32)), t(gt))
    m0, _ = ymx.mean_average_precision(t(boxes), t(score), t(cls), t(keep), t(np.array([0], np.int32)), t(gt))
    assert m1 == pytest.approx(1.0) and m0 == 0.0


def test_map_pipeline_vs_oracle():
    """decode -> NMS -> mAP on noisy copies of the ground truth, GPU path vs the loop oracle."""
    N, G = 16, 6
    rng = np.random.default_rng(3)
    gt = synth.gt_boxes(N, G=G, seed=9, kmin=1, kmax=5, C=4)
    A = 40
    boxes = rng.uniform(0.05, 0.95, (N, A, 4)).astype(np.float32)
    boxes[..., 2:] = boxes[..., 2:] * 0.4 + 0.05
    cls = rng.integers(0, 4, (N, A)).astype(np.int32)
    score = rng.uniform(0.0, 1.0, (N, A)).astype(np.float32)
    for n in range(N):                                   # plant jittered true boxes among the random ones
        for g in gt[n]:
            if g[2] > 0:
                j = rng.integers(0, A)
                boxes[n, j] = g[:4] * rng.uniform(0.9, 1.1, 4).astype(np.float32)
                cls[n, j] = int(g[4])
                score[n, j] = rng.uniform(0.5, 1.0)
    t = lambda a: torch.from_numpy(a).to(DEV)
    keep, count = y.nms(t(boxes), t(score), t(cls), 0.05, 0.5)
    m, aps = ymx.mean_average_precision(t(boxes), t(score), t(cls), keep, count, t(gt))
    m_o, aps_o = eval_ref.mean_average_precision(boxes, score, cls, keep.cpu().numpy(), count.cpu().numpy(), gt)
    assert set(aps) == set(aps_o)
    for c in aps:
        assert aps[c] == pytest.approx(aps_o[c], abs=1e-9)
    assert m == pytest.approx(m_o, abs=1e-9) and 0.0 < m <= 1.0
