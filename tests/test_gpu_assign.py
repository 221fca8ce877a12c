"""Kernel set 4 (YOLOv2 target assignment + ignore-IoU) on the MI355X vs the CPU oracle: masks bit-exact."""
import numpy as np
import pytest
import torch

import oracle
import yolo_from_scratch_amd as y
from yolo_from_scratch_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _case(N, S, G, seed, kmax=3, crowd=False):
    gt = synth.gt_boxes(N, G, seed=seed, kmax=kmax)
    if crowd:   # several gts in one cell / same anchor: lowest index must win
        gt[:, 1, :4] = gt[:, 0, :4] + 1e-3
        gt[:, 2, :2] = gt[:, 0, :2]
    raw = synth.raw_head_v2(N, S, 5, 20, seed=seed + 1)
    pred, _, _ = oracle.decode_grid(raw, synth.VOC_ANCHORS)
    # pull some predictions onto gts so the ignore mask has hits
    rng = np.random.default_rng(seed + 2)
    for n in range(N):
        k = int((gt[n, :, 2] > 0).sum())
        idx = rng.integers(0, pred.shape[1], 40)
        pred[n, idx] = gt[n, rng.integers(0, k, 40), :4] * rng.uniform(0.85, 1.15, (40, 4)).astype(np.float32)
    return gt, pred


@pytest.mark.parametrize("N,S,G,kw", [(4, 13, 32, {}), (3, 10, 32, {}), (3, 19, 32, {"kmax": 20}), (5, 13, 8, {"crowd": True}),
                                      (2, 7, 1, {"kmax": 1}), (64, 13, 32, {"kmax": 32})])
def test_assign_masks_bit_exact(N, S, G, kw):
    gt, pred = _case(N, S, G, seed=N + S, **kw)
    obj_o, ign_o, tgt_o = oracle.assign_targets_v2(gt, synth.VOC_ANCHORS, S, pred, 0.6)
    obj, ign, tgt = y.assign_targets_v2(_t(gt), _t(synth.VOC_ANCHORS), S, _t(pred), 0.6)
    np.testing.assert_array_equal(obj.cpu().numpy(), obj_o)
    np.testing.assert_array_equal(ign.cpu().numpy(), ign_o)
    np.testing.assert_allclose(tgt.cpu().numpy(), tgt_o, rtol=1e-5, atol=1e-6)
    assert ign_o.sum() > 0 and obj_o.sum() > 0


def test_assign_without_pred_and_empty_image():
    gt = synth.gt_boxes(3, 32, seed=4)
    gt[1] = 0                      # an image with no gts
    obj_o, ign_o, tgt_o = oracle.assign_targets_v2(gt, synth.VOC_ANCHORS, 13, None)
    obj, ign, tgt = y.assign_targets_v2(_t(gt), _t(synth.VOC_ANCHORS), 13)
    np.testing.assert_array_equal(obj.cpu().numpy(), obj_o)
    np.testing.assert_array_equal(ign.cpu().numpy(), np.zeros_like(ign_o))
    np.testing.assert_allclose(tgt.cpu().numpy(), tgt_o, rtol=1e-5, atol=1e-6)
    assert obj_o[1].sum() == 0


def test_assign_multiscale_cycle():
    """Config 5: S cycles 10..19 (320..608 / 32)."""
    for S in range(10, 20):
        gt, pred = _case(2, S, 32, seed=S)
        obj_o, ign_o, _ = oracle.assign_targets_v2(gt, synth.VOC_ANCHORS, S, pred, 0.6)
        obj, ign, _ = y.assign_targets_v2(_t(gt), _t(synth.VOC_ANCHORS), S, _t(pred), 0.6)
        np.testing.assert_array_equal(obj.cpu().numpy(), obj_o)
        np.testing.assert_array_equal(ign.cpu().numpy(), ign_o)
