"""ciou_loss (practice/yolo_toy_regressor_3_torch.py:121-138) and yolo_single_box_loss
(practice/yolo_toy_regressor_torch_20251015.py:368-383): the numpy oracle against the fixture the reference itself
produced (tests/golden/g6_boxloss.npz: values and autograd gradients); the HIP kernels against both (1e-5 relative, the
tolerance BASELINE.json states for fp32 loss / gradient)."""
import os

import numpy as np
import pytest
import torch

from oracle import boxloss_ref

DEV = "cuda:0"


def _close(a, b, what, rtol=1e-5, atol=1e-6):
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=atol, err_msg=what)


def test_oracle_against_reference_fixture(golden_dir):
    d = np.load(os.path.join(str(golden_dir), "g6_boxloss.npz"))
    loss, iou, ciou, _ = boxloss_ref.ciou_loss(d["pb"], d["gb"])
    _close(loss, d["ciou_loss"], "ciou loss")
    _close(iou, d["iou"], "iou")
    _close(ciou, d["ciou"], "ciou")
    _close(boxloss_ref.ciou_grad_fd(d["pb"], d["gb"]), d["ciou_grad"], "ciou gradient (finite differences)", rtol=2e-4, atol=2e-5)
    _close(boxloss_ref.yolo_single_box_loss(d["raw"], d["tg"]), d["single_loss"], "single-box loss")
    _close(boxloss_ref.single_box_grad_fd(d["raw"], d["tg"]), d["single_grad"], "single-box gradient", rtol=2e-4, atol=2e-6)


@pytest.mark.gpu
def test_hip_box_losses_against_reference_fixture(golden_dir):
    import yolo_from_scratch_amd as y
    d = np.load(os.path.join(str(golden_dir), "g6_boxloss.npz"))
    pb = torch.from_numpy(d["pb"]).to(DEV).requires_grad_(True)
    loss, iou, ciou = y.ciou_loss(pb, torch.from_numpy(d["gb"]).to(DEV))
    loss.sum().backward()
    _close(loss.detach().cpu(), d["ciou_loss"], "ciou loss")
    _close(iou.cpu(), d["iou"], "iou")
    _close(ciou.cpu(), d["ciou"], "ciou", atol=2e-6)
    _close(pb.grad.cpu(), d["ciou_grad"], "ciou gradient", rtol=2e-5, atol=2e-5)
    raw = torch.from_numpy(d["raw"]).to(DEV).requires_grad_(True)
    sl = y.yolo_single_box_loss(raw, torch.from_numpy(d["tg"]).to(DEV))
    sl.backward()
    _close(sl.detach().cpu(), d["single_loss"], "single-box loss")
    _close(raw.grad.cpu(), d["single_grad"], "single-box gradient", atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 257, 5000])
def test_hip_box_losses_against_oracle_random(n):
    import yolo_from_scratch_amd as y
    rng = np.random.default_rng(n)
    pb = (rng.uniform(size=(n, 4)) * 0.6 + 0.2).astype(np.float32)
    gb = (rng.uniform(size=(n, 4)) * 0.6 + 0.2).astype(np.float32)
    pb[0] = gb[0]                                        # identical boxes: ties in every max / min
    if n > 2:
        pb[1] = [0.1, 0.1, 0.05, 0.05]
        gb[1] = [0.9, 0.9, 0.05, 0.05]                   # disjoint
    pt = torch.from_numpy(pb).to(DEV).requires_grad_(True)
    loss, iou, ciou = y.ciou_loss(pt, torch.from_numpy(gb).to(DEV))
    (loss * 2.0).sum().backward()
    lo, io, co, _ = boxloss_ref.ciou_loss(pb, gb)
    _close(loss.detach().cpu(), lo, "ciou loss", atol=2e-6)
    _close(iou.cpu(), io, "iou")
    sel = slice(1, min(n, 200))                          # finite differences are invalid at the exact ties of row 0
    _close(pt.grad.cpu().numpy()[sel] / 2.0, boxloss_ref.ciou_grad_fd(pb, gb)[sel], "ciou gradient", rtol=5e-4, atol=5e-5)
    raw = rng.standard_normal((n, 4)).astype(np.float32) * 1.5
    tg = (rng.uniform(size=(n, 4)) * 0.8 + 0.1).astype(np.float32)
    rt = torch.from_numpy(raw).to(DEV).requires_grad_(True)
    sl = y.yolo_single_box_loss(rt, torch.from_numpy(tg).to(DEV))
    sl.backward()
    _close(sl.detach().cpu(), boxloss_ref.yolo_single_box_loss(raw, tg), "single-box loss", rtol=2e-5)
    m = min(n, 64)
    _close(rt.grad.cpu().numpy()[:m] * (n / m), boxloss_ref.single_box_grad_fd(raw[:m], tg[:m]), "single-box gradient",
           rtol=5e-4, atol=1e-6)
