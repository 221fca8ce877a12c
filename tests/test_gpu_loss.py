"""Kernel set 2 (fused YOLOv1 loss fwd+grad, MSE) on the MI355X vs the CPU oracle."""
import os

import numpy as np
import pytest
import torch

import oracle
import yolo_from_scratch_amd as y
from yolo_from_scratch_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RTOL = 1e-5   # north_star: loss and gradients within 1e-5 relative in fp32


def _pred(N, S=7, D=30, seed=0):
    rng = np.random.default_rng(seed)
    p = rng.uniform(-0.2, 1.2, (N, S, S, D)).astype(np.float32)
    p[0, 0, 0, 2] = 0.0   # exercises the sign(0) branch of the sqrt transform
    return p


@pytest.mark.parametrize("N,S,B,C", [(1, 7, 2, 20), (8, 7, 2, 20), (64, 7, 2, 20), (5, 13, 1, 3), (3, 7, 3, 20), (700, 7, 2, 20)])
def test_loss_and_grad_vs_oracle(N, S, B, C):
    t = synth.v1_targets(N, S, B, C, seed=1)
    p = _pred(N, S, 5 * B + C, seed=N)
    lo, go = oracle.yolov1_loss(p, t, B, C)
    pt = torch.from_numpy(p).to(DEV).requires_grad_(True)
    crit = y.YoloV1Loss(S, B, C)
    loss = crit(pt, torch.from_numpy(t).to(DEV))
    loss.backward()
    np.testing.assert_allclose(crit.last_terms.cpu().numpy(), lo, rtol=RTOL)
    np.testing.assert_allclose(loss.item(), lo[0], rtol=RTOL)
    np.testing.assert_allclose(pt.grad.cpu().numpy(), go, rtol=RTOL, atol=1e-9)


def test_loss_grad_matches_finite_structure():
    """Scaling the upstream gradient scales the stashed gradient; flat [N,1470] head output is accepted."""
    N = 4
    t = synth.v1_targets(N)
    p = _pred(N, seed=5)
    lo, go = oracle.yolov1_loss(p, t)
    pt = torch.from_numpy(p.reshape(N, -1)).to(DEV).requires_grad_(True)
    (3.0 * y.YoloV1Loss()(pt, torch.from_numpy(t).to(DEV))).backward()
    np.testing.assert_allclose(pt.grad.cpu().numpy().reshape(p.shape), 3.0 * go, rtol=RTOL, atol=1e-9)


def test_loss_large_batch_linearity():
    """Full-size property (N=65,536, the size the HBM roofline is quoted on): the loss of the batch equals
    the mean of the losses of its halves, and the oracle agrees on a 256-image slice."""
    N = 65536
    t = synth.v1_targets(1024)
    t = np.tile(t, (64, 1, 1, 1))
    p = np.tile(_pred(2048, seed=9), (32, 1, 1, 1))
    pt, tt = torch.from_numpy(p).to(DEV), torch.from_numpy(t).to(DEV)
    crit = y.YoloV1Loss()
    full = crit(pt, tt).item()
    a = crit(pt[: N // 2], tt[: N // 2]).item()
    b = crit(pt[N // 2:], tt[N // 2:]).item()
    assert abs(full - 0.5 * (a + b)) <= 1e-5 * abs(full)
    lo, _ = oracle.yolov1_loss(p[:256], t[:256])
    np.testing.assert_allclose(crit(pt[:256], tt[:256]).item(), lo[0], rtol=RTOL)


def test_mse_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "g3_mse.npz"))
    pt = torch.from_numpy(g["pred"]).to(DEV).requires_grad_(True)
    loss = y.MSELoss()(pt, torch.from_numpy(g["target"]).to(DEV))
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=RTOL)
    np.testing.assert_allclose(pt.grad.cpu().numpy(), g["grad"], rtol=RTOL, atol=1e-9)
    assert loss.dim() == 0 and loss.grad_fn is not None
