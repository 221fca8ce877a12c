"""Fused fc.1 epilogue (yolo_toy_regressor_torch.py:258-263: Linear -> LeakyReLU(0.1) -> Dropout(0.2)):
the numpy oracle against torch and the Philox known answer on CPU; the HIP kernels against the oracle on the GPU
(bit-exact for y, the keep mask and dz; 1e-5 relative for the fp32 bias gradient, whose summation order differs)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import fc_ref

DEV = "cuda:0"


def test_philox_known_answer_and_bf16_rounding():
    # Random123 known-answer test for philox4x32-10 with counter 0, key 0
    assert [int(v) for v in fc_ref.philox4x32_10(0, np.array([0], np.uint64))[0]] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C,
                                                                                     0x9B00DBD8]
    x = np.random.default_rng(0).standard_normal(4096).astype(np.float32) * 3
    assert np.array_equal(fc_ref.to_bf16(x), torch.from_numpy(x).bfloat16().float().numpy())


def test_oracle_matches_torch_without_dropout():
    rng = np.random.default_rng(1)
    part = rng.standard_normal((1, 8, 64)).astype(np.float32)
    bias = rng.standard_normal(64).astype(np.float32)
    y, keep = fc_ref.fc_epilogue_fwd(part, bias, 0.1, 0.0)
    ref = F.leaky_relu(torch.from_numpy(part[0]) + torch.from_numpy(bias), 0.1).bfloat16().float().numpy()
    assert np.array_equal(y, ref) and keep.all()
    dy = fc_ref.to_bf16(rng.standard_normal((8, 64)).astype(np.float32))
    dz, db = fc_ref.fc_epilogue_bwd(dy, y, keep, 0.1, 0.0)
    z = (torch.from_numpy(part[0]) + torch.from_numpy(bias)).requires_grad_(True)
    F.leaky_relu(z, 0.1).backward(torch.from_numpy(dy))
    assert np.array_equal(dz, z.grad.bfloat16().float().numpy())
    np.testing.assert_allclose(db, dz.sum(0), rtol=1e-6, atol=1e-6)


def test_oracle_dropout_mask_statistics():
    part = np.ones((2, 64, 512), np.float32)
    y, keep = fc_ref.fc_epilogue_fwd(part, None, 0.1, 0.2, seed=1234)
    frac = 1.0 - keep.mean()
    assert abs(frac - 0.2) < 0.01
    assert np.array_equal(np.unique(y), np.array([0.0, 2.5], np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("s,m,n,p", [(8, 64, 4096, 0.0), (8, 64, 4096, 0.2), (1, 5, 64, 0.5), (3, 7, 1480, 0.2)])
def test_fc_epilogue_kernels_vs_oracle(s, m, n, p):
    from yolo_from_scratch_amd import fc_ops as fo
    rng = np.random.default_rng(s * 1000 + n)
    part = rng.standard_normal((s, m, n)).astype(np.float32)
    bias = rng.standard_normal(n).astype(np.float32)
    seed = 0x1234_5678_9ABC_DEF0 + n
    y, keep = fo.fc_epilogue_fwd(torch.from_numpy(part).to(DEV), torch.from_numpy(bias).to(DEV), 0.1, p, seed)
    y_ref, keep_ref = fc_ref.fc_epilogue_fwd(part, bias, 0.1, p, seed)
    assert np.array_equal(y.float().cpu().numpy(), y_ref)
    if p > 0:
        assert np.array_equal(keep.cpu().numpy(), keep_ref)
    else:
        assert keep is None
    dy = fc_ref.to_bf16(rng.standard_normal((m, n)).astype(np.float32))
    db = torch.full((n,), 1.0, device=DEV)
    dz = fo.fc_epilogue_bwd(torch.from_numpy(dy).to(DEV), y, keep, 0.1, p, db=db, accumulate=True)
    dz_ref, db_ref = fc_ref.fc_epilogue_bwd(dy, y_ref, keep_ref, 0.1, p)
    assert np.array_equal(dz.float().cpu().numpy(), dz_ref)
    np.testing.assert_allclose(db.cpu().numpy() - 1.0, db_ref, rtol=1e-5, atol=1e-4)


@pytest.mark.gpu
def test_fused_linear_module_train_eval_and_gradients():
    """HipLinear(fused_act) + FusedDropout placeholder == Linear -> LeakyReLU -> Dropout: exact semantics in eval
    mode and with p = 0; in train mode the mask drops ~p of the outputs, scales the rest by 1/(1-p) and the backward
    uses the same mask."""
    from yolo_from_scratch_amd import model as ym
    torch.manual_seed(0)
    lin = ym.HipLinear(256, 512, fused_act=0.1).to(DEV)
    drop = ym.FusedDropout(0.25)
    lin.fuse_dropout(drop)
    ref = torch.nn.Linear(256, 512)
    ref.load_state_dict({k: v.cpu() for k, v in lin.state_dict().items()})
    x = torch.rand(32, 256)
    xb = x.bfloat16().float()
    wb = ref.weight.detach().bfloat16().float()
    drop.eval()
    y = drop(lin(x.to(DEV))).detach()
    y_ref = F.leaky_relu(xb @ wb.t() + ref.bias.detach(), 0.1)
    assert float((y.float().cpu() - y_ref).abs().max()) < 2e-2 * float(y_ref.abs().max())
    drop.train()
    xg = x.to(DEV).requires_grad_(True)
    y = drop(lin(xg))
    dropped = (y == 0).float().mean().item()
    assert abs(dropped - 0.25) < 0.02
    kept = (y != 0).cpu()
    assert float(((y.float().cpu() * 0.75 - y_ref).abs() * kept).max()) < 2e-2 * float(y_ref.abs().max())
    y.float().sum().backward()
    # gradient flows only through kept elements: dL/db = sum over rows of keep/(1-p) * leaky'
    gb = ((y_ref > 0).float() + (y_ref <= 0).float() * 0.1) * kept.float() / 0.75
    np.testing.assert_allclose(lin.bias.grad.cpu().numpy(), gb.sum(0).numpy(), rtol=2e-2, atol=2e-2)
    assert xg.grad is not None and torch.isfinite(xg.grad).all()


@pytest.mark.gpu
@pytest.mark.parametrize("b,n_in,n_out,first,clip", [(64, 512, 256, True, False), (64, 1280, 384, False, True),
                                                     (20, 256, 128, False, False), (130, 384, 256, False, True),
                                                     (512, 256, 256, False, True)])   # 512 rows: the 8-GPU gathered step
def test_linear_wgrad_fused_with_sgd_step(b, n_in, n_out, first, clip):
    """yfs_linear_wgrad_sgd_bf16 == (dW = dy^T x in fp32 from the bf16 operands) followed by the clip + SGD formula of
    yolo_toy_regressor_torch.py:331-335, :378-381, to fp32 summation-order tolerance; the bf16 shadow is the rounding of
    the updated weight."""
    from yolo_from_scratch_amd import fc_ops as fo
    g = torch.Generator().manual_seed(b + n_in)
    x = torch.randn(b, n_in, generator=g).bfloat16()
    dy = (torch.randn(b, n_out, generator=g) * 0.1).bfloat16()
    p = torch.randn(n_out, n_in, generator=g)
    m = torch.randn(n_out, n_in, generator=g) * 0.01
    lr, mom, wd, max_norm, gs = 1e-2, 0.9, 5e-4, 1.0, 0.5
    gw = gs * (dy.double().t() @ x.double())
    sumsq = torch.tensor([float((gw ** 2).sum()) + 3.0])          # "+3": the rest of the model
    coef = min(1.0, max_norm / (float(sumsq.sqrt()) + 1e-6)) if clip else 1.0
    d = gw * coef + wd * p.double()
    m_ref = d if first else mom * m.double() + d
    p_ref = p.double() - lr * m_ref
    pd, md = p.to(DEV).clone(), m.to(DEV).clone()
    p16 = torch.zeros(n_out, n_in, device=DEV, dtype=torch.bfloat16)
    fo.linear_wgrad_sgd(x.to(DEV), dy.to(DEV), pd, md, p16, sumsq.to(DEV) if clip else None, max_norm if clip else 0.0, lr, mom,
                        wd, first, gs)
    np.testing.assert_allclose(md.cpu().numpy(), m_ref.float().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pd.cpu().numpy(), p_ref.float().numpy(), rtol=1e-5, atol=1e-6)
    assert torch.equal(p16, pd.bfloat16())
