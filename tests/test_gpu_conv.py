"""Kernel set 1 (bf16 MFMA implicit-GEMM conv fwd / dgrad / wgrad) on the MI355X.

Oracle: the Conv2D block restated on the CPU (oracle.conv_*_nhwc for small shapes — C, double
accumulation; oracle.model_ref.conv_block = torch-CPU fp32 for the larger ones), fed the same
bf16-rounded inputs.  Tolerance from north_star: 1e-2 relative for bf16 conv."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
from oracle import model_ref
from yolo_from_scratch_amd import conv_ops as co

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RTOL = 1e-2


def _bf(t):
    return t.to(torch.bfloat16).float()


def _close(got, ref, what, atol_frac=2e-3):
    got, ref = got.float().cpu(), ref.float().cpu()
    atol = atol_frac * float(ref.abs().max()) + 1e-6
    bad = (got - ref).abs() > RTOL * ref.abs() + atol
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max abs err {float((got - ref).abs().max()):.4g}"


def _case(n, cin, cout, k, hw, seed):
    g = torch.Generator().manual_seed(seed)
    x = _bf(torch.rand(n, cin, hw, hw, generator=g))
    w = _bf((torch.rand(cout, cin, k, k, generator=g) * 2 - 1) / (cin * k * k) ** 0.5 * 2)
    b = (torch.rand(cout, generator=g) * 2 - 1) * 0.1
    return x, w, b


SHAPES = [  # n, cin, cout, k, stride, hw
    (2, 64, 64, 1, 1, 8), (2, 64, 128, 3, 1, 9), (1, 128, 64, 3, 1, 14), (2, 192, 256, 3, 1, 12),
    (3, 256, 128, 1, 1, 7), (2, 128, 192, 3, 1, 10), (2, 64, 128, 3, 2, 14), (1, 64, 64, 3, 2, 7),
    (2, 64, 64, 7, 2, 16), (1, 1024, 512, 1, 1, 14), (1, 512, 1024, 3, 1, 7),
]


@pytest.mark.parametrize("n,cin,cout,k,s,hw", SHAPES)
def test_conv_fwd_dgrad_wgrad(n, cin, cout, k, s, hw):
    x, w, b = _case(n, cin, cout, k, hw, seed=cin + cout + k)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    y_ref = model_ref.conv_block(xr, wr, br, s)
    gy = _bf(torch.randn(y_ref.shape, generator=torch.Generator().manual_seed(5)))
    y_ref.backward(gy)

    xd = co.nhwc(x.to(DEV), torch.bfloat16)
    w32 = co.nhwc(w.to(DEV))
    w16, wt16 = co.weight_prep(w32)
    assert torch.equal(w16.float().cpu(), w)            # bf16-exact inputs survive the cast
    y = co.conv_fwd(xd, w16, b.to(DEV), s)
    _close(y, y_ref.detach(), "forward")

    # backward from the oracle's own y so the LeakyReLU mask is identical on both sides
    yq = co.nhwc(y_ref.detach().to(DEV), torch.bfloat16)
    dz = co.leaky_bwd(co.nhwc(gy.to(DEV), torch.bfloat16), yq)
    dz_ref = gy * torch.where(y_ref.detach() > 0, 1.0, 0.1)
    _close(dz, dz_ref, "leaky backward")
    dzq = _bf(dz_ref)                                   # what the kernels actually consume
    xg, wg, bg = torch.autograd.grad(F.conv2d(xr, wr, br, stride=s, padding=k // 2), (xr, wr, br), dzq)
    dx = co.conv_dgrad(dz, wt16, (hw, hw), s)
    _close(dx, xg, "dgrad")
    dw, db = co.conv_wgrad(xd, dz, k, s)
    _close(dw, wg, "wgrad")
    _close(db, bg, "bias grad")
    dw2, db2 = co.conv_wgrad(xd, dz, k, s, dw=dw.clone(), db=db.clone())     # accumulate path
    _close(dw2, 2 * wg, "wgrad accumulate")
    _close(db2, 2 * bg, "bias grad accumulate")


def test_dgrad_fused_activation_mask():
    n, cin, cout, k, s, hw = 2, 128, 64, 3, 1, 10
    x, w, b = _case(n, cin, cout, k, hw, seed=3)
    x = x - 0.5                                          # x plays the previous block's LeakyReLU output
    dz = _bf(torch.randn(n, cout, hw, hw, generator=torch.Generator().manual_seed(1)))
    xg = torch.nn.grad.conv2d_input(x.shape, w, dz, stride=s, padding=k // 2)
    ref = xg * torch.where(x > 0, 1.0, 0.1)
    _, wt16 = co.weight_prep(co.nhwc(w.to(DEV)))
    dx = co.conv_dgrad(co.nhwc(dz.to(DEV), torch.bfloat16), wt16, (hw, hw), s, x_act=co.nhwc(x.to(DEV), torch.bfloat16))
    _close(dx, ref, "dgrad * leaky'(x)")


def test_conv_vs_c_oracle_and_golden(golden_dir):
    """Small case against the C restatement (double accumulation), and the reference's own G1 block
    Conv2D(8,8,3,s=2)-shaped geometry re-run at a channel count the kernels support."""
    x, w, b = _case(1, 64, 64, 3, 6, seed=11)
    y_c = oracle.conv_fwd_nhwc(x.permute(0, 2, 3, 1).numpy(), w.permute(0, 2, 3, 1).numpy(), b.numpy(), stride=1)
    w16, _ = co.weight_prep(co.nhwc(w.to(DEV)))
    y = co.conv_fwd(co.nhwc(x.to(DEV), torch.bfloat16), w16, b.to(DEV), 1)
    _close(y.permute(0, 2, 3, 1), torch.from_numpy(y_c), "forward vs C oracle")


def test_stem_im2col_path(golden_dir):
    """feat.0 = Conv2D(3,192,7,s=2): im2col (K=147 -> 192) + 1x1 implicit GEMM."""
    g = torch.Generator().manual_seed(2)
    x = _bf(torch.rand(2, 3, 32, 32, generator=g))
    w = _bf((torch.rand(192, 3, 7, 7, generator=g) * 2 - 1) * 0.1)
    b = torch.rand(192, generator=g) * 0.1
    y_ref = model_ref.conv_block(x, w, b, 2)
    cols = co.im2col(x.to(DEV), 7, 2, 192)
    wk = torch.zeros(192, 192, 1, 1)
    wk[:, :147, 0, 0] = w.permute(0, 2, 3, 1).reshape(192, 147)
    w16, _ = co.weight_prep(co.nhwc(wk.to(DEV)), want_wt16=False)
    y = co.conv_fwd(cols, w16, b.to(DEV), 1)
    _close(y, y_ref, "stem forward")
    # reference golden block Conv2D(3,8,7,s=2): same im2col, checked on the column matrix itself
    gg = np.load(os.path.join(golden_dir, "g1_conv_blocks.npz"))
    xg = torch.from_numpy(gg["c2_x"])
    cols = co.im2col(xg.to(DEV), 7, 2, 192).permute(0, 2, 3, 1).float().cpu()       # [N,OH,OW,192]
    wmat = torch.from_numpy(gg["c2_w"]).permute(0, 2, 3, 1).reshape(8, 147)
    y = torch.einsum("nhwk,ok->nohw", cols[..., :147], _bf(wmat)) + torch.from_numpy(gg["c2_b"]).view(1, 8, 1, 1)
    y = F.leaky_relu(y, 0.1)
    # the golden output was computed from un-rounded fp32 x and w; bf16 input rounding (2^-9 relative per
    # operand) shows up as an absolute error on outputs that cancel to ~0, hence the wider absolute term
    _close(y, torch.from_numpy(gg["c2_y"]), "im2col vs reference Conv2D(3,8,7,s=2) output", atol_frac=1e-2)


@pytest.mark.parametrize("hw,layout", [(32, "nchw_f32"), (64, "nhwc_bf16"), (30, "nchw_f32")])
def test_stem_space_to_depth_path(hw, layout, golden_dir):
    """feat.0 = Conv2D(3,192,7,s=2) as a 4x4/s1 conv over the space-to-depth image: forward, weight and bias grads."""
    g = torch.Generator().manual_seed(hw)
    x = _bf(torch.rand(2, 3, hw, hw, generator=g))
    w = _bf((torch.rand(192, 3, 7, 7, generator=g) * 2 - 1) * 0.1)
    b = torch.rand(192, generator=g) * 0.1
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    y_ref = model_ref.conv_block(x, wr, br, 2)
    xd = x.to(DEV) if layout == "nchw_f32" else co.nhwc(x.to(DEV), torch.bfloat16)
    xs = co.stem_space_to_depth(xd)
    oh, ow, hs, ws = co.stem_geometry(hw, hw)
    assert tuple(xs.shape) == (2, 16, hs, ws)
    y = co.stem_fwd(xs, co.stem_weight(co.nhwc(w.to(DEV))), b.to(DEV), oh, ow)
    _close(y, y_ref.detach(), "stem forward")
    dz = _bf(torch.randn(y_ref.shape, generator=g))
    wg, bg = torch.autograd.grad(F.conv2d(x, wr, br, stride=2, padding=3), (wr, br), dz)
    dw, db = co.stem_wgrad(xs, co.nhwc(dz.to(DEV), torch.bfloat16))
    _close(dw.view(192, 7, 7, 3).permute(0, 3, 1, 2), wg, "stem wgrad")
    _close(db, bg, "stem bias grad")


@pytest.mark.gpu
@pytest.mark.parametrize("n,c,hw,fuse", [(2, 64, 8, False), (3, 192, 14, True), (1, 128, 6, True)])
def test_maxpool_both_paths_vs_oracle(n, c, hw, fuse):
    """2x2 max pool (reference pool(), yolo_toy_regressor_torch.py:216): the x-based backward and the saved-index
    backward must agree bit for bit with each other and with the C oracle, ties included (first maximum wins)."""
    from yolo_from_scratch_amd import model as ym
    rng = np.random.default_rng(n * 100 + c + hw)
    x = rng.integers(-3, 4, (n, hw, hw, c)).astype(np.float32) * 0.5          # few distinct values: many ties, zeros
    dy = rng.standard_normal((n, hw // 2, hw // 2, c)).astype(np.float32)
    xb = torch.from_numpy(x).to(torch.bfloat16)
    dyb = torch.from_numpy(dy).to(torch.bfloat16)
    xd = xb.permute(0, 3, 1, 2).to(DEV)            # logical NCHW, physically NHWC (channels_last)
    dyd = dyb.permute(0, 3, 1, 2).to(DEV)
    y1 = ym._pool_fwd(xd)
    y2, idx = ym._pool_fwd_idx(xd)
    y_ref = oracle.maxpool2_fwd_nhwc(xb.float().numpy())
    assert torch.equal(y1, y2)
    np.testing.assert_array_equal(y1.permute(0, 2, 3, 1).float().cpu().numpy(), y_ref)
    dx1 = ym._pool_bwd(xd, dyd, fuse_leaky=fuse)
    dx2 = ym._pool_bwd_idx(y2, idx, dyd, fuse_leaky=fuse)
    assert torch.equal(dx1, dx2)
    dx_ref = oracle.maxpool2_bwd_nhwc(xb.float().numpy(), dyb.float().numpy())
    if fuse:                                       # LeakyReLU'(x) of the routed element, rounded once to bf16
        scale = np.where(xb.float().numpy() > 0, 1.0, 0.1).astype(np.float32)
        dx_ref = (torch.from_numpy(dx_ref) * torch.from_numpy(scale)).to(torch.bfloat16).float().numpy()
    np.testing.assert_array_equal(dx1.permute(0, 2, 3, 1).float().cpu().numpy(), dx_ref)


@pytest.mark.gpu
@pytest.mark.parametrize("n,cin,cout,k,hw", [(2, 64, 128, 3, 12), (1, 128, 192, 3, 10), (3, 64, 64, 1, 8), (2, 192, 256, 3, 16),
                                             (64, 64, 256, 3, 64)])
def test_conv_fwd_pool_fused_is_bit_identical(n, cin, cout, k, hw):
    """Conv2D block + pool() in one launch (pool in the GEMM epilogue, rows enumerated window-major) must equal the
    block followed by the pool kernel bit for bit: pooled values and saved window positions.  The shapes hit the
    128x128, 128x192, 128x64 and 256x256 tiles."""
    from yolo_from_scratch_amd import model as ym
    x, w, b = _case(n, cin, cout, k, hw, seed=cin + cout + k + hw)
    xd = co.nhwc(x.to(DEV), torch.bfloat16)
    w16, _ = co.weight_prep(co.nhwc(w.to(DEV)), want_wt16=False)
    bd = b.to(DEV)
    y_full = co.conv_fwd(xd, w16, bd, 1)
    y_ref, idx_ref = ym._pool_fwd_idx(y_full)
    y, idx = co.conv_fwd_pool(xd, w16, bd, 1)
    assert torch.equal(y, y_ref)
    assert torch.equal(idx, idx_ref)


@pytest.mark.gpu
def test_weight_prep_batch_from_bf16_shadow_matches_fp32_path():
    """The transposed, tap-reversed dgrad operand made from the optimiser's bf16 shadow (16-byte path of
    weight_prep_batch_kernel) equals the one made from the fp32 master weight, bit for bit, for several layers in one
    launch (3x3 and 1x1, Cin != Cout)."""
    g = torch.Generator().manual_seed(11)
    ws = [_bf(torch.randn(co_, ci_, k, k, generator=g)) for co_, ci_, k in ((128, 64, 3), (64, 192, 1), (256, 128, 3))]
    wd = [co.nhwc(w.to(DEV)) for w in ws]
    ref = [co.weight_prep(w)[1] for w in wd]
    out = co.weight_prep_batch([(w.to(torch.bfloat16).contiguous(memory_format=co.CL), True, True) for w in wd])
    for (w16, wt16), r, w in zip(out, ref, wd):
        assert torch.equal(wt16, r)
        assert torch.equal(w16.float(), w)


@pytest.mark.gpu
@pytest.mark.parametrize("n,hw,cout", [(2, 64, 192), (3, 128, 192), (2, 448, 192), (1, 320, 192), (2, 36, 128), (1, 72, 192)])
def test_stem_fwd_pool_fused_is_bit_identical(n, hw, cout):
    """Stem + pool in one launch == stem forward followed by the pool kernel, bit for bit (pooled values and window
    positions).  Cout = 192 with OH % 4 == 0 and OW % 32 == 0 (64, 128, 320, 448) runs the weight-stationary
    stem_pool_kernel, the other shapes the tiled kernel; where the weight-stationary kernel runs it is also compared
    with the tiled kernel forced through the knob."""
    from yolo_from_scratch_amd import _lib
    from yolo_from_scratch_amd import model as ym
    torch.manual_seed(3 + hw)
    x = torch.rand(n, 3, hw, hw) - 0.4
    w = _bf(torch.randn(cout, 3, 7, 7) * 0.05)
    b = torch.randn(cout) * 0.2
    xs = co.stem_space_to_depth(x.to(DEV))
    w2 = co.stem_weight(co.nhwc(w.to(DEV)))
    oh, ow, _, _ = co.stem_geometry(hw, hw)
    y_ref, idx_ref = ym._pool_fwd_idx(co.stem_fwd(xs, w2, b.to(DEV), oh, ow))
    y, idx = co.stem_fwd_pool(xs, w2, b.to(DEV), oh, ow)
    assert torch.equal(y, y_ref) and torch.equal(idx, idx_ref)
    _lib.set_knob("stem_kernel", 1)
    try:
        y_t, idx_t = co.stem_fwd_pool(xs, w2, b.to(DEV), oh, ow)
    finally:
        _lib.set_knob("stem_kernel", 0)
    assert torch.equal(y, y_t) and torch.equal(idx, idx_t)


@pytest.mark.gpu
@pytest.mark.parametrize("leaky", [False, True])
@pytest.mark.parametrize("n,hw,cout", [(2, 64, 192), (3, 36, 128), (1, 20, 64)])
def test_stem_wgrad_from_pooled_side(n, hw, cout, leaky):
    """Weight / bias gradient of [stem -> LeakyReLU -> pool] taken from (pooled gradient, argmax record, pooled
    activation) == pool backward (with the LeakyReLU mask) followed by the ordinary stem wgrad.  The dz tiles hold
    identical bf16 values; only the order in which pixels are summed differs (window-major), hence fp32 tolerance."""
    from yolo_from_scratch_amd import model as ym
    torch.manual_seed(5)
    x = torch.rand(n, 3, hw, hw)
    w = _bf(torch.randn(cout, 3, 7, 7) * 0.05)
    b = torch.randn(cout) * 0.1
    xs = co.stem_space_to_depth(x.to(DEV))
    w2 = co.stem_weight(co.nhwc(w.to(DEV)))
    oh, ow, _, _ = co.stem_geometry(hw, hw)
    yp, idx = co.stem_fwd_pool(xs, w2, b.to(DEV), oh, ow)
    gp = co.nhwc(torch.randn(n, cout, oh // 2, ow // 2).to(DEV), torch.bfloat16)
    dz = ym._pool_bwd_idx(yp, idx, gp, fuse_leaky=leaky)
    dw_ref, db_ref = co.stem_wgrad(xs, dz)
    dw, db = co.stem_wgrad_pool(xs, gp, idx, yp if leaky else None)
    assert float((dw - dw_ref).abs().max()) <= 1e-5 * float(dw_ref.abs().max()) + 1e-6
    assert float((db - db_ref).abs().max()) <= 1e-5 * float(db_ref.abs().max()) + 1e-6


def _forced_tile_outputs():
    outs = []
    for n, cin, cout, k, hw in [(3, 64, 256, 3, 15), (2, 128, 512, 1, 14), (1, 64, 256, 3, 30), (2, 192, 256, 3, 20)]:
        g = torch.Generator().manual_seed(cin + hw)
        x = co.nhwc((torch.rand(n, cin, hw, hw, generator=g) - 0.3).to(DEV), torch.bfloat16)
        w = co.nhwc(((torch.rand(cout, cin, k, k, generator=g) * 2 - 1) / (cin * k * k) ** 0.5).to(DEV))
        b = ((torch.rand(cout, generator=g) * 2 - 1) * 0.1).to(DEV)
        w16, wt16 = co.weight_prep(w)
        y = co.conv_fwd(x, w16, b, 1)
        outs.append(y)
        if hw % 2 == 0:
            outs.extend(co.conv_fwd_pool(x, w16, b, 1))
        act = co.nhwc(torch.randn(n, cout, hw, hw, generator=g).to(DEV), torch.bfloat16)
        w2 = co.nhwc(((torch.rand(cout, cout, k, k, generator=g) * 2 - 1) / (cout * k * k) ** 0.5).to(DEV))
        outs.append(co.conv_dgrad(y, co.weight_prep(w2)[1], (hw, hw), 1, x_act=act))
        outs.append(co.conv_dgrad(y, co.weight_prep(w2)[1], (hw, hw), 1))
    return [t.float().cpu() for t in outs]


@pytest.mark.parametrize("tile", [224, 2562, 64128, 8, 87])
def test_big_igemm_tiles_are_bit_identical_to_the_default(tile):
    """The 224x256 / 256x256 workgroup tiles and the 8-phase 256x256 kernel are picked by a cost model that only fires
    at full-size layers (and in tests/test_gpu_model.py's config-2 case); the YFS_KNOB_IGEMM_TILE tuning knob forces
    one, so forward, pool-fused forward, masked and plain dgrad run on ragged shapes (M not a multiple of the tile, odd
    and even K-step counts) with the forced tile.  Every output element accumulates the same K-steps in the same order
    whatever the tile, so the results must equal the default tile's bit for bit."""
    from yolo_from_scratch_amd import _lib as L
    assert L.get_knob("igemm_tile") == 0
    ref = _forced_tile_outputs()
    L.set_knob("igemm_tile", tile)
    try:
        got = _forced_tile_outputs()
    finally:
        L.set_knob("igemm_tile", 0)
    assert len(ref) == 18
    for a, b in zip(ref, got):
        assert torch.equal(a, b)


@pytest.mark.parametrize("n,cout,k,hw", [(2, 256, 3, 20), (3, 320, 3, 15), (1, 128, 1, 33), (5, 64, 3, 28)])
def test_dgrad_192_channel_8phase_tile_is_bit_identical_to_the_small_kernel(n, cout, k, hw):
    """feat.2's dgrad writes 192 channels: igemm8_kernel<EPI_MASK, FP1, FW1 = 1> (192-channel tile, second weight half-tile
    64 rows) forced through the tuning knob against igemm_kernel<128,192> on ragged pixel counts, odd and even K-step
    counts, 256- and 224-row tiles; masked dgrad only (the one epilogue this tile is built for)."""
    from yolo_from_scratch_amd import _lib as L
    g = torch.Generator().manual_seed(n + cout + hw)
    dz = co.nhwc(torch.randn(n, cout, hw, hw, generator=g).to(DEV), torch.bfloat16)
    act = co.nhwc(torch.randn(n, 192, hw, hw, generator=g).to(DEV), torch.bfloat16)
    w = co.nhwc(((torch.rand(cout, 192, k, k, generator=g) * 2 - 1) / (cout * k * k) ** 0.5).to(DEV))
    wt16 = co.weight_prep(w)[1]
    L.set_knob("igemm_tile", 128)
    try:
        ref = co.conv_dgrad(dz, wt16, (hw, hw), 1, x_act=act)
        L.set_knob("igemm_tile", 192)
        got = co.conv_dgrad(dz, wt16, (hw, hw), 1, x_act=act)
    finally:
        L.set_knob("igemm_tile", 0)
    assert got.shape == ref.shape == (n, 192, hw, hw)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("n,cin,cout,k,s,hw", [(2, 256, 256, 3, 1, 9), (3, 192, 256, 3, 1, 12), (1, 512, 512, 1, 1, 14),
                                               (2, 256, 512, 3, 2, 14), (5, 192, 256, 3, 1, 20), (2, 512, 256, 3, 1, 6)])
def test_wgrad_8phase_kernel_matches_default(n, cin, cout, k, s, hw):
    """wgrad8_kernel (256-wide tiles, forced through the YFS_KNOB_WGRAD_TILE tuning knob; split-K factors 1 and 3, ragged
    pixel counts, odd and even K-step counts, stride 2, the 192-channel ci tile) against the default weight-gradient
    kernel: same bf16 operands, fp32 accumulation, only the summation order over pixels differs."""
    from yolo_from_scratch_amd import _lib as L
    g = torch.Generator().manual_seed(n * 1000 + cin + cout + hw)
    x = co.nhwc((torch.rand(n, cin, hw, hw, generator=g) - 0.4).to(DEV), torch.bfloat16)
    oh, ow = co.out_hw(hw, hw, k, s)
    dz = co.nhwc(torch.randn(n, cout, oh, ow, generator=g).to(DEV), torch.bfloat16)
    L.set_knob("wgrad_tile", 128)          # wgrad_kernel with its own tile choice (auto may already pick wgrad8_kernel)
    try:
        dw_ref, db_ref = co.conv_wgrad(x, dz, k, s)
    finally:
        L.set_knob("wgrad_tile", 0)
    for splits in (1, 3):
        L.set_knob("wgrad_tile", 8)
        L.set_knob("wgrad_splits", splits)
        try:
            dw, db = co.conv_wgrad(x, dz, k, s)
            dw2, db2 = co.conv_wgrad(x, dz, k, s, dw=dw.clone(), db=db.clone())      # accumulate path
            dw3, _ = co.conv_wgrad(x, dz, k, s, want_db=False)
        finally:
            L.set_knob("wgrad_tile", 0)
            L.set_knob("wgrad_splits", 0)
        tol = 2e-5 * float(dw_ref.abs().max()) + 1e-6
        assert float((dw - dw_ref).abs().max()) <= tol, f"splits={splits}"
        assert float((dw3 - dw_ref).abs().max()) <= tol
        assert float((dw2 - 2 * dw_ref).abs().max()) <= 2 * tol
        tolb = 2e-5 * float(db_ref.abs().max()) + 1e-6
        assert float((db - db_ref).abs().max()) <= tolb
        assert float((db2 - 2 * db_ref).abs().max()) <= 2 * tolb


@pytest.mark.parametrize("n,cin,cout,k,s,hw", [(2, 64, 128, 3, 1, 12), (1, 128, 256, 1, 1, 14), (2, 64, 192, 3, 2, 16),
                                               (64, 64, 256, 3, 1, 28)])
def test_conv_folded_batchnorm_leaky_epilogue(n, cin, cout, k, s, hw):
    """Conv + BatchNorm (eval statistics) + LeakyReLU as one launch (north_star's block; README.md:30 of the reference)
    against torch-CPU conv2d -> batch_norm(training=False) -> leaky_relu on the same bf16-rounded operands, 1e-2; and
    scale = None reproduces the reference's bias-only block bit for bit.  The last shape runs on the 8-phase kernel."""
    x, w, b = _case(n, cin, cout, k, hw, seed=cin + cout + hw)
    g = torch.Generator().manual_seed(9)
    bn = torch.nn.BatchNorm2d(cout).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(cout, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(cout, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(cout, generator=g) * 0.2)
        bn.running_var.copy_(torch.rand(cout, generator=g) + 0.3)
        ref = F.leaky_relu(bn(F.conv2d(x, w, b, stride=s, padding=k // 2)), 0.1)
    xd = co.nhwc(x.to(DEV), torch.bfloat16)
    w16, _ = co.weight_prep(co.nhwc(w.to(DEV)), want_wt16=False)
    scale, shift = co.fold_batchnorm(bn, b)
    y = co.conv_fwd_affine(xd, w16, scale.to(DEV), shift.to(DEV), s)
    _close(y, ref, "conv + folded BN + leaky")
    assert torch.equal(co.conv_fwd_affine(xd, w16, None, b.to(DEV), s), co.conv_fwd(xd, w16, b.to(DEV), s))


# ---- the auto-dispatched 8-phase kernels at BASELINE config-2 layer shapes, straight against torch-CPU fp32 --------------
# (reference call sites: Conv2D blocks yolo_toy_regressor_torch.py:206-211, feat.18 = Conv2D(512, 1024, 3) :241,
#  feat.2 = Conv2D(192, 256, 3) :223; the batch is cut to what the CPU finishes in seconds, the per-image geometry and
#  the channel counts — which decide the kernel — are the real ones)
REAL_LAYERS = [  # name, n, cin, cout, k, hw, expected forward kernel, expected masked-dgrad kernel, expected wgrad kernel
    ("feat.18", 32, 512, 1024, 3, 28, "igemm8_kernel<224,256,epi0>", "igemm8_kernel<224,256,epi1>", "wgrad8_kernel<256,256"),
    ("feat.2", 8, 192, 256, 3, 112, "igemm8_kernel<224,256,epi0>", "igemm8_kernel<256,192,epi1>", "wgrad8_kernel<192,256"),
    # feat.26 / feat.27 = Conv2D(1024, 1024, 3) at 7x7 (:253-254), full batch: 3136 pixels, K = 9216 -> split-K form
    # feat.17 = Conv2D(512, 512, 1) at 28x28 (:240), full batch: the 1x1 layers with K >= 512 run on the 8-phase kernel too
    ("feat.17", 64, 512, 512, 1, 28, "igemm8_kernel<224,256,epi0>", "igemm8_kernel<224,256,epi1>", "wgrad_kernel<128,128"),
    ("feat.26", 64, 1024, 1024, 3, 7, "igemm8_kernel<224,256,epi4,ksplit4>+splitk_epilogue<epi0>",
     "igemm8_kernel<224,256,epi4,ksplit4>+splitk_epilogue<epi1>", "wgrad8_kernel<256,256"),
]


@pytest.mark.parametrize("name,n,cin,cout,k,hw,k_fwd,k_dgrad,k_wgrad", REAL_LAYERS)
def test_8phase_kernels_at_real_layer_shapes_vs_torch_cpu(name, n, cin, cout, k, hw, k_fwd, k_dgrad, k_wgrad):
    """Forward, masked dgrad and wgrad + bias gradient of feat.18 / feat.2 on whatever the library dispatches by itself
    (no knob), compared with torch-CPU fp32 on the same bf16-rounded operands at north_star's 1e-2; the dispatch query
    proves that the 8-phase kernels are what ran."""
    from yolo_from_scratch_amd import _lib as L
    assert L.get_knob("igemm_tile") == 0 and L.get_knob("wgrad_tile") == 0 and L.get_knob("wgrad_splits") == 0
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    x, w, b = _case(n, cin, cout, k, hw, seed=cin + hw)
    x = x - 0.35                                       # the block's input is a LeakyReLU output: both signs
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    z_ref = F.conv2d(xr, wr, br, stride=1, padding=k // 2)
    y_ref = F.leaky_relu(z_ref, 0.1)
    dz = _bf(torch.randn(z_ref.shape, generator=torch.Generator().manual_seed(7)))
    xg, wg, bg = torch.autograd.grad(z_ref, (xr, wr, br), dz)
    xg = xg * torch.where(x > 0, 1.0, 0.1)             # the dgrad epilogue applies the previous block's LeakyReLU'

    xd = co.nhwc(x.to(DEV), torch.bfloat16)
    w16, wt16 = co.weight_prep(co.nhwc(w.to(DEV)))
    y = co.conv_fwd(xd, w16, b.to(DEV), 1)
    assert L.last_kernel() == k_fwd, L.last_kernel()
    _close(y, y_ref.detach(), f"{name} forward")
    dzd = co.nhwc(dz.to(DEV), torch.bfloat16)
    dx = co.conv_dgrad(dzd, wt16, (hw, hw), 1, x_act=xd)
    assert L.last_kernel() == k_dgrad, L.last_kernel()
    _close(dx, xg, f"{name} masked dgrad")
    dw, db = co.conv_wgrad(xd, dzd, k, 1)
    assert L.last_kernel().startswith(k_wgrad), L.last_kernel()
    _close(dw, wg, f"{name} wgrad")
    _close(db, bg, f"{name} bias grad")


@pytest.mark.parametrize("n,cin,cout,k,s,hw", [(3, 64, 256, 3, 1, 15), (2, 128, 512, 1, 1, 14), (1, 192, 256, 3, 1, 30),
                                               (2, 64, 256, 3, 2, 14), (4, 320, 256, 3, 1, 7)])
def test_split_k_form_of_the_8phase_kernel_matches_the_unsplit_kernels(n, cin, cout, k, s, hw):
    """igemm8_kernel<EPI_PARTIAL> + splitk_epilogue_kernel forced through the tuning knob (84) on ragged shapes: pixel
    counts that are no multiple of the tile, K-step counts of 2 ... 45 that the split does not divide, stride 2, folded
    BN scale, masked and plain dgrad.  The K-steps are the same; only the fp32 summation order changes before the one
    rounding to bf16, so the outputs may differ by one bf16 ulp where a sum sits on a rounding boundary (or by the fp32
    error of the partial sums where the result cancels to nearly zero)."""
    from yolo_from_scratch_amd import _lib as L
    g = torch.Generator().manual_seed(n + cin + hw)
    x = co.nhwc((torch.rand(n, cin, hw, hw, generator=g) - 0.3).to(DEV), torch.bfloat16)
    w = co.nhwc(((torch.rand(cout, cin, k, k, generator=g) * 2 - 1) / (cin * k * k) ** 0.5).to(DEV))
    b = ((torch.rand(cout, generator=g) * 2 - 1) * 0.1).to(DEV)
    scale = (torch.rand(cout, generator=g) + 0.5).to(DEV)
    w16, _ = co.weight_prep(w)
    oh, ow = co.out_hw(hw, hw, k, s)
    dz = co.nhwc(torch.randn(n, cout, oh, ow, generator=g).to(DEV), torch.bfloat16)
    wd = co.nhwc(((torch.rand(cout, 256, k, k, generator=g) * 2 - 1) / (cout * k * k) ** 0.5).to(DEV))   # dgrad: 256 inputs
    wt16 = co.weight_prep(wd)[1]
    act = co.nhwc(torch.randn(n, 256, hw, hw, generator=g).to(DEV), torch.bfloat16)

    def run():
        outs = [co.conv_fwd(x, w16, b, s), co.conv_fwd_affine(x, w16, scale, b, s)]
        names = [L.last_kernel()]
        if s == 1:
            outs += [co.conv_dgrad(dz, wt16, (hw, hw), 1, x_act=act), co.conv_dgrad(dz, wt16, (hw, hw), 1)]
            names.append(L.last_kernel())
        return [t.float().cpu() for t in outs], names
    ref, _ = run()
    L.set_knob("igemm_tile", 84)
    try:
        got, names = run()
    finally:
        L.set_knob("igemm_tile", 0)
    assert all("epi4,ksplit" in nm for nm in names), names
    for a, r in zip(got, ref):
        err = (a - r).abs()
        # (one ulp of the result, or of the fp32 partial sums where the result itself cancels to almost nothing)
        bad = err > r.abs() * 2.0 ** -7 + 1e-5 * float(r.abs().max())
        i = int((err * bad).argmax())
        assert not bool(bad.any()), (f"{int(bad.sum())}/{bad.numel()} beyond one bf16 ulp; worst {float(a.flatten()[i])!r} vs "
                                     f"{float(r.flatten()[i])!r} at {i}")
        assert float((err > 0).float().mean()) < 0.02          # and the bulk is bit-identical



def test_stride2_dgrad_at_the_real_feat25_shape_vs_torch_cpu():
    """feat.25 = Conv2D(1024, 1024, 3, s=2) at 14x14 (:252), full batch: forward on the split-K 8-phase kernel, masked dgrad
    as four parity classes on the 224 x 256 tile (one workgroup per CU), weight gradient on wgrad8_kernel - all as the
    library dispatches them - against torch-CPU fp32 on the same bf16-rounded operands at 1e-2."""
    from yolo_from_scratch_amd import _lib as L
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    n, c, hw = 64, 1024, 14
    x, w, b = _case(n, c, c, 3, hw, seed=25)
    x = x - 0.35
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    z_ref = F.conv2d(xr, wr, br, stride=2, padding=1)
    dz = _bf(torch.randn(z_ref.shape, generator=torch.Generator().manual_seed(8)))
    xg, wg, bg = torch.autograd.grad(z_ref, (xr, wr, br), dz)
    xg = xg * torch.where(x > 0, 1.0, 0.1)
    xd = co.nhwc(x.to(DEV), torch.bfloat16)
    w16, wt16 = co.weight_prep(co.nhwc(w.to(DEV)))
    y = co.conv_fwd(xd, w16, b.to(DEV), 2)
    assert "epi4,ksplit" in L.last_kernel(), L.last_kernel()
    _close(y, F.leaky_relu(z_ref, 0.1).detach(), "feat.25 forward")
    dzd = co.nhwc(dz.to(DEV), torch.bfloat16)
    dx = co.conv_dgrad(dzd, wt16, (hw, hw), 2, x_act=xd)
    assert L.last_kernel() == "igemm_kernel<224,256,epi1,sub>", L.last_kernel()
    _close(dx, xg, "feat.25 masked dgrad")
    dw, db = co.conv_wgrad(xd, dzd, 3, 2)
    assert L.last_kernel().startswith("wgrad8_kernel<256,256"), L.last_kernel()
    _close(dw, wg, "feat.25 wgrad")
    _close(db, bg, "feat.25 bias grad")


@pytest.mark.gpu
@pytest.mark.parametrize("n,cin,cout,k,hw", [(2, 64, 128, 1, 7), (3, 128, 64, 1, 10), (1, 192, 128, 3, 9), (2, 256, 128, 1, 13),
                                             (5, 128, 256, 3, 6), (64, 256, 128, 1, 56)])
def test_dgrad_carried_through_the_pool_in_front_is_bit_identical(n, cin, cout, k, hw):
    """yfs_conv2d_dgrad_unpool_bf16 (the dgrad of a block behind a 2x2 pool, writing that pool's INPUT gradient) against
    the two launches it replaces: masked dgrad at the pooled resolution, then yfs_maxpool2_bwd_idx_bf16.  Ragged pixel
    counts, 64 / 128 / 192 / 256 output channels (every tile of the small kernel), 1x1 and 3x3, and feat.4's real shape
    (batch 64, 256 channels at 56x56 -> 112x112).  The record comes from the pool's own forward."""
    from yolo_from_scratch_amd import model as ym
    g = torch.Generator().manual_seed(n * 1000 + cin + hw)
    x_full = co.nhwc(torch.randn(n, cin, 2 * hw, 2 * hw, generator=g).to(DEV), torch.bfloat16)
    y_pooled, idx = ym._pool_fwd_idx(x_full)
    dz = co.nhwc(torch.randn(n, cout, hw, hw, generator=g).to(DEV), torch.bfloat16)
    w = co.nhwc(((torch.rand(cout, cin, k, k, generator=g) * 2 - 1) / (cin * k * k) ** 0.5).to(DEV))
    wt16 = co.weight_prep(w)[1]
    ref = ym._pool_bwd_idx(y_pooled, idx, co.conv_dgrad(dz, wt16, (hw, hw), 1, x_act=y_pooled), fuse_leaky=False)
    got = co.conv_dgrad_unpool(dz, wt16, y_pooled, idx)
    assert got.shape == ref.shape == (n, cin, 2 * hw, 2 * hw)
    assert torch.equal(got, ref)
    assert float((got != 0).float().mean()) < 0.2501      # one position per window


@pytest.mark.gpu
@pytest.mark.parametrize("op,n,cin,cout,knob", [("fwd", 5, 64, 256, 87), ("dgrad", 5, 256, 64, 87), ("dgrad_plain", 5, 256, 64, 87),
                                                ("dgrad", 6, 192, 64, 192)])
def test_persistent_8phase_kernel_on_more_tiles_than_cus_with_a_ragged_last_tile(op, n, cin, cout, knob):
    """igemm8_kernel on more tiles than the chip has CUs.  The forward form is persistent: one workgroup per CU walks tiles
    w, w + 256, ... and issues the next tile's first loads from inside the epilogue.  111 x 111 pixels x batch 5 (6) = 276
    tiles of 224 rows (289 of 256 rows x 192 channels), the last one ragged; in the XCD-aware order that tile is some
    workgroup's FIRST, so the row-guarded epilogue is also the one that starts a next tile.  Forward, masked and plain
    dgrad (one workgroup per tile, second round of the grid) against the 128 x 128 kernel: every output element
    accumulates the same K-steps in the same order, so the results are equal bit for bit.  The forward case also runs
    with YFS_KNOB_IGEMM_PERSIST = 1 (one workgroup per tile)."""
    from yolo_from_scratch_amd import _lib as L
    hw = 111
    g = torch.Generator().manual_seed(n * 100 + cin)
    if op == "fwd":
        x = co.nhwc(torch.randn(n, cin, hw, hw, generator=g).to(DEV), torch.bfloat16)
        w = co.nhwc(((torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / (cin * 9) ** 0.5).to(DEV))
        b = (torch.rand(cout, generator=g) - 0.5).to(DEV)
        w16 = co.weight_prep(w, want_wt16=False)[0]
        run = lambda: co.conv_fwd(x, w16, b, 1)
        want = "igemm8_kernel<224,256,epi0>"
    else:
        dz = co.nhwc(torch.randn(n, cout, hw, hw, generator=g).to(DEV), torch.bfloat16)
        act = co.nhwc(torch.randn(n, cin, hw, hw, generator=g).to(DEV), torch.bfloat16) if op == "dgrad" else None
        w = co.nhwc(((torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / (cout * 9) ** 0.5).to(DEV))
        wt16 = co.weight_prep(w)[1]
        run = lambda: co.conv_dgrad(dz, wt16, (hw, hw), 1, x_act=act)
        want = {87: "igemm8_kernel<224,256,epi%d>" % (1 if op == "dgrad" else 2), 192: "igemm8_kernel<256,192,epi1>"}[knob]
    split = co.SPLIT_K
    co.SPLIT_K = False
    try:
        L.set_knob("igemm_tile", 128)
        ref = run()
        L.set_knob("igemm_tile", knob)
        got = run()
        assert L.last_kernel() == want, L.last_kernel()
        again = run()            # and the same bits on a second launch (nothing left behind in LDS or flags)
        L.set_knob("igemm_persist", 1)
        per_tile = run()
    finally:
        L.set_knob("igemm_persist", 0)
        L.set_knob("igemm_tile", 0)
        co.SPLIT_K = split
    assert torch.equal(got, ref)
    assert torch.equal(again, ref)
    assert torch.equal(per_tile, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("name,cin,cout,k,hw,stride", [("feat.2", 192, 256, 3, 112, 1), ("feat.7", 256, 512, 3, 56, 1), ("feat.17", 512, 512, 1, 28, 1),
                                                      ("feat.18", 512, 1024, 3, 28, 1), ("feat.25", 1024, 1024, 3, 14, 2),
                                                      ("feat.26", 1024, 1024, 3, 7, 1)])
def test_full_size_layers_are_homogeneous_under_powers_of_two(name, cin, cout, k, hw, stride):
    """A size-independent property at BASELINE config 2's full sizes (batch 64), where the oracle would take minutes per
    layer: multiplying an operand by a power of two multiplies every bf16 product, every fp32 partial sum and every rounded
    output by it exactly, so dgrad(4 dz) == 4 dgrad(dz), masked or not, wgrad(x, 4 dz) == 4 wgrad(x, dz) == wgrad(4 x, dz), and with a zero
    bias forward(4 x) == 4 forward(x), pooled or not, BIT FOR BIT, whatever tile, split or persistent schedule the dispatch picks for the real shape (the kernels that run:
    8-phase persistent, 192-channel tile, split-K + reduction, strided dgrad, wgrad8 with its slab reduction)."""
    n = 64
    g = torch.Generator().manual_seed(len(name) * 7 + hw)
    ho, wo = co.out_hw(hw, hw, k, stride)
    x = co.nhwc((torch.rand(n, cin, hw, hw, generator=g) - 0.3).to(DEV), torch.bfloat16)
    dz = co.nhwc((torch.randn(n, cout, ho, wo, generator=g) * 0.05).to(DEV), torch.bfloat16)
    w = co.nhwc(((torch.rand(cout, cin, k, k, generator=g) * 2 - 1) / (cin * k * k) ** 0.5).to(DEV))
    wt16 = co.weight_prep(w)[1]
    dz4, x4 = dz * 4, x * 4
    for act in (x, None):
        a = co.conv_dgrad(dz, wt16, (hw, hw), stride, x_act=act)
        b = co.conv_dgrad(dz4, wt16, (hw, hw), stride, x_act=act)
        assert torch.equal(a * 4, b), (name, "dgrad", act is not None)
        assert float(a.float().abs().max()) > 0
    dw, db = co.conv_wgrad(x, dz, k, stride)
    dw4, db4 = co.conv_wgrad(x, dz4, k, stride)
    dwx, _ = co.conv_wgrad(x4, dz, k, stride)
    assert torch.equal(dw * 4, dw4) and torch.equal(db * 4, db4) and torch.equal(dw * 4, dwx), (name, "wgrad")
    assert float(dw.abs().max()) > 0
    # forward with a zero bias: leaky(4 a) == 4 leaky(a); the fused pool picks the same window positions
    w16 = co.weight_prep(w, want_wt16=False)[0]
    zero = torch.zeros(cout, device=DEV)
    y, y4 = co.conv_fwd(x, w16, zero, stride), co.conv_fwd(x4, w16, zero, stride)
    assert torch.equal(y * 4, y4), (name, "forward")
    if ho % 2 == 0 and wo % 2 == 0:
        (p1, i1), (p4, i4) = co.conv_fwd_pool(x, w16, zero, stride), co.conv_fwd_pool(x4, w16, zero, stride)
        assert torch.equal(p1 * 4, p4) and torch.equal(i1, i4), (name, "forward + pool")
