"""Drop-in model surface on the MI355X: chained backbone fwd/bwd, state-dict contract, FlatSGD, and the
reference's own model outputs (golden g2, produced by the imported reference with torch.manual_seed(0))."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import model_ref
import yolo_from_scratch_amd as y
from yolo_from_scratch_amd import model as ym
from yolo_from_scratch_amd import train as yt

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def _bf(t):
    return t.to(torch.bfloat16).float()


def _mini():
    torch.manual_seed(3)
    return ym.Backbone(ym.Conv2D(3, 64, 7, s=2), ym.HipMaxPool2d(), ym.Conv2D(64, 128, 3), ym.Conv2D(128, 64, 1),
                       ym.HipMaxPool2d(), ym.Conv2D(64, 64, 3, s=2), ym.Conv2D(64, 128, 3))


class _RoundBf16(torch.autograd.Function):
    """The GPU chain stores every activation and every activation gradient in bf16; the CPU reference
    chain rounds at the same points so LeakyReLU masks and pool argmaxes are decided on identical values."""

    @staticmethod
    def forward(ctx, x):
        return _bf(x)

    @staticmethod
    def backward(ctx, g):
        return _bf(g)


def _mini_ref(x, params):
    h = x
    cfg = [(0, 2), "P", (2, 1), (3, 1), "P", (5, 2), (6, 1)]
    for c in cfg:
        if c == "P":
            h = F.max_pool2d(h, 2, 2)
        else:
            h = _RoundBf16.apply(model_ref.conv_block(h, params[f"{c[0]}.0.weight"], params[f"{c[0]}.0.bias"], c[1]))
    return h


@pytest.mark.parametrize("use_sinks", [False, True])
def test_chained_backbone_forward_backward(use_sinks):
    net = _mini()
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(_bf(p))                       # bf16-exact weights: both sides see identical operands
    ref_params = {k: v.detach().clone().contiguous().requires_grad_(True) for k, v in net.state_dict().items()}
    net = net.to(DEV)
    opt = yt.FlatSGD(net.parameters(), lr=0.0) if use_sinks else None
    x = _bf(torch.rand(4, 3, 64, 64, generator=torch.Generator().manual_seed(1)))
    out = net(x.to(DEV))
    ref = _mini_ref(x, ref_params)
    assert out.shape == ref.shape and out.dtype == torch.bfloat16
    assert _rel(out, ref) < 1e-2
    gy = _bf(torch.randn(ref.shape, generator=torch.Generator().manual_seed(2)))
    if opt:
        opt.zero_grad()
    out.backward(gy.to(DEV))
    ref.backward(gy)
    for (k, p) in net.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape
        assert _rel(p.grad, ref_params[k].grad) < 2e-2, k
    if opt:      # second backward without zero_grad accumulates into the arena
        net(x.to(DEV)).backward(gy.to(DEV))
        for (k, p) in net.named_parameters():
            assert _rel(p.grad.detach(), 2 * ref_params[k].grad) < 2e-2, k
        opt.zero_grad()
        net(x.to(DEV)).backward(gy.to(DEV))
        for (k, p) in net.named_parameters():
            assert _rel(p.grad.detach(), ref_params[k].grad) < 2e-2, k


def test_single_block_autograd_matches_chain():
    torch.manual_seed(0)
    blk = ym.Conv2D(64, 64, 3).to(DEV)
    x = torch.rand(2, 64, 8, 8, device=DEV, requires_grad=True)
    yb = blk(x)
    yb.float().square().sum().backward()
    w, b = blk[0].weight.detach().cpu().contiguous(), blk[0].bias.detach().cpu()
    xr = _bf(x.detach().cpu()).requires_grad_(True)
    wr = _bf(w).requires_grad_(True)
    yr = model_ref.conv_block(xr, wr, b, 1)
    yr.square().sum().backward()
    assert _rel(yb, yr) < 1e-2 and _rel(x.grad, xr.grad) < 2e-2 and _rel(blk[0].weight.grad, wr.grad) < 2e-2


def test_feat_golden_and_state_dict_contract(golden_dir):
    g = np.load(os.path.join(golden_dir, "g2_feat64.npz"))
    torch.manual_seed(0)
    net = ym.YoloRegressorModel()                 # same construction order as the reference => same weights
    np.testing.assert_array_equal(net.feat[0][0].weight.detach().numpy().ravel(order="K")[:0], [])   # layout differs
    np.testing.assert_array_equal(net.feat[0][0].weight.detach().contiguous().numpy().ravel()[:16], g["w0_head"])
    np.testing.assert_array_equal(net.fc[4].weight.detach().numpy().ravel()[:16], g["fc4_w_head"])
    assert list(net.state_dict().keys()) == [str(k) for k in g["state_keys"]]
    assert sum(p.numel() for p in net.parameters()) == int(g["n_params"]) == 266056324
    net = net.to(DEV).eval()
    with torch.no_grad():
        f = net.feat(torch.from_numpy(g["x"]).to(DEV))
        assert tuple(f.shape) == (1, 1024, 1, 1)
        # 24 bf16 layers deep: rounding compounds; the per-op 1e-2 contract is tested in test_gpu_conv.py
        assert _rel(f, torch.from_numpy(g["y"])) < 4e-2
        x448 = torch.rand(1, 3, 448, 448, generator=torch.Generator().manual_seed(int(g["x448_seed"])))
        out = net(x448.to(DEV))
        ref = torch.from_numpy(g["y448"])
        assert out.shape == ref.shape == (1, 4)
        assert float((out.cpu() - ref).abs().max()) < 4e-2 * float(ref.abs().max()) + 2e-3
    # a reference checkpoint (contiguous NCHW tensors) loads; ours loads into plain torch modules of that shape
    sd = {k: v.detach().cpu().contiguous() for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    conv = nn.Conv2d(3, 192, 7, stride=2, padding=3)
    conv.load_state_dict({"weight": sd["feat.0.0.weight"], "bias": sd["feat.0.0.bias"]})


def test_flat_sgd_matches_torch_sgd_with_clip():
    torch.manual_seed(1)
    shapes = [(64, 64, 3, 3), (64,), (100, 37), (100,)]
    ps = [nn.Parameter(torch.randn(s)) for s in shapes]
    ps[0].data = ps[0].data.contiguous(memory_format=torch.channels_last)
    ref = [nn.Parameter(p.detach().clone().contiguous()) for p in ps]
    ropt = torch.optim.SGD(ref, lr=1e-2, momentum=0.9, weight_decay=5e-4)
    for p in ps:
        p.data = p.data.to(DEV)
    opt = yt.FlatSGD(ps, lr=1e-2, momentum=0.9, weight_decay=5e-4, max_norm=1.0)
    for step in range(3):
        grads = [torch.randn(s, generator=torch.Generator().manual_seed(10 * step + i)) * (5.0 if step == 1 else 0.01)
                 for i, s in enumerate(shapes)]
        opt.zero_grad()
        ropt.zero_grad()
        for p, r, gr in zip(ps, ref, grads):
            (p * gr.to(DEV)).sum().backward()       # plain autograd path: accumulates into the arena views
            r.grad = gr.clone()
        nn.utils.clip_grad_norm_(ref, 1.0)
        ropt.step()
        opt.step()
        for p, r in zip(ps, ref):
            np.testing.assert_allclose(p.detach().cpu().numpy(), r.detach().numpy(), rtol=2e-6, atol=1e-7)


def test_train_one_epoch_and_evaluate_surface():
    """Reference call signatures (yolo_toy_regressor_torch.py:298-339) on a tiny model + MSE."""
    torch.manual_seed(0)
    net = nn.Sequential(ym.Backbone(ym.Conv2D(3, 64, 7, s=2), ym.HipMaxPool2d(), ym.Conv2D(64, 64, 3)), nn.Flatten(),
                        ym.HipLinear(64 * 8 * 8, 4)).to(DEV)
    xs = torch.rand(16, 3, 32, 32)
    ys = torch.rand(16, 4) * 0.5 + 0.25
    loader = [(xs[i:i + 8], ys[i:i + 8]) for i in (0, 8)]
    opt = yt.FlatSGD(net.parameters(), lr=1e-2, momentum=0.9, weight_decay=5e-4)
    loss_fn = y.MSELoss()
    l0 = yt.train_one_epoch(net, loader, DEV, opt, torch.amp.GradScaler(enabled=False), loss_fn, 1.0)
    for _ in range(15):
        l1 = yt.train_one_epoch(net, loader, DEV, opt, None, loss_fn, 1.0)
    vl, viou = yt.evaluate(net, loader, DEV, loss_fn)
    assert l1 < l0 and vl < l0 and 0.0 <= viou <= 1.0


def test_load_state_dict_after_flat_sgd_two_linear_layers():
    """The bf16 shadows are views of one arena and share its version counter: refreshing the second layer's
    shadow (stale after load_state_dict) must not invalidate what the first layer saved for backward, and
    the step must use the loaded weights."""
    torch.manual_seed(3)
    net = nn.Sequential(ym.HipLinear(64, 128), nn.LeakyReLU(0.1), ym.HipLinear(128, 32)).to(DEV)
    opt = yt.FlatSGD(net.parameters(), lr=1e-2, momentum=0.0, weight_decay=0.0)
    torch.manual_seed(4)
    src = nn.Sequential(nn.Linear(64, 128), nn.LeakyReLU(0.1), nn.Linear(128, 32))
    net.load_state_dict(src.state_dict())
    x = torch.rand(16, 64)
    opt.zero_grad()
    out = net(x.to(DEV))
    out.float().pow(2).mean().backward()
    ref = src(_bf(x))
    ref.pow(2).mean().backward()
    assert _rel(out, ref) < 2e-2
    assert _rel(net[0].weight.grad, src[0].weight.grad) < 3e-2
    assert _rel(net[2].weight.grad, src[2].weight.grad) < 3e-2
    opt.step()


def test_config1_forward_loss_decode_nms_vs_cpu_oracle():
    """BASELINE configs[0]: synthetic 224x224, batch 8, S=7 B=2 C=20 — backbone +
    build-defined head -> composite loss -> decode -> class-wise NMS, every stage against the CPU oracle.
    The model output is compared at the bf16 tolerance; the fp32 stages are then checked on the GPU's own
    output (identical inputs on both sides): loss/grad/boxes 1e-5, kept indices bit for bit."""
    import oracle
    from yolo_from_scratch_amd import synth
    import yolo_from_scratch_amd as y
    N = 8
    x = torch.rand(N, 3, 224, 224, generator=torch.Generator().manual_seed(0))
    params = model_ref.init_params(seed=0, head_out=1470, in_hw=224)
    torch.manual_seed(0)
    net = ym.YoloRegressorModel(head_out=1470, image_size=224).to(DEV).eval()
    for k, v in net.state_dict().items():                       # same construction order => same weights
        assert torch.equal(v.cpu().contiguous(), params[k]), k
    with torch.no_grad():
        ref = model_ref.head_forward(model_ref.feat_forward(x, params), params)
        out = net(x.to(DEV)).float()
    assert out.shape == ref.shape == (N, 1470)
    assert _rel(out, ref) < 4e-2                                  # 24 bf16 layers + 2 bf16 linears
    pred = out.view(N, 7, 7, 30)
    target = synth.v1_targets(N, seed=1)
    # fused loss + gradient
    lo, go = oracle.yolov1_loss(pred.cpu().numpy(), target)
    crit = y.YoloV1Loss(7, 2, 20)
    p = pred.clone().requires_grad_(True)
    loss = crit(p, torch.from_numpy(target).to(DEV))
    loss.backward()
    assert abs(float(loss) - lo[0]) <= 1e-5 * abs(lo[0])
    np.testing.assert_allclose(p.grad.cpu().numpy(), go, rtol=1e-5, atol=1e-6)
    # decode -> score -> NMS
    b, s, c = y.decode_v1(pred, B=2, C=20)
    bo, so, co_ = oracle.decode_v1(pred.cpu().numpy(), 2, 20)
    np.testing.assert_allclose(b.cpu().numpy(), bo, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(s.cpu().numpy(), so, rtol=1e-5, atol=1e-7)
    np.testing.assert_array_equal(c.cpu().numpy(), co_)
    keep, cnt = y.nms(b, s, c, 0.05, 0.5)
    keep_o, cnt_o = oracle.nms(b.cpu().numpy(), s.cpu().numpy(), c.cpu().numpy(), 0.05, 0.5)
    np.testing.assert_array_equal(cnt.cpu().numpy(), cnt_o)
    np.testing.assert_array_equal(keep.cpu().numpy(), keep_o)


def test_fit_epoch_loop_checkpoint_and_log(tmp_path):
    """The reference's epoch loop (yolo_toy_regressor_torch.py:397-432): LR schedule per epoch, best-only
    checkpoint with the reference's key names (dense NCHW tensors that plain torch modules load), quality log
    lines in utils/logger_quality.py's format, best weights reloaded at the end."""
    import json
    from yolo_from_scratch_amd import run
    torch.manual_seed(0)
    net = ym.YoloRegressorModel(head_out=4, image_size=64).to(DEV)
    xs = torch.rand(16, 3, 64, 64)
    ys = torch.rand(16, 4) * 0.5 + 0.25
    loader = [(xs[i:i + 8], ys[i:i + 8]) for i in (0, 8)]
    opt = yt.FlatSGD(net.parameters(), lr=1e-3, momentum=0.9, weight_decay=5e-4)
    best = tmp_path / "best_val_iou.pt"
    log = tmp_path / "q.log"
    hist = run.fit(net, loader, loader, DEV, opt, y.MSELoss(), total_epochs=3, warmup=2, patience=15, best_path=best,
                   log_path=log, side_dir=tmp_path, verbose=False)
    assert hist["epoch"] == [1, 2, 3]
    assert hist["lr"] == [yt.yolo_v1_lr_schedule(e, total_epochs=3, warmup=2) for e in range(3)]
    recs = [json.loads(line) for line in log.read_text().splitlines()]
    assert [r["epoch"] for r in recs if r["tag"] == "QUALITY"] == [1, 2, 3]
    assert any(r["tag"] == "QUALITY_BEST" for r in recs)
    sd = torch.load(best, weights_only=True)
    assert list(sd.keys()) == list(net.state_dict().keys())
    assert all(v.is_contiguous() and v.device.type == "cpu" for v in sd.values())
    conv = nn.Conv2d(3, 192, 7, stride=2, padding=3)                      # the reference's own module type
    conv.load_state_dict({"weight": sd["feat.0.0.weight"], "bias": sd["feat.0.0.bias"]})
    # the model now holds the best epoch's weights, and the optimiser's bf16 shadow follows them
    for k, v in net.state_dict().items():
        assert torch.equal(v.cpu().contiguous(), sd[k]), k
    w = net.feat[2][0].weight
    assert torch.equal(yt.bf16_shadow(w).float(), w.detach().to(torch.bfloat16).float())


def test_deferred_linear_gradient_step_equals_materialised(monkeypatch):
    """FlatSGD takes a large Linear weight's step from (x, dy) in one fused kernel (no materialised gradient); the
    trajectory must equal the materialised path (YFS_DEFER_FC=0) to fp32 summation-order tolerance, with clipping
    active, over several steps, and when two backward passes precede one step (accumulation falls back)."""
    def build(defer):
        monkeypatch.setenv("YFS_DEFER_FC", "1" if defer else "0")
        torch.manual_seed(11)
        net = nn.Sequential(ym.HipLinear(8192, 2048), nn.LeakyReLU(0.1), ym.HipLinear(2048, 64)).to(DEV)
        opt = yt.FlatSGD(net.parameters(), lr=5e-2, momentum=0.9, weight_decay=5e-4, max_norm=0.5)
        return net, opt
    (na, oa), (nb, ob) = build(True), build(False)
    assert getattr(na[0].weight, "_yfs_defer", False) and not getattr(nb[0].weight, "_yfs_defer", False)
    assert not getattr(na[2].weight, "_yfs_defer", False)           # small layer: ordinary gradient
    g = torch.Generator().manual_seed(5)
    for step in range(4):
        x = torch.randn(32, 8192, generator=g).to(DEV)
        for net, opt in ((na, oa), (nb, ob)):
            opt.zero_grad()
            net(x).float().pow(2).mean().backward()
            if step == 2:                                             # second backward before the step
                net(x * 0.5).float().pow(2).mean().backward()
            opt.step()
        if step < 2:
            assert float(na[0].weight.grad.abs().max()) == 0.0       # documented: not materialised in deferred mode
    for pa, pb in zip(na.parameters(), nb.parameters()):
        assert _rel(pa, pb) < 2e-5
    assert _rel(oa.flat_m, ob.flat_m) < 2e-5
    assert torch.equal(oa.flat_p16, oa.flat_p.bfloat16())


def test_deferred_linear_gradient_under_data_parallel_exchange(monkeypatch):
    """The N>1 form of the deferred step on one GPU: a GradComm of world size 2 whose collectives are replaced by
    what they return when both ranks hold the same data (all_gather = two copies, all_reduce SUM = x2).  The mean
    gradient then equals the single-process gradient, so the trajectory must match a run without any exchange; this
    drives the bucket layout that skips the deferred weight, the factor gather on the exchange stream, the
    2B-row fused kernel with grad_scale = 1/2 and the Gram-matrix norm with that scale."""
    from yolo_from_scratch_amd import dp
    world = 2

    def fake_all_gather(outs, t, group=None, async_op=False):
        for o in outs:
            o.copy_(t)

    def fake_all_reduce(t, op=None, group=None, async_op=False):
        t.mul_(world)
    monkeypatch.setattr(dp.dist, "all_gather", fake_all_gather)
    monkeypatch.setattr(dp.dist, "all_reduce", fake_all_reduce)

    def build(comm):
        torch.manual_seed(21)
        net = nn.Sequential(ym.HipLinear(8192, 2048), nn.LeakyReLU(0.1), ym.HipLinear(2048, 64)).to(DEV)
        return net, yt.FlatSGD(net.parameters(), lr=5e-2, momentum=0.9, weight_decay=5e-4, max_norm=0.5, comm=comm)
    comm = dp.GradComm(world, bucket_bytes=1 << 20)
    (na, oa), (nb, ob) = build(comm), build(None)
    assert na[0].weight._yfs_defer and id(na[0].weight) not in comm.param_bucket
    g = torch.Generator().manual_seed(6)
    for _ in range(3):
        x = torch.randn(32, 8192, generator=g).to(DEV)
        for net, opt in ((na, oa), (nb, ob)):
            opt.zero_grad()
            net(x).float().pow(2).mean().backward()
            opt.step()
    torch.cuda.synchronize()
    for pa, pb in zip(na.parameters(), nb.parameters()):
        assert _rel(pa, pb) < 2e-5
    assert _rel(oa.flat_m, ob.flat_m) < 2e-5


def test_config2_full_size_batch_properties():
    """BASELINE configs[1] at full size (448x448, batch 64): properties that need no CPU oracle.  (1) Every output pixel's
    reduction is independent of the batch it sits in, so the backbone's forward over 64 images is bit-identical to four
    forwards over 16.  (2) Weight gradients are sums over pixels: the gradient of the whole batch equals the sum of the
    four sub-batch gradients up to fp32 summation order (the split-K partition changes with the batch)."""
    torch.manual_seed(0)
    feat = ym.YoloRegressorModel(head_out=1470).feat.to(DEV)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(64, 3, 448, 448, generator=g).to(DEV)
    dy = (torch.randn(64, 1024, 7, 7, generator=g) * 1e-2).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    def run(xb, dyb):
        for p in feat.parameters():
            p.grad = None
        y = feat(xb)
        y.backward(dyb)
        return y.detach(), [p.grad.detach().float().clone() for p in feat.parameters()]
    # (the split-K form of the few-pixel layers picks its split from the pixel count, i.e. from the batch: without it
    #  the property is exact, with it the sums are re-associated before the one bf16 rounding per layer)
    from yolo_from_scratch_amd import conv_ops as _co
    y_split, _ = run(x, dy)
    _co.SPLIT_K = False
    try:
        y_full, g_full = run(x, dy)
        assert y_full.shape == (64, 1024, 7, 7)
        g_sum = None
        for i in range(0, 64, 16):
            y_part, g_part = run(x[i:i + 16], dy[i:i + 16])
            assert torch.equal(y_part, y_full[i:i + 16])
            g_sum = g_part if g_sum is None else [a + b for a, b in zip(g_sum, g_part)]
    finally:
        _co.SPLIT_K = True
    assert float((y_split.float() - y_full.float()).abs().max()) <= 2e-2 * float(y_full.float().abs().max())
    # the activation gradients are rounded to bf16 per layer identically in both runs (same per-pixel arithmetic),
    # so only the fp32 accumulation order over pixels differs
    for a, b in zip(g_full, g_sum):
        assert _rel(a, b) < 1e-4


def test_wrong_resolution_raises_the_references_shape_error():
    """the reference's full forward at 224x224 fails in fc.1 with torch's matmul shape error; the
    drop-in fails the same way (same exception type and wording) instead of reading out of bounds."""
    torch.manual_seed(0)
    model = ym.YoloRegressorModel(head_out=4, image_size=64).to(DEV)          # fc.1 sized for 64x64 inputs (1024 features)
    with pytest.raises(RuntimeError, match=r"mat1 and mat2 shapes cannot be multiplied \(2x4096 and 1024x4096\)"):
        model(torch.rand(2, 3, 128, 128, device=DEV))


@pytest.mark.parametrize("size,backward", [(320, True), (416, False), (608, False)])
def test_backbone_runs_at_every_multi_scale_resolution(size, backward):
    """BASELINE configs[4] cycles the input over 320 ... 608 (multiples of 32): the convolutional part of the model is
    resolution-free, so ``feat`` must give the reference's result at each of them (only ``fc`` is sized for one image
    size, test_wrong_resolution_raises_the_references_shape_error).  CPU chain of the same torch ops with the activations
    rounded to bf16 where the GPU chain stores them; at 320 also the weight and bias gradients."""
    torch.manual_seed(0)
    net = ym.YoloRegressorModel(head_out=4, image_size=64).feat
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(_bf(p))
    ref_params = {k: v.detach().clone().contiguous().requires_grad_(backward) for k, v in net.state_dict().items()}
    net = net.to(DEV)
    x = _bf(torch.rand(2, 3, size, size, generator=torch.Generator().manual_seed(size)))
    h = x
    for i, cfg in enumerate(model_ref.FEAT_CFG):
        if cfg == "P":
            h = F.max_pool2d(h, 2, 2)
        else:
            h = _RoundBf16.apply(model_ref.conv_block(h, ref_params[f"{i}.0.weight"], ref_params[f"{i}.0.bias"], cfg[3]))
    if backward:
        out = net(x.to(DEV))
    else:
        with torch.no_grad():
            out = net(x.to(DEV))
    assert tuple(out.shape) == tuple(h.shape) == (2, 1024, (size // 32 + 1) // 2, (size // 32 + 1) // 2)
    assert _rel(out, h) < 1e-2
    if backward:
        gy = _bf(torch.randn(h.shape, generator=torch.Generator().manual_seed(2)) * 1e-2)
        out.backward(gy.to(DEV))
        h.backward(gy)
        # Through 24 randomly initialised layers the gradient is chaotic in the forward rounding: two CPU runs of this
        # chain whose pre-activations differ by 1e-3 relative noise disagree by 2.7 % at feat.27 and 77 % at feat.0
        # (LeakyReLU masks and pool winners of near-tie elements flip).  So the layers next to the output are held to
        # the per-op tolerance and the rest to "same direction, same size"; the per-layer kernels are pinned one by
        # one in test_gpu_conv.py and the short chain above.
        for k, p in net.named_parameters():
            r = _rel(p.grad, ref_params[k].grad)
            assert torch.isfinite(p.grad).all() and r < (3e-2 if int(k.split(".")[0]) >= 25 else 0.8), (k, r)
