#!/usr/bin/env python3
"""Generate golden vectors by importing the reference in THIS container.

Run once, here (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

The reference (/root/reference) is a set of Python scripts.  Its module-level
imports pull in two data-only packages that are not installed (`skimage`,
`torchsummary`); they are used by the data loader / a print helper only, never
by the model, loss, IoU or decode code exercised here, so empty module objects
are registered for them before import.  Only *data* is
written: inputs and the outputs the reference's own functions returned.

Fixtures (SURVEY.md §8(c) list):
  g1_conv_blocks.npz   Conv2D(...) blocks fwd + grads       yolo_toy_regressor_torch.py:206-211
  g2_feat64.npz        feat stack at [1,3,64,64]            yolo_toy_regressor_torch.py:218-254
  g3_mse.npz           nn.MSELoss fwd/grad                  yolo_toy_regressor_torch.py:377
  g4_iou.npz           three IoU variants                   :159-182, practice/..._3_torch.py:112-119,
                                                           practice/..._20251015.py:226-278
  g5_decode.npz        decode_head / decode_xywh            practice/..._3_torch.py:290-295, ..._20251015.py:411-421
  g6_boxloss.npz       ciou_loss / yolo_single_box_loss     practice/..._3_torch.py:121-138, ..._20251015.py:368-383
  g7_lr.npz            yolo_v1_lr_schedule(e), e in [0,135) yolo_toy_regressor_torch.py:272-280
  g8_numpy_cnn.npz     numpy_cnn.Conv2D fwd + dx, MaxPool   practice/numpy_cnn.py:22-80, :90-130
"""
import os
import sys
import types
import io
import contextlib

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    sys.dont_write_bytecode = True
    for name in ("skimage", "skimage.measure", "torchsummary"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["skimage.measure"].label = None
    sys.modules["skimage.measure"].regionprops = None
    sys.modules["torchsummary"].summary = None
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "practice"))
    import yolo_toy_regressor_torch as main_ref
    import yolo_toy_regressor_3_torch as ref3
    import yolo_toy_regressor_torch_20251015 as ref_old
    import numpy_cnn
    return main_ref, ref3, ref_old, numpy_cnn


def main():
    torch.set_num_threads(1)
    torch.use_deterministic_algorithms(True)
    main_ref, ref3, ref_old, numpy_cnn = _import_reference()
    meta = np.array([torch.__version__, np.__version__])

    # ---- G1: Conv2D blocks forward + grads ---------------------------------
    g1 = {"torch_version": meta}
    cases = [(8, 16, 3, 1), (16, 8, 1, 1), (3, 8, 7, 2), (8, 8, 3, 2)]
    for ci, (cin, cout, k, s) in enumerate(cases):
        torch.manual_seed(ci)
        blk = main_ref.Conv2D(cin, cout, k, s)
        x = torch.rand(2, cin, 12, 12, requires_grad=True)
        y = blk(x)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(100 + ci))
        y.backward(gy)
        g1[f"c{ci}_cfg"] = np.array([cin, cout, k, s])
        g1[f"c{ci}_w"] = blk[0].weight.detach().numpy().copy()
        g1[f"c{ci}_b"] = blk[0].bias.detach().numpy().copy()
        g1[f"c{ci}_x"] = x.detach().numpy().copy()
        g1[f"c{ci}_y"] = y.detach().numpy().copy()
        g1[f"c{ci}_gy"] = gy.numpy().copy()
        g1[f"c{ci}_gx"] = x.grad.numpy().copy()
        g1[f"c{ci}_gw"] = blk[0].weight.grad.numpy().copy()
        g1[f"c{ci}_gb"] = blk[0].bias.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g1_conv_blocks.npz"), **g1)

    # ---- G2: feat at 64x64 (weights regenerate from the seed) --------------
    torch.manual_seed(0)
    model = main_ref.YoloRegressorModel()
    model.eval()
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(7))
    sums, absmeans, shapes = [], [], []
    h = x
    with torch.no_grad():
        for m in model.feat:
            h = m(h)
            sums.append(float(h.double().sum()))
            absmeans.append(float(h.double().abs().mean()))
            shapes.append(list(h.shape))
        # one full 448 forward pins the fc head too (eval: dropout off)
        x448 = torch.rand(1, 3, 448, 448, generator=torch.Generator().manual_seed(8))
        y448 = model(x448)
    names = [k for k, _ in model.state_dict().items()]
    np.savez_compressed(
        os.path.join(OUT, "g2_feat64.npz"), torch_version=meta, x=x.numpy(),
        y=h.numpy(), layer_sums=np.array(sums), layer_absmeans=np.array(absmeans),
        layer_shapes=np.array(shapes), state_keys=np.array(names),
        n_params=np.array(sum(p.numel() for p in model.parameters())),
        w0_head=model.feat[0][0].weight.detach().numpy().ravel()[:16].copy(),
        fc4_w_head=model.fc[4].weight.detach().numpy().ravel()[:16].copy(),
        y448=y448.numpy(), x448_seed=np.array(8))
    del model

    # ---- G3: MSE -------------------------------------------------------------
    g = torch.Generator().manual_seed(3)
    p = torch.randn(8, 4, generator=g, requires_grad=True)
    t = torch.rand(8, 4, generator=g)
    loss = torch.nn.MSELoss()(p, t)
    loss.backward()
    np.savez_compressed(os.path.join(OUT, "g3_mse.npz"), pred=p.detach().numpy(),
                        target=t.numpy(), loss=loss.detach().numpy(), grad=p.grad.numpy())

    # ---- G4: IoU variants ------------------------------------------------------
    g = torch.Generator().manual_seed(4)
    n = 1000
    pred = torch.rand(n, 4, generator=g)
    gt = torch.rand(n, 4, generator=g)
    pred[:, 2:] = pred[:, 2:] * 0.6 + 0.02
    gt[:, 2:] = gt[:, 2:] * 0.6 + 0.02
    # degenerate rows: identical, disjoint, zero-area, negative width, touching edges
    pred[0] = gt[0]
    pred[1] = torch.tensor([0.1, 0.1, 0.1, 0.1]); gt[1] = torch.tensor([0.9, 0.9, 0.1, 0.1])
    pred[2] = torch.tensor([0.5, 0.5, 0.0, 0.0]); gt[2] = torch.tensor([0.5, 0.5, 0.0, 0.0])
    pred[3] = torch.tensor([0.5, 0.5, -0.2, 0.3]); gt[3] = torch.tensor([0.5, 0.5, 0.2, 0.3])
    pred[4] = torch.tensor([0.25, 0.5, 0.5, 0.5]); gt[4] = torch.tensor([0.75, 0.5, 0.5, 0.5])
    pred[5] = torch.tensor([0.5, 0.5, 0.4, 0.4]); gt[5] = torch.tensor([0.5, 0.5, 0.2, 0.2])
    pred[6] = torch.tensor([0.2, 0.2, 0.1, 0.1]); gt[6] = torch.tensor([0.8, 0.8, 0.1, 0.1])
    pred[7] = torch.tensor([0.5, 0.5, -0.2, -0.3]); gt[7] = torch.tensor([0.5, 0.5, 0.2, 0.3])
    iou_mean_a4 = main_ref.iou_xywh_torch(pred, gt)
    iou_np_rows = np.array([main_ref.iou_xywh_np(pred[i].numpy(), gt[i].numpy()) for i in range(n)],
                           dtype=np.float64)
    iou_a4b = ref3.box_iou_xyxy(ref3.xywh_to_xyxy(pred), ref3.xywh_to_xyxy(gt))
    iou_a4c = ref_old.iou_xywh_torch(pred, gt)
    ka_p = torch.tensor([[.5, .5, .4, .4], [.2, .2, .1, .1]])
    ka_g = torch.tensor([[.5, .5, .2, .2], [.8, .8, .1, .1]])
    np.savez_compressed(
        os.path.join(OUT, "g4_iou.npz"), pred=pred.numpy(), gt=gt.numpy(),
        a4_mean=np.array(iou_mean_a4, dtype=np.float64), a4_np_rows=iou_np_rows,
        a4b=iou_a4b.numpy(), a4c=iou_a4c.numpy(),
        ka_pred=ka_p.numpy(), ka_gt=ka_g.numpy(),
        ka_a4_mean=np.array(main_ref.iou_xywh_torch(ka_p, ka_g), dtype=np.float64),
        ka_a4_np0=np.array(main_ref.iou_xywh_np(ka_p[0].numpy(), ka_g[0].numpy()), dtype=np.float64),
        ka_a4b=ref3.box_iou_xyxy(ref3.xywh_to_xyxy(ka_p), ref3.xywh_to_xyxy(ka_g)).numpy(),
        ka_a4c=ref_old.iou_xywh_torch(ka_p, ka_g).numpy())

    # ---- G5: single-box decoders ----------------------------------------------
    g = torch.Generator().manual_seed(5)
    t = torch.randn(1000, 4, generator=g) * 2.0
    t[0] = torch.tensor([0., 0., 0., 0.]); t[1] = torch.tensor([1., -1., .5, -.5])
    t[2] = torch.tensor([30., -30., 12., -12.])      # saturating inputs exercise both clamps
    np.savez_compressed(
        os.path.join(OUT, "g5_decode.npz"), t=t.numpy(),
        prior=np.array([ref3.PRIOR_W, ref3.PRIOR_H], dtype=np.float32),
        decode_head=ref3.decode_head(t).numpy(), decode_xywh=ref_old.decode_xywh(t).numpy())

    # ---- G6: CIoU + single box loss -------------------------------------------
    g = torch.Generator().manual_seed(6)
    pb = (torch.rand(64, 4, generator=g) * 0.6 + 0.2).requires_grad_(True)
    gb = torch.rand(64, 4, generator=g) * 0.6 + 0.2
    l, iou, ciou = ref3.ciou_loss(pb, gb)
    l.sum().backward()
    raw = torch.randn(64, 4, generator=g, requires_grad=True)
    tg = torch.rand(64, 4, generator=g) * 0.8 + 0.1
    sl = ref_old.yolo_single_box_loss(raw, tg)
    sl.backward()
    np.savez_compressed(
        os.path.join(OUT, "g6_boxloss.npz"), pb=pb.detach().numpy(), gb=gb.numpy(),
        ciou_loss=l.detach().numpy(), iou=iou.detach().numpy(), ciou=ciou.detach().numpy(),
        ciou_grad=pb.grad.numpy(), raw=raw.detach().numpy(), tg=tg.numpy(),
        single_loss=sl.detach().numpy(), single_grad=raw.grad.numpy())

    # ---- G7: LR schedule --------------------------------------------------------
    lr = np.array([main_ref.yolo_v1_lr_schedule(e) for e in range(135)], dtype=np.float64)
    lr3 = np.array([ref3.yolo_v1_lr_schedule(e) for e in range(135)], dtype=np.float64)
    es = main_ref.EarlyStoppingMax(patience=3, min_delta=0.0)
    seq = [0.1, 0.2, 0.2, 0.15, 0.3, 0.1, 0.1, 0.1]
    es_out = np.array([es.step(v) for v in seq], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "g7_lr.npz"), lr=lr, lr_w3=lr3,
                        es_seq=np.array(seq), es_out=es_out)

    # ---- G8: numpy_cnn oracle ----------------------------------------------------
    conv = numpy_cnn.Conv2D(Cin=3, Cout=4, k=3, seed=11)
    rng = np.random.default_rng(12)
    x = rng.standard_normal((2, 6, 6, 3)).astype(np.float32)
    w0 = conv.W.copy()
    y = conv.forward(x)
    dy = rng.standard_normal(y.shape).astype(np.float32)
    dx = conv.backward(dy, lr=0.5)
    dW = (w0 - conv.W) / 0.5       # the reference applies SGD inside backward (:78-79)
    pool = numpy_cnn.MaxPool2x2()
    yp = pool.forward(y)
    dyp = rng.standard_normal(yp.shape).astype(np.float32)
    with contextlib.redirect_stdout(io.StringIO()):   # backward prints (:128-129)
        dxp = pool.backward(dyp, lr=0.0)
    np.savez_compressed(os.path.join(OUT, "g8_numpy_cnn.npz"), x=x, w=w0, y=y, dy=dy, dx=dx,
                        dw=dW.astype(np.float32), pool_y=yp, pool_dy=dyp, pool_dx=dxp)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
