"""The CPU oracle against the golden vectors produced by the imported reference
(tests/golden/gen_golden.py).  Runs on CPU; no GPU, no reference needed."""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import model_ref


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_iou_variants_match_reference(golden_dir):
    g = _load(golden_dir, "g4_iou.npz")
    pred, gt = g["pred"], g["gt"]
    rows = oracle.iou_xywh(pred, gt)
    # iou_xywh_np returns python floats of fp32 arithmetic: bit-exact per row
    np.testing.assert_array_equal(rows.astype(np.float64), g["a4_np_rows"])
    assert abs(oracle.iou_xywh_mean(pred, gt) - float(g["a4_mean"])) < 1e-6
    np.testing.assert_array_equal(oracle.iou_a4b(pred, gt), g["a4b"])
    np.testing.assert_array_equal(oracle.iou_a4c(pred, gt), g["a4c"])


def test_iou_known_answers(golden_dir):
    # SURVEY.md §8(a) A4/A4b/A4c known answers, recorded from the reference
    g = _load(golden_dir, "g4_iou.npz")
    p, t = g["ka_pred"], g["ka_gt"]
    assert abs(oracle.iou_xywh_mean(p, t) - 0.1250000298) < 1e-9
    assert abs(float(oracle.iou_xywh(p, t)[0]) - 0.2500000596) < 1e-9
    assert abs(oracle.iou_xywh_mean(p, t) - float(g["ka_a4_mean"])) < 1e-9
    np.testing.assert_array_equal(oracle.iou_a4b(p, t), g["ka_a4b"])
    np.testing.assert_array_equal(oracle.iou_a4c(p, t), g["ka_a4c"])


def test_pairwise_iou_diagonal_is_paired(golden_dir):
    g = _load(golden_dir, "g4_iou.npz")
    pred, gt = g["pred"][:64], g["gt"][:64]
    m = oracle.pairwise_iou(pred, gt)
    np.testing.assert_array_equal(np.diag(m), oracle.iou_xywh(pred, gt))
    np.testing.assert_array_equal(oracle.pairwise_iou(gt, pred), m.T)   # bit-symmetric


def test_single_box_decoders(golden_dir):
    g = _load(golden_dir, "g5_decode.npz")
    t = g["t"]
    # torch's vectorised exp differs from libm by ulps: 1e-6 relative
    np.testing.assert_allclose(oracle.decode_head(t, *g["prior"]), g["decode_head"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(oracle.decode_xywh(t), g["decode_xywh"], rtol=2e-6, atol=1e-7)
    ka = oracle.decode_head(np.array([[0, 0, 0, 0], [1, -1, .5, -.5]], np.float32))
    np.testing.assert_allclose(ka, [[.5, .5, .25, .35], [.7310586, .26894143, .4121803, .21228573]], rtol=1e-6)


def test_decode_grid_reduces_to_decode_head(golden_dir):
    """S=1, A=1 grid decode is decode_head without its clamp."""
    g = _load(golden_dir, "g5_decode.npz")
    t = g["t"][3:200]
    raw = np.zeros((t.shape[0], 1, 1, 1, 5 + 3), np.float32)
    raw[:, 0, 0, 0, :4] = t
    boxes, score, cls = oracle.decode_grid(raw, g["prior"].reshape(1, 2))
    ref = g["decode_head"][3:200]
    unclamped = (ref[:, 2] > 1e-4) & (ref[:, 2] < 1) & (ref[:, 3] > 1e-4) & (ref[:, 3] < 1)
    np.testing.assert_allclose(boxes[unclamped, 0], ref[unclamped], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(score, 0.5 / 3, rtol=1e-6)   # sigmoid(0) * softmax of equal logits
    assert (cls == 0).all()


def test_mse_and_lr(golden_dir):
    g = _load(golden_dir, "g3_mse.npz")
    v, grad = oracle.mse(g["pred"], g["target"])
    assert abs(v - float(g["loss"])) < 1e-6 * max(1, abs(v))
    np.testing.assert_allclose(grad, g["grad"], rtol=1e-6, atol=1e-8)
    g7 = _load(golden_dir, "g7_lr.npz")
    np.testing.assert_array_equal([oracle.lr_schedule(e, 5) for e in range(135)], g7["lr"])
    np.testing.assert_array_equal([oracle.lr_schedule(e, 3) for e in range(135)], g7["lr_w3"])
    assert oracle.lr_schedule(0) == pytest.approx(0.0028) and oracle.lr_schedule(110) == 1e-4


def test_conv_block_c_oracle_vs_reference(golden_dir):
    g = _load(golden_dir, "g1_conv_blocks.npz")
    for ci in range(4):
        cin, cout, k, s = g[f"c{ci}_cfg"]
        x = g[f"c{ci}_x"].transpose(0, 2, 3, 1)
        w = g[f"c{ci}_w"].transpose(0, 2, 3, 1)
        y = oracle.conv_fwd_nhwc(x, w, g[f"c{ci}_b"], stride=int(s))
        np.testing.assert_allclose(y.transpose(0, 3, 1, 2), g[f"c{ci}_y"], rtol=1e-5, atol=1e-6)
        dx, dw, db = oracle.conv_bwd_nhwc(x, w, y, g[f"c{ci}_gy"].transpose(0, 2, 3, 1), stride=int(s))
        np.testing.assert_allclose(dx.transpose(0, 3, 1, 2), g[f"c{ci}_gx"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(dw.transpose(0, 3, 1, 2), g[f"c{ci}_gw"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(db, g[f"c{ci}_gb"], rtol=1e-4, atol=1e-5)


def test_conv_block_torch_oracle_vs_reference(golden_dir):
    g = _load(golden_dir, "g1_conv_blocks.npz")
    for ci in range(4):
        s = int(g[f"c{ci}_cfg"][3])
        x = torch.from_numpy(g[f"c{ci}_x"]).requires_grad_(True)
        w = torch.from_numpy(g[f"c{ci}_w"]).requires_grad_(True)
        b = torch.from_numpy(g[f"c{ci}_b"]).requires_grad_(True)
        y = model_ref.conv_block(x, w, b, s)
        y.backward(torch.from_numpy(g[f"c{ci}_gy"]))
        np.testing.assert_allclose(y.detach().numpy(), g[f"c{ci}_y"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(x.grad.numpy(), g[f"c{ci}_gx"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(w.grad.numpy(), g[f"c{ci}_gw"], rtol=1e-5, atol=1e-6)


def test_numpy_cnn_oracle(golden_dir):
    """practice/numpy_cnn.py is the reference's second, torch-independent conv/pool (A10)."""
    g = _load(golden_dir, "g8_numpy_cnn.npz")
    w = g["w"].transpose(3, 0, 1, 2)     # (KH,KW,Cin,Cout) -> (Cout,KH,KW,Cin)
    y = oracle.conv_fwd_nhwc(g["x"], w, None, stride=1, slope=1.0)
    np.testing.assert_allclose(y, g["y"], rtol=1e-5, atol=1e-6)
    dx, dw, _ = oracle.conv_bwd_nhwc(g["x"], w, np.ones_like(y), g["dy"], stride=1, slope=1.0)
    np.testing.assert_allclose(dx, g["dx"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dw.transpose(1, 2, 3, 0), g["dw"], rtol=1e-4, atol=2e-5)
    np.testing.assert_array_equal(oracle.maxpool2_fwd_nhwc(g["y"]), g["pool_y"])
    # numpy_cnn.MaxPool2x2 builds its argmax mask from a reshape that interleaves the window and
    # column axes (practice/numpy_cnn.py:104), so its backward is NOT a 2x2-window argmax; the hot
    # path uses torch's MaxPool2d (yolo_toy_regressor_torch.py:216), which pins the backward here.
    yt = torch.from_numpy(g["y"]).permute(0, 3, 1, 2).requires_grad_(True)
    torch.nn.functional.max_pool2d(yt, 2, 2).backward(torch.from_numpy(g["pool_dy"]).permute(0, 3, 1, 2))
    np.testing.assert_array_equal(oracle.maxpool2_bwd_nhwc(g["y"], g["pool_dy"]),
                                  yt.grad.permute(0, 2, 3, 1).numpy())


def test_feat_stack_vs_reference(golden_dir):
    g = _load(golden_dir, "g2_feat64.npz")
    assert str(g["torch_version"][0]) == torch.__version__, "fixture was generated with another torch"
    params = model_ref.init_params(seed=0, head_out=None)
    np.testing.assert_array_equal(params["feat.0.0.weight"].numpy().ravel()[:16], g["w0_head"])
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        h = x
        sums = []
        for i, cfg in enumerate(model_ref.FEAT_CFG):
            h = model_ref.feat_forward(h, params, upto=None) if False else (
                torch.nn.functional.max_pool2d(h, 2, 2) if cfg == "P" else
                model_ref.conv_block(h, params[f"feat.{i}.0.weight"], params[f"feat.{i}.0.bias"], cfg[3]))
            sums.append(float(h.double().sum()))
        np.testing.assert_allclose(sums, g["layer_sums"], rtol=1e-5)
        np.testing.assert_allclose(h.numpy(), g["y"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(model_ref.feat_forward(x, params).numpy(), g["y"], rtol=1e-5, atol=1e-7)
    keys = [k for k in g["state_keys"]]
    assert [k for k in keys if k.startswith("feat")] == [k for k in params]
    assert keys[-4:] == ["fc.1.weight", "fc.1.bias", "fc.4.weight", "fc.4.bias"]
    assert int(g["n_params"]) == 266056324
