"""Torch-CPU restatement of the reference model and train step.  TEST INFRASTRUCTURE ONLY.

The reference's arithmetic for this part lives in PyTorch itself (nn.Conv2d, LeakyReLU,
MaxPool2d, Linear — SURVEY.md §8(c)), so the restatement calls the same third-party ops in
functional form.  Pinned by tests/golden/g1_conv_blocks.npz and g2_feat64.npz, which were
produced by the imported reference.

Follows yolo_toy_regressor_torch.py:206-211 (Conv2D block), :218-254 (feat layer list),
:258-264 (fc head), :314-339 (train step: forward, loss, backward, clip 1.0, SGD).
"""
import torch
import torch.nn.functional as F

# (cin, cout, k, stride) or "P" for MaxPool2d(2, 2) — yolo_toy_regressor_torch.py:218-254
FEAT_CFG = [
    (3, 192, 7, 2), "P",
    (192, 256, 3, 1), "P",
    (256, 128, 1, 1), (128, 256, 3, 1), (256, 256, 1, 1), (256, 512, 3, 1), "P",
    (512, 256, 1, 1), (256, 512, 3, 1), (512, 256, 1, 1), (256, 512, 3, 1),
    (512, 256, 1, 1), (256, 512, 3, 1), (512, 256, 1, 1), (256, 512, 3, 1),
    (512, 512, 1, 1), (512, 1024, 3, 1), "P",
    (1024, 512, 1, 1), (512, 1024, 3, 1), (1024, 512, 1, 1), (512, 1024, 3, 1),
    (1024, 1024, 3, 1), (1024, 1024, 3, 2),
    (1024, 1024, 3, 1), (1024, 1024, 3, 1),
]


def conv_block(x, w, b, stride, slope=0.1):
    """Conv2D(in,out,k,s): Conv2d(pad=k//2) + LeakyReLU(0.1) (:206-211)."""
    return F.leaky_relu(F.conv2d(x, w, b, stride=stride, padding=w.shape[-1] // 2), slope)


def feat_forward(x, params, upto=None):
    """params: dict of reference state-dict names -> tensors ('feat.<i>.0.weight', ...)."""
    for i, cfg in enumerate(FEAT_CFG):
        if upto is not None and i >= upto:
            break
        if cfg == "P":
            x = F.max_pool2d(x, 2, 2)
        else:
            x = conv_block(x, params[f"feat.{i}.0.weight"], params[f"feat.{i}.0.bias"], cfg[3])
    return x


def head_forward(f, params, p_drop=0.0, training=False):
    """fc: Flatten -> Linear -> LeakyReLU(0.1) -> Dropout -> Linear (:258-264)."""
    h = F.linear(f.flatten(1), params["fc.1.weight"], params["fc.1.bias"])
    h = F.leaky_relu(h, 0.1)
    h = F.dropout(h, p_drop, training)
    return F.linear(h, params["fc.4.weight"], params["fc.4.bias"])


def init_params(seed=0, head_out=4, in_hw=448, dtype=torch.float32):
    """Parameters created in the reference's order with torch's default Conv2d/Linear init
    (kaiming_uniform(a=sqrt(5)) weights, uniform(+-1/sqrt(fan_in)) bias), under manual_seed."""
    torch.manual_seed(seed)
    params = {}
    hw = in_hw
    for i, cfg in enumerate(FEAT_CFG):
        if cfg == "P":
            hw //= 2
            continue
        cin, cout, k, s = cfg
        m = torch.nn.Conv2d(cin, cout, k, stride=s, padding=k // 2)   # consumes RNG as the reference does
        params[f"feat.{i}.0.weight"] = m.weight.detach().to(dtype)
        params[f"feat.{i}.0.bias"] = m.bias.detach().to(dtype)
        hw = (hw + 2 * (k // 2) - k) // s + 1
    if head_out is None:
        return params
    flat = 1024 * hw * hw
    for name, (fin, fout) in (("fc.1", (flat, 4096)), ("fc.4", (4096, head_out))):
        m = torch.nn.Linear(fin, fout)
        params[f"{name}.weight"] = m.weight.detach().to(dtype)
        params[f"{name}.bias"] = m.bias.detach().to(dtype)
    return params


def sgd_step(params, grads, bufs, lr=1e-3, momentum=0.9, wd=5e-4, max_norm=1.0):
    """clip_grad_norm_(max_norm) then SGD(momentum, weight_decay) (:333, :378-381)."""
    if max_norm is not None:
        total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    else:
        coef = 1.0
    for k, p in params.items():
        g = grads[k] * coef + wd * p
        if k not in bufs:
            bufs[k] = g.clone()
        else:
            bufs[k].mul_(momentum).add_(g)
        p.sub_(lr * bufs[k])


def train_step(params, bufs, x, target, loss_fn, lr=1e-3):
    """One reference step with dropout off: forward, loss, backward, clip, SGD."""
    leaves = {k: v.detach().requires_grad_(True) for k, v in params.items()}
    out = head_forward(feat_forward(x, leaves), leaves)
    loss = loss_fn(out, target)
    loss.backward()
    grads = {k: v.grad for k, v in leaves.items()}
    with torch.no_grad():
        sgd_step(params, grads, bufs, lr=lr)
    return float(loss)
