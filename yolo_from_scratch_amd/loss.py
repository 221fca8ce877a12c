"""Loss entry points — host side of kernel set 2 (csrc/loss_kernels.hip).

``loss_fn(preds, targets) -> 0-dim tensor with grad_fn`` is the reference's contract
(yolo_toy_regressor_torch.py:329-330, nn.MSELoss() at :377).  ``YoloV1Loss`` is the composite loss the
reference only specifies (assets/yolo_loss_function.png, README.md:97-115); forward and gradient come
out of one kernel launch and backward just scales the stashed gradient.
"""
import torch

from . import _lib as L


class _YoloV1LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, B, C, l_coord, l_noobj, terms_out):
        L.require_cuda(pred, target)
        if pred.dim() != 4 or pred.shape[1] != pred.shape[2] or pred.shape[3] != 5 * B + C:
            raise L.YfsError(f"YoloV1Loss expects pred [N,S,S,{5 * B + C}], got {tuple(pred.shape)}")
        if target.shape != pred.shape:
            raise L.YfsError(f"YoloV1Loss: target shape {tuple(target.shape)} != pred shape {tuple(pred.shape)}")
        p = pred.detach().to(torch.float32).contiguous()
        t = target.detach().to(torch.float32).contiguous()
        N, S = p.shape[0], p.shape[1]
        ws_bytes = L.lib().yfs_yolov1_loss_workspace_bytes(L.i64(N), L.c_int(S))
        ws = torch.empty(ws_bytes // 4, device=p.device, dtype=torch.float32)
        out = torch.empty(5, device=p.device, dtype=torch.float32)
        need_grad = ctx.needs_input_grad[0]
        grad = torch.empty_like(p) if need_grad else None
        L.check(L.lib().yfs_yolov1_loss_fwd_bwd_f32(L.ptr(p), L.ptr(t), L.i64(N), L.c_int(S), L.c_int(B), L.c_int(C),
                                                    L.c_float(l_coord), L.c_float(l_noobj), L.ptr(out), L.ptr(grad),
                                                    L.ptr(ws), L.c_size(ws_bytes), L.stream()),
                "yfs_yolov1_loss_fwd_bwd_f32")
        if terms_out is not None:
            terms_out.copy_(out)
        ctx.grad = grad
        ctx.in_dtype = pred.dtype
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        grad = ctx.grad
        if grad is None:
            return (None,) * 7
        return (grad * g).to(ctx.in_dtype), None, None, None, None, None, None


class YoloV1Loss(torch.nn.Module):
    """Composite YOLOv1 loss: lambda_coord=5, lambda_noobj=0.5, sqrt on w/h, responsible box = argmax IoU.

    pred/target [N,S,S,5B+C]; target box slot 0 = (x,y cell offsets, w,h image-normalised, obj flag),
    classes one-hot.  Returns total/N.  ``last_terms`` holds (total, coord, obj, noobj, cls) of the last call.
    """

    def __init__(self, S=7, B=2, C=20, l_coord=5.0, l_noobj=0.5):
        super().__init__()
        self.S, self.B, self.C, self.l_coord, self.l_noobj = S, B, C, l_coord, l_noobj
        self.last_terms = None

    def forward(self, pred, target):
        if pred.dim() == 2:   # flat head output [N, S*S*(5B+C)]
            pred = pred.view(pred.shape[0], self.S, self.S, 5 * self.B + self.C)
        if target.dim() == 2:
            target = target.view(pred.shape)
        if self.last_terms is None or self.last_terms.device != pred.device:
            self.last_terms = torch.zeros(5, device=pred.device, dtype=torch.float32)
        return _YoloV1LossFn.apply(pred, target, self.B, self.C, self.l_coord, self.l_noobj, self.last_terms)


class _MseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        L.require_cuda(pred, target)
        if pred.shape != target.shape:
            raise L.YfsError(f"MSELoss: shapes differ {tuple(pred.shape)} vs {tuple(target.shape)}")
        p = pred.detach().to(torch.float32).contiguous()
        t = target.detach().to(torch.float32).contiguous()
        n = p.numel()
        ws_bytes = L.lib().yfs_mse_workspace_bytes(L.i64(n))
        ws = torch.empty(max(ws_bytes // 4, 1), device=p.device, dtype=torch.float32)
        out = torch.empty(1, device=p.device, dtype=torch.float32)
        grad = torch.empty_like(p) if ctx.needs_input_grad[0] else None
        L.check(L.lib().yfs_mse_fwd_bwd_f32(L.ptr(p), L.ptr(t), L.i64(n), L.ptr(out), L.ptr(grad), L.ptr(ws),
                                            L.c_size(ws_bytes), L.stream()), "yfs_mse_fwd_bwd_f32")
        ctx.grad = grad
        ctx.in_dtype = pred.dtype
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        if ctx.grad is None:
            return None, None
        return (ctx.grad * g).to(ctx.in_dtype), None


class MSELoss(torch.nn.Module):
    """nn.MSELoss() (mean) — the reference's loss_fn (yolo_toy_regressor_torch.py:377)."""

    def forward(self, pred, target):
        return _MseFn.apply(pred, target)


# ---- single-box losses of the reference's practice scripts ---------------------------------------------------------
class _CiouFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt):
        L.require_cuda(pred, gt)
        if pred.shape != gt.shape or pred.shape[-1] != 4:
            raise L.YfsError(f"ciou_loss: expected two [..., 4] tensors, got {tuple(pred.shape)} and {tuple(gt.shape)}")
        p = pred.detach().to(torch.float32).contiguous()
        g = gt.detach().to(torch.float32).contiguous()
        n = p.numel() // 4
        lead = pred.shape[:-1]
        loss = torch.empty(lead, device=p.device, dtype=torch.float32)
        iou, ciou = torch.empty_like(loss), torch.empty_like(loss)
        grad = torch.empty_like(p) if ctx.needs_input_grad[0] else None
        L.check(L.lib().yfs_ciou_loss_f32(L.ptr(p), L.ptr(g), L.i64(n), L.ptr(loss), L.ptr(iou), L.ptr(ciou), L.ptr(grad),
                                          L.stream()), "yfs_ciou_loss_f32")
        ctx.grad, ctx.in_dtype = grad, pred.dtype
        ctx.mark_non_differentiable(iou, ciou)
        return loss, iou, ciou

    @staticmethod
    def backward(ctx, g_loss, g_iou, g_ciou):
        if ctx.grad is None:
            return None, None
        return (ctx.grad * g_loss.unsqueeze(-1)).to(ctx.in_dtype), None


def ciou_loss(pred_xywh, gt_xywh):
    """practice/yolo_toy_regressor_3_torch.py:121-138: -> (clamp(1 - ciou, 0), iou, ciou), each [...].  The loss is
    differentiable with respect to ``pred_xywh`` (alpha constant, as under the reference's no_grad); iou and ciou are
    returned for logging only."""
    return _CiouFn.apply(pred_xywh, gt_xywh)


class _SingleBoxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, targets):
        L.require_cuda(raw, targets)
        if raw.shape != targets.shape or raw.shape[-1] != 4:
            raise L.YfsError(f"yolo_single_box_loss: expected two [..., 4] tensors, got {tuple(raw.shape)} and "
                             f"{tuple(targets.shape)}")
        r = raw.detach().to(torch.float32).contiguous()
        t = targets.detach().to(torch.float32).contiguous()
        n = r.numel() // 4
        ws_bytes = L.lib().yfs_single_box_loss_workspace_bytes(L.i64(n))
        ws = torch.empty(max(ws_bytes // 4, 1), device=r.device, dtype=torch.float32)
        out = torch.empty(1, device=r.device, dtype=torch.float32)
        grad = torch.empty_like(r) if ctx.needs_input_grad[0] else None
        L.check(L.lib().yfs_single_box_loss_fwd_bwd_f32(L.ptr(r), L.ptr(t), L.i64(n), L.ptr(out), L.ptr(grad), L.ptr(ws),
                                                        L.c_size(ws_bytes), L.stream()), "yfs_single_box_loss_fwd_bwd_f32")
        ctx.grad, ctx.in_dtype = grad, raw.dtype
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        if ctx.grad is None:
            return None, None
        return (ctx.grad * g).to(ctx.in_dtype), None


def yolo_single_box_loss(raw, targets):
    """practice/yolo_toy_regressor_torch_20251015.py:368-383: raw [...,4] = (x logit, y logit, log w, log h), targets
    [...,4] xywh -> scalar 2 (sl1(x) + sl1(y)) + 4 (sl1(w) + sl1(h)), smooth-L1 means over the boxes."""
    return _SingleBoxFn.apply(raw, targets)
