#!/usr/bin/env python3
"""fc.1 (64 x 50176 -> 4096): materialised weight gradient + its share of sumsq + SGD, against the fused
wgrad+SGD kernel that never writes the gradient."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolo_from_scratch_amd import fc_ops as fo, _lib as L
import bench
dev = "cuda:0"
B, I, O = int(os.environ.get("FC_B", "64")), 50176, 4096   # FC_B=512: the 8-GPU step (factors of 8 ranks gathered)
x = torch.randn(B, I, device=dev).bfloat16(); dy = (torch.randn(B, O, device=dev) * 0.01).bfloat16()
p = torch.randn(O, I, device=dev) * 0.01; m = torch.zeros_like(p); g = torch.empty_like(p); p16 = p.bfloat16()
ss = torch.ones(1, device=dev)
n = p.numel()
wsb = L.lib().yfs_sumsq_workspace_bytes(L.i64(n)); ws = torch.empty(wsb // 4 + 1, device=dev)
def old():
    torch.mm(dy.t(), x, out_dtype=torch.float32, out=g)
    L.check(L.lib().yfs_sumsq_f32(L.ptr(g), L.i64(n), L.ptr(ss), L.ptr(ws), L.c_size(wsb), L.stream()), "sumsq")
    L.check(L.lib().yfs_sgd_step_f32(L.ptr(p), L.ptr(g), L.ptr(m), L.i64(n), L.ptr(ss), L.c_float(1.0), L.c_float(1e-3), L.c_float(0.9),
                                     L.c_float(5e-4), L.c_int(0), L.ptr(p16), L.stream()), "sgd")
def new():
    g1 = torch.mm(dy, dy.t(), out_dtype=torch.float32); g2 = torch.mm(x, x.t(), out_dtype=torch.float32)
    s = (g1 * g2).sum().reshape(1)
    fo.linear_wgrad_sgd(x, dy, p, m, p16, s, 1.0, 1e-3, 0.9, 5e-4, False)
print(f"B={B} materialised: {bench._time_ms(old, 10):.3f} ms | fused: {bench._time_ms(new, 10):.3f} ms")
