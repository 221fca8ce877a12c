"""Probe: which formulation of the fc.1 input gradient (dx = dy @ W, M=64, K=4096, N=50176) hipBLASLt runs fastest."""
import time
import torch
dy = torch.randn(64, 4096, device="cuda", dtype=torch.bfloat16)
w = torch.randn(4096, 50176, device="cuda", dtype=torch.bfloat16)
wt = w.t().contiguous()
def t(fn, name):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20): fn()
    torch.cuda.synchronize(); print(f"{name:40s} {(time.time() - t0) / 20 * 1e3:.3f} ms")
ref = torch.mm(dy, w)
t(lambda: torch.mm(dy, w), "mm(dy, w)")
t(lambda: torch.mm(w.t(), dy.t()), "mm(w.t(), dy.t())  -> dx^T")
t(lambda: torch.matmul(dy.unsqueeze(0), w.unsqueeze(0)), "bmm form")
t(lambda: torch.nn.functional.linear(dy, wt), "linear(dy, w^T contiguous)")
out = torch.empty(64, 50176, device="cuda", dtype=torch.bfloat16)
t(lambda: torch.mm(dy, w, out=out), "mm(dy, w, out=)")
print(float((torch.mm(w.t(), dy.t()).t().float() - ref.float()).abs().max()))
