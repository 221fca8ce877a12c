"""Probe: fc.1 forward y = x @ W^T (M=64, N=4096, K=50176): plain addmm vs manual split-K through bmm."""
import time
import torch
x = torch.randn(64, 50176, device="cuda", dtype=torch.bfloat16) * 0.1
w = torch.randn(4096, 50176, device="cuda", dtype=torch.bfloat16) * 0.01
b = torch.randn(4096, device="cuda", dtype=torch.bfloat16)
def t(fn, name):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20): fn()
    torch.cuda.synchronize(); print(f"{name:44s} {(time.time() - t0) / 20 * 1e3:.3f} ms")
ref = torch.addmm(b.float(), x.float(), w.float().t())
t(lambda: torch.addmm(b, x, w.t()), "addmm(b, x, w.t())")
for S in (8, 16, 32, 64):
    ks = 50176 // S
    def f(S=S, ks=ks):
        xa = x.view(64, S, ks).transpose(0, 1)                 # [S, 64, ks]
        wb = w.view(4096, S, ks).permute(1, 2, 0)              # [S, ks, 4096]
        return torch.bmm(xa, wb, out_dtype=torch.float32).sum(0) + b.float()
    try:
        t(f, f"split-K bmm S={S} (fp32 partials)")
        print("   max err vs fp32:", float((f() - ref).abs().max()), " plain:", float((torch.addmm(b, x, w.t()).float() - ref).abs().max()))
    except Exception as e:
        print("S", S, "failed:", str(e)[:200])
