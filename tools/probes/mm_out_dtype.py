"""Probe: does torch.mm accept bf16 inputs with an fp32 result (out_dtype) on this ROCm build?"""
import time
import torch
a = torch.randn(1470, 64, device="cuda", dtype=torch.bfloat16)
b = torch.randn(64, 4096, device="cuda", dtype=torch.bfloat16)
ref = a.float() @ b.float()
try:
    c = torch.mm(a, b, out_dtype=torch.float32)
    print("out_dtype ok", c.dtype, float((c - ref).abs().max()))
    out = torch.empty(1470, 4096, device="cuda")
    try:
        torch.mm(a, b, out_dtype=torch.float32, out=out)
        print("out= ok", float((out - ref).abs().max()))
    except Exception as e:
        print("out= failed:", type(e).__name__, str(e)[:200])
    A = torch.randn(4096, 64, device="cuda", dtype=torch.bfloat16)
    B = torch.randn(64, 50176, device="cuda", dtype=torch.bfloat16)
    for fn, name in ((lambda: torch.mm(A, B), "bf16 out"), (lambda: torch.mm(A, B, out_dtype=torch.float32), "f32 out"),
                     (lambda: torch.mm(A, B).float(), "bf16 out + cast")):
        fn(); torch.cuda.synchronize(); t = time.time()
        for _ in range(10): fn()
        torch.cuda.synchronize(); print(name, (time.time() - t) * 100, "ms")
except Exception as e:
    print("out_dtype failed:", type(e).__name__, str(e)[:300])
