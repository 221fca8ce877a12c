#!/usr/bin/env python3
"""What does the vendor GEMM (hipBLASLt via torch.mm) reach on the GEMM-equivalent shapes of the conv layers?
Gives a practical MFMA+memory ceiling to read the implicit-GEMM numbers of tools/conv_bench.py against
(a dense GEMM has none of the im2col addressing, halo or padding work)."""
import torch

# (M, N, K) = (batch*OH*OW, Cout, KH*KW*Cin) at batch 64
SHAPES = {"L2 3x3 192->256 @112": (802816, 256, 1728), "L4 1x1 256->128 @56": (200704, 128, 256),
          "L7 3x3 256->512 @56": (200704, 512, 2304), "L9 1x1 512->256 @28": (50176, 256, 512),
          "L10 3x3 256->512 @28": (50176, 512, 2304), "L18 3x3 512->1024 @28": (50176, 1024, 4608),
          "L21 3x3 512->1024 @14": (12544, 1024, 4608), "L24 3x3 1024->1024 @14": (12544, 1024, 9216),
          "L26 3x3 1024->1024 @7": (3136, 1024, 9216)}


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = "cuda:0"
    for name, (m, n, k) in SHAPES.items():
        a = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
        bt = torch.randn(n, k, device=dev, dtype=torch.bfloat16)      # weights [Cout, K] as the conv stores them
        b = bt.t().contiguous()
        dz = torch.randn(m, n, device=dev, dtype=torch.bfloat16)
        fl = 2.0 * m * n * k
        us_nt = timed(lambda: torch.mm(a, bt.t()))                     # forward shape
        us_nn = timed(lambda: torch.mm(a, b))
        us_dg = timed(lambda: torch.mm(dz, bt))                        # dgrad shape  [M,N]x[N,K]
        us_wg = timed(lambda: torch.mm(dz.t(), a))                     # wgrad shape  [N,M]x[M,K]
        print(f"{name:26s} GF={fl / 1e9:7.1f} | fwd(NT) {us_nt:8.1f} us {fl / us_nt / 1e6:7.1f} TF | fwd(NN) "
              f"{fl / us_nn / 1e6:7.1f} TF | dgrad {fl / us_dg / 1e6:7.1f} TF | wgrad {fl / us_wg / 1e6:7.1f} TF", flush=True)
        del a, b, bt, dz


if __name__ == "__main__":
    main()
