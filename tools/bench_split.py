"""Dev tool: the NT contractions (plain and gathered) with f32-input MFMA against split bf16 products: time and error vs
float64 (mvae_set_contraction_mode)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvae_amd import functional as Fn
from mvae_amd._lib import load
from mvae_amd.conv import _conv_nhwc
dev = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


g = torch.Generator().manual_seed(0)
for name, M, N, K in [("e2", 4096, 512, 2048), ("e1-like", 16384, 128, 1024), ("d2-like", 16384, 1024, 256), ("big", 16384, 512, 2048)]:
    x = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    ref = (x.double() @ W.double().t() + b.double())
    fl = 2.0 * M * N * K
    for mode in (0, 1):
        load().mvae_set_contraction_mode(mode)
        y = Fn.linear_forward(x, W, b)
        err = float((y.double() - ref).abs().max() / ref.abs().max())
        t = timeit(lambda: Fn.linear_forward(x, W, b))
        print(f"{name} M={M} N={N} K={K} mode {mode}: {t*1e6:7.1f} us {fl/t/1e12:6.1f} TF  max err / max|y| {err:.2e}")
# gathered: e1 forward (B = 256: 64 -> 128 channels on 32x32 -> 16x16 ... as in the step: src [B*16*16... 
B = 256
for name, Cc, IH, OC in [("e1 fwd", 64, 16, 128), ("db1", 64, 16, 256), ("dt0", 256, 8, 128)]:
    src = torch.randn(B * IH * IH, Cc, generator=g).to(dev)
    Wt = (torch.randn(OC, 16 * Cc, generator=g) * 0.05).to(dev)
    outs = []
    for mode in (0, 1):
        load().mvae_set_contraction_mode(mode)
        y = _conv_nhwc(src, Wt, None, None, B, Cc, IH, False)
        outs.append(y)
        t = timeit(lambda: _conv_nhwc(src, Wt, None, None, B, Cc, IH, False))
        fl = 2.0 * y.shape[0] * OC * 16 * Cc
        print(f"{name} mode {mode}: {t*1e6:7.1f} us {fl/t/1e12:6.1f} TF")
    print("   split vs f32-MFMA max diff / max", float((outs[0] - outs[1]).abs().max() / outs[0].abs().max()))
load().mvae_set_contraction_mode(0)
