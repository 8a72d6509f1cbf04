"""Dev tool: the four implicit (operand-gathering) conv contractions of the conv step at B = 256 against the plain NT
contraction of the same M x N x K on a materialised matrix."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvae_amd import functional as Fn
from mvae_amd.conv import _conv_nhwc
dev = torch.device("cuda:0")


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


B = 256
for name, Cc, IH, OC in [("e1 fwd", 64, 16, 128), ("e2 fwd", 128, 8, 512), ("d2 bwd-data", 64, 16, 256),
                         ("d1 bwd-data", 256, 8, 128)]:
    src = torch.randn(B * IH * IH, Cc, device=dev)
    Wt = torch.randn(OC, 16 * Cc, device=dev)
    bias = torch.randn(OC, device=dev)
    M, N, K = B * (IH // 2) ** 2, OC, 16 * Cc
    x = torch.randn(M, K, device=dev)
    fl = 2.0 * M * N * K
    a = timeit(lambda: _conv_nhwc(src, Wt, bias, None, B, Cc, IH, True))
    b = timeit(lambda: Fn.linear_forward(x, Wt, bias))
    print(f"{name}: M={M} N={N} K={K}  implicit {a*1e6:7.1f} us {fl/a/1e12:6.1f} TF | plain NT {b*1e6:7.1f} us {fl/b/1e12:6.1f} TF")

from mvae_amd.conv import _convT_nhwc, _gemm_nn, _col2im, _nhwc
print("transposed convolutions: implicit (4 parity classes) vs product + col2im")
for name, Cc, IH, OC in [("d1 fwd", 128, 4, 256), ("d2 fwd", 256, 8, 64), ("e2 bwd-data", 512, 4, 128),
                         ("e1 bwd-data", 128, 8, 64)]:
    src = torch.randn(B * IH * IH, Cc, device=dev)
    Wt = torch.randn(Cc, 16 * OC, device=dev)
    bias = torch.randn(OC, device=dev)
    fl = 2.0 * (B * IH * IH) * Cc * 16 * OC
    a = timeit(lambda: _convT_nhwc(src, Wt, bias, None, B, Cc, IH, OC, True))
    b = timeit(lambda: _col2im(_gemm_nn(src, Wt), bias, None, B, OC, 2 * IH, _nhwc(2 * IH, OC), True,
                               (B * 4 * IH * IH, OC), True))
    print(f"{name}: C={Cc} IH={IH} OC={OC}  implicit {a*1e6:7.1f} us {fl/a/1e12:6.1f} TF | product+col2im {b*1e6:7.1f} us")
