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
