"""Times the boundary-layer kernels of csrc/mvae_edge.hip against the patch-matrix route (B = 256):  python tools/bench_edge.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvae_amd import conv as C, functional as Fn  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
img = torch.rand(B, 3072, device=dev)
W = torch.randn(64, 48, device=dev) * 0.1
bias = torch.randn(64, device=dev)
mask = torch.randn(B * 256, 64, device=dev)
act = torch.randn(B * 256, 64, device=dev)
yp = C._new_planes(B * 256, 64, dev)
out = torch.empty(64, 48, device=dev)
col = C._im2col(img, None, B, 3, 32, C._nchw(32, 3))


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    C._DEFERRED_WS.clear()
    return e0.elapsed_time(e1) / n * 1e3


rows = [
    ("edge fwd  bias+relu, planes", lambda: C._edge_conv(img, W, bias, None, True, B, yp)),
    ("edge fwd  bias+relu, no planes", lambda: C._edge_conv(img, W, bias, None, True, B)),
    ("edge bwd  mask, planes", lambda: C._edge_conv(img, W, None, mask, False, B, yp)),
    ("edge bwd  mask, no planes", lambda: C._edge_conv(img, W, None, mask, False, B)),
    ("edge wgrad", lambda: C._edge_wgrad(act, img, out, B)),
    ("im2col", lambda: C._im2col(img, None, B, 3, 32, C._nchw(32, 3))),
    ("patch fwd, planes", lambda: C._linear_forward_planes(col, W, bias, True, yp)),
    ("patch fwd, no planes", lambda: Fn.linear_forward(col, W, bias, relu=True)),
    ("patch bwd mask, planes", lambda: C._linear_masked(col, W, mask, planes=yp)),
    ("patch wgrad", lambda: C._gemm_tn(act, col, out=out)),
]
for name, fn in rows:
    print(f"{name:34s} {timeit(fn):8.2f} us")
