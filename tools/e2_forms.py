import os, sys, torch
sys.path.insert(0, "/root/repo")
from mvae_amd import conv as C, functional as Fn
from mvae_amd._lib import load
dev = torch.device("cuda:0")
B = 256
a1 = torch.randn(B * 64, 128, device=dev); w = torch.randn(512, 2048, device=dev) * 0.02; b = torch.zeros(512, device=dev)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("e2 implicit (f32pp gathered)", t(lambda: C._conv_nhwc(a1, w, b, None, B, 128, 8, True)))
col = C._im2col(a1, None, B, 128, 8, C._nhwc(8, 128), True)
print("im2col_tm", t(lambda: C._im2col(a1, None, B, 128, 8, C._nhwc(8, 128), True)))
print("plain NT on the patch matrix", t(lambda: Fn.linear_forward(col, w, b, relu=True)))
load().mvae_set_forward_kernel(0)
print("e2 implicit (k_gemm_tiled)", t(lambda: C._conv_nhwc(a1, w, b, None, B, 128, 8, True)))
print("plain NT (k_gemm_tiled)", t(lambda: Fn.linear_forward(col, w, b, relu=True)))
