"""Dev check: the gathered Conv2d forward on k_gemm_f32pp against k_gemm_tiled, repeated (determinism / races), several shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvae_amd._lib import load
from mvae_amd.conv import _conv_nhwc, _convT_nhwc
dev = torch.device("cuda:0")
load().mvae_set_contraction_mode(0)
g = torch.Generator().manual_seed(5)
for B, Cc, IH, OC in ((48, 128, 8, 512), (256, 128, 8, 512), (16, 64, 16, 128), (48, 64, 16, 128), (8, 128, 8, 512), (24, 256, 8, 128)):
    src = torch.randn(B * IH * IH, Cc, generator=g).to(dev)
    Wt = (torch.randn(OC, 16 * Cc, generator=g) * 0.05).to(dev)
    bias = torch.randn(OC, generator=g).to(dev)
    load().mvae_set_forward_kernel(0)
    ref = _conv_nhwc(src, Wt, bias, None, B, Cc, IH, True)
    load().mvae_set_forward_kernel(1)
    bad = 0
    for it in range(10):
        y = _conv_nhwc(src, Wt, bias, None, B, Cc, IH, True)
        if not torch.equal(y, ref):
            bad += 1
            d = (y - ref).abs()
            idx = torch.nonzero(d > 0)
            rows = torch.unique(idx[:, 0])
            print(f"  it {it}: {idx.shape[0]} entries differ, max {float(d.max()):.3e}, rows {rows[:12].tolist()} ... cols {torch.unique(idx[:, 1])[:12].tolist()}")
    print(f"conv B={B} Cc={Cc} IH={IH} OC={OC} (M={B * IH * IH // 4}): {bad}/10 runs differ from k_gemm_tiled")
