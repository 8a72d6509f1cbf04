"""Dev tool: ONE backward contraction of the conv step at B = 256 on pre-split planes, repeated (for rocprofv3 --pmc runs):
   python tools/p3_one.py e1f|e2f|d1f|d2f|db1|dt0|da1|da0|dWd2|dWd1|dWe2|dWe1 [b3]     (b3: the in-kernel-split kernel instead)
   python tools/p3_one.py pd2|pd1|pe2|pe1     a layer's weight gradient + backward-data as the ONE launch the step issues"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvae_amd._lib import load
from mvae_amd import conv as Cv
dev = torch.device("cuda:0")
B = 256
op = sys.argv[1] if len(sys.argv) > 1 else "db1"
b3 = len(sys.argv) > 2 and sys.argv[2] == "b3"
load().mvae_set_contraction_mode(1 if b3 else 2)
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
P = lambda t: Cv._split_planes([t])[0]  # noqa: E731
if op in ("e1f", "e2f"):  # forward Conv2d layers on the exact f32 MFMA (implicit contraction)
    load().mvae_set_contraction_mode(2)
    Cc, IH, OC = (64, 16, 128) if op == "e1f" else (128, 8, 512)
    src, Wt, bias = rnd(B * IH * IH, Cc), rnd(OC, 16 * Cc) * 0.05, rnd(OC)
    fn = lambda: Cv._conv_nhwc(src, Wt, bias, None, B, Cc, IH, True)  # noqa: E731
elif op in ("d1f", "d2f"):  # forward ConvTranspose2d layers (four parity classes)
    load().mvae_set_contraction_mode(2)
    Cc, IH, OC = (128, 4, 256) if op == "d1f" else (256, 8, 64)
    src, Wt, bias = rnd(B * IH * IH, Cc), rnd(Cc, 16 * OC) * 0.05, rnd(OC)
    fn = lambda: Cv._convT_nhwc(src, Wt, bias, None, B, Cc, IH, OC, True)  # noqa: E731
elif op in ("db1", "dt0"):
    Cc, IH, OC, masked = (64, 16, 256, True) if op == "db1" else (256, 8, 128, False)
    src, Wt = rnd(B * IH * IH, Cc), rnd(OC, 16 * Cc) * 0.05
    mask = rnd(B * (IH // 2) ** 2, OC) if masked else None
    sp, wp = P(src), P(Wt)
    fn = (lambda: Cv._conv_nhwc(src, Wt, None, mask, B, Cc, IH, False, Cv.BACKWARD)) if b3 else \
         (lambda: Cv._conv_nhwc_p3(sp, wp, mask, B, Cc, IH, want_planes=masked))
elif op == "da1":
    x, Wn = rnd(B * 16, 512), rnd(512, 2048) * 0.05
    xp, wp = P(x), P(Wn)
    fn = (lambda: Cv._gemm_nn(x, Wn, Cv.BACKWARD)) if b3 else (lambda: Cv._gemm_nn_p3(xp, wp))
elif op == "da0":
    src, Wt, mask = rnd(B * 64, 128), rnd(128, 16 * 64) * 0.05, rnd(B * 256, 64)
    sp, wp = P(src), P(Wt)
    fn = (lambda: Cv._convT_nhwc(src, Wt, None, mask, B, 128, 8, 64, False, Cv.BACKWARD)) if b3 else \
         (lambda: Cv._convT_nhwc_p3(sp, wp, mask, B, 128, 8, 64))
elif op in ("pd2", "pd1", "pe2", "pe1"):  # the pairs of ConvEngine._backward_body_p3 (same shapes, slicing and epilogues)
    import torch as _t
    if op in ("pd2", "pd1"):  # wgrad of d2 / d1 + backward-data of the transposed convolution (a Conv2d-shaped contraction)
        Cc, IH, OC, masked = (64, 16, 256, True) if op == "pd2" else (256, 8, 128, False)
        dy, Wt, act = rnd(B * IH * IH, Cc), rnd(OC, 16 * Cc) * 0.05, rnd(B * (IH // 2) ** 2, OC)
        dyp, wp, actp = P(dy), P(Wt), P(act)
        out = _t.empty(OC * 16 * Cc, device=dev)
        cs = _t.empty(OC, device=dev)
        def fn():
            with Cv._p3_group(dev):
                Cv._conv_nhwc_wgrad_p3(actp, dyp, out, B, Cc, IH)
                if masked:
                    Cv._conv_nhwc_p3(dyp, wp, act, B, Cc, IH, want_planes=True, colsum_out=cs)
                else:
                    Cv._conv_nhwc_p3(dyp, wp, None, B, Cc, IH, keep_slices=True)
    else:  # wgrad of e2 / e1 + backward-data of the convolution (a transposed-convolution-shaped contraction)
        Cc, IH, OC = (512, 4, 128) if op == "pe2" else (128, 8, 64)
        dy, Wt, act = rnd(B * IH * IH, Cc), rnd(Cc, 16 * OC) * 0.05, rnd(B * 4 * IH * IH, OC)
        dyp, wp, actp = P(dy), P(Wt), P(act)
        out = _t.empty(Cc * 16 * OC, device=dev)
        cs = _t.empty(OC, device=dev)
        def fn():
            with Cv._p3_group(dev):
                Cv._conv_nhwc_wgrad_p3(dyp, actp, out, B, OC, 2 * IH)
                if op == "pe2":
                    Cv._convT_nhwc_p3(dyp, wp, act, B, Cc, IH, OC, want_planes=True, colsum_out=cs, want_y=False)
                else:
                    Cv._convT_nhwc_p3(dyp, wp, act, B, Cc, IH, OC, colsum_out=cs)
else:
    Cc, IH, OC = {"dWd2": (64, 16, 256), "dWd1": (256, 8, 128), "dWe2": (128, 8, 512), "dWe1": (64, 16, 128)}[op]
    dy, src = rnd(B * (IH // 2) ** 2, OC), rnd(B * IH * IH, Cc)
    out = torch.empty(OC, 16 * Cc, device=dev)
    dp, sp = P(dy), P(src)
    fn = (lambda: Cv._conv_nhwc_wgrad(dy, src, out, B, Cc, IH)) if b3 else (lambda: Cv._conv_nhwc_wgrad_p3(dp, sp, out, B, Cc, IH))
from mvae_amd._lib import check, stream_ptr
for _ in range(20):
    if op[0] == "p":  # as in the step: slice sums and column sums are queued and flushed once
        check(load().mvae_slice_sums_defer(1))
    fn()
    if op[0] == "p":
        check(load().mvae_slice_sums_flush(stream_ptr(dev)))
        check(load().mvae_slice_sums_defer(0))
    Cv._DEFERRED_WS.clear()
torch.cuda.synchronize()
