"""The nine backward contractions of the conv step at B = 256 (BASELINE configs[4]): us per call on the f32-input MFMA (mode 0),
on split products with in-kernel splitting (k_gemm_b3, mode 1) and on pre-split planes (k_gemm_p3); TFLOP/s-equivalent of each."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvae_amd._lib import load
from mvae_amd import conv as Cv

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    Cv._DEFERRED_WS.clear()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    Cv._DEFERRED_WS.clear()
    return e0.elapsed_time(e1) / n * 1e3


def modes(fn):
    out = []
    for m in (0, 1):
        load().mvae_set_contraction_mode(m)
        out.append(timeit(fn))
    load().mvae_set_contraction_mode(2)
    return out


rows = []
P = lambda t: Cv._split_planes([t])[0]  # noqa: E731
# backward-data of d2 (db1), of d1 (dt0): gathered conv
for name, Cc, IH, OC, masked in (("db1 = bwd-data d2", 64, 16, 256, True), ("dt0 = bwd-data d1", 256, 8, 128, False)):
    src, Wt = rnd(B * IH * IH, Cc), rnd(OC, 16 * Cc) * 0.05
    mask = rnd(B * (IH // 2) ** 2, OC) if masked else None
    sp, wp = P(src), P(Wt)
    f = modes(lambda: Cv._conv_nhwc(src, Wt, None, mask, B, Cc, IH, False, Cv.BACKWARD))
    p = timeit(lambda: Cv._conv_nhwc_p3(sp, wp, mask, B, Cc, IH, want_planes=masked))
    rows.append((name, 2.0 * B * (IH // 2) ** 2 * OC * 16 * Cc, f[0], f[1], p))
# backward-data of e2: NN product (+ col2im, not timed)
x, Wn = rnd(B * 16, 512), rnd(512, 2048) * 0.05
xp, wp = P(x), P(Wn)
f = modes(lambda: Cv._gemm_nn(x, Wn, Cv.BACKWARD))
rows.append(("da1 product = bwd-data e2", 2.0 * B * 16 * 512 * 2048, f[0], f[1], timeit(lambda: Cv._gemm_nn_p3(xp, wp))))
# backward-data of e1: transposed conv per parity class
src, Wt, mask = rnd(B * 64, 128), rnd(128, 16 * 64) * 0.05, rnd(B * 256, 64)
sp, wp = P(src), P(Wt)
f = modes(lambda: Cv._convT_nhwc(src, Wt, None, mask, B, 128, 8, 64, False, Cv.BACKWARD))
rows.append(("da0 = bwd-data e1", 2.0 * B * 256 * 64 * 4 * 128, f[0], f[1], timeit(lambda: Cv._convT_nhwc_p3(sp, wp, mask, B, 128, 8, 64))))
# weight gradients
for name, Cc, IH, OC in (("dW d2", 64, 16, 256), ("dW d1", 256, 8, 128), ("dW e2", 128, 8, 512), ("dW e1", 64, 16, 128)):
    dy, src = rnd(B * (IH // 2) ** 2, OC), rnd(B * IH * IH, Cc)
    out = torch.empty(OC, 16 * Cc, device=dev)
    dp, sp = P(dy), P(src)

    def f32():
        Cv._conv_nhwc_wgrad(dy, src, out, B, Cc, IH)

    def p3():
        Cv._conv_nhwc_wgrad_p3(dp, sp, out, B, Cc, IH)
    f = modes(f32)
    rows.append((name, 2.0 * B * (IH // 2) ** 2 * OC * 16 * Cc, f[0], f[1], timeit(p3)))
tot = [0.0, 0.0, 0.0]
print(f"{'contraction':28s} {'GFLOP':>7s} | {'f32 MFMA us':>11s} {'TF':>6s} | {'b3 us':>8s} {'TF':>6s} | {'p3 us':>8s} {'TF':>6s}")
for name, fl, a, b, c in rows:
    print(f"{name:28s} {fl / 1e9:7.2f} | {a:11.1f} {fl / a / 1e6:6.1f} | {b:8.1f} {fl / b / 1e6:6.1f} | {c:8.1f} {fl / c / 1e6:6.1f}")
    tot[0] += a; tot[1] += b; tot[2] += c
print(f"{'sum':28s} {sum(r[1] for r in rows) / 1e9:7.2f} | {tot[0]:11.1f}        | {tot[1]:8.1f}        | {tot[2]:8.1f}")
# the split kernel itself: planes of the step's activations and weights
ts = [rnd(B * 256, 64), rnd(B * 64, 128), rnd(B * 64, 256), rnd(B * 16, 128)]
print("planes of a0, a1, b1, t0 in one launch: %.1f us" % timeit(lambda: Cv._split_planes(ts)))
ws = [rnd(128, 1024), rnd(512, 2048), rnd(128, 4096), rnd(256, 1024)]
print("planes of the four conv weights in one launch: %.1f us" % timeit(lambda: Cv._split_planes(ws)))
