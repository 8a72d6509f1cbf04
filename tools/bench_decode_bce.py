"""Dev tool: the fused decoder + BCE launch of the log-likelihood estimator (mvae_decode_bce_rows) alone, 64 000 rows."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvae_amd import functional as Fn
n, B, Z, H, D = 500, 128, int(sys.argv[1]) if len(sys.argv) > 1 else 6, 400, 784
dev = torch.device("cuda:0")
torch.manual_seed(0)
z = torch.randn(n, B, Z, device=dev)
w0, b0 = torch.randn(H, Z, device=dev) * 0.5, torch.randn(H, device=dev) * 0.1
wl, bl = torch.randn(D, H, device=dev) * 0.05, torch.randn(D, device=dev) * 0.1
x = (torch.rand(B, D, device=dev) > 0.7).float()
for _ in range(3): out = Fn.decode_bce_rows(z, w0, b0, wl, bl, x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):
    e0.record()
    for _ in range(10): out = Fn.decode_bce_rows(z, w0, b0, wl, bl, x)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10)
dt = sorted(ts)[2]
fl = 2.0 * n * B * (Z * H + H * D)
print(f"{os.environ.get('MVAE_HIP_LIB', 'default'):60s} {dt * 1e3:7.1f} us  {fl / dt / 1e9:6.1f} TFLOP/s  ({fl / dt / 1e9 / 157.3 * 100:.1f} % of the f32 MFMA peak)  sum {float(out.sum()):.6e}")
