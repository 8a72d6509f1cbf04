"""Dev tool: time of ModelVAE.log_likelihood (IWAE, vae.py:82-123) at the reference's eval setting: B=128, n=500."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvae_amd import utils
from mvae_amd.models import FeedForwardVAE


class _DS:
    in_dim = 784
    img_dims = None

    def reconstruction_loss(self, x_, x):
        from mvae_amd import functional as Fn
        return Fn.bce_rows(x_, x)


model = sys.argv[1] if len(sys.argv) > 1 else "h2,s2,e2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 500
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = FeedForwardVAE(400, utils.parse_components(model, False), _DS(), False).to(dev)
x = (torch.rand(128, 784, device=dev) > 0.7).float()
for _ in range(3):
    out = m.log_likelihood(x, n=n)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    out = m.log_likelihood(x, n=n)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 50
print(f"log_likelihood {model} B=128 n={n}: {dt * 1e3:.4f} ms  (log_px mean {float(out[0].mean()):.3f})")
