"""Dev probe: the latent launches of the conv step repeated back to back (warm instruction cache / L2) -- compare their
rocprofv3 averages with the averages inside the step (tools/bench_conv.py), where 16 other launches run in between."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvae_amd._lib import check, load, ptr, stream_ptr
from mvae_amd.functional import ComponentLayout

dev = torch.device("cuda:0")
B = 256
lay = ComponentLayout([("h", 2), ("s", 2), ("e", 2)], False)
NH, Z, n = lay.heads_dim, lay.z_dim, lay.n
g = torch.Generator().manual_seed(1)
a2 = torch.relu(torch.randn(B, 8192, generator=g)).to(dev)
W = (torch.randn(NH, 8192, generator=g) * 0.01).to(dev)
b = (torch.randn(NH, generator=g) * 0.1).to(dev)
eps = torch.randn(B, lay.eps_dim, generator=g).to(dev)
radii = torch.tensor([1.5, 2.0, 0.0]).to(dev)
Wd = (torch.randn(2048, Z, generator=g) * 0.3).to(dev)
bd = (torch.randn(2048, generator=g) * 0.1).to(dev)
dt0 = torch.randn(B * 16, 128, generator=g).to(dev)
ws = torch.empty(int(load().mvae_conv_latent_workspace_floats(B, n)), device=dev)
heads = torch.empty(B, NH, device=dev); z = torch.empty(B, Z, device=dev); kl = torch.empty(n, B, device=dev)
t0 = torch.empty(B * 16, 128, device=dev)
dW = torch.empty_like(W); dbh = torch.empty(NH, device=dev); da2 = torch.empty_like(a2)
dWd = torch.empty_like(Wd); dbd = torch.empty(2048, device=dev); drad = torch.empty(n, device=dev)
dheads = torch.empty(B, NH, device=dev)
trash = torch.empty(64 << 20, device=dev)  # 256 MB: MODE=cold overwrites it between launches (evicts L2 / MALL)
cold = os.environ.get("MODE") == "cold"
for it in range(30):
    check(load().mvae_conv_latent_forward(lay.descs, n, ptr(a2), ptr(W), ptr(b), ptr(eps), lay.eps_dim, ptr(radii), ptr(Wd),
                                          ptr(bd), ptr(heads), ptr(z), ptr(kl), ptr(t0), None, 0, ptr(ws), B, stream_ptr(dev)))
    if cold: trash.fill_(1.0)
    check(load().mvae_conv_latent_backward(lay.descs, n, ptr(a2), ptr(W), ptr(heads), ptr(eps), lay.eps_dim, ptr(radii),
                                           ptr(z), ptr(Wd), ptr(t0), ptr(dt0), 1, 0, 0.7, ptr(dW), ptr(dbh), ptr(da2),
                                           None, 0, None, None, ptr(dWd), ptr(dbd), ptr(drad), ptr(dheads), ptr(ws), B, stream_ptr(dev)))
    if cold: trash.fill_(1.0)
torch.cuda.synchronize()
print("done")
