"""Dev probe: every output of mvae_conv_latent_forward / _backward against the generic operator sequence, one by one."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvae_amd import functional as Fn
from mvae_amd._lib import check, load, ptr, stream_ptr
from mvae_amd.conv import _permute_rc, _linear_splitk, _relu_mask_
from mvae_amd.functional import ComponentLayout

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
comps = [("h", 2), ("s", 2), ("e", 2)]
lay = ComponentLayout(comps, False)
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
beta = 0.7

def rel(a, b_):
    return float((a - b_).abs().max() / b_.abs().max().clamp_min(1e-30))

# generic
w_cl = _permute_rc(W.view(NH, 512, 16), NH, 512, 16).view(NH, 8192)
heads_g = _linear_splitk(a2, w_cl, b)
co = Fn.component_forward(lay, heads_g, eps, radii, want_kl=True)
d0o = Fn.linear_forward(co["z"], Wd, bd, relu=True)
t0_g = _permute_rc(d0o, B, 128, 16).view(B * 16, 128)
dd0 = _relu_mask_(_permute_rc(dt0, B, 16, 128).view(B, 2048).clone(), d0o)
dWd_g, dbd_g, dz_g = Fn.linear_backward(co["z"], Wd, dd0, relu_in=False, need_dx=True)
dheads_g, drad_g = Fn.component_backward(lay, heads_g, eps, radii, dz_g, None, beta)
dWcl, dbh_g, dh_g = Fn.linear_backward(a2, w_cl, dheads_g, relu_in=True, need_dx=True)
dW_g = _permute_rc(dWcl.view(NH, 16, 512), NH, 16, 512).view(NH, 8192)
# fused
ws = torch.empty(int(load().mvae_conv_latent_workspace_floats(B, n)), device=dev)
heads = torch.empty(B, NH, device=dev); z = torch.empty(B, Z, device=dev); kl = torch.empty(n, B, device=dev)
t0 = torch.empty(B * 16, 128, device=dev)
check(load().mvae_conv_latent_forward(lay.descs, n, ptr(a2), ptr(W), ptr(b), ptr(eps), lay.eps_dim, ptr(radii), ptr(Wd),
                                      ptr(bd), ptr(heads), ptr(z), ptr(kl), ptr(t0), None, 0, ptr(ws), B, stream_ptr(dev)))
torch.cuda.synchronize()
print("heads", rel(heads, heads_g), "z", rel(z, co["z"]), "kl", rel(kl, co["kl"]), "t0", rel(t0, t0_g),
      "mask flips", int(((t0 > 0) != (t0_g > 0)).sum()))
dW = torch.empty_like(W); dbh = torch.empty(NH, device=dev); da2 = torch.empty_like(a2)
dWd = torch.empty_like(Wd); dbd = torch.empty(2048, device=dev); drad = torch.empty(n, device=dev)
dheads = torch.empty(B, NH, device=dev)
# feed the GENERIC forward values so that only the backward kernels are compared
check(load().mvae_conv_latent_backward(lay.descs, n, ptr(a2), ptr(W), ptr(heads_g), ptr(eps), lay.eps_dim, ptr(radii),
                                       ptr(co["z"]), ptr(Wd), ptr(t0_g), ptr(dt0), 1, 0, beta, ptr(dW), ptr(dbh), ptr(da2),
                                       None, 0, None, None, ptr(dWd), ptr(dbd), ptr(drad), ptr(dheads), ptr(ws), B, stream_ptr(dev)))
torch.cuda.synchronize()
print("dheads", rel(dheads, dheads_g), "drad", rel(drad, drad_g), "dW_heads", rel(dW, dW_g), "db_heads", rel(dbh, dbh_g),
      "da2", rel(da2, dh_g), "dW_d0", rel(dWd, dWd_g), "db_d0", rel(dbd, dbd_g))
dd0_f = ws[:B * 2048].view(B, 16, 128).permute(0, 2, 1).reshape(B, 2048)
print("dd0", rel(dd0_f, dd0))
bad = ((dheads - dheads_g).abs() > 1e-5 * dheads_g.abs().max()).nonzero()
print("dheads bad entries", bad[:10].tolist(), len(bad))
bad = ((dWd - dWd_g).abs() > 1e-5 * dWd_g.abs().max()).nonzero()
print("dW_d0 bad entries", bad[:10].tolist(), len(bad))
