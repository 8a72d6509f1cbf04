"""The float64 latent chain (mvae_component_forward_f64 / _backward_f64; the reference CLI's default numerics, run.py:77,98-101):
every intermediate of a component's chain in double between float32 tensors.  Oracle = the CPU restatement evaluated in
float64 on the same float32 inputs (it is dtype-generic torch code); the float32 chain is run beside it to show what the
float64 chain buys at the warm-up radii."""
import numpy as np
import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _cpu(t):
    return t.detach().float().cpu().numpy()


def _oracle64(letter, d, heads, eps, R, wz, wkl):
    from oracle import model as M
    lvd = heads.shape[1] - d
    m = heads[:, :d].double().clone().requires_grad_(True)
    l = heads[:, d:].double().clone().requires_grad_(True)
    rp = None if letter == "e" else torch.tensor(float(R), dtype=torch.float64, requires_grad=True)
    o = M.component_forward(M.ComponentSpec(letter, d), m, l, eps.double(), rp)
    loss = (wz.double() * o.z).sum() + (wkl.double() * o.kl).sum()
    grads = torch.autograd.grad(loss, [m, l] + ([] if rp is None else [rp]))
    return o, grads


@pytest.mark.parametrize("letter,d,R", [("h", 2, 11.0), ("h", 6, 3.0), ("s", 2, 11.0), ("s", 5, 4.0), ("e", 3, 1.0),
                                        ("p", 3, 2.0), ("d", 3, 2.0)])
def test_float64_chain_vs_float64_oracle(dev, letter, d, R):
    """z, kl, d/d(mean head), d/d(logvar head), d/d(radius) of one component: the float64 chain against the oracle in float64
    at float32-OUTPUT precision (2e-6 of each tensor's scale), at radii of the warm-up schedule (R = 11 ... 3) where the
    float32 chain is visibly worse -- asserted for the hyperboloid and the sphere at R = 11."""
    from mvae_amd import functional as Fn
    lay = Fn.ComponentLayout([(letter, d)])
    g = torch.Generator().manual_seed(17 + d)
    B = 64
    heads = torch.randn(B, 2 * d, generator=g) * 0.5
    eps = torch.randn(B, d, generator=g)
    A = d + 1 if letter in ("h", "s") else d
    wz, wkl = torch.randn(B, A, generator=g), torch.rand(B, generator=g) + 0.5
    o, grads = _oracle64(letter, d, heads, eps, R, wz, wkl)
    radii = None if letter == "e" else torch.tensor([R], device=dev)
    res = {}
    for f64 in (True, False):
        with Fn.float64_chain(f64):
            out = Fn.component_forward(lay, heads.to(dev), eps.to(dev), radii)
            dheads, dr = Fn.component_backward(lay, heads.to(dev), eps.to(dev), radii, wz.to(dev), wkl.reshape(1, -1).to(dev),
                                               want_dradii=letter != "e")
        res[f64] = (out["z"], out["kl"][0], dheads[:, :d], dheads[:, d:], dr)
    z, kl, gm, gl, gr = res[True]
    tol = 2e-6
    assert_close(_cpu(z), o.z.detach().numpy(), tol, "z", atol_frac=tol)
    assert_close(_cpu(kl), o.kl.detach().numpy(), tol, "kl", atol_frac=tol)
    assert_close(_cpu(gm), grads[0].numpy(), tol, "d_mean", atol_frac=tol)
    assert_close(_cpu(gl), grads[1].numpy(), tol, "d_logvar", atol_frac=tol)
    if letter != "e":
        assert_close(float(gr[0]), float(grads[2]), 5e-6, "d_radius")
    if R >= 11.0:  # what float32 loses there: the KL of the same inputs
        ref = o.kl.detach().numpy()
        e64 = np.abs(_cpu(kl) - ref).max() / np.abs(ref).max()
        e32 = np.abs(_cpu(res[False][1]) - ref).max() / np.abs(ref).max()
        assert e32 > 20 * e64, f"float32 chain {e32:.2e} vs float64 chain {e64:.2e}"


def test_float64_chain_at_the_acos_boundary(dev):
    """The rows on which float32 rounds <mu_0, z> / R^2 to exactly 1 (test_sphere_backward_finite_at_the_acos_boundary: the
    reference's float32 backward is NaN there, the HIP float32 chain caps the derivative): in float64 the boundary is not
    hit -- finite everywhere and equal to the float64 oracle, no cap involved."""
    from mvae_amd import functional as Fn
    lay = Fn.ComponentLayout([("s", 2)])
    g = torch.Generator().manual_seed(5)
    B = 16
    heads = torch.randn(B, 4, generator=g) * 0.3
    eps = torch.randn(B, 2, generator=g)
    heads[:12, :2] *= 3e-4
    heads[:12, 2:] = -40.0
    R = 10.0
    wz, wkl = torch.randn(B, 3, generator=g), torch.rand(B, generator=g) + 0.5
    o, grads = _oracle64("s", 2, heads, eps, R, wz, wkl)
    assert all(torch.isfinite(x).all() for x in grads), "the float64 oracle itself is finite here"
    radii = torch.tensor([R], device=dev)
    with Fn.float64_chain():
        out = Fn.component_forward(lay, heads.to(dev), eps.to(dev), radii)
        dheads, dr = Fn.component_backward(lay, heads.to(dev), eps.to(dev), radii, wz.to(dev), wkl.reshape(1, -1).to(dev))
    assert torch.isfinite(dheads).all() and torch.isfinite(dr).all()
    assert_close(_cpu(out["kl"][0]), o.kl.detach().numpy(), 2e-6, "kl", atol_frac=2e-6)
    assert_close(_cpu(dheads[:, :2]), grads[0].numpy(), 1e-5, "d_mean", atol_frac=1e-5)
    assert_close(_cpu(dheads[:, 2:]), grads[1].numpy(), 1e-5, "d_logvar", atol_frac=1e-5)


def test_model_step_with_float64_chain(dev):
    """ModelVAE.train_step with float64_chain (run.py --doubles True): the reference's own sequence through the autograd
    operators, components in float64, dense layers float32 -- against the step recorded from the REFERENCE in float64 (g3,
    `f64` record of the small case) at the 1e-4 bar of the float32 path (the dense layers are float32), and against the fused
    float32 step of the same model."""
    from helpers import load_json, load_npz
    from mt.mvae import utils
    from mt.mvae.models import FeedForwardVAE, Trainer
    case = "h2s2e2_learn_ep12"
    meta = load_json("g3_step_small.json")[case]
    g = load_npz("g3_step_small.npz")
    prec = "f64" if any(k.startswith(f"{case}/f64/") for k in g) else "f32"
    k = f"{case}/{prec}/steps1/"
    T = lambda a, dt=torch.float32: torch.tensor(np.asarray(a), dtype=dt)  # noqa: E731

    class DS:
        in_dim = meta["in_dim"]

        def reconstruction_loss(self, x_, x):
            from mvae_amd import functional as Fn
            return Fn.bce_with_logits_rows(x_, x)

    def run(f64):
        state0 = {n[len(k + "state0/"):]: T(v) for n, v in g.items() if n.startswith(k + "state0/")}
        model = FeedForwardVAE(meta["h_dim"], utils.parse_components(meta["model"], meta["fixed_curvature"]), DS(),
                               meta["scalar_parametrization"])
        model.load_state_dict(state0)
        model.to(dev)
        model.float64_chain = f64
        trainer = Trainer(model, chkpt_dir="/tmp/mvae_test_chkpt_f64")
        trainer.epoch = meta["epoch"]
        opt = trainer.build_optimizer(learning_rate=1e-3, fixed_curvature=meta["fixed_curvature"])
        x = T(g[k + "x"])[0].to(dev)
        eps = T(g[k + "eps"])[0].to(dev)
        stats, _ = model.train_step(opt, x, beta=1.0, eps=eps)
        torch.cuda.synchronize()
        return model, float(stats.elbo)

    m64, elbo64 = run(True)
    m32, elbo32 = run(False)
    assert_close(elbo64, float(g[k + "stats"][0][2]), 1e-4, "elbo vs the reference record")
    assert_close(elbo64, elbo32, 1e-5, "elbo vs the fused float32 step")
    p64, p32 = dict(m64.named_parameters()), dict(m32.named_parameters())
    for name in p64:
        assert_close(_cpu(p64[name]), _cpu(p32[name]), 1e-4, "parameter after the step: " + name, atol_frac=1e-5)
    sums = m64.engine.read_stats()["last"]
    assert_close(sums["elbo"], elbo64, 1e-5, "the engine's running sums saw the step")
