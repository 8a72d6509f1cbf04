"""GPU tests of the host-side mirror: the reference's class API (Manifold / Component / ModelVAE / Trainer) driving the
HIP kernels, against the oracle and the golden vectors."""
import os

import numpy as np
import pytest
import torch

from helpers import T, assert_close, assert_close_after_adam, load_npz

pytestmark = pytest.mark.gpu
RTOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


class _DS:
    in_dim = 784

    def __init__(self, in_dim=784):
        self.in_dim = in_dim

    def reconstruction_loss(self, x_, x):
        from mvae_amd import functional as Fn
        return Fn.bce_rows(x_, x)


def _cpu(t):
    return t.detach().cpu().numpy()


def test_manifold_classes_vs_golden(dev):
    from mvae_amd.ops import Euclidean, Hyperboloid, Sphere
    g = load_npz("g1_primitives.npz")
    R = torch.nn.Parameter(torch.tensor(2.0, device=dev))
    for name, man in [("H", Hyperboloid(lambda: R)), ("S", Sphere(lambda: R)), ("E", Euclidean())]:
        k = f"{name}/R2/d5/f32/"
        x, v = T(g[k + "x"]).to(dev), T(g[k + "v"]).to(dev)
        with torch.no_grad():
            mu = man.exp_map_mu0(x)
            z, (u, v_) = man.sample_projection_mu0(v, at_point=mu)
        assert_close(_cpu(mu), g[k + "mu"], RTOL, name + " mu")
        assert_close(_cpu(z), g[k + "z"], RTOL, name + " z")
        if name != "E":
            assert_close(_cpu(man.logdet(mu, None, z, (u, v_))), g[k + "logdet_u"], RTOL, name + " logdet", atol_frac=1e-4)
            assert float(man.radius) == 2.0
    assert float(Hyperboloid(lambda: R).curvature) == -0.25 and float(Sphere(lambda: R).curvature) == 0.25
    # every Manifold method is differentiable (round 1 raised here): gradient flows to the argument and to the radius
    xg = torch.full((2, 2), 0.3, device=dev, requires_grad=True)
    Hyperboloid(lambda: R).exp_map_mu0(xg).sum().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all() and R.grad is not None


def test_component_api(dev):
    """Component.forward -> q.rsample_with_parts -> kl_loss, the reference's call sequence (vae.py:73-76,137)."""
    from mvae_amd import utils
    from oracle import model as M
    comp = utils.parse_components("h3", fixed_curvature=False)[0]
    comp.init_layers(16, scalar_parametrization=False)
    comp.to(dev)
    comp._nradius.data.fill_(2.0)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(8, 16, generator=g)
    eps = torch.randn(8, 3, generator=g)
    q_z, p_z, (loc, scale) = comp(x.to(dev))
    z, data = q_z.rsample_with_parts(eps=eps.to(dev))
    kl = comp.kl_loss(q_z, p_z, z, data)
    mean_raw = torch.nn.functional.linear(x, comp.fc_mean.weight.cpu(), comp.fc_mean.bias.cpu())
    lv_raw = torch.nn.functional.linear(x, comp.fc_logvar.weight.cpu(), comp.fc_logvar.bias.cpu())
    o = M.component_forward(M.ComponentSpec("h", 3), mean_raw, lv_raw, eps, torch.tensor(2.0))
    assert_close(_cpu(z), o.z.detach().numpy(), RTOL, "z")
    assert_close(_cpu(kl), o.kl.detach().numpy(), RTOL, "kl", atol_frac=1e-4)
    assert_close(_cpu(loc), o.mu.detach().numpy(), RTOL, "loc")
    assert_close(_cpu(scale), o.std.detach().numpy(), RTOL, "scale")
    assert tuple(p_z.loc.shape) == (8, 4) and float(p_z.loc[0, 0]) == 2.0


def _model(dev, model_str, in_dim, h_dim, fixed=False, scalar=False):
    from mvae_amd import synthetic, utils
    from mvae_amd.models import FeedForwardVAE
    from oracle import model as M
    spec = M.Spec(model_str, in_dim=in_dim, h_dim=h_dim, fixed_curvature=fixed, scalar_parametrization=scalar)
    state0 = synthetic.synthetic_state(spec.named_shapes(), radius=2.0)
    m = FeedForwardVAE(h_dim, utils.parse_components(model_str, fixed), _DS(in_dim), scalar)
    m.load_state_dict(state0)
    m.to(dev)
    return m, spec, state0


def test_model_train_step_vs_oracle(dev):
    """ModelVAE.train_step through the class API (parameters aliased into the flat HBM buffer) == the oracle."""
    from mvae_amd import synthetic
    from mvae_amd.trainer import Trainer
    from oracle import model as M
    m, spec, state0 = _model(dev, "h2,s2,e2", 784, 400)
    assert list(m.state_dict().keys()) == [n for n, _ in spec.named_shapes()]
    trainer = Trainer(m, chkpt_dir="/tmp/mvae_test_chkpt")
    trainer.epoch = 12
    opt = trainer.build_optimizer(learning_rate=1e-3, fixed_curvature=False)
    orc = M.StepOracle(spec, state0)
    xs = synthetic.binary_batches(3, 128, 784)
    eps = synthetic.eps_batches(3, 128, 6)
    for s in range(3):
        ref = orc.train_step(xs[s], eps[s], 1.0, epoch=12)
        stats, _ = m.train_step(opt, xs[s], beta=1.0, eps=eps[s].to(dev))
        assert_close(stats.elbo, float(ref.elbo.detach()), RTOL, f"elbo step {s}")
        assert_close(stats.component_kl, ref.kl.sum(1).detach().numpy(), RTOL, "component kl", atol_frac=1e-4)
    for n, p in m.named_parameters():
        assert p.data_ptr() >= m.engine.params.data_ptr()  # still a view of the flat buffer
        assert_close_after_adam(_cpu(p), orc.P[n].detach().numpy(), 1e-3, 3, "param " + n)


def test_model_forward_and_batch_stats(dev):
    from mvae_amd import synthetic
    from oracle import model as M
    m, spec, state0 = _model(dev, "h2,s2,e2,p2", 784, 64)
    x = synthetic.binary_batches(1, 32, 784)[0]
    eps = synthetic.eps_batches(1, 32, 8)[0]
    P = {k: v.clone() for k, v in state0.items()}
    ref = M.forward(spec, P, x, eps, beta=0.5)
    with torch.no_grad():
        reps, concat_z, x_ = m(x.to(dev), eps=eps.to(dev))
        bs = m.compute_batch_stats(x.to(dev), x_, reps, beta=0.5)
    assert_close(_cpu(x_), ref.logits.numpy(), RTOL, "logits")
    assert_close(_cpu(concat_z), ref.concat_z.numpy(), RTOL, "concat_z")
    assert_close(float(bs.elbo), float(ref.elbo), RTOL, "elbo")
    assert_close(float(bs.bce), float(ref.bce.sum()), RTOL, "bce")
    for i, r in enumerate(reps):
        assert_close(_cpu(r.q_z.loc), ref.comps[i].mu.numpy(), RTOL, f"loc {i}")
        assert_close(_cpu(m.components[i].kl_loss(r.q_z, r.p_z, r.z, r.data)), ref.kl[i].numpy(), RTOL, f"kl {i}",
                     atol_frac=1e-4)


@pytest.mark.parametrize("name,model", [("h2s2e2", "h2,s2,e2"), ("e6", "e6"), ("h5s3e4", "h5,s3,e4")])
def test_log_likelihood_vs_golden(dev, name, model):
    """ModelVAE.log_likelihood (vae.py:82-123) against the reference's own output (n=8 importance samples)."""
    g = load_npz("g4_loglik.npz")
    m, spec, _ = _model(dev, model, 32, 16)
    key = f"{name}/f32/"
    with torch.no_grad():
        lp, mi, cn = m.log_likelihood(T(g[key + "x"], torch.float32).to(dev), n=8, eps=T(g[key + "eps"]).to(dev))
    assert_close(_cpu(lp), g[key + "log_px"], RTOL, "log_px")
    assert_close(_cpu(mi), g[key + "mi"], RTOL, "mi", atol_frac=1e-4)
    assert_close(float(cn), float(g[key + "cov_norm"]), 5 * RTOL, "cov_norm")


def test_log_likelihood_full_size_properties(dev):
    """n=500, B=128 (the reference's eval setting): finite, log p(x) <= 0 for binary data, IWAE bound >= ELBO-ish,
    and independent of how the sample dim is chunked."""
    from mvae_amd import synthetic
    m, spec, _ = _model(dev, "h2,s2,e2", 784, 400)
    x = synthetic.binary_batches(1, 128, 784)[0].to(dev)
    eps = torch.randn(500, 128, 6, generator=torch.Generator().manual_seed(5)).to(dev)
    with torch.no_grad():
        lp, mi, cn = m.log_likelihood(x, n=500, eps=eps)
        lp_a, _, _ = m.log_likelihood(x, n=250, eps=eps[:250])
        lp_b, _, _ = m.log_likelihood(x, n=250, eps=eps[250:])
    assert torch.isfinite(lp).all() and torch.isfinite(mi).all() and torch.isfinite(cn)
    assert (lp <= 0).all()
    merged = torch.logsumexp(torch.stack([lp_a, lp_b]), dim=0) - np.log(2.0)
    assert float((merged - lp).abs().max()) < 1e-3 * float(lp.abs().max())


def test_trainer_epochs_and_checkpoints(dev, tmp_path):
    """Two tiny epochs through Trainer.train_stopping-style calls: finite stats, warm-up radii (train.py:189-194:
    K = +-1/(11-epoch)^2 even for fixed curvature, reference test_vae.py:296-300), reference-compatible checkpoint."""
    from mvae_amd import utils
    from mvae_amd.data import DeviceLoader
    from mvae_amd.models import FeedForwardVAE
    from mvae_amd.trainer import Trainer
    from mvae_amd import synthetic
    # 300 images at batch 64: four graph-replayed device-pipeline batches + a ragged tail of 44 per epoch
    x = (synthetic.digits_like_batches(5, 64).reshape(-1, 784) * 255).to(torch.uint8).to(dev)[:300]
    y = torch.zeros(300, dtype=torch.int64, device=dev)
    train = DeviceLoader(x, y, 64, train=True, binarize=True, seed=1)
    test = DeviceLoader(x[:64], y[:64], 64, train=False, binarize=True)
    torch.manual_seed(0)
    m = FeedForwardVAE(32, utils.parse_components("h2,s2,e2", True), _DS(), False).to(dev)
    m.seed_sampler(3)
    tr = Trainer(m, chkpt_dir=str(tmp_path))
    opt = tr.build_optimizer(1e-3, fixed_curvature=True)
    res = tr.train_epochs(opt, train, test, betas=None, epochs=2, likelihood_n=4)
    assert tr.global_step == 2 * 5 and int(m.engine.counters[0]) == 10  # 4 pipeline batches + 1 tail batch per epoch
    st = res[1]
    assert np.isfinite([st.bce, st.kl, st.elbo, st.log_likelihood]).all()
    assert st.log_likelihood < 10 and st.log_likelihood > -1e4  # reference test_vae.py:285-304 bounds
    assert abs(float(m.components[0].manifold.curvature) + 0.01) < 1e-6  # R = 11 - 1 = 10 after 2 epochs
    assert abs(float(m.components[1].manifold.curvature) - 0.01) < 1e-6
    sd = torch.load(os.path.join(str(tmp_path), "2.chkpt"))
    assert list(sd.keys()) == list(m.state_dict().keys())
    m2 = FeedForwardVAE(32, utils.parse_components("h2,s2,e2", True), _DS(), False)
    m2.load_state_dict(sd)
    m2.to(dev)
    for (n, a), (_, b) in zip(m.named_parameters(), m2.named_parameters()):
        assert torch.equal(a, b), n


def test_training_improves_elbo(dev):
    """A few hundred steps on structured synthetic images: the ELBO per sample goes up substantially."""
    from mvae_amd import synthetic
    from mvae_amd.engine import StepEngine
    eng = StepEngine([("h", 2), ("s", 2), ("e", 2)], 784, 400, dev, radius_trainable=[True, True, False])
    eng.load_state(synthetic.synthetic_state([(n, s) for n, _, s in eng.flat.entries], radius=2.0))
    xs = synthetic.digits_like_batches(50, 128).to(dev)
    eps = synthetic.eps_batches(50, 128, 6).to(dev)
    first = None
    for it in range(400):
        eng.train_step(xs[it % 50], eps[it % 50], 1.0, it >= 100)
        if it == 0:
            first = eng.read_stats()["last"]["elbo"] / 128
    last = eng.read_stats()["last"]["elbo"] / 128
    assert np.isfinite(last) and last > first + 100, (first, last)


# the wrapped-normal / Euclidean entries of the reference's model table (tests/mvae/models/test_vae.py:200-250)
_REF_MODELS = ["e2", "s2-wn", "h2-wn", "p2-wn", "d2-wn", "h2,s2,e2", "d2-wn,e2,p2-wn", "h40-wn", "s40-wn", "d40-wn",
               "p40-wn", "u2"]


@pytest.mark.parametrize("fixed_curvature", [True, False])
@pytest.mark.parametrize("model", _REF_MODELS)
def test_run_training_like_the_reference(dev, tmp_path, model, fixed_curvature):
    """tests/mvae/models/test_vae.py:255-335 restated: h_dim = 2, two epochs on a tiny fake data set, importance-sampled
    log-likelihood inside (-1e4, eps), and the curvature expectations after the run -- +-1/(11-1)^2 for fixed curvature
    (the warm-up override applies to fixed models too), not -1 / 0 / 1 for learnable ones."""
    from mvae_amd import utils
    from mvae_amd.components import (EuclideanComponent, HyperbolicComponent, PoincareComponent, SphericalComponent,
                                     StereographicallyProjectedSphereComponent, UniversalComponent)
    from mvae_amd.data import DeviceLoader
    from mvae_amd.models import FeedForwardVAE
    from mvae_amd.trainer import Trainer
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(6, 784, generator=g) * 255).to(torch.uint8).to(dev)
    y = torch.zeros(6, dtype=torch.int64, device=dev)
    train = DeviceLoader(x, y, 2, train=True, binarize=True, seed=1)
    test = DeviceLoader(x[:2], y[:2], 2, train=False, binarize=True)
    torch.manual_seed(0)
    m = FeedForwardVAE(2, utils.parse_components(model, fixed_curvature), _DS(), False).to(dev)
    m.seed_sampler(7)
    tr = Trainer(m, chkpt_dir=str(tmp_path))
    opt = tr.build_optimizer(1e-3, fixed_curvature=fixed_curvature)
    res = tr.train_epochs(opt, train, test, betas=[1.0], epochs=2, likelihood_n=500)
    assert len(res) == 1
    for st in res.values():
        for k, v in st.to_print().items():
            assert np.isfinite(v), f"{k} is not finite."
        assert -1e4 < float(st.log_likelihood) < 1e-8
    for c in m.components:
        K = float(c.manifold.curvature)
        if isinstance(c, EuclideanComponent):
            assert K == 0
        elif isinstance(c, UniversalComponent):
            assert np.isfinite(K)
        elif fixed_curvature:
            want = 0.01 if isinstance(c, (SphericalComponent, StereographicallyProjectedSphereComponent)) else -0.01
            assert isinstance(c, (SphericalComponent, StereographicallyProjectedSphereComponent, HyperbolicComponent,
                                  PoincareComponent))
            assert abs(K - want) < 1e-6
        else:
            assert all(abs(K - v) > 1e-6 for v in (-1.0, 0.0, 1.0))


def test_non_finite_training_raises_like_the_reference(dev, tmp_path):
    """The reference asserts isfinite on the loss every step (vae.py:158); this build reports the same condition, with
    the same exception type, from the epoch's running sums (no per-step host sync)."""
    from mvae_amd import utils
    from mvae_amd.data import DeviceLoader
    from mvae_amd.models import FeedForwardVAE
    from mvae_amd.trainer import Trainer
    x = (torch.rand(8, 784, generator=torch.Generator().manual_seed(0)) * 255).to(torch.uint8).to(dev)
    y = torch.zeros(8, dtype=torch.int64, device=dev)
    train = DeviceLoader(x, y, 4, train=True, binarize=True, seed=1)
    m = FeedForwardVAE(8, utils.parse_components("h2,e2", False), _DS(), False).to(dev)
    with torch.no_grad():
        m.fc_e0.bias.fill_(float("nan"))
    tr = Trainer(m, chkpt_dir=str(tmp_path))
    opt = tr.build_optimizer(1e-3, fixed_curvature=False)
    with pytest.raises(AssertionError):
        tr.train_epochs(opt, train, train, betas=[1.0], epochs=1, likelihood_n=0)


def test_stdout_log_lines_parse_like_read_log(dev, tmp_path, capsys):
    """The epoch lines the Trainer prints are what mt/visualization/read_log.py:53-75 parses (lower-cased line starts
    with 'epoch ' / 'trainepoch ', ends with '}', the part after the first ':' is a dict literal whose keys include
    bce / kl / elbo / ll / mi and one 'comp_XXX_<shortcut>/curvature' per component)."""
    from mvae_amd import utils
    from mvae_amd.data import DeviceLoader
    from mvae_amd.models import FeedForwardVAE
    from mvae_amd.trainer import Trainer
    x = (torch.rand(8, 784, generator=torch.Generator().manual_seed(0)) * 255).to(torch.uint8).to(dev)
    y = torch.zeros(8, dtype=torch.int64, device=dev)
    train = DeviceLoader(x, y, 4, train=True, binarize=True, seed=1)
    m = FeedForwardVAE(8, utils.parse_components("e2,h2,s2", True), _DS(), False).to(dev)
    tr = Trainer(m, chkpt_dir=str(tmp_path))
    opt = tr.build_optimizer(1e-3, fixed_curvature=True)
    tr.train_epochs(opt, train, train, betas=[1.0], epochs=2, likelihood_n=2)
    out = capsys.readouterr().out
    parsed = {"epoch": [], "trainepoch": []}
    for line in out.splitlines():
        line = line.strip().lower()
        if not line.endswith("}"):
            continue
        kind = "epoch" if line.startswith("epoch ") else ("trainepoch" if line.startswith("trainepoch ") else None)
        if kind is None:
            continue
        idx = line.find(":")
        d = eval(line[idx + 1:].strip())  # the reference parser does exactly this
        d["epoch"] = int(line[:idx].strip().split(" ")[1])
        parsed[kind].append(d)
    assert [d["epoch"] for d in parsed["trainepoch"]] == [0, 1] and len(parsed["epoch"]) == 1
    want = {"epoch", "bce", "kl", "elbo", "ll", "mi", "comp_000_e2/curvature", "comp_001_h2/curvature",
            "comp_002_s2/curvature"}
    for d in parsed["trainepoch"] + parsed["epoch"]:
        assert want <= set(d.keys())


def test_export_representations(dev, tmp_path):
    """train.py:297-325: files, shapes and contents of the representation export."""
    from mvae_amd import utils
    from mvae_amd.data import DeviceLoader
    from mvae_amd.models import FeedForwardVAE
    from mvae_amd.trainer import Trainer
    x = (torch.rand(10, 784, generator=torch.Generator().manual_seed(0)) * 255).to(torch.uint8).to(dev)
    y = torch.arange(10, device=dev)
    data = DeviceLoader(x, y, 4, train=False, binarize=True)
    m = FeedForwardVAE(8, utils.parse_components("h2,e3", True), _DS(), False).to(dev)
    m.seed_sampler(1)
    tr = Trainer(m, chkpt_dir=str(tmp_path), export_embeddings=1)
    tr._test_epoch(data, likelihood_n=0, beta=1.0)
    rep = os.path.join(str(tmp_path), "repr")
    assert sorted(os.listdir(rep)) == ["eval_comp_000_h2_0.pt", "eval_comp_001_e3_0.pt", "eval_labels_0.pt",
                                       "eval_total_0.pt"]
    assert torch.equal(torch.load(os.path.join(rep, "eval_labels_0.pt")), torch.arange(10))
    h = torch.load(os.path.join(rep, "eval_comp_000_h2_0.pt"))
    assert h.shape == (10, 3) and torch.load(os.path.join(rep, "eval_total_0.pt")).shape == (10, 6)
    R = float(m.components[0].manifold.radius)
    assert torch.allclose(-h[:, 0]**2 + (h[:, 1:]**2).sum(-1), torch.full((10,), -R * R), rtol=1e-4)  # on the hyperboloid


@pytest.mark.parametrize("rows,xr,Z,H,D", [(3 * 128, 128, 6, 400, 784), (70, 7, 10, 128, 96), (64, 64, 3, 16, 32),
                                           (500, 25, 16, 64, 48), (129, 3, 8, 256, 160), (40, 40, 2, 512, 64),
                                           (256, 128, 48, 400, 784), (70, 7, 20, 128, 96), (64, 64, 64, 64, 32)])
def test_fused_decoder_bce_rows_equals_the_three_operators(dev, rows, xr, Z, H, D):
    """mvae_decode_bce_rows (the log-likelihood estimator's decoder + BCE in one launch, vae.py:98-109) against the composed
    operators it replaces (themselves pinned to the reference's vectors above) and against float64."""
    from mvae_amd import functional as Fn
    gen = torch.Generator().manual_seed(rows + H)
    z = torch.randn(rows // xr, xr, Z, generator=gen)
    w0, b0 = torch.randn(H, Z, generator=gen) * 0.7, torch.randn(H, generator=gen) * 0.3
    wl, bl = torch.randn(D, H, generator=gen) * 0.2, torch.randn(D, generator=gen) * 0.3
    x = (torch.rand(xr, D, generator=gen) < 0.4).float()
    zd, xd = z.to(dev), x.to(dev)
    fused = Fn.decode_bce_rows(zd, w0.to(dev), b0.to(dev), wl.to(dev), bl.to(dev), xd)
    assert fused is not None, "shape inside the fused kernel's coverage"
    hd = Fn.linear_forward(zd.reshape(-1, Z), w0.to(dev), b0.to(dev), relu=True)
    composed = Fn.bce_rows(Fn.linear_forward(hd, wl.to(dev), bl.to(dev)).view(rows // xr, xr, D), xd)
    y = torch.relu(z.double() @ w0.double().T + b0.double()) @ wl.double().T + bl.double()
    ref = torch.nn.functional.binary_cross_entropy_with_logits(y, x.double().expand_as(y), reduction="none").sum(-1)
    assert_close(_cpu(fused), ref.numpy(), 1e-5, "fused vs float64")
    assert_close(_cpu(fused), _cpu(composed), 1e-5, "fused vs composed")


def test_fused_decoder_bce_rows_declines_other_shapes(dev):
    from mvae_amd import functional as Fn
    z = torch.randn(4, 8, 6).to(dev)
    assert Fn.decode_bce_rows(z, torch.randn(48, 6).to(dev), torch.randn(48).to(dev), torch.randn(32, 48).to(dev),
                              torch.randn(32).to(dev), torch.rand(8, 32).to(dev)) is None      # H = 48
    assert Fn.decode_bce_rows(z, torch.randn(64, 6).to(dev), torch.randn(64).to(dev), torch.randn(30, 64).to(dev),
                              torch.randn(30).to(dev), torch.rand(8, 30).to(dev)) is None      # D % 16 != 0


def test_fused_decoder_bce_rows_declines_4gb_targets_at_the_c_abi(dev):
    """The kernel addresses the targets with 32-bit byte offsets: a C caller (no Python wrapper in front) with x_rows * D * 4
    >= 2^32 gets MVAE_E_UNSUPPORTED from the entry point itself, before any launch -- the call declines on the shape alone, so
    small buffers stand in for the 4 GB of targets; one row less is accepted (rows = 0: nothing is launched)."""
    from mvae_amd import _lib as L
    lib = L.load()
    Z, H, D = 6, 64, 784
    t = lambda *shape: torch.zeros(*shape, device=dev)
    z, w0, b0, wl, bl, x, out = t(64, Z), t(H, Z), t(H), t(D, H), t(D), t(64, D), t(64)
    big = (1 << 32) // (D * 4) + 1
    args = lambda rows, xr: (L.ptr(z), rows, Z, L.ptr(w0), L.ptr(b0), L.ptr(wl), L.ptr(bl), L.ptr(x), xr, H, D, L.ptr(out),
                             L.stream_ptr(dev))
    assert lib.mvae_decode_bce_rows(*args(64, big)) == L.MVAE_E_UNSUPPORTED
    assert lib.mvae_decode_bce_rows(*args(0, big)) == 0          # rows == 0 returns before the shape checks
    assert lib.mvae_decode_bce_rows(*args(64, 64)) == 0
    torch.cuda.synchronize()


def test_loglik_tail_all_minus_inf_column(dev):
    """torch.logsumexp (vae.py:114,117) of a column whose n terms are all -inf is -inf, not exp(-inf + inf) = NaN: the reduce
    kernels subtract 0 instead of an infinite maximum, as torch does."""
    from mvae_amd import functional as Fn
    n, B, Z, D, C = 40, 16, 6, 32, 2
    gen = torch.Generator().manual_seed(3)
    bce = torch.rand(n, B, generator=gen) * 10 + 100
    lp, lq = torch.randn(C, n, B, generator=gen), torch.randn(C, n, B, generator=gen)
    lp[:, :, 5] = -float("inf")  # log p(z) = -inf for every sample of batch column 5
    z, x = torch.randn(n, B, Z, generator=gen), (torch.rand(B, D, generator=gen) < 0.3).float()
    out = Fn.loglik_tail(bce.to(dev), lp.to(dev), lq.to(dev), z.to(dev), x.to(dev))
    a = -bce.double() + lp.double().sum(0) - lq.double().sum(0)
    ref = torch.logsumexp(a, dim=0) - np.log(n)
    got = _cpu(out[0])
    assert got[5] == -np.inf and ref[5] == -np.inf
    keep = np.arange(B) != 5
    assert_close(got[keep], ref.numpy()[keep], 1e-5, "log p(x), finite columns")
    got2 = _cpu(Fn.loglik_reduce(bce.to(dev), lp.sum(0).to(dev), lq.sum(0).to(dev))[0])
    assert got2[5] == -np.inf
    assert_close(got2[keep], ref.numpy()[keep], 1e-5, "log p(x), finite columns (mvae_loglik_reduce)")


@pytest.mark.parametrize("model", ["h2,s2,e2", "6h2,6s2,6e2"])
def test_log_likelihood_full_size_vs_oracle(dev, model):
    """ModelVAE.log_likelihood at the reference's evaluation size (n = 500 importance samples, B = 128, H = 400, D = 784:
    64 000 decoded rows through mvae_decode_bce_rows, the component sums / logsumexp / covariance tail in two launches) --
    the WHOLE call against oracle.model.log_likelihood (vae.py:82-123) at 1e-4."""
    from mvae_amd import synthetic
    from oracle import model as M
    m, spec, state0 = _model(dev, model, 784, 400)
    x = synthetic.binary_batches(1, 128, 784)[0]
    eps = torch.randn(500, 128, spec.total_true_dim, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        lp, mi, cn = m.log_likelihood(x.to(dev), n=500, eps=eps.to(dev))
        P = {k: v.detach().clone().float() for k, v in state0.items()}
        rlp, rmi, rcn = M.log_likelihood(spec, P, x, eps)
    assert_close(_cpu(lp), rlp.numpy(), RTOL, "log p(x)")
    assert_close(_cpu(mi), rmi.numpy(), RTOL, "mi", atol_frac=1e-4)
    assert_close(float(cn), float(rcn), 5 * RTOL, "cov_norm")


@pytest.mark.parametrize("n,B,Z,D,C", [(500, 128, 6, 784, 3), (8, 32, 6, 32, 3), (37, 100, 12, 200, 5), (5, 4, 6, 3072, 3),
                                       (300, 1, 3, 17, 1), (50, 128, 48, 784, 18), (9, 16, 33, 48, 4)])
def test_loglik_tail_equals_the_composed_operators(dev, n, B, Z, D, C):
    """mvae_loglik_reduce_comps + mvae_cov_norm (the estimator's tail, vae.py:110-121, in two launches) against float64 and
    against the composition they replace (mvae_loglik_reduce on torch sums, torch means + mvae_linear_backward + norm);
    twice in a row: the arrival counter of the covariance launch re-arms itself."""
    from mvae_amd import functional as Fn
    gen = torch.Generator().manual_seed(n * 7 + B)
    bce = torch.rand(n, B, generator=gen) * 40 + 500
    lp, lq = torch.randn(C, n, B, generator=gen) * 2 - 3, torch.randn(C, n, B, generator=gen) * 2 + 1
    z = torch.randn(n, B, Z, generator=gen)
    x = (torch.rand(B, D, generator=gen) < 0.3).float()
    for _ in range(2):
        out = Fn.loglik_tail(bce.to(dev), lp.to(dev), lq.to(dev), z.to(dev), x.to(dev))
        assert out is not None
        torch.cuda.synchronize()
    a = -bce.double() + lp.double().sum(0) - lq.double().sum(0)
    ref_lp = torch.logsumexp(a, dim=0) - np.log(n)
    ref_mi = torch.logsumexp(lq.double().sum(0) - lp.double().sum(0), dim=0) - np.log(n)
    zn = z.double().mean(0)
    ref_cn = ((x.double() - x.double().mean(0, keepdim=True)).T @ (zn - zn.mean(0, keepdim=True))).norm()
    assert_close(_cpu(out[0]), ref_lp.numpy(), 1e-5, "log p(x)")
    assert_close(_cpu(out[1]), ref_mi.numpy(), 1e-5, "mi", atol_frac=1e-5)
    assert_close(float(out[2]), float(ref_cn), 1e-4, "cov norm", atol_frac=1e-5)
    lp2, mi2 = Fn.loglik_reduce(bce.to(dev), lp.to(dev).sum(dim=0), lq.to(dev).sum(dim=0))
    assert_close(_cpu(out[0]), _cpu(lp2), 1e-5, "log p(x) vs mvae_loglik_reduce")
    assert_close(_cpu(out[1]), _cpu(mi2), 1e-5, "mi vs mvae_loglik_reduce", atol_frac=1e-5)
