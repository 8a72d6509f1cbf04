"""Device-side input pipeline (scope row f-2): gather + dynamic binarisation + eps draw in one launch, and a whole epoch
as HIP-graph replays.  The RNG is Philox (no parity with torch's generator is intended): the tests are statistical and
structural -- the semantics of ImageDynamicBinarization (image_reconstruction.py:44-53) and of N(0,1) draws."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


def _prepare(images, perm, counters, B, E, seed, nb, train, dev):
    from mvae_amd._lib import check, load, ptr, stream_ptr
    N, D = images.shape
    x = torch.zeros(B, D, device=dev)
    eps = torch.zeros(B, E, device=dev)
    check(load().mvae_prepare_batch(ptr(images), ptr(perm), N, D, B, E, C.c_uint64(seed), ptr(counters), nb, train,
                                    ptr(x), ptr(eps), stream_ptr(dev)))
    return x, eps


def test_prepare_batch_semantics(dev):
    N, D, B, E = 512, 784, 128, 6
    g = torch.Generator().manual_seed(0)
    images = torch.randint(0, 256, (N, D), generator=g, dtype=torch.uint8)
    images[:, 0] = torch.arange(N, dtype=torch.int64).remainder(256).to(torch.uint8)  # row tag in pixel 0
    images[:, 1] = (torch.arange(N) // 256).to(torch.uint8) * 255  # second tag: 0 or 255 -> deterministic bit
    images_d = images.to(dev)
    perm = torch.randperm(N, generator=g).to(torch.int32).to(dev)
    counters = torch.zeros(32, dtype=torch.int32, device=dev)
    # eval mode: fixed threshold at 0.5, rows gathered through the permutation of batch (cursor % nb)
    for cursor in (0, 1, 3, 6):
        counters[8] = cursor
        x, _ = _prepare(images_d, perm, counters, B, E, 7, N // B, 0, dev)
        rows = perm[(cursor % 4) * B:(cursor % 4 + 1) * B].long().cpu()
        want = (images[rows].float() / 255.0 > 0.5).float()
        assert torch.equal(x.cpu(), want)
    # train mode: P(x=1) = pixel/255, independent draws per cursor, deterministic per (seed, cursor)
    acc = torch.zeros(B, D, device=dev)
    K = 400
    first = None
    for k in range(K):
        counters[8] = 4 * k  # same batch every time (cursor % nb == 0), different stream position
        x, eps = _prepare(images_d, perm, counters, B, E, 7, N // B, 1, dev)
        assert set(torch.unique(x).tolist()) <= {0.0, 1.0}
        if k == 0:
            first = (x.clone(), eps.clone())
            x2, eps2 = _prepare(images_d, perm, counters, B, E, 7, N // B, 1, dev)
            assert torch.equal(x, x2) and torch.equal(eps, eps2)  # reproducible
            x3, eps3 = _prepare(images_d, perm, counters, B, E, 8, N // B, 1, dev)
            assert not torch.equal(eps, eps3)  # seed matters
        acc += x
    p = images[perm[:B].long().cpu()].float() / 255.0
    freq = (acc / K).cpu()
    assert float((freq - p).abs().max()) < 5 * 0.5 / np.sqrt(K) + 1e-3  # 5 sigma of a Bernoulli mean
    assert float((freq - p).abs().mean()) < 0.03
    assert not torch.equal(first[0], x)
    # eps ~ N(0,1)
    counters[8] = 12345
    _, eps = _prepare(images_d, perm, counters, 4096, 64, 7, 1, 1, dev) if False else (None, None)
    big = torch.cat([_prepare(images_d, perm, counters.index_fill_(0, torch.tensor([8], device=dev), 100 + k), B, E, 7,
                              N // B, 1, dev)[1].flatten() for k in range(200)])
    assert abs(float(big.mean())) < 0.02 and abs(float(big.var()) - 1.0) < 0.03
    assert abs(float((big**4).mean()) - 3.0) < 0.2  # kurtosis of a normal
    assert float(big.abs().max()) < 6.5


def test_epoch_runner_graph_equals_eager(dev):
    """One epoch as graph replays == the same [prepare, step] pairs launched eagerly (bit-identical parameters), the
    step counter / cursor advance by the number of batches, and training makes progress."""
    from mvae_amd import synthetic
    from mvae_amd.engine import StepEngine
    from mvae_amd.runner import EpochRunner
    imgs = (synthetic.digits_like_batches(10, 100).reshape(-1, 784) * 230 + 12).to(torch.uint8).to(dev)  # 1000 images
    results = []
    for use_graphs in (True, False):
        eng = StepEngine([("h", 2), ("s", 2), ("e", 2)], 784, 400, dev, radius_trainable=[True, True, False])
        eng.load_state(synthetic.synthetic_state([(n, s) for n, _, s in eng.flat.entries], radius=2.0))
        er = EpochRunner(eng, imgs, batch=128, seed=3, graph_steps=3)
        assert er.nb == 7
        for ep in range(2):
            n = er.run_epoch(1.0, ep >= 1, use_graphs=use_graphs)
            assert n == 7
        torch.cuda.synchronize()
        st = eng.read_stats()
        assert st["sum"]["steps"] == 14
        assert int(eng.counters[0]) == 14 and int(eng.counters[8]) == 14
        assert np.isfinite(st["last"]["elbo"])
        results.append((eng.params.clone(), st))
    assert torch.equal(results[0][0], results[1][0])
    assert results[0][1]["sum"]["elbo"] == results[1][1]["sum"]["elbo"]


def test_split_step_keeps_cursor_and_counter(dev):
    """Data-parallel order (gradients, then k_optim): the Adam counter and the batch cursor both advance by one per
    step -- the arrival scratch words of k_optim must not overlap either of them."""
    from mvae_amd import synthetic
    from mvae_amd.engine import StepEngine
    eng = StepEngine([("h", 2), ("s", 2), ("e", 2)], 784, 400, dev, radius_trainable=[True, True, False])
    eng.load_state(synthetic.synthetic_state([(n, s) for n, _, s in eng.flat.entries], radius=1.0))
    x = synthetic.binary_batches(1, 128, 784)[0].to(dev)
    eps = synthetic.eps_batches(1, 128, 6)[0].to(dev)
    for k in range(5):
        eng.forward_backward(x, eps, 1.0)
        eng.optimizer_step(True)
        torch.cuda.synchronize()
        assert int(eng.counters[0]) == k + 1 and int(eng.counters[8]) == k + 1
        assert int(eng.counters[1]) == 0 and int(eng.counters[16:].abs().sum()) == 0


def test_prepare_batch_without_binarisation(dev):
    """mvae_prepare_batch(train = 2): the CIFAR pipeline of the reference (ToTensor only, image_reconstruction.py:123-127): the
    gathered rows as pixel / 255 -- bit-exact against torch's division --, the same eps stream as the binarising mode."""
    N, D, B, E = 384, 3072, 64, 6
    g = torch.Generator().manual_seed(1)
    images = torch.randint(0, 256, (N, D), generator=g, dtype=torch.uint8)
    images[:, :256] = torch.arange(256, dtype=torch.uint8)  # every pixel value in every row
    images_d = images.to(dev)
    perm = torch.randperm(N, generator=g).to(torch.int32).to(dev)
    counters = torch.zeros(32, dtype=torch.int32, device=dev)
    for cursor in (0, 2, 7):
        counters[8] = cursor
        x, eps = _prepare(images_d, perm, counters, B, E, 5, N // B, 2, dev)
        rows = perm[(cursor % 6) * B:(cursor % 6 + 1) * B].long().cpu()
        assert torch.equal(x.cpu(), images[rows].float() / 255.0)
        _, eps1 = _prepare(images_d, perm, counters, B, E, 5, N // B, 1, dev)
        assert torch.equal(eps, eps1)
    # identity permutation (perm = NULL)
    counters[8] = 1
    x, _ = _prepare(images_d, None, counters, B, E, 5, N // B, 2, dev)
    assert torch.equal(x.cpu(), images[B:2 * B].float() / 255.0)


def test_conv_epoch_runner_graph_equals_eager(dev):
    """Scope row f-2 for the conv architecture: whole CIFAR-shaped epochs as HIP-graph replays of [mvae_prepare_batch(train = 2),
    ConvEngine.train_step] equal the same pairs launched eagerly bit for bit; the optimizer launch advances the batch cursor
    (counters[8]) and the Adam counter by one per step; the first batch of the epoch is the gather the permutation says."""
    from mvae_amd import synthetic
    from mvae_amd.conv import ConvEngine
    from mvae_amd.runner import EpochRunner
    g = torch.Generator().manual_seed(4)
    imgs = torch.randint(0, 256, (400, 3072), generator=g, dtype=torch.uint8).to(dev)
    results = []
    for use_graphs in (True, False):
        eng = ConvEngine([("h", 2), ("s", 2), ("e", 2)], dev, radius_trainable=[True, True, False])
        shapes = [(name, tuple(v.shape)) for name, v in eng.param_views().items()]
        eng.load_state(synthetic.synthetic_state(shapes, radius=2.0, transposed_conv=("d1", "d2", "d3")))
        er = EpochRunner(eng, imgs, batch=64, seed=9, graph_steps=2, binarize=False)
        assert er.nb == 6 and er.mode == 2
        for ep in range(2):
            assert er.run_epoch(1.0, ep >= 1, use_graphs=use_graphs) == 6
        torch.cuda.synchronize()
        st = eng.read_stats()
        assert st["sum"]["steps"] == 12 and int(eng.counters[0]) == 12 and int(eng.counters[8]) == 12
        assert np.isfinite(st["last"]["elbo"])
        # the LAST batch the pipeline prepared = batch 5 of the second epoch's permutation, unbinarised
        rows = er.perm[5 * 64:6 * 64].long()
        assert torch.equal(er.x.cpu(), imgs[rows].cpu().float() / 255.0)  # (ATen's device division by a scalar multiplies by 1/255)
        results.append((eng.params.clone(), st))
    assert torch.equal(results[0][0], results[1][0])
    assert results[0][1]["sum"]["elbo"] == results[1][1]["sum"]["elbo"]


def test_trainer_takes_the_device_pipeline_for_the_conv_architecture(dev, tmp_path):
    """Trainer._train_epoch with a CIFAR-shaped uint8 DeviceLoader (no binarisation) and the ConvEngine: the epoch runs through
    runner.EpochRunner (mode 2: pixel / 255), the ragged last batch through train_step, statistics count every sample."""
    from mvae_amd import utils
    from mvae_amd.data import DeviceLoader, VaeDataset
    from mvae_amd.models import ConvolutionalVAE
    from mvae_amd.trainer import Trainer
    g = torch.Generator().manual_seed(2)
    x = torch.randint(0, 256, (150, 3072), generator=g, dtype=torch.uint8).to(dev)
    y = torch.zeros(150, dtype=torch.int64, device=dev)
    train = DeviceLoader(x, y, 32, train=True, binarize=False, seed=1)
    torch.manual_seed(0)
    m = ConvolutionalVAE(8192, utils.parse_components("h2,s2,e2", False), VaeDataset(32, 3072, (3, 32, 32)), False).to(dev)
    m.seed_sampler(3)
    tr = Trainer(m, chkpt_dir=str(tmp_path))
    opt = tr.build_optimizer(1e-3, fixed_curvature=False)
    st = tr._train_epoch(opt, train, beta=1.0)
    er = tr._epoch_runner
    assert er.mode == 2 and er.nb == 4 and er.eng is m.engine
    assert tr.global_step == 5  # four full batches through the pipeline + the 22-image tail
    assert int(m.engine.counters[0]) == 5
    d = st.to_print()
    assert all(np.isfinite(v) for v in d.values()) and d["elbo"] < 0


@pytest.mark.parametrize("comps,mode", [("h2,s2,e2", 1), ("h2,s2,e2", 0), ("h2,s2,e2", 2), ("6h2,6s2,6e2", 1), ("e5", 1)])
def test_next_batch_feed_equals_prepare_batch(dev, comps, mode):
    """mvae_set_next_batch_feed: a step that also prepares its successor's batch (spare workgroups of launch 4) writes the
    bits mvae_prepare_batch writes when called after the step, advances nothing else, and is one-shot; fused step and
    gradients-only step; every kernel path (fused forward, block kernels, per-row)."""
    from mvae_amd import synthetic
    from mvae_amd.engine import StepEngine
    spec = {"h2,s2,e2": [("h", 2), ("s", 2), ("e", 2)], "6h2,6s2,6e2": [("h", 2)] * 6 + [("s", 2)] * 6 + [("e", 2)] * 6,
            "e5": [("e", 5)]}[comps]
    N, D, B = 1024, 784, 128
    g = torch.Generator().manual_seed(5)
    images = torch.randint(0, 256, (N, D), generator=g, dtype=torch.uint8).to(dev)
    perm = torch.randperm(N, generator=g).to(torch.int32).to(dev)
    res = []
    for armed in (True, False):
        eng = StepEngine(spec, D, 400, dev)
        eng.load_state(synthetic.synthetic_state([(n, s) for n, _, s in eng.flat.entries], radius=1.5))
        E = eng.layout.eps_dim
        eng.counters[8] = 5
        x, eps = _prepare(images, perm, eng.counters, B, E, 11, N // B, mode, dev)
        nx, ne = torch.full((B, D), -7.0, device=dev), torch.full((B, E), -7.0, device=dev)
        for k in range(2):  # k = 0: fused step, k = 1: gradients + optimizer
            if armed:
                eng.set_next_batch_feed(B, images, perm, 11, N // B, mode, nx, ne)
            if k == 0:
                eng.train_step(x, eps, 1.0, True)
            else:
                eng.forward_backward(x, eps, 1.0)
                eng.optimizer_step(True)
            torch.cuda.synchronize()
            assert int(eng.counters[8]) == 6 + k
            wx, we = _prepare(images, perm, eng.counters, B, E, 11, N // B, mode, dev)
            if armed:
                assert torch.equal(nx, wx) and torch.equal(ne, we)
                nx.fill_(-7.0), ne.fill_(-7.0)
        # one-shot: an un-armed step leaves the buffers alone
        eng.train_step(x, eps, 1.0, False)
        torch.cuda.synchronize()
        assert float(nx.max()) == -7.0 and float(ne.max()) == -7.0
        res.append((eng.params.clone(), eng.stats.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("gs", [2, 3])
def test_epoch_runner_in_step_preparation_equals_pairs(dev, gs):
    """EpochRunner with the batch prepared inside the previous step == [mvae_prepare_batch, step] pairs: bit-identical
    parameters and statistics over two epochs, for even and odd graph lengths (odd: the graphs of both buffer parities)."""
    from mvae_amd import synthetic
    from mvae_amd.engine import StepEngine
    from mvae_amd.runner import EpochRunner
    imgs = (synthetic.digits_like_batches(10, 100).reshape(-1, 784) * 230 + 12).to(torch.uint8).to(dev)  # 1000 images
    results = []
    for fold in (True, False):
        eng = StepEngine([("h", 2), ("s", 2), ("e", 2)], 784, 400, dev, radius_trainable=[True, True, False])
        eng.load_state(synthetic.synthetic_state([(n, s) for n, _, s in eng.flat.entries], radius=2.0))
        er = EpochRunner(eng, imgs, batch=128, seed=3, graph_steps=gs, fold=fold)
        assert er.fold == fold
        for ep in range(3):
            assert er.run_epoch(1.0, ep >= 1) == 7
        torch.cuda.synchronize()
        assert int(eng.counters[0]) == 21 and int(eng.counters[8]) == 21
        results.append((eng.params.clone(), eng.stats.clone()))
    assert torch.equal(results[0][0], results[1][0]) and torch.equal(results[0][1], results[1][1])


def test_next_batch_feed_refuses_the_buffers_the_step_reads(dev):
    from mvae_amd import synthetic
    from mvae_amd._lib import MvaeHipError
    from mvae_amd.engine import StepEngine
    eng = StepEngine([("h", 2), ("s", 2), ("e", 2)], 784, 400, dev)
    eng.load_state(synthetic.synthetic_state([(n, s) for n, _, s in eng.flat.entries], radius=1.5))
    images = torch.randint(0, 256, (256, 784), dtype=torch.uint8).to(dev)
    x = synthetic.binary_batches(1, 128, 784)[0].to(dev)
    eps = synthetic.eps_batches(1, 128, 6)[0].to(dev)
    eng.set_next_batch_feed(128, images, None, 1, 2, 1, x, torch.empty_like(eps))
    with pytest.raises(MvaeHipError, match="buffers this step reads"):
        eng.train_step(x, eps, 1.0, False)
    eng.train_step(x, eps, 1.0, False)  # the refused arming is gone
    torch.cuda.synchronize()
    assert int(eng.counters[0]) == 1


def test_epoch_runner_at_the_reference_batch_size_pads_rows(dev, monkeypatch):
    """`--batch_size 100` (the reference CLI's default, mt/examples/run.py:32) through the device-side pipeline: the buffers are
    padded to 112 rows, the pipeline prepares 100 of them and the step masks the rest (four launches on the fused kernels
    instead of the one-row-per-workgroup path).  Same permutation, same Philox items -> the same batches: after an epoch the
    parameters and the epoch statistics agree with a runner that does not pad (MVAE_NO_PAD_ROWS=1) to float32 rounding."""
    from helpers import assert_close, assert_close_after_adam
    from mvae_amd import synthetic
    from mvae_amd.engine import StepEngine
    from mvae_amd.runner import EpochRunner
    from oracle import model as M
    spec = M.Spec("h2,s2,e2", in_dim=784, h_dim=400, fixed_curvature=False)
    state0 = synthetic.synthetic_state(spec.named_shapes(), radius=2.0)
    imgs = (synthetic.digits_like_batches(7, 100).reshape(-1, 784) * 255).to(torch.uint8).to(dev)[:650]  # 6 batches + a tail
    res = {}
    for mode in ("padded", "exact"):
        if mode == "exact":
            monkeypatch.setenv("MVAE_NO_PAD_ROWS", "1")
        eng = StepEngine([("h", 2), ("s", 2), ("e", 2)], 784, 400, dev, radius_trainable=[True, True, False])
        eng.load_state(state0)
        er = EpochRunner(eng, imgs, 100, seed=5, graph_steps=3)
        assert er.Bp == (112 if mode == "padded" else 100) and er.nb == 6
        assert eng.kernel_path(er.Bp) == ("fused" if mode == "padded" else "row")
        for _ in range(2):
            er.run_epoch(0.8, True)
        torch.cuda.synchronize()
        assert tuple(er.x.shape) == (100, 784)
        st = eng.read_stats()
        res[mode] = ({n: t.detach().cpu().numpy().copy() for n, t in eng.param_views().items()}, st["sum"]["elbo"], st["sum"]["steps"])
    assert res["padded"][2] == res["exact"][2] == 12
    assert_close(res["padded"][1], res["exact"][1], 1e-4, "sum of the ELBO over two epochs")
    for n, v in res["padded"][0].items():
        if n.endswith("radius"):
            assert_close(v, res["exact"][0][n], 2e-4, n)
        else:
            assert_close_after_adam(v, res["exact"][0][n], 1e-3, 12, f"param {n}: padded rows vs exact batch", bad_frac=2e-3)


def test_pipeline_eps_is_never_exactly_zero(dev):
    """Round 6: with Box-Muller's first uniform on (0, 1] an eps PAIR is exactly (0, 0) with probability 2^-24, and a sphere
    component whose eps is the zero vector is 0 / 0 in the reference's formula (spherical.py:87-88) -- the float32 CLI run went
    non-finite at a random epoch (2 of 16 seeds within 100 epochs).  The two draws that did it (found by tools/eps_zero_scan.py in the
    old stream: seed 8, batch 100, cursor 51138, row 50; seed 9, batch 128, cursor 16186, row 122 -- both the s2 component)
    and a scan of 3000 batches: no entry of eps is exactly zero, and the draws are still standard normal."""
    from mvae_amd.engine import StepEngine
    from mvae_amd.runner import EpochRunner
    os.environ.pop("MVAE_NO_PAD_ROWS", None)
    images = (torch.rand(60000, 784, device=dev) * 255).to(torch.uint8)
    for seed, B, cursors in ((8, 100, [51138] + list(range(1500))), (9, 128, [16186] + list(range(1500)))):
        eng = StepEngine([("h", 2), ("s", 2), ("e", 2)], 784, 400, dev, radius_trainable=[True, True, False])
        er = EpochRunner(eng, images, B, seed=seed, fold=False)
        zeros, acc = 0, []
        for cur in cursors:
            eng.counters[8] = cur
            er._prepare()
            e = er.eps
            zeros += int((e == 0).sum())
            if cur < 200:
                acc.append(e.clone())
        assert zeros == 0, f"seed {seed}: {zeros} exactly-zero eps entries"
        allv = torch.cat(acc).double()
        assert abs(float(allv.mean())) < 0.01 and abs(float(allv.var()) - 1.0) < 0.02
        assert float(allv.abs().max()) < 6.0


def test_randn_nonzero_one_launch_draw(dev):
    """`Fn.randn_nonzero` (mvae_randn: the eps draw of an eager step / of log_likelihood's samples in one launch): standard
    normal (moments of 4 M draws, both tails), never exactly zero, any size and alignment, seeded from -- and advancing -- the
    torch generator (re-seeding reproduces the draws, a second call continues the stream), and the same bits for the same
    (seed, offset) through the C ABI whatever the buffer's alignment."""
    import ctypes as C
    from mvae_amd import functional as Fn
    from mvae_amd._lib import check, load, ptr, stream_ptr
    g = torch.Generator(device=dev).manual_seed(1234)
    a = Fn.randn_nonzero((4, 1 << 20), dev, g)
    b = Fn.randn_nonzero((4, 1 << 20), dev, g)
    assert a.shape == (4, 1 << 20) and a.dtype == torch.float32 and not torch.equal(a, b)
    v = torch.cat([a.flatten(), b.flatten()]).double()
    assert int((v == 0).sum()) == 0 and bool(torch.isfinite(v).all())
    m, s2 = float(v.mean()), float(v.var())
    assert abs(m) < 2e-3 and abs(s2 - 1.0) < 3e-3, (m, s2)
    assert abs(float((v ** 3).mean())) < 1e-2 and abs(float((v ** 4).mean()) - 3.0) < 3e-2
    assert 4.5 < float(v.max()) < 6.5 and -6.5 < float(v.min()) < -4.5  # 8 M draws: both tails are populated
    assert abs(float((v.abs() < 1.0).double().mean()) - 0.682689) < 1e-3
    g.manual_seed(1234)
    assert torch.equal(Fn.randn_nonzero((4, 1 << 20), dev, g), a) and torch.equal(Fn.randn_nonzero((4, 1 << 20), dev, g), b)
    # the default generator: torch.manual_seed reproduces
    torch.manual_seed(7)
    c1 = Fn.randn_nonzero((3, 5, 7), dev)
    torch.manual_seed(7)
    assert torch.equal(Fn.randn_nonzero((3, 5, 7), dev), c1) and c1.shape == (3, 5, 7)
    # C ABI: same (seed, offset) = same bits for an unaligned buffer and a ragged count; another offset = another stream
    lib = load()
    buf = torch.zeros(1003 + 1, device=dev)
    check(lib.mvae_randn(ptr(buf[1:]), 1003, 99, 8, stream_ptr(dev)))
    al = torch.zeros(1003, device=dev)
    check(lib.mvae_randn(ptr(al), 1003, 99, 8, stream_ptr(dev)))
    assert torch.equal(buf[1:], al) and float(buf[0]) == 0.0
    other = torch.zeros(1003, device=dev)
    check(lib.mvae_randn(ptr(other), 1003, 99, 12, stream_ptr(dev)))
    assert not torch.equal(other, al)
    assert lib.mvae_randn(None, 4, 0, 0, stream_ptr(dev)) != 0  # null pointer: MVAE_E_BADARG
