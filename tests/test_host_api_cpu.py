"""CPU tests of the host-side mirror of the reference API: model-string grammar, parameter naming/counting, the
early-stopping rule, the CLI surface -- everything that does not need a kernel launch."""
import numpy as np
import pytest
import torch

from helpers import load_json
from mvae_amd import utils
from mvae_amd._lib import MvaeHipError
from mvae_amd.models import ConvolutionalVAE, FeedForwardVAE
from mvae_amd.trainer import Trainer


class _DS:
    in_dim = 784

    def reconstruction_loss(self, a, b):
        raise AssertionError


def test_parse_components_table():
    """Reference tests/mvae/test_utils.py + golden table recorded from the reference parser."""
    tab = load_json("g5_parser.json")
    for s, ref in tab["parse"].items():
        comps = utils.parse_components(s, fixed_curvature=False)
        assert utils.canonical_name(comps) == ref["canonical"]
        for c, r in zip(comps, ref["components"]):
            assert type(c).__name__ == r["class"]
            assert (c.dim, c.true_dim, c._shortcut()) == (r["dim"], r["true_dim"], r["shortcut"])
            assert [n for n, _ in c.named_parameters()] == r["params"]
    for s, err in tab["errors"].items():
        with pytest.raises({"ValueError": ValueError, "NotImplementedError": NotImplementedError}[err]):
            utils.parse_components(s, False)
    assert utils.parse_components("", False) == []
    for k, v in tab["linear_betas"].items():
        a, b, c, d = k.split(",")
        got = utils.linear_betas(float(a), float(b), int(c), int(d))
        assert np.allclose(got[:len(v)], v)


def test_constant_component_fails_where_the_reference_fails():
    """`c` parses, but Component.init_layers cannot build its sampling procedure (missing `dim`): a TypeError in the
    reference (component.py:48-50 + sampling_procedures.py:119-131) and here."""
    comp = utils.parse_components("c3", False)[0]
    assert type(comp).__name__ == "ConstantComponent" and comp._shortcut() == "c3"
    with pytest.raises(TypeError):
        comp.init_layers(8, scalar_parametrization=False)


def test_fixed_curvature_freezes_radii():
    comps = utils.parse_components("h2,s2,p2,e2", fixed_curvature=True)
    assert [c._radius_param().requires_grad for c in comps[:3]] == [False] * 3
    assert comps[3]._radius_param() is None
    comps = utils.parse_components("d2,u2", fixed_curvature=True)
    assert [c._radius_param().requires_grad for c in comps] == [False] * 2
    assert [n for c in utils.parse_components("d2,u2", False) for n, p in c.named_parameters() if p.dim() == 0] == \
        ["_pradius", "_curvature"]
    comps = utils.parse_components("h2,s2", fixed_curvature=False)
    assert all(c._radius_param().requires_grad for c in comps)
    for c in comps:
        c.init_layers(8, scalar_parametrization=False)  # the manifold is created here, as in the reference
    assert float(comps[0].manifold.curvature) == -1.0 and float(comps[1].manifold.curvature) == 1.0


def test_state_dict_contract_conv():
    tab = load_json("g5_parser.json")["state_shapes"]
    m = ConvolutionalVAE(8192, utils.parse_components("h2,s2,e2", False), _DS(), False)
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == tab["h2,s2,e2|conv"]
    from mvae_amd.conv import ConvFlatLayout
    from mvae_amd.functional import ComponentLayout
    flat = ConvFlatLayout(ComponentLayout([("h", 2), ("s", 2), ("e", 2)]))
    assert [[n, list(s)] for n, _, s in flat.entries] == tab["h2,s2,e2|conv"]
    assert sum(int(torch.tensor(s).prod()) if s else 1 for _, _, s in flat.entries) == 2090001


@pytest.mark.parametrize("model", ["h2,s2,e2", "6h2,6s2,6e2", "e6", "u2,d2,e2"])
def test_state_dict_contract(model):
    """Parameter names, shapes and registration order are the reference's (checkpoint compatibility)."""
    tab = load_json("g5_parser.json")["state_shapes"]
    torch.manual_seed(0)
    m = FeedForwardVAE(400, utils.parse_components(model, False), _DS(), False)
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == tab[f"{model}|ff"]
    assert m.total_z_dim == sum(c.dim for c in m.components)


def test_no_cpu_execution_path():
    m = FeedForwardVAE(16, utils.parse_components("h2,e2", False), _DS(), False)
    with pytest.raises(MvaeHipError):
        m(torch.zeros(2, 784))
    with pytest.raises(ValueError):  # the reference only warns (run.py:117-119); a wrong h_dim cannot work
        ConvolutionalVAE(400, utils.parse_components("e2", False), _DS(), False)


def test_should_stop_rule():
    """train.py:79-95 restated: stop when no epoch in the lookahead window beats the window start."""
    class R:
        def __init__(self, e):
            self.elbo = e
    res = {i: R(v) for i, v in enumerate([-10, -9, -8, -8.5, -8.6, -8.7])}
    assert Trainer._should_stop(res, 5, 3, 100) == 2
    assert Trainer._should_stop(res, 4, 3, 100) is None
    assert Trainer._should_stop(res, 4, 3, 4) == 2
    res2 = {i: R(v) for i, v in enumerate([-10, -9, -8, -7])}
    assert Trainer._should_stop(res2, 3, 2, 3) == 3


def test_cli_flags_match_reference():
    import argparse
    from mvae_amd import run
    flags = ("--device --data --batch_size --learning_rate --epochs --warmup --lookahead --model --architecture "
             "--universal --dataset --h_dim --seed --show_embeddings --export_embeddings --test_every "
             "--train_statistics --scalar_parametrization --fixed_curvature --doubles --beta_start --beta_end "
             "--beta_end_epoch --likelihood_n").split()
    src = open(run.__file__).read()
    for f in flags:  # mt/examples/run.py:30-84
        assert f'"{f}"' in src, f
    assert run.str2bool("True") is True and run.str2bool("false") is False
    with pytest.raises(argparse.ArgumentTypeError):
        run.str2bool("maybe")


def test_cifar_readers(tmp_path):
    """Both on-disk layouts of CIFAR-10 decode to the same [N,3072] uint8 matrix and labels."""
    import pickle
    from mvae_amd.data import CifarVaeDataset
    rng = np.random.default_rng(0)
    data = rng.integers(0, 256, size=(7, 3072), dtype=np.uint8)
    labels = rng.integers(0, 10, size=7)
    py = tmp_path / "a" / "cifar-10-batches-py"
    py.mkdir(parents=True)
    with open(py / "test_batch", "wb") as fh:
        pickle.dump({"data": data, "labels": labels.tolist()}, fh)
    bn = tmp_path / "b" / "cifar-10-batches-bin"
    bn.mkdir(parents=True)
    np.concatenate([labels.astype(np.uint8)[:, None], data], axis=1).tofile(bn / "test_batch.bin")
    for sub in ("a", "b"):
        x, y = CifarVaeDataset(4, str(tmp_path / sub), device="cpu")._read(train=False)
        assert np.array_equal(x, data) and np.array_equal(y, labels)
    assert CifarVaeDataset(4, str(tmp_path / "none"), device="cpu")._read(train=True) is None
