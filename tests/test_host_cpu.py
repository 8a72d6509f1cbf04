"""CPU-side checks of the host layer: the C-ABI library loads and exports every declared symbol, argument validation
works without a GPU, the flat layout reproduces the reference's state-dict contract, and the product path refuses
to run without a HIP device (no silent fallback)."""
import ctypes as C
import os
import re

import pytest
import torch

from helpers import load_json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from mvae_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build()
    return _lib


def test_library_exports_every_declared_symbol():
    L = _lib()
    lib = L.load()
    header = open(os.path.join(ROOT, "include", "mvae_hip.h")).read()
    declared = set(re.findall(r"\b(mvae_[a-z0-9_]+)\s*\(", header))
    declared -= {"mvae_ctx"}
    assert declared == set(L.PROTOTYPES), declared ^ set(L.PROTOTYPES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mvae_abi_version() == L.ABI_VERSION


def test_argument_validation_without_gpu():
    L = _lib()
    lib = L.load()
    # NULL pointers / bad kinds are rejected before anything touches the device
    assert lib.mvae_exp_map_mu0(1, None, None, 4, 2, None, None) == -1
    assert lib.mvae_exp_map_mu0(9, 1, 1, 4, 2, 1, None) == -1
    assert b"kind" in lib.mvae_last_error()
    assert lib.mvae_exp_map_mu0(1, 16, 16, 4, 100, 16, None) == -2  # true_dim > MVAE_MAX_TRUE_DIM
    assert lib.mvae_linear_forward(None, None, None, None, 4, 4, 4, 0, None) == -1
    # the conv architecture's fused entry points: NULL pointers, unsupported geometries and shapes
    assert lib.mvae_conv_latent_forward(None, 1, None, None, None, None, 6, None, None, None, None, None, None, None, None, 0,
                                        None, 4, None) == -1
    assert lib.mvae_conv_latent_backward(None, 1, None, None, None, None, 6, None, None, None, None, None, 1, 0, 1.0, None, None,
                                         None, None, 0, None, None, None, None, None, None, None, 4, None) == -1
    assert lib.mvae_conv3_k4s2p1_nchw(None, None, None, None, 1, None, None, 0, 4, 3, 32, 32, 64, None) == -1
    assert lib.mvae_conv3_k4s2p1_nchw(16, 16, None, None, 1, 16, None, 0, 4, 3, 64, 64, 64, None) == -2  # 3 x 32 x 32 only
    assert lib.mvae_conv3_k4s2p1_nchw_wgrad_workspace_floats(256, 3, 32, 32, 64) == 256 * 64 * 48
    assert lib.mvae_conv3_k4s2p1_nchw_wgrad_workspace_floats(300, 3, 32, 32, 64) == 150 * 64 * 48
    assert lib.mvae_conv_bce_stats(None, None, None, None, None, None, 1.0, 4, 3072, 1024, 3, None, None, None, None) == -1
    assert lib.mvae_conv_bce_stats(16, 16, 16, 16, 16, 16, 1.0, 4, 3072, 1000, 3, 16, 16, 16, None) == -2  # HW % 1024
    assert lib.mvae_convt_to3_k4s2p1_forward(None, None, None, None, 4, 64, 16, 16, 3, None) == -1
    assert lib.mvae_convt_to3_k4s2p1_forward(16, 16, 16, 16, 4, 32, 16, 16, 3, None) == -2  # 64 features only
    assert lib.mvae_conv_latent_workspace_floats(256, 3) == max(64 * 256 * 16, 256 * 2048 + 3 * 256)
    assert lib.mvae_set_contraction_mode(-1) in (0, 1, 2)
    with pytest.raises(L.MvaeHipError):
        L.check(-1)


def test_model_desc_validation_without_gpu():
    L = _lib()
    lib = L.load()
    from mvae_amd.engine import FlatLayout
    from mvae_amd.functional import ComponentLayout
    lay = ComponentLayout([("h", 2), ("s", 2), ("e", 2)])
    flat = FlatLayout(lay, 784, 400, False)
    d = L.ModelDesc()
    d.abi_version, d.arch, d.batch, d.in_dim, d.h_dim = L.ABI_VERSION, 0, 128, 784, 400
    d.ncomp, d.heads_dim, d.z_dim, d.eps_dim = 3, lay.heads_dim, lay.z_dim, lay.eps_dim
    d.n_params = flat.n_params
    d.comps = lay.descs
    for f in ("off_w_heads", "off_b_heads", "off_w_e0", "off_b_e0", "off_w_d0", "off_b_d0", "off_w_logits",
              "off_b_logits"):
        setattr(d, f, getattr(flat, f))
    assert lib.mvae_workspace_floats(C.byref(d)) > 128 * 784
    h = C.c_void_p()
    assert lib.mvae_create(C.byref(d), C.byref(h)) == -1  # null buffers
    fake = 0x10000  # never dereferenced by create
    for f in ("params", "grads", "adam_m", "adam_v", "step_count", "workspace", "stats"):
        setattr(d, f, fake)
    assert lib.mvae_create(C.byref(d), C.byref(h)) == 0
    lib.mvae_destroy(h)
    d.heads_dim += 1
    assert lib.mvae_create(C.byref(d), C.byref(h)) == -1  # inconsistent with the component table
    d.heads_dim -= 1
    d.off_w_e0 += 2
    assert lib.mvae_create(C.byref(d), C.byref(h)) == -3  # misaligned segment


def test_layouts_match_reference_state_dict_contract():
    from mvae_amd.engine import FlatLayout
    from mvae_amd.functional import ComponentLayout
    tab = load_json("g5_parser.json")["state_shapes"]
    for key, comps in [("h2,s2,e2|ff", [("h", 2), ("s", 2), ("e", 2)]),
                       ("6h2,6s2,6e2|ff", [("h", 2)] * 6 + [("s", 2)] * 6 + [("e", 2)] * 6), ("e6|ff", [("e", 6)])]:
        lay = ComponentLayout(comps)
        flat = FlatLayout(lay, 784, 400, False)
        assert [[n, list(s)] for n, _, s in flat.entries] == tab[key]
        views = flat.views(torch.zeros(flat.n_params))
        # views tile the head matrix without overlap: writing ones everywhere covers exactly the logical count
        for v in views.values():
            v.fill_(1.0)
        assert int(sum(v.numel() for v in views.values())) == flat.n_logical_params()
    lay = ComponentLayout([("h", 2), ("s", 2), ("e", 2)])
    assert (lay.heads_dim, lay.z_dim, lay.eps_dim) == (12, 8, 6)
    flat = FlatLayout(lay, 784, 400, False)
    assert flat.n_logical_params() == 636798  # P for h2,s2,e2
    d = lay.descs[1]
    assert (d.kind, d.mean_col, d.logvar_col, d.eps_col, d.z_col) == (2, 2, 8, 2, 3)


def test_no_cpu_fallback():
    _lib()
    from mvae_amd import functional as Fn
    from mvae_amd._lib import MvaeHipError
    from mvae_amd.engine import StepEngine
    with pytest.raises(MvaeHipError):
        Fn.exp_map_mu0(1, torch.zeros(4, 2), torch.tensor(1.0))
    with pytest.raises(MvaeHipError):
        Fn.linear_forward(torch.zeros(4, 4), torch.zeros(4, 4), None)
    with pytest.raises(MvaeHipError):
        StepEngine([("h", 2)], 8, 8, "cpu")


def test_product_does_not_import_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "mvae_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "/root/reference" not in src or f.endswith((".hip", ".hpp", ".py")) and \
                    not re.search(r"open\(.*/root/reference", src), f


def test_counter_summaries_are_quoted_only_for_identical_machine_code(tmp_path, monkeypatch):
    """bench.py quotes profiles/r*_pmc_traffic.json for the build it runs: same `source_hash`, or -- a summary of another
    source revision -- identical machine code of every launch it is quoted for (`build.kernel_isa` fingerprints recorded in
    the summary).  A changed kernel, a summary without fingerprints, or another revision's conv translation units: refused."""
    import importlib.util
    import json
    import sys
    from mvae_amd import build
    isa = build.kernel_isa("mvae_step")
    if isa is None:
        pytest.skip("no csrc/_obj/mvae_step.o or no llvm tools: the fingerprints cannot be taken here")
    for k in ("k_enc_fwd", "k_fwd23", "k_dec1_bwd", "k_bwd56"):
        assert k in isa and len(isa[k]) == 16
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(b, "ROOT", str(tmp_path))

    class _Flat:
        def n_logical_params(self):
            return 636000

    class _Lay:
        heads_dim, z_dim, eps_dim = 12, 8, 6

    class _Eng:
        layout, flat = _Lay(), _Flat()

    prof = {"enc_fwd": 0.0046, "latent_fwd": 0.0, "dec1_fwd": 0.0093, "dec1_bwd": 0.0049, "latent_bwd": 0.0, "enc_bwd": 0.0083}
    kern = {k: {"traffic_bytes": 1000 * (i + 1)} for i, k in enumerate(("k_enc_fwd", "k_fwd23", "k_dec1_bwd", "k_bwd56"))}

    def line(doc):
        with open(tmp_path / "profiles" / "r99_pmc_traffic.json", "w") as fh:
            json.dump(doc, fh)
        r = b.mlp_roofline(_Eng(), prof, 28.8e-6, False)
        return r["traffic_step"], r["traffic_source"]

    same = {"source_hash": build.source_hash(), "kernels": kern}
    assert line(same) == (10000, "profiles/r99_pmc_traffic.json")
    other = {"source_hash": "0" * 16, "kernels": kern}
    step, src = line(other)  # another revision, no fingerprints: refused
    assert step is None and "source_hash mismatch" in src
    other["kernel_isa"] = {"unit": "mvae_step", "hashes": dict(isa)}
    step, src = line(other)  # another revision, identical machine code: quoted, and the line says so
    assert step == 10000 and "collected from source_hash " + "0" * 16 in src and "identical" in src
    other["kernel_isa"]["hashes"]["k_bwd56"] = "f" * 16
    step, src = line(other)  # one launch of the step has other machine code: refused
    assert step is None and "source_hash mismatch" in src
    # conv summaries: the conv translation units and every header byte-identical
    ck = {"e2f": {"traffic_bytes": 123}}
    with open(tmp_path / "profiles" / "r99_conv_pmc_traffic.json", "w") as fh:
        json.dump({"source_hash": "0" * 16, "kernels": ck, "conv_file_hashes": build.conv_file_hashes()}, fh)
    val, src = b.conv_kernel_traffic("e2f")
    assert val == 123 and "unchanged" in src
    bad = dict(build.conv_file_hashes(), **{"mvae_p3.hip": "0" * 16})
    with open(tmp_path / "profiles" / "r99_conv_pmc_traffic.json", "w") as fh:
        json.dump({"source_hash": "0" * 16, "kernels": ck, "conv_file_hashes": bad}, fh)
    val, src = b.conv_kernel_traffic("e2f")
    assert val is None and "source_hash mismatch" in src
