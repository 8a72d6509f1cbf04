"""Accuracy of the short float32 elementary functions (mvae_amd/csrc/mvae_fastmath.hpp) against float64, through a
host build of the same header (g++; CPU-only test)."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include "mvae_fastmath.hpp"
extern "C" {
void t_sinhcosh(const float* x, float* s, float* c, int n) { for (int i = 0; i < n; ++i) mvf::sinhcosh(x[i], s + i, c + i); }
void t_sincos(const float* x, float* s, float* c, int n) { for (int i = 0; i < n; ++i) mvf::sincos_fast(x[i], s + i, c + i); }
void t_log1p(const float* x, float* o, int n) { for (int i = 0; i < n; ++i) o[i] = mvf::log1p_pos(x[i]); }
}
'''


@pytest.fixture(scope="module")
def lib():
    d = tempfile.mkdtemp()
    with open(os.path.join(d, "t.cpp"), "w") as fh:
        fh.write(SRC)
    so = os.path.join(d, "t.so")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I", os.path.join(ROOT, "mvae_amd", "csrc"),
                    os.path.join(d, "t.cpp"), "-o", so], check=True)
    return C.CDLL(so)


def _call2(fn, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    a, b = np.empty_like(x), np.empty_like(x)
    fn(x.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), C.c_int(x.size))
    return a, b


def _ulps(got, want):
    want32 = want.astype(np.float32)
    return np.abs(got.astype(np.float64) - want) / np.maximum(np.spacing(np.abs(want32)).astype(np.float64), 1e-45)


def test_sinhcosh(lib):
    rng = np.random.RandomState(0)
    x = np.concatenate([rng.uniform(-85, 85, 20000), rng.uniform(-1, 1, 20000), rng.uniform(-0.4, 0.4, 20000),
                        [0.0, 1e-8, -1e-8, 0.35, 0.3499999, 85.0, -85.0, 1e-20]]).astype(np.float32)
    s, c = _call2(lib.t_sinhcosh, x)
    xd = x.astype(np.float64)
    assert _ulps(c, np.cosh(xd)).max() <= 4
    assert _ulps(s, np.sinh(xd)).max() <= 4
    assert s[np.where(x == 0)[0][0]] == 0.0


def test_sincos(lib):
    rng = np.random.RandomState(1)
    x = np.concatenate([rng.uniform(-8000, 8000, 30000), rng.uniform(-10, 10, 30000), rng.uniform(-1e-3, 1e-3, 1000),
                        [0.0, np.pi / 2, np.pi, 3 * np.pi / 2, 1e-10]]).astype(np.float32)
    s, c = _call2(lib.t_sincos, x)
    xd = x.astype(np.float64)
    # absolute error bound (values near the zeros of sin/cos are limited by the reduction, as in any f32 sincos)
    assert np.abs(s - np.sin(xd)).max() < 2.5e-7
    assert np.abs(c - np.cos(xd)).max() < 2.5e-7
    small = np.abs(xd) < 3.0
    assert _ulps(s[small], np.sin(xd[small]))[np.abs(np.sin(xd[small])) > 1e-3].max() <= 4


def test_log1p_pos(lib):
    rng = np.random.RandomState(2)
    e = np.concatenate([np.exp(rng.uniform(-40, 20, 50000)), [0.0, 1e-30, 1e-8, 1.0, 3e38]]).astype(np.float32)
    out = np.empty_like(e)
    lib.t_log1p(e.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_int(e.size))
    assert _ulps(out, np.log1p(e.astype(np.float64))).max() <= 4


def test_tile_index_division_by_float_reciprocal_is_exact():
    """`fast_div` (csrc/mvae_common.hpp): a / b as int((a + 0.5) * rcp(b)) in float32, used for the tile index of a
    workgroup.  Exact for every grid size in use, also with a reciprocal that is off by one or two ulps (v_rcp_f32)."""
    a = np.arange(0, 1 << 16, dtype=np.int64)
    for b in list(range(1, 600)) + [784, 1024, 4096]:
        r = np.float32(1) / np.float32(b)
        for rr in (r, np.nextafter(r, np.float32(0)), np.nextafter(r, np.float32(2)),
                   np.nextafter(np.nextafter(r, np.float32(2)), np.float32(2))):
            q = ((a.astype(np.float32) + np.float32(0.5)) * rr).astype(np.int64)
            assert np.array_equal(q, a // b), b
