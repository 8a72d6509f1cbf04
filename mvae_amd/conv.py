"""ConvEngine: the CIFAR conv architecture of the reference (mt/mvae/models/conv_vae.py:28-79) on the HIP kernels.

Layers (conv_vae.py:47-55): e0 Conv(3->64) e1 Conv(64->128) e2 Conv(128->512), all k4 s2 p1 + ReLU -> flatten 8192 ->
component heads -> latent components -> d0 Linear(Z->2048)+ReLU -> view [128,4,4] -> d1 ConvT(128->256) d2 ConvT(256->64)
(+ReLU) d3 ConvT(64->3) -> flatten 3072 -> BCE-with-logits (soft targets).

Every Conv2d / ConvTranspose2d runs on an LDS-tiled f32-MFMA contraction (`k_gemm_tiled`, csrc/mvae_conv.hip);
activations between layers are channel-last ([B*H*W, C]).  Where a convolution GATHERS (Conv2d forward, its weight
gradient, ConvTranspose2d backward-data and weight gradient) the contraction is implicit: the operand fetch of the tiled
kernel reads the activation directly (`mvae_conv_k4s2p1_nhwc`, `..._wgrad`), no patch matrix exists in memory (round 1
wrote and re-read 16x the activation for each of them: ~2 GB of HBM traffic per step at B = 256).  Where it SCATTERS
(ConvTranspose2d forward, Conv2d backward-data) the contraction writes a patch matrix that `mvae_col2im_k4s2p1` folds
(<= 4 terms per output, no atomics) or runs as four implicit contractions, one per output parity class, whichever measured
faster for the layer.  The 3-channel NCHW boundary layers (e0, d3) have no patch matrix either: their kernels read the image
straight into MFMA fragments (csrc/mvae_edge.hip), and the forward of d3 runs inside the loss-end launch
(`mvae_convt_to3_bce_stats`).  Between the last encoder and the first decoder convolution the latent
section (flatten -> heads -> components -> decoder fc) is two fused launches each way (`mvae_conv_latent_forward /
_backward`), the loss end (BCE, statistics, d3.bias) one (`mvae_conv_bce_stats`).  Parameters, gradients and optimizer
state live in flat buffers laid out like StepEngine's (first 64 floats = radii), so the optimizer is the same streaming
kernel and data-parallel training all-reduces one buffer.

Weight layout in HBM: the four channel-last layers (e1, e2, d1, d2) keep their weights TAPS-MAJOR in the flat buffers --
the matrix [rows, (ky, kx, c)] the coalesced gathers contract with -- and expose them to the host model as STRIDED views
of the reference's logical shape ([OC, IC, 4, 4] / [IC, OC, 4, 4]); `state_dict()` / `load_state_dict()` / checkpoints
see the reference's tensors, the kernels see the layout they want, and no weight or gradient is permuted per step
(round 1 re-permuted 8 weight matrices and 4 gradients every step).  e0 / d3 sit on the 3-channel NCHW model boundary
and keep the reference's (c, ky, kx) order.
"""
import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _lib
from . import functional as Fn
from ._lib import check, load, ptr, stream_ptr
from .functional import ComponentLayout

H_DIM = 8192  # conv_vae.py: the encoder output is 512*4*4


def _up64(n: int) -> int:
    return (n + 63) // 64 * 64


class ConvFlatLayout:

    def __init__(self, layout: ComponentLayout):
        NH, Z = layout.heads_dim, layout.z_dim
        self.comp_layout = layout
        o = _lib.RADII_REGION
        self.off: Dict[str, int] = {}

        def take(name, n):
            nonlocal o
            self.off[name] = o
            o += _up64(n)

        take("w_heads", NH * H_DIM)
        take("b_heads", NH)
        shapes = [("e0", (64, 3, 4, 4)), ("e1", (128, 64, 4, 4)), ("e2", (512, 128, 4, 4)), ("d0", (2048, Z)),
                  ("d1", (128, 256, 4, 4)), ("d2", (256, 64, 4, 4)), ("d3", (64, 3, 4, 4))]
        bias = {"e0": 64, "e1": 128, "e2": 512, "d0": 2048, "d1": 256, "d2": 64, "d3": 3}
        self.shapes = dict(shapes)
        for name, shp in shapes:
            n = 1
            for v in shp:
                n *= v
            take(name + ".weight", n)
            take(name + ".bias", bias[name])
        self.n_params = o
        self.entries: List[Tuple[str, int, Tuple[int, ...]]] = []
        radius_name = {"h": "_nradius", "p": "_nradius", "s": "_pradius", "d": "_pradius", "u": "_curvature"}
        for i, (letter, d) in enumerate(layout.comps):
            desc = layout.descs[i]
            pre = f"components.{i}."
            if letter in radius_name:
                self.entries.append((pre + radius_name[letter], i, ()))
            self.entries.append((pre + "fc_mean.weight", self.off["w_heads"] + desc.mean_col * H_DIM, (d, H_DIM)))
            self.entries.append((pre + "fc_mean.bias", self.off["b_heads"] + desc.mean_col, (d,)))
            self.entries.append((pre + "fc_logvar.weight", self.off["w_heads"] + desc.logvar_col * H_DIM,
                                 (desc.logvar_dim, H_DIM)))
            self.entries.append((pre + "fc_logvar.bias", self.off["b_heads"] + desc.logvar_col, (desc.logvar_dim,)))
        for name, shp in shapes:
            self.entries.append((name + ".weight", self.off[name + ".weight"], shp))
            self.entries.append((name + ".bias", self.off[name + ".bias"], (bias[name],)))

    TAPS_MAJOR = ("e1.weight", "e2.weight", "d1.weight", "d2.weight")

    def views(self, flat: Tensor) -> Dict[str, Tensor]:
        """Name -> tensor of the reference's logical shape aliasing `flat`.  The taps-major weights are strided views:
        element (r, c, ky, kx) of [R, Cc, 4, 4] lives at r * 16 Cc + (4 ky + kx) * Cc + c."""
        out = {}
        for name, off, shape in self.entries:
            n = 1
            for v in shape:
                n *= v
            if name in self.TAPS_MAJOR:
                Cc = shape[1]
                out[name] = flat[off:off + n].as_strided(shape, (16 * Cc, 1, 4 * Cc, Cc))
            else:
                out[name] = flat[off:off + n].view(shape)
        return out

    def matrix(self, flat: Tensor, name: str) -> Tensor:
        """The [rows, 16 * Cc] taps-major matrix of a channel-last layer's weight (or gradient), as stored."""
        shape = self.shapes[name]
        off = self.off[name + ".weight"]
        return flat[off:off + shape[0] * shape[1] * 16].view(shape[0], shape[1] * 16)


# ---- pre-split operands ("planes", csrc/mvae_p3.hip): tensor [rows, cols] f32 -> bfloat16 [3, rows, cols] (hi, mid, lo pieces)
def _new_planes(rows: int, cols: int, device) -> Tensor:
    return torch.empty(3, rows, cols, dtype=torch.bfloat16, device=device)


def _pptr(planes: Optional[Tensor]) -> Optional[int]:
    if planes is None:
        return None
    assert planes.is_cuda and planes.dtype == torch.bfloat16 and planes.is_contiguous() and planes.shape[0] == 3
    return planes.data_ptr()


def _ps(planes: Optional[Tensor]) -> int:
    return 0 if planes is None else planes[0].numel()


def _split_planes(tensors: Sequence[Tensor], outs: Optional[Sequence[Tensor]] = None, queue: bool = False) -> List[Tensor]:
    """Planes of up to 12 f32 tensors ([rows, cols], element count a multiple of 4) in ONE launch (mvae_split3_planes).
    queue: no launch of their own -- the jobs ride on the next fused latent forward as extra workgroups
    (mvae_split3_planes_queue; whoever reads planes first performs them otherwise)."""
    n = len(tensors)
    if outs is None:
        outs = [_new_planes(t.shape[0], t.numel() // t.shape[0], t.device) for t in tensors]
    src = (C.c_void_p * n)(*[ptr(t) for t in tensors])
    dst = (C.c_void_p * n)(*[_pptr(o) for o in outs])
    cnt = (C.c_int64 * n)(*[t.numel() for t in tensors])
    fn = load().mvae_split3_planes_queue if queue else load().mvae_split3_planes
    check(fn(n, src, dst, cnt, stream_ptr(tensors[0].device)))
    return list(outs)


class _p3_group:
    """with _p3_group(device): a layer's weight gradient and its backward-data (independent plane contractions) leave as ONE
    launch (mvae_p3_group); MVAE_P3_NO_PAIR=1: one after the other as before."""

    def __init__(self, device):
        self.stream = stream_ptr(device)

    def __enter__(self):
        check(load().mvae_p3_group(1, self.stream))

    def __exit__(self, *exc):
        check(load().mvae_p3_group(0, self.stream))
        return False


def _conv_nhwc_p3(src_p: Tensor, Wt_p: Tensor, mask: Optional[Tensor], B: int, Cc: int, IH: int,
                  want_planes: bool = False, bias: Optional[Tensor] = None, relu: bool = False,
                  out_planes: Optional[Tensor] = None, keep_slices: bool = False,
                  colsum_out: Optional[Tensor] = None) -> Tuple[Optional[Tensor], Optional[Tensor]]:
    """_conv_nhwc (a Conv2d forward with bias / relu, or the backward-data of a ConvTranspose2d with mask) on planes: src_p
    [3, B*IH*IH, Cc], Wt_p [3, OC, 16 Cc] -> (y [B*(IH/2)^2, OC] f32, its planes or None).  keep_slices: when the call cuts K
    into slices, return them un-added, [slices, M, OC], for a consumer that adds them while reading (the latent backward).
    colsum_out [OC] (with planes, unsliced): the column sums of y -- a bias gradient -- from the epilogue's per-tile partial
    sums (added by the deferred slice sum); the f32 y is then not written and None is returned for it."""
    OC = Wt_p.shape[1]
    M = B * (IH // 2) * (IH // 2)
    epilogue = mask is not None or bias is not None or relu
    nws = int(load().mvae_conv_k4s2p1_nhwc_p3_workspace_floats(B, Cc, IH, IH, OC, 1 if epilogue else 0))
    if keep_slices and nws > 0:
        ws = torch.empty(nws // (M * OC), M, OC, dtype=torch.float32, device=src_p.device)
        check(load().mvae_conv_k4s2p1_nhwc_p3(_pptr(src_p), _ps(src_p), _pptr(Wt_p), _ps(Wt_p), None, None, 0, None, None, 0,
                                              None, None, B, Cc, IH, IH, OC, ptr(ws), stream_ptr(ws.device)))
        return ws, None
    if colsum_out is not None and nws == 0 and (want_planes or out_planes is not None):
        yp = out_planes if out_planes is not None else _new_planes(M, OC, src_p.device)
        part = _keep(torch.empty(M // 128, OC, dtype=torch.float32, device=src_p.device))
        check(load().mvae_conv_k4s2p1_nhwc_p3(_pptr(src_p), _ps(src_p), _pptr(Wt_p), _ps(Wt_p), ptr(mask), ptr(bias),
                                              1 if relu else 0, None, _pptr(yp), _ps(yp), ptr(colsum_out), ptr(part), B, Cc, IH, IH,
                                              OC, None, stream_ptr(src_p.device)))
        return None, yp
    y = torch.empty(M, OC, dtype=torch.float32, device=src_p.device)
    ws = y.new_empty(nws) if nws > 0 else None  # split-K slices, added in index order right away (y is an intermediate)
    yp = out_planes if out_planes is not None else (_new_planes(M, OC, y.device) if (want_planes and nws == 0) else None)
    check(load().mvae_conv_k4s2p1_nhwc_p3(_pptr(src_p), _ps(src_p), _pptr(Wt_p), _ps(Wt_p), ptr(mask), ptr(bias), 1 if relu else 0,
                                          ptr(y), _pptr(yp), _ps(yp), None, None, B, Cc, IH, IH, OC, ptr(ws), stream_ptr(y.device)))
    return y, yp


def _gemm_nn_p3(G_p: Tensor, W_p: Tensor) -> Tensor:
    M, K = G_p.shape[1], G_p.shape[2]
    N = W_p.shape[2]
    out = torch.empty(M, N, dtype=torch.float32, device=G_p.device)
    check(load().mvae_gemm_nn_p3(_pptr(G_p), _ps(G_p), _pptr(W_p), _ps(W_p), ptr(out), M, K, N, stream_ptr(out.device)))
    return out


def _convT_nhwc_p3(src_p: Tensor, Wt_p: Tensor, mask: Optional[Tensor], B: int, Cc: int, IH: int, OC: int,
                   want_planes: bool = False, bias: Optional[Tensor] = None, relu: bool = False,
                   out_planes: Optional[Tensor] = None, colsum_out: Optional[Tensor] = None,
                   want_y: bool = True) -> Tuple[Optional[Tensor], Optional[Tensor]]:
    """_convT_nhwc (a ConvTranspose2d forward with bias / relu, or a Conv2d's backward-data with mask; four parity classes)
    on planes: src_p [3, B*IH*IH, Cc], Wt_p [3, Cc, 16 OC].  colsum_out [OC]: the column sums of the result (a bias gradient)
    from the epilogue's per-tile partial sums; want_y=False (with planes and colsum_out): the f32 result is not written."""
    M = B * (2 * IH) * (2 * IH)
    yp = out_planes if out_planes is not None else (_new_planes(M, OC, src_p.device) if want_planes else None)
    cws = None
    if colsum_out is not None:
        cws = _keep(torch.empty(int(load().mvae_conv_transpose_k4s2p1_nhwc_p3_colsum_floats(B, IH, IH, OC)),
                                dtype=torch.float32, device=src_p.device))
    y = torch.empty(M, OC, dtype=torch.float32, device=src_p.device) if (want_y or cws is None or yp is None) else None
    check(load().mvae_conv_transpose_k4s2p1_nhwc_p3(_pptr(src_p), _ps(src_p), _pptr(Wt_p), _ps(Wt_p), ptr(mask), ptr(bias),
                                                    1 if relu else 0, ptr(y), _pptr(yp), _ps(yp),
                                                    ptr(colsum_out) if cws is not None else None, ptr(cws), B, Cc, IH, IH, OC,
                                                    stream_ptr(src_p.device)))
    return y, yp


def _conv_nhwc_wgrad_p3(dy_p: Tensor, src_p: Tensor, out: Tensor, B: int, Cc: int, IH: int) -> Tensor:
    """_conv_nhwc_wgrad on planes: dy_p [3, B*(IH/2)^2, OC], src_p [3, B*IH*IH, Cc] -> out [OC, 16 Cc] (taps-major)."""
    OC = dy_p.shape[2]
    assert out.is_contiguous() and out.numel() == OC * 16 * Cc
    nws = int(load().mvae_conv_k4s2p1_nhwc_wgrad_p3_workspace_floats(B, Cc, IH, IH, OC))
    ws = _keep(out.new_empty(nws)) if nws > 0 else None
    check(load().mvae_conv_k4s2p1_nhwc_wgrad_p3(_pptr(dy_p), _ps(dy_p), _pptr(src_p), _ps(src_p), ptr(out), B, Cc, IH, IH, OC,
                                                ptr(ws), stream_ptr(out.device)))
    return out


def _im2col(src: Tensor, mask: Optional[Tensor], B: int, Cc: int, IH: int, strides, taps_major: bool = False) -> Tensor:
    col = src.new_empty(B * (IH // 2) * (IH // 2), Cc * 16)
    check(load().mvae_im2col_k4s2p1(ptr(src), ptr(mask), ptr(col), B, Cc, IH, IH, *strides, 1 if taps_major else 0,
                                    stream_ptr(src.device)))
    return col


def _col2im(col: Tensor, bias: Optional[Tensor], mask: Optional[Tensor], B: int, Cc: int, Hh: int, strides, relu: bool,
            out_shape, taps_major: bool = False, planes: Optional[Tensor] = None) -> Tensor:
    """planes (taps-major only): bfloat16 [3, rows, Cc] receiving the bf16 planes of the result (csrc/mvae_p3.hpp)."""
    dst = col.new_empty(out_shape)
    check(load().mvae_col2im_k4s2p1(ptr(col), ptr(bias), ptr(mask), ptr(dst), B, Cc, Hh, Hh, *strides,
                                    1 if relu else 0, 1 if taps_major else 0, _pptr(planes), _ps(planes),
                                    stream_ptr(col.device)))
    return dst


def _nhwc(Hh: int, Cc: int):
    return (Hh * Hh * Cc, 1, Hh * Cc, Cc)  # (sb, sc, sy, sx)


def _nchw(Hh: int, Cc: int):
    return (Cc * Hh * Hh, Hh * Hh, Hh, 1)


def _permute_rc(x: Tensor, B: int, R: int, Cc: int, out: Optional[Tensor] = None) -> Tensor:
    """out[b][c][r] = x[b][r][c]."""
    if out is None:
        out = torch.empty_like(x)
    assert out.is_contiguous() and out.numel() == x.numel()
    check(load().mvae_permute_rc(ptr(x), ptr(out), B, R, Cc, stream_ptr(x.device)))
    return out


def _linear_splitk(x: Tensor, W: Tensor, b: Optional[Tensor]) -> Tensor:
    """x W^T + b for few rows and a long contraction (the heads on the 8192-wide flatten)."""
    M, K = x.shape
    N = W.shape[0]
    y = x.new_empty(M, N)
    nws = load().mvae_linear_forward_splitk_workspace_floats(M, N, K)
    ws = x.new_empty(int(nws)) if nws > 0 else None
    check(load().mvae_linear_forward_splitk(ptr(x), ptr(W), ptr(b), ptr(y), M, N, K, 0, ptr(ws), stream_ptr(x.device)))
    return y


def _taps_major(W: Tensor, rows: int, Cc: int) -> Tensor:
    """[rows, Cc, 16] weight block (Conv2d: [OC, IC, 4, 4]; ConvTranspose2d: [IC, OC, 4, 4]) -> [rows, 16 * Cc] with
    the patch axis ordered (ky, kx, c): what the taps-major gathers of the channel-last layers contract with."""
    return _permute_rc(W.reshape(rows, Cc, 16), rows, Cc, 16).view(rows, 16 * Cc)


def _from_taps_major(dWt: Tensor, rows: int, Cc: int, out: Tensor) -> Tensor:
    """Inverse of _taps_major for a gradient, written straight into its slot of the flat gradient buffer."""
    return _permute_rc(dWt.reshape(rows, 16, Cc), rows, 16, Cc, out=out)


# Workspaces whose final slice sum is queued (mvae_slice_sums_defer): alive until the flush.
_DEFERRED_WS: list = []


def _keep(ws: Optional[Tensor]) -> Optional[Tensor]:
    if ws is not None:
        _DEFERRED_WS.append(ws)
    return ws


def _gemm_tn(P: Tensor, Q: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """out[NP, NQ] = P^T Q; `out` may be a (contiguous) view into the flat gradient buffer."""
    M, NP = P.shape
    NQ = Q.shape[1]
    if out is None:
        out = P.new_empty(NP, NQ)
    assert out.is_contiguous() and out.numel() == NP * NQ
    nws = load().mvae_gemm_tn_workspace_floats(M, NP, NQ)
    ws = _keep(P.new_empty(int(nws))) if nws > 0 else None
    check(load().mvae_gemm_tn(ptr(P), ptr(Q), ptr(out), M, NP, NQ, ptr(ws), stream_ptr(P.device)))
    return out


_EDGE_DIRECT = os.environ.get("MVAE_CONV_EDGE_DIRECT", "1") != "0"  # (read once, at import)


def _edge_direct() -> bool:
    """The 3-channel boundary layers straight from the image (csrc/mvae_edge.hip, the default) or through the patch matrix
    (MVAE_CONV_EDGE_DIRECT=0: mvae_im2col_k4s2p1 + generic contractions; same bits on the activation side)."""
    return _EDGE_DIRECT


def _edge_conv(img: Tensor, W: Tensor, bias: Optional[Tensor], mask: Optional[Tensor], relu: bool, B: int,
               planes: Optional[Tensor] = None) -> Tensor:
    """[B*256, 64] = mask(relu(patches(img [B, 3, 32, 32]) W[64, 48]^T + bias)): e0 forward / d3 backward-data, no patch matrix."""
    y = img.new_empty(B * 256, 64)
    check(load().mvae_conv3_k4s2p1_nchw(ptr(img), ptr(W), ptr(bias), ptr(mask), 1 if relu else 0, ptr(y), _pptr(planes),
                                        _ps(planes), B, 3, 32, 32, 64, stream_ptr(img.device)))
    return y


def _edge_wgrad(act: Tensor, img: Tensor, out: Tensor, B: int) -> Tensor:
    """out[64, 48] = act[B*256, 64]^T patches(img [B, 3, 32, 32]): the weight gradient of e0 / d3, no patch matrix."""
    assert out.is_contiguous() and out.numel() == 64 * 48
    nws = int(load().mvae_conv3_k4s2p1_nchw_wgrad_workspace_floats(B, 3, 32, 32, 64))
    ws = _keep(act.new_empty(nws))
    check(load().mvae_conv3_k4s2p1_nchw_wgrad(ptr(act), ptr(img), ptr(out), B, 3, 32, 32, 64, ptr(ws), stream_ptr(act.device)))
    return out


def _edge_backward(act: Tensor, img: Tensor, W: Tensor, out_dW: Tensor, B: int, planes: Optional[Tensor] = None,
                   colsum_out: Optional[Tensor] = None) -> Optional[Tensor]:
    """_edge_wgrad(act, img, out_dW) and _edge_conv(img, W, None, act, False) in ONE launch: the backward pass of d3.
    colsum_out [64] (with planes): the column sums of the backward-data result (the bias gradient of the layer below) from
    per-workgroup partial sums; the f32 result is then not written and None is returned."""
    assert out_dW.is_contiguous() and out_dW.numel() == 64 * 48
    nws = int(load().mvae_conv3_k4s2p1_nchw_wgrad_workspace_floats(B, 3, 32, 32, 64))
    ws = _keep(act.new_empty(nws))
    cws = None
    if colsum_out is not None and planes is not None:
        cws = _keep(act.new_empty(int(load().mvae_conv3_k4s2p1_nchw_backward_colsum_floats(B))))
    y = None if cws is not None else img.new_empty(B * 256, 64)
    check(load().mvae_conv3_k4s2p1_nchw_backward(ptr(act), ptr(img), ptr(W), ptr(out_dW), ptr(y), _pptr(planes), _ps(planes),
                                                 ptr(colsum_out) if cws is not None else None, ptr(cws), B, 3, 32, 32, 64, ptr(ws),
                                                 stream_ptr(act.device)))
    return y


FORWARD, BACKWARD = 0, 1  # MVAE_PASS_FORWARD / MVAE_PASS_BACKWARD: which pass a shared contraction belongs to


def _gemm_nn(G: Tensor, W: Tensor, pass_: int = FORWARD) -> Tensor:
    M, K = G.shape
    N = W.shape[1]
    out = G.new_empty(M, N)
    check(load().mvae_gemm_nn(ptr(G), ptr(W), None, ptr(out), M, K, N, pass_, stream_ptr(G.device)))
    return out


def _colsum(G: Tensor, out: Optional[Tensor] = None) -> Tensor:
    if out is None:
        out = G.new_empty(G.shape[1])
    assert out.is_contiguous() and out.numel() == G.shape[1]
    nws = load().mvae_colsum_workspace_floats(G.shape[0], G.shape[1])
    ws = _keep(G.new_empty(int(nws))) if nws > 0 else None
    if nws > 0:
        _keep(G)  # while deferral is on the column sum itself is queued: its input lives until the flush
    check(load().mvae_colsum(ptr(G), ptr(out), G.shape[0], G.shape[1], ptr(ws), stream_ptr(G.device)))
    return out


def _conv_nhwc(src: Tensor, Wt: Tensor, bias: Optional[Tensor], mask: Optional[Tensor], B: int, Cc: int, IH: int,
               relu: bool, pass_: int = FORWARD, planes: Optional[Tensor] = None) -> Tensor:
    """Implicit contraction (no patch matrix): src [B*IH*IH, Cc] channel-last -> [B*(IH/2)^2, OC]; Wt [OC, 16 Cc]
    taps-major.  Conv2d forward, or ConvTranspose2d backward-data with `mask` = the previous ReLU's output."""
    OC = Wt.shape[0]
    y = src.new_empty(B * (IH // 2) * (IH // 2), OC)
    nws = int(load().mvae_conv_k4s2p1_nhwc_workspace_floats(B, Cc, IH, IH, OC, 0 if mask is None else 1))
    ws = src.new_empty(nws) if nws > 0 else None  # split-K slices of a layer with few output tiles
    check(load().mvae_conv_k4s2p1_nhwc(ptr(src), ptr(Wt), ptr(bias), ptr(mask), ptr(y), B, Cc, IH, IH, OC,
                                       1 if relu else 0, ptr(ws), pass_, _pptr(planes), _ps(planes),
                                       stream_ptr(src.device)))
    return y


def _convT_nhwc(src: Tensor, Wt: Tensor, bias: Optional[Tensor], mask: Optional[Tensor], B: int, Cc: int, IH: int,
                OC: int, relu: bool, pass_: int = FORWARD, planes: Optional[Tensor] = None) -> Tensor:
    """Transposed convolution as four implicit contractions (one per output parity class; no [M, 16 OC] product, no
    col2im): src [B*IH*IH, Cc] channel-last -> [B*(2 IH)^2, OC]; Wt [Cc, 16 OC] with columns (ky, kx, oc).
    ConvTranspose2d forward, or Conv2d backward-data with `mask` = the previous ReLU's output."""
    y = src.new_empty(B * (2 * IH) * (2 * IH), OC)
    check(load().mvae_conv_transpose_k4s2p1_nhwc(ptr(src), ptr(Wt), ptr(bias), ptr(mask), ptr(y), B, Cc, IH, IH, OC,
                                        1 if relu else 0, pass_, _pptr(planes), _ps(planes), stream_ptr(src.device)))
    return y


def _convT_to3(src: Tensor, W48: Tensor, bias: Tensor, R: int) -> Tensor:
    """Direct ConvTranspose2d(64 -> 3, k4 s2 p1): channel-last [R * 256, 64] -> NCHW logits [R, 3072]."""
    y = src.new_empty(R, 3072)
    check(load().mvae_convt_to3_k4s2p1_forward(ptr(src), ptr(W48), ptr(bias), ptr(y), R, 64, 16, 16, 3,
                                               stream_ptr(src.device)))
    return y


def _conv_e2(a1: Tensor, We2: Tensor, bias: Tensor, B: int) -> Tensor:
    """The e2 layer forward (128 -> 512 channels on 8 x 8): the implicit contraction (operand gather in the LDS-DMA requests
    of k_gemm_f32pp).  (Until round 4 the patch matrix + plain contraction was faster on the register-staged kernel, 12 + 83 us
    against 105 us, and stayed selectable; that form is gone.)"""
    return _conv_nhwc(a1, We2, bias, None, B, 128, 8, True)


def _conv_nhwc_wgrad(dy: Tensor, src: Tensor, out: Tensor, B: int, Cc: int, IH: int) -> Tensor:
    """out[OC, 16 Cc] (taps-major, e.g. a slot of the flat gradient buffer) = dy^T im2col(src), patch matrix implicit."""
    OC = dy.shape[1]
    assert out.is_contiguous() and out.numel() == OC * 16 * Cc
    nws = int(load().mvae_conv_k4s2p1_nhwc_wgrad_workspace_floats(B, Cc, IH, IH, OC))
    ws = _keep(dy.new_empty(nws)) if nws > 0 else None
    check(load().mvae_conv_k4s2p1_nhwc_wgrad(ptr(dy), ptr(src), ptr(out), B, Cc, IH, IH, OC, ptr(ws),
                                             stream_ptr(dy.device)))
    return out


def _linear_masked(x: Tensor, W: Tensor, mask: Tensor, planes: Optional[Tensor] = None) -> Tensor:
    """(x W^T) zeroed where mask <= 0: a Linear backward-data with the previous ReLU's mask applied in the epilogue."""
    M, K = x.shape
    N = W.shape[0]
    y = x.new_empty(M, N)
    check(load().mvae_linear_forward_masked(ptr(x), ptr(W), ptr(mask), ptr(y), M, N, K, _pptr(planes), _ps(planes),
                                            stream_ptr(x.device)))
    return y


def _linear_forward_planes(x: Tensor, W: Tensor, b: Optional[Tensor], relu: bool, planes: Tensor) -> Tensor:
    """Fn.linear_forward on the LDS-tiled kernel with the result's planes written by the epilogue."""
    M, K = x.shape
    N = W.shape[0]
    y = x.new_empty(M, N)
    check(load().mvae_linear_forward_planes(ptr(x), ptr(W), ptr(b), ptr(y), _pptr(planes), _ps(planes), M, N, K,
                                            1 if relu else 0, stream_ptr(x.device)))
    return y


def _relu_mask_(dy: Tensor, y: Tensor) -> Tensor:
    check(load().mvae_relu_mask(ptr(dy), ptr(y), dy.numel(), stream_ptr(dy.device)))
    return dy


class ConvEngine:
    in_dim = 3072

    def __init__(self, comps: Sequence[Tuple[str, int]], device, scalar_parametrization: bool = False,
                 radius_trainable: Optional[Sequence[bool]] = None, lr: float = 1e-3, curvature_lr: float = 1e-4):
        load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.MvaeHipError("ConvEngine needs a HIP device: the product path has no CPU fallback")
        self.layout = ComponentLayout(comps, scalar_parametrization)
        self.flat = ConvFlatLayout(self.layout)
        self.lr, self.curvature_lr = float(lr), float(curvature_lr)
        n = self.layout.n
        tr = [False] * n if radius_trainable is None else [bool(t) for t in radius_trainable]
        self.radius_trainable = [t and letter != "e" for t, (letter, _) in zip(tr, self.layout.comps)]
        self._trainable_arr = (C.c_uint8 * n)()
        self.set_radius_trainable(self.radius_trainable)
        P = self.flat.n_params
        z = lambda k, dt=torch.float32: torch.zeros(k, dtype=dt, device=self.device)  # noqa: E731
        self.params, self.grads, self.adam_m, self.adam_v = z(P), z(P), z(P), z(P)
        self.counters = z(32, torch.int32)
        self.stats = z(3 * (4 + n))  # sums | last step | Kahan compensation of the sums
        self._arrive = z(32, torch.int32)  # arrival counters of mvae_conv_bce_stats (0 between launches)
        self.generation = 0  # bumped by set_lr: graph caches are keyed on it (like StepEngine's)
        # Switches (environment, read once per engine; DESIGN section 4, conv architecture):
        # MVAE_CONV_FUSED (default 1): the loss end as one launch (mvae_conv_bce_stats), the last transposed convolution
        #   direct (mvae_convt_to3_k4s2p1_forward) and -- where the model's shapes fit, mvae_conv_latent_supported -- the
        #   latent section (flatten -> heads -> components -> decoder fc) as 2 + 2 fused launches; 0: the generic operators.
        self.direct = os.environ.get("MVAE_CONV_FUSED", "1") != "0"
        self.fused = self.direct and bool(load().mvae_conv_latent_supported(self.layout.descs, n))
        # MVAE_CONV_SPLIT_BF16 (default: leave the library's mode alone = 2): how the large contractions multiply
        #   (process-wide, mvae_set_contraction_mode).  2: the backward pass through exact three-way bf16 splits on the bf16
        #   MFMA, the forward pass -- whose outputs decide the ReLU masks -- on the exact f32-input MFMA; 1: split products
        #   everywhere; 0: the f32-input MFMA everywhere.
        if os.environ.get("MVAE_CONV_SPLIT_BF16", "") in ("0", "1", "2"):
            load().mvae_set_contraction_mode(int(os.environ["MVAE_CONV_SPLIT_BF16"]))
        # MVAE_CONV_PLANES (default 1): in contraction mode 2 the backward pass runs on PRE-SPLIT operands (csrc/mvae_p3.hip):
        #   the forward epilogues write the bf16 planes of the activations next to them, the backward contractions stage the
        #   planes by LDS-DMA; 0: the split happens inside the backward kernels (k_gemm_b3).  Same arithmetic either way.
        self.planes = os.environ.get("MVAE_CONV_PLANES", "1") != "0"
        # A/B switches of the plane backward pass, read ONCE here (INTEGRATION.md lists every surviving switch):
        # MVAE_CONV_D3_FUSED (1): the last transposed convolution inside the loss-end launch; MVAE_CONV_EPI_COLSUM (1; "2": db1
        # only): bias gradients from the producing launch's epilogue; MVAE_CONV_DT0_SLICES (1): dt0's K slices added by the
        # latent backward; MVAE_CONV_DA1_IMPLICIT (1): da1 as four implicit parity-class contractions
        self._sw_d3_fused = os.environ.get("MVAE_CONV_D3_FUSED", "1") != "0"
        self._sw_epi_colsum = os.environ.get("MVAE_CONV_EPI_COLSUM", "1")
        self._sw_dt0_slices = os.environ.get("MVAE_CONV_DT0_SLICES", "1") != "0"
        self._sw_da1_implicit = os.environ.get("MVAE_CONV_DA1_IMPLICIT", "1") != "0"
        self._sw_split_ride = os.environ.get("MVAE_SPLIT_RIDE", "1") != "0"  # weight planes ride on the latent forward
        # (Rounds 3-4 also carried a backward pass on three HIP streams -- weight gradients and bias sums forked onto side
        # streams --: same bits, measured slower in every form (1.08 -> 1.21 ms eager, no overlap inside a captured graph);
        # removed in round 5.)

    def set_radius_trainable(self, radius_trainable: Sequence[bool]) -> None:
        """0 fixed / 1 trainable radius / 3 trainable universal curvature (clip group), see mvae_optimizer_step_flat."""
        self.radius_trainable = [bool(t) and letter != "e" for t, (letter, _) in zip(radius_trainable, self.layout.comps)]
        for i, (t, (letter, _)) in enumerate(zip(self.radius_trainable, self.layout.comps)):
            self._trainable_arr[i] = (3 if letter == "u" else 1) if t else 0

    # ---- state (same surface as StepEngine)
    def param_views(self) -> Dict[str, Tensor]:
        return self.flat.views(self.params)

    def grad_views(self) -> Dict[str, Tensor]:
        return self.flat.views(self.grads)

    def load_state(self, state: Dict[str, Tensor]) -> None:
        for name, v in self.param_views().items():
            v.copy_(state[name].to(device=self.device, dtype=torch.float32).reshape(v.shape))

    def set_radii(self, value: float) -> None:
        for i, (letter, _) in enumerate(self.layout.comps):
            if letter in ("h", "p", "s", "d"):  # not the universal curvature, train.py:189-194
                self.params[i] = value

    def set_lr(self, lr: float, curvature_lr: Optional[float] = None) -> None:
        self.lr = float(lr)
        if curvature_lr is not None:
            self.curvature_lr = float(curvature_lr)
        self.generation += 1  # captured graphs carry the learning rates in their kernel arguments (runner.EpochRunner)

    def _w(self, name: str) -> Tensor:
        return self.param_views()[name]

    def _use_p3(self, B: int) -> int:
        """Which passes of a B-row step run on pre-split operands: 0 none, 2 the backward pass (contraction mode 2), 1 both
        passes (contraction mode 1: split products everywhere).  Needs the switch on and every plane
        contraction's shape made of whole tiles."""
        mode = load().mvae_set_contraction_mode(-1)
        if not self.planes or mode not in (1, 2):
            return 0
        sup = load().mvae_p3_supported
        ok = bool(sup(0, B * 64, 256, 1024, 64) and sup(0, B * 16, 128, 4096, 256) and sup(1, B * 16, 2048, 512, 0) and
                  sup(2, B * 64, 64, 512, 128) and sup(3, B * 64, 256, 1024, 64) and sup(3, B * 16, 128, 4096, 256) and
                  sup(3, B * 16, 512, 2048, 128) and sup(3, B * 64, 128, 1024, 64) and B * 256 >= 512)
        if ok and mode == 1:  # the four forward layers
            ok = bool(sup(0, B * 64, 128, 1024, 64) and sup(0, B * 16, 512, 2048, 128) and sup(2, B * 16, 256, 512, 128) and
                      sup(2, B * 64, 64, 1024, 256))
        return mode if ok else 0

    # ---- forward (keeps what backward needs)
    def _forward(self, x: Tensor, eps: Tensor, want_kl: bool = True, planes: bool = False, planes_forward: bool = False,
                 defer_logits: bool = False):
        """planes: write the bf16 planes the backward pass on pre-split operands reads; planes_forward (with planes, fused
        latent section): the four channel-last layers themselves run on planes (contraction mode 1)."""
        PV = self.param_views()
        lay = self.layout
        B = x.shape[0]
        NH = lay.heads_dim
        c = {}
        # e0 / d3 touch the 3-channel NCHW boundary and keep the weight layout's (c,ky,kx) patch order; the four
        # channel-last layers in between run taps-major (coalesced gathers) against permuted weight matrices
        c["We1"], c["We2"] = self.flat.matrix(self.params, "e1"), self.flat.matrix(self.params, "e2")
        c["Wd1"], c["Wd2"] = self.flat.matrix(self.params, "d1"), self.flat.matrix(self.params, "d2")
        c["x"] = x
        c["col0"] = None if _edge_direct() else _im2col(x, None, B, 3, 32, _nchw(32, 3))
        if planes:
            # the backward pass will run on pre-split operands: the epilogues below write the bf16 planes of the activations
            # its weight gradients gather (a0, a1) or contract with (t0, b1) next to the f32 tensors
            c["a0_p"], c["a1_p"] = _new_planes(B * 256, 64, self.device), _new_planes(B * 64, 128, self.device)
        if c["col0"] is None:
            c["a0"] = _edge_conv(x, PV["e0.weight"].view(64, 48), PV["e0.bias"], None, True, B, c.get("a0_p"))
        elif planes:
            c["a0"] = _linear_forward_planes(c["col0"], PV["e0.weight"].view(64, 48), PV["e0.bias"], True, c["a0_p"])
        else:
            c["a0"] = Fn.linear_forward(c["col0"], PV["e0.weight"].view(64, 48), PV["e0.bias"], relu=True)
        planes_forward = planes_forward and planes and self.fused and want_kl and eps.dim() == 2
        if planes_forward:
            # planes of the four weight matrices from the CURRENT parameters (one launch; the backward pass reuses them)
            c["W_p"] = _split_planes([c["We1"], c["We2"], c["Wd1"], c["Wd2"]])
            c["a1"], _ = _conv_nhwc_p3(c["a0_p"], c["W_p"][0], None, B, 64, 16, bias=PV["e1.bias"], relu=True, out_planes=c["a1_p"])
            c["a2"], _ = _conv_nhwc_p3(c["a1_p"], c["W_p"][1], None, B, 128, 8, bias=PV["e2.bias"], relu=True)
        else:
            c["a1"] = _conv_nhwc(c["a0"], c["We1"], PV["e1.bias"], None, B, 64, 16, True, FORWARD, c.get("a1_p"))   # [B*64, 128]
            c["a2"] = _conv_e2(c["a1"], c["We2"], PV["e2.bias"], B)   # [B*16, 512]
        # The reference flattens NCHW (conv_vae.py:65: column c * 16 + p of the head matrices); the activation here is
        # channel-last (column p * 512 + c).  Re-ordering the head matrix (NH x 8192: 0.4 MB) instead of the activation
        # and its gradient (8 MB each) gives the same products.
        c["hflat"] = c["a2"].view(B, H_DIM)
        if self.fused and want_kl and eps.dim() == 2:
            ow, ob = self.flat.off["w_heads"], self.flat.off["b_heads"]
            c["heads"] = x.new_empty(B, NH)
            c["z"], c["kl"] = x.new_empty(B, lay.z_dim), x.new_empty(lay.n, B)
            c["t0"] = x.new_empty(B * 16, 128)
            if planes:
                c["t0_p"] = _new_planes(B * 16, 128, self.device)  # written by the same launch
            ws = x.new_empty(int(load().mvae_conv_latent_workspace_floats(B, lay.n)))
            eps = eps.contiguous()
            if planes and "W_p" not in c and self._sw_split_ride:
                # the weight planes of the backward pass (contraction mode 2) as extra workgroups of the latent forward: a launch
                # of one small workgroup per batch row that leaves the memory system idle (no k_split3 launch in the backward)
                c["W_p"] = _split_planes([c["We1"], c["We2"], c["Wd1"], c["Wd2"]], queue=True)
            check(load().mvae_conv_latent_forward(lay.descs, lay.n, ptr(c["hflat"]), ptr(self.params[ow:ow + NH * H_DIM]),
                                                  ptr(self.params[ob:ob + NH]), ptr(eps), eps.shape[1],
                                                  ptr(self.params[:lay.n]), ptr(PV["d0.weight"]), ptr(PV["d0.bias"]),
                                                  ptr(c["heads"]), ptr(c["z"]), ptr(c["kl"]), ptr(c["t0"]), _pptr(c.get("t0_p")),
                                                  _ps(c.get("t0_p")), ptr(ws), B, stream_ptr(self.device)))
            c["co"], c["fused"] = None, True
            R = B
        else:
            c["w_heads_cl"], b_heads = self._heads_channel_last()
            c["heads"] = _linear_splitk(c["hflat"], c["w_heads_cl"], b_heads)
            co = Fn.component_forward(lay, c["heads"], eps, self.params[:lay.n], want_kl=want_kl,
                                      want_log_probs=not want_kl, want_params=False)
            c["z"], c["kl"], c["co"] = co["z"], co["kl"], co
            zz = co["z"].reshape(-1, lay.z_dim)
            R = zz.shape[0]
            c["d0o"] = Fn.linear_forward(zz, PV["d0.weight"], PV["d0.bias"], relu=True)  # [R, 2048] = [R,128,4,4]
            c["t0"] = _permute_rc(c["d0o"], R, 128, 16).view(R * 16, 128)  # channel-last rows
        if planes:
            c["b1_p"] = _new_planes(R * 64, 256, self.device)
        if planes_forward:
            c["b1"], _ = _convT_nhwc_p3(c["t0_p"], c["W_p"][2], None, R, 128, 4, 256, bias=PV["d1.bias"], relu=True,
                                        out_planes=c["b1_p"])
            c["b2"], _ = _convT_nhwc_p3(c["b1_p"], c["W_p"][3], None, R, 256, 8, 64, bias=PV["d2.bias"], relu=True)
        else:
            c["b1"] = _convT_nhwc(c["t0"], c["Wd1"], PV["d1.bias"], None, R, 128, 4, 256, True, FORWARD, c.get("b1_p"))   # [R*64, 256]
            c["b2"] = self._d2_forward(c["b1"], c["Wd2"], PV["d2.bias"], R)
        if self.direct and defer_logits and R == B:
            c["logits"] = None  # forward_backward computes them in the launch of the loss end (mvae_convt_to3_bce_stats)
        elif self.direct:
            c["logits"] = _convT_to3(c["b2"], PV["d3.weight"].view(64, 48), PV["d3.bias"], R)
        else:
            c["cT3"] = _gemm_nn(c["b2"], PV["d3.weight"].view(64, 3 * 16))
            c["logits"] = _col2im(c["cT3"], PV["d3.bias"], None, R, 3, 32, _nchw(32, 3), False, (R, 3072))
        return c

    @staticmethod
    def _d2_forward(b1: Tensor, Wd2: Tensor, bias: Tensor, R: int) -> Tensor:
        """ConvTranspose2d(256 -> 64) + ReLU (conv_vae.py:53,73): [R * 64, 256] -> [R * 256, 64].  Four implicit contractions,
        one per output parity class (no [R * 64, 1024] product, no col2im): 0.86 -> 0.84 ms per step once the ping-pong
        kernel took the gathers (the product + col2im form it replaced is gone)."""
        return _convT_nhwc(b1, Wd2, bias, None, R, 256, 8, 64, True)

    def _heads_channel_last(self):
        """(W_heads with its 8192 columns re-ordered from the reference's (c, y, x) to channel-last (y, x, c), b_heads)."""
        NH = self.layout.heads_dim
        ow, ob = self.flat.off["w_heads"], self.flat.off["b_heads"]
        w = self.params[ow:ow + NH * H_DIM]
        return _permute_rc(w.view(NH, 512, 16), NH, 512, 16).view(NH, H_DIM), self.params[ob:ob + NH]

    def encode_heads(self, x: Tensor) -> Tensor:
        """conv_vae.py:57-66 + the fused head matrix: x[B,3072] -> heads[B, NH]."""
        PV = self.param_views()
        B, NH = x.shape[0], self.layout.heads_dim
        a0 = Fn.linear_forward(_im2col(x, None, B, 3, 32, _nchw(32, 3)), PV["e0.weight"].view(64, 48), PV["e0.bias"],
                               relu=True)
        a1 = _conv_nhwc(a0, self.flat.matrix(self.params, "e1"), PV["e1.bias"], None, B, 64, 16, True)
        a2 = _conv_e2(a1, self.flat.matrix(self.params, "e2"), PV["e2.bias"], B)
        w_heads_cl, b_heads = self._heads_channel_last()
        return _linear_splitk(a2.view(B, H_DIM), w_heads_cl, b_heads)

    def decode(self, z: Tensor) -> Tensor:
        """[..., B, Z] -> [..., B, 3072] (conv_vae.py:68-79)."""
        PV = self.param_views()
        zz = z.reshape(-1, self.layout.z_dim).contiguous()
        R = zz.shape[0]
        d0o = Fn.linear_forward(zz, PV["d0.weight"], PV["d0.bias"], relu=True)
        t0 = _permute_rc(d0o, R, 128, 16).view(R * 16, 128)
        b1 = _convT_nhwc(t0, self.flat.matrix(self.params, "d1"), PV["d1.bias"], None, R, 128, 4, 256, True)
        b2 = self._d2_forward(b1, self.flat.matrix(self.params, "d2"), PV["d2.bias"], R)
        if self.direct:
            lo = _convT_to3(b2, PV["d3.weight"].view(64, 48), PV["d3.bias"], R)
        else:
            lo = _col2im(_gemm_nn(b2, PV["d3.weight"].view(64, 48)), PV["d3.bias"], None, R, 3, 32, _nchw(32, 3), False,
                         (R, 3072))
        return lo.view(z.shape[:-1] + (3072,))

    def forward_backward(self, x: Tensor, eps: Tensor, beta: float = 1.0, want_outputs: bool = False):
        x = x.contiguous()
        B = x.shape[0]
        lay = self.layout
        p3 = self._use_p3(B) if eps.dim() == 2 else 0
        use_p3 = p3 != 0
        fuse_d3 = self.direct and self._sw_d3_fused
        c = self._forward(x, eps, planes=use_p3, planes_forward=(p3 == 1), defer_logits=fuse_d3)
        c["p3"] = use_p3
        bce = x.new_empty(B)
        # From here to the end of the backward pass the final "add the slices" of every weight gradient / bias column sum, and
        # the tail of the loss end (d3.bias, batch statistics), are queued and performed by ONE launch at the end (nobody reads
        # them before the optimizer).
        _DEFERRED_WS.clear()
        check(load().mvae_slice_sums_defer(1))
        try:
            g = self._loss_end(x, c, bce, beta, B, lay)
            return self._backward(x, eps, beta, want_outputs, c, g, bce, self.param_views(), self.grad_views(), B, lay)
        except BaseException:
            # an aborted pass may have left arrival counters of mvae_conv_bce_stats non-zero (the kernel re-arms them
            # itself only when it completes): without this no workgroup of the next step would ever see itself as the
            # last one, and d3.bias / the batch statistics would silently stop updating
            try:
                self._arrive.zero_()
            except Exception:  # noqa: BLE001  (a dead device: the original exception is the one to report)
                pass
            raise
        finally:
            # (planes queued by the forward pass and never consumed -- an aborted pass: performed while their tensors are alive)
            load().mvae_split3_planes_flush(stream_ptr(self.device))
            check(load().mvae_slice_sums_defer(0))

    def _loss_end(self, x, c, bce, beta, B, lay):
        """BCE, its gradient g, the batch statistics and d3.bias (the last two queued with the deferred sums)."""
        if c["logits"] is None:
            # the last transposed convolution, BCE + its gradient, the batch statistics and d3.bias in ONE launch
            PV = self.param_views()
            c["logits"], g, chan = x.new_empty(B, 3072), x.new_empty(B, 3072), _keep(x.new_empty(B, 3))  # (read at the flush)
            check(load().mvae_convt_to3_bce_stats(ptr(c["b2"]), ptr(PV["d3.weight"]), ptr(PV["d3.bias"]), ptr(x), ptr(c["logits"]),
                                                  ptr(bce), ptr(g), ptr(c["kl"]), ptr(self.stats), float(beta), B, 64, 16, 16, 3,
                                                  lay.n, ptr(chan), ptr(self.grad_views()["d3.bias"]), ptr(self._arrive),
                                                  stream_ptr(self.device)))
            c["d3_bias_done"] = True
        elif self.direct:
            g = torch.empty_like(c["logits"])
            # BCE + its gradient, the batch statistics and the bias gradient of d3 (sum of g per channel) in one launch
            chan = _keep(x.new_empty(B, 3))  # (read at the flush)
            check(load().mvae_conv_bce_stats(ptr(c["logits"]), ptr(x), ptr(bce), ptr(g), ptr(c["kl"]), ptr(self.stats),
                                             float(beta), B, 3072, 1024, lay.n, ptr(chan),
                                             ptr(self.grad_views()["d3.bias"]), ptr(self._arrive),
                                             stream_ptr(self.device)))
            c["d3_bias_done"] = True
        else:
            g = torch.empty_like(c["logits"])
            check(load().mvae_bce_forward_backward(ptr(c["logits"]), ptr(x), ptr(bce), ptr(g), B, 3072,
                                                   stream_ptr(self.device)))
            check(load().mvae_batch_stats(ptr(bce), ptr(c["kl"]), ptr(self.stats), float(beta), B, lay.n,
                                          stream_ptr(self.device)))
        return g

    def _backward(self, x, eps, beta, want_outputs, c, g, bce, PV, GV, B, lay):
        body = self._backward_body_p3 if c.get("p3") else self._backward_body
        return body(x, eps, beta, want_outputs, c, g, bce, PV, GV, B, lay)

    def _backward_body(self, x, eps, beta, want_outputs, c, g, bce, PV, GV, B, lay):
        # ---- decoder backward
        def d3_bias():
            # d3.bias gradient = sum over (b, y, x) of g[b, c, y, x]: column sums over the batch first ([B, 3072] ->
            # [3072]), then the 1024 pixels of each channel -- instead of permuting the whole gradient to [B * 1024, 3]
            # (an INTERMEDIATE: read two lines below, so its slice sum -- B > 512 rows are summed in slices -- must not
            # wait for the flush: deferral is suspended around it)
            check(load().mvae_slice_sums_defer(2))
            gpix = _colsum(g.view(B, 3072))
            check(load().mvae_slice_sums_defer(1))
            _colsum(_permute_rc(gpix, 1, 3, 1024).view(1024, 3), out=GV["d3.bias"])

        if not c.get("d3_bias_done"):
            d3_bias()
        if c["col0"] is None:  # the boundary layers straight from the images (csrc/mvae_edge.hip)
            _edge_wgrad(c["b2"], g, GV["d3.weight"].view(64, 48), B)
            db2 = _edge_conv(g, PV["d3.weight"].view(64, 48), None, c["b2"], False, B)
        else:
            dcol3 = _im2col(g, None, B, 3, 32, _nchw(32, 3))  # ConvT backward = im2col of the incoming gradient
            _gemm_tn(c["b2"], dcol3, out=GV["d3.weight"].view(64, 48))
            db2 = _linear_masked(dcol3, PV["d3.weight"].view(64, 48), c["b2"])  # ReLU mask in the contraction's epilogue
        # ConvTranspose2d backward = a Conv2d of the incoming gradient: implicit contractions, no patch matrices
        _conv_nhwc_wgrad(c["b1"], db2, self.flat.matrix(self.grads, "d2"), B, 64, 16)
        _colsum(db2, out=GV["d2.bias"])
        db1 = _conv_nhwc(db2, c["Wd2"], None, c["b1"], B, 64, 16, False, BACKWARD)  # [B*64, 256], ReLU mask in the epilogue
        _conv_nhwc_wgrad(c["t0"], db1, self.flat.matrix(self.grads, "d1"), B, 256, 8)
        _colsum(db1, out=GV["d1.bias"])
        dt0 = _conv_nhwc(db1, c["Wd1"], None, None, B, 256, 8, False, BACKWARD)  # [B*16, 128]
        NH = lay.heads_dim
        ow, ob = self.flat.off["w_heads"], self.flat.off["b_heads"]
        if c.get("fused"):
            # decoder fc backward, the components, the heads' backward against the channel-last flatten: two launches
            dhflat = torch.empty_like(c["hflat"])
            dheads = dt0.new_empty(B, NH)
            ws = dt0.new_empty(int(load().mvae_conv_latent_workspace_floats(B, lay.n)))
            epsc = eps.contiguous()
            check(load().mvae_conv_latent_backward(
                lay.descs, lay.n, ptr(c["hflat"]), ptr(self.params[ow:ow + NH * H_DIM]), ptr(c["heads"]), ptr(epsc),
                epsc.shape[1], ptr(self.params[:lay.n]), ptr(c["z"]), ptr(PV["d0.weight"]), ptr(c["t0"]), ptr(dt0), 1, 0,
                float(beta), ptr(self.grads[ow:ow + NH * H_DIM]), ptr(self.grads[ob:ob + NH]), ptr(dhflat), None, 0, None, None,
                ptr(GV["d0.weight"]), ptr(GV["d0.bias"]), ptr(self.grads[:lay.n]), ptr(dheads), ptr(ws), B,
                stream_ptr(self.device)))
        else:
            dhflat = self._latent_backward_generic(c, dt0, eps, beta, PV, GV, B, lay)
        # ---- encoder backward (Conv2d backward-data = col2im)
        da2 = dhflat.view(B * 16, 512)
        _conv_nhwc_wgrad(da2, c["a1"], self.flat.matrix(self.grads, "e2"), B, 128, 8)
        _colsum(da2, out=GV["e2.bias"])
        da1 = _col2im(_gemm_nn(da2, c["We2"], BACKWARD), None, c["a1"], B, 128, 8, _nhwc(8, 128), False, (B * 64, 128),
                      True)
        _conv_nhwc_wgrad(da1, c["a0"], self.flat.matrix(self.grads, "e1"), B, 64, 16)
        _colsum(da1, out=GV["e1.bias"])
        da0 = _convT_nhwc(da1, c["We1"], None, c["a0"], B, 128, 8, 64, False, BACKWARD)    # [B*256, 64], ReLU mask of a0
        if c["col0"] is None:
            _edge_wgrad(da0, c["x"], GV["e0.weight"].view(64, 48), B)
        else:
            _gemm_tn(da0, c["col0"], out=GV["e0.weight"].view(64, 48))
        _colsum(da0, out=GV["e0.bias"])
        check(load().mvae_slice_sums_flush(stream_ptr(self.device)))
        _DEFERRED_WS.clear()
        if want_outputs:
            return {"logits": c["logits"], "concat_z": c["z"], "bce": bce, "kl": c["kl"]}
        return None

    def _latent_backward(self, c, dt0, eps, beta, PV, GV, B, lay, planes=None, chansum_out=None):
        """Decoder fc backward, the components, the heads' backward: -> dhflat = the gradient of the channel-last a2 (+ its
        bf16 planes into `planes` [3, B*16, 512], fused latent section only).  chansum_out [512] (with planes): the sum of that
        gradient over rows and pixels per channel (e2.bias) from the same launch; the f32 gradient is then not written and None
        is returned."""
        NH = lay.heads_dim
        ow, ob = self.flat.off["w_heads"], self.flat.off["b_heads"]
        if not c.get("fused"):
            return self._latent_backward_generic(c, dt0, eps, beta, PV, GV, B, lay)
        skip = planes is not None and chansum_out is not None
        dhflat = None if skip else torch.empty_like(c["hflat"])
        cws = _keep(dt0.new_empty(H_DIM)) if skip else None
        dheads = dt0.new_empty(B, NH)
        ws = dt0.new_empty(int(load().mvae_conv_latent_workspace_floats(B, lay.n)))
        epsc = eps.contiguous()
        check(load().mvae_conv_latent_backward(
            lay.descs, lay.n, ptr(c["hflat"]), ptr(self.params[ow:ow + NH * H_DIM]), ptr(c["heads"]), ptr(epsc),
            epsc.shape[1], ptr(self.params[:lay.n]), ptr(c["z"]), ptr(PV["d0.weight"]), ptr(c["t0"]), ptr(dt0),
            dt0.shape[0] if dt0.dim() == 3 else 1, dt0[0].numel() if dt0.dim() == 3 else 0, float(beta),
            ptr(self.grads[ow:ow + NH * H_DIM]), ptr(self.grads[ob:ob + NH]), ptr(dhflat), _pptr(planes), _ps(planes),
            ptr(chansum_out) if skip else None, ptr(cws), ptr(GV["d0.weight"]), ptr(GV["d0.bias"]), ptr(self.grads[:lay.n]),
            ptr(dheads), ptr(ws), B, stream_ptr(self.device)))
        return dhflat

    def _backward_body_p3(self, x, eps, beta, want_outputs, c, g, bce, PV, GV, B, lay):
        """The backward pass on pre-split operands (contraction mode 2, csrc/mvae_p3.hip): the same sequence of contractions
        as _backward_body -- autograd of conv_vae.py:57-79 -- with every large one reading bf16 planes: of the forward
        activations (written by the forward epilogues), of the activation gradients (written by the epilogue that produces
        each) and of the four channel-last weight matrices (split here, one launch, from the CURRENT parameters)."""
        dev = self.device
        # planes of the weights and of the one forward activation whose producer does not write them (t0: the fused latent
        # section), in ONE launch
        if "W_p" in c:  # (contraction mode 1: the forward pass ran on them already)
            We1_p, We2_p, Wd1_p, Wd2_p = c["W_p"]
        else:
            We1_p, We2_p, Wd1_p, Wd2_p = _split_planes([c["We1"], c["We2"], c["Wd1"], c["Wd2"]])
        t0_p = c["t0_p"] if "t0_p" in c else _split_planes([c["t0"]])[0]
        # ---- decoder backward
        if not c.get("d3_bias_done"):
            check(load().mvae_slice_sums_defer(2))
            gpix = _colsum(g.view(B, 3072))
            check(load().mvae_slice_sums_defer(1))
            _colsum(_permute_rc(gpix, 1, 3, 1024).view(1024, 3), out=GV["d3.bias"])
        db2_p = _new_planes(B * 256, 64, dev)
        if c["col0"] is None:  # the boundary layers straight from the images (csrc/mvae_edge.hip)
            epi2 = self._sw_epi_colsum == "1"  # d2.bias from the same launch, no f32 db2 ("2": db1 only)
            db2 = _edge_backward(c["b2"], g, PV["d3.weight"].view(64, 48), GV["d3.weight"].view(64, 48), B, db2_p,
                                 colsum_out=GV["d2.bias"] if epi2 else None)
        else:
            dcol3 = _im2col(g, None, B, 3, 32, _nchw(32, 3))  # ConvT backward = im2col of the incoming gradient
            _gemm_tn(c["b2"], dcol3, out=GV["d3.weight"].view(64, 48))
            db2 = _linear_masked(dcol3, PV["d3.weight"].view(64, 48), c["b2"], planes=db2_p)  # ReLU mask in the epilogue
        # each layer's weight gradient and backward-data read the same incoming gradient and not each other: ONE launch per pair
        with _p3_group(dev):
            _conv_nhwc_wgrad_p3(c["b1_p"], db2_p, self.flat.matrix(self.grads, "d2"), B, 64, 16)
            # [B*64, 256], ReLU mask of b1; only its planes and its column sums (d1.bias) are ever read: the epilogue delivers
            # both and the f32 tensor is not written (MVAE_CONV_EPI_COLSUM=0: f32 result + the batched column sum)
            epi = self._sw_epi_colsum != "0"
            db1, db1_p = _conv_nhwc_p3(db2_p, Wd2_p, c["b1"], B, 64, 16, want_planes=True,
                                       colsum_out=GV["d1.bias"] if epi else None)
        if db2 is not None:
            _colsum(db2, out=GV["d2.bias"])
        with _p3_group(dev):
            _conv_nhwc_wgrad_p3(t0_p, db1_p, self.flat.matrix(self.grads, "d1"), B, 256, 8)
            # [B*16, 128]; with the fused latent section its K slices stay un-added (the latent backward adds them as it reads)
            dt0, _ = _conv_nhwc_p3(db1_p, Wd1_p, None, B, 256, 8, keep_slices=bool(c.get("fused")) and self._sw_dt0_slices)
        if db1 is not None:
            _colsum(db1, out=GV["d1.bias"])
        # ---- latent section
        da2_p = _new_planes(B * 16, 512, dev) if c.get("fused") else None
        epi_l = da2_p is not None and self._sw_epi_colsum != "0"  # e2.bias from the latent launch
        dhflat = self._latent_backward(c, dt0, eps, beta, PV, GV, B, lay, planes=da2_p,
                                       chansum_out=GV["e2.bias"] if epi_l else None)
        # ---- encoder backward
        if dhflat is not None:
            da2 = dhflat.view(B * 16, 512)
            if da2_p is None:
                da2_p = _split_planes([da2])[0]
            _colsum(da2, out=GV["e2.bias"])
        if self._sw_da1_implicit and load().mvae_p3_supported(2, B * 16, 128, 2048, 512):
            # backward-data of e2 as four implicit contractions per output parity class (no [B * 16, 2048] product, no col2im:
            # 256 workgroups with 64 K steps each instead of two rounds of 16-step ones).  Equal to the product form until the
            # transposed-convolution kernels took alternate K steps; now 0.744 -> 0.717 ms (mode 2), 0.631 -> 0.609 (mode 1).
            # MVAE_CONV_DA1_IMPLICIT=0: product + col2im.
            with _p3_group(dev):
                _conv_nhwc_wgrad_p3(da2_p, c["a1_p"], self.flat.matrix(self.grads, "e2"), B, 128, 8)
                epi = self._sw_epi_colsum != "0"  # e1.bias from the epilogue, no f32 da1
                da1, da1_p = _convT_nhwc_p3(da2_p, We2_p, c["a1"], B, 512, 4, 128, want_planes=True,
                                            colsum_out=GV["e1.bias"] if epi else None, want_y=not epi)
        else:
            da1_p = _new_planes(B * 64, 128, dev)
            with _p3_group(dev):
                _conv_nhwc_wgrad_p3(da2_p, c["a1_p"], self.flat.matrix(self.grads, "e2"), B, 128, 8)
                prod = _gemm_nn_p3(da2_p, We2_p)
            da1 = _col2im(prod, None, c["a1"], B, 128, 8, _nhwc(8, 128), False, (B * 64, 128), True, planes=da1_p)
        with _p3_group(dev):
            _conv_nhwc_wgrad_p3(da1_p, c["a0_p"], self.flat.matrix(self.grads, "e1"), B, 64, 16)
            epi0 = self._sw_epi_colsum != "0"  # e0.bias from the epilogue (da0 itself is still read)
            da0, _ = _convT_nhwc_p3(da1_p, We1_p, c["a0"], B, 128, 8, 64,
                                    colsum_out=GV["e0.bias"] if epi0 else None)  # [B*256, 64], ReLU mask of a0
        if da1 is not None:
            _colsum(da1, out=GV["e1.bias"])
        if c["col0"] is None:
            _edge_wgrad(da0, c["x"], GV["e0.weight"].view(64, 48), B)
        else:
            _gemm_tn(da0, c["col0"], out=GV["e0.weight"].view(64, 48))
        if not epi0:
            _colsum(da0, out=GV["e0.bias"])
        check(load().mvae_slice_sums_flush(stream_ptr(self.device)))
        _DEFERRED_WS.clear()
        if want_outputs:
            return {"logits": c["logits"], "concat_z": c["z"], "bce": bce, "kl": c["kl"]}
        return None

    def _latent_backward_generic(self, c, dt0, eps, beta, PV, GV, B, lay):
        dd0 = _relu_mask_(_permute_rc(dt0, B, 16, 128).view(B, 2048), c["d0o"])
        _, _, dz = Fn.linear_backward(c["z"], PV["d0.weight"], dd0, relu_in=False, need_dx=True,
                                      out_dW=GV["d0.weight"], out_db=GV["d0.bias"])
        # ---- latent components
        # d/d(radius) lands in the radii region of the flat gradient buffer (fixed-order row sums); the optimizer kernel
        # only applies it to trainable radii, the rest of the region stays 0
        dheads, _ = Fn.component_backward(lay, c["heads"], eps, self.params[:lay.n], dz, None, float(beta),
                                          out_dradii=self.grads[:lay.n])
        NH = lay.heads_dim
        ow, ob = self.flat.off["w_heads"], self.flat.off["b_heads"]
        # heads backward against the channel-last head matrix of the forward pass; its gradient is re-ordered into the
        # reference's column order on the way into the flat gradient buffer, dhflat IS the channel-last da2
        dW_cl = dheads.new_empty(NH, H_DIM)
        _, _, dhflat = Fn.linear_backward(c["hflat"], c["w_heads_cl"], dheads, relu_in=True, need_dx=True,
                                          out_dW=dW_cl, out_db=self.grads[ob:ob + NH])
        _permute_rc(dW_cl.view(NH, 16, 512), NH, 16, 512, out=self.grads[ow:ow + NH * H_DIM])
        return dhflat

    def optimizer_step(self, do_curvature_step: bool, batch: Optional[int] = None) -> None:
        check(load().mvae_optimizer_step_flat(ptr(self.params), ptr(self.grads), ptr(self.adam_m), ptr(self.adam_v),
                                              self.flat.n_params, ptr(self.counters), self.layout.n,
                                              self._trainable_arr, self.lr, self.curvature_lr,
                                              1 if do_curvature_step else 0, 1, stream_ptr(self.device)))

    def train_step(self, x: Tensor, eps: Tensor, beta: float = 1.0, do_curvature_step: bool = False) -> None:
        self.forward_backward(x, eps, beta)
        self.optimizer_step(do_curvature_step)

    def read_stats(self, reset: bool = False):
        s = self.stats.cpu()
        n = 4 + self.layout.n
        rec = lambda v: {"bce": float(v[0]), "kl": float(v[1]), "elbo": float(v[2]), "steps": int(v[3]),  # noqa: E731
                         "component_kl": [float(t) for t in v[4:n]]}
        out = {"sum": rec(s[:n]), "last": rec(s[n:2 * n])}
        if reset:
            self.stats.zero_()
        return out
