"""ctypes binding of libmvae_hip.so (C ABI: include/mvae_hip.h).

There is deliberately NO fallback: if the shared library is missing or an entry point fails, the product path raises.
"""
import ctypes as C
import os
from typing import Optional

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MVAE_HIP_LIB") or os.path.join(HERE, "libmvae_hip.so")  # override: A/B builds

EUCLIDEAN, HYPERBOLOID, SPHERE, POINCARE, PROJ_SPHERE, UNIVERSAL = 0, 1, 2, 3, 4, 5
ABI_VERSION = 12
# return codes of the C ABI (include/mvae_hip.h)
MVAE_OK, MVAE_E_BADARG, MVAE_E_UNSUPPORTED, MVAE_E_ALIGN, MVAE_E_SYSTEM = 0, -1, -2, -3, -4
MAX_TRUE_DIM = 64
MAX_COMPONENTS = 64
RADII_REGION = 64
# MVAE_OP_* (mvae_primitive_backward) and MVAE_FN_* (mvae_scalar_fn) of include/mvae_hip.h
(OP_EXP0, OP_LOG0, OP_PT0, OP_IPT0, OP_SAMPLE, OP_ISAMPLE, OP_LOGDET, OP_EXP, OP_LOG, OP_DIST, OP_DIST_GYRO, OP_LPROD,
 OP_LNORM, OP_TO_BALL, OP_TO_AMBIENT, OP_LAMBDA, OP_MOBADD, OP_NORMAL_LOGPROB, OP_NORMAL_RSAMPLE,
 OP_NORMAL_KL) = range(20)
DIST_GEODESIC, DIST_GYRO = 0, 1
SCALAR_FNS = {name: i for i, name in enumerate(
    ["clamp", "atanh", "acosh", "cosh", "sinh", "sqrt", "logsinh", "logcosh", "cosh_sinh_pair", "cos_sin_pair",
     "softplus", "acos", "tan", "log1p_pos", "exp", "log", "std"])}


class ComponentDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("true_dim", C.c_int32), ("mean_col", C.c_int32), ("logvar_col", C.c_int32),
                ("logvar_dim", C.c_int32), ("eps_col", C.c_int32), ("z_col", C.c_int32), ("radius_idx", C.c_int32)]


class ModelDesc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("arch", C.c_int32), ("batch", C.c_int32), ("in_dim", C.c_int32),
                ("h_dim", C.c_int32), ("ncomp", C.c_int32), ("heads_dim", C.c_int32), ("z_dim", C.c_int32),
                ("eps_dim", C.c_int32), ("n_params", C.c_int32), ("comps", C.POINTER(ComponentDesc)),
                ("off_radii", C.c_int64), ("off_w_heads", C.c_int64), ("off_b_heads", C.c_int64),
                ("off_w_e0", C.c_int64), ("off_b_e0", C.c_int64), ("off_w_d0", C.c_int64), ("off_b_d0", C.c_int64),
                ("off_w_logits", C.c_int64), ("off_b_logits", C.c_int64), ("params", C.c_void_p),
                ("grads", C.c_void_p), ("adam_m", C.c_void_p), ("adam_v", C.c_void_p), ("step_count", C.c_void_p),
                ("workspace", C.c_void_p), ("stats", C.c_void_p), ("radius_trainable", C.POINTER(C.c_uint8)),
                ("lr", C.c_double), ("curvature_lr", C.c_double)]


# name -> (restype, argtypes): every symbol include/mvae_hip.h declares
_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
PROTOTYPES = {
    "mvae_abi_version": (C.c_int, []),
    "mvae_last_error": (C.c_char_p, []),
    "mvae_exp_map_mu0": (C.c_int, [_I, _P, _P, _L, _I, _P, _P]),
    "mvae_inverse_exp_map_mu0": (C.c_int, [_I, _P, _P, _L, _I, _P, _P]),
    "mvae_parallel_transport_mu0": (C.c_int, [_I, _P, _P, _P, _L, _I, _P, _P]),
    "mvae_inverse_parallel_transport_mu0": (C.c_int, [_I, _P, _P, _P, _L, _I, _P, _P]),
    "mvae_sample_projection_mu0": (C.c_int, [_I, _P, _P, _P, _P, _L, _L, _I, _P, _P]),
    "mvae_inverse_sample_projection_mu0": (C.c_int, [_I, _P, _P, _P, _P, _L, _L, _I, _P, _P]),
    "mvae_logdet": (C.c_int, [_I, _P, _P, _P, _P, _L, _L, _I, _P, _P]),
    "mvae_exp_map": (C.c_int, [_I, _P, _P, _P, _L, _L, _I, _P, _P]),
    "mvae_inverse_exp_map": (C.c_int, [_I, _P, _P, _P, _L, _L, _I, _P, _P]),
    "mvae_geodesic_distance": (C.c_int, [_I, _I, _P, _P, _P, _L, _L, _I, _P, _P]),
    "mvae_manifold_aux": (C.c_int, [_I, _I, _P, _P, _P, _L, _I, _P, _P]),
    "mvae_normal_op": (C.c_int, [_I, _P, _P, _P, _P, _L, _L, _I, _P]),
    "mvae_primitive_backward": (C.c_int, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _I, _P, _P]),
    "mvae_scalar_fn": (C.c_int, [_I, _P, _P, _P, _L, _F, _F, _P]),
    "mvae_mul": (C.c_int, [_P, _P, _P, _L, _P]),
    "mvae_component_forward": (C.c_int, [C.POINTER(ComponentDesc), _I, _P, _I, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P,
                                         _L, _L, _P]),
    "mvae_component_backward_workspace_floats": (C.c_int64, [_I, _L]),
    "mvae_component_backward": (C.c_int, [C.POINTER(ComponentDesc), _I, _P, _I, _P, _I, _P, _P, _I, _P, _F, _P, _P, _P,
                                          _L, _P]),
    "mvae_component_forward_f64": (C.c_int, [C.POINTER(ComponentDesc), _I, _P, _I, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P,
                                         _L, _L, _P]),
    "mvae_component_backward_f64": (C.c_int, [C.POINTER(ComponentDesc), _I, _P, _I, _P, _I, _P, _P, _I, _P, _F, _P, _P, _P,
                                          _L, _P]),
    "mvae_scale_rows": (C.c_int, [_P, _P, _P, _L, _I, _P]),
    "mvae_linear_forward": (C.c_int, [_P, _P, _P, _P, _L, _I, _I, _I, _P]),
    "mvae_linear_forward_masked": (C.c_int, [_P, _P, _P, _P, _L, _I, _I, _P, _L, _P]),
    "mvae_linear_forward_planes": (C.c_int, [_P, _P, _P, _P, _P, _L, _L, _I, _I, _I, _P]),
    "mvae_linear_backward": (C.c_int, [_P, _P, _P, _I, _P, _P, _P, _L, _I, _I, _P]),
    "mvae_im2col_k4s2p1": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _L, _I, _P]),
    "mvae_col2im_k4s2p1": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _L, _I, _I, _P, _L, _P]),
    "mvae_conv_transpose_k4s2p1_nhwc": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _L, _P]),
    "mvae_conv_k4s2p1_nhwc_workspace_floats": (C.c_int64, [_I, _I, _I, _I, _I, _I]),
    "mvae_conv_k4s2p1_nhwc": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P, _L, _P]),
    "mvae_conv_k4s2p1_nhwc_wgrad_workspace_floats": (C.c_int64, [_I, _I, _I, _I, _I]),
    "mvae_conv_k4s2p1_nhwc_wgrad": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "mvae_linear_forward_splitk_workspace_floats": (C.c_int64, [_L, _I, _I]),
    "mvae_linear_forward_splitk": (C.c_int, [_P, _P, _P, _P, _L, _I, _I, _I, _P, _P]),
    "mvae_permute_rc": (C.c_int, [_P, _P, _L, _I, _I, _P]),
    "mvae_gemm_tn_workspace_floats": (C.c_int64, [_L, _I, _I]),
    "mvae_gemm_tn": (C.c_int, [_P, _P, _P, _L, _I, _I, _P, _P]),
    "mvae_relu_mask": (C.c_int, [_P, _P, _L, _P]),
    "mvae_gemm_nn": (C.c_int, [_P, _P, _P, _P, _L, _I, _I, _I, _P]),
    "mvae_colsum_workspace_floats": (C.c_int64, [_L, _I]),
    "mvae_colsum": (C.c_int, [_P, _P, _L, _I, _P, _P]),
    "mvae_bce_forward_backward": (C.c_int, [_P, _P, _P, _P, _L, _I, _P]),
    "mvae_batch_stats": (C.c_int, [_P, _P, _P, _F, _I, _I, _P]),
    "mvae_set_contraction_mode": (C.c_int, [_I]),
    "mvae_set_forward_kernel": (C.c_int, [_I]),
    "mvae_p3_supported": (C.c_int, [_I, _L, _I, _I, _I]),
    "mvae_split3_planes": (C.c_int, [_I, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64), _P]),
    "mvae_split3_planes_queue": (C.c_int, [_I, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64), _P]),
    "mvae_split3_planes_flush": (C.c_int, [_P]),
    "mvae_conv3_k4s2p1_nchw": (C.c_int, [_P, _P, _P, _P, _I, _P, _P, _L, _I, _I, _I, _I, _I, _P]),
    "mvae_conv3_k4s2p1_nchw_wgrad_workspace_floats": (C.c_int64, [_I, _I, _I, _I, _I]),
    "mvae_conv3_k4s2p1_nchw_wgrad": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "mvae_conv3_k4s2p1_nchw_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "mvae_conv3_k4s2p1_nchw_backward_colsum_floats": (C.c_int64, [_I]),
    "mvae_p3_group": (C.c_int, [_I, _P]),
    "mvae_conv_k4s2p1_nhwc_p3_workspace_floats": (C.c_int64, [_I, _I, _I, _I, _I, _I]),
    "mvae_conv_k4s2p1_nhwc_p3": (C.c_int, [_P, _L, _P, _L, _P, _P, _I, _P, _P, _L, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "mvae_gemm_nn_p3": (C.c_int, [_P, _L, _P, _L, _P, _L, _I, _I, _P]),
    "mvae_conv_transpose_k4s2p1_nhwc_p3": (C.c_int, [_P, _L, _P, _L, _P, _P, _I, _P, _P, _L, _P, _P, _I, _I, _I, _I, _I, _P]),
    "mvae_conv_transpose_k4s2p1_nhwc_p3_colsum_floats": (C.c_int64, [_I, _I, _I, _I]),
    "mvae_conv_k4s2p1_nhwc_wgrad_p3_workspace_floats": (C.c_int64, [_I, _I, _I, _I, _I]),
    "mvae_conv_k4s2p1_nhwc_wgrad_p3": (C.c_int, [_P, _L, _P, _L, _P, _I, _I, _I, _I, _I, _P, _P]),
    "mvae_convt_to3_k4s2p1_forward": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "mvae_conv_bce_stats": (C.c_int, [_P, _P, _P, _P, _P, _P, _F, _L, _I, _I, _I, _P, _P, _P, _P]),
    "mvae_convt_to3_bce_stats": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _L, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "mvae_conv_latent_supported": (C.c_int, [_P, _I]),
    "mvae_conv_latent_workspace_floats": (_L, [_L, _I]),
    "mvae_conv_latent_forward": (C.c_int, [_P, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _L, _P]),
    "mvae_conv_latent_backward": (C.c_int, [_P, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _L, _F, _P, _P, _P, _P, _L, _P, _P,
                                            _P, _P, _P, _P, _P, _L, _P]),
    "mvae_optimizer_step_flat": (C.c_int, [_P, _P, _P, _P, _L, _P, _I, C.POINTER(C.c_uint8), C.c_double, C.c_double,
                                           _I, _I, _P]),
    "mvae_bce_rows": (C.c_int, [_P, _P, _P, _L, _L, _I, _P]),
    "mvae_decode_bce_rows": (C.c_int, [_P, _L, _I, _P, _P, _P, _P, _P, _L, _I, _I, _P, _P]),
    "mvae_loglik_reduce": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "mvae_loglik_reduce_comps": (C.c_int, [_P, _P, _P, _I, _P, _I, _P, _P, _P, _I, _I, _P]),
    "mvae_cov_norm_workspace_floats": (_L, [_I]),
    "mvae_cov_norm": (C.c_int, [_P, _P, _I, _I, _I, _P, _P, _P]),
    "mvae_workspace_floats": (C.c_int64, [C.POINTER(ModelDesc)]),
    "mvae_create": (C.c_int, [C.POINTER(ModelDesc), C.POINTER(C.c_void_p)]),
    "mvae_destroy": (None, [C.c_void_p]),
    "mvae_set_radius_trainable": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint8)]),
    "mvae_step_forward_backward": (C.c_int, [_P, _P, _P, _F, _I, _P, _P, _P, _P, _P]),
    "mvae_step_optimizer": (C.c_int, [_P, _I, _P]),
    "mvae_train_step": (C.c_int, [_P, _P, _P, _F, _I, _P]),
    "mvae_randn": (C.c_int, [_P, C.c_int64, C.c_uint64, C.c_uint64, _P]),
    "mvae_prepare_batch": (C.c_int, [_P, _P, _I, _I, _I, _I, C.c_uint64, _P, _I, _I, _P, _P, _P]),
    "mvae_set_next_batch_feed": (C.c_int, [_P, _P, _P, _I, C.c_uint64, _I, _I, _P, _P]),
    "mvae_slice_sums_defer": (C.c_int, [_I]),
    "mvae_slice_sums_flush": (C.c_int, [_P]),
    "mvae_step_kernel_path": (C.c_int, [_P]),
    "mvae_set_valid_rows": (C.c_int, [_P, _I]),
    "mvae_peer_create": (C.c_int, [C.c_int64, _I, _I, C.c_char_p, C.c_double, C.POINTER(C.c_void_p)]),
    "mvae_peer_destroy": (None, [C.c_void_p]),
    "mvae_peer_export": (C.c_int, [_P, C.POINTER(C.c_uint8)]),
    "mvae_peer_import": (C.c_int, [_P, _I, C.POINTER(C.c_uint8)]),
    "mvae_peer_publish": (C.c_int, [_P, _P, _P]),
    "mvae_peer_set_two_shot": (C.c_int, [_P, _I]),
    "mvae_step_optimizer_peer": (C.c_int, [_P, _P, _I, _P]),
    "mvae_step_optimizer_slice": (C.c_int, [_P, _I, _I, _I, _P]),
    "mvae_peer_timeouts": (C.c_int, [_P]),
    "mvae_step_profile": (C.c_int, [_P, _P, _P, _F, _I, _I, C.POINTER(C.c_float), _P]),
    "mvae_rccl_load": (C.c_int, [C.c_char_p]),
    "mvae_rccl_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "mvae_rccl_create": (C.c_int, [C.POINTER(C.c_uint8), _I, _I, C.POINTER(C.c_void_p)]),
    "mvae_rccl_destroy": (None, [C.c_void_p]),
    "mvae_flat_allreduce": (C.c_int, [_P, _P, C.c_int64, _P]),
    "mvae_flat_reduce_scatter": (C.c_int, [_P, _P, C.c_int64, _P]),
    "mvae_flat_allgather": (C.c_int, [_P, _P, C.c_int64, _P]),
    "mvae_flat_broadcast": (C.c_int, [_P, _P, C.c_int64, _I, _P]),
    "mvae_rccl_group": (C.c_int, [_I]),
}

_lib: Optional[C.CDLL] = None


class MvaeHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Loads the library once; raises if it is missing (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MvaeHipError(f"{LIB_PATH} is missing: build it with `python -m mvae_amd.build` "
                           "(or __graft_entry__.build()). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.mvae_abi_version() != ABI_VERSION:
        raise MvaeHipError("libmvae_hip.so ABI version mismatch: rebuild")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().mvae_last_error().decode(errors="replace")
        raise MvaeHipError(f"libmvae_hip error {rc}: {msg}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a contiguous fp32 CUDA(HIP) tensor (None passes NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MvaeHipError("mvae_amd ops need tensors on a HIP device (there is no CPU path)")
    if t.dtype != torch.float32 and t.dtype != torch.int32 and t.dtype != torch.uint8:
        raise MvaeHipError(f"unsupported dtype {t.dtype}: the HIP path computes in float32")
    if not t.is_contiguous():
        raise MvaeHipError("tensor must be contiguous")
    return t.data_ptr()


def stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream
