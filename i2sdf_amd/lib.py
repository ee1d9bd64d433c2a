"""ctypes binding of the C ABI declared in include/i2sdf.h (libi2sdf_hip.so, built in-tree by
i2sdf_amd/csrc/build.sh or __graft_entry__.build()).  No CPU fallback: a missing library or a failing call raises."""
from __future__ import annotations

import ctypes as C
import os

MAX_LAYERS = 12
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("I2SDF_LIB_PATH") or os.path.join(_HERE, "lib", "libi2sdf_hip.so")     # override: A/B of two builds on one box


class MlpDesc(C.Structure):
    _fields_ = [("n_lin", C.c_int32), ("hidden", C.c_int32), ("d_in", C.c_int32), ("in0", C.c_int32), ("d_out", C.c_int32),
                ("multires", C.c_int32), ("skip_layer", C.c_int32), ("reserved", C.c_int32),
                ("out_dim", C.c_int32 * MAX_LAYERS), ("in_dim", C.c_int32 * MAX_LAYERS),
                ("off_bias", C.c_int64 * MAX_LAYERS), ("off_g", C.c_int64 * MAX_LAYERS), ("off_v", C.c_int64 * MAX_LAYERS)]


class NetDesc(C.Structure):
    _fields_ = [("sdf", MlpDesc), ("rgb", MlpDesc), ("light", MlpDesc), ("off_beta", C.c_int64), ("n_params", C.c_int64),
                ("beta_min", C.c_float), ("scene_bounding_sphere", C.c_float)]


class TrainBuffers(C.Structure):
    _fields_ = [("M_sdf", C.c_int64), ("M_main", C.c_int64), ("Mp", C.c_int64)] + [(n, C.c_void_p) for n in (
        "pe", "hs", "abars", "gus", "gpbar", "gas", "ga_last4", "ones4", "fbar", "pev", "feat", "rs", "gar", "ga_last_rgb",
        "hl", "gal0", "gal_last")]


class SamplerCfg(C.Structure):
    _fields_ = [("near", C.c_float), ("eps", C.c_float), ("add_tiny", C.c_float), ("N_samples", C.c_int32), ("N_samples_eval", C.c_int32),
                ("N_samples_extra", C.c_int32), ("beta_iters", C.c_int32), ("max_total_iters", C.c_int32)]


# int allreduce(void* ctx, void* buf, int64_t n, int32_t dtype, int32_t op, void* stream)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p)


class Exchange(C.Structure):
    _fields_ = [("allreduce", EXCHANGE_FN), ("ctx", C.c_void_p)]


class LossCfg(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("eikonal_w", "smooth_w", "mask_w", "depth_w", "normal_w", "angular_w", "bubble_w", "light_w")] + \
               [("smooth_on", C.c_int32), ("reserved", C.c_int32), ("exchange", C.POINTER(Exchange))]


XCHG_F32, XCHG_I32 = 0, 1
XCHG_SUM, XCHG_AVG, XCHG_MAX = 0, 1, 2
DP_GLOBAL_SAMPLER = 1
COMM_UNIQUE_ID_BYTES = 128


class RayTables(C.Structure):
    _fields_ = [("intrinsics", C.c_void_p), ("pose", C.c_void_p), ("pose_is_quat", C.c_int32), ("n_images", C.c_int32),
                ("height", C.c_int32), ("width", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("rgb", "depth", "normal", "mask", "light_mask", "depth_mask", "normal_mask")]


class RayBatch(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("image_idx", "uv", "cam_loc", "dirs", "dnorm", "rgb", "depth", "normal", "mask", "light_mask",
                                          "depth_mask", "normal_mask", "n_bad")]


OPT_SDF_FWD_BF16X3 = 1
OPT_WGRAD_BF16X3 = 2
OPT_TRAIN_FWD_BF16X3 = 4
OPT_SDF_BWD_BF16X3 = 8
OPT_RGB_BF16X3 = 16
OPT_TAIL_OVERLAP = 32
OPT_BLOCKED_SAVES = 128
OPT_WGRAD_BF16X2 = 256
OPT_PARTS = 512
OPT_SAMPLER_BF16X2 = 1024
OPT_SAVES24 = 2048
MAX_PARTS = 4
GRID_ORDER_MESHGRID, GRID_ORDER_VOLUME = 0, 1


class I2SDFError(RuntimeError):
    pass


_lib = None

# name -> (restype, argtypes).  Must list every symbol include/i2sdf.h declares (tests/test_cabi.py checks it).
_P, _I64, _I32, _F, _D = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_double
SIGNATURES = {
    "i2sdf_version": (C.c_int, []),
    "i2sdf_strerror": (C.c_char_p, [C.c_int]),
    "i2sdf_last_hip_error": (C.c_char_p, []),
    "i2sdf_plan_create": (C.c_int, [C.POINTER(NetDesc), C.POINTER(_P)]),
    "i2sdf_plan_destroy": (None, [_P]),
    "i2sdf_plan_set_option": (C.c_int, [_P, _I32, _I32]),
    "i2sdf_chain_begin": (C.c_int, [_P, _I64, _P]),
    "i2sdf_chain_fence": (C.c_int, [_P, _P]),
    "i2sdf_chain_end": (C.c_int, [_P, _P]),
    "i2sdf_blocked_points": (_I64, [_P, _I32, _I64, _I64, _I32]),
    "i2sdf_plan_pack_floats": (_I64, [_P]),
    "i2sdf_plan_wgrad_floats": (_I64, [_P]),
    "i2sdf_pack_weights": (C.c_int, [_P, _P, _P, _P]),
    "i2sdf_sdf_forward": (C.c_int, [_P, _P, _P, _I64, _P, _P, _I64, _P]),
    # plan, packed, points, cam, dirs, z, ldz, n_per_ray, n_ray_pts, M, Mp, sdf, feat, grad, hs, abars, pe_save, stream
    "i2sdf_sdf_forward_grad": (C.c_int, [_P] * 6 + [_I64, _I32, _I64, _I64, _I64] + [_P] * 7),
    # plan, packed, dirs, n_per_ray, feat, M, Mp, rgb, rs, pev_save, stream
    "i2sdf_rgb_forward": (C.c_int, [_P, _P, _P, _I32, _P, _I64, _I64, _P, _P, _P, _P]),
    # plan, packed, rgb, rgb_bar, rs, M, Mp, gar, ga_last, fbar, stream
    "i2sdf_rgb_backward": (C.c_int, [_P] * 5 + [_I64, _I64] + [_P] * 4),
    # plan, packed, points, cam, dirs, z, ldz, n_per_ray, n_ray_pts, M, Mp, hs, abars, sbar, fbar, m_fbar, nbar, gus, gpbar, gas,
    # ga_last4, ones4, stream
    "i2sdf_sdf_backward": (C.c_int, [_P] * 6 + [_I64, _I32, _I64, _I64, _I64] + [_P] * 4 + [_I64] + [_P] * 7),
    "i2sdf_wgrad_chunk_points": (_I64, []),
    "i2sdf_weight_grads": (C.c_int, [_P, C.POINTER(TrainBuffers), _P, _P, _I64, _P, _P]),
    "i2sdf_comm_available": (_I32, []),
    "i2sdf_comm_unique_id": (C.c_int, [_P, _I64]),
    "i2sdf_comm_init_rank": (C.c_int, [_P, _I32, _I32, C.POINTER(_P)]),
    "i2sdf_comm_destroy": (None, [_P]),
    "i2sdf_comm_size": (_I32, [_P]),
    "i2sdf_comm_rank": (_I32, [_P]),
    "i2sdf_last_comm_error": (C.c_char_p, []),
    "i2sdf_allreduce_grads": (C.c_int, [_P, _I64, _P, _P]),
    "i2sdf_allreduce_max_i32": (C.c_int, [_P, _I64, _P, _P]),
    "i2sdf_broadcast": (C.c_int, [_P, _I64, _I32, _P, _P]),
    "i2sdf_comm_as_exchange": (C.c_int, [_P, C.POINTER(Exchange)]),
    "i2sdf_plan_set_exchange": (C.c_int, [_P, C.POINTER(Exchange), _I32]),
    "i2sdf_adam_step": (C.c_int, [_P, _P, _P, _P, _I64, _D, _D, _D, _D, _D, _I64, _D, _P]),
    "i2sdf_error_bound": (C.c_int, [_P, _P, _I64, _I32, _P, _I64, _P, _P, _P, _P]),
    "i2sdf_sampler_workspace_floats": (_I64, [_I64]),
    # plan, packed, params, cfg, cam, dirs, B, training, t_lin, u_more, u_final, ldu_final, extra_tab, strat_u, extra_idx, eik_idx,
    # force_iters, workspace, z_out, ldz, z_eik, iters_out, stream
    "i2sdf_sample_rays": (C.c_int, [_P, _P, _P, C.POINTER(SamplerCfg), _P, _P, _I64, _I32, _P, _P, _P, _I64, _P, _P, _P, _P, _I32, _P, _P,
                                    _I64, _P, _P, _P]),
    "i2sdf_render_image_workspace_floats": (_I64, [_P, C.POINTER(SamplerCfg), _I64]),
    # plan, packed, params, cfg, uv, pose, pose_is_quat, intrinsics, P, chunk, t_lin, u_more, u_final, extra_tab, workspace,
    # o_rgb, o_depth, o_wsum, o_normal, o_lmask, o_z, o_iters, stream
    "i2sdf_render_image": (C.c_int, [_P, _P, _P, C.POINTER(SamplerCfg), _P, _P, _I32, _P, _I64, _I64] + [_P] * 13),
    "i2sdf_sdf_grid_workspace_floats": (_I64, [_I64]),
    # plan, packed, x, y, z, nx, ny, nz, order, rot (host), trans (host), first, count, sdf_out, workspace, chunk_points, stream
    "i2sdf_sdf_grid": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _P, _I64, _I64, _P, _P, _I64, _P]),
    # pred, target, channels, pixel_idx, first_pixel, n, pointlinks, n_links, pdf_max, pdf_prune, pdf, n_pdf, n_bad, stream
    "i2sdf_pdf_update": (C.c_int, [_P, _P, _I32, _P, _I64, _I64, _P, _I64, C.c_double, C.c_double, _P, _I64, _P, _P]),
    # seed, B, n_eval, n_samples, n_extra, max_iters, n_z, eik_radius, nbr_half_width, strat_u, cdf_u, extra_idx, eik_idx, eik_pts, nbr_off, stream
    "i2sdf_training_draws": (C.c_int, [C.c_uint64, _I64, _I32, _I32, _I32, _I32, _I32, C.c_float, C.c_float, _P, _P, _P, _P, _P, _P, _P]),
    "i2sdf_light_forward": (C.c_int, [_P, _P, _P, _I64, _I64, _P, _P, _P]),
    "i2sdf_light_backward": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, _P, _P, _P]),
    "i2sdf_loss_scratch_floats": (_I64, []),
    "i2sdf_loss_forward_backward": (C.c_int, [C.POINTER(LossCfg), _I64, _I64] + [_P] * 27),
    "i2sdf_eikonal_outputs_forward": (C.c_int, [_P, _I64, _P, _P, _P]),
    "i2sdf_extra_points": (C.c_int, [_P, _P, _P, _P, _P, _I64, _P, _P]),
    "i2sdf_render_loss_scratch_floats": (_I64, [_I64]),
    # cfg, B, n, n_pc, M_main, M_sdf, n_eik | beta_param, beta_min, z, ldz, sdf, rgb_pts, grad_pts, dnorm, nsum_save | 8 outputs | 7 ground truth |
    # scratch, losses, loss_value | 8 seeds | sdf_bar, rgb_bar, grad_bar, normal_term, lmask_bar, beta_grad, stream
    "i2sdf_render_loss_backward": (C.c_int, [C.POINTER(LossCfg), _I64, _I32, _I64, _I64, _I64, _I64, _P, C.c_float, _P, _I64] + [_P] * 5 + [_P] * 8 + [_P] * 7 +
                                   [_P] * 3 + [_P] * 8 + [_P, _P, _P, _I32, _P, _P, _P]),
    "i2sdf_scale_seeds": (C.c_int, [_P, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _P, _P]),
    "i2sdf_backward_seeds": (C.c_int, [_P, _I64, _P, _P, _I64, _I64, _P, _I64, _P, _I64, C.c_int32, _P]),
    "i2sdf_eikonal_outputs_backward": (C.c_int, [_P, _P, _P, _I64, _P, _P]),
    "i2sdf_ray_setup": (C.c_int, [_P, _P, _P, _I64, _I32, _P, _P, _P, _P]),
    "i2sdf_ray_setup_ex": (C.c_int, [_P, _P, _I32, _P, _I64, _I32, _P, _P, _P, _P]),
    "i2sdf_ray_batch": (C.c_int, [C.POINTER(RayTables), _P, _I64, C.POINTER(RayBatch), _P]),
    "i2sdf_sphere_intersections": (C.c_int, [_P, _P, _I64, _F, _P, _P, _P]),
    "i2sdf_composite_forward": (C.c_int, [_P, _F, _P, _I64, _P, _P, _P, _P, _P, _I64, _I32] + [_P] * 8),
    "i2sdf_composite_forward_eik": (C.c_int, [_P, _F, _P, _I64, _P, _P, _P, _P, _P, _I64, _I32] + [_P] * 11),
    "i2sdf_composite_backward": (C.c_int, [_P, _F, _P, _I64] + [_P] * 5 + [_I64, _I32] + [_P] * 12),
}


def load():
    """Load the HIP library or raise -- there is deliberately no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so.7; it must be in the process before our library so that both bind to the
    # SAME HIP runtime instance (streams and device pointers are shared between them).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise I2SDFError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        lib = load()
        msg = lib.i2sdf_strerror(rc).decode()
        if rc == -2 or (rc == -1 and what == "i2sdf_plan_create"):      # (plan creation leaves the reason of a refused shape there)
            msg += ": " + lib.i2sdf_last_hip_error().decode()
        if rc == -5:
            msg += ": " + lib.i2sdf_last_comm_error().decode()
        raise I2SDFError(f"{what} failed ({rc}): {msg}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), "C ABI needs contiguous tensors"
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
