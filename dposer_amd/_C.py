"""ctypes binding of libdposer_hip.so (C ABI: include/dposer_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C dposer_amd/csrc``.  There is
NO fallback: if the shared object is missing or a call fails, the caller gets an exception.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DPOSER_LIB_PATH") or os.path.join(_HERE, "libdposer_hip.so")      # (override: the host-ASan build of the tests)

PREC_BF16, PREC_FP32, PREC_BF16X3 = 0, 1, 2
PRECISIONS = {"bf16": PREC_BF16, "fp32": PREC_FP32, "bf16x3": PREC_BF16X3}
EMB_POSITIONAL, EMB_FOURIER = 0, 1
ACTIVATIONS = {"swish": 0, "elu": 1, "relu": 2, "lrelu": 3}      # config.model.nonlinearity -> DPOSER_ACT_*
SDE_SUBVP, SDE_VP, SDE_VE, SDE_VE_DISCRETE, SDE_VP_DISCRETE = 0, 1, 2, 3, 4
WS_INFER, WS_SHARED_T, WS_TRAIN = 0, 1, 2


class ScoreFCDesc(C.Structure):
    _fields_ = [("data_dim", C.c_int32), ("hidden_dim", C.c_int32), ("embed_dim", C.c_int32),
                ("n_blocks", C.c_int32), ("embedding", C.c_int32), ("scale_by_sigma", C.c_int32),
                ("num_scales", C.c_int32), ("precision", C.c_int32), ("dropout_p", C.c_float), ("activation", C.c_int32)]


class MlpDesc(C.Structure):
    """dposer_mlp_desc (include/dposer_hip.h)."""
    _fields_ = [("in_dim", C.c_int32), ("out_dim", C.c_int32), ("hidden_dim", C.c_int32), ("n_blocks", C.c_int32),
                ("precision", C.c_int32), ("activation", C.c_int32), ("dropout_p", C.c_float)]


class SdeDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("N", C.c_int32), ("beta_min", C.c_double),
                ("beta_max", C.c_double), ("T", C.c_double)]


class BodyDesc(C.Structure):
    _fields_ = [("num_joints", C.c_int32), ("num_vertices", C.c_int32), ("num_shape", C.c_int32),
                ("num_extra", C.c_int32), ("num_landmarks", C.c_int32)]


class MotionDenoiseArgs(C.Structure):
    """dposer_motion_denoise_args (include/dposer_hip.h), field for field."""
    _fields_ = [("net", C.c_void_p), ("flat_params", C.c_void_p), ("packed", C.c_void_p), ("net_ws", C.c_void_p), ("sde", C.POINTER(SdeDesc)),
                ("freq", C.c_void_p), ("sigmas", C.c_void_p), ("body", C.c_void_p), ("lbs_ws_fwd", C.c_void_p), ("lbs_ws_bwd", C.c_void_p),
                ("posedirs_packed", C.c_void_p), ("posedirs_bwd_packed", C.c_void_p), ("j_rest", C.c_void_p), ("v_shaped", C.c_void_p),
                ("rest_batched", C.c_int32), ("skin_idx", C.c_void_p), ("skin_w", C.c_void_p), ("skin_k", C.c_int32), ("joint_ptr", C.c_void_p),
                ("joint_vidx", C.c_void_p), ("joint_w", C.c_void_p), ("extra_vertex_ids", C.c_void_p), ("lmk_tri", C.c_void_p),
                ("lmk_bary", C.c_void_p), ("segment_joints_host", C.POINTER(C.c_int32)), ("num_segments", C.c_int32), ("body_segment", C.c_int32),
                ("num_vertices", C.c_int32), ("num_joints", C.c_int32), ("joint_rows", C.c_int32), ("frames", C.c_int64), ("frames_per_sequence", C.c_int64), ("pose", C.c_void_p),
                ("adam_m", C.c_void_p), ("adam_v", C.c_void_p), ("joints_obs", C.c_void_p), ("n_obs_joints", C.c_int32), ("norm_mode", C.c_int32),
                ("norm_a", C.c_void_p), ("norm_b", C.c_void_p), ("n_steps", C.c_int32), ("weighted", C.c_int32), ("t_host", C.POINTER(C.c_float)),
                ("w_temp_host", C.POINTER(C.c_float)), ("w_data_host", C.POINTER(C.c_float)), ("w_prior_host", C.POINTER(C.c_float)),
                ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("adam_step0", C.c_int32),
                ("step0", C.c_uint32), ("seed", C.c_uint64), ("noise", C.c_void_p), ("scratch", C.c_void_p), ("loss_log", C.c_void_p),
                ("rot6d", C.c_int32)]


class LbsJointFold(C.Structure):
    """dposer_lbs_joint_fold (include/dposer_hip.h): which vertices the extra joints / landmarks read, as device tables."""
    _fields_ = [("vertex_slot", C.c_void_p), ("slot_vertex", C.c_void_p), ("slot_ptr", C.c_void_p), ("entry_row", C.c_void_p),
                ("entry_weight", C.c_void_p), ("n_slots", C.c_int32)]


class DPoserHipError(RuntimeError):
    pass


_lib = None
vp, i32, i64, u32, u64, f32, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_float, C.c_double
# dposer_ranges_final_fn(user, first_bucket, n_ranges, lo, hi, event): called from inside dposer_dsm_loss_fwd_bwd_notify
RANGES_FINAL_FN = C.CFUNCTYPE(None, vp, i32, i32, C.POINTER(i64), C.POINTER(i64), vp)

# name -> (restype, argtypes); every symbol include/dposer_hip.h declares
SIGNATURES = {
    "dposer_abi_version": (C.c_int, []),
    "dposer_last_error": (C.c_char_p, []),
    "dposer_scorefc_create": (C.c_int, [C.POINTER(ScoreFCDesc), C.POINTER(vp)]),
    "dposer_scorefc_destroy": (None, [vp]),
    "dposer_scorefc_num_params": (i64, [vp]),
    "dposer_scorefc_num_tensors": (i32, [vp]),
    "dposer_scorefc_tensor_offset": (i64, [vp, i32]),
    "dposer_scorefc_tensor_numel": (i64, [vp, i32]),
    "dposer_scorefc_nograd_ranges": (i32, [vp, C.POINTER(i64), C.POINTER(i64)]),
    "dposer_scorefc_packed_bytes": (i64, [vp, i32]),
    "dposer_scorefc_pack": (C.c_int, [vp, vp, vp, i32, vp]),
    "dposer_scorefc_workspace_bytes": (i64, [vp, i64, i32, i32]),
    "dposer_scorefc_forward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp]),
    "dposer_scorefc_forward_train": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, u64, u32, vp]),
    "dposer_scorefc_backward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, u64, u32, vp]),
    "dposer_em_sampler": (C.c_int, [vp, vp, vp, vp, C.POINTER(SdeDesc), vp, vp, vp, i32, vp, vp, vp, u64, vp, i32,
                                    vp, vp, i64, vp]),
    "dposer_em_sampler_steps": (C.c_int, [vp, vp, vp, vp, C.POINTER(SdeDesc), vp, vp, vp, i32, i32, vp, vp, vp, u64, vp, i32,
                                          vp, vp, i64, vp]),
    "dposer_langevin_step": (C.c_int, [vp, vp, vp, vp, C.POINTER(SdeDesc), vp, vp, f32, f32, f32, vp, u64, u32, vp, i32, f64, vp, vp,
                                       i64, vp]),
    "dposer_prior_loss": (C.c_int, [vp, vp, vp, vp, C.POINTER(SdeDesc), vp, vp, f32, i32, f32, vp, vp, vp, u64,
                                    u32, vp, vp, i64, vp]),
    "dposer_prior_table_build": (C.c_int, [vp, vp, vp, vp, C.POINTER(f32), i32, vp, i64, vp]),
    "dposer_prior_table_build_sde": (C.c_int, [vp, vp, vp, vp, C.POINTER(SdeDesc), C.POINTER(f32), i32, vp, i64, vp]),
    "dposer_prior_loss_tabled": (C.c_int, [vp, vp, vp, vp, C.POINTER(SdeDesc), vp, vp, f32, i32, i32, i32, f32, vp, vp, vp, u64, u32, vp, i64, vp]),
    "dposer_completion_optimize": (C.c_int, [vp, vp, vp, vp, C.POINTER(SdeDesc), vp, vp, vp, vp, vp, C.POINTER(f32), C.POINTER(i32),
                                             C.POINTER(f32), C.POINTER(f32), i32, f64, f64, f64, f64, vp, u64, u32, vp, vp, i64, vp]),
    "dposer_motion_denoise_scratch_bytes": (i64, [i64, i32, i32, i32]),
    "dposer_motion_denoise_optimize": (C.c_int, [C.POINTER(MotionDenoiseArgs), vp]),
    "dposer_dsm_loss_fwd_bwd": (C.c_int, [vp, vp, vp, vp, C.POINTER(SdeDesc), vp, vp, vp, f32, u64, u32, vp, vp,
                                          vp, vp, i64, vp]),
    "dposer_dsm_loss_fwd_bwd_bucketed": (C.c_int, [vp, vp, vp, vp, C.POINTER(SdeDesc), vp, vp, vp, f32, u64, u32, vp, vp,
                                                   vp, vp, i64, C.POINTER(vp), i32, vp]),
    "dposer_dsm_loss_fwd_bwd_notify": (C.c_int, [vp, vp, vp, vp, C.POINTER(SdeDesc), vp, vp, vp, f32, u64, u32, vp, vp,
                                                 vp, vp, i64, C.POINTER(vp), i32, RANGES_FINAL_FN, vp, vp]),
    "dposer_scorefc_grad_buckets": (i32, [vp, C.POINTER(i64), C.POINTER(i64), i32]),
    "dposer_event_create": (C.c_int, [C.POINTER(vp)]),
    "dposer_event_destroy": (None, [vp]),
    "dposer_stream_wait_event": (C.c_int, [vp, vp]),
    "dposer_adam_ema_clip_step": (C.c_int, [vp, vp, vp, vp, vp, i64, C.POINTER(i64), C.POINTER(i64), i32, f64, f64, f64, f64,
                                            f64, f64, i64, f64, vp, vp]),
    "dposer_grad_sqnorm": (C.c_int, [vp, i64, vp, vp]),
    "dposer_rk_combine_f64": (C.c_int, [vp, vp, C.POINTER(vp), C.POINTER(C.c_double), i32, C.c_double, i64, vp]),
    "dposer_pf_ode_rhs_begin": (C.c_int, [C.POINTER(SdeDesc), f32, vp, vp, vp, vp, vp, i64, i32, vp]),
    "dposer_pf_ode_rhs_end": (C.c_int, [C.POINTER(SdeDesc), f32, vp, vp, vp, vp, vp, i64, i32, vp]),
    "dposer_adam_ema_clip_step_presummed": (C.c_int, [vp, vp, vp, vp, vp, i64, C.POINTER(i64), C.POINTER(i64), i32, f64, f64, f64, f64,
                                                      f64, f64, i64, f64, vp, vp]),
    "dposer_adam_ema_clip_step_wd": (C.c_int, [vp, vp, vp, vp, vp, i64, C.POINTER(i64), C.POINTER(i64), i32, f64, f64, f64, f64,
                                               f64, f64, f64, i64, f64, vp, i32, vp]),
    "dposer_scorefc_adam_pack_step": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.POINTER(i64), C.POINTER(i64), i32, f64, f64, f64, f64, f64, f64, f64,
                                                i64, f64, vp, i32, vp]),
    "dposer_profile_enable": (None, [i32]),
    "dposer_profile_num_kinds": (i32, []),
    "dposer_profile_collect": (C.c_int, [C.POINTER(f64), C.POINTER(i64), C.POINTER(f64)]),
    "dposer_profile_kind_name": (None, [i32, C.c_char_p, i32]),
    "dposer_rot6d_to_rotmat": (C.c_int, [vp, vp, i64, vp]),
    "dposer_rodrigues": (C.c_int, [vp, vp, i64, vp]),
    "dposer_rot6d_to_axis_angle": (C.c_int, [vp, vp, i64, vp]),
    "dposer_rotmat_to_axis_angle": (C.c_int, [vp, vp, i64, vp]),
    "dposer_body_create": (C.c_int, [C.POINTER(BodyDesc), C.POINTER(i32), C.POINTER(vp)]),
    "dposer_body_destroy": (None, [vp]),
    "dposer_shape_blend_forward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i64, vp]),
    "dposer_shape_blend_scratch_floats": (i64, [i32, i32, i64]),
    "dposer_shape_blend_backward": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i64, vp]),
    "dposer_fk_joints": (C.c_int, [vp, C.POINTER(vp), C.POINTER(i32), i32, vp, i32, vp, vp, vp, i32, i64, vp]),
    "dposer_lbs_posedirs_packed_bytes": (i64, [vp]),
    "dposer_lbs_pack_posedirs": (C.c_int, [vp, vp, vp, vp]),
    "dposer_lbs_workspace_bytes": (i64, [vp, i64]),
    "dposer_lbs_forward": (C.c_int, [vp, vp, vp, C.POINTER(vp), C.POINTER(i32), i32, vp, i32, vp, i32, vp, vp, i32,
                                     vp, vp, vp, vp, vp, vp, i64, vp]),
    "dposer_lbs_forward_temporal_grad": (C.c_int, [vp, vp, vp, C.POINTER(vp), C.POINTER(i32), i32, vp, i32, vp, i32, vp, vp, i32,
                                                   vp, i64, f32, vp, vp, vp, i64, vp]),
    "dposer_lbs_temporal_in_backward_ok": (i32, [vp, i32, i64]),
    "dposer_lbs_forward_front": (C.c_int, [vp, vp, vp, C.POINTER(vp), C.POINTER(i32), i32, vp, i32, vp, vp, i64, vp]),
    "dposer_lbs_backward_temporal": (C.c_int, [vp, vp, vp, vp, C.POINTER(vp), C.POINTER(i32), i32, vp, i32, vp, i32, vp, vp, i32, vp, vp, vp,
                                               i64, f32, vp, vp, i64, C.POINTER(vp), i64, vp]),
    "dposer_lbs_posedirs_bwd_packed_bytes": (i64, [vp]),
    "dposer_lbs_pack_posedirs_bwd": (C.c_int, [vp, vp, vp, vp]),
    "dposer_lbs_backward_workspace_bytes": (i64, [vp, i64]),
    "dposer_lbs_prepare_joint_lists": (C.c_int, [vp, vp, vp, vp, vp]),
    "dposer_body_tuning_reload": (None, []),
    "dposer_scorefc_tuning_reload": (None, []),
    "dposer_scorefc_debug_set_dropout_masks": (C.c_int, [vp, vp, i64]),
    "dposer_mlp_create": (C.c_int, [C.POINTER(MlpDesc), C.POINTER(vp)]),
    "dposer_mlp_destroy": (None, [vp]),
    "dposer_mlp_num_params": (i64, [vp]),
    "dposer_mlp_packed_bytes": (i64, [vp]),
    "dposer_mlp_workspace_bytes": (i64, [vp, i64]),
    "dposer_mlp_pack": (C.c_int, [vp, vp, vp, vp]),
    "dposer_mlp_forward": (C.c_int, [vp, vp, vp, vp, vp, vp, i64, i32, i32, u64, u32, vp]),
    "dposer_mlp_backward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i64, i32, u64, u32, vp]),
    "dposer_lbs_backward": (C.c_int, [vp, vp, vp, vp, C.POINTER(vp), C.POINTER(i32), i32, vp, i32, vp, i32, vp, vp, i32, vp, vp, vp,
                                      vp, vp, i64, C.POINTER(vp), vp, vp, i64, vp]),
    "dposer_lbs_backward_fold": (C.c_int, [vp, vp, vp, vp, C.POINTER(vp), C.POINTER(i32), i32, vp, i32, vp, i32, vp, vp, i32, vp, vp, vp,
                                           vp, vp, i64, C.POINTER(LbsJointFold), C.POINTER(vp), vp, vp, i64, vp]),
}


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DPoserHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C dposer_amd/csrc` (hipcc --offload-arch=gfx950). dposer_amd has no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.dposer_abi_version() != 1:
            raise DPoserHipError("libdposer_hip.so ABI version mismatch")
        _lib = l
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().dposer_last_error().decode("utf-8", "replace")
        raise DPoserHipError(f"{what or 'libdposer_hip'} failed (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a torch tensor (or None -> NULL)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(t, name="tensor"):
    if not t.is_cuda:
        raise DPoserHipError(
            f"{name} lives on {t.device}: dposer_amd runs the hot path only as HIP kernels on an AMD GPU "
            "(no CPU fallback). Move the module / tensors to 'cuda'.")


PROFILE_EPI_KINDS = ("gn_fwd", "gn_fwd_train", "bias_silu", "rowmajor", "plain_ft", "gn_bwd_dgrad", "silu_bwd_dgrad", "wgrad",
                     "post_em_step")


# Every library routine that rewrites model parameters through ``.data`` (EMA copy_to / restore, re-flattening) bumps this epoch;
# together with the parameters' version counters it tells the engine whether the packed weights the fused optimizer step wrote are
# still those of the parameters (engine.ScoreEngine.packed).  Code that writes ``p.data`` by other means calls ``bump_param_epoch()``.
PARAM_EPOCH = [0]


def bump_param_epoch():
    PARAM_EPOCH[0] += 1


def profile_enable(on: bool, only: str = None):
    """HIP events around the GEMM launches; ``only`` = one of PROFILE_EPI_KINDS brackets that kind alone (cheap enough
    for a timed region: bracketing every launch costs about 3 % of a training step)."""
    lib().dposer_profile_enable((2 + PROFILE_EPI_KINDS.index(only) if only else 1) if on else 0)


def profile_pause():
    """Stop recording without discarding what has been recorded (profile_enable(True, ...) resumes)."""
    lib().dposer_profile_enable(-1)


def profile_collect():
    """{kernel kind name: (total_ms, launches, algorithmic_flops)} of the GEMM launches since the last collect."""
    l = lib()
    n = l.dposer_profile_num_kinds()
    ms, cnt, fl = (C.c_double * n)(), (C.c_int64 * n)(), (C.c_double * n)()
    check(l.dposer_profile_collect(ms, cnt, fl), "dposer_profile_collect")
    out = {}
    buf = C.create_string_buffer(128)
    for k in range(n):
        if cnt[k]:
            l.dposer_profile_kind_name(k, buf, 128)
            out[buf.value.decode()] = (ms[k], cnt[k], fl[k])
    return out
