"""ctypes binding of libeamrl_hip.so (include/eamrl.h).  There is no fallback: if the library is
missing or cannot be loaded every native op raises."""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# EAMRL_HIP_LIB: development override (instrumented builds made by tools/); the default is the in-tree library
LIB_PATH = os.environ.get("EAMRL_HIP_LIB") or os.path.join(_PKG, "lib", "libeamrl_hip.so")

ENV_TSP, ENV_CVRP, ENV_SDVRP, ENV_PCTSP, ENV_OP, ENV_CVRPTW = 0, 1, 2, 3, 4, 5
GREEDY, SAMPLE, EVALUATE = 0, 1, 2
NORM_BATCH_EVAL, NORM_INSTANCE = 0, 1
ST_NAN_LOGITS, ST_INFEASIBLE, ST_STEP_OVERRUN = 1, 2, 4

_vp, _i64, _i32, _f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float


class Cache(C.Structure):
    """struct eamrl_cache"""
    _fields_ = [("K", _vp), ("V", _vp), ("Lp", _vp), ("Pa", _vp), ("Pb", _vp), ("cvec", _vp), ("gctx", _vp),
                ("ld", _i64), ("B", _i64), ("M", C.c_int32), ("E", C.c_int32), ("H", C.c_int32),
                ("dyn", _vp)]


class EncoderLayer(C.Structure):
    """struct eamrl_encoder_layer"""
    _fields_ = [(n, _vp) for n in ("Wqkv", "bqkv", "Wo", "bo", "W1", "b1", "W2", "b2", "n1_gamma", "n1_beta", "n1_mean",
                                   "n1_var", "n2_gamma", "n2_beta", "n2_mean", "n2_var")]


class EncoderCache(C.Structure):
    """struct eamrl_encoder_cache"""
    _fields_ = [("Wc", _vp), ("WoutT", _vp), ("out", _vp), ("ld", _i64), ("nproj", C.c_int32), ("Wg", _vp), ("gctx", _vp)]


class EncoderInit(C.Structure):
    """struct eamrl_encoder_init"""
    _fields_ = [("feat", _vp), ("F", C.c_int32), ("W", _vp), ("b", _vp), ("depot", _vp), ("depot_ld", _i64), ("Wd", _vp),
                ("bd", _vp), ("init_out", _vp)]


class Reeval(C.Structure):
    """struct eamrl_reeval"""
    _fields_ = [("K", _vp), ("V", _vp), ("Lp", _vp), ("Pa", _vp), ("Pb", _vp), ("ld", _i64),
                ("gctx", _vp), ("Cvec", _vp), ("NC", C.c_int32),
                ("idxA", _vp), ("idxB", _vp), ("sc", _vp), ("maskbits", _vp), ("actions", _vp),
                ("B", _i64), ("R", _i64), ("S", C.c_int32), ("T", C.c_int32), ("M", C.c_int32), ("tstart", C.c_int32),
                ("nchunk", C.c_int32), ("clip", _f32), ("temp", _f32),
                ("logp", _vp), ("lse", _vp), ("glogp", _vp), ("dheads", _vp), ("heads", _vp), ("heads_T", C.c_int32),
                ("entropy", _vp),
                ("dK", _vp), ("dV", _vp), ("dLp", _vp), ("dPa", _vp), ("dPb", _vp), ("ldg", _i64),
                ("dgctx", _vp), ("dCvec", _vp), ("rem", _vp), ("dyn", _vp), ("ddyn", _vp),
                ("nkc", C.c_int32), ("scratch", _vp), ("mc_koff", C.c_int32), ("mc_mstride", C.c_int32),
                ("mc_part_o", _vp), ("mc_part_s", _vp), ("mc_gstat", _vp), ("mc_lpart", _vp), ("mc_rsq", _vp), ("mc_dq", _vp)]


class State(C.Structure):
    """struct eamrl_state"""
    _fields_ = [("first", _vp), ("cur", _vp), ("istep", _vp), ("used", _vp), ("vcap", _vp), ("demand", _vp),
                ("mask", _vp), ("visited", _vp), ("done", _vp), ("rem", _vp), ("locs", _vp),
                ("time", _vp), ("tw", _vp), ("dur", _vp), ("heads_out", _vp)]


# name -> argtypes (all return int unless listed in _RESTYPES); mirrors include/eamrl.h one to one
PROTOTYPES = {
    "eamrl_version": [],
    "eamrl_last_error": [],
    "eamrl_debug_set": [_i32, _i32],
    "eamrl_tsp_step": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp],
    "eamrl_cvrp_mask": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp],
    "eamrl_cvrp_step_mask": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp],
    "eamrl_sdvrp_step_mask": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp],
    "eamrl_cvrptw_step_mask": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp],
    "eamrl_cvrptw_check_time": [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp],
    "eamrl_op_step_mask": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp],
    "eamrl_op_reward": [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp],
    "eamrl_op_check_solution": [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp],
    "eamrl_pctsp_step_mask": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp],
    "eamrl_pctsp_reward": [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp],
    "eamrl_linear": [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _vp],
    "eamrl_linear_bn": [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _f32, _vp],
    "eamrl_matmul_right": [_vp, _i64, _vp, _vp, _i64, _i64, _i32, _i32, _vp],
    "eamrl_linear_wgrad_scratch": [_i64, _i32, _i32],
    "eamrl_linear_wgrad": [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _i64, _vp],
    "eamrl_augment_xy": [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _f32, _vp],
    "eamrl_small_linear_wgrad_scratch": [_i64, _i32],
    "eamrl_small_linear_wgrad": [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _i64, _vp],
    "eamrl_batchnorm_backward_scratch": [_i64, _i32],
    "eamrl_batchnorm_backward": [_vp, _vp, _vp, _vp, _vp, _f32, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _vp],
    "eamrl_mha_encoder": [_vp, _vp, _i64, _i32, _i32, _i32, _vp],
    "eamrl_mha_encoder_backward_supported": [_i32, _i32, _i32],
    "eamrl_mha_encoder_backward": [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp],
    "eamrl_normalize": [_vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _f32, _vp],
    "eamrl_batchnorm_train": [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _i64, _vp],
    "eamrl_pointer_attention": [_vp, _vp, _vp, _vp, _i64, _vp, _i32, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp],
    "eamrl_pack_linear_weight": [_vp, _vp, _i32, _i32, _vp],
    "eamrl_encoder_fused_supported": [_i32, _i32, _i32, _i32, _i32],
    "eamrl_encoder_fused": [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp],
    "eamrl_encoder_fused_init": [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp],
    "eamrl_reeval_supported": [_i32, _i32, _i32],
    "eamrl_reeval_scratch_floats": [_i64, _i32, _i32],
    "eamrl_pack_mask_bits_chunked": [_vp, _vp, _i64, _i32, _i32, _i32, _vp],
    "eamrl_tsp_mask_bits_chunked": [_vp, _vp, _i64, _i32, _i32, _vp],
    "eamrl_reeval_forward": [_vp, _vp],
    "eamrl_reeval_backward": [_vp, _vp],
    "eamrl_pack_mask_bits": [_vp, _vp, _i64, _i32, _i32, _i32, _vp],
    "eamrl_tsp_mask_bits": [_vp, _vp, _i64, _i32, _i32, _vp],
    "eamrl_mean_nodes": [_vp, _vp, _i64, _i32, _i32, _vp],
    "eamrl_am_decode_step": [_i32, C.POINTER(Cache), C.POINTER(State), _i64, _i32, _vp, _vp, _f32, _f32, _i32, _f32, _i32,
                             _vp, _vp, _vp, _vp, _vp, _vp],
    "eamrl_am_rollout": [_i32, C.POINTER(Cache), C.POINTER(State), _i64, _i32, _vp, _vp, _i32, _f32, _f32, _i32, _f32, _i32,
                         _vp, _vp, _vp, _vp, _vp],
    "eamrl_exp1_noise": [C.c_uint64, _vp, _vp, _i64, _i32, _i32, _vp],
    "eamrl_rollout_rng_native": [_i32, C.POINTER(Cache), _i64],
    "eamrl_am_rollout_seeded": [_i32, C.POINTER(Cache), C.POINTER(State), _i64, C.c_uint64, _vp, _vp, _f32, _f32, _i32,
                                _vp, _vp, _vp, _vp, _vp],
    "eamrl_tour_length": [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp],
    "eamrl_sum_logp": [_vp, _i64, _vp, _i64, _i32, _vp],
    "eamrl_rollout_finish": [_i32, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp],
    "eamrl_multi_copy": [_i32, _vp, _vp, _vp, _vp],
    "eamrl_instance_norm_forward": [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _f32, _vp],
    "eamrl_instance_norm_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp],
    "eamrl_replay_states": [_i32, C.POINTER(State), _i64, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _vp],
    "eamrl_replay_states_sdvrp": [C.POINTER(State), _i64, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp],
    "eamrl_check_solution": [_i32, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp],
    "eamrl_beam_topk": [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp],
    "eamrl_ea_cvrp_run": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, C.c_double, C.c_double, C.c_double, _i32,
                          _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "eamrl_ea_prize_run": [_i32, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, C.c_double, C.c_double, C.c_double, _i32,
                           _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "eamrl_ea_tsp_run": [_vp, _vp, _vp, _i64, _i32, _i32, _i32, C.c_double, C.c_double, C.c_double, _vp, _vp, _vp, _vp, _vp],
}
_RESTYPES = {"eamrl_last_error": C.c_char_p, "eamrl_linear_wgrad_scratch": C.c_int64,
             "eamrl_small_linear_wgrad_scratch": C.c_int64, "eamrl_batchnorm_backward_scratch": C.c_int64,
             "eamrl_reeval_scratch_floats": C.c_int64}

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def load():
    """Load libeamrl_hip.so.  torch must be imported first so that its bundled HIP runtime
    (same SONAME libamdhip64.so.7) is the one the library binds to: one runtime per process."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (loads libamdhip64 with the SONAME we link against)

    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m eam_rl4co_amd.build` (hipcc, gfx950). "
            "eam_rl4co_amd has no CPU or PyTorch fallback for the rollout path.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, argtypes in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().eamrl_last_error()
        raise RuntimeError(f"{what or 'eamrl'} failed ({rc}): {msg.decode() if msg else '?'}")
