"""ctypes loader for libpdmphip.so (C ABI: include/pdmp_hip.h).

There is no CPU fallback: if the library is missing or no GPU is visible every compute entry point
raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` (or ``make -C
pdmpflux.jl_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpdmphip.so")

OK, ERR_ARGUMENT, ERR_DIMENSION, ERR_RUNTIME, ERR_ITERCAP, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5


class ArgumentError(ValueError):
    """Julia ``ArgumentError`` (sample.jl:262-264,476-478; ZigZagSamplers.jl:62-68)."""


class DimensionMismatch(ValueError):
    """Julia ``DimensionMismatch`` (AbstractPDMP.jl:96-98)."""


class EngineError(RuntimeError):
    """HIP runtime failure, missing library / device, or iteration cap."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"[pdmphip {code}] {msg}")
        self.code = code


class Config(C.Structure):
    _fields_ = [
        ("sampler", C.c_int32), ("dim", C.c_int32), ("grid_size", C.c_int32),
        ("vectorized_bound", C.c_int32), ("signed_bound", C.c_int32), ("adaptive", C.c_int32),
        ("gaussian_velocity", C.c_int32), ("deriv_mode", C.c_int32), ("target_id", C.c_int32),
        ("device", C.c_int32), ("tmax", C.c_double), ("refresh_rate", C.c_double),
        ("target_params", C.POINTER(C.c_double)), ("n_target_params", C.c_int64),
        ("max_iter_per_event", C.c_int64),
        ("fe_mix_p", C.c_double), ("fe_ran_p", C.c_int32), ("fe_switch", C.c_int32), ("fe_positive", C.c_int32),
        ("fe_reserved", C.c_int32), ("rh_step_size", C.c_double), ("rh_phi", C.c_double),
        ("fe_speed_factor", C.c_double), ("rng_version", C.c_int32), ("reserved0", C.c_int32),
    ]


class Diag(C.Structure):
    _fields_ = [("horizon", C.c_void_p), ("ar", C.c_void_p), ("errored_bound", C.c_void_p),
                ("error_value_ar", C.c_void_p), ("rejected", C.c_void_p), ("hitting_horizon", C.c_void_p),
                ("is_active", C.c_void_p)]


class Timing(C.Structure):
    _fields_ = [("kernel_ms_total", C.c_double), ("kernel_ms_last", C.c_double), ("launches", C.c_int64)]


# every symbol include/pdmp_hip.h declares; tests check that the library exports all of them
SYMBOLS = (
    "pdmp_abi_version", "pdmp_last_error", "pdmp_device_count", "pdmp_create", "pdmp_create_from_source", "pdmp_get_config",
    "pdmp_destroy", "pdmp_get_group_lanes", "pdmp_suggest_group_lanes", "pdmp_set_group_lanes", "pdmp_reload_policy", "pdmp_last_launch_form", "pdmp_set_tape", "pdmp_clear_tape", "pdmp_sample_skeleton", "pdmp_sample_skeleton_device", "pdmp_sample_from_skeleton",
    "pdmp_sample_from_skeleton_device", "pdmp_sample_from_skeleton_dt", "pdmp_sample_from_skeleton_dt_device",
    "pdmp_sample_from_skeleton_prefix", "pdmp_sample_from_skeleton_prefix_device", "pdmp_sample_moments_device",
    "pdmp_close_at_time_device", "pdmp_compact_offsets_device", "pdmp_compact_history_device",
    "pdmp_sample_skeleton_until_device", "pdmp_close_chunk_device", "pdmp_until_advance_device", "pdmp_compact_chunk_device",
    "pdmp_multi_shard_range", "pdmp_multi_create", "pdmp_multi_destroy", "pdmp_multi_sample_skeleton", "pdmp_multi_sample_skeleton_device", "pdmp_multi_init_states", "pdmp_multi_get_timing",
    "pdmp_check_status", "pdmp_get_timing", "pdmp_reset_timing",
    "pdmp_init_states_device", "pdmp_device_alloc", "pdmp_device_free", "pdmp_device_alloc_error", "pdmp_device_alloc_streams", "pdmp_device_pool_trim", "pdmp_debug_rng", "pdmp_dense_batched_last_phases", "pdmp_dense_batched_last_columns", "pdmp_dense_batched_last_mode",
)

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError(ERR_RUNTIME, f"{LIB_PATH} not found: build the HIP extension first "
                                       "(__graft_entry__.build()); there is no CPU fallback")
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7.  Two HIP runtimes in one process cannot
    # both own the GPU, so when torch is installed make sure ITS runtime is the one already loaded:
    # our DT_NEEDED libamdhip64.so.7 then resolves to it by soname (else /opt/rocm's is used).
    if not os.environ.get("PDMP_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    vp, i64, u64, i32 = C.c_void_p, C.c_int64, C.c_uint64, C.c_int32
    L.pdmp_abi_version.restype = C.c_int
    L.pdmp_last_error.restype = C.c_char_p
    L.pdmp_device_count.restype = C.c_int
    L.pdmp_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.pdmp_create_from_source.argtypes = [C.POINTER(Config), C.c_char_p, i32, C.POINTER(vp)]
    L.pdmp_get_config.argtypes = [vp, C.POINTER(Config)]
    L.pdmp_destroy.argtypes = [vp]
    L.pdmp_get_group_lanes.argtypes = [vp]
    L.pdmp_get_group_lanes.restype = C.c_int
    L.pdmp_last_launch_form.argtypes = [vp]
    L.pdmp_last_launch_form.restype = C.c_int
    L.pdmp_suggest_group_lanes.argtypes = [vp, i64]
    L.pdmp_suggest_group_lanes.restype = C.c_int
    L.pdmp_set_group_lanes.argtypes = [vp, i32]
    L.pdmp_reload_policy.argtypes = [vp]
    L.pdmp_set_tape.argtypes = [vp, i64, i64, i64, i64, vp, vp, vp]
    L.pdmp_clear_tape.argtypes = [vp]
    L.pdmp_sample_skeleton.argtypes = [vp, i64, i64, i64, vp, vp, u64, vp, vp, vp, C.POINTER(Diag)]
    L.pdmp_sample_skeleton_device.argtypes = [vp, i64, i64, i64, vp, vp, u64, vp, vp, vp, C.POINTER(Diag), vp]
    f64 = C.c_double
    L.pdmp_sample_from_skeleton.argtypes = [vp, i64, i64, vp, vp, vp, vp, i64, i32, vp]
    L.pdmp_sample_from_skeleton_device.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp, i64, i32, vp, vp]
    L.pdmp_sample_from_skeleton_dt.argtypes = [vp, i64, i64, vp, vp, vp, vp, i64, f64, i32, vp, vp]
    L.pdmp_sample_from_skeleton_dt_device.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp, i64, f64, i32, vp, vp, vp]
    L.pdmp_sample_from_skeleton_prefix.argtypes = [vp, i64, i64, vp, vp, vp, vp, i64, f64, i64, i32, vp, vp]
    L.pdmp_sample_from_skeleton_prefix_device.argtypes = [vp, i64, i64, vp, vp, vp, vp, i64, f64, i64, i32, vp, vp, vp]
    L.pdmp_sample_moments_device.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp, i64, vp, vp, vp]
    L.pdmp_close_at_time_device.argtypes = [vp, i64, i64, vp, vp, vp, C.POINTER(Diag), f64, vp, vp]
    L.pdmp_compact_offsets_device.argtypes = [vp, i64, vp, vp, vp]
    L.pdmp_compact_history_device.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp, C.POINTER(Diag), vp, vp, vp, C.POINTER(Diag), vp]
    L.pdmp_sample_skeleton_until_device.argtypes = [vp, i64, i64, i64, vp, vp, u64, f64, i32, vp, vp, vp, vp, C.POINTER(Diag), vp, vp]
    L.pdmp_close_chunk_device.argtypes = [vp, i64, i64, vp, vp, vp, C.POINTER(Diag), f64, vp, vp, vp]
    L.pdmp_until_advance_device.argtypes = [vp, i64, vp, i64, vp, vp, vp, vp, vp, vp]
    L.pdmp_compact_chunk_device.argtypes = [vp, i64, i64, vp, i64, vp, vp, vp, vp, vp, vp, C.POINTER(Diag), vp, vp, vp, C.POINTER(Diag), vp]
    L.pdmp_multi_shard_range.argtypes = [i64, i32, i32, C.POINTER(i64), C.POINTER(i64)]
    L.pdmp_multi_create.argtypes = [C.POINTER(Config), i32, C.POINTER(i32), C.POINTER(vp)]
    L.pdmp_multi_destroy.argtypes = [vp]
    L.pdmp_multi_sample_skeleton.argtypes = [vp, i64, i64, i64, vp, vp, u64, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.pdmp_multi_get_timing.argtypes = [vp, C.POINTER(f64), C.POINTER(f64)]
    L.pdmp_multi_sample_skeleton_device.argtypes = [vp, i64, i64, i64, C.POINTER(vp), C.POINTER(vp), u64, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.pdmp_multi_init_states.argtypes = [vp, i64, i64, u64, C.c_double, C.c_double, i32, C.POINTER(vp), C.POINTER(vp)]
    L.pdmp_check_status.argtypes = [vp, C.POINTER(i64)]
    L.pdmp_get_timing.argtypes = [vp, C.POINTER(Timing)]
    L.pdmp_reset_timing.argtypes = [vp]
    L.pdmp_debug_rng.argtypes = [vp, u64, i64, i32, i64, vp]
    L.pdmp_device_alloc.argtypes = [i32, C.c_size_t, C.c_size_t, C.POINTER(vp)]
    L.pdmp_device_free.argtypes = [vp]
    L.pdmp_device_alloc_error.restype = C.c_char_p
    L.pdmp_device_alloc_streams.argtypes = [i32, i32, C.POINTER(C.c_size_t), C.c_size_t, C.c_size_t, i32, C.POINTER(vp), C.POINTER(C.c_float), C.POINTER(i32)]
    L.pdmp_init_states_device.argtypes = [vp, i64, i64, u64, C.c_double, C.c_double, i32, vp, vp, vp]
    L.pdmp_dense_batched_last_phases.restype = C.c_longlong
    L.pdmp_dense_batched_last_columns.restype = C.c_longlong
    L.pdmp_dense_batched_last_mode.restype = C.c_int
    if L.pdmp_abi_version() != 11:
        raise EngineError(ERR_RUNTIME, "ABI version mismatch")
    _lib = L
    return L


def check(rc: int) -> None:
    if rc == OK:
        return
    msg = lib().pdmp_last_error().decode(errors="replace")
    if rc == ERR_ARGUMENT:
        raise ArgumentError(msg)
    if rc == ERR_DIMENSION:
        raise DimensionMismatch(msg)
    raise EngineError(rc, msg)
