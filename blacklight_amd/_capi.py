"""ctypes binding of the C-ABI in include/blacklight_amd.h (libblacklight_amd.so).

Plumbing only: the product is the HIP library. Loading fails loudly if the library has not been
built (python -c "import __graft_entry__ as g; g.build()"); there is no CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BLACKLIGHT_AMD_LIB: another build of the same library (kernel A/B runs, tools/gpu_ab.sh)
LIB_PATH = os.environ.get("BLACKLIGHT_AMD_LIB") or os.path.join(_HERE, "libblacklight_amd.so")

BL_OK, BL_E_INPUT, BL_E_MISSING, BL_E_UNSUPPORTED, BL_E_DEVICE, BL_E_ARG, BL_E_STATE = range(7)


class GridDesc(C.Structure):
    _fields_ = [
        ("n_blocks", C.c_int32), ("n_i", C.c_int32), ("n_j", C.c_int32), ("n_k", C.c_int32),
        ("n_var", C.c_int32),
        ("prim", C.c_void_p),
        ("x1f", C.c_void_p), ("x2f", C.c_void_p), ("x3f", C.c_void_p),
        ("x1v", C.c_void_p), ("x2v", C.c_void_p), ("x3v", C.c_void_p),
        ("ind_rho", C.c_int32), ("ind_pgas", C.c_int32), ("ind_kappa", C.c_int32),
        ("ind_uu1", C.c_int32), ("ind_uu2", C.c_int32), ("ind_uu3", C.c_int32),
        ("ind_bb1", C.c_int32), ("ind_bb2", C.c_int32), ("ind_bb3", C.c_int32),
        ("plasma_gamma", C.c_double), ("plasma_gamma_i", C.c_double), ("plasma_gamma_e", C.c_double),
        ("levels", C.c_void_p), ("locations", C.c_void_p), ("n_3_root", C.c_int32),
        ("sks_map", C.c_void_p), ("sks_map_n1", C.c_int32), ("sks_map_n2", C.c_int32),
        ("sks_map_r_in", C.c_double), ("sks_map_dr", C.c_double), ("sks_map_dtheta", C.c_double),
        ("simulation_bounds", C.c_double * 6),
    ]


class CameraFrame(C.Structure):
    _fields_ = [(name, C.c_double * 4) for name in
                ("cam_x", "u_con", "u_cov", "norm_con", "norm_con_c", "hor_con_c", "vert_con_c")] + \
               [(name, C.c_double) for name in ("bh_m", "bh_a", "r_horizon", "r_terminate", "mass_msun")]


class RenderDesc(C.Structure):
    _fields_ = [
        ("level", C.c_int32), ("n_blocks", C.c_int32), ("block_locs", C.c_void_p),
        ("n_rays", C.c_int64), ("pixel_map", C.c_void_p), ("outputs_on_device", C.c_int32),
        ("image", C.c_void_p), ("sample_num", C.c_void_p), ("sample_flags", C.c_void_p),
        ("camera_pos", C.c_void_p), ("camera_dir", C.c_void_p), ("render", C.c_void_p),
    ]


# BL_SWITCH_* of include/blacklight_amd.h: measurement switches (bl_stats.switches, bl_debug_set_switches)
SWITCHES = {"TENSOR_TRANSPORT": 1 << 0, "SPLIT_RECORDS": 1 << 1, "RECORD_EVERY_STEP": 1 << 2,
            "GENERAL_LOCATE": 1 << 4, "LANE_TRANSFER": 1 << 5, "NO_FUSED_LOCATE": 1 << 6, "SAMPLE_RECORDS": 1 << 8, "QUAD_EVERY_RAY": 1 << 11}

BL_MAX_LEVELS = 16


class OutputLevel(C.Structure):
    _fields_ = [("n_blocks", C.c_int32), ("block_locs", C.c_void_p), ("image", C.c_void_p), ("camera", C.c_void_p),
                ("render", C.c_void_p)]


class OutputDesc(C.Structure):
    _fields_ = [("adaptive_num_levels", C.c_int32), ("level", OutputLevel * (BL_MAX_LEVELS + 1)),
                ("snapshot", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [
        ("n_rays", C.c_int64), ("n_samples", C.c_int64), ("n_samples_emitted", C.c_int64),
        ("n_gathers", C.c_int64), ("n_flagged", C.c_int64), ("max_sample_num", C.c_int32),
        ("n_chunks", C.c_int32), ("algorithmic_bytes", C.c_double),
        ("ms_geodesic", C.c_float), ("ms_shade", C.c_float), ("ms_transfer", C.c_float),
        ("ms_total", C.c_float),
        ("launches_geodesic", C.c_int32), ("launches_shade", C.c_int32),
        ("launches_transfer", C.c_int32),
        ("ms_locate", C.c_float), ("ms_wall", C.c_float), ("launches_locate", C.c_int32),
        ("arithmetic", C.c_int32), ("n_deferred", C.c_int64), ("n_undefined", C.c_int64),
        ("switches", C.c_uint32), ("fused_variant", C.c_int32), ("n_parked", C.c_int64),
        ("composed_maps", C.c_int32), ("tail_policy", C.c_int32), ("geodesics_reused", C.c_int32), ("sampling_reused", C.c_int32),
    ]


_lib = None


def lib():
    """Load libblacklight_amd.so (once) and declare prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP library first "
            "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.bl_params_clear.argtypes = [C.c_void_p]
    L.bl_params_sizeof.restype = C.c_size_t
    L.bl_params_set_line.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]
    L.bl_params_read_file.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
    L.bl_params_get.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.bl_params_get_string.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]
    L.bl_init.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    L.bl_set_grid.argtypes = [C.c_void_p, C.POINTER(GridDesc)]
    L.bl_image_num_quantities.argtypes = [C.c_void_p]
    L.bl_camera_frame_get.argtypes = [C.c_void_p, C.POINTER(CameraFrame)]
    L.bl_frequencies.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
    L.bl_set_scratch_limit.argtypes = [C.c_void_p, C.c_uint64]
    L.bl_set_overlap.argtypes = [C.c_void_p, C.c_int]
    L.bl_set_arithmetic.argtypes = [C.c_void_p, C.c_int]
    L.bl_set_reproducible.argtypes = [C.c_void_p, C.c_int]
    L.bl_set_tail_policy.argtypes = [C.c_void_p, C.c_int]
    L.bl_set_geodesic_reuse.argtypes = [C.c_void_p, C.c_int]
    L.bl_host_alloc.argtypes = [C.c_void_p, C.c_size_t]
    L.bl_host_alloc.restype = C.c_void_p
    L.bl_host_free.argtypes = [C.c_void_p, C.c_void_p]
    L.bl_set_caller_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.bl_device_count.restype = C.c_int
    L.bl_set_undefined_policy.argtypes = [C.c_void_p, C.c_int]
    L.bl_debug_set_guard_band.argtypes = [C.c_void_p, C.c_double]
    L.bl_debug_set_switches.argtypes = [C.c_void_p, C.c_uint32]
    L.bl_debug_math.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bl_render.argtypes = [C.c_void_p, C.POINTER(RenderDesc)]
    L.bl_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    L.bl_last_error.argtypes = [C.c_void_p]
    L.bl_last_error.restype = C.c_char_p
    L.bl_last_global_error.restype = C.c_char_p
    L.bl_warnings.argtypes = [C.c_void_p]
    L.bl_warnings.restype = C.c_char_p
    L.bl_adaptive_refine.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.POINTER(C.c_int32), C.c_void_p]
    L.bl_write_output.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(OutputDesc)]
    L.bl_free.argtypes = [C.c_void_p]
    L.bl_snapshot_open.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
    L.bl_snapshot_grid.argtypes = [C.c_void_p]
    L.bl_snapshot_grid.restype = C.POINTER(GridDesc)
    L.bl_snapshot_time.argtypes = [C.c_void_p]
    L.bl_snapshot_time.restype = C.c_double
    L.bl_snapshot_warnings.argtypes = [C.c_void_p]
    L.bl_snapshot_warnings.restype = C.c_char_p
    L.bl_snapshot_file.argtypes = [C.c_void_p]
    L.bl_snapshot_file.restype = C.c_char_p
    L.bl_snapshot_blocks.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.POINTER(C.c_int32))]
    L.bl_snapshot_close.argtypes = [C.c_void_p]
    L.bl_snapshot_open_number.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
    L.bl_slow_light_read.argtypes = [C.c_void_p, C.c_int]
    L.bl_set_grid_slice.argtypes = [C.c_void_p, C.c_int, C.POINTER(GridDesc), C.c_double]
    L.bl_shift_grid_slices.argtypes = [C.c_void_p, C.c_int]
    L.bl_set_snapshot.argtypes = [C.c_void_p, C.c_int]
    L.bl_warnings_clear.argtypes = [C.c_void_p]
    L.bl_build_info.restype = C.c_char_p
    _lib = L
    return L


class BlacklightError(RuntimeError):
    """Raised with the reference-style 'Error: ...' text of a failed C-ABI call."""

    def __init__(self, code, message):
        super().__init__(message.strip() or f"blacklight_amd error code {code}")
        self.code = code
