"""ctypes access to the CPU oracle (oracle/libbl_oracle*.so). TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")


class Extra(C.Structure):
    _fields_ = [
        ("num_threads", C.c_int32), ("dump_ray", C.c_int64),
        ("dump_pos", C.c_void_p), ("dump_dir", C.c_void_p), ("dump_len", C.c_void_p),
        ("dump_num", C.c_int32),
        ("n_samples", C.c_int64), ("n_gathers", C.c_int64), ("n_flagged", C.c_int64),
        ("max_sample_num", C.c_int32), ("seconds", C.c_double),
        ("slow_n", C.c_int32), ("slow_grids", C.c_void_p), ("slow_times", C.c_void_p),
        ("slow_snapshot_time", C.c_double), ("slow_count", C.c_int64 * 4), ("slow_val", C.c_double * 4),
        ("define_kappa_aa_high_i", C.c_int32),
    ]


def _host_has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            return " fma " in f.read()
    except OSError:
        return False


_libs = {}


def load(variant="blmath"):
    """variant: 'blmath' (tier B, pinned math) or 'libm' (tier A, host glibc)."""
    if variant in _libs:
        return _libs[variant]
    if variant == "libm":
        name = "libbl_oracle_libm.so"
    else:
        name = "libbl_oracle_fma.so" if _host_has_fma() else "libbl_oracle.so"
    path = os.path.join(ORACLE_DIR, name)
    if not os.path.exists(path):
        subprocess.run(["make", "-C", ORACLE_DIR, "oracle", "oracle_libm", "oracle_fma"], check=True,
                       capture_output=True)
    L = C.CDLL(path)
    L.blo_build_info.restype = C.c_char_p
    L.blo_image_num_quantities.argtypes = [C.c_void_p]
    L.blo_image_num_frequencies.argtypes = [C.c_void_p]
    _libs[variant] = L
    return L


def render(params_ptr, grid_desc, desc_cls, camera_frame_cls, *, n_rays, level=0, block_locs=None,
           pixel_map=None, variant="blmath", num_threads=0, dump_ray=-1, max_steps=0, n_freq=1,
           want_camera=False, n_render=0, slow=None, define_kappa=False):
    """Run the oracle. Returns dict(image, sample_num, sample_flags, frame, frequencies, extra...).
    n_render > 0: also the false-colour renderings, (n_render, 3, n_rays).
    slow = dict(grids=[bl_grid_desc, ...] latest first, times=[...], snapshot_time=t): slow light.
    define_kappa: unpolarized kappa-distribution electrons with kappa_aa_high_i as in polarized runs (blo_extra)."""
    L = load(variant)
    n_q = L.blo_image_num_quantities(params_ptr)
    image = np.zeros((max(n_q, 1), n_rays), dtype=np.float64)
    sample_num = np.zeros(n_rays, dtype=np.int32)
    sample_flags = np.zeros(n_rays, dtype=np.uint8)
    d = desc_cls()
    d.level = level
    keep = []
    if block_locs is not None:
        bl = np.ascontiguousarray(block_locs, dtype=np.int32)
        keep.append(bl)
        d.block_locs = bl.ctypes.data_as(C.c_void_p)
        d.n_blocks = bl.shape[0]
    if pixel_map is not None:
        pm = np.ascontiguousarray(pixel_map, dtype=np.int32)
        keep.append(pm)
        d.pixel_map = pm.ctypes.data_as(C.c_void_p)
    d.n_rays = n_rays
    d.outputs_on_device = 0
    d.image = image.ctypes.data_as(C.c_void_p)
    d.sample_num = sample_num.ctypes.data_as(C.c_void_p)
    d.sample_flags = sample_flags.ctypes.data_as(C.c_void_p)
    camera_pos = camera_dir = None
    if want_camera:
        camera_pos = np.zeros((n_rays, 4))
        camera_dir = np.zeros((n_rays, 4))
        d.camera_pos = camera_pos.ctypes.data_as(C.c_void_p)
        d.camera_dir = camera_dir.ctypes.data_as(C.c_void_p)
    rendering = None
    if n_render > 0:
        rendering = np.zeros((n_render, 3, n_rays))
        d.render = rendering.ctypes.data_as(C.c_void_p)
    frame = camera_frame_cls()
    # (the oracle writes the whole frequency list: sized from the parameters, whatever the caller said - a sweep with several
    # frequencies and the default n_freq once wrote past an 8-byte buffer, found by tools/gpu_fuzz_tiers.py under ASan)
    freqs = np.zeros(max(n_freq, L.blo_image_num_frequencies(params_ptr), 1))
    extra = Extra()
    extra.num_threads = num_threads
    extra.define_kappa_aa_high_i = 1 if define_kappa else 0
    extra.dump_ray = dump_ray
    dump = None
    if dump_ray >= 0:
        dump = dict(pos=np.zeros((max_steps, 4)), dir=np.zeros((max_steps, 4)), len=np.zeros(max_steps))
        extra.dump_pos = dump["pos"].ctypes.data_as(C.c_void_p)
        extra.dump_dir = dump["dir"].ctypes.data_as(C.c_void_p)
        extra.dump_len = dump["len"].ctypes.data_as(C.c_void_p)
    if slow is not None:
        grid_ptrs = (C.c_void_p * len(slow["grids"]))(*[C.addressof(g) for g in slow["grids"]])
        times = np.ascontiguousarray(slow["times"], dtype=np.float64)
        keep += [grid_ptrs, times]
        extra.slow_n = len(slow["grids"])
        extra.slow_grids = C.cast(grid_ptrs, C.c_void_p)
        extra.slow_times = times.ctypes.data_as(C.c_void_p)
        extra.slow_snapshot_time = float(slow["snapshot_time"])
    err = C.create_string_buffer(1024)
    grid_ptr = C.byref(grid_desc) if grid_desc is not None else None
    rc = L.blo_render(params_ptr, grid_ptr, C.byref(d), C.byref(frame),
                      freqs.ctypes.data_as(C.c_void_p), C.byref(extra), err, C.c_size_t(len(err)))
    if rc != 0:
        raise RuntimeError(f"oracle failed ({rc}): {err.value.decode()}")
    out = dict(image=image[:n_q], sample_num=sample_num, sample_flags=sample_flags, frame=frame,
               frequencies=freqs, n_samples=extra.n_samples, n_gathers=extra.n_gathers,
               n_flagged=extra.n_flagged, max_sample_num=extra.max_sample_num, seconds=extra.seconds,
               camera_pos=camera_pos, camera_dir=camera_dir, rendering=rendering,
               slow_count=list(extra.slow_count), slow_val=list(extra.slow_val))
    if dump is not None:
        n = extra.dump_num
        out["dump"] = dict(pos=dump["pos"][:n], dir=dump["dir"][:n], len=dump["len"][:n])
    return out
