"""Snapshot files read by the library's own reader (bl_snapshot_open): Athena++ .athdf.

Mirrors what the reference's main() does with SimulationReader (src/blacklight.cpp:93, :181-186):
construct from the input parameters, Read(snapshot), hand the arrays to the radiation integrator.
"""
import ctypes as C

import numpy as np

from . import _capi
from .params import Params


class Snapshot:
    """One file of the run: `Snapshot(params, snapshot=0)`; pass it to Context.set_grid(). `file_number`: that file of a
    numbered series instead (bl_snapshot_open_number - what the slow-light window reads, simulation_reader.cpp:225-232)."""

    def __init__(self, params: Params, snapshot: int = 0, file_number: int = None):
        L = _capi.lib()
        self._lib = L
        self._params = params   # keeps simulation_kappa_name etc. alive
        handle = C.c_void_p()
        err = C.create_string_buffer(1024)
        if file_number is None:
            rc = L.bl_snapshot_open(params.ptr, int(snapshot), C.byref(handle), err, len(err))
        else:
            rc = L.bl_snapshot_open_number(params.ptr, int(file_number), C.byref(handle), err, len(err))
        if rc != 0:
            raise _capi.BlacklightError(rc, err.value.decode())
        self._h = handle

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bl_snapshot_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def desc(self):
        """bl_grid_desc view of the arrays this object owns (valid until close())."""
        return self._lib.bl_snapshot_grid(self._h).contents

    @property
    def time(self):
        return self._lib.bl_snapshot_time(self._h)

    @property
    def warnings(self):
        return self._lib.bl_snapshot_warnings(self._h).decode()

    @property
    def file(self):
        return self._lib.bl_snapshot_file(self._h).decode()

    @property
    def blocks(self):
        """(levels [n_b], logical locations [n_b][3]) of the MeshBlocks."""
        levels = C.POINTER(C.c_int32)()
        locations = C.POINTER(C.c_int32)()
        n = self._lib.bl_snapshot_blocks(self._h, C.byref(levels), C.byref(locations))
        return (np.ctypeslib.as_array(levels, (n,)).copy(), np.ctypeslib.as_array(locations, (n, 3)).copy())

    def arrays(self):
        """Copies of the arrays as numpy: prim [n_var][n_b][n_k][n_j][n_i] float32 and the six coordinate
        arrays [n_b][...] float64, plus the variable indices."""
        d = self.desc()
        shape = (d.n_var, d.n_blocks, d.n_k, d.n_j, d.n_i)
        out = {"prim": np.ctypeslib.as_array(C.cast(d.prim, C.POINTER(C.c_float)), shape).copy()}
        for name, n in (("x1f", d.n_i + 1), ("x2f", d.n_j + 1), ("x3f", d.n_k + 1), ("x1v", d.n_i), ("x2v", d.n_j),
                        ("x3v", d.n_k)):
            out[name] = np.ctypeslib.as_array(C.cast(getattr(d, name), C.POINTER(C.c_double)), (d.n_blocks, n)).copy()
        out["indices"] = {k: getattr(d, k) for k in ("ind_rho", "ind_pgas", "ind_kappa", "ind_uu1", "ind_uu2", "ind_uu3",
                                                      "ind_bb1", "ind_bb2", "ind_bb3")}
        return out
