"""Synthetic GRMHD snapshot: native restatement of the reference's workload generator.

Follows reference scripts/generate_mock_simulation.py:26-77 (grid, cutoff, perturbation, cell
values) and :158-198 (what the Athena++ .athdf branch stores: float32 primitives in the order
rho, press, vel1, vel2, vel3, Bcc1, Bcc2, Bcc3 with index [var][block][k=phi][j=theta][i=r], and
float32 face / cell-centre coordinates), then applies what the reference's reader does on load
(src/simulation_reader/simulation_reader.cpp:615-620: coordinates promoted float32 -> double;
:724-758: theta / phi end faces snapped to [0, pi] / [0, 2 pi] only when off by more than a tenth
of a cell, which never triggers for this grid).

The arrays this returns are exactly what RadiationIntegrator::ObtainGridData() would see
(simulation_sampling.cpp:38-78), i.e. the input of bl_set_grid().

numpy's exp / power can differ from the script's run-time numpy by one ulp in a few 1-D factors;
after rounding to float32 that flips essentially no stored value (tests/test_mock.py pins the
small-grid output against arrays produced by the script itself).
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _capi

# argparse defaults of the script (:352-425)
DEFAULTS = dict(
    r_min=2.0 * 25.0 ** (-1.0 / 75.0), r_max=2.0 * 25.0 ** (76.0 / 75.0), n_r=77, n_th=64, n_ph=128,
    rho_amp=1.0, rho_r_power=0.5, rho_th_scale=np.pi / 8.0, rho_floor=1.0e-8,
    pgas_amp=0.1, pgas_r_power=1.25, pgas_th_scale=np.pi / 8.0, pgas_floor=1.0e-9,
    uph_amp=None, uph_r_power=1.5, uph_th_scale=np.pi / 8.0,
    Bph_amp=0.2, Bph_r_power=1.75, Bph_th_scale=np.pi / 8.0, Bph_no_flip=False,
    Bz_amp=0.02, Bz_R_power=0.625,
    cutoff_r_min=2.0, cutoff_r_max=50.0, cutoff_th_min=np.pi / 16.0,
    pert_amp=0.1, pert_n_r=3.0, pert_n_th=2.0, pert_n_ph=4,
)


def _default_uph_amp(uph_r_power):
    # :378-382
    r_isco = 6.0
    omega_isco = r_isco ** -1.5
    gamma_isco = (1.0 - 2.0 / r_isco - r_isco ** 2 * omega_isco ** 2) ** -0.5
    return gamma_isco * omega_isco * r_isco ** uph_r_power


@dataclass
class Grid:
    """Grid in the reference reader's layout (see bl_grid_desc): n_b equal blocks."""
    prim: np.ndarray   # float32 [8][n_b][n_k][n_j][n_i]
    x1f: np.ndarray    # float64 [n_b][n_i+1]
    x2f: np.ndarray
    x3f: np.ndarray
    x1v: np.ndarray    # float64 [1][n_i]
    x2v: np.ndarray
    x3v: np.ndarray
    plasma_gamma: float = 0.0
    plasma_gamma_i: float = 0.0
    plasma_gamma_e: float = 0.0
    ind_kappa: int = -1   # index of the electron entropy in prim (plasma_model = code_kappa), -1: none
    # MeshBlock table (simulation_block_interp): int32 [n_b], int32 [n_b][3], cells of the root grid in x3
    levels: np.ndarray = None
    locations: np.ndarray = None
    n_3_root: int = 0

    @property
    def shape(self):
        return self.prim.shape[2:]

    def save_raw(self, path):
        """Raw grid file read by the command-line driver (bl_main.cpp). One block: magic "BLGRID1", int32 n_i,
        n_j, n_k, n_var; several equal blocks: magic "BLGRID2", int32 n_b, n_i, n_j, n_k, n_var. Then float64
        x1f x2f x3f x1v x2v x3v (each [n_b][...]) and float32 prim[n_var][n_b][n_k][n_j][n_i]."""
        n_var, n_b, n_k, n_j, n_i = self.prim.shape
        with open(path, "wb") as f:
            if n_b == 1:
                f.write(b"BLGRID1\0")
                np.array([n_i, n_j, n_k, n_var], dtype=np.int32).tofile(f)
            else:
                f.write(b"BLGRID2\0")
                np.array([n_b, n_i, n_j, n_k, n_var], dtype=np.int32).tofile(f)
            for name in ("x1f", "x2f", "x3f", "x1v", "x2v", "x3v"):
                np.ascontiguousarray(getattr(self, name), dtype=np.float64).tofile(f)
            np.ascontiguousarray(self.prim, dtype=np.float32).tofile(f)

    def desc(self):
        """bl_grid_desc borrowing this object's arrays (keep `self` alive while it is used)."""
        d = _capi.GridDesc()
        n_var, n_b, n_k, n_j, n_i = self.prim.shape
        d.n_blocks, d.n_i, d.n_j, d.n_k, d.n_var = n_b, n_i, n_j, n_k, n_var
        for name in ("prim", "x1f", "x2f", "x3f", "x1v", "x2v", "x3v"):
            arr = getattr(self, name)
            assert arr.flags["C_CONTIGUOUS"]
            setattr(d, name, arr.ctypes.data_as(C.c_void_p))
        # VerifyVariablesAthena (simulation_reader.cpp:1141-1216) for the order written above
        d.ind_rho, d.ind_pgas, d.ind_uu1, d.ind_uu2, d.ind_uu3 = 0, 1, 2, 3, 4
        d.ind_bb1, d.ind_bb2, d.ind_bb3 = 5, 6, 7
        d.ind_kappa = max(self.ind_kappa, 0)
        d.plasma_gamma, d.plasma_gamma_i, d.plasma_gamma_e = (
            self.plasma_gamma, self.plasma_gamma_i, self.plasma_gamma_e)
        if self.levels is not None:
            assert self.levels.dtype == np.int32 and self.locations.dtype == np.int32
            assert self.levels.flags["C_CONTIGUOUS"] and self.locations.flags["C_CONTIGUOUS"]
            d.levels = self.levels.ctypes.data_as(C.c_void_p)
            d.locations = self.locations.ctypes.data_as(C.c_void_p)
            d.n_3_root = int(self.n_3_root)
        return d


ENTROPY_SCALE = np.float32(2.0 ** 26)


def electron_entropy(rho, pgas):
    """A stand-in electron entropy variable for plasma_model = code_kappa (the reference's mock script writes
    none): 2^26 p / rho^(3/2) in single precision - only correctly rounded operations, so that any host
    reproduces it bit for bit. Gives Theta_e of order 1..10 on the mock torus."""
    rho = np.asarray(rho, dtype=np.float32)
    pgas = np.asarray(pgas, dtype=np.float32)
    return (ENTROPY_SCALE * (pgas / (rho * np.sqrt(rho)))).astype(np.float32)


def with_entropy(grid):
    """The grid with electron_entropy() appended as a ninth variable."""
    import dataclasses
    kappa = electron_entropy(grid.prim[0], grid.prim[1])
    prim = np.ascontiguousarray(np.concatenate([grid.prim[:8], kappa[None]], axis=0), dtype=np.float32)
    return dataclasses.replace(grid, prim=prim, ind_kappa=8)


def generate(**overrides):
    """Return the Grid the reference would hold after reading the script's .athdf output."""
    kw = dict(DEFAULTS)
    kw.update(overrides)
    if kw["uph_amp"] is None:
        kw["uph_amp"] = _default_uph_amp(kw["uph_r_power"])

    # Construct grid (:26-39)
    lr_min = np.log(kw["r_min"])
    lr_max = np.log(kw["r_max"])
    lrf = np.linspace(lr_min, lr_max, kw["n_r"] + 1)
    rf = np.exp(lrf)
    thf = np.linspace(0.0, np.pi, kw["n_th"] + 1)
    phf = np.linspace(0.0, 2.0 * np.pi, kw["n_ph"] + 1)
    r = 0.5 * (rf[:-1] + rf[1:])
    th = 0.5 * (thf[:-1] + thf[1:])
    ph = 0.5 * (phf[:-1] + phf[1:])

    # Cutoff (:42-46)
    cutoff_r = np.where((r < kw["cutoff_r_min"]) | (r > kw["cutoff_r_max"]), 0.0, 1.0)
    cutoff_th = np.where((th < kw["cutoff_th_min"]) | (th > np.pi - kw["cutoff_th_min"]), 0.0, 1.0)
    cutoff_ph = np.ones_like(ph)
    cutoff = cutoff_r[None, None, :] * cutoff_th[None, :, None] * cutoff_ph[:, None, None]

    # Perturbation (:49-55)
    pert_r = np.cos(2.0 * np.pi * kw["pert_n_r"] * np.log(r / kw["cutoff_r_min"])
                    / np.log(kw["cutoff_r_max"] / kw["cutoff_r_min"]))
    pert_th = -np.cos(2.0 * np.pi * kw["pert_n_th"] * (th - kw["cutoff_th_min"])
                      / (np.pi - 2.0 * kw["cutoff_th_min"]))
    pert_ph = np.cos(kw["pert_n_ph"] * ph)
    pert = 1.0 + kw["pert_amp"] * pert_r[None, None, :] * pert_th[None, :, None] * pert_ph[:, None, None]

    # Cell values (:58-77)
    rho = kw["rho_amp"] * r[None, None, :] ** -kw["rho_r_power"] \
        * np.exp(-np.abs(th[None, :, None] - np.pi / 2.0) / kw["rho_th_scale"]) * pert * cutoff
    rho = np.maximum(rho, kw["rho_floor"])
    pgas = kw["pgas_amp"] * r[None, None, :] ** -kw["pgas_r_power"] \
        * np.exp(-np.abs(th[None, :, None] - np.pi / 2.0) / kw["pgas_th_scale"]) * pert ** 2 * cutoff
    pgas = np.maximum(pgas, kw["pgas_floor"])
    uur = np.zeros_like(rho)
    uuth = np.zeros_like(rho)
    uuph = kw["uph_amp"] * r[None, None, :] ** -kw["uph_r_power"] \
        * np.exp(-np.abs(th[None, :, None] - np.pi / 2.0) / kw["uph_th_scale"]) * cutoff
    rr = np.maximum(r[None, None, :] * np.sin(th[None, :, None]), kw["cutoff_r_min"])
    bbz = kw["Bz_amp"] * rr ** -kw["Bz_R_power"]
    bbr = np.cos(th[None, :, None]) * bbz * np.ones_like(ph[:, None, None])
    bbth = -np.sin(th[None, :, None]) / r[None, None, :] * bbz * np.ones_like(ph[:, None, None])
    bbph = kw["Bph_amp"] * r[None, None, :] ** -kw["Bph_r_power"] \
        * np.exp(-np.abs(th[None, :, None] - np.pi / 2.0) / kw["Bph_th_scale"]) \
        * np.ones_like(ph[:, None, None])
    if not kw["Bph_no_flip"]:
        bbph = bbph * np.where(th > np.pi / 2.0, -1.0, 1.0)[None, :, None]

    # What the .athdf holds (:181-198) and what the reader hands on
    n_ph, n_th, n_r = len(ph), len(th), len(r)
    prim = np.empty((8, 1, n_ph, n_th, n_r), dtype=np.float32)
    for index, field in enumerate((rho, pgas, uur, uuth, uuph, bbr, bbth, bbph)):
        prim[index, 0] = np.broadcast_to(field, (n_ph, n_th, n_r)).astype(np.float32)

    def as_read(values):
        return np.ascontiguousarray(values.astype(np.float32).astype(np.float64)[None, :])

    return Grid(prim=prim, x1f=as_read(rf), x2f=as_read(thf), x3f=as_read(phf),
                x1v=as_read(r), x2v=as_read(th), x3v=as_read(ph))
