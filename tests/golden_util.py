"""Helpers shared by the parity tests: load a golden fixture (tests/golden/*.npz, produced by
tools/make_goldens.py from the compiled reference) and turn it into inputs for the oracle and the
HIP library."""
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = sorted(f[:-4] for f in os.listdir(GOLDEN_DIR)
               if f.endswith(".npz") and not f.startswith("mock_") and not f.startswith("window_"))
# cases whose parameters are inside the scope of the HIP path today
GPU_CASES = list(CASES)

FRAME_KEYS = ("cam_x", "u_con", "u_cov", "norm_con", "norm_con_c", "hor_con_c", "vert_con_c")


def load_case(name):
    fx = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"), allow_pickle=False)
    params = json.loads(str(fx["params"]))
    mock_args = json.loads(str(fx["mock_args"])) if "mock_args" in fx.files else None
    return fx, params, mock_args


def golden_grid(mock_args):
    """Grid built from the arrays the reference's own script wrote (mock_small.npz), not from the
    build's generator, so that parity tests do not depend on blacklight_amd.mock."""
    from blacklight_amd.mock import Grid
    fx = np.load(os.path.join(GOLDEN_DIR, "mock_small.npz"), allow_pickle=False)
    mock_args = dict(mock_args)
    blocks = mock_args.pop("_blocks", None)
    entropy = mock_args.pop("_entropy", None)
    assert json.loads(str(fx["mock_args"])) == mock_args
    prim = np.ascontiguousarray(fx["prim"], dtype=np.float32)

    def coord(name):
        return np.ascontiguousarray(fx[name].astype(np.float32).astype(np.float64).reshape(1, -1))

    grid = Grid(prim=prim, x1f=coord("x1f"), x2f=coord("x2f"), x3f=coord("x3f"),
                x1v=coord("x1v"), x2v=coord("x2v"), x3v=coord("x3v"))
    if entropy:
        from blacklight_amd.mock import with_entropy
        grid = with_entropy(grid)
    return split_grid(grid, *blocks) if blocks is not None else grid


def split_grid(grid, nbi, nbj, nbk):
    """The single-block grid as nbi x nbj x nbk equal blocks in the scrambled order that
    tools/make_goldens.py split_into_blocks wrote for the reference (same permutation)."""
    from blacklight_amd.mock import Grid
    n_var, _, n_k, n_j, n_i = grid.prim.shape
    ni, nj, nk = n_i // nbi, n_j // nbj, n_k // nbk
    blocks = [(bk, bj, bi) for bk in range(nbk) for bj in range(nbj) for bi in range(nbi)]
    order = np.random.default_rng(3).permutation(len(blocks))
    blocks = [blocks[o] for o in order]
    prim = np.empty((n_var, len(blocks), nk, nj, ni), dtype=np.float32)
    for n, (bk, bj, bi) in enumerate(blocks):
        prim[:, n] = grid.prim[:, 0, bk * nk:(bk + 1) * nk, bj * nj:(bj + 1) * nj, bi * ni:(bi + 1) * ni]

    def cut(arr, n, sel, extra):
        return np.ascontiguousarray(np.array([arr[0, b[sel] * n: b[sel] * n + n + extra] for b in blocks]))

    return Grid(prim=prim, x1f=cut(grid.x1f, ni, 2, 1), x2f=cut(grid.x2f, nj, 1, 1), x3f=cut(grid.x3f, nk, 0, 1),
                x1v=cut(grid.x1v, ni, 2, 0), x2v=cut(grid.x2v, nj, 1, 0), x3v=cut(grid.x3v, nk, 0, 0),
                ind_kappa=grid.ind_kappa)


IMAGE_ROW_NAMES = ["I_nu", "time", "length", "lambda", "emission", "tau", "lambda_ave_rho", "lambda_ave_n_e",
                   "lambda_ave_p_gas", "lambda_ave_Theta_e", "lambda_ave_B", "lambda_ave_sigma", "lambda_ave_beta_inverse",
                   "emission_ave_rho", "emission_ave_n_e", "emission_ave_p_gas", "emission_ave_Theta_e", "emission_ave_B",
                   "emission_ave_sigma", "emission_ave_beta_inverse", "tau_int_rho", "tau_int_n_e", "tau_int_p_gas",
                   "tau_int_Theta_e", "tau_int_B", "tau_int_sigma", "tau_int_beta_inverse", "crossings"]


def expected_image(fx, tier, n_pix):
    """All image rows in the reference's row order (radiation_integrator.cpp:436-520): (n_q, n_pix)."""
    rows = []
    for name in IMAGE_ROW_NAMES:
        key = f"{tier}_npz_{name}"
        if key in fx.files:
            rows.append(fx[key].reshape(-1, n_pix))
    if not rows:
        return np.zeros((0, n_pix))
    return np.concatenate(rows, axis=0)


def expected_rendering(fx, tier, n_pix):
    """False-colour renderings (n_images, 3, n_pix), or None."""
    key = f"{tier}_npz_rendering"
    if key not in fx.files:
        return None
    arr = fx[key]
    return arr.reshape(arr.shape[0], 3, n_pix)


def same_bits(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b)) | ((a == 0) & (b == 0))
