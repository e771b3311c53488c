"""Helpers shared by the parity tests: load a golden fixture (tests/golden/*.npz, produced by
tools/make_goldens.py from the compiled reference) and turn it into inputs for the oracle and the
HIP library."""
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = sorted(f[:-4] for f in os.listdir(GOLDEN_DIR)
               if f.endswith(".npz") and not f.startswith(("mock_", "window_", "slow_")))
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
    refined = mock_args.pop("_refined", None)
    last = mock_args.pop("_last", None)   # block to move to the end of the list: see move_last
    assert json.loads(str(fx["mock_args"])) == mock_args
    prim = np.ascontiguousarray(fx["prim"], dtype=np.float32)

    def coord(name):
        return np.ascontiguousarray(fx[name].astype(np.float32).astype(np.float64).reshape(1, -1))

    grid = Grid(prim=prim, x1f=coord("x1f"), x2f=coord("x2f"), x3f=coord("x3f"),
                x1v=coord("x1v"), x2v=coord("x2v"), x3v=coord("x3v"))
    if entropy:
        from blacklight_amd.mock import with_entropy
        grid = with_entropy(grid)
    if refined:
        return refined_grid(grid, last)
    return split_grid(grid, *blocks, last=last) if blocks is not None else single_block_table(grid)


def single_block_table(grid):
    import dataclasses
    return dataclasses.replace(grid, levels=np.zeros(1, dtype=np.int32), locations=np.zeros((1, 3), dtype=np.int32),
                               n_3_root=int(grid.prim.shape[2]))


def move_last(blocks, last, key):
    """Put the block for which key(block) == tuple(last) at the end of the list. With inter-block interpolation the
    reference reads one element past a block's cell-centre rows at its upper edges - the first element of the next
    block's row, or, for the last block of the file, whatever follows the array - so the goldens for it keep a block
    the rays never reach in last place (tools/make_goldens.py applies the same move)."""
    if last is None:
        return blocks
    at = [n for n, b in enumerate(blocks) if key(b) == tuple(last)]
    assert len(at) == 1
    return blocks[:at[0]] + blocks[at[0] + 1:] + [blocks[at[0]]]


def split_grid(grid, nbi, nbj, nbk, last=None):
    """The single-block grid as nbi x nbj x nbk equal blocks in the scrambled order that
    tools/make_goldens.py split_into_blocks wrote for the reference (same permutation)."""
    from blacklight_amd.mock import Grid
    n_var, _, n_k, n_j, n_i = grid.prim.shape
    ni, nj, nk = n_i // nbi, n_j // nbj, n_k // nbk
    blocks = [(bk, bj, bi) for bk in range(nbk) for bj in range(nbj) for bi in range(nbi)]
    order = np.random.default_rng(3).permutation(len(blocks))
    blocks = move_last([blocks[o] for o in order], last, lambda b: (b[2], b[1], b[0]))   # last = (bi, bj, bk)
    prim = np.empty((n_var, len(blocks), nk, nj, ni), dtype=np.float32)
    for n, (bk, bj, bi) in enumerate(blocks):
        prim[:, n] = grid.prim[:, 0, bk * nk:(bk + 1) * nk, bj * nj:(bj + 1) * nj, bi * ni:(bi + 1) * ni]

    def cut(arr, n, sel, extra):
        return np.ascontiguousarray(np.array([arr[0, b[sel] * n: b[sel] * n + n + extra] for b in blocks]))

    return Grid(prim=prim, x1f=cut(grid.x1f, ni, 2, 1), x2f=cut(grid.x2f, nj, 1, 1), x3f=cut(grid.x3f, nk, 0, 1),
                x1v=cut(grid.x1v, ni, 2, 0), x2v=cut(grid.x2v, nj, 1, 0), x3v=cut(grid.x3v, nk, 0, 0),
                ind_kappa=grid.ind_kappa, levels=np.zeros(len(blocks), dtype=np.int32),
                locations=np.ascontiguousarray([(bi, bj, bk) for bk, bj, bi in blocks], dtype=np.int32), n_3_root=int(n_k))


REFINED_BLOCK = (8, 6, 8)   # cells per block (i, j, k) of the refined version of the 32 x 24 x 32 mock


def refined_blocks(prim, xf, xv, last=None, block=None):
    """A two-level mesh from one block of data: prim [n_var][n_k][n_j][n_i] float32, xf / xv = three float32
    face / centre rows. The domain is cut into 2 x 2 x 2 octants; octants with an odd index sum become one
    coarse block each (level 0: pairwise averages of the fine cells in single precision, every second face),
    the others eight fine blocks each (level 1), all of REFINED_BLOCK cells, in a scrambled order. Only
    correctly rounded single-precision operations, so that tools/make_goldens.py (which writes the result as
    an .athdf for the reference) and the tests build the same bits. Returns a dict of arrays."""
    bi, bj, bk = block or REFINED_BLOCK   # (block: a quarter of another grid's cells per axis, bench.py --workload refined256)
    n_var, n_k, n_j, n_i = prim.shape
    assert (n_i, n_j, n_k) == (4 * bi, 4 * bj, 4 * bk)
    f32 = np.float32
    prim = prim.astype(f32)
    xf = [np.asarray(a, dtype=f32) for a in xf]
    xv = [np.asarray(a, dtype=f32) for a in xv]
    c = prim
    pairs_i = (c[..., 0::2] + c[..., 1::2]).astype(f32)
    pairs_j = (pairs_i[..., 0::2, :] + pairs_i[..., 1::2, :]).astype(f32)
    coarse = ((pairs_j[:, 0::2] + pairs_j[:, 1::2]).astype(f32) * f32(0.125)).astype(f32)
    cxf = [a[0::2] for a in xf]
    cxv = [(0.5 * (a[:-1].astype(np.float64) + a[1:].astype(np.float64))).astype(f32) for a in cxf]
    blocks = []   # (level, (li, lj, lk), source arrays, cell offsets)
    for ok in range(2):
        for oj in range(2):
            for oi in range(2):
                if (oi + oj + ok) % 2 == 1:
                    blocks.append((0, (oi, oj, ok), coarse, cxf, cxv, (oi * bi, oj * bj, ok * bk)))
                else:
                    for fk in range(2):
                        for fj in range(2):
                            for fi in range(2):
                                loc = (2 * oi + fi, 2 * oj + fj, 2 * ok + fk)
                                blocks.append((1, loc, prim, xf, xv, (loc[0] * bi, loc[1] * bj, loc[2] * bk)))
    order = np.random.default_rng(5).permutation(len(blocks))
    blocks = move_last([blocks[o] for o in order], last, lambda b: (b[0],) + tuple(b[1]))   # last = (level, li, lj, lk)
    n_b = len(blocks)
    out = dict(prim=np.empty((n_var, n_b, bk, bj, bi), dtype=f32), levels=np.empty(n_b, dtype=np.int32),
               locations=np.empty((n_b, 3), dtype=np.int64))
    for name, n in (("x1f", bi + 1), ("x2f", bj + 1), ("x3f", bk + 1), ("x1v", bi), ("x2v", bj), ("x3v", bk)):
        out[name] = np.empty((n_b, n), dtype=f32)
    for n, (level, loc, data, faces, centres, (i0, j0, k0)) in enumerate(blocks):
        out["prim"][:, n] = data[:, k0:k0 + bk, j0:j0 + bj, i0:i0 + bi]
        out["levels"][n] = level
        out["locations"][n] = loc
        for axis, (start, size) in enumerate(((i0, bi), (j0, bj), (k0, bk))):
            out[f"x{axis + 1}f"][n] = faces[axis][start:start + size + 1]
            out[f"x{axis + 1}v"][n] = centres[axis][start:start + size]
    return out


def refined_grid(grid, last=None, block=None):
    """The single-block Grid as the two-level mesh of refined_blocks()."""
    from blacklight_amd.mock import Grid
    blocks = refined_blocks(grid.prim[:, 0], [grid.x1f[0], grid.x2f[0], grid.x3f[0]], [grid.x1v[0], grid.x2v[0], grid.x3v[0]], last, block)

    def row(name):
        return np.ascontiguousarray(blocks[name].astype(np.float64))

    return Grid(prim=np.ascontiguousarray(blocks["prim"]), x1f=row("x1f"), x2f=row("x2f"), x3f=row("x3f"),
                x1v=row("x1v"), x2v=row("x2v"), x3v=row("x3v"), ind_kappa=grid.ind_kappa,
                levels=np.ascontiguousarray(blocks["levels"], dtype=np.int32),
                locations=np.ascontiguousarray(blocks["locations"], dtype=np.int32), n_3_root=2 * (block or REFINED_BLOCK)[2])


def subdivide_blocks(grid, split):
    """Every MeshBlock of a Grid cut into split^3 (or split_i x split_j x split_k) smaller ones of the same level (what a run with smaller MeshBlocks of the same mesh
    writes): cells and coordinates are the blocks' own values, locations multiply by split. A deep hierarchy's table sizes - thousands
    of blocks, dozens of distinct coordinate rows - from the two-level mesh of refined_blocks()."""
    import dataclasses
    n_var, n_b, n_k, n_j, n_i = grid.prim.shape
    split_i, split_j, split_k = (split, split, split) if np.isscalar(split) else split
    assert n_i % split_i == 0 and n_j % split_j == 0 and n_k % split_k == 0
    si, sj, sk = n_i // split_i, n_j // split_j, n_k // split_k
    n_new = n_b * split_i * split_j * split_k
    prim = np.empty((n_var, n_new, sk, sj, si), dtype=np.float32)
    rows = {name: np.empty((n_new, n), dtype=np.float64) for name, n in
            (("x1f", si + 1), ("x2f", sj + 1), ("x3f", sk + 1), ("x1v", si), ("x2v", sj), ("x3v", sk))}
    levels = np.empty(n_new, dtype=np.int32) if grid.levels is not None else None
    locations = np.empty((n_new, 3), dtype=np.int32) if grid.locations is not None else None
    n = 0
    for b in range(n_b):
        for c in range(split_k):
            for bb in range(split_j):
                for a in range(split_i):
                    prim[:, n] = grid.prim[:, b, c * sk:(c + 1) * sk, bb * sj:(bb + 1) * sj, a * si:(a + 1) * si]
                    rows["x1f"][n] = grid.x1f[b, a * si:(a + 1) * si + 1]
                    rows["x2f"][n] = grid.x2f[b, bb * sj:(bb + 1) * sj + 1]
                    rows["x3f"][n] = grid.x3f[b, c * sk:(c + 1) * sk + 1]
                    rows["x1v"][n] = grid.x1v[b, a * si:(a + 1) * si]
                    rows["x2v"][n] = grid.x2v[b, bb * sj:(bb + 1) * sj]
                    rows["x3v"][n] = grid.x3v[b, c * sk:(c + 1) * sk]
                    if levels is not None:
                        levels[n] = grid.levels[b]
                        locations[n] = grid.locations[b] * np.array([split_i, split_j, split_k]) + np.array([a, bb, c])
                    n += 1
    return dataclasses.replace(grid, prim=np.ascontiguousarray(prim), levels=levels, locations=locations, **rows)


SLOW_CASES = ["slow_interp", "slow_nearest"]


def slow_light_grids(fx):
    """The eleven snapshots of a slow-light fixture, regenerated (blacklight_amd.mock.generate restates the
    reference's generator bit for bit; checked here against the hashes of the files the reference read)."""
    import hashlib
    from blacklight_amd import mock
    grids = []
    for args, want in zip(json.loads(str(fx["mock_args"])), json.loads(str(fx["prim_sha256"]))):
        grid = mock.generate(**args)
        assert hashlib.sha256(np.ascontiguousarray(grid.prim[:8, 0]).tobytes()).hexdigest() == want
        grids.append(grid)
    return grids


def slow_light_windows(params, file_times):
    """Test-side restatement of the reader's sliding window (simulation_reader.cpp:211-303): for every
    snapshot, (camera time, file numbers held, latest first)."""
    start, end, chunk = int(params["simulation_start"]), int(params["simulation_end"]), int(params["slow_chunk_size"])
    latest_number, windows = None, []
    for snapshot in range(int(params["slow_num_images"])):
        t = float(params["slow_t_start"]) + float(params["slow_dt"]) * snapshot
        if latest_number is None:
            latest_time, latest_number = t - 2.0, start + chunk - 2
        else:
            latest_time = file_times[latest_number]
        while latest_time < t and latest_number < end:
            latest_number += 1
            latest_time = file_times[latest_number]
        assert latest_time >= t - 1.0
        windows.append((t, [latest_number - n for n in range(chunk)]))
    return windows


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
        if key not in fx.files:
            continue
        if name == "I_nu" and f"{tier}_npz_Q_nu" in fx.files:   # polarized: rows 4 l + (I, Q, U, V)
            stokes = [fx[f"{tier}_npz_{s}_nu"].reshape(-1, n_pix) for s in "IQUV"]
            rows.append(np.stack(stokes, axis=1).reshape(-1, n_pix))
        else:
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


def per_pixel_relative(got, want):
    """north_star's parity bar as it words it - "per-pixel L-infinity < 1e-6 vs reference": max_m |got_m - want_m| / |want_m| over the
    finite pixels with want_m > 0, the number of pixels above 1e-6, and whether the two frames are finite and positive in the same
    pixels (bench.py's relative_distance is the same function)."""
    got = np.asarray(got, dtype=np.float64).reshape(-1)
    want = np.asarray(want, dtype=np.float64).reshape(-1)
    use = np.isfinite(want) & np.isfinite(got) & (want > 0.0)
    with np.errstate(invalid="ignore", divide="ignore"):
        rel = np.abs(got[use] - want[use]) / want[use]
    same_support = bool(np.array_equal(np.isfinite(got) & (got > 0.0), np.isfinite(want) & (want > 0.0)))
    return (float(rel.max()) if rel.size else 0.0), int((rel > 1.0e-6).sum()), int(use.sum()), same_support
