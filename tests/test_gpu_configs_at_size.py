"""BASELINE.json's configurations 4 and 5 at the camera sizes it names, through properties that do not depend on size and,
where the reference can reach them (forced adaptive refinement of a small root camera evaluates windows of a fine pixel
lattice), against reference windows:

  4: 2048 x 2048 adaptive camera, full-Stokes polarized transfer (example_adaptive.input's physics over the 256^3 mock):
     the adaptive loop over eight emulated ranks (blacklight_amd.distributed, the product path of a multi-GPU run: tiles
     dealt centre-first, buffers, reassembly, refinement on rank 0) equals the one-call loop level by level, row by row;
     a window of the 2048^2 lattice equals the reference's (tests/golden/window_2048_polarized.npz).
  5: 4096 x 4096 x 64 frequencies (example_true_color.input's parameter set): windows of that lattice equal the matching
     single-frequency renders in both arithmetic tiers; a ten-frequency window of the 1024^2 lattice equals the
     reference's (tests/golden/window_1024_multifreq.npz).
"""
import json
import os

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu

ADAPTIVE = dict(adaptive_max_level=1, adaptive_block_size=8, adaptive_frequency_num=1, adaptive_val_cut=0.0, adaptive_val_frac=-1.0,
                adaptive_abs_grad_cut=0.0, adaptive_abs_grad_frac=-1.0, adaptive_rel_grad_cut=0.0, adaptive_rel_grad_frac=-1.0,
                adaptive_abs_lapl_cut=0.0, adaptive_abs_lapl_frac=-1.0, adaptive_rel_lapl_cut=1.0, adaptive_rel_lapl_frac=0.25,
                adaptive_num_regions=0)


@pytest.fixture(scope="module")
def big_grid():
    from blacklight_amd import mock
    return mock.generate(n_r=256, n_th=256, n_ph=256)


def test_config_4_at_size_over_eight_emulated_ranks(built_library, big_grid):
    """example_adaptive.input's physics (polarized + tau, one refined level of 8 x 8 blocks by the relative-Laplacian
    criterion) with a 2048^2 root camera: blacklight_amd.distributed.render_adaptive over eight emulated ranks == the
    one-call Context.render_adaptive, every level and every row bit for bit, in the tolerant tier the frame is timed in."""
    import blacklight_amd as bl
    from blacklight_amd import distributed as bd
    import bench
    params = dict(bench.WORKLOAD, camera_resolution=2048, image_polarization=True, image_tau=True, **ADAPTIVE)
    with bl.Context(bl.Params.from_dict(params)) as ctx:
        ctx.set_grid(big_grid)
        ctx.set_arithmetic("tolerant")
        ctx.set_scratch_limit(96 << 30)
        one = ctx.render_adaptive()
        one_warnings = ctx.warnings
        ctx.clear_warnings()
        levels, warnings = bd.render_adaptive(ctx, bd.EmulatedComm(8))
    # (a few rays of the 2048^2 camera run into ray_max_steps: the warning carries the level's totals either way)
    assert len(levels) == len(one) == 2 and warnings == one_warnings and "geodesics terminate unexpectedly" in warnings
    for n, (got, want) in enumerate(zip(levels, one)):
        assert got["image"].shape == want["image"].shape and want["image"].shape[0] == 5     # I, Q, U, V, tau
        assert gu.same_bits(got["image"], want["image"]).all(), n
        assert np.array_equal(got["sample_num"], want["sample_num"]) and np.array_equal(got["sample_flags"], want["sample_flags"])
        if n > 0:
            assert np.array_equal(got["block_locs"], want["block_locs"])
        else:
            assert np.array_equal(got["refinement_flags"], want["refinement_flags"])
    assert one[0]["image"].shape[1] == 2048 * 2048 and one[1]["image"].shape[1] >= 64 and one[1]["image"].shape[1] % 64 == 0   # refined 8 x 8 blocks
    assert np.isfinite(one[0]["image"][:, one[0]["sample_flags"] == 0]).all()


def test_config_5_windows_of_the_4096_lattice(built_library, big_grid):
    """Windows of the 4096^2 x 64-frequency lattice (photon ring, disc, image corner): each row equals the single-frequency
    render of the same pixels - bit for bit in the exact tier, within the tolerant tier's rounding-level bound otherwise -
    and the two tiers agree on every integer result."""
    import blacklight_amd as bl
    import bench
    res, n_freq = 4096, 64
    params = dict(bench.WORKLOAD, camera_resolution=res, image_num_frequencies=n_freq, image_frequency_start=1.5e11,
                  image_frequency_end=3.3e11, image_frequency_spacing="lin_wave")
    params.pop("image_frequency", None)
    windows = []
    for v0, u0 in ((1930, 1600), (2300, 900), (8, 16)):
        iv, iu = np.mgrid[v0:v0 + 48, u0:u0 + 48]
        windows.append((iv * res + iu).reshape(-1).astype(np.int32))
    window = np.concatenate(windows)
    out = {}
    with bl.Context(bl.Params.from_dict(params)) as ctx:
        ctx.set_grid(big_grid)
        freqs = ctx.frequencies
        for tier in ("exact", "tolerant"):
            ctx.set_arithmetic(tier)
            out[tier] = ctx.render(pixel_map=window)
    assert out["exact"]["image"].shape == (n_freq, window.size)
    assert np.array_equal(out["exact"]["sample_num"], out["tolerant"]["sample_num"])
    assert np.array_equal(np.isnan(out["exact"]["image"]), np.isnan(out["tolerant"]["image"]))
    scale = np.nanmax(np.abs(out["exact"]["image"]), axis=1, keepdims=True)
    assert float(np.nanmax(np.abs(out["tolerant"]["image"] - out["exact"]["image"]) / scale)) < 1.0e-11
    assert out["exact"]["sample_num"].max() > 1000 and np.nanmax(out["exact"]["image"][:, -48 * 48:]) < np.nanmax(out["exact"]["image"])
    for l in (0, 17, 63):
        single = dict(bench.WORKLOAD, camera_resolution=res, image_frequency=float(freqs[l]))
        with bl.Context(bl.Params.from_dict(single)) as ctx:
            ctx.set_grid(big_grid)
            one = ctx.render(pixel_map=window)
            ctx.set_arithmetic("tolerant")
            one_tolerant = ctx.render(pixel_map=window)
        assert gu.same_bits(one["image"][0], out["exact"]["image"][l]).all(), l
        assert np.array_equal(one["sample_num"], out["exact"]["sample_num"])
        peak = np.nanmax(np.abs(one["image"][0]))
        assert float(np.nanmax(np.abs(one_tolerant["image"][0] - out["tolerant"]["image"][l])) / peak) < 1.0e-11, l


def _window_pixels(block_locs, lattice, bs=16):
    iv, iu = np.mgrid[0:bs, 0:bs]
    return np.concatenate([((bv * bs + iv) * lattice + (bu * bs + iu)).reshape(-1) for bv, bu in block_locs]).astype(np.int32)


POLARIZED_WINDOW = os.path.join(gu.GOLDEN_DIR, "window_2048_polarized.npz")
MULTIFREQ_WINDOW = os.path.join(gu.GOLDEN_DIR, "window_1024_multifreq.npz")


@pytest.mark.skipif(not os.path.exists(MULTIFREQ_WINDOW), reason="window_1024_multifreq fixture not generated")
def test_reference_window_of_the_ten_frequency_frame(built_library, big_grid):
    """The reference's pixels of the 1024^2 x 10-frequency lattice (example_true_color.input's frequencies) inside two forced
    refinement regions, over the 256^3 mock: bit-exact in the exact tier, <= 1e-9 of each row's peak in the tolerant one."""
    import blacklight_amd as bl
    fx = np.load(MULTIFREQ_WINDOW, allow_pickle=False)
    params = json.loads(str(fx["params"]))
    lattice = int(fx["lattice"])
    pixels = _window_pixels(fx["B_block_locs"], lattice)
    want = fx["B_I_nu"].reshape(fx["B_I_nu"].shape[0], -1)
    with bl.Context(bl.Params.from_dict(params)) as ctx:
        ctx.set_grid(big_grid)
        exact = ctx.render(pixel_map=pixels)
        ctx.set_arithmetic("tolerant")
        tolerant = ctx.render(pixel_map=pixels)
    assert exact["image"].shape == want.shape == (10, pixels.size)
    same = gu.same_bits(exact["image"], want)
    assert same.all(), f"{(~same).sum()} of {same.size} values differ from the reference"
    peak = np.nanmax(np.abs(want), axis=1, keepdims=True)
    assert float(np.nanmax(np.abs(tolerant["image"] - want) / peak)) < 1.0e-9


@pytest.mark.skipif(not os.path.exists(POLARIZED_WINDOW), reason="window_2048_polarized fixture not generated")
def test_reference_window_of_the_polarized_2048_frame(built_library, big_grid):
    """The reference's full-Stokes pixels of the 2048^2 lattice inside a forced refinement region at the photon ring:
    I, Q, U, V bit-exact in the exact tier, <= 1e-9 of each row's peak in the tolerant one (transport matrices)."""
    import blacklight_amd as bl
    fx = np.load(POLARIZED_WINDOW, allow_pickle=False)
    params = json.loads(str(fx["params"]))
    lattice = int(fx["lattice"])
    pixels = _window_pixels(fx["B_block_locs"], lattice)
    with bl.Context(bl.Params.from_dict(params)) as ctx:
        ctx.set_grid(big_grid)
        exact = ctx.render(pixel_map=pixels)
        ctx.set_arithmetic("tolerant")
        tolerant = ctx.render(pixel_map=pixels)
    for row, key in enumerate(("I_nu", "Q_nu", "U_nu", "V_nu")):
        want = fx["B_" + key].reshape(-1)
        same = gu.same_bits(exact["image"][row], want)
        assert same.all(), f"{key}: {(~same).sum()} of {same.size} window pixels differ from the reference"
        assert float(np.nanmax(np.abs(tolerant["image"][row] - want)) / np.nanmax(np.abs(want))) < 1.0e-9, key


def test_tolerant_tier_per_pixel_at_the_benchmark_size(built_library, big_grid):
    """north_star: "per-pixel L-infinity < 1e-6 vs reference". The tier bench.py quotes against the exact tier - the reference's bits,
    tests/test_gpu_window_1024.py - on all 1 048 576 pixels of the benchmark frame, each relative to its own intensity."""
    import bench
    import blacklight_amd as bl
    with bl.Context(bl.Params.from_dict(dict(bench.WORKLOAD))) as ctx:
        ctx.set_grid(big_grid)
        ctx.set_arithmetic("exact")
        exact = ctx.render()
        ctx.set_arithmetic("tolerant")
        tolerant = ctx.render()
        assert exact["stats"].arithmetic == 0 and tolerant["stats"].arithmetic == 1 and tolerant["stats"].composed_maps == 1
    assert np.array_equal(exact["sample_num"], tolerant["sample_num"]) and np.array_equal(exact["sample_flags"], tolerant["sample_flags"])
    worst, above, compared, same_support = gu.per_pixel_relative(tolerant["image"][0], exact["image"][0])
    assert same_support and compared > 1_000_000
    assert above == 0 and worst < 1.0e-6, (worst, above)
    assert worst < 1.0e-10, worst
