"""Windows of the 1024^2 benchmark frame computed by the unmodified reference (tools/make_goldens.py
window_1024: a 256^2 base camera with forced adaptive refinement to level 2 evaluates exactly the pixels of
the 1024^2 lattice inside the refined blocks) against the HIP path rendering those pixels of
a plain 1024^2 camera over the same 256^3 grid - the workload bench.py times."""
import json
import os

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu

FIXTURE = os.path.join(gu.GOLDEN_DIR, "window_1024.npz")


@pytest.mark.skipif(not os.path.exists(FIXTURE), reason="window_1024 fixture not generated")
def test_reference_windows_of_the_benchmark_frame(built_library):
    import blacklight_amd as bl
    from blacklight_amd import mock
    fx = np.load(FIXTURE, allow_pickle=False)
    params = json.loads(str(fx["params"]))
    mock_args = json.loads(str(fx["mock_args"]))
    # the build's generator reproduces the reference script's arrays bit for bit (tests/test_mock.py)
    grid = mock.generate(**mock_args)
    p = bl.Params.from_dict(params)
    res, bs = int(p.get("camera_resolution")), 16
    assert res == 1024
    locs = fx["B_block_locs"]
    assert np.array_equal(locs, fx["A_block_locs"])
    iv, iu = np.mgrid[0:bs, 0:bs]
    pixels = np.concatenate([((bv * bs + iv) * res + (bu * bs + iu)).reshape(-1) for bv, bu in locs]).astype(np.int32)
    with bl.Context(p) as ctx:
        ctx.set_grid(grid)
        out = ctx.render(pixel_map=pixels)
    got = out["image"][0]
    want_b = fx["B_I_nu"].reshape(-1)
    assert got.shape == want_b.shape
    same = gu.same_bits(got, want_b)
    assert same.all(), f"{(~same).sum()} of {same.size} window pixels differ from the reference (pinned math)"
    # stock glibc reference: a = 0, stated tolerance
    want_a = fx["A_I_nu"].reshape(-1)
    assert np.nanmax(np.abs(got - want_a)) / np.nanmax(np.abs(want_a)) < 1.0e-6
    # north_star's bar the way it is worded - per pixel, relative to that pixel's own intensity (the window spans decades of
    # intensity): the tier bench.py quotes against the reference (pinned math), every window pixel
    with bl.Context(p) as ctx:
        ctx.set_grid(grid)
        ctx.set_arithmetic("tolerant")
        tolerant = ctx.render(pixel_map=pixels)
        assert tolerant["stats"].arithmetic == 1 and tolerant["stats"].fused_variant == 2
    worst, above, compared, same_support = gu.per_pixel_relative(tolerant["image"][0], want_b)
    assert same_support and compared > 0.9 * want_b.size and np.array_equal(tolerant["sample_num"], out["sample_num"])
    assert above == 0 and worst < 1.0e-6, (worst, above, compared)
    assert worst < 1.0e-10, worst   # (measured 2e-14: what the tier actually delivers here)
    dim = want_b[np.isfinite(want_b) & (want_b > 0.0)]
    assert dim.max() / dim.min() > 3.0e2   # (the windows do hold dim pixels - 1 / 600 of the brightest: an L-infinity over the image maximum would not see them)


FORMULA_FIXTURE = os.path.join(gu.GOLDEN_DIR, "window_512_formula.npz")


@pytest.mark.skipif(not os.path.exists(FORMULA_FIXTURE), reason="window_512_formula fixture not generated")
def test_reference_windows_of_configuration_2_at_its_own_lattice(built_library):
    """BASELINE's configuration 2 (example_formula.input, a = 0.9, 512^2) against the reference at that lattice: two 32 x 32 windows
    - the edge of the shadow, where rays orbit for thousands of samples, and the periphery - which the unmodified reference computed
    through forced refinement of a 128^2 root camera (tools/make_goldens.py window_512_formula; radiation_adaptive.cpp:50-69). The HIP
    path renders those pixels of the plain 512^2 camera: bit for bit in the exact tier (pinned math library), 1e-6 from the stock
    reference, and the tolerant tier within 1e-9 of it with the same counts and NaN mask."""
    import blacklight_amd as bl
    fx = np.load(FORMULA_FIXTURE, allow_pickle=False)
    p = bl.Params.from_dict(json.loads(str(fx["params"])))
    res, bs = int(p.get("camera_resolution")), 16
    assert res == 512 == int(fx["lattice"])
    locs = fx["B_block_locs"]
    assert np.array_equal(locs, fx["A_block_locs"]) and locs.shape[0] == 8
    iv, iu = np.mgrid[0:bs, 0:bs]
    pixels = np.concatenate([((bv * bs + iv) * res + (bu * bs + iu)).reshape(-1) for bv, bu in locs]).astype(np.int32)
    with bl.Context(p) as ctx:
        exact = ctx.render(pixel_map=pixels)
        # one ray of the window runs into ray_max_steps (NaN with fallback_nan): the reference's warning, with its count
        assert str(fx["B_warnings"]) == "Warning: 1 out of 2048 geodesics terminate unexpectedly.\n" and str(fx["B_warnings"]) in ctx.warnings
        ctx.set_arithmetic("tolerant")
        tolerant = ctx.render(pixel_map=pixels)
        assert tolerant["stats"].arithmetic == 1
    assert exact["sample_flags"].sum() == 1 and np.isnan(exact["image"][0]).sum() == 1
    want_b, want_a = fx["B_I_nu"].reshape(-1), fx["A_I_nu"].reshape(-1)
    got = exact["image"][0]
    same = gu.same_bits(got, want_b)
    assert same.all(), f"{(~same).sum()} of {same.size} window pixels differ from the reference (pinned math)"
    assert exact["sample_num"].max() > 5000   # (the window does hold the long orbits)
    peak = np.nanmax(np.abs(want_b))
    assert np.nanmax(np.abs(got - want_a)) / np.nanmax(np.abs(want_a)) < 1.0e-5   # (a = 0.9: the envelope SURVEY.md 7 measured between math libraries)
    assert np.array_equal(np.isnan(tolerant["image"][0]), np.isnan(want_b))
    assert np.nanmax(np.abs(tolerant["image"][0] - want_b)) / peak < 1.0e-9
    assert np.array_equal(tolerant["sample_num"], exact["sample_num"]) and np.array_equal(tolerant["sample_flags"], exact["sample_flags"])


def test_full_size_frame_is_independent_of_how_it_is_split(built_library):
    """Size-independent properties at the benchmark's own size (1024^2 camera, 256^3 grid): every ray is independent,
    so the frame rendered in one call, the frame assembled from the 32 x 32-pixel tiles of eight emulated ranks
    (blacklight_amd.distributed, the strong-scaling mode of bench.py), and the frame rendered in many small chunks
    must agree bit for bit - image, sample counts and flags. Also: the gather counter of the whole frame equals
    the sum over the ranks, and no pixel of the plain frame is NaN except along flagged rays."""
    import blacklight_amd as bl
    from blacklight_amd import distributed as bd, mock
    import bench
    grid = mock.generate(n_r=256, n_th=256, n_ph=256)
    p = bl.Params.from_dict(dict(bench.WORKLOAD))
    res, world = 1024, 8
    with bl.Context(p) as ctx:
        ctx.set_grid(grid)
        full = ctx.render()
        assert full["image"].shape == (1, res * res)
        assembled = np.full(res * res, np.nan)
        counts = np.full(res * res, -1, dtype=np.int32)
        gathers = 0
        seen = 0
        for rank in range(world):
            pixels = bd.tile_pixels(res, rank, world, bench.TILE)
            part = ctx.render(pixel_map=pixels)
            assembled[pixels] = part["image"][0]
            counts[pixels] = part["sample_num"]
            gathers += part["stats"].n_gathers
            seen += pixels.size
        assert seen == res * res and (counts >= 0).all()
        assert gu.same_bits(assembled, full["image"][0]).all()
        assert np.array_equal(counts, full["sample_num"])
        assert gathers == full["stats"].n_gathers
        ctx.set_scratch_limit(8 << 30)   # ~ 40 chunks instead of 2
        chunked = ctx.render()
        assert chunked["stats"].n_chunks > 10
        assert gu.same_bits(chunked["image"][0], full["image"][0]).all()
        assert np.array_equal(chunked["sample_num"], full["sample_num"])
        assert np.array_equal(chunked["sample_flags"], full["sample_flags"])
    nan = np.isnan(full["image"][0])
    assert np.array_equal(nan, full["sample_flags"].astype(bool) & nan) and nan.sum() <= full["sample_flags"].sum()
    # algorithmic bytes of the roofline: 256 B per gathered sample + 13 B per ray
    assert full["stats"].algorithmic_bytes == 256 * full["stats"].n_gathers + 13 * res * res


def test_saturated_rays_fill_the_record_buffer_exactly(built_library):
    """Every ray of a 1024^2 camera runs into a low ray_max_steps: the sample records then need all of
    chunk_rays * ray_max_steps slots. The geodesic kernel hands slots out without losing any at a block switch, so
    that bound (plus one block per wave) holds - the reference only warns in this situation, and so does bl_render."""
    import blacklight_amd as bl
    fx, params, _ = gu.load_case("formula_dp")
    for steps in (96, 257):
        p = bl.Params.from_dict(dict(params, camera_resolution=1024, ray_max_steps=steps, fallback_nan="false"))
        with bl.Context(p) as ctx:
            out = ctx.render()
            st = out["stats"]
            assert st.n_flagged == 1024 * 1024 and (out["sample_num"] == steps).all() and out["sample_flags"].all()
            assert st.n_samples == 1024 * 1024 * steps
            assert "1048576 out of 1048576 geodesics terminate unexpectedly." in ctx.warnings
            assert np.isfinite(out["image"]).all()


def _split_property(ctx, res, world, tile, rows=None):
    """Frame in one call == frame assembled from the tiles of `world` emulated ranks == frame in many chunks."""
    from blacklight_amd import distributed as bd
    full = ctx.render()
    n_q = full["image"].shape[0]
    assembled = np.full((n_q, res * res), np.nan)
    counts = np.full(res * res, -1, dtype=np.int32)
    for rank in range(world):
        pixels = bd.tile_pixels(res, rank, world, tile)
        part = ctx.render(pixel_map=pixels)
        assembled[:, pixels] = part["image"]
        counts[pixels] = part["sample_num"]
    assert np.array_equal(counts, full["sample_num"])
    assert gu.same_bits(assembled, full["image"]).all()
    return full


def test_config_2_at_size_is_independent_of_how_it_is_split(built_library):
    """BASELINE.json configuration 2 - example_formula.input at 512^2 (formula mode, a = 0.9, camera at r = 1000,
    ray_max_steps = 7000) - at its own size: split independence, flags only where rays ran out of steps."""
    import blacklight_amd as bl
    fx, params, _ = gu.load_case("formula_dp")
    p = bl.Params.from_dict(dict(params, camera_resolution=512))
    with bl.Context(p) as ctx:
        full = _split_property(ctx, 512, 4, 32)
        ctx.set_scratch_limit(4 << 30)   # 5e7 sample records at a time of the frame's 3.7e8
        chunked = ctx.render()
        assert chunked["stats"].n_chunks > 4
        assert gu.same_bits(chunked["image"], full["image"]).all() and np.array_equal(chunked["sample_num"], full["sample_num"])
    flagged = full["sample_flags"].astype(bool)
    assert np.array_equal(np.isnan(full["image"][0]), flagged) and 0 < flagged.sum() < 4096
    assert (full["sample_num"][~flagged] < 7000).all() and full["sample_num"].mean() > 1000


def test_polarized_frame_at_size_is_independent_of_how_it_is_split(built_library):
    """Configuration 4's physics - full-Stokes polarized transfer with image_tau - on one 1024^2 frame over the 256^3 mock:
    the frame in one call equals the frame assembled from the tiles of three emulated ranks (unequal shares), row by row
    (I, Q, U, V, tau), and |Q|, |U|, |V| never exceed I."""
    import blacklight_amd as bl
    from blacklight_amd import mock
    import bench
    grid = mock.generate(n_r=256, n_th=256, n_ph=256)
    p = bl.Params.from_dict(dict(bench.WORKLOAD, image_polarization=True, image_rotation_split=False, image_tau=True))
    with bl.Context(p) as ctx:
        ctx.set_grid(grid)
        full = _split_property(ctx, 1024, 3, 32)
        # the tolerant tier on the same frame - transport matrices instead of the tensor transport (docs/notebook.md section 5d), rays of
        # every length across the 64-sample segments of bl_transport_matrix_kernel: north_star's tolerance row by row
        ctx.set_arithmetic("tolerant")
        tolerant = ctx.render()
        assert tolerant["stats"].arithmetic == 1
    image = full["image"]
    assert np.array_equal(tolerant["sample_num"], full["sample_num"]) and np.array_equal(tolerant["sample_flags"], full["sample_flags"])
    assert np.array_equal(np.isnan(tolerant["image"]), np.isnan(image))
    for row, name in enumerate(["I", "Q", "U", "V", "tau"]):
        scale = np.nanmax(np.abs(image[row]))
        distance = np.nanmax(np.abs(tolerant["image"][row] - image[row])) / scale
        print(f"polarized 1024^2, tolerant vs exact, row {name}: {distance:.2e} of the row's peak")
        assert distance < 1.0e-6, name
    assert image.shape == (5, 1024 * 1024)
    ok = np.isfinite(image[0])
    assert ok.mean() > 0.999
    polarized = np.sqrt(image[1] ** 2 + image[2] ** 2 + image[3] ** 2)
    assert (polarized[ok] <= image[0][ok] * (1.0 + 1.0e-12) + 1.0e-300).all() and polarized[ok].max() > 0.0
