"""GPU: geodesics once per series (bl_set_geodesic_reuse).

The reference integrates the root camera's geodesics once, before its loop over snapshots (blacklight.cpp:93-94 against :178-250), and
locates the samples once while the mesh does not change (`first_time`, radiation_integrator.cpp:693-704). Here: a root-level render of
an unchanged camera after a new bl_set_grid shades the sample records the previous render left in HBM. What must hold: the images are
the bits a fresh render gives (exact tier, and the tolerant tier under bl_set_reproducible; to rounding with composed maps), the stepper
does not run (launches_geodesic == 0), renders of refined levels in between leave the records alone, and any change of the camera, the
record layout or the switch itself makes the next render integrate its rays again."""
import dataclasses

import numpy as np
import pytest
import torch   # (before the library: see tests/test_gpu_defaults.py)

import golden_util as gu

pytestmark = pytest.mark.gpu


def _snapshots(grid, count):
    """`count` snapshots on one geometry: density and pressure scaled, the rest kept"""
    out = []
    for n in range(count):
        prim = grid.prim.copy()
        prim[0:2] *= np.float32(1.0 + 0.11 * n)
        out.append(dataclasses.replace(grid, prim=prim))
    return out


def _fresh(params, grid, tier, reproducible=False, undefined=None, **render_args):
    import blacklight_amd as bl
    with bl.Context(bl.Params.from_dict(params)) as ctx:
        ctx.set_geodesic_reuse(False)
        if undefined:
            ctx.set_undefined_policy(undefined)
        ctx.set_grid(grid)
        ctx.set_arithmetic(tier)
        ctx.set_reproducible(reproducible)
        out = ctx.render(**render_args)
        assert out["stats"].geodesics_reused == 0 and out["stats"].launches_geodesic >= 1
        return out


def _same_frame(got, want, bitwise=True, tolerance=1.0e-13):
    assert np.array_equal(got["sample_num"], want["sample_num"]) and np.array_equal(got["sample_flags"], want["sample_flags"])
    if bitwise:
        assert gu.same_bits(got["image"], want["image"]).all()
    else:
        assert np.array_equal(np.isnan(got["image"]), np.isnan(want["image"]))
        with np.errstate(invalid="ignore"):
            assert np.nanmax(np.abs(got["image"] - want["image"])) <= tolerance * np.nanmax(np.abs(want["image"]))
    assert got["stats"].n_samples == want["stats"].n_samples and got["stats"].n_gathers == want["stats"].n_gathers


@pytest.mark.parametrize("tier", ["exact", "tolerant"])
def test_two_snapshots_second_frame_reuses_the_records(tier):
    """The benchmark's path (locate step inside the coefficient kernel) at small size: frame 2 of a series"""
    import bench
    import blacklight_amd as bl
    from blacklight_amd import mock
    params = dict(bench.WORKLOAD, camera_resolution=96)
    snaps = _snapshots(mock.generate(n_r=48, n_th=48, n_ph=48), 3)
    with bl.Context(bl.Params.from_dict(params)) as ctx:
        ctx.set_arithmetic(tier)
        ctx.set_reproducible(True)   # tolerant tier: one record per sample, so that frames are the same bits however they were made
        frames = []
        for n, grid in enumerate(snaps):
            ctx.clear_warnings()
            ctx.set_grid(grid)
            out = ctx.render(want_camera=True)
            st = out["stats"]
            assert st.fused_variant == (3 if tier == "exact" else 2)
            if n == 0:
                assert st.geodesics_reused == 0 and st.launches_geodesic == 1 and st.ms_geodesic > 0.0
                assert "geodesics terminate unexpectedly" in ctx.warnings or st.n_flagged == 0
            else:
                assert st.geodesics_reused == 1 and st.launches_geodesic == 0 and st.ms_geodesic == 0.0 and st.sampling_reused == 0
                assert "geodesics terminate unexpectedly" not in ctx.warnings   # raised once, by the render that integrated them
            assert st.n_flagged == frames[0]["stats"].n_flagged if frames else True
            frames.append(out)
        for n, grid in enumerate(snaps):
            want = _fresh(params, grid, tier, reproducible=True, want_camera=True)
            _same_frame(frames[n], want)
            assert gu.same_bits(frames[n]["camera_pos"], want["camera_pos"]).all() and gu.same_bits(frames[n]["camera_dir"], want["camera_dir"]).all()
        assert not gu.same_bits(frames[0]["image"], frames[1]["image"]).all()   # (the snapshots do differ)
        # composed maps (the tolerant tier's default): the same frame to rounding
        if tier == "tolerant":
            ctx.set_reproducible(False)
            ctx.set_grid(snaps[1])
            first = ctx.render()      # the record layout changed (segments numbered): integrated again
            assert first["stats"].geodesics_reused == 0 and first["stats"].composed_maps == 1
            ctx.set_grid(snaps[2])
            again = ctx.render()
            assert again["stats"].geodesics_reused == 1 and again["stats"].composed_maps == 1
            _same_frame(again, frames[2], bitwise=False)


def test_what_invalidates_the_records():
    import bench
    import blacklight_amd as bl
    from blacklight_amd import distributed as bd
    from blacklight_amd import mock
    params = dict(bench.WORKLOAD, camera_resolution=64)
    grid = mock.generate(n_r=32, n_th=32, n_ph=32)
    with bl.Context(bl.Params.from_dict(params)) as ctx:
        ctx.set_arithmetic("exact")
        ctx.set_grid(grid)
        whole = ctx.render()
        assert ctx.render()["stats"].geodesics_reused == 1
        # another set of pixels: integrated; the same set again: reused; the whole frame again: integrated
        pixels = bd.tile_pixels(64, 1, 2, 32)
        share = ctx.render(pixel_map=pixels)
        assert share["stats"].geodesics_reused == 0
        again = ctx.render(pixel_map=pixels)
        assert again["stats"].geodesics_reused == 1
        assert gu.same_bits(again["image"], share["image"]).all() and gu.same_bits(share["image"], whole["image"][:, pixels]).all()
        other = ctx.render(pixel_map=bd.tile_pixels(64, 0, 2, 32))
        assert other["stats"].geodesics_reused == 0
        back = ctx.render()
        assert back["stats"].geodesics_reused == 0 and gu.same_bits(back["image"], whole["image"]).all()
        # the tail policy, a measurement switch, the tier's record layout
        ctx.set_tail_policy("quad")
        assert ctx.render()["stats"].geodesics_reused == 0
        assert ctx.render()["stats"].geodesics_reused == 1
        ctx.set_tail_policy("auto")
        ctx.debug_set_switches("RECORD_EVERY_STEP")
        assert ctx.render()["stats"].geodesics_reused == 0
        ctx.debug_set_switches()
        assert ctx.render()["stats"].geodesics_reused == 0
        # switched off: every render integrates; switched on again: the first one does
        ctx.set_geodesic_reuse(False)
        for _ in range(2):
            out = ctx.render()
            assert out["stats"].geodesics_reused == 0 and out["stats"].launches_geodesic == 1
        ctx.set_geodesic_reuse(True)
        assert ctx.render()["stats"].geodesics_reused == 0
        last = ctx.render()
        assert last["stats"].geodesics_reused == 1 and gu.same_bits(last["image"], whole["image"]).all()


@pytest.mark.parametrize("layout", ["refined", "blocks", "block_interp"])
def test_located_samples_are_kept_while_the_geometry_is(layout):
    """Grids that go through a locate kernel of its own (mesh refinement, blocks with holes, inter-block interpolation): frame 2 keeps the
    located samples too (the reference's first_time); a snapshot on another geometry has them located again over the same records"""
    import blacklight_amd as bl
    fx, params, mock_args = gu.load_case("sim_dp_interp")
    base = gu.golden_grid(mock_args)
    params = dict(params, camera_resolution=48)
    undefined = None
    if layout == "refined":
        grid = gu.refined_grid(base)
    elif layout == "blocks":
        grid = gu.split_grid(base, 2, 2, 2)
    else:
        grid = gu.split_grid(base, 2, 2, 2)
        params["simulation_block_interp"] = "true"
        undefined = "edge"
    snaps = _snapshots(grid, 2)
    # a third snapshot on another geometry: the radial faces stretched by one part in a thousand (coordinates are the file's floats)
    def stretched(values):
        return (values * 1.001).astype(np.float32).astype(np.float64)
    moved = dataclasses.replace(snaps[1], x1f=stretched(grid.x1f), x1v=stretched(grid.x1v))
    for tier in ("exact", "tolerant"):
        with bl.Context(bl.Params.from_dict(params)) as ctx:
            if undefined:
                ctx.set_undefined_policy(undefined)
            ctx.set_arithmetic(tier)
            ctx.set_reproducible(True)
            got = []
            for grid_n in (snaps[0], snaps[1], moved, moved):
                ctx.set_grid(grid_n)
                got.append(ctx.render())
            flags = [(g["stats"].geodesics_reused, g["stats"].sampling_reused, g["stats"].launches_locate) for g in got]
            located_inside = got[0]["stats"].fused_variant != 0
            if located_inside:   # (equal blocks merged into one array: the locate step runs inside the coefficient kernel, nothing to keep)
                assert flags == [(0, 0, 0), (1, 0, 0), (1, 0, 0), (1, 0, 0)], flags
            else:
                assert flags == [(0, 0, 1), (1, 1, 0), (1, 0, 1), (1, 1, 0)], flags
            for g, grid_n in zip(got, (snaps[0], snaps[1], moved, moved)):
                _same_frame(g, _fresh(params, grid_n, tier, reproducible=True, undefined=undefined))


def test_refined_mesh_series_with_the_locate_step_inside():
    """The tolerant tier's default over a mesh with refinement (bl_shade_fused2_kernel<..., kRefined>: nothing located outside the
    coefficient kernel, composed maps): frames 2 and 3 of a series run that kernel over the resident records"""
    import blacklight_amd as bl
    fx, params, mock_args = gu.load_case("sim_dp_interp")
    params = dict(params, camera_resolution=48)
    snaps = _snapshots(gu.refined_grid(gu.golden_grid(mock_args)), 3)
    with bl.Context(bl.Params.from_dict(params)) as ctx:
        ctx.set_arithmetic("tolerant")
        frames = []
        for grid in snaps:
            ctx.set_grid(grid)
            frames.append(ctx.render())
    flags = [(f["stats"].geodesics_reused, f["stats"].sampling_reused, f["stats"].launches_locate, f["stats"].fused_variant, f["stats"].composed_maps) for f in frames]
    assert flags == [(0, 0, 0, 2, 1), (1, 0, 0, 2, 1), (1, 0, 0, 2, 1)], flags
    for frame, grid in zip(frames, snaps):
        _same_frame(frame, _fresh(params, grid, "tolerant", reproducible=True), bitwise=False)   # (through the locate kernel, a record per sample)
        _same_frame(frame, _fresh(params, grid, "exact"), bitwise=False, tolerance=1.0e-11)


def test_adaptive_levels_leave_the_root_records_alone():
    """The adaptive loop of every snapshot renders refined levels between two root-level renders (blacklight.cpp:196-233): they work in
    buffers of their own, and snapshot 2's root level still finds the records of snapshot 1's"""
    import blacklight_amd as bl
    fx, params, mock_args = gu.load_case("sim_adaptive")
    snaps = _snapshots(gu.golden_grid(mock_args), 3)
    for tier in ("exact", "tolerant"):
        with bl.Context(bl.Params.from_dict(params)) as ctx:
            ctx.set_arithmetic(tier)
            ctx.set_reproducible(True)
            series = []
            for grid in snaps:
                ctx.set_grid(grid)
                series.append(ctx.render_adaptive())
        assert len(series[0]) > 1   # (the case does refine)
        assert [lv["stats"].geodesics_reused for lv in series[0]] == [0] * len(series[0])
        for run in series[1:]:
            assert run[0]["stats"].geodesics_reused == 1 and all(lv["stats"].geodesics_reused == 0 for lv in run[1:])
        for grid, run in zip(snaps, series):
            with bl.Context(bl.Params.from_dict(params)) as ctx:
                ctx.set_geodesic_reuse(False)
                ctx.set_arithmetic(tier)
                ctx.set_reproducible(True)
                ctx.set_grid(grid)
                want = ctx.render_adaptive()
            assert len(want) == len(run)
            for a, b in zip(run, want):
                assert np.array_equal(a["block_locs"], b["block_locs"]) if a["block_locs"] is not None else b["block_locs"] is None
                assert gu.same_bits(a["image"], b["image"]).all() and np.array_equal(a["sample_num"], b["sample_num"])


def test_polarized_series_and_auxiliary_rows():
    """Polarized transfer (bl_shade_polarized2_kernel: locate step inside) and an auxiliary-image run over resident records"""
    import blacklight_amd as bl
    for case in ("sim_polarized", "sim_aux_images"):
        fx, params, mock_args = gu.load_case(case)
        snaps = _snapshots(gu.golden_grid(mock_args), 2)
        with bl.Context(bl.Params.from_dict(params)) as ctx:
            ctx.set_arithmetic("exact")
            frames = []
            for grid in snaps:
                ctx.set_grid(grid)
                frames.append(ctx.render())
            assert frames[0]["stats"].geodesics_reused == 0 and frames[1]["stats"].geodesics_reused == 1, case
        for grid, frame in zip(snaps, frames):
            _same_frame(frame, _fresh(params, grid, "exact"))


def test_formula_frames_reuse_too():
    """Formula mode has no snapshots, but a second render of the same frame (another frequency list would be another context) shades
    the same records: the same bits"""
    import blacklight_amd as bl
    fx, params, _ = gu.load_case("formula_64")
    with bl.Context(bl.Params.from_dict(dict(params, camera_resolution=32))) as ctx:
        ctx.set_arithmetic("exact")
        first = ctx.render()
        second = ctx.render()
        assert first["stats"].geodesics_reused == 0 and second["stats"].geodesics_reused == 1
        assert second["stats"].tail_policy == first["stats"].tail_policy and second["stats"].n_parked == first["stats"].n_parked
        _same_frame(second, first)


def test_next_snapshot_staged_beside_the_render():
    """bl_set_grid on a second host thread while bl_render runs (same geometry: the cells go up into the second cell array and take
    effect with the next render): the frames of a series staged that way are the frames of the series rendered step by step, and a
    change of geometry in between still waits its turn."""
    import threading
    import bench
    import blacklight_amd as bl
    from blacklight_amd import mock
    params = dict(bench.WORKLOAD, camera_resolution=192)
    base = mock.generate(n_r=64, n_th=64, n_ph=64)
    snaps = _snapshots(base, 6)
    want = []
    with bl.Context(bl.Params.from_dict(params)) as ctx:
        ctx.set_arithmetic("exact")
        for grid in snaps:
            ctx.set_grid(grid)
            want.append(ctx.render())
    with bl.Context(bl.Params.from_dict(params)) as ctx:
        ctx.set_arithmetic("exact")
        ctx.set_grid(snaps[0])
        got, staged = [], None

        def stage(grid, go):
            go.wait()          # (set by the rendering thread on its way into bl_render, which takes the context's render lock first thing:
            ctx.set_grid(grid)  # this thread gets the interpreter when that call releases it - the cells go up beside the render)

        for n in range(len(snaps)):
            if staged is not None:
                staged.join()
            staged = None
            go = threading.Event()
            if n + 1 < len(snaps):
                staged = threading.Thread(target=stage, args=(snaps[n + 1], go))
                staged.start()
            go.set()
            got.append(ctx.render())
        assert [g["stats"].geodesics_reused for g in got] == [0, 1, 1, 1, 1, 1]
        for a, b in zip(got, want):
            _same_frame(a, b)
        # another geometry from the second thread: excluded by the render, complete before the next one
        other = mock.generate(n_r=48, n_th=48, n_ph=48)
        worker = threading.Thread(target=ctx.set_grid, args=(other,))
        worker.start()
        during = ctx.render()          # (either grid, whole: the two calls exclude each other)
        worker.join()
        after = ctx.render()
    alone = _fresh(params, other, "exact")
    _same_frame(after, alone)
    assert gu.same_bits(during["image"], want[-1]["image"]).all() or gu.same_bits(during["image"], alone["image"]).all()
