"""bl_geodesic_quad_kernel (a ray per quad of lanes: the last rays of a chunk) against bl_geodesic_kernel (a ray per lane).

The quad kernel performs the operations of the ray-per-lane kernel on the same operands, spread over four lanes: positions, step
lengths, sample counts and flags must be the same bits. BL_SWITCH_QUAD_EVERY_RAY parks every ray before its first step - the
whole frame is stepped by the quad kernel -, bl_set_tail_policy(BL_TAIL_QUAD) parks the last rays of a chunk, each somewhere along
its way; BL_TAIL_WIDE parks none (docs/notebook.md: what the quad kernel gains and where it costs). All three must give the same frame, and the oracle's (geodesics.cpp:39-396).
"""
import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


def _three_ways(ctx, extra=()):
    # the baseline is the ray-per-lane stepper alone (BL_TAIL_AUTO would park rays of formula frames itself)
    out = {}
    for name, policy, switches in (("lane", "wide", ()), ("tail", "quad", ()), ("quad", "wide", ("QUAD_EVERY_RAY",))):
        ctx.set_tail_policy(policy)
        ctx.debug_set_switches(*switches, *extra)
        out[name] = ctx.render()
    ctx.debug_set_switches()
    ctx.set_tail_policy("auto")
    return out


def _assert_same_frames(out, tier, what):
    lane = out["lane"]
    assert lane["stats"].n_parked == 0
    assert out["quad"]["stats"].n_parked == lane["stats"].n_rays, what
    for name in ("tail", "quad"):
        got = out[name]
        assert np.array_equal(got["sample_num"], lane["sample_num"]), (what, name)
        assert np.array_equal(got["sample_flags"], lane["sample_flags"]), (what, name)
        assert got["stats"].n_samples == lane["stats"].n_samples and got["stats"].n_gathers == lane["stats"].n_gathers, (what, name)
        if tier == "exact":   # the transfer equation runs along a ray in sample order whatever the order of the records
            assert gu.same_bits(got["image"], lane["image"]).all(), (what, name)
        else:   # the tolerant tier composes neighbouring records: the grouping follows the order of the records
            assert np.array_equal(np.isnan(got["image"]), np.isnan(lane["image"])), (what, name)
            with np.errstate(invalid="ignore"):
                assert np.nanmax(np.abs(got["image"] - lane["image"])) <= 1.0e-13 * np.nanmax(np.abs(lane["image"])), (what, name)


@pytest.mark.parametrize("seed", range(6))
def test_formula_frames_through_the_quad_kernel(seed):
    """Formula mode (nothing of a ray is skipped), spin and camera drawn at random; the oracle's frame beside them"""
    import blacklight_amd as bl
    from blacklight_amd import _capi
    import oracle_api
    rng = np.random.default_rng(7100 + seed)
    fx, params, _ = gu.load_case("formula_dp")
    res = 24
    over = dict(camera_resolution=res, camera_r=float(rng.uniform(40.0, 1000.0)), camera_th=float(rng.uniform(5.0, 175.0)),
                camera_ph=float(rng.uniform(0.0, 360.0)), camera_width=float(rng.uniform(12.0, 40.0)),
                formula_spin=float([0.0, 0.9, 0.5, 0.0, 0.99, 0.3][seed]), ray_step=float(rng.choice([0.01, 0.03])),
                ray_max_steps=int(rng.choice([400, 2500])), ray_flat="true" if seed == 5 else "false")
    params = dict(params, **over)
    p = bl.Params.from_dict(params)
    with bl.Context(p) as ctx:
        for tier in ("exact", "tolerant"):
            ctx.set_arithmetic(tier)
            out = _three_ways(ctx)
            _assert_same_frames(out, tier, over)
            if tier == "exact":
                exact = out
    want = oracle_api.render(p.ptr, None, _capi.RenderDesc, _capi.CameraFrame, n_rays=res * res, max_steps=int(p.get("ray_max_steps")), n_freq=1)
    for name in ("lane", "quad"):
        assert np.array_equal(exact[name]["sample_num"], want["sample_num"]) and np.array_equal(exact[name]["sample_flags"], want["sample_flags"]), over
        assert gu.same_bits(exact[name]["image"], want["image"]).all(), over
    assert want["sample_num"].max() > 30


@pytest.mark.parametrize("seed", range(4))
def test_simulation_frames_through_the_quad_kernel(seed):
    """Simulation mode with every step recorded (the instantiation that skips the empty shell parks nothing), both tiers"""
    import blacklight_amd as bl
    rng = np.random.default_rng(7200 + seed)
    fx, params, mock_args = gu.load_case("sim_dp_interp")
    over = dict(camera_resolution=32, camera_r=float(rng.uniform(60.0, 300.0)), camera_th=float(rng.uniform(10.0, 170.0)),
                camera_ph=float(rng.uniform(0.0, 360.0)), camera_width=float(rng.uniform(20.0, 100.0)),
                simulation_a=float([0.0, 0.9, 0.0, 0.5][seed]), ray_integrator="dp", fallback_nan="false", fallback_rho=1.0e-6,
                fallback_pgas=1.0e-8)
    p = bl.Params.from_dict(dict(params, **over))
    grid = gu.golden_grid(mock_args)
    with bl.Context(p) as ctx:
        ctx.set_grid(grid)
        for tier in ("exact", "tolerant"):
            ctx.set_arithmetic(tier)
            _assert_same_frames(_three_ways(ctx, extra=("RECORD_EVERY_STEP",)), tier, over)


def test_the_last_rays_of_a_frame_are_parked():
    """BL_TAIL_QUAD: the waves of a frame that has run out of rays hand their last ones to the quad kernel"""
    import blacklight_amd as bl
    fx, params, _ = gu.load_case("formula_dp")
    p = bl.Params.from_dict(dict(params, camera_resolution=96))
    with bl.Context(p) as ctx:
        out = _three_ways(ctx)
    _assert_same_frames(out, "exact", "formula_dp at 96^2")
    assert 0 < out["tail"]["stats"].n_parked < out["tail"]["stats"].n_rays


@pytest.mark.parametrize("spin", [0.0, 0.7])
def test_rays_predicted_long_on_their_own_compute_units(spin):
    """BL_TAIL_SPLIT: the rays whose impact parameter lies in the band around the photon ring's are parked before their first
    step and stepped by bl_geodesic_quad_kernel on a CU-masked stream beside the other stepper (hipExtStreamCreateWithCUMask): the
    same frame, bit for bit in the exact tier, to rounding in the tolerant one."""
    import bench
    import blacklight_amd as bl
    from blacklight_amd import mock
    params = dict(bench.WORKLOAD, camera_resolution=128, simulation_a=spin)
    grid = mock.generate(n_r=32, n_th=32, n_ph=32)
    with bl.Context(bl.Params.from_dict(params)) as ctx:
        ctx.set_grid(grid)
        for tier in ("exact", "tolerant"):
            ctx.set_arithmetic(tier)
            ctx.set_tail_policy("wide")
            plain = ctx.render()
            ctx.set_tail_policy("split")
            split = ctx.render()
            assert plain["stats"].tail_policy == 1 and split["stats"].tail_policy == 3
            assert plain["stats"].n_parked == 0 and 100 < split["stats"].n_parked < 128 * 128 // 4, split["stats"].n_parked
            assert np.array_equal(split["sample_num"], plain["sample_num"]) and np.array_equal(split["sample_flags"], plain["sample_flags"])
            assert split["stats"].n_samples == plain["stats"].n_samples and split["stats"].n_gathers == plain["stats"].n_gathers
            if tier == "exact":
                assert gu.same_bits(split["image"], plain["image"]).all()
            else:
                assert np.array_equal(np.isnan(split["image"]), np.isnan(plain["image"]))
                with np.errstate(invalid="ignore"):
                    assert np.nanmax(np.abs(split["image"] - plain["image"])) <= 1.0e-13 * np.nanmax(np.abs(plain["image"]))
