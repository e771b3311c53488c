"""The tolerant arithmetic tier (bl_set_arithmetic, include/blacklight_amd.h): north_star's parity bar for intensities
is a tolerance - per-pixel L-infinity < 1e-6 of the image maximum - while ray-step counts and termination masks are
bit-exact. The tolerant coefficient kernel uses that room (fused multiply-adds, lighter exp / expm1 / cbrt, the
fluid-frame angle and frequency as invariants). Checked here, on every golden case it applies to and at the
benchmark's size:
  * sample_num, sample_flags, NaN masks: identical to the exact tier and to the reference;
  * intensities: within TOLERANCE of the exact tier, of the reference run with the pinned math library (tier B) and
    of the stock reference (tier A) - measured distances are ~1e-13, printed with -s;
  * configurations outside its scope run in exact arithmetic and stay bit-exact;
  * the cut decisions it defers (guard band around active thresholds, list and list overflow) change nothing."""
import json
import os

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu

TOLERANCE = 1.0e-6      # north_star: per-pixel L-infinity relative to the image maximum
EXPECTED = 1.0e-11      # what rounding-level differences amount to; a regression beyond this is a bug, not noise


def _applies(params):
    """Scope of bl_shade_fast_kernel (bl_api.hip: `fast`)."""
    def on(key):
        return str(params.get(key, "false")) == "true"
    aux = any(on(k) for k in ("image_time", "image_length", "image_lambda", "image_emission", "image_lambda_ave",
                              "image_emission_ave", "image_tau_int", "image_crossings")) or int(params.get("render_num_images", 0)) > 0
    if on("image_tau") and params["model_type"] == "formula":   # (simulation mode: an optical-depth image rides along, bl_tau_kernel)
        aux = True
    if params["model_type"] == "formula":   # bl_shade_formula_fast_kernel: plain images, no optional geometric cut
        cuts = (on("cut_omit_near") or on("cut_omit_far") or float(params.get("cut_omit_in", -1.0)) >= 0.0 or float(params.get("cut_omit_out", -1.0)) >= 0.0
                or float(params.get("cut_midplane_theta", 0.0)) != 0.0 or float(params.get("cut_midplane_z", 0.0)) != 0.0 or on("cut_plane"))
        return not aux and not cuts
    # (inter-block interpolation and slow light since round 6: bl_shade_fast_kernel behind their locate kernels)
    return (params["model_type"] == "simulation" and not aux and not on("image_polarization")
            and float(params.get("plasma_kappa_frac", 0.0)) == 0.0 and params.get("plasma_model", "ti_te_beta") == "ti_te_beta"
            and not on("ray_flat") and on("image_light"))


def _both(params, mock_args, **render_args):
    import blacklight_amd as bl
    p = bl.Params.from_dict(params)
    grid = gu.golden_grid(mock_args) if mock_args is not None else None
    out = {}
    with bl.Context(p) as ctx:
        if grid is not None:
            ctx.set_grid(grid)
        for tier in ("exact", "tolerant"):
            ctx.set_arithmetic(tier)
            out[tier] = ctx.render(**render_args)
    return out


def _distance(a, b):
    scale = np.nanmax(np.abs(b), axis=-1, keepdims=True)
    scale = np.where(scale > 0, scale, 1.0)
    with np.errstate(invalid="ignore"):
        d = np.abs(a - b) / scale
    return float(np.nanmax(d)) if np.isfinite(d).any() else 0.0


ADAPTIVE = [c for c in gu.GPU_CASES if "adaptive" in c]


@pytest.mark.parametrize("case", [c for c in gu.GPU_CASES if c not in ADAPTIVE])
def test_tolerant_tier_on_the_goldens(case, built_library):
    fx, params, mock_args = gu.load_case(case)
    out = _both(params, mock_args)
    exact, tol = out["exact"], out["tolerant"]
    n_pix = exact["sample_num"].size
    assert exact["stats"].arithmetic == 0
    assert np.array_equal(tol["sample_num"], exact["sample_num"])
    assert np.array_equal(tol["sample_flags"], exact["sample_flags"])
    assert np.array_equal(np.isnan(tol["image"]), np.isnan(exact["image"]))
    if not _applies(params):
        polarized = params["model_type"] == "simulation" and str(params.get("image_polarization", "false")) == "true"
        if not polarized:
            assert tol["stats"].arithmetic == 0
            assert gu.same_bits(tol["image"], exact["image"]).all()
            return
        # polarized runs: the transport between two couplings is one 4 x 4 matrix per sample, built sample-parallel in closed
        # form (bl_transport_matrix_kernel), instead of the tensor walked along the ray; frame, coupling and the per-frequency
        # coefficient formulas stay exact (the reference's polarized step can amplify last-place differences of the coefficients far
        # beyond this bound away from the goldens: docs/notebook.md).
        assert tol["stats"].arithmetic == 1
        d_exact = _distance(tol["image"], exact["image"])
        d_b = _distance(tol["image"], gu.expected_image(fx, "B", n_pix))
        print(f"{case}: polarized, tolerant vs exact {d_exact:.2e}, vs reference (pinned math) {d_b:.2e}")
        assert d_exact < 1.0e-9 and d_b < 1.0e-9
        return
    assert tol["stats"].arithmetic == 1
    d_exact = _distance(tol["image"], exact["image"])
    d_b = _distance(tol["image"], gu.expected_image(fx, "B", n_pix))
    d_a = _distance(tol["image"], gu.expected_image(fx, "A", n_pix))
    print(f"{case}: tolerant vs exact {d_exact:.2e}, vs reference (pinned math) {d_b:.2e}, vs stock reference {d_a:.2e}, "
          f"deferred {tol['stats'].n_deferred}")
    assert d_exact < EXPECTED and d_b < EXPECTED
    if params["model_type"] == "simulation" and float(params.get("simulation_a", 0.0)) == 0.0:   # spinning cases: glibc's hypot / pow move sample counts
        assert np.array_equal(tol["sample_num"], fx["A_sample_num"])
        assert d_a < TOLERANCE


@pytest.mark.parametrize("seed", range(12))
def test_tolerant_tier_on_seeded_configurations(seed, built_library):
    """Cameras, spins, sampling modes, temperature models, cuts and frequency lists drawn inside the tier's scope."""
    rng = np.random.default_rng(7000 + seed)
    fx, params, mock_args = gu.load_case("sim_dp_interp")
    over = dict(camera_resolution=20, camera_th=float(rng.uniform(5.0, 175.0)), camera_ph=float(rng.uniform(0.0, 360.0)),
                camera_r=float(rng.uniform(30.0, 100.0)), camera_width=float(rng.uniform(8.0, 40.0)),
                camera_type=str(rng.choice(["plane", "pinhole"])),
                simulation_a=float(rng.choice([0.0, 0.0, 0.3, 0.9])), simulation_interp=str(rng.choice(["true", "false"])),
                plasma_use_p=str(rng.choice(["true", "false"])), plasma_rat_high=float(rng.uniform(3.0, 40.0)),
                fallback_rho=1.0e-6, fallback_pgas=1.0e-8,
                cut_sigma_max=float(rng.choice([-1.0, 1.0, 10.0])), cut_theta_e_max=float(rng.choice([-1.0, 50.0])),
                cut_beta_inverse_min=float(rng.choice([-1.0, 1.0e-3])), image_num_frequencies=int(rng.choice([1, 4])))
    # NaN for off-grid samples only where the rays stay inside the grid's outer edge (r = 52): a camera beyond it with
    # fallback_nan would make the whole image NaN and test nothing but the mask
    over["fallback_nan"] = str(rng.choice(["true", "false"])) if over["camera_r"] < 50.0 else "false"
    if over["camera_type"] == "pinhole":
        over["camera_width"] = float(rng.uniform(0.05, 0.4)) * over["camera_r"]
    if over["image_num_frequencies"] > 1:
        over.update(image_frequency_start=1.0e11, image_frequency_end=float(10.0 ** rng.uniform(11.3, 12.0)), image_frequency_spacing="log")
    mesh = [{}, dict(_blocks=[2, 2, 2]), dict(_refined=1)][seed % 3]
    params = dict(params, **over)
    assert _applies(params)
    out = _both(params, dict(mock_args, **mesh))
    exact, tol = out["exact"], out["tolerant"]
    assert tol["stats"].arithmetic == 1
    assert np.array_equal(tol["sample_num"], exact["sample_num"]) and np.array_equal(tol["sample_flags"], exact["sample_flags"])
    assert np.array_equal(np.isnan(tol["image"]), np.isnan(exact["image"])), over
    d = _distance(tol["image"], exact["image"])
    print(f"seed {seed}: {d:.2e}, deferred {tol['stats'].n_deferred}, NaN pixels {int(np.isnan(exact['image']).sum())} of {exact['image'].size}")
    assert d < EXPECTED, over
    assert np.isfinite(exact["image"]).mean() > 0.5


@pytest.mark.parametrize("seed", range(8))
def test_tolerant_tier_with_power_laws_and_cartesian_grids(seed, built_library):
    """The general instantiation of bl_shade_fast_kernel: power-law electrons beside the thermal ones
    (simulation_coefficients.cpp:556-584) and simulations in Cartesian Kerr-Schild coordinates, separately and together."""
    rng = np.random.default_rng(9100 + seed)
    fx, params, mock_args = gu.load_case("sim_dp_interp")
    over = dict(camera_resolution=24, camera_th=float(rng.uniform(10.0, 170.0)), camera_ph=float(rng.uniform(0.0, 360.0)),
                simulation_a=float(rng.choice([0.0, 0.5, 0.9])), simulation_interp=str(rng.choice(["true", "false"])),
                fallback_nan="false", fallback_rho=1.0e-6, fallback_pgas=1.0e-8, cut_sigma_max=float(rng.choice([-1.0, 5.0])),
                image_num_frequencies=int(rng.choice([1, 3, 6])))   # (the mock grid read as Cartesian covers one octant: no NaN fallback)
    if seed % 4 != 0:
        over.update(plasma_power_frac=float(rng.uniform(0.05, 0.6)), plasma_p=float(rng.uniform(2.1, 3.5)),
                    plasma_gamma_min=float(rng.uniform(1.0, 10.0)), plasma_gamma_max=float(rng.uniform(500.0, 5000.0)))
    if seed % 2 == 0:
        over["simulation_coord"] = "cks"
    if over["image_num_frequencies"] > 1:
        over.update(image_frequency_start=1.0e11, image_frequency_end=float(10.0 ** rng.uniform(11.3, 12.0)), image_frequency_spacing="log")
    params = dict(params, **over)
    assert _applies(params)
    out = _both(params, mock_args)
    exact, tol = out["exact"], out["tolerant"]
    assert tol["stats"].arithmetic == 1 and exact["stats"].arithmetic == 0
    assert np.array_equal(tol["sample_num"], exact["sample_num"]) and np.array_equal(tol["sample_flags"], exact["sample_flags"])
    assert np.array_equal(np.isnan(tol["image"]), np.isnan(exact["image"])), over
    d = _distance(tol["image"], exact["image"])
    print(f"seed {seed}: {d:.2e}, deferred {tol['stats'].n_deferred}")
    assert d < EXPECTED, over
    assert np.isfinite(exact["image"]).mean() > 0.5 and np.nanmax(exact["image"]) > 0.0


@pytest.mark.parametrize("seed", [235, 355, 427, 462, 3, 11])
def test_tolerant_tier_keeps_the_nan_mask_behind_thick_steps(seed, built_library):
    """An optically thick step replaces the intensity behind it (unpolarized.cpp:103-104) - a NaN from an off-grid sample farther
    along the ray included. The tolerant tier's records are affine maps I <- a I + c with a = 0 for such a step, and 0 x NaN is NaN:
    its transfer kernels (lane per ray, four lanes per ray with composed maps, per-frequency) have to treat a = 0 as the
    replacement it stands for. Configurations with fallback_nan and thick steps in front of off-grid samples, found by
    tools/gpu_fuzz_wide.py (its generator is the parity test's); counts, flags and NaN mask against the exact tier."""
    from test_gpu_parity import _random_configuration
    base, over, mesh = _random_configuration(seed)
    fx, params, mock_args = gu.load_case(base)
    params = dict(params, **over)
    out = _both(params, dict(mock_args, **mesh) if mock_args is not None else None)
    exact, tol = out["exact"], out["tolerant"]
    assert np.array_equal(tol["sample_num"], exact["sample_num"]) and np.array_equal(tol["sample_flags"], exact["sample_flags"])
    assert np.array_equal(np.isnan(tol["image"]), np.isnan(exact["image"])), over
    if tol["stats"].arithmetic == 1:
        assert _distance(tol["image"], exact["image"]) < EXPECTED, over
    else:
        assert gu.same_bits(tol["image"], exact["image"]).all()


@pytest.mark.parametrize("frequencies,spin", [(1, 0.0), (1, 0.9), (5, 0.0)])
def test_tolerant_tier_on_a_cartesian_grid_with_thermal_electrons(frequencies, spin, built_library):
    """One block, trilinear sampling, thermal electrons, no auxiliary row - everything the fused kernel asks for except the
    coordinates: a Cartesian Kerr-Schild grid has to go through the locate kernel (the fused kernel's locate step is the spherical
    one; a randomised sweep, tools/gpu_fuzz_tiers.py, found such runs sent to it). Counts, S_in and NaN mask equal, image at
    rounding level."""
    fx, params, mock_args = gu.load_case("sim_dp_interp")
    params = dict(params, camera_resolution=24, camera_th=75.0, camera_ph=40.0, simulation_a=spin, simulation_interp="true", simulation_coord="cks",
                  fallback_nan="false", fallback_rho=1.0e-6, fallback_pgas=1.0e-8, image_num_frequencies=frequencies)
    if frequencies > 1:
        params.update(image_frequency_start=1.0e11, image_frequency_end=6.0e11, image_frequency_spacing="log")
    assert _applies(params)
    out = _both(params, mock_args)
    exact, tol = out["exact"], out["tolerant"]
    assert tol["stats"].arithmetic == 1 and exact["stats"].arithmetic == 0
    assert tol["stats"].launches_locate == exact["stats"].launches_locate > 0
    assert tol["stats"].n_gathers == exact["stats"].n_gathers
    assert np.array_equal(tol["sample_num"], exact["sample_num"]) and np.array_equal(np.isnan(tol["image"]), np.isnan(exact["image"]))
    d = _distance(tol["image"], exact["image"])
    print(f"cks, thermal, {frequencies} frequencies, a = {spin}: {d:.2e}")
    assert d < EXPECTED and np.nanmax(exact["image"]) > 0.0


@pytest.mark.parametrize("seed", range(8))
def test_tolerant_tier_with_an_optical_depth_image(seed, built_library):
    """image_tau as the only auxiliary row: the fast coefficient kernel leaves alpha x length of every sample beside its
    transfer records and bl_tau_kernel sums them far -> near (unpolarized.cpp:63-151); every other auxiliary row still
    sends the run to the exact kernels. Rows I_nu and tau against the exact tier, NaN pixels (flagged rays with
    fallback_nan) in the same places; cameras, spins, frequency lists, power laws, Cartesian grids, few-step rays and
    an unbounded guard band (every sample through the exact second pass) among the draws."""
    import blacklight_amd as bl
    rng = np.random.default_rng(5200 + seed)
    fx, params, mock_args = gu.load_case("sim_dp_interp")
    over = dict(camera_resolution=24, camera_th=float(rng.uniform(10.0, 170.0)), camera_ph=float(rng.uniform(0.0, 360.0)),
                simulation_a=float(rng.choice([0.0, 0.5, 0.9])), simulation_interp=str(rng.choice(["true", "false"])),
                fallback_nan=str(rng.choice(["true", "false"])), fallback_rho=1.0e-6, fallback_pgas=1.0e-8,
                cut_sigma_max=float(rng.choice([-1.0, 5.0])), image_num_frequencies=int(rng.choice([1, 3, 6])), image_tau="true")
    if seed >= 4:
        over["cut_sigma_max"] = 5.0   # (a threshold for the guard band to act on)
    if seed % 4 == 1:
        over.update(plasma_power_frac=0.3, plasma_p=2.8, plasma_gamma_min=2.0, plasma_gamma_max=2000.0)
    if seed % 4 == 2:
        over.update(simulation_coord="cks", fallback_nan="false")
    if seed % 4 == 3:
        over.update(ray_max_steps=560, fallback_nan="true")   # rays that run into the step limit: NaN pixels in both rows
    if over["image_num_frequencies"] > 1:
        over.update(image_frequency_start=1.0e11, image_frequency_end=float(10.0 ** rng.uniform(11.3, 12.0)), image_frequency_spacing="log")
    params = dict(params, **over)
    n_nu = over["image_num_frequencies"]
    with bl.Context(bl.Params.from_dict(params)) as ctx:
        ctx.set_grid(gu.golden_grid(mock_args))
        exact = ctx.render()
        ctx.set_arithmetic("tolerant")
        tol = ctx.render()
        if seed >= 4:
            ctx.debug_set_guard_band(1.0e30)
            wide = ctx.render()
            assert wide["stats"].n_deferred > 10 * max(tol["stats"].n_deferred, 1)
            tol = wide
    assert exact["stats"].arithmetic == 0 and tol["stats"].arithmetic == 1
    assert exact["image"].shape == tol["image"].shape == (2 * n_nu, 24 * 24)
    assert np.array_equal(tol["sample_num"], exact["sample_num"]) and np.array_equal(tol["sample_flags"], exact["sample_flags"])
    assert np.array_equal(np.isnan(tol["image"]), np.isnan(exact["image"])), over
    d = _distance(tol["image"], exact["image"])   # (every row against its own maximum)
    print(f"seed {seed}: {d:.2e}, NaN pixels {int(np.isnan(exact['image'][0]).sum())}, deferred {tol['stats'].n_deferred}")
    assert d < EXPECTED, over
    assert np.nanmax(exact["image"][n_nu:]) > 0.0 and np.nanmax(exact["image"][:n_nu]) > 0.0
    if seed % 4 == 3:
        assert np.isnan(exact["image"]).any()
    # any other auxiliary row: exact arithmetic, as before
    with bl.Context(bl.Params.from_dict(dict(params, image_lambda="true"))) as ctx:
        ctx.set_grid(gu.golden_grid(mock_args))
        ctx.set_arithmetic("tolerant")
        assert ctx.render()["stats"].arithmetic == 0


@pytest.mark.parametrize("warp", ["theta", "phi", "both"])
def test_tolerant_locate_on_unevenly_spaced_axes(warp, built_library):
    """The fused kernel guesses the cell along evenly spaced axes and searches the others (BlGridDevice::uniform_mask). The mock's
    polar and azimuthal faces are even; here they are warped smoothly (cell data unchanged - a test of the search, not a disc), so
    that the tolerant locate step takes its search path: counts, NaN mask and S_in equal to the exact tier's and the oracle's,
    image at rounding level."""
    import dataclasses
    import blacklight_amd as bl
    from blacklight_amd import _capi
    import oracle_api
    fx, params, mock_args = gu.load_case("sim_dp_interp")
    params = dict(params, camera_resolution=32, simulation_a=0.5, camera_th=70.0, fallback_nan="false", fallback_rho=1.0e-6, fallback_pgas=1.0e-8)
    grid = gu.golden_grid(mock_args)

    def warped(faces, amplitude):
        lo, hi = faces[0, 0], faces[0, -1]
        u = (faces - lo) / (hi - lo)
        out = lo + (hi - lo) * (u + amplitude * np.sin(2.0 * np.pi * u) / (2.0 * np.pi))
        out = out.astype(np.float32).astype(np.float64)
        out[0, 0], out[0, -1] = lo, hi
        return np.ascontiguousarray(out)

    changes = {}
    if warp in ("theta", "both"):
        x2f = warped(grid.x2f, 0.6)
        changes.update(x2f=x2f, x2v=np.ascontiguousarray((0.5 * (x2f[:, :-1] + x2f[:, 1:])).astype(np.float32).astype(np.float64)))
    if warp in ("phi", "both"):
        x3f = warped(grid.x3f, -0.5)
        changes.update(x3f=x3f, x3v=np.ascontiguousarray((0.5 * (x3f[:, :-1] + x3f[:, 1:])).astype(np.float32).astype(np.float64)))
    grid = dataclasses.replace(grid, **changes)
    p = bl.Params.from_dict(params)
    with bl.Context(p) as ctx:
        ctx.set_grid(grid)
        exact = ctx.render()
        ctx.set_arithmetic("tolerant")
        tol = ctx.render()
    want = oracle_api.render(p.ptr, grid.desc(), _capi.RenderDesc, _capi.CameraFrame, n_rays=32 * 32, max_steps=int(p.get("ray_max_steps")))
    assert gu.same_bits(exact["image"], want["image"]).all() and exact["stats"].n_gathers == want["n_gathers"]
    assert tol["stats"].arithmetic == 1 and tol["stats"].n_gathers == exact["stats"].n_gathers
    assert np.array_equal(tol["sample_num"], exact["sample_num"]) and np.array_equal(np.isnan(tol["image"]), np.isnan(exact["image"]))
    d = _distance(tol["image"], exact["image"])
    print(f"{warp}: {d:.2e}, deferred {tol['stats'].n_deferred}")
    assert d < EXPECTED


@pytest.mark.parametrize("spin,camera_th", [(0.0, 75.0), (0.9, 20.0), (0.0, 179.0)])
def test_tolerant_tier_over_a_refined_mesh_locates_inside_the_coefficient_kernel(spin, camera_th, built_library):
    """A two-level mesh whose boxes and rows are evenly spaced in log r, theta and phi: bl_shade_fused2_kernel<..., kRefined> guesses
    the box of the block lattice, then the cell in the block's rows, and confirms both by the rows' faces (no locate kernel, no
    located samples). Counts, flags, S_in and NaN mask are the exact tier's (whose locate kernel does the reference's search,
    simulation_sampling.cpp:352-394, :458-490); the image at rounding level; the same with every sample left to the exact kernel's
    second pass (which then locates them itself on the tables in HBM), and through the locate kernel + bl_shade_fast_kernel."""
    import blacklight_amd as bl
    fx, params, mock_args = gu.load_case("sim_dp_interp")
    params = dict(params, camera_resolution=40, camera_th=camera_th, camera_ph=130.0, simulation_a=spin, simulation_interp="true",
                  fallback_nan="false", fallback_rho=1.0e-6, fallback_pgas=1.0e-8)
    grid = gu.golden_grid(dict(mock_args, _refined=1))
    with bl.Context(bl.Params.from_dict(params)) as ctx:
        ctx.set_grid(grid)
        ctx.set_arithmetic("exact")
        exact = ctx.render()
        ctx.debug_set_switches("NO_FUSED_LOCATE")
        exact_outside = ctx.render()
        ctx.set_arithmetic("tolerant")
        outside = ctx.render()
        ctx.debug_set_switches()
        inside = ctx.render()
        ctx.debug_set_guard_band(1.0e30)
        deferred = ctx.render()
    # (the exact tier's kernel with the locate step inside takes the mesh too - bl_shade_exact2_kernel, locate_exact<., kMeshes> - and gives
    # the bits of the path through the locate kernel)
    assert exact["stats"].launches_locate == 0 and exact["stats"].fused_variant == 3
    assert exact_outside["stats"].launches_locate == 1 and exact_outside["stats"].fused_variant == 0
    assert gu.same_bits(exact["image"], exact_outside["image"]).all() and np.array_equal(exact["sample_num"], exact_outside["sample_num"])
    assert exact["stats"].n_gathers == exact_outside["stats"].n_gathers
    assert inside["stats"].arithmetic == 1 and inside["stats"].fused_variant == 2 and inside["stats"].launches_locate == 0 and inside["stats"].composed_maps == 1
    assert outside["stats"].arithmetic == 1 and outside["stats"].fused_variant == 0 and outside["stats"].launches_locate == 1
    assert deferred["stats"].fused_variant == 2 and deferred["stats"].n_deferred > 100 * max(inside["stats"].n_deferred, 1)
    for got in (inside, outside, deferred):
        assert got["stats"].n_gathers == exact["stats"].n_gathers
        assert np.array_equal(got["sample_num"], exact["sample_num"]) and np.array_equal(got["sample_flags"], exact["sample_flags"])
        assert np.array_equal(np.isnan(got["image"]), np.isnan(exact["image"]))
        assert _distance(got["image"], exact["image"]) < EXPECTED
    # few samples are left undecided by the guesses (a float log2 r, boxes and cells evenly spaced to 1e-4)
    assert inside["stats"].n_deferred < 2.0e-3 * inside["stats"].n_gathers, (inside["stats"].n_deferred, inside["stats"].n_gathers)
    print(f"a = {spin}: inside {_distance(inside['image'], exact['image']):.2e}, deferred {inside['stats'].n_deferred} of {inside['stats'].n_gathers}")


@pytest.mark.parametrize("split,block_interp", [(2, False), ((4, 3, 4), False), (2, True)])
def test_meshes_of_many_small_blocks(split, block_interp, built_library):
    """The two-level mesh cut into smaller MeshBlocks (288 / 2 304 blocks): more distinct coordinate rows, a finer block lattice - at
    split 4 beyond what the locate kernel stages in LDS, so that it searches the tables where they lie in HBM. Exact tier against the
    CPU oracle bit for bit (whose search is the reference's scan over all blocks), tolerant tier against the exact one."""
    import blacklight_amd as bl
    from blacklight_amd import _capi
    import oracle_api
    fx, params, mock_args = gu.load_case("sim_dp_interp")
    params = dict(params, camera_resolution=24, camera_th=70.0, camera_ph=200.0, simulation_interp="true", fallback_nan="false",
                  fallback_rho=1.0e-6, fallback_pgas=1.0e-8, simulation_block_interp="true" if block_interp else "false")
    grid = gu.subdivide_blocks(gu.golden_grid(dict(mock_args, _refined=1)), split)
    p = bl.Params.from_dict(params)
    with bl.Context(p) as ctx:
        if block_interp:
            ctx.set_undefined_policy("edge")
        ctx.set_grid(grid)
        ctx.set_arithmetic("exact")
        exact = ctx.render()
        ctx.set_arithmetic("tolerant")
        tol = ctx.render()
        # ... and with the mesh's tables searched where they lie in HBM (what a mesh beyond the LDS budgets gets), both tiers
        ctx.debug_set_switches("GENERAL_LOCATE", "NO_FUSED_LOCATE")
        tol_hbm = ctx.render()
        ctx.set_arithmetic("exact")
        exact_hbm = ctx.render()
    assert exact_hbm["stats"].launches_locate == 1 and gu.same_bits(exact_hbm["image"], exact["image"]).all()
    assert np.array_equal(exact_hbm["sample_num"], exact["sample_num"]) and exact_hbm["stats"].n_gathers == exact["stats"].n_gathers
    assert tol_hbm["stats"].launches_locate == 1 and tol_hbm["stats"].arithmetic == 1 and tol_hbm["stats"].n_gathers == exact["stats"].n_gathers
    assert np.array_equal(tol_hbm["sample_num"], exact["sample_num"]) and _distance(tol_hbm["image"], exact["image"]) < EXPECTED
    if not block_interp:   # (the oracle refuses the undefined reads that BL_UNDEFINED_EDGE defines)
        want = oracle_api.render(p.ptr, grid.desc(), _capi.RenderDesc, _capi.CameraFrame, n_rays=24 * 24, max_steps=int(p.get("ray_max_steps")))
        assert gu.same_bits(exact["image"], want["image"]).all() and exact["stats"].n_gathers == want["n_gathers"]
        assert np.array_equal(exact["sample_num"], want["sample_num"])
    assert tol["stats"].arithmetic == 1 and tol["stats"].n_gathers == exact["stats"].n_gathers
    assert np.array_equal(tol["sample_num"], exact["sample_num"]) and np.array_equal(np.isnan(tol["image"]), np.isnan(exact["image"]))
    d = _distance(tol["image"], exact["image"])
    print(f"split {split}: {grid.prim.shape[1]} blocks, fused_variant {tol['stats'].fused_variant}, {d:.2e}, deferred {tol['stats'].n_deferred}")
    assert d < EXPECTED and np.nanmax(exact["image"]) > 0.0


@pytest.mark.parametrize("band,resolution,frequencies,variant", [(1.0e30, 24, 1, ""), (1.0e30, 56, 1, ""), (1.0e30, 24, 5, ""), (1.0e30, 56, 5, ""),
                                                                 (1.0e30, 24, 1, "power"), (1.0e30, 24, 3, "cks"), (1.0e30, 24, 3, "cks power")])
def test_deferred_cut_decisions(band, resolution, frequencies, variant, built_library):
    """An unbounded guard band defers every sample that reaches the cell cuts to the exact kernel: through the list
    (24^2 rays), and past its capacity (56^2 rays x ~700 samples > 2^20 entries: every record is then shaded by the
    exact kernel). Same image either way - with one frequency (transfer records) and with five (per-sample factors for
    bl_transfer_freq_kernel, which the exact pass then has to leave as well)."""
    import blacklight_amd as bl
    # sim_cuts: rho, B and 1 / beta thresholds behind geometric cuts; sim_dp_interp (sigma < 1 only) keeps nearly every sample
    fx, params, mock_args = gu.load_case("sim_cuts" if resolution < 40 else "sim_dp_interp")
    params = dict(params, camera_resolution=resolution)
    # (the exact kernel's instantiations behind the general fast kernel: extended for power laws, general for Cartesian grids)
    if "power" in variant:
        params.update(plasma_power_frac=0.3, plasma_p=2.5, plasma_gamma_min=1.0, plasma_gamma_max=1000.0)
    if "cks" in variant:
        params.update(simulation_coord="cks", simulation_a=0.5, fallback_nan="false", fallback_rho=1.0e-6, fallback_pgas=1.0e-8)
    if frequencies > 1:
        params.update(image_num_frequencies=frequencies, image_frequency_start=1.0e11, image_frequency_end=8.0e11, image_frequency_spacing="log")
        params.pop("image_frequency", None)
    p = bl.Params.from_dict(params)
    with bl.Context(p) as ctx:
        ctx.set_grid(gu.golden_grid(mock_args))
        exact = ctx.render()
        ctx.set_arithmetic("tolerant")
        plain = ctx.render()
        ctx.debug_set_guard_band(band)
        wide = ctx.render()
    assert wide["stats"].arithmetic == 1 and wide["stats"].n_deferred > 10 * max(plain["stats"].n_deferred, 1)
    if resolution > 40:
        assert wide["stats"].n_deferred > (1 << 20)
    assert np.array_equal(wide["sample_num"], exact["sample_num"])
    assert np.array_equal(np.isnan(wide["image"]), np.isnan(exact["image"]))
    assert _distance(wide["image"], exact["image"]) < EXPECTED
    assert _distance(plain["image"], exact["image"]) < EXPECTED


def test_tolerant_functions_are_accurate(built_library):
    """exp, expm1, cbrt, reciprocal, reciprocal square root, acos and atan2 of the tolerant tier against numpy's long double, its
    Bessel functions against scipy."""
    import blacklight_amd as bl
    fx, params, _ = gu.load_case("formula_flat")
    rng = np.random.default_rng(3)
    ld = np.longdouble
    with bl.Context(bl.Params.from_dict(params)) as ctx:
        def worst(op, x, ref):
            got = ctx.debug_math(op, x)
            want = ref(x.astype(ld))
            ok = np.isfinite(want.astype(np.float64)) & (want != 0)
            return float(np.max(np.abs((got[ok].astype(ld) - want[ok]) / want[ok])))
        x = np.concatenate([rng.uniform(-700.0, 700.0, 200000), rng.uniform(-1.0, 1.0, 200000), 10.0 ** rng.uniform(-300, -1, 1000)])
        assert worst(20, x, np.exp) < 4.0e-16
        xm = np.concatenate([rng.uniform(-40.0, 700.0, 200000), rng.uniform(-1.0, 1.0, 200000),
                             10.0 ** rng.uniform(-300, -1, 1000), -(10.0 ** rng.uniform(-300, -1, 1000))])
        assert worst(21, xm, np.expm1) < 6.0e-16
        xp = 10.0 ** rng.uniform(-300.0, 300.0, 400000)
        assert worst(22, xp, np.cbrt) < 4.0e-16
        assert worst(23, xp, lambda v: 1 / v) < 2.5e-15          # one Newton step on the hardware's 26 bits (bl_fastmath.h)
        xq = 10.0 ** rng.uniform(-290.0, 300.0, 400000)
        assert worst(24, xq, lambda v: 1 / np.sqrt(v)) < 4.0e-16
        special = np.array([0.0, np.inf, np.nan, 5e-324, 1.0, 8.0, 27.0e300])
        got = ctx.debug_math(22, special)
        assert got[0] == 0.0 and got[1] == np.inf and np.isnan(got[2]) and got[4] == 1.0 and abs(got[5] - 2.0) < 1e-15
        assert abs(got[3] / np.cbrt(ld(5e-324)) - 1) < 1e-15
        sat = ctx.debug_math(20, np.array([-800.0, 800.0, np.nan]))
        assert sat[0] == 0.0 and sat[1] == np.inf and np.isnan(sat[2])
        sat = ctx.debug_math(21, np.array([-800.0, 800.0, np.nan, 0.0]))
        assert sat[0] == -1.0 and sat[1] == np.inf and np.isnan(sat[2]) and sat[3] == 0.0
        rc = ctx.debug_math(23, np.array([0.0, np.inf, -np.inf, np.nan]))
        assert rc[0] == np.inf and rc[1] == 0.0 and rc[2] == 0.0 and np.isnan(rc[3])
        # acos and atan2 of the tolerant locate step (ops 28, 29): absolute error, against long double
        xa = np.concatenate([rng.uniform(-1.0, 1.0, 300000), 1.0 - 10.0 ** rng.uniform(-16, 0, 50000), -1.0 + 10.0 ** rng.uniform(-16, 0, 50000),
                             np.array([0.0, 1.0, -1.0, 0.5, -0.5, 0.4999999999999999, 1e-300])])
        got = ctx.debug_math(28, xa)
        assert float(np.max(np.abs(got.astype(ld) - np.arccos(xa.astype(ld))))) < 1.0e-15
        ya = np.concatenate([rng.uniform(-60.0, 60.0, 300000), 10.0 ** rng.uniform(-12, 2, 100000) * rng.choice([-1.0, 1.0], 100000)])
        xb2 = np.concatenate([rng.uniform(-60.0, 60.0, 300000), 10.0 ** rng.uniform(-12, 2, 100000) * rng.choice([-1.0, 1.0], 100000)])
        got = ctx.debug_math(29, ya, xb2)
        assert float(np.max(np.abs(got.astype(ld) - np.arctan2(ya.astype(ld), xb2.astype(ld))))) < 1.0e-15
        edge = ctx.debug_math(29, np.array([0.0, 0.0, 1.0, -1.0, 0.0, 7.0, -7.0]), np.array([1.0, -1.0, 0.0, 0.0, 0.0, 7.0, -7.0]))
        assert edge[0] == 0.0 and abs(edge[1] - np.pi) < 5e-16 and abs(edge[2] - np.pi / 2) < 5e-16 and abs(edge[3] + np.pi / 2) < 5e-16
        assert edge[4] == 0.0 and abs(edge[5] - np.pi / 4) < 5e-16 and abs(edge[6] + 3 * np.pi / 4) < 5e-16
        # K_0, K_1, K_2 (ops 25-27) over the arguments 1 / Theta_e takes (Theta_e from 0.01 to 1e4), against scipy
        from scipy import special as sp
        xb = np.concatenate([10.0 ** rng.uniform(-4.0, 2.0, 100000), rng.uniform(1.5, 2.5, 20000)])
        for op, order in ((25, 0), (26, 1), (27, 2)):
            got = ctx.debug_math(op, xb)
            want = sp.kn(order, xb)
            ok = want > 1.0e-300
            assert np.max(np.abs(got[ok] / want[ok] - 1.0)) < 5.0e-14, order


WINDOWS = os.path.join(gu.GOLDEN_DIR, "window_1024.npz")


def test_tolerant_tier_at_the_benchmark_size(built_library):
    """The 1024^2 frame over the 256^3 grid that bench.py times: tolerant vs exact over the whole frame, and the three
    windows the unmodified reference computed (tests/golden/window_1024.npz) in the tolerant tier."""
    import blacklight_amd as bl
    from blacklight_amd import mock
    import bench
    grid = mock.generate(n_r=256, n_th=256, n_ph=256)
    p = bl.Params.from_dict(dict(bench.WORKLOAD))
    res, bs = 1024, 16
    with bl.Context(p) as ctx:
        ctx.set_grid(grid)
        exact = ctx.render()
        ctx.set_arithmetic("tolerant")
        tol = ctx.render()
        assert tol["stats"].arithmetic == 1
        assert np.array_equal(tol["sample_num"], exact["sample_num"]) and np.array_equal(tol["sample_flags"], exact["sample_flags"])
        assert np.array_equal(np.isnan(tol["image"]), np.isnan(exact["image"]))
        assert tol["stats"].n_gathers == exact["stats"].n_gathers
        d = _distance(tol["image"], exact["image"])
        print(f"1024^2 frame: tolerant vs exact {d:.2e}, deferred {tol['stats'].n_deferred} of {tol['stats'].n_samples_emitted}")
        assert d < EXPECTED
        if os.path.exists(WINDOWS):
            fx = np.load(WINDOWS, allow_pickle=False)
            iv, iu = np.mgrid[0:bs, 0:bs]
            pixels = np.concatenate([((bv * bs + iv) * res + (bu * bs + iu)).reshape(-1) for bv, bu in fx["B_block_locs"]]).astype(np.int32)
            got = tol["image"][0][pixels]
            for tier in ("B", "A"):
                want = fx[f"{tier}_I_nu"].reshape(-1)
                dist = np.nanmax(np.abs(got - want)) / np.nanmax(np.abs(want))
                print(f"windows vs reference tier {tier}: {dist:.2e}")
                assert dist < (EXPECTED if tier == "B" else TOLERANCE)


@pytest.mark.parametrize("seed", range(8))
def test_tolerant_tier_with_inter_block_interpolation(seed, built_library):
    """simulation_block_interp (simulation_sampling.cpp:505-546, :1068-1321) in the tolerant tier (round 6): the primitives are the exact
    tier's - eight anchor cells named by the locate kernel, InterpolateAdvanced's weights - the arithmetic behind them the tier's.
    Meshes: equal blocks, and the two-level refined mesh; cameras, spins, cuts and frequency lists drawn."""
    rng = np.random.default_rng(9300 + seed)
    fx, params, mock_args = gu.load_case("sim_blockinterp" if seed % 2 == 0 else "sim_blockinterp_refined")
    over = dict(camera_resolution=20, camera_th=float(rng.uniform(10.0, 170.0)), camera_ph=float(rng.uniform(0.0, 360.0)),
                camera_width=float(rng.uniform(10.0, 30.0)), simulation_a=float(rng.choice([0.0, 0.5, 0.9])),
                plasma_rat_high=float(rng.uniform(3.0, 40.0)), cut_sigma_max=float(rng.choice([-1.0, 1.0, 10.0])),
                cut_theta_e_max=float(rng.choice([-1.0, 50.0])), image_num_frequencies=int(rng.choice([1, 1, 5])))
    if over["image_num_frequencies"] > 1:
        over.update(image_frequency_start=1.0e11, image_frequency_end=float(10.0 ** rng.uniform(11.3, 12.0)), image_frequency_spacing="log")
    params = dict(params, **over, plasma_model="ti_te_beta")   # (the refined golden reads electron entropy from the grid: outside the tier)
    assert _applies(params) and str(params["simulation_block_interp"]) == "true"
    import blacklight_amd as bl
    grid = gu.golden_grid(mock_args)
    out = {}
    with bl.Context(bl.Params.from_dict(params)) as ctx:
        ctx.set_undefined_policy("edge")   # (the drawn cameras may reach the upper edges of the file's last block)
        ctx.set_grid(grid)
        for tier in ("exact", "tolerant"):
            ctx.set_arithmetic(tier)
            ctx.clear_warnings()
            out[tier] = ctx.render()
    exact, tol = out["exact"], out["tolerant"]
    assert tol["stats"].arithmetic == 1 and exact["stats"].arithmetic == 0 and tol["stats"].launches_locate == 1
    assert np.array_equal(tol["sample_num"], exact["sample_num"]) and np.array_equal(tol["sample_flags"], exact["sample_flags"])
    assert tol["stats"].n_gathers == exact["stats"].n_gathers and tol["stats"].n_undefined == exact["stats"].n_undefined
    assert np.array_equal(np.isnan(tol["image"]), np.isnan(exact["image"])), over
    d = _distance(tol["image"], exact["image"])
    print(f"seed {seed}: {d:.2e}, deferred {tol['stats'].n_deferred}")
    assert d < EXPECTED, over
    assert np.isfinite(exact["image"]).mean() > 0.5 and np.nanmax(exact["image"]) > 0.0


@pytest.mark.parametrize("spin,camera_th", [(0.0, 60.0), (0.9, 100.0)])
def test_inter_block_interpolation_with_the_locate_step_inside(spin, camera_th, built_library):
    """MeshBlocks of twelve cells and more per axis, evenly spaced: bl_shade_fused2_kernel<..., kRefined> shades the samples whose eight
    anchors lie in their own block (ordinary trilinear samples: InterpolateAdvanced = InterpolateSimple there) and hands the others - the
    outer half cell of every block, a sixth of the samples here - to the exact pass through the waves' lists; that pass finds their
    anchor cells (FindNearbyInds) and shades them. Counts, S_in, undefined-read counts and NaN mask are the exact tier's, the image at
    rounding level, and the same through the locate kernel + bl_shade_fast_kernel."""
    import blacklight_amd as bl
    fx, params, mock_args = gu.load_case("sim_blockinterp")
    params = dict(params, camera_resolution=40, camera_th=camera_th, camera_ph=250.0, simulation_a=spin, plasma_model="ti_te_beta",
                  fallback_nan="false", fallback_rho=1.0e-6, fallback_pgas=1.0e-8, image_tau="false")
    grid = gu.golden_grid(dict({k: v for k, v in mock_args.items() if not k.startswith("_")}, _blocks=[2, 2, 2]))   # blocks of 16 x 12 x 16 cells
    with bl.Context(bl.Params.from_dict(params)) as ctx:
        ctx.set_undefined_policy("edge")
        ctx.set_grid(grid)
        ctx.set_arithmetic("exact")
        exact = ctx.render()
        ctx.set_arithmetic("tolerant")
        inside = ctx.render()
        ctx.debug_set_switches("NO_FUSED_LOCATE")
        outside = ctx.render()
    assert inside["stats"].fused_variant == 2 and inside["stats"].launches_locate == 0 and inside["stats"].composed_maps == 1
    assert outside["stats"].fused_variant == 0 and outside["stats"].launches_locate == 1
    assert 0.02 * inside["stats"].n_gathers < inside["stats"].n_deferred < 0.4 * inside["stats"].n_gathers   # (the outer half cells)
    for got in (inside, outside):
        assert got["stats"].arithmetic == 1 and got["stats"].n_gathers == exact["stats"].n_gathers and got["stats"].n_undefined == exact["stats"].n_undefined
        assert np.array_equal(got["sample_num"], exact["sample_num"]) and np.array_equal(np.isnan(got["image"]), np.isnan(exact["image"]))
        assert _distance(got["image"], exact["image"]) < EXPECTED
    print(f"a = {spin}: {_distance(inside['image'], exact['image']):.2e}, to the exact pass {inside['stats'].n_deferred} of {inside['stats'].n_gathers}")


@pytest.mark.parametrize("case", ["slow_interp", "slow_nearest"])
def test_tolerant_tier_with_slow_light(case, built_library):
    """slow_light_on (simulation_sampling.cpp:296-349, :736-912) in the tolerant tier (round 6): every image of the fixtures' series,
    one frequency and four, against the exact tier - whose images are the reference's (tests/test_gpu_slow_light.py)."""
    import blacklight_amd as bl
    fx = np.load(os.path.join(gu.GOLDEN_DIR, f"{case}.npz"), allow_pickle=False)
    base = json.loads(str(fx["params"]))
    grids = gu.slow_light_grids(fx)
    file_times = [float(t) for t in fx["file_times"]]
    for n_freq in (1, 4):
        params = dict(base, image_num_frequencies=n_freq)
        if n_freq > 1:
            params.update(image_frequency_start=1.0e11, image_frequency_end=6.0e11, image_frequency_spacing="log")
        assert _applies(params)
        images = {}
        for tier in ("exact", "tolerant"):
            with bl.Context(bl.Params.from_dict(params)) as ctx:
                ctx.set_arithmetic(tier)
                held, got = None, []
                for image, (t_cam, files) in enumerate(gu.slow_light_windows(params, file_times)):
                    new = len(files) if held is None or files[0] - held[0] >= len(files) else files[0] - held[0]
                    if 0 < new < len(files):
                        ctx.shift_grid_slices(new)
                    for n in range(new):
                        ctx.set_grid_slice(n, grids[files[n]], file_times[files[n]])
                    held = files
                    ctx.set_snapshot(image)
                    out = ctx.render()
                    assert out["stats"].arithmetic == (1 if tier == "tolerant" else 0)
                    got.append(out)
                images[tier] = got
        for image, (e, t) in enumerate(zip(images["exact"], images["tolerant"])):
            assert np.array_equal(t["sample_num"], e["sample_num"]) and np.array_equal(t["sample_flags"], e["sample_flags"])
            assert np.array_equal(np.isnan(t["image"]), np.isnan(e["image"])), (case, n_freq, image)
            assert _distance(t["image"], e["image"]) < EXPECTED, (case, n_freq, image)
        assert np.nanmax(images["exact"][0]["image"]) > 0.0
