"""GPU parity: the HIP path (through the C-ABI) against the reference's golden vectors and the
CPU oracle. Integer outputs (sample_num, flags) and, with the pinned math library, every pixel
must be bit-exact (tier B); against the stock glibc-linked reference (tier A) the stated fp64
tolerance is per-pixel L-infinity < 1e-6 relative to the image maximum for the a = 0 cases."""
import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


def _render(case, **extra):
    import blacklight_amd as bl
    fx, params, mock_args = gu.load_case(case)
    params = dict(params)
    params.update(extra)
    p = bl.Params.from_dict(params)
    ctx = bl.Context(p)
    if mock_args is not None:
        ctx.set_grid(gu.golden_grid(mock_args))
    out = ctx.render(want_camera=True)
    out["frame"] = ctx.camera_frame
    out["frequencies"] = ctx.frequencies
    out["warnings"] = ctx.warnings
    ctx.close()
    return fx, p, out


@pytest.mark.parametrize("case", gu.GPU_CASES)
def test_tier_b_bit_exact(case, built_library):
    fx, p, out = _render(case)
    n_pix = out["sample_num"].size
    # camera frame and frequencies (host side of the boundary)
    for key in gu.FRAME_KEYS:
        assert np.array_equal(np.array(getattr(out["frame"], key)), fx[f"B_{key}"]), key
    assert np.array_equal(out["frequencies"], fx["B_image_frequencies"])
    # per-pixel initial conditions at the recorded pixels
    picks = fx["B_camera_pos_pixels"]
    assert np.array_equal(out["camera_pos"][picks], fx["B_camera_pos"])
    assert np.array_equal(out["camera_dir"][picks], fx["B_camera_dir"])
    # integer outputs: bit-exact
    assert np.array_equal(out["sample_num"], fx["B_sample_num"])
    assert np.array_equal(out["sample_flags"], fx["B_sample_flags"])
    assert out["stats"].max_sample_num == int(fx["B_geodesic_num_steps"])
    # image: bit-exact (NaN positions included)
    want = gu.expected_image(fx, "B", n_pix)
    got = out["image"]
    assert got.shape == want.shape
    assert np.array_equal(np.isnan(got), np.isnan(want))
    same = gu.same_bits(got, want)
    assert same.all(), f"{(~same).sum()} of {same.size} pixels differ"
    # false-colour renderings (rendering.cpp)
    want_render = gu.expected_rendering(fx, "B", n_pix)
    if want_render is not None:
        assert gu.same_bits(out["rendering"], want_render).all()
    # warning text of the reference
    n_bad = int(fx["B_sample_flags"].sum())
    if n_bad:
        assert f"Warning: {n_bad} out of {n_pix} geodesics terminate unexpectedly." in out["warnings"]


# sim_polarized_powerlaw and sim_polarized_kappa_mix (a = 0.9): rows move by 1.6e-5 between glibc and the pinned
# library, like the other strongly spinning cases the name filter leaves out
@pytest.mark.parametrize("case", [c for c in gu.GPU_CASES if c.startswith("sim_") and "spin" not in c
                                  and c not in ("sim_polarized_powerlaw", "sim_polarized_kappa_mix")])
def test_tier_a_tolerance(case, built_library):
    """Against the stock (glibc) reference: a = 0 cases keep sample counts and stay within 1e-6."""
    fx, p, out = _render(case)
    n_pix = out["sample_num"].size
    assert np.array_equal(out["sample_num"], fx["A_sample_num"])
    assert np.array_equal(out["sample_flags"], fx["A_sample_flags"])
    want = gu.expected_image(fx, "A", n_pix)
    got = out["image"]
    assert np.array_equal(np.isnan(got), np.isnan(want))
    if want.size:
        scale = np.nanmax(np.abs(want))
        if np.isfinite(scale) and scale > 0:
            assert np.nanmax(np.abs(got - want)) / scale < 1.0e-6
    want_render = gu.expected_rendering(fx, "A", n_pix)
    if want_render is not None:
        assert np.nanmax(np.abs(out["rendering"] - want_render)) / np.nanmax(np.abs(want_render)) < 1.0e-6


# Spinning cases against the STOCK reference (tier A): glibc's hypot / pow differ from the pinned library's in the last bit
# often enough for the step-size controller to move a few sample counts (the reference itself does that
# between an FMA and a non-FMA host), and the images then differ by more than 1e-6 in those pixels. The measured distances
# are asserted with their real bounds instead of being left out.
# (fraction of sample counts that may move, image distance) - measured on MI355X: no count moves on these small cameras,
# distances 2e-9 ... 9e-9, and 2.2e-5 in one auxiliary row of sim_polarized_powerlaw (a = 0.9, power-law electrons)
SPIN_BOUNDS = {"sim_spin_fallback": (0.0, 1.0e-6), "sim_spin_nan": (0.0, 1.0e-6), "sim_polarized_powerlaw": (0.0, 4.0e-5),
               "sim_polarized_kappa_mix": (0.0, 1.0e-6), "sim_code_kappa_fallback": (0.0, 1.0e-6)}


@pytest.mark.parametrize("case", sorted(SPIN_BOUNDS))
def test_tier_a_spinning_cases(case, built_library):
    fx, p, out = _render(case)
    n_pix = out["sample_num"].size
    max_moved, max_distance = SPIN_BOUNDS[case]
    moved = np.mean(out["sample_num"] != fx["A_sample_num"])
    assert moved <= max_moved and np.max(np.abs(out["sample_num"] - fx["A_sample_num"])) <= 2
    assert np.array_equal(out["sample_flags"], fx["A_sample_flags"])
    want = gu.expected_image(fx, "A", n_pix)
    got = out["image"]
    assert np.array_equal(np.isnan(got), np.isnan(want))
    scale = np.nanmax(np.abs(want), axis=1, keepdims=True)
    scale = np.where(scale > 0, scale, 1.0)
    distance = float(np.nanmax(np.abs(got - want) / scale))
    print(f"{case}: {moved * 100:.2f} % of sample counts moved, image distance {distance:.2e}")
    assert distance < max_distance


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("case", ["sim_dp_interp", "formula_dp"])
def test_pixel_map_and_chunking(case, overlap, built_library):
    """A shuffled pixel subset rendered in several small chunks gives the same bits per pixel, with the
    chunks back to back on one stream and with the geodesic kernel of the next chunk overlapped."""
    import blacklight_amd as bl
    fx, params, mock_args = gu.load_case(case)
    p = bl.Params.from_dict(params)
    with bl.Context(p) as ctx:
        if mock_args is not None:
            ctx.set_grid(gu.golden_grid(mock_args))
        full = ctx.render()
        rng = np.random.default_rng(7)
        n_pix = full["sample_num"].size
        subset = rng.permutation(n_pix)[: n_pix // 3].astype(np.int32)
        # a few rays' worth of sample records per scratch set (two sets when chunks overlap): several chunks
        ctx.set_scratch_limit(max(1 << 20, int(p.get("ray_max_steps")) * 600))
        ctx.set_overlap(overlap)
        part = ctx.render(pixel_map=subset)
        assert part["stats"].n_chunks > 1
    assert np.array_equal(part["sample_num"], full["sample_num"][subset])
    assert np.array_equal(part["sample_flags"], full["sample_flags"][subset])
    assert gu.same_bits(part["image"], full["image"][:, subset]).all()


AUX_ALL = dict(image_time="true", image_length="true", image_lambda="true", image_emission="true", image_tau="true",
               image_crossings="true")
AUX_SIM = dict(AUX_ALL, image_lambda_ave="true", image_emission_ave="true", image_tau_int="true")


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("mode", ["empty shell", "optical depth", "kappa", "power law cks"])
def test_round_three_paths_in_several_chunks(mode, overlap, built_library):
    """The paths added in round 3 under a scratch budget that forces several chunks (rays the record gate refused go to the next
    chunk; with bl_set_overlap two scratch sets alternate): same bits as the unconstrained render, in both tiers."""
    import blacklight_amd as bl
    fx, params, mock_args = gu.load_case("sim_dp_interp")
    params = dict(params, camera_resolution=48, fallback_nan="false", fallback_rho=1.0e-6, fallback_pgas=1.0e-8, image_num_frequencies=3,
                  image_frequency_start=1.0e11, image_frequency_end=6.0e11, image_frequency_spacing="log")
    params.pop("image_frequency", None)
    if mode == "empty shell":
        params.update(camera_r=150.0, camera_width=60.0)
    elif mode == "optical depth":
        params.update(image_tau="true", simulation_a=0.5)
    elif mode == "kappa":
        params.update(plasma_kappa_frac=0.3, plasma_kappa=4.0, plasma_w=10.0, image_num_frequencies=5)
    else:
        params.update(simulation_coord="cks", plasma_power_frac=0.2, plasma_p=3.0, plasma_gamma_min=1.0, plasma_gamma_max=1000.0, simulation_a=0.5)
    p = bl.Params.from_dict(params)
    with bl.Context(p) as ctx:
        ctx.set_grid(gu.golden_grid(mock_args))
        if mode == "kappa":
            ctx.set_undefined_policy("kappa")
        for tier in ("exact", "tolerant"):
            ctx.set_arithmetic(tier)
            ctx.set_scratch_limit(1 << 40)
            ctx.set_overlap(False)
            whole = ctx.render()
            assert whole["stats"].n_chunks == 1
            ctx.set_scratch_limit(int(p.get("ray_max_steps")) * 16384)
            ctx.set_overlap(overlap)
            split = ctx.render()
            assert split["stats"].n_chunks >= 4, split["stats"].n_chunks
            assert split["stats"].arithmetic == whole["stats"].arithmetic == (0 if (mode == "kappa" or tier == "exact") else 1)
            assert gu.same_bits(split["image"], whole["image"]).all(), (mode, tier)
            assert np.array_equal(split["sample_num"], whole["sample_num"]) and np.array_equal(split["sample_flags"], whole["sample_flags"])
            assert split["stats"].n_samples == whole["stats"].n_samples and split["stats"].n_gathers == whole["stats"].n_gathers
            if mode == "empty shell":
                assert whole["stats"].n_samples_emitted < 0.9 * whole["stats"].n_samples
    assert np.nanmax(whole["image"]) > 0.0


@pytest.mark.parametrize("case,extra", [
    ("sim_multifreq", AUX_SIM),                                    # three frequencies, every auxiliary image
    ("sim_few_steps", AUX_SIM),                                    # flagged rays: NaN primitives along the whole ray
    ("sim_cuts", dict(AUX_SIM, image_light="false")),              # no image_light: j, alpha read as NaN where unused
    ("sim_spin_fallback", dict(image_lambda_ave="true", image_crossings="true")),   # no coefficients needed at all
    ("sim_rk4", AUX_SIM),
    ("sim_powerlaw", AUX_SIM),                                     # thermal + power-law electrons
    ("sim_powerlaw", dict(plasma_power_frac=1.0)),                 # power-law electrons only (no thermal part, theta_e NaN)
    ("sim_powerlaw", dict(plasma_power_frac=0.5, plasma_p=3.5, plasma_gamma_min=10.0, plasma_gamma_max=1.0e5,
                          image_tau="true", image_light="false")),
    ("formula_dp", AUX_ALL),
    ("formula_absorb", dict(AUX_ALL, image_light="false", image_time="false")),
    ("formula_flat", AUX_ALL),
])
def test_auxiliary_images_against_oracle(case, extra, built_library):
    """Auxiliary images (unpolarized.cpp:113-196) on configurations beyond the golden one: HIP vs the CPU
    oracle (itself bit-exact against the reference on the golden auxiliary case), every row bit-exact."""
    import blacklight_amd as bl
    from blacklight_amd import _capi
    import oracle_api
    fx, params, mock_args = gu.load_case(case)
    params = dict(params)
    params.update(extra)
    p = bl.Params.from_dict(params)
    grid = gu.golden_grid(mock_args) if mock_args is not None else None
    with bl.Context(p) as ctx:
        if grid is not None:
            ctx.set_grid(grid)
        got = ctx.render()
    res = int(p.get("camera_resolution"))
    want = oracle_api.render(p.ptr, grid.desc() if grid is not None else None, _capi.RenderDesc, _capi.CameraFrame,
                             n_rays=res * res, max_steps=int(p.get("ray_max_steps")),
                             n_freq=int(p.get("image_num_frequencies")))
    assert np.array_equal(got["sample_num"], want["sample_num"])
    assert np.array_equal(got["sample_flags"], want["sample_flags"])
    assert got["image"].shape == want["image"].shape
    same = gu.same_bits(got["image"], want["image"])
    bad_rows = sorted(set(np.nonzero(~same)[0].tolist()))
    assert same.all(), f"{(~same).sum()} values differ in rows {bad_rows}"


def test_oracle_agreement_other_configuration(built_library):
    """Seeded variation away from the golden cases: HIP vs CPU oracle, bit-exact."""
    import blacklight_amd as bl
    from blacklight_amd import _capi
    import oracle_api
    fx, params, mock_args = gu.load_case("sim_dp_interp")
    params = dict(params)
    params.update(camera_resolution=40, camera_th=63.0, camera_ph=111.0, camera_rotation=-20.0, simulation_a=0.3,
                  fallback_nan="false", fallback_rho=1.0e-7, fallback_pgas=1.0e-9, image_frequency=8.6e10,
                  ray_step=0.02, camera_width=30.0)
    p = bl.Params.from_dict(params)
    grid = gu.golden_grid(mock_args)
    with bl.Context(p) as ctx:
        ctx.set_grid(grid)
        got = ctx.render()
    desc = grid.desc()
    want = oracle_api.render(p.ptr, desc, _capi.RenderDesc, _capi.CameraFrame, n_rays=40 * 40)
    assert np.array_equal(got["sample_num"], want["sample_num"])
    assert np.array_equal(got["sample_flags"], want["sample_flags"])
    assert gu.same_bits(got["image"], want["image"]).all()
    assert got["stats"].n_gathers == want["n_gathers"]
    assert got["stats"].n_samples == want["n_samples"]


def _random_configuration(seed):
    """A configuration drawn from the supported parameter space, seeded."""
    rng = np.random.default_rng(1000 + seed)
    simulation = bool(rng.integers(0, 4))          # 3 in 4 simulation
    base = "sim_dp_interp" if simulation else "formula_dp"
    over = dict(
        camera_resolution=int(rng.choice([16, 24])),
        camera_type=str(rng.choice(["plane", "pinhole"])),
        camera_r=float(rng.uniform(30.0, 120.0)),
        camera_th=float(rng.uniform(5.0, 175.0)),
        camera_ph=float(rng.uniform(0.0, 360.0)),
        camera_rotation=float(rng.uniform(-90.0, 90.0)),
        camera_urn=float(rng.uniform(-0.2, 0.2)), camera_uthn=float(rng.uniform(-0.1, 0.1)),
        camera_uphn=float(rng.uniform(-0.2, 0.2)),
        camera_width=float(rng.uniform(8.0, 40.0)),
        image_normalization=str(rng.choice(["camera", "infinity"])),
        ray_integrator=str(rng.choice(["dp", "dp", "rk4", "rk2"])),
        ray_step=float(rng.choice([0.01, 0.02, 0.05])),
        ray_terminate=str(rng.choice(["photon", "multiplicative", "additive"])),
        image_num_frequencies=int(rng.choice([1, 1, 3])),
    )
    if over["camera_type"] == "pinhole":
        over["camera_width"] = float(rng.uniform(0.05, 0.4)) * over["camera_r"]
    if over["ray_terminate"] == "multiplicative":
        over["ray_factor"] = float(rng.uniform(1.001, 1.1))
    elif over["ray_terminate"] == "additive":
        over["ray_factor"] = float(rng.uniform(0.01, 0.5))
    if over["image_num_frequencies"] > 1:
        over.update(image_frequency_start=float(10.0 ** rng.uniform(10.5, 11.2)), image_frequency_end=float(10.0 ** rng.uniform(11.3, 12.0)),
                    image_frequency_spacing=str(rng.choice(["lin_freq", "lin_wave", "log"])))
    else:
        over["image_frequency"] = float(10.0 ** rng.uniform(10.5, 12.0))
    spin = float(rng.choice([0.0, 0.3, 0.7, 0.95]))
    if simulation:
        over.update(simulation_a=spin, simulation_interp=str(rng.choice(["true", "false"])),
                    fallback_nan=str(rng.choice(["true", "false"])), fallback_rho=1.0e-6, fallback_pgas=1.0e-8,
                    plasma_rat_low=float(rng.uniform(1.0, 3.0)), plasma_rat_high=float(rng.uniform(5.0, 40.0)),
                    plasma_use_p=str(rng.choice(["true", "false"])),
                    cut_sigma_max=float(rng.choice([-1.0, 1.0, 10.0])), cut_omit_near=str(rng.choice(["true", "false", "false"])),
                    cut_midplane_theta=float(rng.choice([0.0, 0.0, 60.0, -20.0])))
        if rng.integers(0, 3) == 0:
            over.update(plasma_power_frac=float(rng.uniform(0.05, 0.6)), plasma_p=float(rng.uniform(2.2, 3.8)),
                        plasma_gamma_min=float(rng.uniform(1.0, 30.0)), plasma_gamma_max=float(10.0 ** rng.uniform(3.0, 6.0)))
    else:
        over.update(formula_spin=spin, formula_h=float(rng.uniform(0.0, 4.0)), formula_l0=float(rng.uniform(0.0, 1.0)),
                    formula_a=float(rng.choice([0.0, 1.0e3, 1.0e6])), formula_alpha=float(rng.uniform(-1.0, 1.0)),
                    ray_flat=str(rng.choice(["false", "false", "false", "true"])))
        if over["ray_flat"] == "true":
            over["formula_spin"] = 0.0
    if rng.integers(0, 3) == 0:
        over.update(image_tau="true", image_lambda="true", image_crossings="true")
    # mesh layout and electron model (drawn last, so that earlier seeds keep their other parameters)
    mesh = {}
    if simulation:
        layout = int(rng.integers(0, 4))
        if layout == 1:
            mesh["_blocks"] = [2, 2, 2]
        elif layout >= 2:
            mesh["_refined"] = 1
        if layout == 3:
            mesh["_entropy"] = 1
            over.update(plasma_model="code_kappa", simulation_kappa_name="r0", fallback_kappa=2.0e6)
        if int(rng.integers(0, 4)) == 0:   # the same arrays read as a Cartesian Kerr-Schild box, looked at along x
            over.update(simulation_coord="cks", camera_th=float(rng.uniform(75.0, 100.0)), camera_ph=float(rng.uniform(-15.0, 15.0)))
    return base, over, mesh


@pytest.mark.parametrize("seed", list(range(24)) + [205, 294, 312])   # (the last three: found by tools/gpu_fuzz_wide.py - formula mode,
# auxiliary rows, rays that run into ray_max_steps with fallback_nan: NaN coefficients at frequency 0, formula_coefficients.cpp:51-59)
def test_randomised_configurations_against_oracle(seed, built_library):
    """Seeded draws from the supported parameter space (cameras, spins, integrators, termination rules,
    frequency lists, plasma / formula parameters, cuts, power-law electrons, auxiliary images, single-block /
    multi-block / refined meshes, electron entropy): HIP vs the CPU oracle, every output bit-exact."""
    import blacklight_amd as bl
    from blacklight_amd import _capi
    import oracle_api
    base, over, mesh = _random_configuration(seed)
    fx, params, mock_args = gu.load_case(base)
    params = dict(params)
    params.update(over)
    if mock_args is not None:
        mock_args = dict(mock_args, **mesh)
    p = bl.Params.from_dict(params)
    grid = gu.golden_grid(mock_args) if mock_args is not None else None
    with bl.Context(p) as ctx:
        if grid is not None:
            ctx.set_grid(grid)
        got = ctx.render()
    res = int(p.get("camera_resolution"))
    want = oracle_api.render(p.ptr, grid.desc() if grid is not None else None, _capi.RenderDesc, _capi.CameraFrame,
                             n_rays=res * res, max_steps=int(p.get("ray_max_steps")),
                             n_freq=int(p.get("image_num_frequencies")))
    assert np.array_equal(got["sample_num"], want["sample_num"]), over
    assert np.array_equal(got["sample_flags"], want["sample_flags"]), over
    assert got["image"].shape == want["image"].shape
    same = gu.same_bits(got["image"], want["image"])
    assert same.all(), f"{(~same).sum()} of {same.size} values differ for {over}"


@pytest.mark.parametrize("seed", range(8))
def test_empty_shell_steps_leave_no_records(seed, built_library, monkeypatch):
    """Camera outside the grid, fallback values beyond it: the geodesic kernel records nothing of the steps that lie between the
    grid's outer edge and the camera's sphere (BlTraceArgs::skip_low) - off the grid, no field, nothing added
    (simulation_sampling.cpp:352-394, simulation_coefficients.cpp:394). Same image, counts and flags as the oracle and as the
    same library recording every step, in both tiers; integrators, spins, cameras, meshes and frequency lists drawn."""
    import blacklight_amd as bl
    from blacklight_amd import _capi
    import oracle_api
    rng = np.random.default_rng(4400 + seed)
    fx, params, mock_args = gu.load_case("sim_dp_interp")
    over = dict(camera_resolution=20, camera_r=float(rng.uniform(60.0, 400.0)), camera_th=float(rng.uniform(5.0, 175.0)),
                camera_ph=float(rng.uniform(0.0, 360.0)), camera_type=str(rng.choice(["plane", "pinhole"])),
                camera_width=float(rng.uniform(20.0, 130.0)), simulation_a=float(rng.choice([0.0, 0.5, 0.95])),
                ray_integrator=str(rng.choice(["dp", "dp", "rk4", "rk2"])), ray_step=float(rng.choice([0.01, 0.03])),
                ray_terminate=str(rng.choice(["photon", "multiplicative"])), simulation_interp=str(rng.choice(["true", "false"])),
                fallback_nan="false", fallback_rho=1.0e-6, fallback_pgas=1.0e-8, image_num_frequencies=int(rng.choice([1, 1, 5])))
    if over["camera_type"] == "pinhole":
        over["camera_width"] = float(rng.uniform(0.2, 0.8)) * over["camera_r"]
    if over["image_num_frequencies"] > 1:
        over.update(image_frequency_start=1.0e11, image_frequency_end=9.0e11, image_frequency_spacing="log")
    mesh = [{}, dict(_blocks=[2, 2, 2]), dict(_refined=1)][seed % 3]
    params = dict(params, **over)
    p = bl.Params.from_dict(params)
    grid = gu.golden_grid(dict(mock_args, **mesh))
    out = {}
    with bl.Context(p) as ctx:
        ctx.set_grid(grid)
        for tier in ("exact", "tolerant"):
            ctx.set_arithmetic(tier)
            out[tier] = ctx.render()
            ctx.debug_set_switches("RECORD_EVERY_STEP")
            out[tier + " all"] = ctx.render()
            ctx.debug_set_switches()
    res = 20
    want = oracle_api.render(p.ptr, grid.desc(), _capi.RenderDesc, _capi.CameraFrame, n_rays=res * res, max_steps=int(p.get("ray_max_steps")),
                             n_freq=int(p.get("image_num_frequencies")))
    for tier in ("exact", "tolerant"):
        got, every = out[tier], out[tier + " all"]
        assert np.array_equal(got["sample_num"], want["sample_num"]) and np.array_equal(got["sample_flags"], want["sample_flags"]), over
        assert got["stats"].n_samples == want["n_samples"] == every["stats"].n_samples and got["stats"].n_gathers == want["n_gathers"]
        assert got["stats"].max_sample_num == every["stats"].max_sample_num == int(want["sample_num"].max())
        if tier == "exact":
            assert gu.same_bits(got["image"], every["image"]).all(), over     # (the skipped records are identities)
        else:   # the tolerant tier composes a ray's records four at a time (bl_transfer_quad_kernel): without the identities the groups differ
            with np.errstate(invalid="ignore"):
                assert np.nanmax(np.abs(got["image"] - every["image"])) <= 1.0e-13 * np.nanmax(np.abs(every["image"])), over
            assert np.array_equal(np.isnan(got["image"]), np.isnan(every["image"]))
        assert every["stats"].n_samples_emitted >= every["stats"].n_samples
        if over["ray_integrator"] == "dp":
            assert got["stats"].n_samples_emitted < 0.9 * every["stats"].n_samples_emitted, over
        else:   # (the fixed-step steppers have no instantiation that skips the shell since round 6: every step is recorded)
            # (record slots are handed out in blocks of 1 024 per wave: equal up to the blocks' slack)
            assert abs(got["stats"].n_samples_emitted - every["stats"].n_samples_emitted) <= 8 * 1024, over
    assert gu.same_bits(out["exact"]["image"], want["image"]).all(), over
    assert np.nanmax(want["image"]) > 0.0


@pytest.mark.parametrize("seed", range(6))
def test_unpolarized_kappa_electrons_under_the_opt_in_policy(seed, built_library):
    """plasma_kappa_frac != 0 without polarization: the reference's absorptivity reads kappa_aa_high_i (simulation_coefficients.
    cpp:652), set for polarized runs only (:108-121) - no image of its to compare with. Refused by default; under
    bl_set_undefined_policy(BL_UNDEFINED_KAPPA) rendered with the polarized definition and a warning, bit-identical to the
    oracle under the same definition - plain images, auxiliary rows and frequency lists (the exact tier's per-frequency
    kernel), with thermal and power-law electrons beside them."""
    import blacklight_amd as bl
    from blacklight_amd import _capi
    import oracle_api
    rng = np.random.default_rng(8800 + seed)
    fx, params, mock_args = gu.load_case("sim_dp_interp")
    over = dict(camera_resolution=16, camera_th=float(rng.uniform(20.0, 160.0)), camera_ph=float(rng.uniform(0.0, 360.0)),
                simulation_a=float(rng.choice([0.0, 0.6])), plasma_kappa_frac=float(rng.uniform(0.1, 0.7)),
                plasma_kappa=float(rng.choice([3.5, 4.2, 5.0, 7.0])), plasma_w=float(rng.uniform(5.0, 40.0)),
                image_num_frequencies=int(rng.choice([1, 5])), simulation_interp=str(rng.choice(["true", "false"])))
    if seed % 3 == 1:
        over.update(plasma_power_frac=0.2, plasma_p=3.0, plasma_gamma_min=1.0, plasma_gamma_max=1000.0)
    if seed % 3 == 2:
        over.update(image_tau="true", image_emission="true", image_tau_int="true")
    if over["image_num_frequencies"] > 1:
        over.update(image_frequency_start=1.0e11, image_frequency_end=8.0e11, image_frequency_spacing="log")
    params = dict(params, **over)
    p = bl.Params.from_dict(params)
    grid = gu.golden_grid(mock_args)
    with bl.Context(p) as ctx:
        ctx.set_grid(grid)
        with pytest.raises(bl.BlacklightError, match="BL_UNDEFINED_KAPPA"):
            ctx.render()
        ctx.set_undefined_policy("kappa")
        ctx.clear_warnings()
        got = ctx.render()
        assert ctx.warnings.count("kappa_aa_high_i") == 1 and "(3 / kappa)^4.75 + 0.6" in ctx.warnings
        ctx.set_arithmetic("tolerant")
        again = ctx.render()
        assert again["stats"].arithmetic == 0 and gu.same_bits(again["image"], got["image"]).all()
    want = oracle_api.render(p.ptr, grid.desc(), _capi.RenderDesc, _capi.CameraFrame, n_rays=256, max_steps=int(p.get("ray_max_steps")),
                             n_freq=over["image_num_frequencies"], define_kappa=True)
    with pytest.raises(RuntimeError, match="uninitialised kappa_aa_high_i"):
        oracle_api.render(p.ptr, grid.desc(), _capi.RenderDesc, _capi.CameraFrame, n_rays=256, max_steps=int(p.get("ray_max_steps")),
                          n_freq=over["image_num_frequencies"])
    assert np.array_equal(got["sample_num"], want["sample_num"])
    assert got["image"].shape == want["image"].shape
    same = gu.same_bits(got["image"], want["image"])
    assert same.all(), f"{(~same).sum()} of {same.size} values differ for {over}"
    # the electrons are seen: the same run without them gives another image
    with bl.Context(bl.Params.from_dict(dict(params, plasma_kappa_frac=0.0))) as ctx:
        ctx.set_grid(grid)
        assert not gu.same_bits(ctx.render()["image"], got["image"]).all()


@pytest.mark.parametrize("seed", range(10))
def test_randomised_polarized_configurations_against_oracle(seed, built_library):
    """Seeded draws of polarized runs (joint coupling / rotation split, spins, one or two frequencies, nearest or
    trilinear sampling, thermal + power-law + kappa-distribution electron mixes, kappa on and between the fitted
    values): every Stokes row and every auxiliary row, HIP vs the CPU oracle, bit-exact."""
    import blacklight_amd as bl
    from blacklight_amd import _capi
    import oracle_api
    rng = np.random.default_rng(5000 + seed)
    fx, params, mock_args = gu.load_case("sim_polarized")
    over = dict(camera_resolution=12, image_rotation_split=str(rng.choice(["true", "false"])),
                simulation_a=float(rng.choice([0.0, 0.5, 0.9])), simulation_interp=str(rng.choice(["true", "false"])),
                camera_th=float(rng.uniform(20.0, 160.0)), image_tau=str(rng.choice(["true", "false"])),
                image_num_frequencies=int(rng.choice([1, 2])))
    if over["image_num_frequencies"] > 1:
        over.update(image_frequency_start=float(10.0 ** rng.uniform(10.5, 11.2)), image_frequency_end=float(10.0 ** rng.uniform(11.3, 12.0)),
                    image_frequency_spacing="log")
    else:
        over["image_frequency"] = float(10.0 ** rng.uniform(10.8, 11.8))
    if rng.integers(0, 2) == 0:
        over.update(plasma_power_frac=float(rng.uniform(0.05, 0.4)), plasma_p=float(rng.uniform(2.2, 3.8)),
                    plasma_gamma_min=float(rng.uniform(1.5, 30.0)), plasma_gamma_max=float(10.0 ** rng.uniform(3.0, 6.0)))
    if seed % 5 != 4:
        over.update(plasma_kappa_frac=float(rng.uniform(0.05, 0.5)), plasma_w=float(rng.uniform(1.0, 3.0)),
                    plasma_kappa=float(rng.choice([3.5, 4.0, 4.5, 5.0, float(rng.uniform(3.5, 5.0))])))
    params = dict(params)
    params.update(over)
    p = bl.Params.from_dict(params)
    grid = gu.golden_grid(mock_args)
    with bl.Context(p) as ctx:
        ctx.set_grid(grid)
        got = ctx.render()
    want = oracle_api.render(p.ptr, grid.desc(), _capi.RenderDesc, _capi.CameraFrame, n_rays=144,
                             max_steps=int(p.get("ray_max_steps")), n_freq=int(p.get("image_num_frequencies")))
    assert np.array_equal(got["sample_num"], want["sample_num"]), over
    assert got["image"].shape == want["image"].shape
    same = gu.same_bits(got["image"], want["image"])
    assert same.all(), f"{(~same).sum()} of {same.size} values differ for {over}"


@pytest.mark.parametrize("seed", range(6))
def test_polarized_runs_over_a_refined_mesh(seed, built_library):
    """The polarized coefficient kernel's locate step over the two-level mesh (and the same mesh in smaller blocks): box descriptors and
    row chunks in LDS, cells confirmed exactly, a walk where the guesses do not hold. Exact tier: every Stokes row equal to the CPU
    oracle's and to the path through the locate kernel, bit for bit; the tolerant tier (transport matrices) within its tolerance."""
    import blacklight_amd as bl
    from blacklight_amd import _capi
    import oracle_api
    rng = np.random.default_rng(8800 + seed)
    fx, params, mock_args = gu.load_case("sim_polarized")
    over = dict(camera_resolution=14, simulation_a=float(rng.choice([0.0, 0.0, 0.6])), simulation_interp="true", image_tau=str(rng.choice(["true", "false"])),
                camera_th=float(rng.uniform(15.0, 165.0)), camera_ph=float(rng.uniform(0.0, 360.0)), image_rotation_split=str(rng.choice(["true", "false"])))
    if seed % 3 == 2:
        over.update(plasma_power_frac=float(rng.uniform(0.05, 0.4)), plasma_p=2.8, plasma_gamma_min=2.0, plasma_gamma_max=1.0e4)
    params = dict(params, **over)
    p = bl.Params.from_dict(params)
    grid = gu.golden_grid(dict(mock_args, _refined=1))
    if seed % 2 == 1:
        grid = gu.subdivide_blocks(grid, (2, 3, 2))
    with bl.Context(p) as ctx:
        ctx.set_grid(grid)
        ctx.set_arithmetic("exact")
        inside = ctx.render()
        ctx.debug_set_switches("NO_FUSED_LOCATE")
        outside = ctx.render()
        ctx.debug_set_switches()
        ctx.set_arithmetic("tolerant")
        tolerant = ctx.render()
    assert inside["stats"].fused_variant == 4 and inside["stats"].launches_locate == 0
    assert outside["stats"].fused_variant == 0 and outside["stats"].launches_locate == 1
    want = oracle_api.render(p.ptr, grid.desc(), _capi.RenderDesc, _capi.CameraFrame, n_rays=14 * 14, max_steps=int(p.get("ray_max_steps")))
    assert np.array_equal(inside["sample_num"], want["sample_num"]) and inside["stats"].n_gathers == outside["stats"].n_gathers, over
    if not (inside["sample_flags"] != 0).any():   # (a ray that runs into ray_max_steps is NaN unsampled in the reference; the device reads the grid for it)
        assert inside["stats"].n_gathers == want["n_gathers"], over
    assert gu.same_bits(inside["image"], outside["image"]).all(), over
    assert gu.same_bits(inside["image"], want["image"]).all(), over
    assert tolerant["stats"].fused_variant == 4 and np.array_equal(tolerant["sample_num"], want["sample_num"])
    scale = np.nanmax(np.abs(inside["image"]), axis=-1, keepdims=True)
    with np.errstate(invalid="ignore"):
        assert np.nanmax(np.abs(tolerant["image"] - inside["image"]) / np.where(scale > 0, scale, 1.0)) < 1.0e-9, over


@pytest.mark.parametrize("seed", range(8))
def test_block_interpolation_against_oracle(seed, built_library):
    """simulation_block_interp = true (FindNearbyInds / InterpolateAdvanced) on equal blocks and on the two-level mesh,
    narrow cameras whose rays are all captured so that the far-side block kept last in the list is never reached
    (tests/golden_util.move_last): anchors across block faces, refinement levels and the periodic seam, GPU vs the
    CPU oracle, bit-exact; auxiliary rows, an entropy variable and polarized transfer among the draws."""
    import blacklight_amd as bl
    from blacklight_amd import _capi
    import oracle_api
    rng = np.random.default_rng(9000 + seed)
    fx, params, mock_args = gu.load_case("sim_dp_interp")
    layouts = [dict(_blocks=[2, 2, 4], _last=[1, 1, 2]), dict(_blocks=[4, 2, 2], _last=[3, 1, 1]), dict(_blocks=[2, 3, 8], _last=[1, 2, 4]),
               dict(_refined=1, _last=[0, 1, 1, 1]), dict(_refined=1, _entropy=1, _last=[0, 1, 1, 1])]
    mesh = layouts[seed % len(layouts)]
    over = dict(camera_resolution=12, simulation_block_interp="true", simulation_interp="true",
                camera_th=float(rng.uniform(35.0, 80.0)), camera_ph=float(rng.uniform(-20.0, 20.0)),
                camera_width=float(rng.uniform(1.5, 4.5)), image_tau=str(rng.choice(["true", "false"])),
                fallback_nan=str(rng.choice(["true", "false"])), fallback_rho=1.0e-6, fallback_pgas=1.0e-8, fallback_kappa=2.0e6)
    if "_entropy" in mesh:
        over.update(plasma_model="code_kappa", simulation_kappa_name="r0")
    if seed % 4 == 3:
        over.update(image_polarization="true", image_rotation_split="false")
    params = dict(params)
    params.update(over)
    p = bl.Params.from_dict(params)
    grid = gu.golden_grid(dict(mock_args, **mesh))
    with bl.Context(p) as ctx:
        ctx.set_grid(grid)
        got = ctx.render()
    want = oracle_api.render(p.ptr, grid.desc(), _capi.RenderDesc, _capi.CameraFrame, n_rays=144,
                             max_steps=int(p.get("ray_max_steps")), n_freq=1)
    assert np.array_equal(got["sample_num"], want["sample_num"]), over
    assert got["image"].shape == want["image"].shape
    same = gu.same_bits(got["image"], want["image"])
    assert same.all(), f"{(~same).sum()} of {same.size} values differ for {over} on {mesh}"


def test_block_interpolation_refuses_undefined_reads(built_library):
    """A camera that sees the whole grid reaches the upper edges of the last MeshBlock, where the reference reads past
    its cell-centre arrays: refused (never approximated). Without the MeshBlock table bl_set_grid fails."""
    import dataclasses
    import blacklight_amd as bl
    fx, params, mock_args = gu.load_case("sim_blockinterp")
    params = dict(params, camera_width=24.0)
    p = bl.Params.from_dict(params)
    grid = gu.golden_grid(mock_args)
    with bl.Context(p) as ctx:
        ctx.set_grid(grid)
        with pytest.raises(bl.BlacklightError, match="last MeshBlock"):
            ctx.render()
        ctx.set_undefined_policy("edge")   # usable for a camera that sees the whole grid: edge rule + a warning
        out = ctx.render()
        assert np.isfinite(out["image"]).all() and "where the reference reads past its arrays" in ctx.warnings
        ctx.set_undefined_policy("refuse")
        with pytest.raises(bl.BlacklightError, match="MeshBlock table"):
            ctx.set_grid(dataclasses.replace(grid, levels=None, locations=None, n_3_root=0))


def test_refined_mesh_holes_and_overlaps(built_library):
    """A mesh with a block missing is still a mesh (samples in the hole are off the grid, as in the reference's
    scan over blocks): HIP vs oracle bit-exact. Blocks that overlap are refused."""
    import dataclasses
    import blacklight_amd as bl
    from blacklight_amd import _capi
    import oracle_api
    fx, params, mock_args = gu.load_case("sim_refined")
    params = dict(params, camera_resolution=16, fallback_nan="false", fallback_rho=1.0e-6, fallback_pgas=1.0e-8)
    p = bl.Params.from_dict(params)
    full = gu.golden_grid(mock_args)

    def subset(keep):
        return dataclasses.replace(full, prim=np.ascontiguousarray(full.prim[:, keep]),
                                   **{n: np.ascontiguousarray(getattr(full, n)[keep]) for n in ("x1f", "x2f", "x3f", "x1v", "x2v", "x3v")})

    n_b = full.prim.shape[1]
    holed = subset([b for b in range(n_b) if b not in (3, 17)])
    with bl.Context(p) as ctx:
        ctx.set_grid(holed)
        got = ctx.render()
        want = oracle_api.render(p.ptr, holed.desc(), _capi.RenderDesc, _capi.CameraFrame, n_rays=256,
                                 max_steps=int(p.get("ray_max_steps")), n_freq=1)
        assert np.array_equal(got["sample_num"], want["sample_num"])
        assert gu.same_bits(got["image"], want["image"]).all()
        with pytest.raises(bl.BlacklightError, match="overlap"):
            ctx.set_grid(subset(list(range(n_b)) + [5]))


def test_contexts_give_their_memory_back(built_library):
    """bl_free releases everything a context allocated (records, grids, block tables, polarized scratch, redo lists): device
    memory after a series of contexts - plain, polarized + tolerant, refined mesh with inter-block interpolation - is back
    where it was (DeviceBuffer owns its allocation)."""
    import ctypes
    import blacklight_amd as bl
    hip = ctypes.CDLL("libamdhip64.so")

    def free_bytes():
        free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipDeviceSynchronize() == 0 and hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value

    def run(case, arithmetic):
        fx, params, mock_args = gu.load_case(case)
        with bl.Context(bl.Params.from_dict(params)) as ctx:
            if mock_args is not None:
                ctx.set_grid(gu.golden_grid(mock_args))
            ctx.set_arithmetic(arithmetic)
            ctx.render()

    run("sim_dp_interp", "exact")            # first use: the runtime's own one-off allocations
    before = free_bytes()
    for case, arithmetic in (("sim_dp_interp", "tolerant"), ("sim_polarized", "tolerant"), ("sim_polarized", "exact"),
                             ("sim_blockinterp_refined", "exact"), ("formula_dp", "exact")):
        run(case, arithmetic)
    after = free_bytes()
    assert abs(before - after) <= 64 << 20, (before, after)
