"""GPU: slow light (slow_light_on) against the reference's images of the same window of files, bit-exact
(tier B), through the slice calls of the C-ABI a caller with its own reader would use."""
import json
import os

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", gu.SLOW_CASES)
def test_slow_light_images(case, built_library):
    import blacklight_amd as bl
    fx = np.load(os.path.join(gu.GOLDEN_DIR, f"{case}.npz"), allow_pickle=False)
    params = json.loads(str(fx["params"]))
    p = bl.Params.from_dict(params)
    grids = gu.slow_light_grids(fx)
    file_times = [float(t) for t in fx["file_times"]]
    rows = [name for name in ("I_nu", "tau") if f"B_0_{name}" in fx.files]
    warnings = ""
    with bl.Context(p) as ctx:
        warnings += ctx.warnings
        ctx.clear_warnings()
        with pytest.raises(bl.BlacklightError, match="bl_set_grid_slice"):
            ctx.set_grid(grids[0])
        with pytest.raises(bl.BlacklightError, match="before bl_set_grid"):
            ctx.render()
        held = None
        for image, (t_cam, files) in enumerate(gu.slow_light_windows(params, file_times)):
            # what the reader does: keep the slices still in the window, read the new ones (simulation_reader.cpp:281-302)
            new = len(files) if held is None or files[0] - held[0] >= len(files) else files[0] - held[0]
            if 0 < new < len(files):
                ctx.shift_grid_slices(new)
            for n in range(new):
                ctx.set_grid_slice(n, grids[files[n]], file_times[files[n]])
            held = files
            if t_cam > file_times[files[0]]:
                warnings += f"Warning: Snapshot {image} at time {t_cam:g} requires moderate extrapolation.\n"
            ctx.set_snapshot(image)
            out = ctx.render()
            warnings += ctx.warnings
            ctx.clear_warnings()
            want = np.stack([fx[f"B_{image}_{name}"].reshape(-1) for name in rows])
            assert gu.same_bits(out["image"], want).all(), (image, t_cam)
    assert warnings == str(fx["B_warnings"])


def test_significant_extrapolation_is_an_error(built_library):
    """A window that ends long before the camera time: the reference's exception text
    (simulation_sampling.cpp:577-596)."""
    import blacklight_amd as bl
    fx = np.load(os.path.join(gu.GOLDEN_DIR, "slow_nearest.npz"), allow_pickle=False)
    params = dict(json.loads(str(fx["params"])), camera_resolution=8, slow_chunk_size=2)
    p = bl.Params.from_dict(params)
    grids = gu.slow_light_grids(fx)
    with bl.Context(p) as ctx:
        ctx.set_grid_slice(0, grids[1], 20.0)
        ctx.set_grid_slice(1, grids[0], 0.0)
        ctx.set_snapshot(0)   # camera time 170.5
        with pytest.raises(bl.BlacklightError) as info:
            ctx.render()
        text = str(info.value)
        assert text.startswith("Error: Snapshot 0 at time 170.5 requires significant extrapolation forward in time (")
        assert text.endswith(" gravitational times).") and "(64/64 pixels, by up to 150." in text


@pytest.mark.parametrize("n_freq", [1, 5])
@pytest.mark.parametrize("case", gu.SLOW_CASES)
def test_slow_light_with_block_interpolation_against_oracle(case, n_freq, built_library):
    """(With one frequency and with five.) slow_light_on together with simulation_block_interp (simulation_sampling.cpp:960-1033: InterpolateAdvanced per time
    slice, "<= 0 -> first anchor" per slice, then the blend in time): GPU vs the CPU oracle, bit-exact, on the slow-light
    fixtures' snapshots. These are single blocks - the file's last block - so the camera keeps clear of its upper
    half-cells in r and phi (where the reference reads past its centre rows): it sits at r = 40, phi = 180 degrees, and is
    narrow enough for the plunging rays' sweep in phi to stay below the last half-cell."""
    import dataclasses
    import blacklight_amd as bl
    from blacklight_amd import _capi
    import oracle_api
    fx = np.load(os.path.join(gu.GOLDEN_DIR, f"{case}.npz"), allow_pickle=False)
    params = dict(json.loads(str(fx["params"])), simulation_block_interp="true", simulation_interp="true", camera_resolution=10,
                  camera_r=40.0, camera_ph=180.0, camera_width=3.0, slow_num_images=2, simulation_a=0.0)
    if n_freq > 1:   # several frequencies: the exact tier's frequency loop runs as lanes of bl_coefficients_freq_kernel
        params.update(image_num_frequencies=n_freq, image_frequency_start=1.0e11, image_frequency_end=6.0e11, image_frequency_spacing="log")
        params.pop("image_frequency", None)
    p = bl.Params.from_dict(params)
    grids = [gu.single_block_table(g) for g in gu.slow_light_grids(fx)]
    file_times = [float(t) for t in fx["file_times"]]
    with bl.Context(p) as ctx:
        for image, (t_cam, files) in enumerate(gu.slow_light_windows(params, file_times)):
            for n, f in enumerate(files):
                ctx.set_grid_slice(n, grids[f], file_times[f])
            ctx.set_snapshot(image)
            got = ctx.render()
            descs = [grids[f].desc() for f in files]
            want = oracle_api.render(p.ptr, descs[0], _capi.RenderDesc, _capi.CameraFrame, n_rays=100, max_steps=int(params["ray_max_steps"]),
                                     n_freq=n_freq, slow=dict(grids=descs, times=[file_times[f] for f in files], snapshot_time=t_cam))
            assert np.array_equal(got["sample_num"], want["sample_num"])
            assert gu.same_bits(got["image"], want["image"]).all(), (image, t_cam)
            assert np.nanmax(got["image"][0]) > 0.0
            ctx.clear_warnings()
