"""BASELINE.json configuration 5: the multi-frequency render of input/example_true_color.input (simulation mode,
frequencies equally spaced in wavelength between 1.5e11 and 3.3e11 Hz, camera.cpp:30-50; the per-frequency loop of
simulation_coefficients.cpp:458). The reference's own ten-frequency golden (tests/golden/sim_true_color.npz) runs with
the other goldens in test_gpu_parity.py; here: the 64-frequency regime (1 KiB of transfer records per sample, one lane
per (ray, frequency) in the transfer kernel) against the CPU oracle, and at the configuration's full width - 64
frequencies over the 256^3 grid at 1024^2 - the size-independent split property."""
import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


def _true_color(n_freq, **extra):
    fx, params, mock_args = gu.load_case("sim_true_color")
    params = dict(params, image_num_frequencies=n_freq)
    params.update(extra)
    return params, mock_args


@pytest.mark.parametrize("n_freq,res,extra", [
    (64, 16, {}),
    (64, 12, dict(simulation_a=0.5, image_normalization="camera", camera_urn=-0.05)),
    (33, 16, dict(simulation_interp="false", image_frequency_spacing="lin_freq")),
    # thermal + power-law electrons: the extended instantiation of the per-frequency kernel (bl_coefficients_freq_kernel<true>)
    (12, 14, dict(plasma_power_frac=0.3, plasma_p=2.5, plasma_gamma_min=1.0, plasma_gamma_max=1000.0)),
])
def test_many_frequencies_against_oracle(n_freq, res, extra, built_library):
    import blacklight_amd as bl
    from blacklight_amd import _capi
    import oracle_api
    params, mock_args = _true_color(n_freq, camera_resolution=res, **extra)
    p = bl.Params.from_dict(params)
    grid = gu.golden_grid(mock_args)
    with bl.Context(p) as ctx:
        ctx.set_grid(grid)
        got = ctx.render()
        freqs = ctx.frequencies
    want = oracle_api.render(p.ptr, grid.desc(), _capi.RenderDesc, _capi.CameraFrame, n_rays=res * res,
                             max_steps=int(p.get("ray_max_steps")), n_freq=n_freq)
    assert got["image"].shape == (n_freq, res * res) == want["image"].shape
    assert np.array_equal(freqs, want["frequencies"])
    if "image_frequency_spacing" not in extra:   # lin_wave: equal steps in wavelength, descending (camera.cpp:39-45)
        wavelength = 1.0 / freqs
        assert np.allclose(np.diff(wavelength), (1.0 / 3.3e11 - 1.0 / 1.5e11) / (n_freq - 1), rtol=1e-12)
    assert np.array_equal(got["sample_num"], want["sample_num"])
    assert np.array_equal(got["sample_flags"], want["sample_flags"])
    same = gu.same_bits(got["image"], want["image"])
    assert same.all(), f"{(~same).sum()} of {same.size} values differ"


def test_true_color_at_size_is_independent_of_how_it_is_split(built_library):
    """1024^2 x 64 frequencies over the 256^3 mock (a quarter of configuration 5's 4096^2 pixels - one GPU's share of
    the frame twice over): the frame in one call (several chunks of the default scratch budget in the exact tier), the frame assembled from
    the tiles of eight emulated ranks, and a window of it in many small chunks agree bit for bit; row l of the 64-row
    image equals the single-frequency render at frequency l for a sample of rows."""
    import blacklight_amd as bl
    from blacklight_amd import distributed as bd, mock
    import bench
    n_freq, res, world = 64, 1024, 8
    params = dict(bench.WORKLOAD, image_num_frequencies=n_freq, image_frequency_start=1.5e11, image_frequency_end=3.3e11,
                  image_frequency_spacing="lin_wave")
    params.pop("image_frequency", None)
    grid = mock.generate(n_r=256, n_th=256, n_ph=256)
    p = bl.Params.from_dict(params)
    with bl.Context(p) as ctx:
        ctx.set_grid(grid)
        full = ctx.render()
        freqs = ctx.frequencies
        assert full["image"].shape == (n_freq, res * res)
        assert full["stats"].n_chunks > 1           # 1 KiB of transfer records per sample
        # the tolerant tier on the same frame: per-sample factors and one lane per (ray, frequency), no transfer records
        # (docs/notebook.md section 5f) - every row within north_star's tolerance of the exact tier's, integer results identical
        ctx.set_arithmetic("tolerant")
        tolerant = ctx.render()
        ctx.set_arithmetic("exact")
        assert tolerant["stats"].arithmetic == 1 and tolerant["stats"].n_chunks < full["stats"].n_chunks
        assert np.array_equal(tolerant["sample_num"], full["sample_num"])
        assert np.array_equal(np.isnan(tolerant["image"]), np.isnan(full["image"]))
        scale = np.nanmax(np.abs(full["image"]), axis=1, keepdims=True)
        worst = float(np.nanmax(np.abs(tolerant["image"] - full["image"]) / scale))
        print(f"1024^2 x 64 frequencies, tolerant vs exact: {worst:.2e} of each row's peak")
        assert worst < 1.0e-6
        del tolerant
        assembled = np.empty_like(full["image"])
        counts = np.full(res * res, -1, dtype=np.int32)
        for rank in range(world):
            pixels = bd.tile_pixels(res, rank, world, bench.TILE)
            part = ctx.render(pixel_map=pixels)
            assembled[:, pixels] = part["image"]
            counts[pixels] = part["sample_num"]
        assert np.array_equal(counts, full["sample_num"])
        same = gu.same_bits(assembled, full["image"])
        assert same.all(), f"{(~same).sum()} values differ between the tiled and the plain frame"
        del assembled
        # a 64 x 64 window through the photon ring in ~30 chunks
        iv, iu = np.mgrid[480:544, 400:464]
        window = (iv * res + iu).reshape(-1).astype(np.int32)
        ctx.set_scratch_limit(256 << 20)
        chunked = ctx.render(pixel_map=window)
        assert chunked["stats"].n_chunks > 10
        assert gu.same_bits(chunked["image"], full["image"][:, window]).all()
        assert np.array_equal(chunked["sample_num"], full["sample_num"][window])
    # single-frequency renders of the same window reproduce their rows
    for l in (0, 31, 63):
        single = dict(bench.WORKLOAD, image_frequency=float(freqs[l]))
        with bl.Context(bl.Params.from_dict(single)) as ctx:
            ctx.set_grid(grid)
            one = ctx.render(pixel_map=window)
        assert gu.same_bits(one["image"][0], full["image"][l, window]).all(), l
    nan = np.isnan(full["image"])
    assert not nan[:, full["sample_flags"] == 0].any()


def test_many_frequency_frames_download_chunk_by_chunk_with_the_same_bits():
    """Large host outputs in many rows (>= 8 image rows, >= 256 MiB: configuration 5's kind) are traced in pixel order and leave chunk by
    chunk while the next chunk renders (bl_render.hip: RenderJob::raster, DownloadChunk), into pageable or pinned memory
    (Context.pinned_array / bl_host_alloc). The frame must be the one a render into device memory gives - tile order, one download:
    bit for bit in the exact tier; in the tolerant tier to rounding (its many-frequency kernel is reproducible from run to run to the
    last bit but one or two pixels in a million, whichever way the frame is produced: DESIGN.md, open points)."""
    import torch
    import bench
    import blacklight_amd as bl
    from blacklight_amd import mock
    res, n_freq = 1024, 32
    params = dict(bench.WORKLOAD, camera_resolution=res, image_num_frequencies=n_freq, image_frequency_start=1.5e11, image_frequency_end=3.3e11,
                  image_frequency_spacing="lin_wave")
    params.pop("image_frequency", None)
    grid = mock.generate(n_r=64, n_th=64, n_ph=64)
    n = res * res
    with bl.Context(bl.Params.from_dict(params)) as ctx:
        ctx.set_geodesic_reuse(False)
        ctx.set_grid(grid)
        pinned = dict(image=ctx.pinned_array((n_freq, n)), sample_num=ctx.pinned_array(n, np.int32), sample_flags=ctx.pinned_array(n, np.uint8))
        for tier, limit in (("exact", 48 << 30), ("tolerant", 24 << 30)):
            ctx.set_arithmetic(tier)
            ctx.set_scratch_limit(200 << 30)
            on_device = torch.zeros((n_freq, n), dtype=torch.float64, device="cuda")
            num_device = torch.zeros(n, dtype=torch.int32, device="cuda")
            st = ctx.render_device(on_device.data_ptr(), n, sample_num_ptr=num_device.data_ptr())
            torch.cuda.synchronize()
            want, want_num = on_device.cpu().numpy(), num_device.cpu().numpy()
            whole = ctx.render()                      # pageable memory, the download after the last chunk
            ctx.set_scratch_limit(limit)
            chunked = ctx.render()                    # several chunks, each downloaded as it finishes, pageable memory
            assert chunked["stats"].n_chunks >= 3 and chunked["stats"].n_chunks > whole["stats"].n_chunks, (chunked["stats"].n_chunks, whole["stats"].n_chunks)
            pinned["image"][:] = -1.0
            into_pinned = ctx.render(out=pinned)      # ... into pinned memory
            assert into_pinned["stats"].n_chunks >= 3 and into_pinned["image"] is pinned["image"]
            for got in (whole, chunked, into_pinned):
                assert np.array_equal(got["sample_num"], want_num)
                if tier == "exact":
                    assert gu.same_bits(got["image"], want).all()
                else:
                    assert np.array_equal(np.isnan(got["image"]), np.isnan(want))
                    with np.errstate(invalid="ignore"):
                        assert np.nanmax(np.abs(got["image"] - want) / np.abs(want)) < 1.0e-14
