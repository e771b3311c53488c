"""What the library does when nobody sets a switch: the fast bit-identical choices the host can make are the
defaults, the choices that trade a property away are API calls whose effect bl_stats reports.

  * bl_set_reproducible: the tolerant tier without composed transfer maps - two renders of a frame, and a frame and its tiles, are the
    same bits (the reference is one deterministic loop, blacklight.cpp:196-233, and bit-deterministic across thread counts);
  * bl_set_tail_policy / BL_TAIL_AUTO: formula-mode frames finish their last rays in bl_geodesic_quad_kernel without any switch
    (bl_stats.switches == 0), bit-identical to the wide stepper alone;
  * bl_set_caller_stream: a render ordered behind work on the caller's stream.
"""
import numpy as np
import pytest
import torch   # (before the library: torch brings a HIP runtime of its own, which has to be the one the process loads first)

import golden_util as gu

pytestmark = pytest.mark.gpu


def _benchmark_like(res=128, grid_n=64):
    import bench
    import blacklight_amd as bl
    from blacklight_amd import mock
    params = dict(bench.WORKLOAD, camera_resolution=res)
    return bl.Params.from_dict(params), mock.generate(n_r=grid_n, n_th=grid_n, n_ph=grid_n)


def test_tolerant_renders_are_bit_equal_under_set_reproducible():
    import blacklight_amd as bl
    from blacklight_amd import distributed as bd
    p, grid = _benchmark_like()
    with bl.Context(p) as ctx:
        ctx.set_grid(grid)
        ctx.set_arithmetic("tolerant")
        composed = ctx.render()
        assert composed["stats"].fused_variant == 2 and composed["stats"].composed_maps == 1 and composed["stats"].switches == 0
        ctx.set_reproducible(True)
        first = ctx.render()
        assert first["stats"].fused_variant == 2 and first["stats"].composed_maps == 0 and first["stats"].switches == 0
        for _ in range(3):
            again = ctx.render()
            assert gu.same_bits(again["image"], first["image"]).all()
            assert np.array_equal(again["sample_num"], first["sample_num"])
        # a frame and its tiles (three ranks' shares, shares of different lengths)
        res = 128
        tiled = np.empty_like(first["image"])
        for rank in range(3):
            pixels = bd.tile_pixels(res, rank, 3, 32)
            share = ctx.render(pixel_map=pixels)
            tiled[:, pixels] = share["image"]
        assert gu.same_bits(tiled, first["image"]).all()
        # ... and the composed maps differ from it by rounding only
        with np.errstate(invalid="ignore"):
            assert np.nanmax(np.abs(composed["image"] - first["image"])) <= 1.0e-13 * np.nanmax(np.abs(first["image"]))
        ctx.set_reproducible(False)
        assert ctx.render()["stats"].composed_maps == 1


def test_formula_frames_take_the_quad_tail_by_default():
    import blacklight_amd as bl
    fx, params, _ = gu.load_case("formula_dp")
    p = bl.Params.from_dict(dict(params, camera_resolution=96))
    with bl.Context(p) as ctx:
        for tier in ("exact", "tolerant"):
            ctx.set_arithmetic(tier)
            auto = ctx.render()
            assert auto["stats"].switches == 0 and auto["stats"].tail_policy == 2 and auto["stats"].n_parked > 0, tier
            ctx.set_tail_policy("wide")
            wide = ctx.render()
            assert wide["stats"].tail_policy == 1 and wide["stats"].n_parked == 0
            ctx.set_tail_policy("auto")
            assert np.array_equal(auto["sample_num"], wide["sample_num"]) and np.array_equal(auto["sample_flags"], wide["sample_flags"])
            assert gu.same_bits(auto["image"], wide["image"]).all(), tier
    # small frames and simulation frames stay with the wide stepper
    with bl.Context(bl.Params.from_dict(dict(params, camera_resolution=32))) as ctx:
        assert ctx.render()["stats"].tail_policy == 1
    p, grid = _benchmark_like(96, 32)
    with bl.Context(p) as ctx:
        ctx.set_grid(grid)
        assert ctx.render()["stats"].tail_policy == 1
        ctx.set_tail_policy("quad")
        quad = ctx.render()
        ctx.set_tail_policy("wide")
        assert gu.same_bits(quad["image"], ctx.render()["image"]).all()


def test_render_waits_for_the_callers_stream():
    """A fill queued on a torch stream just before the render must not land after the render's own writes."""
    import blacklight_amd as bl
    torch.cuda.init()
    p, grid = _benchmark_like(64, 32)
    with bl.Context(p) as ctx:
        ctx.set_grid(grid)
        want = ctx.render()
        n = want["sample_num"].size
        side = torch.cuda.Stream()
        image = torch.empty((1, n), dtype=torch.float64, device="cuda")
        sample_num = torch.empty(n, dtype=torch.int32, device="cuda")
        ballast = torch.empty(1 << 28, dtype=torch.float64, device="cuda")   # 2 GiB of fills ahead of the small ones
        with torch.cuda.stream(side):
            for _ in range(4):
                ballast.fill_(1.0)
            image.fill_(-7.0)
            sample_num.fill_(-7)
            ctx.follow_torch_stream()
        ctx.render_device(image.data_ptr(), n, sample_num_ptr=sample_num.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(sample_num.cpu().numpy(), want["sample_num"])
        assert gu.same_bits(image.cpu().numpy(), want["image"]).all()
        del ballast, image, sample_num
        torch.cuda.empty_cache()   # (the suite's large-frame tests size their scratch by what is free)


def test_a_share_of_a_frame_takes_the_split_steppers_by_default():
    """BL_TAIL_AUTO over a simulation grid: a call with at most eight rays per lane of the device (a share of a tiled frame), a plane camera and no spin gives the
    photon ring's rays to the quad stepper on compute units of its own (BL_TAIL_SPLIT) - same bits; with spin, or more rays, it does not."""
    import blacklight_amd as bl
    p, grid = _benchmark_like(256, 32)
    with bl.Context(p) as ctx:
        ctx.set_grid(grid)
        auto = ctx.render()
        assert auto["stats"].tail_policy == 3 and auto["stats"].switches == 0 and 0 < auto["stats"].n_parked < 4096
        ctx.set_tail_policy("wide")
        wide = ctx.render()
        assert wide["stats"].tail_policy == 1 and wide["stats"].n_parked == 0
        assert np.array_equal(auto["sample_num"], wide["sample_num"]) and np.array_equal(auto["sample_flags"], wide["sample_flags"])
        assert gu.same_bits(auto["image"], wide["image"]).all()
        ctx.set_tail_policy("auto")
        # a window of the frame that the ring does not cross: nothing to split
        corner = (np.arange(64)[:, None] * 256 + np.arange(64)[None, :]).reshape(-1).astype(np.int32)
        corner = np.tile(corner, 8)   # (32 768 rays, so that the size alone would qualify)
        assert ctx.render(pixel_map=corner)["stats"].tail_policy == 1
    import bench
    from blacklight_amd import mock
    with bl.Context(bl.Params.from_dict(dict(bench.WORKLOAD, camera_resolution=256, simulation_a=0.5))) as ctx:
        ctx.set_grid(mock.generate(n_r=32, n_th=32, n_ph=32))
        assert ctx.render()["stats"].tail_policy == 1


def test_a_context_starts_in_the_tolerant_tier(monkeypatch):
    """north_star's tolerance is what a caller gets who sets nothing (the default was the slow tier); the exact tier is
    bl_set_arithmetic(BL_ARITH_EXACT) or BLACKLIGHT_AMD_ARITHMETIC=exact, which is how this test session pins it (conftest.py)."""
    import blacklight_amd as bl
    p, grid = _benchmark_like()
    monkeypatch.delenv("BLACKLIGHT_AMD_ARITHMETIC", raising=False)
    with bl.Context(p) as ctx:
        ctx.set_grid(grid)
        fast = ctx.render()
        assert fast["stats"].arithmetic == 1 and fast["stats"].fused_variant == 2
        ctx.set_arithmetic("exact")
        exact = ctx.render()
        assert exact["stats"].arithmetic == 0 and exact["stats"].fused_variant == 3
    assert np.array_equal(fast["sample_num"], exact["sample_num"]) and np.array_equal(np.isnan(fast["image"]), np.isnan(exact["image"]))
    with np.errstate(invalid="ignore"):
        assert np.nanmax(np.abs(fast["image"] - exact["image"])) <= 1.0e-11 * np.nanmax(np.abs(exact["image"]))
    monkeypatch.setenv("BLACKLIGHT_AMD_ARITHMETIC", "exact")
    with bl.Context(p) as ctx:
        ctx.set_grid(grid)
        again = ctx.render()
        assert again["stats"].arithmetic == 0 and gu.same_bits(again["image"], exact["image"]).all()


def test_polarized_frames_build_their_matrices_beside_the_coefficients_and_get_the_same_bits(monkeypatch):
    """Polarized runs in the tolerant tier build the transport matrices on a second stream beside the per-frequency coefficient kernel
    (both read bl_shade_polarized2_kernel's output only; polarized.cpp:150-198 needs the geometry, :387-779 the coefficients): the same
    kernels on the same data as one after the other - BLACKLIGHT_AMD_POLARIZED_OVERLAP=0, read when a context is made -, so the same bits;
    several chunks as well."""
    import blacklight_amd as bl
    import bench
    _, grid = _benchmark_like(res=96)
    params = dict(bench.WORKLOAD, camera_resolution=96, image_polarization=True, image_tau=True)
    images = {}
    for overlap in ("1", "0"):
        monkeypatch.setenv("BLACKLIGHT_AMD_POLARIZED_OVERLAP", overlap)
        with bl.Context(bl.Params.from_dict(params)) as ctx:
            ctx.set_grid(grid)
            ctx.set_arithmetic("tolerant")
            whole = ctx.render()
            assert whole["stats"].arithmetic == 1 and whole["stats"].n_chunks == 1
            ctx.set_scratch_limit(1 << 30)
            chunked = ctx.render()
            assert chunked["stats"].n_chunks > 1
            assert gu.same_bits(chunked["image"], whole["image"]).all()
            images[overlap] = whole["image"]
    assert images["1"].shape[0] >= 5 and np.isfinite(images["1"]).any()
    assert gu.same_bits(images["1"], images["0"]).all()


def test_a_context_made_with_no_environment_starts_in_the_tolerant_tier():
    """the parity suite pins the exact tier through BLACKLIGHT_AMD_ARITHMETIC (tests/conftest.py), so the path a caller gets
    with nothing set is tested here, in a process without the variable: tolerant tier, composed maps, the benchmark's kernel, and the
    stats say so."""
    import os
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import torch, bench, blacklight_amd as bl\n"
        "from blacklight_amd import mock\n"
        "ctx = bl.Context(bl.Params.from_dict(dict(bench.WORKLOAD, camera_resolution=64)))\n"
        "ctx.set_grid(mock.generate(n_r=32, n_th=32, n_ph=32))\n"
        "st = ctx.render()['stats']\n"
        "print('TIER', st.arithmetic, st.composed_maps, st.fused_variant, st.switches, st.geodesics_reused)\n"
        "st = ctx.render()['stats']\n"
        "print('AGAIN', st.arithmetic, st.composed_maps, st.fused_variant, st.switches, st.geodesics_reused)\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("BLACKLIGHT_AMD_")}
    run = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "TIER 1 1 2 0 0" in run.stdout and "AGAIN 1 1 2 0 1" in run.stdout, run.stdout
