"""The adaptive refinement decision (host C++ behind bl_adaptive_refine; reference radiation_adaptive.cpp:19-321, camera.cpp:445-458)
over random images against a numpy restatement of the reference (tests/refine_restatement.py): every criterion on and off, block
sizes, levels, regions, several frequencies and polarized row layouts, NaN / infinite / zero / negative pixels, blocks without a
finite pixel. The reference's own decisions pin both on the goldens (test_host_steps.py); this covers the parameter space."""
import numpy as np
import pytest

import golden_util as gu
import refine_restatement as rr

BL_DEVICE_NONE = -2


def _draw(seed):
    rng = np.random.default_rng(31000 + seed)
    polarized = bool(rng.integers(0, 4) == 0)
    fx, params, _ = gu.load_case("sim_polarized_adaptive" if polarized else "sim_adaptive")
    bs = int(rng.choice([4, 8]))
    res = bs * int(rng.choice([2, 3, 4]))
    n_nu = int(rng.choice([1, 1, 3]))
    over = dict(camera_resolution=res, adaptive_block_size=bs, adaptive_max_level=int(rng.integers(1, 4)), camera_width=float(rng.uniform(8.0, 40.0)),
                image_num_frequencies=n_nu, adaptive_frequency_num=int(rng.integers(1, n_nu + 1)))
    if n_nu > 1:
        over.update(image_frequency_start=1.0e11, image_frequency_end=4.0e11, image_frequency_spacing="log")
    for name, scale in (("val", 1.0), ("abs_grad", 0.5), ("rel_grad", 1.0), ("abs_lapl", 0.5), ("rel_lapl", 2.0)):
        over[f"adaptive_{name}_cut"] = float(rng.uniform(0.0, scale))
        over[f"adaptive_{name}_frac"] = float(rng.choice([-1.0, -1.0, 0.0, 0.1, 0.25, 0.5, 0.9]))
    regions = int(rng.choice([0, 0, 1, 2]))
    for key in [k for k in params if k.startswith("adaptive_region_")]:
        params.pop(key)
    over["adaptive_num_regions"] = regions
    half = over["camera_width"] / 2.0
    for r in range(1, regions + 1):
        x0, x1 = sorted(rng.uniform(-half, half, 2))
        y0, y1 = sorted(rng.uniform(-half, half, 2))
        over.update({f"adaptive_region_{r}_level": int(rng.integers(0, 3)), f"adaptive_region_{r}_x_min": float(x0), f"adaptive_region_{r}_x_max": float(x1),
                     f"adaptive_region_{r}_y_min": float(y0), f"adaptive_region_{r}_y_max": float(y1)})
    return dict(params, **over), polarized, rng


def _image(rng, rows, pixels):
    kind = int(rng.integers(0, 4))
    smooth = np.abs(np.sin(np.linspace(0.0, rng.uniform(1.0, 30.0), pixels) + rng.uniform(0.0, 6.0))) * rng.uniform(0.1, 2.0)
    image = np.stack([smooth * rng.uniform(0.2, 2.0) + rng.uniform(0.0, 0.3) * rng.random(pixels) for _ in range(rows)])
    if kind == 1:
        image[:, rng.random(pixels) < 0.2] = np.nan
    elif kind == 2:
        image[:, rng.random(pixels) < 0.1] = 0.0
        image[:, rng.random(pixels) < 0.05] = np.inf
    elif kind == 3:
        image = image - 0.7   # negative values: the relative criteria divide by sums that pass through zero
        image[:, rng.random(pixels) < 0.5] = np.nan
    return image


@pytest.mark.filterwarnings("ignore::RuntimeWarning")   # (NaN and infinite pixels are part of the draw)
@pytest.mark.parametrize("seed", range(160))
def test_refinement_decision_over_random_images(seed, built_library):
    import blacklight_amd as bl
    params, polarized, rng = _draw(seed)
    p = bl.Params.from_dict(params)
    res, bs = int(params["camera_resolution"]), int(params["adaptive_block_size"])
    with bl.Context(p, device=BL_DEVICE_NONE) as ctx:
        rows = ctx.num_quantities
        image, locs = _image(rng, rows, res * res), None
        for level in range(int(params["adaptive_max_level"]) + 1):
            flags, nxt = ctx.adaptive_refine(level, image, locs)
            want_flags, want_nxt = rr.check_refinement(params, level, image, locs, polarized)
            assert np.array_equal(flags, want_flags), (seed, level)
            assert np.array_equal(nxt, want_nxt), (seed, level)
            if nxt.shape[0] == 0:
                break
            locs = nxt
            image = _image(rng, rows, nxt.shape[0] * bs * bs)
