#!/usr/bin/env python3
"""Randomised sweep of the two arithmetic tiers against each other and against the oracle (a tool, not a test: the seeded cases
of tests/test_gpu_tolerant.py are the part of this that runs every time).

    python3 tools/gpu_fuzz_tiers.py [n_seeds] [first_seed] [oracle_every]

Every seed draws a camera (plane / pinhole, anywhere around the hole, inside or outside the grid), a spin, a mock grid of its own
size (evenly spaced or warped polar / azimuthal faces, one block, split into blocks, or a two-level mesh), sampling mode, temperature model, cuts,
fallback values, frequencies, optionally power-law electrons, a Cartesian reading of the grid, an optical-depth image. It then
checks what the tiers promise: sample counts, flags, NaN masks and S_in identical, intensities within 1e-11 of a row's maximum; and
for every `oracle_every`-th seed the exact tier bit for bit against the CPU oracle. Prints one line per violation and a summary."""
import dataclasses
import json
import os
os.environ.setdefault("BLACKLIGHT_AMD_ARITHMETIC", "exact")   # (a context starts in this tier; the tool names the tolerant one where it wants it)
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import blacklight_amd as bl          # noqa: E402
from blacklight_amd import _capi, mock   # noqa: E402
import golden_util as gu             # noqa: E402
import oracle_api                    # noqa: E402

EXPECTED = 1.0e-11


def distance(a, b):
    scale = np.nanmax(np.abs(b), axis=-1, keepdims=True)
    scale = np.where(scale > 0, scale, 1.0)
    with np.errstate(invalid="ignore"):
        d = np.abs(a - b) / scale
    return float(np.nanmax(d)) if np.isfinite(d).any() else 0.0


def warped(faces, amplitude):
    lo, hi = faces[0, 0], faces[0, -1]
    u = (faces - lo) / (hi - lo)
    out = lo + (hi - lo) * (u + amplitude * np.sin(2.0 * np.pi * u) / (2.0 * np.pi))
    out = out.astype(np.float32).astype(np.float64)
    out[0, 0], out[0, -1] = lo, hi
    return np.ascontiguousarray(out)


def centres(faces):
    return np.ascontiguousarray((0.5 * (faces[:, :-1] + faces[:, 1:])).astype(np.float32).astype(np.float64))


def draw(seed):
    rng = np.random.default_rng(424200 + seed)
    _, params, _ = gu.load_case("sim_dp_interp")
    res = int(rng.choice([16, 24, 33]))
    if os.environ.get("FUZZ_RES"):   # larger frames: many waves, refills of the persistent kernel, record blocks, the reservation gate
        res = int(rng.choice([int(v) for v in os.environ["FUZZ_RES"].split(",")]))
    over = dict(camera_resolution=res, camera_th=float(rng.uniform(3.0, 177.0)), camera_ph=float(rng.uniform(0.0, 360.0)),
                camera_r=float(rng.uniform(25.0, 110.0)), camera_width=float(rng.uniform(6.0, 45.0)),
                camera_type=str(rng.choice(["plane", "pinhole"])),
                simulation_a=float(rng.choice([0.0, 0.0, 0.0, 0.3, 0.9, 0.998])), simulation_interp=str(rng.choice(["true", "true", "false"])),
                plasma_use_p=str(rng.choice(["true", "false"])), plasma_rat_high=float(rng.uniform(3.0, 40.0)),
                plasma_rat_low=float(rng.uniform(1.0, 3.0)),
                fallback_rho=1.0e-6, fallback_pgas=1.0e-8, fallback_nan="false",
                cut_sigma_max=float(rng.choice([-1.0, 1.0, 10.0])), cut_theta_e_max=float(rng.choice([-1.0, 50.0])),
                cut_beta_inverse_min=float(rng.choice([-1.0, 1.0e-3])), image_num_frequencies=int(rng.choice([1, 1, 3, 5])),
                image_frequency=float(10.0 ** rng.uniform(10.8, 12.3)))
    if over["camera_type"] == "pinhole":
        over["camera_width"] = float(rng.uniform(0.05, 0.4)) * over["camera_r"]
    if over["image_num_frequencies"] > 1:
        over.update(image_frequency_start=1.0e11, image_frequency_end=float(10.0 ** rng.uniform(11.3, 12.0)), image_frequency_spacing="log")
    if over["camera_r"] < 50.0 and rng.random() < 0.4:
        over["fallback_nan"] = "true"
    kind = int(rng.integers(0, 8))
    cks = kind == 1 or kind == 2
    if kind in (2, 3):
        over.update(plasma_power_frac=float(rng.uniform(0.05, 0.6)), plasma_p=float(rng.uniform(2.1, 3.5)),
                    plasma_gamma_min=float(rng.uniform(1.0, 10.0)), plasma_gamma_max=float(rng.uniform(500.0, 5000.0)))
    if cks:
        over.update(simulation_coord="cks", fallback_nan="false")
    if kind == 4:
        over["image_tau"] = "true"
    if kind == 5:
        over["ray_max_steps"] = int(rng.integers(300, 700))
    if rng.random() < 0.15 and not cks:   # optional geometric cuts: the general locate path
        over.update(cut_omit_near="true" if rng.random() < 0.5 else "false", cut_midplane_theta=float(rng.choice([0.0, 20.0])))
    n_r, n_th, n_ph = int(rng.choice([12, 20, 32, 48])), int(rng.choice([8, 16, 24, 40])), int(rng.choice([8, 16, 32]))
    grid = mock.generate(n_r=n_r, n_th=n_th, n_ph=n_ph)
    layout = int(rng.integers(0, 8))   # (5, 6: the two-level mesh of golden_util.refined_blocks, evenly spaced / over warped angles; 7: in smaller blocks)
    changes = {}
    if layout in (1, 3, 6) and not cks:
        x2f = warped(grid.x2f, float(rng.uniform(-0.6, 0.6)))
        changes.update(x2f=x2f, x2v=centres(x2f))
    if layout in (2, 3, 6) and not cks:
        x3f = warped(grid.x3f, float(rng.uniform(-0.6, 0.6)))
        changes.update(x3f=x3f, x3v=centres(x3f))
    if changes:
        grid = dataclasses.replace(grid, **changes)
    if layout == 4:
        grid = gu.split_grid(grid, 2, 2, 2)
    elif layout in (5, 6, 7):
        grid = gu.refined_grid(grid, block=(n_r // 4, n_th // 4, n_ph // 4))
        if layout == 7:   # every block cut in two or four along the axes that allow it (blocks of at least two cells)
            def cut(n):
                return int(rng.choice([c for c in (1, 2, 4) if (n // 4) % c == 0 and (n // 4) // c >= 2]))
            grid = gu.subdivide_blocks(grid, (cut(n_r), cut(n_th), cut(n_ph)))
    else:
        grid = gu.single_block_table(grid)
    return dict(params, **over), grid, dict(kind=kind, layout=layout, grid=[n_r, n_th, n_ph])


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    oracle_every = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    bad = []
    worst = 0.0
    ran_tolerant = 0
    deferred = 0
    t0 = time.time()
    only_layout = os.environ.get("FUZZ_LAYOUT")   # bisecting aids: keep only the draws of one grid layout / one kind
    only_kind = os.environ.get("FUZZ_KIND")
    for seed in range(first, first + n_seeds):
        params, grid, what = draw(seed)
        if (only_layout is not None and what["layout"] != int(only_layout)) or (only_kind is not None and what["kind"] != int(only_kind)):
            continue
        try:
            p = bl.Params.from_dict(params)
            chunk_problems = []
            with bl.Context(p) as ctx:
                ctx.set_grid(grid)
                exact = ctx.render()
                ctx.set_arithmetic("tolerant")
                tol = ctx.render()
                if os.environ.get("FUZZ_CHUNKS") and seed % 2 == 0:   # the same frames under a scratch limit of several chunks: same bits
                    limit = int(max(exact["stats"].n_samples, 1) * 40 + (1 << 20))
                    while True:
                        ctx.set_scratch_limit(limit)
                        try:
                            tol_chunked = ctx.render()
                            ctx.set_arithmetic("exact")
                            exact_chunked = ctx.render()
                            break
                        except bl.BlacklightError as exc:
                            if "Scratch budget too small" not in str(exc):
                                raise
                            ctx.set_arithmetic("tolerant")
                            limit *= 4
                    ctx.set_scratch_limit(144 << 30)
                    ctx.set_arithmetic("tolerant")
                    if not (gu.same_bits(exact_chunked["image"], exact["image"]).all() and np.array_equal(exact_chunked["sample_num"], exact["sample_num"])):
                        chunk_problems.append(f"exact tier in {exact_chunked['stats'].n_chunks} chunks differs")
                    if not (np.array_equal(tol_chunked["sample_num"], tol["sample_num"]) and distance(tol_chunked["image"], tol["image"]) < EXPECTED
                            and np.array_equal(np.isnan(tol_chunked["image"]), np.isnan(tol["image"]))):
                        chunk_problems.append(f"tolerant tier in {tol_chunked['stats'].n_chunks} chunks differs {distance(tol_chunked['image'], tol['image']):.2e}")
                subset = None
                if seed % 4 == 0:   # a shuffled subset of the pixels (what a rank of a tiled job renders): the same pixels
                    rng = np.random.default_rng(99000 + seed)
                    n_all = exact["sample_num"].size
                    subset = rng.permutation(n_all)[: int(rng.integers(1, n_all + 1))].astype(np.int32)
                    part_tol = ctx.render(pixel_map=subset)
                    ctx.set_arithmetic("exact")
                    part = ctx.render(pixel_map=subset)
            problems = list(chunk_problems)
            if subset is not None:
                for name, got, full in (("exact", part, exact), ("tolerant", part_tol, tol)):
                    # (the exact tier bit for bit; the tolerant tier to rounding level since round 4: which of a ray's affine maps are
                    # composed with which follows the order in which the persistent geodesic kernel emitted the records)
                    same_image = (gu.same_bits(got["image"], full["image"][:, subset]).all() if name == "exact" else
                                  (np.array_equal(np.isnan(got["image"]), np.isnan(full["image"][:, subset])) and distance(got["image"], full["image"][:, subset]) < 1.0e-13))
                    if not (same_image and np.array_equal(got["sample_num"], full["sample_num"][subset])
                            and np.array_equal(got["sample_flags"], full["sample_flags"][subset])):
                        problems.append(f"{name} tier: a subset of {subset.size} pixels differs from the full frame "
                                        f"({distance(got['image'], full['image'][:, subset]):.2e})")
            if not np.array_equal(tol["sample_num"], exact["sample_num"]):
                problems.append("sample_num")
            if not np.array_equal(tol["sample_flags"], exact["sample_flags"]):
                problems.append("flags")
            if not np.array_equal(np.isnan(tol["image"]), np.isnan(exact["image"])):
                problems.append("nan mask")
            if str(params["fallback_nan"]) == "false" and tol["stats"].n_gathers != exact["stats"].n_gathers:
                problems.append(f"S_in {tol['stats'].n_gathers} != {exact['stats'].n_gathers}")
            d = distance(tol["image"], exact["image"])
            worst = max(worst, d)
            if not d < EXPECTED:
                problems.append(f"distance {d:.2e}")
            ran_tolerant += int(tol["stats"].arithmetic == 1)
            deferred += int(tol["stats"].n_deferred)
            if oracle_every > 0 and seed % oracle_every == 0:
                n_rays = exact["sample_num"].size
                want = oracle_api.render(p.ptr, grid.desc(), _capi.RenderDesc, _capi.CameraFrame, n_rays=n_rays, max_steps=int(p.get("ray_max_steps")))
                if not np.array_equal(want["sample_num"], exact["sample_num"]):
                    problems.append("oracle sample_num")
                if not gu.same_bits(exact["image"][: want["image"].shape[0]], want["image"]).all():
                    problems.append(f"oracle image {distance(exact['image'][: want['image'].shape[0]], want['image']):.2e}")
            if problems:
                bad.append(seed)
                print(f"seed {seed}: {problems} {what} " + json.dumps({k: v for k, v in params.items() if k.startswith(('camera', 'simulation', 'cut', 'image', 'plasma', 'fallback', 'ray_max'))}), flush=True)
        except Exception as exc:   # noqa: BLE001 - a refusal or an error is a finding too
            bad.append(seed)
            print(f"seed {seed}: raised {type(exc).__name__}: {exc} {what}", flush=True)
        if (seed - first) % 25 == 24:
            print(f"... {seed - first + 1} seeds, {len(bad)} findings, worst distance {worst:.2e}, {time.time() - t0:.0f} s", flush=True)
    print(json.dumps(dict(seeds=n_seeds, first=first, findings=bad, worst_distance=worst, tolerant_ran=ran_tolerant, deferred_samples=deferred,
                          seconds=round(time.time() - t0, 1))))


if __name__ == "__main__":
    main()
