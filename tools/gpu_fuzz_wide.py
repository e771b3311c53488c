#!/usr/bin/env python3
"""Randomised sweep over the whole supported parameter space (a tool, not a test): the generator of
tests/test_gpu_parity.py::test_randomised_configurations_against_oracle - cameras, spins, integrators, termination rules, frequency
lists, plasma / formula parameters, cuts, power-law electrons, auxiliary images, single-block / multi-block / refined meshes,
electron entropy, Cartesian grids - run for many more seeds than the test keeps, and on top of it:
  * the exact tier against the CPU oracle, every output bit for bit;
  * the tolerant tier against the exact one: counts, flags and NaN masks identical, image rows within 1e-11 of their maximum
    (bit-identical where the tier does not apply);
  * every third seed with a scratch limit that forces the frame through several chunks: same bits as in one chunk.

    python3 tools/gpu_fuzz_wide.py [n_seeds] [first_seed]"""
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
from blacklight_amd import _capi     # noqa: E402
import golden_util as gu             # noqa: E402
import oracle_api                    # noqa: E402
from test_gpu_parity import _random_configuration   # noqa: E402

EXPECTED = 1.0e-11


def distance(a, b):
    with np.errstate(invalid="ignore", all="ignore"):
        scale = np.nanmax(np.abs(np.where(np.isfinite(b), b, np.nan)), axis=-1, keepdims=True)
        scale = np.where(np.isfinite(scale) & (scale > 0), scale, 1.0)
        d = np.abs(a - b) / scale
    return float(np.nanmax(d)) if np.isfinite(d).any() else 0.0


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    polarized = os.environ.get("FUZZ_POLARIZED") is not None   # simulation draws with full-Stokes transfer (tolerant tier: 1e-9)
    bad = []
    worst = 0.0
    tolerant_ran = chunked = amplified = refined = 0
    spread_ratio = 0.0
    t0 = time.time()
    for seed in range(first, first + n_seeds):
        base, over, mesh = _random_configuration(seed)
        if polarized:
            if base != "sim_dp_interp" or over.get("plasma_model") == "code_kappa":
                continue
            rng = np.random.default_rng(77000 + seed)
            over = dict(over, image_polarization="true", image_rotation_split=str(rng.choice(["true", "false"])), camera_resolution=12)
            if rng.integers(0, 3) == 0:
                over.update(plasma_kappa_frac=float(rng.uniform(0.05, 0.5)), plasma_kappa=float(rng.uniform(3.55, 4.95)), plasma_w=float(rng.uniform(1.0, 30.0)))
        if os.environ.get("FUZZ_RES") and not polarized:   # larger frames: many waves, several chunks, refills of the persistent kernel
            over = dict(over, camera_resolution=int(np.random.default_rng(66000 + seed).choice([int(v) for v in os.environ["FUZZ_RES"].split(",")])))
        adaptive = os.environ.get("FUZZ_ADAPTIVE") is not None and not polarized
        if adaptive:
            rng_a = np.random.default_rng(55000 + seed)
            over = dict(over, camera_resolution=int(rng_a.choice([16, 24])), adaptive_max_level=int(rng_a.integers(1, 3)), adaptive_block_size=int(rng_a.choice([4, 8])),
                        adaptive_frequency_num=1, adaptive_val_cut=0.0, adaptive_val_frac=-1.0, adaptive_abs_grad_cut=0.0, adaptive_abs_grad_frac=-1.0,
                        adaptive_rel_grad_cut=float(rng_a.uniform(0.05, 0.5)), adaptive_rel_grad_frac=float(rng_a.choice([0.1, 0.25, 0.5])),
                        adaptive_abs_lapl_cut=0.0, adaptive_abs_lapl_frac=-1.0, adaptive_rel_lapl_cut=float(rng_a.uniform(0.2, 2.0)),
                        adaptive_rel_lapl_frac=float(rng_a.choice([-1.0, 0.25])), adaptive_num_regions=0)
        try:
            fx, params, mock_args = gu.load_case(base)
            params = dict(params, **over)
            if mock_args is not None:
                mock_args = dict(mock_args, **mesh)
            p = bl.Params.from_dict(params)
            grid = gu.golden_grid(mock_args) if mock_args is not None else None
            res = int(p.get("camera_resolution"))
            problems = []
            with bl.Context(p) as ctx:
                if grid is not None:
                    ctx.set_grid(grid)
                exact = ctx.render()
                ctx.set_arithmetic("tolerant")
                tol = ctx.render()
                if seed % 3 == 0:
                    limit = int(max(exact["stats"].n_samples, 1) * 40 + (1 << 20))   # a fraction of what one chunk would take
                    while True:
                        ctx.set_scratch_limit(limit)
                        try:
                            again_tol = ctx.render()
                            ctx.set_arithmetic("exact")
                            again = ctx.render()
                            break
                        except bl.BlacklightError as exc:   # (a budget below one ray's records is refused: not a finding)
                            if "Scratch budget too small" not in str(exc):
                                raise
                            ctx.set_arithmetic("tolerant")
                            limit *= 4
                    chunked += int(again["stats"].launches_shade > exact["stats"].launches_shade)
                    if not (gu.same_bits(again["image"], exact["image"]).all() and np.array_equal(again["sample_num"], exact["sample_num"])):
                        problems.append(f"chunked exact differs ({again['stats'].launches_shade} launches)")
                    if not np.array_equal(again_tol["sample_num"], tol["sample_num"]) or not distance(again_tol["image"], tol["image"]) < (1.0e-9 if polarized else EXPECTED):
                        problems.append(f"chunked tolerant differs {distance(again_tol['image'], tol['image']):.2e}")
                if adaptive:   # refined levels (exact tier): every level's rays against the oracle's for the same block list
                    ctx.set_arithmetic("exact")
                    ctx.set_scratch_limit(144 << 30)
                    levels = ctx.render_adaptive()
                    refined += int(len(levels) > 1)
                    bs = int(p.get("adaptive_block_size"))
                    for level, lv in enumerate(levels[1:], start=1):
                        w = oracle_api.render(p.ptr, grid.desc() if grid is not None else None, _capi.RenderDesc, _capi.CameraFrame,
                                              n_rays=lv["block_locs"].shape[0] * bs * bs, level=level, block_locs=lv["block_locs"],
                                              max_steps=int(p.get("ray_max_steps")), n_freq=int(p.get("image_num_frequencies")))
                        if not (np.array_equal(lv["sample_num"], w["sample_num"]) and gu.same_bits(lv["image"], w["image"]).all()):
                            problems.append(f"adaptive level {level} ({lv['block_locs'].shape[0]} blocks) differs from the oracle")
            want = oracle_api.render(p.ptr, grid.desc() if grid is not None else None, _capi.RenderDesc, _capi.CameraFrame, n_rays=res * res,
                                     max_steps=int(p.get("ray_max_steps")), n_freq=int(p.get("image_num_frequencies")))
            if not np.array_equal(exact["sample_num"], want["sample_num"]) or not np.array_equal(exact["sample_flags"], want["sample_flags"]):
                problems.append("oracle counts / flags")
            if exact["image"].shape != want["image"].shape or not gu.same_bits(exact["image"], want["image"]).all():
                problems.append("oracle image")
            if not np.array_equal(tol["sample_num"], exact["sample_num"]) or not np.array_equal(tol["sample_flags"], exact["sample_flags"]):
                problems.append("tolerant counts / flags")
            if not np.array_equal(np.isnan(tol["image"]), np.isnan(exact["image"])):
                problems.append("tolerant NaN mask")
            if tol["stats"].arithmetic == 1:
                tolerant_ran += 1
                d = distance(tol["image"], exact["image"])
                worst = max(worst, d)
                if polarized and not d < 1.0e-9:
                    # The reference's polarized step amplifies last-place differences of its elementary functions in some
                    # configurations (optically and Faraday thick steps): measured here as the distance between the oracle with the
                    # pinned math library and the same oracle with the host's libm - the reference against itself. The tolerant
                    # tier has to stay within a small multiple of that, row by row.
                    other = oracle_api.render(p.ptr, grid.desc(), _capi.RenderDesc, _capi.CameraFrame, n_rays=res * res, variant="libm",
                                              max_steps=int(p.get("ray_max_steps")), n_freq=int(p.get("image_num_frequencies")))
                    if np.array_equal(other["sample_num"], want["sample_num"]):
                        rows_tol = [distance(tol["image"][r:r + 1], exact["image"][r:r + 1]) for r in range(exact["image"].shape[0])]
                        rows_ref = [distance(other["image"][r:r + 1], want["image"][r:r + 1]) for r in range(exact["image"].shape[0])]
                        ratio = max(t / max(f, 1.0e-11) for t, f in zip(rows_tol, rows_ref))
                        spread_ratio = max(spread_ratio, ratio)
                        amplified += 1
                        if ratio > 30.0:
                            problems.append(f"tolerant distance {d:.2e}, {ratio:.1f} x the reference's own spread between math libraries")
                    # (different sample counts between the two oracles: glibc's hypot / pow move steps of spinning rays - no measure)
                elif not polarized and not d < EXPECTED:
                    problems.append(f"tolerant distance {d:.2e}")
            elif not gu.same_bits(tol["image"], exact["image"]).all():
                problems.append("tolerant tier fell back to exact kernels but differs")
            if problems:
                bad.append(seed)
                print(f"seed {seed}: {problems} base {base} mesh {mesh} {json.dumps(over)}", flush=True)
        except Exception as exc:   # noqa: BLE001 - a refusal or an error is a finding too
            bad.append(seed)
            print(f"seed {seed}: raised {type(exc).__name__}: {exc} base {base} mesh {mesh} {json.dumps(over)}", flush=True)
        if (seed - first) % 25 == 24:
            print(f"... {seed - first + 1} seeds, {len(bad)} findings, worst tolerant distance {worst:.2e}, {time.time() - t0:.0f} s", flush=True)
    print(json.dumps(dict(seeds=n_seeds, first=first, findings=bad, worst_tolerant_distance=worst, tolerant_ran=tolerant_ran, chunked=chunked, amplified=amplified, refined=refined, worst_ratio_to_reference_spread=spread_ratio,
                          seconds=round(time.time() - t0, 1))))


if __name__ == "__main__":
    main()
