#!/usr/bin/env python3
"""Randomised slow-light frames over the slow-light fixtures' eleven snapshots (tests/golden/slow_*.npz): cameras, spins, camera
times, window handling, time interpolation on and off, trilinear / nearest sampling, frequency lists, auxiliary rows - GPU against
the CPU oracle bit for bit, warnings and refusals included (a tool, not a test).   python3 tools/gpu_fuzz_slow.py [n_seeds] [first]"""
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


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    fixtures = {}
    for case in gu.SLOW_CASES:
        fx = np.load(os.path.join(gu.GOLDEN_DIR, f"{case}.npz"), allow_pickle=False)
        fixtures[case] = (fx, [gu.single_block_table(g) for g in gu.slow_light_grids(fx)], [float(t) for t in fx["file_times"]])
    bad = []
    frames = refused = 0
    t0 = time.time()
    for seed in range(first, first + n_seeds):
        rng = np.random.default_rng(818000 + seed)
        case = str(rng.choice(gu.SLOW_CASES))
        fx, grids, file_times = fixtures[case]
        res = int(rng.choice([8, 12]))
        params = dict(json.loads(str(fx["params"])), camera_resolution=res, camera_r=float(rng.uniform(25.0, 50.0)), camera_th=float(rng.uniform(10.0, 170.0)),
                      camera_ph=float(rng.uniform(0.0, 360.0)), camera_width=float(rng.uniform(6.0, 30.0)), simulation_a=float(rng.choice([0.0, 0.0, 0.5, 0.9])),
                      simulation_interp=str(rng.choice(["true", "false"])), slow_interp=str(rng.choice(["true", "false"])),
                      slow_t_start=float(rng.uniform(150.0, 170.0)), slow_dt=float(rng.uniform(2.0, 15.0)), slow_num_images=int(rng.integers(1, 4)),
                      image_tau=str(rng.choice(["true", "false"])), image_time=str(rng.choice(["true", "false"])),
                      fallback_nan=str(rng.choice(["true", "false"])), fallback_rho=1.0e-6, fallback_pgas=1.0e-8)
        n_freq = int(rng.choice([1, 1, 3]))
        if n_freq > 1:
            params.update(image_num_frequencies=n_freq, image_frequency_start=1.0e11, image_frequency_end=6.0e11, image_frequency_spacing="log")
            params.pop("image_frequency", None)
        try:
            p = bl.Params.from_dict(params)
            with bl.Context(p) as ctx:
                for image, (t_cam, files) in enumerate(gu.slow_light_windows(params, file_times)):
                    for n, f in enumerate(files):
                        ctx.set_grid_slice(n, grids[f], file_times[f])
                    ctx.set_snapshot(image)
                    descs = [grids[f].desc() for f in files]
                    want_error = got_error = None
                    try:
                        want = oracle_api.render(p.ptr, descs[0], _capi.RenderDesc, _capi.CameraFrame, n_rays=res * res, max_steps=int(params["ray_max_steps"]),
                                                 n_freq=n_freq, slow=dict(grids=descs, times=[file_times[f] for f in files], snapshot_time=t_cam))
                    except RuntimeError as exc:
                        want_error = str(exc)
                    try:
                        got = ctx.render()
                    except bl.BlacklightError as exc:
                        got_error = str(exc)
                    frames += 1
                    if want_error is not None:
                        raise RuntimeError("oracle: " + want_error)
                    # what the reference would say about extrapolation in time (simulation_sampling.cpp:577-616), from the oracle's counts
                    def message(kind, direction, count, value):
                        return (f"Snapshot {image} at time {format(t_cam, '.6g')} requires {kind} extrapolation {direction} in time ({count}/{res * res} pixels, "
                                f"by up to {format(value, '.6g')} gravitational times).")
                    count, value = want["slow_count"], want["slow_val"]
                    expect_error = None
                    if count[1] > 0:
                        expect_error = message("significant", "forward", count[1], value[1])
                    elif count[3] > 0:
                        expect_error = message("significant", "backward", count[3], value[3])
                    if expect_error is not None or got_error is not None:
                        refused += 1
                        if got_error is None or expect_error is None or expect_error not in got_error:
                            bad.append(seed)
                            print(f"seed {seed} image {image}: expected error {expect_error!r}, library {got_error!r}", flush=True)
                        break
                    expect_warnings = [message("moderate", d, count[e], value[e]) for e, d in ((0, "forward"), (2, "backward")) if count[e] > 0]
                    for text in expect_warnings:
                        if text not in ctx.warnings:
                            bad.append(seed)
                            print(f"seed {seed} image {image}: missing warning {text!r} in {ctx.warnings!r}", flush=True)
                    if not expect_warnings and "extrapolation" in ctx.warnings:
                        bad.append(seed)
                        print(f"seed {seed} image {image}: unexpected warning {ctx.warnings!r}", flush=True)
                    problems = []
                    if not np.array_equal(got["sample_num"], want["sample_num"]) or not np.array_equal(got["sample_flags"], want["sample_flags"]):
                        problems.append("counts / flags")
                    if got["image"].shape != want["image"].shape or not gu.same_bits(got["image"], want["image"]).all():
                        problems.append("image")
                    if problems:
                        bad.append(seed)
                        print(f"seed {seed} image {image} t = {t_cam}: {problems} " + json.dumps({k: params[k] for k in params if k.startswith(("camera_r", "camera_t", "camera_p", "camera_w", "slow", "simulation_a", "simulation_interp", "image_t", "fallback_nan"))}), flush=True)
                    ctx.clear_warnings()
        except Exception as exc:   # noqa: BLE001
            bad.append(seed)
            print(f"seed {seed}: raised {type(exc).__name__}: {exc}", flush=True)
        if (seed - first) % 10 == 9:
            print(f"... {seed - first + 1} seeds, {frames} frames, {len(bad)} findings, {refused} refused, {time.time() - t0:.0f} s", flush=True)
    print(json.dumps(dict(seeds=n_seeds, first=first, frames=frames, findings=sorted(set(bad)), refused=refused, seconds=round(time.time() - t0, 1))))


if __name__ == "__main__":
    main()
