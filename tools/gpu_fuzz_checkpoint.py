#!/usr/bin/env python3
"""Geodesic checkpoints over the parity test's random configurations (a tool, not a test): render; render again while saving the
checkpoint (geodesic_checkpoint.cpp:28-64); load it in a fresh context (:66-108) in place of the geodesic kernel - the three images
bit for bit in the exact tier, and the loaded frame in the tolerant tier equal to the plain tolerant frame; every fourth draw
saved and loaded under a scratch limit of several chunks.    python3 tools/gpu_fuzz_checkpoint.py [n_seeds] [first_seed]"""
import json
import os
os.environ.setdefault("BLACKLIGHT_AMD_ARITHMETIC", "exact")   # (a context starts in this tier; the tool names the tolerant one where it wants it)
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import blacklight_amd as bl          # noqa: E402
import golden_util as gu             # noqa: E402
from test_gpu_parity import _random_configuration   # noqa: E402


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    bad = []
    chunked = 0
    t0 = time.time()
    work = tempfile.mkdtemp(prefix="bl_ckpt_")
    path = os.path.join(work, "geodesics.ckpt")
    for seed in range(first, first + n_seeds):
        base, over, mesh = _random_configuration(seed)
        try:
            fx, params, mock_args = gu.load_case(base)
            params = dict(params, **over)
            grid = gu.golden_grid(dict(mock_args, **mesh)) if mock_args is not None else None

            def context(**extra):
                ctx = bl.Context(bl.Params.from_dict(dict(params, **extra)))
                if grid is not None:
                    ctx.set_grid(grid)
                return ctx

            problems = []
            with context() as ctx:
                plain = ctx.render()
                ctx.set_arithmetic("tolerant")
                plain_tol = ctx.render()
            limit = int(max(plain["stats"].n_samples, 1) * 60 + (1 << 20)) if seed % 4 == 0 else 0
            with context(checkpoint_geodesic_save="true", checkpoint_geodesic_file=path) as ctx:
                if limit:
                    ctx.set_scratch_limit(limit)
                try:
                    saved = ctx.render()
                except bl.BlacklightError as exc:
                    if "Scratch budget too small" not in str(exc):
                        raise
                    ctx.set_scratch_limit(144 << 30)   # the default
                    limit = 0
                    saved = ctx.render()
                chunked += int(saved["stats"].n_chunks > 1)
            if not (gu.same_bits(saved["image"], plain["image"]).all() and np.array_equal(saved["sample_num"], plain["sample_num"])):
                problems.append("the frame rendered while saving differs")
            with context(checkpoint_geodesic_load="true", checkpoint_geodesic_file=path) as ctx:
                if limit:
                    ctx.set_scratch_limit(limit)
                try:
                    loaded = ctx.render()
                except bl.BlacklightError as exc:   # (a budget below one ray's records is refused: not a finding)
                    if "Scratch budget too small" not in str(exc):
                        raise
                    ctx.set_scratch_limit(144 << 30)   # the default
                    loaded = ctx.render()
                ctx.set_arithmetic("tolerant")
                try:
                    loaded_tol = ctx.render()
                except bl.BlacklightError as exc:   # (the tolerant tier's records per ray are larger)
                    if "Scratch budget too small" not in str(exc):
                        raise
                    ctx.set_scratch_limit(144 << 30)   # the default
                    loaded_tol = ctx.render()
            if not (gu.same_bits(loaded["image"], plain["image"]).all() and np.array_equal(loaded["sample_num"], plain["sample_num"])
                    and np.array_equal(loaded["sample_flags"], plain["sample_flags"])):
                problems.append("the frame from the loaded checkpoint differs")
            with np.errstate(invalid="ignore"):
                scale = np.nanmax(np.abs(np.where(np.isfinite(plain_tol["image"]), plain_tol["image"], np.nan)), axis=-1, keepdims=True)
                scale = np.where(np.isfinite(scale) & (scale > 0), scale, 1.0)
                d = np.abs(loaded_tol["image"] - plain_tol["image"]) / scale
            d = float(np.nanmax(d)) if np.isfinite(d).any() else 0.0
            if not np.array_equal(np.isnan(loaded_tol["image"]), np.isnan(plain_tol["image"])) or not d < 1.0e-11:
                problems.append(f"tolerant tier from the loaded checkpoint: {d:.2e}")
            if problems:
                bad.append(seed)
                print(f"seed {seed}: {problems} base {base} mesh {mesh} {json.dumps(over)}"[:700], flush=True)
        except Exception as exc:   # noqa: BLE001
            bad.append(seed)
            print(f"seed {seed}: raised {type(exc).__name__}: {exc} base {base} mesh {mesh} {json.dumps(over)}"[:700], flush=True)
        if (seed - first) % 20 == 19:
            print(f"... {seed - first + 1} seeds, {len(bad)} findings, {chunked} in several chunks, {time.time() - t0:.0f} s", flush=True)
    print(json.dumps(dict(seeds=n_seeds, first=first, findings=bad, chunked=chunked, seconds=round(time.time() - t0, 1))))


if __name__ == "__main__":
    main()
