#!/usr/bin/env python3
"""Per-row distances of the tolerant tier in polarized draws of tools/gpu_fuzz_wide.py: python3 tools/gpu_fuzz_detail3.py seed [...]"""
import json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import blacklight_amd as bl
import golden_util as gu
from test_gpu_parity import _random_configuration

for seed in [int(s) for s in sys.argv[1:]]:
    base, over, mesh = _random_configuration(seed)
    rng = np.random.default_rng(77000 + seed)
    over = dict(over, image_polarization="true", image_rotation_split=str(rng.choice(["true", "false"])), camera_resolution=12)
    if rng.integers(0, 3) == 0:
        over.update(plasma_kappa_frac=float(rng.uniform(0.05, 0.5)), plasma_kappa=float(rng.uniform(3.6, 7.5)), plasma_w=float(rng.uniform(1.0, 30.0)))
    fx, params, mock_args = gu.load_case(base)
    params = dict(params, **over)
    p = bl.Params.from_dict(params)
    for switch in ("", "BLACKLIGHT_AMD_TENSOR_TRANSPORT", "BLACKLIGHT_AMD_TOLERANT_POLARIZED_COEFFICIENTS"):
        for name in ("BLACKLIGHT_AMD_TENSOR_TRANSPORT", "BLACKLIGHT_AMD_TOLERANT_POLARIZED_COEFFICIENTS"):
            os.environ.pop(name, None)
        if switch:
            os.environ[switch] = "1"
        with bl.Context(p) as ctx:
            ctx.set_grid(gu.golden_grid(dict(mock_args, **mesh)))
            exact = ctx.render()
            ctx.set_arithmetic("tolerant")
            tol = ctx.render()
        e, t = exact["image"], tol["image"]
        with np.errstate(invalid="ignore"):
            per_row = [float(np.nanmax(np.abs(t[r] - e[r])) / max(np.nanmax(np.abs(e[r])), 1e-300)) for r in range(e.shape[0])]
        print(f"   [{switch or 'default'}] per-row distance: " + " ".join(f"{d:.1e}" for d in per_row))
    print(f"== seed {seed} rows {e.shape[0]} " + json.dumps({k: over[k] for k in over if k.startswith(("image", "simulation", "plasma", "ray_int", "ray_step", "cut", "fallback_nan"))}))
    i_max = np.nanmax(np.abs(e[0]))
    for r in range(e.shape[0]):
        with np.errstate(invalid="ignore"):
            d = np.abs(t[r] - e[r])
        if not np.isfinite(d).any():
            continue
        c = int(np.nanargmax(d))
        row_max = np.nanmax(np.abs(e[r]))
        print(f"   row {r}: max |exact| {row_max:.3e} worst diff {np.nanmax(d):.3e} = {np.nanmax(d) / max(row_max, 1e-300):.2e} of row, {np.nanmax(d) / i_max:.2e} of I max; at pixel {c}: exact {e[r, c]:.6e} tolerant {t[r, c]:.6e} samples {exact['sample_num'][c]}")
