#!/usr/bin/env python3
"""Details of findings of tools/gpu_fuzz_wide.py: python3 tools/gpu_fuzz_detail.py seed [seed ...]"""
import json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import blacklight_amd as bl
from blacklight_amd import _capi
import golden_util as gu
import oracle_api
from test_gpu_parity import _random_configuration

for seed in [int(s) for s in sys.argv[1:]]:
    base, over, mesh = _random_configuration(seed)
    fx, params, mock_args = gu.load_case(base)
    params = dict(params, **over)
    if mock_args is not None:
        mock_args = dict(mock_args, **mesh)
    p = bl.Params.from_dict(params)
    grid = gu.golden_grid(mock_args) if mock_args is not None else None
    res = int(p.get("camera_resolution"))
    with bl.Context(p) as ctx:
        if grid is not None:
            ctx.set_grid(grid)
        exact = ctx.render()
        ctx.set_arithmetic("tolerant")
        tol = ctx.render()
    want = oracle_api.render(p.ptr, grid.desc() if grid is not None else None, _capi.RenderDesc, _capi.CameraFrame, n_rays=res * res,
                             max_steps=int(p.get("ray_max_steps")), n_freq=int(p.get("image_num_frequencies")))
    print(f"== seed {seed} base {base} mesh {mesh} tier ran {tol['stats'].arithmetic} deferred {tol['stats'].n_deferred}")
    print("   " + json.dumps({k: over[k] for k in over if k.startswith(("image", "ray", "formula", "fallback", "simulation", "cut", "plasma"))}))
    same = gu.same_bits(exact["image"], want["image"])
    if not same.all():
        rows, cols = np.nonzero(~same)
        print(f"   exact vs oracle: {rows.size} values differ, rows {sorted(set(rows.tolist()))}")
        for r, c in list(zip(rows, cols))[:6]:
            print(f"     row {r} pixel {c}: gpu {exact['image'][r, c]!r} ({exact['image'][r, c].hex() if np.isfinite(exact['image'][r, c]) else ''}) oracle {want['image'][r, c]!r} "
                  f"({want['image'][r, c].hex() if np.isfinite(want['image'][r, c]) else ''}) sample_num {exact['sample_num'][c]} flag {exact['sample_flags'][c]}")
    nan_e, nan_t = np.isnan(exact["image"]), np.isnan(tol["image"])
    if not np.array_equal(nan_e, nan_t):
        rows, cols = np.nonzero(nan_e != nan_t)
        print(f"   tolerant vs exact NaN mask: {rows.size} differ, rows {sorted(set(rows.tolist()))}")
        for r, c in list(zip(rows, cols))[:6]:
            print(f"     row {r} pixel {c}: exact {exact['image'][r, c]!r} tolerant {tol['image'][r, c]!r} oracle {want['image'][r, c]!r} sample_num {exact['sample_num'][c]} flag {exact['sample_flags'][c]}")
