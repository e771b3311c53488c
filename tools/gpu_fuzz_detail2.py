#!/usr/bin/env python3
"""NaN-mask findings of tools/gpu_fuzz_wide.py under the tolerant tier's switches: python3 tools/gpu_fuzz_detail2.py seed [seed ...]"""
import json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import blacklight_amd as bl
import golden_util as gu
from test_gpu_parity import _random_configuration

for seed in [int(s) for s in sys.argv[1:]]:
    base, over, mesh = _random_configuration(seed)
    fx, params, mock_args = gu.load_case(base)
    params = dict(params, **over)
    mock_args = dict(mock_args, **mesh)
    print(f"== seed {seed} mesh {mesh} " + json.dumps({k: over[k] for k in over if k.startswith(("camera_r", "camera_type", "ray_t", "ray_f", "fallback_nan", "simulation", "cut", "plasma_power", "image_num"))}))
    for variant in ("default", "no_fused", "band", "exact_again"):
        os.environ.pop("BLACKLIGHT_AMD_NO_FUSED_LOCATE", None)
        if variant == "no_fused":
            os.environ["BLACKLIGHT_AMD_NO_FUSED_LOCATE"] = "1"
        p = bl.Params.from_dict(params)
        with bl.Context(p) as ctx:
            ctx.set_grid(gu.golden_grid(mock_args))
            exact = ctx.render()
            ctx.set_arithmetic("tolerant")
            if variant == "band":
                ctx.debug_set_guard_band(1.0e30)
            tol = ctx.render()
        ne, nt = np.isnan(exact["image"]), np.isnan(tol["image"])
        print(f"   {variant}: NaN exact {int(ne.sum())} tolerant {int(nt.sum())} differ {int((ne != nt).sum())}; launches locate {tol['stats'].launches_locate} deferred {tol['stats'].n_deferred} "
              f"S_in exact {exact['stats'].n_gathers} tolerant {tol['stats'].n_gathers}")
