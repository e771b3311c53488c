#!/usr/bin/env python3
"""Tolerant tier against the exact tier on the benchmark's frame (or a smaller one): the worst pixels, the distribution of the
distances, how many samples went to the exact second pass.  python3 tools/gpu_tier_distance.py [resolution] [grid]
BLACKLIGHT_AMD_GENERAL_FUSED=1 in the environment selects the general fused kernel (A/B of the two)."""
import os
os.environ.setdefault("BLACKLIGHT_AMD_ARITHMETIC", "exact")   # (a context starts in this tier; the tool names the tolerant one where it wants it)
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
import blacklight_amd as bl
from blacklight_amd import mock

res = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_grid = int(sys.argv[2]) if len(sys.argv) > 2 else 256
params = dict(bench.WORKLOAD, camera_resolution=res)
p = bl.Params.from_dict(params)
grid = mock.generate(n_r=n_grid, n_th=n_grid, n_ph=n_grid)
with bl.Context(p) as ctx:
    ctx.set_grid(grid)
    exact = ctx.render()
    ctx.set_arithmetic("tolerant")
    tol = ctx.render()
st = tol["stats"]
e, t = exact["image"][0], tol["image"][0]
peak = np.nanmax(np.abs(e))
both_nan = np.isnan(e) & np.isnan(t)
d = np.where(both_nan, 0.0, np.abs(t - e)) / peak
print(f"frame {res}^2 over {n_grid}^3: fused_variant {st.fused_variant} switches {st.switches} deferred {st.n_deferred} of {st.n_gathers} gathered samples "
      f"({st.n_deferred / max(st.n_gathers, 1):.2e}); sample_num equal {np.array_equal(exact['sample_num'], tol['sample_num'])}; "
      f"NaN masks equal {np.array_equal(np.isnan(e), np.isnan(t))}")
print("distance / image maximum: max %.3e, 99.99 %% %.3e, 99 %% %.3e, median %.3e" % (np.nanmax(d), np.nanquantile(d, 0.9999), np.nanquantile(d, 0.99), np.nanmedian(d)))
order = np.argsort(-np.nan_to_num(d))[:8]
for pix in order:
    print(f"   pixel ({pix // res}, {pix % res}): exact {e[pix]:.17e} tolerant {t[pix]:.17e} distance {d[pix]:.3e} relative {abs(t[pix] - e[pix]) / abs(e[pix]):.3e} samples {exact['sample_num'][pix]}")
print("pixels beyond 1e-13: %d, beyond 1e-12: %d" % ((d > 1e-13).sum(), (d > 1e-12).sum()))
