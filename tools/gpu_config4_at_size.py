"""BASELINE.json's config 4 on ONE GPU at size: example_adaptive.input (full-Stokes polarized transfer + image_tau,
adaptive refinement with 8x8 blocks, one level, relative-Laplacian criterion) with a 2048^2 root camera over the 256^3
mock. Prints the time of the whole adaptive loop and the number of refined blocks."""
import json
import os
os.environ.setdefault("BLACKLIGHT_AMD_ARITHMETIC", "exact")   # (a context starts in this tier; the tool names the tolerant one where it wants it)
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import blacklight_amd as bl
from blacklight_amd import mock
import bench

res = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
grid = mock.generate(n_r=256, n_th=256, n_ph=256)
p = dict(bench.WORKLOAD)
p.update(camera_resolution=res, image_polarization=True, image_tau=True, adaptive_max_level=1, adaptive_block_size=8,
         adaptive_frequency_num=1, adaptive_val_cut=0.0, adaptive_val_frac=-1.0, adaptive_abs_grad_cut=0.0,
         adaptive_abs_grad_frac=-1.0, adaptive_rel_grad_cut=0.0, adaptive_rel_grad_frac=-1.0, adaptive_abs_lapl_cut=0.0,
         adaptive_abs_lapl_frac=-1.0, adaptive_rel_lapl_cut=1.0, adaptive_rel_lapl_frac=0.25, adaptive_num_regions=0)
with bl.Context(bl.Params.from_dict(p)) as ctx:
    ctx.set_geodesic_reuse(False)   # a measurement of whole renders: every one integrates its geodesics
    ctx.set_grid(grid)
    ctx.set_arithmetic(os.environ.get("ARITH", "exact"))
    ctx.render_adaptive()                      # first run: allocations
    t0 = time.perf_counter()
    levels = ctx.render_adaptive()
    sec = time.perf_counter() - t0
    rays = sum(int(lv["image"].shape[1]) for lv in levels)
    print(json.dumps(dict(root_resolution=res, arithmetic=os.environ.get("ARITH", "exact"), seconds=sec, levels=len(levels), rays=rays, mrays_per_s=rays / sec / 1e6,
                          blocks_per_level=[int(lv["image"].shape[1]) // 64 for lv in levels],
                          finite_fraction=[float(np.isfinite(lv["image"]).mean()) for lv in levels]), indent=1))
