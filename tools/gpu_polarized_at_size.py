"""Config 4's physics at size on one GPU: full-Stokes polarized transfer + image_tau (example_adaptive.input)
over the 256^3 mock, plain camera of the given resolution (default 1024). Prints per-kernel times."""
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

res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
grid = mock.generate(n_r=256, n_th=256, n_ph=256)
p = dict(bench.WORKLOAD)
p.update(camera_resolution=res, image_polarization=True, image_tau=True)
with bl.Context(bl.Params.from_dict(p)) as ctx:
    ctx.set_geodesic_reuse(False)   # a measurement of whole renders: every one integrates its geodesics
    ctx.set_grid(grid)
    ctx.set_arithmetic(os.environ.get("ARITH", "exact"))
    if "SCRATCH_GB" in os.environ:
        ctx.set_scratch_limit(int(float(os.environ["SCRATCH_GB"]) * 1e9))
    ctx.render()
    t0 = time.perf_counter()
    out = ctx.render()
    sec = time.perf_counter() - t0
    st = out["stats"]
    fields = {k: getattr(st, k) for k in dir(st) if k.startswith(("ms_", "n_"))}
    img = out["image"]
    print(json.dumps(dict(resolution=res, seconds=sec, mrays_per_s=res * res / sec / 1e6, stats=fields,
                          finite_fraction=float(np.isfinite(img).mean()), rows=int(img.shape[0])), indent=1, default=float))
