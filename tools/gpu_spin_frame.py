"""The benchmark frame with a spinning hole (simulation_a = 0.9375; the mock is accepted for any spin): per-kernel times of the
general-spin instantiations. python tools/gpu_spin_frame.py [exact|tolerant]"""
import json
import os
os.environ.setdefault("BLACKLIGHT_AMD_ARITHMETIC", "exact")   # (a context starts in this tier; the tool names the tolerant one where it wants it)
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import blacklight_amd as bl
from blacklight_amd import mock
import bench

grid = mock.generate(n_r=256, n_th=256, n_ph=256)
p = dict(bench.WORKLOAD)
p.update(simulation_a=0.9375)
with bl.Context(bl.Params.from_dict(p)) as ctx:
    ctx.set_geodesic_reuse(False)   # a measurement of whole renders: every one integrates its geodesics
    ctx.set_grid(grid)
    ctx.set_arithmetic(sys.argv[1] if len(sys.argv) > 1 else "tolerant")
    ctx.render()
    out = ctx.render()
    st = out["stats"]
    print(json.dumps({k: round(getattr(st, k), 2) for k in ("ms_geodesic", "ms_locate", "ms_shade", "ms_transfer", "ms_total")}
                     | dict(samples_per_ray=st.n_samples / st.n_rays)))
