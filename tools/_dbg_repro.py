import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, blacklight_amd as bl
from blacklight_amd import mock
res, n_freq = 512, 8
params = dict(bench.WORKLOAD, camera_resolution=res, image_num_frequencies=n_freq, image_frequency_start=1.5e11, image_frequency_end=3.3e11, image_frequency_spacing="lin_wave")
params.pop("image_frequency", None)
grid = mock.generate(n_r=64, n_th=64, n_ph=64)
def count(a, b):
    bad = (a.view(np.uint64) != b.view(np.uint64)) & ~(np.isnan(a) & np.isnan(b))
    return int(bad.sum()), np.nonzero(bad.any(axis=0))[0][:5]
with bl.Context(bl.Params.from_dict(params)) as ctx:
    ctx.set_geodesic_reuse(False)
    ctx.set_grid(grid)
    ctx.set_arithmetic("tolerant")
    for name, switches, band in (("default", (), None), ("no fused locate", ("NO_FUSED_LOCATE",), None), ("default, guard band 0", (), 0.0), ("default, wide band", (), 1e-3)):
        ctx.debug_set_switches(*switches)
        if band is not None:
            ctx.debug_set_guard_band(band)
        first = ctx.render()
        out = []
        for _ in range(8):
            again = ctx.render()
            out.append(count(again["image"], first["image"]))
        print(name, "fused", first["stats"].fused_variant, "deferred", first["stats"].n_deferred, "diffs", [o[0] for o in out], "pixels", [list(o[1]) for o in out if o[0]][:3], flush=True)
    ctx.debug_set_switches()
    ctx.debug_set_guard_band(1e-9)
    # pixel order vs tile order
    ctx.set_geodesic_reuse(True)
    a = ctx.render(); b = ctx.render()
    print("reused records:", count(b["image"], a["image"]), b["stats"].geodesics_reused)
