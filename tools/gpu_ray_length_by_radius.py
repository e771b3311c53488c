"""How long the benchmark frame's rays take by impact parameter: 64 pixels of the 1024^2 plane camera on a ring of radius b (one wave of
bl_geodesic_kernel; its time is its longest ray's), b in steps.   python3 tools/gpu_ray_length_by_radius.py"""
import os
os.environ.setdefault("BLACKLIGHT_AMD_ARITHMETIC", "exact")   # (a context starts in this tier; the tool names the tolerant one where it wants it)
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import blacklight_amd as bl
from blacklight_amd import mock
import bench

res = 1024
width = bench.WORKLOAD["camera_width"]
grid = mock.generate(n_r=64, n_th=64, n_ph=64)
with bl.Context(bl.Params.from_dict(dict(bench.WORKLOAD))) as ctx:
    ctx.set_geodesic_reuse(False)   # a measurement of whole renders: every one integrates its geodesics
    ctx.set_grid(grid)
    ctx.set_arithmetic("exact")
    for b in (list(np.arange(0.5, 4.0, 0.5)) + list(np.arange(4.0, 7.01, 0.2)) + list(np.arange(7.5, 12.1, 0.5)) if not os.environ.get("FINE") else list(np.arange(4.9, 5.7, 0.025))):
        ang = np.linspace(0.0, 2.0 * np.pi, 64, endpoint=False)
        r_px = b / width * res
        m1 = np.clip(np.round(res / 2 - 0.5 + r_px * np.cos(ang)).astype(np.int64), 0, res - 1)
        m2 = np.clip(np.round(res / 2 - 0.5 + r_px * np.sin(ang)).astype(np.int64), 0, res - 1)
        pixels = (m2 * res + m1).astype(np.int32)
        best = None
        for rep in range(3):
            out = ctx.render(pixel_map=pixels)
            t = out["stats"].ms_geodesic
            best = t if best is None else min(best, t)
        print(f"b = {b:5.2f} M: geodesic stage {best:6.3f} ms, samples per ray mean {out['sample_num'].mean():7.1f} max {out['sample_num'].max()}")
