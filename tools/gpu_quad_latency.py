"""The time of a ray alone: frames of 64 rays around the photon ring (one wave of bl_geodesic_kernel, four of bl_geodesic_quad_kernel with
BL_SWITCH_QUAD_EVERY_RAY) - the geodesic stage's time is its longest ray's.   python3 tools/gpu_quad_latency.py"""
import os
os.environ.setdefault("BLACKLIGHT_AMD_ARITHMETIC", "exact")   # (a context starts in this tier; the tool names the tolerant one where it wants it)
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import blacklight_amd as bl
import golden_util as gu

fx, params, _ = gu.load_case("formula_dp")
for spin in (0.9, 0.0):
    for width, res in ((12.0, 8), (12.0, 16), (30.0, 64)):
        p = bl.Params.from_dict(dict(params, camera_resolution=res, camera_width=width, formula_spin=spin))
        with bl.Context(p) as ctx:
            ctx.set_geodesic_reuse(False)   # a measurement of whole renders: every one integrates its geodesics
            row = []
            for name, switches in (("ray per lane", ()), ("ray per quad", ("QUAD_EVERY_RAY",))):
                ctx.debug_set_switches(*switches)
                times = []
                for rep in range(4):
                    out = ctx.render()
                    times.append(out["stats"].ms_geodesic)
                row.append((name, min(times), int(out["sample_num"].max()), int(out["stats"].n_samples)))
            print(f"spin {spin} {res}^2 rays over width {width}: " + "; ".join(f"{n}: geodesic stage {t:.2f} ms" for n, t, _, _ in row)
                  + f"; longest ray {row[0][2]} samples, {row[0][3]} samples in all; ratio {row[1][1] / row[0][1]:.2f}")
