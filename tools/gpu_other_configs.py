"""Run BASELINE.json's other single-GPU configurations at size (parity is covered by the tests; this
checks that they run at size and reports their speed): config 1 = example_formula at 512^2, config 4's
physics = example_true_color (10 frequencies, lin_wave) at 1024^2 over the 256^3 mock."""
import json
import os
os.environ.setdefault("BLACKLIGHT_AMD_ARITHMETIC", "exact")   # (a context starts in this tier; the tool names the tolerant one where it wants it)
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import blacklight_amd as bl
from blacklight_amd import mock
import bench
import golden_util as gu


def timed(ctx, n=3):
    ctx.render()
    best = 1e30
    for _ in range(n):
        t0 = time.perf_counter()
        out = ctx.render()
        best = min(best, time.perf_counter() - t0)
    return out, best


out = {}
fx, params, _ = gu.load_case("formula_dp")
params = dict(params)
params.update(camera_resolution=512)
with bl.Context(bl.Params.from_dict(params)) as ctx:
    ctx.set_geodesic_reuse(False)   # a measurement of whole renders: every one integrates its geodesics
    res, sec = timed(ctx)
    st = res["stats"]
    out["formula_512"] = dict(seconds=sec, mrays_per_s=512 * 512 / sec / 1e6, samples_per_ray=st.n_samples / (512 * 512),
                              ms_geodesic=st.ms_geodesic, ms_shade=st.ms_shade, ms_transfer=st.ms_transfer, chunks=st.n_chunks)
grid = mock.generate(n_r=256, n_th=256, n_ph=256)
for nf, tier in ((10, "exact"), (10, "tolerant"), (64, "exact"), (64, "tolerant")):
    p = dict(bench.WORKLOAD)
    p.update(image_num_frequencies=nf, image_frequency_start=1.5e11, image_frequency_end=3.3e11, image_frequency_spacing="lin_wave")
    with bl.Context(bl.Params.from_dict(p)) as ctx:
        ctx.set_geodesic_reuse(False)   # a measurement of whole renders: every one integrates its geodesics
        ctx.set_grid(grid)
        ctx.set_arithmetic(tier)
        res, sec = timed(ctx, n=2)
        st = res["stats"]
        out[f"true_color_1024_{nf}freq_{tier}"] = dict(seconds=sec, mrays_per_s=1024 * 1024 / sec / 1e6, ms_geodesic=st.ms_geodesic,
                                                ms_locate=st.ms_locate, ms_shade=st.ms_shade, ms_transfer=st.ms_transfer, chunks=st.n_chunks,
                                                finite_fraction=float(np.isfinite(res["image"]).mean()))
# camera outside the grid (r = 100, the grid ends at 52; fallback values beyond it): the benchmark frame from farther away.
# Steps in the empty shell leave no records (BlTraceArgs::skip_low); the RECORD_EVERY_STEP switch = the same frame without that.
for tier in ("exact", "tolerant"):
    p = dict(bench.WORKLOAD)
    p.update(camera_r=100.0, fallback_nan=False, fallback_rho=1.0e-6, fallback_pgas=1.0e-8)
    with bl.Context(bl.Params.from_dict(p)) as ctx:
        ctx.set_geodesic_reuse(False)   # a measurement of whole renders: every one integrates its geodesics
        ctx.set_grid(grid)
        ctx.set_arithmetic(tier)
        for every in (False, True):
            if every:
                ctx.debug_set_switches("RECORD_EVERY_STEP")
            res, sec = timed(ctx, n=2)
            ctx.debug_set_switches()
            st = res["stats"]
            out[f"camera_at_100_{tier}" + ("_every_step_recorded" if every else "")] = dict(
                seconds=sec, mrays_per_s=1024 * 1024 / sec / 1e6, samples_per_ray=st.n_samples / (1024 * 1024),
                records_per_ray=st.n_samples_emitted / (1024 * 1024), ms_geodesic=st.ms_geodesic, ms_locate=st.ms_locate,
                ms_shade=st.ms_shade, ms_transfer=st.ms_transfer, chunks=st.n_chunks, image_sum=float(np.nansum(res["image"])))
print(json.dumps(out, indent=1))
