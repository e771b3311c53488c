"""BASELINE.json's configuration 2 at size: example_formula.input's model (formula mode, a = 0.9, camera at r = 1000,
ray_max_steps = 7000) with a 512^2 camera, one MI355X. Prints per-kernel times; `python tools/gpu_formula_frame.py [exact|tolerant] [reps]`."""
import json
import os
os.environ.setdefault("BLACKLIGHT_AMD_ARITHMETIC", "exact")   # (a context starts in this tier; the tool names the tolerant one where it wants it)
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import blacklight_amd as bl
import golden_util as gu

tier = sys.argv[1] if len(sys.argv) > 1 else "exact"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
fx, params, _ = gu.load_case("formula_dp")
params = dict(params, camera_resolution=512)
with bl.Context(bl.Params.from_dict(params)) as ctx:
    ctx.set_geodesic_reuse(False)   # a measurement of whole renders: every one integrates its geodesics
    ctx.set_arithmetic(tier)
    ctx.render()
    best, st = 1e30, None
    for _ in range(reps):
        t0 = time.perf_counter()
        out = ctx.render()
        dt = time.perf_counter() - t0
        if dt < best:
            best, st = dt, out["stats"]
    print(json.dumps(dict(tier=tier, tier_ran="tolerant" if st.arithmetic == 1 else "exact", seconds=best, mrays_per_s=512 * 512 / best / 1e6,
                          samples_per_ray=st.n_samples / (512 * 512), max_sample_num=st.max_sample_num, n_flagged=st.n_flagged,
                          ms_geodesic=round(st.ms_geodesic, 2), ms_shade=round(st.ms_shade, 2), ms_transfer=round(st.ms_transfer, 2), chunks=st.n_chunks)))
