#!/usr/bin/env python3
"""Polarized 1024^2 frame (bench.py's polarized1024 workload) with one scratch set (the default) and with bl_set_overlap's two sets of half
the budget, where the geodesic stage of chunk c + 1 runs beside the shading of chunk c: ms per frame, chunks, kernel sums.
    gpurun -- 'python3 tools/gpu_polarized_two_sets.py [resolution]'"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench                      # noqa: E402
import blacklight_amd as bl       # noqa: E402
from blacklight_amd import mock   # noqa: E402

res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
params = dict(bench.WORKLOAD, camera_resolution=res, image_polarization=True, image_tau=True)
grid = mock.generate(n_r=256, n_th=256, n_ph=256)
with bl.Context(bl.Params.from_dict(params)) as ctx:
    ctx.set_geodesic_reuse(False)   # a measurement of whole renders: every one integrates its geodesics
    ctx.set_grid(grid)
    ctx.set_arithmetic("tolerant")
    reference = None
    for overlap in (False, True, False, True):
        ctx.set_overlap(overlap)
        ctx.render()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            out = ctx.render()
        ms = 1000.0 * (time.perf_counter() - t0) / n
        st = out["stats"]
        if reference is None:
            reference = out["image"]
        same = bool((out["image"].view("u8") == reference.view("u8")).all())
        print(f"two sets {overlap}: {ms:.1f} ms per frame, {st.n_chunks} chunks, geodesic {st.ms_geodesic:.1f} shade {st.ms_shade:.1f} transfer {st.ms_transfer:.1f} "
              f"wall {st.ms_wall:.1f}; same bits as the first render: {same}", flush=True)
