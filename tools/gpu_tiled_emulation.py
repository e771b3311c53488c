"""Emulate `bench.py --mode tiled` ranks on one GPU: time each rank's share of one 1024^2 frame split
into 32x32 tiles dealt block-cyclically over `world` ranks (strong scaling estimate: max over ranks)."""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import blacklight_amd as bl
from blacklight_amd import distributed as bd, mock
import bench

res = 1024
grid = mock.generate(n_r=256, n_th=256, n_ph=256)
p = dict(bench.WORKLOAD)
out = {}
with bl.Context(bl.Params.from_dict(p)) as ctx:
    ctx.set_grid(grid)
    ctx.set_arithmetic(os.environ.get("ARITH", "tolerant"))
    if os.environ.get("OVERLAP"):
        ctx.set_overlap(True)
    if os.environ.get("SCRATCH_GB"):
        ctx.set_scratch_limit(int(float(os.environ["SCRATCH_GB"]) * 1e9))
    for world in [int(w) for w in os.environ.get("WORLDS", "1,2,4,8").split(",")]:
        times = []
        for rank in range(world):
            pixels = bd.tile_pixels(res, rank, world, bench.TILE) if world > 1 else None
            n_rays = res * res if pixels is None else int(pixels.size)
            image = torch.empty((1, n_rays), dtype=torch.float64, device="cuda")
            dt = 1.0e30
            for rep in range(6):   # the fastest of six: a rank's share is a few milliseconds, and the clock takes a render or two to settle
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                st_rep = ctx.render_device(image.data_ptr(), n_rays, pixel_map=pixels)
                torch.cuda.synchronize()
                if time.perf_counter() - t0 < dt:
                    dt, st = time.perf_counter() - t0, st_rep
            times.append(dict(rank=rank, ms=1e3 * dt, geodesic=st.ms_geodesic, locate=st.ms_locate, shade=st.ms_shade, transfer=st.ms_transfer, wall=st.ms_wall, chunks=st.n_chunks, emitted=st.n_samples_emitted, samples=st.n_samples))
        worst = max(t["ms"] for t in times)
        out[f"world_{world}"] = dict(max_ms=worst, mrays_per_s=res * res / worst / 1e3, ranks=times)
print(json.dumps(out, indent=1))
