#!/usr/bin/env python3
"""Host time of one bl_render of an eighth-frame share (131 072 rays of the 1024^2 benchmark camera, rank 0's tiles) against the time its
kernels take: what the call costs outside them.   gpurun -- 'python3 tools/gpu_share_overhead.py'"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench                                   # noqa: E402
import blacklight_amd as bl                    # noqa: E402
from blacklight_amd import distributed as bd   # noqa: E402
from blacklight_amd import mock                # noqa: E402

params = dict(bench.WORKLOAD)
grid = mock.generate(n_r=256, n_th=256, n_ph=256)
with bl.Context(bl.Params.from_dict(params)) as ctx:
    ctx.set_geodesic_reuse(False)   # a measurement of whole renders: every one integrates its geodesics
    ctx.set_grid(grid)
    ctx.set_arithmetic("tolerant")
    for world in (8, 1):
        pixels = bd.tile_pixels(1024, 0, world, 32)
        image = torch.empty((1, pixels.size), dtype=torch.float64, device="cuda")
        for _ in range(3):
            ctx.render_device(image.data_ptr(), pixels.size, pixel_map=pixels)
        torch.cuda.synchronize()
        host, kernels, wall = [], [], []
        for _ in range(20):
            t0 = time.perf_counter()
            st = ctx.render_device(image.data_ptr(), pixels.size, pixel_map=pixels)
            host.append(1000.0 * (time.perf_counter() - t0))
            kernels.append(st.ms_geodesic + st.ms_locate + st.ms_shade + st.ms_transfer)
            wall.append(st.ms_wall)
        print(f"world {world}: host {np.median(host):.3f} ms per call, event wall {np.median(wall):.3f}, kernels {np.median(kernels):.3f}; "
              f"outside the kernels {np.median(host) - np.median(kernels):.3f} ms", flush=True)
