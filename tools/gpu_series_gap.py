"""Does the coefficient kernel of a reused frame run slower after an idle gap (bl_set_grid on the host between two frames) than
back to back? Frames over the resident records with (a) nothing between them, (b) a sleep, (c) bl_set_grid.   gpurun -- python tools/gpu_series_gap.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import blacklight_amd as bl
from blacklight_amd import mock
grid = mock.generate(n_r=256, n_th=256, n_ph=256)
n = 1024 * 1024
dev = torch.device("cuda", 0)
image = torch.zeros((1, n), dtype=torch.float64, device=dev)
with bl.Context(bl.Params.from_dict(dict(bench.WORKLOAD)), device=0) as ctx:
    ctx.set_grid(grid)
    ctx.render_device(image.data_ptr(), n)
    for name, between in (("back to back", lambda: None), ("sleep 0.3 s", lambda: time.sleep(0.3)), ("bl_set_grid", lambda: ctx.set_grid(grid)),
                          ("back to back", lambda: None)):
        shade, wall = [], []
        for _ in range(6):
            between()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = ctx.render_device(image.data_ptr(), n)
            torch.cuda.synchronize()
            wall.append(1000.0 * (time.perf_counter() - t0))
            shade.append(st.ms_shade)
            assert st.geodesics_reused == 1
        print(f"{name:14s} shade ms {' '.join(f'{v:.2f}' for v in shade)} | render ms {' '.join(f'{v:.2f}' for v in wall)}", flush=True)
