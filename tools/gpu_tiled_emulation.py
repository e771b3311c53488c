"""EMULATION of `bench.py --gpus N` (one 1024^2 frame in 32 x 32 tiles dealt centre-first round-robin over N ranks) on ONE GPU:
every rank's share is rendered by the same library calls a real rank makes (Context.render_device into HBM tensors, then the
gather of the padded share through an RCCL group - of world size 1 here, which costs the collective's launch and copy but no
link time - and, for rank 0, the de-tiling of the gathered (world, n_padded) buffer into the frame exactly as bench.py's
gather_image() does it: ShareLayout.detile through the cached permutation), one rank after the other. A frame's time is the slowest rank's. Per rank: three warm-up renders, then NINE timed
ones back to back - the clock stays where a rank's own frame loop would hold it - and the MEDIAN is what counts (minimum and
maximum beside it). No 8-GPU node was available to this builder; everything this prints is an estimate from one GPU.

    python3 tools/gpu_tiled_emulation.py gpurun_out/tiled_emulation.json      WORLDS=1,2,4,8  ARITH=tolerant  REPS=9
(the result goes to the named file: RCCL prints its banner on stdout)
"""
import json
import os
os.environ.setdefault("BLACKLIGHT_AMD_ARITHMETIC", "exact")   # (a context starts in this tier; the tool names the tolerant one where it wants it)
import statistics
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import blacklight_amd as bl
from blacklight_amd import distributed as bd, mock
import bench

res = 1024
reps = int(os.environ.get("REPS", "9"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
grid = mock.generate(n_r=256, n_th=256, n_ph=256)
p = dict(bench.WORKLOAD)
out = {"note": "emulated on one GPU, rank by rank (see the docstring of tools/gpu_tiled_emulation.py); median of %d renders per rank, gather of the share through "
               "a world-size-1 RCCL group included" % reps}
with bl.Context(bl.Params.from_dict(p)) as ctx:
    ctx.set_geodesic_reuse(False)   # a measurement of whole renders: every one integrates its geodesics
    ctx.set_grid(grid)
    ctx.set_arithmetic(os.environ.get("ARITH", "tolerant"))
    for world in [int(w) for w in os.environ.get("WORLDS", "1,2,4,8").split(",")]:
        ranks = []
        layout = bd.frame_layout(res, world, bench.TILE) if world > 1 else None
        n_padded = layout.n_padded if world > 1 else res * res
        ctx.follow_torch_stream("cuda")
        for rank in range(world):
            pixels = layout.pixels[rank] if world > 1 else None
            n_rays = res * res if pixels is None else int(pixels.size)
            image = torch.zeros((1, n_padded), dtype=torch.float64, device="cuda")
            gathered = torch.empty((1, n_padded), dtype=torch.float64, device="cuda")
            everyone = torch.zeros((world, n_padded), dtype=torch.float64, device="cuda")   # rank 0's receive buffer
            torch.cuda.synchronize()
            times, kernel = [], []
            for rep in range(3 + reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                st = ctx.render_device(image.data_ptr(), n_rays, pixel_map=pixels)
                dist.all_gather_into_tensor(gathered, image)
                if rank == 0 and world > 1:   # rank 0 alone de-tiles, every frame, as in bench.py
                    frame = layout.detile(everyone, 1)
                torch.cuda.synchronize()
                if rep >= 3:
                    times.append(1e3 * (time.perf_counter() - t0))
                    kernel.append((st.ms_geodesic, st.ms_shade, st.ms_transfer, st.ms_wall))
            mid = sorted(range(reps), key=lambda i: times[i])[reps // 2]
            ranks.append(dict(rank=rank, rays=n_rays, median_ms=statistics.median(times), min_ms=min(times), max_ms=max(times),
                              geodesic=kernel[mid][0], shade=kernel[mid][1], transfer=kernel[mid][2], kernels_wall=kernel[mid][3],
                              chunks=st.n_chunks, samples=st.n_samples))
        worst = max(r["median_ms"] for r in ranks)
        out[f"world_{world}"] = dict(frame_ms_median=worst, mrays_per_s=res * res / worst / 1e3, frame_ms_best=max(r["min_ms"] for r in ranks),
                                     frame_ms_worst=max(r["max_ms"] for r in ranks), ranks=ranks)
    base = out["world_1"]["frame_ms_median"] if "world_1" in out else None   # (strong-scaling efficiency needs the one-rank frame)
    for world in [int(w) for w in os.environ.get("WORLDS", "1,2,4,8").split(",")]:
        out[f"world_{world}"]["strong_scaling_efficiency"] = None if base is None else base / (world * out[f"world_{world}"]["frame_ms_median"])
dist.destroy_process_group()
with open(sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout", "w") as f:
    json.dump(out, f, indent=1)
for key, value in out.items():
    if key.startswith("world"):
        print(key, "median %.2f ms (best %.2f, worst %.2f) = %.1f Mrays/s, strong-scaling efficiency %s; slowest rank's kernels: geodesic %.2f coefficient %.2f transfer %.2f"
              % (value["frame_ms_median"], value["frame_ms_best"], value["frame_ms_worst"], value["mrays_per_s"], value["strong_scaling_efficiency"],
                 *max(((r["geodesic"], r["shade"], r["transfer"]) for r in value["ranks"]), key=lambda k: sum(k))), file=sys.stderr)
