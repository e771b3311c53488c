"""Multi-process path on CPU (gloo, world_size 2): tile ownership, padding, gather and reassembly
of blacklight_amd.distributed, i.e. everything of the N > 1 path that is not a kernel."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, resolution, tile, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from blacklight_amd import distributed as bd
    pixels = bd.tile_pixels(resolution, rank, world, tile)
    n_local = bd.padded_count(resolution, world, tile)
    # stand-in for the render: row 0 = pixel index, row 1 = -pixel index / 3
    local = torch.full((2, n_local), float("nan"), dtype=torch.float64)
    local[0, : pixels.size] = torch.from_numpy(pixels.astype(np.float64))
    local[1, : pixels.size] = torch.from_numpy(-pixels.astype(np.float64) / 3.0)
    parts = bd.gather_rows(local, dst=0)
    # scalar reductions used for the reference's warning text / geodesic_num_steps
    stats = torch.tensor([float(pixels.size), float(rank + 5)], dtype=torch.float64)
    total = stats.clone()
    dist.all_reduce(total, op=dist.ReduceOp.SUM)
    peak = stats.clone()
    dist.all_reduce(peak, op=dist.ReduceOp.MAX)
    if rank == 0:
        image = bd.assemble(parts, resolution, tile)
        torch.save({"image": image, "total": total, "peak": peak}, result_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("resolution,tile", [(64, 16), (96, 32)])
def test_tiled_gather_world2(tmp_path, resolution, tile):
    world = 2
    result_path = str(tmp_path / "result.pt")
    mp.spawn(_worker, args=(world, _free_port(), resolution, tile, result_path), nprocs=world, join=True)
    out = torch.load(result_path)
    want = torch.arange(resolution * resolution, dtype=torch.float64)
    assert torch.equal(out["image"][0], want)
    assert torch.equal(out["image"][1], -want / 3.0)
    assert out["total"][0].item() == resolution * resolution
    assert out["peak"][1].item() == 6.0


def test_tiles_partition_the_image():
    from blacklight_amd import distributed as bd
    for world in (1, 2, 3, 4, 8):
        seen = np.concatenate([bd.tile_pixels(128, r, world, 32) for r in range(world)])
        assert np.array_equal(np.sort(seen), np.arange(128 * 128))
        first = bd.tile_pixels(128, 0, world, 32)[:32 * 32].reshape(32, 32)
        # row-major inside a tile; rank 0's first tile is one of the four central ones (centre first)
        assert np.array_equal(first[0], first[0, 0] + np.arange(32)) and first[1, 0] == first[0, 0] + 128
        assert first[0, 0] // 128 in (32, 64) and first[0, 0] % 128 in (32, 64)
        # every rank gets the same number of tiles (+-1), dealt so that each holds central and peripheral ones
        counts = [bd.tile_pixels(128, r, world, 32).size // (32 * 32) for r in range(world)]
        assert max(counts) - min(counts) <= 1
