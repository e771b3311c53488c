"""Multi-GPU sharding of the camera: one process per GPU (torch.distributed; backend "nccl" is
RCCL on ROCm, "gloo" in the CPU tests), grid replicated on every GPU, rays independent.

The reference has no distributed layer (single process + OpenMP). Rays are independent through
the whole path, so the camera is cut into square tiles that are dealt block-cyclically to the
ranks - cost per ray varies ~4x across the image (sample counts 420 ... 1702), so contiguous bands
would be imbalanced - and the only communication is the final gather of the image rows on rank 0.
"""
import numpy as np

TILE = 32


def tile_pixels(resolution, rank, world, tile=TILE):
    """Pixel indices (m = m2 * resolution + m1, as the reference's camera, camera.cpp:393-396) of the
    tiles owned by `rank`. Tile-major, row-major inside a tile, which is also the order the geodesic
    kernel wants (each wave starts as a compact 2-D patch)."""
    if resolution % tile != 0:
        raise ValueError("camera_resolution must be a multiple of the tile size")
    tiles_per_side = resolution // tile
    # All tiles sorted by distance from the image centre, then dealt round-robin: every rank gets the same
    # mix of long (central) and short (peripheral) rays, and traces its long ones first, so that its
    # persistent geodesic waves end on short rays (same reason as the library's centre-first tile order).
    all_ids = np.arange(tiles_per_side * tiles_per_side)
    centre = 0.5 * (tiles_per_side - 1)
    dist2 = (all_ids // tiles_per_side - centre) ** 2 + (all_ids % tiles_per_side - centre) ** 2
    ids = all_ids[np.argsort(dist2, kind="stable")][rank::world]
    ty, tx = ids // tiles_per_side, ids % tiles_per_side
    yy, xx = np.meshgrid(np.arange(tile), np.arange(tile), indexing="ij")
    m2 = (ty[:, None, None] * tile + yy[None]).reshape(-1)
    m1 = (tx[:, None, None] * tile + xx[None]).reshape(-1)
    return (m2 * resolution + m1).astype(np.int32)


def padded_count(resolution, world, tile=TILE):
    """Rays per rank after padding to the largest share (gather needs equal sizes)."""
    tiles = (resolution // tile) ** 2
    return ((tiles + world - 1) // world) * tile * tile


def gather_rows(local, dst=0):
    """Gather a (n_q, n_local) tensor from every rank on `dst` (RCCL gather over xGMI on GPUs)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    if dist.get_rank() == dst:
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.gather(local, parts, dst=dst)
        return parts
    dist.gather(local, None, dst=dst)
    return None


def assemble(parts, resolution, tile=TILE):
    """Rank 0: scatter the gathered (n_q, n_local) blocks back into (n_q, resolution**2)."""
    import torch
    world = len(parts)
    n_q = parts[0].shape[0]
    image = torch.empty((n_q, resolution * resolution), dtype=parts[0].dtype, device=parts[0].device)
    for rank, part in enumerate(parts):
        pixels = torch.from_numpy(tile_pixels(resolution, rank, world, tile).astype(np.int64)).to(part.device)
        image[:, pixels] = part[:, : pixels.numel()]
    return image
