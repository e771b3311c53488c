"""Multi-GPU sharding of the camera: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm,
"gloo" in the CPU tests and when several ranks share one GPU), grid replicated on every GPU, rays independent.

The reference has no distributed layer (single process + OpenMP). Rays are independent through the whole path, so the
camera is cut into square tiles that are dealt block-cyclically to the ranks - cost per ray varies ~4x across the image
(sample counts 420 ... 1702), so contiguous bands would be imbalanced - and the only communication is, per adaptive
level:

  * a gather of the level's image rows (and sample counts, flags, camera rows, renderings) on rank 0;
  * two reductions, max of max_sample_num and sum of the flagged-ray count, so that rank 0 can raise the reference's
    "N out of M geodesics terminate unexpectedly." with the totals of the level (geodesics.cpp:389-394);
  * a broadcast of the next level's block list: rank 0 alone evaluates the refinement criteria on the gathered image
    (bl_adaptive_refine; radiation_adaptive.cpp:19-139) and lists the children in AugmentCamera's order
    (camera.cpp:445-458); every rank then traces the blocks  rank, rank + world, ...  of that list.

`render_tiled` is one level-0 frame (what bench.py times); `render_adaptive` is the reference's
do { Integrate; AddGeodesics } while (!done) loop (blacklight.cpp:196-233) over all ranks, returning on rank 0 exactly
what `Context.render_adaptive` returns on one GPU, so `Context.write_output` writes the same file.
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


def default_tile(resolution, block_size=1):
    """Largest of 32, 16, 8, ... that divides the camera (and is a multiple of the adaptive block where there is one)."""
    tile = TILE
    while tile > 1 and (resolution % tile != 0 or tile % max(block_size, 1) != 0):
        tile //= 2
    return tile if resolution % tile == 0 and tile % max(block_size, 1) == 0 else resolution


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
    """Rank 0: scatter the gathered (n_q, n_padded) blocks back into (n_q, resolution**2)."""
    import torch
    world = len(parts)
    n_q = parts[0].shape[0]
    image = torch.empty((n_q, resolution * resolution), dtype=parts[0].dtype, device=parts[0].device)
    for rank, part in enumerate(parts):
        pixels = torch.from_numpy(tile_pixels(resolution, rank, world, tile).astype(np.int64)).to(part.device)
        image[:, pixels] = part[:, : pixels.numel()]
    return image


class Comm:
    """The three collectives of the protocol on torch.distributed's default group. `device`: where the tensors that
    travel live - the rank's GPU for nccl (RCCL over xGMI), the CPU for gloo."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        self.device = device

    def gather_columns(self, local, n_padded):
        """local: numpy (rows, n_local). Rank 0 gets the list of every rank's (rows, n_padded) array (zero padded)."""
        torch = self.torch
        rows = local.shape[0]
        buf = torch.zeros((rows, n_padded), dtype=torch.from_numpy(local[:0].copy()).dtype, device=self.device)
        if local.shape[1]:
            buf[:, : local.shape[1]] = torch.from_numpy(np.ascontiguousarray(local)).to(self.device)
        parts = gather_rows(buf, dst=0)
        return None if parts is None else [p.cpu().numpy() for p in parts]

    def reduce_counts(self, max_value, sum_value):
        """(max over ranks of max_value, sum over ranks of sum_value), on every rank."""
        torch, dist = self.torch, self.dist
        t_max = torch.tensor([int(max_value)], dtype=torch.int64, device=self.device)
        t_sum = torch.tensor([int(sum_value)], dtype=torch.int64, device=self.device)
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(t_sum, op=dist.ReduceOp.SUM)
        return int(t_max.item()), int(t_sum.item())

    def broadcast_blocks(self, block_locs):
        """Rank 0's (n, 2) int32 block list to every rank (n = 0: the adaptive loop is complete)."""
        torch, dist = self.torch, self.dist
        count = torch.tensor([0 if block_locs is None else int(block_locs.shape[0])], dtype=torch.int64, device=self.device)
        dist.broadcast(count, src=0)
        n = int(count.item())
        data = torch.zeros((max(n, 1), 2), dtype=torch.int32, device=self.device)
        if self.rank == 0 and n:
            data[:] = torch.from_numpy(np.ascontiguousarray(block_locs, dtype=np.int32)).to(self.device)
        dist.broadcast(data, src=0)
        return data.cpu().numpy()[:n].copy()


ROW_KEYS = ("image", "sample_num", "sample_flags", "camera_pos", "camera_dir", "rendering")


def _as_rows(key, value):
    """Per-pixel outputs of Context.render as (rows, n_pixels) arrays."""
    if key in ("sample_num", "sample_flags"):
        return value.reshape(1, -1)
    if key in ("camera_pos", "camera_dir"):
        return np.ascontiguousarray(value.T)                       # (n, 4) -> (4, n)
    if key == "rendering":
        return value.reshape(-1, value.shape[-1])                   # (n_images, 3, n) -> (3 n_images, n)
    return value


def _from_rows(key, rows, like):
    if key in ("sample_num", "sample_flags"):
        return rows[0]
    if key in ("camera_pos", "camera_dir"):
        return np.ascontiguousarray(rows.T)
    if key == "rendering":
        return rows.reshape(like.shape[0], 3, -1)
    return rows


def render_level(ctx, comm, level=0, block_locs=None, want_camera=False, tile=None):
    """One adaptive level over all ranks. Level 0: the camera in tiles (tile_pixels); refined levels: blocks
    rank, rank + world, ... of `block_locs`. Returns on rank 0 the dict Context.render returns for the whole level
    (image, sample_num, sample_flags, camera rows, rendering) plus max_sample_num / n_flagged of the level; None on
    the other ranks. The per-rank "geodesics terminate unexpectedly" warnings are replaced by one with the level's totals."""
    rank, world = comm.rank, comm.world
    bs = int(ctx.params.get("adaptive_block_size") or 1) if int(ctx.params.get("adaptive_max_level") or 0) > 0 else 1
    if level == 0:
        res = ctx.resolution
        tile = tile or default_tile(res, bs)
        pixels = tile_pixels(res, rank, world, tile)
        n_total = res * res
        n_padded = padded_count(res, world, tile)
        out = ctx.render(pixel_map=pixels, want_camera=want_camera) if pixels.size else None
        n_local = int(pixels.size)
    else:
        mine = np.ascontiguousarray(block_locs[rank::world], dtype=np.int32)
        n_blocks = int(block_locs.shape[0])
        n_total = n_blocks * bs * bs
        n_padded = ((n_blocks + world - 1) // world) * bs * bs
        out = ctx.render(level=level, block_locs=mine, want_camera=want_camera) if mine.shape[0] else None
        n_local = int(mine.shape[0]) * bs * bs
    if hasattr(ctx, "clear_warnings"):
        ctx.clear_warnings()   # per-rank counts; the level's totals are reported below
    max_num = out["stats"].max_sample_num if out is not None else 0
    n_flagged = out["stats"].n_flagged if out is not None else 0
    max_num, n_flagged = comm.reduce_counts(max_num, n_flagged)
    # which rows exist is the same on every rank (it follows from the parameters); a rank without rays sends zeros
    template = out if out is not None else ctx.render_template(want_camera)
    result = dict(max_sample_num=max_num, n_flagged=n_flagged, n_rays=n_total) if rank == 0 else None
    for key in ROW_KEYS:
        if template.get(key) is None:
            continue
        like = template[key]
        rows = _as_rows(key, like)
        if out is None:
            rows = np.zeros((rows.shape[0], 0), dtype=rows.dtype)
        parts = comm.gather_columns(rows[:, :n_local], n_padded)
        if rank != 0:
            continue
        full = np.empty((rows.shape[0], n_total), dtype=rows.dtype)
        for r, part in enumerate(parts):
            if level == 0:
                where = tile_pixels(ctx.resolution, r, world, tile).astype(np.int64)
            else:
                blocks = np.arange(r, n_total // (bs * bs), world, dtype=np.int64)
                where = (blocks[:, None] * (bs * bs) + np.arange(bs * bs)[None, :]).reshape(-1)
            full[:, where] = part[:, : where.size]
        result[key] = _from_rows(key, full, like)
    if rank == 0:
        for key in ROW_KEYS:
            result.setdefault(key, None)
        if n_flagged > 0:   # geodesics.cpp:389-394, with the totals of the level
            result["warning"] = f"Warning: {n_flagged} out of {n_total} geodesics terminate unexpectedly.\n"
    return result


def render_tiled(ctx, comm, want_camera=False, tile=None):
    """One full frame (level 0) over all ranks; the image lands on rank 0."""
    return render_level(ctx, comm, 0, None, want_camera, tile)


def render_adaptive(ctx, comm=None, want_camera=False, tile=None):
    """The reference's adaptive loop over all ranks. Rank 0 returns the list of per-level dicts Context.render_adaptive
    returns on one GPU (block_locs, refinement_flags included) and the concatenated warnings; other ranks (None, "")."""
    comm = comm or Comm()
    rank = comm.rank
    warnings = ""
    levels = []
    first = render_level(ctx, comm, 0, None, want_camera, tile)
    if rank == 0:
        first["block_locs"] = None
        warnings += first.pop("warning", "")
        levels.append(first)
    max_level = int(ctx.params.get("adaptive_max_level") or 0)
    level = 0
    block_locs = None
    while max_level > 0:
        nxt = None
        if rank == 0:
            flags, nxt = ctx.adaptive_refine(level, levels[level]["image"], levels[level]["block_locs"])
            levels[level]["refinement_flags"] = flags
        block_locs = comm.broadcast_blocks(nxt)
        if block_locs.shape[0] == 0:
            break
        level += 1
        out = render_level(ctx, comm, level, block_locs, want_camera, tile)
        if rank == 0:
            out["block_locs"] = block_locs
            warnings += out.pop("warning", "")
            levels.append(out)
    return (levels, warnings) if rank == 0 else (None, "")
