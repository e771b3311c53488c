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
import functools

import numpy as np

TILE = 32


class ShareLayout:
    """How the rays of one level are dealt to the ranks, and the way back: `where[r]` = destination pixel of every ray of
    rank r, in the order the rank traces them. Built once (level 0: cached per (resolution, world, tile) by `frame_layout`;
    a refined level: once per level from its block list), vectorised over all ranks; nothing here is recomputed per
    frame, per output or per rank. The inverse - for every pixel of the level, where its value sits in the gathered buffer -
    lives on the device the gathered buffers live on (`_index`), so that de-tiling is index_select on that device:
    no numpy, no upload, no loop over ranks in the frame loop."""
    builds = 0   # how many layouts were ever built (tests: a second frame builds none)

    def __init__(self, where, n_total, n_padded):
        ShareLayout.builds += 1
        self.world = len(where)
        self.n_total, self.n_padded = int(n_total), int(n_padded)
        self.pixels = where
        self.counts = [int(w.size) for w in where]
        self.even = all(c == self.n_padded for c in self.counts)   # every share fills its padded buffer: rows keep one stride
        n_rays = sum(self.counts)
        if n_rays != self.n_total:
            raise ValueError("the shares do not cover the level")
        dest = np.concatenate(where).astype(np.int64) if self.world > 1 else np.asarray(where[0], dtype=np.int64)
        rank_of_ray = np.repeat(np.arange(self.world, dtype=np.int64), self.counts)
        slot_of_ray = np.arange(n_rays, dtype=np.int64) - np.repeat(np.cumsum([0] + self.counts[:-1]), self.counts)
        # for pixel m of the level: the rank that traced it, its position in that rank's share, the length of that share
        self._rank = np.empty(self.n_total, dtype=np.int64)
        self._slot = np.empty(self.n_total, dtype=np.int64)
        self._rank[dest] = rank_of_ray
        self._slot[dest] = slot_of_ray
        self._cache = {}

    def _index(self, device):
        """Per pixel of the level, as int64 tensors on `device` (uploaded once): its place rank * n_padded + slot in buffers whose
        rows are n_padded apart and, where the shares differ in length, (rank * n_padded, slot, length of the rank's share)."""
        import torch
        key = str(device)
        if key not in self._cache:
            padded = torch.from_numpy(self._rank * self.n_padded + self._slot).to(device)
            packed = None
            if not self.even:
                count_of = np.asarray(self.counts, dtype=np.int64)[self._rank]
                packed = tuple(torch.from_numpy(v).to(device) for v in (self._rank * self.n_padded, self._slot, count_of))
            self._cache[key] = (padded, packed)
        return self._cache[key]

    def detile(self, gathered, rows, ray_major=False, packed=True):
        """gathered: (world, rows * n_padded), rank r's flat buffer in row r. packed (what bl_render leaves when it is given
        counts[r] rays): `rows` rows of counts[r] values back to back in front of the buffer; not packed: rows n_padded apart;
        ray_major: counts[r] rays of `rows` values each. Returns (rows, n_total) - or (n_total, rows) - on the same device."""
        import torch
        padded, uneven = self._index(gathered.device)
        if ray_major:
            return gathered.reshape(self.world * self.n_padded, rows).index_select(0, padded)
        if uneven is None or not packed or rows == 1:
            by_row = gathered.reshape(self.world, rows, self.n_padded)
            by_row = by_row[0] if self.world == 1 else by_row.permute(1, 0, 2).reshape(rows, self.world * self.n_padded)
            return by_row.index_select(1, padded)
        # shares of different lengths: row q of rank r starts at q * counts[r] of its buffer
        rank_base, slot, count_of = uneven
        flat = gathered.reshape(-1)
        out = torch.empty((rows, self.n_total), dtype=gathered.dtype, device=gathered.device)
        first = rank_base * rows + slot
        for q in range(rows):
            torch.index_select(flat, 0, first + q * count_of, out=out[q])
        return out


def _tile_shares(resolution, world, tile):
    """Pixel indices (m = m2 * resolution + m1, as the reference's camera, camera.cpp:393-396) of every rank's tiles:
    tile-major, row-major inside a tile, which is also the order the geodesic kernel wants (each wave starts as a compact
    2-D patch)."""
    if resolution % tile != 0:
        raise ValueError("camera_resolution must be a multiple of the tile size")
    tiles_per_side = resolution // tile
    # All tiles sorted by distance from the image centre, then dealt round-robin: every rank gets the same
    # mix of long (central) and short (peripheral) rays, and traces its long ones first, so that its
    # persistent geodesic waves end on short rays (same reason as the library's centre-first tile order).
    all_ids = np.arange(tiles_per_side * tiles_per_side)
    centre = 0.5 * (tiles_per_side - 1)
    dist2 = (all_ids // tiles_per_side - centre) ** 2 + (all_ids % tiles_per_side - centre) ** 2
    order = all_ids[np.argsort(dist2, kind="stable")]
    yy, xx = np.meshgrid(np.arange(tile), np.arange(tile), indexing="ij")
    shares = []
    for rank in range(world):
        ids = order[rank::world]
        ty, tx = ids // tiles_per_side, ids % tiles_per_side
        m2 = (ty[:, None, None] * tile + yy[None]).reshape(-1)
        m1 = (tx[:, None, None] * tile + xx[None]).reshape(-1)
        pixels = (m2 * resolution + m1).astype(np.int32)
        pixels.flags.writeable = False
        shares.append(pixels)
    return shares


@functools.lru_cache(maxsize=16)
def frame_layout(resolution, world, tile=TILE):
    """The level-0 layout of a camera (cached: a frame loop asks for it every frame and gets the same object)."""
    return ShareLayout(_tile_shares(resolution, world, tile), resolution * resolution, padded_count(resolution, world, tile))


def block_layout(n_blocks, block_size, world):
    """A refined level: blocks rank, rank + world, ... of the level's block list, block_size^2 pixels each."""
    per_block = block_size * block_size
    within = np.arange(per_block, dtype=np.int64)[None, :]
    where = [(np.arange(r, n_blocks, world, dtype=np.int64)[:, None] * per_block + within).reshape(-1) for r in range(world)]
    return ShareLayout(where, n_blocks * per_block, ((n_blocks + world - 1) // world) * per_block)


def tile_pixels(resolution, rank, world, tile=TILE):
    """Pixel indices of the tiles owned by `rank` (read-only view of the cached layout's array)."""
    return frame_layout(resolution, world, tile).pixels[rank]


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


def gather_shares(local, dst=0, buffer=None):
    """Gather one flat tensor of the same length from every rank into ONE (world, n) tensor on `dst` (RCCL gather over xGMI
    on GPUs; the receive buffers are the rows of that tensor, nothing is copied afterwards). `buffer`: a (world, n) tensor
    to receive into (a frame loop keeps one). Returns it on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    flat = local.reshape(-1)
    if dist.get_rank() == dst:
        if buffer is None or buffer.shape != (world, flat.numel()) or buffer.dtype != flat.dtype or buffer.device != flat.device:
            buffer = torch.empty((world, flat.numel()), dtype=flat.dtype, device=flat.device)
        dist.gather(flat, list(buffer.unbind(0)), dst=dst)
        return buffer
    dist.gather(flat, None, dst=dst)
    return None


def gather_rows(local, dst=0):
    """Gather a (n_q, n_local) tensor from every rank on `dst`: the list of the ranks' tensors (views of one buffer)."""
    gathered = gather_shares(local, dst)
    return None if gathered is None else [part.view(local.shape) for part in gathered.unbind(0)]


def assemble(parts, resolution, tile=TILE):
    """Rank 0: the gathered (n_q, n_padded) shares of a frame - a list of them, or one (world, n_q * n_padded) tensor as
    gather_shares returns it - back into (n_q, resolution**2), on the device they are on, through the cached layout."""
    import torch
    if isinstance(parts, (list, tuple)):
        world, n_q = len(parts), parts[0].shape[0]
        gathered = torch.stack([part.reshape(-1) for part in parts])
    else:
        world, gathered = parts.shape[0], parts
        n_q = gathered.shape[1] // padded_count(resolution, world, tile)
    return frame_layout(resolution, world, tile).detile(gathered, n_q, packed=False)


class RankError(RuntimeError):
    """A rank's render failed: raised on EVERY rank (the others would otherwise block in the next collective until the
    backend's timeout), with the failing ranks' messages."""


class Comm:
    """The collectives of the protocol on torch.distributed's default group. `device`: where the tensors that
    travel live - the rank's GPU for nccl (RCCL over xGMI), the CPU for gloo."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        self.device = device

    @property
    def on_gpu(self):
        return self.device.type == "cuda"

    def agree(self, error):
        """Every rank passes its error text (None: none). If any rank failed, every rank raises RankError with all
        the texts - before any data collective, so that nobody waits for a rank that has left."""
        torch, dist = self.torch, self.dist
        flag = torch.tensor([0 if error is None else 1], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()) == 0:
            return
        texts = [None] * self.world
        dist.all_gather_object(texts, error)
        raise RankError("; ".join(f"rank {r}: {t.strip()}" for r, t in enumerate(texts) if t is not None))

    def gather_flat(self, local, dst=0):
        """local: a flat tensor on self.device, the same length on every rank. Rank dst gets them as ONE (world, n) tensor - a view
        of the one receive buffer this object keeps per element type, grown to the largest gather seen (a frame loop allocates
        nothing after its first frame; refined levels, whose length differs from level to level and frame to frame, share it
        instead of leaving a buffer each behind). The view is valid until the next gather of that type."""
        torch = self.torch
        pools = self.__dict__.setdefault("_gather_pools", {})
        view = None
        if self.dist.get_rank() == dst:
            need = self.world * local.numel()
            pool = pools.get(local.dtype)
            if pool is None or pool.numel() < need or pool.device != local.device:
                pool = pools[local.dtype] = torch.empty(need, dtype=local.dtype, device=local.device)
            view = pool[:need].view(self.world, local.numel())
        return gather_shares(local, dst=dst, buffer=view)

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
            data[:] = torch.as_tensor(np.ascontiguousarray(block_locs, dtype=np.int32), device=self.device)
        dist.broadcast(data, src=0)
        return data.cpu().numpy()[:n].copy()


class EmulatedComm:
    """`world` ranks emulated by ONE process on one GPU: render_level / render_adaptive run every rank's share one after
    the other through the same tiling, padding, buffer layouts and reassembly as a real group, with the collectives
    replaced by their serial meaning. For tests and strong-scaling estimates at sizes where a box allows no more than a
    few processes on its GPU; nothing here touches torch.distributed."""
    emulated = True

    def __init__(self, world, device=None):
        import torch
        self.torch = torch
        self.rank, self.world = 0, int(world)
        self.device = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))

    @property
    def on_gpu(self):
        return self.device.type == "cuda"

    def broadcast_blocks(self, block_locs):
        return np.zeros((0, 2), dtype=np.int32) if block_locs is None else np.ascontiguousarray(block_locs, dtype=np.int32)


# per-pixel outputs of a level, as flat (rows x rays) buffers: name, rows per ray as a function of the context, dtype
def _row_specs(ctx, want_camera):
    import torch
    specs = [("image", ctx.num_quantities, torch.float64), ("sample_num", 1, torch.int32), ("sample_flags", 1, torch.uint8)]
    if want_camera:
        specs += [("camera_pos", 4, torch.float64), ("camera_dir", 4, torch.float64)]
    if ctx.num_render_images > 0:
        specs.append(("rendering", 3 * ctx.num_render_images, torch.float64))
    return specs


def _render_into_buffers(ctx, comm, specs, n_local, n_padded, level, pixels, blocks, want_camera):
    """This rank's share of a level into one flat buffer per output on comm.device, each rows x n_padded long with the
    share's rows x n_local values in front. On a GPU the library writes straight into the buffers (bl_render with device
    pointers): nothing crosses PCIe before the gather. Returns (buffers, stats)."""
    import torch
    buffers = {name: torch.zeros(rows * n_padded, dtype=dtype, device=comm.device) for name, rows, dtype in specs}
    if n_local == 0:
        return buffers, None
    if comm.on_gpu and hasattr(ctx, "render_device"):
        # torch.zeros queued its fill kernels on torch's current stream; the library renders on non-blocking streams of its own:
        # bl_set_caller_stream puts the render behind those fills (and behind the previous level's gathers) on the device
        ctx.follow_torch_stream(comm.device)
        ptr = {name: buffers[name].data_ptr() for name in buffers}
        stats = ctx.render_device(ptr["image"], n_local, level=level, pixel_map=pixels, block_locs=blocks,
                                  sample_num_ptr=ptr["sample_num"], sample_flags_ptr=ptr["sample_flags"],
                                  camera_pos_ptr=ptr.get("camera_pos", 0), camera_dir_ptr=ptr.get("camera_dir", 0),
                                  render_ptr=ptr.get("rendering", 0))
        return buffers, stats
    out = ctx.render(level=level, block_locs=blocks, pixel_map=pixels, want_camera=want_camera)
    for name, rows, dtype in specs:
        value = out[name]
        # the library's layouts: image / rendering rows x rays, camera rows rays x 4, counts and flags one row
        flat = torch.from_numpy(np.ascontiguousarray(value).reshape(-1))
        buffers[name][: flat.numel()] = flat
    return buffers, out["stats"]


def render_level(ctx, comm, level=0, block_locs=None, want_camera=False, tile=None):
    """One adaptive level over all ranks. Level 0: the camera in tiles (tile_pixels); refined levels: blocks
    rank, rank + world, ... of `block_locs`. Returns on rank 0 the dict Context.render returns for the whole level
    (image, sample_num, sample_flags, camera rows, rendering) plus max_sample_num / n_flagged of the level; None on
    the other ranks. The per-rank "geodesics terminate unexpectedly" warnings are replaced by one with the level's totals.
    A failure on any rank (a refusal that depends on the rank's own rays, a full record buffer, no memory) is agreed on
    before the first data collective and raised on every rank (RankError)."""
    rank, world = comm.rank, comm.world
    bs = int(ctx.params.get("adaptive_block_size") or 1) if int(ctx.params.get("adaptive_max_level") or 0) > 0 else 1
    if level == 0:
        layout = frame_layout(ctx.resolution, world, tile or default_tile(ctx.resolution, bs))   # cached: built once per camera
    else:
        layout = block_layout(int(block_locs.shape[0]), bs, world)                                # once per level
    counts, n_total, n_padded = layout.counts, layout.n_total, layout.n_padded

    def share_of(r):
        if level == 0:
            return layout.pixels[r], None
        return None, np.ascontiguousarray(block_locs[r::world], dtype=np.int32)

    specs = _row_specs(ctx, want_camera)
    emulated = getattr(comm, "emulated", False)
    if emulated:
        # every rank's share in turn, into buffers of its own: what the gathers below would deliver
        shares, max_num, n_flagged = [], 0, 0
        for r in range(world):
            pixels_r, blocks_r = share_of(r)
            buffers_r, stats_r = _render_into_buffers(ctx, comm, specs, counts[r], n_padded, level, pixels_r, blocks_r, want_camera)
            shares.append(buffers_r)
            if stats_r is not None:
                max_num, n_flagged = max(max_num, stats_r.max_sample_num), n_flagged + stats_r.n_flagged
        if hasattr(ctx, "clear_warnings"):
            ctx.clear_warnings()
    else:
        n_local = counts[rank]
        pixels, blocks = share_of(rank)
        buffers, stats, error = None, None, None
        try:
            buffers, stats = _render_into_buffers(ctx, comm, specs, n_local, n_padded, level, pixels, blocks, want_camera)
        except Exception as failure:   # agreed on below: every rank raises, none is left waiting in a collective
            error = f"{type(failure).__name__}: {failure}"
        comm.agree(error)
        if hasattr(ctx, "clear_warnings"):
            ctx.clear_warnings()   # per-rank counts; the level's totals are reported below
        max_num = stats.max_sample_num if stats is not None else 0
        n_flagged = stats.n_flagged if stats is not None else 0
        max_num, n_flagged = comm.reduce_counts(max_num, n_flagged)
    result = dict(max_sample_num=max_num, n_flagged=n_flagged, n_rays=n_total) if rank == 0 else None
    import torch
    landed = {}
    for name, rows, dtype in specs:
        gathered = torch.stack([share[name] for share in shares]) if emulated else comm.gather_flat(buffers[name])
        if rank != 0:
            continue
        # de-tiled where the gather left it (the GPU for RCCL); the downloads of all outputs of the level are queued - into pinned
        # memory from torch's caching host allocator, without waiting - and waited for once, below
        ray_major = name in ("camera_pos", "camera_dir")          # the library writes rays x 4 there, rows x rays elsewhere
        whole = layout.detile(gathered, rows, ray_major=ray_major)
        if whole.is_cuda:
            host = torch.empty(whole.shape, dtype=whole.dtype, pin_memory=True)
            host.copy_(whole, non_blocking=True)
            whole = host
        landed[name] = whole
    if any(t.is_pinned() for t in landed.values()):
        torch.cuda.current_stream().synchronize()
    for name, rows, dtype in specs:
        if rank != 0:
            continue
        full = landed[name].numpy()
        if name in ("sample_num", "sample_flags"):
            result[name] = full[0]
        elif name == "rendering":
            result[name] = full.reshape(ctx.num_render_images, 3, -1)
        else:
            result[name] = full
    if rank == 0:
        for key in ROW_KEYS:
            result.setdefault(key, None)
        if n_flagged > 0:   # geodesics.cpp:389-394, with the totals of the level
            result["warning"] = f"Warning: {n_flagged} out of {n_total} geodesics terminate unexpectedly.\n"
    return result


ROW_KEYS = ("image", "sample_num", "sample_flags", "camera_pos", "camera_dir", "rendering")


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
        error = None
        if rank == 0:
            try:
                flags, nxt = ctx.adaptive_refine(level, levels[level]["image"], levels[level]["block_locs"])
                levels[level]["refinement_flags"] = flags
            except Exception as failure:   # (a bad image, out of memory, a library error: every rank has to hear of it)
                error = f"{type(failure).__name__}: {failure}"
        if hasattr(comm, "agree"):   # (rank 0 alone refines: the others would wait in the broadcast below for a rank that has left)
            comm.agree(error)
        elif error is not None:
            raise RankError(f"rank 0: {error}")
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
