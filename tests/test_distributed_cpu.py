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


ADAPTIVE_STUB = dict(
    model_type="formula", num_threads=1, output_format="npz", output_file="stub.npz", output_camera="true",
    checkpoint_geodesic_save="false", checkpoint_geodesic_load="false", formula_mass=6.0e11, formula_spin=0.9, formula_r0=10.0,
    formula_h=0.0, formula_l0=0.0, formula_q=0.5, formula_nup=2.3e11, formula_cn0=3.0e-18, formula_alpha=-3.0, formula_a=0.0,
    formula_beta=2.5, camera_type="plane", camera_r=100.0, camera_th=60.0, camera_ph=0.0, camera_urn=0.0, camera_uthn=0.0,
    camera_uphn=0.0, camera_k_r=1.0, camera_k_th=0.0, camera_k_ph=0.0, camera_rotation=0.0, camera_width=30.0,
    camera_resolution=48, ray_flat="false", ray_terminate="additive", ray_factor=5.0e-4, ray_integrator="dp", ray_step=0.01,
    ray_max_steps=2000, ray_max_retries=20, ray_tol_abs=1.0e-8, ray_tol_rel=1.0e-8, image_light="true",
    image_num_frequencies=2, image_frequency_start=1.0e11, image_frequency_end=3.0e11, image_frequency_spacing="log",
    image_normalization="camera", image_time="false", image_length="false", image_lambda="false", image_emission="false",
    image_tau="false", image_lambda_ave="false", image_emission_ave="false", image_tau_int="false", image_crossings="false",
    render_num_images=0, adaptive_max_level=2, adaptive_block_size=8, adaptive_frequency_num=1, adaptive_val_cut=0.0,
    adaptive_val_frac=-1.0, adaptive_abs_grad_cut=0.0, adaptive_abs_grad_frac=-1.0, adaptive_rel_grad_cut=0.25,
    adaptive_rel_grad_frac=0.2, adaptive_abs_lapl_cut=0.0, adaptive_abs_lapl_frac=-1.0, adaptive_rel_lapl_cut=0.5,
    adaptive_rel_lapl_frac=0.2, adaptive_num_regions=0, cut_omit_near="false", cut_omit_far="false", cut_omit_in=-1.0,
    cut_omit_out=-1.0, cut_midplane_theta=0.0, cut_midplane_z=0.0, cut_plane="false", fallback_nan="true",
)


@pytest.mark.parametrize("world", [2, 3])
def test_adaptive_orchestration_world(tmp_path, world, built_library):
    """blacklight_amd.distributed.render_adaptive - the code a multi-GPU run executes: per-level tiling, gathers of
    image / counts / flags / camera rows, max / sum reductions, refinement on rank 0 (the real bl_adaptive_refine on a
    host-only context), broadcast of the next block list - on 2 and 3 gloo ranks with a stub renderer, against the
    same loop on one rank. 3 ranks: shares of unequal size (36 tiles, block counts not divisible by 3)."""
    import dist_worker
    out_path = str(tmp_path / "levels.npz")
    mp.spawn(dist_worker.worker, args=(world, _free_port(), "stub", ADAPTIVE_STUB, None, True, out_path), nprocs=world, join=True)
    got = np.load(out_path)
    ctx = dist_worker.StubContext(ADAPTIVE_STUB)
    want = dist_worker.serial_adaptive(ctx, True)
    assert int(got["n_levels"]) == len(want) == 3          # the ring refines twice
    total_flagged = 0
    for n, level in enumerate(want):
        for key in ("image", "sample_num", "sample_flags", "camera_pos", "camera_dir"):
            assert np.array_equal(got[f"{key}_{n}"], level[key]), (n, key)
        if n > 0:
            assert np.array_equal(got[f"block_locs_{n}"], level["block_locs"])
            assert level["block_locs"].shape[0] % world != 0 or world == 2
        if n < len(want) - 1:
            assert np.array_equal(got[f"refinement_flags_{n}"], level["refinement_flags"])
        assert int(got[f"count_max_sample_num_{n}"]) == level["stats"].max_sample_num
        assert int(got[f"count_n_flagged_{n}"]) == level["stats"].n_flagged
        assert int(got[f"count_n_rays_{n}"]) == level["sample_num"].size
        total_flagged += level["stats"].n_flagged
    # one warning per level with the level's totals, in the reference's words (geodesics.cpp:389-394)
    expected = "".join(f"Warning: {lv['stats'].n_flagged} out of {lv['sample_num'].size} geodesics terminate unexpectedly.\n"
                       for lv in want if lv["stats"].n_flagged > 0)
    assert str(got["warnings"]) == expected and total_flagged > 0


def test_a_failing_rank_fails_every_rank(tmp_path, built_library):
    """One rank's render raises at level 1 (a refusal that depends on the rank's own rays): every rank raises RankError with
    that rank's text before the level's first data collective - nobody hangs until the backend's timeout."""
    import dist_worker
    out_path = str(tmp_path / "outcome")
    mp.spawn(dist_worker.worker, args=(3, _free_port(), "stub_failing", ADAPTIVE_STUB, None, False, out_path), nprocs=3, join=True)
    for rank in range(3):
        text = open(f"{out_path}.rank{rank}").read()
        assert "rank 1: RuntimeError: Error: this rank's rays reach a place" in text and "rank 0" not in text and "rank 2" not in text


def test_rank_0_failing_between_levels_fails_every_rank(tmp_path, built_library):
    """Rank 0 alone runs the refinement decision between two levels; an exception there (a bad image, out of memory) reaches every
    rank as RankError before the broadcast of the next block list, in which the others would otherwise wait for a rank that has left."""
    import dist_worker
    out_path = str(tmp_path / "outcome")
    mp.spawn(dist_worker.worker, args=(3, _free_port(), "stub_failing_refine", ADAPTIVE_STUB, None, False, out_path), nprocs=3, join=True)
    for rank in range(3):
        text = open(f"{out_path}.rank{rank}").read()
        assert "rank 0: MemoryError: no room for the next level's block list" in text and "rank 1" not in text


def test_launcher_refuses_a_mismatched_world(tmp_path):
    """bench.py --gpus N must equal the number of ranks it was started with, and without a launcher it must not run a
    smaller job under that name: both fail before anything touches a GPU."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    run = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True)
    assert run.returncode != 0 and "must agree" in run.stderr
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    run = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "64"], env=env, capture_output=True, text=True)
    assert run.returncode != 0 and "GPU(s) visible" in run.stderr


def test_frame_layout_is_built_once_and_detiles_without_index_building():
    """bench.py's gather_image() / render_level at world 8: the second frame builds no layout, no index and uploads nothing -
    what is left on the host is a cache lookup (the per-frame numpy de-tiling took 11.9 ms against an 8 ms frame)."""
    import time
    from blacklight_amd import distributed as bd
    res, world = 1024, 8
    bd.frame_layout.cache_clear()
    layout = bd.frame_layout(res, world, 32)
    builds = bd.ShareLayout.builds
    gathered = torch.empty((world, layout.n_padded), dtype=torch.float64)
    for r in range(world):
        gathered[r, : layout.counts[r]] = torch.from_numpy(layout.pixels[r].astype(np.float64))
    first = layout.detile(gathered, 1)     # builds the device-side permutation
    assert torch.equal(first[0], torch.arange(res * res, dtype=torch.float64))
    index_before = layout._index(gathered.device)
    t0 = time.perf_counter()
    for _ in range(20):
        again = bd.frame_layout(res, world, 32)
        pixels = bd.tile_pixels(res, 3, world, 32)
        index = again._index(gathered.device)
    host_ms = 1e3 * (time.perf_counter() - t0) / 20
    assert again is layout and pixels is layout.pixels[3] and index is index_before
    assert bd.ShareLayout.builds == builds, "a later frame rebuilt the layout"
    assert host_ms < 0.2, f"{host_ms:.3f} ms of host work per frame before the index_select"
    assert not layout.pixels[0].flags.writeable


@pytest.mark.parametrize("world", [1, 2, 3, 5, 8])
def test_detile_packed_rows_of_uneven_shares(world):
    """What bl_render leaves in a rank's buffer: `rows` rows of counts[r] values back to back (not n_padded apart), camera rows
    rays x 4. Shares of different lengths (9 tiles over 2, 5, 8 ranks) take the general index."""
    from blacklight_amd import distributed as bd
    res, tile, rows = 96, 32, 3
    layout = bd.frame_layout(res, world, tile)
    assert layout.even == (9 % world == 0)
    gathered = torch.full((world, rows * layout.n_padded), float("nan"), dtype=torch.float64)
    camera = torch.full((world, 4 * layout.n_padded), float("nan"), dtype=torch.float64)
    for r in range(world):
        pix = torch.from_numpy(layout.pixels[r].astype(np.float64))
        n = layout.counts[r]
        gathered[r, : rows * n] = torch.stack([pix * (q + 1) + 0.25 * q for q in range(rows)]).reshape(-1)
        camera[r, : 4 * n] = torch.stack([pix, -pix, pix * 0.5, pix + 7.0], dim=1).reshape(-1)
    m = torch.arange(res * res, dtype=torch.float64)
    full = layout.detile(gathered, rows)
    for q in range(rows):
        assert torch.equal(full[q], m * (q + 1) + 0.25 * q)
    cam = layout.detile(camera, 4, ray_major=True)
    assert cam.shape == (res * res, 4) and torch.equal(cam[:, 0], m) and torch.equal(cam[:, 3], m + 7.0)


def test_block_layout_matches_the_round_robin_deal():
    from blacklight_amd import distributed as bd
    n_blocks, bs, world = 11, 4, 3
    layout = bd.block_layout(n_blocks, bs, world)
    assert layout.counts == [4 * 16, 4 * 16, 3 * 16] and layout.n_padded == 4 * 16 and layout.n_total == 11 * 16
    gathered = torch.zeros((world, 2 * layout.n_padded), dtype=torch.float64)
    for r in range(world):
        ids = np.arange(r, n_blocks, world)
        where = (ids[:, None] * 16 + np.arange(16)[None]).reshape(-1).astype(np.float64)
        gathered[r, : 2 * where.size] = torch.from_numpy(np.concatenate([where, -where]))
    full = layout.detile(gathered, 2)
    m = torch.arange(n_blocks * 16, dtype=torch.float64)
    assert torch.equal(full[0], m) and torch.equal(full[1], -m)


def _frame_loop_worker(rank, world, port, resolution, tile, result_path):
    """bench.py's N > 1 frame loop on CPU tensors over gloo: Comm.gather_flat into the buffer rank 0 keeps, ShareLayout.detile, twice -
    the second frame must reuse buffer, layout and index."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from blacklight_amd import distributed as bd
    comm = bd.Comm(device=torch.device("cpu"))
    layout = bd.frame_layout(resolution, world, tile)
    pixels = layout.pixels[rank]
    image = torch.zeros((1, layout.n_padded), dtype=torch.float64)
    frames, buffers = [], []
    for frame in range(2):
        image[0, : pixels.size] = torch.from_numpy(pixels.astype(np.float64)) + 1000.0 * frame
        builds = bd.ShareLayout.builds
        gathered = comm.gather_flat(image.reshape(-1), dst=0)
        if rank == 0:
            frames.append(layout.detile(gathered, 1).clone())
            buffers.append(gathered.data_ptr())
        assert bd.ShareLayout.builds == builds and bd.frame_layout(resolution, world, tile) is layout
    if rank == 0:
        torch.save({"frames": frames, "same_buffer": buffers[0] == buffers[1]}, result_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_frame_loop_of_the_benchmark_over_gloo(tmp_path, world):
    resolution, tile = 96, 32   # nine tiles: shares of 5 + 4 (world 2) or 3 + 3 + 3 tiles
    result_path = str(tmp_path / "frames.pt")
    mp.spawn(_frame_loop_worker, args=(world, _free_port(), resolution, tile, result_path), nprocs=world, join=True)
    out = torch.load(result_path)
    want = torch.arange(resolution * resolution, dtype=torch.float64)
    assert torch.equal(out["frames"][0][0], want) and torch.equal(out["frames"][1][0], want + 1000.0)
    assert out["same_buffer"]
