"""Worker of the multi-process tests (spawned with torch.multiprocessing, gloo backend): runs the REAL orchestration
of blacklight_amd.distributed.render_adaptive - tiling, gathers, reductions, rank-0 refinement, block-list broadcast -
either on a GPU context (several ranks may share one GPU: the collectives go over gloo on the CPU) or on a stub
renderer whose images are an analytic function of the pixel position (CPU suite)."""
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for path in (REPO, os.path.join(REPO, "tests")):
    if path not in sys.path:
        sys.path.insert(0, path)


def stub_value(u, v):
    """Test image: smooth background plus a sharp ring (triggers the gradient / Laplacian criteria)."""
    rho = np.hypot(u - 0.47, v - 0.55)
    return 1.0 + 0.3 * u - 0.2 * v + 5.0 * np.exp(-((rho - 0.23) / 0.02) ** 2)


class StubContext:
    """Stands in for blacklight_amd.Context in render_level / render_adaptive: same render() contract, images from
    stub_value, refinement decisions and the writer from a host-only context (BL_DEVICE_NONE) of the real library."""

    fail_on_level = None   # a level at which render() raises (one rank's refusal: test of the error agreement)
    fail_refine = False    # adaptive_refine() raises (rank 0's own step between two levels)

    def __init__(self, params_dict):
        import blacklight_amd as bl
        self.params = bl.Params.from_dict(params_dict)
        self.host = bl.Context(self.params, device=-2)
        self.resolution = int(self.params.get("camera_resolution"))
        self.block_size = int(self.params.get("adaptive_block_size") or 1)
        self.n_q = self.host.num_quantities
        self.calls = []

    num_render_images = 0

    @property
    def num_quantities(self):
        return self.n_q

    def render_template(self, want_camera=False):
        return dict(image=np.empty((self.n_q, 0)), sample_num=np.empty(0, dtype=np.int32), sample_flags=np.empty(0, dtype=np.uint8),
                    camera_pos=np.empty((0, 4)) if want_camera else None, camera_dir=np.empty((0, 4)) if want_camera else None,
                    rendering=None)

    def render(self, level=0, block_locs=None, pixel_map=None, want_camera=False):
        res, bs = self.resolution, self.block_size
        if self.fail_on_level == level:
            raise RuntimeError("Error: this rank's rays reach a place the reference reads past its arrays.")
        if level == 0:
            pixels = np.arange(res * res) if pixel_map is None else np.asarray(pixel_map, dtype=np.int64)
            iu, iv, eff = pixels % res, pixels // res, res
        else:
            locs = np.asarray(block_locs, dtype=np.int64).reshape(-1, 2)
            jv, ju = np.mgrid[0:bs, 0:bs]
            iv = (locs[:, 0, None, None] * bs + jv[None]).reshape(-1)
            iu = (locs[:, 1, None, None] * bs + ju[None]).reshape(-1)
            eff = res << level
        u, v = (iu + 0.5) / eff, (iv + 0.5) / eff
        base = stub_value(u, v)
        image = np.stack([base * (q + 1) for q in range(self.n_q)])
        sample_num = (100 + 900 * u * v).astype(np.int32)
        sample_flags = ((iu * 7 + iv * 13) % 97 == 0).astype(np.uint8)
        stats = types.SimpleNamespace(max_sample_num=int(sample_num.max()) if sample_num.size else 0, n_flagged=int(sample_flags.sum()))
        self.calls.append((level, int(u.size)))
        cam = np.stack([u, v, u * v, u - v], axis=1) if want_camera else None
        return dict(image=image, sample_num=sample_num, sample_flags=sample_flags, camera_pos=cam,
                    camera_dir=None if cam is None else -cam, rendering=None, stats=stats)

    def adaptive_refine(self, level, image, block_locs=None):
        if self.fail_refine:
            raise MemoryError("no room for the next level's block list")
        return self.host.adaptive_refine(level, image, block_locs)

    def clear_warnings(self):
        pass


def serial_adaptive(ctx, want_camera):
    """Context.render_adaptive on one rank (blacklight.cpp:196-233), for the stub."""
    levels = [ctx.render(want_camera=want_camera)]
    levels[0]["block_locs"] = None
    level = 0
    while int(ctx.params.get("adaptive_max_level") or 0) > 0:
        flags, nxt = ctx.adaptive_refine(level, levels[level]["image"], levels[level]["block_locs"])
        levels[level]["refinement_flags"] = flags
        if nxt.shape[0] == 0:
            break
        level += 1
        out = ctx.render(level=level, block_locs=nxt, want_camera=want_camera)
        out["block_locs"] = nxt
        levels.append(out)
    return levels


def worker(rank, world, port, mode, params_dict, mock_args, want_camera, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from blacklight_amd import distributed as bd
    comm = bd.Comm(device=torch.device("cpu"))
    if mode in ("stub_failing", "stub_failing_refine"):
        # rank 1 alone fails at level 1 - or rank 0 in the refinement step, which it runs alone: every rank must raise, none may be
        # left waiting in a collective
        ctx = StubContext(params_dict)
        if mode == "stub_failing" and rank == 1:
            ctx.fail_on_level = 1
        if mode == "stub_failing_refine":
            ctx.fail_refine = True
        try:
            bd.render_adaptive(ctx, comm, want_camera)
            outcome = "no error"
        except bd.RankError as failure:
            outcome = str(failure)
        with open(f"{out_path}.rank{rank}", "w") as f:
            f.write(outcome)
    elif mode == "stub":
        ctx = StubContext(params_dict)
        levels, warnings = bd.render_adaptive(ctx, comm, want_camera)
        if rank == 0:
            np.savez(out_path, n_levels=len(levels), warnings=warnings,
                     **{f"{key}_{n}": lv[key] for n, lv in enumerate(levels) for key in lv
                        if isinstance(lv.get(key), np.ndarray)},
                     **{f"count_{key}_{n}": lv[key] for n, lv in enumerate(levels) for key in ("max_sample_num", "n_flagged", "n_rays")})
    else:
        import blacklight_amd as bl
        import golden_util as gu
        p = bl.Params.from_dict(params_dict)
        with bl.Context(p, device=0) as ctx:
            if mock_args is not None:
                ctx.set_grid(gu.golden_grid(mock_args))
            levels = ctx.render_adaptive(want_camera=want_camera, distributed=True, comm=comm)
            if rank == 0:
                ctx.write_output(levels, path=out_path)
                with open(out_path + ".warnings", "w") as f:
                    f.write(ctx.distributed_warnings)
                np.save(out_path + ".counts.npy", np.array([levels[0]["max_sample_num"], levels[0]["n_flagged"]]))
    dist.barrier()
    dist.destroy_process_group()
