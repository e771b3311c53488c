"""Host-side driver object over the C-ABI (bl_init / bl_set_grid / bl_render).

Mirrors how the reference's main() drives the hot path (src/blacklight.cpp:93-94, 178-233):
construct once from the input parameters, hand over the grid once per snapshot, render one adaptive
level per call. All computation happens in libblacklight_amd.so on the GPU.
"""
import ctypes as C

import numpy as np

from . import _capi
from .params import Params


class Context:
    def __init__(self, params: Params, device: int = -1):
        L = _capi.lib()
        self._lib = L
        self.params = params
        handle = C.c_void_p()
        rc = L.bl_init(params.ptr, device, C.byref(handle))
        if rc != 0:
            raise _capi.BlacklightError(rc, L.bl_last_global_error().decode())
        self._ctx = handle
        self._grid_keepalive = None
        self.n_freq = int(params.get("image_num_frequencies"))
        self.resolution = int(params.get("camera_resolution"))

    def close(self):
        if self._ctx:
            for ptr in self.__dict__.pop("_pinned", []):   # (arrays from pinned_array() end here: copy what is to outlive the context)
                self._lib.bl_host_free(self._ctx, ptr)
            self._lib.bl_free(self._ctx)
            self._ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise _capi.BlacklightError(rc, self._lib.bl_last_error(self._ctx).decode())

    # ------------------------------------------------------------------ queries
    @property
    def num_quantities(self):
        return self._lib.bl_image_num_quantities(self._ctx)

    @property
    def num_render_images(self):
        return self._lib.bl_render_num_images(self._ctx)

    @property
    def camera_frame(self):
        frame = _capi.CameraFrame()
        self._check(self._lib.bl_camera_frame_get(self._ctx, C.byref(frame)))
        return frame

    @property
    def frequencies(self):
        out = np.zeros(self.n_freq)
        self._check(self._lib.bl_frequencies(self._ctx, out.ctypes.data_as(C.POINTER(C.c_double)), self.n_freq))
        return out

    @property
    def warnings(self):
        return self._lib.bl_warnings(self._ctx).decode()

    @property
    def stats(self):
        st = _capi.Stats()
        self._check(self._lib.bl_get_stats(self._ctx, C.byref(st)))
        return st

    def set_scratch_limit(self, nbytes):
        self._check(self._lib.bl_set_scratch_limit(self._ctx, int(nbytes)))

    def debug_math(self, op, x, y=None):
        """Apply device math function `op` (see bl_debug_math in the header) element-wise; returns float64 array."""
        import numpy as np
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty_like(x)
        yp = None
        if y is not None:
            y = np.ascontiguousarray(y, dtype=np.float64)
            yp = y.ctypes.data
        self._check(self._lib.bl_debug_math(self._ctx, int(op), x.size, x.ctypes.data, yp, out.ctypes.data))
        return out

    def set_undefined_policy(self, policy):
        """"refuse" (default), "edge" (samples where the reference reads past its arrays), "kappa" (unpolarized kappa-distribution
        electrons, whose absorptivity the reference computes from an uninitialised constant), or "edge,kappa" (bl_set_undefined_policy)."""
        flags = 0
        for word in str(policy).replace("|", ",").split(","):
            flags |= {"refuse": 0, "edge": 1, "kappa": 2}[word.strip()]
        self._check(self._lib.bl_set_undefined_policy(self._ctx, flags))

    def set_arithmetic(self, mode):
        """"tolerant" (what a context starts in unless BLACKLIGHT_AMD_ARITHMETIC=exact) or "exact" (bit-identical to the reference):
        the arithmetic tier (bl_set_arithmetic)."""
        self._check(self._lib.bl_set_arithmetic(self._ctx, {"exact": 0, "tolerant": 1}[mode]))

    def set_reproducible(self, on=True):
        """Tolerant tier: bit-reproducible images (one transfer record per sample instead of composed maps; bl_set_reproducible)."""
        self._check(self._lib.bl_set_reproducible(self._ctx, 1 if on else 0))

    def set_tail_policy(self, policy):
        """"auto" (default), "wide", "quad" or "split": who steps the rays a chunk waits for longest (bl_set_tail_policy)."""
        self._check(self._lib.bl_set_tail_policy(self._ctx, {"auto": 0, "wide": 1, "quad": 2, "split": 3}[policy]))

    def set_geodesic_reuse(self, on=True):
        """Geodesics once per series (default on): a root-level render of an unchanged camera shades the sample records the last one
        left in HBM instead of integrating its rays again (bl_set_geodesic_reuse; stats.geodesics_reused says which way it went)."""
        self._check(self._lib.bl_set_geodesic_reuse(self._ctx, 1 if on else 0))

    def set_caller_stream(self, stream=None, enabled=True):
        """Every later render starts behind the work queued so far on `stream` (a raw hipStream_t handle, e.g.
        torch.cuda.current_stream().cuda_stream; None / 0: the NULL stream) - bl_set_caller_stream."""
        self._check(self._lib.bl_set_caller_stream(self._ctx, C.c_void_p(int(stream or 0)), 1 if enabled else 0))

    def follow_torch_stream(self, device=None):
        """set_caller_stream(torch's current stream on `device`): tensors torch filled, and collectives torch has waited for on
        that stream, are complete before a render touches the buffers."""
        import torch
        self.set_caller_stream(torch.cuda.current_stream(device).cuda_stream)

    def debug_set_switches(self, *names):
        """Measurement switches of this context by name (_capi.SWITCHES: "RECORD_EVERY_STEP", "NO_FUSED_LOCATE", ...); none: all off.
        They select another kernel or layout with the same results (bl_stats.switches echoes them)."""
        mask = 0
        for name in names:
            mask |= _capi.SWITCHES[name]
        self._check(self._lib.bl_debug_set_switches(self._ctx, mask))

    def debug_set_guard_band(self, relative_width):
        self._check(self._lib.bl_debug_set_guard_band(self._ctx, float(relative_width)))

    def set_overlap(self, on):
        """Overlap the geodesic kernel of the next chunk with the shading of the current one."""
        self._check(self._lib.bl_set_overlap(self._ctx, 1 if on else 0))

    # ------------------------------------------------------------------ grid
    def set_grid(self, grid):
        """grid: blacklight_amd.mock.Grid (or anything with .desc())."""
        desc = grid.desc()
        self._grid_keepalive = grid
        self._check(self._lib.bl_set_grid(self._ctx, C.byref(desc)))

    # ------------------------------------------------------------------ slow light
    def slow_light_read(self, snapshot):
        """SimulationReader::Read(snapshot) with slow_light_on: advance the window of files to the camera time of
        image `snapshot` (reads .athdf files itself) and select that image for the next render()."""
        self._check(self._lib.bl_slow_light_read(self._ctx, int(snapshot)))

    def set_grid_slice(self, n, grid, time):
        """Slice n of the slow-light window (n = 0: latest file) from a caller-side reader."""
        desc = grid.desc()
        self._check(self._lib.bl_set_grid_slice(self._ctx, int(n), C.byref(desc), float(time)))

    def shift_grid_slices(self, count):
        self._check(self._lib.bl_shift_grid_slices(self._ctx, int(count)))

    def set_snapshot(self, snapshot):
        self._check(self._lib.bl_set_snapshot(self._ctx, int(snapshot)))

    def clear_warnings(self):
        self._lib.bl_warnings_clear(self._ctx)

    # ------------------------------------------------------------------ render
    def level_pixels(self, level=0, n_blocks=0):
        if level == 0:
            return self.resolution * self.resolution
        bs = int(self.params.get("adaptive_block_size"))
        return n_blocks * bs * bs

    def pinned_array(self, shape, dtype=np.float64):
        """A numpy array over host memory the GPU downloads into at the link's rate (bl_host_alloc: pinned), owned by this context and
        freed with it (close(): the array must not be used afterwards) - what a frame loop hands to render(out=...) frame after frame.
        Falls back to an ordinary array where pinning is refused."""
        shape = tuple(int(n) for n in np.atleast_1d(shape))
        count = int(np.prod(shape))
        nbytes = count * np.dtype(dtype).itemsize
        ptr = self._lib.bl_host_alloc(self._ctx, nbytes) if nbytes > 0 else None
        if not ptr:
            return np.empty(shape, dtype=dtype)
        self.__dict__.setdefault("_pinned", []).append(ptr)
        raw = (C.c_char * nbytes).from_address(ptr)
        return np.frombuffer(raw, dtype=dtype, count=count).reshape(shape)

    def render(self, level=0, block_locs=None, pixel_map=None, want_camera=False, out=None):
        """Trace one adaptive level into host arrays. Returns a dict. out: a dict of arrays to receive "image" (n_q, n_rays) f64,
        "sample_num" (n_rays) i32 and "sample_flags" (n_rays) u8 instead of fresh ones - e.g. pinned_array()s a frame loop reuses:
        a download into touched, pinned pages takes a third of the time of one into new pageable memory."""
        d = _capi.RenderDesc()
        d.level = level
        keep = []
        n_blocks = 0
        if block_locs is not None:
            bl = np.ascontiguousarray(block_locs, dtype=np.int32).reshape(-1, 2)
            keep.append(bl)
            d.block_locs = bl.ctypes.data_as(C.c_void_p)
            n_blocks = bl.shape[0]
            d.n_blocks = n_blocks
        if pixel_map is not None:
            pm = np.ascontiguousarray(pixel_map, dtype=np.int32)
            keep.append(pm)
            d.pixel_map = pm.ctypes.data_as(C.c_void_p)
            n_rays = pm.size
        else:
            n_rays = self.level_pixels(level, n_blocks)
        d.n_rays = n_rays
        d.outputs_on_device = 0
        n_q = self.num_quantities
        out = out or {}
        image = out.get("image") if out.get("image") is not None else np.empty((n_q, n_rays), dtype=np.float64)
        sample_num = out.get("sample_num") if out.get("sample_num") is not None else np.empty(n_rays, dtype=np.int32)
        sample_flags = out.get("sample_flags") if out.get("sample_flags") is not None else np.empty(n_rays, dtype=np.uint8)
        if (image.shape != (n_q, n_rays) or image.dtype != np.float64 or not image.flags.c_contiguous or sample_num.shape != (n_rays,)
                or sample_num.dtype != np.int32 or sample_flags.shape != (n_rays,) or sample_flags.dtype != np.uint8):
            raise ValueError("render(out=...): image (n_q, n_rays) float64, sample_num (n_rays) int32, sample_flags (n_rays) uint8, C-contiguous")
        d.image = image.ctypes.data_as(C.c_void_p)
        d.sample_num = sample_num.ctypes.data_as(C.c_void_p)
        d.sample_flags = sample_flags.ctypes.data_as(C.c_void_p)
        camera_pos = camera_dir = None
        if want_camera:
            camera_pos = np.empty((n_rays, 4))
            camera_dir = np.empty((n_rays, 4))
            d.camera_pos = camera_pos.ctypes.data_as(C.c_void_p)
            d.camera_dir = camera_dir.ctypes.data_as(C.c_void_p)
        rendering = None
        n_render = self.num_render_images
        if n_render > 0:
            rendering = np.empty((n_render, 3, n_rays))
            d.render = rendering.ctypes.data_as(C.c_void_p)
        self._check(self._lib.bl_render(self._ctx, C.byref(d)))
        return dict(image=image, sample_num=sample_num, sample_flags=sample_flags,
                    camera_pos=camera_pos, camera_dir=camera_dir, rendering=rendering, stats=self.stats)

    # ------------------------------------------------------------------ host steps of the reference loop
    def adaptive_refine(self, level, image, block_locs=None):
        """CheckAdaptiveRefinement for the level just rendered (reference radiation_adaptive.cpp:19-139)
        plus AugmentCamera's block list of the next level (camera.cpp:445-458).
        Returns (refine_flags, next_block_locs); an empty next list ends the adaptive loop."""
        image = np.ascontiguousarray(image, dtype=np.float64)
        if level == 0:
            bs = int(self.params.get("adaptive_block_size"))
            n_blocks = (self.resolution // bs) ** 2
            locs_ptr = None
        else:
            bl = np.ascontiguousarray(block_locs, dtype=np.int32).reshape(-1, 2)
            n_blocks = bl.shape[0]
            locs_ptr = bl.ctypes.data_as(C.c_void_p)
        flags = np.zeros(n_blocks, dtype=np.uint8)
        nxt = np.zeros((4 * n_blocks, 2), dtype=np.int32)
        count = C.c_int32(0)
        self._check(self._lib.bl_adaptive_refine(self._ctx, level, n_blocks, locs_ptr, image.ctypes.data_as(C.c_void_p),
                                                 flags.ctypes.data_as(C.c_void_p), C.byref(count),
                                                 nxt.ctypes.data_as(C.c_void_p)))
        return flags.astype(bool), nxt[: 4 * count.value].copy()

    def render_template(self, want_camera=False):
        """The per-pixel outputs of render() for zero rays: which rows a level has and their leading shapes
        (blacklight_amd.distributed uses it on a rank that holds no rays of a level)."""
        n_render = self.num_render_images
        return dict(image=np.empty((self.num_quantities, 0)), sample_num=np.empty(0, dtype=np.int32),
                    sample_flags=np.empty(0, dtype=np.uint8),
                    camera_pos=np.empty((0, 4)) if want_camera else None, camera_dir=np.empty((0, 4)) if want_camera else None,
                    rendering=np.empty((n_render, 3, 0)) if n_render > 0 else None)

    def render_adaptive(self, want_camera=False, distributed=False, comm=None):
        """The reference's do { Integrate; AddGeodesics } while (!done) loop (blacklight.cpp:196-233).
        Returns a list of per-level dicts (level 0 first), each with image / block_locs / ...
        distributed=True: every rank of torch.distributed's default group calls this; the camera is tiled over the
        ranks level by level (blacklight_amd.distributed.render_adaptive) and rank 0 gets the same list a single
        GPU returns (the other ranks get None); warnings with the levels' totals are in .distributed_warnings."""
        if distributed:
            from . import distributed as bd
            levels, self.distributed_warnings = bd.render_adaptive(self, comm, want_camera)
            return levels
        levels = [self.render(want_camera=want_camera)]
        levels[0]["block_locs"] = None
        if int(self.params.get("adaptive_max_level") or 0) <= 0:
            return levels
        level = 0
        while True:
            flags, nxt = self.adaptive_refine(level, levels[level]["image"], levels[level]["block_locs"])
            levels[level]["refinement_flags"] = flags
            if nxt.shape[0] == 0:
                return levels
            level += 1
            out = self.render(level=level, block_locs=nxt, want_camera=want_camera)
            out["block_locs"] = nxt
            levels.append(out)

    def write_output(self, levels, path=None, snapshot=0):
        """OutputWriter::Write (reference output_writer.cpp:169-274); `levels` as from render_adaptive."""
        d = _capi.OutputDesc()
        d.adaptive_num_levels = len(levels) - 1
        d.snapshot = snapshot
        keep = []
        plane = int(self.params.get("camera_type")) == 0
        for index, lv in enumerate(levels):
            image = np.ascontiguousarray(lv["image"], dtype=np.float64)
            keep.append(image)
            d.level[index].image = image.ctypes.data_as(C.c_void_p)
            if index > 0:
                bl = np.ascontiguousarray(lv["block_locs"], dtype=np.int32)
                keep.append(bl)
                d.level[index].n_blocks = bl.shape[0]
                d.level[index].block_locs = bl.ctypes.data_as(C.c_void_p)
            if lv.get("rendering") is not None:
                rendering = np.ascontiguousarray(lv["rendering"], dtype=np.float64)
                keep.append(rendering)
                d.level[index].render = rendering.ctypes.data_as(C.c_void_p)
            camera = lv.get("camera_pos") if plane else lv.get("camera_dir")
            if camera is not None:
                camera = np.ascontiguousarray(camera, dtype=np.float64)
                keep.append(camera)
                d.level[index].camera = camera.ctypes.data_as(C.c_void_p)
        self._check(self._lib.bl_write_output(self._ctx, None if path is None else str(path).encode(), C.byref(d)))

    def render_device(self, image_ptr, n_rays, level=0, pixel_map=None, sample_num_ptr=0, sample_flags_ptr=0,
                      block_locs=None, camera_pos_ptr=0, camera_dir_ptr=0, render_ptr=0):
        """Trace into caller-owned HBM (raw device pointers, e.g. torch.Tensor.data_ptr()): image (n_q, n_rays), sample_num,
        sample_flags (n_rays), camera_pos / camera_dir (n_rays, 4), renderings (n_images, 3, n_rays) - the layouts of render()."""
        d = _capi.RenderDesc()
        d.level = level
        keep = []
        if block_locs is not None:
            bl = np.ascontiguousarray(block_locs, dtype=np.int32).reshape(-1, 2)
            keep.append(bl)
            d.block_locs = bl.ctypes.data_as(C.c_void_p)
            d.n_blocks = bl.shape[0]
        if pixel_map is not None:
            pm = np.ascontiguousarray(pixel_map, dtype=np.int32)
            keep.append(pm)
            d.pixel_map = pm.ctypes.data_as(C.c_void_p)
        d.n_rays = n_rays
        d.outputs_on_device = 1
        d.image = C.c_void_p(image_ptr)
        d.sample_num = C.c_void_p(sample_num_ptr) if sample_num_ptr else None
        d.sample_flags = C.c_void_p(sample_flags_ptr) if sample_flags_ptr else None
        d.camera_pos = C.c_void_p(camera_pos_ptr) if camera_pos_ptr else None
        d.camera_dir = C.c_void_p(camera_dir_ptr) if camera_dir_ptr else None
        d.render = C.c_void_p(render_ptr) if render_ptr else None
        self._check(self._lib.bl_render(self._ctx, C.byref(d)))
        return self.stats
