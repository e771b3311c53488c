"""Geodesic checkpoints (reference src/geodesic_integrator/geodesic_checkpoint.cpp:19-103, utils/file_io.cpp:65-131).
checkpoint_geodesic_load: the reference's own files (tests/golden/reader/geodesic_*.ckpt, written by tools/make_goldens.py
checkpoint) replace the geodesic kernel, and the image is the reference's bit for bit. checkpoint_geodesic_save: the file this
library writes holds the reference's values wherever the reference defines them (it leaves the tails of its sample arrays
unset), and the reference binary, where it travelled to the box, reads that file and reproduces its own image."""
import json
import os
import subprocess

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READER_DIR = os.path.join(gu.GOLDEN_DIR, "reader")
REFERENCE = os.path.join(REPO, "oracle", "_ref", "blacklight")
PRELOAD = os.path.join(REPO, "oracle", "_ref", "libblmath_preload.so")
ROWS = {"sim": ["I_nu", "time", "tau"], "formula": ["I_nu"]}


@pytest.fixture(scope="module")
def expected():
    return np.load(os.path.join(READER_DIR, "expected_checkpoint.npz"), allow_pickle=False)


def read_checkpoint(path):
    """The reference's layout: seven 4-vectors, then Arrays as five int32 extents (fastest first) + data."""
    out = {}
    with open(path, "rb") as f:
        for name in ("cam_x", "u_con", "u_cov", "norm_con", "norm_con_c", "hor_con_c", "vert_con_c"):
            out[name] = np.frombuffer(f.read(32), dtype="<f8").copy()

        def array(dtype):
            dims = [int(d) for d in np.frombuffer(f.read(20), dtype="<i4")]
            shape = dims[::-1]
            while len(shape) > 1 and shape[0] == 1:
                shape = shape[1:]                     # unused slow extents are 1
            count = int(np.prod(shape))
            return np.frombuffer(f.read(count * np.dtype(dtype).itemsize), dtype=dtype).reshape(shape).copy()

        out["camera_pos"] = array("<f8")
        out["camera_dir"] = array("<f8")
        out["image_frequencies"] = array("<f8")
        out["momentum_factors"] = array("<f8")
        out["geodesic_num_steps"] = int(np.frombuffer(f.read(4), dtype="<i4")[0])
        out["sample_flags"] = array("u1")
        out["sample_num"] = array("<i4")
        out["sample_pos"] = array("<f8")
        out["sample_dir"] = array("<f8")
        out["sample_len"] = array("<f8")
        assert f.read() == b""
    return out


def _context(expected, case, **overrides):
    from blacklight_amd import Context, Params
    params = json.loads(str(expected[f"{case}_params"]))
    params.update(checkpoint_geodesic_save="false", checkpoint_geodesic_load="false")
    params.update(overrides)
    ctx = Context(Params.from_dict(params))
    if f"{case}_mock_args" in expected.files:
        ctx.set_grid(gu.golden_grid(json.loads(str(expected[f"{case}_mock_args"]))))
    return ctx


def _want(expected, case):
    return np.stack([expected[f"{case}_npz_{name}"].reshape(-1) for name in ROWS[case]])


@pytest.mark.parametrize("case", ["sim", "formula"])
def test_load_the_reference_checkpoint(built_library, expected, case):
    path = os.path.join(READER_DIR, f"geodesic_{case}.ckpt")
    with _context(expected, case, checkpoint_geodesic_load="true", checkpoint_geodesic_file=path) as ctx:
        out = ctx.render(want_camera=True)
        assert out["stats"].ms_geodesic == 0.0                       # nothing was integrated
        assert gu.same_bits(out["image"], _want(expected, case)).all()
        ref = read_checkpoint(path)
        assert np.array_equal(out["sample_num"], ref["sample_num"])
        assert np.array_equal(out["sample_flags"].astype(bool), ref["sample_flags"].astype(bool))
        assert gu.same_bits(out["camera_pos"], ref["camera_pos"]).all() and gu.same_bits(out["camera_dir"], ref["camera_dir"]).all()
        # the reference's words for the flagged rays come from the file's flags
        assert ctx.warnings == str(expected[f"{case}_warnings"])


@pytest.mark.parametrize("case", ["sim", "formula"])
def test_save_matches_the_reference_file(built_library, expected, case, tmp_path):
    mine = str(tmp_path / "mine.ckpt")
    with _context(expected, case, checkpoint_geodesic_save="true", checkpoint_geodesic_file=mine) as ctx:
        out = ctx.render()
        assert gu.same_bits(out["image"], _want(expected, case)).all()
    got, ref = read_checkpoint(mine), read_checkpoint(os.path.join(READER_DIR, f"geodesic_{case}.ckpt"))
    assert os.path.getsize(mine) == os.path.getsize(os.path.join(READER_DIR, f"geodesic_{case}.ckpt"))
    for name in ("cam_x", "u_con", "u_cov", "norm_con", "norm_con_c", "hor_con_c", "vert_con_c", "camera_pos", "camera_dir",
                 "image_frequencies", "momentum_factors", "sample_len"):
        assert got[name].shape == ref[name].shape and gu.same_bits(got[name], ref[name]).all(), name
    assert got["geodesic_num_steps"] == ref["geodesic_num_steps"]
    assert np.array_equal(got["sample_num"], ref["sample_num"]) and np.array_equal(got["sample_flags"], ref["sample_flags"])
    # positions and directions: the reference sets entries [0, sample_num) of each pixel and leaves the rest as allocated
    for name in ("sample_pos", "sample_dir"):
        assert got[name].shape == ref[name].shape
        for m, num in enumerate(ref["sample_num"]):
            assert gu.same_bits(got[name][m, :num], ref[name][m, :num]).all(), (name, m)
            assert not got[name][m, num:].any()
    # sample_len is zero-initialised by the reference, so it compares whole (above)

    # ... and what was saved loads back to the same image
    with _context(expected, case, checkpoint_geodesic_load="true", checkpoint_geodesic_file=mine) as ctx:
        again = ctx.render()
        assert gu.same_bits(again["image"], _want(expected, case)).all()


def test_the_reference_loads_our_checkpoint(built_library, expected, tmp_path):
    """The compiled reference (oracle/_ref/blacklight with the pinned math library preloaded), where it travelled with the tree, loads a
    geodesic checkpoint this library saved and renders the reference's own image from it. Skipped - visibly - where the binary is not
    there. (Formula case: the simulation case would need its athdf file, which is not a fixture.)"""
    case = "formula"
    if not (os.path.exists(REFERENCE) and os.path.exists(PRELOAD)):
        pytest.skip("oracle/_ref/blacklight and libblmath_preload.so did not travel with this tree")
    mine = str(tmp_path / "mine.ckpt")
    with _context(expected, case, checkpoint_geodesic_save="true", checkpoint_geodesic_file=mine) as ctx:
        ctx.render()
    work = tmp_path / "reference"
    (work / "data").mkdir(parents=True)
    (work / "output").mkdir()
    params = json.loads(str(expected[f"{case}_params"]))
    params.update(checkpoint_geodesic_save="false", checkpoint_geodesic_load="true", checkpoint_geodesic_file=mine,
                  output_file=str(work / "output" / "out.npz"))
    with open(work / "load.input", "w") as f:
        for key, value in params.items():
            f.write(f"{key} = {value}\n")
    run = subprocess.run([REFERENCE, "load.input"], cwd=work, env=dict(os.environ, LD_PRELOAD=PRELOAD, OMP_NUM_THREADS="4"),
                         capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    npz = np.load(params["output_file"])
    for name in ROWS[case]:
        assert gu.same_bits(npz[name], expected[f"{case}_npz_{name}"]).all(), name


@pytest.mark.parametrize("case", ["sim", "formula"])
def test_round_trip_in_several_chunks(built_library, expected, case, tmp_path):
    """A 16 x 16 camera saved and loaded with the smallest scratch budget (a few thousand sample records per chunk): the chunked
    save writes the file the unchunked one writes, and the chunked load gives the image of the plain render."""
    path, whole = str(tmp_path / "chunked.ckpt"), str(tmp_path / "whole.ckpt")
    with _context(expected, case, camera_resolution=16) as ctx:
        plain = ctx.render()
        limit = 1 << 20
    with _context(expected, case, camera_resolution=16, checkpoint_geodesic_save="true", checkpoint_geodesic_file=whole) as ctx:
        ctx.render()
    with _context(expected, case, camera_resolution=16, checkpoint_geodesic_save="true", checkpoint_geodesic_file=path) as ctx:
        ctx.set_scratch_limit(limit)
        saved = ctx.render()
        assert saved["stats"].n_chunks > 1 and gu.same_bits(saved["image"], plain["image"]).all()
    with open(path, "rb") as a, open(whole, "rb") as b:
        assert a.read() == b.read()
    with _context(expected, case, camera_resolution=16, checkpoint_geodesic_load="true", checkpoint_geodesic_file=path) as ctx:
        ctx.set_scratch_limit(limit)
        loaded = ctx.render()
        assert loaded["stats"].n_chunks > 1 and gu.same_bits(loaded["image"], plain["image"]).all()
        assert np.array_equal(loaded["sample_num"], plain["sample_num"])


def test_checkpoint_subsets_and_damaged_files(built_library, expected, tmp_path):
    """Loading serves any pixel subset (a rank's tiles) from the one file; saving needs the whole camera, because the file has
    no way to say which pixels it holds; a truncated or absent file is an error in the reference's words."""
    from blacklight_amd import BlacklightError
    path = os.path.join(READER_DIR, "geodesic_formula.ckpt")
    subset = np.random.default_rng(3).permutation(64)[:20].astype(np.int32)
    with _context(expected, "formula", checkpoint_geodesic_load="true", checkpoint_geodesic_file=path) as ctx:
        part = ctx.render(pixel_map=subset)
        assert gu.same_bits(part["image"], _want(expected, "formula")[:, subset]).all()
        with pytest.raises(BlacklightError, match="does not hold"):
            ctx.render(pixel_map=np.array([3, 64], dtype=np.int32))
    with _context(expected, "formula", checkpoint_geodesic_save="true", checkpoint_geodesic_file=str(tmp_path / "part.ckpt")) as ctx:
        with pytest.raises(BlacklightError, match="whole root camera"):
            ctx.render(pixel_map=subset)
    bad = tmp_path / "short.ckpt"
    with open(path, "rb") as src:
        bad.write_bytes(src.read()[:100000])
    with _context(expected, "formula", checkpoint_geodesic_load="true", checkpoint_geodesic_file=str(bad)) as ctx:
        with pytest.raises(BlacklightError, match="checkpoint"):
            ctx.render()
    with _context(expected, "formula", checkpoint_geodesic_load="true", checkpoint_geodesic_file=str(tmp_path / "none")) as ctx:
        with pytest.raises(BlacklightError, match="Could not open"):
            ctx.render()


# ---- sample checkpoints (sample_checkpoint.cpp:22-46)
def _read_sample_checkpoint(path, interp, block_interp=False):
    data = open(path, "rb").read()
    out, off = {}, 0
    for name, dtype in [("inds", "<i4")] + ([("fracs", "<f8")] if interp else []) + [("nan", "u1"), ("fallback", "u1")]:
        dims = [int(v) for v in np.frombuffer(data[off:off + 20], dtype="<i4")]
        off += 20
        shape = dims[::-1]
        while len(shape) > 1 and shape[0] == 1:
            shape = shape[1:]
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        out[name] = np.frombuffer(data[off:off + nbytes], dtype=dtype).reshape(shape)
        off += nbytes
    assert off == len(data)
    return out


@pytest.mark.parametrize("case", ["interp", "nearest_blocks_fallback", "few_steps_nan"])
@pytest.mark.parametrize("tier,chunked", [("exact", False), ("exact", True), ("tolerant", False)])
def test_sample_checkpoint_holds_what_the_reference_defines(built_library, case, tier, chunked, tmp_path):
    """checkpoint_sample_save: the file has the reference's layout and, wherever the reference defines an entry (a kept sample
    inside camera_r and on the grid: its MeshBlock and cell indices, its trilinear fractions), the reference's value bit for
    bit; sample_nan and sample_fallback, which the reference zeroes first, are equal throughout. The image is the one a run
    without the checkpoint gives, and the second image of a context does not write the file again."""
    from blacklight_amd import Context, Params
    fx = np.load(os.path.join(READER_DIR, "expected_sample_checkpoint.npz"), allow_pickle=False)
    params = json.loads(str(fx[f"{case}_params"]))
    mock_args = json.loads(str(fx[f"{case}_mock_args"]))
    path = str(tmp_path / "sample.ckpt")
    plain_params = dict(params, checkpoint_sample_save="false")
    with Context(Params.from_dict(plain_params)) as ctx:
        ctx.set_grid(gu.golden_grid(mock_args))
        ctx.set_arithmetic(tier)
        plain = ctx.render()
    with Context(Params.from_dict(dict(params, checkpoint_sample_file=path))) as ctx:
        ctx.set_grid(gu.golden_grid(mock_args))
        ctx.set_arithmetic(tier)
        if chunked:
            ctx.set_scratch_limit(1 << 20)
        saved = ctx.render()
        assert (saved["stats"].n_chunks > 1) == chunked
        if tier == "exact":
            assert gu.same_bits(saved["image"], plain["image"]).all()
        else:   # a run that saves located samples takes the locate kernel + bl_shade_fast_kernel, the plain one a fused kernel: the
            # tier's arithmetic in another association of its operations (rounding level), the same NaN mask
            assert np.array_equal(np.isnan(saved["image"]), np.isnan(plain["image"]))
            with np.errstate(invalid="ignore"):
                assert np.nanmax(np.abs(saved["image"] - plain["image"])) <= 1.0e-13 * np.nanmax(np.abs(plain["image"]))
        assert np.array_equal(saved["sample_num"], plain["sample_num"])
        assert np.array_equal(saved["sample_num"], fx[f"{case}_sample_num"])
        stamp = os.path.getmtime(path)
        size = os.path.getsize(path)
        ctx.render()
        assert os.path.getmtime(path) == stamp and os.path.getsize(path) == size
    interp = params["simulation_interp"] == "true"
    got = _read_sample_checkpoint(path, interp)
    m, n = fx[f"{case}_pixels"].astype(np.int64), fx[f"{case}_steps"].astype(np.int64)
    assert got["nan"].shape == fx[f"{case}_nan"].shape and got["inds"].shape == got["nan"].shape + (4,)
    assert np.array_equal(got["nan"], fx[f"{case}_nan"]) and np.array_equal(got["fallback"], fx[f"{case}_fallback"])
    assert np.array_equal(got["inds"][m, n], fx[f"{case}_inds"])
    if interp:
        assert gu.same_bits(got["fracs"][m, n], fx[f"{case}_fracs"]).all()
    # what the reference leaves unset is zero here: nothing beyond a pixel's samples
    beyond = np.arange(got["nan"].shape[1])[None, :] >= fx[f"{case}_sample_num"][:, None]
    assert not got["inds"][beyond].any() and not got["nan"][beyond].any()


def test_sample_checkpoint_load_is_refused_with_the_reason(built_library):
    from blacklight_amd import Context, Params, BlacklightError
    fx = np.load(os.path.join(READER_DIR, "expected_sample_checkpoint.npz"), allow_pickle=False)
    params = json.loads(str(fx["interp_params"]))
    params.update(checkpoint_sample_save="false", checkpoint_sample_load="true", checkpoint_sample_file="x.ckpt")
    with pytest.raises(BlacklightError, match="cannot load its own sample checkpoints"):
        Context(Params.from_dict(params))
