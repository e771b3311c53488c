"""Host-side steps of the boundary on a host-only context (no GPU): camera frame and frequency list
against the reference's checkpoint, the adaptive refinement decision + next-level block list against
the reference's adaptive_block_locs, and the .npz / .npy / raw writer against the reference's own
output files (record names, order, shapes, bytes of every array; 128-byte .npy headers; stored ZIP
entries with valid CRC-32)."""
import io
import struct
import zipfile

import numpy as np
import pytest

import golden_util as gu

BL_DEVICE_NONE = -2


def _host_context(case):
    import blacklight_amd as bl
    fx, params, mock_args = gu.load_case(case)
    p = bl.Params.from_dict(params)
    return fx, p, bl.Context(p, device=BL_DEVICE_NONE)


@pytest.mark.parametrize("case", gu.GPU_CASES)
def test_camera_frame_and_frequencies(case, built_library):
    fx, p, ctx = _host_context(case)
    frame = ctx.camera_frame
    for key in gu.FRAME_KEYS:
        assert np.array_equal(np.array(getattr(frame, key)), fx[f"B_{key}"]), key
    assert np.array_equal(ctx.frequencies, fx["B_image_frequencies"])
    assert frame.mass_msun == float(fx["B_npz_mass_msun"][0])
    assert ctx.num_quantities == gu.expected_image(fx, "B", int(p.get("camera_resolution")) ** 2).shape[0]


def test_host_only_context_cannot_render(built_library):
    import blacklight_amd as bl
    fx, p, ctx = _host_context("formula_flat")
    with pytest.raises(bl.BlacklightError) as err:
        ctx.render()
    assert err.value.code == 4


def _levels_from_golden(fx, n_nu):
    """Per-level dicts in the shape Context.render_adaptive returns, from the reference's npz."""
    res = fx["B_npz_I_nu"].shape[-1]
    levels = [dict(image=fx["B_npz_I_nu"].reshape(n_nu, res * res), block_locs=None)]
    for key in ("positions", "directions"):
        if f"B_npz_{key}" in fx.files:
            levels[0]["camera_pos" if key == "positions" else "camera_dir"] = fx[f"B_npz_{key}"].reshape(-1, 4)
    for level in range(1, int(fx["B_npz_adaptive_num_levels"][0]) + 1):
        img = fx[f"B_npz_adaptive_I_nu_{level}"]
        lv = dict(image=img.reshape(n_nu, -1), block_locs=fx[f"B_npz_adaptive_block_locs_{level}"])
        for key in ("positions", "directions"):
            name = f"B_npz_adaptive_{key}_{level}"
            if name in fx.files:
                lv["camera_pos" if key == "positions" else "camera_dir"] = fx[name].reshape(-1, 4)
        levels.append(lv)
    return levels


@pytest.mark.parametrize("case", ["sim_adaptive", "formula_adaptive_multifreq"])
def test_refinement_decision_matches_reference(case, built_library):
    fx, p, ctx = _host_context(case)
    n_nu = int(p.get("image_num_frequencies"))
    levels = _levels_from_golden(fx, n_nu)
    for level, lv in enumerate(levels):
        flags, nxt = ctx.adaptive_refine(level, lv["image"], lv["block_locs"])
        if level + 1 < len(levels):
            assert np.array_equal(nxt, levels[level + 1]["block_locs"])
            assert flags.sum() * 4 == levels[level + 1]["block_locs"].shape[0]
        else:
            assert nxt.shape[0] == 0
    assert np.array_equal(fx["B_npz_adaptive_num_blocks"][1:], [lv["block_locs"].shape[0] for lv in levels[1:]])


@pytest.mark.parametrize("case", ["sim_adaptive", "formula_adaptive_multifreq", "sim_multifreq", "sim_dp_interp"])
def test_npz_writer_matches_reference_file(case, built_library, tmp_path):
    fx, p, ctx = _host_context(case)
    n_nu = int(p.get("image_num_frequencies"))
    levels = _levels_from_golden(fx, n_nu)
    path = tmp_path / "out.npz"
    ctx.write_output(levels, path=path)
    want_names = [k[len("B_npz_"):] for k in fx.files if k.startswith("B_npz_")]
    with zipfile.ZipFile(path) as z:
        assert z.testzip() is None                       # CRC-32 of every entry
        assert [i.filename for i in z.infolist()] == [n + ".npy" for n in want_names]
        for info in z.infolist():
            assert info.compress_type == zipfile.ZIP_STORED and info.create_system == 3
            raw = z.read(info.filename)
            assert raw[:8] == b"\x93NUMPY\x01\x00" and struct.unpack("<H", raw[8:10])[0] == 118
            assert raw[127:128] == b"\n"
    got = np.load(path)
    for name in want_names:
        want = fx["B_npz_" + name]
        assert got[name].dtype == want.dtype and got[name].shape == want.shape, name
        assert np.array_equal(got[name].view(np.uint8 if want.dtype.itemsize == 1 else f"u{want.dtype.itemsize}"),
                              want.view(np.uint8 if want.dtype.itemsize == 1 else f"u{want.dtype.itemsize}")), name


def test_npy_and_raw_formats(built_library, tmp_path):
    import blacklight_amd as bl
    fx, params, _ = gu.load_case("sim_multifreq")
    image = fx["B_npz_I_nu"].reshape(3, -1)
    for fmt in ("npy", "raw"):
        q = dict(params, output_format=fmt)
        ctx = bl.Context(bl.Params.from_dict(q), device=BL_DEVICE_NONE)
        path = tmp_path / f"out.{fmt}"
        ctx.write_output([dict(image=image, block_locs=None)], path=path)
        if fmt == "npy":
            arr = np.load(path)
            assert arr.shape == (3, 16, 16) and np.array_equal(arr.reshape(3, -1), image)
        else:
            assert np.array_equal(np.fromfile(path, dtype="<f8").reshape(3, -1), image)


def test_output_file_pattern_for_multiple_runs(built_library, tmp_path):
    import blacklight_amd as bl
    fx, params, _ = gu.load_case("sim_dp_interp")
    q = dict(params, simulation_multiple="true", simulation_start=7, simulation_end=9,
             output_file=str(tmp_path / "img_{04d}.npz"))
    ctx = bl.Context(bl.Params.from_dict(q), device=BL_DEVICE_NONE)
    ctx.write_output([dict(image=fx["B_npz_I_nu"].reshape(1, -1), block_locs=None)], snapshot=2)
    assert (tmp_path / "img_0009.npz").exists()
