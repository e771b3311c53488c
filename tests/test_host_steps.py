"""Host-side steps of the boundary on a host-only context (no GPU): camera frame and frequency list
against the reference's checkpoint, the adaptive refinement decision + next-level block list against
the reference's adaptive_block_locs, and the .npz / .npy / raw writer against the reference's own
output files (record names, order, shapes, bytes of every array; 128-byte .npy headers; stored ZIP
entries with valid CRC-32)."""
import io
import os
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


def test_npz_zip64_fields_read_back(built_library, tmp_path, monkeypatch):
    """BLACKLIGHT_AMD_ZIP64=always: every record carries the ZIP64 extended-information field and the file ends with the ZIP64
    end-of-central-directory record + locator; zipfile and numpy read the reference's arrays out of it unchanged."""
    fx, p, ctx = _host_context("sim_adaptive")
    levels = _levels_from_golden(fx, int(p.get("image_num_frequencies")))
    plain, forced = tmp_path / "plain.npz", tmp_path / "zip64.npz"
    ctx.write_output(levels, path=plain)
    monkeypatch.setenv("BLACKLIGHT_AMD_ZIP64", "always")
    ctx.write_output(levels, path=forced)
    monkeypatch.delenv("BLACKLIGHT_AMD_ZIP64")
    raw = forced.read_bytes()
    assert raw.count(b"PK\x06\x06") >= 1 and raw.count(b"PK\x06\x07") >= 1 and raw[-22:-18] == b"PK\x05\x06"
    with zipfile.ZipFile(forced) as z:
        assert z.testzip() is None
        for info in z.infolist():
            assert info.extract_version == 45 and info.extra[:4] == struct.pack("<HH", 1, 24)
    a, b = np.load(plain), np.load(forced)
    assert a.files == b.files
    for name in a.files:
        assert a[name].dtype == b[name].dtype and a[name].shape == b[name].shape and a[name].tobytes() == b[name].tobytes(), name


def test_npz_record_beyond_4_gib(built_library, tmp_path, monkeypatch):
    """A 33-frequency 4100^2 image (4.4 GB in one I_nu record; config 5's 4096^2 x 64 is twice that): the reference stops
    with "too large for ZIP" (numpy_format.cpp:44-45) - kept under BLACKLIGHT_AMD_ZIP64=never - and the ZIP64 writer hands
    numpy the array. The writer checksums and writes the rows where they lie, so the only large thing here is the file
    (the image is untouched zero pages with a few marked values)."""
    import blacklight_amd as bl
    if os.statvfs(tmp_path).f_bavail * os.statvfs(tmp_path).f_frsize < 6 << 30:
        pytest.skip("needs 6 GiB of disk")
    fx, params, _ = gu.load_case("formula_dp")
    res, n_nu = 4100, 33
    q = dict(params, camera_resolution=res, image_num_frequencies=n_nu, image_frequency_start=1.0e11, image_frequency_end=1.0e12,
             image_frequency_spacing="log", adaptive_max_level=0)
    ctx = bl.Context(bl.Params.from_dict(q), device=BL_DEVICE_NONE)
    image = np.zeros((n_nu, res * res))
    marks = [(0, 0, 1.5), (0, res * res - 1, -2.25), (16, 12345, 3.0e-7), (n_nu - 1, 7, 42.0), (n_nu - 1, res * res - 1, -1.0e300)]
    for l, m, value in marks:
        image[l, m] = value
    path = tmp_path / "big.npz"
    monkeypatch.setenv("BLACKLIGHT_AMD_ZIP64", "never")
    with pytest.raises(bl.BlacklightError, match="too large for ZIP"):
        ctx.write_output([dict(image=image, block_locs=None)], path=path)
    monkeypatch.delenv("BLACKLIGHT_AMD_ZIP64")
    ctx.write_output([dict(image=image, block_locs=None)], path=path)
    assert os.path.getsize(path) > 1 << 32
    with zipfile.ZipFile(path) as z:
        info = z.getinfo("I_nu.npy")
        assert info.file_size == 128 + 8 * n_nu * res * res and info.CRC != 0
        assert [i.filename for i in z.infolist()] == ["mass_msun.npy", "width.npy", "frequency.npy", "adaptive_num_levels.npy", "I_nu.npy"]
    with np.load(path, mmap_mode=None) as got:
        assert got["frequency"].shape == (n_nu,) and got["width"].shape == (1,)
    # numpy's own reader on the big member, without holding 4.4 GB: header through numpy, values by position
    with zipfile.ZipFile(path) as z, z.open("I_nu.npy") as member:
        version = np.lib.format.read_magic(member)
        shape, fortran, dtype = np.lib.format.read_array_header_1_0(member)
        assert version == (1, 0) and shape == (n_nu, res, res) and not fortran and dtype == np.dtype("<f8")
        for l, m, value in marks:
            member.seek(128 + 8 * (l * res * res + m))
            assert struct.unpack("<d", member.read(8))[0] == value
        member.seek(128 + 8 * (5 * res * res))
        assert member.read(1 << 20) == bytes(1 << 20)
    with zipfile.ZipFile(path) as z, z.open("I_nu.npy") as member:
        while member.read(1 << 26):     # read to the end: zipfile compares the CRC-32 of 4.4 GB with the header's
            pass


def test_the_arithmetic_tier_from_the_environment_is_parsed_strictly(built_library, monkeypatch):
    """BLACKLIGHT_AMD_ARITHMETIC = exact | tolerant in any case; anything else fails bl_init with the reason (a typo used to
    select the tolerant tier silently). Same for BLACKLIGHT_AMD_TAIL_POLICY."""
    import blacklight_amd as bl
    fx, p, ctx = _host_context("formula_dp")
    ctx.close()
    for good in ("exact", "EXACT", "Tolerant"):
        monkeypatch.setenv("BLACKLIGHT_AMD_ARITHMETIC", good)
        bl.Context(p, device=BL_DEVICE_NONE).close()
    for bad in ("exakt", "exact ", "1", ""):
        monkeypatch.setenv("BLACKLIGHT_AMD_ARITHMETIC", bad)
        with pytest.raises(bl.BlacklightError, match="BLACKLIGHT_AMD_ARITHMETIC must be exact or tolerant"):
            bl.Context(p, device=BL_DEVICE_NONE)
    monkeypatch.setenv("BLACKLIGHT_AMD_ARITHMETIC", "exact")
    monkeypatch.setenv("BLACKLIGHT_AMD_TAIL_POLICY", "quadd")
    with pytest.raises(bl.BlacklightError, match="BLACKLIGHT_AMD_TAIL_POLICY"):
        bl.Context(p, device=BL_DEVICE_NONE)


def test_large_records_are_written_by_several_threads_with_the_same_bytes(built_library, tmp_path):
    """Records of 32 MiB and more get their CRC-32 from slices computed side by side and combined (GF(2) algebra of the CRC register),
    and their bytes from several pwrite calls (bl_host.cpp): the file must be the ZIP the single-threaded writer produces - every
    member's stored CRC equal to zlib's over its bytes, the array equal to what went in, stored entries at the offsets the headers name."""
    import zlib
    import blacklight_amd as bl
    fx, params, _ = gu.load_case("formula_64")
    res = 512
    q = dict(params, camera_resolution=res, image_num_frequencies=20, image_frequency_start=1.0e11, image_frequency_end=3.0e11,
             image_frequency_spacing="log")
    ctx = bl.Context(bl.Params.from_dict(q), device=BL_DEVICE_NONE)
    image = np.random.default_rng(5).standard_normal((ctx.num_quantities, res * res))
    assert image.nbytes > (32 << 20)
    path = tmp_path / "large.npz"
    ctx.write_output([dict(image=image, block_locs=None)], path=path)
    ctx.close()
    with zipfile.ZipFile(path) as z:
        assert z.testzip() is None
        info = z.getinfo("I_nu.npy")
        assert info.compress_type == zipfile.ZIP_STORED and info.file_size == image.nbytes + 128
        with open(path, "rb") as f:   # the member's bytes where the local header says they are
            f.seek(info.header_offset)
            local = f.read(30)
            name_len, extra_len = struct.unpack("<HH", local[26:30])
            f.seek(info.header_offset + 30 + name_len + extra_len)
            member = f.read(info.file_size)
        assert zlib.crc32(member) == info.CRC
        assert member[128:] == image.tobytes()
    assert np.array_equal(np.load(path)["I_nu"].reshape(image.shape), image)
