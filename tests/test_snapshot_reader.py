"""Host side: the library's own .athdf reader (bl_snapshot_open, blacklight_amd/csrc/bl_snapshot.cpp) against
files written by h5py with the reference's mock script (tests/golden/reader/*.athdf, made by
tools/make_goldens.py reader) and the arrays h5py itself reads back from them (expected.npz). The reference
reads the same files: its images of the two-file series are the GPU test in test_gpu_adaptive_cli.py."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import golden_util as gu
from blacklight_amd import BlacklightError, Params, Snapshot

READER_DIR = os.path.join(gu.GOLDEN_DIR, "reader")


@pytest.fixture(scope="module")
def expected():
    return np.load(os.path.join(READER_DIR, "expected.npz"), allow_pickle=False)


def _params(expected, **overrides):
    params = json.loads(str(expected["series_params"]))
    params.update(simulation_multiple="false", simulation_file=os.path.join(READER_DIR, "series_0003.athdf"))
    params.pop("simulation_start"), params.pop("simulation_end")
    params.update(overrides)
    return Params.from_dict({k: v for k, v in params.items() if v is not None})


def _check_arrays(snapshot, expected, stem):
    arrays = snapshot.arrays()
    assert arrays["prim"].dtype == np.float32 and np.array_equal(
        arrays["prim"].view(np.uint32), expected[f"{stem}_prim"].view(np.uint32))
    for name in ("x1f", "x2f", "x3f", "x1v", "x2v", "x3v"):
        want = expected[f"{stem}_{name}"].astype(np.float64)   # file float32 promoted to double
        assert arrays[name].dtype == np.float64 and np.array_equal(arrays[name], want), name
    levels, locations = snapshot.blocks
    assert np.array_equal(levels, expected[f"{stem}_levels"])
    assert np.array_equal(locations, expected[f"{stem}_locations"])
    assert snapshot.time == float(expected[f"{stem}_time"])
    # the MeshBlock table travels in the grid description too (inter-block interpolation reads it): same arrays,
    # and the root grid's cells along x3 (one level here: blocks along x3 times cells per block)
    d = snapshot.desc()
    import ctypes
    assert np.array_equal(np.ctypeslib.as_array(ctypes.cast(d.levels, ctypes.POINTER(ctypes.c_int32)), (d.n_blocks,)), levels)
    assert np.array_equal(np.ctypeslib.as_array(ctypes.cast(d.locations, ctypes.POINTER(ctypes.c_int32)), (d.n_blocks, 3)), locations)
    if levels.max() == 0:
        assert d.n_3_root == (int(locations[:, 2].max()) + 1) * d.n_k
    return arrays


def test_single_block_file(built_library, expected):
    with Snapshot(_params(expected)) as s:
        arrays = _check_arrays(s, expected, "series_0003")
        assert s.warnings == "" and s.time == 3.5
        assert arrays["indices"] == dict(ind_rho=0, ind_pgas=1, ind_kappa=0, ind_uu1=2, ind_uu2=3, ind_uu3=4,
                                         ind_bb1=5, ind_bb2=6, ind_bb3=7)
        d = s.desc()
        assert (d.n_blocks, d.n_i, d.n_j, d.n_k, d.n_var) == (1, 16, 12, 16, 8)


def test_blocks_and_entropy_variable(built_library, expected):
    path = os.path.join(READER_DIR, "blocks_entropy.athdf")
    p = _params(expected, simulation_file=path, plasma_model="code_kappa", simulation_kappa_name="r0")
    with Snapshot(p) as s:
        arrays = _check_arrays(s, expected, "blocks_entropy")
        names = json.loads(str(expected["blocks_entropy_variable_names"]))
        assert arrays["indices"]["ind_kappa"] == names.index("r0") == 5
        assert [arrays["indices"][k] for k in ("ind_bb1", "ind_bb2", "ind_bb3")] == [6, 7, 8]
        assert s.desc().n_blocks == 4 and s.time == 11.0
    with pytest.raises(BlacklightError, match="Unable to locate electron entropy slice of \"prim\" in data file."):
        Snapshot(_params(expected, simulation_file=path, plasma_model="code_kappa", simulation_kappa_name="s_e"))
    # without code_kappa the extra variable is carried along and ignored
    with Snapshot(_params(expected, simulation_file=path)) as s:
        assert s.desc().n_var == 9 and s.desc().ind_kappa == 0


def test_file_series(built_library, expected):
    pattern = os.path.join(READER_DIR, "series_{04d}.athdf")
    p = _params(expected, simulation_multiple="true", simulation_start=3, simulation_end=4, simulation_file=pattern)
    for snapshot, stem in ((0, "series_0003"), (1, "series_0004")):
        with Snapshot(p, snapshot) as s:
            assert s.file.endswith(stem + ".athdf")
            _check_arrays(s, expected, stem)
    with pytest.raises(BlacklightError, match="Could not open file for reading."):
        Snapshot(p, 2)
    for bad in ("series.athdf", "series_{04}.athdf", "series_{04d.athdf"):
        with pytest.raises(BlacklightError, match="Invalid simulation_file for multiple runs."):
            Snapshot(_params(expected, simulation_multiple="true", simulation_start=3, simulation_end=4,
                             simulation_file=os.path.join(READER_DIR, bad)))
    with pytest.raises(BlacklightError, match="Must have nonnegative index simulation_start."):
        Snapshot(_params(expected, simulation_multiple="true", simulation_start=-1, simulation_end=4, simulation_file=pattern))
    with pytest.raises(BlacklightError, match="Must have simulation_end at least as large as simulation_start."):
        Snapshot(_params(expected, simulation_multiple="true", simulation_start=5, simulation_end=4, simulation_file=pattern))


def test_adiabatic_indices_and_constructor_rules(built_library, expected):
    with Snapshot(_params(expected, plasma_gamma=1.5, plasma_gamma_i=1.6, plasma_gamma_e=1.3)) as s:
        d = s.desc()
        assert (d.plasma_gamma, d.plasma_gamma_i, d.plasma_gamma_e) == (1.5, 0.0, 0.0)
        assert s.warnings == "Warning: Ignoring plasma_gamma_i selection.\nWarning: Ignoring plasma_gamma_e selection.\n"
    with Snapshot(_params(expected, plasma_use_p="false", plasma_gamma=1.5, plasma_gamma_i=1.6, plasma_gamma_e=1.3)) as s:
        d = s.desc()
        assert (d.plasma_gamma, d.plasma_gamma_i, d.plasma_gamma_e) == (1.5, 1.6, 1.3) and s.warnings == ""
    with pytest.raises(BlacklightError, match="SimulationReader unable to find all needed values in input file."):
        Snapshot(_params(expected, plasma_use_p="false", plasma_gamma=1.5))
    # slow light: the constructor's rules (simulation_reader.cpp:66-82); its window of files is read by
    # bl_slow_light_read (GPU tests), a single bl_snapshot_open refuses
    slow = dict(simulation_multiple="true", simulation_start=3, simulation_end=4, slow_light_on="true", slow_chunk_size=2,
                slow_t_start=4.0, slow_dt=0.5, simulation_file=os.path.join(READER_DIR, "series_{04d}.athdf"))
    with pytest.raises(BlacklightError, match="read by bl_slow_light_read") as info:
        Snapshot(_params(expected, **slow))
    assert info.value.code == 6
    for change, message in ((dict(simulation_multiple="false"), "Must enable simulation_multiple to use slow light."),
                            (dict(slow_chunk_size=1), "Must have slow_chunk_size be at least 2."),
                            (dict(slow_chunk_size=3), "Not enough simulation files for given slow_chunk_size."),
                            (dict(slow_dt=0.0), "Must have positive time interval slow_dt."),
                            (dict(slow_dt=None), "SimulationReader unable to find all needed values in input file.")):
        with pytest.raises(BlacklightError) as info:
            Snapshot(_params(expected, **dict(slow, **change)))
        assert str(info.value) == "Error: " + message
    with pytest.raises(BlacklightError):   # an .athdf file is not a harm3d dump
        Snapshot(_params(expected, simulation_format="harm3d"))


def _patched(tmp_path, name, edit):
    data = bytearray(open(os.path.join(READER_DIR, "series_0003.athdf"), "rb").read())
    data = edit(data) or data
    path = tmp_path / name
    path.write_bytes(bytes(data))
    return str(path)


def test_angular_range_fix(built_library, expected, tmp_path):
    x2f = expected["series_0003_x2f"].astype(np.float32)

    def edit(data):
        raw = x2f.tobytes()
        at = bytes(data).find(raw)
        assert at > 0 and bytes(data).find(raw, at + 1) < 0
        data[at + len(raw) - 4: at + len(raw)] = np.float32(3.0).tobytes()

    path = _patched(tmp_path, "short_theta.athdf", edit)
    with Snapshot(_params(expected, simulation_file=path)) as s:
        assert s.warnings == ("Warning: Changing theta range from [0.0000000000000000e+00, 3.0000000000000000e+00] "
                              "to [0, pi].\n")
        got = s.arrays()["x2f"]
        assert got[0, 0] == 0.0 and got[0, -1] == 3.141592653589793 and got[0, -2] == float(x2f[0, -2])
    # Cartesian grids keep whatever the file says
    with Snapshot(_params(expected, simulation_file=path, simulation_coord="cks")) as s:
        assert s.warnings == "" and s.arrays()["x2f"][0, -1] == 3.0


def test_malformed_files(built_library, expected, tmp_path):
    def byte(offset, value):
        def edit(data):
            data[offset] = value
        return edit

    cases = [
        (byte(1, ord("X")), "Unexpected HDF5 format signature."),
        (byte(8, 2), "Unexpected HDF5 superblock version."),
        (byte(9, 1), "Unexpected HDF5 file free space storage version."),
        (byte(10, 1), "Unexpected HDF5 root group symbol table entry version."),
        (byte(12, 1), "Unexpected HDF5 shared header message format version."),
        (byte(13, 4), "Unexpected HDF5 size of offsets."),
        (byte(14, 4), "Unexpected HDF5 size of lengths."),
        (byte(56 + 16, 0), "Unexpected HDF5 root group symbol table entry cache type."),
        (lambda data: data[:4096], "Unexpected end of HDF5 file."),
        (lambda data: bytearray(bytes(data).replace(b"VariableNames", b"VariableNamez")), "Could not find needed file-level attributes."),
        (lambda data: bytearray(bytes(data).replace(b"LogicalLocations", b"LogicalLocationz")), "Could not find HDF5 dataset in file."),
        (lambda data: bytearray(bytes(data).replace(b"press", b"presz")), "Unable to locate \"press\" slice of \"prim\" in data file."),
        (lambda data: bytearray(bytes(data).replace(b"Bcc3", b"Bcc4")), "Unable to locate \"Bcc3\" slice of \"prim\" in data file."),
    ]
    for index, (edit, message) in enumerate(cases):
        path = _patched(tmp_path, f"bad_{index}.athdf", edit)
        with pytest.raises(BlacklightError) as info:
            Snapshot(_params(expected, simulation_file=path))
        assert str(info.value) == "Error: " + message, (index, str(info.value))
    with pytest.raises(BlacklightError, match="Could not open file for reading."):
        Snapshot(_params(expected, simulation_file=str(tmp_path / "absent.athdf")))
    empty = tmp_path / "empty.athdf"
    empty.write_bytes(b"")
    with pytest.raises(BlacklightError, match="Unexpected end of HDF5 file."):
        Snapshot(_params(expected, simulation_file=str(empty)))


def test_athdf_reader_sizes_nothing_after_a_damaged_count(expected, tmp_path):
    """Every 8-byte field of the file's first 12 KiB (superblock, object headers, dataspace dimensions, attribute counts) in turn
    replaced by 2^40 + 5: the file is read as before or refused with an error text - the decoder sizes its arrays after what
    the file holds, never after a count alone (a robustness sweep over mutated files, tools/fuzz_snapshot_reader.py under
    AddressSanitizer, found two counts that were allocated first and checked afterwards)."""
    import resource
    import time
    data = np.fromfile(os.path.join(READER_DIR, "series_0003.athdf"), dtype=np.uint8)
    path = str(tmp_path / "damaged.athdf")
    soft, hard = resource.getrlimit(resource.RLIMIT_AS)
    resource.setrlimit(resource.RLIMIT_AS, (8 << 30, hard))   # a terabyte-sized request fails at once instead of being tried
    try:
        refused = read = 0
        t0 = time.time()
        for at in range(0, 12 * 1024, 8):
            damaged = data.copy()
            damaged[at:at + 8] = np.frombuffer(((1 << 40) + 5).to_bytes(8, "little"), dtype=np.uint8)
            damaged.tofile(path)
            try:
                with Snapshot(_params(expected, simulation_file=path)):
                    read += 1
            except BlacklightError as exc:
                assert str(exc).startswith("Error: "), str(exc)
                assert "alloc" not in str(exc).lower(), (at, str(exc))
                refused += 1
        assert refused > 20 and read > 20 and time.time() - t0 < 120.0
    finally:
        resource.setrlimit(resource.RLIMIT_AS, (soft, hard))


# ---------------------------------------------------------------------------------------------- AthenaK dumps
@pytest.fixture(scope="module")
def athenak():
    return np.load(os.path.join(READER_DIR, "expected_athenak.npz"), allow_pickle=False)


def _athenak_params(athenak, stem, **overrides):
    params = json.loads(str(athenak[f"{stem}_params"]))
    params["simulation_file"] = os.path.join(READER_DIR, stem + ".bin")
    params.update(overrides)
    return Params.from_dict({k: v for k, v in params.items() if v is not None})


@pytest.mark.parametrize("stem", ["athenak_single", "athenak_blocks"])
def test_athenak_files(built_library, athenak, stem):
    """simulation_format = athenak (simulation_reader.cpp:915-1131, :434-589): 4- and 8-byte locations and variables,
    variables in any order, an entropy variable, several MeshBlocks; the arrays the reader hands over against what
    tools/make_goldens.py put into the file, in the reference's internal order (rho, v, p = (gamma - 1) e_int, B, kappa)."""
    names = json.loads(str(athenak[f"{stem}_names"]))
    order = athenak[f"{stem}_order"]
    nbi, nbj, nbk = (int(v) for v in athenak[f"{stem}_blocks"])
    bounds = athenak[f"{stem}_bounds"]
    gamma = float(athenak[f"{stem}_gamma"])
    with Snapshot(_athenak_params(athenak, stem)) as s:
        arrays = s.arrays()
        d = s.desc()
        assert s.time == float(athenak[f"{stem}_time"])
        assert s.warnings == str(athenak[f"{stem}_B_warnings"])
        assert d.plasma_gamma == gamma
        levels, locations = s.blocks
    assert np.array_equal(levels, np.zeros(len(order), dtype=np.int32)) and np.array_equal(locations, order)
    n_k, n_j, n_i = athenak["source_dens"].shape
    ni, nj, nk = n_i // nbi, n_j // nbj, n_k // nbk
    internal = ["dens", "velx", "vely", "velz", "eint", "bcc1", "bcc2", "bcc3"] + (["s_00"] if "s_00" in names else [])
    assert arrays["prim"].shape == (len(internal), len(order), nk, nj, ni)
    assert arrays["indices"] == dict(ind_rho=0, ind_uu1=1, ind_uu2=2, ind_uu3=3, ind_pgas=4, ind_bb1=5, ind_bb2=6, ind_bb3=7,
                                     ind_kappa=8 if "s_00" in names else 0)
    for b, (bi, bj, bk) in enumerate(order):
        for v, name in enumerate(internal):
            want = athenak[f"source_{name}"][bk * nk:(bk + 1) * nk, bj * nj:(bj + 1) * nj, bi * ni:(bi + 1) * ni]
            if name == "eint":
                want = (want * np.float32(gamma - 1.0)).astype(np.float32)
            assert np.array_equal(arrays["prim"][v, b].view(np.uint32), want.view(np.uint32)), (name, b)
        for axis, (n_blocks_axis, n, index) in enumerate(((nbi, ni, bi), (nbj, nj, bj), (nbk, nk, bk))):
            edges = np.linspace(bounds[2 * axis], bounds[2 * axis + 1], n_blocks_axis + 1)
            lo, hi = edges[index], edges[index + 1]
            if stem == "athenak_single":   # 4-byte locations
                lo, hi = float(np.float32(lo)), float(np.float32(hi))
            faces = arrays[f"x{axis + 1}f"][b]
            dx = (hi - lo) / n
            want_faces = np.array([lo] + [lo + i * dx for i in range(1, n)] + [hi])
            assert np.array_equal(faces, want_faces)
            assert np.array_equal(arrays[f"x{axis + 1}v"][b], 0.5 * (want_faces[:-1] + want_faces[1:]))


def test_athenak_malformed_files(built_library, athenak, tmp_path):
    """The reference's error texts for files it rejects."""
    good = open(os.path.join(READER_DIR, "athenak_single.bin"), "rb").read()

    def failing(data, **overrides):
        path = tmp_path / "bad.bin"
        path.write_bytes(data)
        with pytest.raises(BlacklightError) as err:
            Snapshot(_athenak_params(athenak, "athenak_single", simulation_file=str(path), **overrides))
        return str(err.value)

    assert failing(good.replace(b"version=1.1", b"version=1.0", 1)) == "Error: Unknown AthenaK file format."
    assert failing(good.replace(b"  time=", b"  tyme=", 1)) == "Error: Invalid AthenaK file header."
    assert failing(good.replace(b"size of location=4", b"size of location=2", 1)) == "Error: Unsupported size of location."
    assert failing(good.replace(b" eint ", b" etot ", 1)) == "Error: Unable to locate \"eint\" values in data file."
    assert failing(good.replace(b"gamma = ", b"gamna = ", 1)) == "Error: Missing adiabatic index."
    assert failing(good.replace(b"eos = ideal", b"eos : ideal", 1)) == "Error: Error parsing inputs in AthenaK file."
    assert failing(good, plasma_model="code_kappa", simulation_kappa_name="s_00") == \
        "Error: Unable to locate electron entropy values in data file."
    header_end = good.index(b"gamma = ")
    assert "Error:" in failing(good[:header_end + 40])   # data cut off
    # a mismatch between the input's adiabatic index and the file's: the reference's warning, the input's value kept
    with Snapshot(_athenak_params(athenak, "athenak_single", plasma_gamma=1.5)) as s:
        assert "Warning: Given total adiabatic index of 1.5 does not match file value of 1.66667; ignoring the latter.\n" in s.warnings
        assert s.desc().plasma_gamma == 1.5


# ---------------------------------------------------------------------------------------------- iharm3d dumps
@pytest.fixture(scope="module")
def iharm3d():
    return np.load(os.path.join(READER_DIR, "expected_iharm3d.npz"), allow_pickle=False)


def _iharm3d_params(iharm3d, case, **overrides):
    params = json.loads(str(iharm3d[f"{case}_params"]))
    params["simulation_file"] = os.path.join(READER_DIR, "iharm3d_mock.h5")
    params.update(overrides)
    return Params.from_dict({k: v for k, v in params.items() if v is not None})


def test_iharm3d_file_against_its_athena_twin(built_library, iharm3d):
    """simulation_format = iharm3d, modified Kerr-Schild coordinates (simulation_reader.cpp:354-428, :622-656, :782-807;
    simulation_geometry.cpp:29-83, :95-230). The reference's mock script writes the same fields once as an Athena++ file
    (normal-frame velocity and field on the spherical Kerr-Schild basis) and once as an iharm3d dump (internal energy,
    components on the log-r / x2 basis): after the reader's conversions the two must agree to single precision. The
    bit-level pin is the reference's image of this file (tests/test_gpu_adaptive_cli.py)."""
    with Snapshot(_iharm3d_params(iharm3d, "plain")) as s:
        arrays = s.arrays()
        d = s.desc()
        assert s.time == 0.0 and s.warnings == ""
        assert d.plasma_gamma == 13.0 / 9.0
        levels, locations = s.blocks
    assert np.array_equal(levels, [0]) and np.array_equal(locations, [[0, 0, 0]])
    assert arrays["indices"] == dict(ind_rho=0, ind_pgas=1, ind_kappa=0, ind_uu1=2, ind_uu2=3, ind_uu3=4, ind_bb1=5, ind_bb2=6, ind_bb3=7)
    twin = iharm3d["twin_prim"]
    assert arrays["prim"].shape == (8, 1) + twin.shape[1:]
    for name in ("x1f", "x2f", "x3f", "x1v", "x2v", "x3v"):
        # the twin holds float32 coordinates, the iharm3d header doubles: agreement to single precision
        # (cell centres of the log-spaced radial grid are geometric means here, arithmetic ones in the twin)
        tolerance = 1.0e-2 if name == "x1v" else 1.0e-6
        assert np.allclose(arrays[name][0], iharm3d[f"twin_{name}"], rtol=tolerance, atol=1.0e-6), name
    # scalars are the same numbers; vector components pass through the radial Jacobian at the reader's cell centres
    # (geometric means of the faces, the script's are arithmetic: 0.5 % apart on this coarse grid)
    scale = np.abs(twin).reshape(8, -1).max(axis=1)
    for v in range(8):
        tolerance = 2.0e-6 if v < 2 else 1.0e-2
        assert np.abs(arrays["prim"][v, 0] - twin[v]).max() <= tolerance * scale[v] + 1.0e-12, v


def test_iharm3d_messages(built_library, iharm3d, tmp_path):
    with Snapshot(_iharm3d_params(iharm3d, "spin")) as s:   # the reader's part of the reference's stderr for this case
        assert s.warnings == ("Warning: Given spin of 0.5 does not match file value of 0; ignoring the latter.\n"
                              "Warning: Given total adiabatic index of 1.5 does not match file value of 1.44444; ignoring the latter.\n")
        d = s.desc()
        assert (d.plasma_gamma, d.plasma_gamma_i, d.plasma_gamma_e) == (1.5, 1.6666666666666667, 1.3333333333333333)
    with pytest.raises(BlacklightError, match="Could not find ion adiabatic index in input or data file."):
        Snapshot(_iharm3d_params(iharm3d, "spin", plasma_gamma_i=None))
    with pytest.raises(BlacklightError, match="Invalid simulation_coord for Harm format."):
        Snapshot(_iharm3d_params(iharm3d, "plain", simulation_coord="cks"))
    with pytest.raises(BlacklightError, match="header/metric is FMKS or MMKS"):
        Snapshot(_iharm3d_params(iharm3d, "plain", simulation_coord="fmks"))
    with pytest.raises(BlacklightError, match="electron entropy slice"):
        Snapshot(_iharm3d_params(iharm3d, "plain", plasma_model="code_kappa", simulation_kappa_name="KEL"))


# ---------------------------------------------------------------------------------------------- harm3d dumps
def test_harm3d_file_against_its_athena_twin(built_library):
    """simulation_format = harm3d (simulation_reader.cpp:360-361, :661-716, :808-846; ConvertPrimitives4,
    simulation_geometry.cpp:242-311): the reference's mock script with --format harm3d. As for iharm3d, the converted
    arrays agree with the script's Athena++ rendition of the same fields; the bit-level pin is the reference's image."""
    fx = np.load(os.path.join(READER_DIR, "expected_harm3d.npz"), allow_pickle=False)
    params = json.loads(str(fx["plain_params"]))
    params["simulation_file"] = os.path.join(READER_DIR, "harm3d_mock.bin")
    with Snapshot(Params.from_dict(params)) as s:
        arrays = s.arrays()
        assert s.time == 0.0 and s.warnings == "" and s.desc().plasma_gamma == 13.0 / 9.0
    assert arrays["indices"] == dict(ind_rho=0, ind_pgas=1, ind_kappa=0, ind_uu1=3, ind_uu2=4, ind_uu3=5, ind_bb1=7, ind_bb2=8, ind_bb3=9)
    twin = fx["twin_prim"]
    assert arrays["prim"].shape == (10, 1) + twin.shape[1:]
    # vectors arrive as four-vectors on the modified basis; the radial Jacobian at the reader's (geometric-mean) cell
    # centres against the script's arithmetic ones leaves 0.5 % of the coordinate-frame u^r in the normal-frame one
    scale = np.abs(twin).reshape(8, -1).max(axis=1)
    scale[2:5] = max(scale[2:5].max(), 1.0)
    scale[5:8] = scale[5:8].max()
    for v, w in enumerate((0, 1, 3, 4, 5, 7, 8, 9)):
        tolerance = 2.0e-6 if v < 2 else 1.0e-2
        assert np.abs(arrays["prim"][w, 0] - twin[v]).max() <= tolerance * scale[v] + 1.0e-12, v
    params["simulation_coord"] = "cks"
    with pytest.raises(BlacklightError, match="Invalid simulation_coord for Harm format."):
        Snapshot(Params.from_dict(params))


# ---------------------------------------------------------------------------------------------- FMKS iharm3d dumps
FMKS_FILE = os.path.join(READER_DIR, "iharm3d_fmks.h5")


@pytest.fixture(scope="module")
def fmks():
    return np.load(os.path.join(READER_DIR, "expected_fmks.npz"), allow_pickle=False)


def _fmks_params(fmks, case, **overrides):
    params = json.loads(str(fmks[f"{case}_params"]))
    params["simulation_file"] = FMKS_FILE
    params.update(overrides)
    return Params.from_dict({k: v for k, v in params.items() if v is not None})


def test_fmks_reader(built_library, fmks):
    """simulation_coord = fmks on an iharm3d FMKS dump (simulation_reader.cpp:371-427; ConvertCoordinates, GenerateSKSMap,
    GetSKSCoordinates, SetJacobianFactors: simulation_geometry.cpp:36-57, :321-483): native coordinates kept, the SKS ->
    FMKS table and the grid's SKS bounds built."""
    with Snapshot(_fmks_params(fmks, "interp")) as s:
        d = s.desc()
        arrays = s.arrays()
        assert s.warnings == ""
        n1, n2 = d.sks_map_n1, d.sks_map_n2
        assert (n1, n2) == (2048, 2048) and d.sks_map
        table = np.ctypeslib.as_array(C.cast(d.sks_map, C.POINTER(C.c_double)), shape=(2, n2, n1)).copy()
        bounds = np.array(d.simulation_bounds)
        r_in, dr, dtheta = d.sks_map_r_in, d.sks_map_dr, d.sks_map_dtheta
    # native coordinates: uniform in log r and in x^2 from 0 to 1
    assert np.allclose(np.diff(arrays["x1f"][0]), arrays["x1f"][0][1] - arrays["x1f"][0][0], rtol=1e-12)
    assert arrays["x2f"][0][0] == 0.0 and abs(arrays["x2f"][0][-1] - 1.0) < 1e-14
    assert abs(np.exp(arrays["x1f"][0][0]) - r_in) < 1e-12 and abs(bounds[0] - r_in) < 1e-12
    assert abs(bounds[1] - np.exp(arrays["x1f"][0][-1])) < 1e-9 and abs(dr * (n1 - 1) - (bounds[1] - bounds[0])) < 1e-9
    assert abs(bounds[2]) < 1e-12 and abs(bounds[3] - np.pi) < 1e-12 and bounds[4] == 0.0 and bounds[5] == 2.0 * np.pi
    assert abs(dtheta * (n2 - 1) - np.pi) < 1e-14
    # x^1 = log r along a row; x^2 from 0 at the north pole to 1 at the south pole, monotonic, and the FMKS theta of
    # (x^1, x^2) is the row's theta to the bisection's tolerance
    assert np.allclose(table[0, 7], np.log(r_in + dr * np.arange(n1)), rtol=1e-13)
    assert np.all(table[1, 0] == 0.0) and np.all(table[1, -1] == 1.0) and np.all(np.diff(table[1, :, 100]) > 0)
    hslope, xt, alpha, smooth = 0.3, 0.82, 14.0, 0.5
    norm = 0.5 * np.pi * ((alpha + 1) * xt ** alpha) / ((alpha + 1) * xt ** alpha + 1)
    j = np.arange(5, n2 - 5, 37)
    x1, x2 = table[0, j, 900], table[1, j, 900]
    y = 2 * x2 - 1
    theta_g = np.pi * x2 + (1 - hslope) / 2 * np.sin(2 * np.pi * x2)
    theta_j = 0.5 * np.pi + norm * y * (1 + (y / xt) ** alpha / (alpha + 1))
    theta = theta_g + np.exp(smooth * (np.log(r_in) - x1)) * (theta_j - theta_g)
    assert np.max(np.abs(theta - j * dtheta)) < 2e-5   # the bisection stops at 1e-8 in theta, then takes one more midpoint in x^2


@pytest.mark.parametrize("case", ["interp", "nearest"])
def test_fmks_oracle_against_the_reference(built_library, fmks, case):
    """Reader + FMKS sampler of the oracle (simulation_sampling.cpp:190-198, :396-456) against the reference's images of
    the FMKS fixture: bit-exact with the pinned math library (tier B), within 1e-6 of the stock reference (whose reader
    builds the table with glibc's exp / sin / pow / log)."""
    import oracle_api
    from blacklight_amd import _capi
    p = _fmks_params(fmks, case)
    with Snapshot(p) as s:
        assert s.warnings + "Warning: Ignoring simulation_block_interp selection.\n" == str(fmks[f"{case}_B_warnings"]) \
            or "Warning: Ignoring simulation_block_interp selection.\n" + s.warnings == str(fmks[f"{case}_B_warnings"])
        out = oracle_api.render(p.ptr, s.desc(), _capi.RenderDesc, _capi.CameraFrame, n_rays=256, max_steps=2000, n_freq=1)
        out_a = oracle_api.render(p.ptr, s.desc(), _capi.RenderDesc, _capi.CameraFrame, n_rays=256, max_steps=2000, n_freq=1, variant="libm")
    rows = ["I_nu"] + (["tau"] if f"{case}_B_tau" in fmks.files else [])
    want = np.stack([fmks[f"{case}_B_{name}"].reshape(-1) for name in rows])
    assert gu.same_bits(out["image"], want).all()
    want_a = np.stack([fmks[f"{case}_A_{name}"].reshape(-1) for name in rows])
    assert np.max(np.abs(out_a["image"] - want_a) / np.max(np.abs(want_a), axis=1, keepdims=True)) < 1.0e-6


@pytest.mark.parametrize("case", ["interp", "nearest"])
def test_fmks_slow_light_oracle_against_the_reference(built_library, case):
    """slow_light_on over a series of FMKS dumps (tests/golden/reader/fmksslow_*.h5, tools/make_goldens.py fmks_slow): the
    time-slice lookup (simulation_sampling.cpp:296-349) in front of the FMKS branch (:396-456), values from the slice(s)
    (:710-786, :809-912). The oracle on the reader's slices against the reference's images of every camera time: bit-exact
    with the pinned math library."""
    import oracle_api
    from blacklight_amd import _capi
    fx = np.load(os.path.join(READER_DIR, "expected_fmks_slow.npz"), allow_pickle=False)
    params = json.loads(str(fx[f"{case}_params"]))
    params["simulation_file"] = os.path.join(READER_DIR, "fmksslow_{02d}.h5")
    p = Params.from_dict(params)
    file_times = [float(t) for t in fx["file_times"]]
    res = int(params["camera_resolution"])
    rows = ["I_nu"] + (["tau"] if f"{case}_B_0_tau" in fx.files else [])
    for image, (t_cam, files) in enumerate(gu.slow_light_windows(params, file_times)):
        snaps = [Snapshot(p, file_number=f) for f in files]
        try:
            assert [s.time for s in snaps] == [file_times[f] for f in files]
            descs = [s.desc() for s in snaps]
            out = oracle_api.render(p.ptr, descs[0], _capi.RenderDesc, _capi.CameraFrame, n_rays=res * res, max_steps=int(params["ray_max_steps"]),
                                    n_freq=1, slow=dict(grids=descs, times=[file_times[f] for f in files], snapshot_time=t_cam))
        finally:
            for s in snaps:
                s.close()
        want = np.stack([fx[f"{case}_B_{image}_{name}"].reshape(-1) for name in rows])
        assert gu.same_bits(out["image"], want).all(), (image, t_cam)
        assert out["slow_count"] == [0, 0, 0, 0]
    assert not gu.same_bits(fx[f"{case}_B_0_I_nu"], fx[f"{case}_B_2_I_nu"]).all()   # (the slices do differ)


def test_fmks_undefined_reads_are_refused(built_library, fmks):
    """Without the polar cut of the goldens the camera sees the last polar zone of the last azimuthal plane, where the
    reference's unbounded Array hands back another variable's data: the oracle (like the GPU path) refuses."""
    import oracle_api
    from blacklight_amd import _capi
    p = _fmks_params(fmks, "interp", cut_midplane_theta=0.0)
    with Snapshot(p) as s:
        with pytest.raises(RuntimeError, match="reads past"):
            oracle_api.render(p.ptr, s.desc(), _capi.RenderDesc, _capi.CameraFrame, n_rays=256, max_steps=2000, n_freq=1)
