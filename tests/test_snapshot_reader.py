"""Host side: the library's own .athdf reader (bl_snapshot_open, blacklight_amd/csrc/bl_snapshot.cpp) against
files written by h5py with the reference's mock script (tests/golden/reader/*.athdf, made by
tools/make_goldens.py reader) and the arrays h5py itself reads back from them (expected.npz). The reference
reads the same files: its images of the two-file series are the GPU test in test_gpu_adaptive_cli.py."""
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
    with pytest.raises(BlacklightError, match="Only simulation_format = athena") as info:
        Snapshot(_params(expected, simulation_format="iharm3d"))
    assert info.value.code == 3


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
