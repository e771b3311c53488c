#!/usr/bin/env python3
"""Robustness sweep of the native snapshot readers (blacklight_amd/csrc/bl_snapshot.cpp: its own HDF5 decoder for .athdf and
iharm3d files, the AthenaK and harm3d binary readers) on the CPU under AddressSanitizer + UBSan: the test fixtures with random
bytes flipped, fields overwritten by extreme integers, files truncated. A damaged file must end in an error message (or be read,
if the damage is harmless) - never in a read or write outside the file's mapping or the reader's arrays.

    g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -shared -pthread -Iinclude -Iblacklight_amd/csrc \\
        blacklight_amd/csrc/bl_snapshot.cpp blacklight_amd/csrc/bl_params.cpp -o /tmp/snapfuzz/libsnap_asan.so
    LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so)" ASAN_OPTIONS=detect_leaks=0 \\
        python3 tools/fuzz_snapshot_reader.py /tmp/snapfuzz/libsnap_asan.so [mutations per fixture] [seed]

Without the sanitizer build it still checks that nothing crashes the process. A tool, not a test (the suite has the readers' error
cases one by one, tests/test_snapshot_reader.py)."""
import ctypes as C
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
READER_DIR = os.path.join(REPO, "tests", "golden", "reader")
WORK = "/tmp/snapfuzz"


class GridDesc(C.Structure):   # bl_grid_desc (include/blacklight_amd.h), as in blacklight_amd/_capi.py
    _fields_ = [("n_blocks", C.c_int32), ("n_i", C.c_int32), ("n_j", C.c_int32), ("n_k", C.c_int32), ("n_var", C.c_int32), ("prim", C.c_void_p),
                ("x1f", C.c_void_p), ("x2f", C.c_void_p), ("x3f", C.c_void_p), ("x1v", C.c_void_p), ("x2v", C.c_void_p), ("x3v", C.c_void_p),
                ("ind", C.c_int32 * 9), ("gamma", C.c_double * 3), ("levels", C.c_void_p), ("locations", C.c_void_p), ("n_3_root", C.c_int32),
                ("sks_map", C.c_void_p), ("sks_map_n1", C.c_int32), ("sks_map_n2", C.c_int32), ("sks_map_geom", C.c_double * 3), ("bounds", C.c_double * 6)]


def fixtures():
    out = []
    for npz, key, file_name in (("expected.npz", "params", "series_0003.athdf"), ("expected.npz", "params", "blocks_entropy.athdf"),
                                ("expected_athenak.npz", "single_params", "athenak_single.bin"), ("expected_athenak.npz", "blocks_params", "athenak_blocks.bin"),
                                ("expected_iharm3d.npz", "plain_params", "iharm3d_mock.h5"), ("expected_harm3d.npz", "plain_params", "harm3d_mock.bin"),
                                ("expected_fmks.npz", "interp_params", "iharm3d_fmks.h5")):
        fx = np.load(os.path.join(READER_DIR, npz), allow_pickle=False)
        name = key if key in fx.files else [k for k in fx.files if k.endswith("params")][0]
        params = json.loads(str(fx[name]))
        params.update(simulation_multiple="false")
        out.append((file_name, params))
    return out


def main():
    lib_path = sys.argv[1]
    n_mut = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    os.makedirs(WORK, exist_ok=True)
    L = C.CDLL(lib_path)
    L.bl_params_sizeof.restype = C.c_size_t
    L.bl_snapshot_grid.restype = C.POINTER(GridDesc)
    L.bl_snapshot_grid.argtypes = [C.c_void_p]
    L.bl_snapshot_close.argtypes = [C.c_void_p]
    rng = np.random.default_rng(seed)
    totals = dict(opened=0, refused=0)
    for file_name, params in fixtures():
        data = np.fromfile(os.path.join(READER_DIR, file_name), dtype=np.uint8)
        path = os.path.join(WORK, "mutated_" + file_name)
        pbuf = C.create_string_buffer(L.bl_params_sizeof())
        L.bl_params_clear(pbuf)
        err = C.create_string_buffer(2048)
        for key, value in dict(params, simulation_file=path).items():
            if value is None:
                continue
            rc = L.bl_params_set_line(pbuf, f"{key} = {value}".encode(), err, C.c_size_t(len(err)))
            assert rc == 0, (key, value, err.value)
        opened = refused = 0
        for m in range(n_mut + 1):
            mutated = data.copy()
            kind = int(rng.integers(0, 5)) if m > 0 else -1   # (the first pass reads the fixture as it is)
            if kind == 0:      # a few random bytes anywhere
                at = rng.integers(0, mutated.size, int(rng.integers(1, 8)))
                mutated[at] = rng.integers(0, 256, at.size, dtype=np.uint8)
            elif kind == 1:    # random bytes in the first 4 KiB, where headers and object tables live
                at = rng.integers(0, min(4096, mutated.size), int(rng.integers(1, 16)))
                mutated[at] = rng.integers(0, 256, at.size, dtype=np.uint8)
            elif kind == 2:    # an aligned 8-byte field replaced by an extreme integer (sizes, offsets, counts)
                at = int(rng.integers(0, mutated.size // 8 - 1)) * 8
                value = [0, 1, 0xffffffff, 0x7fffffffffffffff, 0xffffffffffffffff, mutated.size, mutated.size + 1, 1 << 40][int(rng.integers(0, 8))]
                mutated[at:at + 8] = np.frombuffer(int(value).to_bytes(8, "little"), dtype=np.uint8)
            elif kind == 3:    # the same for a 4-byte field
                at = int(rng.integers(0, mutated.size // 4 - 1)) * 4
                value = [0, 1, 0xffff, 0x7fffffff, 0xffffffff, 1 << 20][int(rng.integers(0, 6))]
                mutated[at:at + 4] = np.frombuffer(int(value).to_bytes(4, "little"), dtype=np.uint8)
            elif kind == 4:    # truncated
                mutated = mutated[: int(rng.integers(0, mutated.size))]
            mutated.tofile(path)
            snap = C.c_void_p()
            rc = L.bl_snapshot_open(pbuf, 0, C.byref(snap), err, C.c_size_t(len(err)))
            if rc != 0:
                refused += 1
                assert err.value.startswith(b"Error"), err.value
                assert m > 0, (file_name, err.value)
                continue
            opened += 1
            g = L.bl_snapshot_grid(snap).contents
            # touch what the descriptor describes: every array end to end
            n_b = max(g.n_blocks, 1)
            cells = n_b * g.n_k * g.n_j * g.n_i
            assert 0 < cells < (1 << 31) and 0 < g.n_var < 64, (cells, g.n_var)
            total = float(np.nansum(np.ctypeslib.as_array(C.cast(g.prim, C.POINTER(C.c_float)), shape=(g.n_var * cells,)).astype(np.float64)))
            for ptr, n in ((g.x1f, g.n_i + 1), (g.x2f, g.n_j + 1), (g.x3f, g.n_k + 1), (g.x1v, g.n_i), (g.x2v, g.n_j), (g.x3v, g.n_k)):
                total += float(np.nansum(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(n_b * n,))))
            if g.sks_map:
                total += float(np.nansum(np.ctypeslib.as_array(C.cast(g.sks_map, C.POINTER(C.c_double)), shape=(g.sks_map_n1 * g.sks_map_n2,))))
            L.bl_snapshot_close(snap)
        print(f"{file_name}: {opened} read, {refused} refused of {n_mut + 1}", flush=True)
        totals["opened"] += opened
        totals["refused"] += refused
    print(json.dumps(totals))


if __name__ == "__main__":
    main()
