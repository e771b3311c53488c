#!/usr/bin/env python3
"""Robustness sweep of the .input parser (blacklight_amd/csrc/bl_params.cpp) under AddressSanitizer + UBSan on the CPU: assignments
built from the goldens' keys with hostile values - very long strings, extreme and malformed numbers, indexed keys (adaptive regions,
render images and features) with indices far outside the tables, control characters. A line is accepted or refused with an
"Error: ..." text; it never writes outside the parameter block.  Usage as tools/fuzz_snapshot_reader.py: <library> [lines] [seed]"""
import ctypes as C
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
import golden_util as gu   # noqa: E402


def main():
    L = C.CDLL(sys.argv[1])
    n_lines = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    L.bl_params_sizeof.restype = C.c_size_t
    keys = set()
    for case in gu.GPU_CASES:
        keys.update(gu.load_case(case)[1].keys())
    keys = sorted(keys)
    indexed = ["adaptive_region_{}_level", "adaptive_region_{}_x_min", "render_{}_num_features", "render_{}_{}_type", "render_{}_{}_quantity",
               "render_{}_{}_min", "render_{}_{}_rgb", "render_{}_{}_opacity", "render_{}_{}_x_values", "render_{}_{}_thresh_vals"]
    values = ["", " ", "true", "false", "0", "-1", "1e308", "-1e308", "1e-320", "nan", "inf", "-inf", "0x10", "99999999999999999999999999", "-99999999999",
              "2147483648", "4294967296", "1,2,3", "1,2", "1,2,3,4", ",,", "1;2;3", "abc", "sks", "cks", "fmks", "dp", "rk4", "plane", "pinhole",
              "a" * 300, "b" * 5000, "/" + "x" * 9000 + "/{05d}.athdf", "{d}", "{999999d}", "\x01\x02", "1e", "--1", "+-2", "1.0.0", "=", "= =", "#", "x # y"]
    pbuf = C.create_string_buffer(L.bl_params_sizeof() + 64)
    guard = bytes(pbuf[L.bl_params_sizeof():])
    err = C.create_string_buffer(4096)
    accepted = refused = 0
    for n in range(n_lines):
        if n % 500 == 0:
            L.bl_params_clear(pbuf)
        kind = int(rng.integers(0, 4))
        if kind == 0:
            key = str(rng.choice(keys))
        elif kind == 1:
            key = str(rng.choice(indexed)).format(int(rng.choice([-1, 0, 1, 2, 7, 8, 9, 15, 16, 17, 255, 65536, 2147483647, 99999999999])), int(rng.choice([-1, 0, 1, 7, 8, 9, 64, 1000000])))
        elif kind == 2:
            key = str(rng.choice(keys))[: int(rng.integers(0, 12))] + str(rng.choice(["", "_", "x", "_1", " "]))
        else:
            key = "".join(chr(int(c)) for c in rng.integers(32, 127, int(rng.integers(0, 40))))
        value = values[int(rng.integers(0, len(values)))]
        line = f"{key} {str(rng.choice(['=', '=', '=', '', '==', ' = ']))} {value}"
        rc = L.bl_params_set_line(pbuf, line.encode("latin-1"), err, C.c_size_t(len(err)))
        if rc == 0:
            accepted += 1
        else:
            refused += 1
            assert err.value.startswith(b"Error"), (line[:80], err.value)
        assert bytes(pbuf[L.bl_params_sizeof():]) == guard, line[:80]
    print(json.dumps(dict(lines=n_lines, accepted=accepted, refused=refused)))


if __name__ == "__main__":
    main()
