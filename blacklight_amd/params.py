"""Parameter block of the drop-in boundary (mirror of the reference's InputReader).

The block itself lives in C (bl_params, include/blacklight_amd.h); Python only holds the bytes and
goes through the same `key = value` grammar as a line of a .input file, so there is exactly one
parser (blacklight_amd/csrc/bl_params.cpp).
"""
import ctypes as C

from . import _capi


def _format(value):
    if isinstance(value, bool):
        return "true" if value else "false"
    if isinstance(value, float):
        return repr(value)
    return str(value)


class Params:
    def __init__(self):
        L = _capi.lib()
        self._buf = C.create_string_buffer(L.bl_params_sizeof())
        L.bl_params_clear(self._buf)
        self.num_runs = 1

    @property
    def ptr(self):
        return C.cast(self._buf, C.c_void_p)

    @classmethod
    def from_file(cls, path):
        """InputReader::Read() (reference src/input_reader/input_reader.cpp:72)."""
        self = cls()
        err = C.create_string_buffer(1024)
        runs = C.c_int(1)
        rc = _capi.lib().bl_params_read_file(self._buf, str(path).encode(), C.byref(runs), err, len(err))
        if rc != 0:
            raise _capi.BlacklightError(rc, err.value.decode())
        self.num_runs = runs.value
        return self

    @classmethod
    def from_text(cls, text):
        self = cls()
        for line in text.splitlines():
            self.set_line(line)
        return self

    @classmethod
    def from_dict(cls, mapping):
        self = cls()
        for key, value in mapping.items():
            self.set(key, value)
        return self

    def set_line(self, line):
        err = C.create_string_buffer(1024)
        rc = _capi.lib().bl_params_set_line(self._buf, line.encode(), err, len(err))
        if rc != 0:
            raise _capi.BlacklightError(rc, err.value.decode())

    def set(self, key, value):
        self.set_line(f"{key} = {_format(value)}")

    def update(self, mapping):
        for key, value in mapping.items():
            self.set(key, value)
        return self

    def copy(self):
        other = Params()
        C.memmove(other._buf, self._buf, len(self._buf))
        other.num_runs = self.num_runs
        return other

    def has(self, key):
        present = C.c_int(0)
        rc = _capi.lib().bl_params_get(self._buf, key.encode(), None, C.byref(present))
        if rc != 0:
            raise KeyError(key)
        return bool(present.value)

    def get(self, key):
        """Numeric value as stored (angles in radians, enums as their integer)."""
        value = C.c_double(0.0)
        present = C.c_int(0)
        rc = _capi.lib().bl_params_get(self._buf, key.encode(), C.byref(value), C.byref(present))
        if rc != 0:
            out = C.create_string_buffer(512)
            if _capi.lib().bl_params_get_string(self._buf, key.encode(), out, len(out)) != 0:
                raise KeyError(key)
            return out.value.decode()
        if not present.value:
            return None
        return value.value
