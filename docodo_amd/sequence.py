"""IndexSequence mirror (reference: Docodo.NET/IndexSequence.cs).

Holds the 15-bit var-ushort units exactly as the reference's List<ushort> does; encode / decode /
Count() / Shift / operators run through libdocodo_hip (HIP kernels), never on the CPU.
"""
import ctypes as C

import numpy as np

from . import _lib


def _ptr(a):
    return a.ctypes.data if a.size else None


class IndexSequence:
    """IndexSequence.cs:10 -- ascending ulong values stored as delta-coded ushorts; R as in :163."""

    def __init__(self, units=None, R=0):
        self.units = np.ascontiguousarray(units if units is not None else [], dtype=np.uint16)
        self.R = int(R)

    # IndexSequence.Builder().AddRange(values), IndexSequence.cs:63-90
    @classmethod
    def from_values(cls, values, R=0):
        v = np.ascontiguousarray(values, dtype=np.uint64)
        n = C.c_size_t(0)
        out = np.empty(max(1, 5 * v.size), np.uint16)
        _lib.check(_lib.lib().dcd_seq_encode(_ptr(v), v.size, out.ctypes.data, out.size, C.byref(n)))
        return cls(out[:n.value].copy(), R)

    @property
    def Count(self):  # number of UNITS (IndexSequence.cs:45)
        return int(self.units.size)

    def __getitem__(self, i):  # IndexSequence.cs:158
        return int(self.units[i])

    def values(self):  # enumerator, IndexSequence.cs:114-136
        n = C.c_size_t(0)
        out = np.empty(max(1, self.units.size), np.uint64)
        _lib.check(_lib.lib().dcd_seq_decode(_ptr(self.units), self.units.size, out.ctypes.data, out.size, C.byref(n)))
        return out[:n.value].copy()

    def __iter__(self):
        return iter(self.values().tolist())

    def count_values(self):  # LINQ Count(), Search.cs:307
        n = C.c_size_t(0)
        _lib.check(_lib.lib().dcd_seq_count(_ptr(self.units), self.units.size, C.byref(n)))
        return n.value

    def Shift(self, shift):  # IndexSequence.cs:191-202 (in place)
        out = np.empty(self.units.size + 8, np.uint16)
        n = C.c_size_t(0)
        _lib.check(_lib.lib().dcd_seq_shift(_ptr(self.units), self.units.size, int(shift), out.ctypes.data, out.size, C.byref(n)))
        self.units = out[:n.value].copy()

    def _binop(self, fn, other):
        cap = max(1, 5 * (self.units.size + other.units.size))
        out = np.empty(cap, np.uint16)
        n = C.c_size_t(0)
        R = C.c_int32(0)
        _lib.check(fn(_ptr(self.units), self.units.size, self.R, _ptr(other.units), other.units.size, other.R,
                      out.ctypes.data, out.size, C.byref(n), C.byref(R)))
        return IndexSequence(out[:n.value].copy(), R.value)

    def __mul__(self, other):  # operator *, IndexSequence.cs:218-284
        return self._binop(_lib.lib().dcd_seq_and, other)

    __and__ = __mul__          # operator &, IndexSequence.cs:205-215

    @property
    def order(self):  # IndexSequence.cs:161: "exact" sequence
        return self.R < 0

    @order.setter
    def order(self, value):  # (the reference's setter keeps the sign: `(order ? -1 : 1) * Math.Abs(R)`, the VALUE is ignored)
        self.R = (-1 if self.R < 0 else 1) * abs(self.R)

    def Write(self, stream):  # IndexSequence.cs:167-173: i32 count of ushort units + the units, little endian
        import struct
        stream.write(struct.pack("<i", int(self.units.size)))
        stream.write(np.ascontiguousarray(self.units, "<u2").tobytes())

    def Read(self, stream):  # IndexSequence.cs:175-188 (a short read leaves the sequence unchanged, as the swallowed exception does)
        import struct
        head = stream.read(4)
        if len(head) < 4:
            return
        n = struct.unpack("<i", head)[0]
        data = stream.read(2 * n) if n > 0 else b""
        if n < 0 or len(data) < 2 * n:
            return
        self.units = np.frombuffer(data, "<u2").astype(np.uint16)

    def __add__(self, other):  # operator +, IndexSequence.cs:286-322
        return self._binop(_lib.lib().dcd_seq_or, other)
