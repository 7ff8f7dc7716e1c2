"""IndexSequence.Write / Read (IndexSequence.cs:167-188): i32 unit count + little-endian ushort units -- host-side
serialisation of the mirror class, no GPU involved."""
import io
import struct

import numpy as np


def test_write_read_round_trip_and_short_reads():
    import docodo_amd
    units = np.array([5, 0x8001, 0x7FFF, 0xFFFF, 0x0003, 42], np.uint16)
    s = docodo_amd.IndexSequence(units, R=-7)
    buf = io.BytesIO()
    s.Write(buf)
    raw = buf.getvalue()
    assert raw[:4] == struct.pack("<i", 6) and raw[4:] == units.astype("<u2").tobytes()
    t = docodo_amd.IndexSequence()
    t.Read(io.BytesIO(raw))
    assert np.array_equal(t.units, units) and t.Count == 6 and t[1] == 0x8001
    # a truncated stream leaves the sequence as it was (the reference swallows the exception, IndexSequence.cs:184-187)
    u = docodo_amd.IndexSequence([1, 2, 3])
    u.Read(io.BytesIO(raw[:9]))
    assert list(u.units) == [1, 2, 3]
    # order: R < 0 <=> "exact"; the setter keeps the sign of R whatever is assigned (IndexSequence.cs:161)
    assert s.order and not docodo_amd.IndexSequence(units, R=3).order
    s.order = False
    assert s.R == -7
