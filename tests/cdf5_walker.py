"""An independent reader of NetCDF classic files in the CDF-5 variant, written from the format
grammar (netcdf-c docs/file_format_specifications.md, "The CDF-5 file format"; the grammar is quoted at
the top of tests/golden/make_cdf5_column.py) -- test infrastructure.  It shares no code with
wrf_partmc_amd/ncio.py: tests hold ncio's writer to what THIS walker reads back, and compare that with
the oracle's cells.

    walk(path) -> (dims {name: length}, gatts {name: value}, vars {name: (dims, atts, ndarray)})
Fixed-size variables only (the column files have no record dimension)."""
import struct

import numpy as np

_TYPES = {1: ("i1", 1), 2: ("S1", 1), 3: (">i2", 2), 4: (">i4", 4), 5: (">f4", 4), 6: (">f8", 8),
          7: ("u1", 1), 8: (">u2", 2), 9: (">u4", 4), 10: (">i8", 8), 11: (">u8", 8)}


class _Cursor:
    def __init__(self, buf):
        self.b, self.o = buf, 0

    def i32(self):
        v = struct.unpack_from(">i", self.b, self.o)[0]; self.o += 4
        return v

    def i64(self):
        v = struct.unpack_from(">q", self.b, self.o)[0]; self.o += 8
        return v

    def name(self):
        n = self.i64()
        s = bytes(self.b[self.o:self.o + n]).decode(); self.o += n + (-n % 4)
        return s

    def values(self, t, n):
        dt, size = _TYPES[t]
        raw = bytes(self.b[self.o:self.o + n * size]); self.o += n * size + (-(n * size) % 4)
        if t == 2:
            return raw.decode()
        return np.frombuffer(raw, dtype=dt).copy()

    def att_list(self):
        tag, n = self.i32(), self.i64()
        if tag == 0:
            assert n == 0, "ABSENT list with a count"
            return {}
        assert tag == 0x0C, f"expected NC_ATTRIBUTE, found tag {tag:#x}"
        out = {}
        for _ in range(n):
            nm = self.name(); t = self.i32(); k = self.i64()
            out[nm] = self.values(t, k)
        return out


def walk(path):
    buf = memoryview(open(path, "rb").read())
    c = _Cursor(buf)
    assert bytes(buf[:4]) == b"CDF\x05", "not a CDF-5 file"
    c.o = 4
    assert c.i64() == 0, "record variables are not expected in a column file"
    tag, n = c.i32(), c.i64()
    dims, order = {}, []
    if tag:
        assert tag == 0x0A, f"expected NC_DIMENSION, found tag {tag:#x}"
        for _ in range(n):
            nm = c.name(); ln = c.i64()
            dims[nm] = ln; order.append(nm)
    gatts = c.att_list()
    tag, n = c.i32(), c.i64()
    out = {}
    if tag:
        assert tag == 0x0B, f"expected NC_VARIABLE, found tag {tag:#x}"
        for _ in range(n):
            nm = c.name()
            nd = c.i64()
            vd = [order[c.i64()] for _ in range(nd)]
            atts = c.att_list()
            t = c.i32(); vsize = c.i64(); begin = c.i64()
            shape = [dims[d] for d in vd]
            count = int(np.prod(shape)) if shape else 1
            dt, size = _TYPES[t]
            assert vsize == count * size + (-(count * size) % 4), f"{nm}: vsize {vsize} for {count} x {size} bytes"
            assert begin + count * size <= len(buf), f"{nm}: data beyond the end of the file"
            raw = bytes(buf[begin:begin + count * size])
            arr = np.frombuffer(raw, dtype=dt).reshape(shape).copy() if t != 2 else raw.decode()
            out[nm] = (vd, atts, arr)
    return dims, gatts, out
