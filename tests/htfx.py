"""Reader/writer for the HTFX named-array fixture container (layout documented in oracle/htfx.h)."""
import struct

import numpy as np

_DT = {0: np.float32, 1: np.int32, 2: np.uint16, 3: np.uint8}
_CODE = {np.dtype(np.float32): 0, np.dtype(np.int32): 1, np.dtype(np.uint16): 2, np.dtype(np.uint8): 3}


def load(path):
    """Return {name: ndarray} for every array in the file."""
    out = {}
    with open(path, "rb") as f:
        buf = f.read()
    assert buf[:8] == b"HTFX0001", "not an HTFX file: %s" % path
    (count,) = struct.unpack_from("<I", buf, 8)
    off = 12
    for _ in range(count):
        name = buf[off:off + 48].split(b"\0", 1)[0].decode()
        dtype, ndim, d0, d1, d2, d3, nbytes = struct.unpack_from("<IIIIIIQ", buf, off + 48)
        off += 48 + 24 + 8
        dims = (d0, d1, d2, d3)[:ndim]
        arr = np.frombuffer(buf, dtype=_DT[dtype], count=nbytes // np.dtype(_DT[dtype]).itemsize, offset=off)
        out[name] = arr.reshape(dims).copy()
        off += nbytes + ((8 - (nbytes & 7)) & 7)
    return out


def save(path, arrays):
    """Write {name: ndarray} (dtypes f32/i32/u16/u8, ndim<=4)."""
    with open(path, "wb") as f:
        f.write(b"HTFX0001")
        f.write(struct.pack("<I", len(arrays)))
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            code = _CODE[a.dtype]
            dims = list(a.shape) + [1] * (4 - a.ndim)
            raw = a.tobytes()
            f.write(name.encode()[:47].ljust(48, b"\0"))
            f.write(struct.pack("<IIIIIIQ", code, a.ndim, *dims, len(raw)))
            f.write(raw)
            f.write(b"\0" * ((8 - (len(raw) & 7)) & 7))
