"""Minimal PLY reader/writer for 3DGS point clouds (the data format on the input side of the render path:
/root/reference/src/gs/gaussian_model.py:231-288 reads it with ``plyfile``, which is not available here).
Supports ``binary_little_endian`` and ``ascii`` vertex elements with scalar properties."""
from __future__ import annotations

from pathlib import Path

import numpy as np

_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
          "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
          "double": "f8", "float64": "f8"}


def read_ply_vertices(path) -> np.ndarray:
    """Returns the 'vertex' element as a numpy structured array."""
    data = Path(path).read_bytes()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    header = data[:end].decode("ascii").splitlines()
    if header[0].strip() != "ply":
        raise ValueError("not a PLY file")
    fmt = None
    elements = []   # (name, count, [(prop, dtype)])
    for line in header[1:]:
        tok = line.split()
        if not tok or tok[0] == "comment":
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            elements.append((tok[1], int(tok[2]), []))
        elif tok[0] == "property":
            if tok[1] == "list":
                raise ValueError("list properties are not supported")
            elements[-1][2].append((tok[2], _TYPES[tok[1]]))
    if fmt not in ("binary_little_endian", "ascii"):
        raise ValueError(f"unsupported PLY format {fmt}")
    off = end
    for name, count, props in elements:
        dt = np.dtype([(p, "<" + t) for p, t in props])
        if fmt == "binary_little_endian":
            arr = np.frombuffer(data, dtype=dt, count=count, offset=off)
            off += count * dt.itemsize
        else:
            text = data[off:].decode("ascii").split("\n")
            rows = [tuple(float(x) for x in text[i].split()) for i in range(count)]
            arr = np.array(rows, dtype=dt) if count else np.zeros(0, dtype=dt)
            off += sum(len(text[i]) + 1 for i in range(count))
        if name == "vertex":
            return arr
    raise ValueError("no vertex element")


def write_ply_vertices(path, columns: dict):
    """columns: ordered {property name: 1-D array}; written as float32 binary_little_endian."""
    names = list(columns)
    n = len(columns[names[0]]) if names else 0
    dt = np.dtype([(k, "<f4") for k in names])
    arr = np.empty(n, dtype=dt)
    for k in names:
        arr[k] = np.asarray(columns[k], dtype=np.float32)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n
    header += "".join(f"property float {k}\n" for k in names) + "end_header\n"
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(arr.tobytes())
