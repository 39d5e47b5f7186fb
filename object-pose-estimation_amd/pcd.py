"""Minimal PCD v0.7 reader / writer (the on-disk format either side of the path).

Handles what the reference's data uses: `DATA binary` (and `ascii`), float32 fields, `FIELDS x y z [rgb]`
(DetectAndLocalize/3DModel/*.pcd: 16 B per point, rgb packed in a float32).  The reference reads and
writes these with pcl::io::loadPCDFile / savePCDFile (rosinterface.cpp:80, BuildModel/src/main.cpp:113-153,221).
"""
from __future__ import annotations

import numpy as np


def read_pcd(path: str):
    """Returns (xyz float32 (n,3), rgb uint32 (n,) or None)."""
    with open(path, "rb") as f:
        header = {}
        while True:
            line = f.readline()
            if not line:
                raise ValueError("PCD header ended without a DATA line")
            s = line.decode("ascii", "replace").strip()
            if not s or s.startswith("#"):
                continue
            key, _, val = s.partition(" ")
            header[key.upper()] = val.split()
            if key.upper() == "DATA":
                break
        fields = header["FIELDS"]
        sizes = [int(v) for v in header["SIZE"]]
        types = header["TYPE"]
        counts = [int(v) for v in header.get("COUNT", ["1"] * len(fields))]
        n = int(header["POINTS"][0]) if "POINTS" in header else int(header["WIDTH"][0]) * int(header["HEIGHT"][0])
        kind = header["DATA"][0].lower()
        np_types = {("F", 4): "<f4", ("F", 8): "<f8", ("U", 4): "<u4", ("U", 2): "<u2", ("U", 1): "u1",
                    ("I", 4): "<i4", ("I", 2): "<i2", ("I", 1): "i1"}
        dt = np.dtype([(name, np_types[(t, sz)], (c,)) if c > 1 else (name, np_types[(t, sz)])
                       for name, t, sz, c in zip(fields, types, sizes, counts)])
        if kind == "binary":
            data = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
        elif kind == "ascii":
            raw = np.loadtxt(f, dtype=np.float64, ndmin=2)
            data = np.zeros(len(raw), dtype=dt)
            for i, name in enumerate(fields):
                data[name] = raw[:, i]
        else:
            raise ValueError(f"unsupported PCD DATA kind {kind!r} (binary_compressed is not used by the reference data)")
    xyz = np.stack([data["x"], data["y"], data["z"]], axis=1).astype(np.float32)
    rgb = None
    if "rgb" in fields:
        rgb = np.ascontiguousarray(data["rgb"]).view(np.uint32) if data["rgb"].dtype == np.float32 else data["rgb"].astype(np.uint32)
    return xyz, rgb


def write_pcd(path: str, xyz, rgb=None) -> None:
    """Binary PCD v0.7, FIELDS x y z [rgb], like pcl::io::savePCDFile(..., binary=true)."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    n = len(xyz)
    fields = "x y z" + (" rgb" if rgb is not None else "")
    k = 4 if rgb is not None else 3
    header = (f"# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS {fields}\nSIZE {' '.join(['4'] * k)}\n"
              f"TYPE {' '.join(['F'] * k)}\nCOUNT {' '.join(['1'] * k)}\nWIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\n"
              f"POINTS {n}\nDATA binary\n")
    rec = np.zeros((n, k), np.float32)
    rec[:, :3] = xyz
    if rgb is not None:
        rec[:, 3] = np.ascontiguousarray(rgb, np.uint32).view(np.float32)
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(rec.tobytes())
