#!/usr/bin/env python3
"""Extract vertex positions from the reference's binary PLY scans into raw float32 fixtures.

Reads  /root/reference/data/**.ply  (binary_little_endian 1.0; vertex = x,y,z,nx,ny,nz float32 +
flags int32 = 28-byte records, as declared in each file's header) and writes the first 12 bytes of
every vertex record (x, y, z) to tests/golden/<name>.xyz.f32 (little-endian float32, N x 3, C order).
These are input DATA files of the reference (src/main.cpp:255-258 loads them); no source is copied.

Run in the build container only (the GPU box has no /root/reference).
"""
import os
import struct
import sys
import numpy as np

REF = "/root/reference/data"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

SCANS = {
    "bunny000": "bunny000.ply",            # ICP source   (src/main.cpp argv[1])
    "bunny045": "bunny045.ply",            # ICP target / EDT model
    # bun315/bun000.ply holds the same 40256 vertices as bunny000.ply (checked byte for byte on the
    # extracted positions), so the Go-ICP model reuses bunny000.xyz.f32.
    "bun270": "bun315/bun270.ply",         # Go-ICP data (rotated 270 deg)
}


def read_ply_xyz(path):
    with open(path, "rb") as f:
        blob = f.read()
    end = blob.index(b"end_header\n") + len(b"end_header\n")
    header = blob[:end].decode("ascii", "replace").splitlines()
    assert any(l.strip() == "format binary_little_endian 1.0" for l in header), path
    nvert, stride, in_vertex, props = None, 0, False, []
    sizes = {"float": 4, "int": 4, "uchar": 1, "double": 8, "short": 2, "uint": 4, "char": 1, "ushort": 2}
    for l in header:
        t = l.split()
        if not t:
            continue
        if t[0] == "element":
            in_vertex = t[1] == "vertex"
            if in_vertex:
                nvert = int(t[2])
        elif t[0] == "property" and in_vertex:
            props.append((t[1], t[2]))
            stride += sizes[t[1]]
    assert [p[1] for p in props[:3]] == ["x", "y", "z"] and all(p[0] == "float" for p in props[:3])
    rec = np.frombuffer(blob, dtype=np.uint8, count=nvert * stride, offset=end).reshape(nvert, stride)
    xyz = rec[:, :12].copy().view("<f4").reshape(nvert, 3)
    return xyz


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, rel in SCANS.items():
        xyz = read_ply_xyz(os.path.join(REF, rel))
        assert np.isfinite(xyz).all(), name
        dst = os.path.join(OUT, name + ".xyz.f32")
        xyz.astype("<f4").tofile(dst)
        print(f"{name}: {xyz.shape[0]} vertices -> {os.path.relpath(dst)}  "
              f"bbox min {xyz.min(0)} max {xyz.max(0)}")


if __name__ == "__main__":
    sys.exit(main())
