#!/usr/bin/env python3
"""tools/cubetool.py — .cube and HRIR-sphere file tooling for the two on-disk formats of the path.

  cubetool.py cube-validate FILE        parse with the product's host reader (mirror of CubeLut::parse,
                                        video/colorlut/src/parser.rs:110-281) and print kind / size / domain / value range
  cubetool.py cube-identity N [OUT]     write an N^3 identity 3D LUT (red fastest, as the reader expects)
  cubetool.py cube-1d N GAMMA [OUT]     write an N-entry 1D gamma LUT
  cubetool.py hrir-info FILE            header, mesh and impulse-response statistics of an `hrtf`-crate sphere file
                                        ("HRIR" | u32 rate | u32 len | u32 n_vertices | u32 n_indices | indices | vertices)
  cubetool.py hrir-synth MESH LEN OUT   write a synthetic sphere on the mesh of MESH with LEN-tap responses (test material)
No GPU needed."""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))


def cube_validate(path):
    from mi355fx.cube import parse_cube_file, CubeParseError
    try:
        lut = parse_cube_file(path)
    except CubeParseError as e:
        print("INVALID: %s" % e)
        return 1
    t = lut.table.reshape(-1, 4)[:, :3] if lut.is3d else lut.table.reshape(3, -1).T
    print("%s LUT, size %d, %d entries" % ("3D" if lut.is3d else "1D", lut.size, t.shape[0]))
    print("domain scale %s offset %s" % (lut.domain_scale.tolist(), lut.domain_offset.tolist()))
    print("value range [%g, %g], finite: %s" % (np.nanmin(t), np.nanmax(t), bool(np.isfinite(t).all())))
    if lut.is3d:
        lds = lut.size <= 33 and np.isfinite(t).all()
        print("device kernel: %s" % ("LDS three-pass (plane fits the CU's 160 KB)" if lds else "gather through L2"))
    return 0


def cube_identity(n, out):
    g = np.arange(n, dtype=np.float64) / (n - 1)
    lines = ['TITLE "identity %d"' % n, "LUT_3D_SIZE %d" % n]
    for b in g:
        for gg in g:
            for r in g:
                lines.append("%.6f %.6f %.6f" % (r, gg, b))
    out.write("\n".join(lines) + "\n")


def cube_1d(n, gamma, out):
    g = np.arange(n, dtype=np.float64) / (n - 1)
    out.write("\n".join(['TITLE "gamma %g"' % gamma, "LUT_1D_SIZE %d" % n] + ["%.6f %.6f %.6f" % (v, v, v) for v in g ** gamma]) + "\n")


def hrir_info(path):
    b = open(path, "rb").read()
    if len(b) < 20 or b[:4] != b"HRIR":
        print("INVALID: bad magic")
        return 1
    rate, length, nv, ni = struct.unpack_from("<4I", b, 4)
    need = 20 + 4 * ni + nv * (12 + 8 * length)
    print("rate %d Hz, %d taps, %d vertices, %d faces, %d bytes (%s)" % (rate, length, nv, ni // 3, len(b), "complete" if len(b) >= need else "TRUNCATED, need %d" % need))
    if len(b) < need or length == 0:
        return 1
    idx = np.frombuffer(b, "<u4", ni, 20)
    print("index range [%d, %d]%s" % (idx.min(), idx.max(), "" if idx.max() < nv else "  OUT OF RANGE"))
    off = 20 + 4 * ni
    v = np.frombuffer(b, "<f4", nv * (3 + 2 * length), off).reshape(nv, 3 + 2 * length)
    radius = np.linalg.norm(v[:, :3], axis=1)
    print("vertex radius [%g, %g]  (the crate's ray reaches 10x the direction length: radii must stay below 10)" % (radius.min(), radius.max()))
    print("response peak %g, energy per vertex [%g, %g]" % (np.abs(v[:, 3:]).max(), (v[:, 3:] ** 2).sum(1).min(), (v[:, 3:] ** 2).sum(1).max()))
    return 0


def main(argv):
    if len(argv) < 2:
        print(__doc__)
        return 2
    cmd = argv[1]
    if cmd == "cube-validate":
        return cube_validate(argv[2])
    if cmd == "cube-identity":
        out = open(argv[3], "w") if len(argv) > 3 else sys.stdout
        cube_identity(int(argv[2]), out)
        return 0
    if cmd == "cube-1d":
        out = open(argv[4], "w") if len(argv) > 4 else sys.stdout
        cube_1d(int(argv[2]), float(argv[3]), out)
        return 0
    if cmd == "hrir-info":
        return hrir_info(argv[2])
    if cmd == "hrir-synth":
        from mi355fx import synth
        open(argv[4], "wb").write(synth.hrir_sphere_bytes(open(argv[2], "rb").read(), int(argv[3])))
        return 0
    print(__doc__)
    return 2


if __name__ == "__main__":
    sys.exit(main(sys.argv))
