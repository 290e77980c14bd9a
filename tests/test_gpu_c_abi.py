"""The C ABI used from plain C: tests/c_abi/chain_example.c is compiled with gcc -std=c99 against include/mi355fx.h and
linked to libmi355fx.so exactly as a reference-side shim would; its output is compared with the CPU
oracle on the same inputs (regenerated here with the same LCG)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gst-plugins-rs_amd")
SRC = os.path.join(ROOT, "tests", "c_abi", "chain_example.c")
W, H, S = 640, 360, 17


def _build(tmp_path):
    exe = str(tmp_path / "chain_example")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), SRC,
                           "-L", PKG, "-lmi355fx", "-Wl,-rpath," + PKG, "-o", exe])
    return exe


def test_c_example_compiles_as_c99(tmp_path, mi355lib):
    """CPU check: the header is valid C99 and every symbol the example uses links."""
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_c_example_matches_oracle(tmp_path, mi355lib, oracle):
    exe = _build(tmp_path)
    out_path = str(tmp_path / "out.bin")
    r = subprocess.run([exe, out_path], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    got = np.fromfile(out_path, dtype=np.uint8)
    # same inputs as the C program
    n = W * H * 4
    seed = np.uint32(12345)
    vals = np.empty(n, np.uint8)
    s = int(seed)
    for i in range(n):  # LCG, 32-bit wrap
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        vals[i] = (s >> 8) & 255
    g = np.arange(S, dtype=np.float32) / np.float32(S - 1)
    table = np.zeros((S, S, S, 4), np.float32)  # [z][y][x]
    r_, g_, b_ = g[None, None, :], g[None, :, None], g[:, None, None]
    table[..., 0] = np.float32(0.75) * r_ + np.float32(0.25) * g_
    table[..., 1] = np.float32(0.75) * g_ + np.float32(0.25) * b_
    table[..., 2] = np.float32(0.75) * b_ + np.float32(0.25) * r_
    table[..., 3] = 1.0
    cube = oracle.Cube.from_table(True, S, table.reshape(-1, 4))
    st = (45.0, 1.25, -0.05, 0.9, 0.02)
    mid = vals.copy()
    oracle.hsvfilter(mid, W, W * 4, 4, 0, False, st, nthreads=4)
    exp = np.zeros_like(mid)
    oracle.colorlut_rgba8(cube, mid, W * 4, exp, W * 4, W, H, nthreads=4)
    assert got.size == exp.size and (got == exp).all(), int((got != exp).sum())
