"""CPU tests of the hrtfrender oracle (parity unpinned: third-party `hrtf` crate; see oracle/hrtf_oracle.c).
What can be pinned here: the file format of the reference's own fixture, FFT correctness, and that the FFT
overlap-save restatement equals the exact streaming convolution it is mathematically defined as."""
import os
import struct

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "test.hrir")


def _mesh():
    return open(GOLDEN, "rb").read()


def test_reference_fixture_parses(oracle):
    """audio/hrtf/tests/test.hrir (the reference's own test data): 44.1 kHz, 1-tap all-zero HRIRs, 187 vertices,
    370 faces; total size 8200 bytes is exactly header + indices + vertices."""
    b = _mesh()
    assert len(b) == 8200
    s = oracle.HrirSphere(b, 44100)
    assert (s.len, s.vertices, s.faces) == (1, 187, 370)
    assert 20 + 4 * 1110 + 187 * (12 + 8) == 8200


def test_parse_errors(oracle):
    b = _mesh()
    with pytest.raises(ValueError):
        oracle.HrirSphere(b"XXXX" + b[4:], 44100)
    with pytest.raises(ValueError):
        oracle.HrirSphere(b[:100], 44100)
    with pytest.raises(ValueError):          # rate mismatch: resampling not restated
        oracle.HrirSphere(b, 48000)
    zero_len = b[:8] + struct.pack("<I", 0) + b[12:]
    with pytest.raises(ValueError):
        oracle.HrirSphere(zero_len, 44100)


def test_every_direction_hits_exactly_the_enclosing_face(oracle):
    """The sphere mesh is closed: any non-zero direction hits a face, the weights are a partition of unity and
    reproduce the hit point."""
    b = _mesh()
    s = oracle.HrirSphere(b, 44100)
    idx = np.frombuffer(b, "<u4", 1110, 20).reshape(-1, 3)
    pos = np.array([np.frombuffer(b, "<f4", 3, 20 + 4 * 1110 + v * 20) for v in range(187)])
    rng = np.random.default_rng(3)
    for _ in range(300):
        d = rng.standard_normal(3).astype(np.float32)
        d /= np.linalg.norm(d)
        face, uvw = s.sample(d * rng.uniform(0.5, 3.0))
        assert face >= 0
        assert abs(uvw.sum() - 1.0) < 1e-5 and (uvw > -1e-6).all()
        p = (pos[idx[face]] * uvw[:, None]).sum(0)
        assert np.allclose(p / np.linalg.norm(p), d, atol=2e-4)
    assert s.sample(np.zeros(3, np.float32))[0] == -1


@pytest.mark.parametrize("length,block", [(1, 64), (32, 64), (128, 256), (100, 77)])
def test_fft_overlap_save_equals_exact_streaming_convolution(oracle, synth, length, block):
    """FFT sizes 64, 95, 383 (prime), 176: the f32 FFT restatement stays within 2e-5 of full scale of the f64
    time-domain value over several blocks with moving sources and changing gains."""
    sphere = oracle.HrirSphere(synth.hrir_sphere_bytes(_mesh(), length), 44100)
    C, steps = 3, 4
    r = oracle.HrtfRender(sphere, C, steps, block)
    ex = oracle.HrtfExact(sphere, C, steps, block)
    rng = np.random.default_rng(length * 1000 + block)
    pos = rng.standard_normal((C, 3)).astype(np.float32)
    worst, scale = 0.0, 0.0
    for blk in range(4):
        x = rng.uniform(-1, 1, (steps * block, C)).astype(np.float32)
        gains = rng.uniform(0.2, 1.0, C).astype(np.float32)
        pos = (pos + 0.7 * rng.standard_normal((C, 3))).astype(np.float32)
        a = r.process_block(x, pos, gains)
        e = ex.process_block(x, pos, gains)
        worst = max(worst, float(np.abs(a - e).max()))
        scale = max(scale, float(np.abs(e).max()))
    assert scale > 0.1
    assert worst <= 2e-5 * max(scale, 1.0), (worst, scale)


# ---- audio/hrtf/src/spatial.rs:235-287: the reference's known answers for the coordinate conversions, replayed against
# the oracle restatement and against the product's host function (libmi355fx_host.so)
SPATIAL_KATS = [  # (from, to, input, expected): 0 Cartesian, 1 LeftHanded, 2 RightHanded
    (0, 1, (1.0, 2.0, 3.0), (-2.0, 3.0, 1.0)),    # cartesian_to_left_handed  (spatial.rs:235-252)
    (0, 2, (1.0, 2.0, 3.0), (-2.0, 3.0, -1.0)),   # cartesian_to_right_handed (spatial.rs:254-270)
    (1, 0, (1.0, 2.0, 3.0), (3.0, -1.0, 2.0)),    # left_handed_to_cartesian  (spatial.rs:272-287)
]


def test_spatial_known_answers_oracle(oracle):
    for f, t, v, exp in SPATIAL_KATS:
        assert oracle.position_convert(f, t, v) == exp


def test_spatial_known_answers_host_library():
    import ctypes as C
    from mi355fx import cube
    L = cube.load_host_library()
    L.mi355host_position_convert.restype = C.c_int
    L.mi355host_position_convert.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    for f, t, v, exp in SPATIAL_KATS:
        i, o = (C.c_float * 3)(*v), (C.c_float * 3)()
        assert L.mi355host_position_convert(f, t, i, o) == 0
        assert tuple(o) == exp
    # round trips through every pair of systems are the identity (the three maps are signed permutations)
    import itertools
    for a, b in itertools.permutations(range(3), 2):
        i, m, o = (C.c_float * 3)(0.25, -7.5, 3.0), (C.c_float * 3)(), (C.c_float * 3)()
        assert L.mi355host_position_convert(a, b, i, m) == 0 and L.mi355host_position_convert(b, a, m, o) == 0
        assert tuple(o) == tuple(i)
    assert L.mi355host_position_convert(3, 0, (C.c_float * 3)(), (C.c_float * 3)()) == -1


def test_sofa_oracle_is_a_streaming_linear_convolution(oracle):
    """The sofalizer oracle, block by block, equals one long np.convolve of the whole signal (per channel and ear), including
    a whole-sample delay; a dropped channel contributes nothing."""
    rng = np.random.default_rng(0)
    C, L, B, n_blocks = 3, 40, 64, 5
    r = oracle.SofaRenderer(C, L, B)
    hs = [(rng.standard_normal(L).astype(np.float32), rng.standard_normal(L).astype(np.float32)) for _ in range(C)]
    for c, (l, rr) in enumerate(hs):
        r.set_filter(c, l, rr, delay_left=c, delay_right=0)
    r.drop[1] = True
    x = rng.standard_normal((n_blocks * B, C)).astype(np.float32)
    g = np.array([0.5, 9.0, 2.0], np.float32)
    out = np.concatenate([r.process_block(x[k * B:(k + 1) * B], g) for k in range(n_blocks)])
    exp = np.zeros((n_blocks * B, 2))
    for c in (0, 2):
        hl = np.zeros(L); hl[c:] = hs[c][0][:L - c]
        exp[:, 0] += np.convolve(x[:, c].astype(np.float64), hl)[: n_blocks * B] * g[c]
        exp[:, 1] += np.convolve(x[:, c].astype(np.float64), hs[c][1].astype(np.float64))[: n_blocks * B] * g[c]
    assert np.abs(out - exp).max() < 1e-4


def test_resampling_preserves_a_band_limited_impulse_response(oracle):
    """Property of the method (CPU, oracle side): a low-pass impulse response sampled at 44.1 kHz and converted to 48 kHz equals
    the same continuous response sampled at 48 kHz (within the window's stop-band error)."""
    import struct
    flen, fr, dr = 400, 44100, 48000
    t = (np.arange(flen) - 100) / fr
    h = (np.sinc(2 * 6000.0 * t) * np.hanning(flen)).astype(np.float32)   # 6 kHz low-pass, far below both Nyquist rates
    mesh = open(GOLDEN, "rb").read()
    magic, rate, l0, nv, ni = struct.unpack_from("<4s4I", mesh, 0)
    body, off = [struct.pack("<4s4I", b"HRIR", fr, flen, nv, ni), mesh[20:20 + 4 * ni]], 20 + 4 * ni
    for _ in range(nv):
        body.append(mesh[off:off + 12]); off += 12 + 8 * l0
        body.append(h.tobytes()); body.append(h.tobytes())
    conv = oracle.resample_hrir_sphere_bytes(b"".join(body), dr)
    _, r2, l2, _, _ = struct.unpack_from("<4s4I", conv, 0)
    assert r2 == dr and l2 == int(np.floor(flen * dr / fr + 0.5))
    got = np.frombuffer(conv, "<f4", l2, 20 + 4 * ni + 12)
    t2 = np.arange(l2) / dr - 100 / fr
    win = np.interp(np.arange(l2) * fr / dr, np.arange(flen), np.hanning(flen))
    exp = np.sinc(2 * 6000.0 * t2) * win
    assert np.abs(got[20:-20] - exp[20:-20]).max() < 2e-3
