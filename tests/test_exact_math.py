"""CPU-only checks of the exactness arguments behind the FAST device arithmetic: the same operation
sequences replayed in C (tests/cpu_replay/fast_path_replay.c) against IEEE division / the oracle.
The GPU parity tests repeat the end-to-end comparison on the real hardware reciprocal."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def replay():
    src = os.path.join(HERE, "cpu_replay", "fast_path_replay.c")
    out = os.path.join(HERE, "cpu_replay", "libreplay.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-o", out, src, "-lm"])
    L = C.CDLL(out)
    for name in ("replay_div255", "replay_div65535", "replay_div60"):
        getattr(L, name).restype = C.c_float
        getattr(L, name).argtypes = [C.c_float]
    L.replay_hsvfilter_fast.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_float), C.c_int]
    L.replay_hsvfilter_fast2.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_float), C.c_int]
    L.replay_float_to_u8_fast.restype = C.c_uint32
    L.replay_float_to_u8_fast.argtypes = [C.c_float]
    L.replay_float_to_u8_ref.restype = C.c_uint32
    L.replay_float_to_u8_ref.argtypes = [C.c_float]
    return L


def test_div255_and_div65535_are_exact(replay):
    for n in range(256):
        assert np.float32(replay.replay_div255(float(n))) == np.float32(n) / np.float32(255.0)
    xs = np.arange(65536, dtype=np.float32)
    ref = xs / np.float32(65535.0)
    got = np.array([replay.replay_div65535(float(x)) for x in xs[::7]], np.float32)
    assert (got == ref[::7]).all()


def test_div60_exact_on_fast_domain(replay):
    """RN(h/60) for h == 0 or 1e-30 <= h <= 360 — the hue values the FAST path can see (hue-shift with
    |shift| < 1e-30 takes the GENERIC kernel). Dense sample: every 4096th float + the binade edges."""
    lo = np.float32(1e-30).view(np.uint32)
    hi = np.float32(360.0).view(np.uint32)
    bits = np.concatenate([np.arange(lo, hi + 1, 4096, dtype=np.uint64), np.arange(hi - 5000, hi + 1, dtype=np.uint64)]).astype(np.uint32)
    hs = np.concatenate([bits.view(np.float32), np.array([0.0], np.float32)])
    ref = hs / np.float32(60.0)
    got = np.array([replay.replay_div60(float(h)) for h in hs], np.float32)
    assert (got == ref).all()


@pytest.mark.parametrize("setting", ["defaults", "hue90", "mixed", "neg"])
def test_fast_hsvfilter_algorithm_matches_oracle_on_all_colours(replay, oracle, synth, setting):
    """The whole FAST pixel algorithm (rotation, 2-op constant divisions, reciprocal + one residual step,
    sign-bit wraps, floor-indexed sextant select) == the oracle on all 2^24 colours, given a correctly
    rounded reciprocal. (The real v_rcp_f32 is checked by the GPU all-colours tests.)"""
    st = synth.HSV_SETTINGS.get(setting) or (-200.25, 0.8, 0.1, 1.1, -0.05)
    ac = synth.allcolors()
    exp = ac.copy().reshape(-1)
    oracle.hsvfilter(exp, 4096, 4096 * 4, 4, 0, False, st, nthreads=8)
    got = ac.copy().reshape(-1).view(np.uint32)
    s = (C.c_float * 5)(*[float(v) for v in st])
    replay.replay_hsvfilter_fast(got.ctypes.data, got.size, s, 0)
    assert (got.view(np.uint8) == exp).all()


R3_SETTINGS = {"neg": (-200.25, 0.8, 0.1, 1.1, -0.05), "wide+": (725.5, 1.0, 0.0, 1.0, 0.0), "wide-": (-400.5, 1.1, 0.0, 0.9, 0.01),
               "just over 360": (360.00003, 1.0, 0.0, 1.0, 0.0), "just under -360": (-360.00003, 1.0, 0.0, 1.0, 0.0),
               "2^22": (4194304.0, 1.0, 0.0, 1.0, 0.0), "-2 x 360": (-720.0, 1.0, 0.0, 1.0, 0.0)}


@pytest.mark.parametrize("setting", ["defaults", "hue90", "mixed"] + sorted(R3_SETTINGS))
def test_round3_fast_hsvfilter_algorithm_matches_oracle_on_all_colours(replay, oracle, synth, setting):
    """The round-3 form of the FAST pixel algorithm (csrc/hsv_device.hpp: value and its correctly rounded reciprocal from
    a table, chroma + 1e-30 for the zero denominator, sign wraps as one unsigned minimum, hp - odd for hp % 2 - 1, exact
    one-fma fmod for 360 < |hue-shift| <= 2^22) == the oracle on all 2^24 colours."""
    st = synth.HSV_SETTINGS.get(setting) or R3_SETTINGS[setting]
    ac = synth.allcolors()
    exp = ac.copy().reshape(-1)
    oracle.hsvfilter(exp, 4096, 4096 * 4, 4, 0, False, st, nthreads=8)
    got = ac.copy().reshape(-1).view(np.uint32)
    s = (C.c_float * 5)(*[float(v) for v in st])
    replay.replay_hsvfilter_fast2(got.ctypes.data, got.size, s, 0)
    assert (got.view(np.uint8) == exp).all()


@pytest.mark.parametrize("rcp_mode", [1, -1])
def test_fast_division_sensitivity_to_reciprocal_error(replay, oracle, synth, rcp_mode):
    """Documents WHY the GPU all-colours test is the gate for div_rcp_refine: with a reciprocal that is a
    full ulp off, rcp + one residual step mis-rounds a handful of the 2^24 colours (57-68), so the
    sequence is exact only for the reciprocal the hardware actually returns on the operands the HSV
    conversion produces (all of which the all-colours frame contains)."""
    st = synth.HSV_SETTINGS["defaults"]
    ac = synth.allcolors()
    exp = ac.copy().reshape(-1)
    oracle.hsvfilter(exp, 4096, 4096 * 4, 4, 0, False, st, nthreads=8)
    got = ac.copy().reshape(-1).view(np.uint32)
    replay.replay_hsvfilter_fast(got.ctypes.data, got.size, (C.c_float * 5)(*st), rcp_mode)
    bad = int((got.view(np.uint8).reshape(-1, 4) != exp.reshape(-1, 4)).any(axis=1).sum())
    assert 0 < bad < 200


def test_round_half_away_via_floor_plus_half(replay):
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.uniform(-0.2, 1.2, 200000).astype(np.float32),
                           ((np.arange(0, 256) + 0.5) / 255.0).astype(np.float32),
                           np.nextafter(((np.arange(0, 256) + 0.5) / 255.0).astype(np.float32), np.float32(0)),
                           np.array([0.0, 1.0, 0.0019607842, 0.00196078419], np.float32)])
    for v in vals:
        assert replay.replay_float_to_u8_fast(float(v)) == replay.replay_float_to_u8_ref(float(v))
