"""Pins the ebur128 oracle on published known answers: the EBU Tech 3341 (loudness, +-0.1 LU) and
Tech 3342 (loudness range, +-1 LU) minimum-requirement test signals — synthetic 1 kHz sines with
specified levels and durations — plus the BS.1770 K-weighting coefficients at 48 kHz.
The crate the reference calls (ebur128 0.1.10) is not vendored: parity with it is unpinned."""
import numpy as np
import pytest


def sine(db, secs, rate=48000, ch=2, f=1000.0):
    t = np.arange(int(round(secs * rate))) / rate
    s = (10 ** (db / 20.0)) * np.sin(2 * np.pi * f * t)
    return np.repeat(s[:, None], ch, axis=1).reshape(-1)


def test_k_weighting_coefficients_48k(oracle):
    """BS.1770-4 Table 1/2 stage coefficients at 48 kHz, convolved: b0 = 1.53512485958697 * 1.0, a1 = sum of a1s..."""
    b, a = oracle.EbuR128(2, 48000).filter_coeffs()
    s1b, s1a = [1.53512485958697, -2.69169618940638, 1.19839281085285], [1.0, -1.69065929318241, 0.73248077421585]
    s2b, s2a = [1.0, -2.0, 1.0], [1.0, -1.99004745483398, 0.99007225036621]
    assert np.allclose(b, np.convolve(s1b, s2b), rtol=0, atol=2e-7)
    assert np.allclose(a, np.convolve(s1a, s2a), rtol=0, atol=2e-7)


@pytest.mark.parametrize("db", [-23.0, -33.0])
def test_tech3341_case1_2(oracle, db):
    e = oracle.EbuR128(2, 48000)
    e.add_frames(sine(db, 20).astype(np.float32))
    assert abs(e.loudness_momentary() - db) <= 0.1
    assert abs(e.loudness_shortterm() - db) <= 0.1
    assert abs(e.loudness_global() - db) <= 0.1
    assert abs(e.sample_peak(0) - 10 ** (db / 20)) < 1e-6 and abs(e.true_peak(1) - 10 ** (db / 20)) < 2e-4


@pytest.mark.parametrize("parts", [
    [(-36, 10), (-23, 60), (-36, 10)],                                   # case 3
    [(-72, 10), (-36, 10), (-23, 60), (-36, 10), (-72, 10)],             # case 4
    [(-26, 20), (-20, 20.1), (-26, 20)],                                 # case 5
])
def test_tech3341_gating_cases(oracle, parts):
    e = oracle.EbuR128(2, 48000)
    e.add_frames(np.concatenate([sine(db, s) for db, s in parts]))
    assert abs(e.loudness_global() - (-23.0)) <= 0.1


@pytest.mark.parametrize("a,b,lra", [(-20, -30, 10), (-20, -15, 5), (-40, -20, 20)])
def test_tech3342_loudness_range(oracle, a, b, lra):
    e = oracle.EbuR128(2, 48000)
    e.add_frames(np.concatenate([sine(a, 20), sine(b, 20)]))
    assert abs(e.loudness_range() - lra) <= 1.0


def test_chunking_is_transparent_and_reset_clears(oracle):
    x = (sine(-20, 7.3) * (1 + 0.3 * np.sin(np.arange(int(round(7.3 * 48000)) * 2) / 5000.0))).astype(np.float32)
    a, b = oracle.EbuR128(2, 48000), oracle.EbuR128(2, 48000)
    a.add_frames(x)
    pos = 0
    for n in (1, 479, 4800, 4801, 19200, 7, 100000):
        b.add_frames(x[pos * 2:(pos + n) * 2])
        pos += n
    b.add_frames(x[pos * 2:])
    assert a.loudness_global() == b.loudness_global() and a.loudness_momentary() == b.loudness_momentary()
    assert a.loudness_range() == b.loudness_range() and a.true_peak(0) == b.true_peak(0)
    a.reset()
    assert a.loudness_global() == -np.inf and a.relative_threshold() == -70.0 and a.sample_peak(0) == 0.0


def test_channel_weights_and_unused(oracle):
    x6 = sine(-23, 5, ch=6)
    full = oracle.EbuR128(6, 48000, channel_classes=[1, 1, 1, 0, 2, 2])   # L R C LFE Ls Rs
    full.add_frames(x6)
    exp = -23.0 + 10 * np.log10((3 + 2 * 1.41) / 2.0)                      # relative to the stereo reading
    assert abs(full.loudness_momentary() - exp) <= 0.1
    mono = oracle.EbuR128(1, 48000, channel_classes=[3])                    # dual mono counts twice
    mono.add_frames(sine(-23, 5, ch=1))
    assert abs(mono.loudness_momentary() - (-23.0)) <= 0.1


# ---- EBU Tech 3341 / 3342 as committed data (tests/golden/ebu_tech_334x.json): third-party conformance numbers for the BS.1770
# chain that do not depend on anybody's memory of the ebur128 crate. The same cases run on the device (tests/test_gpu_ebur128.py).
from ebu_cases import check_case, load_cases  # noqa: E402

_RATE, _CASES = load_cases()


class _OracleMeter:
    def __init__(self, oracle, case, rate):
        self.e = oracle.EbuR128(case["channels"], rate, 63, case.get("channel_class"))
    add = lambda self, x: self.e.add_frames(x)
    momentary = lambda self: self.e.loudness_momentary()
    shortterm = lambda self: self.e.loudness_shortterm()
    integrated = lambda self: self.e.loudness_global()
    lra = lambda self: self.e.loudness_range()
    true_peak = lambda self, c: self.e.true_peak(c)


@pytest.mark.parametrize("case", _CASES, ids=[c["id"] for c in _CASES])
def test_ebu_tech_334x_conformance_oracle(oracle, synth, case):
    check_case(case, _RATE, _OracleMeter(oracle, case, _RATE), synth)


def test_ebu_golden_file_is_complete_and_says_what_it_leaves_out():
    import json
    from ebu_cases import GOLDEN
    doc = json.load(open(GOLDEN))
    ids = [c["id"] for c in doc["cases"]]
    assert ids == sorted(set(ids), key=ids.index) and len(ids) == 17
    for c in doc["cases"]:
        assert c["doc"].startswith("EBU Tech 334") and ("tolerance_lu" in c or "tolerance_db" in c)
    # every case number of the two documents is either a recipe here or named with the reason it is not
    left_out = " ".join(doc["not_included"])
    for n in (7, 8, 10, 11, 13, 14, 20, 23):
        assert "3341-%d" % n in left_out
    for n in (5, 6):
        assert "3342-%d" % n in left_out
