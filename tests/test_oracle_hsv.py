"""Pins the CPU oracle for the hsv path: the reference's own known-answer tests
(video/hsv/src/hsvutils.rs:203-279) + agreement with the independent numpy restatement +
structural facts the FAST GPU path relies on."""
import numpy as np
import pytest

EPS = 0.00001

RGB = {"white": [255, 255, 255], "black": [0, 0, 0], "red": [255, 0, 0], "green": [0, 255, 0], "blue": [0, 0, 255]}
BGR = {"white": [255, 255, 255], "black": [0, 0, 0], "red": [0, 0, 255], "green": [0, 255, 0], "blue": [255, 0, 0]}
HSV = {"white": [0.0, 0.0, 1.0], "black": [0.0, 0.0, 0.0], "red": [0.0, 1.0, 1.0], "green": [120.0, 1.0, 1.0],
       "blue": [240.0, 1.0, 1.0]}


def is_equivalent(hsv, expected, eps):
    # hsvutils.rs:202-215 (hue is circular)
    shifted = np.float32(hsv[0]) + (np.float32(180.0) - np.float32(expected[0]))
    if shifted < 0:
        shifted += np.float32(360.0)
    shifted = np.fmod(shifted, np.float32(360.0))
    return abs(shifted - 180.0) < eps and abs(hsv[1] - expected[1]) < eps and abs(hsv[2] - expected[2]) < eps


@pytest.mark.parametrize("name", list(RGB))
def test_from_rgb_kat(oracle, name):  # hsvutils.rs:238-247
    assert is_equivalent(oracle.from_rgb(RGB[name]), HSV[name], EPS)


@pytest.mark.parametrize("name", list(BGR))
def test_from_bgr_kat(oracle, name):  # hsvutils.rs:249-258
    assert is_equivalent(oracle.from_rgb(BGR[name], bgr=True), HSV[name], EPS)


@pytest.mark.parametrize("name", list(RGB))
def test_to_rgb_kat(oracle, name):  # hsvutils.rs:260-269
    assert list(oracle.to_rgb(HSV[name])) == RGB[name]


@pytest.mark.parametrize("name", list(BGR))
def test_to_bgr_kat(oracle, name):  # hsvutils.rs:271-279
    assert list(oracle.to_rgb(HSV[name], bgr=True)) == BGR[name]


def test_defaults_are_not_identity(oracle, synth):
    """the truncating cast makes default settings lossy by 1 LSB."""
    ac = synth.allcolors()
    out = ac.copy().reshape(-1)
    oracle.hsvfilter(out, 4096, 4096 * 4, 4, 0, False, synth.HSV_SETTINGS["defaults"])
    d = out.reshape(-1, 4).astype(np.int16) - ac.reshape(-1, 4).astype(np.int16)
    assert int((d[:, :3] != 0).any(axis=1).sum()) == 11093274
    assert int(np.abs(d[:, :3]).max()) == 1
    assert (d[:, 3] == 0).all()  # 4th byte untouched (hsvfilter/imp.rs:331-333)


def _sample_colours(n, seed=7):
    rng = np.random.default_rng(seed)
    edge = np.array([[a, b, c] for a in (0, 1, 127, 128, 254, 255) for b in (0, 1, 127, 128, 254, 255)
                     for c in (0, 1, 127, 128, 254, 255)], np.uint8)
    rnd = rng.integers(0, 256, size=(n, 3), dtype=np.uint8)
    grey = np.repeat(np.arange(256, dtype=np.uint8)[:, None], 3, axis=1)
    return np.concatenate([edge, grey, rnd])


@pytest.mark.parametrize("setting", ["defaults", "hue90", "mixed", "big", "nan", "inf"])
def test_c_oracle_matches_numpy_restatement(oracle, synth, setting):
    from oracle import np_restate as N
    extra = {"big": (725.5, 3.0, -0.5, 0.5, 0.6), "nan": (float("nan"), 1.0, float("nan"), 1.0, 0.0),
             "inf": (float("inf"), float("inf"), 0.0, -1.0, 2.0)}
    st = synth.HSV_SETTINGS.get(setting) or extra[setting]
    cols = _sample_colours(1 << 17)
    frame = np.zeros((cols.shape[0], 4), np.uint8)
    frame[:, :3] = cols
    frame[:, 3] = 0x5A
    buf = frame.copy().reshape(-1)
    oracle.hsvfilter(buf, cols.shape[0], cols.shape[0] * 4, 4, 0, False, st)
    r, g, b = N.hsvfilter_rgb(cols[:, 0], cols[:, 1], cols[:, 2], st)
    out = buf.reshape(-1, 4)
    assert (out[:, 0] == r).all() and (out[:, 1] == g).all() and (out[:, 2] == b).all()
    assert (out[:, 3] == 0x5A).all()


def test_hue_range_facts_fast_path_relies_on(oracle):
    """FAST GPU path: hue in [0,360) before the shift, sat/value in [0,1] — exhaustive over 2^24 via
    the per-(max,min,mid) structure is expensive in Python; check a dense sample incl. all greys/primaries."""
    cols = _sample_colours(1 << 15)
    for c in cols[::7]:
        h, s, v = oracle.from_rgb(c)
        assert 0.0 <= h < 360.0 and 0.0 <= s <= 1.0 and 0.0 <= v <= 1.0


@pytest.mark.parametrize("fmt", ["RGBx", "xRGB", "BGRx", "xBGR", "RGB", "BGR"])
def test_formats_are_byte_permutations_of_rgbx(oracle, synth, fmt):
    """All format arms of transform_frame_ip (hsvfilter/imp.rs:327-373) are the RGB arithmetic on
    permuted bytes."""
    from mi355fx import FMT_LAYOUT
    ps, first, bgr = FMT_LAYOUT[fmt]
    rng = np.random.default_rng(3)
    w, h = 37, 5
    stride = w * ps + 8  # padded rows
    buf = rng.integers(0, 256, size=h * stride, dtype=np.uint8)
    orig = buf.copy()
    st = synth.HSV_SETTINGS["mixed"]
    oracle.hsvfilter(buf, w, stride, ps, first, bool(bgr), st)
    # reference result via the RGBx arm
    exp = orig.copy()
    for row in range(h):
        for x in range(w):
            o = row * stride + x * ps + first
            tri = orig[o:o + 3][::-1] if bgr else orig[o:o + 3]
            px = np.array([tri[0], tri[1], tri[2], 0], np.uint8)
            oracle.hsvfilter(px, 1, 4, 4, 0, False, st)
            res = px[:3][::-1] if bgr else px[:3]
            exp[o:o + 3] = res
    assert (buf == exp).all()


def test_trailing_partial_row_is_skipped(oracle, synth):
    """chunks_exact_mut(stride) drops a trailing partial row (hsvfilter/imp.rs:96)."""
    w, stride = 4, 16
    buf = np.full(stride * 2 + 7, 200, np.uint8)
    buf[0::4] = 10
    orig = buf.copy()
    oracle.hsvfilter(buf, w, stride, 4, 0, False, synth.HSV_SETTINGS["hue90"])
    assert (buf[: 2 * stride] != orig[: 2 * stride]).any()
    assert (buf[2 * stride:] == orig[2 * stride:]).all()


def test_hsvdetect_defaults(oracle):
    """hsvdetector/imp.rs:126-156: near-black pixels match the default reference (0,0,0) window."""
    src = np.array([0, 0, 0, 9, 20, 20, 20, 9, 255, 0, 0, 9, 200, 200, 200, 9], np.uint8)
    dst = np.zeros(16, np.uint8)
    oracle.hsvdetect(src, 16, 4, 0, False, dst, 16, False, False, 4, (0.0, 10.0, 0.0, 0.15, 0.0, 0.3))
    px = dst.reshape(4, 4)
    assert list(px[:, 3]) == [255, 255, 0, 0]
    assert (px[:, :3] == src.reshape(4, 4)[:, :3]).all()
