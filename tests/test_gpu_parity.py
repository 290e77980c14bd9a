"""GPU parity tests: the HIP path, called through the C ABI (libmi355fx.so), against the CPU oracle
on the same seeded inputs. Bit-exact for every u8/u16 pixel result; bit-exact f32/f64 for echo.

All tests here need a real MI355X: `pytest -m gpu`.
"""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W4K, H4K = 3840, 2160


def _mismatch_report(got, exp, limit=5):
    bad = np.nonzero(got.reshape(-1) != exp.reshape(-1))[0]
    return "mismatching bytes: %d, first at %s got %s exp %s" % (
        bad.size, bad[:limit], got.reshape(-1)[bad[:limit]], exp.reshape(-1)[bad[:limit]])


# ------------------------------------------------------------------ hsvfilter

FOUR_BYTE = ["RGBx", "xRGB", "BGRx", "xBGR", "RGBA", "ARGB", "BGRA", "ABGR"]


@pytest.mark.parametrize("setting", ["defaults", "hue90", "mixed"])
def test_hsvfilter_allcolors_rgba_fast(ctx, oracle, synth, setting):
    """Every one of the 2^24 colours, FAST kernel, host entry point (H2D, kernel, D2H)."""
    st = synth.HSV_SETTINGS[setting]
    ac = synth.allcolors()
    exp = ac.copy().reshape(-1)
    oracle.hsvfilter(exp, 4096, 4096 * 4, 4, 0, False, st, nthreads=8)
    got = ac.copy().reshape(-1)
    ctx.hsvfilter_frame_ip(got, 4096, 4096 * 4, "RGBA", st)
    assert (got == exp).all(), _mismatch_report(got, exp)


@pytest.mark.parametrize("st", [
    (-360.0, 1.0, 0.0, 1.0, 0.0), (360.0, 1.0, 0.0, 1.0, 0.0), (359.99997, 0.5, 0.25, 2.0, -0.5),
    (1e-30, 1.0, 0.0, 1.0, 0.0), (-180.0, -1.0, 0.5, 0.0, 0.5), (120.5, 1e9, -1e9, 1.0, 0.0),
    (45.0, float("nan"), 0.0, float("inf"), float("-inf")), (0.0, float("inf"), float("-inf"), 1.0, 0.0),
])
def test_hsvfilter_allcolors_fast_edge_settings(ctx, oracle, synth, st):
    """FAST-path boundary settings (|hue_shift| = 360, tiny shift, NaN/inf sat/val parameters)."""
    ac = synth.allcolors()
    exp = ac.copy().reshape(-1)
    oracle.hsvfilter(exp, 4096, 4096 * 4, 4, 0, False, st, nthreads=8)
    got = ac.copy().reshape(-1)
    ctx.hsvfilter_frame_ip(got, 4096, 4096 * 4, "RGBA", st)
    assert (got == exp).all(), _mismatch_report(got, exp)


@pytest.mark.parametrize("st", [
    (725.5, 1.0, 0.0, 1.0, 0.0), (-1e6, 1.1, 0.0, 0.9, 0.0), (float("inf"), 1.0, 0.0, 1.0, 0.0),
    (float("nan"), 1.0, 0.0, 1.0, 0.0), (1e-40, 1.0, 0.0, 1.0, 0.0), (3.4e38, 1.0, 0.0, 1.0, 0.0),
])
def test_hsvfilter_allcolors_generic_settings(ctx, oracle, synth, st):
    """Settings outside the FAST envelope take the GENERIC (literal) kernel: huge / non-finite /
    denormal hue-shift (SURVEY.md Appendix A1 last bullet)."""
    ac = synth.allcolors()
    exp = ac.copy().reshape(-1)
    oracle.hsvfilter(exp, 4096, 4096 * 4, 4, 0, False, st, nthreads=8)
    got = ac.copy().reshape(-1)
    ctx.hsvfilter_frame_ip(got, 4096, 4096 * 4, "RGBA", st)
    assert (got == exp).all(), _mismatch_report(got, exp)


WIDE_SHIFTS = [(720.0, 1.0, 0.0, 1.0, 0.0), (-400.5, 1.1, 0.0, 0.9, 0.01), (360.00003, 1.0, 0.0, 1.0, 0.0),
               (-360.00003, 1.0, 0.0, 1.0, 0.0), (4194304.0, 1.0, 0.0, 1.0, 0.0), (-123456.7, 0.7, 0.2, 1.0, 0.0),
               (-720.0, 1.0, 0.0, 1.0, 0.0), (1080.0, 1.0, 0.0, 1.0, 0.0), (4194304.5, 1.0, 0.0, 1.0, 0.0),
               (-4194304.0, 1.0, 0.0, 1.0, 0.0)]


@pytest.mark.parametrize("st", WIDE_SHIFTS)
def test_hsvfilter_allcolors_wide_hue_shift(ctx, oracle, synth, st):
    """360 < |hue_shift| <= 2^22 runs the FAST arithmetic with an exact one-fma fmod (round 3; these settings used to
    take the literal kernel); the reference's `% 360` then `< 0 -> + 360` on every colour, both signs, exact multiples
    of 360 (fmod gives -0 for a negative one) and the class boundaries (2^22 + 0.5 is GENERIC again)."""
    ac = synth.allcolors()
    exp = ac.copy().reshape(-1)
    oracle.hsvfilter(exp, 4096, 4096 * 4, 4, 0, False, st, nthreads=8)
    got = ac.copy().reshape(-1)
    ctx.hsvfilter_frame_ip(got, 4096, 4096 * 4, "RGBA", st)
    assert (got == exp).all(), _mismatch_report(got, exp)


@pytest.mark.parametrize("fmt,w,h,stride,n", [("RGBA", 1918, 9, 7680, 1), ("BGRx", 1921, 5, 7696, 3), ("xRGB", 637, 33, 2560, 2),
                                              ("ABGR", 4, 4, 64, 2), ("ARGB", 3, 2, 16, 1), ("RGBx", 3840, 17, 15424, 2)])
@pytest.mark.parametrize("setting", ["hue90", "generic"])
@pytest.mark.parametrize("gap", [48, 0])
def test_hsvfilter_padded_rows_16_byte_aligned(ctx, oracle, synth, fmt, w, h, stride, n, setting, gap):
    """4-byte formats whose stride / frame pitch are padded to multiples of 16 (what aligned allocators negotiate) run
    the strided 16-B-per-lane kernel (round 3), not the one-pixel-per-lane rows kernel: widths that end inside a
    4-pixel group, padding and inter-frame gaps untouched. gap 0 (frames back to back, rows padded: one tall picture) and single
    frames take hsvfilter_rowpad_kernel (round 5: no divisions), frames with a gap between them hsvfilter_strided_kernel."""
    from mi355fx import FMT_LAYOUT
    ps, first, bgr = FMT_LAYOUT[fmt]
    st = synth.HSV_SETTINGS["hue90"] if setting == "hue90" else (float("inf"), 1.0, 0.0, 1.0, 0.0)
    pitch = stride * h + gap
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, size=pitch * n, dtype=np.uint8)
    exp = frames.copy()
    for f in range(n):
        oracle.hsvfilter(exp[f * pitch: f * pitch + stride * h], w, stride, ps, first, bool(bgr), st)
    d = ctx.alloc(frames.nbytes)
    try:
        ctx.h2d(d, frames)
        ctx.hsvfilter_frames_device(d, n, pitch, w, h, stride, fmt, st)
        ctx.synchronize()
        got = np.empty_like(frames)
        ctx.d2h(got, d)
    finally:
        ctx.free(d)
    assert (got == exp).all(), _mismatch_report(got, exp)


def test_hsvfilter_generic_kernel_equals_fast_kernel(ctx, oracle, synth):
    """A/B of the two kernels on the same input (FORCE_GENERIC flag)."""
    import mi355fx
    st = synth.HSV_SETTINGS["mixed"]
    ac = synth.allcolors()
    fast = ac.copy().reshape(-1)
    ctx.hsvfilter_frame_ip(fast, 4096, 4096 * 4, "RGBA", st)
    ctx.set_flag(mi355fx.FLAG_FORCE_GENERIC, 1)
    gen = ac.copy().reshape(-1)
    ctx.hsvfilter_frame_ip(gen, 4096, 4096 * 4, "RGBA", st)
    ctx.set_flag(mi355fx.FLAG_FORCE_GENERIC, 0)
    assert (fast == gen).all(), _mismatch_report(fast, gen)


@pytest.mark.parametrize("fmt", FOUR_BYTE + ["RGB", "BGR"])
def test_hsvfilter_all_formats(ctx, oracle, synth, fmt):
    """The 10 caps formats of hsvfilter (hsvfilter/imp.rs:274-312), contiguous 4-byte frames hit the
    flat kernel, 3-byte frames the row kernel. 1920x1080 noise (config 2 shape for BGRx)."""
    from mi355fx import FMT_LAYOUT
    ps, first, bgr = FMT_LAYOUT[fmt]
    w, h = 1920, 1080
    stride = (w * ps + 3) & ~3
    frame = synth.noise_frame(stride, h, channels=1).reshape(-1).copy()
    st = synth.HSV_SETTINGS["hue90"]
    exp = frame.copy()
    oracle.hsvfilter(exp, w, stride, ps, first, bool(bgr), st, nthreads=8)
    got = frame.copy()
    ctx.hsvfilter_frame_ip(got, w, stride, fmt, st)
    assert (got == exp).all(), _mismatch_report(got, exp)


@pytest.mark.parametrize("fmt,w,h,pad", [("RGBA", 37, 5, 12), ("xBGR", 1, 1, 0), ("BGRx", 3, 7, 4), ("RGB", 5, 3, 1),
                                         ("BGR", 33, 2, 3), ("ARGB", 640, 3, 64)])
def test_hsvfilter_ragged_and_padded(ctx, oracle, synth, fmt, w, h, pad):
    """Row padding must come back untouched; odd sizes exercise the row kernel; a trailing partial row
    in the plane is skipped (chunks_exact_mut)."""
    from mi355fx import FMT_LAYOUT
    ps, first, bgr = FMT_LAYOUT[fmt]
    stride = w * ps + pad
    tail = (ps * 2) if stride > ps * 2 else 0  # partial trailing row, multiple of the pixel stride
    rng = np.random.default_rng(11)
    frame = rng.integers(0, 256, size=stride * h + tail, dtype=np.uint8)
    st = synth.HSV_SETTINGS["mixed"]
    exp = frame.copy()
    oracle.hsvfilter(exp, w, stride, ps, first, bool(bgr), st)
    got = frame.copy()
    ctx.hsvfilter_frame_ip(got, w, stride, fmt, st)
    assert (got == exp).all(), _mismatch_report(got, exp)


@pytest.mark.parametrize("fmt", ["RGB", "BGR"])
@pytest.mark.parametrize("w,h", [(1000, 7), (4096, 3), (1024, 1), (340, 3), (1028, 5)])
@pytest.mark.parametrize("setting", ["mixed", "wide", "generic"])
def test_hsvfilter_rgb24_chunks_and_tails(ctx, oracle, synth, fmt, w, h, setting):
    """3-byte formats: whole 3 KB chunks (1024 pixels) go through the coalescing kernel (three 16-byte loads per lane, LDS
    transpose), what is left through the 12-bytes-per-lane one; sizes with no whole chunk, only whole chunks, and both."""
    from mi355fx import FMT_LAYOUT
    ps, first, bgr = FMT_LAYOUT[fmt]
    st = {"mixed": synth.HSV_SETTINGS["mixed"], "wide": (777.25, 1.0, 0.0, 1.0, 0.0), "generic": (float("nan"), 1.0, 0.0, 1.0, 0.0)}[setting]
    frame = synth.noise_frame(w * 3, h, seed=w + h, channels=1).reshape(-1).copy()
    exp = frame.copy()
    oracle.hsvfilter(exp, w, w * 3, ps, first, bool(bgr), st)
    got = frame.copy()
    ctx.hsvfilter_frame_ip(got, w, w * 3, fmt, st)
    assert (got == exp).all(), _mismatch_report(got, exp)


def test_hsvfilter_empty_and_errors(ctx, synth):
    import mi355fx
    st = synth.HSV_SETTINGS["hue90"]
    empty = np.zeros(0, np.uint8)
    ctx.hsvfilter_frame_ip(empty, 0, 16, "RGBA", st)          # no rows: no-op
    short = np.zeros(8, np.uint8)
    ctx.hsvfilter_frame_ip(short, 4, 16, "RGBA", st)          # less than one row: no-op
    assert (short == 0).all()
    with pytest.raises(mi355fx.Mi355Error) as e:
        ctx.hsvfilter_frame_ip(np.zeros(64, np.uint8), 8, 16, "RGBA", st)  # line_bytes > stride
    assert e.value.status == mi355fx.ERR_INVALID_ARG
    with pytest.raises(mi355fx.Mi355Error):
        ctx.hsvfilter_frame_ip(np.zeros(10, np.uint8), 1, 5, "RGBA", st)   # len % pixel_stride != 0


def test_hsvfilter_4k_batch_device_is_pure_per_pixel_function(ctx, oracle, synth):
    """Full BASELINE size (3840x2160 RGBA, batch of 3 frames, device-resident entry point): the output
    must equal T[input] where T is the 2^24-entry table produced by the oracle (size-independent
    property: the element is a pure per-pixel function)."""
    st = synth.HSV_SETTINGS["hue90"]
    ac = synth.allcolors()
    tab = ac.copy().reshape(-1)
    oracle.hsvfilter(tab, 4096, 4096 * 4, 4, 0, False, st, nthreads=8)
    table = tab.view(np.uint32) & np.uint32(0x00FFFFFF)       # index = r | g<<8 | b<<16
    n = 3
    frames = np.stack([synth.noise_frame(W4K, H4K, seed=100 + i) for i in range(n - 1)] + [synth.smooth_frame(W4K, H4K)])
    nbytes = frames.nbytes
    d = ctx.alloc(nbytes)
    try:
        ctx.h2d(d, frames)
        ctx.hsvfilter_frames_device(d, n, W4K * H4K * 4, W4K, H4K, W4K * 4, "RGBA", st)
        ctx.synchronize()
        got = np.empty_like(frames)
        ctx.d2h(got, d)
    finally:
        ctx.free(d)
    inp = frames.reshape(-1).view(np.uint32)
    exp = table[inp & np.uint32(0x00FFFFFF)] | (inp & np.uint32(0xFF000000))
    assert (got.reshape(-1).view(np.uint32) == exp).all()


@pytest.mark.parametrize("fmt", ["RGBA", "BGRx"])
@pytest.mark.parametrize("st", [(90.0, 1.0, 0.0, 1.0, 0.0), (-33.3, 1.2, -0.05, 0.9, 0.02), (725.5, 1.0, 0.0, 1.0, 0.0),
                                (float("inf"), 1.3, -0.1, 0.9, 0.05)])
def test_hsvfilter_table_kernel_allcolors(ctx, oracle, synth, fmt, st):
    """MI355_FLAG_HSV_TABLE = 2: hsvfilter through the memoised table (built by the arithmetic kernel for the format's byte
    order) on every colour with a varying 4th byte, FAST and GENERIC settings."""
    import mi355fx
    from mi355fx import FMT_LAYOUT
    ps, first, bgr = FMT_LAYOUT[fmt]
    ac = synth.allcolors().copy().reshape(-1)
    ac[3::4] = (np.arange(4096 * 4096, dtype=np.uint32) * 2654435761 >> 11).astype(np.uint8)
    exp = ac.copy()
    oracle.hsvfilter(exp, 4096, 4096 * 4, ps, first, bool(bgr), st, nthreads=8)
    ctx.set_flag(mi355fx.FLAG_HSV_TABLE, 2)
    got = ac.copy()
    ctx.hsvfilter_frame_ip(got, 4096, 4096 * 4, fmt, st)
    assert (got == exp).all(), _mismatch_report(got, exp)
    assert ctx.colorlut_kernel_choice(fused=2)[0]


@pytest.mark.parametrize("fmt,w,h,pad", [("xRGB", 256, 64, 0), ("RGB", 256, 64, 0), ("RGBA", 250, 7, 24), ("ABGR", 128, 8, 0)])
def test_hsvfilter_table_mode_ineligible_geometry_falls_back(ctx, oracle, synth, fmt, w, h, pad):
    """Alpha-first / 3-byte formats and padded rows never use the table; the flag must not change the result."""
    import mi355fx
    from mi355fx import FMT_LAYOUT
    ps, first, bgr = FMT_LAYOUT[fmt]
    stride = w * ps + pad
    rng = np.random.default_rng(3)
    frame = rng.integers(0, 256, size=stride * h, dtype=np.uint8)
    st = synth.HSV_SETTINGS["mixed"]
    exp = frame.copy()
    oracle.hsvfilter(exp, w, stride, ps, first, bool(bgr), st)
    ctx.set_flag(mi355fx.FLAG_HSV_TABLE, 2)
    got = frame.copy()
    ctx.hsvfilter_frame_ip(got, w, stride, fmt, st)
    assert (got == exp).all(), _mismatch_report(got, exp)


def test_hsvfilter_auto_table_per_buffer_launches(ctx, oracle, synth):
    """MI355_FLAG_HSV_TABLE = 1 (auto), one 4K frame per launch: compute kernel while the settings are young, then both kinds are measured and
    one is chosen; the output is exact at every call and a settings / byte-order change is honoured at once."""
    import mi355fx
    ctx.set_flag(mi355fx.FLAG_HSV_TABLE, 1)
    w, h = W4K, H4K
    frame = synth.smooth_frame(w, h, seed=91).reshape(-1)
    sts = [synth.HSV_SETTINGS["hue90"], synth.HSV_SETTINGS["mixed"]]
    exps = []
    for st in sts:
        e = frame.copy()
        oracle.hsvfilter(e, w, w * 4, 4, 0, False, st, nthreads=8)
        exps.append(e)
    d = ctx.alloc(frame.nbytes)
    got = np.zeros_like(frame)
    try:
        def run(st, fmt="RGBA"):
            ctx.h2d(d, frame)
            ctx.hsvfilter_frames_device(d, 1, w * h * 4, w, h, w * 4, fmt, st)
            ctx.synchronize()
            ctx.d2h(got, d)
        for k in range(20):
            run(sts[0])
            assert (got == exps[0]).all(), "call %d" % k
        on_table, t_c, t_t = ctx.colorlut_kernel_choice(fused=2)
        assert t_c > 0.0 and t_t > 0.0
        assert on_table == (t_t < t_c) or abs(t_t - t_c) <= 0.03 * t_c  # 3 % hysteresis
        run(sts[1])
        assert (got == exps[1]).all()
        e = frame.copy()
        oracle.hsvfilter(e, w, w * 4, 4, 0, True, sts[1], nthreads=8)
        run(sts[1], "BGRA")
        assert (got == e).all()
    finally:
        ctx.free(d)


def test_hsvfilter_generic_settings_get_the_table_by_default(ctx, oracle, synth):
    """Default flags: settings outside the FAST envelope (non-finite hue shift: literal GENERIC arithmetic) are candidates
    for the memoised table once they have been stable, and are served by whichever kind measured faster; FAST settings -
    since round 3 that includes 360 < |hue-shift| <= 2^22 - never are."""
    w, h, n = 1920, 1080, 8
    frames = np.stack([synth.smooth_frame(w, h, seed=60 + i) for i in range(n)]).reshape(-1)
    d = ctx.alloc(frames.nbytes)
    got = np.zeros_like(frames)
    try:
        for st, candidate in (((float("inf"), 1.1, 0.0, 0.9, 0.0), True), ((725.5, 1.1, 0.0, 0.9, 0.0), False), (synth.HSV_SETTINGS["hue90"], False)):
            exp = frames.copy()
            for f in range(n):
                oracle.hsvfilter(exp[f * w * h * 4:(f + 1) * w * h * 4], w, w * 4, 4, 0, True, st, nthreads=8)
            for k in range(16):
                ctx.h2d(d, frames)
                ctx.hsvfilter_frames_device(d, n, w * h * 4, w, h, w * 4, "BGRx", st)
                ctx.synchronize()
                ctx.d2h(got, d)
                assert (got == exp).all(), "call %d" % k
            on_table, t_c, t_t = ctx.colorlut_kernel_choice(fused=2)
            if candidate:
                assert t_c > 0.0 and t_t > 0.0 and on_table == (t_t < t_c), (on_table, t_c, t_t)
            else:
                assert not on_table
    finally:
        ctx.free(d)


# ------------------------------------------------------------------ colorlut

def _load_cube(ctx, oracle, text):
    cube = oracle.Cube.parse(text)
    sc, of = cube.domain
    ctx.colorlut_load(cube.is3d, cube.size, cube.table, sc, of)
    return cube


@pytest.mark.parametrize("force_generic", [0, 1])
def test_colorlut_allcolors_33(ctx, oracle, synth, force_generic):
    """BASELINE config 3 LUT (33^3 trilinear) on every 8-bit colour: LDS kernel and generic kernel."""
    import mi355fx
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(33))
    ac = synth.allcolors()
    exp = np.zeros_like(ac)
    oracle.colorlut_rgba8(cube, ac, 4096 * 4, exp, 4096 * 4, 4096, 4096, nthreads=8)
    ctx.set_flag(mi355fx.FLAG_FORCE_GENERIC, force_generic)
    got = np.zeros_like(ac)
    ctx.colorlut_frame(ac, 4096 * 4, got, 4096 * 4, 4096, 4096, "RGBA")
    assert (got == exp).all(), _mismatch_report(got, exp)
    assert zlib.crc32(got.tobytes()) == zlib.crc32(exp.tobytes())


@pytest.mark.parametrize("size", [2, 3, 17, 32, 33, 34, 35, 65])
def test_colorlut_identity_is_passthrough(ctx, oracle, synth, size):
    """SURVEY.md §8 a6 KAT: an identity LUT of any size is an exact pass-through (sizes <= 34 use the
    LDS kernel, larger ones the gather kernel)."""
    _load_cube(ctx, oracle, synth.cube_text_3d(size, identity=True))
    ac = synth.allcolors()
    got = np.zeros_like(ac)
    ctx.colorlut_frame(ac, 4096 * 4, got, 4096 * 4, 4096, 4096, "RGBA")
    assert (got == ac).all(), _mismatch_report(got, ac)


@pytest.mark.parametrize("size,domain", [(17, None), (33, ((-0.25, 0.0, 0.1), (1.5, 1.0, 0.9))), (34, None),
                                         (5, ((0.0, 0.0, 0.0), (2.0, 0.5, 1.0))), (64, None)])
def test_colorlut_sizes_and_domains(ctx, oracle, synth, size, domain):
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(size, amp=0.07, domain=domain))
    ac = synth.allcolors()
    exp = np.zeros_like(ac)
    oracle.colorlut_rgba8(cube, ac, 4096 * 4, exp, 4096 * 4, 4096, 4096, nthreads=8)
    got = np.zeros_like(ac)
    ctx.colorlut_frame(ac, 4096 * 4, got, 4096 * 4, 4096, 4096, "RGBA")
    assert (got == exp).all(), _mismatch_report(got, exp)


def test_colorlut_out_of_range_and_nonfinite_lut(ctx, oracle):
    """LUT entries outside [0,1], huge, inf and nan (str::parse::<f32> accepts them): the load-time
    check routes these to the literal kernel; results must still match the oracle."""
    rng = np.random.default_rng(5)
    size = 9
    vals = rng.uniform(-0.5, 1.5, size=(size ** 3, 3))
    lines = ["LUT_3D_SIZE %d" % size]
    for i, v in enumerate(vals):
        if i % 97 == 3:
            lines.append("inf %.6f -inf" % v[1])
        elif i % 89 == 5:
            lines.append("%.6f nan %.6f" % (v[0], v[2]))
        elif i % 83 == 7:
            lines.append("3e38 -3e38 %.6f" % v[2])
        else:
            lines.append("%.6f %.6f %.6f" % tuple(v))
    cube = _load_cube(ctx, oracle, "\n".join(lines) + "\n")
    rgb = rng.integers(0, 256, size=(512, 512, 4), dtype=np.uint8).reshape(512, 2048)
    exp = np.zeros_like(rgb)
    oracle.colorlut_rgba8(cube, rgb, 2048, exp, 2048, 512, 512)
    got = np.zeros_like(rgb)
    ctx.colorlut_frame(rgb, 2048, got, 2048, 512, 512, "RGBA")
    assert (got == exp).all(), _mismatch_report(got, exp)


def test_colorlut_1d(ctx, oracle, synth):
    cube = _load_cube(ctx, oracle, synth.cube_text_1d(64))
    assert not cube.is3d
    ac = synth.allcolors()[:512]
    exp = np.zeros_like(ac)
    oracle.colorlut_rgba8(cube, ac, 4096 * 4, exp, 4096 * 4, 4096, 512)
    got = np.zeros_like(ac)
    ctx.colorlut_frame(ac, 4096 * 4, got, 4096 * 4, 4096, 512, "RGBA")
    assert (got == exp).all(), _mismatch_report(got, exp)


@pytest.mark.parametrize("le", [True, False])
@pytest.mark.parametrize("kind", ["3d", "1d"])
def test_colorlut_rgba64(ctx, oracle, synth, le, kind):
    """RGBA64_LE / RGBA64_BE arms (colorlut/imp.rs:296-397): byte order handling + raw alpha copy."""
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(17) if kind == "3d" else synth.cube_text_1d(1024))
    rng = np.random.default_rng(9)
    w, h = 333, 41
    stride = w * 8 + 16
    src = rng.integers(0, 256, size=h * stride, dtype=np.uint8)
    exp = np.full(h * stride, 0xCD, np.uint8)
    got = exp.copy()
    oracle.colorlut_rgba64(cube, src, stride, exp, stride, w, h, le=le)
    ctx.colorlut_frame(src, stride, got, stride, w, h, "RGBA64_LE" if le else "RGBA64_BE")
    assert (got == exp).all(), _mismatch_report(got, exp)


def test_colorlut_independent_strides_and_untouched_padding(ctx, oracle, synth):
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(33))
    rng = np.random.default_rng(13)
    w, h = 250, 9
    ss, ds = w * 4 + 24, w * 4 + 8
    src = rng.integers(0, 256, size=h * ss, dtype=np.uint8)
    exp = np.full(h * ds, 0xEE, np.uint8)
    got = exp.copy()
    oracle.colorlut_rgba8(cube, src, ss, exp, ds, w, h)
    ctx.colorlut_frame(src, ss, got, ds, w, h, "RGBA")
    assert (got == exp).all(), _mismatch_report(got, exp)


@pytest.mark.parametrize("variant", [0, 5, 6, 7])
@pytest.mark.parametrize("w,h,spad,dpad,n", [(640, 37, 64, 128, 1), (1920, 16, 16, 16, 2), (132, 9, 112, 0, 3)])
def test_colorlut_padded_rows_take_the_fast_kernels(ctx, oracle, synth, variant, w, h, spad, dpad, n):
    """RGBA frames whose strides are padded to multiples of 16 B (independent for source and destination,
    colorlut/imp.rs:275-286) run the brick / memoised-table kernels with a row-stride argument (round 3) instead of the rows
    kernels; a batch is frames `stride * height` apart. Padding and the destination's padding stay untouched; fused chain too."""
    import mi355fx
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(33))
    ss, ds = w * 4 + spad, w * 4 + dpad
    rng = np.random.default_rng(w + variant)
    base = synth.smooth_frame(w, h, seed=9).reshape(h, w * 4)
    src = rng.integers(0, 256, size=(n, h, ss), dtype=np.uint8)
    for f in range(n):
        src[f, :, :w * 4] = np.roll(base, 4 * f, axis=1)
    src = src.reshape(-1)
    st = synth.HSV_SETTINGS["hue90"]
    for fused in (False, True):
        exp = np.full(n * h * ds, 0xEE, np.uint8)
        for f in range(n):
            s_f = src[f * h * ss:(f + 1) * h * ss].copy()
            if fused:
                oracle.hsvfilter(s_f, w, ss, 4, 0, False, st)
            oracle.colorlut_rgba8(cube, s_f, ss, exp[f * h * ds:(f + 1) * h * ds], ds, w, h)
        d_src, d_dst = ctx.alloc(src.nbytes), ctx.alloc(exp.nbytes)
        ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, variant)
        try:
            ctx.h2d(d_src, src)
            got = np.full(n * h * ds, 0xEE, np.uint8)
            for rep in range(12 if variant == 0 else 1):   # auto: let the choice settle and change kernels on the way
                ctx.h2d(d_dst, np.full(n * h * ds, 0xEE, np.uint8))
                if fused:
                    ctx.hsv_colorlut_frames_device(d_src, h * ss, ss, d_dst, h * ds, ds, n, w, h, st)
                else:
                    ctx.colorlut_frames_device(d_src, h * ss, ss, d_dst, h * ds, ds, n, w, h, "RGBA")
                ctx.synchronize()
                ctx.d2h(got, d_dst)
                assert (got == exp).all(), (fused, rep, _mismatch_report(got, exp))
            if not fused and variant in (5, 7):
                assert ctx.colorlut_kernel_name().startswith("colorlut_table_tiled_kernel" if variant == 5 else "colorlut3d_brick_kernel")
        finally:
            ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 0)
            ctx.free(d_src)
            ctx.free(d_dst)


def test_colorlut_without_lut_fails_like_reference(ctx):
    """transform_frame without a LUT is FlowError::Error (colorlut/imp.rs:209-212)."""
    import mi355fx
    a = np.zeros(16, np.uint8)
    with pytest.raises(mi355fx.Mi355Error) as e:
        ctx.colorlut_frame(a, 16, a.copy(), 16, 4, 1, "RGBA")
    assert e.value.status == mi355fx.ERR_NOT_CONFIGURED
    assert "No LUT configured" in str(e.value)


def test_colorlut_4k_batch_device_matches_table(ctx, oracle, synth):
    """Full BASELINE size, device entry point, batch of 2 frames (noise + smooth): output == T[input]
    with T from the oracle on all colours."""
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(33))
    ac = synth.allcolors()
    tab = np.zeros_like(ac)
    oracle.colorlut_rgba8(cube, ac, 4096 * 4, tab, 4096 * 4, 4096, 4096, nthreads=8)
    table = tab.reshape(-1).view(np.uint32) & np.uint32(0x00FFFFFF)
    frames = np.stack([synth.noise_frame(W4K, H4K, seed=77), synth.smooth_frame(W4K, H4K)])
    d_src, d_dst = ctx.alloc(frames.nbytes), ctx.alloc(frames.nbytes)
    try:
        ctx.h2d(d_src, frames)
        pitch = W4K * H4K * 4
        ctx.colorlut_frames_device(d_src, pitch, W4K * 4, d_dst, pitch, W4K * 4, 2, W4K, H4K, "RGBA")
        ctx.synchronize()
        got = np.empty_like(frames)
        ctx.d2h(got, d_dst)
    finally:
        ctx.free(d_src)
        ctx.free(d_dst)
    inp = frames.reshape(-1).view(np.uint32)
    exp = table[inp & np.uint32(0x00FFFFFF)] | (inp & np.uint32(0xFF000000))
    assert (got.reshape(-1).view(np.uint32) == exp).all()


def test_hsv_then_lut_chain_4k(ctx, oracle, synth):
    """The headline chain (hsvfilter hue-shift=90 -> colorlut 33^3) on one 4K smooth frame vs the
    oracle chain."""
    st = synth.HSV_SETTINGS["hue90"]
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(33))
    frame = synth.smooth_frame(W4K, H4K)
    mid = frame.copy().reshape(-1)
    oracle.hsvfilter(mid, W4K, W4K * 4, 4, 0, False, st, nthreads=8)
    exp = np.zeros_like(frame)
    oracle.colorlut_rgba8(cube, mid.reshape(H4K, -1), W4K * 4, exp, W4K * 4, W4K, H4K, nthreads=8)
    got_mid = frame.copy().reshape(-1)
    ctx.hsvfilter_frame_ip(got_mid, W4K, W4K * 4, "RGBA", st)
    got = np.zeros_like(frame)
    ctx.colorlut_frame(got_mid, W4K * 4, got, W4K * 4, W4K, H4K, "RGBA")
    assert (got_mid == mid).all()
    assert (got == exp).all(), _mismatch_report(got, exp)


# ------------------------------------------------------------------ rsaudioecho

def _echo_pair(ctx, oracle, max_delay_ns, rate, channels):
    e = oracle.Echo(max_delay_ns, rate, channels)
    ctx.echo_setup(e.ring_len)
    return e


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("delay_ms,intensity,feedback", [(250, 0.6, 0.4), (250, 0.5, 0.0), (1000, 0.9, 0.9),
                                                         (0, 0.3, 0.2), (500000, 0.5, 0.0), (7, 1.0, 0.99)])
def test_echo_config1(ctx, oracle, synth, dtype, delay_ms, intensity, feedback):
    """BASELINE config 1: 48 kHz stereo sine, 10 s in ONE buffer; plus the default-parameter quirk
    (delay 500 s clamped to max-delay 1 s), delay 0 and a short comb."""
    rate, ch = 48000, 2
    e = _echo_pair(ctx, oracle, 10 ** 9, rate, ch)
    x = synth.sine_stereo_f32(480000 if feedback == 0.0 or delay_ms >= 250 or delay_ms == 0 else 48000).astype(dtype)
    delay_ns = delay_ms * 10 ** 6
    exp = x.copy()
    e.process(exp, delay_ns, intensity, feedback)
    got = x.copy()
    d = oracle.lib().oracle_echo_delay_samples(delay_ns, 10 ** 9, rate, ch)
    ctx.echo_process(got, d, intensity, feedback)
    assert got.tobytes() == exp.tobytes()
    ring, pos = ctx.echo_state(e.ring_len)
    assert pos == e.pos
    assert ring[: e.ring_len].tobytes() == e.ring[: e.ring_len].tobytes()


def test_echo_streaming_many_buffers(ctx, oracle, synth):
    """Element state carried across buffers of ragged sizes (ring position + contents)."""
    rate, ch = 44100, 2
    e = _echo_pair(ctx, oracle, 300 * 10 ** 6, rate, ch)
    rng = np.random.default_rng(21)
    delay_ns = 123456789
    d = oracle.lib().oracle_echo_delay_samples(delay_ns, 300 * 10 ** 6, rate, ch)
    for n in [1, 2, 4410, 0, 26460, 13, 88200, 7]:
        x = rng.standard_normal(n).astype(np.float32)
        exp, got = x.copy(), x.copy()
        e.process(exp, delay_ns, 0.7, 0.5)
        ctx.echo_process(got, d, 0.7, 0.5)
        assert got.tobytes() == exp.tobytes()
    ring, pos = ctx.echo_state(e.ring_len)
    assert pos == e.pos and ring[: e.ring_len].tobytes() == e.ring[: e.ring_len].tobytes()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_echo_batch_of_streams(ctx, oracle, synth, dtype):
    """mi355_echo_*_batch: independent AudioEcho instances advanced together, each with its own delay / intensity / feedback
    (feedback 0 and != 0 mixed in one batch, delay 0, a comb shorter than the buffer), ragged buffer sizes across calls,
    padded stream stride. Every stream must equal its own oracle instance bit for bit, ring and position included."""
    rate, ch, S = 48000, 2, 7
    max_ns = 500 * 10 ** 6
    es = [oracle.Echo(max_ns, rate, ch) for _ in range(S)]
    ctx.echo_setup_batch(S, es[0].ring_len)
    delays_ns = [250 * 10 ** 6, 0, 10 ** 6, 123456789, 499999999, 7 * 10 ** 6, 250 * 10 ** 6]
    inten = [0.6, 0.3, 1.0, 0.7, 0.5, 0.9, 0.0]
    fb = [0.4, 0.2, 0.99, 0.0, 0.0, 0.5, 0.0]
    d = [oracle.lib().oracle_echo_delay_samples(t, max_ns, rate, ch) for t in delays_ns]
    rng = np.random.default_rng(5)
    stride = 96000 + 32
    dev = ctx.alloc(S * stride * np.dtype(dtype).itemsize)
    try:
        for n in [960, 2, 96000, 0, 4801, 19200]:
            x = rng.standard_normal((S, stride)).astype(dtype)
            exp = x.copy()
            for s in range(S):
                row = np.ascontiguousarray(exp[s, :n])
                es[s].process(row, delays_ns[s], inten[s], fb[s])
                exp[s, :n] = row
            ctx.h2d(dev, x.reshape(-1).view(np.uint8))
            ctx.echo_process_batch_device(dev, stride, n, dtype == np.float64, d, inten, fb)
            got = np.zeros_like(x)
            ctx.synchronize()
            ctx.d2h(got.reshape(-1).view(np.uint8), dev)
            assert got.tobytes() == exp.tobytes(), "buffer of %d samples" % n   # padding beyond n untouched as well
        for s in range(S):
            ring, pos = ctx.echo_state(es[s].ring_len, stream=s)
            assert pos == es[s].pos and ring[: es[s].ring_len].tobytes() == es[s].ring[: es[s].ring_len].tobytes(), s
        with pytest.raises(Exception):   # the single-stream entry point on a batch
            ctx.echo_process(np.zeros(4, np.float32), 1, 0.5, 0.0)
        with pytest.raises(Exception):   # delay beyond the ring (RingBufferIter::new's assertion)
            ctx.echo_process_batch_device(dev, stride, 16, dtype == np.float64, [es[0].ring_len + 1] * S, inten, fb)
    finally:
        ctx.free(dev)
        ctx.echo_reset()


def test_echo_not_negotiated(ctx):
    """transform_ip before setup is FlowError::NotNegotiated (audioecho/imp.rs:210)."""
    import mi355fx
    with pytest.raises(mi355fx.Mi355Error) as e:
        ctx.echo_process(np.zeros(4, np.float32), 1, 0.5, 0.0)
    assert e.value.status == mi355fx.ERR_NOT_CONFIGURED


# ------------------------------------------------------------------ hsvdetector

@pytest.mark.parametrize("st", [
    (0.0, 10.0, 0.0, 0.15, 0.0, 0.3),            # defaults (hsvdetector/imp.rs:25-30)
    (120.0, 40.0, 0.8, 0.5, 0.7, 0.6),
    (-180.0, 180.0, 0.5, 0.5, 0.5, 0.5), (180.0, 1.0, 1.0, 0.25, 1.0, 0.25), (-0.0, 0.0, 0.0, 1.0, 0.0, 1.0),
    (359.0, 15.0, 0.5, 0.4, 0.5, 0.4), (240.0, 25.0, 0.6, 0.4, 0.5, 0.5), (180.00002, 90.0, 0.5, 0.5, 0.5, 0.5),   # (180, 540]: FAST, negative offset
    (360.0, 0.0, 0.0, 1.0, 0.5, 0.5), (540.0, 180.0, 0.5, 0.5, 0.5, 0.5),
    (540.00006, 30.0, 0.5, 0.5, 0.5, 0.5), (-725.0, 30.0, 0.5, 0.5, 0.5, 0.5),   # outside [-180, 540]: literal kernel
])
@pytest.mark.parametrize("in_fmt,out_fmt", [("RGBx", "RGBA"), ("xBGR", "ARGB"), ("BGRx", "ABGR"), ("xRGB", "BGRA")])
def test_hsvdetect_allcolors(ctx, oracle, synth, st, in_fmt, out_fmt):
    """hsvdetector on every colour (vectorised FAST kernel for hue-ref in [-180, 540] in its two offset classes, literal otherwise)."""
    from mi355fx import FMT_LAYOUT
    ps, first, bgr = FMT_LAYOUT[in_fmt]
    af, obgr = {"RGBA": (0, 0), "ARGB": (1, 0), "BGRA": (0, 1), "ABGR": (1, 1)}[out_fmt]
    ac = synth.allcolors()
    if first == 1:   # put the colour triple at bytes 1..3
        ac = np.roll(ac.reshape(-1, 4), 1, axis=1).reshape(4096, -1).copy()
    if bgr:
        v = ac.reshape(-1, 4)
        v[:, first:first + 3] = v[:, first:first + 3][:, ::-1].copy()
    exp = np.zeros_like(ac)
    oracle.hsvdetect(ac, 4096 * 4, ps, first, bool(bgr), exp, 4096 * 4, bool(af), bool(obgr), 4096, st)
    got = np.zeros_like(ac)
    ctx.hsvdetect_frame(ac, 4096 * 4, in_fmt, got, 4096 * 4, out_fmt, 4096, st)
    assert (got == exp).all(), _mismatch_report(got, exp)


@pytest.mark.parametrize("size,force_generic", [(2, 0), (64, 0), (64, 1), (1024, 0), (12000, 0), (65536, 0)])
def test_colorlut_1d_allcolors(ctx, oracle, synth, size, force_generic):
    """1D LUTs on every colour: LDS kernel for sizes whose three tables fit LDS, literal kernel otherwise."""
    import mi355fx
    cube = _load_cube(ctx, oracle, synth.cube_text_1d(size, gamma=0.45))
    ac = synth.allcolors()
    exp = np.zeros_like(ac)
    oracle.colorlut_rgba8(cube, ac, 4096 * 4, exp, 4096 * 4, 4096, 4096, nthreads=8)
    ctx.set_flag(mi355fx.FLAG_FORCE_GENERIC, force_generic)
    got = np.zeros_like(ac)
    ctx.colorlut_frame(ac, 4096 * 4, got, 4096 * 4, 4096, 4096, "RGBA")
    ctx.set_flag(mi355fx.FLAG_FORCE_GENERIC, 0)
    assert (got == exp).all(), _mismatch_report(got, exp)


@pytest.mark.parametrize("fmt", ["RGB", "BGR"])
@pytest.mark.parametrize("setting", ["hue90", "mixed", "generic"])
def test_hsvfilter_rgb24_allcolors(ctx, oracle, synth, fmt, setting):
    """3-byte formats on contiguous storage take the 12-byte (4 pixel) vector kernel: every colour, FAST and
    GENERIC arithmetic."""
    st = synth.HSV_SETTINGS.get(setting) or (1234.5, 0.7, 0.1, 1.3, -0.1)
    ac = synth.allcolors().reshape(-1, 4)[:, :3]
    if fmt == "BGR":
        ac = ac[:, ::-1]
    frame = np.ascontiguousarray(ac).reshape(-1)          # 4096x4096 RGB, stride 12288
    exp = frame.copy()
    oracle.hsvfilter(exp, 4096, 4096 * 3, 3, 0, fmt == "BGR", st, nthreads=8)
    got = frame.copy()
    ctx.hsvfilter_frame_ip(got, 4096, 4096 * 3, fmt, st)
    assert (got == exp).all(), _mismatch_report(got, exp)


def test_independent_contexts_are_concurrent_safe(oracle, synth, mi355lib):
    """Many element instances on one GPU, each driven by its own streaming thread (SURVEY.md section 8b
    'Threading'): four contexts in four Python threads, different settings/LUTs, results must match the
    oracle for every stream."""
    import threading
    import mi355fx
    from mi355fx.cube import parse_cube
    w, h = 1280, 720
    frames = [synth.noise_frame(w, h, seed=500 + i) for i in range(4)]
    settings = [(90.0, 1.0, 0.0, 1.0, 0.0), (-45.0, 1.2, -0.05, 0.9, 0.02), (0.0, 0.5, 0.25, 1.0, 0.0), (200.0, 1.0, 0.0, 1.1, -0.1)]
    texts = [synth.cube_text_3d(33), synth.cube_text_3d(17, amp=0.07), synth.cube_text_3d(33, identity=True), synth.cube_text_1d(256)]
    results, errors = [None] * 4, []

    def worker(i):
        try:
            with mi355fx.Context(0) as c:
                lut = parse_cube(texts[i])
                c.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
                out = None
                for _ in range(5):
                    mid = frames[i].copy().reshape(-1)
                    c.hsvfilter_frame_ip(mid, w, w * 4, "RGBA", settings[i])
                    out = np.zeros_like(frames[i])
                    c.colorlut_frame(mid, w * 4, out, w * 4, w, h, "RGBA")
                results[i] = out
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(4):
        mid = frames[i].copy().reshape(-1)
        oracle.hsvfilter(mid, w, w * 4, 4, 0, False, settings[i])
        exp = np.zeros_like(frames[i])
        oracle.colorlut_rgba8(oracle.Cube.parse(texts[i]), mid, w * 4, exp, w * 4, w, h)
        assert (results[i] == exp).all(), "stream %d" % i


@pytest.mark.parametrize("size", [33, 17, 5, 2])
@pytest.mark.parametrize("le", [True, False])
@pytest.mark.parametrize("domain", [None, ((-0.25, 0.0, 0.1), (1.5, 1.0, 0.9))])
def test_colorlut_rgba64_lds_kernel(ctx, oracle, synth, le, domain, size):
    """RGBA64 + a 3D LUT whose plane fits LDS, on contiguous frames, takes the LDS three-pass kernel (VALU coordinates,
    exact /65535): random 16-bit pixels plus every channel value 0..65535 on the diagonal."""
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(size, amp=0.06, domain=domain))
    rng = np.random.default_rng(23)
    w, h = 1024, 320
    px = rng.integers(0, 65536, size=(h * w, 4), dtype=np.uint16)
    ramp = np.arange(65536, dtype=np.uint16)
    px[:65536, 0] = ramp; px[:65536, 1] = ramp[::-1]; px[:65536, 2] = (ramp * 3) & 0xffff
    px[65536:131072, :3] = ramp[:, None]
    src = (px if le else px.byteswap()).reshape(-1).view(np.uint8).copy()
    exp = np.zeros_like(src)
    oracle.colorlut_rgba64(cube, src, w * 8, exp, w * 8, w, h, le=le)
    import mi355fx
    got = np.zeros_like(src)
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 3)   # the three-pass kernel pinned (round 3: RGBA64 goes to the brick kernel first)
    try:
        ctx.colorlut_frame(src, w * 8, got, w * 8, w, h, "RGBA64_LE" if le else "RGBA64_BE")
        assert ctx.colorlut_kernel_name() == "colorlut3d_lds64_kernel"
    finally:
        ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 0)
    assert (got == exp).all(), _mismatch_report(got, exp)


@pytest.mark.parametrize("le", [True, False])
@pytest.mark.parametrize("size,domain", [(33, None), (17, ((-0.25, 0.0, 0.1), (1.5, 1.0, 0.9))), (2, None), (65, None)])
@pytest.mark.parametrize("sets", [32, 64])
def test_colorlut_rgba64_brick_kernel(ctx, oracle, synth, le, size, domain, sets):
    """RGBA64 through the brick-cache kernel (round 3: x0 / t computed from the 16-bit sample instead of tabled, two pixels per
    16-byte group): random 16-bit pixels - every step misses, fills, overflows - every channel value 0..65535 on ramps, and a
    smooth picture (hits); both endiannesses, both cache geometries, LUT sizes 2..65, a non-unit domain, padded rows."""
    import mi355fx
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(size, amp=0.06, domain=domain))
    rng = np.random.default_rng(size + sets)
    w, h = 1024, 400
    px = rng.integers(0, 65536, size=(h * w, 4), dtype=np.uint16)
    ramp = np.arange(65536, dtype=np.uint16)
    px[:65536, 0] = ramp; px[:65536, 1] = ramp[::-1]; px[:65536, 2] = (ramp * 3) & 0xffff
    px[65536:131072, :3] = ramp[:, None]
    sm = synth.smooth_frame(w, 128, seed=3).reshape(128 * w, 4).astype(np.uint16)          # coherent content: cache hits
    px[131072 * 2:131072 * 2 + 128 * w] = sm * 257 + rng.integers(0, 200, size=sm.shape, dtype=np.uint16)
    ss, ds = w * 8 + 64, w * 8 + 16
    src = np.zeros((h, ss), np.uint8)
    src[:, :w * 8] = (px if le else px.byteswap()).reshape(h, w * 4).view(np.uint8)
    src = src.reshape(-1)
    exp = np.full(h * ds, 0xEE, np.uint8)
    oracle.colorlut_rgba64(cube, src, ss, exp, ds, w, h, le=le)
    got = np.full(h * ds, 0xEE, np.uint8)
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 7)
    ctx.set_flag(mi355fx.FLAG_BRICK_SETS, sets)
    try:
        ctx.colorlut_frame(src, ss, got, ds, w, h, "RGBA64_LE" if le else "RGBA64_BE")
        assert ctx.colorlut_kernel_name().startswith("colorlut3d_brick_kernel<RGBA64>")
    finally:
        ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 0)
        ctx.set_flag(mi355fx.FLAG_BRICK_SETS, 0)
    assert (got == exp).all(), _mismatch_report(got, exp)


def test_colorlut_rgba64_content_watch(ctx, oracle, synth):
    """default flags on RGBA64: coherent frames stay on the brick kernel, noise climbs to the three-pass 16-bit kernel, every
    output exact on the way"""
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(33))
    w, h = 1920, 270
    rng = np.random.default_rng(4)
    smooth = (synth.smooth_frame(w, h, seed=5).reshape(h * w, 4).astype(np.uint16) * 257).reshape(-1).view(np.uint8).copy()
    noise = rng.integers(0, 65536, size=h * w * 4, dtype=np.uint16).view(np.uint8).copy()
    seen = set()
    for name, src, reps in (("smooth", smooth, 6), ("noise", noise, 40), ("smooth", smooth, 6)):
        exp = np.zeros_like(src)
        oracle.colorlut_rgba64(cube, src, w * 8, exp, w * 8, w, h, le=True)
        for _ in range(reps):
            got = np.zeros_like(src)
            ctx.colorlut_frame(src, w * 8, got, w * 8, w, h, "RGBA64_LE")
            assert (got == exp).all(), (name, ctx.colorlut_kernel_name())
            seen.add((name, ctx.colorlut_kernel_name()))
    assert ("smooth", "colorlut3d_brick_kernel<RGBA64>") in seen
    assert ("noise", "colorlut3d_lds64_kernel") in seen


# ------------------------------------------------------------------ hsvfilter ! colorlut, one pass

def _fused_device(ctx, frames, w, h, st, src_stride=None, dst_stride=None, in_place=False):
    """Run mi355_hsv_colorlut_frames_device on host frames (n, h, stride) -> host result."""
    n = frames.shape[0]
    sstride = w * 4 if src_stride is None else src_stride
    dstride = w * 4 if dst_stride is None else dst_stride
    src = np.ascontiguousarray(frames).reshape(-1)
    out = np.full(n * h * dstride, 0x5A, np.uint8)
    d_src = ctx.alloc(src.nbytes)
    d_dst = d_src if in_place else ctx.alloc(out.nbytes)
    try:
        ctx.h2d(d_src, src)
        if not in_place:
            ctx.h2d(d_dst, out)
        ctx.hsv_colorlut_frames_device(d_src, h * sstride, sstride, d_dst, h * dstride, dstride, n, w, h, st)
        ctx.synchronize()
        ctx.d2h(out, d_dst)
        src_after = np.zeros_like(src)
        ctx.d2h(src_after, d_src)
    finally:
        ctx.free(d_src)
        if not in_place:
            ctx.free(d_dst)
    return out, src_after


def _oracle_chain(oracle, cube, frame_bytes, w, h, st):
    mid = frame_bytes.copy().reshape(-1)
    oracle.hsvfilter(mid, w, w * 4, 4, 0, False, st, nthreads=8)
    exp = np.zeros_like(mid)
    oracle.colorlut_rgba8(cube, mid, w * 4, exp, w * 4, w, h, nthreads=8)
    return exp


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("setting", ["defaults", "hue90", "mixed", "neg", "nonfinite"])
def test_fused_chain_allcolors(ctx, oracle, synth, setting, variant):
    """One-pass hsvfilter+colorlut over every 8-bit colour == oracle hsvfilter then oracle colorlut, for each
    arithmetic variant of the hsv stage (identity, +shift, -shift with affine s/v, generic/non-finite)."""
    st = {"neg": (-77.5, 1.0, 0.0, 1.0, 0.0), "nonfinite": (float("inf"), 1.3, -0.1, 0.9, 0.05)}.get(setting) or synth.HSV_SETTINGS[setting]
    import mi355fx
    ctx.set_flag(mi355fx.FLAG_FUSED_VARIANT, variant)
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(33))
    ac = synth.allcolors()
    exp = _oracle_chain(oracle, cube, ac, 4096, 4096, st)
    got, src_after = _fused_device(ctx, ac.reshape(1, 4096, 4096 * 4), 4096, 4096, st)
    assert (got == exp).all(), _mismatch_report(got, exp)
    assert (src_after == ac.reshape(-1)).all(), "fused chain must not modify its source"


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("size", [2, 17, 33, 40])
def test_fused_chain_lut_sizes_and_in_place(ctx, oracle, synth, size, variant):
    """All-resident small LUTs, the 33^3 fast path and a LUT too large for LDS (two-kernel route), in place; both
    fused kernels (inline hsv stage / software-pipelined)."""
    import mi355fx
    ctx.set_flag(mi355fx.FLAG_FUSED_VARIANT, variant)
    st = synth.HSV_SETTINGS["mixed"]
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(size))
    w, h, n = 1920, 540, 3
    frames = np.stack([synth.noise_frame(w, h, seed=11 + i) for i in range(n)])
    exp = np.concatenate([_oracle_chain(oracle, cube, frames[i], w, h, st) for i in range(n)])
    got, _ = _fused_device(ctx, frames, w, h, st, in_place=True)
    assert (got == exp).all(), _mismatch_report(got, exp)


def test_fused_chain_padded_rows_and_1d_lut_take_two_kernel_route(ctx, oracle, synth):
    """Row padding (stride > width*4) and 1D LUTs are not eligible for the fused kernel: same results through
    the element kernels, padding bytes of dst untouched, src untouched."""
    st = synth.HSV_SETTINGS["hue90"]
    w, h = 333, 37
    sstride, dstride = w * 4 + 12, w * 4 + 20
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, (1, h, sstride), dtype=np.uint8)
    for text in (synth.cube_text_3d(33), synth.cube_text_1d(64)):
        cube = _load_cube(ctx, oracle, text)
        tight = np.ascontiguousarray(src[0, :, : w * 4])
        exp = _oracle_chain(oracle, cube, tight, w, h, st).reshape(h, w * 4)
        got, src_after = _fused_device(ctx, src, w, h, st, src_stride=sstride, dst_stride=dstride)
        got = got.reshape(h, dstride)
        assert (got[:, : w * 4] == exp).all()
        assert (got[:, w * 4:] == 0x5A).all()
        assert (src_after == src.reshape(-1)).all()


def test_fused_chain_4k_batch_equals_two_element_launches(ctx, oracle, synth):
    """BASELINE headline shape (8 x 4K RGBA, hue-shift=90, 33^3): fused launch == hsvfilter launch then colorlut
    launch on the device, and both equal the oracle chain on the first frame."""
    st = synth.HSV_SETTINGS["hue90"]
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(33))
    n = 8
    frames = np.stack([synth.smooth_frame(W4K, H4K, seed=100 + i) if i % 2 == 0 else synth.noise_frame(W4K, H4K, seed=100 + i) for i in range(n)])
    fused, _ = _fused_device(ctx, frames, W4K, H4K, st)
    nbytes = frames.nbytes
    d_a, d_b = ctx.alloc(nbytes), ctx.alloc(nbytes)
    two = np.zeros(nbytes, np.uint8)
    try:
        ctx.h2d(d_a, frames.reshape(-1))
        ctx.hsvfilter_frames_device(d_a, n, H4K * W4K * 4, W4K, H4K, W4K * 4, "RGBA", st)
        ctx.colorlut_frames_device(d_a, H4K * W4K * 4, W4K * 4, d_b, H4K * W4K * 4, W4K * 4, n, W4K, H4K, "RGBA")
        ctx.synchronize()
        ctx.d2h(two, d_b)
    finally:
        ctx.free(d_a)
        ctx.free(d_b)
    assert (fused == two).all(), _mismatch_report(fused, two)
    exp0 = _oracle_chain(oracle, cube, frames[0], W4K, H4K, st)
    assert (fused[: exp0.size] == exp0).all()


@pytest.mark.parametrize("variant", [4, 5, 8])
@pytest.mark.parametrize("setting,kind,size", [("hue90", "3d", 33), ("mixed", "3d", 65), ("nonfinite", "3d", 17), ("mixed", "1d", 128)])
def test_fused_chain_table_kernel_allcolors(ctx, oracle, synth, variant, setting, kind, size):
    """The memoised table of the COMPOSED function (hsvfilter then colorlut) on every colour, for LUTs the fused
    compute kernel handles (17, 33) and ones that take the two-kernel route when the table is built (65, 1D)."""
    import mi355fx
    st = {"nonfinite": (float("inf"), 1.3, -0.1, 0.9, 0.05)}.get(setting) or synth.HSV_SETTINGS[setting]
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(size) if kind == "3d" else synth.cube_text_1d(size))
    ac = synth.allcolors()
    exp = _oracle_chain(oracle, cube, ac, 4096, 4096, st)
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, variant)
    got, src_after = _fused_device(ctx, ac.reshape(1, 4096, 4096 * 4), 4096, 4096, st)
    assert (got == exp).all(), _mismatch_report(got, exp)
    assert (src_after == ac.reshape(-1)).all()


def test_fused_chain_auto_builds_table_for_stable_settings_only(ctx, oracle, synth):
    """Auto mode on the fused entry point: no table while the hsv settings keep changing; once they have been the same
    for 8 calls the table is built and (on natural-like content) used; a settings change is honoured immediately."""
    import mi355fx
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(33))
    w, h = W4K, H4K
    frame = synth.smooth_frame(w, h, seed=77).reshape(1, h, w * 4)
    sts = [synth.HSV_SETTINGS["hue90"], synth.HSV_SETTINGS["mixed"]]
    exps = [_oracle_chain(oracle, cube, frame[0], w, h, st) for st in sts]
    for k in range(6):  # alternating settings: compute kernel every time
        got, _ = _fused_device(ctx, frame, w, h, sts[k % 2])
        assert (got == exps[k % 2]).all()
    assert ctx.colorlut_kernel_choice(fused=True) == (False, 0.0, 0.0)
    for k in range(24):
        got, _ = _fused_device(ctx, frame, w, h, sts[0])
        assert (got == exps[0]).all(), "call %d" % k
    on_table, t_c, t_t = ctx.colorlut_kernel_choice(fused=True)
    assert t_c > 0.0 and t_t > 0.0
    assert on_table == (t_t < t_c) or abs(t_t - t_c) <= 0.03 * t_c  # 3 % hysteresis
    assert on_table, (t_c, t_t)
    got, _ = _fused_device(ctx, frame, w, h, sts[1])  # new settings: must not be served from the old table
    assert (got == exps[1]).all()


def test_fused_chain_errors(ctx, synth):
    import mi355fx
    ctx.colorlut_unload()
    d = ctx.alloc(64)
    try:
        with pytest.raises(mi355fx.Mi355Error) as e:
            ctx.hsv_colorlut_frames_device(d, 64, 16, d, 64, 16, 1, 4, 4, synth.HSV_SETTINGS["defaults"])
        assert e.value.status == mi355fx.ERR_NOT_CONFIGURED
    finally:
        ctx.free(d)


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("size", [17, 33])
def test_colorlut_kernel_variants_allcolors(ctx, oracle, synth, variant, size):
    """MI355_FLAG_LUT_VARIANT: the late-prefetch and lean-state forms of the 3D LDS kernel are bit-exact too."""
    import mi355fx
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(size))
    ac = synth.allcolors()
    exp = np.zeros_like(ac)
    oracle.colorlut_rgba8(cube, ac, 4096 * 4, exp, 4096 * 4, 4096, 4096, nthreads=8)
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, variant)
    got = np.zeros_like(ac)
    ctx.colorlut_frame(ac, 4096 * 4, got, 4096 * 4, 4096, 4096, "RGBA")
    assert (got == exp).all(), _mismatch_report(got, exp)


@pytest.mark.parametrize("variant", [4, 5, 8, 0])
@pytest.mark.parametrize("kind,size", [("3d", 2), ("3d", 33), ("3d", 65), ("1d", 256)])
def test_colorlut_table_kernel_allcolors(ctx, oracle, synth, variant, kind, size):
    """The 2^24-entry memoised-table kernels (MI355_FLAG_LUT_VARIANT 4 linear / 5 Morton index through the gather kernels,
    8 Morton index through the LDS-cached kernel, and whatever 0 = auto picks on the second and later launches) on every
    colour, with a varying alpha that must pass through."""
    import mi355fx
    text = synth.cube_text_3d(size) if kind == "3d" else synth.cube_text_1d(size)
    cube = _load_cube(ctx, oracle, text)
    ac = synth.allcolors().copy()
    ac.reshape(-1, 4)[:, 3] = (np.arange(4096 * 4096, dtype=np.uint32) * 2654435761 >> 13).astype(np.uint8)
    exp = np.zeros_like(ac)
    oracle.colorlut_rgba8(cube, ac, 4096 * 4, exp, 4096 * 4, 4096, 4096, nthreads=8)
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, variant)
    for _ in range(6 if variant == 0 else 1):  # auto: interpolating kernel twice, table build + table kernel twice, then its pick
        got = np.zeros_like(ac)
        ctx.colorlut_frame(ac, 4096 * 4, got, 4096 * 4, 4096, 4096, "RGBA")
        assert (got == exp).all(), _mismatch_report(got, exp)
    if variant == 0:
        _, t_c, t_t = ctx.colorlut_kernel_choice()
        assert t_c > 0.0 and t_t > 0.0  # both kinds ran and were measured


def test_colorlut_table_follows_lut_reload(ctx, oracle, synth):
    """A new LUT invalidates the memoised table."""
    import mi355fx
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 5)
    px = synth.noise_frame(256, 64, seed=3)
    for size in (17, 33):
        cube = _load_cube(ctx, oracle, synth.cube_text_3d(size, amp=0.01 * size / 8))
        exp = np.zeros_like(px)
        oracle.colorlut_rgba8(cube, px, 256 * 4, exp, 256 * 4, 256, 64)
        got = np.zeros_like(px)
        ctx.colorlut_frame(px, 256 * 4, got, 256 * 4, 256, 64, "RGBA")
        assert (got == exp).all()


@pytest.mark.parametrize("variant", [4, 5, 8])
@pytest.mark.parametrize("w,h", [(4, 1), (100, 37), (128, 4), (516, 3), (1920, 1081), (3840, 7), (1000, 9), (256, 8), (260, 17), (1924, 1083), (132, 4000)])
def test_colorlut_table_kernel_chunk_tails(ctx, oracle, synth, variant, w, h):
    """Frame sizes around the table kernels' 512-pixel wave patches: flat kernel (width < 128 or not a multiple of 4) and the
    tiled kernel with 128- and 256-pixel-wide patches, masked last columns and rows."""
    import mi355fx
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(17))
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, variant)
    src = synth.noise_frame(w, h, seed=w + h).reshape(-1)
    exp = np.zeros_like(src)
    oracle.colorlut_rgba8(cube, src, w * 4, exp, w * 4, w, h)
    got = np.full_like(src, 0x5A)
    ctx.colorlut_frame(src, w * 4, got, w * 4, w, h, "RGBA")
    assert (got == exp).all(), _mismatch_report(got, exp)


def test_colorlut_table_geometry_fallback(ctx, oracle, synth):
    """Padded rows and odd sizes are not eligible for the table kernel; the flag must not change the result."""
    import mi355fx
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(33))
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 5)
    w, h, stride = 37, 11, 37 * 4 + 12
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, size=h * stride, dtype=np.uint8)
    exp = np.zeros_like(src)
    oracle.colorlut_rgba8(cube, src, stride, exp, stride, w, h)
    got = np.zeros_like(src)
    ctx.colorlut_frame(src, stride, got, stride, w, h, "RGBA")
    rows = lambda a: np.stack([a[y * stride: y * stride + w * 4] for y in range(h)])
    assert (rows(got) == rows(exp)).all()


def test_colorlut_auto_stays_exact_when_the_content_changes(ctx, oracle, synth):
    """Auto mode across a change of content (natural-like frames, then uniform noise): whatever kernels the policy picks on
    the way, the output is the oracle's. (What it picks, and when, is a matter of in-stream timings: tests/test_gpu_zz_timing.py.)"""
    import mi355fx
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(33))
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 0)
    n = 2
    nb = n * W4K * H4K * 4
    d_s, d_n, d_o = ctx.alloc(nb), ctx.alloc(nb), ctx.alloc(nb)
    try:
        smooth = np.stack([synth.smooth_frame(W4K, H4K, seed=40 + i) for i in range(n)]).reshape(-1)
        noise = np.stack([synth.noise_frame(W4K, H4K, seed=50 + i) for i in range(n)]).reshape(-1)
        ctx.h2d(d_s, smooth)
        ctx.h2d(d_n, noise)
        run = lambda d: ctx.colorlut_frames_device(d, H4K * W4K * 4, W4K * 4, d_o, H4K * W4K * 4, W4K * 4, n, W4K, H4K, "RGBA")
        exp = np.zeros(W4K * H4K * 4, np.uint8)
        out = np.zeros_like(smooth)
        seen = set()
        for src_host, d in ((smooth, d_s), (noise, d_n), (smooth, d_s)):
            oracle.colorlut_rgba8(cube, src_host[: exp.size], W4K * 4, exp, W4K * 4, W4K, H4K, nthreads=8)
            for k in range(40):
                run(d)
                ctx.synchronize()
                seen.add(ctx.colorlut_kernel_name())
                if k % 8 == 7 or k < 6:   # the learning launches and a sample of the rest
                    ctx.d2h(out, d_o)
                    assert (out[: exp.size] == exp).all(), (k, ctx.colorlut_kernel_name())
        assert len(seen) >= 2, seen   # more than one kernel kind did serve (learning alone runs two)
    finally:
        for d in (d_s, d_n, d_o):
            ctx.free(d)


@pytest.mark.parametrize("st", [(0.0, 10.0, 0.0, 0.15, 0.0, 0.3), (120.0, 40.0, 0.8, 0.5, 0.7, 0.6), (300.0, 45.0, 0.5, 0.5, 0.5, 0.5),
                                (-725.0, 30.0, 0.5, 0.5, 0.5, 0.5)])
@pytest.mark.parametrize("in_fmt,out_fmt", [("RGB", "RGBA"), ("BGR", "ARGB"), ("RGB", "ABGR"), ("BGR", "BGRA")])
def test_hsvdetect_rgb24_allcolors(ctx, oracle, synth, st, in_fmt, out_fmt):
    """3-byte input formats on every colour: the 12-byte vector kernel (and the literal kernel for hue-ref outside
    [-180,180]) against the oracle."""
    bgr = in_fmt == "BGR"
    af, obgr = {"RGBA": (0, 0), "ARGB": (1, 0), "BGRA": (0, 1), "ABGR": (1, 1)}[out_fmt]
    rgb = synth.allcolors().reshape(-1, 4)[:, :3]
    src = np.ascontiguousarray(rgb[:, ::-1] if bgr else rgb).reshape(4096, 4096 * 3)
    exp = np.zeros((4096, 4096 * 4), np.uint8)
    oracle.hsvdetect(src, 4096 * 3, 3, 0, bgr, exp, 4096 * 4, bool(af), bool(obgr), 4096, st)
    got = np.zeros_like(exp)
    ctx.hsvdetect_frame(src, 4096 * 3, in_fmt, got, 4096 * 4, out_fmt, 4096, st)
    assert (got == exp).all(), _mismatch_report(got, exp)


def test_device_entry_points_reject_overlapping_frames(ctx, oracle, synth):
    """n_frames > 1 with a frame pitch smaller than one frame would make launches read and write overlapping frames:
    INVALID_ARG instead."""
    import mi355fx
    _load_cube(ctx, oracle, synth.cube_text_3d(5))
    w, h = 64, 8
    d = ctx.alloc(4 * w * h * 4)
    try:
        st = synth.HSV_SETTINGS["hue90"]
        for call in (lambda: ctx.hsvfilter_frames_device(d, 2, w * 4 * (h - 1), w, h, w * 4, "RGBA", st),
                     lambda: ctx.colorlut_frames_device(d, w * 4, w * 4, d, w * 4 * h, w * 4, 2, w, h, "RGBA"),
                     lambda: ctx.colorlut_frames_device(d, w * 4 * h, w * 4, d, w * 4, w * 4, 2, w, h, "RGBA"),
                     lambda: ctx.hsv_colorlut_frames_device(d, w * 4 * h, w * 4, d, 16, w * 4, 2, w, h, st)):
            with pytest.raises(mi355fx.Mi355Error) as e:
                call()
            assert e.value.status == mi355fx.ERR_INVALID_ARG
        # exactly one frame apart is fine, and a single frame ignores the pitch
        ctx.hsvfilter_frames_device(d, 2, w * 4 * h, w, h, w * 4, "RGBA", st)
        ctx.colorlut_frames_device(d, 0, w * 4, d, 0, w * 4, 1, w, h, "RGBA")
        ctx.synchronize()
    finally:
        ctx.free(d)


def test_memoised_tables_are_shared_between_contexts(oracle, synth, mi355lib):
    """Contexts that load the same LUT use ONE 64 MiB table (32 streams with one LUT: one table, not 32); a different LUT
    gets its own; tables go away with their last user; results stay exact throughout."""
    import mi355fx
    base = mi355lib.mi355_shared_table_count()
    text_a, text_b = synth.cube_text_3d(33), synth.cube_text_3d(17, amp=0.09)
    cube_a, cube_b = oracle.Cube.parse(text_a), oracle.Cube.parse(text_b)
    frame = synth.smooth_frame(512, 64)
    exp_a, exp_b = np.zeros_like(frame), np.zeros_like(frame)
    oracle.colorlut_rgba8(cube_a, frame, 512 * 4, exp_a, 512 * 4, 512, 64)
    oracle.colorlut_rgba8(cube_b, frame, 512 * 4, exp_b, 512 * 4, 512, 64)
    ctxs = [mi355fx.Context(0) for _ in range(5)]
    try:
        for i, c in enumerate(ctxs):
            cube = cube_b if i == 4 else cube_a
            sc, of = cube.domain
            c.colorlut_load(cube.is3d, cube.size, cube.table, sc, of)
            c.set_flag(mi355fx.FLAG_LUT_VARIANT, 5)
        for rep in range(2):
            for i, c in enumerate(ctxs):
                got = np.zeros_like(frame)
                c.colorlut_frame(frame, 512 * 4, got, 512 * 4, 512, 64, "RGBA")
                assert (got == (exp_b if i == 4 else exp_a)).all(), (rep, i)
        assert mi355lib.mi355_shared_table_count() == base + 2
        ctxs[4].close()
        assert mi355lib.mi355_shared_table_count() == base + 1
        ctxs[0].colorlut_unload()
        assert mi355lib.mi355_shared_table_count() == base + 1      # three users left
    finally:
        for c in ctxs[:4]:
            c.close()
    assert mi355lib.mi355_shared_table_count() == base


def test_a_table_rebuilt_in_place_waits_for_the_streams_that_used_to_share_it(oracle, synth, mi355lib):
    """Two contexts share the table of hsvfilter -> colorlut under the same settings. B queues a long run of launches on it
    and moves to other settings WITHOUT waiting (its reference goes); A, now the sole owner, changes its settings too and
    rebuilds the 64 MiB buffer in place on ITS stream. B's queued launches must still read the old contents: the table
    remembers an event on B's stream and A's build waits for it in stream order (no host wait anywhere)."""
    import mi355fx
    cube = oracle.Cube.parse(synth.cube_text_3d(33))
    sc, of = cube.domain
    st_old, st_b, st_a = synth.HSV_SETTINGS["hue90"], synth.HSV_SETTINGS["mixed"], (45.0, 0.9, 0.01, 1.1, -0.02)
    n = 4
    frames = np.stack([synth.smooth_frame(W4K, H4K, seed=300 + i) for i in range(n)]).reshape(-1)
    fb = W4K * H4K * 4
    exp_old = _oracle_chain(oracle, cube, frames[:fb], W4K, H4K, st_old)
    exp_a = _oracle_chain(oracle, cube, frames[:fb], W4K, H4K, st_a)
    base = mi355lib.mi355_shared_table_count()
    a, b = mi355fx.Context(0), mi355fx.Context(0)
    bufs = []
    try:
        for c in (a, b):
            c.colorlut_load(cube.is3d, cube.size, cube.table, sc, of)
            c.set_flag(mi355fx.FLAG_LUT_VARIANT, 5)
        d_src, d_a, d_b, d_b2 = (a.alloc(frames.nbytes) for _ in range(4))
        bufs = [d_src, d_a, d_b, d_b2]
        a.h2d(d_src, frames)
        a.synchronize()
        run = lambda c, dst, st: c.hsv_colorlut_frames_device(d_src, fb, W4K * 4, dst, fb, W4K * 4, n, W4K, H4K, st)
        run(a, d_a, st_old)
        run(b, d_b, st_old)
        a.synchronize(); b.synchronize()
        assert mi355lib.mi355_shared_table_count() == base + 1          # one table, two users
        for _ in range(40):                                              # B: ~10 ms of queued work on the shared table
            run(b, d_b, st_old)
        run(b, d_b2, st_b)                                               # B lets go of it (builds its own), nothing waited for
        run(a, d_a, st_a)                                                # A: sole owner now -> rebuilds the buffer in place
        a.synchronize(); b.synchronize()
        got_b, got_a = np.zeros(fb, np.uint8), np.zeros(fb, np.uint8)
        b.d2h(got_b, d_b)
        a.d2h(got_a, d_a)
        assert (got_b == exp_old).all(), _mismatch_report(got_b, exp_old)
        assert (got_a == exp_a).all(), _mismatch_report(got_a, exp_a)
    finally:
        for d in bufs:
            a.free(d)
        a.close(); b.close()
    assert mi355lib.mi355_shared_table_count() == base


def test_issue_streams_round_is_the_two_calls_per_stream(ctx, oracle, synth):
    """mi355_issue_streams_round (bench plumbing: one native loop instead of 2 x n interpreter calls) gives every stream exactly
    what mi355_hsvfilter_frames_device + mi355_colorlut_frames_device give it."""
    import mi355fx
    w, h, n = 256, 64, 3
    st = synth.HSV_SETTINGS["hue90"]
    ctxs = [mi355fx.Context(0) for _ in range(n)]
    cube = None
    for c in ctxs:
        cube = _load_cube(c, oracle, synth.cube_text_3d(9))
    rng = np.random.default_rng(11)
    frames = [rng.integers(0, 256, (h, w * 4), dtype=np.uint8) for _ in range(n)]
    d_src = [c.alloc(f.nbytes) for c, f in zip(ctxs, frames)]
    d_dst = [c.alloc(f.nbytes) for c, f in zip(ctxs, frames)]
    try:
        for c, d, f in zip(ctxs, d_src, frames):
            c.h2d(d, f.reshape(-1))
        mi355fx.StreamsRound(ctxs, w, h, w * 4, "RGBA", st).issue(d_src, d_dst)
        for c, f, ds, dd in zip(ctxs, frames, d_src, d_dst):
            c.synchronize()
            filt = f.copy()
            oracle.hsvfilter(filt, w, w * 4, 4, 0, False, st)
            exp = np.zeros_like(f)
            oracle.colorlut_rgba8(cube, filt, w * 4, exp, w * 4, w, h)
            got_f, got = np.zeros_like(f), np.zeros_like(f)
            c.d2h(got_f, ds); c.d2h(got, dd)
            assert (got_f == filt).all() and (got == exp).all()
    finally:
        for c, a, b in zip(ctxs, d_src, d_dst):
            c.free(a); c.free(b); c.close()


@pytest.mark.parametrize("le", [True, False])
@pytest.mark.parametrize("size,domain", [(2, None), (64, None), (1024, None), (4096, ((-0.1, 0.0, 0.2), (1.2, 1.0, 0.7))), (12000, None)])
def test_colorlut_rgba64_1d_lds_kernel(ctx, oracle, synth, le, size, domain):
    """RGBA64 + a 1D LUT whose three tables fit LDS, on contiguous frames (round 3: colorlut1d_lds64_kernel instead of the
    one-pixel-per-lane literal kernel): random 16-bit pixels plus every channel value 0..65535 on ramps, both byte orders, LUT
    sizes 2..12000, a non-unit domain; a host frame with padded rows agrees too (padding untouched)."""
    text = synth.cube_text_1d(size, gamma=0.6)
    if domain:
        text = text.replace("LUT_1D_SIZE", "DOMAIN_MIN %g %g %g\nDOMAIN_MAX %g %g %g\nLUT_1D_SIZE" % (domain[0] + domain[1]), 1)
    cube = _load_cube(ctx, oracle, text)
    rng = np.random.default_rng(size)
    w, h = 1024, 160
    px = rng.integers(0, 65536, size=(h * w, 4), dtype=np.uint16)
    ramp = np.arange(65536, dtype=np.uint16)
    px[:65536, 0] = ramp; px[:65536, 1] = ramp[::-1]; px[:65536, 2] = (ramp * 3) & 0xffff
    px[65536:131072, :3] = ramp[:, None]
    src = (px if le else px.byteswap()).reshape(-1).view(np.uint8).copy()
    fmt = "RGBA64_LE" if le else "RGBA64_BE"
    exp = np.zeros_like(src)
    oracle.colorlut_rgba64(cube, src, w * 8, exp, w * 8, w, h, le=le)
    got = np.zeros_like(src)
    ctx.colorlut_frame(src, w * 8, got, w * 8, w, h, fmt)
    assert ctx.colorlut_kernel_name() == "colorlut1d_lds64_kernel"
    assert (got == exp).all(), _mismatch_report(got, exp)
    # a host frame with padded rows
    ss = w * 8 + 24
    srcp = np.zeros((h, ss), np.uint8); srcp[:, : w * 8] = src.reshape(h, w * 8)
    expp, gotp = np.full(h * ss, 0xAB, np.uint8), np.full(h * ss, 0xAB, np.uint8)
    oracle.colorlut_rgba64(cube, srcp.reshape(-1), ss, expp, ss, w, h, le=le)
    ctx.colorlut_frame(srcp.reshape(-1), ss, gotp, ss, w, h, fmt)
    assert (gotp == expp).all()


@pytest.mark.parametrize("fmt", ["RGBA", "xBGR"])
def test_hsvfilter_rowpad_kernel_4k_batch_equals_the_packed_kernel(ctx, synth, fmt):
    """Full size: 3 x 4K frames with rows padded by 64 bytes (frames back to back: hsvfilter_rowpad_kernel, round 5) against the same
    pixels packed (hsvfilter_flat_kernel): identical pixel bytes, padding untouched. (Both against the oracle at small sizes above.)"""
    w, h, n, pad = 3840, 2160, 3, 64
    st = synth.HSV_SETTINGS["mixed"]
    rng = np.random.default_rng(77)
    packed = np.concatenate([synth.smooth_frame(w, h, seed=60 + f).reshape(-1) if f != 1 else synth.noise_frame(w, h, seed=61).reshape(-1) for f in range(n)])
    stride = w * 4 + pad
    padded = rng.integers(0, 256, size=(n * h, stride), dtype=np.uint8)
    padded[:, :w * 4] = packed.reshape(n * h, w * 4)
    keep = padded.copy()
    dp, dq = ctx.alloc(packed.nbytes), ctx.alloc(padded.nbytes)
    try:
        ctx.h2d(dp, packed); ctx.h2d(dq, padded.reshape(-1))
        ctx.hsvfilter_frames_device(dp, n, w * h * 4, w, h, w * 4, fmt, st)
        ctx.hsvfilter_frames_device(dq, n, stride * h, w, h, stride, fmt, st)
        ctx.synchronize()
        a, b = np.zeros_like(packed), np.zeros(padded.size, np.uint8)
        ctx.d2h(a, dp); ctx.d2h(b, dq)
    finally:
        ctx.free(dp); ctx.free(dq)
    b = b.reshape(n * h, stride)
    assert (b[:, :w * 4] == a.reshape(n * h, w * 4)).all()
    assert (b[:, w * 4:] == keep[:, w * 4:]).all()
