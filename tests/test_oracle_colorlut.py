"""Pins the CPU oracle (and the product-side host parser) for the colorlut path: the reference's own
parser known-answer tests (video/colorlut/src/parser.rs:381-473), grammar edge cases derived from
the source, and the pixel math against the independent numpy restatement."""
import numpy as np
import pytest

# --- the five reference parser tests, text verbatim as data (parser.rs:381-473)
LUT3D_2 = """
            LUT_3D_SIZE 2

            0.0 0.0 0.0
            1.0 0.0 0.0
            0.0 1.0 0.0
            1.0 1.0 0.0
            0.0 0.0 1.0
            1.0 0.0 1.0
            0.0 1.0 1.0
            1.0 1.0 1.0
        """
KEYWORD_AFTER_SIZE = """
            LUT_1D_SIZE 2

            TITLE "test"
            DOMAIN_MIN 0.0 0.0 0.0
            DOMAIN_MAX 1.0 1.0 1.0

            0.0 0.0 0.0
            1.0 0.5 0.7
        """
KEYWORD_AFTER_DATA = """
            LUT_1D_SIZE 2

            0.0 0.0 0.0
            1.0 0.0 0.0
            TITLE "invalid"
        """
KEYWORD_BETWEEN_DATA = """
            LUT_1D_SIZE 2

            0.0 0.0 0.0
            TITLE "invalid"
            1.0 0.0 0.0
        """
MULTIPLE_SIZES = """
            LUT_1D_SIZE 2
            LUT_3D_SIZE 2

            0.0 0.0 0.0
            1.0 1.0 1.0
        """


class _Parsers:
    """Both parsers under test behind one face: the oracle's (C) and the product's host mirror (C++)."""

    def __init__(self, oracle):
        from mi355fx.cube import parse_cube, CubeParseError
        self.items = {
            "oracle": (lambda t: self._norm_o(oracle.Cube.parse(t)), ValueError),
            "host": (lambda t: self._norm_h(parse_cube(t)), CubeParseError),
        }

    @staticmethod
    def _norm_o(c):
        s, o = c.domain
        return dict(is3d=c.is3d, size=c.size, table=c.table, scale=s, offset=o)

    @staticmethod
    def _norm_h(c):
        return dict(is3d=c.is3d, size=c.size, table=c.table, scale=c.domain_scale, offset=c.domain_offset)


@pytest.fixture(scope="module")
def parsers(oracle):
    return _Parsers(oracle).items


@pytest.mark.parametrize("which", ["oracle", "host"])
def test_parse_3d_lut(parsers, which):  # parser.rs:381-409
    parse, _ = parsers[which]
    lut = parse(LUT3D_2)
    assert lut["is3d"] and lut["size"] == 2 and lut["table"].size == 8 * 4
    flat = lut["table"].reshape(8, 4)
    assert list(flat[0]) == [0.0, 0.0, 0.0, 1.0]          # at(0,0,0)
    assert list(flat[1 + 1 * 2 + 1 * 4]) == [1.0, 1.0, 1.0, 1.0]  # at(1,1,1)


@pytest.mark.parametrize("which", ["oracle", "host"])
def test_keyword_after_lut_size(parsers, which):  # parser.rs:411-436
    parse, _ = parsers[which]
    lut = parse(KEYWORD_AFTER_SIZE)
    assert not lut["is3d"] and lut["size"] == 2
    r, g, b = lut["table"].reshape(3, 2)
    assert list(r) == [np.float32(0.0), np.float32(1.0)]
    assert list(g) == [np.float32(0.0), np.float32(0.5)]
    assert list(b) == [np.float32(0.0), np.float32(0.7)]


@pytest.mark.parametrize("which", ["oracle", "host"])
@pytest.mark.parametrize("text", [KEYWORD_AFTER_DATA, KEYWORD_BETWEEN_DATA, MULTIPLE_SIZES])
def test_reference_rejections(parsers, which, text):  # parser.rs:438-473
    parse, exc = parsers[which]
    with pytest.raises(exc):
        parse(text)


ACCEPT = [
    "LUT_1D_SIZE 2\n0 0 0\n1 1 1",                       # no trailing newline
    "LUT_1D_SIZE 2\r\n0 0 0\r\n1 1 1\r\n",               # CRLF (str::lines strips \r)
    "# c\n\n  # indented comment\nLUT_1D_SIZE +2\n0 0 0\n1 1 1\n",   # usize accepts '+'
    "TITLE a b c d\nLUT_1D_SIZE 2\n.5 5. +1e0\ninf -inf nan\n",      # float spellings Rust accepts
    "LUT_1D_SIZE 2\nINFINITY NaN 1E2\n0 0 0\n",
    "LUT_1D_SIZE 2\n0 0　0\n1\t1\x0b1\n",                    # Unicode whitespace separators
    "DOMAIN_MIN -1 -1 -1\nDOMAIN_MAX 2 2 2\nLUT_1D_SIZE 2\n0 0 0\n1 1 1\n",
    "LUT_1D_SIZE 2\nDOMAIN_MAX nan 1 1\n0 0 0\n1 1 1\n",               # NaN bound passes `min >= max`
    "LUT_3D_SIZE 2\n" + "1e40 -1e40 1e-50\n" * 8,                       # overflow -> inf, underflow -> 0
]
REJECT = [
    "", "LUT_1D_SIZE 2\n", "LUT_1D_SIZE 1\n0 0 0\n", "LUT_1D_SIZE 65537\n", "LUT_3D_SIZE 257\n", "LUT_3D_SIZE 1\n",
    "LUT_1D_SIZE 2\n0 0\n1 1 1\n", "LUT_1D_SIZE 2\n0 0 0 0\n1 1 1\n", "LUT_1D_SIZE 2\n0 0 0\n",
    "LUT_1D_SIZE 2\n0 0 0\n1 1 1\n2 2 2\n", "LUT_1D_SIZE -2\n", "LUT_1D_SIZE 2 3\n", "LUT_1D_SIZE\n",
    "LUT_1D_SIZE 2.0\n", "LUT_1D_SIZE 2\n0x10 0 0\n1 1 1\n", "LUT_1D_SIZE 2\n1_0 0 0\n1 1 1\n",
    "LUT_1D_SIZE 2\n. 0 0\n1 1 1\n", "LUT_1D_SIZE 2\n1e 0 0\n1 1 1\n", "LUT_1D_SIZE 2\ne5 0 0\n1 1 1\n",
    "LUT_1D_SIZE 2\nnan(1) 0 0\n1 1 1\n", "LUT_1D_SIZE 2\n+ 0 0\n1 1 1\n",
    "DOMAIN_MIN 0 0\nLUT_1D_SIZE 2\n0 0 0\n1 1 1\n", "DOMAIN_MIN 0 0 0 0\nLUT_1D_SIZE 2\n0 0 0\n1 1 1\n",
    "DOMAIN_MIN 1 0 0\nLUT_1D_SIZE 2\n0 0 0\n1 1 1\n",               # min >= max
    "DOMAIN_MAX 0 1 1\nLUT_1D_SIZE 2\n0 0 0\n1 1 1\n",
    "lut_1d_size 2\n0 0 0\n1 1 1\n",                                   # keywords are case-sensitive -> data before size
    "﻿LUT_1D_SIZE 2\n0 0 0\n1 1 1\n",                             # BOM is not whitespace
    "LUT_1D_SIZE 2\n0 0 0\n1 1 1\nLUT_1D_SIZE 2\n",
    "LUT_3D_SIZE 2\n0 0 0\n",
]


@pytest.mark.parametrize("which", ["oracle", "host"])
@pytest.mark.parametrize("idx", range(len(ACCEPT)))
def test_grammar_accepts(parsers, which, idx):
    parse, _ = parsers[which]
    parse(ACCEPT[idx])


@pytest.mark.parametrize("which", ["oracle", "host"])
@pytest.mark.parametrize("idx", range(len(REJECT)))
def test_grammar_rejects(parsers, which, idx):
    parse, exc = parsers[which]
    with pytest.raises(exc):
        parse(REJECT[idx])


def test_invalid_utf8_is_an_error(oracle):
    from mi355fx.cube import parse_cube, CubeParseError
    bad = b"LUT_1D_SIZE 2\n0 0 0\n1 1 \xff1\n"
    with pytest.raises(ValueError):
        oracle.Cube.parse(bad)
    with pytest.raises(CubeParseError):
        parse_cube(bad)


@pytest.mark.parametrize("text_fn", ["3d33", "3d17dom", "1d"])
def test_host_parser_equals_oracle_parser_bitwise(oracle, synth, text_fn):
    from mi355fx.cube import parse_cube
    text = {"3d33": synth.cube_text_3d(33), "3d17dom": synth.cube_text_3d(17, domain=((-0.5, 0.0, 0.25), (1.5, 2.0, 0.75))),
            "1d": synth.cube_text_1d(1024)}[text_fn]
    a, b = parse_cube(text), oracle.Cube.parse(text)
    sc, of = b.domain
    assert a.is3d == b.is3d and a.size == b.size
    assert a.table.tobytes() == b.table.tobytes()
    assert a.domain_scale.tobytes() == sc.tobytes() and a.domain_offset.tobytes() == of.tobytes()


def test_parse_file(tmp_path, synth):
    from mi355fx.cube import parse_cube_file, CubeParseError
    p = tmp_path / "x.cube"
    p.write_text(synth.cube_text_3d(5))
    assert parse_cube_file(str(p)).size == 5
    with pytest.raises(CubeParseError):
        parse_cube_file(str(tmp_path / "missing.cube"))


# ------------------------------------------------------------------ pixel math

@pytest.mark.parametrize("size", [2, 17, 33, 65])
def test_identity_lut_is_passthrough(oracle, synth, size):
    """SURVEY.md §8 a6 survey-derived KAT."""
    cube = oracle.Cube.parse(synth.cube_text_3d(size, identity=True))
    ac = synth.allcolors()[:1024]  # 4M colours: all r,g and b in 0..63
    edge = synth.allcolors()[-256:]
    for src in (ac, edge):
        dst = np.zeros_like(src)
        oracle.colorlut_rgba8(cube, src, 4096 * 4, dst, 4096 * 4, 4096, src.shape[0], nthreads=8)
        assert (dst == src).all()


@pytest.mark.parametrize("case", ["33", "17dom", "9wild"])
def test_c_oracle_matches_numpy_restatement_3d(oracle, synth, case):
    from oracle import np_restate as N
    if case == "33":
        text = synth.cube_text_3d(33)
    elif case == "17dom":
        text = synth.cube_text_3d(17, amp=0.1, domain=((-0.25, 0.0, 0.1), (1.5, 1.0, 0.9)))
    else:
        rng = np.random.default_rng(5)
        text = "LUT_3D_SIZE 9\n" + "\n".join("%.7g %.7g %.7g" % tuple(v) for v in rng.uniform(-0.5, 1.5, (729, 3))) + "\n"
    cube = oracle.Cube.parse(text)
    sc, of = cube.domain
    rng = np.random.default_rng(1)
    n = 1 << 18
    px = rng.integers(0, 256, size=(n, 4), dtype=np.uint8)
    px[:4096, :3] = np.array([[a, b, c] for a in (0, 1, 7, 8, 127, 128, 254, 255) * 2 for b in (0, 8, 15, 16, 247, 248, 254, 255) * 2
                              for c in (0, 1, 127, 128, 135, 136, 254, 255) * 2], np.uint8)[:4096]
    dst = np.zeros_like(px)
    oracle.colorlut_rgba8(cube, px, n * 4, dst, n * 4, n, 1)
    r, g, b = N.colorlut3d_rgb(px[:, 0], px[:, 1], px[:, 2], cube.size, cube.table.reshape(-1, 4), sc, of)
    assert (dst[:, 0] == r).all() and (dst[:, 1] == g).all() and (dst[:, 2] == b).all()
    assert (dst[:, 3] == px[:, 3]).all()  # alpha copied (imp.rs:291)


def test_c_oracle_matches_numpy_restatement_1d(oracle, synth):
    from oracle import np_restate as N
    cube = oracle.Cube.parse(synth.cube_text_1d(64))
    sc, of = cube.domain
    ramp = np.arange(256, dtype=np.uint8)
    px = np.stack([ramp, ramp[::-1], (ramp * 7) & 255, ramp], axis=1).copy()
    dst = np.zeros_like(px)
    oracle.colorlut_rgba8(cube, px, 1024, dst, 1024, 256, 1)
    planes = cube.table.reshape(3, -1)
    for c in range(3):
        assert (dst[:, c] == N.colorlut1d(px[:, c], c, cube.size, planes, sc, of)).all()


def test_rgba64_le_be_are_byte_swaps_of_each_other(oracle, synth):
    """colorlut/imp.rs:350-396: BE = byteswap(in) -> same math -> byteswap(out); alpha word raw."""
    cube = oracle.Cube.parse(synth.cube_text_3d(17))
    rng = np.random.default_rng(2)
    w, h = 64, 3
    le = rng.integers(0, 65536, size=(h, w * 4), dtype=np.uint16)
    be = le.byteswap()
    out_le, out_be = np.zeros_like(le), np.zeros_like(be)
    oracle.colorlut_rgba64(cube, le.view(np.uint8), w * 8, out_le.view(np.uint8), w * 8, w, h, le=True)
    oracle.colorlut_rgba64(cube, be.view(np.uint8), w * 8, out_be.view(np.uint8), w * 8, w, h, le=False)
    assert (out_be.byteswap() == out_le).all()
    assert (out_le[:, 3::4] == le[:, 3::4]).all()
    # 16-bit identity: value v -> clamp(v/65535)*... identity LUT passes through
    ident = oracle.Cube.parse(synth.cube_text_3d(33, identity=True))
    out = np.zeros_like(le)
    oracle.colorlut_rgba64(ident, le.view(np.uint8), w * 8, out.view(np.uint8), w * 8, w, h, le=True)
    assert np.abs(out.astype(np.int32) - le.astype(np.int32)).max() <= 1


def test_parsers_agree_on_fuzzed_cube_texts(parsers):
    """Property test (hypothesis, seeded/derandomised): the two independent restatements of CubeLut::parse — the oracle's C
    parser and the product's C++ host reader — agree on accept/reject and, when both accept, on every parsed value, for
    texts assembled from valid and invalid .cube ingredients (keywords in any order, comments, blank and odd-whitespace
    lines, malformed numbers, wrong counts)."""
    from hypothesis import given, settings, strategies as st, HealthCheck

    num = st.one_of(st.sampled_from(["0", "1", "0.5", "1.0", "-0.25", "1e-3", "+0.75", ".5", "5.", "1e400", "nan", "inf", "abc", "0x10", "1,0", ""]),
                    st.floats(-2, 2, allow_nan=False, width=32).map(lambda v: "%.6f" % v))
    data_line = st.lists(num, min_size=1, max_size=5).map(" ".join)
    keyword = st.sampled_from(["LUT_3D_SIZE 2", "LUT_3D_SIZE 3", "LUT_1D_SIZE 2", "LUT_1D_SIZE 4", "LUT_3D_SIZE", "LUT_3D_SIZE x", "LUT_1D_SIZE -1",
                               "LUT_3D_SIZE 1", "LUT_3D_SIZE 300", "LUT_1D_SIZE 70000", 'TITLE "t"', "TITLE", "DOMAIN_MIN 0 0 0", "DOMAIN_MAX 1 1 1",
                               "DOMAIN_MIN 0.1 0.2 0.3", "DOMAIN_MAX 2 2 2", "DOMAIN_MIN 1 1 1", "DOMAIN_MAX 0 0", "DOMAIN_MAX a b c",
                               "LUT_1D_INPUT_RANGE 0 1", "LUT_3D_INPUT_RANGE 0 1", "UNKNOWN_KEY 1", "# comment", "#", "", "   ", "\t",
                               " LUT_3D_SIZE 2", "LUT_3D_SIZE 2"])
    sep = st.sampled_from(["\n", "\r\n", "\n\n", " \n"])

    @settings(max_examples=400, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(st.lists(st.one_of(keyword, data_line), min_size=0, max_size=14), sep, st.integers(0, 27))
    def run(lines, joiner, pad_rows):
        body = joiner.join(lines)
        if pad_rows:   # often complete the table so that acceptance paths are exercised too
            body += joiner + joiner.join("%.3f %.3f %.3f" % (k / 27.0, (k * 7 % 27) / 27.0, (k * 11 % 27) / 27.0) for k in range(pad_rows))
        res = {}
        for name, (fn, exc) in parsers.items():
            try:
                res[name] = fn(body)
            except exc:
                res[name] = None
        a, b = res["oracle"], res["host"]
        assert (a is None) == (b is None), (body, a is None, b is None)
        if a is not None:
            assert a["is3d"] == b["is3d"] and a["size"] == b["size"]
            assert np.array_equal(np.asarray(a["table"]).view(np.uint32), np.asarray(b["table"]).view(np.uint32))
            assert np.array_equal(np.asarray(a["scale"]).view(np.uint32), np.asarray(b["scale"]).view(np.uint32))
            assert np.array_equal(np.asarray(a["offset"]).view(np.uint32), np.asarray(b["offset"]).view(np.uint32))

    run()


def test_host_parser_does_not_depend_on_the_process_locale(synth):
    """A GStreamer process runs under setlocale(LC_ALL, ""); Rust's f32 parser knows no locale. The host parser must accept
    the same '.'-decimal .cube text under a comma-decimal LC_NUMERIC (plain strtof stopped at the '.')."""
    import ctypes
    import locale
    from mi355fx.cube import parse_cube
    text = synth.cube_text_3d(5)
    ref = parse_cube(text)
    libc = ctypes.CDLL(None)
    libc.setlocale.restype = ctypes.c_char_p
    old = libc.setlocale(locale.LC_NUMERIC, None)
    switched = None
    for name in (b"de_DE.UTF-8", b"de_DE.utf8", b"fr_FR.UTF-8", b"de_DE", b"ru_RU.UTF-8"):
        if libc.setlocale(locale.LC_NUMERIC, name):
            switched = name
            break
    try:
        got = parse_cube(text)  # must parse whatever LC_NUMERIC is
        assert got.size == ref.size and (np.asarray(got.table) == np.asarray(ref.table)).all()
    finally:
        libc.setlocale(locale.LC_NUMERIC, old)
    if switched is None:
        pytest.skip("no comma-decimal locale installed in this image: parsed under the C locale only")
