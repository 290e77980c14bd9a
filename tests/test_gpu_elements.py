"""Element-level tests on the GPU: the host-side mirror of the reference elements (properties, caps,
start/stop, transform vfuncs) driving the HIP path, checked against the oracle. They read like the
reference's Harness tests: make element, set properties, push a buffer, inspect the output."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "element_surface.json")


def test_introspection_matches_plugin_cache():
    """Factory/GType names, klass, property types/defaults/ranges/mutability and pad-template formats,
    compared with tests/golden/element_surface.json (extracted from docs/plugins/gst_plugins_cache.json)."""
    from mi355fx.elements import Element
    want = json.load(open(GOLDEN))
    for factory, spec in want.items():
        e = Element(factory)
        assert e.type_name == spec["type_name"] and e.klass == spec["klass"]
        props = e.properties()
        assert sorted(props) == sorted(spec["properties"])
        for name, p in spec["properties"].items():
            got = props[name]
            assert got["mutable"] == p["mutable"], (factory, name)
            if p["type"] in ("gfloat", "gdouble", "guint64"):
                assert got["default"] == pytest.approx(float(p["default"]), rel=1e-6)
                assert got["min"] == pytest.approx(float(p["min"]), rel=1e-5) and got["max"] == pytest.approx(float(p["max"]), rel=1e-5)
        if "sink_formats" in spec:
            assert e.formats(False) == spec["sink_formats"] and e.formats(True) == spec["src_formats"]
        e.close()


def test_hsvfilter_element(oracle, synth):
    from mi355fx.elements import Element, FLOW_OK, FLOW_NOT_NEGOTIATED
    e = Element("hsvfilter")
    assert e.get_property("saturation-mul") == 1.0 and e.get_property("hue-shift") == 0.0
    assert e.set_property("hue-shift", 90.0) and e.set_property("value-off", 0.02)
    assert not e.set_property("hue-shift", float("nan"))      # GLib rejects NaN / inf / out of range
    assert not e.set_property("hue-shift", float("inf"))
    assert not e.set_property("no-such", 1.0)
    assert e.get_property("hue-shift") == 90.0
    w, h = 640, 360
    frame = synth.smooth_frame(w, h).reshape(-1).copy()
    exp = frame.copy()
    oracle.hsvfilter(exp, w, w * 4, 4, 0, True, (90.0, 1.0, 0.0, 1.0, np.float32(0.02)))
    assert e.transform_frame_ip("BGRx", w, h, w * 4, frame) == FLOW_OK
    assert (frame == exp).all()
    # properties are mutable in PLAYING: the next buffer sees the new snapshot
    assert e.set_property("hue-shift", -45.5)
    exp2 = exp.copy()
    oracle.hsvfilter(exp2, w, w * 4, 4, 0, True, (-45.5, 1.0, 0.0, 1.0, np.float32(0.02)))
    assert e.transform_frame_ip("BGRx", w, h, w * 4, frame) == FLOW_OK
    assert (frame == exp2).all()
    assert e.transform_frame_ip("RGBA64_LE", w, h, w * 8, frame) == FLOW_NOT_NEGOTIATED
    e.close()


def test_colorlut_element_lifecycle(tmp_path, oracle, synth):
    from mi355fx.elements import Element, FLOW_OK, FLOW_ERROR
    e = Element("colorlut")
    w, h = 320, 200
    src = synth.noise_frame(w, h)
    dst = np.zeros_like(src)
    # start() without a location: ResourceError::Settings (colorlut/imp.rs:175-180)
    assert not e.start() and "not configured" in e.last_error
    # transform without a LUT: FlowError::Error (colorlut/imp.rs:209-212)
    assert e.transform_frame("RGBA", w, h, w * 4, src, "RGBA", w * 4, dst) == FLOW_ERROR
    bad = tmp_path / "bad.cube"
    bad.write_text("LUT_3D_SIZE 2\n0 0 0\n")
    assert e.set_property("location", str(bad))
    assert not e.start() and "Failed to parse LUT file" in e.last_error
    good = tmp_path / "good.cube"
    text = synth.cube_text_3d(33)
    good.write_text(text)
    assert e.set_property("location", str(good)) and e.start()
    assert not e.set_property("location", str(bad))           # mutable only in READY
    assert e.transform_frame("RGBA", w, h, w * 4, src, "RGBA", w * 4, dst) == FLOW_OK
    exp = np.zeros_like(src)
    oracle.colorlut_rgba8(oracle.Cube.parse(text), src, w * 4, exp, w * 4, w, h)
    assert (dst == exp).all()
    assert e.stop()
    assert e.transform_frame("RGBA", w, h, w * 4, src, "RGBA", w * 4, dst) == FLOW_ERROR   # LUT dropped by stop()
    e.close()


def test_hsvdetector_element(oracle, synth):
    from mi355fx.elements import Element, FLOW_OK
    from mi355fx import FMT_LAYOUT
    e = Element("hsvdetector")
    assert e.set_property("hue-ref", 120.0) and e.set_property("hue-var", 40.0)
    assert e.set_property("saturation-ref", 0.8) and e.set_property("saturation-var", 0.5)
    assert e.set_property("value-ref", 0.7) and e.set_property("value-var", 0.6)
    assert not e.set_property("hue-var", 181.0) and not e.set_property("value-ref", -0.1)   # bounded ranges
    st = (120.0, 40.0, np.float32(0.8), 0.5, np.float32(0.7), np.float32(0.6))
    rng = np.random.default_rng(17)
    # h is a multiple of 3: the reference asserts plane_len % pixel_stride == 0 (hsvdetector/imp.rs:124),
    # so a padded 3-byte-pixel plane whose length is not a multiple of 3 is a panic there (an error here)
    w, h = 97, 12
    for in_fmt in ("RGBx", "xRGB", "BGRx", "xBGR", "RGB", "BGR"):
        ps, first, bgr = FMT_LAYOUT[in_fmt]
        ss = (w * ps + 3) & ~3
        src = rng.integers(0, 256, size=h * ss, dtype=np.uint8)
        for out_fmt, (af, obgr) in {"RGBA": (0, 0), "ARGB": (1, 0), "BGRA": (0, 1), "ABGR": (1, 1)}.items():
            ds = w * 4 + 4
            dst = np.full(h * ds, 0x3C, np.uint8)
            exp = dst.copy()
            oracle.hsvdetect(src, ss, ps, first, bool(bgr), exp, ds, bool(af), bool(obgr), w, st)
            assert e.transform_frame(in_fmt, w, h, ss, src, out_fmt, ds, dst) == FLOW_OK
            assert (dst == exp).all(), (in_fmt, out_fmt)
    e.close()


def test_rsaudioecho_element(oracle, synth):
    from mi355fx.elements import Element, FLOW_OK, FLOW_NOT_NEGOTIATED
    e = Element("rsaudioecho")
    assert e.get_property("delay") == 500 * 10 ** 9 and e.get_property("max-delay") == 10 ** 9   # audioecho/imp.rs:31-34
    x = synth.sine_stereo_f32(48000)
    assert e.audio_transform_ip(x.copy()) == FLOW_NOT_NEGOTIATED                                   # before setup
    assert e.set_property("max-delay", 2 * 10 ** 9) and e.set_property("delay", 300 * 10 ** 6)
    assert e.set_property("intensity", 0.6) and e.set_property("feedback", 0.4)
    assert not e.set_property("feedback", 1.5)                                                     # range 0..1
    assert e.audio_setup(48000, 2, f64=False)
    assert e.set_property("max-delay", 5 * 10 ** 9)            # accepted but ignored once there is state
    assert e.get_property("max-delay") == 2 * 10 ** 9
    ref = oracle.Echo(2 * 10 ** 9, 48000, 2)
    for chunk in np.array_split(x, 5):
        got, exp = chunk.copy(), chunk.copy()
        ref.process(exp, 300 * 10 ** 6, 0.6, 0.4)
        assert e.audio_transform_ip(got) == FLOW_OK
        assert got.tobytes() == exp.tobytes()
    assert e.stop()
    assert e.audio_transform_ip(x.copy()) == FLOW_NOT_NEGOTIATED                                   # state dropped
    e.close()


def test_unknown_factory():
    from mi355fx.elements import Element, ElementError
    with pytest.raises(ElementError):
        Element("agingradio")


@pytest.mark.parametrize("dtype", [np.int16, np.int32, np.float32, np.float64])
@pytest.mark.parametrize("planar", [False, True])
def test_ebur128level_element_like_the_reference_test(oracle, dtype, planar):
    """Replays audio/audiofx/tests/ebur128level.rs:93-153 on the element mirror: 5 one-second buffers of a
    440 Hz test tone (audiotestsrc default: sine, volume 0.8), 2 channels 48 kHz, interval=500 ms ->
    exactly 10 `ebur128-level` messages with timestamps k*500 ms carrying every field; in addition the
    values must equal the oracle's readings at the same instants."""
    from mi355fx.elements import Element, FLOW_OK
    rate, ch = 48000, 2
    e = Element("ebur128level")
    assert e.set_property("interval", 500_000_000)
    assert e.get_property("interval") == 500_000_000 and e.get_property("mode") == 63 and e.get_property("post-messages") == 1.0
    assert e.ebur128_setup(rate, ch, dtype, planar)
    ref = oracle.EbuR128(ch, rate, 63, [1, 1])   # no channel positions -> Center weighting for all (imp.rs:589-595)
    t = np.arange(5 * rate) / rate
    tone = 0.8 * np.sin(2 * np.pi * 440.0 * t)
    if dtype == np.int16:
        x = np.round(tone * 32767).astype(np.int16)
    elif dtype == np.int32:
        x = np.round(tone * 2147483647).astype(np.int32)
    else:
        x = tone.astype(dtype)
    x = np.repeat(x[:, None], ch, axis=1)
    msgs, expected = [], []
    for b in range(5):
        buf = x[b * rate:(b + 1) * rate]
        data = np.ascontiguousarray(buf.T) if planar else buf.reshape(-1)
        assert e.ebur128_push(data, b * 10 ** 9, planar) == FLOW_OK
        msgs += e.ebur128_pop_messages()
        for half in range(2):
            part = buf[half * rate // 2:(half + 1) * rate // 2]
            ref.add_frames(np.ascontiguousarray(part.T) if planar else part.reshape(-1), planar=planar)
            expected.append((ref.loudness_momentary(), ref.loudness_shortterm(), ref.loudness_global(), ref.relative_threshold(),
                             ref.loudness_range(), [ref.sample_peak(c) for c in range(ch)], [ref.true_peak(c) for c in range(ch)]))
    assert len(msgs) == 10
    for k, (m, ex) in enumerate(zip(msgs, expected), start=1):
        assert m["timestamp"] == k * 500_000_000
        for field in ("momentary-loudness", "shortterm-loudness", "global-loudness", "relative-threshold", "loudness-range"):
            assert isinstance(m[field], float)
        assert len(m["sample-peak"]) == 2 and len(m["true-peak"]) == 2
        got = (m["momentary-loudness"], m["shortterm-loudness"], m["global-loudness"], m["relative-threshold"], m["loudness-range"])
        for g, w in zip(got, ex[:5]):
            assert (g == w) or abs(g - w) <= 1e-9
        assert m["sample-peak"] == ex[5] and m["true-peak"] == ex[6]
    # post-messages=false is honoured per buffer (mutable in PLAYING); the `reset` action restarts the interval
    assert e.set_property("post-messages", False)
    assert e.ebur128_push(x[:rate].reshape(-1) if not planar else np.ascontiguousarray(x[:rate].T), 5 * 10 ** 9, planar) == FLOW_OK
    assert e.ebur128_pop_messages() == []
    assert e.stop()
    from mi355fx.elements import FLOW_NOT_NEGOTIATED
    assert e.ebur128_push(x[:10].reshape(-1) if not planar else np.ascontiguousarray(x[:10].T), 0, planar) == FLOW_NOT_NEGOTIATED
    e.close()
