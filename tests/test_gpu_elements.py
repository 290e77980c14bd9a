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


# ------------------------------------------------------------------ hrtfrender (audio/hrtf/tests/hrtfrender.rs)

HRIR_FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "test.hrir")
FLOW_OK, FLOW_NOT_NEGOTIATED = 0, -4


def _hrtf_element(channels, positions=None, objects=None, rate=44100, hrir=None):
    from mi355fx.elements import Element
    e = Element("hrtfrender")
    assert e.hrtf_set_hrir_raw(open(HRIR_FIXTURE, "rb").read() if hrir is None else hrir)
    if objects is not None:
        assert e.hrtf_set_spatial_objects(objects)
    return e, e.hrtf_set_caps(rate, channels, positions)


def test_hrtfrender_samples_in_samples_out():
    """test_hrtfrender_samples_in_samples_out: one full block + 20 frames in -> one full block out, the 20-frame
    residue comes out at EOS (drain), zero-padded internally and truncated to 20 frames."""
    e, ok = _hrtf_element(1, positions=[2])  # channel-mask 0x1... FRONT_LEFT in the reference; any positioned mono works
    assert ok, e.last_error
    full_block = 512 * 8
    buf = np.zeros(full_block + 20, np.float32)
    assert e.hrtf_transform_size(buf.nbytes) == full_block * 8
    flow, out = e.hrtf_transform(buf)
    assert flow == FLOW_OK and out.nbytes == full_block * 8
    flow, out = e.hrtf_drain()
    assert flow == FLOW_OK and out.nbytes == 20 * 8
    flow, out = e.hrtf_drain()          # adapter empty: nothing more
    assert flow == FLOW_OK and out.size == 0


def test_hrtfrender_implicit_spatial_objects():
    """8 channels, channel-mask 0xc3f (FL FR FC LFE RL RR SL SR): objects are inferred from the positions."""
    e, ok = _hrtf_element(8, positions=[0, 1, 2, 3, 4, 5, 10, 11])
    assert ok, e.last_error
    objs = e.hrtf_spatial_objects()
    assert len(objs) == 8
    assert (objs[0]["x"], objs[0]["y"], objs[0]["z"]) == (pytest.approx(-1.45), 0.0, 2.5)   # FrontLeft (spatial.rs:184)
    assert (objs[3]["x"], objs[3]["y"], objs[3]["z"]) == (0.0, 0.0, 0.0)                    # Lfe1
    assert objs[6]["x"] == -2.5 and objs[6]["z"] == pytest.approx(-0.44)                    # SideLeft
    assert all(o["coordinate-system"] == "left-handed" and o["distance-gain"] == 1.0 for o in objs)


def test_hrtfrender_explicit_spatial_objects():
    objs = [{"x": -1.0 + x / 8.0, "y": 0.0, "z": 1.0, "distance-gain": 0.1} for x in range(8)]
    e, ok = _hrtf_element(8, objects=objs)   # unpositioned caps are fine once objects are set
    assert ok, e.last_error
    got = e.hrtf_spatial_objects()
    assert len(got) == 8 and got[7]["x"] == pytest.approx(-0.125) and got[0]["distance-gain"] == pytest.approx(0.1)


def test_hrtfrender_caps_negotiation_fail():
    """6 input channels but 2 spatial objects: set_caps fails ("Wrong number of spatial objects") and pushing a
    buffer returns NotNegotiated."""
    objs = [{"x": 0.0, "y": 0.0, "z": 1.0, "distance-gain": 0.1} for _ in range(2)]
    e, ok = _hrtf_element(6, objects=objs)
    assert not ok and "Wrong number of spatial objects" in e.last_error
    flow, out = e.hrtf_transform(np.zeros(512, np.float32))
    assert flow == FLOW_NOT_NEGOTIATED and out.size == 0


def test_hrtfrender_errors_like_the_reference():
    from mi355fx.elements import Element
    e = Element("hrtfrender")
    assert not e.hrtf_set_caps(44100, 1, [2]) and "Impulse response not set" in e.last_error
    e, ok = _hrtf_element(2, positions=None)
    assert not ok and "Cannot infer object positions" in e.last_error
    e, ok = _hrtf_element(1, positions=[-1])                      # GST_AUDIO_CHANNEL_POSITION_INVALID
    assert not ok and "Unsupported channel position" in e.last_error
    e, ok = _hrtf_element(1, positions=[2], rate=48000)           # sphere is 44.1 kHz: resampled at load (HrirSphere::new(bytes, rate))
    assert ok, e.last_error
    # once negotiated, a spatial-objects update with the wrong count is ignored with a warning (imp.rs:440-451)
    e, ok = _hrtf_element(2, positions=[0, 1])
    assert ok
    assert not e.hrtf_set_spatial_objects([{"x": 0.0, "y": 0.0, "z": 1.0}])
    assert len(e.hrtf_spatial_objects()) == 2
    assert e.properties()["interpolation-steps"]["default"] == 8 and e.properties()["block-length"]["default"] == 512
    assert e.get_property("use-rayon") == 0.0


def test_hrtfrender_element_stream_matches_oracle(oracle, synth):
    """Arbitrary buffer sizes through the adapter, moving objects between buffers, EOS drain: the element's output
    stream equals the oracle fed block by block (tolerance as tests/test_gpu_hrtf.py)."""
    mesh = open(HRIR_FIXTURE, "rb").read()
    data = synth.hrir_sphere_bytes(mesh, 64)
    C, steps, bl = 3, 4, 128
    from mi355fx.elements import Element
    e = Element("hrtfrender")
    e.hrtf_set_hrir_raw(data)
    assert e.set_property("interpolation-steps", steps) and e.set_property("block-length", bl)
    objs = [{"x": 1.0, "y": 0.2, "z": 0.5, "coordinate-system": "cartesian"}, {"x": -0.3, "y": 0.1, "z": 1.0, "distance-gain": 0.7},
            {"x": 0.4, "y": -0.9, "z": -0.2, "coordinate-system": "right-handed", "distance-gain": 0.5}]
    assert e.hrtf_set_spatial_objects(objs)
    assert e.hrtf_set_caps(44100, C), e.last_error
    sphere = oracle.HrirSphere(data, 44100)
    ref = oracle.HrtfRender(sphere, C, steps, bl)

    def rh(o):  # Position::to_right_handed (spatial.rs:58-66)
        cs = o.get("coordinate-system", "left-handed")
        x, y, z = o["x"], o["y"], o["z"]
        return {"cartesian": (-y, z, -x), "left-handed": (x, y, -z), "right-handed": (x, y, z)}[cs]

    rng = np.random.default_rng(12)
    blk = steps * bl
    total = 3 * blk + 57
    x = rng.uniform(-1, 1, (total, C)).astype(np.float32)
    got = []
    fed = 0
    for n in (100, blk, 2 * blk - 100 + 30, total - (100 + blk + 2 * blk - 100 + 30)):
        flow, out = e.hrtf_transform(x[fed: fed + n])
        assert flow == FLOW_OK
        got.append(out)
        fed += n
    flow, out = e.hrtf_drain()
    assert flow == FLOW_OK and out.size == 57 * 2
    got.append(out)
    got = np.concatenate(got)
    pos = np.array([rh(o) for o in objs], np.float32)
    gains = np.array([o.get("distance-gain", 1.0) for o in objs], np.float32)
    xp = np.concatenate([x, np.zeros((4 * blk - total, C), np.float32)])
    exp = np.concatenate([ref.process_block(xp[k * blk: (k + 1) * blk], pos, gains) for k in range(4)])[: total * 2]
    assert got.size == exp.size
    assert np.abs(got - exp).max() <= 4e-5 * max(1.0, np.abs(exp).max())


# ------------------------------------------------------------------ videocompare (video/videofx/tests/videocompare.rs)

def _solid_rgba(w, h, rgb):
    f = np.zeros((h, w * 4), np.uint8)
    f[:, 0::4], f[:, 1::4], f[:, 2::4], f[:, 3::4] = rgb[0], rgb[1], rgb[2], 255
    return f


def test_videocompare_can_find_similar_frames():
    """pattern=red on both pads, max-dist-threshold 0, Blockhash: a message is posted and sink_1's distance is <= 0."""
    from mi355fx.elements import Element
    e = Element("videocompare")
    assert e.set_property("max-dist-threshold", 0.0) and e.set_property("hash-algo", "blockhash")
    red = _solid_rgba(320, 240, (255, 0, 0))
    flow, posted, dist = e.videocompare_aggregate([red, red.copy()], "RGBA", 320, 240, 1280)
    assert flow == 0 and posted and dist == [0.0]


def test_videocompare_do_not_send_message_when_image_not_found():
    """reference = snow, secondary = red: no message at threshold 0."""
    from mi355fx.elements import Element
    e = Element("videocompare")
    snow = np.random.default_rng(3).integers(0, 256, (240, 320), dtype=np.uint8).repeat(4, axis=1)
    snow[:, 3::4] = 255
    flow, posted, dist = e.videocompare_aggregate([snow, _solid_rgba(320, 240, (255, 0, 0))], "RGBA", 320, 240, 1280)
    assert flow == 0 and not posted
    # with a generous threshold the same pair does post, carrying the real distance
    assert e.set_property("max-dist-threshold", 64.0)
    flow, posted, dist = e.videocompare_aggregate([snow, _solid_rgba(320, 240, (255, 0, 0))], "RGBA", 320, 240, 1280)
    assert posted and 0 < dist[0] <= 64


def test_videocompare_several_pads_and_unsupported_algo(oracle):
    from mi355fx.elements import Element
    e = Element("videocompare")
    assert e.set_property("max-dist-threshold", 3.0)
    rng = np.random.default_rng(8)
    ref = rng.integers(0, 256, (64, 96 * 4), dtype=np.uint8)
    near = ref.copy(); near[:8, :48] ^= 0x80
    far = rng.integers(0, 256, (64, 96 * 4), dtype=np.uint8)
    flow, posted, dist = e.videocompare_aggregate([ref, far, near, ref], "RGBA", 96, 64, 384)
    assert flow == 0 and posted and len(dist) == 3 and dist[2] == 0.0
    exp = [oracle.hash_distance(oracle.blockhash(ref, 96, 64, 384, 4), oracle.blockhash(x, 96, 64, 384, 4)) for x in (far, near, ref)]
    assert dist == exp
    assert e.get_property("max-dist-threshold") == 3.0
    assert e.set_property("hash-algo", "gradient") and e.set_property("max-dist-threshold", 0.0)
    flow, posted, dist = e.videocompare_aggregate([ref, ref.copy()], "RGBA", 96, 64, 384)
    assert flow == 0 and posted and dist == [0.0]
    # test_use_dssim_to_find_similar_frames: Dssim engine, identical frames, threshold 0
    assert e.set_property("hash-algo", "dssim") and e.set_property("max-dist-threshold", 0.0)
    oref, ofar = ref.copy(), far.copy()
    oref[:, 3::4] = 255; ofar[:, 3::4] = 255
    flow, posted, dist = e.videocompare_aggregate([oref, oref.copy(), ofar], "RGBA", 96, 64, 384)
    assert flow == 0 and posted and dist[0] == 0.0 and dist[1] > 0.1
    # translucent pixels are compared too (create_image_rgba whatever the alpha, hashed_image.rs:54-55): no flow error
    flow, posted, dist = e.videocompare_aggregate([ref, ref.copy(), far], "RGBA", 96, 64, 384)
    assert flow == 0 and posted and dist[0] == 0.0 and dist[1] > 0.0
    assert not e.set_property("hash-algo", "nonsense")


# ------------------------------------------------------------------ audioloudnorm

def test_audioloudnorm_element(oracle):
    """Properties with the reference's ranges, caps restricted to 192 kHz, chain/drain through the adapter == oracle."""
    from mi355fx.elements import Element
    e = Element("audioloudnorm")
    assert e.type_name == "GstAudioLoudNorm" and e.klass == "Filter/Effect/Audio"
    p = e.properties()
    assert (p["loudness-target"]["default"], p["loudness-target"]["min"], p["loudness-target"]["max"]) == (-24.0, -70.0, -5.0)
    assert (p["max-true-peak"]["default"], p["max-true-peak"]["min"], p["max-true-peak"]["max"]) == (-2.0, -9.0, 0.0)
    assert not e.set_property("loudness-target", -3.0)               # out of range: rejected, unchanged
    assert e.set_property("loudness-target", -18.0) and e.set_property("max-true-peak", -1.5)
    flow, out = e.loudnorm_chain(np.zeros(100))
    assert flow == -4 and out.size == 0                               # NotNegotiated before caps
    assert not e.loudnorm_set_caps(48000, 2)                          # only 192 kHz is accepted
    assert e.loudnorm_set_caps(192000, 2), e.last_error
    rate = 192000
    t = np.arange(int(3.7 * rate)) / rate
    x = np.stack([0.03 * np.sin(2 * np.pi * 330 * t), 0.03 * np.sin(2 * np.pi * 331 * t)], 1)
    x[int(3.2 * rate): int(3.2 * rate) + 500] *= 50
    ln = oracle.LoudNorm(2, loudness_target=-18.0, max_true_peak=-1.5)
    got, exp = [], []
    for k in range(0, len(x), 100000):
        flow, o = e.loudnorm_chain(x[k:k + 100000])
        assert flow == 0
        got.append(o); exp.append(ln.push(x[k:k + 100000]))
    flow, o = e.loudnorm_drain()
    assert flow == 0
    got.append(o); exp.append(ln.drain())
    got, exp = np.concatenate(got), np.concatenate(exp)
    assert got.size == x.size == exp.size
    assert np.abs(got - exp).max() <= 1e-9 * np.abs(exp).max()
    assert np.abs(got).max() <= 10 ** (-1.5 / 20)
    assert e.stop()
    flow, o = e.loudnorm_drain()
    assert flow == 0 and o.size == 0


# ------------------------------------------------------------------ roundedcorners (host only, cairo-rendered mask)

def test_roundedcorners_element_masks_match_cairo_goldens():
    """The alpha plane the element attaches equals the committed cairo renderings byte for byte (CRC of the whole plane +
    the corner bytes), radius 0 gives an opaque plane, the mask is regenerated when border-radius-px changes."""
    import zlib
    from mi355fx.elements import Element
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cairo_masks.json")))
    e = Element("roundedcorners")
    assert e.type_name == "GstRoundedCorners" and e.klass == "Filter/Effect/Converter/Video"
    assert e.roundedcorners_src_formats() == ["I420", "A420"]       # radius 0: both offered (imp.rs:408-416)
    for key, g in gold["cases"].items():
        dims, r = key.split("_r")
        w, h = (int(v) for v in dims.split("x"))
        assert e.set_property("border-radius-px", int(r))
        assert e.roundedcorners_src_formats() == (["I420", "A420"] if int(r) == 0 else ["A420"])
        assert e.roundedcorners_set_caps(w, h, True)
        flow, passthrough, alpha = e.roundedcorners_prepare()
        assert flow == 0 and not passthrough
        assert list(alpha.shape) == g["shape"]
        assert alpha[:12, :12].tolist() == g["corner"], key
        assert zlib.crc32(alpha.tobytes()) == g["crc32"], key
    # property change while negotiated: next buffer carries the new mask
    assert e.roundedcorners_set_caps(40, 24, True) and e.set_property("border-radius-px", 8)
    _, _, a8 = e.roundedcorners_prepare()
    assert e.set_property("border-radius-px", 0)
    _, _, a0 = e.roundedcorners_prepare()
    assert (a0 == 255).all() and (a8 != a0).any() and a8[0, 0] == 0 and a8[12, 20] == 255
    # I420 output caps: passthrough, no alpha plane
    assert e.roundedcorners_set_caps(40, 24, False)
    flow, passthrough, alpha = e.roundedcorners_prepare()
    assert flow == 0 and passthrough and alpha is None
