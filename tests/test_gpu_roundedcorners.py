"""roundedcorners on device-resident frames (csrc/roundedcorners.hip). The reference renders ONE A8 plane
per caps / radius change with cairo (video/videofx/src/border/imp.rs:57-180) and appends that shared memory to every buffer as
plane 3 of A420 (imp.rs:482-559; stride[3] x round_up_2(height) bytes, :469-470). The host mirror renders the plane the same way
(through the system libcairo; tests/golden/cairo_masks.json are cairo's own renderings); these tests check what the device adds:
the plane kept in HBM equals the golden, and the append launch writes exactly that behind the I420 planes of every frame of a
batch and nothing else."""
import json
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cairo_masks.json")))


def _i420_size(w, h):
    """GStreamer's I420 layout: strides round_up_4(w), round_up_4(round_up_2(w) / 2); heights h, round_up_2(h) / 2 (x 2)."""
    s0, s1 = (w + 3) & ~3, ((((w + 1) & ~1) // 2) + 3) & ~3
    return s0 * ((h + 1) & ~1) + 2 * s1 * (((h + 1) & ~1) // 2)


@pytest.mark.parametrize("key", sorted(GOLD["cases"]))
def test_device_plane_and_append_match_the_cairo_goldens(ctx, key):
    from mi355fx.elements import Element
    g = GOLD["cases"][key]
    dims, r = key.split("_r")
    w, h = (int(v) for v in dims.split("x"))
    e = Element("roundedcorners")
    assert e.set_property("border-radius-px", int(r)) and e.roundedcorners_set_caps(w, h, True)
    flow, passthrough, alpha = e.roundedcorners_prepare()      # the host-rendered plane (cairo)
    assert flow == 0 and not passthrough and list(alpha.shape) == g["shape"] and zlib.crc32(alpha.tobytes()) == g["crc32"]
    rows, stride = alpha.shape
    assert rows == (h + 1) & ~1 and stride == (w + 3) & ~3      # round_up_2(height) rows (imp.rs:469), A420 plane-3 stride
    ctx.roundedcorners_set_mask(alpha, w, h, stride)
    d_mask, n, st = ctx.roundedcorners_mask_device()
    assert n == alpha.size and st == stride
    back = np.zeros(n, np.uint8)
    ctx.d2h(back, d_mask)
    assert zlib.crc32(back.tobytes()) == g["crc32"] and back.reshape(alpha.shape)[:12, :12].tolist() == g["corner"]
    if int(r) == 0:
        assert (back == 255).all()                                # radius 0: opaque
    # a batch of three A420 frames: I420 payload (random), then the alpha plane; odd pitch on purpose for the unaligned paths
    n_frames, off = 3, _i420_size(w, h)
    for pad in (0, 3, 16):
        pitch = off + n + pad
        rng = np.random.default_rng(w + h + pad)
        frames = rng.integers(0, 256, size=n_frames * pitch, dtype=np.uint8)
        d = ctx.alloc(frames.size)
        try:
            ctx.h2d(d, frames)
            ctx.roundedcorners_append_device(d, pitch, off, n_frames)
            ctx.synchronize()
            got = np.zeros_like(frames)
            ctx.d2h(got, d)
        finally:
            ctx.free(d)
        exp = frames.copy()
        for f in range(n_frames):
            exp[f * pitch + off: f * pitch + off + n] = alpha.reshape(-1)
        assert (got == exp).all(), (key, pad)                     # plane 3 of every frame == the mask; Y, U, V and the padding untouched
    e.close() if hasattr(e, "close") else None


def test_mask_changes_and_passthrough(ctx):
    """A new radius replaces the plane in stream order; I420 output (passthrough) drops it; append without a plane is NOT_CONFIGURED."""
    import mi355fx
    from mi355fx.elements import Element
    e = Element("roundedcorners")
    assert e.roundedcorners_set_caps(40, 24, True) and e.set_property("border-radius-px", 8)
    _, _, a8 = e.roundedcorners_prepare()
    ctx.roundedcorners_set_mask(a8, 40, 24, a8.shape[1])
    d = ctx.alloc(2 * a8.size)
    try:
        ctx.roundedcorners_append_device(d, a8.size, 0, 2)
        assert e.set_property("border-radius-px", 0)
        _, _, a0 = e.roundedcorners_prepare()
        ctx.roundedcorners_set_mask(a0, 40, 24, a0.shape[1])
        out = np.zeros(2 * a8.size, np.uint8)
        ctx.d2h(out, d)
        assert (out.reshape(2, -1) == a8.reshape(-1)).all()       # the first append saw the first mask
        ctx.roundedcorners_append_device(d, a8.size, 0, 2)
        ctx.d2h(out, d)
        assert (out == 255).all()
        ctx.roundedcorners_set_mask(None, 0, 0, 0)
        with pytest.raises(mi355fx.Mi355Error) as err:
            ctx.roundedcorners_append_device(d, a8.size, 0, 2)
        assert err.value.status == mi355fx.ERR_NOT_CONFIGURED
        with pytest.raises(mi355fx.Mi355Error):
            ctx.roundedcorners_mask_device()
    finally:
        ctx.free(d)


def test_element_prepare_output_buffer_device(ctx):
    """The element's device form of prepare_output_buffer: A420 batch in HBM in, plane 3 == the element's (cairo) plane out;
    a radius change while negotiated shows in the next batch; I420 output caps: passthrough, nothing written."""
    from mi355fx.elements import Element
    w, h, n_frames = 64, 48, 2
    e = Element("roundedcorners")
    assert e.set_property("border-radius-px", 10) and e.roundedcorners_set_caps(w, h, True)
    off = _i420_size(w, h)
    _, _, alpha = e.roundedcorners_prepare()
    pitch = off + alpha.size
    frames = np.random.default_rng(5).integers(0, 256, size=n_frames * pitch, dtype=np.uint8)
    d = ctx.alloc(frames.size)
    try:
        ctx.h2d(d, frames)
        assert e.roundedcorners_prepare_device(d, pitch, off, n_frames) == 0
        got = np.zeros_like(frames)
        ctx.synchronize()
        import mi355fx
        mi355fx.Context  # (the element launches on its own context's stream: wait for the device, not for ours)
        ctx.d2h(got, d)
        exp = frames.copy()
        for f in range(n_frames):
            exp[f * pitch + off:(f + 1) * pitch] = alpha.reshape(-1)
        assert zlib.crc32(alpha.tobytes()) == GOLD["cases"]["64x48_r10"]["crc32"]
        # (stream ordering between the element's context and ours is the caller's business: poll until the launch has landed)
        for _ in range(200):
            if (got == exp).all():
                break
            ctx.d2h(got, d)
        assert (got == exp).all()
        assert e.set_property("border-radius-px", 0)
        assert e.roundedcorners_prepare_device(d, pitch, off, n_frames) == 0
        for _ in range(200):
            ctx.d2h(got, d)
            if (got.reshape(n_frames, -1)[:, off:] == 255).all():
                break
        assert (got.reshape(n_frames, -1)[:, off:] == 255).all() and (got.reshape(n_frames, -1)[:, :off] == frames.reshape(n_frames, -1)[:, :off]).all()
        assert e.roundedcorners_set_caps(w, h, False)
        ctx.h2d(d, frames)
        assert e.roundedcorners_prepare_device(d, pitch, off, n_frames) == 0
        ctx.d2h(got, d)
        assert (got == frames).all()
    finally:
        ctx.free(d)
