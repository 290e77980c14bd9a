"""GPU tests for videocompare's Dssim engine through the C ABI vs the numpy restatement (parity unpinned w.r.t. the
crate). The per-pixel f32 pipeline is written in the same operation order on both sides; the scores involve f64
reductions whose order differs: tolerance 1e-9 relative (values are O(1e-4 .. 1))."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frame(rng, w, h, block=8):
    base = np.kron(rng.integers(0, 256, (h // block, w // block, 4), dtype=np.uint8), np.ones((block, block, 1), np.uint8)).reshape(h, w * 4)
    base[:, 3::4] = 255
    return base


def _noisy(rng, base, amp):
    n = np.clip(base.astype(int) + rng.integers(-amp, amp + 1, base.shape), 0, 255).astype(np.uint8)
    n[:, 3::4] = 255
    return n


@pytest.mark.parametrize("w,h", [(128, 96), (322, 246), (1920, 1080)])
def test_dssim_matches_restatement(ctx, w, h):
    from oracle import dssim_restate as D
    rng = np.random.default_rng(w + h)
    base = _frame(rng, w - w % 8, h - h % 8)
    base = np.pad(base, ((0, h % 8), (0, (w % 8) * 4)), mode="edge") if (w % 8 or h % 8) else base
    base[:, 3::4] = 255
    ga = ctx.dssim_create_image(base, w * 4, w, h)
    oa = D.DssimImage(base, w, h, w * 4, 4)
    try:
        assert ctx.dssim_compare(ga, ga) == 0.0
        for amp in (3, 20, 90):
            mod = _noisy(rng, base, amp)
            gb = ctx.dssim_create_image(mod, w * 4, w, h)
            got = ctx.dssim_compare(ga, gb)
            ctx.dssim_free_image(gb)
            exp = D.compare(oa, D.DssimImage(mod, w, h, w * 4, 4))
            assert got == pytest.approx(exp, rel=1e-9, abs=1e-13), (amp, got, exp)
    finally:
        ctx.dssim_free_image(ga)


def test_dssim_identical_4k_frames_exactly_zero_and_rgb(ctx):
    """Reference test test_use_dssim_to_find_similar_frames: red vs red at threshold 0 -> distance <= 0.0."""
    from oracle import dssim_restate as D
    w, h = 3840, 2160
    red = np.zeros((h, w * 4), np.uint8); red[:, 0::4] = 255; red[:, 3::4] = 255
    a, b = ctx.dssim_create_image(red, w * 4, w, h), ctx.dssim_create_image(red.copy(), w * 4, w, h)
    try:
        assert ctx.dssim_compare(a, b) == 0.0
    finally:
        ctx.dssim_free_image(a); ctx.dssim_free_image(b)
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (72, 64 * 3), dtype=np.uint8)
    pad = np.zeros((72, 64 * 3 + 5), np.uint8); pad[:, : 192] = rgb
    other = rng.integers(0, 256, (72, 64 * 3), dtype=np.uint8)
    ga, gb = ctx.dssim_create_image(pad, 197, 64, 72, "RGB"), ctx.dssim_create_image(other, 192, 64, 72, "RGB")
    try:
        exp = D.compare(D.DssimImage(rgb, 64, 72, 192, 3), D.DssimImage(other, 64, 72, 192, 3))
        assert ctx.dssim_compare(ga, gb) == pytest.approx(exp, rel=1e-9)
    finally:
        ctx.dssim_free_image(ga); ctx.dssim_free_image(gb)


def test_dssim_errors(ctx):
    import mi355fx
    f = np.zeros((16, 64), np.uint8); f[:, 3::4] = 255
    g = np.zeros((32, 128), np.uint8); g[:, 3::4] = 255
    with pytest.raises(mi355fx.Mi355Error):
        ctx.dssim_create_image(f, 64, 16, 16, "BGRx")
    a, b = ctx.dssim_create_image(f, 64, 16, 16), ctx.dssim_create_image(g, 128, 32, 32)
    try:
        with pytest.raises(mi355fx.Mi355Error):
            ctx.dssim_compare(a, b)
    finally:
        ctx.dssim_free_image(a); ctx.dssim_free_image(b)


@pytest.mark.parametrize("pattern", [True, False])
@pytest.mark.parametrize("w,h,fmt", [(128, 96, "RGBA"), (322, 246, "RGBA"), (100, 50, "RGB"), (37, 19, "RGBA")])
def test_dssim_image_planes_bit_identical(ctx, w, h, fmt, pattern):
    """Every plane of every scale (LAB image, mu, img_sq_blur) equals the numpy restatement bit for bit: pins the fused
    per-scale kernel (tiles, halos, per-pass edge replication) incl. sizes that are not multiples of the tile, and the
    composition of translucent pixels - over dssim's coloured pattern (default) or over black (MI355_FLAG_DSSIM_TRANSLUCENT)."""
    from oracle import dssim_restate as D
    rng = np.random.default_rng(w * h)
    ch = 4 if fmt == "RGBA" else 3
    f = rng.integers(0, 256, (h, w * ch), dtype=np.uint8)
    if ch == 4:
        f[:, 3::4] = np.where(rng.random((h, w)) < 0.2, rng.integers(0, 256, (h, w)), 255)   # some translucent pixels
    import mi355fx
    ctx.set_flag(mi355fx.FLAG_DSSIM_TRANSLUCENT, 0 if pattern else 1)
    g = ctx.dssim_create_image(f, w * ch, w, h, fmt)
    o = D.DssimImage(f, w, h, w * ch, ch, pattern=pattern)
    try:
        for s, chans in enumerate(o.scales):
            for c in range(3):
                for kind in ("img", "mu", "sq"):
                    got = ctx.dssim_image_plane(g, s, c, kind)
                    exp = chans[c][kind]
                    assert got.shape == exp.shape
                    assert (got.view(np.uint32) == exp.view(np.uint32)).all(), (s, c, kind, int((got != exp).sum()))
    finally:
        ctx.dssim_free_image(g)
        ctx.set_flag(mi355fx.FLAG_DSSIM_TRANSLUCENT, 0)


def test_dssim_translucent_frames_are_compared(ctx):
    """hashed_image.rs:54-55 hands RGBA frames to create_image_rgba whatever their alpha: a frame with translucent pixels gets a
    value (round 2 refused it), the value agrees with the restatement, a translucent pixel moves it away from the opaque
    frame's, and an all-opaque RGBA frame equals the RGB one."""
    from oracle import dssim_restate as D
    w, h = 200, 120
    rng = np.random.default_rng(1)
    base = rng.integers(0, 256, (h, w * 4), dtype=np.uint8)
    base[:, 3::4] = 255
    mod = base.copy()
    mod[20:70, 4 * 30 + 3:4 * 150 + 3:4] = rng.integers(0, 200, (50, 120), dtype=np.uint8)   # a translucent patch
    a, b = ctx.dssim_create_image(base, w * 4, w, h, "RGBA"), ctx.dssim_create_image(mod, w * 4, w, h, "RGBA")
    try:
        got = ctx.dssim_compare(a, b)
        same = ctx.dssim_compare(a, a)
    finally:
        ctx.dssim_free_image(a)
        ctx.dssim_free_image(b)
    exp = D.compare(D.DssimImage(base, w, h, w * 4, 4), D.DssimImage(mod, w, h, w * 4, 4))
    assert same == 0.0 and got > 1e-3 and got == pytest.approx(exp, rel=1e-9, abs=1e-13)
    rgb = np.ascontiguousarray(base.reshape(h, w, 4)[..., :3]).reshape(h, w * 3)
    x, y = ctx.dssim_create_image(base, w * 4, w, h, "RGBA"), ctx.dssim_create_image(rgb, w * 3, w, h, "RGB")
    try:
        assert ctx.dssim_compare(x, y) == 0.0
    finally:
        ctx.dssim_free_image(x)
        ctx.dssim_free_image(y)


def test_dssim_matches_restatement_at_4k(ctx, synth):
    """BASELINE config 5's frame size: the natural-like 4K frame against itself + N(0, 2) noise and against a heavily
    changed copy, device vs the numpy restatement (same tolerance as the smaller sizes)."""
    from oracle import dssim_restate as D
    w, h = 3840, 2160
    rng = np.random.default_rng(5)
    base = synth.smooth_frame(w, h)
    base[:, 3::4] = 255
    ga = ctx.dssim_create_image(base, w * 4, w, h)
    oa = D.DssimImage(base, w, h, w * 4, 4)
    try:
        for sigma in (2.0, 25.0):
            mod = base.reshape(h, w, 4).astype(np.int16)
            mod[..., :3] += np.rint(rng.normal(0.0, sigma, size=(h, w, 3))).astype(np.int16)
            mod = np.clip(mod, 0, 255).astype(np.uint8).reshape(h, w * 4)
            mod[:, 3::4] = 255
            gb = ctx.dssim_create_image(mod, w * 4, w, h)
            got = ctx.dssim_compare(ga, gb)
            ctx.dssim_free_image(gb)
            exp = D.compare(oa, D.DssimImage(mod, w, h, w * 4, 4))
            assert got == pytest.approx(exp, rel=1e-9, abs=1e-13), (sigma, got, exp)
    finally:
        ctx.dssim_free_image(ga)


def test_config5_bench_leg_runs(tmp_path):
    """bench.py --config 5 at BASELINE's shape: 32 concurrent 4K streams per GPU through the SSIM engine, 8 worker contexts
    (2 timed steps = 64 comparisons): one JSON line with comparisons/s, the roofline object and a plausible dssim value."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "5", "--streams", "32", "--workers", "8", "--steps", "2", "--warmup", "1",
                        "--ramp-seconds", "0.05", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["unit"] == "comparisons/s" and d["value"] > 0 and d["config"]["streams_per_gpu"] == 32 and d["config"]["worker_contexts_per_gpu"] == 8
    # the engine is VALU-bound (exact f32, ~1,300 instructions per pixel pair): that is the roofline it is priced against; the HBM figure rides along
    assert d["roofline"]["bound"] == "valu" and 0 < d["roofline"]["frac"] < 1 and d["roofline"]["hbm"]["frac"] < 0.05 and 0 < d["dssim_of_stream_0"] < 0.1
    # ... and through the dispatcher: the same value for stream 0
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "5", "--group", "--streams", "32", "--workers", "8", "--steps", "2", "--warmup", "1",
                        "--ramp-seconds", "0.05", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    g = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert g["dispatcher"]["pairs_in_the_largest"] == 32 and g["dssim_of_stream_0"] == d["dssim_of_stream_0"]


@pytest.mark.parametrize("w,h,fmt", [(128, 96, "RGBA"), (322, 246, "RGBA"), (35, 20, "RGB"), (7, 50, "RGBA"), (1920, 1080, "RGBA"), (641, 363, "RGB"),
                                     (3840, 2160, "RGBA")])
def test_dssim_compare_frames_is_create_plus_compare_bit_for_bit(ctx, w, h, fmt):
    """mi355_dssim_compare_frames (hash and compare in one pass, videocompare/imp.rs:316-345) returns the f64 bits of
    mi355_dssim_create_image + mi355_dssim_compare for every frame of the call - interior tiles, border tiles, sizes below one
    tile, fewer than five scales, RGB and RGBA, padded rows."""
    ch = 4 if fmt == "RGBA" else 3
    rng = np.random.default_rng(w * 7 + h)
    smooth = np.kron(rng.integers(0, 256, ((h + 7) // 8, (w + 7) // 8, ch), dtype=np.uint8), np.ones((8, 8, 1), np.uint8))[:h, :w].reshape(h, w * ch)
    stride = w * ch + 12
    def padded(a):
        p = np.zeros((h, stride), np.uint8); p[:, : w * ch] = a
        return p
    base = smooth.copy()
    if ch == 4:
        base[:, 3::4] = 255
    frames = []
    for amp in (0, 2, 25, 120):
        f = np.clip(base.astype(int) + rng.integers(-amp, amp + 1, base.shape), 0, 255).astype(np.uint8)
        if ch == 4:
            f[:, 3::4] = 255 if amp != 25 else rng.integers(0, 256, (h, w), dtype=np.uint8)   # one translucent frame
        frames.append(padded(f))
    ref = ctx.dssim_create_image(padded(base), stride, w, h, fmt)
    try:
        two_step = []
        for f in frames:
            g = ctx.dssim_create_image(f, stride, w, h, fmt)
            two_step.append(ctx.dssim_compare(ref, g))
            ctx.dssim_free_image(g)
        fused = ctx.dssim_compare_frames(ref, frames, stride, w, h, fmt)
        assert fused == two_step, (fused, two_step)
        assert fused[0] == 0.0
        # one frame per call, and the reference's planes are untouched by the fused passes
        assert ctx.dssim_compare_frames(ref, frames[2:3], stride, w, h, fmt) == two_step[2:3]
        assert ctx.dssim_compare(ref, ref) == 0.0
        assert ctx.dssim_compare_frames(ref, [], stride, w, h, fmt) == []
    finally:
        ctx.dssim_free_image(ref)


def test_dssim_compare_frames_device_and_errors(ctx):
    import mi355fx
    w, h = 256, 144
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (h, w * 4), dtype=np.uint8); a[:, 3::4] = 255
    b = rng.integers(0, 256, (h, w * 4), dtype=np.uint8); b[:, 3::4] = 255
    ref = ctx.dssim_create_image(a, w * 4, w, h)
    d = ctx.alloc(2 * a.nbytes)
    try:
        ctx.h2d(d, np.concatenate([a.reshape(-1), b.reshape(-1)]))
        got = ctx.dssim_compare_frames_device(ref, [d, d + a.nbytes], w * 4, w, h)
        assert got == ctx.dssim_compare_frames(ref, [a, b], w * 4, w, h)
        assert got[0] == 0.0 and got[1] > 0.1
        with pytest.raises(mi355fx.Mi355Error):
            ctx.dssim_compare_frames(ref, [a[:, : 128 * 4].copy()], 128 * 4, 128, h)      # size differs from the original's
        with pytest.raises(mi355fx.Mi355Error):
            ctx.dssim_compare_frames(ref, [a], w * 4, w, h, "BGRx")
    finally:
        ctx.free(d); ctx.dssim_free_image(ref)


@pytest.mark.parametrize("w,h,fmt", [(35, 20, "RGB"), (35, 20, "RGBA"), (641, 363, "RGB"), (1920, 1080, "RGBA")])
def test_dssim_create_image_is_done_with_the_host_frame_when_it_returns(ctx, w, h, fmt):
    """The element unmaps its buffer right after HashedImage::new: the host variant of create_image may not read `data` after
    it has returned (round-3 regression: without the final wait the upload of a 105-byte-row frame was still in flight)."""
    ch = 4 if fmt == "RGBA" else 3
    rng = np.random.default_rng(w + h)
    stride = w * ch + 4
    frame = rng.integers(0, 256, (h, stride), dtype=np.uint8)
    if ch == 4:
        frame[:, 3: w * 4: 4] = 255
    keep = frame.copy()
    a = ctx.dssim_create_image(frame, stride, w, h, fmt)
    frame[:] = 0                                            # the caller's buffer is recycled at once
    b = ctx.dssim_create_image(keep, stride, w, h, fmt)
    try:
        assert ctx.dssim_compare(a, b) == 0.0
    finally:
        ctx.dssim_free_image(a); ctx.dssim_free_image(b)


def test_dssim_cube_root_with_trimmed_division_over_its_whole_domain(ctx):
    """The LAB conversion divides without v_div_scale / v_div_fixup (csrc/dssim_kernels.hip: dssim_div_unscaled). Every f32 the
    conversion can feed the cube root - (216/24389, 1.0x] - and an octave beyond goes through both forms: no bit differs."""
    eps = np.nextafter(np.float32(216.0 / 24389.0), np.float32(1))
    assert ctx.selftest_dssim_cbrt(eps, 2.0) == 0
    assert ctx.selftest_dssim_cbrt(2.0, 64.0) == 0          # far beyond what an image can produce
