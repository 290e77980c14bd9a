"""GPU parity of the LDS-cached memoised-table kernel (csrc/colorlut_window.hip, MI355_FLAG_LUT_VARIANT 8): packed RGBA8
through the Morton-indexed 2^24-entry table, table bricks cached per block in LDS, misses served from the table in
global memory. Replaces the per-pixel loop of video/colorlut/src/colorlut/imp.rs:267-294 (and, for the fused entry point,
video/hsv/src/hsvfilter/imp.rs:323-376 in front of it); the oracle is the C restatement of both.

The kernel has no barrier after its prologue - waves install bricks while others read - so besides the usual geometry
cases these tests hammer it with content that keeps many installs in flight (noise of growing amplitude, two colour
clusters per strip, every colour once) and compare with the oracle AND with the gather kernels on the same table."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W4K, H4K = 3840, 2160


@pytest.fixture(autouse=True, params=[1, 0], ids=["fronts", "shares"])
def window_kind(request, ctx):
    """Both ways the kernel's blocks share the picture (MI355_FLAG_WINDOW_ORDER: aligned fronts - the default - and round 4's
    contiguous shares), with the diagnostic counters on. Returns the name the library must report."""
    import mi355fx
    ctx.set_flag(mi355fx.FLAG_WINDOW_ORDER, request.param)
    ctx.set_flag(mi355fx.FLAG_WINDOW_STATS, 1)
    return "colorlut_window_kernel"


def _load_cube(ctx, oracle, text):
    cube = oracle.Cube.parse(text)
    sc, of = cube.domain
    ctx.colorlut_load(cube.is3d, cube.size, cube.table, sc, of)
    return cube


def _window_ctx(ctx, min_steps=0):
    import mi355fx
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 8)
    ctx.set_flag(mi355fx.FLAG_WINDOW_MIN_STEPS, min_steps)


def _run(ctx, src, w, h, n=1, sstride=None, dstride=None, fused_st=None, reps=1):
    ss = w * 4 if sstride is None else sstride
    ds = w * 4 if dstride is None else dstride
    out = np.full(n * h * ds, 0x5A, np.uint8)
    d_src, d_dst = ctx.alloc(src.nbytes), ctx.alloc(out.nbytes)
    try:
        ctx.h2d(d_src, src)
        ctx.h2d(d_dst, out)
        for _ in range(reps):
            if fused_st is not None:
                ctx.hsv_colorlut_frames_device(d_src, h * ss, ss, d_dst, h * ds, ds, n, w, h, fused_st)
            else:
                ctx.colorlut_frames_device(d_src, h * ss, ss, d_dst, h * ds, ds, n, w, h, "RGBA")
        ctx.synchronize()
        ctx.d2h(out, d_dst)
    finally:
        ctx.free(d_src)
        ctx.free(d_dst)
    return out


@pytest.mark.parametrize("w,h", [(4, 1), (8, 3), (100, 37), (128, 4), (256, 32), (260, 33), (516, 3), (1000, 65), (1920, 1081), (3840, 7), (132, 4000)])
def test_window_kernel_geometry(ctx, oracle, synth, w, h, window_kind):
    """Sizes around the kernel's 256 x 32 pixel steps: masked last strip, masked last rows, fewer steps than CUs, one strip."""
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(17))
    _window_ctx(ctx)
    src = synth.noise_frame(w, h, seed=w + h).reshape(-1)
    exp = np.zeros_like(src)
    oracle.colorlut_rgba8(cube, src, w * 4, exp, w * 4, w, h)
    got = _run(ctx, src, w, h)
    assert ctx.colorlut_kernel_name() == window_kind
    assert (got == exp).all()


@pytest.mark.parametrize("w,h,spad,dpad,n", [(640, 70, 16, 48, 3), (1280, 33, 0, 32, 2), (516, 40, 64, 0, 1)])
def test_window_kernel_padded_rows_and_batches(ctx, oracle, synth, w, h, spad, dpad, n, window_kind):
    """Row strides padded to multiples of 16 B, independent for source and destination; a batch is one tall picture; the
    destination's padding stays untouched."""
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(33))
    _window_ctx(ctx)
    ss, ds = w * 4 + spad, w * 4 + dpad
    rng = np.random.default_rng(w)
    src = rng.integers(0, 256, size=(n, h, ss), dtype=np.uint8)
    base = synth.smooth_frame(w, h, seed=3).reshape(h, w * 4)
    for f in range(n):
        src[f, :, :w * 4] = np.roll(base, 8 * f, axis=1)
    src = src.reshape(-1)
    exp = np.full(n * h * ds, 0x5A, np.uint8)
    for f in range(n):
        oracle.colorlut_rgba8(cube, src[f * h * ss:(f + 1) * h * ss], ss, exp[f * h * ds:(f + 1) * h * ds], ds, w, h)
    got = _run(ctx, src, w, h, n=n, sstride=ss, dstride=ds)
    assert ctx.colorlut_kernel_name() == window_kind
    assert (got == exp).all()


def _noisy(synth, amp, seed, n=1):
    base = np.stack([synth.smooth_frame(W4K, H4K, seed=seed + i) for i in range(n)]).reshape(n, H4K, W4K, 4).astype(np.int16)
    if amp:
        rng = np.random.default_rng(seed)
        base[..., :3] += rng.integers(-amp, amp + 1, size=base[..., :3].shape, dtype=np.int16)
    return np.clip(base, 0, 255).astype(np.uint8).reshape(-1)


@pytest.mark.parametrize("amp", [0, 4, 8, 16, 48])
def test_window_kernel_4k_content_sweep(ctx, oracle, synth, amp, window_kind):
    """The bench's natural-like frame plus uniform noise of growing amplitude: from "nearly every pixel found in LDS" to
    "installs in flight everywhere, most pixels past the cache". Exact against the oracle, identical to the gather
    kernel on the same table, launch after launch (the cache is rebuilt by every launch; timing differs every time)."""
    import mi355fx
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(33))
    src = _noisy(synth, amp, seed=20 + amp)
    exp = np.zeros_like(src)
    oracle.colorlut_rgba8(cube, src, W4K * 4, exp, W4K * 4, W4K, H4K, nthreads=8)
    _window_ctx(ctx, min_steps=3)
    ctx.colorlut_window_stats(reset=True)
    for rep in range(4):
        got = _run(ctx, src, W4K, H4K)
        assert ctx.colorlut_kernel_name() == window_kind
        assert (got == exp).all(), "launch %d" % rep
    px, past, installs = ctx.colorlut_window_stats()
    assert px >= 4 * W4K * H4K and installs > 0  # lookups (lanes of masked rows included); bricks were installed
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 5)
    assert (_run(ctx, src, W4K, H4K) == exp).all()
    assert ctx.colorlut_kernel_name() == "colorlut_table_tiled_kernel"


def test_window_kernel_two_clusters_per_strip(ctx, oracle, synth):
    """Vertical stripes 24 pixels wide alternating between two colour clusters that are a multiple of the cache's box apart
    (32 levels in r and g, 16 in b): every set is asked for two bricks all the time - one per way."""
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(33))
    _window_ctx(ctx)
    w, h = 1024, 12000  # four strips, six steps per block: the first one is cold
    rng = np.random.default_rng(11)
    img = np.empty((h, w, 4), np.uint8)
    a = np.array([40, 90, 130]) + rng.integers(-5, 6, size=(h, w, 3))
    b = np.array([40 + 64, 90 + 96, 130 + 48]) + rng.integers(-5, 6, size=(h, w, 3))
    stripe = ((np.arange(w) // 24) % 2).astype(bool)
    img[..., :3] = np.where(stripe[None, :, None], b, a).astype(np.uint8)
    img[..., 3] = rng.integers(0, 256, size=(h, w))
    src = img.reshape(-1)
    exp = np.zeros_like(src)
    oracle.colorlut_rgba8(cube, src, w * 4, exp, w * 4, w, h, nthreads=8)
    ctx.colorlut_window_stats(reset=True)
    for _ in range(3):
        assert (_run(ctx, src, w, h) == exp).all()
    px, past, installs = ctx.colorlut_window_stats()
    assert past < 0.3 * px, (px, past, installs)  # two ways hold two clusters


def test_window_kernel_three_clusters_thrash_and_stay_exact(ctx, oracle, synth):
    """Three clusters that all map to the same sets: two ways cannot hold them, bricks are evicted while other waves read
    them. Results stay exact (a reader that overlaps an install notices the generation change and goes to the table)."""
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(33))
    _window_ctx(ctx)
    w, h = 1024, 1500
    rng = np.random.default_rng(12)
    img = np.empty((h, w, 4), np.uint8)
    centres = np.array([[30, 60, 20], [30 + 32, 60 + 64, 20 + 16], [30 + 96, 60 + 32, 20 + 64]])
    which = rng.integers(0, 3, size=(h, w))
    img[..., :3] = (centres[which] + rng.integers(-3, 4, size=(h, w, 3))).astype(np.uint8)
    img[..., 3] = 7
    src = img.reshape(-1)
    exp = np.zeros_like(src)
    oracle.colorlut_rgba8(cube, src, w * 4, exp, w * 4, w, h, nthreads=8)
    for rep in range(6):
        assert (_run(ctx, src, w, h) == exp).all(), "launch %d" % rep


def test_window_kernel_fused_chain_batch(ctx, oracle, synth, window_kind):
    """The fused entry point (table of hsvfilter -> colorlut) through the LDS-cached kernel on a 4 x 4K batch: first and
    last frame against the oracle chain, everything against the gather kernel."""
    import mi355fx
    st = synth.HSV_SETTINGS["hue90"]
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(33))
    n = 4
    src = _noisy(synth, 2, seed=60, n=n)
    _window_ctx(ctx, min_steps=3)
    ctx.colorlut_window_stats(reset=True)
    got = _run(ctx, src, W4K, H4K, n=n, fused_st=st)
    assert ctx.colorlut_kernel_name() == window_kind
    px, past, installs = ctx.colorlut_window_stats()
    assert past < 0.1 * px and installs > 0, (px, past, installs)  # the cache serves the pixels (16 steps per block, the first one cold)
    fb = W4K * H4K * 4
    for f in (0, n - 1):
        mid = src[f * fb:(f + 1) * fb].copy()
        oracle.hsvfilter(mid, W4K, W4K * 4, 4, 0, False, st, nthreads=8)
        exp = np.zeros_like(mid)
        oracle.colorlut_rgba8(cube, mid, W4K * 4, exp, W4K * 4, W4K, H4K, nthreads=8)
        assert (got[f * fb:(f + 1) * fb] == exp).all(), "frame %d" % f
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 5)
    assert (_run(ctx, src, W4K, H4K, n=n, fused_st=st) == got).all()


def test_window_kernel_in_place(ctx, oracle, synth):
    """src == dst (the element's in-place mode): every pixel group is read before it is written by the same lane."""
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(33))
    _window_ctx(ctx)
    w, h = 1920, 1080
    src = synth.smooth_frame(w, h, seed=5).reshape(-1)
    exp = np.zeros_like(src)
    oracle.colorlut_rgba8(cube, src, w * 4, exp, w * 4, w, h, nthreads=8)
    d = ctx.alloc(src.nbytes)
    try:
        ctx.h2d(d, src)
        ctx.colorlut_frames_device(d, h * w * 4, w * 4, d, h * w * 4, w * 4, 1, w, h, "RGBA")
        ctx.synchronize()
        got = np.zeros_like(src)
        ctx.d2h(got, d)
    finally:
        ctx.free(d)
    assert (got == exp).all()


def test_the_table_kernel_is_chosen_by_where_the_pixels_come_from(ctx, oracle, synth, window_kind):
    """MI355_FLAG_LUT_VARIANT 9 (what auto does once it is on the table): ONE Morton table, read by the LDS-cached kernel when the
    frames come from HBM and by the gather kernel when the launch before wrote them (they are on-die then: hsvfilter ! colorlut) -
    a rule on the input's provenance (round 6; rounds 4-5 measured one kernel against the other with samples up to 1024 launches
    old). Results exact either way; a 720p frame goes through the tiled gather kernel on the same table."""
    import mi355fx
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(33))
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 9)
    src = _noisy(synth, 0, seed=70, n=2)
    exp = np.zeros(W4K * H4K * 4, np.uint8)
    oracle.colorlut_rgba8(cube, src[: exp.size], W4K * 4, exp, W4K * 4, W4K, H4K, nthreads=8)
    st = synth.HSV_SETTINGS["hue90"]
    mid = src[: exp.size].copy()
    oracle.hsvfilter(mid, W4K, W4K * 4, 4, 0, False, st)
    exp_chain = np.zeros_like(exp)
    oracle.colorlut_rgba8(cube, mid, W4K * 4, exp_chain, W4K * 4, W4K, H4K, nthreads=8)
    d_s, d_o = ctx.alloc(src.nbytes), ctx.alloc(src.nbytes)
    pitch = H4K * W4K * 4
    try:
        for _ in range(3):
            # frames uploaded by a copy: from HBM -> the LDS-cached kernel, every time (nothing to settle, nothing to probe)
            ctx.h2d(d_s, src)
            for _ in range(3):
                ctx.colorlut_frames_device(d_s, pitch, W4K * 4, d_o, pitch, W4K * 4, 2, W4K, H4K, "RGBA")
                assert ctx.colorlut_kernel_name() == window_kind and ctx.colorlut_kernel_choice(fused=10)[0]
            out = np.zeros_like(src)
            ctx.d2h(out, d_o)
            assert (out[: exp.size] == exp).all()
            # the same frames behind hsvfilter on this device: on-die -> the gather kernel
            ctx.hsvfilter_frames_device(d_s, 2, pitch, W4K, H4K, W4K * 4, "RGBA", st)
            ctx.colorlut_frames_device(d_s, pitch, W4K * 4, d_o, pitch, W4K * 4, 2, W4K, H4K, "RGBA")
            assert ctx.colorlut_kernel_name() == "colorlut_table_tiled_kernel" and not ctx.colorlut_kernel_choice(fused=10)[0]
            ctx.d2h(out, d_o)
            assert (out[: exp.size] == exp_chain).all()
            # ... also when ANOTHER context of the device wrote them (hsvfilter and colorlut are two elements, two contexts)
            other = mi355fx.Context(0)
            ctx.h2d(d_s, src)
            other.hsvfilter_frames_device(d_s, 2, pitch, W4K, H4K, W4K * 4, "RGBA", st)
            other.synchronize()
            ctx.colorlut_frames_device(d_s, pitch, W4K * 4, d_o, pitch, W4K * 4, 2, W4K, H4K, "RGBA")
            assert ctx.colorlut_kernel_name() == "colorlut_table_tiled_kernel"
            other.close()
            ctx.d2h(out, d_o)
            assert (out[: exp.size] == exp_chain).all()
        tables = mi355fx.load_library().mi355_shared_table_count()
        small = synth.smooth_frame(1280, 720, seed=71).reshape(-1)
        got = _run(ctx, small, 1280, 720)
        assert ctx.colorlut_kernel_name() == "colorlut_table_tiled_kernel"
        exp_s = np.zeros_like(small)
        oracle.colorlut_rgba8(cube, small, 1280 * 4, exp_s, 1280 * 4, 1280, 720, nthreads=8)
        assert (got == exp_s).all()
        assert mi355fx.load_library().mi355_shared_table_count() == tables  # the same table served both
    finally:
        ctx.free(d_s)
        ctx.free(d_o)
