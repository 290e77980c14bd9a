"""GPU parity of the block-shared brick cache (colorlut3d_shared_kernel in csrc/colorlut_brick.hip; MI355_FLAG_BRICK_SETS =
512 pins it) against the CPU oracle, through the C ABI: every 8-bit colour (every LUT cell, constant misses and evictions:
the lock / generation protocol under the heaviest traffic it can see), LUT sizes 2..65 and non-unit domains, non-finite
entries, 4K batches of smooth / noisy / noise content in and out of place, ragged geometries, concurrent contexts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_gpu_brick import _report, _load, _device_lut, _varying_alpha  # noqa: E402


def _pin(ctx):
    import mi355fx
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 7)
    ctx.set_flag(mi355fx.FLAG_BRICK_SETS, 512)


@pytest.mark.parametrize("size,domain", [(33, None), (2, None), (3, None), (9, None), (17, None), (34, None), (64, None), (65, None),
                                         (33, ((-0.25, 0.0, 0.1), (1.5, 1.0, 0.9))), (5, ((0.0, 0.0, 0.0), (2.0, 0.5, 1.0)))])
def test_shared_kernel_allcolors(ctx, oracle, synth, size, domain):
    cube = _load(ctx, oracle, synth.cube_text_3d(size, amp=0.07, domain=domain))
    ac = _varying_alpha(synth.allcolors())
    exp = np.zeros_like(ac)
    oracle.colorlut_rgba8(cube, ac, 4096 * 4, exp, 4096 * 4, 4096, 4096, nthreads=8)
    _pin(ctx)
    got = np.zeros_like(ac)
    ctx.colorlut_frame(ac, 4096 * 4, got, 4096 * 4, 4096, 4096, "RGBA")
    assert ctx.colorlut_kernel_name() == "colorlut3d_shared_kernel"
    assert (got == exp).all(), _report(got, exp)


def test_shared_kernel_nonfinite_lut_entries(ctx, oracle, synth):
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
    cube = _load(ctx, oracle, "\n".join(lines) + "\n")
    ac = synth.allcolors()
    exp = np.zeros_like(ac)
    oracle.colorlut_rgba8(cube, ac, 4096 * 4, exp, 4096 * 4, 4096, 4096, nthreads=8)
    _pin(ctx)
    got = np.zeros_like(ac)
    ctx.colorlut_frame(ac, 4096 * 4, got, 4096 * 4, 4096, 4096, "RGBA")
    assert ctx.colorlut_kernel_name() == "colorlut3d_shared_kernel"
    assert (got == exp).all(), _report(got, exp)


@pytest.mark.parametrize("content", ["smooth", "amp4", "amp8", "amp32", "noise"])
@pytest.mark.parametrize("in_place", [False, True])
def test_shared_kernel_4k_batch(ctx, oracle, synth, content, in_place):
    """3 x 3840x2160 through the device entry point: output == T[input] with T = the oracle's output on every colour."""
    cube = _load(ctx, oracle, synth.cube_text_3d(33))
    if content == "noise":
        frames = np.stack([synth.noise_frame(3840, 2160, seed=synth.SEED + 3 + i) for i in range(3)])
    else:
        frames = np.stack([synth.smooth_frame(3840, 2160, seed=synth.SEED + 3 + i) for i in range(3)])
        if content != "smooth":
            amp = int(content[3:])
            f = frames.reshape(3, 2160, 3840, 4).astype(np.int16)
            f[..., :3] += np.random.default_rng(9).integers(-amp, amp + 1, size=f[..., :3].shape, dtype=np.int16)
            frames = np.clip(f, 0, 255).astype(np.uint8).reshape(frames.shape)
    ac = synth.allcolors()
    table = np.zeros_like(ac)
    oracle.colorlut_rgba8(cube, ac, 4096 * 4, table, 4096 * 4, 4096, 4096, nthreads=8)
    t = table.reshape(-1, 4)
    px = frames.reshape(-1, 4)
    idx = px[:, 0].astype(np.uint32) | (px[:, 1].astype(np.uint32) << 8) | (px[:, 2].astype(np.uint32) << 16)
    exp = t[idx].copy()
    exp[:, 3] = px[:, 3]
    _pin(ctx)
    ctx.colorlut_brick_stats(reset=True)
    got = _device_lut(ctx, frames, 3840, 2160, in_place=in_place)
    assert ctx.colorlut_kernel_name() == "colorlut3d_shared_kernel"
    assert (got.reshape(-1, 4) == exp).all(), _report(got, exp)
    steps, slow, _, _ = ctx.colorlut_brick_stats()
    total_steps = frames.size // 4 // 256
    if content in ("smooth", "amp4", "amp8"):
        assert slow < 0.10 * total_steps, "%d of %d steps read bricks past the cache" % (slow, total_steps)


@pytest.mark.parametrize("w,h,n", [(256, 96, 260), (260, 97, 130), (1920, 1081, 4), (3840, 7, 500), (1000, 9, 700), (4, 50000, 1), (8, 4097, 7), (516, 300, 30)])
def test_shared_kernel_ragged_geometry(ctx, oracle, synth, w, h, n):
    """Widths that are not whole 256-pixel strips, heights that are not whole 32-row steps, many frames per launch (the
    kernel needs enough steps per block to be chosen at all: these are all above that bar)."""
    cube = _load(ctx, oracle, synth.cube_text_3d(17, amp=0.1))
    rng = np.random.default_rng(w * 1000 + h)
    base = synth.smooth_frame(max(w, 16), max(h, 4))[:h, : w * 4]
    frames = np.stack([np.roll(base, 4 * (i % 61), axis=1) for i in range(n)]).copy()
    frames[:, ::3, ::7] = rng.integers(0, 256, size=frames[:, ::3, ::7].shape, dtype=np.uint8)
    exp = np.zeros_like(frames)
    for i in range(n):
        oracle.colorlut_rgba8(cube, frames[i], w * 4, exp[i], w * 4, w, h, nthreads=8)
    _pin(ctx)
    got = _device_lut(ctx, frames, w, h).reshape(frames.shape)
    assert ctx.colorlut_kernel_name() == "colorlut3d_shared_kernel"
    assert (got == exp).all(), _report(got, exp)


def test_shared_kernel_small_launches_fall_back(ctx, oracle, synth):
    """Too few steps for a block's cache to warm up: the per-wave brick kernel serves, results unchanged."""
    cube = _load(ctx, oracle, synth.cube_text_3d(17, amp=0.1))
    frames = synth.smooth_frame(640, 480)[None].copy()
    exp = np.zeros_like(frames)
    oracle.colorlut_rgba8(cube, frames[0], 640 * 4, exp[0], 640 * 4, 640, 480)
    _pin(ctx)
    got = _device_lut(ctx, frames, 640, 480).reshape(frames.shape)
    assert ctx.colorlut_kernel_name().startswith("colorlut3d_brick_kernel")
    assert (got == exp).all(), _report(got, exp)


def test_shared_kernels_of_concurrent_contexts(mi355lib, oracle, synth):
    """Four contexts, four host threads and streams, different LUTs: blocks of different launches interleave on the CUs;
    each block's cache, locks and queue are its own LDS."""
    import threading
    import mi355fx
    w, h, n = 1920, 1080, 4
    jobs = []
    for k in range(4):
        cube = oracle.Cube.parse(synth.cube_text_3d([33, 17, 65, 9][k], amp=0.05 + 0.01 * k))
        frames = np.stack([np.roll(synth.smooth_frame(w, h, seed=40 + k), 4 * 37 * i, axis=1) for i in range(n)]).copy()
        exp = np.zeros_like(frames)
        for i in range(n):
            oracle.colorlut_rgba8(cube, frames[i], w * 4, exp[i], w * 4, w, h, nthreads=2)
        jobs.append((cube, frames, exp))
    errors = []
    barrier = threading.Barrier(4)

    def worker(k):
        cube, frames, exp = jobs[k]
        c = mi355fx.Context(0)
        try:
            sc, of = cube.domain
            c.colorlut_load(cube.is3d, cube.size, cube.table, sc, of)
            _pin(c)
            barrier.wait()
            for it in range(6):
                got = _device_lut(c, frames, w, h).reshape(frames.shape)
                if c.colorlut_kernel_name() != "colorlut3d_shared_kernel":
                    errors.append("context %d served by %s" % (k, c.colorlut_kernel_name()))
                if not (got == exp).all():
                    errors.append("context %d launch %d: %s" % (k, it, _report(got, exp)))
        except Exception as e:  # noqa: BLE001 - reported below
            errors.append("context %d: %r" % (k, e))
        finally:
            c.close()

    th = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_content_watch_ladder_goes_through_the_shared_cache(ctx, oracle, synth):
    """MI355_FLAG_LUT_VARIANT 6 on 2 x 4K per launch: smooth frames stay on the per-wave brick kernel, +-8 of noise moves
    the stream to the block-shared cache (level 1) within a few launches and keeps it there, uniform noise moves it on to
    the three-pass kernel, calm content brings it back down. Every output stays exact on the way."""
    import mi355fx
    cube = _load(ctx, oracle, synth.cube_text_3d(33))
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 6)
    ctx.set_flag(mi355fx.FLAG_BRICK_SETS, 0)
    w, h, n = 3840, 2160, 2
    ac = synth.allcolors()
    table = np.zeros_like(ac)
    oracle.colorlut_rgba8(cube, ac, 4096 * 4, table, 4096 * 4, 4096, 4096, nthreads=8)
    t = table.reshape(-1, 4)

    def expect(frames):
        px = frames.reshape(-1, 4)
        idx = px[:, 0].astype(np.uint32) | (px[:, 1].astype(np.uint32) << 8) | (px[:, 2].astype(np.uint32) << 16)
        e = t[idx].copy()
        e[:, 3] = px[:, 3]
        return e

    smooth = np.stack([synth.smooth_frame(w, h, seed=5 + i) for i in range(n)])
    f = smooth.reshape(n, h, w, 4).astype(np.int16)
    f[..., :3] += np.random.default_rng(3).integers(-8, 9, size=f[..., :3].shape, dtype=np.int16)
    noisy = np.clip(f, 0, 255).astype(np.uint8).reshape(smooth.shape)
    noise = np.stack([synth.noise_frame(w, h, seed=9 + i) for i in range(n)])
    es, ey, en = expect(smooth), expect(noisy), expect(noise)

    def run(frames, exp, count):
        names = []
        for _ in range(count):
            got = _device_lut(ctx, frames, w, h)
            assert (got.reshape(-1, 4) == exp).all(), _report(got, exp)
            names.append(ctx.colorlut_kernel_name())
        return names

    names = run(smooth, es, 10)
    assert all(k == "colorlut3d_brick_kernel" for k in names), names
    names = run(noisy, ey, 24)
    assert names[-1] == "colorlut3d_shared_kernel" and names[-8:] == ["colorlut3d_shared_kernel"] * 8, names
    names = run(noise, en, 24)
    assert names[-1] == "colorlut3d_lds_kernel", names
    names = run(noisy, ey, 80)   # one probation of level 1 after 64 launches at level 2
    assert names[-1] == "colorlut3d_shared_kernel", names[-12:]
    names = run(smooth, es, 80)  # ... and of level 0 after 64 at level 1
    assert names[-1] == "colorlut3d_brick_kernel", names[-12:]


def test_single_frame_launches_start_on_the_shared_cache(ctx, oracle, synth):
    """One 4K frame per launch (what a pipeline hands over per buffer), interpolating path: the 16 waves of a CU warm ONE cache
    instead of one each, so level 1 is where such launches start - on clean content too - and the results are the oracle's."""
    import mi355fx
    cube = _load(ctx, oracle, synth.cube_text_3d(33))
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 6)
    ctx.set_flag(mi355fx.FLAG_BRICK_SETS, 0)
    w, h = 3840, 2160
    frame = synth.smooth_frame(w, h, seed=3)[None]
    exp = np.zeros_like(frame)
    oracle.colorlut_rgba8(cube, frame[0], w * 4, exp[0], w * 4, w, h, nthreads=8)
    names = []
    for _ in range(12):
        got = _device_lut(ctx, frame, w, h).reshape(frame.shape)
        assert (got == exp).all(), _report(got, exp)
        names.append(ctx.colorlut_kernel_name())
    assert names == ["colorlut3d_shared_kernel"] * 12, names


@pytest.mark.parametrize("which", ["shared", "window"])
@pytest.mark.parametrize("w,h,n,spad,dpad", [(1920, 1080, 4, 16, 48), (1000, 700, 12, 96, 0), (3840, 2160, 1, 0, 64)])
def test_lds_cache_kernels_take_padded_rows(ctx, oracle, synth, which, w, h, n, spad, dpad):
    """Source and destination strides padded independently to multiples of 16 B (colorlut/imp.rs:275-286), frames
    `stride * height` apart, launches large enough for the per-CU caches: the block-shared brick cache and the LDS-cached
    table kernel walk rows by stride; the padding of the destination stays untouched."""
    import mi355fx
    cube = _load(ctx, oracle, synth.cube_text_3d(33))
    ss, ds = w * 4 + spad, w * 4 + dpad
    rng = np.random.default_rng(w + n)
    base = synth.smooth_frame(w, h, seed=9).reshape(h, w * 4)
    src = rng.integers(0, 256, size=(n, h, ss), dtype=np.uint8)
    for f in range(n):
        src[f, :, :w * 4] = np.roll(base, 4 * 41 * f, axis=1)
    src = src.reshape(-1)
    exp = np.full(n * h * ds, 0xEE, np.uint8)
    for f in range(n):
        oracle.colorlut_rgba8(cube, src[f * h * ss:(f + 1) * h * ss].copy(), ss, exp[f * h * ds:(f + 1) * h * ds], ds, w, h, nthreads=8)
    if which == "shared":
        _pin(ctx)
    else:
        ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 8)
    d_src, d_dst = ctx.alloc(src.nbytes), ctx.alloc(exp.nbytes)
    try:
        ctx.h2d(d_src, src)
        got = np.empty_like(exp)
        for rep in range(2):   # (the table kernel's first launch builds the table)
            ctx.h2d(d_dst, np.full(n * h * ds, 0xEE, np.uint8))
            ctx.colorlut_frames_device(d_src, h * ss, ss, d_dst, h * ds, ds, n, w, h, "RGBA")
            ctx.synchronize()
            ctx.d2h(got, d_dst)
            assert (got == exp).all(), (rep, _report(got, exp))
        assert ctx.colorlut_kernel_name() == ("colorlut3d_shared_kernel" if which == "shared" else "colorlut_window_kernel"), ctx.colorlut_kernel_name()
    finally:
        ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 0)
        ctx.set_flag(mi355fx.FLAG_BRICK_SETS, 0)
        ctx.free(d_src)
        ctx.free(d_dst)
