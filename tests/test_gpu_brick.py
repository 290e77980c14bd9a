"""GPU parity of the brick-cache interpolating colorlut kernel (csrc/colorlut_brick.hip, MI355_FLAG_LUT_VARIANT 7 pins
it) against the CPU oracle, through the C ABI: every 8-bit colour, LUT sizes 2..65, non-unit domains, non-finite LUT
entries, natural-like and noise frames at 4K, ragged geometries, in place, the fused hsvfilter -> colorlut form, and the
content watch that hands noise-like streams to the three-pass kernel."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _report(got, exp, limit=5):
    bad = np.nonzero(got.reshape(-1) != exp.reshape(-1))[0]
    return "mismatching bytes: %d, first at %s got %s exp %s" % (bad.size, bad[:limit], got.reshape(-1)[bad[:limit]], exp.reshape(-1)[bad[:limit]])


def _load(ctx, oracle, text):
    cube = oracle.Cube.parse(text)
    sc, of = cube.domain
    ctx.colorlut_load(cube.is3d, cube.size, cube.table, sc, of)
    return cube


def _device_lut(ctx, frames, w, h, in_place=False):
    """mi355_colorlut_frames_device on host frames (n, h, w*4) -> host result."""
    n = frames.shape[0]
    src = np.ascontiguousarray(frames).reshape(-1)
    out = np.full(src.size, 0x5A, np.uint8)
    d_src = ctx.alloc(src.nbytes)
    d_dst = d_src if in_place else ctx.alloc(out.nbytes)
    try:
        ctx.h2d(d_src, src)
        if not in_place:
            ctx.h2d(d_dst, out)
        ctx.colorlut_frames_device(d_src, h * w * 4, w * 4, d_dst, h * w * 4, w * 4, n, w, h, "RGBA")
        ctx.synchronize()
        ctx.d2h(out, d_dst)
    finally:
        ctx.free(d_src)
        if not in_place:
            ctx.free(d_dst)
    return out


def _varying_alpha(ac):
    ac = ac.copy()
    ac.reshape(-1, 4)[:, 3] = (np.arange(ac.size // 4, dtype=np.uint32) * 2654435761 >> 13).astype(np.uint8)
    return ac


@pytest.mark.parametrize("sets", [32, 48, 64])
@pytest.mark.parametrize("size,domain", [(33, None), (2, None), (3, None), (17, None), (34, None), (65, None),
                                         (33, ((-0.25, 0.0, 0.1), (1.5, 1.0, 0.9))), (5, ((0.0, 0.0, 0.0), (2.0, 0.5, 1.0)))])
def test_brick_kernel_allcolors(ctx, oracle, synth, size, domain, sets):
    """Every 8-bit colour (varying alpha passes through). The all-colours frame touches every LUT cell and misses the
    wave caches constantly, so this drives the careful path (global brick reads, elected refills) as well as the fast one."""
    import mi355fx
    cube = _load(ctx, oracle, synth.cube_text_3d(size, amp=0.07, domain=domain))
    ac = _varying_alpha(synth.allcolors())
    exp = np.zeros_like(ac)
    oracle.colorlut_rgba8(cube, ac, 4096 * 4, exp, 4096 * 4, 4096, 4096, nthreads=8)
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 7)
    ctx.set_flag(mi355fx.FLAG_BRICK_SETS, sets)
    got = np.zeros_like(ac)
    ctx.colorlut_frame(ac, 4096 * 4, got, 4096 * 4, 4096, 4096, "RGBA")
    assert ctx.colorlut_kernel_name().startswith("colorlut3d_brick_kernel")
    assert (got == exp).all(), _report(got, exp)


def test_brick_kernel_nonfinite_lut_entries(ctx, oracle):
    """inf / nan / huge LUT entries: the brick differences and lerps are the reference's own IEEE operations, so the
    brick kernel takes such tables too (the three-pass kernel does not)."""
    import mi355fx
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
    from mi355fx import synth
    ac = synth.allcolors()
    exp = np.zeros_like(ac)
    oracle.colorlut_rgba8(cube, ac, 4096 * 4, exp, 4096 * 4, 4096, 4096, nthreads=8)
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 7)
    got = np.zeros_like(ac)
    ctx.colorlut_frame(ac, 4096 * 4, got, 4096 * 4, 4096, 4096, "RGBA")
    assert ctx.colorlut_kernel_name() == "colorlut3d_brick_kernel"
    assert (got == exp).all(), _report(got, exp)


@pytest.mark.parametrize("sets", [32, 48, 64])
@pytest.mark.parametrize("content", ["smooth", "noise"])
@pytest.mark.parametrize("in_place", [False, True])
def test_brick_kernel_4k_batch(ctx, oracle, synth, content, in_place, sets):
    """Full-size batch (3 x 3840x2160) through the device entry point: output == T[input] with T from the oracle."""
    import mi355fx
    cube = _load(ctx, oracle, synth.cube_text_3d(33))
    gen = synth.smooth_frame if content == "smooth" else synth.noise_frame
    frames = np.stack([gen(3840, 2160, seed=synth.SEED + 3 + i) for i in range(3)])
    ac = synth.allcolors()
    table = np.zeros_like(ac)
    oracle.colorlut_rgba8(cube, ac, 4096 * 4, table, 4096 * 4, 4096, 4096, nthreads=8)
    t = table.reshape(-1, 4)
    px = frames.reshape(-1, 4)
    idx = px[:, 0].astype(np.uint32) | (px[:, 1].astype(np.uint32) << 8) | (px[:, 2].astype(np.uint32) << 16)
    exp = t[idx].copy()
    exp[:, 3] = px[:, 3]
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 7)
    ctx.set_flag(mi355fx.FLAG_BRICK_SETS, sets)
    got = _device_lut(ctx, frames, 3840, 2160, in_place=in_place)
    assert ctx.colorlut_kernel_name().startswith("colorlut3d_brick_kernel")
    assert (got.reshape(-1, 4) == exp).all(), _report(got, exp)
    steps, slow, _, _ = ctx.colorlut_brick_stats()
    total_steps = frames.size // 4 // 256
    if content == "smooth":
        assert steps < 0.3 * total_steps, "natural-like frames should mostly hit the wave caches (%d of %d steps missed)" % (steps, total_steps)
        assert slow < 0.05 * total_steps, "%d of %d steps on the slow path" % (slow, total_steps)
    else:
        assert slow > 0.9 * total_steps


@pytest.mark.parametrize("w,h,n", [(4, 1, 1), (8, 3, 2), (100, 37, 1), (128, 4, 1), (132, 5, 3), (516, 3, 1), (1920, 1081, 1), (3840, 7, 2), (1000, 9, 1), (260, 17, 5)])
def test_brick_kernel_ragged_geometry(ctx, oracle, synth, w, h, n):
    """Widths that are not whole 128-pixel strips, heights that are not whole tiles, several frames per launch."""
    import mi355fx
    cube = _load(ctx, oracle, synth.cube_text_3d(17, amp=0.1))
    rng = np.random.default_rng(w * 1000 + h)
    base = synth.smooth_frame(max(w, 16), max(h, 4))[:h, : w * 4]
    frames = np.stack([np.roll(base, 4 * i, axis=1) for i in range(n)]).copy()
    frames[:, ::3, ::7] = rng.integers(0, 256, size=frames[:, ::3, ::7].shape, dtype=np.uint8)  # a few outliers: cache misses inside tiles
    exp = np.zeros_like(frames)
    for i in range(n):
        oracle.colorlut_rgba8(cube, frames[i], w * 4, exp[i], w * 4, w, h)
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 7)
    got = _device_lut(ctx, frames, w, h).reshape(frames.shape)
    assert ctx.colorlut_kernel_name() == "colorlut3d_brick_kernel"
    assert (got == exp).all(), _report(got, exp)


@pytest.mark.parametrize("sets", [32, 64])
@pytest.mark.parametrize("share,tpr", [(0, 0), (1, 0), (2, 0), (3, 0), (3, 1), (3, 3), (3, 11), (3, 12), (2, 500), (0, 2)])
def test_brick_kernel_work_sharing_modes(ctx, oracle, synth, share, tpr, sets):
    """The waves of a block share their runs (tile deques in LDS: FLAG_BRICK_PRIO bit 1 = stealing, bit 0 = progress-based
    priorities). Every tile must be worked on exactly once whatever the mode and the run length: runs of one tile (every
    wave lives on stolen tiles after its first), runs longer than the picture (one run per strip, the other waves of the
    block steal all their work), frames of very different difficulty per strip (left half smooth, right half noise)."""
    import mi355fx
    cube = _load(ctx, oracle, synth.cube_text_3d(33, amp=0.05))
    w, h, n = 1280, 203, 3
    base = synth.smooth_frame(w, h)
    frames = np.stack([np.roll(base, 4 * 11 * i, axis=1) for i in range(n)]).copy()
    frames[:, :, w * 2:] = synth.noise_frame(w, h, seed=3)[:, w * 2:]
    exp = np.zeros_like(frames)
    for i in range(n):
        oracle.colorlut_rgba8(cube, frames[i], w * 4, exp[i], w * 4, w, h)
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 7)
    ctx.set_flag(mi355fx.FLAG_BRICK_SETS, sets)
    ctx.set_flag(mi355fx.FLAG_BRICK_PRIO, share)
    ctx.set_flag(mi355fx.FLAG_BRICK_TILES_PER_RUN, tpr)
    try:
        for in_place in (False, True):
            got = _device_lut(ctx, frames, w, h, in_place=in_place).reshape(frames.shape)
            assert (got == exp).all(), _report(got, exp)
    finally:
        ctx.set_flag(mi355fx.FLAG_BRICK_PRIO, 3)
        ctx.set_flag(mi355fx.FLAG_BRICK_TILES_PER_RUN, 0)
        ctx.set_flag(mi355fx.FLAG_BRICK_SETS, 0)


def test_brick_kernels_of_concurrent_contexts(mi355lib, oracle, synth):
    """Four contexts on four host threads, each with its own HIP stream, LUT and frames, launching the brick kernel (both
    geometries in turn) at the same time: a brick block takes a whole CU, so the kernels of different streams interleave
    block by block. Every output must still equal the oracle's (nothing is shared between the contexts' caches, deques and
    counters)."""
    import threading
    import mi355fx
    w, h, n = 1920, 1080, 2
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
            c.set_flag(mi355fx.FLAG_LUT_VARIANT, 7)
            barrier.wait()
            for it in range(6):
                c.set_flag(mi355fx.FLAG_BRICK_SETS, 64 if (it + k) % 2 else 32)
                got = _device_lut(c, frames, w, h).reshape(frames.shape)
                if not (got == exp).all():
                    errors.append("context %d launch %d: %s" % (k, it, _report(got, exp)))
        except Exception as e:  # noqa: BLE001 - reported below
            errors.append("context %d: %r" % (k, e))
        finally:
            c.close()

    th = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in th: t.start()
    for t in th: t.join()
    assert not errors, errors


def test_brick_kernel_width_not_multiple_of_4_takes_another_kernel(ctx, oracle, synth):
    import mi355fx
    cube = _load(ctx, oracle, synth.cube_text_3d(33))
    w, h = 37, 5
    frame = synth.smooth_frame(64, 8)[:h, : w * 4].copy()
    exp = np.zeros_like(frame)
    oracle.colorlut_rgba8(cube, frame, w * 4, exp, w * 4, w, h)
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 7)
    got = np.zeros_like(frame)
    ctx.colorlut_frame(frame, w * 4, got, w * 4, w, h, "RGBA")
    assert (got == exp).all(), _report(got, exp)
    assert ctx.colorlut_kernel_name() != "colorlut3d_brick_kernel"


@pytest.mark.parametrize("setting", ["defaults", "hue90", "mixed", "neg", "nonfinite"])
def test_brick_kernel_fused_chain_allcolors(ctx, oracle, synth, setting):
    """hsvfilter -> colorlut in one launch of the brick kernel == oracle hsvfilter then oracle colorlut, every colour."""
    import mi355fx
    st = {"neg": (-77.5, 1.0, 0.0, 1.0, 0.0), "nonfinite": (float("inf"), 1.3, -0.1, 0.9, 0.05)}.get(setting) or synth.HSV_SETTINGS[setting]
    cube = _load(ctx, oracle, synth.cube_text_3d(33))
    ac = _varying_alpha(synth.allcolors())
    mid = ac.copy().reshape(-1)
    oracle.hsvfilter(mid, 4096, 4096 * 4, 4, 0, False, st, nthreads=8)
    exp = np.zeros_like(mid)
    oracle.colorlut_rgba8(cube, mid, 4096 * 4, exp, 4096 * 4, 4096, 4096, nthreads=8)
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 7)
    src = ac.reshape(-1)
    out = np.zeros_like(src)
    d_src, d_dst = ctx.alloc(src.nbytes), ctx.alloc(src.nbytes)
    try:
        ctx.h2d(d_src, src)
        ctx.hsv_colorlut_frames_device(d_src, 4096 * 4096 * 4, 4096 * 4, d_dst, 4096 * 4096 * 4, 4096 * 4, 1, 4096, 4096, st)
        ctx.synchronize()
        ctx.d2h(out, d_dst)
    finally:
        ctx.free(d_src); ctx.free(d_dst)
    assert ctx.colorlut_kernel_name() == "colorlut3d_brick_kernel<HSV>"
    assert (out == exp).all(), _report(out, exp)


def test_content_watch_hands_noise_to_the_three_pass_kernel_and_back(ctx, oracle, synth):
    """MI355_FLAG_LUT_VARIANT 6 (interpolating kernels only): natural-like frames stay on the brick kernel; a stream that
    turns into noise is handed to the three-pass kernel within a few launches (no host wait involved), and comes back when
    the content calms down (the brick kernel is re-tried after the probation period). Every output stays exact."""
    import mi355fx
    cube = _load(ctx, oracle, synth.cube_text_3d(33))
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 6)
    w, h = 1920, 1080
    ac = synth.allcolors()
    table = np.zeros_like(ac)
    oracle.colorlut_rgba8(cube, ac, 4096 * 4, table, 4096 * 4, 4096, 4096, nthreads=8)
    t = table.reshape(-1, 4)

    def expect(frame):
        px = frame.reshape(-1, 4)
        idx = px[:, 0].astype(np.uint32) | (px[:, 1].astype(np.uint32) << 8) | (px[:, 2].astype(np.uint32) << 16)
        e = t[idx].copy()
        e[:, 3] = px[:, 3]
        return e

    smooth = synth.smooth_frame(w, h)[None]
    noise = synth.noise_frame(w, h)[None]
    es, en = expect(smooth), expect(noise)
    names = []
    for _ in range(10):
        got = _device_lut(ctx, smooth, w, h)
        assert (got.reshape(-1, 4) == es).all()
        names.append(ctx.colorlut_kernel_name())
    assert all(n.startswith("colorlut3d_brick_kernel") for n in names), names   # 32 or 64 sets: the watch's call
    names = []
    for _ in range(20):
        got = _device_lut(ctx, noise, w, h)
        assert (got.reshape(-1, 4) == en).all()
        names.append(ctx.colorlut_kernel_name())
    assert names[-1] == "colorlut3d_lds_kernel", names
    assert "colorlut3d_brick_kernel (64 sets)" in names   # the ladder goes through the 64-set cache first
    names = []
    for _ in range(170):  # two probations (64 sets, then 32 sets), each after 64 launches of the level above
        got = _device_lut(ctx, smooth, w, h)
        assert (got.reshape(-1, 4) == es).all()
        names.append(ctx.colorlut_kernel_name())
    assert names[-1].startswith("colorlut3d_brick_kernel"), names[-10:]
