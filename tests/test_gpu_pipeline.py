"""Asynchronous host-buffer pipeline (mi355_pipe_*) and pinned host memory: results identical to the synchronous
entry points / the oracle, in any submit/wait order the ring allows."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _load_cube(ctx, oracle, text):
    cube = oracle.Cube.parse(text)
    sc, of = cube.domain
    ctx.colorlut_load(cube.is3d, cube.size, cube.table, sc, of)
    return cube


def test_pipeline_chain_many_frames_pinned(ctx, oracle, synth):
    """12 different 1080p frames through hsvfilter!colorlut with a depth-3 ring and pinned buffers; every output
    equals the oracle chain; inputs are left untouched."""
    w, h, n = 1920, 1080, 12
    st = synth.HSV_SETTINGS["mixed"]
    cube = _load_cube(ctx, oracle, synth.cube_text_3d(33))
    pipe = ctx.pipe_create(3, w * h * 4)
    srcs = [ctx.host_array(w * h * 4) for _ in range(n)]
    dsts = [ctx.host_array(w * h * 4) for _ in range(n)]
    try:
        for k in range(n):
            srcs[k][:] = (synth.noise_frame(w, h, seed=50 + k) if k % 2 else synth.smooth_frame(w, h, seed=50 + k)).reshape(-1)
            dsts[k][:] = 0
        keep = [s.copy() for s in srcs]
        tickets = [ctx.pipe_submit_hsv_colorlut(pipe, srcs[k], w * 4, dsts[k], w * 4, w, h, st) for k in range(n)]
        assert tickets == list(range(1, n + 1))
        for t in reversed(tickets):      # any order; early tickets were already reclaimed by back-pressure
            ctx.pipe_wait(pipe, t)
        for k in range(n):
            mid = keep[k].copy()
            oracle.hsvfilter(mid, w, w * 4, 4, 0, False, st, nthreads=8)
            exp = np.zeros_like(mid)
            oracle.colorlut_rgba8(cube, mid, w * 4, exp, w * 4, w, h, nthreads=8)
            assert (dsts[k] == exp).all(), k
            assert (srcs[k] == keep[k]).all()
    finally:
        ctx.pipe_destroy(pipe)
        for a in srcs + dsts:
            ctx.host_free(a)


def test_pipeline_elements_match_sync_entry_points(ctx, oracle, synth):
    """hsvfilter in place (BGRx, padded stride) and colorlut (RGBA64) through the pipeline == synchronous calls;
    pageable numpy buffers work too."""
    w, h, pad = 333, 77, 20
    stride = w * 4 + pad
    st = synth.HSV_SETTINGS["hue90"]
    rng = np.random.default_rng(4)
    frame = rng.integers(0, 256, (h, stride), dtype=np.uint8)
    sync = frame.copy().reshape(-1)
    ctx.hsvfilter_frame_ip(sync, w, stride, "BGRx", st)
    pipe = ctx.pipe_create(2, 1 << 22)
    try:
        a = frame.copy().reshape(-1)
        b = frame.copy().reshape(-1)
        ta = ctx.pipe_submit_hsvfilter(pipe, a, w, stride, "BGRx", st)
        tb = ctx.pipe_submit_hsvfilter(pipe, b, w, stride, "BGRx", st)
        ctx.pipe_wait(pipe, ta); ctx.pipe_wait(pipe, tb)
        assert (a == sync).all() and (b == sync).all()       # padding bytes untouched as well
        cube = _load_cube(ctx, oracle, synth.cube_text_3d(17))
        src = rng.integers(0, 65536, (h, w * 4), dtype=np.uint16).view(np.uint8).reshape(h, w * 8)
        exp = np.zeros_like(src)
        ctx.colorlut_frame(src, w * 8, exp, w * 8, w, h, "RGBA64_LE")
        got = np.zeros_like(src)
        t = ctx.pipe_submit_colorlut(pipe, src, w * 8, got, w * 8, w, h, "RGBA64_LE")
        ctx.pipe_wait_all(pipe)
        assert (got == exp).all()
        ctx.pipe_wait(pipe, t)   # waiting twice is fine
    finally:
        ctx.pipe_destroy(pipe)


def test_pipeline_errors(ctx, synth):
    import mi355fx
    ctx.colorlut_unload()
    pipe = ctx.pipe_create(2, 4096)
    try:
        buf = np.zeros(64 * 64 * 4, np.uint8)
        with pytest.raises(mi355fx.Mi355Error) as e:      # frame larger than the slots
            ctx.pipe_submit_hsvfilter(pipe, buf, 64, 256, "RGBA", synth.HSV_SETTINGS["defaults"])
        assert e.value.status == mi355fx.ERR_INVALID_ARG
        small = np.zeros(16 * 16 * 4, np.uint8)
        with pytest.raises(mi355fx.Mi355Error) as e:      # "No LUT configured"
            ctx.pipe_submit_colorlut(pipe, small, 64, small.copy(), 64, 16, 16, "RGBA")
        assert e.value.status == mi355fx.ERR_NOT_CONFIGURED
        with pytest.raises(mi355fx.Mi355Error):
            ctx.pipe_wait(pipe, 99)
    finally:
        ctx.pipe_destroy(pipe)
    with pytest.raises(mi355fx.Mi355Error):
        ctx.pipe_create(0, 4096)
