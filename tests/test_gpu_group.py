"""GPU parity of the multi-stream dispatcher (csrc/group.hip, mi355_group_*): frames of many streams batched into multi-frame
launches give every stream exactly what its own two element launches give it (hsvfilter in place,
video/hsv/src/hsvfilter/imp.rs:323-376, then colorlut, video/colorlut/src/colorlut/imp.rs:203-223) - bit for bit, in order,
whatever mix of sizes, settings, LUTs and formats the streams submit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctxs(mi355fx, oracle, synth, n, texts=None):
    out = []
    for i in range(n):
        c = mi355fx.Context(0)
        cube = oracle.Cube.parse((texts[i] if texts else synth.cube_text_3d(33)))
        sc, of = cube.domain
        c.colorlut_load(cube.is3d, cube.size, cube.table, sc, of)
        out.append((c, cube))
    return out


def _expect(oracle, cube, frame, w, h, st, stride=None, ps=4, fmt_first=0, bgr=False):
    stride = stride or w * ps
    mid = frame.copy()
    oracle.hsvfilter(mid, w, stride, ps, fmt_first, bgr, st, nthreads=8)
    out = np.zeros_like(mid)
    oracle.colorlut_rgba8(cube, mid, stride, out, stride, w, h, nthreads=8)
    return mid, out


def test_round_of_streams_equals_the_two_calls_per_stream(oracle, synth, mi355lib):
    """12 streams, one 1080p frame each per round, three rounds: the group's batched launches == mi355_issue_streams_round ==
    the oracle chain; the source frames end up hsv-filtered in place exactly as the element leaves them."""
    import mi355fx
    w, h, n = 1920, 1080, 12
    st = synth.HSV_SETTINGS["hue90"]
    pairs = _ctxs(mi355fx, oracle, synth, n)
    ctxs = [p[0] for p in pairs]
    g = mi355fx.Group(0)
    fb = w * h * 4
    d_src, d_dst, d_src2, d_dst2 = [], [], [], []
    try:
        frames = [synth.smooth_frame(w, h, seed=500 + i).reshape(-1) if i % 3 else synth.noise_frame(w, h, seed=500 + i).reshape(-1) for i in range(n)]
        for i, c in enumerate(ctxs):
            for lst in (d_src, d_dst, d_src2, d_dst2):
                lst.append(c.alloc(fb))
        for rnd in range(3):
            for i, c in enumerate(ctxs):
                c.h2d(d_src[i], frames[i])
                c.h2d(d_src2[i], frames[i])
                c.synchronize()
            g.submit_round(ctxs, d_src, d_dst, w, h, w * 4, "RGBA", st)
            g.wait_all()
            mi355fx.StreamsRound(ctxs, w, h, w * 4, "RGBA", st).issue(d_src2, d_dst2)
            for i, c in enumerate(ctxs):
                c.synchronize()
                a, b, s_a, s_b = (np.zeros(fb, np.uint8) for _ in range(4))
                c.d2h(a, d_dst[i]); c.d2h(b, d_dst2[i]); c.d2h(s_a, d_src[i]); c.d2h(s_b, d_src2[i])
                assert (a == b).all() and (s_a == s_b).all(), (rnd, i)
                if rnd == 0 and i < 3:
                    mid, exp = _expect(oracle, pairs[i][1], frames[i], w, h, st)
                    assert (a == exp).all() and (s_a == mid).all(), i
        frames_n, batched, single = g.stats()
        assert frames_n == 3 * n and single == 0 and batched == 3 * 2   # 12 frames = one launch pair of 8 + one of 4, per round
    finally:
        g.close()
        for i, c in enumerate(ctxs):
            for lst in (d_src, d_dst, d_src2, d_dst2):
                if i < len(lst):
                    c.free(lst[i])
            c.close()


def test_mixed_streams_stay_exact_and_in_order(oracle, synth, mi355lib):
    """Streams that do not agree: two sizes, two hsv settings, two LUTs, a padded-rows stream (not the batched kernels'
    business), a BGRx frame (refused at submit: the chain exists for RGBA), and one stream that submits three dependent frames in a row (frame k+1 reads what frame k
    wrote: order within a stream). Every result is the oracle's."""
    import mi355fx
    texts = [synth.cube_text_3d(33), synth.cube_text_3d(33), synth.cube_text_3d(17, amp=0.08), synth.cube_text_3d(33), synth.cube_text_3d(33), synth.cube_text_3d(33)]
    pairs = _ctxs(mi355fx, oracle, synth, 6, texts)
    ctxs = [p[0] for p in pairs]
    g = mi355fx.Group(0, max_batch=4)
    st_a, st_b = synth.HSV_SETTINGS["hue90"], synth.HSV_SETTINGS["mixed"]
    bufs, jobs = [], []   # jobs: (ctx index, d_src, d_dst, host frame, w, h, stride, fmt, settings, ticket)
    try:
        def job(i, frame, w, h, stride, fmt, st):
            c = ctxs[i]
            ds, dd = c.alloc(frame.nbytes), c.alloc(frame.nbytes)
            bufs.extend([(c, ds), (c, dd)])
            c.h2d(ds, frame)
            c.h2d(dd, np.full(frame.nbytes, 0xEE, np.uint8))
            t = g.submit_chain(c, ds, dd, w, h, stride, fmt, st)
            jobs.append([i, ds, dd, frame, w, h, stride, fmt, st, t])
        job(0, synth.smooth_frame(1920, 1080, seed=1).reshape(-1), 1920, 1080, 1920 * 4, "RGBA", st_a)
        job(1, synth.smooth_frame(1920, 1080, seed=2).reshape(-1), 1920, 1080, 1920 * 4, "RGBA", st_a)
        job(2, synth.smooth_frame(1920, 1080, seed=3).reshape(-1), 1920, 1080, 1920 * 4, "RGBA", st_a)      # another LUT: its own launch
        job(3, synth.smooth_frame(1280, 720, seed=4).reshape(-1), 1280, 720, 1280 * 4, "RGBA", st_a)        # another size
        job(1, synth.noise_frame(1920, 1080, seed=5).reshape(-1), 1920, 1080, 1920 * 4, "RGBA", st_b)        # same stream, other settings: after its first frame
        job(0, synth.noise_frame(1920, 1080, seed=6).reshape(-1), 1920, 1080, 1920 * 4, "RGBA", st_a)
        rng = np.random.default_rng(9)
        pad = rng.integers(0, 256, size=(360, 640 * 4 + 64), dtype=np.uint8)
        pad[:, :640 * 4] = synth.smooth_frame(640, 360, seed=7).reshape(360, 640 * 4)
        job(4, pad.reshape(-1).copy(), 640, 360, 640 * 4 + 64, "RGBA", st_a)                                 # padded rows: context's own path
        job(5, synth.smooth_frame(1920, 1080, seed=8).reshape(-1), 1920, 1080, 1920 * 4, "RGBA", st_a)
        with pytest.raises(mi355fx.Mi355Error):   # not queued, nobody else's batch fails for it
            g.submit_chain(ctxs[5], jobs[-1][1], jobs[-1][2], 1920, 1080, 1920 * 4, "BGRx", st_a)
        # a stream whose frames depend on each other: dst of frame k is src of frame k + 1
        c = ctxs[5]
        f0 = synth.smooth_frame(1280, 720, seed=9).reshape(-1)
        chain_bufs = [c.alloc(f0.nbytes) for _ in range(4)]
        bufs.extend((c, b) for b in chain_bufs)
        c.h2d(chain_bufs[0], f0)
        chain_tickets = [g.submit_chain(c, chain_bufs[k], chain_bufs[k + 1], 1280, 720, 1280 * 4, "RGBA", st_b) for k in range(3)]
        g.wait(chain_tickets[-1])
        g.wait_all()
        for i, ds, dd, frame, w, h, stride, fmt, st, t in jobs:
            g.wait(t)   # (already done: must be a no-op)
            got, src_after = np.zeros(frame.nbytes, np.uint8), np.zeros(frame.nbytes, np.uint8)
            ctxs[i].d2h(got, dd)
            ctxs[i].d2h(src_after, ds)
            mid, exp = _expect(oracle, pairs[i][1], frame, w, h, st, stride=stride)
            rows = lambda a: a.reshape(h, stride)[:, :w * 4]
            assert (rows(got) == rows(exp)).all(), (i, w, h, stride)
            assert (rows(src_after) == rows(mid)).all(), (i, "source after hsvfilter")
            if stride != w * 4:
                assert (got.reshape(h, stride)[:, w * 4:] == 0xEE).all()   # padding of the destination untouched
        cur = f0
        for k in range(3):
            _, cur = _expect(oracle, pairs[5][1], cur, 1280, 720, st_b)
        got = np.zeros(f0.nbytes, np.uint8)
        c.d2h(got, chain_bufs[3])
        assert (got == cur).all(), "three dependent frames of one stream ran out of order"
        frames_n, batched, single = g.stats()
        assert frames_n == len(jobs) + 3 and single == 1 and batched >= 6
    finally:
        g.close()
        for c, b in bufs:
            c.free(b)
        for c in ctxs:
            c.close()


def test_one_frame_deep_streams_fill_batches(oracle, synth, mi355lib):
    """The way an element uses it (gst/gstcolorlut.c is one frame deep): every stream submits frame n and then waits for its
    frame n-1. Nobody ever calls flush - the first wait of a round launches what all streams have submitted since."""
    import mi355fx
    w, h, n, rounds = 1280, 720, 6, 5
    st = synth.HSV_SETTINGS["hue90"]
    pairs = _ctxs(mi355fx, oracle, synth, n)
    ctxs = [p[0] for p in pairs]
    g = mi355fx.Group(0)
    fb = w * h * 4
    try:
        frames = [[synth.smooth_frame(w, h, seed=700 + 10 * r + i).reshape(-1) for i in range(n)] for r in range(rounds)]
        d_src = [[c.alloc(fb) for c in ctxs] for _ in range(rounds)]
        d_dst = [[c.alloc(fb) for c in ctxs] for _ in range(rounds)]
        tickets = [[0] * n for _ in range(rounds)]
        for r in range(rounds):
            for i, c in enumerate(ctxs):
                c.h2d(d_src[r][i], frames[r][i])
                tickets[r][i] = g.submit_chain(c, d_src[r][i], d_dst[r][i], w, h, w * 4, "RGBA", st)
                if r:
                    if i % 2:
                        g.wait(tickets[r - 1][i])                 # host wait ...
                    else:
                        g.order_after(c, tickets[r - 1][i])       # ... or the stream's own download ordered behind the frame
                    got = np.zeros(fb, np.uint8)
                    c.d2h(got, d_dst[r - 1][i])
                    _, exp = _expect(oracle, pairs[i][1], frames[r - 1][i], w, h, st)
                    assert (got == exp).all(), (r - 1, i)
        g.wait_all()
        frames_n, batched, single = g.stats()
        assert frames_n == rounds * n and single == 0
        assert batched <= rounds + 1, batched   # one launch pair per round (the first wait of round r flushes round r-1's leftovers + what round r has so far)
    finally:
        g.close()
        for r in range(rounds):
            for i, c in enumerate(ctxs):
                c.free(d_src[r][i]); c.free(d_dst[r][i])
        for c in ctxs:
            c.close()


def test_host_buffer_pipelines_share_launches_through_a_group(oracle, synth, mi355lib):
    """mi355_pipe_set_group: four host-buffer pipelines (pinned upload -> kernels -> download, one frame deep like
    gst/gstcolorlut.c) hand their frames to one group (as the fused pair); every output is the oracle chain's, inputs untouched, and
    once the settings have stayed for eight frames the streams' frames share launches."""
    import mi355fx
    w, h, n, rounds = 1920, 1080, 4, 12   # (a stream's first seven frames with new settings take its own fused path, §mi355_group_submit_fused)
    st = synth.HSV_SETTINGS["mixed"]
    pairs = _ctxs(mi355fx, oracle, synth, n)
    ctxs = [p[0] for p in pairs]
    g = mi355fx.Group(0)
    pipes = [c.pipe_create(3, w * h * 4) for c in ctxs]
    srcs = [[c.host_array(w * h * 4) for _ in range(rounds)] for c in ctxs]
    dsts = [[c.host_array(w * h * 4) for _ in range(rounds)] for c in ctxs]
    try:
        for i, c in enumerate(ctxs):
            c.pipe_set_group(pipes[i], g)
            for r in range(rounds):
                srcs[i][r][:] = (synth.noise_frame(w, h, seed=900 + 10 * r + i) if (r + i) % 2 else synth.smooth_frame(w, h, seed=900 + 10 * r + i)).reshape(-1)
                dsts[i][r][:] = 0
        keep = [[a.copy() for a in row] for row in srcs]
        tickets = [[0] * rounds for _ in range(n)]
        for r in range(rounds):
            for i, c in enumerate(ctxs):
                tickets[i][r] = c.pipe_submit_hsv_colorlut(pipes[i], srcs[i][r], w * 4, dsts[i][r], w * 4, w, h, st)
                if r:
                    c.pipe_wait(pipes[i], tickets[i][r - 1])
        for i, c in enumerate(ctxs):
            c.pipe_wait_all(pipes[i])
        for i in range(n):
            for r in range(rounds):
                _, exp = _expect(oracle, pairs[i][1], keep[i][r], w, h, st)
                assert (dsts[i][r] == exp).all(), (i, r)
                assert (srcs[i][r] == keep[i][r]).all()
        frames_n, batched, single = g.stats()
        assert frames_n == n * rounds and single == 7 * n and batched < n * (rounds - 7), (frames_n, batched, single)
        ctxs[0].pipe_set_group(pipes[0], None)   # and back to its own launches
        t = ctxs[0].pipe_submit_hsv_colorlut(pipes[0], srcs[0][0], w * 4, dsts[0][1], w * 4, w, h, st)
        ctxs[0].pipe_wait(pipes[0], t)
        assert (dsts[0][1] == dsts[0][0]).all() and g.stats()[0] == n * rounds
    finally:
        for i, c in enumerate(ctxs):
            c.pipe_destroy(pipes[i])
            for a in srcs[i] + dsts[i]:
                c.host_free(a)
        g.close()
        for c in ctxs:
            c.close()


def test_fused_frames_share_one_launch_and_leave_their_sources_alone(oracle, synth, mi355lib):
    """mi355_group_submit_fused: frames of streams that agree in size, LUT and hsv settings go through ONE launch (the composed
    table of those settings); a stream with other settings, one with another LUT, one in place, one with padded rows (its
    context's own fused path) and a second frame of a stream all get the oracle chain's bytes, and no source is modified."""
    import mi355fx
    texts = [synth.cube_text_3d(33)] * 5 + [synth.cube_text_3d(17, amp=0.08)] + [synth.cube_text_3d(33)] * 2
    pairs = _ctxs(mi355fx, oracle, synth, 8, texts)
    ctxs = [p[0] for p in pairs]
    g = mi355fx.Group(0)
    st_a, st_b = synth.HSV_SETTINGS["hue90"], synth.HSV_SETTINGS["mixed"]
    w, h = 1920, 1080
    bufs, jobs = [], []
    try:
        def job(i, frame, w_, h_, stride, st, in_place=False):
            c = ctxs[i]
            ds = c.alloc(frame.nbytes)
            dd = ds if in_place else c.alloc(frame.nbytes)
            bufs.append((c, ds))
            if not in_place:
                bufs.append((c, dd))
                c.h2d(dd, np.full(frame.nbytes, 0xEE, np.uint8))
            c.h2d(ds, frame)
            t = g.submit_fused(c, ds, dd, w_, h_, stride, "RGBA", st)
            jobs.append((i, ds, dd, frame, w_, h_, stride, st, t, in_place))
        # a composed table is built for settings that STAY: eight submits in a row with them (until then a frame takes its
        # context's own fused path) - the streams have been running for a while
        warm = {}
        for i, st in [(0, st_a), (1, st_a), (2, st_a), (3, st_a), (4, st_b), (5, st_a), (6, st_a)]:
            c = ctxs[i]
            ws, wd = c.alloc(w * h * 4), c.alloc(w * h * 4)
            bufs.extend([(c, ws), (c, wd)])
            c.h2d(ws, synth.smooth_frame(w, h, seed=90 + i).reshape(-1))
            for _ in range(8):
                g.submit_fused(c, ws, wd, w, h, w * 4, "RGBA", st)
        g.wait_all()
        warm_frames, warm_batched, warm_single = g.stats()
        assert warm_single == 7 * 7, (warm_frames, warm_batched, warm_single)   # seven submits of each stream before its settings count as settled
        for i in range(4):                                                                         # four streams: one launch
            job(i, synth.smooth_frame(w, h, seed=20 + i).reshape(-1), w, h, w * 4, st_a)
        job(4, synth.smooth_frame(w, h, seed=30).reshape(-1), w, h, w * 4, st_b)                  # other settings: another table
        job(5, synth.smooth_frame(w, h, seed=31).reshape(-1), w, h, w * 4, st_a)                  # another LUT
        job(6, synth.noise_frame(w, h, seed=32).reshape(-1), w, h, w * 4, st_a, in_place=True)    # in place: allowed for the fused form
        rng = np.random.default_rng(3)
        pad = rng.integers(0, 256, size=(360, 640 * 4 + 64), dtype=np.uint8)
        pad[:, :640 * 4] = synth.smooth_frame(640, 360, seed=33).reshape(360, 640 * 4)
        job(7, pad.reshape(-1).copy(), 640, 360, 640 * 4 + 64, st_a)                              # padded rows: own path
        job(0, synth.noise_frame(w, h, seed=34).reshape(-1), w, h, w * 4, st_a)                   # second frame of stream 0: next batch
        g.wait_all()
        for i, ds, dd, frame, w_, h_, stride, st, t, in_place in jobs:
            g.wait(t)
            got = np.zeros(frame.nbytes, np.uint8)
            ctxs[i].d2h(got, dd)
            _, exp = _expect(oracle, pairs[i][1], frame, w_, h_, st, stride=stride)
            rows = lambda a: a.reshape(h_, stride)[:, :w_ * 4]
            assert (rows(got) == rows(exp)).all(), (i, w_, h_, stride)
            if not in_place:
                src_after = np.zeros(frame.nbytes, np.uint8)
                ctxs[i].d2h(src_after, ds)
                assert (src_after == frame).all(), (i, "the fused form leaves its source alone")
                if stride != w_ * 4:
                    assert (got.reshape(h_, stride)[:, w_ * 4:] == 0xEE).all()
        frames_n, batched, single = g.stats()
        assert frames_n - warm_frames == len(jobs) and single - warm_single == 1 and batched - warm_batched <= 6, (frames_n, batched, single)
        # the same frames through mi355_hsv_colorlut_frames_device on the streams' own contexts: identical bytes
        for i, ds, dd, frame, w_, h_, stride, st, t, in_place in jobs[:5]:
            c = ctxs[i]
            d2 = c.alloc(frame.nbytes); bufs.append((c, d2))
            c.hsv_colorlut_frames_device(ds, frame.nbytes, stride, d2, frame.nbytes, stride, 1, w_, h_, st)
            c.synchronize()
            a, b = np.zeros(frame.nbytes, np.uint8), np.zeros(frame.nbytes, np.uint8)
            c.d2h(a, dd); c.d2h(b, d2)
            assert (a == b).all(), i
    finally:
        g.close()
        for c, b in bufs:
            c.free(b)
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("max_batch", [16, 8])
def test_settings_change_without_waiting_keeps_queued_frames_on_their_own_table(oracle, synth, mi355lib, max_batch):
    """one stream, fused submits with settings A (enough for A's composed table), then - without waiting for anything -
    with settings B until B's table is built. Frames of A are still queued (max_batch 16: never launched yet) or in flight (8)
    when the context moves on to B: they hold a reference to A's table, so B gets a table of its own instead of a rebuild in place,
    and every frame comes out with the settings it was submitted with."""
    import mi355fx
    (c, cube), = _ctxs(mi355fx, oracle, synth, 1)
    g = mi355fx.Group(0, max_batch)
    st_a, st_b = synth.HSV_SETTINGS["hue90"], synth.HSV_SETTINGS["mixed"]
    w, h = 1280, 720
    n_a, n_b = 12, 10
    frames = [synth.smooth_frame(w, h, seed=700 + k).reshape(-1) for k in range(n_a + n_b)]
    srcs, dsts, tickets = [], [], []
    try:
        for k, f in enumerate(frames):
            srcs.append(c.alloc(f.nbytes)); dsts.append(c.alloc(f.nbytes))
            c.h2d(srcs[k], f)
        c.synchronize()
        for k in range(n_a + n_b):   # no wait, no flush in between
            tickets.append(g.submit_fused(c, srcs[k], dsts[k], w, h, w * 4, "RGBA", st_a if k < n_a else st_b))
        g.wait_all()
        tables = mi355fx.load_library().mi355_shared_table_count()
        assert tables >= 1
        for k, f in enumerate(frames):
            g.wait(tickets[k])
            got = np.zeros(f.nbytes, np.uint8)
            c.d2h(got, dsts[k])
            _, exp = _expect(oracle, cube, f, w, h, st_a if k < n_a else st_b)
            assert (got == exp).all(), (max_batch, k, "A" if k < n_a else "B")
    finally:
        g.close()
        for p in srcs + dsts:
            c.free(p)
        c.close()


def test_pipeline_destroyed_while_its_frames_wait_in_the_group(oracle, synth, mi355lib):
    """mi355_pipe_destroy in group mode. Two pipelines hand frames to a group that has not launched them yet
    (max_batch 16, nobody waits); one pipeline is destroyed - its slot buffers are freed - before the group flushes. The destroy
    makes the group finish that pipeline's frames first; the other pipeline's frames come out exact afterwards and nothing
    touches freed memory (the run would fault or corrupt the survivor's output otherwise)."""
    import mi355fx
    w, h = 1280, 720
    st = synth.HSV_SETTINGS["hue90"]
    pairs = _ctxs(mi355fx, oracle, synth, 2)
    (a, cube_a), (b, cube_b) = pairs
    g = mi355fx.Group(0, 16)
    pa, pb = a.pipe_create(4, w * h * 4), b.pipe_create(4, w * h * 4)
    bufs = []
    try:
        a.pipe_set_group(pa, g); b.pipe_set_group(pb, g)
        fa = [a.host_array(w * h * 4) for _ in range(3)]; oa = [a.host_array(w * h * 4) for _ in range(3)]
        fb = [b.host_array(w * h * 4) for _ in range(3)]; ob = [b.host_array(w * h * 4) for _ in range(3)]
        bufs = [(a, x) for x in fa + oa] + [(b, x) for x in fb + ob]
        for k in range(3):
            fa[k][:] = synth.smooth_frame(w, h, seed=800 + k).reshape(-1)
            fb[k][:] = synth.smooth_frame(w, h, seed=810 + k).reshape(-1)
        tb = []
        for k in range(3):
            a.pipe_submit_hsv_colorlut(pa, fa[k], w * 4, oa[k], w * 4, w, h, st)
            tb.append(b.pipe_submit_hsv_colorlut(pb, fb[k], w * 4, ob[k], w * 4, w, h, st))
        a.pipe_destroy(pa)          # frames of `a` are still with the group
        pa = None
        for k in range(3):
            b.pipe_wait(pb, tb[k])
            _, exp = _expect(oracle, cube_b, np.array(fb[k]), w, h, st)
            assert (ob[k] == exp).all(), k
        g.wait_all()
    finally:
        if pa is not None:
            a.pipe_destroy(pa)
        b.pipe_destroy(pb)
        for c, x in bufs:
            c.host_free(x)
        g.close()
        a.close(); b.close()


def test_fused_batches_from_hbm_go_through_the_lds_cached_kernel(oracle, synth, mi355lib):
    """Round 6: the fused pair of many streams in one launch reads frames that come from HBM (each stream's own source): the
    multi-frame form of colorlut_window_kernel (rows of up to 16 separate frames, base pointers in the kernel arguments) serves
    them where the launch is large enough - the provenance rule of the single-buffer entry points. Five 4K streams with odd
    content (noise, two-colour bars, every stream different), twice (the second round on a warm cache), in and out of place:
    every frame == the oracle chain; the two-launch form of the same streams keeps the gather kernel (its input is on-die)."""
    import mi355fx
    w, h, n = 3840, 2160, 5
    pairs = _ctxs(mi355fx, oracle, synth, n, [synth.cube_text_3d(33)] * n)
    ctxs, cube = [p[0] for p in pairs], pairs[0][1]
    st = synth.HSV_SETTINGS["hue90"]
    g = mi355fx.Group(0, 8)
    rng = np.random.default_rng(12)
    bufs = []
    try:
        frames = []
        for i in range(n):
            f = synth.smooth_frame(w, h, seed=300 + i).reshape(h, w * 4).copy()
            if i == 1:
                f[:, : w * 2] = rng.integers(0, 256, size=(h, w * 2), dtype=np.uint8)          # half noise: misses, installs, past-cache gathers
            if i == 2:
                f.reshape(h, w, 4)[:, ::2, :3] = (255, 0, 128)
                f.reshape(h, w, 4)[:, 1::2, :3] = (127, 255, 0)                                # two colours 128 levels apart: one set, two ways
            frames.append(f.reshape(-1))
        exp = []
        for f in frames:
            mid = f.copy()
            oracle.hsvfilter(mid, w, w * 4, 4, 0, False, st)
            e = np.zeros_like(f)
            oracle.colorlut_rgba8(cube, mid, w * 4, e, w * 4, w, h, nthreads=8)
            exp.append(e)
        srcs, dsts = [], []
        for c, f in zip(ctxs, frames):
            ds, dd = c.alloc(f.nbytes), c.alloc(f.nbytes)
            bufs.extend([(c, ds), (c, dd)])
            c.h2d(ds, f)
            srcs.append(ds); dsts.append(dd)
        for _ in range(8):      # settings that stay: the composed table is built (or found) at the eighth submit
            for c, ds, dd in zip(ctxs, srcs, dsts):
                g.submit_fused(c, ds, dd, w, h, w * 4, "RGBA", st)
            g.wait_all()
        for rnd in range(2):
            for c, dd in zip(ctxs, dsts):
                c.h2d(dd, np.full(w * h * 4, 0x11 + rnd, np.uint8))
            tk = [g.submit_fused(c, ds, ds if (rnd == 1 and i == 4) else dd, w, h, w * 4, "RGBA", st) for i, (c, ds, dd) in enumerate(zip(ctxs, srcs, dsts))]
            for t in tk:
                g.wait(t)
            assert ctxs[0].colorlut_kernel_name() == "colorlut_window_kernel"
            for i, (c, ds, dd) in enumerate(zip(ctxs, srcs, dsts)):
                out = np.zeros(w * h * 4, np.uint8)
                c.d2h(out, ds if (rnd == 1 and i == 4) else dd)
                assert (out == exp[i]).all(), (rnd, i)
                if not (rnd == 1 and i == 4):
                    c.d2h(out, ds)
                    assert (out == frames[i]).all()          # the source is left alone
        # the two-launch form of the same streams: hsvfilter's output is on-die -> the gather kernel
        c4 = ctxs[4]
        c4.h2d(srcs[4], frames[4])
        tk = [g.submit_chain(c, ds, dd, w, h, w * 4, "RGBA", st) for c, ds, dd in zip(ctxs, srcs, dsts)]
        for t in tk:
            g.wait(t)
        assert ctxs[0].colorlut_kernel_name() == "colorlut_table_tiled_multi_kernel"
        for i, (c, dd) in enumerate(zip(ctxs, dsts)):
            out = np.zeros(w * h * 4, np.uint8)
            c.d2h(out, dd)
            assert (out == exp[i]).all(), i
    finally:
        g.close()
        for c, p in bufs:
            c.free(p)
        for c in ctxs:
            c.close()
