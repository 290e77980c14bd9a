"""Device buffers (mi355_buf, include/mi355fx.h "device buffers"; csrc/buf.hip): what a device GstMemory wraps so that adjacent
mi355 elements hand frames over in HBM - the analogue of the GPU pool the reference's d3d12colorlut offers in
propose_allocation / decide_allocation (video/colorlut/src/d3d12colorlut/imp.rs:385-492).

The chain test runs hsvdetector -> colorlut -> videocompare, each element on its OWN context (as three GStreamer elements
would), on ONE upload and ONE download, and compares every stage with the host entry points and the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H = 1280, 720
DETECT = (120.0, 40.0, 0.8, 0.5, 0.7, 0.6)


def test_dirty_tracking_uploads_and_downloads_only_when_needed(ctx):
    import mi355fx
    rng = np.random.default_rng(1)
    data = rng.integers(0, 256, size=4096 * 3 + 5, dtype=np.uint8)   # (not a multiple of 16)
    b = ctx.buf_alloc(data.size)
    try:
        assert b.size == data.size and b.state() == 0 and ctx.transfer_counts() == (0, 0)
        b.write(data)
        assert b.state() == 1 and ctx.transfer_counts() == (0, 0)      # host newer, nothing moved yet
        p = b.device_ptr(flags=mi355fx.MAP_READ)
        assert ctx.transfer_counts() == (1, 0) and b.state() == 0       # uploaded once
        assert b.device_ptr(flags=mi355fx.MAP_READ) == p and ctx.transfer_counts() == (1, 0)
        b.commit()
        assert (b.read() == data).all() and ctx.transfer_counts() == (1, 0)   # in sync: a READ map moves nothing
        # the device side changes (the reversed bytes through the library's own copy entry point, counted as an upload)
        q = b.device_ptr(flags=mi355fx.MAP_WRITE)
        assert b.state() == 2
        ctx.h2d(q, data[::-1].copy())
        h2d0, d2h0 = ctx.transfer_counts()
        assert (h2d0, d2h0) == (2, 0)
        b.commit()
        assert (b.read() == data[::-1]).all()
        assert ctx.transfer_counts() == (h2d0, d2h0 + 1) and b.state() == 0   # downloaded once
        assert (b.read() == data[::-1]).all() and ctx.transfer_counts() == (h2d0, d2h0 + 1)
        # a WRITE-only device use of a host-newer buffer skips the upload (the kernel overwrites it)
        b.write(data)
        b.device_ptr(flags=mi355fx.MAP_WRITE)
        assert ctx.transfer_counts() == (h2d0, d2h0 + 1) and b.state() == 2
        # memory mapped for reading may be read by kernels too (an aggregator maps its pads' frames); writes on either side: no
        b.map(mi355fx.MAP_READ)
        assert b.device_ptr(flags=mi355fx.MAP_READ)
        with pytest.raises(mi355fx.Mi355Error):
            b.device_ptr(flags=mi355fx.MAP_READ | mi355fx.MAP_WRITE)
        b.unmap()
        b.map(mi355fx.MAP_WRITE)
        with pytest.raises(mi355fx.Mi355Error):
            b.device_ptr(flags=mi355fx.MAP_READ)
        b.unmap()
    finally:
        b.close()


def test_three_elements_one_upload_one_download(ctx, oracle, synth):
    """hsvdetector (RGBx -> RGBA) -> colorlut (RGBA) -> videocompare (blockhash against the source's), three contexts, three
    device buffers: 1 H2D + 1 D2H in total; every stage equals its host entry point and the oracle."""
    import mi355fx
    from mi355fx import FMT_LAYOUT
    cube = oracle.Cube.parse(synth.cube_text_3d(17))
    sc, of = cube.domain
    src = synth.smooth_frame(W, H, seed=9).reshape(-1)
    det, lut, cmp_ = mi355fx.Context(0), mi355fx.Context(0), mi355fx.Context(0)
    lut.colorlut_load(cube.is3d, cube.size, cube.table, sc, of)
    b_in, b_mid, b_out = ctx.buf_alloc(src.size), ctx.buf_alloc(src.size), ctx.buf_alloc(src.size)
    try:
        b_in.write(src)                                                  # upstream wrote the frame into our memory
        s = mi355fx.HsvDetectSettings(*DETECT)
        import ctypes as C
        pitch, stride = W * H * 4, W * 4
        # hsvdetector: its own context, device pointers of OUR memory
        det._ck(det.L.mi355_hsvdetect_frames_device(det.h, b_in.device_ptr(det, mi355fx.MAP_READ), pitch, stride, mi355fx.FMT["RGBx"],
                                                    b_mid.device_ptr(det, mi355fx.MAP_WRITE), pitch, stride, mi355fx.FMT["RGBA"], 1, W, H, C.byref(s)))
        b_in.commit(det); b_mid.commit(det)
        # colorlut
        lut.colorlut_frames_device(b_mid.device_ptr(lut, mi355fx.MAP_READ), pitch, stride, b_out.device_ptr(lut, mi355fx.MAP_WRITE), pitch, stride, 1, W, H, "RGBA")
        b_mid.commit(lut); b_out.commit(lut)
        # videocompare: the hash of the graded frame against the source's
        h_out = cmp_.videocompare_hash_frames_device(b_out.device_ptr(cmp_, mi355fx.MAP_READ), pitch, stride, 1, W, H, "RGBA", "blockhash")[0]
        h_src = cmp_.videocompare_hash_frames_device(b_in.device_ptr(cmp_, mi355fx.MAP_READ), pitch, stride, 1, W, H, "RGBA", "blockhash")[0]
        b_out.commit(cmp_); b_in.commit(cmp_)
        dist = cmp_.videocompare_distance(h_src, h_out, "blockhash")
        assert ctx.transfer_counts() == (1, 0)                           # one upload so far, nothing has come back
        got = b_out.read()                                               # downstream maps the memory
        assert ctx.transfer_counts() == (1, 1)
        # the same through the host entry points (two PCIe crossings per element) and through the oracle
        ps, first, bgr = FMT_LAYOUT["RGBx"]
        mid_o = np.zeros_like(src)
        oracle.hsvdetect(src.reshape(H, -1), stride, ps, first, bool(bgr), mid_o.reshape(H, -1), stride, False, False, W, DETECT)
        exp = np.zeros_like(src)
        oracle.colorlut_rgba8(cube, mid_o, stride, exp, stride, W, H)
        assert (got == exp).all()
        mid_h = np.zeros_like(src)
        ctx.hsvdetect_frame(src, stride, "RGBx", mid_h, stride, "RGBA", W, DETECT)
        assert (mid_h == mid_o).all() and (b_mid.read() == mid_o).all()
        lut2 = mi355fx.Context(0)
        lut2.colorlut_load(cube.is3d, cube.size, cube.table, sc, of)
        out_h = np.zeros_like(src)
        lut2.colorlut_frame(mid_h, stride, out_h, stride, W, H, "RGBA")
        lut2.close()
        assert (out_h == got).all()
        assert h_out == cmp_.videocompare_hash_frame(out_h, stride, W, H, "RGBA", "blockhash") == oracle.blockhash(out_h, W, H, stride, 4)
        assert h_src == oracle.blockhash(src, W, H, stride, 4)
        assert dist == bin(h_src ^ h_out).count("1")
    finally:
        for b in (b_in, b_mid, b_out):
            b.close()
        for c in (det, lut, cmp_):
            c.close()


def test_two_contexts_are_ordered_through_commits(ctx, synth):
    """A producer context writes the buffer on its stream, a consumer context reads it on another: device_ptr orders the
    consumer behind the producer's commit without a host wait. 40 rounds of hsvfilter (ctx A, in place) then hsvfilter (ctx B)."""
    import mi355fx
    src = synth.smooth_frame(W, H, seed=3).reshape(-1)
    a, b_ctx = mi355fx.Context(0), mi355fx.Context(0)
    st = synth.HSV_SETTINGS["mixed"]
    buf = ctx.buf_alloc(src.size)
    try:
        buf.write(src)
        for _ in range(20):
            for c in (a, b_ctx):
                c.hsvfilter_frames_device(buf.device_ptr(c), 1, src.size, W, H, W * 4, "RGBA", st)
                buf.commit(c)
        got = buf.read()
        exp = src.copy()
        one = mi355fx.Context(0)
        d = one.alloc(src.size)
        one.h2d(d, exp)
        for _ in range(40):
            one.hsvfilter_frames_device(d, 1, src.size, W, H, W * 4, "RGBA", st)
        one.synchronize()
        one.d2h(exp, d)
        one.free(d); one.close()
        assert (got == exp).all()
        assert ctx.transfer_counts() == (1, 1)
    finally:
        buf.close(); a.close(); b_ctx.close()


def test_concurrent_readers_are_both_seen_by_a_later_writer(ctx, oracle, synth):
    """A tee into two mi355 branches: two contexts READ one buffer concurrently (each commits), then a third context WRITES it in
    place and the buffer is freed. The writer must start behind BOTH readers (round 5 kept only the last commit: the first
    reader's kernels could still be reading when the writer started - ADVICE r05). 4K x 12 rounds so that the readers are in flight
    when the writer is enqueued; every reader output is checked against the content it must have seen."""
    import ctypes as C
    import mi355fx
    w, h = 3840, 2160
    src = synth.smooth_frame(w, h, seed=5).reshape(-1)
    st = synth.HSV_SETTINGS["hue90"]
    ra, rb, wr = mi355fx.Context(0), mi355fx.Context(0), mi355fx.Context(0)
    buf = ctx.buf_alloc(src.size)
    out_a, out_b = ra.alloc(src.size), rb.alloc(src.size)
    try:
        buf.write(src)
        exp_in = src.copy()
        for rnd in range(12):
            # reader A: hsvdetect of the buffer into out_a; reader B: the same with another target format byte order
            ds = mi355fx.HsvDetectSettings(*DETECT)
            ra._ck(ra.L.mi355_hsvdetect_frames_device(ra.h, buf.device_ptr(ra, mi355fx.MAP_READ), src.size, w * 4, mi355fx.FMT["RGBx"], out_a, src.size, w * 4,
                                                      mi355fx.FMT["RGBA"], 1, w, h, C.byref(ds)))
            buf.commit(ra)
            rb._ck(rb.L.mi355_hsvdetect_frames_device(rb.h, buf.device_ptr(rb, mi355fx.MAP_READ), src.size, w * 4, mi355fx.FMT["RGBx"], out_b, src.size, w * 4,
                                                      mi355fx.FMT["ARGB"], 1, w, h, C.byref(ds)))
            buf.commit(rb)
            # the writer: hsvfilter in place, enqueued at once (no host wait anywhere)
            wr.hsvfilter_frames_device(buf.device_ptr(wr, mi355fx.MAP_READ | mi355fx.MAP_WRITE), 1, src.size, w, h, w * 4, "RGBA", st)
            buf.commit(wr)
            if rnd in (0, 11):
                ga, gb = np.zeros_like(src), np.zeros_like(src)
                ra.synchronize(); rb.synchronize()
                ra.d2h(ga, out_a); rb.d2h(gb, out_b)
                ea, eb = np.zeros_like(src), np.zeros_like(src)
                oracle.hsvdetect(exp_in, w * 4, 4, 0, False, ea, w * 4, False, False, w, DETECT)
                oracle.hsvdetect(exp_in, w * 4, 4, 0, False, eb, w * 4, True, False, w, DETECT)
                assert (ga == ea).all() and (gb == eb).all(), rnd
            oracle.hsvfilter(exp_in, w, w * 4, 4, 0, False, st)
        assert (buf.read() == exp_in).all()
    finally:
        buf.close()      # waits for every stream that used it
        ra.free(out_a); rb.free(out_b)
        for c in (ra, rb, wr):
            c.close()
