"""mi355_hsv_colorlut_chain_batches_device: hsvfilter in place then colorlut for n independent batches from ONE native call, on one
lane (the context's stream) or two (odd batches on a side stream that forks after batch 0 and joins before the call returns). What it
replaces is n pairs of the two element calls (video/hsv/src/hsvfilter/imp.rs:323-376, video/colorlut/src/colorlut/imp.rs:203-223 per
buffer): same kernels, same bytes - checked against those calls and against the oracle chain, including what the stream does next
(the join) and an error in the middle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lanes", [1, 2])
@pytest.mark.parametrize("w,h,n_frames,n_batches", [(1920, 1080, 2, 5), (640, 360, 1, 3), (1280, 720, 3, 1)])
def test_chain_batches_equal_the_element_calls_and_the_oracle(ctx, oracle, synth, lanes, w, h, n_frames, n_batches):
    import mi355fx
    cube = oracle.Cube.parse(synth.cube_text_3d(33))
    sc, of = cube.domain
    ctx.colorlut_load(cube.is3d, cube.size, cube.table, sc, of)
    st = synth.HSV_SETTINGS["mixed"]
    fb = w * h * 4
    batches = [np.concatenate([(synth.smooth_frame(w, h, seed=40 + 7 * b + f) if (b + f) % 3 else synth.noise_frame(w, h, seed=40 + 7 * b + f)).reshape(-1)
                               for f in range(n_frames)]) for b in range(n_batches)]
    srcs = [ctx.alloc(x.nbytes) for x in batches]
    dsts = [ctx.alloc(x.nbytes) for x in batches]
    srcs2 = [ctx.alloc(x.nbytes) for x in batches]
    dsts2 = [ctx.alloc(x.nbytes) for x in batches]
    try:
        for rep in range(3):   # (the first calls run while the kernel choice is still learning)
            for b, x in enumerate(batches):
                ctx.h2d(srcs[b], x); ctx.h2d(srcs2[b], x)
            ctx.chain_batches_device(srcs, dsts, n_frames, fb, w * 4, w, h, "RGBA", st, lanes=lanes)
            # what the stream does next is ordered behind BOTH lanes: read the results back on the context's stream right away
            got = [np.zeros_like(x) for x in batches]
            mid = [np.zeros_like(x) for x in batches]
            for b in range(n_batches):
                ctx.d2h(got[b], dsts[b]); ctx.d2h(mid[b], srcs[b])
            for b in range(n_batches):
                ctx.hsvfilter_frames_device(srcs2[b], n_frames, fb, w, h, w * 4, "RGBA", st)
                ctx.colorlut_frames_device(srcs2[b], fb, w * 4, dsts2[b], fb, w * 4, n_frames, w, h, "RGBA")
            for b, x in enumerate(batches):
                ref, refmid = np.zeros_like(x), np.zeros_like(x)
                ctx.d2h(ref, dsts2[b]); ctx.d2h(refmid, srcs2[b])
                assert (got[b] == ref).all() and (mid[b] == refmid).all(), (rep, b)
                if rep == 0:
                    for f in range(n_frames):
                        m = x[f * fb:(f + 1) * fb].copy()
                        oracle.hsvfilter(m, w, w * 4, 4, 0, False, st, nthreads=8)
                        e = np.zeros_like(m)
                        oracle.colorlut_rgba8(cube, m, w * 4, e, w * 4, w, h, nthreads=8)
                        assert (mid[b][f * fb:(f + 1) * fb] == m).all() and (got[b][f * fb:(f + 1) * fb] == e).all(), (b, f)
    finally:
        for p in srcs + dsts + srcs2 + dsts2:
            ctx.free(p)


def test_chain_batches_argument_errors(ctx, oracle, synth):
    import mi355fx
    st = synth.HSV_SETTINGS["hue90"]
    d = ctx.alloc(64 * 64 * 4)
    try:
        with pytest.raises(mi355fx.Mi355Error) as e:   # no LUT
            ctx.chain_batches_device([d], [d], 1, 64 * 64 * 4, 64 * 4, 64, 64, "RGBA", st)
        assert e.value.status == mi355fx.ERR_NOT_CONFIGURED
        cube = oracle.Cube.parse(synth.cube_text_3d(9))
        sc, of = cube.domain
        ctx.colorlut_load(cube.is3d, cube.size, cube.table, sc, of)
        with pytest.raises(mi355fx.Mi355Error) as e:   # three lanes do not exist
            ctx.chain_batches_device([d], [d], 1, 64 * 64 * 4, 64 * 4, 64, 64, "RGBA", st, lanes=3)
        assert e.value.status == mi355fx.ERR_INVALID_ARG
        with pytest.raises(mi355fx.Mi355Error) as e:   # a null batch in the middle: the lanes are joined, the error is reported
            ctx.chain_batches_device([d, 0, d], [d, d, d], 1, 64 * 64 * 4, 64 * 4, 64, 64, "RGBA", st, lanes=2)
        assert e.value.status == mi355fx.ERR_INVALID_ARG
        ctx.synchronize()
    finally:
        ctx.free(d)
