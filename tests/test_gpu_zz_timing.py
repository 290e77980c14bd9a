"""Behaviour that depends on in-stream TIMINGS, kept apart from the parity suites (and last in collection order, so that a
slow or noisy box cannot hide a parity result behind `-x`): the run-time kernel choice of colorlut following the content.
Exactness of the same scenario is asserted in tests/test_gpu_parity.py::test_colorlut_auto_stays_exact_when_the_content_changes;
the policy itself is exercised deterministically on the CPU against a scripted device (tests/test_autopick.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W4K, H4K = 3840, 2160


def test_colorlut_auto_follows_the_content(ctx, oracle, synth):
    """Natural-like frames end up on the memoised table, uniform noise flips the choice to the interpolating path within the
    sampling interval (video/colorlut/src/colorlut/imp.rs:267-294 is the loop both replace)."""
    import mi355fx
    cube = oracle.Cube.parse(synth.cube_text_3d(33))
    sc, of = cube.domain
    ctx.colorlut_load(cube.is3d, cube.size, cube.table, sc, of)
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 0)
    n = 8      # (the bench's launch size: the two kinds are 35 % apart there; at two frames they are within 10-20 % and a scattered sample can hold the choice)
    nb = n * W4K * H4K * 4
    d_s, d_n, d_o = ctx.alloc(nb), ctx.alloc(nb), ctx.alloc(nb)
    try:
        ctx.h2d(d_s, np.stack([synth.smooth_frame(W4K, H4K, seed=40 + i % 2) for i in range(n)]).reshape(-1))
        ctx.h2d(d_n, np.stack([synth.noise_frame(W4K, H4K, seed=50 + i % 2) for i in range(n)]).reshape(-1))
        run = lambda d: ctx.colorlut_frames_device(d, H4K * W4K * 4, W4K * 4, d_o, H4K * W4K * 4, W4K * 4, n, W4K, H4K, "RGBA")
        # (the device is kept busy - a wait after every eighth launch only, as a streaming host does: with a wait after every
        # launch the clocks drop between launches and both kernels measure the same ~45 us of ramp-up for these two frames)
        for k in range(160):
            run(d_s)
            if k % 8 == 7:
                ctx.synchronize()
        on_table, t_c, t_t = ctx.colorlut_kernel_choice()
        assert on_table and 0.0 < t_t < t_c, (on_table, t_c, t_t)
        for k in range(160):
            run(d_n)
            if k % 8 == 7:
                ctx.synchronize()
        on_table, t_c, t_t = ctx.colorlut_kernel_choice()
        assert not on_table and t_t > t_c, (on_table, t_c, t_t)
    finally:
        for d in (d_s, d_n, d_o):
            ctx.free(d)
