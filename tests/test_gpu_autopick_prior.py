"""A LUT reload keeps what the kernel choice has learnt (csrc/colorlut_kernels.hip: lut_upload): which of the interpolating and the
memoised-table kernels is faster depends on the content, hardly on the LUT. Without the prior a used context spent its first
dozen launches after every reload re-learning through the slower kind (profiles/r05_configs_elements.txt: 17^3 at 0.134 ms next to
33^3 at 0.096). Replaces nothing in the reference (colorlut/imp.rs:168-194 installs a new `State { lut }`); output stays exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H, B = 3840, 2160, 4


def test_reload_starts_on_the_kind_that_served_the_previous_lut(ctx, oracle, synth):
    frame = synth.smooth_frame(W, H, seed=9)
    src = np.stack([frame] * B).reshape(-1)
    pitch = W * H * 4
    d_s, d_o = ctx.alloc(src.nbytes), ctx.alloc(src.nbytes)
    try:
        ctx.h2d(d_s, src)

        def load(size):
            cube = oracle.Cube.parse(synth.cube_text_3d(size))
            sc, of = cube.domain
            ctx.colorlut_load(cube.is3d, cube.size, cube.table, sc, of)
            return cube

        def launch():
            ctx.colorlut_frames_device(d_s, pitch, W * 4, d_o, pitch, W * 4, B, W, H, "RGBA")
            return ctx.colorlut_kernel_name()

        load(33)
        names = []
        for _ in range(40):
            names.append(launch())
            ctx.synchronize()      # (every measurement is readable at the next call: the learning phase is four launches)
        assert names[0].startswith("colorlut3d_") and names[-1] in ("colorlut_window_kernel", "colorlut_table_tiled_kernel"), names
        assert ctx.colorlut_kernel_choice()[0]
        # natural content: the table serves. Reload: the very first launches stay on the table (built for the new LUT in-stream)
        for size in (17, 65, 33):
            cube = load(size)
            after = [launch() for _ in range(10)]
            assert all(n in ("colorlut_window_kernel", "colorlut_table_tiled_kernel") for n in after), (size, after)
            out = np.zeros(pitch, np.uint8)
            ctx.synchronize()
            ctx.d2h(out, d_o)
            exp = np.zeros(pitch, np.uint8)
            oracle.colorlut_rgba8(cube, src[:pitch], W * 4, exp, W * 4, W, H, nthreads=8)
            assert (out == exp).all(), size
        # a 1D LUT has no table path: nothing carried over, nothing broken
        cube = oracle.Cube.parse(synth.cube_text_1d(256))
        sc, of = cube.domain
        ctx.colorlut_load(cube.is3d, cube.size, cube.table, sc, of)
        launch()
        load(33)
        assert launch().startswith("colorlut3d_")     # after a 1D LUT the 3D choice is learnt afresh
    finally:
        ctx.free(d_s)
        ctx.free(d_o)

