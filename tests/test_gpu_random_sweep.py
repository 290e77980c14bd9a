"""Randomised parity sweeps (seeded): many settings / LUTs on random pixels, GPU through the C ABI vs the oracle,
bit-exact. These hunt for holes in the FAST-path envelope of hsvfilter (exact_math.hpp) and in the LDS colorlut
layout logic that the structured tests might miss."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _random_settings(rng):
    kind = rng.integers(0, 8)
    if kind == 0:
        hue = 0.0
    elif kind == 1:
        hue = float(rng.uniform(-360, 360))
    elif kind == 2:
        hue = float(rng.choice([360.0, -360.0, 1e-30, -1e-30, 359.99997, -359.99997, 180.0, 60.0, -60.0, 120.00001]))
    elif kind == 3:
        hue = float(rng.choice([1e-31, -1e-38, 360.00003, -400.0, 1e6, -1e9, 3.4e38]))     # outside the FAST envelope -> generic
    elif kind == 4:
        hue = float(rng.choice([np.inf, -np.inf, np.nan]))
    else:
        hue = float(np.float32(rng.normal(0, 120)))
    def sv():
        k = rng.integers(0, 6)
        if k == 0:
            return 1.0, 0.0
        if k == 1:
            return float(rng.uniform(0, 3)), float(rng.uniform(-1, 1))
        if k == 2:
            return float(rng.choice([0.0, -1.0, 1e-20, 1e20, np.inf, -np.inf, np.nan])), float(rng.choice([0.0, 0.5, -0.5, np.nan, np.inf]))
        return float(np.float32(rng.normal(1, 0.5))), float(np.float32(rng.normal(0, 0.2)))
    sm, so = sv()
    vm, vo = sv()
    return (hue, sm, so, vm, vo)


@pytest.mark.parametrize("table", [False, True])
def test_hsvfilter_random_settings_sweep(ctx, oracle, table):
    """table=True: the same sweep through the memoised-table kernel (MI355_FLAG_HSV_TABLE = 2; the table is rebuilt by
    the arithmetic kernel for every settings / byte-order change, alpha-first formats fall back)."""
    import mi355fx
    if table:
        ctx.set_flag(mi355fx.FLAG_HSV_TABLE, 2)
    rng = np.random.default_rng(20260101)
    w, h = 512, 128
    base = rng.integers(0, 256, (h, w * 4), dtype=np.uint8)
    # salt with edge colours: greys, primaries, near-ties
    edge = np.array([[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [1, 0, 0], [254, 255, 255], [128, 128, 127], [10, 10, 10],
                     [255, 254, 255], [0, 1, 1], [200, 200, 199]], np.uint8)
    for k, e in enumerate(edge):
        base[0, 4 * k: 4 * k + 3] = e
    bad = []
    for it in range(160):
        st = _random_settings(rng)
        fmt = ["RGBA", "BGRx", "xRGB", "ABGR"][it % 4]
        first, bgr = {"RGBA": (0, False), "BGRx": (0, True), "xRGB": (1, False), "ABGR": (1, True)}[fmt]
        exp = base.copy().reshape(-1)
        oracle.hsvfilter(exp, w, w * 4, 4, first, bgr, st, nthreads=4)
        got = base.copy().reshape(-1)
        ctx.hsvfilter_frame_ip(got, w, w * 4, fmt, st)
        if not (got == exp).all():
            bad.append((st, fmt, int((got != exp).sum())))
    assert not bad, bad[:5]


@pytest.mark.parametrize("variant", [6, 5, 4])
def test_colorlut_random_luts_sweep(ctx, oracle, variant):
    """variant 6: interpolating kernels; 5 / 4: the memoised-table kernel (Morton / linear index), rebuilt per LUT."""
    import mi355fx
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, variant)
    rng = np.random.default_rng(77)
    w, h = 256, 96
    frame = rng.integers(0, 256, (h, w * 4), dtype=np.uint8)
    bad = []
    for it in range(40):
        size = int(rng.choice([2, 3, 4, 5, 7, 9, 16, 17, 18, 19, 25, 32, 33, 34, 36, 40]))
        table = rng.uniform(-0.2, 1.2, (size ** 3, 4)).astype(np.float32)
        table[:, 3] = 1.0
        if it % 5 == 0:
            table[rng.integers(0, size ** 3, 5), rng.integers(0, 3, 5)] = rng.choice([np.inf, -np.inf, np.nan, 1e30, -1e30], 5)
        lo = rng.uniform(-0.3, 0.2, 3).astype(np.float32)
        hi = (lo + rng.uniform(0.5, 1.5, 3)).astype(np.float32)
        scale = (np.float32(1.0) / (hi - lo)).astype(np.float32)
        offset = (-lo * scale).astype(np.float32)
        cube = oracle.Cube.from_table(True, size, table, scale, offset)
        ctx.colorlut_load(True, size, table, scale, offset)
        exp = np.zeros_like(frame)
        oracle.colorlut_rgba8(cube, frame, w * 4, exp, w * 4, w, h, nthreads=4)
        got = np.zeros_like(frame)
        ctx.colorlut_frame(frame, w * 4, got, w * 4, w, h, "RGBA")
        if not (got == exp).all():
            bad.append((it, size, int((got != exp).sum())))
    assert not bad, bad[:5]


@pytest.mark.parametrize("variant", [6, 5])
def test_fused_chain_random_sweep(ctx, oracle, variant):
    """variant 6: fused interpolating kernel; 5: the table of the composed function, rebuilt per LUT / settings."""
    import mi355fx
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, variant)
    rng = np.random.default_rng(5150)
    w, h = 256, 64
    frame = rng.integers(0, 256, (h, w * 4), dtype=np.uint8)
    size = 33
    g = np.linspace(0, 1, size, dtype=np.float32)
    bad = []
    for it in range(30):
        table = np.zeros((size ** 3, 4), np.float32)
        bb, gg, rr = np.meshgrid(g, g, g, indexing="ij")
        table[:, 0] = (rr.reshape(-1) ** rng.uniform(0.5, 2.0)).astype(np.float32)
        table[:, 1] = (gg.reshape(-1) * rng.uniform(0.5, 1.1)).astype(np.float32)
        table[:, 2] = (1.0 - (1.0 - bb.reshape(-1)) ** rng.uniform(0.5, 2.0)).astype(np.float32)
        table[:, 3] = 1.0
        ctx.colorlut_load(True, size, table)
        cube = oracle.Cube.from_table(True, size, table)
        st = _random_settings(rng)
        mid = frame.copy().reshape(-1)
        oracle.hsvfilter(mid, w, w * 4, 4, 0, False, st, nthreads=4)
        exp = np.zeros_like(mid)
        oracle.colorlut_rgba8(cube, mid, w * 4, exp, w * 4, w, h, nthreads=4)
        d_a, d_b = ctx.alloc(frame.nbytes), ctx.alloc(frame.nbytes)
        try:
            ctx.h2d(d_a, frame.reshape(-1))
            ctx.hsv_colorlut_frames_device(d_a, frame.nbytes, w * 4, d_b, frame.nbytes, w * 4, 1, w, h, st)
            ctx.synchronize()
            got = np.zeros_like(mid)
            ctx.d2h(got, d_b)
        finally:
            ctx.free(d_a); ctx.free(d_b)
        if not (got == exp).all():
            bad.append((st, int((got != exp).sum())))
    assert not bad, bad[:5]


def test_dssim_fused_against_two_step_random_geometry(ctx):
    """Random frame sizes (below a tile, ragged right / bottom tiles, odd sizes whose down-sampled scales drop a column, sizes
    with fewer than five scales), formats, strides and contents: mi355_dssim_compare_frames == create_image + compare, to the
    last bit, and the detector of the restatement agrees on a few of them."""
    from oracle import dssim_restate as D
    rng = np.random.default_rng(20261002)
    for trial in range(24):
        w = int(rng.choice([1, 3, 7, 8, 9, 15, 16, 31, 32, 33, 40, 63, 64, 65, 100, 129, 257, int(rng.integers(34, 700))]))
        h = int(rng.choice([1, 2, 7, 8, 9, 16, 17, 24, 33, 47, 48, 49, 80, 130, int(rng.integers(18, 400))]))
        fmt = "RGBA" if rng.integers(0, 2) else "RGB"
        ch = 4 if fmt == "RGBA" else 3
        stride = w * ch + int(rng.choice([0, 0, 1, 5, 16]))
        def frame(kind):
            f = np.zeros((h, stride), np.uint8)
            if kind == 0:
                v = rng.integers(0, 256, (h, w * ch), dtype=np.uint8)
            elif kind == 1:
                v = np.full((h, w * ch), int(rng.integers(0, 256)), np.uint8)
            else:
                g = np.linspace(0, 255, w * ch)[None, :] * np.linspace(0.3, 1.0, h)[:, None]
                v = np.clip(g + rng.normal(0, 3, (h, w * ch)), 0, 255).astype(np.uint8)
            f[:, : w * ch] = v
            if ch == 4 and rng.integers(0, 3):
                f[:, 3: w * 4: 4] = 255
            return f
        ref = frame(int(rng.integers(0, 3)))
        others = [frame(int(rng.integers(0, 3))) for _ in range(3)] + [ref.copy()]
        a = ctx.dssim_create_image(ref, stride, w, h, fmt)
        try:
            two_step = []
            for f in others:
                b = ctx.dssim_create_image(f, stride, w, h, fmt)
                two_step.append(ctx.dssim_compare(a, b))
                ctx.dssim_free_image(b)
            fused = ctx.dssim_compare_frames(a, others, stride, w, h, fmt)
            assert fused == two_step, (trial, w, h, fmt, stride, fused, two_step)
            assert fused[-1] == 0.0
            if trial % 6 == 0 and w * h < 40000:
                oa = D.DssimImage(np.ascontiguousarray(ref[:, : w * ch]), w, h, w * ch, ch)
                ob = D.DssimImage(np.ascontiguousarray(others[0][:, : w * ch]), w, h, w * ch, ch)
                assert fused[0] == pytest.approx(D.compare(oa, ob), rel=1e-9, abs=1e-13)
        finally:
            ctx.dssim_free_image(a)
