"""GPU loudness meter (mi355_ebur128_*) vs the serial f64 oracle. The K-weighting recurrence is evaluated
as a chunked parallel scan, so filtered samples differ from the serial run by rounding of the chained
states (~1e-16 relative): loudness values are compared with |delta| <= 1e-9 LU (far inside 1 ulp of
the f32 the element's consumers print), peaks and histogram-derived quantities must be identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-9


def sine(db, secs, rate=48000, ch=2, f=1000.0):
    t = np.arange(int(round(secs * rate))) / rate
    s = (10 ** (db / 20.0)) * np.sin(2 * np.pi * f * t)
    return np.repeat(s[:, None], ch, axis=1).reshape(-1)


def programme(secs, rate, ch, seed=3):
    """Speech-like synthetic programme: band-limited noise bursts with a slow envelope, per-channel gains."""
    rng = np.random.default_rng(seed)
    n = int(secs * rate)
    x = rng.standard_normal((n, ch))
    k = np.ones(16) / 16.0
    for c in range(ch):
        x[:, c] = np.convolve(x[:, c], k, mode="same")
    env = 0.05 + 0.45 * (0.5 + 0.5 * np.sin(2 * np.pi * np.arange(n) / (rate * 2.7))) ** 3
    gains = np.linspace(1.0, 0.3, ch)
    return (x * env[:, None] * gains[None, :]).reshape(-1)


def close(a, b):
    if np.isinf(a) or np.isinf(b):
        return a == b
    return abs(a - b) <= TOL


def compare_all(ctx, ref, channels, mode=63):
    if mode & 1:
        assert close(ctx.ebur128_loudness_momentary(), ref.loudness_momentary())
    if mode & 2:
        assert close(ctx.ebur128_loudness_shortterm(), ref.loudness_shortterm())
    if mode & 4:
        assert close(ctx.ebur128_loudness_global(), ref.loudness_global())
        assert close(ctx.ebur128_relative_threshold(), ref.relative_threshold())
    if mode & 8:
        assert close(ctx.ebur128_loudness_range(), ref.loudness_range())
    for c in range(channels):
        if mode & 16:
            assert ctx.ebur128_sample_peak(c) == ref.sample_peak(c)
        if mode & 32:
            assert ctx.ebur128_true_peak(c) == ref.true_peak(c)


@pytest.mark.parametrize("db", [-23.0, -33.0])
def test_tech3341_sines(ctx, oracle, db):
    """EBU Tech 3341 cases 1 and 2 through the GPU meter: M, S, I within +-0.1 LU of the level."""
    x = sine(db, 20).astype(np.float32)
    ctx.ebur128_setup(2, 48000)
    ctx.ebur128_add_frames(x)
    assert abs(ctx.ebur128_loudness_momentary() - db) <= 0.1
    assert abs(ctx.ebur128_loudness_shortterm() - db) <= 0.1
    assert abs(ctx.ebur128_loudness_global() - db) <= 0.1
    ref = oracle.EbuR128(2, 48000)
    ref.add_frames(x)
    compare_all(ctx, ref, 2)


@pytest.mark.parametrize("dtype,planar", [(np.int16, False), (np.int32, False), (np.float32, False), (np.float64, False),
                                          (np.int16, True), (np.int32, True), (np.float32, True), (np.float64, True)])
def test_eight_format_layout_combinations(ctx, oracle, dtype, planar):
    """The 8 format x layout combinations the reference test feeds (audio/audiofx/tests/ebur128level.rs:93-153),
    one-second buffers like the element's default interval."""
    rate, ch = 48000, 2
    x = programme(6, rate, ch)
    if dtype == np.int16:
        x = np.clip(x * 32768.0 * 2, -32768, 32767).astype(np.int16)
    elif dtype == np.int32:
        x = np.clip(x * 2147483648.0 * 2, -2147483648, 2147483647).astype(np.int32)
    else:
        x = x.astype(dtype)
    ctx.ebur128_setup(ch, rate)
    ref = oracle.EbuR128(ch, rate)
    frames = x.size // ch
    for k in range(0, frames, rate):
        chunk = x[k * ch:(k + rate) * ch]
        if planar:
            pl = np.ascontiguousarray(chunk.reshape(-1, ch).T)
            ctx.ebur128_add_frames(pl, planar=True)
            ref.add_frames(pl, planar=True)
        else:
            ctx.ebur128_add_frames(chunk)
            ref.add_frames(chunk)
        compare_all(ctx, ref, ch)


@pytest.mark.parametrize("rate,ch,classes", [(44100, 1, [3]), (48000, 6, [1, 1, 1, 0, 2, 2]), (96000, 2, None), (192000, 2, [1, 1]),
                                             (32000, 3, [1, 1, 1]), (22050, 2, [1, 0])])
def test_rates_channels_and_weights(ctx, oracle, rate, ch, classes):
    x = programme(8, rate, ch, seed=rate % 97).astype(np.float32)
    ctx.ebur128_setup(ch, rate, 63, classes)
    ref = oracle.EbuR128(ch, rate, 63, classes)
    # ragged chunking across 100 ms boundaries
    pos, sizes = 0, [1, 17, rate // 10 - 1, rate // 10, rate // 10 + 1, rate // 3, 5, rate * 2, 3]
    frames = x.size // ch
    i = 0
    while pos < frames:
        n = min(sizes[i % len(sizes)], frames - pos)
        ctx.ebur128_add_frames(x[pos * ch:(pos + n) * ch])
        ref.add_frames(x[pos * ch:(pos + n) * ch])
        pos += n
        i += 1
    compare_all(ctx, ref, ch)


def test_loudness_range_and_gating(ctx, oracle):
    x = np.concatenate([sine(-20, 20), sine(-30, 20)]).astype(np.float32)      # Tech 3342 case 1: LRA 10
    ctx.ebur128_setup(2, 48000)
    ctx.ebur128_add_frames(x)
    assert abs(ctx.ebur128_loudness_range() - 10.0) <= 1.0
    x = np.concatenate([sine(-72, 10), sine(-36, 10), sine(-23, 60), sine(-36, 10), sine(-72, 10)])   # Tech 3341 case 4
    ctx.ebur128_reset()
    ctx.ebur128_add_frames(x)
    assert abs(ctx.ebur128_loudness_global() - (-23.0)) <= 0.1
    ref = oracle.EbuR128(2, 48000)
    ref.add_frames(x)
    compare_all(ctx, ref, 2)


def test_modes_reset_and_errors(ctx, oracle):
    import mi355fx
    with pytest.raises(mi355fx.Mi355Error) as e:
        ctx.ebur128_loudness_momentary()                                       # "Have no state yet" (imp.rs:303-306)
    assert e.value.status == mi355fx.ERR_NOT_CONFIGURED
    ctx.ebur128_setup(2, 48000, 1 | 16)                                        # momentary + sample peak only
    x = programme(2, 48000, 2).astype(np.float32)
    ctx.ebur128_add_frames(x)
    ref = oracle.EbuR128(2, 48000, 1 | 16)
    ref.add_frames(x)
    compare_all(ctx, ref, 2, mode=1 | 16)
    with pytest.raises(mi355fx.Mi355Error):
        ctx.ebur128_loudness_global()                                          # mode not enabled
    with pytest.raises(mi355fx.Mi355Error):
        ctx.ebur128_true_peak(0)
    ctx.ebur128_setup(2, 48000)
    ctx.ebur128_add_frames(x)
    ctx.ebur128_reset()
    assert ctx.ebur128_loudness_global() == -np.inf and ctx.ebur128_relative_threshold() == -70.0
    assert ctx.ebur128_sample_peak(0) == 0.0 and ctx.ebur128_loudness_range() == 0.0
    ctx.ebur128_add_frames(np.zeros(0, np.float32))                            # empty buffer: no-op


# ------------------------------------------------------------------ batch of streams

@pytest.mark.parametrize("dtype,channels,rate", [(np.float32, 2, 48000), (np.int16, 6, 44100), (np.float64, 1, 96000)])
def test_batch_equals_separate_meters(ctx, dtype, channels, rate):
    """n_streams meters fed in lock step through the batch entry points give, stream for stream, exactly what separate
    single-stream meters give (every loudness value, relative threshold, range and peak bit-identical): the batch only adds
    a grid dimension, the per-stream arithmetic and its order are unchanged."""
    import mi355fx
    S, mode = 5, 63
    rng = np.random.default_rng(11)
    secs = 4.3
    n = int(rate * secs)
    t = np.arange(n) / rate
    sig = np.empty((S, n, channels))
    for s in range(S):
        for c in range(channels):
            sig[s, :, c] = (0.05 + 0.17 * s) * np.sin(2 * np.pi * (180.0 + 97 * s + 13 * c) * t) + 0.01 * rng.standard_normal(n)
        sig[s, : n // 3] *= 0.2 + 0.1 * s   # level changes: gating and loudness range have something to do
    if dtype == np.int16:
        data = np.clip(sig * 32767, -32768, 32767).astype(np.int16)
    else:
        data = sig.astype(dtype)
    chunk = rate // 7 + 3                      # ragged buffers: segments cross the 100 ms boundaries
    # separate meters
    singles = []
    for s in range(S):
        with mi355fx.Context(0) as c1:
            c1.ebur128_setup(channels, rate, mode)
            for a in range(0, n, chunk):
                c1.ebur128_add_frames(data[s, a:a + chunk])
            singles.append([c1.ebur128_loudness_momentary(), c1.ebur128_loudness_shortterm(), c1.ebur128_loudness_global(),
                            c1.ebur128_relative_threshold(), c1.ebur128_loudness_range()] +
                           [c1.ebur128_sample_peak(c) for c in range(channels)] + [c1.ebur128_true_peak(c) for c in range(channels)])
    ctx.ebur128_setup_batch(S, channels, rate, mode)
    for a in range(0, n, chunk):
        ctx.ebur128_add_frames_batch(data[:, a:a + chunk])
    got = [ctx.ebur128_loudness_batch(w) for w in range(5)]
    sp, tp = ctx.ebur128_peak_batch(False), ctx.ebur128_peak_batch(True)
    for s in range(S):
        row = [got[w][s] for w in range(5)] + list(sp[s]) + list(tp[s])
        assert row == singles[s], (s, row, singles[s])
    # reset applies to all streams
    ctx.ebur128_reset()
    assert (ctx.ebur128_loudness_batch(2) == -np.inf).all() and (ctx.ebur128_peak_batch(False) == 0).all()


def test_batch_and_single_entry_points_do_not_mix(ctx):
    import mi355fx
    ctx.ebur128_setup_batch(3, 2, 48000, 63)
    with pytest.raises(mi355fx.Mi355Error):
        ctx.ebur128_add_frames(np.zeros((480, 2), np.float32))


# ---- EBU Tech 3341 / 3342 conformance on the device (tests/golden/ebu_tech_334x.json: published recipes, expected readings and
# tolerances; the CPU oracle runs the same cases in tests/test_oracle_ebur128.py)
from ebu_cases import check_case, load_cases  # noqa: E402

_RATE, _CASES = load_cases()


class _DeviceMeter:
    def __init__(self, ctx, case, rate):
        self.c = ctx
        ctx.ebur128_setup(case["channels"], rate, 63, case.get("channel_class"))
    add = lambda self, x: self.c.ebur128_add_frames(x)
    momentary = lambda self: self.c.ebur128_loudness_momentary()
    shortterm = lambda self: self.c.ebur128_loudness_shortterm()
    integrated = lambda self: self.c.ebur128_loudness_global()
    lra = lambda self: self.c.ebur128_loudness_range()
    true_peak = lambda self, ch: self.c.ebur128_true_peak(ch)


@pytest.mark.parametrize("case", _CASES, ids=[c["id"] for c in _CASES])
def test_ebu_tech_334x_conformance_device(ctx, synth, case):
    check_case(case, _RATE, _DeviceMeter(ctx, case, _RATE), synth)
