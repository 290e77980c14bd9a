"""GPU parity tests for audioloudnorm through the C ABI against the CPU oracle.

The State machine (gain history, limiter states) is transcribed on the host; the per-sample loops and the peak search
run as kernels. Everything in-tree is f64 and unfused, so the samples must be BIT-IDENTICAL to the oracle as long
as the two loudness meters agree; the meters (device vs oracle) agree to ~1e-9 LU (tests/test_gpu_ebur128.py), which
enters the samples only through pow(10, x/20) of the gain history: tolerance 1e-9 relative, and bit-exact is asserted
where no meter-dependent gain is involved (linear path excluded: offset depends on the global loudness)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RATE = 192000


def _tone(seconds, ch, amp=0.02, f=440.0):
    t = np.arange(int(seconds * RATE)) / RATE
    return np.stack([amp * np.sin(2 * np.pi * (f + 3 * c) * t) for c in range(ch)], 1)


def _both(ctx, oracle, x, chunk, **kw):
    ch = x.shape[1]
    ln = oracle.LoudNorm(ch, **kw)
    ctx.loudnorm_setup(ch, **kw)
    got, exp = [], []
    for k in range(0, len(x), chunk):
        g, e = ctx.loudnorm_push(x[k:k + chunk]), ln.push(x[k:k + chunk])
        assert g.size == e.size
        got.append(g); exp.append(e)
    g, e = ctx.loudnorm_drain(), ln.drain()
    assert (g is None) == (e is None)
    if g is not None:
        assert g.size == e.size
        got.append(g); exp.append(e)
    return np.concatenate(got), np.concatenate(exp)


def _close(got, exp):
    scale = max(1e-12, float(np.abs(exp).max()))
    return float(np.abs(got - exp).max()) / scale


@pytest.mark.parametrize("ch", [1, 2, 6])
def test_limiter_heavy_stream_matches_oracle(ctx, oracle, ch):
    """Bursts far above the ceiling in every limiter situation: isolated, rising series (attack restarts), falling
    series (sustain), a burst inside the release window, one in the first 10 ms, one near the very end."""
    x = _tone(7.3, ch)
    rng = np.random.default_rng(ch)
    for start, n, k in ((0.001, 40, 80.0), (3.5, 2000, 60.0), (3.52, 300, 90.0), (3.56, 100, 40.0), (3.7, 50, 30.0), (4.4, 10, 70.0),
                        (4.41, 10, 75.0), (4.42, 10, 85.0), (5.9, 500, 55.0), (7.28, 100, 65.0)):
        i = int(start * RATE)
        x[i:i + n] *= k
    x += 1e-4 * rng.standard_normal(x.shape)
    got, exp = _both(ctx, oracle, x, 123457)
    assert got.size == x.size
    assert np.abs(got).max() <= 10 ** (-2.0 / 20)
    assert _close(got, exp) <= 1e-9


def test_quiet_then_loud_gain_tracking(ctx, oracle):
    """Silence (below -70 LUFS: above_threshold stays false), then programme material: exercises the delta history,
    the gaussian smoothing and the 1.0058 creep (imp.rs:560-575)."""
    x = np.concatenate([_tone(3.4, 2, amp=1e-6), _tone(4.0, 2, amp=0.2), _tone(1.1, 2, amp=0.01)])
    got, exp = _both(ctx, oracle, x, 96000)
    assert _close(got, exp) <= 1e-9


def test_short_stream_linear_path_and_empty(ctx, oracle):
    x = _tone(1.3, 2, amp=0.05)
    got, exp = _both(ctx, oracle, x, 50000)
    assert got.size == x.size and _close(got, exp) <= 1e-9
    ctx.loudnorm_setup(2)
    assert ctx.loudnorm_drain() is None   # FlowError::Eos


def test_settings_and_exactly_three_seconds(ctx, oracle):
    x = _tone(3.0, 1, amp=0.3)
    x[int(2.95 * RATE)] = 0.999
    got, exp = _both(ctx, oracle, x, 192000, loudness_target=-16.0, loudness_range_target=11.0, max_true_peak=-1.0, offset=0.5)
    assert got.size == x.size
    assert np.abs(got).max() <= 10 ** (-1.0 / 20)
    assert _close(got, exp) <= 1e-9


def test_not_negotiated(ctx):
    import mi355fx
    with pytest.raises(mi355fx.Mi355Error) as e:
        ctx._ln_channels = 1
        ctx.loudnorm_push(np.zeros(10))
    assert e.value.status == mi355fx.ERR_NOT_CONFIGURED


# ------------------------------------------------------------------ batches of streams (round 3)

def _burst_stream(seed, seconds, ch):
    """programme material with limiter work in it: a tone at a per-stream level with bursts far above the ceiling"""
    rng = np.random.default_rng(seed)
    x = _tone(seconds, ch, amp=0.01 * (1 + seed % 5), f=300.0 + 37 * seed)
    for _ in range(6):
        i = int(rng.uniform(0.0, seconds - 0.05) * RATE)
        x[i:i + int(rng.integers(10, 3000))] *= rng.uniform(20.0, 90.0)
    return x + 1e-4 * rng.standard_normal(x.shape)


def _run_batch(ctx, streams, ch, **kw):
    """feeds the batch the way drain_full_frames does: whole frames, then the rest at drain"""
    S = len(streams)
    ctx.loudnorm_setup_batch(S, ch, **kw)
    n = len(streams[0])
    outs, pos = [], 0
    while True:
        need = ctx.loudnorm_batch_frame_size()
        if n - pos < need:
            break
        outs.append(ctx.loudnorm_process_batch(np.stack([s[pos:pos + need].reshape(-1) for s in streams])))
        pos += need
    outs.append(ctx.loudnorm_process_batch(np.stack([s[pos:].reshape(-1) for s in streams]), final=True))
    ctx.loudnorm_teardown()
    return np.concatenate(outs, axis=1)


@pytest.mark.parametrize("ch,seconds", [(2, 5.23), (1, 3.0), (6, 3.71)])
def test_batch_equals_separate_streams_and_oracle(ctx, oracle, ch, seconds):
    """n streams in lock step, the limiter's state machine on the device: every stream's samples equal those of a separate
    single-stream context BIT FOR BIT (same expressions, same meters) and the oracle's within the meters' 1e-9."""
    import mi355fx
    S = 5
    streams = [_burst_stream(10 * ch + s, seconds, ch) for s in range(S)]
    streams[3] = _tone(seconds, ch, amp=1e-6)           # below -70 LUFS throughout: above_threshold never set
    got = _run_batch(ctx, streams, ch)
    assert got.shape == (S, int(seconds * RATE) * ch)
    for s in range(S):
        with mi355fx.Context(0) as c1:
            single, exp = _both(c1, oracle, streams[s], 200000)
        assert (got[s] == single).all(), (s, int((got[s] != single).sum()))
        assert _close(got[s], exp) <= 1e-9
        assert np.abs(got[s]).max() <= 10 ** (-2.0 / 20)


def test_batch_short_streams_take_the_linear_path(ctx, oracle):
    """less than 3 s in all: process_first_frame_is_last (imp.rs:334-366), a per-stream linear gain"""
    streams = [_tone(1.3, 2, amp=0.05 * (s + 1)) for s in range(3)]
    got = _run_batch(ctx, streams, 2, loudness_target=-20.0)
    for s in range(3):
        ln = oracle.LoudNorm(2, loudness_target=-20.0)
        assert ln.push(streams[s]).size == 0
        assert _close(got[s], ln.drain()) <= 1e-9


def test_batch_argument_checks(ctx):
    import mi355fx
    ctx.loudnorm_setup_batch(2, 2)
    assert ctx.loudnorm_batch_frame_size() == 3 * RATE
    with pytest.raises(mi355fx.Mi355Error):
        ctx.loudnorm_process_batch(np.zeros((2, 100 * 2)))   # not a whole frame
    ctx.loudnorm_teardown()
    assert ctx.loudnorm_batch_frame_size() == 0
