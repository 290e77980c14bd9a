"""GPU parity tests for hrtfrender (BASELINE config 4) through the C ABI.

Parity is tolerance-based: the reference evaluates the per-step streaming convolution with
an f32 FFT overlap-save inside the `hrtf` crate; the device evaluates the same sums in the time domain. Both are
compared with the f64 exact value: the device error must stay within the same 2e-5-of-full-scale bound the oracle's
FFT restatement is held to (tests/test_oracle_hrtf.py), and the two must agree with each other within 4e-5.
The mesh search (which face, which weights) is integer/f32-exact and compared bit-for-bit.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "test.hrir")
TOL_EXACT = 2e-5   # of full scale, vs the f64 time-domain value
TOL_ORACLE = 4e-5  # device vs the f32 FFT restatement (both carry up to TOL_EXACT)


def _mesh():
    return open(GOLDEN, "rb").read()


def test_reference_fixture_loads(ctx):
    ctx.hrtf_load_sphere(_mesh(), 44100)
    assert ctx.hrtf_sphere_info() == (1, 187, 370)


def test_load_errors(ctx):
    import mi355fx
    b = _mesh()
    for bad, status in ((b"XXXX" + b[4:], mi355fx.ERR_INVALID_ARG), (b[:64], mi355fx.ERR_INVALID_ARG)):
        with pytest.raises(mi355fx.Mi355Error) as e:
            ctx.hrtf_load_sphere(bad, 44100)
        assert e.value.status == status
    with pytest.raises(mi355fx.Mi355Error) as e:   # "Impulse response not set" (imp.rs:93)
        ctx.hrtf_setup(2, 512, 8)
    assert e.value.status == mi355fx.ERR_NOT_CONFIGURED
    ctx.hrtf_load_sphere(b, 48000)   # a sphere at another rate is resampled (1 tap at 44.1 kHz -> round(48/44.1) = 1 tap)
    assert ctx.hrtf_sphere_info() == (1, 187, 370)


def _run(ctx, oracle, synth, length, channels, steps, block, n_blocks, seed, static=False, method=0):
    import mi355fx
    data = synth.hrir_sphere_bytes(_mesh(), length)
    sphere = oracle.HrirSphere(data, 44100)
    ctx.hrtf_load_sphere(data, 44100)
    ctx.set_flag(mi355fx.FLAG_HRTF_METHOD, method)   # 0 = by HRIR length, 1 = overlap-save FFT, 2 = time-domain FIR (read at setup)
    try:
        ctx.hrtf_setup(channels, block, steps)
    finally:
        ctx.set_flag(mi355fx.FLAG_HRTF_METHOD, 0)
    r = oracle.HrtfRender(sphere, channels, steps, block)
    ex = oracle.HrtfExact(sphere, channels, steps, block)
    rng = np.random.default_rng(seed)
    pos = rng.standard_normal((channels, 3)).astype(np.float32)
    gains = rng.uniform(0.2, 1.0, channels).astype(np.float32)
    worst_exact = worst_oracle = oracle_exact = scale = 0.0
    prev_pos = None
    for blk in range(n_blocks):
        x = rng.uniform(-1, 1, (steps * block, channels)).astype(np.float32)
        if not static:
            pos = (pos + 0.6 * rng.standard_normal((channels, 3))).astype(np.float32)
            gains = rng.uniform(0.2, 1.0, channels).astype(np.float32)
        got = ctx.hrtf_process_block(x, pos, gains)
        a = r.process_block(x, pos, gains)
        e = ex.process_block(x, pos, gains)
        faces, uvw = ctx.hrtf_last_lookup()
        # mesh search: bit-exact against the oracle's scan for the last step (t == 1 -> the new vector itself)
        for c in range(channels):
            pv = pos[c] if prev_pos is None else prev_pos[c]
            f, w = sphere.sample((pv + (pos[c] - pv) * np.float32(1.0)).astype(np.float32))  # lerp(prev, new, t=1) in f32
            assert faces[c, steps - 1] == f
            assert (uvw[c, steps - 1] == w).all()
        prev_pos = pos.copy()
        worst_exact = max(worst_exact, float(np.abs(got - e).max()))
        oracle_exact = max(oracle_exact, float(np.abs(a - e).max()))
        worst_oracle = max(worst_oracle, float(np.abs(got - a).max()))
        scale = max(scale, float(np.abs(e).max()))
    return worst_exact, worst_oracle, oracle_exact, max(scale, 1.0)


@pytest.mark.parametrize("method", [1, 2])
@pytest.mark.parametrize("length,channels,steps,block", [(1, 1, 8, 512), (32, 2, 8, 64), (128, 8, 8, 512), (100, 3, 4, 77), (512, 4, 2, 1500)])
def test_blocks_match_oracle_and_exact(ctx, oracle, synth, length, channels, steps, block, method):
    """both convolution forms - the overlap-save FFT in LDS (round 3; where block + HRIR fit 4096 points, else the call falls
    back to the FIR) and the time-domain FIR - against the f64 exact value and the f32 FFT restatement"""
    we, wo, oe, scale = _run(ctx, oracle, synth, length, channels, steps, block, 4, 17 * length + block, method=method)
    assert we <= TOL_EXACT * scale, (we, scale)
    assert wo <= TOL_ORACLE * scale, (wo, scale)


@pytest.mark.parametrize("length,block,steps", [(1024, 512, 4), (2048, 512, 2), (3000, 1024, 2), (1500, 256, 4)])
def test_long_hrirs_through_the_fft(ctx, oracle, synth, length, block, steps):
    """HRIRs of 1 k - 3 k taps (hrtf/imp.rs:221-230 convolves whatever length the sphere file holds): O(N log N) per step on
    the device as well since round 3; default method selection."""
    we, wo, oe, scale = _run(ctx, oracle, synth, length, 3, steps, block, 3, length + block)
    assert we <= 2 * TOL_EXACT * scale, (we, scale)
    assert wo <= 2 * TOL_ORACLE * scale, (wo, scale)


@pytest.mark.parametrize("method", [0, 1, 2])
def test_config4_64_sources_default_block(ctx, oracle, synth, method):
    """BASELINE config 4: 64 sources, block 512 x 8 steps, 256-tap HRIRs; moving sources over 3 blocks; the default method
    (FIR at this length: the measured crossover is near 384 taps) and both forms pinned."""
    we, wo, oe, scale = _run(ctx, oracle, synth, 256, 64, 8, 512, 3, 4242, method=method)
    assert we <= TOL_EXACT * scale, (we, scale)
    assert wo <= TOL_ORACLE * scale, (wo, scale)


def test_static_sources_streaming_equals_one_long_convolution(ctx, oracle, synth):
    """Size-independent property: with fixed positions and gains the blocks tile one long convolution
    (history carried across blocks); the device result over 6 blocks matches the f64 evaluation end to end."""
    we, _, _, scale = _run(ctx, oracle, synth, 64, 2, 8, 128, 6, 99, static=True)
    assert we <= TOL_EXACT * scale


def test_reset_clears_tails_only(ctx, oracle, synth):
    data = synth.hrir_sphere_bytes(_mesh(), 48)
    ctx.hrtf_load_sphere(data, 44100)
    ctx.hrtf_setup(1, 64, 2)
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, (128, 1)).astype(np.float32)
    pos, g = np.array([[0.3, 0.2, 1.0]], np.float32), np.array([0.8], np.float32)
    first = ctx.hrtf_process_block(x, pos, g)
    second = ctx.hrtf_process_block(x, pos, g)          # has the first block's tail in front
    assert np.abs(second[:40] - first[:40]).max() > 1e-4
    ctx.hrtf_reset()
    third = ctx.hrtf_process_block(x, pos, g)
    assert (third == first).all()


def test_zero_direction_keeps_previous_taps(ctx, oracle, synth):
    """A direction that hits no face (the zero vector) leaves the processor's HRTF untouched: silence before any
    successful lookup, the previous taps afterwards."""
    data = synth.hrir_sphere_bytes(_mesh(), 16)
    ctx.hrtf_load_sphere(data, 44100)
    ctx.hrtf_setup(1, 32, 2)
    x = np.ones((64, 1), np.float32)
    zero, g = np.zeros((1, 3), np.float32), np.array([1.0], np.float32)
    assert (ctx.hrtf_process_block(x, zero, g) == 0).all()
    faces, _ = ctx.hrtf_last_lookup()
    assert (faces == -1).all()
    pos = np.array([[0.0, 0.0, 1.0]], np.float32)
    ctx.hrtf_process_block(x, pos, g)
    a = ctx.hrtf_process_block(x, pos, g)
    ctx.hrtf_process_block(x, zero, g)   # lerp towards zero: later steps still hit, the last step (t=1) does not
    b = ctx.hrtf_process_block(x, zero, g)  # prev == new == zero: no hit at all -> taps from the last hit persist
    assert np.abs(b).max() > 0
    assert np.isfinite(a).all() and np.isfinite(b).all()


def test_not_negotiated(ctx):
    import mi355fx
    with pytest.raises(mi355fx.Mi355Error) as e:
        ctx._hrtf_shape = (1, 64, 2)
        ctx.hrtf_process_block(np.zeros((64, 1), np.float32), np.zeros((1, 3), np.float32), np.ones(1, np.float32))
    assert e.value.status == mi355fx.ERR_NOT_CONFIGURED


@pytest.mark.parametrize("file_rate,device_rate,length", [(44100, 48000, 200), (96000, 48000, 400), (44100, 44100, 64), (48000, 192000, 50)])
def test_sphere_at_another_rate_is_resampled_at_load(ctx, oracle, synth, file_rate, device_rate, length):
    """HrirSphere::new(bytes, rate) (audio/hrtf/src/hrtf/imp.rs:83-93) converts a sphere measured at another rate. The product
    does it in mi355_hrtf_load_sphere (windowed-sinc interpolation, C++ loop); the oracle's numpy restatement of the same
    method re-serialises the sphere at the stream rate. Rendering the same input through both must agree: the taps differ
    only by f64 summation order (1e-6 of full scale)."""
    import mi355fx
    data = synth.hrir_sphere_bytes(_mesh(), length, rate=file_rate)
    conv = oracle.resample_hrir_sphere_bytes(data, device_rate)
    exp_len = max(1, int(np.floor(length * device_rate / file_rate + 0.5)))
    ctx.hrtf_load_sphere(data, device_rate)
    assert ctx.hrtf_sphere_info() == (exp_len, 187, 370)
    channels, steps, block = 3, 4, 96
    ctx.hrtf_setup(channels, block, steps)
    rng = np.random.default_rng(length)
    pos = rng.standard_normal((channels, 3)).astype(np.float32)
    gains = rng.uniform(0.2, 1.0, channels).astype(np.float32)
    xs = [rng.uniform(-1, 1, (steps * block, channels)).astype(np.float32) for _ in range(3)]
    got = [ctx.hrtf_process_block(x, pos, gains) for x in xs]
    with mi355fx.Context(0) as c2:
        c2.hrtf_load_sphere(conv, device_rate)     # already at the stream rate: loaded as is
        assert c2.hrtf_sphere_info() == (exp_len, 187, 370)
        c2.hrtf_setup(channels, block, steps)
        exp = [c2.hrtf_process_block(x, pos, gains) for x in xs]
    scale = max(1.0, max(float(np.abs(e).max()) for e in exp))
    assert max(float(np.abs(g - e).max()) for g, e in zip(got, exp)) <= 1e-6 * scale
