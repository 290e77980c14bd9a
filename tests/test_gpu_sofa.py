"""sofalizer's per-block loop on the device (uniformly partitioned FFT convolution in LDS, csrc/sofa_kernels.hip) through the
C ABI against the time-domain oracle (oracle.SofaRenderer). The crate that does this in the reference (sofar) is not in the
reference tree: parity is that of a streaming linear convolution, tolerance 2e-6 of full scale."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 2e-6


def _filters(rng, channels, L):
    k = np.arange(L)
    out = []
    for c in range(channels):
        env = np.exp(-k / (0.1 * L + 4.0))
        out.append(((0.5 * env * rng.standard_normal(L)).astype(np.float32), (0.4 * env * rng.standard_normal(L)).astype(np.float32)))
    return out


@pytest.mark.parametrize("channels,L,P,B", [(2, 200, 64, 256), (6, 512, 64, 256), (1, 33, 16, 64), (8, 1024, 128, 512), (3, 64, 64, 64), (2, 2048, 256, 256)])
def test_sofalizer_blocks_match_time_domain_convolution(ctx, oracle, channels, L, P, B):
    rng = np.random.default_rng(channels * 1000 + L)
    flt = _filters(rng, channels, L)
    ref = oracle.SofaRenderer(channels, L, B)
    ctx.sofa_setup(channels, L, P, B)
    for c, (l, r) in enumerate(flt):
        d = (c % 3, (2 * c) % 5)
        ctx.sofa_set_filter(c, l, r, *d)
        ref.set_filter(c, l, r, *d)
    gains = (0.5 + 0.5 * rng.random(channels)).astype(np.float32)
    worst = 0.0
    for blk in range(7):
        x = (0.5 * rng.standard_normal((B, channels))).astype(np.float32)
        if blk == 3:   # a source moves: new filter from this block on (State::update_filters, sofa/imp.rs:129-160)
            l2, r2 = _filters(rng, 1, L)[0]
            ctx.sofa_set_filter(0, l2, r2, 1, 0)
            ref.set_filter(0, l2, r2, 1, 0)
        got, exp = ctx.sofa_process_block(x, gains), ref.process_block(x, gains)
        worst = max(worst, float(np.abs(got - exp).max()))
    scale = max(1.0, float(np.abs(exp).max()))
    assert worst <= TOL * scale * max(1, L // 64), (worst, scale)


def test_sofalizer_lfe_channels_are_dropped_and_reset_clears_history(ctx, oracle):
    rng = np.random.default_rng(3)
    C, L, P, B = 6, 128, 64, 256
    flt = _filters(rng, C, L)
    ref = oracle.SofaRenderer(C, L, B)
    ctx.sofa_setup(C, L, P, B)
    for c, (l, r) in enumerate(flt):
        if c == 3:   # LFE1: ChannelProcessor::Drop, never gets a filter (sofa/imp.rs:808-821)
            ctx.sofa_set_drop(c)
            ref.drop[c] = True
            continue
        ctx.sofa_set_filter(c, l, r)
        ref.set_filter(c, l, r)
    g = np.ones(C, np.float32)
    x = (0.5 * rng.standard_normal((B, C))).astype(np.float32)
    x[:, 3] = 100.0   # whatever the LFE carries must not reach the output
    for _ in range(2):
        got, exp = ctx.sofa_process_block(x, g), ref.process_block(x, g)
        assert np.abs(got - exp).max() <= TOL * max(1.0, np.abs(exp).max())
    ctx.sofa_reset()
    ref.reset()
    z = np.zeros((B, C), np.float32)
    assert (ctx.sofa_process_block(z, g) == 0).all()   # no tail of the earlier blocks after a flush


def test_sofalizer_argument_errors(ctx):
    import mi355fx
    with pytest.raises(mi355fx.Mi355Error) as e:
        ctx.sofa_setup(2, 128, 64, 200)          # "Block Length is not multiple of Partition Length" (sofa/imp.rs:775-781)
    assert e.value.status == mi355fx.ERR_INVALID_ARG and "multiple of Partition Length" in str(e.value)
    with pytest.raises(mi355fx.Mi355Error) as e:
        ctx.sofa_setup(2, 128, 48, 96)           # not a power of two
    assert e.value.status == mi355fx.ERR_UNSUPPORTED
    ctx.sofa_setup(2, 16, 16, 32)
    with pytest.raises(mi355fx.Mi355Error) as e:  # channel 1 has no filter and is not dropped
        ctx.sofa_set_filter(0, np.ones(16, np.float32), np.ones(16, np.float32))
        ctx.sofa_process_block(np.zeros((32, 2), np.float32), np.ones(2, np.float32))
    assert e.value.status == mi355fx.ERR_NOT_CONFIGURED
    ctx.sofa_teardown()
