"""CPU tests of the audioloudnorm oracle (oracle/loudnorm_oracle.c). The reference has no tests with sample values for
this element; what is pinned here are the invariants of the reference's design: latency/length
bookkeeping of the three frame types, the hard ceiling, the linear path for short streams, and loudness moving
towards the target."""
import numpy as np
import pytest

RATE = 192000


def _tone(seconds, ch, amp=0.02, f=440.0):
    t = np.arange(int(seconds * RATE)) / RATE
    return np.stack([amp * np.sin(2 * np.pi * (f + 3 * c) * t) for c in range(ch)], 1)


def _run(ln, x, chunk):
    outs = []
    for k in range(0, len(x), chunk):
        outs.append(ln.push(x[k:k + chunk]))
    outs.append(ln.drain())
    return np.concatenate([o for o in outs if o is not None]).reshape(-1, x.shape[1])


def test_gaussian_weights_sum_to_one_and_lengths(oracle):
    ln = oracle.LoudNorm(2)
    x = _tone(4.25, 2)
    assert ln.push(x[:RATE]).size == 0 and ln.frame_type == 0           # first 3 s only collected
    o = ln.push(x[RATE:3 * RATE])
    assert o.size == 19200 * 2 and ln.frame_type == 1                    # first frame -> 100 ms out, now Inner
    o = ln.push(x[3 * RATE:])                                            # 1.25 s more: 12 full frames
    assert o.size == 12 * 19200 * 2
    d = ln.drain()                                                       # Final: the remaining 3 s - 100 ms + residue
    assert ln.frame_type == 2
    assert 19200 + 12 * 19200 + d.size // 2 == len(x)                    # every input frame comes out exactly once


def test_short_stream_takes_linear_path(oracle):
    ln = oracle.LoudNorm(1)
    x = _tone(1.0, 1, amp=0.05)
    assert ln.push(x).size == 0
    y = ln.drain()
    assert ln.frame_type == 3 and y.size == x.size
    assert np.allclose(y, x.reshape(-1) * ln.offset, rtol=0, atol=0)     # one multiply per sample
    m = oracle.EbuR128(1, RATE); m.add_frames(y)
    assert abs(m.loudness_global() - (-24.0)) < 0.1                      # short, peak-safe: lands on the target
    assert oracle.LoudNorm(1).drain() is None                            # nothing at all: FlowError::Eos


@pytest.mark.parametrize("ch", [1, 2])
def test_ceiling_and_loudness(oracle, ch):
    x = _tone(6.35, ch)
    x[int(4.0 * RATE): int(4.0 * RATE) + 2000] *= 60.0                   # a burst far above the ceiling
    x[int(5.2 * RATE): int(5.2 * RATE) + 30] *= 45.0
    y = _run(oracle.LoudNorm(ch), x, 50000)
    assert y.shape == x.shape
    tp = 10 ** (-2.0 / 20)
    assert np.abs(y).max() <= tp                                         # target_tp is a hard ceiling (imp.rs:1418-1423)
    m_in, m_out = oracle.EbuR128(ch, RATE), oracle.EbuR128(ch, RATE)
    m_in.add_frames(x.reshape(-1)); m_out.add_frames(y.reshape(-1))
    assert abs(m_out.loudness_global() + 24.0) < abs(m_in.loudness_global() + 24.0)


def test_chunking_does_not_matter(oracle):
    x = _tone(3.9, 2, amp=0.1)
    x[int(3.3 * RATE)] = 0.99
    a = _run(oracle.LoudNorm(2), x, 7777)
    b = _run(oracle.LoudNorm(2), x, 400000)
    assert (a == b).all()
