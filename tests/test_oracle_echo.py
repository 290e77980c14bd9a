"""rsaudioecho oracle vs a direct pure-Python restatement of the reference loop
(audio/audiofx/src/audioecho/imp.rs:69-85, ring_buffer.rs:37-82). The reference has no echo test
(SURVEY.md §4): parity is pinned by source semantics only."""
import numpy as np
import pytest


def py_echo(data, ring, pos, delay, intensity, feedback):
    size = len(ring)
    assert size >= delay and size != 0
    read_pos = (size - delay + pos) % size
    write_pos = pos % size
    out = data.copy()
    for i in range(len(data)):
        e = ring[read_pos]
        inp = float(data[i])
        o = inp + intensity * e
        ring[write_pos] = inp + feedback * e
        out[i] = data.dtype.type(o)
        write_pos = (write_pos + 1) % size
        read_pos = (read_pos + 1) % size
    return out, write_pos


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("delay_ns,max_ns", [(250_000_000, 10 ** 9), (500 * 10 ** 9, 10 ** 9), (0, 10 ** 8), (1, 10 ** 9)])
def test_echo_matches_python_loop(oracle, dtype, delay_ns, max_ns):
    rate, ch = 8000, 2
    e = oracle.Echo(max_ns, rate, ch)
    ring = [0.0] * e.ring_len
    pos = 0
    rng = np.random.default_rng(4)
    d = oracle.lib().oracle_echo_delay_samples(delay_ns, max_ns, rate, ch)
    for n in (5, 3000, 0, 7001):
        x = rng.standard_normal(n).astype(dtype)
        exp, pos = py_echo(x, ring, pos, d, 0.6, 0.4)
        got = x.copy()
        e.process(got, delay_ns, 0.6, 0.4)
        assert got.tobytes() == exp.tobytes()
        assert e.pos == pos
    assert np.array(ring).tobytes() == e.ring[: e.ring_len].tobytes()


def test_ring_sizing_and_delay_quirks(oracle):
    L = oracle.lib()
    # setup: size = (max_delay*rate).seconds() * channels (imp.rs:250-251)
    assert L.oracle_echo_ring_len(10 ** 9, 48000, 2) == 96000
    assert L.oracle_echo_ring_len(1_500_000_000, 44100, 1) == 66150
    # default delay 500 s is clamped to max-delay 1 s (imp.rs:34,207)
    assert L.oracle_echo_delay_samples(500 * 10 ** 9, 10 ** 9, 48000, 2) == 96000
    # delay counts interleaved samples, floor of ns*ch*rate/1e9 (imp.rs:74-77)
    assert L.oracle_echo_delay_samples(250_000_000, 10 ** 9, 48000, 2) == 24000
    assert L.oracle_echo_delay_samples(10_417, 10 ** 9, 48000, 2) == 1  # odd: mixes channels


def test_delay_zero_and_full_behave_as_ring_length(oracle):
    """ring_buffer.rs:44-45: delay 0 and delay == size both read the slot about to be overwritten."""
    a, b = oracle.Echo(10 ** 7, 1000, 1), oracle.Echo(10 ** 7, 1000, 1)  # ring of 10
    x = np.arange(1, 36, dtype=np.float32)
    xa, xb = x.copy(), x.copy()
    a.process(xa, 0, 0.5, 0.25)
    b.process(xb, 10 ** 7, 0.5, 0.25)
    assert xa.tobytes() == xb.tobytes()
    assert xa[0] == 1.0 and xa[10] == np.float32(11 + 0.5 * 1.0)
