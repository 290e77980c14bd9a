"""Host logic of the run-time kernel choice (csrc/autopick.hpp) against a scripted device, through the C ABI test hook
mi355_selftest_autopick - no GPU needed. The same policy drives colorlut, the fused chain and hsvfilter on the device
(tests/test_gpu_parity.py::test_colorlut_auto_*, tools/stress_auto.py)."""
import ctypes as C

import numpy as np

NV = 16_588_800  # pixel groups of an 8 x 4K launch


def run(mi355lib, n_vec, ms_c, ms_t, lag=0):
    n = len(n_vec)
    nv = (C.c_uint64 * n)(*[int(v) for v in n_vec])
    a = (C.c_double * n)(*[float(v) for v in ms_c])
    b = (C.c_double * n)(*[float(v) for v in ms_t])
    kind = (C.c_int * n)()
    meas = (C.c_int * n)()
    f = mi355lib.mi355_selftest_autopick
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    assert f(n, nv, a, b, lag, kind, meas) == 0
    return np.array(kind[:]), np.array(meas[:])


def test_learning_phase_then_faster_kind(mi355lib):
    n = 60
    kind, meas = run(mi355lib, [NV] * n, [0.25] * n, [0.12] * n)
    assert list(kind[:4]) == [0, 0, 1, 1] and list(meas[:4]) == [1, 1, 1, 1]
    assert (kind[4:] == 1).all()           # table is faster: it serves everything until the first probe (call 64+)
    kind, _ = run(mi355lib, [NV] * n, [0.25] * n, [0.60] * n)
    assert list(kind[:4]) == [0, 0, 1, 1] and (kind[4:] == 0).all()


def test_first_launch_of_each_kind_is_discarded(mi355lib):
    """One-off costs in the first compute / table launch must not decide: here they would point the wrong way."""
    n = 40
    ms_c = [5.0, 0.25] + [0.25] * (n - 2)      # first compute launch inflated
    ms_t = [0.12] * 2 + [0.05, 0.6] + [0.6] * (n - 4)   # first table launch (call 2) flatteringly fast, real ones slow
    kind, _ = run(mi355lib, [NV] * n, ms_c, ms_t)
    assert (kind[4:] == 0).all()


def test_probe_of_the_other_kind_backs_off(mi355lib):
    n = 2600
    kind, meas = run(mi355lib, [NV] * n, [0.25] * n, [0.12] * n)
    other = [i for i in range(4, n) if kind[i] == 0]
    assert other, "the kind not in use must be tried again"
    # a probe is two consecutive launches of the other kind: the first warms its working set and is not measured, the second is
    assert len(other) % 2 == 0 and all(b == a + 1 for a, b in zip(other[0::2], other[1::2]))
    assert all(meas[i] == 0 for i in other[0::2]) and all(meas[i] == 1 for i in other[1::2])
    probes = other[0::2]
    gaps = np.diff([4] + probes)
    assert 60 <= gaps[0] <= 70
    assert all(g2 >= g1 for g1, g2 in zip(gaps, gaps[1:])) and gaps[-1] <= 1030   # 64, 128, 256, 512, 1024, 1024 ...


def test_content_change_flips_within_a_sample_interval(mi355lib):
    n, change = 400, 200
    ms_t = [0.12] * change + [1.2] * (n - change)   # the table kernel becomes 10x slower (noise)
    kind, meas = run(mi355lib, [NV] * n, [0.4] * n, ms_t, lag=3)
    after = kind[change:]
    first_compute = int(np.argmax(after == 0))
    assert after[first_compute] == 0 and first_compute <= 8 + 7 + 1   # next sample (every 8th launch) + read-back lag
    assert (after[first_compute:first_compute + 60] == 0).all()
    # and back when the content becomes friendly again. Noise that made the table kernel 10 x slower is a DECISIVE answer (round 6):
    # its next probe is 1024 launches away - unless the kind in use says that the content has changed: the interpolating kernels
    # get faster on friendly content too (uniform noise 0.40 ms, natural frames 0.13), and a sample of the kind in use that moves by
    # more than 25 % brings the probe forward to now
    ms_t2 = ms_t + [0.12] * 400
    ms_c2 = [0.4] * n + [0.13] * 400
    kind2, _ = run(mi355lib, [NV] * (n + 400), ms_c2, ms_t2, lag=3)
    back = kind2[n:]
    first_table = int(np.argmax(back == 1))
    assert back[first_table] == 1 and first_table <= 32 + 3 + 2      # next sample of the kind in use + read-back lag + the probe's two launches
    assert kind2[-1] == 1 and (back[first_table + 8:] == 1).sum() >= len(back) - first_table - 8 - 4
    # a change the kind in use cannot see (its own time stays) is found by the long-period probe
    kind3, _ = run(mi355lib, [NV] * (n + 1400), [0.4] * (n + 1400), ms_t + [0.12] * 1400, lag=3)
    assert kind3[n + 300] == 0 and kind3[-1] == 1


def test_a_cold_first_launch_does_not_bias_the_probe(mi355lib):
    """What the kind not in use keeps in the caches is gone by the time it is probed: its first launch after a pause is slow
    (the table kernel: 0.156 ms cold against 0.10 warm, the interpolating kernel 0.131). A one-launch probe would measure the
    cold launch and never let the table back in once something had flipped the choice; the two-launch probe measures the
    second, warm one."""
    n = 600
    ms_c = [0.131] * n
    ms_t = [0.100] * n
    for i in range(8, 100):
        ms_t[i] = 0.20            # a disturbance while the table is in use (content, a neighbour): the choice flips to compute
    kind0, _ = run(mi355lib, [NV] * n, ms_c, ms_t)
    assert (kind0[40:100] == 0).sum() > 50
    # from then on every table launch that follows a compute launch is cold, every table launch that follows a table launch warm
    kind = None
    for _ in range(3):        # the scripted times depend on the decisions: iterate to the fixed point
        prev = kind0 if kind is None else kind
        ms_t2 = list(ms_t)
        for i in range(100, n):
            ms_t2[i] = 0.156 if prev[i - 1] == 0 else 0.100
        kind, _ = run(mi355lib, [NV] * n, ms_c, ms_t2)
    assert (kind[300:] == 1).sum() > 280, "the table must win back once the disturbance is over"


def test_hysteresis_no_flip_flop_on_equal_kernels(mi355lib):
    rng = np.random.default_rng(0)
    n = 3000
    ms_c = 0.200 * (1 + rng.uniform(-0.01, 0.01, n))
    ms_t = 0.200 * (1 + rng.uniform(-0.01, 0.01, n))
    kind, _ = run(mi355lib, [NV] * n, ms_c, ms_t)
    steady = kind[4:]
    # ignore the two-launch probes: count changes of the kind that serves runs of >= 3 launches
    runs = [k for k, g, h in zip(steady[:-2], steady[1:-1], steady[2:]) if k == g == h]
    assert np.count_nonzero(np.diff(runs)) <= 2


def test_size_jump_restarts_learning(mi355lib):
    nv = [NV] * 50 + [NV // 8] * 50 + [NV // 7] * 10
    n = len(nv)
    kind, meas = run(mi355lib, nv, [0.25] * n, [0.12] * n, lag=0)
    assert list(kind[:4]) == [0, 0, 1, 1]
    assert list(kind[50:54]) == [0, 0, 1, 1] and list(meas[50:54]) == [1, 1, 1, 1]   # > 2x smaller: start over
    assert (kind[54:100] == 1).all()
    assert (kind[100:] == 1).all()          # a 14 % change does not


def test_sampling_cadence(mi355lib):
    n = 400
    _, meas = run(mi355lib, [NV] * n, [0.25] * n, [0.12] * n, lag=1)
    steady = np.nonzero(meas[8:64])[0]
    assert len(steady) and (np.diff(steady) == 8).all()          # big launches: every 8th
    small = NV // 64
    _, meas = run(mi355lib, [small] * n, [0.25] * n, [0.12] * n, lag=1)
    steady = np.nonzero(meas[8:64])[0]
    assert len(steady) and (np.diff(steady) == 32).all()         # small launches: every 32nd


def test_a_measurement_is_never_waited_for(mi355lib):
    """A host that never finds the events complete (huge lag) is never blocked: the first measured launch stays in flight,
    nothing else is measured, and every launch runs the kernel being learned (the interpolating one)."""
    n = 300
    kind, meas = run(mi355lib, [NV] * n, [0.4] * n, [0.12] * n, lag=10 ** 6)
    assert (kind == 0).all() and meas[0] == 1 and meas[1:].sum() == 0


def test_learning_spans_more_launches_when_results_arrive_late(mi355lib):
    """Read-back lag L: no launch is measured while another measurement is in flight, the four learning measurements are
    L + 1 launches apart, and the steady state is reached all the same."""
    for lag in (0, 1, 5, 17):
        n = 300
        kind, meas = run(mi355lib, [NV] * n, [0.4] * n, [0.12] * n, lag=lag)
        m = np.nonzero(meas)[0]
        assert list(m[:4]) == [0, lag + 1, 2 * (lag + 1), 3 * (lag + 1)], (lag, m[:6])
        assert (np.diff(m) >= lag + 1).all()
        steady = kind[5 * (lag + 1):200]
        assert kind[0] == 0 and (steady == 1).sum() >= len(steady) - 6, (lag, kind[:40])   # all but the odd two-launch probe of the other kind


def test_property_steady_timings_pick_the_faster_kind(mi355lib):
    """Hypothesis: for any constant pair of timings at least 10 % apart, any launch size and any read-back lag, every
    launch after the learning phase runs the faster kind except two-launch probes of the other one (the first launch warms
    and is not measured, the second is measured unless another measurement is still in flight), at least 64 launches apart."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=150, deadline=None)
    @given(tc=st.floats(0.01, 10.0), ratio=st.floats(1.1, 20.0), table_faster=st.booleans(), nv=st.integers(16384, 1 << 27),
           lag=st.integers(0, 40), n=st.integers(80, 1500))
    def prop(tc, ratio, table_faster, nv, lag, n):
        tt = tc / ratio if table_faster else tc * ratio
        kind, meas = run(mi355lib, [nv] * n, [tc] * n, [tt] * n, lag=lag)
        want = 1 if table_faster else 0
        learned = 5 * (lag + 1)         # four measurements, each readable `lag` calls later; nothing is ever waited for
        if n <= learned:
            return
        assert kind[0] == 0
        other = [i for i in range(learned, n - 1) if kind[i] != want]
        starts = [i for i in other if i - 1 not in other]
        assert all(i + 1 in other or i + 1 >= n - 1 for i in starts) and len(other) <= 2 * len(starts), "probes of the slower kind come in pairs"
        assert all(meas[i] == 0 for i in starts), "the warming half of a probe is never measured"
        assert all(b - a >= 60 for a, b in zip([3 * (lag + 1)] + starts, starts)), starts[:5]

    prop()


def test_a_decisive_answer_is_not_asked_again_for_1024_launches(mi355lib):
    """Uniform noise: interpolating 0.40 ms, table 1.2 ms. Rounds 4-5 probed the table after 64, 128, 256, 512, 1024 launches - ten
    launches at three times the cost in the first two thousand, 10 % of BENCH_r05's 32-launch uniform leg when two of them fell into
    it (auto 14.6 k against 16.3 k pinned). A decisive answer now waits the longest period at once: at most one probe (two launches)
    per 1024 launches after the learning phase, i.e. <= 0.2 % of the launches and <= 0.6 % of the time."""
    n = 4300
    kind, meas = run(mi355lib, [NV] * n, [0.40] * n, [1.2] * n, lag=2)
    assert kind[0] == 0 and (kind[:16] == 1).sum() >= 2      # the learning phase (spread over more launches by the read-back lag)
    table_launches = [i for i in range(16, n) if kind[i] == 1]
    assert len(table_launches) <= 2 * 4 and len(table_launches) >= 2 * 3        # probes at ~1030, ~2055, ~3080, ~4105
    assert table_launches[0] >= 1000
    for w0 in range(16, n - 1024, 97):
        assert sum(1 for i in table_launches if w0 <= i < w0 + 1024) <= 2
    # ... and a close call (10 % apart) still climbs the 64, 128, ... ladder
    kind, _ = run(mi355lib, [NV] * 600, [0.110] * 600, [0.100] * 600, lag=2)
    probes = [i for i in range(16, 600) if kind[i] == 0][0::2]
    assert 60 <= probes[0] <= 84 and len(probes) >= 3
