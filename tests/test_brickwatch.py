"""Content-watch policy of the interpolating colorlut path (csrc/brickwatch.hpp) against scripted streams, on the CPU:
mi355_selftest_brickwatch feeds the policy the miss / slow step fractions each call would show on the 32-set and on the
64-set brick kernel and reports the level (0 brick kernel 32 sets, 1 brick kernel 64 sets, 2 three-pass kernel) of every call."""
import ctypes as C

import numpy as np
import pytest


def _run(mi355lib, miss0, slow0, miss1, slow1, lag=2):
    n = len(miss0)
    arr = lambda v: (C.c_double * n)(*[float(x) for x in v])
    out = (C.c_int * n)()
    rc = mi355lib.mi355_selftest_brickwatch(n, arr(miss0), arr(slow0), arr(miss1), arr(slow1), lag, out)
    assert rc == 0
    return list(out)


def test_coherent_content_stays_on_the_32_set_kernel(mi355lib):
    n = 400
    lv = _run(mi355lib, [0.045] * n, [0.0] * n, [0.04] * n, [0.0] * n)
    assert set(lv) == {0}


def test_edge_heavy_content_moves_to_64_sets_and_stays(mi355lib):
    n = 400
    lv = _run(mi355lib, [0.2] * n, [0.08] * n, [0.1] * n, [0.0] * n)
    first1 = lv.index(1)
    assert 4 <= first1 <= 8                      # one snapshot (4 launches) + lag
    assert 2 not in lv
    # level 0 is probed again after 64 launches, found bad, then after 128 ...
    probes = [i for i in range(first1 + 1, n) if lv[i] == 0 and lv[i - 1] == 1]
    assert len(probes) >= 2 and probes[1] - probes[0] > probes[0] - first1


def test_noise_climbs_to_the_three_pass_kernel(mi355lib):
    n = 300
    lv = _run(mi355lib, [1.0] * n, [1.0] * n, [1.0] * n, [1.0] * n)
    assert lv[0] == 0 and 1 in lv and 2 in lv
    assert lv.index(2) <= 16
    assert lv.count(2) > 0.8 * n                  # probes of level 1 are rare and short
    assert 0 not in lv[lv.index(2):]              # never all the way down while level 1 is bad


def test_stream_that_calms_down_walks_back_to_level_0(mi355lib):
    n = 600
    bad = 60
    miss = [1.0] * bad + [0.03] * (n - bad)
    slow = [1.0] * bad + [0.0] * (n - bad)
    lv = _run(mi355lib, miss, slow, miss, slow)
    assert 2 in lv[:bad]
    assert lv[-1] == 0
    i1 = max(i for i, v in enumerate(lv) if v == 2)
    assert i1 < bad + 64 + 80                     # left the three-pass kernel at the first probe after the content changed


def test_snapshot_lag_does_not_break_the_ladder(mi355lib):
    n = 300
    for lag in (0, 1, 7, 20):
        lv = _run(mi355lib, [1.0] * n, [1.0] * n, [0.1] * n, [0.0] * n, lag=lag)
        assert lv[-1] == 1 and 2 not in lv


@pytest.mark.parametrize("seed", range(5))
def test_levels_are_always_valid_under_random_content(mi355lib, seed):
    rng = np.random.default_rng(seed)
    n = 500
    m0, s0, m1, s1 = rng.random(n), rng.random(n) * 0.2, rng.random(n) * 0.6, rng.random(n) * 0.08
    lv = _run(mi355lib, m0, np.minimum(s0, m0), m1, np.minimum(s1, m1), lag=int(rng.integers(0, 6)))
    assert set(lv) <= {0, 1, 2}
    assert lv[0] == 0


def test_probes_are_single_launches_even_when_the_host_runs_far_ahead(mi355lib):
    """A host that enqueues 150 launches ahead of the device sees every snapshot 150 calls late. The first move up still
    takes that long, but afterwards a probe of the lower level is ONE launch - not one queue depth of launches on the kernel
    believed slower."""
    n, lag = 1500, 150
    lv = _run(mi355lib, [0.2] * n, [0.05] * n, [0.08] * n, [0.0] * n, lag=lag)
    first1 = lv.index(1)
    assert first1 <= lag + 6
    tail = lv[first1:]
    zeros = [i for i, v in enumerate(tail) if v == 0]
    assert 2 <= len(zeros) <= 12                                   # a handful of probes in 1300 launches
    assert all(b - a > 1 for a, b in zip(zeros, zeros[1:]))        # never two in a row
    assert 2 not in lv


SHARED1 = 1 << 16


def test_shared_cache_thresholds(mi355lib):
    """Level 1 = the block-shared cache (round 4): level 0 leaves above 30 % miss steps (35 % before), level 1 holds up to 50 %
    miss / 30 % slow steps (32 % / 2 % for the per-wave 64-set cache)."""
    n = 300
    # 32 % miss steps at level 0: stays with the per-wave level 1, moves with the shared one
    assert set(_run(mi355lib, [0.32] * n, [0.0] * n, [0.1] * n, [0.0] * n)) == {0}
    lv = _run(mi355lib, [0.32] * n, [0.0] * n, [0.1] * n, [0.0] * n, lag=2 | SHARED1)
    assert lv[-1] == 1 and 2 not in lv
    # 45 % miss / 20 % slow steps at level 1: too much for the per-wave cache, fine for the shared one
    lv = _run(mi355lib, [0.9] * n, [0.2] * n, [0.45] * n, [0.2] * n)
    assert lv[-1] == 2
    lv = _run(mi355lib, [0.9] * n, [0.2] * n, [0.45] * n, [0.2] * n, lag=2 | SHARED1)
    assert lv[-1] == 1 and 2 not in lv
    # 60 % miss steps at level 1: on to the three-pass kernel either way
    lv = _run(mi355lib, [1.0] * n, [1.0] * n, [0.6] * n, [0.35] * n, lag=2 | SHARED1)
    assert lv[-1] == 2


def test_small_launches_never_run_below_their_lowest_level(mi355lib):
    """min_level 1 (one or two frames per launch with the shared cache as level 1): level 0 is neither used nor probed, the
    ladder above it works as usual (noise climbs to the three-pass kernel, calm content comes back to level 1)."""
    n = 500
    flags = 2 | SHARED1 | (1 << 17)
    lv = _run(mi355lib, [0.02] * n, [0.0] * n, [0.02] * n, [0.0] * n, lag=flags)
    assert set(lv) == {1}
    miss1 = [0.9] * 100 + [0.05] * 400
    slow1 = [0.6] * 100 + [0.0] * 400
    lv = _run(mi355lib, [1.0] * n, [1.0] * n, miss1, slow1, lag=flags)
    assert 0 not in lv and 2 in lv[:40] and lv[-1] == 1
