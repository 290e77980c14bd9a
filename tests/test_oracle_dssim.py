"""CPU tests of the Dssim restatement (oracle/dssim_restate.py; parity unpinned — see its header). Pinned: the
reference's own test property (identical frames -> distance <= 0.0, video/videofx/tests/videocompare.rs:140-182) and
the defining behaviour of a dissimilarity measure."""
import numpy as np


def _frame(rng, w, h):
    base = np.kron(rng.integers(0, 256, (h // 8, w // 8, 4), dtype=np.uint8), np.ones((8, 8, 1), np.uint8)).reshape(h, w * 4)
    base[:, 3::4] = 255
    return base


def test_identical_is_exactly_zero_like_the_reference_test():
    from oracle import dssim_restate as D
    red = np.zeros((240, 320 * 4), np.uint8); red[:, 0::4] = 255; red[:, 3::4] = 255
    a, b = D.DssimImage(red, 320, 240, 1280, 4), D.DssimImage(red.copy(), 320, 240, 1280, 4)
    assert len(a.scales) == 5
    assert D.compare(a, b) == 0.0 <= 0.0


def test_monotone_in_distortion_and_rgb_equals_opaque_rgba():
    from oracle import dssim_restate as D
    rng = np.random.default_rng(3)
    w, h = 128, 96
    base = _frame(rng, w, h)
    a = D.DssimImage(base, w, h, w * 4, 4)
    prev = 0.0
    for amp in (2, 8, 32, 96):
        n = np.clip(base.astype(int) + rng.integers(-amp, amp + 1, base.shape), 0, 255).astype(np.uint8)
        n[:, 3::4] = 255
        d = D.compare(a, D.DssimImage(n, w, h, w * 4, 4))
        assert d > prev
        prev = d
    rgb = base.reshape(h, w, 4)[:, :, :3].reshape(h, w * 3).copy()
    assert D.compare(a, D.DssimImage(rgb, w, h, w * 3, 3)) == 0.0


def test_scale_count_for_small_frames():
    from oracle import dssim_restate as D
    f = np.full((20, 40 * 4), 128, np.uint8)
    assert [s[0]["img"].shape for s in D.DssimImage(f, 40, 20, 160, 4).scales] == [(20, 40), (10, 20), (5, 10)]
