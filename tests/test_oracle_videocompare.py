"""CPU tests of the videocompare oracle (image_hasher Blockhash restatement; parity unpinned for arbitrary content,
see oracle/videocompare_oracle.c). Pinned here: the properties the reference's own tests assert
(video/videofx/tests/videocompare.rs) and the algorithm's defining cases."""
import numpy as np
import pytest


def _solid(w, h, rgba):
    f = np.zeros((h, w * 4), np.uint8)
    for c in range(4):
        f[:, c::4] = rgba[c]
    return f


def test_identical_frames_have_distance_zero(oracle):
    """test_can_find_similar_frames: videotestsrc pattern=red on both pads -> distance <= 0.0."""
    red = _solid(320, 240, (255, 0, 0, 255))
    h = oracle.blockhash(red, 320, 240, 320 * 4, 4)
    assert oracle.hash_distance(h, oracle.blockhash(red.copy(), 320, 240, 320 * 4, 4)) == 0.0


def test_snow_is_far_from_red(oracle):
    """test_do_not_send_message_when_image_not_found: snow vs red -> distance > 0 (no message at threshold 0)."""
    red = _solid(320, 240, (255, 0, 0, 255))
    snow = np.random.default_rng(1).integers(0, 256, (240, 320), dtype=np.uint8).repeat(4, axis=1)
    snow[:, 3::4] = 255
    assert oracle.hash_distance(oracle.blockhash(red, 320, 240, 1280, 4), oracle.blockhash(snow, 320, 240, 1280, 4)) > 0


def test_uniform_frames_tie_rule(oracle):
    """All blocks equal the median: bits are set only when the median is above half scale (765*area/2)."""
    assert oracle.blockhash(_solid(64, 64, (255, 255, 255, 255)), 64, 64, 256, 4) == 0xFFFFFFFFFFFFFFFF
    assert oracle.blockhash(_solid(64, 64, (10, 10, 10, 255)), 64, 64, 256, 4) == 0
    # transparent pixels count as white (sum_px: a == 0 -> 765)
    assert oracle.blockhash(_solid(64, 64, (0, 0, 0, 0)), 64, 64, 256, 4) == 0xFFFFFFFFFFFFFFFF


def test_known_pattern_bits(oracle):
    """Left half bright, right half dark: in every row of blocks the 4 left blocks exceed the band median."""
    f = _solid(64, 64, (0, 0, 0, 255))
    f[:, : 32 * 4] = 200
    f[:, 3::4] = 255
    h = oracle.blockhash(f, 64, 64, 256, 4)
    # upper median of 16 bright + 16 dark values is a bright value: nothing is strictly greater, and the tie rule
    # (median > half scale) decides: bright = 600*64 = 38400 > 765*64/2 = 24480 -> bright bits set
    assert h == 0x0F0F0F0F0F0F0F0F  # bit index = block_row*8 + block_col: columns 0..3 of every row


def test_rgb_and_stride(oracle):
    rng = np.random.default_rng(5)
    rgb = rng.integers(0, 256, (48, 40 * 3), dtype=np.uint8)
    padded = np.zeros((48, 40 * 3 + 13), np.uint8)
    padded[:, : 120] = rgb
    assert oracle.blockhash(rgb, 40, 48, 120, 3) == oracle.blockhash(padded, 40, 48, 133, 3)
    with pytest.raises(ValueError):
        oracle.blockhash(rgb, 7, 48, 120, 3)   # below 8 x 8 (sizes that are merely not divisible by 8 take the float path now)


def test_resize_based_hashes_basic_properties(oracle):
    """Mean/Gradient/VertGradient/DoubleGradient restatement (oracle/imghash_oracle.c): bit counts, a horizontal ramp has
    all Gradient bits set and no VertGradient bits, identical frames have distance 0."""
    w, h = 320, 200
    ramp = np.zeros((h, w, 4), np.uint8)
    ramp[..., 0] = ramp[..., 1] = ramp[..., 2] = (np.arange(w) * 255 // (w - 1))[None, :]
    ramp[..., 3] = 255
    f = ramp.reshape(h, w * 4)
    hg, nb, small = oracle.imghash(f, w, h, w * 4, 4, "gradient")
    assert nb == 64 and hg == 0xFFFFFFFFFFFFFFFF and small.shape == (8, 9)
    assert (np.diff(small.astype(int), axis=1) > 0).all()
    hv, nb, _ = oracle.imghash(f, w, h, w * 4, 4, "vertgradient")
    assert nb == 64 and hv == 0
    hd, nb, _ = oracle.imghash(f, w, h, w * 4, 4, "doublegradient")
    assert nb == 40 and hd == (1 << 20) - 1          # 5 rows x 4 gradient bits set, 4 x 5 vertical bits clear
    hm, nb, sm = oracle.imghash(f, w, h, w * 4, 4, "mean")
    assert nb == 64 and bin(hm).count("1") in (32, 40)  # right half (>= mean) of every row
    assert oracle.hash_distance(hg, oracle.imghash(f.copy(), w, h, w * 4, 4, "gradient")[0]) == 0.0
    # resizing a constant image returns the constant (weights are normalised)
    const = np.full((64, 64 * 3), 77, np.uint8)
    assert (oracle.imghash(const, 64, 64, 192, 3, "mean")[2] == 77).all()


def test_any_size_path_against_a_numpy_restatement(oracle):
    """The floating-point path (sizes not divisible by 8) restated a second time in numpy: every pixel whole to block
    (floor(x / (w/8)), floor(y / (h/8))) with f32 quotients, each block's f32 sum accumulated in row-major pixel order
    (np.add.accumulate in float32 is sequential), upper median of each band of 32, the crate's float comparison."""
    rng = np.random.default_rng(4)
    for w, h, c in ((63, 64, 4), (101, 57, 3), (1283, 721, 4), (3841, 2161, 4)):
        f = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
        f[: h // 2] //= 3
        if c == 4:
            f[..., 3] = np.where(rng.random((h, w)) < 0.1, 0, 255)
        s = f[..., :3].astype(np.uint32).sum(axis=2)
        if c == 4:
            s[f[..., 3] == 0] = 765
        bw, bh = np.float32(w) / np.float32(8), np.float32(h) / np.float32(8)
        bx = np.floor(np.arange(w, dtype=np.float32) / bw).astype(np.int64)
        by = np.floor(np.arange(h, dtype=np.float32) / bh).astype(np.int64)
        blocks = np.zeros(64, np.float32)
        for j in range(8):
            rows = s[by == j]
            for i in range(8):
                vals = rows[:, bx == i].reshape(-1).astype(np.float32)     # row-major within the block = pixel order
                blocks[j * 8 + i] = np.add.accumulate(vals, dtype=np.float32)[-1]
        half = np.float32(765.0) * bw * bh / np.float32(2.0)
        bits = 0
        for g in range(2):
            band = blocks[32 * g: 32 * g + 32]
            median = np.sort(band)[16]
            for i, v in enumerate(band):
                if v > median or (abs(v - median) < 1.0 and median > half):
                    bits |= 1 << (32 * g + i)
        assert oracle.blockhash(f.reshape(h, w * c), w, h, w * c, c) == bits, (w, h, c)
    with pytest.raises(ValueError):
        oracle.blockhash(np.zeros((7, 28), np.uint8), 7, 7, 28, 4)
