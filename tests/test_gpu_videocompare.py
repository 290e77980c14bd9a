"""GPU parity tests for videocompare's Blockhash path through the C ABI: bit-exact 64-bit hashes and distances
against the oracle (integer arithmetic), including the BASELINE config-5 shape (a batch of concurrent 4K streams)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frames(rng, n, w, h, c):
    return rng.integers(0, 256, (n, h, w * c), dtype=np.uint8)


@pytest.mark.parametrize("w,h,c", [(64, 64, 4), (320, 240, 4), (1920, 1080, 4), (3840, 2160, 4), (40, 48, 3), (1280, 720, 3), (8, 8, 4), (24, 16, 4)])
def test_hash_frame_matches_oracle(ctx, oracle, w, h, c):
    rng = np.random.default_rng(w * 7 + h + c)
    f = _frames(rng, 1, w, h, c)[0]
    if c == 4:
        f[:, 3::4] = np.where(rng.random((h, w)) < 0.1, 0, 255)   # some fully transparent pixels (count as 765)
    # give the blocks structure so that the medians separate them
    f[: h // 2] //= 3
    fmt = "RGBA" if c == 4 else "RGB"
    assert ctx.videocompare_hash_frame(f, w * c, w, h, fmt) == oracle.blockhash(f, w, h, w * c, c)


def test_padded_rows_and_solid_frames(ctx, oracle):
    rng = np.random.default_rng(2)
    w, h = 200, 104
    f = np.zeros((h, w * 4 + 36), np.uint8)
    f[:, : w * 4] = rng.integers(0, 256, (h, w * 4), dtype=np.uint8)
    f[:, w * 4:] = 0xEE   # padding must not be hashed
    assert ctx.videocompare_hash_frame(f, w * 4 + 36, w, h, "RGBA") == oracle.blockhash(f, w, h, w * 4 + 36, 4)
    red = np.zeros((240, 320 * 4), np.uint8); red[:, 0::4] = 255; red[:, 3::4] = 255
    hr = ctx.videocompare_hash_frame(red, 1280, 320, 240, "RGBA")
    assert hr == oracle.blockhash(red, 320, 240, 1280, 4)
    assert ctx.videocompare_distance(hr, ctx.videocompare_hash_frame(red.copy(), 1280, 320, 240, "RGBA")) == 0.0   # reference test 1
    snow = rng.integers(0, 256, (240, 320), dtype=np.uint8).repeat(4, axis=1); snow[:, 3::4] = 255
    assert ctx.videocompare_distance(hr, ctx.videocompare_hash_frame(snow, 1280, 320, 240, "RGBA")) > 0                # reference test 2
    white = np.full((64, 256), 255, np.uint8)
    assert ctx.videocompare_hash_frame(white, 256, 64, 64, "RGBA") == 0xFFFFFFFFFFFFFFFF


def test_config5_batch_of_4k_streams_on_device(ctx, oracle):
    """32 streams per GPU, one comparison per stream per frame: 16 reference + 16 secondary 4K RGBA frames hashed in one
    launch; distances equal the oracle's; identical pairs give exactly 0.0."""
    rng = np.random.default_rng(9)
    n, w, h = 16, 3840, 2160
    ref = np.zeros((n, h, w * 4), np.uint8)
    for k in range(n):
        base = rng.integers(0, 256, (h // 40, w // 40, 4), dtype=np.uint8).repeat(40, axis=0).repeat(40, axis=1)
        ref[k] = base.reshape(h, w * 4)
    sec = ref.copy()
    for k in range(0, n, 2):    # every other stream differs in one quadrant
        sec[k, : h // 2, : w * 2] = 255 - sec[k, : h // 2, : w * 2]
    both = np.concatenate([ref, sec]).reshape(-1)
    d = ctx.alloc(both.nbytes)
    try:
        ctx.h2d(d, both)
        hashes = ctx.videocompare_hash_frames_device(d, h * w * 4, w * 4, 2 * n, w, h, "RGBA")
    finally:
        ctx.free(d)
    for k in range(n):
        er, es = oracle.blockhash(ref[k], w, h, w * 4, 4), oracle.blockhash(sec[k], w, h, w * 4, 4)
        assert hashes[k] == er and hashes[n + k] == es
        dist = ctx.videocompare_distance(hashes[k], hashes[n + k])
        assert dist == oracle.hash_distance(er, es)
        assert (dist == 0.0) == (k % 2 == 1)


@pytest.mark.parametrize("w,h,c", [(60, 15, 4), (1921, 1083, 4), (3841, 2161, 4), (101, 57, 3), (1366, 768, 3), (9, 8, 4)])
def test_blockhash_any_size_matches_oracle(ctx, oracle, w, h, c):
    """MI355_FLAG_BLOCKHASH_ANY_SIZE: frames that do not divide into 8 x 8 whole blocks take the crate's floating-point path
    (video/videofx/src/videocompare/hashed_image.rs:37-44 -> image_hasher blockhash_slow, restated from memory: parity unpinned).
    Every pixel whole to block (floor(x / (w/8)), floor(y / (h/8))), f32 block sums accumulated in pixel order: one lane per
    block on the device, bit-exact against the C restatement - including 4K+1, where a block sum passes 2^24 and the order of
    the additions shows in its low bits. Off by default: the same call is refused without the flag."""
    import mi355fx
    rng = np.random.default_rng(w + 3 * h + c)
    f = _frames(rng, 1, w, h, c)[0]
    if c == 4:
        f[:, 3::4] = np.where(rng.random((h, w)) < 0.1, 0, 255)
    f[: h // 2] //= 3
    fmt = "RGBA" if c == 4 else "RGB"
    with pytest.raises(mi355fx.Mi355Error) as e:
        ctx.videocompare_hash_frame(f, w * c, w, h, fmt)
    assert e.value.status == mi355fx.ERR_UNSUPPORTED
    ctx.set_flag(mi355fx.FLAG_BLOCKHASH_ANY_SIZE, 1)
    got = ctx.videocompare_hash_frame(f, w * c, w, h, fmt)
    assert got == oracle.blockhash(f, w, h, w * c, c)
    assert ctx.videocompare_distance(got, ctx.videocompare_hash_frame(f.copy(), w * c, w, h, fmt)) == 0.0
    # sizes that DO divide still take the integer path, flag or not
    g = _frames(rng, 1, 64, 64, c)[0]
    assert ctx.videocompare_hash_frame(g, 64 * c, 64, 64, fmt) == oracle.blockhash(g, 64, 64, 64 * c, c)


def test_unsupported_and_errors(ctx):
    import mi355fx
    f = np.zeros((16, 64), np.uint8)
    with pytest.raises(mi355fx.Mi355Error) as e:      # Dssim has no 64-bit hash (mi355_dssim_* is its API)
        ctx.videocompare_hash_frame(f, 64, 16, 16, "RGBA", "dssim")
    assert e.value.status == mi355fx.ERR_UNSUPPORTED
    with pytest.raises(mi355fx.Mi355Error) as e:      # fast path only
        ctx.videocompare_hash_frame(np.zeros((16, 60), np.uint8), 60, 15, 16, "RGBA")
    assert e.value.status == mi355fx.ERR_UNSUPPORTED
    with pytest.raises(mi355fx.Mi355Error) as e:      # pad template: RGB / RGBA only
        ctx.videocompare_hash_frame(f, 64, 16, 16, "BGRx")
    assert e.value.status == mi355fx.ERR_INVALID_ARG
    assert ctx.videocompare_distance(1, 2, "dssim") < 0
    assert ctx.videocompare_distance(0b1011, 0b0001, "gradient") == 2.0


@pytest.mark.parametrize("algo,bits", [("mean", 64), ("gradient", 64), ("vertgradient", 64), ("doublegradient", 40)])
@pytest.mark.parametrize("w,h,c", [(640, 360, 4), (3840, 2160, 4), (101, 57, 3), (8, 9, 4), (5, 4, 3)])
def test_resize_based_hashes_match_oracle(ctx, oracle, algo, bits, w, h, c):
    """image_hasher Mean / Gradient / VertGradient / DoubleGradient: integer luma, Lanczos3 resize in the image crate's
    f32 accumulation order, bit rule — bit-exact against the C restatement (same libm sinf for the weights), including
    up-sampling cases (frames smaller than the hash grid)."""
    rng = np.random.default_rng(w * 3 + h + c)
    yy, xx = np.mgrid[0:h, 0:w]
    f = np.zeros((h, w, c), np.uint8)
    f[..., 0] = (xx * 255 // max(w - 1, 1)) ^ rng.integers(0, 32, (h, w))
    f[..., 1] = (yy * 255 // max(h - 1, 1))
    f[..., 2] = rng.integers(0, 256, (h, w))
    if c == 4:
        f[..., 3] = 255
    frame = f.reshape(h, w * c)
    exp, nb, small = oracle.imghash(frame, w, h, w * c, c, algo)
    assert nb == bits
    got = ctx.videocompare_hash_frame(frame, w * c, w, h, "RGBA" if c == 4 else "RGB", algo)
    assert got == exp, (hex(got), hex(exp))
    other = frame[::-1].copy()
    assert ctx.videocompare_distance(got, ctx.videocompare_hash_frame(other, w * c, w, h, "RGBA" if c == 4 else "RGB", algo), algo) == \
        oracle.hash_distance(exp, oracle.imghash(other, w, h, w * c, c, algo)[0])
