"""oracle/np_restate.py — second, independent CPU restatement (numpy float32, vectorised).

TEST INFRASTRUCTURE ONLY. Its single job is to cross-check oracle/*.c: per-pixel outputs of
hsvfilter / colorlut are not pinned by any reference test, so the C oracle is
pinned by (a) the reference's known-answer tests and (b) bit-for-bit agreement with this
independently written restatement over the exhaustive 2^24 colour domain.

Written from the reference source, not from the C oracle:
  video/hsv/src/hsvutils.rs:44-84,132-163      from_rgb / to_rgb
  video/hsv/src/hsvfilter/imp.rs:96-118        filter body
  video/colorlut/src/colorlut/imp.rs:431-539   apply_3d / apply_1d / sample / lerp / float_to_u8

numpy float32 arithmetic is IEEE single with no contraction; np.fmod is C fmodf.
"""
import numpy as np

F = np.float32


def _clamp_inherent(x, lo, hi):
    # f32::clamp: NaN passes through
    out = x.copy()
    out[x < lo] = lo
    out[x > hi] = hi
    return out


def _as_u8(x):
    # Rust `as u8`: trunc, saturate, NaN -> 0
    y = np.where(np.isnan(x), F(0), x)
    y = np.clip(y, F(0), F(255))
    return np.trunc(y).astype(np.uint8)


def from_rgb(r8, g8, b8):
    r = r8.astype(F) / F(255.0)
    g = g8.astype(F) / F(255.0)
    b = b8.astype(F) / F(255.0)
    mx = np.maximum(np.maximum(r8, g8), b8)
    mn = np.minimum(np.minimum(r8, g8), b8)
    value = mx.astype(F) / F(255.0)
    chroma = value - mn.astype(F) / F(255.0)
    eps = F(0.00001)
    with np.errstate(divide="ignore", invalid="ignore"):
        h_r = F(60.0) * ((g - b) / chroma)
        h_g = F(60.0) * (F(2.0) + ((b - r) / chroma))
        h_b = F(60.0) * (F(4.0) + ((r - g) / chroma))
    is_r = np.abs(value - r) < eps
    is_g = np.abs(value - g) < eps
    is_b = np.abs(value - b) < eps
    hue = np.where(chroma == F(0), F(0), np.where(is_r, h_r, np.where(is_g, h_g, np.where(is_b, h_b, F(0))))).astype(F)
    hue = np.where(hue < F(0), hue + F(360.0), hue).astype(F)
    with np.errstate(divide="ignore", invalid="ignore"):
        sat = np.where(value == F(0), F(0), chroma / value).astype(F)
    return np.fmod(hue, F(360.0)).astype(F), _clamp_inherent(sat, F(0), F(1)), _clamp_inherent(value, F(0), F(1))


def to_rgb(h, s, v):
    c = v * s
    hp = h / F(60.0)
    x = c * (F(1.0) - np.abs(np.fmod(hp, F(2.0)) - F(1.0)))
    z = np.zeros_like(c)
    conds = [hp < 0, hp <= 1, hp <= 2, hp <= 3, hp <= 4, hp <= 5, hp <= 6]
    rp = np.select(conds, [z, c, x, z, z, x, c], default=z)
    gp = np.select(conds, [z, x, c, c, x, z, z], default=z)
    bp = np.select(conds, [z, z, z, x, c, c, x], default=z)
    m = v - c
    out = []
    for p in (rp, gp, bp):
        out.append(_as_u8(_clamp_inherent(((p + m) * F(255.0)).astype(F), F(0), F(255))))
    return out


def hsvfilter_rgb(r8, g8, b8, settings):
    hs, sm, so, vm, vo = [F(t) for t in settings]
    h, s, v = from_rgb(r8, g8, b8)
    with np.errstate(invalid="ignore"):
        h = np.fmod(h + hs, F(360.0)).astype(F)
        h = np.where(h < 0, h + F(360.0), h).astype(F)
        # crate Clamp trait: max then min, NaN -> bound
        s = np.fmin(np.fmax(sm * s + so, F(0)), F(1)).astype(F)
        v = np.fmin(np.fmax(vm * v + vo, F(0)), F(1)).astype(F)
        return to_rgb(h, s, v)


def _float_to_u8(v):
    t = _clamp_inherent(v, F(0), F(1)) * F(255.0)
    # f32::round = half away from zero; t >= 0 here (or NaN)
    fl = np.floor(t)
    r = np.where(t - fl >= F(0.5), fl + F(1), fl).astype(F)
    return _as_u8(r)


def colorlut3d_rgb(r8, g8, b8, size, table4, scale, offset):
    """table4: (size^3, 4) float32, index x + y*size + z*size^2."""
    sm1 = F(size) - F(1.0)
    coords = []
    for comp, c8 in enumerate((r8, g8, b8)):
        v = c8.astype(F) / F(255.0)
        n = _clamp_inherent((v * F(scale[comp]) + F(offset[comp])).astype(F), F(0), F(1))
        coords.append((n * sm1).astype(F))
    idx0, idx1, ts = [], [], []
    for c in coords:
        fl = np.floor(c)
        i0 = np.minimum(np.where(np.isnan(fl), 0, fl).astype(np.int64), size - 1)
        i1 = np.minimum(i0 + 1, size - 1)
        idx0.append(i0)
        idx1.append(i1)
        ts.append((c - i0.astype(F)).astype(F))
    (x0, y0, z0), (x1, y1, z1), (tx, ty, tz) = idx0, idx1, ts

    def at(x, y, z):
        return table4[x + y * size + z * size * size]

    def lerp(a, b, t):
        return (a + ((b - a) * t[:, None]).astype(F)).astype(F)

    c00 = lerp(at(x0, y0, z0), at(x1, y0, z0), tx)
    c10 = lerp(at(x0, y1, z0), at(x1, y1, z0), tx)
    c01 = lerp(at(x0, y0, z1), at(x1, y0, z1), tx)
    c11 = lerp(at(x0, y1, z1), at(x1, y1, z1), tx)
    c0 = lerp(c00, c10, ty)
    c1 = lerp(c01, c11, ty)
    o = lerp(c0, c1, tz)
    return [_float_to_u8(o[:, k]) for k in range(3)]


def colorlut1d(c8, comp, size, planes, scale, offset):
    """planes: (3, size) float32."""
    v = c8.astype(F) / F(255.0)
    n = _clamp_inherent((v * F(scale[comp]) + F(offset[comp])).astype(F), F(0), F(1))
    x = (n * (F(size) - F(1.0))).astype(F)
    fl = np.floor(x)
    i0 = np.minimum(np.where(np.isnan(fl), 0, fl).astype(np.int64), size - 1)
    i1 = np.minimum(i0 + 1, size - 1)
    t = (x - i0.astype(F)).astype(F)
    lut = planes[comp]
    return _float_to_u8((lut[i0] + ((lut[i1] - lut[i0]) * t).astype(F)).astype(F))


# ---- rsaudioecho (audio/audiofx/src/audioecho/imp.rs:69-85, ring_buffer.rs:37-82), written from the reference source
# independently of oracle/echo_oracle.c: the sample loop there is serial; here the recurrence is evaluated a whole delay
# period at a time (sample i only depends on ring entries written >= D samples earlier), in float64 like the reference.
class Echo:
    def __init__(self, max_delay_ns, rate, channels):
        # AudioFilterImpl::setup (imp.rs:246-259): size = (max_delay * rate).seconds(), times channels
        self.rate, self.channels = rate, channels
        self.ring = np.zeros((max_delay_ns * rate // 10 ** 9) * channels, np.float64)
        self.pos = 0

    def process(self, data, delay_ns, intensity, feedback):
        """In place on an interleaved float32 / float64 array; delay already clamped to max-delay by the caller (imp.rs:207)."""
        size = self.ring.size
        delay = delay_ns * self.channels * self.rate // 10 ** 9          # imp.rs:74-77 (`.seconds()` truncates)
        assert size >= delay and size != 0                               # ring_buffer.rs:41-42
        # read == write index when delay is 0 or size: the sample read is the one written `size` samples ago (ring_buffer.rs:44-45)
        period = delay if 0 < delay < size else size
        n, done = data.size, 0
        while done < n:
            m = min(period, n - done)
            w = (self.pos + np.arange(m)) % size
            r = (w + size - delay) % size
            e = self.ring[r]                                             # all written at least one period ago
            inp = data[done:done + m].astype(np.float64)
            data[done:done + m] = (inp + intensity * e).astype(data.dtype)   # imp.rs:81,83
            self.ring[w] = inp + feedback * e                            # imp.rs:82
            self.pos = (self.pos + m) % size
            done += m
        return data
