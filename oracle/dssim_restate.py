"""oracle/dssim_restate.py — numpy-f32 restatement of videocompare's optional Dssim engine. TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED. HashAlgorithm::Dssim (cargo feature `dssim`, off by default: video/videofx/Cargo.toml:18,39) hands the
packed frame to the third-party crate dssim-core 3.4.0 (`Dssim::new()`, `create_image_rgb/rgba`, `compare`;
call sites video/videofx/src/videocompare/hashed_image.rs:41-53,66-70,92-94), whose sources are not under
/root/reference. The only reference test (video/videofx/tests/videocompare.rs:140-182) pins: identical frames ->
distance <= 0.0. This file restates the crate's published multi-scale SSIM-in-LAB algorithm:

  * 8-bit sRGB -> linear light through a 256-entry table (s <= 0.04045 ? s/12.92 : ((s+0.055)/1.055)^2.4);
    RGBA is premultiplied by a/255; translucent pixels are composed over dssim's coloured background pattern: with
    n = (x + 11) ^ (y + 11) in the coordinates of the scale being converted, r += 1 - a where n & 16, g += 1 - a where
    n & 8, b += 1 - a where n & 32 (from memory of dssim-core's to_lab for RGBA, unverified; `pattern=False` composes
    over black). Opaque frames - what videocompare's tests feed - do not depend on it.
  * 5 scales (weights 0.028, 0.197, 0.322, 0.298, 0.155), each the 2x2 box average ((a+b+c+d)*0.25) of the previous
    one in linear RGB, floor(w/2) x floor(h/2), stopping early below 8 pixels.
  * per scale: linear RGB -> the crate's LAB variant (D65-normalised XYZ, polynomial+2xHalley cube root,
    L = 1.05*Y', a = 500/220*(X'-Y') + 86.2/220, b = 200/220*(Y'-Z') + 107.9/220), chroma planes blurred once;
    mu = blur(plane), sq = blur(plane^2), blur = TWO passes of the 3x3 kernel
    [0.095332 0.118095 0.095332; 0.118095 0.146293 0.118095; ...] with replicated edges.
  * compare: i12 = blur(img1*img2) per channel; with X = mean over the 3 LAB channels of each quantity,
    ssim = (2*mu1mu2 + c1)(2*sigma12 + c2) / ((mu1^2+mu2^2 + c1)(sigma1^2+sigma2^2 + c2)), c1 = 0.01^2, c2 = 0.03^2;
    per scale n: avg = max(mean(ssim), 0)^(0.5^n); score = 1 - mean(|avg - ssim|) (f64);
    ssim_total = sum(score*w)/sum(w); dssim = 1/max(ssim_total, eps) - 1.
Tap order inside the blur (row-major, running f32 sum) and every f32 operation order are the ones the device
kernels use, so the per-pixel maps are comparable bit for bit; the f64 reductions are order-dependent (1e-12)."""
import numpy as np

F = np.float32
WEIGHTS = [0.028, 0.197, 0.322, 0.298, 0.155]
KERNEL = np.array([0.095332, 0.118095, 0.095332, 0.118095, 0.146293, 0.118095, 0.095332, 0.118095, 0.095332], F)
D65 = (F(0.9505), F(1.0), F(1.089))
EPSILON = F(216.0) / F(24389.0)
K = F(24389.0) / (F(27.0) * F(116.0))


def gamma_lut():
    s = np.arange(256, dtype=np.float64) / 255.0
    lin = np.where(s <= 0.04045, s / 12.92, ((s + 0.055) / 1.055) ** 2.4)
    return lin.astype(F)


def to_linear(frame, width, height, stride, channels):
    """-> (h, w, 4) f32 premultiplied linear RGBA (alpha 1 for RGB)."""
    lut = gamma_lut()
    px = np.ascontiguousarray(frame, np.uint8).reshape(-1)[: height * stride].reshape(height, stride)[:, : width * channels].reshape(height, width, channels)
    out = np.ones((height, width, 4), F)
    if channels == 4:
        a = px[..., 3].astype(F) / F(255.0)
        for c in range(3):
            out[..., c] = lut[px[..., c]] * a
        out[..., 3] = a
    else:
        for c in range(3):
            out[..., c] = lut[px[..., c]]
    return out


def downsample(img):
    h, w = img.shape[:2]
    if w < 8 or h < 8:
        return None
    h2, w2 = h // 2, w // 2
    a = img[0:2 * h2:2, 0:2 * w2:2]; b = img[0:2 * h2:2, 1:2 * w2:2]; c = img[1:2 * h2:2, 0:2 * w2:2]; d = img[1:2 * h2:2, 1:2 * w2:2]
    return (((a + b) + c) + d) * F(0.25)


def cbrt_poly(x):
    y = (F(-0.5) * x + F(1.51)) * x + F(0.2)
    for _ in range(2):
        y3 = y * y * y
        y = y * (y3 + F(2.0) * x) / (F(2.0) * y3 + x)
    return y


def to_lab(img, pattern=True):
    r, g, b = img[..., 0], img[..., 1], img[..., 2]
    if pattern:
        h, w = img.shape[:2]
        n = (np.arange(w)[None, :] + 11) ^ (np.arange(h)[:, None] + 11)
        t = F(1.0) - img[..., 3]
        r = np.where(n & 16, r + t, r).astype(F)
        g = np.where(n & 8, g + t, g).astype(F)
        b = np.where(n & 32, b + t, b).astype(F)
    def mat(rx, gx, bx, d):
        return (r * (F(rx) / d) + g * (F(gx) / d)) + b * (F(bx) / d)
    fx = mat(0.4124, 0.3576, 0.1805, D65[0]); fy = mat(0.2126, 0.7152, 0.0722, D65[1]); fz = mat(0.0193, 0.1192, 0.9505, D65[2])
    def f(t):
        with np.errstate(all="ignore"):
            return np.where(t > EPSILON, cbrt_poly(t) - F(16.0) / F(116.0), K * t).astype(F)
    X, Y, Z = f(fx), f(fy), f(fz)
    L = Y * F(1.05)
    A = (F(500.0) / F(220.0)) * (X - Y) + F(86.2) / F(220.0)
    B = (F(200.0) / F(220.0)) * (Y - Z) + F(107.9) / F(220.0)
    return [L.astype(F), A.astype(F), B.astype(F)]


def blur_pass(p):
    q = np.pad(p, 1, mode="edge")
    h, w = p.shape
    acc = np.zeros((h, w), F)
    k = 0
    for dy in range(3):
        for dx in range(3):
            acc = acc + q[dy:dy + h, dx:dx + w] * KERNEL[k]
            k += 1
    return acc


def blur(p):
    return blur_pass(blur_pass(p))


class DssimImage:
    def __init__(self, frame, width, height, stride, channels, pattern=True):
        lin = to_linear(frame, width, height, stride, channels)
        self.scales = []
        cur = lin
        while cur is not None and len(self.scales) < len(WEIGHTS):
            planes = to_lab(cur, pattern and channels == 4)
            chans = []
            for n, p in enumerate(planes):
                if n > 0:
                    p = blur(p)
                chans.append({"img": p, "mu": blur(p), "sq": blur(p * p)})
            self.scales.append(chans)
            cur = downsample(cur)


def compare(img1, img2, return_maps=False):
    c1, c2 = F(0.01 * 0.01), F(0.03 * 0.03)
    third = F(1.0 / 3.0)
    ssim_sum = weight_sum = 0.0
    maps = []
    for n, (w, s1, s2) in enumerate(zip(WEIGHTS, img1.scales, img2.scales)):
        def avg3(key, src):
            return ((src[0][key] + src[1][key]) + src[2][key]) * third
        i12 = [blur(s1[c]["img"] * s2[c]["img"]) for c in range(3)]
        mu1mu1 = ((s1[0]["mu"] * s1[0]["mu"] + s1[1]["mu"] * s1[1]["mu"]) + s1[2]["mu"] * s1[2]["mu"]) * third
        mu2mu2 = ((s2[0]["mu"] * s2[0]["mu"] + s2[1]["mu"] * s2[1]["mu"]) + s2[2]["mu"] * s2[2]["mu"]) * third
        mu1mu2 = ((s1[0]["mu"] * s2[0]["mu"] + s1[1]["mu"] * s2[1]["mu"]) + s1[2]["mu"] * s2[2]["mu"]) * third
        sig1 = (((s1[0]["sq"] - s1[0]["mu"] * s1[0]["mu"]) + (s1[1]["sq"] - s1[1]["mu"] * s1[1]["mu"])) + (s1[2]["sq"] - s1[2]["mu"] * s1[2]["mu"])) * third
        sig2 = (((s2[0]["sq"] - s2[0]["mu"] * s2[0]["mu"]) + (s2[1]["sq"] - s2[1]["mu"] * s2[1]["mu"])) + (s2[2]["sq"] - s2[2]["mu"] * s2[2]["mu"])) * third
        sig12 = (((i12[0] - s1[0]["mu"] * s2[0]["mu"]) + (i12[1] - s1[1]["mu"] * s2[1]["mu"])) + (i12[2] - s1[2]["mu"] * s2[2]["mu"])) * third
        ssim = ((F(2.0) * mu1mu2 + c1) * (F(2.0) * sig12 + c2)) / (((mu1mu1 + mu2mu2) + c1) * ((sig1 + sig2) + c2))
        ssim = ssim.astype(F)
        m = ssim.astype(np.float64)
        avg = max(float(m.sum()) / m.size, 0.0) ** (0.5 ** n)
        score = 1.0 - float(np.abs(avg - m).sum()) / m.size
        ssim_sum += score * w
        weight_sum += w
        maps.append(ssim)
    total = ssim_sum / weight_sum
    d = 1.0 / max(total, np.finfo(np.float64).eps) - 1.0
    return (d, maps) if return_maps else d
