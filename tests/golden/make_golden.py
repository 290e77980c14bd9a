#!/usr/bin/env python3
"""Regenerates the fixtures in tests/golden/.
  surface : element_surface.json from /root/reference/docs/plugins/gst_plugins_cache.json (introspection DATA)
  crc     : pixel_crc.json — CRC-32 of the oracle's outputs on the seeded synthetic inputs
  masks   : cairo_masks.json — roundedcorners alpha masks rendered with the system libcairo via ctypes
"""
import json
import os
import re
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))


def surface():
    d = json.load(open("/root/reference/docs/plugins/gst_plugins_cache.json"))
    out = {}
    for plug, els in (("hsv", ["hsvfilter", "hsvdetector"]), ("colorlut", ["colorlut"]), ("rsaudiofx", ["rsaudioecho", "ebur128level", "audioloudnorm"]),
                      ("rsvideofx", ["videocompare", "roundedcorners"]), ("hrtf", ["hrtfrender"])):
        for e in els:
            el = d[plug]["elements"][e]
            props = {n: {k: p[k] for k in ("type", "default", "min", "max", "mutable") if k in p}
                     for n, p in el["properties"].items() if n not in ("name", "parent", "qos")}
            spec = {"type_name": el["hierarchy"][0], "klass": el["klass"], "properties": props}
            if plug in ("hsv", "colorlut"):
                def fmts(caps):
                    return [t.strip() for t in re.search(r"format:\s*\{([^}]*)\}", caps).group(1).split(",")]
                spec["sink_formats"] = fmts(el["pad-templates"]["sink"]["caps"])
                spec["src_formats"] = fmts(el["pad-templates"]["src"]["caps"])
            out[e] = spec
    json.dump(out, open(os.path.join(HERE, "element_surface.json"), "w"), indent=1, sort_keys=True)


def crc_cases():
    """name -> bytes of the oracle output. Shared with tests/test_golden.py."""
    import numpy as np
    from oracle import oracle as O
    from mi355fx import synth
    ac = synth.allcolors()
    out = {}
    for name, st in synth.HSV_SETTINGS.items():
        buf = ac.copy().reshape(-1)
        O.hsvfilter(buf, 4096, 4096 * 4, 4, 0, False, st, nthreads=8)
        out["hsvfilter_allcolors_rgba_" + name] = buf.tobytes()
    buf = ac.copy().reshape(-1)
    O.hsvfilter(buf, 4096, 4096 * 4, 4, 1, True, synth.HSV_SETTINGS["mixed"], nthreads=8)
    out["hsvfilter_allcolors_xbgr_mixed"] = buf.tobytes()
    for tag, text in (("lut33", synth.cube_text_3d(33)), ("lut17_domain", synth.cube_text_3d(17, amp=0.07, domain=((-0.25, 0.0, 0.1), (1.5, 1.0, 0.9)))),
                      ("lut1d64", synth.cube_text_1d(64))):
        dst = np.zeros_like(ac)
        O.colorlut_rgba8(O.Cube.parse(text), ac, 4096 * 4, dst, 4096 * 4, 4096, 4096, nthreads=8)
        out["colorlut_allcolors_" + tag] = dst.tobytes()
    e = O.Echo(10 ** 9, 48000, 2)
    x = synth.sine_stereo_f32()
    e.process(x, 250 * 10 ** 6, 0.6, 0.4)
    out["echo_config1_f32"] = x.tobytes()
    return out


def crc():
    vals = {k: zlib.crc32(v) for k, v in crc_cases().items()}
    json.dump(vals, open(os.path.join(HERE, "pixel_crc.json"), "w"), indent=1, sort_keys=True)


def cairo_mask(w, h, r):
    """The alpha mask roundedcorners renders (video/videofx/src/border/imp.rs:57-180), through the system libcairo with ctypes:
    A8 surface of stride round_up_4(w) and round_up_2(h) rows, four arcs, antialiased fill_preserve + 1 px stroke."""
    import ctypes as C
    import math
    import numpy as np
    cairo = C.CDLL("libcairo.so.2")
    cairo.cairo_image_surface_create_for_data.restype = C.c_void_p
    cairo.cairo_image_surface_create_for_data.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    cairo.cairo_create.restype = C.c_void_p
    cairo.cairo_create.argtypes = [C.c_void_p]
    for f in ("cairo_new_sub_path", "cairo_close_path", "cairo_fill_preserve", "cairo_stroke", "cairo_destroy", "cairo_surface_flush", "cairo_surface_destroy"):
        getattr(cairo, f).argtypes = [C.c_void_p]
    cairo.cairo_arc.argtypes = [C.c_void_p] + [C.c_double] * 5
    cairo.cairo_set_source_rgb.argtypes = [C.c_void_p] + [C.c_double] * 3
    cairo.cairo_set_source_rgba.argtypes = [C.c_void_p] + [C.c_double] * 4
    cairo.cairo_set_line_width.argtypes = [C.c_void_p, C.c_double]
    stride, rows = (w + 3) & ~3, (h + 1) & ~1
    buf = np.zeros((rows, stride), np.uint8)
    if r == 0:
        buf[:] = 255
        return buf
    s = cairo.cairo_image_surface_create_for_data(buf.ctypes.data, 2, w, h, stride)
    cr = cairo.cairo_create(s)
    d, R, W, H = math.pi / 180.0, float(r), float(w), float(h)
    cairo.cairo_new_sub_path(cr)
    cairo.cairo_arc(cr, W - R, R, R, -90 * d, 0 * d)
    cairo.cairo_arc(cr, W - R, H - R, R, 0 * d, 90 * d)
    cairo.cairo_arc(cr, R, H - R, R, 90 * d, 180 * d)
    cairo.cairo_arc(cr, R, R, R, 180 * d, 270 * d)
    cairo.cairo_close_path(cr)
    cairo.cairo_set_source_rgb(cr, 0, 0, 0)
    cairo.cairo_fill_preserve(cr)
    cairo.cairo_set_source_rgba(cr, 0, 0, 0, 1)
    cairo.cairo_set_line_width(cr, 1.0)
    cairo.cairo_stroke(cr)
    cairo.cairo_destroy(cr)
    cairo.cairo_surface_flush(s)
    cairo.cairo_surface_destroy(s)
    return buf


MASK_CASES = [(40, 24, 8), (64, 48, 10), (321, 241, 30), (1920, 1080, 100), (3840, 2160, 250), (100, 60, 0), (50, 50, 25)]


def masks():
    """cairo_masks.json: CRC-32 of the full A8 plane per (width, height, radius) + the first 12x12 corner bytes of each,
    rendered with the libcairo of this image (version recorded)."""
    import ctypes as C
    cairo = C.CDLL("libcairo.so.2")
    cairo.cairo_version_string.restype = C.c_char_p
    out = {"cairo_version": cairo.cairo_version_string().decode(), "cases": {}}
    for w, h, r in MASK_CASES:
        m = cairo_mask(w, h, r)
        out["cases"]["%dx%d_r%d" % (w, h, r)] = {"crc32": zlib.crc32(m.tobytes()), "shape": list(m.shape), "corner": m[:12, :12].tolist()}
    json.dump(out, open(os.path.join(HERE, "cairo_masks.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("surface", "all") and os.path.exists("/root/reference"):
        surface()
    if what in ("crc", "all"):
        crc()
    if what in ("masks", "all"):
        masks()
