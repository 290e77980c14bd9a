"""The GStreamer C shim (gst/) is compile-gated on `pkg-config gstreamer-video-1.0`: where the development files exist it must
build; where they do not (this image) `make -C gst` must say so and succeed. Either way the sources must carry the
reference's registration surface (factory / GType names, plugin names, the vfunc slots that define the transform mode)."""
import os
import re
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GST = os.path.join(ROOT, "gst")


def _have_gst_dev():
    return shutil.which("pkg-config") is not None and subprocess.call(
        ["pkg-config", "--exists", "gstreamer-1.0", "gstreamer-base-1.0", "gstreamer-video-1.0"]) == 0


def test_make_builds_or_explains():
    r = subprocess.run(["make", "-C", GST, "all"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    if _have_gst_dev():
        assert os.path.exists(os.path.join(GST, "libgsthsv.so")) and os.path.exists(os.path.join(GST, "libgstcolorlut.so"))
    else:
        assert "gst shim not built" in r.stdout


def test_sources_carry_the_reference_registration_surface():
    src = {f: open(os.path.join(GST, f)).read() for f in os.listdir(GST) if f.endswith((".c", ".h"))}
    hsv, lut = src["gsthsvfilter.c"], src["gstcolorlut.c"]
    # factory names, rank NONE, GType names (same as the Rust plugins: video/hsv/src/hsvfilter/imp.rs:67-72, colorlut/imp.rs:61-66)
    assert re.search(r'gst_element_register\(plugin, "hsvfilter", GST_RANK_NONE', hsv)
    assert re.search(r'gst_element_register\(plugin, "colorlut", GST_RANK_NONE', lut)
    assert "G_DEFINE_TYPE(GstHsvFilter," in hsv and "G_DEFINE_TYPE(GstColorLut," in lut
    # plugin names (video/hsv/src/lib.rs:33, video/colorlut/src/lib.rs:33)
    assert re.search(r"GST_PLUGIN_DEFINE\(GST_VERSION_MAJOR, GST_VERSION_MINOR, hsv,", src["plugin_hsv.c"])
    assert re.search(r"GST_PLUGIN_DEFINE\(GST_VERSION_MAJOR, GST_VERSION_MINOR, colorlut,", src["plugin_colorlut.c"])
    # transform modes: hsvfilter installs only transform_frame_ip (AlwaysInPlace), colorlut only transform_frame (NeverInPlace)
    assert "vfilter->transform_frame_ip =" in hsv and "vfilter->transform_frame =" not in hsv
    assert "vfilter->transform_frame =" in lut and "vfilter->transform_frame_ip =" not in lut
    # properties of the reference (docs/plugins/gst_plugins_cache.json via tests/golden/element_surface.json)
    import json
    surface = json.load(open(os.path.join(ROOT, "tests", "golden", "element_surface.json")))
    for name in surface["hsvfilter"]["properties"]:
        assert '"%s"' % name in hsv, name
    for name in surface["colorlut"]["properties"]:
        assert '"%s"' % name in lut, name
    for fmt in surface["hsvfilter"]["sink_formats"]:
        assert fmt in hsv
    for fmt in surface["colorlut"]["sink_formats"]:
        assert fmt in lut
    # the ABI entry points the vfuncs call, and the pinned pool offered upstream
    assert "mi355_hsvfilter_frame_ip(" in hsv and "mi355_colorlut_frame(" in lut and "mi355_colorlut_load(" in lut
    assert "mi355_host_alloc(" in src["gstmi355allocator.c"] and "propose_allocation" in hsv and "propose_allocation" in lut
