/* gst/plugin_hrtf.c — plugin "hrtf" = libgsthrtf.so (audio/hrtf/src/lib.rs:49-70, Cargo.toml lib name gsthrtf). plugin_init
 * registers hrtfrender; the `sofalizer` element (lib.rs:58) has its kernels behind the C ABI (mi355_sofa_*) but no shim
 * element yet. */
#include <gst/gst.h>
#ifndef PACKAGE
#define PACKAGE "gst-plugin-hrtf"
#endif
gboolean gst_hrtf_render_register(GstPlugin *plugin);

static gboolean plugin_init(GstPlugin *plugin) { return gst_hrtf_render_register(plugin); }

GST_PLUGIN_DEFINE(GST_VERSION_MAJOR, GST_VERSION_MINOR, hrtf, "GStreamer Rust Head Related Transform Function (HRTF) Plugin (MI355X kernels)", plugin_init,
                  "0.16.0-alpha.1-mi355fx", "MPL", "gst-plugin-hrtf", "https://gitlab.freedesktop.org/gstreamer/gst-plugins-rs")
