/* gst/plugin_rsaudiofx.c — plugin "rsaudiofx" = libgstrsaudiofx.so (audio/audiofx/src/lib.rs:23-46, Cargo.toml lib name
 * gstrsaudiofx). plugin_init registers the elements of the hot path in the reference's order; agingradio and audiornnoise
 * (lib.rs:27, :30) are not part of it and stay with the reference's plugin. */
#include <gst/gst.h>
#ifndef PACKAGE
#define PACKAGE "gst-plugin-audiofx"
#endif
gboolean gst_rs_audio_echo_register(GstPlugin *plugin);
gboolean gst_audio_loud_norm_register(GstPlugin *plugin);
gboolean gst_ebur128_level_register(GstPlugin *plugin);

static gboolean plugin_init(GstPlugin *plugin) {
  return gst_rs_audio_echo_register(plugin) && gst_audio_loud_norm_register(plugin) && gst_ebur128_level_register(plugin);
}

GST_PLUGIN_DEFINE(GST_VERSION_MAJOR, GST_VERSION_MINOR, rsaudiofx, "GStreamer Rust Audio Effects Plugin (MI355X kernels)", plugin_init,
                  "0.16.0-alpha.1-mi355fx", "MPL", "gst-plugin-audiofx", "https://gitlab.freedesktop.org/gstreamer/gst-plugins-rs")
