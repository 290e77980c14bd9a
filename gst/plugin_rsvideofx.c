/* gst/plugin_rsvideofx.c — plugin "rsvideofx" = libgstrsvideofx.so (video/videofx/src/lib.rs:25-48, Cargo.toml lib name
 * gstrsvideofx). plugin_init registers roundedcorners (border::register) and videocompare; colordetect (lib.rs:33) is not part
 * of the hot path and stays with the reference's plugin. */
#include <gst/gst.h>
#ifndef PACKAGE
#define PACKAGE "gst-plugin-videofx"
#endif
gboolean gst_rounded_corners_register(GstPlugin *plugin);
gboolean gst_video_compare_register(GstPlugin *plugin);

static gboolean plugin_init(GstPlugin *plugin) { return gst_rounded_corners_register(plugin) && gst_video_compare_register(plugin); }

GST_PLUGIN_DEFINE(GST_VERSION_MAJOR, GST_VERSION_MINOR, rsvideofx, "GStreamer Rust Video Effects Plugin (MI355X kernels)", plugin_init,
                  "0.16.0-alpha.1-mi355fx", "MPL", "gst-plugin-videofx", "https://gitlab.freedesktop.org/gstreamer/gst-plugins-rs")
