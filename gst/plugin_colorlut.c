/* gst/plugin_colorlut.c — plugin "colorlut" = libgstcolorlut.so (video/colorlut/src/lib.rs:22-43). */
#include <gst/gst.h>
#ifndef PACKAGE
#define PACKAGE "gst-plugin-colorlut"
#endif
gboolean gst_color_lut_register(GstPlugin *plugin);

static gboolean plugin_init(GstPlugin *plugin) { return gst_color_lut_register(plugin); }

GST_PLUGIN_DEFINE(GST_VERSION_MAJOR, GST_VERSION_MINOR, colorlut, "GStreamer colorlut plugin (MI355X kernels)", plugin_init, "0.16.0-alpha.1-mi355fx",
                  "MPL-2.0", "gst-plugin-colorlut", "https://gitlab.freedesktop.org/gstreamer/gst-plugins-rs")
