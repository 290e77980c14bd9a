/* gst/plugin_hsv.c — plugin "hsv" = libgsthsv.so (video/hsv/src/lib.rs:23-42, Cargo.toml lib name gsthsv):
 * GST_PLUGIN_DEFINE exports gst_plugin_hsv_get_desc() / gst_plugin_hsv_register(), which is what plugin_define! does. */
#include <gst/gst.h>
#ifndef PACKAGE
#define PACKAGE "gst-plugin-hsv"
#endif
gboolean gst_hsv_filter_register(GstPlugin *plugin);
gboolean gst_hsv_detector_register(GstPlugin *plugin);

/* plugin_init (video/hsv/src/lib.rs:23-30): hsvfilter::register, then hsvdetector::register */
static gboolean plugin_init(GstPlugin *plugin) { return gst_hsv_filter_register(plugin) && gst_hsv_detector_register(plugin); }

GST_PLUGIN_DEFINE(GST_VERSION_MAJOR, GST_VERSION_MINOR, hsv, "GStreamer plugin with HSV manipulation elements (MI355X kernels)", plugin_init,
                  "0.16.0-alpha.1-mi355fx", "MIT/X11", "gst-plugin-hsv", "https://gitlab.freedesktop.org/gstreamer/gst-plugins-rs")
