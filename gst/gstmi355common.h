/* gst/gstmi355common.h — shared by the two shim elements: a process-wide-by-element mi355 context, the mapping from
 * GstVideoFormat to mi355_video_format, and a GstAllocator whose memory is page-locked (mi355_host_alloc) so that mapped
 * GstBuffer payloads can be DMA'd without a staging copy.
 *
 * This directory is the thin C shim of SURVEY.md §7 step 6 / §8(b): GObject subclasses that register under the
 * reference's factory and GType names and call the C ABI of include/mi355fx.h in their transform vfuncs. It is built only
 * where `pkg-config gstreamer-video-1.0` exists (gst/Makefile); this image has neither GStreamer headers nor pkg-config,
 * so here it is source only. Two shared objects, as in the reference: libgsthsv.so (plugin "hsv": hsvfilter) and
 * libgstcolorlut.so (plugin "colorlut": colorlut) — video/hsv/src/lib.rs:23-42, video/colorlut/src/lib.rs:22-43.
 * Same GType names as the reference plugins: one process can load this shim OR the Rust plugins, not both. */
#ifndef GST_MI355_COMMON_H
#define GST_MI355_COMMON_H

#include <gst/gst.h>
#include <gst/video/video.h>
#include <gst/video/gstvideofilter.h>

#include "../include/mi355fx.h"

G_BEGIN_DECLS

/* ---- pinned allocator (precedent for an element offering its own memory upstream:
 * video/colorlut/src/d3d12colorlut/imp.rs:385-424 propose_allocation) */
#define GST_TYPE_MI355_ALLOCATOR (gst_mi355_allocator_get_type())
G_DECLARE_FINAL_TYPE(GstMi355Allocator, gst_mi355_allocator, GST, MI355_ALLOCATOR, GstAllocator)
GstAllocator *gst_mi355_allocator_new(mi355_ctx *ctx);

/* adds a video buffer pool over the pinned allocator (and GstVideoMeta support) to an allocation query */
gboolean gst_mi355_propose_pinned_pool(GstBaseTransform *trans, mi355_ctx *ctx, GstQuery *query);

/* GstVideoFormat -> mi355_video_format, -1 when the format is not one of the reference's */
static inline int gst_mi355_format(GstVideoFormat f) {
  switch (f) {
    case GST_VIDEO_FORMAT_RGBx: return MI355_FMT_RGBx;
    case GST_VIDEO_FORMAT_xRGB: return MI355_FMT_xRGB;
    case GST_VIDEO_FORMAT_BGRx: return MI355_FMT_BGRx;
    case GST_VIDEO_FORMAT_xBGR: return MI355_FMT_xBGR;
    case GST_VIDEO_FORMAT_RGBA: return MI355_FMT_RGBA;
    case GST_VIDEO_FORMAT_ARGB: return MI355_FMT_ARGB;
    case GST_VIDEO_FORMAT_BGRA: return MI355_FMT_BGRA;
    case GST_VIDEO_FORMAT_ABGR: return MI355_FMT_ABGR;
    case GST_VIDEO_FORMAT_RGB: return MI355_FMT_RGB;
    case GST_VIDEO_FORMAT_BGR: return MI355_FMT_BGR;
    case GST_VIDEO_FORMAT_RGBA64_LE: return MI355_FMT_RGBA64_LE;
    case GST_VIDEO_FORMAT_RGBA64_BE: return MI355_FMT_RGBA64_BE;
    default: return -1;
  }
}

G_END_DECLS
#endif
