/* gst/gstmi355common.h — shared by the shim elements: the mapping from GstVideoFormat to mi355_video_format, a GstAllocator
 * whose memory is page-locked (mi355_host_alloc) so that mapped GstBuffer payloads can be DMA'd without a staging copy, and
 * the buffer meta + peer query with which `hsvfilter ! colorlut` collapse into ONE upload, ONE fused launch and ONE
 * download (mi355_pipe_submit_hsv_colorlut) instead of two PCIe round trips.
 *
 * This directory is the thin C shim of SURVEY.md §7 step 6 / §8(b): GObject subclasses that register under the
 * reference's factory and GType names and call the C ABI of include/mi355fx.h in their transform vfuncs. It is built only
 * where `pkg-config gstreamer-video-1.0` exists (gst/Makefile); this image has neither GStreamer headers nor pkg-config,
 * so here it is source only. Two shared objects, as in the reference: libgsthsv.so (plugin "hsv": hsvfilter, hsvdetector)
 * and libgstcolorlut.so (plugin "colorlut": colorlut) — video/hsv/src/lib.rs:23-42, video/colorlut/src/lib.rs:22-43.
 * Same GType names as the reference plugins: one process can load this shim OR the Rust plugins, not both. */
#ifndef GST_MI355_COMMON_H
#define GST_MI355_COMMON_H

#include <gst/gst.h>
#include <gst/video/video.h>
#include <gst/video/gstvideofilter.h>

#include "../include/mi355fx.h"

G_BEGIN_DECLS

/* ---- pinned allocator (precedent for an element offering its own memory upstream:
 * video/colorlut/src/d3d12colorlut/imp.rs:385-424 propose_allocation). The allocator owns its mi355 context: pools handed
 * upstream in propose_allocation belong to the upstream element and their buffers can outlive this element's stop(). */
#define GST_TYPE_MI355_ALLOCATOR (gst_mi355_allocator_get_type())
G_DECLARE_FINAL_TYPE(GstMi355Allocator, gst_mi355_allocator, GST, MI355_ALLOCATOR, GstAllocator)
GstAllocator *gst_mi355_allocator_new(void);

/* adds a video buffer pool over the pinned allocator (and GstVideoMeta support) to an allocation query */
gboolean gst_mi355_propose_pinned_pool(GstBaseTransform *trans, GstQuery *query);
/* a pinned video pool for `caps`, configured (decide_allocation of an element that allocates its own output) */
GstBufferPool *gst_mi355_pinned_pool_new(GstCaps *caps, const GstVideoInfo *info);

/* ---- hsvfilter -> colorlut fusion.
 * hsvfilter asks its downstream peer `mi355-fuse-hsv` (a custom query); this shim's colorlut answers it when it runs on
 * RGBA frames. From then on hsvfilter does NOT touch the pixels: it attaches a GstMi355HsvMeta with the settings snapshot
 * of that frame to the buffer it passes through, and colorlut runs mi355_pipe_submit_hsv_colorlut (the fused kernel, bit-
 * identical to the two launches: tests/test_gpu_parity.py) when it finds the meta. Queries travel (queue, tee, capsfilter,
 * every GstBaseTransform forward what they do not know), so an answer alone says nothing about the neighbour: the answering
 * colorlut writes its own address into the query's "who" field and hsvfilter defers only if that is the element owning the
 * pad linked to its source pad; it asks again after every relink, RECONFIGURE and renegotiation. Nothing but that colorlut
 * ever sees an unfiltered frame carrying the meta. */
#define GST_MI355_FUSE_QUERY_NAME "mi355-fuse-hsv"
#define GST_MI355_FUSE_QUERY_WHO "who"
typedef struct {
  GstMeta meta;
  mi355_hsv_settings settings;
} GstMi355HsvMeta;
GType gst_mi355_hsv_meta_api_get_type(void);
const GstMetaInfo *gst_mi355_hsv_meta_get_info(void);
#define GST_MI355_HSV_META_API_TYPE (gst_mi355_hsv_meta_api_get_type())
static inline GstMi355HsvMeta *gst_buffer_get_mi355_hsv_meta(GstBuffer *b) {
  return (GstMi355HsvMeta *)gst_buffer_get_meta(b, GST_MI355_HSV_META_API_TYPE);
}
GstMi355HsvMeta *gst_buffer_add_mi355_hsv_meta(GstBuffer *b, const mi355_hsv_settings *s);

/* GstVideoFormat -> mi355_video_format, -1 when the format is not one of the reference's */
static inline int gst_mi355_format(GstVideoFormat f) {
  switch (f) {
    case GST_VIDEO_FORMAT_RGBx: return MI355_FMT_RGBX;
    case GST_VIDEO_FORMAT_xRGB: return MI355_FMT_XRGB;
    case GST_VIDEO_FORMAT_BGRx: return MI355_FMT_BGRX;
    case GST_VIDEO_FORMAT_xBGR: return MI355_FMT_XBGR;
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
