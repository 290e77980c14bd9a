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

/* ---- device memory: GstMemory over mi355_buf (include/mi355fx.h "device buffers"; csrc/buf.hip). Precedent:
 * video/colorlut/src/d3d12colorlut/imp.rs:385-492 - propose_allocation offers upstream a pool of GPU memory, decide_allocation
 * picks one for the output, and an element whose input memory is "ours" works on the GPU resource directly while anything else
 * maps the memory and gets bytes. Here: mem_map = mi355_buf_map_host (the pinned shadow, downloaded only if the device side is
 * newer), mem_unmap = mi355_buf_unmap_host (a WRITE map makes the host side newer: the next device use uploads it). The video
 * shims look at their input / output buffers BEFORE GstVideoFilter maps them (they override GstBaseTransformClass::transform):
 * two buffers of ours run the `_device` entry point on mi355_buf_device_ptr and are committed; anything else chains up and is
 * mapped as before. `hsvdetector ! colorlut ! videocompare` then costs one upload and one download in total
 * (tests/test_gpu_buf.py runs exactly that through the C ABI). NOTHING of this file has ever run inside GStreamer: this image
 * has no GStreamer; `make -C gst syntax` compiles it against tests/gst_stub. */
#define GST_MI355_DEVICE_MEMORY_TYPE "Mi355DeviceMemory"
#define GST_CAPS_FEATURE_MEMORY_MI355 "memory:Mi355DeviceMemory"
#define GST_TYPE_MI355_DEVICE_ALLOCATOR (gst_mi355_device_allocator_get_type())
G_DECLARE_FINAL_TYPE(GstMi355DeviceAllocator, gst_mi355_device_allocator, GST, MI355_DEVICE_ALLOCATOR, GstAllocator)
GstAllocator *gst_mi355_device_allocator_new(void);
/* the mi355_buf behind `mem` (borrowed) if the memory is ours and covers its whole buffer object, else NULL */
mi355_buf *gst_mi355_device_memory_get_buf(GstMemory *mem);
/* the mi355_buf of a GstBuffer that consists of exactly one whole memory of ours, else NULL (the caller maps the buffer) */
mi355_buf *gst_mi355_buffer_peek_device(GstBuffer *buffer);
/* propose_allocation: the device allocator and (if asked for) a video pool over it are added IN FRONT of the pinned ones */
gboolean gst_mi355_propose_device_pool(GstBaseTransform *trans, GstQuery *query);
/* decide_allocation, BEFORE chaining up to the parent class (which adds a system-memory pool of its own to a query without one): if
 * downstream offered no pool, the output comes from a device pool of ours (a downstream element that is not ours maps it: one lazy
 * download) */
gboolean gst_mi355_decide_device_pool(GstBaseTransform *trans, GstQuery *query);

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
