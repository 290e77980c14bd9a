/* gst/gstmi355allocator.c — GstAllocator over mi355_host_alloc (page-locked host memory). Buffers of a pool built on it
 * map to pointers the library can hipMemcpyAsync from / to directly (47 GB/s each way), instead of going
 * through a pageable staging copy. */
#include "gstmi355common.h"

struct _GstMi355Allocator {
  GstAllocator parent;
  mi355_ctx *ctx; /* borrowed: the element outlives its pools (it drops them in stop) */
};
G_DEFINE_TYPE(GstMi355Allocator, gst_mi355_allocator, GST_TYPE_ALLOCATOR)

typedef struct {
  GstMemory mem;
  gpointer data;
} GstMi355Memory;

static GstMemory *gst_mi355_allocator_alloc(GstAllocator *allocator, gsize size, GstAllocationParams *params) {
  GstMi355Allocator *self = GST_MI355_ALLOCATOR(allocator);
  const gsize maxsize = size + params->prefix + params->padding;
  gpointer data = mi355_host_alloc(self->ctx, maxsize); /* page-aligned: satisfies any params->align */
  if (!data) return NULL;
  GstMi355Memory *m = g_new0(GstMi355Memory, 1);
  gst_memory_init(GST_MEMORY_CAST(m), params->flags, allocator, NULL, maxsize, params->align, params->prefix, size);
  m->data = data;
  if (params->prefix && (params->flags & GST_MEMORY_FLAG_ZERO_PREFIXED)) memset(data, 0, params->prefix);
  if (params->padding && (params->flags & GST_MEMORY_FLAG_ZERO_PADDED)) memset((guint8 *)data + params->prefix + size, 0, params->padding);
  return GST_MEMORY_CAST(m);
}

static void gst_mi355_allocator_free(GstAllocator *allocator, GstMemory *memory) {
  GstMi355Allocator *self = GST_MI355_ALLOCATOR(allocator);
  GstMi355Memory *m = (GstMi355Memory *)memory;
  (void)mi355_host_free(self->ctx, m->data);
  g_free(m);
}

static gpointer gst_mi355_mem_map(GstMemory *mem, gsize maxsize, GstMapFlags flags) { return ((GstMi355Memory *)mem)->data; }
static void gst_mi355_mem_unmap(GstMemory *mem) {}

static void gst_mi355_allocator_class_init(GstMi355AllocatorClass *klass) {
  GstAllocatorClass *a = GST_ALLOCATOR_CLASS(klass);
  a->alloc = gst_mi355_allocator_alloc;
  a->free = gst_mi355_allocator_free;
}

static void gst_mi355_allocator_init(GstMi355Allocator *self) {
  GstAllocator *a = GST_ALLOCATOR_CAST(self);
  a->mem_type = "Mi355PinnedMemory";
  a->mem_map = gst_mi355_mem_map;
  a->mem_unmap = gst_mi355_mem_unmap;
  /* no mem_share / mem_copy: the default copy goes through map, sub-memories are not offered */
  GST_OBJECT_FLAG_SET(self, GST_ALLOCATOR_FLAG_CUSTOM_ALLOC);
}

GstAllocator *gst_mi355_allocator_new(mi355_ctx *ctx) {
  GstMi355Allocator *self = g_object_new(GST_TYPE_MI355_ALLOCATOR, NULL);
  self->ctx = ctx;
  gst_object_ref_sink(self);
  return GST_ALLOCATOR_CAST(self);
}

/* propose_allocation: offer upstream a GstVideoBufferPool whose buffers live in pinned memory (the shape of
 * video/colorlut/src/d3d12colorlut/imp.rs:385-424: parse the caps of the query, build a pool, add it and the metas). */
gboolean gst_mi355_propose_pinned_pool(GstBaseTransform *trans, mi355_ctx *ctx, GstQuery *query) {
  GstCaps *caps = NULL;
  gboolean need_pool = FALSE;
  GstVideoInfo info;
  gst_query_parse_allocation(query, &caps, &need_pool);
  if (!caps || !gst_video_info_from_caps(&info, caps)) return FALSE;
  GstAllocator *alloc = gst_mi355_allocator_new(ctx);
  GstAllocationParams params;
  gst_allocation_params_init(&params);
  params.align = 15; /* 16-byte rows: the flat kernels take whole uint4 groups */
  gst_query_add_allocation_param(query, alloc, &params);
  if (need_pool) {
    GstBufferPool *pool = gst_video_buffer_pool_new();
    GstStructure *config = gst_buffer_pool_get_config(pool);
    gst_buffer_pool_config_set_params(config, caps, GST_VIDEO_INFO_SIZE(&info), 2, 0);
    gst_buffer_pool_config_set_allocator(config, alloc, &params);
    gst_buffer_pool_config_add_option(config, GST_BUFFER_POOL_OPTION_VIDEO_META);
    if (!gst_buffer_pool_set_config(pool, config)) {
      gst_object_unref(pool);
      gst_object_unref(alloc);
      return FALSE;
    }
    gst_query_add_allocation_pool(query, pool, GST_VIDEO_INFO_SIZE(&info), 2, 0);
    gst_object_unref(pool);
  }
  gst_query_add_allocation_meta(query, GST_VIDEO_META_API_TYPE, NULL);
  gst_object_unref(alloc);
  (void)trans;
  return TRUE;
}
