/* gst/gstmi355allocator.c — GstAllocator over mi355_host_alloc (page-locked host memory). Buffers of a pool built on it
 * map to pointers the library can hipMemcpyAsync from / to directly (47 GB/s each way), instead of going
 * through a pageable staging copy. Also here: the GstMi355HsvMeta registration shared by the two plugins. */
#include "gstmi355common.h"

struct _GstMi355Allocator {
  GstAllocator parent;
  /* OWN context, created with the allocator and destroyed in finalize. (Round 2 borrowed the element's: the pool offered in
   * propose_allocation belongs to the UPSTREAM element, its buffers are freed whenever upstream lets go of them - after this
   * element's stop() has destroyed that context on a READY <-> PAUSED cycle: use after free. ADVICE r02.) */
  mi355_ctx *ctx;
};
G_DEFINE_TYPE(GstMi355Allocator, gst_mi355_allocator, GST_TYPE_ALLOCATOR)

typedef struct {
  GstMemory mem;
  gpointer data;
} GstMi355Memory;

static GstMemory *gst_mi355_allocator_alloc(GstAllocator *allocator, gsize size, GstAllocationParams *params) {
  GstMi355Allocator *self = GST_MI355_ALLOCATOR(allocator);
  const gsize maxsize = size + params->prefix + params->padding;
  if (!self->ctx) return NULL;
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
  /* every memory holds a reference on its allocator, so self (and self->ctx) are alive here */
  (void)mi355_host_free(self->ctx, m->data);
  g_free(m);
}

static gpointer gst_mi355_mem_map(GstMemory *mem, gsize maxsize, GstMapFlags flags) { return ((GstMi355Memory *)mem)->data; }
static void gst_mi355_mem_unmap(GstMemory *mem) {}

static void gst_mi355_allocator_finalize(GObject *object) {
  GstMi355Allocator *self = GST_MI355_ALLOCATOR(object);
  if (self->ctx) mi355_ctx_destroy(self->ctx);
  self->ctx = NULL;
  G_OBJECT_CLASS(gst_mi355_allocator_parent_class)->finalize(object);
}

static void gst_mi355_allocator_class_init(GstMi355AllocatorClass *klass) {
  GstAllocatorClass *a = GST_ALLOCATOR_CLASS(klass);
  a->alloc = gst_mi355_allocator_alloc;
  a->free = gst_mi355_allocator_free;
  G_OBJECT_CLASS(klass)->finalize = gst_mi355_allocator_finalize;
}

static void gst_mi355_allocator_init(GstMi355Allocator *self) {
  GstAllocator *a = GST_ALLOCATOR_CAST(self);
  a->mem_type = "Mi355PinnedMemory";
  a->mem_map = gst_mi355_mem_map;
  a->mem_unmap = gst_mi355_mem_unmap;
  /* no mem_share / mem_copy: the default copy goes through map, sub-memories are not offered */
  GST_OBJECT_FLAG_SET(self, GST_ALLOCATOR_FLAG_CUSTOM_ALLOC);
}

GstAllocator *gst_mi355_allocator_new(void) {
  GstMi355Allocator *self = g_object_new(GST_TYPE_MI355_ALLOCATOR, NULL);
  int status = 0;
  self->ctx = mi355_ctx_create(0, &status); /* device choice: HIP_VISIBLE_DEVICES, one process per GPU */
  gst_object_ref_sink(self);
  if (!self->ctx) {
    gst_object_unref(self);
    return NULL;
  }
  return GST_ALLOCATOR_CAST(self);
}

GstBufferPool *gst_mi355_pinned_pool_new(GstCaps *caps, const GstVideoInfo *info) {
  GstAllocator *alloc = gst_mi355_allocator_new();
  if (!alloc) return NULL;
  GstAllocationParams params;
  gst_allocation_params_init(&params);
  params.align = 15; /* 16-byte rows: the flat kernels take whole uint4 groups */
  GstBufferPool *pool = gst_video_buffer_pool_new();
  GstStructure *config = gst_buffer_pool_get_config(pool);
  gst_buffer_pool_config_set_params(config, caps, GST_VIDEO_INFO_SIZE(info), 2, 0);
  gst_buffer_pool_config_set_allocator(config, alloc, &params);
  gst_buffer_pool_config_add_option(config, GST_BUFFER_POOL_OPTION_VIDEO_META);
  gst_object_unref(alloc); /* the config holds its own reference */
  if (!gst_buffer_pool_set_config(pool, config)) {
    gst_object_unref(pool);
    return NULL;
  }
  return pool;
}

/* propose_allocation: offer upstream a GstVideoBufferPool whose buffers live in pinned memory (the shape of
 * video/colorlut/src/d3d12colorlut/imp.rs:385-424: parse the caps of the query, build a pool, add it and the metas). */
gboolean gst_mi355_propose_pinned_pool(GstBaseTransform *trans, GstQuery *query) {
  GstCaps *caps = NULL;
  gboolean need_pool = FALSE;
  GstVideoInfo info;
  gst_query_parse_allocation(query, &caps, &need_pool);
  if (!caps || !gst_video_info_from_caps(&info, caps)) return FALSE;
  GstAllocator *alloc = gst_mi355_allocator_new();
  if (!alloc) return FALSE;
  GstAllocationParams params;
  gst_allocation_params_init(&params);
  params.align = 15;
  gst_query_add_allocation_param(query, alloc, &params);
  gst_object_unref(alloc);
  if (need_pool) {
    GstBufferPool *pool = gst_mi355_pinned_pool_new(caps, &info);
    if (!pool) return FALSE;
    gst_query_add_allocation_pool(query, pool, GST_VIDEO_INFO_SIZE(&info), 2, 0);
    gst_object_unref(pool);
  }
  gst_query_add_allocation_meta(query, GST_VIDEO_META_API_TYPE, NULL);
  (void)trans;
  return TRUE;
}

/* ---- device memory (gstmi355common.h "device memory") */
struct _GstMi355DeviceAllocator {
  GstAllocator parent;
  mi355_ctx *ctx; /* owns the buffers' device memory and pinned shadows; outlives every element (see GstMi355Allocator) */
};
G_DEFINE_TYPE(GstMi355DeviceAllocator, gst_mi355_device_allocator, GST_TYPE_ALLOCATOR)

typedef struct {
  GstMemory mem;
  mi355_buf *buf;
} GstMi355DeviceMemory;

static GstMemory *gst_mi355_device_allocator_alloc(GstAllocator *allocator, gsize size, GstAllocationParams *params) {
  GstMi355DeviceAllocator *self = GST_MI355_DEVICE_ALLOCATOR(allocator);
  /* no prefix / padding games on device memory: the kernels address whole frames (hipMalloc is 256-byte aligned) */
  if (!self->ctx || params->prefix || params->padding) return NULL;
  mi355_buf *buf = mi355_buf_alloc(self->ctx, size);
  if (!buf) return NULL;
  GstMi355DeviceMemory *m = g_new0(GstMi355DeviceMemory, 1);
  gst_memory_init(GST_MEMORY_CAST(m), params->flags, allocator, NULL, size, params->align, 0, size);
  m->buf = buf;
  return GST_MEMORY_CAST(m);
}

static void gst_mi355_device_allocator_free(GstAllocator *allocator, GstMemory *memory) {
  GstMi355DeviceMemory *m = (GstMi355DeviceMemory *)memory;
  mi355_buf_unref(m->buf); /* waits for the device work committed on it */
  g_free(m);
}

static gpointer gst_mi355_device_mem_map(GstMemory *mem, gsize maxsize, GstMapFlags flags) {
  const int f = ((flags & GST_MAP_READ) ? MI355_MAP_READ : 0) | ((flags & GST_MAP_WRITE) ? MI355_MAP_WRITE : 0);
  return mi355_buf_map_host(((GstMi355DeviceMemory *)mem)->buf, f ? f : MI355_MAP_READ);
}
static void gst_mi355_device_mem_unmap(GstMemory *mem) { (void)mi355_buf_unmap_host(((GstMi355DeviceMemory *)mem)->buf); }

static void gst_mi355_device_allocator_finalize(GObject *object) {
  GstMi355DeviceAllocator *self = GST_MI355_DEVICE_ALLOCATOR(object);
  if (self->ctx) mi355_ctx_destroy(self->ctx);
  self->ctx = NULL;
  G_OBJECT_CLASS(gst_mi355_device_allocator_parent_class)->finalize(object);
}

static void gst_mi355_device_allocator_class_init(GstMi355DeviceAllocatorClass *klass) {
  GstAllocatorClass *a = GST_ALLOCATOR_CLASS(klass);
  a->alloc = gst_mi355_device_allocator_alloc;
  a->free = gst_mi355_device_allocator_free;
  G_OBJECT_CLASS(klass)->finalize = gst_mi355_device_allocator_finalize;
}

static void gst_mi355_device_allocator_init(GstMi355DeviceAllocator *self) {
  GstAllocator *a = GST_ALLOCATOR_CAST(self);
  a->mem_type = GST_MI355_DEVICE_MEMORY_TYPE;
  a->mem_map = gst_mi355_device_mem_map;
  a->mem_unmap = gst_mi355_device_mem_unmap;
  /* no mem_share (a sub-memory would need its own dirty state), default mem_copy goes through map */
  GST_OBJECT_FLAG_SET(self, GST_ALLOCATOR_FLAG_CUSTOM_ALLOC);
}

/* ONE device allocator - one mi355_ctx with its stream - per process: allocation params, pools and roundedcorners' plane all hold a
 * reference to the same object; the last unref destroys it, the next caller makes a new one. (Statically allocated: a zero-filled GMutex
 * and GWeakRef are valid. g_weak_ref_get hands out a strong reference or NULL, never an object that is being finalized.) */
static GMutex gst_mi355_device_allocator_lock;
static GWeakRef gst_mi355_device_allocator_ref;

GstAllocator *gst_mi355_device_allocator_new(void) {
  g_mutex_lock(&gst_mi355_device_allocator_lock);
  GstAllocator *alloc = (GstAllocator *)g_weak_ref_get(&gst_mi355_device_allocator_ref);
  if (!alloc) {
    GstMi355DeviceAllocator *self = g_object_new(GST_TYPE_MI355_DEVICE_ALLOCATOR, NULL);
    int status = 0;
    self->ctx = mi355_ctx_create(0, &status);
    gst_object_ref_sink(self);
    if (!self->ctx) {
      gst_object_unref(self);
    } else {
      g_weak_ref_set(&gst_mi355_device_allocator_ref, self);
      alloc = GST_ALLOCATOR_CAST(self);
    }
  }
  g_mutex_unlock(&gst_mi355_device_allocator_lock);
  return alloc;
}

mi355_buf *gst_mi355_device_memory_get_buf(GstMemory *mem) {
  if (!mem || !gst_memory_is_type(mem, GST_MI355_DEVICE_MEMORY_TYPE)) return NULL;
  GstMi355DeviceMemory *m = (GstMi355DeviceMemory *)mem;
  if (mem->offset != 0 || mem->size != mi355_buf_size(m->buf)) return NULL; /* resized: let the caller map it */
  return m->buf;
}

mi355_buf *gst_mi355_buffer_peek_device(GstBuffer *buffer) {
  if (!buffer || gst_buffer_n_memory(buffer) != 1) return NULL;
  return gst_mi355_device_memory_get_buf(gst_buffer_peek_memory(buffer, 0));
}

static GstBufferPool *gst_mi355_device_pool_new(GstCaps *caps, const GstVideoInfo *info) {
  GstAllocator *alloc = gst_mi355_device_allocator_new();
  if (!alloc) return NULL;
  GstAllocationParams params;
  gst_allocation_params_init(&params);
  params.align = 15;
  GstBufferPool *pool = gst_video_buffer_pool_new();
  GstStructure *config = gst_buffer_pool_get_config(pool);
  gst_buffer_pool_config_set_params(config, caps, GST_VIDEO_INFO_SIZE(info), 2, 0);
  gst_buffer_pool_config_set_allocator(config, alloc, &params);
  gst_buffer_pool_config_add_option(config, GST_BUFFER_POOL_OPTION_VIDEO_META);
  gst_object_unref(alloc);
  if (!gst_buffer_pool_set_config(pool, config)) {
    gst_object_unref(pool);
    return NULL;
  }
  return pool;
}

/* the shape of d3d12colorlut/imp.rs:385-424: parse the query's caps, build a pool over our memory, add it and the metas */
gboolean gst_mi355_propose_device_pool(GstBaseTransform *trans, GstQuery *query) {
  GstCaps *caps = NULL;
  gboolean need_pool = FALSE;
  GstVideoInfo info;
  gst_query_parse_allocation(query, &caps, &need_pool);
  if (!caps || !gst_video_info_from_caps(&info, caps)) return FALSE;
  GstAllocator *alloc = gst_mi355_device_allocator_new();
  if (!alloc) return FALSE;
  GstAllocationParams params;
  gst_allocation_params_init(&params);
  params.align = 15;
  gst_query_add_allocation_param(query, alloc, &params);
  gst_object_unref(alloc);
  if (need_pool) {
    GstBufferPool *pool = gst_mi355_device_pool_new(caps, &info);
    if (!pool) return FALSE;
    gst_query_add_allocation_pool(query, pool, GST_VIDEO_INFO_SIZE(&info), 2, 0);
    gst_object_unref(pool);
  }
  gst_query_add_allocation_meta(query, GST_VIDEO_META_API_TYPE, NULL);
  (void)trans;
  return TRUE;
}

/* the shape of d3d12colorlut/imp.rs:426-492: keep what downstream offered; with no pool on offer the output is ours */
gboolean gst_mi355_decide_device_pool(GstBaseTransform *trans, GstQuery *query) {
  if (gst_query_get_n_allocation_pools(query) > 0) return TRUE;
  GstCaps *caps = NULL;
  GstVideoInfo info;
  gst_query_parse_allocation(query, &caps, NULL);
  if (!caps || !gst_video_info_from_caps(&info, caps)) return TRUE; /* (not a failure: the default allocation stays) */
  GstBufferPool *pool = gst_mi355_device_pool_new(caps, &info);
  if (!pool) return TRUE;
  gst_query_add_allocation_pool(query, pool, GST_VIDEO_INFO_SIZE(&info), 2, 0);
  gst_object_unref(pool);
  (void)trans;
  return TRUE;
}

/* ---- GstMi355HsvMeta. Both plugins carry this code; whichever is loaded first registers the types, the other finds them
 * by name (a second registration under the same name fails). */
GType gst_mi355_hsv_meta_api_get_type(void) {
  static GType type = 0;
  if (g_once_init_enter(&type)) {
    static const gchar *tags[] = {NULL};
    GType t = g_type_from_name("GstMi355HsvMetaAPI");
    if (!t) t = gst_meta_api_type_register("GstMi355HsvMetaAPI", tags);
    g_once_init_leave(&type, t);
  }
  return type;
}

static gboolean gst_mi355_hsv_meta_init(GstMeta *meta, gpointer params, GstBuffer *buffer) {
  GstMi355HsvMeta *m = (GstMi355HsvMeta *)meta;
  memset(&m->settings, 0, sizeof m->settings);
  return TRUE;
}

/* the meta describes work still to be done on THESE pixels: it follows a plain copy of the buffer and nothing else */
static gboolean gst_mi355_hsv_meta_transform(GstBuffer *dest, GstMeta *meta, GstBuffer *buffer, GQuark type, gpointer data) {
  if (!GST_META_TRANSFORM_IS_COPY(type)) return FALSE;
  const GstMetaTransformCopy *copy = data;
  if (copy->region) return FALSE;
  return gst_buffer_add_mi355_hsv_meta(dest, &((GstMi355HsvMeta *)meta)->settings) != NULL;
}

const GstMetaInfo *gst_mi355_hsv_meta_get_info(void) {
  static const GstMetaInfo *info = NULL;
  if (g_once_init_enter((GstMetaInfo **)&info)) {
    const GstMetaInfo *mi = gst_meta_get_info("GstMi355HsvMeta");
    if (!mi)
      mi = gst_meta_register(GST_MI355_HSV_META_API_TYPE, "GstMi355HsvMeta", sizeof(GstMi355HsvMeta), gst_mi355_hsv_meta_init, NULL,
                             gst_mi355_hsv_meta_transform);
    g_once_init_leave((GstMetaInfo **)&info, (GstMetaInfo *)mi);
  }
  return info;
}

GstMi355HsvMeta *gst_buffer_add_mi355_hsv_meta(GstBuffer *b, const mi355_hsv_settings *s) {
  GstMi355HsvMeta *m = (GstMi355HsvMeta *)gst_buffer_add_meta(b, gst_mi355_hsv_meta_get_info(), NULL);
  if (m) m->settings = *s;
  return m;
}
