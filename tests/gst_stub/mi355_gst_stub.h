/* tests/gst_stub/mi355_gst_stub.h — TEST SCAFFOLDING, not GStreamer.
 *
 * This image has no GLib / GStreamer development files, so the shim under gst/ cannot be built here. To still have a
 * compiler over every line of it, `make -C gst syntax` runs `gcc -fsyntax-only` against THIS header (through the forwarding
 * files next to it: gst/gst.h, gst/video/video.h, ...). It declares - by hand, with the signatures of the GLib 2.x /
 * GStreamer 1.x C API as documented - exactly the types, macros and functions the shim uses. What a green `make syntax`
 * proves: the shim is valid C, every mi355_* call matches include/mi355fx.h (the real header), every vfunc has the shape of
 * the class member it is assigned to as declared here. What it does NOT prove: anything about GStreamer's behaviour, or that
 * these declarations agree with a real installation - on a box that has one, `make -C gst` builds against the real headers.
 * Nothing in here is linked, shipped or used by the product. */
#ifndef MI355_GST_STUB_H
#define MI355_GST_STUB_H
#include <stddef.h>
#include <stdint.h>
#include <string.h>

/* ---------------------------------------------------------------- GLib */
typedef int gint; typedef unsigned int guint; typedef int gboolean; typedef char gchar; typedef unsigned char guchar;
typedef float gfloat; typedef double gdouble; typedef void *gpointer; typedef const void *gconstpointer;
typedef uint8_t guint8; typedef uint16_t guint16; typedef uint32_t guint32; typedef uint64_t guint64; typedef int64_t gint64; typedef int32_t gint32;
typedef size_t gsize; typedef ptrdiff_t gssize; typedef unsigned long gulong;
typedef gsize GType; typedef guint32 GQuark;
#define TRUE 1
#define FALSE 0
#define G_BEGIN_DECLS
#define G_END_DECLS
#define G_MAXFLOAT 3.402823466e+38f
#define G_MAXDOUBLE 1.7976931348623157e+308
#define G_MAXINT 2147483647
#define G_MAXUINT 4294967295u
#define G_MAXUINT64 18446744073709551615ull
#define G_BIG_ENDIAN 4321
#define G_LITTLE_ENDIAN 1234
#define G_BYTE_ORDER G_LITTLE_ENDIAN
#define G_N_ELEMENTS(a) (sizeof(a) / sizeof((a)[0]))
#define MIN(a, b) (((a) < (b)) ? (a) : (b))
#define MAX(a, b) (((a) > (b)) ? (a) : (b))
#define G_CALLBACK(f) ((GCallback)(f))
typedef void (*GCallback)(void);
typedef struct _GList { gpointer data; struct _GList *next, *prev; } GList;
typedef struct { gpointer p; guint i[2]; } GMutex;
typedef struct { GList *head, *tail; guint length; } GQueue;
typedef struct _GBytes GBytes;
typedef struct _GError GError;
typedef struct { GType g_type; guint64 data[2]; } GValue;
#define G_VALUE_INIT {0, {0, 0}}
typedef struct { gint value; const gchar *value_name, *value_nick; } GEnumValue;
typedef struct { guint value; const gchar *value_name, *value_nick; } GFlagsValue;
typedef enum { G_PARAM_READABLE = 1, G_PARAM_WRITABLE = 2, G_PARAM_READWRITE = 3, G_PARAM_STATIC_STRINGS = 0xe0 } GParamFlags;
typedef enum { G_SIGNAL_RUN_LAST = 2, G_SIGNAL_ACTION = 32 } GSignalFlags;
typedef struct _GParamSpec GParamSpec;
typedef struct _GTypeClass { GType g_type; } GTypeClass;
typedef struct _GObject { GTypeClass *g_class; guint ref_count; gpointer qdata; } GObject;
typedef struct _GObjectClass {
  GTypeClass g_type_class;
  void (*set_property)(GObject *object, guint property_id, const GValue *value, GParamSpec *pspec);
  void (*get_property)(GObject *object, guint property_id, GValue *value, GParamSpec *pspec);
  void (*dispose)(GObject *object);
  void (*finalize)(GObject *object);
} GObjectClass;
#define G_OBJECT_CLASS(k) ((GObjectClass *)(k))
#define G_TYPE_FROM_CLASS(k) (((GTypeClass *)(k))->g_type)
#define G_TYPE_NONE ((GType)4)
#define G_TYPE_INT ((GType)24)
#define G_TYPE_UINT64 ((GType)44)
#define G_TYPE_FLOAT ((GType)56)
#define G_TYPE_DOUBLE ((GType)60)
#define G_TYPE_STRING ((GType)64)
#define G_TYPE_POINTER ((GType)68)
GType g_bytes_get_type(void);
#define G_TYPE_BYTES (g_bytes_get_type())
void g_object_warn_invalid_property_id_stub(gpointer object, guint id, GParamSpec *pspec);
#define G_OBJECT_WARN_INVALID_PROPERTY_ID(o, id, p) g_object_warn_invalid_property_id_stub((o), (id), (p))
void g_mutex_init(GMutex *m); void g_mutex_clear(GMutex *m); void g_mutex_lock(GMutex *m); void g_mutex_unlock(GMutex *m);
typedef struct { union { gpointer p; } priv; } GWeakRef;   /* (gobject.h) */
void g_weak_ref_set(GWeakRef *weak_ref, gpointer object); gpointer g_weak_ref_get(GWeakRef *weak_ref);
gint g_atomic_int_get_stub(const volatile gint *p); void g_atomic_int_set_stub(volatile gint *p, gint v); gboolean g_atomic_int_cas_stub(volatile gint *p, gint o, gint n);
#define g_atomic_int_get(p) g_atomic_int_get_stub(p)
#define g_atomic_int_set(p, v) g_atomic_int_set_stub((p), (v))
#define g_atomic_int_compare_and_exchange(p, o, n) g_atomic_int_cas_stub((p), (o), (n))
gboolean g_once_init_enter_stub(volatile gsize *loc); void g_once_init_leave_stub(volatile gsize *loc, gsize v);
#define g_once_init_enter(l) g_once_init_enter_stub((volatile gsize *)(l))
#define g_once_init_leave(l, v) g_once_init_leave_stub((volatile gsize *)(l), (gsize)(v))
void g_free(gpointer p); gpointer g_malloc0_n(gsize n, gsize s); gpointer g_malloc_n(gsize n, gsize s);
#define g_new0(T, n) ((T *)g_malloc0_n((n), sizeof(T)))
#define g_new(T, n) ((T *)g_malloc_n((n), sizeof(T)))
gchar *g_strdup(const gchar *s); const gchar *g_getenv(const gchar *name);
gboolean g_file_get_contents(const gchar *filename, gchar **contents, gsize *length, GError **error);
gboolean g_uint64_checked_mul(guint64 *dest, guint64 a, guint64 b);
gconstpointer g_bytes_get_data(GBytes *bytes, gsize *size); void g_bytes_unref(GBytes *bytes);
void g_queue_init(GQueue *q); gboolean g_queue_is_empty(GQueue *q); guint g_queue_get_length(GQueue *q); gpointer g_queue_pop_head(GQueue *q); void g_queue_push_tail(GQueue *q, gpointer data);
gpointer g_object_new(GType type, const gchar *first_property_name, ...); void g_object_unref(gpointer object);
void g_object_class_install_property(GObjectClass *oclass, guint property_id, GParamSpec *pspec);
GParamSpec *g_param_spec_float(const gchar *name, const gchar *nick, const gchar *blurb, gfloat min, gfloat max, gfloat def, GParamFlags flags);
GParamSpec *g_param_spec_double(const gchar *name, const gchar *nick, const gchar *blurb, gdouble min, gdouble max, gdouble def, GParamFlags flags);
GParamSpec *g_param_spec_uint(const gchar *name, const gchar *nick, const gchar *blurb, guint min, guint max, guint def, GParamFlags flags);
GParamSpec *g_param_spec_uint64(const gchar *name, const gchar *nick, const gchar *blurb, guint64 min, guint64 max, guint64 def, GParamFlags flags);
GParamSpec *g_param_spec_boolean(const gchar *name, const gchar *nick, const gchar *blurb, gboolean def, GParamFlags flags);
GParamSpec *g_param_spec_string(const gchar *name, const gchar *nick, const gchar *blurb, const gchar *def, GParamFlags flags);
GParamSpec *g_param_spec_enum(const gchar *name, const gchar *nick, const gchar *blurb, GType enum_type, gint def, GParamFlags flags);
GParamSpec *g_param_spec_flags(const gchar *name, const gchar *nick, const gchar *blurb, GType flags_type, guint def, GParamFlags flags);
GParamSpec *g_param_spec_boxed(const gchar *name, const gchar *nick, const gchar *blurb, GType boxed_type, GParamFlags flags);
GType g_enum_register_static(const gchar *name, const GEnumValue *values); GType g_flags_register_static(const gchar *name, const GFlagsValue *values);
GType g_type_from_name(const gchar *name);
GValue *g_value_init(GValue *value, GType type); void g_value_unset(GValue *value);
gfloat g_value_get_float(const GValue *v); void g_value_set_float(GValue *v, gfloat f);
gdouble g_value_get_double(const GValue *v); void g_value_set_double(GValue *v, gdouble d);
guint g_value_get_uint(const GValue *v); void g_value_set_uint(GValue *v, guint u);
guint64 g_value_get_uint64(const GValue *v); void g_value_set_uint64(GValue *v, guint64 u);
gboolean g_value_get_boolean(const GValue *v); void g_value_set_boolean(GValue *v, gboolean b);
gint g_value_get_enum(const GValue *v); void g_value_set_enum(GValue *v, gint e);
guint g_value_get_flags(const GValue *v); void g_value_set_flags(GValue *v, guint f);
void g_value_set_string(GValue *v, const gchar *s); gchar *g_value_dup_string(const GValue *v);
gpointer g_value_dup_boxed(const GValue *v); void g_value_set_boxed(GValue *v, gconstpointer boxed); void g_value_take_boxed(GValue *v, gconstpointer boxed);
gulong g_signal_connect_stub(gpointer instance, const gchar *signal, GCallback handler, gpointer data);
#define g_signal_connect(i, s, h, d) g_signal_connect_stub((i), (s), (h), (d))
guint g_signal_new_class_handler(const gchar *name, GType itype, GSignalFlags flags, GCallback class_handler, gpointer accumulator, gpointer accu_data,
                                 gpointer c_marshaller, GType return_type, guint n_params, ...);
/* type definition macros: the shapes GLib's expand to, without the registration machinery */
GType mi355_stub_register_type(const gchar *name, GType parent, gsize class_size, gsize instance_size);
#define G_DECLARE_FINAL_TYPE(ModuleObjName, module_obj_name, MODULE, OBJ_NAME, ParentName) \
  GType module_obj_name##_get_type(void);                                                  \
  typedef struct _##ModuleObjName ModuleObjName;                                           \
  typedef struct { ParentName##Class parent_class; } ModuleObjName##Class;                 \
  static inline ModuleObjName *MODULE##_##OBJ_NAME(gpointer ptr) { return (ModuleObjName *)ptr; }
#define G_DEFINE_TYPE(TypeName, type_name, TYPE_PARENT)                                                                   \
  static void type_name##_init(TypeName *self);                                                                           \
  static void type_name##_class_init(TypeName##Class *klass);                                                             \
  static gpointer type_name##_parent_class = NULL;                                                                        \
  GType type_name##_get_type(void) {                                                                                      \
    static GType t = 0;                                                                                                   \
    if (!t) {                                                                                                             \
      void (*ci)(TypeName##Class *) = type_name##_class_init;                                                             \
      void (*ii)(TypeName *) = type_name##_init;                                                                          \
      (void)ci; (void)ii; (void)type_name##_parent_class;                                                                 \
      t = mi355_stub_register_type(#TypeName, TYPE_PARENT, sizeof(TypeName##Class), sizeof(TypeName));                    \
    }                                                                                                                     \
    return t;                                                                                                             \
  }

/* ---------------------------------------------------------------- GStreamer core */
typedef guint64 GstClockTime;
#define GST_SECOND ((GstClockTime)1000000000)
#define GST_CLOCK_TIME_NONE ((GstClockTime)-1)
#define GST_CLOCK_TIME_IS_VALID(t) (((GstClockTime)(t)) != GST_CLOCK_TIME_NONE)
#define GST_VERSION_MAJOR 1
#define GST_VERSION_MINOR 24
typedef enum { GST_FLOW_OK = 0, GST_FLOW_EOS = -3, GST_FLOW_NOT_NEGOTIATED = -4, GST_FLOW_ERROR = -5 } GstFlowReturn;
typedef enum { GST_PAD_UNKNOWN, GST_PAD_SRC, GST_PAD_SINK } GstPadDirection;
typedef enum { GST_PAD_ALWAYS, GST_PAD_SOMETIMES, GST_PAD_REQUEST } GstPadPresence;
typedef enum { GST_MAP_READ = 1, GST_MAP_WRITE = 2, GST_MAP_READWRITE = 3 } GstMapFlags;
typedef enum { GST_RANK_NONE = 0 } GstRank;
typedef enum { GST_FORMAT_UNDEFINED = 0, GST_FORMAT_TIME = 3 } GstFormat;
typedef enum { GST_CAPS_INTERSECT_ZIG_ZAG = 0, GST_CAPS_INTERSECT_FIRST = 1 } GstCapsIntersectMode;
typedef enum { GST_EVENT_FLUSH_STOP = 1, GST_EVENT_CAPS, GST_EVENT_SEGMENT, GST_EVENT_EOS, GST_EVENT_GAP, GST_EVENT_RECONFIGURE } GstEventType;
typedef enum { GST_QUERY_LATENCY = 1, GST_QUERY_CUSTOM } GstQueryType;
typedef enum { GST_STATE_CHANGE_FAILURE = 0, GST_STATE_CHANGE_SUCCESS = 1 } GstStateChangeReturn;
typedef enum { GST_STATE_CHANGE_NULL_TO_READY = 10, GST_STATE_CHANGE_READY_TO_PAUSED = 19, GST_STATE_CHANGE_PAUSED_TO_READY = 26, GST_STATE_CHANGE_READY_TO_NULL = 17 } GstStateChange;
typedef enum { GST_BUFFER_COPY_ALL = 0x1f } GstBufferCopyFlags;
typedef enum { GST_MEMORY_FLAG_ZERO_PREFIXED = 64, GST_MEMORY_FLAG_ZERO_PADDED = 128 } GstMemoryFlags;
#define GST_PARAM_MUTABLE_READY ((GParamFlags)(1 << 10))
#define GST_PARAM_MUTABLE_PLAYING ((GParamFlags)(1 << 12))
typedef struct _GstObject { GObject object; GMutex lock; gchar *name; struct _GstObject *parent; guint flags; } GstObject;
typedef struct _GstCaps GstCaps; typedef struct _GstStructure GstStructure; typedef struct _GstEvent GstEvent; typedef struct _GstQuery GstQuery;
typedef struct _GstMessage GstMessage; typedef struct _GstPlugin GstPlugin; typedef struct _GstPadTemplate GstPadTemplate; typedef struct _GstBufferPool GstBufferPool;
typedef struct _GstDebugCategory GstDebugCategory;
typedef struct { guint flags; gdouble rate, applied_rate; GstFormat format; guint64 base, offset, start, stop, time, position, duration; } GstSegment;
typedef struct _GstPad { GstObject object; gpointer element_private; GstPadTemplate *padtemplate; GstPadDirection direction; } GstPad;
typedef struct _GstElement { GstObject object; GMutex state_lock; guint16 numpads; GList *pads; guint16 numsrcpads; GList *srcpads; guint16 numsinkpads; GList *sinkpads; } GstElement;
typedef struct _GstElementClass {
  GObjectClass parent_class;
  GstPad *(*request_new_pad)(GstElement *element, GstPadTemplate *templ, const gchar *name, const GstCaps *caps);
  void (*release_pad)(GstElement *element, GstPad *pad);
  GstStateChangeReturn (*change_state)(GstElement *element, GstStateChange transition);
} GstElementClass;
typedef struct { const gchar *string; GstCaps *caps; } GstStaticCaps;
typedef struct { const gchar *name_template; GstPadDirection direction; GstPadPresence presence; GstStaticCaps static_caps; } GstStaticPadTemplate;
#define GST_STATIC_CAPS(s) {(s), NULL}
#define GST_STATIC_PAD_TEMPLATE(n, d, p, c) {(n), (d), (p), c}
typedef struct _GstAllocator GstAllocator;
typedef struct { guint flags; gsize align, prefix, padding; } GstAllocationParams;
typedef struct _GstMemory { gpointer mini_object[8]; GstAllocator *allocator; struct _GstMemory *parent; gsize maxsize, align, offset, size; } GstMemory;
typedef gpointer (*GstMemoryMapFunction)(GstMemory *mem, gsize maxsize, GstMapFlags flags);
typedef void (*GstMemoryUnmapFunction)(GstMemory *mem);
struct _GstAllocator { GstObject object; const gchar *mem_type; GstMemoryMapFunction mem_map; GstMemoryUnmapFunction mem_unmap; };
typedef struct { GObjectClass object_class; GstMemory *(*alloc)(GstAllocator *allocator, gsize size, GstAllocationParams *params); void (*free)(GstAllocator *allocator, GstMemory *memory); } GstAllocatorClass;
typedef struct { GstMemory *memory; GstMapFlags flags; guint8 *data; gsize size, maxsize; gpointer user_data[4]; } GstMapInfo;
typedef struct _GstBuffer { gpointer mini_object[8]; GstBufferPool *pool; GstClockTime pts, dts, duration; guint64 offset, offset_end; } GstBuffer;
typedef struct _GstMetaInfo GstMetaInfo;
typedef struct { guint flags; const GstMetaInfo *info; } GstMeta;
typedef struct { gboolean region; gsize offset, size; } GstMetaTransformCopy;
typedef gboolean (*GstMetaInitFunction)(GstMeta *meta, gpointer params, GstBuffer *buffer);
typedef void (*GstMetaFreeFunction)(GstMeta *meta, GstBuffer *buffer);
typedef gboolean (*GstMetaTransformFunction)(GstBuffer *transbuf, GstMeta *meta, GstBuffer *buffer, GQuark type, gpointer data);
typedef GstFlowReturn (*GstPadChainFunction)(GstPad *pad, GstObject *parent, GstBuffer *buffer);
typedef gboolean (*GstPadEventFunction)(GstPad *pad, GstObject *parent, GstEvent *event);
typedef gboolean (*GstPadQueryFunction)(GstPad *pad, GstObject *parent, GstQuery *query);
GType gst_element_get_type(void); GType gst_allocator_get_type(void); GType gst_pad_get_type(void); GType gst_structure_get_type(void);
GType gst_value_list_get_type(void); GType gst_value_array_get_type(void); GType gst_bitmask_get_type(void); GType gst_int_range_get_type(void);
#define GST_TYPE_ELEMENT (gst_element_get_type())
#define GST_TYPE_ALLOCATOR (gst_allocator_get_type())
#define GST_TYPE_PAD (gst_pad_get_type())
#define GST_TYPE_STRUCTURE (gst_structure_get_type())
#define GST_TYPE_LIST (gst_value_list_get_type())
#define GST_TYPE_ARRAY (gst_value_array_get_type())
#define GST_TYPE_BITMASK (gst_bitmask_get_type())
#define GST_TYPE_INT_RANGE (gst_int_range_get_type())
#define GST_TYPE_CLOCK_TIME G_TYPE_UINT64
#define GST_OBJECT(o) ((GstObject *)(o))
#define GST_ELEMENT(o) ((GstElement *)(o))
#define GST_PAD(o) ((GstPad *)(o))
#define GST_ELEMENT_CLASS(k) ((GstElementClass *)(k))
#define GST_ALLOCATOR_CAST(o) ((GstAllocator *)(o))
#define GST_ALLOCATOR_CLASS(k) ((GstAllocatorClass *)(k))
#define GST_MEMORY_CAST(m) ((GstMemory *)(m))
#define GST_ALLOCATOR_FLAG_CUSTOM_ALLOC (1u << 4)
#define GST_OBJECT_FLAG_SET(o, f) (GST_OBJECT(o)->flags |= (f))
#define GST_OBJECT_LOCK(o) g_mutex_lock(&GST_OBJECT(o)->lock)
#define GST_OBJECT_UNLOCK(o) g_mutex_unlock(&GST_OBJECT(o)->lock)
gboolean gst_is_element_stub(gconstpointer o);
#define GST_IS_ELEMENT(o) gst_is_element_stub(o)
#define GST_PAD_NAME(p) (GST_OBJECT(p)->name)
#define GST_PAD_DIRECTION(p) (GST_PAD(p)->direction)
void gst_pad_set_proxy_caps_stub(GstPad *pad);
#define GST_PAD_SET_PROXY_CAPS(p) gst_pad_set_proxy_caps_stub(p)
#define GST_EVENT_TYPE(e) gst_event_type_stub(e)
#define GST_QUERY_TYPE(q) gst_query_type_stub(q)
GstEventType gst_event_type_stub(GstEvent *e); GstQueryType gst_query_type_stub(GstQuery *q);
#define GST_BUFFER_PTS(b) (((GstBuffer *)(b))->pts)
#define GST_BUFFER_DURATION(b) (((GstBuffer *)(b))->duration)
#define GST_BUFFER_PTS_IS_VALID(b) GST_CLOCK_TIME_IS_VALID(GST_BUFFER_PTS(b))
gboolean gst_buffer_is_discont_stub(GstBuffer *b);
#define GST_BUFFER_IS_DISCONT(b) gst_buffer_is_discont_stub(b)
#define GST_META_TRANSFORM_IS_COPY(type) ((type) == gst_meta_transform_copy_quark_stub())
GQuark gst_meta_transform_copy_quark_stub(void);
#define GST_VALUE_HOLDS_STRUCTURE(v) ((v)->g_type == GST_TYPE_STRUCTURE)
/* debug + error reporting: the arguments are evaluated as a printf call so that format strings are checked */
void mi355_stub_log(gconstpointer obj, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
#define GST_DEBUG_CATEGORY_STATIC(cat) static GstDebugCategory *cat = NULL
#define GST_DEBUG_CATEGORY_INIT(cat, name, color, desc) ((void)(cat), (void)(name), (void)(desc))
#define GST_ERROR_OBJECT(obj, ...) mi355_stub_log((obj), __VA_ARGS__)
#define GST_WARNING_OBJECT(obj, ...) mi355_stub_log((obj), __VA_ARGS__)
#define GST_INFO_OBJECT(obj, ...) mi355_stub_log((obj), __VA_ARGS__)
#define GST_DEBUG_OBJECT(obj, ...) mi355_stub_log((obj), __VA_ARGS__)
gchar *mi355_stub_element_message_printf(const gchar *format, ...) __attribute__((format(printf, 1, 2))); /* _gst_element_error_printf: NULL allowed */
#define GST_ELEMENT_ERROR(el, domain, code, text, debug) do { (void)(el); (void)mi355_stub_element_message_printf text; (void)mi355_stub_element_message_printf debug; } while (0)
#define GST_ELEMENT_WARNING(el, domain, code, text, debug) do { (void)(el); (void)mi355_stub_element_message_printf text; (void)mi355_stub_element_message_printf debug; } while (0)
#define GST_PLUGIN_DEFINE(major, minor, name, description, init, version, license, package, origin) \
  gboolean gst_plugin_##name##_register(void);                                                      \
  const void *gst_plugin_##name##_get_desc(void);                                                   \
  gboolean gst_plugin_##name##_register(void) { gboolean (*f)(GstPlugin *) = init; return f != NULL; } \
  const void *gst_plugin_##name##_get_desc(void) { return description version license package origin PACKAGE; }
gpointer gst_object_ref(gpointer object); void gst_object_unref(gpointer object); gpointer gst_object_ref_sink(gpointer object);
guint64 gst_util_uint64_scale(guint64 val, guint64 num, guint64 denom);
gboolean gst_element_register(GstPlugin *plugin, const gchar *name, guint rank, GType type);
void gst_element_class_set_static_metadata(GstElementClass *klass, const gchar *longname, const gchar *classification, const gchar *description, const gchar *author);
void gst_element_class_add_static_pad_template(GstElementClass *klass, GstStaticPadTemplate *templ);
void gst_element_class_add_static_pad_template_with_gtype(GstElementClass *klass, GstStaticPadTemplate *templ, GType pad_type);
gboolean gst_element_post_message(GstElement *element, GstMessage *message); gboolean gst_element_add_pad(GstElement *element, GstPad *pad);
GstMessage *gst_message_new_latency(GstObject *src); GstMessage *gst_message_new_element(GstObject *src, GstStructure *structure);
GstPad *gst_pad_new_from_static_template(GstStaticPadTemplate *templ, const gchar *name);
void gst_pad_set_chain_function_stub(GstPad *pad, GstPadChainFunction f); void gst_pad_set_event_function_stub(GstPad *pad, GstPadEventFunction f); void gst_pad_set_query_function_stub(GstPad *pad, GstPadQueryFunction f);
#define gst_pad_set_chain_function(p, f) gst_pad_set_chain_function_stub((p), (f))
#define gst_pad_set_event_function(p, f) gst_pad_set_event_function_stub((p), (f))
#define gst_pad_set_query_function(p, f) gst_pad_set_query_function_stub((p), (f))
GstFlowReturn gst_pad_push(GstPad *pad, GstBuffer *buffer); gboolean gst_pad_peer_query(GstPad *pad, GstQuery *query); gboolean gst_pad_query(GstPad *pad, GstQuery *query);
gboolean gst_pad_query_default(GstPad *pad, GstObject *parent, GstQuery *query); gboolean gst_pad_event_default(GstPad *pad, GstObject *parent, GstEvent *event);
GstPad *gst_pad_get_peer(GstPad *pad); GstObject *gst_pad_get_parent_stub(GstPad *pad); GstCaps *gst_pad_get_current_caps(GstPad *pad);
#define gst_pad_get_parent(p) gst_pad_get_parent_stub(p)
void gst_event_parse_caps(GstEvent *event, GstCaps **caps); void gst_event_unref(GstEvent *event);
GstQuery *gst_query_new_custom(GstQueryType type, GstStructure *structure); GstQuery *gst_query_new_latency(void); void gst_query_unref(GstQuery *q);
const GstStructure *gst_query_get_structure(GstQuery *query); GstStructure *gst_query_writable_structure(GstQuery *query);
void gst_query_parse_latency(GstQuery *query, gboolean *live, GstClockTime *min_latency, GstClockTime *max_latency);
void gst_query_set_latency(GstQuery *query, gboolean live, GstClockTime min_latency, GstClockTime max_latency);
void gst_query_parse_allocation(GstQuery *query, GstCaps **caps, gboolean *need_pool); guint gst_query_get_n_allocation_pools(GstQuery *query);
void gst_query_add_allocation_pool(GstQuery *query, GstBufferPool *pool, guint size, guint min_buffers, guint max_buffers);
void gst_query_add_allocation_param(GstQuery *query, GstAllocator *allocator, const GstAllocationParams *params);
void gst_query_add_allocation_meta(GstQuery *query, GType api, const GstStructure *params);
GstStructure *gst_structure_new(const gchar *name, const gchar *firstfield, ...); GstStructure *gst_structure_new_empty(const gchar *name);
gboolean gst_structure_has_name(const GstStructure *structure, const gchar *name); gboolean gst_structure_get(const GstStructure *structure, const char *first_fieldname, ...);
void gst_structure_set(GstStructure *structure, const gchar *fieldname, ...); void gst_structure_set_value(GstStructure *structure, const gchar *fieldname, const GValue *value);
void gst_structure_take_value(GstStructure *structure, const gchar *fieldname, GValue *value); void gst_structure_remove_field(GstStructure *structure, const gchar *fieldname);
gboolean gst_structure_get_enum(const GstStructure *structure, const gchar *fieldname, GType enumtype, gint *value);
GstCaps *gst_caps_copy(const GstCaps *caps); GstCaps *gst_caps_ref(GstCaps *caps); void gst_caps_unref(GstCaps *caps); GstCaps *gst_caps_from_string(const gchar *string);
guint gst_caps_get_size(const GstCaps *caps); GstStructure *gst_caps_get_structure(const GstCaps *caps, guint index);
GstCaps *gst_caps_intersect_full(GstCaps *caps1, GstCaps *caps2, GstCapsIntersectMode mode); gboolean gst_caps_can_intersect(const GstCaps *caps1, const GstCaps *caps2);
void gst_value_list_append_and_take_value(GValue *value, GValue *append_value); void gst_value_array_append_and_take_value(GValue *value, GValue *append_value);
guint gst_value_array_get_size(const GValue *value); const GValue *gst_value_array_get_value(const GValue *value, guint index); const GstStructure *gst_value_get_structure(const GValue *value);
GParamSpec *gst_param_spec_array(const gchar *name, const gchar *nick, const gchar *blurb, GParamSpec *element_spec, GParamFlags flags);
guint64 gst_segment_to_running_time(const GstSegment *segment, GstFormat format, guint64 position); guint64 gst_segment_to_stream_time(const GstSegment *segment, GstFormat format, guint64 position);
GstBuffer *gst_buffer_ref(GstBuffer *buf); void gst_buffer_unref(GstBuffer *buf); GstBuffer *gst_buffer_copy(const GstBuffer *buf);
GstBuffer *gst_buffer_new_wrapped(gpointer data, gsize size); GstBuffer *gst_buffer_new_allocate(GstAllocator *allocator, gsize size, GstAllocationParams *params);
gboolean gst_buffer_map(GstBuffer *buffer, GstMapInfo *info, GstMapFlags flags); void gst_buffer_unmap(GstBuffer *buffer, GstMapInfo *info);
gsize gst_buffer_get_size(GstBuffer *buffer); void gst_buffer_set_size(GstBuffer *buffer, gssize size); gsize gst_buffer_memset(GstBuffer *buffer, gsize offset, guint8 val, gsize size);
gboolean gst_buffer_is_writable_stub(const GstBuffer *buf);
#define gst_buffer_is_writable(b) gst_buffer_is_writable_stub(b)
void gst_buffer_append_memory(GstBuffer *buffer, GstMemory *mem); void gst_buffer_remove_all_memory(GstBuffer *buffer);
guint gst_buffer_n_memory(GstBuffer *buffer); GstMemory *gst_buffer_peek_memory(GstBuffer *buffer, guint idx); gboolean gst_memory_is_type(GstMemory *mem, const gchar *mem_type);
gboolean gst_buffer_copy_into(GstBuffer *dest, GstBuffer *src, GstBufferCopyFlags flags, gsize offset, gsize size);
GstMeta *gst_buffer_get_meta(GstBuffer *buffer, GType api); GstMeta *gst_buffer_add_meta(GstBuffer *buffer, const GstMetaInfo *info, gpointer params); gboolean gst_buffer_remove_meta(GstBuffer *buffer, GstMeta *meta);
GType gst_meta_api_type_register(const gchar *api, const gchar **tags); const GstMetaInfo *gst_meta_get_info(const gchar *impl);
const GstMetaInfo *gst_meta_register(GType api, const gchar *impl, gsize size, GstMetaInitFunction init_func, GstMetaFreeFunction free_func, GstMetaTransformFunction transform_func);
void gst_memory_init(GstMemory *mem, guint flags, GstAllocator *allocator, GstMemory *parent, gsize maxsize, gsize align, gsize offset, gsize size);
GstMemory *gst_memory_ref(GstMemory *mem); void gst_memory_unref(GstMemory *mem); gboolean gst_memory_map(GstMemory *mem, GstMapInfo *info, GstMapFlags flags); void gst_memory_unmap(GstMemory *mem, GstMapInfo *info);
GstMemory *gst_allocator_alloc(GstAllocator *allocator, gsize size, GstAllocationParams *params); void gst_allocation_params_init(GstAllocationParams *params);
GstStructure *gst_buffer_pool_get_config(GstBufferPool *pool); gboolean gst_buffer_pool_set_config(GstBufferPool *pool, GstStructure *config);
void gst_buffer_pool_config_set_params(GstStructure *config, GstCaps *caps, guint size, guint min_buffers, guint max_buffers);
void gst_buffer_pool_config_set_allocator(GstStructure *config, GstAllocator *allocator, const GstAllocationParams *params); void gst_buffer_pool_config_add_option(GstStructure *config, const gchar *option);

/* ---------------------------------------------------------------- gst-base */
typedef struct _GstBaseTransform { GstElement element; GstPad *sinkpad, *srcpad; gboolean have_segment; GstSegment segment; GstBuffer *queued_buf; } GstBaseTransform;
typedef struct _GstBaseTransformClass {
  GstElementClass parent_class;
  gboolean passthrough_on_same_caps, transform_ip_on_passthrough;
  GstCaps *(*transform_caps)(GstBaseTransform *trans, GstPadDirection direction, GstCaps *caps, GstCaps *filter);
  GstCaps *(*fixate_caps)(GstBaseTransform *trans, GstPadDirection direction, GstCaps *caps, GstCaps *othercaps);
  gboolean (*accept_caps)(GstBaseTransform *trans, GstPadDirection direction, GstCaps *caps);
  gboolean (*set_caps)(GstBaseTransform *trans, GstCaps *incaps, GstCaps *outcaps);
  gboolean (*query)(GstBaseTransform *trans, GstPadDirection direction, GstQuery *query);
  gboolean (*decide_allocation)(GstBaseTransform *trans, GstQuery *query);
  gboolean (*filter_meta)(GstBaseTransform *trans, GstQuery *query, GType api, const GstStructure *params);
  gboolean (*propose_allocation)(GstBaseTransform *trans, GstQuery *decide_query, GstQuery *query);
  gboolean (*transform_size)(GstBaseTransform *trans, GstPadDirection direction, GstCaps *caps, gsize size, GstCaps *othercaps, gsize *othersize);
  gboolean (*get_unit_size)(GstBaseTransform *trans, GstCaps *caps, gsize *size);
  gboolean (*start)(GstBaseTransform *trans);
  gboolean (*stop)(GstBaseTransform *trans);
  gboolean (*sink_event)(GstBaseTransform *trans, GstEvent *event);
  gboolean (*src_event)(GstBaseTransform *trans, GstEvent *event);
  GstFlowReturn (*prepare_output_buffer)(GstBaseTransform *trans, GstBuffer *input, GstBuffer **outbuf);
  gboolean (*copy_metadata)(GstBaseTransform *trans, GstBuffer *input, GstBuffer *outbuf);
  gboolean (*transform_meta)(GstBaseTransform *trans, GstBuffer *outbuf, GstMeta *meta, GstBuffer *inbuf);
  void (*before_transform)(GstBaseTransform *trans, GstBuffer *buffer);
  GstFlowReturn (*transform)(GstBaseTransform *trans, GstBuffer *inbuf, GstBuffer *outbuf);
  GstFlowReturn (*transform_ip)(GstBaseTransform *trans, GstBuffer *buf);
  GstFlowReturn (*submit_input_buffer)(GstBaseTransform *trans, gboolean is_discont, GstBuffer *input);
  GstFlowReturn (*generate_output)(GstBaseTransform *trans, GstBuffer **outbuf);
} GstBaseTransformClass;
GType gst_base_transform_get_type(void);
#define GST_TYPE_BASE_TRANSFORM (gst_base_transform_get_type())
#define GST_BASE_TRANSFORM(o) ((GstBaseTransform *)(o))
#define GST_BASE_TRANSFORM_CLASS(k) ((GstBaseTransformClass *)(k))
#define GST_BASE_TRANSFORM_GET_CLASS(o) ((GstBaseTransformClass *)(((GObject *)(o))->g_class))
#define GST_BASE_TRANSFORM_SINK_PAD(o) (GST_BASE_TRANSFORM(o)->sinkpad)
#define GST_BASE_TRANSFORM_SRC_PAD(o) (GST_BASE_TRANSFORM(o)->srcpad)
void gst_base_transform_set_passthrough(GstBaseTransform *trans, gboolean passthrough); gboolean gst_base_transform_is_passthrough(GstBaseTransform *trans);
void gst_base_transform_reconfigure_src(GstBaseTransform *trans);
typedef struct _GstAdapter GstAdapter;
GstAdapter *gst_adapter_new(void); void gst_adapter_clear(GstAdapter *adapter); void gst_adapter_push(GstAdapter *adapter, GstBuffer *buf);
gconstpointer gst_adapter_map(GstAdapter *adapter, gsize size); void gst_adapter_unmap(GstAdapter *adapter); void gst_adapter_flush(GstAdapter *adapter, gsize flush); gsize gst_adapter_available(GstAdapter *adapter);
typedef struct _GstAggregatorPad { GstPad parent; GstSegment segment; } GstAggregatorPad;
typedef struct _GstAggregator { GstElement parent; GstPad *srcpad; } GstAggregator;
typedef struct _GstAggregatorClass {
  GstElementClass parent_class;
  GstFlowReturn (*flush)(GstAggregator *aggregator);
  GstBuffer *(*clip)(GstAggregator *aggregator, GstAggregatorPad *aggregator_pad, GstBuffer *buf);
  GstFlowReturn (*finish_buffer)(GstAggregator *aggregator, GstBuffer *buffer);
  gboolean (*sink_event)(GstAggregator *aggregator, GstAggregatorPad *aggregator_pad, GstEvent *event);
  gboolean (*sink_query)(GstAggregator *aggregator, GstAggregatorPad *aggregator_pad, GstQuery *query);
  gboolean (*src_event)(GstAggregator *aggregator, GstEvent *event);
  gboolean (*src_query)(GstAggregator *aggregator, GstQuery *query);
  gboolean (*src_activate)(GstAggregator *aggregator, gint mode, gboolean active);
  GstFlowReturn (*aggregate)(GstAggregator *aggregator, gboolean timeout);
  gboolean (*stop)(GstAggregator *aggregator);
  gboolean (*start)(GstAggregator *aggregator);
  GstClockTime (*get_next_time)(GstAggregator *aggregator);
  GstAggregatorPad *(*create_new_pad)(GstAggregator *self, GstPadTemplate *templ, const gchar *req_name, const GstCaps *caps);
  GstFlowReturn (*update_src_caps)(GstAggregator *self, GstCaps *caps, GstCaps **ret);
  GstCaps *(*fixate_src_caps)(GstAggregator *self, GstCaps *caps);
  gboolean (*negotiated_src_caps)(GstAggregator *self, GstCaps *caps);
} GstAggregatorClass;
GType gst_aggregator_pad_get_type(void);
#define GST_TYPE_AGGREGATOR_PAD (gst_aggregator_pad_get_type())
#define GST_AGGREGATOR_PAD(o) ((GstAggregatorPad *)(o))
#define GST_AGGREGATOR_CLASS(k) ((GstAggregatorClass *)(k))
gboolean gst_aggregator_pad_is_eos(GstAggregatorPad *pad);

/* ---------------------------------------------------------------- gst-video */
typedef enum {
  GST_VIDEO_FORMAT_UNKNOWN, GST_VIDEO_FORMAT_ENCODED, GST_VIDEO_FORMAT_I420, GST_VIDEO_FORMAT_YV12, GST_VIDEO_FORMAT_YUY2, GST_VIDEO_FORMAT_UYVY, GST_VIDEO_FORMAT_AYUV,
  GST_VIDEO_FORMAT_RGBx, GST_VIDEO_FORMAT_BGRx, GST_VIDEO_FORMAT_xRGB, GST_VIDEO_FORMAT_xBGR, GST_VIDEO_FORMAT_RGBA, GST_VIDEO_FORMAT_BGRA, GST_VIDEO_FORMAT_ARGB, GST_VIDEO_FORMAT_ABGR,
  GST_VIDEO_FORMAT_RGB, GST_VIDEO_FORMAT_BGR, GST_VIDEO_FORMAT_A420 = 34, GST_VIDEO_FORMAT_RGBA64_LE = 118, GST_VIDEO_FORMAT_RGBA64_BE = 119
} GstVideoFormat;
typedef enum { GST_VIDEO_FRAME_FLAG_NONE = 0 } GstVideoFrameFlags;
#define GST_VIDEO_MAX_PLANES 4
typedef struct { const void *finfo; gint interlace_mode; guint flags; gint width, height; gsize size; gint views; gint par_n, par_d, fps_n, fps_d; gsize offset[GST_VIDEO_MAX_PLANES]; gint stride[GST_VIDEO_MAX_PLANES]; GstVideoFormat format_stub; guint n_planes_stub; } GstVideoInfo;
typedef struct { GstVideoInfo info; GstVideoFrameFlags flags; GstBuffer *buffer; gpointer meta; gint id; gpointer data[GST_VIDEO_MAX_PLANES]; GstMapInfo map[GST_VIDEO_MAX_PLANES]; } GstVideoFrame;
typedef struct { GstMeta meta; GstBuffer *buffer; GstVideoFrameFlags flags; GstVideoFormat format; gint id; guint width, height; guint n_planes; gsize offset[GST_VIDEO_MAX_PLANES]; gint stride[GST_VIDEO_MAX_PLANES]; } GstVideoMeta;
#define GST_VIDEO_INFO_FORMAT(i) ((i)->format_stub)
#define GST_VIDEO_INFO_WIDTH(i) ((i)->width)
#define GST_VIDEO_INFO_HEIGHT(i) ((i)->height)
#define GST_VIDEO_INFO_SIZE(i) ((i)->size)
#define GST_VIDEO_INFO_FPS_N(i) ((i)->fps_n)
#define GST_VIDEO_INFO_FPS_D(i) ((i)->fps_d)
#define GST_VIDEO_INFO_N_PLANES(i) ((i)->n_planes_stub)
#define GST_VIDEO_INFO_PLANE_OFFSET(i, p) ((i)->offset[p])
#define GST_VIDEO_INFO_PLANE_STRIDE(i, p) ((i)->stride[p])
#define GST_VIDEO_FRAME_FORMAT(f) (GST_VIDEO_INFO_FORMAT(&(f)->info))
#define GST_VIDEO_FRAME_WIDTH(f) (GST_VIDEO_INFO_WIDTH(&(f)->info))
#define GST_VIDEO_FRAME_HEIGHT(f) (GST_VIDEO_INFO_HEIGHT(&(f)->info))
#define GST_VIDEO_FRAME_PLANE_DATA(f, p) ((f)->data[p])
#define GST_VIDEO_FRAME_PLANE_STRIDE(f, p) (GST_VIDEO_INFO_PLANE_STRIDE(&(f)->info, (p)))
#define GST_VIDEO_CAPS_MAKE(format) "video/x-raw, format = (string) " format ", width = (int) [ 1, max ], height = (int) [ 1, max ], framerate = (fraction) [ 0, max ]"
#define GST_BUFFER_POOL_OPTION_VIDEO_META "GstBufferPoolOptionVideoMeta"
GType gst_video_meta_api_get_type(void);
#define GST_VIDEO_META_API_TYPE (gst_video_meta_api_get_type())
gboolean gst_video_info_from_caps(GstVideoInfo *info, const GstCaps *caps);
gboolean gst_video_frame_map(GstVideoFrame *frame, const GstVideoInfo *info, GstBuffer *buffer, GstMapFlags flags); void gst_video_frame_unmap(GstVideoFrame *frame);
GstBufferPool *gst_video_buffer_pool_new(void);
GstVideoMeta *gst_buffer_get_video_meta(GstBuffer *buffer);
GstVideoMeta *gst_buffer_add_video_meta_full(GstBuffer *buffer, GstVideoFrameFlags flags, GstVideoFormat format, guint width, guint height, guint n_planes,
                                             const gsize offset[GST_VIDEO_MAX_PLANES], const gint stride[GST_VIDEO_MAX_PLANES]);
typedef struct _GstVideoFilter { GstBaseTransform element; gboolean negotiated; GstVideoInfo in_info, out_info; } GstVideoFilter;
typedef struct _GstVideoFilterClass {
  GstBaseTransformClass parent_class;
  gboolean (*set_info)(GstVideoFilter *filter, GstCaps *incaps, GstVideoInfo *in_info, GstCaps *outcaps, GstVideoInfo *out_info);
  GstFlowReturn (*transform_frame)(GstVideoFilter *filter, GstVideoFrame *inframe, GstVideoFrame *outframe);
  GstFlowReturn (*transform_frame_ip)(GstVideoFilter *trans, GstVideoFrame *frame);
} GstVideoFilterClass;
GType gst_video_filter_get_type(void);
#define GST_TYPE_VIDEO_FILTER (gst_video_filter_get_type())
#define GST_VIDEO_FILTER(o) ((GstVideoFilter *)(o))
#define GST_VIDEO_FILTER_CLASS(k) ((GstVideoFilterClass *)(k))
typedef struct _GstVideoAggregatorPad { GstAggregatorPad parent; GstVideoInfo info; } GstVideoAggregatorPad;
typedef struct _GstVideoAggregator { GstAggregator aggregator; GstVideoInfo info; } GstVideoAggregator;
typedef struct _GstVideoAggregatorClass {
  GstAggregatorClass parent_class;
  GstCaps *(*update_caps)(GstVideoAggregator *videoaggregator, GstCaps *caps);
  GstFlowReturn (*aggregate_frames)(GstVideoAggregator *videoaggregator, GstBuffer *outbuffer);
  GstFlowReturn (*create_output_buffer)(GstVideoAggregator *videoaggregator, GstBuffer **outbuffer);
} GstVideoAggregatorClass;
GType gst_video_aggregator_get_type(void); GType gst_video_aggregator_pad_get_type(void);
#define GST_TYPE_VIDEO_AGGREGATOR (gst_video_aggregator_get_type())
#define GST_TYPE_VIDEO_AGGREGATOR_PAD (gst_video_aggregator_pad_get_type())
#define GST_VIDEO_AGGREGATOR_PAD(o) ((GstVideoAggregatorPad *)(o))
#define GST_VIDEO_AGGREGATOR_CLASS(k) ((GstVideoAggregatorClass *)(k))
GstVideoFrame *gst_video_aggregator_pad_get_prepared_frame(GstVideoAggregatorPad *pad);

/* ---------------------------------------------------------------- gst-audio */
typedef enum { GST_AUDIO_FORMAT_UNKNOWN, GST_AUDIO_FORMAT_ENCODED, GST_AUDIO_FORMAT_S8, GST_AUDIO_FORMAT_U8, GST_AUDIO_FORMAT_S16LE, GST_AUDIO_FORMAT_S32LE = 10, GST_AUDIO_FORMAT_F32LE = 28, GST_AUDIO_FORMAT_F64LE = 30 } GstAudioFormat;
#define GST_AUDIO_FORMAT_S16 GST_AUDIO_FORMAT_S16LE
#define GST_AUDIO_FORMAT_S32 GST_AUDIO_FORMAT_S32LE
#define GST_AUDIO_FORMAT_F32 GST_AUDIO_FORMAT_F32LE
#define GST_AUDIO_FORMAT_F64 GST_AUDIO_FORMAT_F64LE
#define GST_AUDIO_NE(s) #s "LE"
typedef enum { GST_AUDIO_LAYOUT_INTERLEAVED = 0, GST_AUDIO_LAYOUT_NON_INTERLEAVED } GstAudioLayout;
typedef enum {
  GST_AUDIO_CHANNEL_POSITION_NONE = -3, GST_AUDIO_CHANNEL_POSITION_MONO, GST_AUDIO_CHANNEL_POSITION_INVALID,
  GST_AUDIO_CHANNEL_POSITION_FRONT_LEFT = 0, GST_AUDIO_CHANNEL_POSITION_FRONT_RIGHT, GST_AUDIO_CHANNEL_POSITION_FRONT_CENTER, GST_AUDIO_CHANNEL_POSITION_LFE1,
  GST_AUDIO_CHANNEL_POSITION_REAR_LEFT, GST_AUDIO_CHANNEL_POSITION_REAR_RIGHT, GST_AUDIO_CHANNEL_POSITION_FRONT_LEFT_OF_CENTER, GST_AUDIO_CHANNEL_POSITION_FRONT_RIGHT_OF_CENTER,
  GST_AUDIO_CHANNEL_POSITION_REAR_CENTER, GST_AUDIO_CHANNEL_POSITION_LFE2, GST_AUDIO_CHANNEL_POSITION_SIDE_LEFT, GST_AUDIO_CHANNEL_POSITION_SIDE_RIGHT,
  GST_AUDIO_CHANNEL_POSITION_TOP_FRONT_LEFT, GST_AUDIO_CHANNEL_POSITION_TOP_FRONT_RIGHT, GST_AUDIO_CHANNEL_POSITION_TOP_FRONT_CENTER, GST_AUDIO_CHANNEL_POSITION_TOP_CENTER,
  GST_AUDIO_CHANNEL_POSITION_TOP_REAR_LEFT, GST_AUDIO_CHANNEL_POSITION_TOP_REAR_RIGHT, GST_AUDIO_CHANNEL_POSITION_TOP_SIDE_LEFT, GST_AUDIO_CHANNEL_POSITION_TOP_SIDE_RIGHT,
  GST_AUDIO_CHANNEL_POSITION_TOP_REAR_CENTER, GST_AUDIO_CHANNEL_POSITION_BOTTOM_FRONT_CENTER, GST_AUDIO_CHANNEL_POSITION_BOTTOM_FRONT_LEFT, GST_AUDIO_CHANNEL_POSITION_BOTTOM_FRONT_RIGHT,
  GST_AUDIO_CHANNEL_POSITION_WIDE_LEFT, GST_AUDIO_CHANNEL_POSITION_WIDE_RIGHT, GST_AUDIO_CHANNEL_POSITION_SURROUND_LEFT, GST_AUDIO_CHANNEL_POSITION_SURROUND_RIGHT
} GstAudioChannelPosition;
typedef struct { const void *finfo; guint flags; GstAudioLayout layout; gint rate, channels, bpf; GstAudioChannelPosition position[64]; GstAudioFormat format_stub; } GstAudioInfo;
#define GST_AUDIO_INFO_FORMAT(i) ((i)->format_stub)
#define GST_AUDIO_INFO_RATE(i) ((i)->rate)
#define GST_AUDIO_INFO_CHANNELS(i) ((i)->channels)
#define GST_AUDIO_INFO_BPF(i) ((i)->bpf)
#define GST_AUDIO_INFO_LAYOUT(i) ((i)->layout)
#define GST_AUDIO_INFO_POSITION(i, c) ((i)->position[c])
#define GST_AUDIO_INFO_IS_UNPOSITIONED(i) (((i)->flags & 1u) != 0)
typedef struct { GstAudioInfo info; gsize n_samples; gint n_planes; gpointer *planes; GstBuffer *buffer; } GstAudioBuffer;
gboolean gst_audio_info_from_caps(GstAudioInfo *info, const GstCaps *caps);
gboolean gst_audio_buffer_map(GstAudioBuffer *buffer, const GstAudioInfo *info, GstBuffer *gstbuffer, GstMapFlags flags); void gst_audio_buffer_unmap(GstAudioBuffer *buffer);
typedef struct _GstAudioFilter { GstBaseTransform basetransform; GstAudioInfo info; } GstAudioFilter;
typedef struct _GstAudioFilterClass { GstBaseTransformClass basetransformclass; gboolean (*setup)(GstAudioFilter *filter, const GstAudioInfo *info); } GstAudioFilterClass;
GType gst_audio_filter_get_type(void);
#define GST_TYPE_AUDIO_FILTER (gst_audio_filter_get_type())
#define GST_AUDIO_FILTER_CLASS(k) ((GstAudioFilterClass *)(k))
void gst_audio_filter_class_add_pad_templates(GstAudioFilterClass *klass, GstCaps *allowed_caps);
#endif
