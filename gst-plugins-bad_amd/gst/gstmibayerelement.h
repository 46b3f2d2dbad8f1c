/* Shared implementation of the two elements of plugin `bayer` (private header).
 *
 * `bayer2rgb` and `rgb2bayer` are mirror images of each other above the kernel:
 * the same caps transform with the pad roles swapped (reference
 * gstbayer2rgb.c:289-322 vs gstrgb2bayer.c:128-159), the same unit sizes
 * (:324-352 vs :161-188), one 8-bit mosaic buffer and one 4-byte-per-pixel
 * video frame per conversion.  Both GTypes derive directly from
 * GstBaseTransform, as in the reference, and share this instance/class layout;
 * the class carries the direction.
 */
#ifndef MI_GST_BAYER_ELEMENT_H
#define MI_GST_BAYER_ELEMENT_H

#include <gst/gst.h>
#include <gst/base/gstbasetransform.h>
#include <gst/video/video.h>

#include "mibayer.h"

G_BEGIN_DECLS

typedef struct _GstMiBayerElement GstMiBayerElement;
typedef struct _GstMiBayerElementClass GstMiBayerElementClass;

/* Negotiated state = the reference's struct _GstBayer2RGB (gstbayer2rgb.c:115-127)
 * / struct _GstRGB2Bayer (gstrgb2bayer.h:43-52), plus the GPU pool that replaces
 * the CPU frame loops. */
struct _GstMiBayerElement
{
  GstBaseTransform basetransform;

  GstVideoInfo info;            /* the video/x-raw side: output of bayer2rgb, input of rgb2bayer */
  gint width;
  gint height;
  gint r_off;                   /* byte offset of red inside a 4-byte pixel */
  gint g_off;
  gint b_off;
  gint format;                  /* mibayer_pattern == reference enum (gstbayer2rgb.c:95-101) */

  /* additive, optional properties (the reference has none); the defaults give
   * the reference's behaviour: one device, strictly 1-in/1-out synchronous.
   * Written under the object lock by set_property at any time ... */
  gint device_id;
  gchar *devices;               /* "0,1,2,..." round-robin frame sharding; NULL = device-id */
  gint inflight;                /* frames in flight per device; 1 = synchronous */
  gboolean use_hipgraph;
  gboolean pinned_pool;
  gint timeout_ms;              /* deadline of every wait for a GPU; 0 = none */
  /* ... and latched into these by start(): the streaming thread only ever reads
   * the latched copies, so a property changed while PLAYING takes effect at the
   * next READY -> PAUSED and never races with the data flow */
  struct
  {
    gint device_id;
    gchar *devices;
    gint inflight;
    gboolean use_hipgraph;
    gboolean pinned_pool;
    gint timeout_ms;
  } act;

  /* GPU side: one shard (mibayer_ctx) per device behind a round-robin pool;
   * (re)created when caps or the mapped video-frame stride change */
  mibayer_pool *pool;
  gint pool_stride;
  gint capacity;                /* frames the pool may hold in flight (shrinks when a device is dropped) */
  GQueue pending;               /* PendingFrame*, oldest first */
  GQueue ready;                 /* GstBuffer*: finished outputs collected early (the pool shrank), oldest first */
  GQueue quarantine;            /* PendingFrame*: frames lost on a GPU that ran into the wait deadline; both buffers
                                   stay mapped and referenced until the pool says the device has let go of them */
  guint frames_lost;            /* since start */
  GMutex flow_lock;             /* pool / pending / ready: the streaming thread vs. FLUSH_START, which
                                   arrives on another thread */
  volatile gint flushing;       /* between FLUSH_START and FLUSH_STOP: nothing is submitted or pushed */
  gboolean prerolled;           /* a frame has left since start / flush; the first one is never held back */
  gchar *failure_note;          /* a device was dropped: posted as ONE element warning (flow_lock) */
  /* an error found while flow_lock was held: posted once the lock is released (a bus sync handler runs
   * application code, which may call back into the element) */
  GQuark error_domain;
  gint error_code;
  gchar *error_text;
  gchar *error_debug;
};

struct _GstMiBayerElementClass
{
  GstBaseTransformClass parent;

  GstBaseTransformClass *base_class;    /* GstBaseTransform's vfuncs, to chain up */
  gboolean inverse;             /* FALSE: bayer2rgb, TRUE: rgb2bayer */
  const gchar *label;           /* element name used in messages */
  GstDebugCategory *cat;        /* debug category of that name */
};

/* to be called from the concrete class_init / instance init */
void gst_mi_bayer_element_class_setup (GstMiBayerElementClass * klass,
    gboolean inverse, const gchar * label);
void gst_mi_bayer_element_instance_setup (GstMiBayerElement * self);

G_END_DECLS
#endif
