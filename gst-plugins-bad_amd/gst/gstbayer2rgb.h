/* MI355X-native bayer2rgb element: type declaration.
 * Mirrors the private declarations of reference gst/bayer/gstbayer2rgb.c:103-132. */
#ifndef MI_GST_BAYER2RGB_H
#define MI_GST_BAYER2RGB_H

#include <gst/gst.h>
#include <gst/base/gstbasetransform.h>
#include <gst/video/video.h>

#include "mibayer.h"

G_BEGIN_DECLS

#define GST_TYPE_BAYER2RGB (gst_bayer2rgb_get_type ())
#define GST_BAYER2RGB(obj) \
  (G_TYPE_CHECK_INSTANCE_CAST ((obj), GST_TYPE_BAYER2RGB, GstBayer2RGB))
#define GST_IS_BAYER2RGB(obj) \
  (G_TYPE_CHECK_INSTANCE_TYPE ((obj), GST_TYPE_BAYER2RGB))

typedef struct _GstBayer2RGB GstBayer2RGB;
typedef struct _GstBayer2RGBClass GstBayer2RGBClass;

/* Same negotiated state as the reference's struct _GstBayer2RGB
 * (gstbayer2rgb.c:115-127) plus the handle of the GPU context that replaces
 * gst_bayer2rgb_process. */
struct _GstBayer2RGB
{
  GstBaseTransform basetransform;

  GstVideoInfo info;            /* output video info */
  gint width;
  gint height;
  gint r_off;                   /* byte offset of red in an output pixel */
  gint g_off;
  gint b_off;
  gint format;                  /* mibayer_pattern == reference enum :95-101 */

  /* additive, optional properties (the reference has none); the defaults give
   * the reference's behaviour: one device, strictly 1-in/1-out synchronous */
  gint device_id;
  gchar *devices;               /* "0,1,2,..." round-robin frame sharding; NULL = device-id */
  gint inflight;                /* frames in flight per device; 1 = synchronous */
  gboolean use_hipgraph;
  gboolean pinned_pool;

  /* GPU side: one shard (mibayer_ctx) per device behind a round-robin pool;
   * (re)created when caps or the mapped output stride change */
  mibayer_pool *pool;
  gint pool_dst_stride;
  gint capacity;                /* frames the pool may hold in flight */
  GQueue pending;               /* Bayer2RGBPending*, oldest first */
};

struct _GstBayer2RGBClass
{
  GstBaseTransformClass parent;
};

GType gst_bayer2rgb_get_type (void);
gboolean gst_bayer2rgb_register (GstPlugin * plugin);

G_END_DECLS
#endif
