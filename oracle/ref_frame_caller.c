/* TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * Caller of the reference's OWN frame-level functions, compiled where they lie:
 *
 *   gst_bayer2rgb_process    /root/reference/gst/bayer/gstbayer2rgb.c:387-451  (+ its row helper :354-381, the
 *                            enum :95-101 and struct _GstBayer2RGB :115-127 it needs)
 *   gst_rgb2bayer_transform  /root/reference/gst/bayer/gstrgb2bayer.c:229-278  (+ its debug category :31-32 and the
 *                            reference's real gstrgb2bayer.h)
 *
 * Both are `static` inside translation units that also hold the element registration, which needs GStreamer >= 1.20
 * macros this image's 1.14 headers lack.  `oracle/Makefile: ref_frame` therefore extracts exactly those line ranges
 * AT BUILD TIME into a temporary directory (`sed -n`, nothing is copied into the repository and the temporary files
 * are removed), and this file #includes them between the image's real GStreamer / GLib headers and the entry points
 * below.  No compat header, no macro or type stand-in: everything the extracted lines reference is declared by
 * GStreamer 1.14, GLib, the reference's own gstbayerorc-dist.h / gstrgb2bayer.h, or the extracted lines themselves.
 * -DG_DISABLE_CAST_CHECKS (a GLib build option) makes GST_RGB_2_BAYER() a plain cast, so no GType is registered.
 *
 * Only the three functions at the bottom are this repository's own code.  The result,
 * oracle/_ref/libbayer_frame_ref.so, is git-ignored and travels to the GPU box as a binary (checker only).
 */
#include <stdint.h>
#include <string.h>

#include <gst/gst.h>
#include <gst/base/gstbasetransform.h>
#include <gst/video/video.h>

#include "gstbayerorc-dist.h"           /* the reference's header, -I$(REF)/gst/bayer */
#include "gstrgb2bayer.h"               /* the reference's header */

#include "ref_bayer2rgb_lines.inc"      /* extracted at build time: gstbayer2rgb.c:95-127,354-451 */
#include "ref_rgb2bayer_lines.inc"      /* extracted at build time: gstrgb2bayer.c:31-32,229-278 */

static void
ref_frame_init (void)
{
  static gsize once = 0;
  if (g_once_init_enter (&once)) {
    if (!gst_is_initialized ()) {
      gst_registry_fork_set_enabled (FALSE);
      gst_init (NULL, NULL);
    }
    GST_DEBUG_CATEGORY_INIT (gst_rgb2bayer_debug, "rgb2bayer_ref", 0, "reference rgb2bayer (oracle pin)");
    g_once_init_leave (&once, 1);
  }
}

/* One frame through the reference's gst_bayer2rgb_process.  `format` is the reference's enum
 * (gstbayer2rgb.c:95-101: 0 bggr, 1 gbrg, 2 grbg, 3 rggb); offsets as set_caps derives them (:253-256). */
int
ref_frame_bayer2rgb (uint8_t * dst, int dst_stride, const uint8_t * src, int src_stride, int width, int height,
    int format, int r_off, int g_off, int b_off)
{
  GstBayer2RGB f;

  if (width < 4 || (width & 1) || height < 3)
    return -1;                  /* outside the domain where the reference is defined */
  memset (&f, 0, sizeof (f));
  f.width = width;
  f.height = height;
  f.format = format;
  f.r_off = r_off;
  f.g_off = g_off;
  f.b_off = b_off;
  gst_bayer2rgb_process (&f, dst, dst_stride, (uint8_t *) src, src_stride);
  return 0;
}

/* One ARGB frame through the reference's gst_rgb2bayer_transform.  The destination row pitch is the reference's
 * GST_ROUND_UP_4 (width) (gstrgb2bayer.c:255); a source pitch other than 4 * width travels as a GstVideoMeta, the way
 * an upstream element would announce it. */
int
ref_frame_rgb2bayer (uint8_t * dst, const uint8_t * src, int src_stride, int width, int height, int format)
{
  GstRGB2Bayer f;
  GstBuffer *in, *out;
  GstFlowReturn ret;
  gsize in_size = (gsize) src_stride * height, out_size = (gsize) GST_ROUND_UP_4 (width) * height;

  if (width < 1 || height < 1 || src_stride < 4 * width)
    return -1;
  ref_frame_init ();
  memset (&f, 0, sizeof (f));
  f.width = width;
  f.height = height;
  f.format = format;
  gst_video_info_init (&f.info);
  if (!gst_video_info_set_format (&f.info, GST_VIDEO_FORMAT_ARGB, width, height))
    return -2;
  in = gst_buffer_new_wrapped_full (GST_MEMORY_FLAG_READONLY, (gpointer) src, in_size, 0, in_size, NULL, NULL);
  if (src_stride != 4 * width) {
    gsize offset[GST_VIDEO_MAX_PLANES] = { 0, };
    gint stride[GST_VIDEO_MAX_PLANES] = { src_stride, };
    gst_buffer_add_video_meta_full (in, GST_VIDEO_FRAME_FLAG_NONE, GST_VIDEO_FORMAT_ARGB, width, height, 1, offset,
        stride);
  }
  out = gst_buffer_new_wrapped_full (0, dst, out_size, 0, out_size, NULL, NULL);
  ret = gst_rgb2bayer_transform ((GstBaseTransform *) & f, in, out);
  gst_buffer_unref (in);
  gst_buffer_unref (out);
  return ret == GST_FLOW_OK ? 0 : -3;
}

const char *
ref_frame_describe (void)
{
  return "gst_bayer2rgb_process (gstbayer2rgb.c:387-451) + gst_rgb2bayer_transform (gstrgb2bayer.c:229-278), "
      "gst-plugins-bad 1.19.2, -DDISABLE_ORC, lines extracted at build time";
}
