/* A plain C99 consumer of include/mibayer.h, built with -std=c99 -pedantic -Wall -Wextra -Werror by
 * tests/test_abi_c.py: proves the boundary is a C ABI (no C++ / GLib / torch types) and exercises it the way a
 * GStreamer element written in C would.  With a GPU it converts the hand-checkable 4x4 frame of SURVEY.md
 * Appendix B.4 and compares with the known answer; without one it must be refused with MIBAYER_ERR_NO_DEVICE. */
#include <stdio.h>
#include <string.h>

#include "mibayer.h"

static const uint8_t frame[16] = {
  10, 200, 30, 180,
  90, 250, 70, 5,
  50, 120, 255, 0,
  33, 77, 141, 222
};

/* SURVEY.md Appendix B.4, bggr -> RGBx (R,G,B per pixel; x = 255) */
static const uint8_t want_rgb[16][3] = {
  {250, 145, 10}, {250, 200, 20}, {250, 135, 30}, {5, 180, 30},
  {250, 90, 30}, {250, 120, 87}, {250, 70, 143}, {5, 80, 143},
  {164, 91, 50}, {164, 120, 153}, {164, 113, 255}, {114, 0, 255},
  {77, 33, 30}, {77, 124, 87}, {77, 141, 143}, {222, 116, 143}
};

int
main (void)
{
  mibayer_cfg cfg;
  mibayer_ctx *ctx = NULL;
  uint8_t out[64];
  int rc, i;

  if (mibayer_abi_version () != MIBAYER_ABI_VERSION) {
    fprintf (stderr, "ABI version mismatch\n");
    return 2;
  }
  memset (&cfg, 0, sizeof cfg);
  cfg.struct_size = sizeof cfg;
  cfg.width = 4;
  cfg.height = 4;
  cfg.pattern = MIBAYER_BGGR;
  cfg.r_off = 0;
  cfg.g_off = 1;
  cfg.b_off = 2;
  cfg.device = 0;
  rc = mibayer_create (&cfg, &ctx);
  if (mibayer_device_count () == 0) {
    printf ("no device: mibayer_create -> %d (%s)\n", rc, mibayer_strerror (rc));
    return rc == MIBAYER_ERR_NO_DEVICE && ctx == NULL ? 0 : 3;
  }
  if (rc != MIBAYER_OK) {
    fprintf (stderr, "create: %s %s\n", mibayer_strerror (rc), mibayer_last_hip_error ());
    return 4;
  }
  {
    /* ABI v4 from C: the launch plan as a value, the host-wait policy and the host statistics */
    int variant = -1, band = 0, align = -1;
    mibayer_host_stats st;

    if (mibayer_is_lab_build () != 0 || mibayer_plan_source (ctx) != MIBAYER_PLAN_DEFAULT
        || mibayer_get_plan (ctx, &variant, &band, &align) != MIBAYER_OK || variant < 1 || align != 0
        || mibayer_set_plan (ctx, variant, 0, 0) != MIBAYER_OK || mibayer_plan_source (ctx) != MIBAYER_PLAN_SET
        || mibayer_set_wait_spin (ctx, 0) != MIBAYER_OK || mibayer_get_host_stats (ctx, &st) != MIBAYER_OK
        || st.submits != 0 || mibayer_wedged_contexts () != 0 || mibayer_deferred_frees () != 0) {
      fprintf (stderr, "ABI v4 calls: unexpected answer (variant %d band %d align %d)\n", variant, band, align);
      return 7;
    }
  }
  {
    /* ABI v5 from C: the plan of a launch class, the frame queues */
    int variant = -1, band = 0, align = -1, source = -1;

    if (mibayer_get_plan_for (ctx, 1, &variant, &band, &align, &source) != MIBAYER_OK || variant < 1
        || source != MIBAYER_PLAN_SET || mibayer_set_plan_for (ctx, 1, variant, band, align) != MIBAYER_OK
        || mibayer_get_plan_for (ctx, 0, NULL, NULL, NULL, NULL) != MIBAYER_ERR_ARG
        || mibayer_ctx_frame_queue (ctx, 0) == NULL || mibayer_ctx_frame_queue (ctx, 0) == mibayer_ctx_stream (ctx)
        || mibayer_ctx_frame_queue (ctx, MIBAYER_FRAME_QUEUES - 1) == mibayer_ctx_frame_queue (ctx, 0)
        || mibayer_ctx_frame_queue (ctx, 1) != mibayer_ctx_frame_queue (ctx, 1)
        || mibayer_ctx_frame_queue (ctx, MIBAYER_FRAME_QUEUES) != NULL || mibayer_ctx_frame_queue (ctx, -1) != NULL) {
      fprintf (stderr, "ABI v5 calls: unexpected answer (variant %d band %d align %d source %d)\n", variant, band,
          align, source);
      return 9;
    }
  }
  memset (out, 0, sizeof out);
  rc = mibayer_process_host (ctx, frame, out);
  {
    mibayer_host_stats st;

    if (mibayer_get_host_stats (ctx, &st) != MIBAYER_OK || st.submits != 1 || st.waits < 1 || st.polls < 1) {
      fprintf (stderr, "host stats after one frame: submits %lu waits %lu\n", (unsigned long) st.submits,
          (unsigned long) st.waits);
      return 8;
    }
  }
  mibayer_destroy (ctx);
  if (rc != MIBAYER_OK) {
    fprintf (stderr, "process: %s\n", mibayer_strerror (rc));
    return 5;
  }
  for (i = 0; i < 16; i++) {
    if (out[4 * i] != want_rgb[i][0] || out[4 * i + 1] != want_rgb[i][1]
        || out[4 * i + 2] != want_rgb[i][2] || out[4 * i + 3] != 255) {
      fprintf (stderr, "pixel %d: got %u %u %u %u\n", i, out[4 * i], out[4 * i + 1], out[4 * i + 2],
          out[4 * i + 3]);
      return 6;
    }
  }
  printf ("gpu: 4x4 known answer ok\n");
  return 0;
}
