/* Development tool: which launch plan should ONE frame per launch run?  (the frame-class shape rule,
 * mibayer_frame_class_variant, was derived on sector-aligned widths; 3838x2160 runs at 40 % per frame where aligned 4K
 * runs at 54 %.)  For a geometry: every production shape x store policy (ids 1-9) x block order (the shape's default,
 * identity, band 1, one chunk per XCD) x store alignment (0 / 64 / 128), one frame per launch on the context's stream,
 * separately allocated frames, C caller; fastest first, with the context's default plan marked.
 *
 *   gcc -O2 -Wall -I include tools/csrc/frame_plan_sweep.c -o /tmp/frame_plan_sweep -Lgst-plugins-bad_amd -lmibayer \
 *       -Wl,-rpath,$PWD/gst-plugins-bad_amd
 *   /tmp/frame_plan_sweep 3838 2160
 */
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "mibayer.h"

#define MAXN 256

static double
now_us (void)
{
  struct timespec t;

  clock_gettime (CLOCK_MONOTONIC, &t);
  return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}

typedef struct
{
  int v, band, align;
  double us;
} arm;

static int
by_us (const void *a, const void *b)
{
  const double d = ((const arm *) a)->us - ((const arm *) b)->us;

  return d < 0 ? -1 : d > 0;
}

int
main (int argc, char **argv)
{
  const int w = argc > 2 ? atoi (argv[1]) : 3838, h = argc > 2 ? atoi (argv[2]) : 2160;
  const int reps = argc > 3 ? atoi (argv[3]) : 12;
  static const int bands[4] = { INT_MIN, 0, 1, -1 };
  static const int aligns[3] = { 0, 64, 128 };
  mibayer_cfg cfg;
  mibayer_ctx *ctx = NULL;
  void *src[MAXN], *dst[MAXN];
  arm arms[9 * 4 * 3];
  int narms = 0, n, i, v, bi, ai, r, dv = 0, db = 0, da = 0, dsrc = 0, round;
  size_t sb, dbytes;
  long long want;

  memset (&cfg, 0, sizeof cfg);
  cfg.struct_size = sizeof cfg;
  cfg.width = w;
  cfg.height = h;
  cfg.pattern = MIBAYER_RGGB;
  cfg.r_off = 2;
  cfg.g_off = 1;
  cfg.b_off = 0;
  cfg.device = 0;
  if (mibayer_create (&cfg, &ctx) != MIBAYER_OK) {
    fprintf (stderr, "create: %s\n", mibayer_last_hip_error ());
    return 1;
  }
  mibayer_get_cfg (ctx, &cfg);
  sb = (size_t) cfg.src_stride * h;
  dbytes = (size_t) cfg.dst_stride * h;
  want = 64LL * 3840 * 2160 / ((long long) w * h);
  n = want < 8 ? 8 : want > MAXN ? MAXN : (int) want;
  for (i = 0; i < n; i++) {
    src[i] = mibayer_device_alloc (ctx, sb);
    dst[i] = mibayer_device_alloc (ctx, dbytes);
    if (!src[i] || !dst[i])
      return 2;
    mibayer_fill_synthetic (ctx, src[i], 0, (uint32_t) i, 1, 2, mibayer_ctx_stream (ctx));
  }
  mibayer_sync (ctx);
  mibayer_get_plan_for (ctx, 1, &dv, &db, &da, &dsrc);
  printf ("# %dx%d, %d separately allocated frames per pass, ONE frame per launch on the context's stream, %d passes, best of 2 rounds\n",
      w, h, n, reps);
  printf ("# default plan of the frame class: %s band %d align %d\n", mibayer_variant_name (dv), db == INT_MIN ? -999 : db, da);
  for (v = 1; v <= 9; v++)
    for (bi = 0; bi < 4; bi++)
      for (ai = 0; ai < 3; ai++) {
        if (aligns[ai] && (cfg.dst_stride % 64 == 0 || cfg.dst_stride % 8 != 0))
          continue;             /* the shifted arm exists for rows off the sector grid only */
        arms[narms].v = v;
        arms[narms].band = bands[bi];
        arms[narms].align = aligns[ai];
        arms[narms].us = 1e30;
        narms++;
      }
  for (round = 0; round < 2; round++)
    for (i = 0; i < narms; i++) {
      double t0 = 0;
      int k;

      if (mibayer_set_plan_for (ctx, 1, arms[i].v, arms[i].band, arms[i].align) != MIBAYER_OK) {
        arms[i].us = 1e29;
        continue;
      }
      for (r = 0; r < reps + 2; r++) {
        if (r == 2) {
          mibayer_sync (ctx);
          t0 = now_us ();
        }
        for (k = 0; k < n; k++)
          if (mibayer_process_device (ctx, src[k], 0, dst[k], 0, 1, mibayer_ctx_stream (ctx)) != MIBAYER_OK)
            return 3;
      }
      mibayer_sync (ctx);
      {
        const double us = (now_us () - t0) / ((double) reps * n);

        if (us < arms[i].us)
          arms[i].us = us;
      }
    }
  qsort (arms, (size_t) narms, sizeof (arm), by_us);
  for (i = 0; i < narms; i++) {
    const int is_default = arms[i].v == dv && arms[i].band == db && arms[i].align == da;

    if (i < 14 || is_default)
      printf ("%s %-20s band %4d align %3d  %8.3f us per frame  %5.1f %% of 8 TB/s\n", is_default ? "DEFAULT" : "       ",
          mibayer_variant_name (arms[i].v), arms[i].band == INT_MIN ? -999 : arms[i].band, arms[i].align, arms[i].us,
          5.0 * w * h / arms[i].us / 1e3 / 80.0);
  }
  for (i = 0; i < n; i++) {
    mibayer_device_free (ctx, src[i]);
    mibayer_device_free (ctx, dst[i]);
  }
  mibayer_destroy (ctx);
  return 0;
}
