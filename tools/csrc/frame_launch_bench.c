/* Development tool: what a C caller (an element) gets from one launch per frame -- the python harness of
 * tools/list_launch_bench.py spends several microseconds per call in ctypes, which is the same order as a 4K kernel.
 * 64 device-resident frames, each its own allocation; wall time per pass, issue time per launch (host side).
 *   gcc -O2 -I include tools/csrc/frame_launch_bench.c -o /tmp/frame_launch_bench -Lgst-plugins-bad_amd -lmibayer \
 *       -Wl,-rpath,$PWD/gst-plugins-bad_amd
 *   /tmp/frame_launch_bench [width height [inverse]] */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "mibayer.h"

static double now_us (void)
{
  struct timespec t;
  clock_gettime (CLOCK_MONOTONIC, &t);
  return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}

#define N 64
#define REPS 30

int main (int argc, char **argv)
{
  const int w = argc > 2 ? atoi (argv[1]) : 3840, h = argc > 2 ? atoi (argv[2]) : 2160;
  const int inverse = argc > 3;
  mibayer_cfg cfg;
  mibayer_ctx *ctx = NULL;
  void *src[N], *dst[N], *q[2];
  size_t sb, db;
  int i, r, mode;

  memset (&cfg, 0, sizeof cfg);
  cfg.struct_size = sizeof cfg;
  cfg.width = w;
  cfg.height = h;
  cfg.pattern = MIBAYER_RGGB;
  cfg.r_off = inverse ? 1 : 2;
  cfg.g_off = inverse ? 2 : 1;
  cfg.b_off = inverse ? 3 : 0;
  cfg.device = 0;
  cfg.flags = inverse ? MIBAYER_FLAG_RGB2BAYER : 0;
  if (mibayer_create (&cfg, &ctx) != MIBAYER_OK) {
    fprintf (stderr, "create: %s\n", mibayer_last_hip_error ());
    return 1;
  }
  mibayer_get_cfg (ctx, &cfg);
  sb = (size_t) cfg.src_stride * h;
  db = (size_t) cfg.dst_stride * h;
  for (i = 0; i < N; i++) {
    src[i] = mibayer_device_alloc (ctx, sb);
    dst[i] = mibayer_device_alloc (ctx, db);
    if (!src[i] || !dst[i])
      return 2;
  }
  q[0] = mibayer_ctx_stream (ctx);
  q[1] = mibayer_ctx_stream2 (ctx);
  printf ("# %dx%d %s, %d frames per pass, one launch per frame, C caller, %d passes\n", w, h,
      inverse ? "rgb2bayer" : "bayer2rgb", N, REPS);
  for (mode = 1; mode <= 2; mode++) {
    double t0 = 0, t1, issue = 0;
    for (r = 0; r < REPS + 3; r++) {
      double a;
      if (r == 3) {
        mibayer_sync (ctx);
        t0 = now_us ();
        issue = 0;
      }
      a = now_us ();
      for (i = 0; i < N; i++)
        if (mibayer_process_device (ctx, src[i], 0, dst[i], 0, 1, q[mode == 2 ? (i & 1) : 0]) != MIBAYER_OK)
          return 3;
      issue += now_us () - a;
    }
    mibayer_sync (ctx);
    t1 = now_us ();
    printf ("%d queue(s): %8.3f us per frame  %6.1f %% of 8 TB/s   host issue %6.3f us per launch\n", mode,
        (t1 - t0) / (REPS * N), 5.0 * w * h / ((t1 - t0) / (REPS * N)) / 1e3 / 80, issue / (REPS * N));
  }
  mibayer_destroy (ctx);
  return 0;
}
