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
#ifdef WITH_CU_MASK
#include <hip/hip_runtime_api.h>
#endif

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
  void *src[N], *dst[N], *q[1 + MIBAYER_FRAME_QUEUES];
#define MAXSCH 16
#define MAXQ 16
  void *mq[MAXSCH][MAXQ];       /* CU-masked streams: [scheme][k] */
  int nmq[MAXSCH] = { 0 }, nsch = 0;
  char sch_name[MAXSCH][16];
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
  cfg.variant = getenv ("FLB_VARIANT") ? atoi (getenv ("FLB_VARIANT")) : 0;
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
  for (i = 0; i < MIBAYER_FRAME_QUEUES; i++)
    q[1 + i] = mibayer_ctx_frame_queue (ctx, i);
  printf ("# %dx%d %s, %d frames per pass, one launch per frame, C caller, %d passes\n", w, h,
      inverse ? "rgb2bayer" : "bayer2rgb", N, REPS);
#ifdef WITH_CU_MASK
  {
    /* experiment: each stream owns a share of the CUs, so that several one-frame kernels are resident side by side and
     * run out of phase (one in its store burst while another loads) instead of one after the other.
     * FLB_SCHEMES = comma list of mod:K (stream k owns the CUs with index % K == k), blk:K (index / (256/K) == k),
     * all:K (K streams that each own every CU: the masked-stream API without a partition) */
    const char *e = getenv ("FLB_SCHEMES") ? getenv ("FLB_SCHEMES") : "mod:2,mod:4,mod:8,blk:2,blk:4,blk:8,all:4";
    char buf[256];
    char *tok;
    int k, b;
    snprintf (buf, sizeof buf, "%s", e);
    for (tok = strtok (buf, ","); tok && nsch < MAXSCH; tok = strtok (NULL, ",")) {
      const int parts = atoi (tok + 4);
      if (parts < 1 || parts > MAXQ)
        continue;
      snprintf (sch_name[nsch], sizeof sch_name[nsch], "%s", tok);
      for (k = 0; k < parts; k++) {
        uint32_t mask[8] = { 0 };
        hipStream_t st = NULL;
        for (b = 0; b < 256; b++) {
          const int mine = tok[0] == 'm' ? (b % parts == k) : tok[0] == 'b' ? (b / (256 / parts) == k) : 1;
          if (mine)
            mask[b / 32] |= 1u << (b % 32);
        }
        if (hipExtStreamCreateWithCUMask (&st, 8, mask) != hipSuccess) {
          fprintf (stderr, "hipExtStreamCreateWithCUMask failed (%s)\n", tok);
          break;
        }
        mq[nsch][nmq[nsch]++] = st;
      }
      nsch++;
    }
  }
#endif
  for (mode = 0; mode <= 2 + nsch; mode++) {
    double t0 = 0, t1, issue = 0;
    void **qs = mode == 0 ? q : q + 1;          /* 0: the context's stream; 1: one frame queue; 2: all of them */
    int nq = mode <= 1 ? 1 : MIBAYER_FRAME_QUEUES;
    if (mode > 2) {
      if (nmq[mode - 3] == 0)
        continue;
      qs = mq[mode - 3];
      nq = nmq[mode - 3];
    }
    for (r = 0; r < REPS + 3; r++) {
      double a;
      if (r == 3) {
        mibayer_sync (ctx);
#ifdef WITH_CU_MASK
        if (mode > 2)
          for (i = 0; i < nq; i++)
            (void) hipStreamSynchronize ((hipStream_t) qs[i]);
#endif
        t0 = now_us ();
        issue = 0;
      }
      a = now_us ();
      for (i = 0; i < N; i++)
        if (mibayer_process_device (ctx, src[i], 0, dst[i], 0, 1, qs[i % nq]) != MIBAYER_OK)
          return 3;
      issue += now_us () - a;
    }
    mibayer_sync (ctx);
#ifdef WITH_CU_MASK
    if (mode > 2)
      for (i = 0; i < nq; i++)
        (void) hipStreamSynchronize ((hipStream_t) qs[i]);
#endif
    t1 = now_us ();
    if (mode > 2)
      printf ("CU mask %-6s ", sch_name[mode - 3]);
    if (mode <= 2)
      printf ("%-14s ", mode == 0 ? "ctx stream" : "frame queues");
    printf ("%d queue(s): %8.3f us per frame  %6.1f %% of 8 TB/s   host issue %6.3f us per launch\n", nq,
        (t1 - t0) / (REPS * N), 5.0 * w * h / ((t1 - t0) / (REPS * N)) / 1e3 / 80, issue / (REPS * N));
  }
  mibayer_destroy (ctx);
  return 0;
}
