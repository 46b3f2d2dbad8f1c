/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  See bayer2rgb_oracle.h.
 *
 * Plain-C restatement of the reference bayer2rgb frame path.  Every function
 * cites the reference lines it restates (paths relative to /root/reference).
 * The structure deliberately follows the reference (two horizontal lines per
 * source row kept in a ring of four row slots, one merge per output row) so
 * that the reference's edge behaviour -- in particular the bottom row pairing
 * with row H-4 -- falls out of the ring arithmetic instead of being
 * hard-coded; the independent closed form lives in oracle/bayer2rgb_np.py.
 */
#define _GNU_SOURCE
#include "bayer2rgb_oracle.h"

#include <dlfcn.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ORC opcode avgub, gstbayerorc-dist.c:225-226: (a + b + 1) >> 1 on uint8 */
static inline uint8_t
avgub (uint8_t a, uint8_t b)
{
  return (uint8_t) (((unsigned) a + (unsigned) b + 1u) >> 1);
}

/* ---- horizontal pass -------------------------------------------------- */

/* gst_bayer2rgb_split_and_upsample_horiz, gstbayer2rgb.c:354-381, with the
 * x86 inner loop bayer_orc_horiz_upsample_unaligned (gstbayerorc.orc:3-19;
 * C semantics gstbayerorc-dist.c:183-249) folded in.
 *   ev[x]: the colour sampled at even columns, at every x
 *   od[x]: the colour sampled at odd columns, at every x */
static void
own_row_lines (uint8_t *ev, uint8_t *od, const uint8_t *s, int w)
{
  int x, i, npairs;

  /* head, :360-363 */
  ev[0] = s[0];
  od[0] = s[1];
  ev[1] = avgub (s[0], s[2]);
  od[1] = s[1];

  /* body, :365-367: d0 = ev+2, d1 = od+2, src pointer s+1, (w-4)>>1 pairs.
   * orc:10-19 with t = s+1: c=t[2i] b=t[2i+1] e'=t[2i+2] d=t[2i+3]
   *   d0 pair = (b, avg(b,d));  d1 pair = (avg(c,e'), e')   */
  npairs = (w - 4) >> 1;
  for (i = 0; i < npairs; i++) {
    const uint8_t *t = s + 1 + 2 * i;
    ev[2 + 2 * i] = t[1];
    ev[3 + 2 * i] = avgub (t[1], t[3]);
    od[2 + 2 * i] = avgub (t[0], t[2]);
    od[3 + 2 * i] = t[2];
  }

  /* tail, :372-380 */
  for (x = w - 2; x < w; x++) {
    if ((x & 1) == 0) {
      ev[x] = s[x];
      od[x] = s[x - 1];
    } else {
      ev[x] = s[x - 1];
      od[x] = s[x];
    }
  }
}

/* ---- vertical merge ----------------------------------------------------- */

/* bayer_orc_merge_bg_* (gstbayerorc.orc:43-66 and the three layout siblings
 * :95-118, :147-170, :199-222): output row whose even columns carry the
 * "b"-named sample.  Lines are (ev,od) of the rows above, at and below.
 * pr/pg/pb = byte position of the "r"-named, green and "b"-named value in the
 * 4-byte output pixel; the 4th byte is 255 (orc:65 `mergebw ra, r, 255`). */
static void
own_merge_bg (uint8_t *d, const uint8_t *ev_u, const uint8_t *od_u,
    const uint8_t *ev_c, const uint8_t *od_c, const uint8_t *ev_d,
    const uint8_t *od_d, int npairs, int pr, int pg, int pb)
{
  int i, k, pa = 6 - pr - pg - pb;
  for (i = 0; i < npairs; i++) {
    for (k = 0; k < 2; k++) {
      int x = 2 * i + k;
      uint8_t r = avgub (od_u[x], od_d[x]);       /* orc:57 */
      uint8_t g = avgub (avgub (ev_u[x], ev_d[x]), od_c[x]);   /* orc:58-60 */
      if (k == 1)
        g = od_c[x];                /* orc:61-63: odd pixel keeps g1 */
      d[4 * x + pb] = ev_c[x];      /* orc:64 */
      d[4 * x + pg] = g;
      d[4 * x + pr] = r;
      d[4 * x + pa] = 255;
    }
  }
}

/* bayer_orc_merge_gr_* (gstbayerorc.orc:69-92, :121-144, :173-196, :225-248):
 * output row whose even columns are green. */
static void
own_merge_gr (uint8_t *d, const uint8_t *ev_u, const uint8_t *od_u,
    const uint8_t *ev_c, const uint8_t *od_c, const uint8_t *ev_d,
    const uint8_t *od_d, int npairs, int pr, int pg, int pb)
{
  int i, k, pa = 6 - pr - pg - pb;
  for (i = 0; i < npairs; i++) {
    for (k = 0; k < 2; k++) {
      int x = 2 * i + k;
      uint8_t b = avgub (ev_u[x], ev_d[x]);       /* orc:83 */
      uint8_t g = avgub (avgub (od_u[x], od_d[x]), ev_c[x]);   /* orc:84-86 */
      if (k == 0)
        g = ev_c[x];                /* orc:87-89: even pixel keeps g1 */
      d[4 * x + pb] = b;
      d[4 * x + pg] = g;
      d[4 * x + pr] = od_c[x];      /* orc:91 */
      d[4 * x + pa] = 255;
    }
  }
}

/* ---- reference row kernels loaded from oracle/_ref ------------------------ */

typedef void (*ref_upsample_fn) (uint8_t *, uint8_t *, const uint8_t *, int);
typedef void (*ref_merge_fn) (uint8_t *, const uint8_t *, const uint8_t *,
    const uint8_t *, const uint8_t *, const uint8_t *, const uint8_t *, int);

static struct
{
  void *handle;
  ref_upsample_fn upsample_unaligned;
  /* [layout][0=bg,1=gr]; layout order bgra, abgr, argb, rgba as in
   * gstbayer2rgb.c:409-421 */
  ref_merge_fn merge[4][2];
} g_ref;

int
oracle_load_ref_rows (const char *so_path)
{
  static const char *names[4][2] = {
    {"bayer_orc_merge_bg_bgra", "bayer_orc_merge_gr_bgra"},
    {"bayer_orc_merge_bg_abgr", "bayer_orc_merge_gr_abgr"},
    {"bayer_orc_merge_bg_argb", "bayer_orc_merge_gr_argb"},
    {"bayer_orc_merge_bg_rgba", "bayer_orc_merge_gr_rgba"},
  };
  int l, t;
  if (g_ref.handle)
    return 0;
  void *h = dlopen (so_path, RTLD_NOW | RTLD_LOCAL);
  if (!h)
    return -1;
  g_ref.upsample_unaligned =
      (ref_upsample_fn) dlsym (h, "bayer_orc_horiz_upsample_unaligned");
  if (!g_ref.upsample_unaligned)
    goto fail;
  for (l = 0; l < 4; l++)
    for (t = 0; t < 2; t++) {
      g_ref.merge[l][t] = (ref_merge_fn) dlsym (h, names[l][t]);
      if (!g_ref.merge[l][t])
        goto fail;
    }
  g_ref.handle = h;
  return 0;
fail:
  dlclose (h);
  return -1;
}

/* gstbayer2rgb.c:354-381 with the body delegated to the reference's kernel */
static void
ref_row_lines (uint8_t *ev, uint8_t *od, const uint8_t *s, int w)
{
  int x;
  ev[0] = s[0];
  od[0] = s[1];
  ev[1] = avgub (s[0], s[2]);
  od[1] = s[1];
  g_ref.upsample_unaligned (ev + 2, od + 2, s + 1, (w - 4) >> 1);
  for (x = w - 2; x < w; x++) {
    if ((x & 1) == 0) {
      ev[x] = s[x];
      od[x] = s[x - 1];
    } else {
      ev[x] = s[x - 1];
      od[x] = s[x];
    }
  }
}

/* ---- frame driver --------------------------------------------------------- */

/* layout index per gstbayer2rgb.c:409-421, or -1 */
static int
layout_of (int r, int g, int b)
{
  if (r == 2 && g == 1 && b == 0)
    return 0;                   /* bgra */
  if (r == 3 && g == 2 && b == 1)
    return 1;                   /* abgr */
  if (r == 1 && g == 2 && b == 3)
    return 2;                   /* argb */
  if (r == 0 && g == 1 && b == 2)
    return 3;                   /* rgba */
  return -1;
}

/* ---- SIMD row kernels (bayer2rgb_simd.c) ----------------------------------- */

void simd_sse2_horiz_upsample_unaligned (uint8_t *d0, uint8_t *d1,
    const uint8_t *s, int n);
void simd_avx2_horiz_upsample_unaligned (uint8_t *d0, uint8_t *d1,
    const uint8_t *s, int n);
void simd_merge (int isa, int layout, int type, uint8_t *d, const uint8_t *u0,
    const uint8_t *u1, const uint8_t *c0, const uint8_t *c1, const uint8_t *d0,
    const uint8_t *d1, int n);
int oracle_simd_best_isa (void);

/* gstbayer2rgb.c:354-381 with the body delegated to the SIMD kernel */
static void
simd_row_lines (int isa, uint8_t *ev, uint8_t *od, const uint8_t *s, int w)
{
  int x;
  ev[0] = s[0];
  od[0] = s[1];
  ev[1] = avgub (s[0], s[2]);
  od[1] = s[1];
  if (isa == 2)
    simd_avx2_horiz_upsample_unaligned (ev + 2, od + 2, s + 1, (w - 4) >> 1);
  else
    simd_sse2_horiz_upsample_unaligned (ev + 2, od + 2, s + 1, (w - 4) >> 1);
  for (x = w - 2; x < w; x++) {
    if ((x & 1) == 0) {
      ev[x] = s[x];
      od[x] = s[x - 1];
    } else {
      ev[x] = s[x - 1];
      od[x] = s[x];
    }
  }
}

/* row kernels: 0 = the scalar restatement above, 1 = the reference's own
 * compiled kernels (oracle/_ref), 2 = SSE2, 3 = AVX2 restatement of the ORC
 * programs (bayer2rgb_simd.c) */
enum { ROWS_OWN = 0, ROWS_REF = 1, ROWS_SSE2 = 2, ROWS_AVX2 = 3 };

static void
any_row_lines (int mode, uint8_t *ev, uint8_t *od, const uint8_t *s, int w)
{
  if (mode == ROWS_REF)
    ref_row_lines (ev, od, s, w);
  else if (mode == ROWS_OWN)
    own_row_lines (ev, od, s, w);
  else
    simd_row_lines (mode - 1, ev, od, s, w);
}

/* gst_bayer2rgb_process, gstbayer2rgb.c:387-451; output rows y0 <= j < y1
 * only (a horizontal band of the frame; 0, h = the reference's whole-frame
 * loop).  A band primes the ring the way the reference primes it for row 0. */
static int
frame_driver_rows (uint8_t *dst, int dst_stride, const uint8_t *src,
    int src_stride, int w, int h, int pattern, int r_off, int g_off, int b_off,
    int mode, int y0, int y1)
{
  int j, layout, swap_rows;
  uint8_t *ring;

  if (w < 4 || (w & 1) || h < 3)
    return -1;
  if (pattern < 0 || pattern > 3)
    return -1;
  if (mode == ROWS_REF && !g_ref.handle)
    return -1;
  if (mode < ROWS_OWN || mode > ROWS_AVX2)
    return -1;
  if (mode == ROWS_AVX2 && oracle_simd_best_isa () < 2)
    return -1;
  if (y0 < 0 || y1 > h || y0 > y1)
    return -1;
  if (y0 == y1)
    return 0;

  /* :400-407 "For RGGB, we swap the red offset and blue offset" (also GBRG) */
  if (pattern == ORACLE_BAYER_RGGB || pattern == ORACLE_BAYER_GBRG) {
    int t = r_off;
    r_off = b_off;
    b_off = t;
  }
  layout = layout_of (r_off, g_off, b_off);     /* :409-421 */
  if (layout < 0)
    return -1;
  /* :422-427 "For GRBG, we swap the order of the merge functions" (also GBRG) */
  swap_rows = (pattern == ORACLE_BAYER_GRBG || pattern == ORACLE_BAYER_GBRG);

  /* :429-430 eight w-byte lines = ring of 4 rows x (ev, od) */
  ring = (uint8_t *) malloc ((size_t) 8 * w);
  if (!ring)
    return -1;
#define SLOT(n) (ring + (size_t) ((n) & 7) * w)

  /* :432-436 prime: row 1 stands in for row -1 (slot 3), then row 0; a band
   * that starts lower primes with rows y0-1 and y0 in the slots the whole-frame
   * loop would hold them in */
  any_row_lines (mode, SLOT (2 * y0 - 2), SLOT (2 * y0 - 1),
      src + (size_t) (y0 > 0 ? y0 - 1 : 1) * src_stride, w);
  any_row_lines (mode, SLOT (2 * y0), SLOT (2 * y0 + 1),
      src + (size_t) y0 * src_stride, w);

  /* :438-448 */
  for (j = y0; j < y1; j++) {
    int type = (j & 1) ^ swap_rows;     /* 0 = merge_bg, 1 = merge_gr */
    uint8_t *d = dst + (size_t) j * dst_stride;
    if (j < h - 1) {
      const uint8_t *s = src + (size_t) (j + 1) * src_stride;
      any_row_lines (mode, SLOT (2 * j + 2), SLOT (2 * j + 3), s, w);
    } else if (j - y0 < 3) {
      /* the last row pairs with whatever the ring slot (2j+2)&7 still holds:
       * row h-4 in the whole-frame loop (row 1, the stand-in for row -1, when
       * h == 3; :430-447).  A band that has not been running for three rows
       * yet computes that row explicitly. */
      any_row_lines (mode, SLOT (2 * j + 2), SLOT (2 * j + 3),
          src + (size_t) (h >= 4 ? h - 4 : 1) * src_stride, w);
    }
    if (mode == ROWS_REF) {
      g_ref.merge[layout][type] (d, SLOT (2 * j - 2), SLOT (2 * j - 1),
          SLOT (2 * j), SLOT (2 * j + 1), SLOT (2 * j + 2), SLOT (2 * j + 3),
          w >> 1);
    } else if (mode >= ROWS_SSE2) {
      simd_merge (mode - 1, layout, type, d, SLOT (2 * j - 2),
          SLOT (2 * j - 1), SLOT (2 * j), SLOT (2 * j + 1), SLOT (2 * j + 2),
          SLOT (2 * j + 3), w >> 1);
    } else if (type == 0) {
      own_merge_bg (d, SLOT (2 * j - 2), SLOT (2 * j - 1), SLOT (2 * j),
          SLOT (2 * j + 1), SLOT (2 * j + 2), SLOT (2 * j + 3), w >> 1,
          r_off, g_off, b_off);
    } else {
      own_merge_gr (d, SLOT (2 * j - 2), SLOT (2 * j - 1), SLOT (2 * j),
          SLOT (2 * j + 1), SLOT (2 * j + 2), SLOT (2 * j + 3), w >> 1,
          r_off, g_off, b_off);
    }
  }
#undef SLOT
  free (ring);
  return 0;
}

static int
frame_driver (uint8_t *dst, int dst_stride, const uint8_t *src, int src_stride,
    int w, int h, int pattern, int r_off, int g_off, int b_off, int mode)
{
  return frame_driver_rows (dst, dst_stride, src, src_stride, w, h, pattern,
      r_off, g_off, b_off, mode, 0, h);
}

int
oracle_bayer2rgb (uint8_t *dst, int dst_stride, const uint8_t *src,
    int src_stride, int width, int height, int pattern, int r_off, int g_off,
    int b_off)
{
  return frame_driver (dst, dst_stride, src, src_stride, width, height,
      pattern, r_off, g_off, b_off, ROWS_OWN);
}

int
oracle_bayer2rgb_refrows (uint8_t *dst, int dst_stride, const uint8_t *src,
    int src_stride, int width, int height, int pattern, int r_off, int g_off,
    int b_off)
{
  return frame_driver (dst, dst_stride, src, src_stride, width, height,
      pattern, r_off, g_off, b_off, ROWS_REF);
}

int
oracle_bayer2rgb_mode (uint8_t *dst, int dst_stride, const uint8_t *src,
    int src_stride, int width, int height, int pattern, int r_off, int g_off,
    int b_off, int mode, int y0, int y1)
{
  return frame_driver_rows (dst, dst_stride, src, src_stride, width, height,
      pattern, r_off, g_off, b_off, mode, y0, y1 < 0 ? height : y1);
}

/* ---- batch (frame- and band-parallel) ------------------------------------------ */

struct batch_job
{
  uint8_t *dst;
  size_t dst_frame_bytes;
  int dst_stride;
  const uint8_t *src;
  size_t src_frame_bytes;
  int src_stride;
  int w, h, pattern, r, g, b, nframes, nbands, nthreads, tid, mode, rc, repeat;
};

static void *
batch_worker (void *arg)
{
  struct batch_job *jb = (struct batch_job *) arg;
  long job, njobs = (long) jb->nframes * jb->nbands;
  int rep;
  for (rep = 0; rep < jb->repeat; rep++)
  for (job = jb->tid; job < njobs; job += jb->nthreads) {
    const int f = (int) (job / jb->nbands), band = (int) (job % jb->nbands);
    const int y0 = (int) ((long) jb->h * band / jb->nbands);
    const int y1 = (int) ((long) jb->h * (band + 1) / jb->nbands);
    int rc = frame_driver_rows (jb->dst + (size_t) f * jb->dst_frame_bytes,
        jb->dst_stride, jb->src + (size_t) f * jb->src_frame_bytes,
        jb->src_stride, jb->w, jb->h, jb->pattern, jb->r, jb->g, jb->b,
        jb->mode, y0, y1);
    if (rc)
      jb->rc = rc;
  }
  return NULL;
}

/* nframes x nbands independent jobs (frame f, rows h*b/nbands .. h*(b+1)/nbands),
 * job k on thread k mod nthreads */
static int batch_bands_repeat (uint8_t *dst, size_t dst_frame_bytes,
    int dst_stride, const uint8_t *src, size_t src_frame_bytes, int src_stride,
    int width, int height, int pattern, int r_off, int g_off, int b_off,
    int nframes, int nbands, int nthreads, int mode, int repeat);

int
oracle_bayer2rgb_batch_bands (uint8_t *dst, size_t dst_frame_bytes,
    int dst_stride, const uint8_t *src, size_t src_frame_bytes, int src_stride,
    int width, int height, int pattern, int r_off, int g_off, int b_off,
    int nframes, int nbands, int nthreads, int mode)
{
  return batch_bands_repeat (dst, dst_frame_bytes, dst_stride, src,
      src_frame_bytes, src_stride, width, height, pattern, r_off, g_off, b_off,
      nframes, nbands, nthreads, mode, 1);
}

/* the same conversion `repeat` times over inside the worker threads: for timing on many cores, where
 * creating and joining a few hundred threads per pass would otherwise be a large part of a pass */
int
oracle_bayer2rgb_batch_bands_repeat (uint8_t *dst, size_t dst_frame_bytes,
    int dst_stride, const uint8_t *src, size_t src_frame_bytes, int src_stride,
    int width, int height, int pattern, int r_off, int g_off, int b_off,
    int nframes, int nbands, int nthreads, int mode, int repeat)
{
  return batch_bands_repeat (dst, dst_frame_bytes, dst_stride, src,
      src_frame_bytes, src_stride, width, height, pattern, r_off, g_off, b_off,
      nframes, nbands, nthreads, mode, repeat < 1 ? 1 : repeat);
}

static int
batch_bands_repeat (uint8_t *dst, size_t dst_frame_bytes,
    int dst_stride, const uint8_t *src, size_t src_frame_bytes, int src_stride,
    int width, int height, int pattern, int r_off, int g_off, int b_off,
    int nframes, int nbands, int nthreads, int mode, int repeat)
{
  int t, rc = 0;
  long njobs;
  if (nbands < 1)
    nbands = 1;
  if (nbands > height)
    nbands = height;
  njobs = (long) nframes * nbands;
  if (nthreads < 1)
    nthreads = 1;
  if (nthreads > njobs)
    nthreads = njobs > 0 ? (int) njobs : 1;
  struct batch_job *jobs = calloc ((size_t) nthreads, sizeof *jobs);
  pthread_t *th = calloc ((size_t) nthreads, sizeof *th);
  if (!jobs || !th) {
    free (jobs);
    free (th);
    return -1;
  }
  for (t = 0; t < nthreads; t++) {
    struct batch_job jb = { dst, dst_frame_bytes, dst_stride, src,
      src_frame_bytes, src_stride, width, height, pattern, r_off, g_off,
      b_off, nframes, nbands, nthreads, t, mode, 0, repeat
    };
    jobs[t] = jb;
    if (t > 0 && pthread_create (&th[t], NULL, batch_worker, &jobs[t]) != 0) {
      batch_worker (&jobs[t]);  /* no thread to be had: do its share here */
      th[t] = 0;
    }
  }
  batch_worker (&jobs[0]);
  for (t = 1; t < nthreads; t++)
    if (th[t])
      pthread_join (th[t], NULL);
  for (t = 0; t < nthreads; t++)
    if (jobs[t].rc)
      rc = jobs[t].rc;
  free (jobs);
  free (th);
  return rc;
}

int
oracle_bayer2rgb_batch (uint8_t *dst, size_t dst_frame_bytes, int dst_stride,
    const uint8_t *src, size_t src_frame_bytes, int src_stride,
    int width, int height, int pattern, int r_off, int g_off, int b_off,
    int nframes, int nthreads, int use_ref_rows)
{
  return oracle_bayer2rgb_batch_bands (dst, dst_frame_bytes, dst_stride, src,
      src_frame_bytes, src_stride, width, height, pattern, r_off, g_off, b_off,
      nframes, 1, nthreads, use_ref_rows ? ROWS_REF : ROWS_OWN);
}

/* ---- rgb2bayer --------------------------------------------------------------- */

/* gst_rgb2bayer_transform, gstrgb2bayer.c:254-268 (pinned against the compiled reference function, see header) */
int
oracle_rgb2bayer (uint8_t *dst, int dst_stride, const uint8_t *src,
    int src_stride, int width, int height, int pattern, int r_off, int g_off,
    int b_off)
{
  int i, j;
  if (width < 1 || height < 1 || pattern < 0 || pattern > 3)
    return -1;
  for (j = 0; j < height; j++) {
    uint8_t *d = dst + (size_t) j * dst_stride;
    const uint8_t *s = src + (size_t) j * src_stride;
    for (i = 0; i < width; i++) {
      int site = ((j & 1) << 1) | (i & 1);      /* "is_blue", :259 */
      if (site == pattern)
        d[i] = s[4 * i + b_off];                /* :261, +3 for ARGB */
      else if ((site ^ 3) == pattern)
        d[i] = s[4 * i + r_off];                /* :263, +1 */
      else
        d[i] = s[4 * i + g_off];                /* :265, +2 */
    }
  }
  return 0;
}

/* ---- synthetic input --------------------------------- */

static inline uint32_t
fmix32 (uint32_t z)
{
  z ^= z >> 16;
  z *= 0x85EBCA6Bu;
  z ^= z >> 13;
  z *= 0xC2B2AE35u;
  z ^= z >> 16;
  return z;
}

void
oracle_fill_synthetic (uint8_t *buf, int width, int height, int stride,
    size_t frame_bytes, uint32_t first_frame, int nframes, uint32_t seed)
{
  int f, y, x;
  for (f = 0; f < nframes; f++) {
    uint8_t *fr = buf + (size_t) f * frame_bytes;
    uint32_t base = (first_frame + (uint32_t) f) * (uint32_t) height
        * (uint32_t) width;
    for (y = 0; y < height; y++) {
      uint8_t *row = fr + (size_t) y * stride;
      uint32_t idx = base + (uint32_t) y * (uint32_t) width;
      for (x = 0; x < width; x++)
        row[x] = (uint8_t) (fmix32 ((idx + (uint32_t) x) * 2654435761u
                + seed * 0x9E3779B9u) & 0xffu);
      for (; x < stride; x++)
        row[x] = 0;
    }
  }
}
