/* TEST DOUBLE of libmibayer.so -- NOT a conversion path, NOT shipped, NOT a fallback.
 *
 * The elements of plugin `bayer` (gst-plugins-bad_amd/gst/gstmibayerelement.c) talk to the GPU only through ten
 * entry points of include/mibayer.h.  This file implements exactly those ten with NO demosaic in them, so that
 * the elements' own logic -- buffer ownership in the synchronous and the queued mode, ordering, draining on
 * EOS / caps / segment events, dropping on flush, pool re-creation on renegotiation -- can be exercised on a
 * machine without a GPU, under AddressSanitizer (tests/test_gst_element_logic.py).
 *
 * What a "conversion" does here: nothing at submit time; at wait time (the moment the real library would have
 * finished its asynchronous work) it READS every source byte and WRITES every destination byte the real kernel
 * would write, so a buffer the element unmapped or released too early is a sanitizer report.  The output is a
 * stamp, not an image: bytes 0..3 of the frame = submission sequence number, every other written byte = the
 * first source byte of that frame.
 */
#include "mibayer.h"

#include <stdlib.h>
#include <string.h>

#define MOCK_MAX_PENDING 1024

typedef struct
{
  const uint8_t *src;
  uint8_t *dst;
  void *tag;
  uint32_t seq;
} mock_frame;

struct mibayer_pool
{
  mibayer_cfg cfg;
  int capacity;
  mock_frame fifo[MOCK_MAX_PENDING];
  int head, count;
  uint32_t seq;
};

int
mibayer_device_count (void)
{
  const char *e = getenv ("MOCK_MIBAYER_DEVICES");

  return e ? atoi (e) : 1;
}

const char *
mibayer_strerror (int status)
{
  return status == MIBAYER_OK ? "ok" : "mock error";
}

const char *
mibayer_last_hip_error (void)
{
  return "";
}

void *
mibayer_host_alloc (size_t bytes)
{
  return malloc (bytes ? bytes : 1);    /* plain heap: the sanitizer sees its bounds */
}

void
mibayer_host_free (void *p)
{
  free (p);
}

int
mibayer_pool_create (const mibayer_pool_cfg * cfg, mibayer_pool ** out)
{
  mibayer_pool *p;
  const mibayer_cfg *f;
  int inverse;

  if (!cfg || !out || cfg->struct_size != sizeof (*cfg) || cfg->ndevices < 1)
    return MIBAYER_ERR_ARG;
  f = &cfg->stream;
  inverse = (f->flags & MIBAYER_FLAG_RGB2BAYER) != 0;
  if (mibayer_device_count () <= 0)
    return MIBAYER_ERR_NO_DEVICE;
  /* the real library's geometry domain for bayer2rgb */
  if (!inverse && (f->width < 4 || (f->width & 1) || f->height < 3))
    return MIBAYER_ERR_GEOMETRY;
  p = calloc (1, sizeof *p);
  p->cfg = *f;
  p->capacity = cfg->ndevices * (f->inflight > 0 ? f->inflight : 2);
  if (p->capacity > MOCK_MAX_PENDING)
    p->capacity = MOCK_MAX_PENDING;
  *out = p;
  return MIBAYER_OK;
}

void
mibayer_pool_destroy (mibayer_pool * p)
{
  free (p);
}

int
mibayer_pool_capacity (const mibayer_pool * p)
{
  return p ? p->capacity : MIBAYER_ERR_ARG;
}

int
mibayer_pool_pending (const mibayer_pool * p)
{
  return p ? p->count : MIBAYER_ERR_ARG;
}

int
mibayer_pool_submit (mibayer_pool * p, const uint8_t * src, uint8_t * dst,
    void *tag)
{
  mock_frame *fr;

  if (!p || !src || !dst)
    return MIBAYER_ERR_ARG;
  if (p->count == p->capacity)
    return MIBAYER_ERR_BUSY;
  fr = &p->fifo[(p->head + p->count) % MOCK_MAX_PENDING];
  fr->src = src;
  fr->dst = dst;
  fr->tag = tag;
  fr->seq = p->seq++;
  p->count++;
  return MIBAYER_OK;
}

int
mibayer_pool_wait (mibayer_pool * p, void **tag)
{
  const mibayer_cfg *f;
  mock_frame fr;
  int inverse, y, src_row, dst_row;
  unsigned sum = 0;

  if (!p)
    return MIBAYER_ERR_ARG;
  if (p->count == 0)
    return MIBAYER_ERR_EMPTY;
  fr = p->fifo[p->head];
  p->head = (p->head + 1) % MOCK_MAX_PENDING;
  p->count--;
  f = &p->cfg;
  inverse = (f->flags & MIBAYER_FLAG_RGB2BAYER) != 0;
  /* bytes per row the real path reads / writes */
  src_row = inverse ? 4 * f->width : ((f->width + 3) & ~3);
  dst_row = inverse ? ((f->width + 3) & ~3) : 4 * f->width;
  for (y = 0; y < f->height; y++) {
    const uint8_t *s = fr.src + (size_t) y * f->src_stride;
    int x;

    for (x = 0; x < src_row; x++)
      sum += s[x];              /* the source must still be mapped and alive */
  }
  for (y = 0; y < f->height; y++)
    memset (fr.dst + (size_t) y * f->dst_stride, fr.src[0], (size_t) dst_row);
  memcpy (fr.dst, &fr.seq, 4);
  if (sum == 0xffffffffu)       /* keep the reads */
    fr.dst[4] ^= 1;
  if (tag)
    *tag = fr.tag;
  return MIBAYER_OK;
}
