/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  See bayer2rgb_oracle.h.
 *
 * "ORC-equivalent" SIMD row kernels for the CPU baseline leg of bench.py
 * (BASELINE.md section 4): what the reference's ORC
 * JIT emits on x86-64 cannot be built or run here (no liborc, no orcc), so
 * the two ORC programs on the frame path are restated as the instruction
 * sequences they compile to -- `avgub` IS pavgb (_mm_avg_epu8), `mergebw` /
 * `mergewl` ARE punpck{l,h}bw / punpck{l,h}wd, `splitwb` is a shift and a
 * mask on 16-bit lanes -- one SSE2 form (ORC's x86-64 backend of the 1.19 era
 * is SSE) and one AVX2 form of the same programs:
 *
 *   bayer_orc_horiz_upsample_unaligned   gst/bayer/gstbayerorc.orc:3-19
 *       (JIT wrapper gst/bayer/gstbayerorc-dist.c:913-1007)
 *   bayer_orc_merge_bg_{bgra,abgr,argb,rgba}   gstbayerorc.orc:43-66 (+ :95-118,
 *       :147-170, :199-222)
 *   bayer_orc_merge_gr_{bgra,abgr,argb,rgba}   gstbayerorc.orc:69-92 (+ :121-144,
 *       :173-196, :225-248)
 *
 * Signatures are those of the reference's row kernels (n = number of 2-byte
 * elements, i.e. pixel PAIRS), so the restated frame driver in
 * bayer2rgb_oracle.c runs over them exactly as it runs over oracle/_ref.
 * tests/test_oracle.py proves every function byte-equal to oracle/_ref (the
 * reference's own compiled row kernels) and to the scalar restatement.
 */
#include "bayer2rgb_oracle.h"

#include <emmintrin.h>
#include <immintrin.h>
#include <string.h>

static inline uint8_t
avgub1 (uint8_t a, uint8_t b)
{
  return (uint8_t) (((unsigned) a + (unsigned) b + 1u) >> 1);
}

/* ---- scalar tails (the ORC executor's "region 3" loop) ------------------- */

static void
upsample_tail (uint8_t *d0, uint8_t *d1, const uint8_t *s, int i, int n)
{
  for (; i < n; i++) {
    /* orc:10-19: s word = (b lo, c hi); word at offset 1 = (d lo, e hi) */
    uint8_t b = s[2 * i], c = s[2 * i + 1], d = s[2 * i + 2], e = s[2 * i + 3];
    d0[2 * i] = c;
    d0[2 * i + 1] = avgub1 (c, e);
    d1[2 * i] = avgub1 (b, d);
    d1[2 * i + 1] = d;
  }
}

/* type 0 = merge_bg (orc:57-66), 1 = merge_gr (orc:83-92); pr/pg/pb/pa = byte
 * position of the r-named, green, b-named and 255 value */
static void
merge_tail (uint8_t *d, const uint8_t *u0, const uint8_t *u1,
    const uint8_t *c0, const uint8_t *c1, const uint8_t *d0, const uint8_t *d1,
    int i, int n, int type, int pr, int pg, int pb, int pa)
{
  for (i *= 2; i < 2 * n; i++) {
    uint8_t r, g, b;
    if (type == 0) {
      r = avgub1 (u1[i], d1[i]);
      g = (i & 1) ? c1[i] : avgub1 (avgub1 (u0[i], d0[i]), c1[i]);
      b = c0[i];
    } else {
      b = avgub1 (u0[i], d0[i]);
      g = (i & 1) ? avgub1 (avgub1 (u1[i], d1[i]), c0[i]) : c0[i];
      r = c1[i];
    }
    d[4 * i + pr] = r;
    d[4 * i + pg] = g;
    d[4 * i + pb] = b;
    d[4 * i + pa] = 255;
  }
}

/* ---- SSE2 ------------------------------------------------------------------ */

void
simd_sse2_horiz_upsample_unaligned (uint8_t *d0, uint8_t *d1, const uint8_t *s,
    int n)
{
  const __m128i lo = _mm_set1_epi16 (0x00ff);
  int i = 0;
  for (; i + 8 <= n; i += 8) {
    const __m128i v0 = _mm_loadu_si128 ((const __m128i *) (s + 2 * i));
    const __m128i v1 = _mm_loadu_si128 ((const __m128i *) (s + 2 * i + 2)); /* loadoffw t, s, 1 */
    const __m128i b = _mm_and_si128 (v0, lo);   /* splitwb c, b, s */
    const __m128i c = _mm_srli_epi16 (v0, 8);
    const __m128i d = _mm_and_si128 (v1, lo);   /* splitwb e, d, t */
    __m128i e = _mm_srli_epi16 (v1, 8);
    e = _mm_avg_epu8 (c, e);                    /* avgub e, c, e */
    _mm_storeu_si128 ((__m128i *) (d0 + 2 * i),
        _mm_or_si128 (c, _mm_slli_epi16 (e, 8)));       /* mergebw d0, c, e */
    const __m128i bb = _mm_avg_epu8 (b, d);     /* avgub b, b, d */
    _mm_storeu_si128 ((__m128i *) (d1 + 2 * i),
        _mm_or_si128 (bb, _mm_slli_epi16 (d, 8)));      /* mergebw d1, b, d */
  }
  upsample_tail (d0, d1, s, i, n);
}

/* 16 pixels: planes p[0..3] = the byte written at offset 0..3 of each pixel */
static inline void
store16_sse2 (uint8_t *d, __m128i p0, __m128i p1, __m128i p2, __m128i p3)
{
  const __m128i a_lo = _mm_unpacklo_epi8 (p0, p1);      /* x2 mergebw */
  const __m128i a_hi = _mm_unpackhi_epi8 (p0, p1);
  const __m128i b_lo = _mm_unpacklo_epi8 (p2, p3);
  const __m128i b_hi = _mm_unpackhi_epi8 (p2, p3);
  _mm_storeu_si128 ((__m128i *) (d + 0), _mm_unpacklo_epi16 (a_lo, b_lo));   /* x2 mergewl */
  _mm_storeu_si128 ((__m128i *) (d + 16), _mm_unpackhi_epi16 (a_lo, b_lo));
  _mm_storeu_si128 ((__m128i *) (d + 32), _mm_unpacklo_epi16 (a_hi, b_hi));
  _mm_storeu_si128 ((__m128i *) (d + 48), _mm_unpackhi_epi16 (a_hi, b_hi));
}

static void
merge_sse2 (uint8_t *d, const uint8_t *u0, const uint8_t *u1, const uint8_t *c0,
    const uint8_t *c1, const uint8_t *d0, const uint8_t *d1, int n, int type,
    int pr, int pg, int pb)
{
  const __m128i even = _mm_set1_epi16 (0x00ff);         /* andw g, g, 255 */
  const __m128i ff = _mm_set1_epi8 ((char) 0xff);
  const int pa = 6 - pr - pg - pb;
  int i = 0;
  for (; i + 8 <= n; i += 8) {
    const __m128i U0 = _mm_loadu_si128 ((const __m128i *) (u0 + 2 * i));
    const __m128i U1 = _mm_loadu_si128 ((const __m128i *) (u1 + 2 * i));
    const __m128i C0 = _mm_loadu_si128 ((const __m128i *) (c0 + 2 * i));
    const __m128i C1 = _mm_loadu_si128 ((const __m128i *) (c1 + 2 * i));
    const __m128i D0 = _mm_loadu_si128 ((const __m128i *) (d0 + 2 * i));
    const __m128i D1 = _mm_loadu_si128 ((const __m128i *) (d1 + 2 * i));
    __m128i r, g, b;
    if (type == 0) {
      r = _mm_avg_epu8 (U1, D1);                        /* x2 avgub r, r0, r2 */
      g = _mm_avg_epu8 (_mm_avg_epu8 (U0, D0), C1);     /* x2 avgub g, g0, g2; x2 avgub g, g, t */
      g = _mm_or_si128 (_mm_and_si128 (g, even), _mm_andnot_si128 (even, C1));
      b = C0;
    } else {
      b = _mm_avg_epu8 (U0, D0);
      g = _mm_avg_epu8 (_mm_avg_epu8 (U1, D1), C0);
      g = _mm_or_si128 (_mm_andnot_si128 (even, g), _mm_and_si128 (even, C0));
      r = C1;
    }
    __m128i p[4];
    p[pr] = r;
    p[pg] = g;
    p[pb] = b;
    p[pa] = ff;
    store16_sse2 (d + 8 * i, p[0], p[1], p[2], p[3]);
  }
  merge_tail (d, u0, u1, c0, c1, d0, d1, i, n, type, pr, pg, pb, pa);
}

/* ---- AVX2 ------------------------------------------------------------------ */

__attribute__ ((target ("avx2")))
void
simd_avx2_horiz_upsample_unaligned (uint8_t *d0, uint8_t *d1, const uint8_t *s,
    int n)
{
  const __m256i lo = _mm256_set1_epi16 (0x00ff);
  int i = 0;
  for (; i + 16 <= n; i += 16) {
    const __m256i v0 = _mm256_loadu_si256 ((const __m256i *) (s + 2 * i));
    const __m256i v1 = _mm256_loadu_si256 ((const __m256i *) (s + 2 * i + 2));
    const __m256i b = _mm256_and_si256 (v0, lo);
    const __m256i c = _mm256_srli_epi16 (v0, 8);
    const __m256i d = _mm256_and_si256 (v1, lo);
    __m256i e = _mm256_srli_epi16 (v1, 8);
    e = _mm256_avg_epu8 (c, e);
    _mm256_storeu_si256 ((__m256i *) (d0 + 2 * i),
        _mm256_or_si256 (c, _mm256_slli_epi16 (e, 8)));
    const __m256i bb = _mm256_avg_epu8 (b, d);
    _mm256_storeu_si256 ((__m256i *) (d1 + 2 * i),
        _mm256_or_si256 (bb, _mm256_slli_epi16 (d, 8)));
  }
  upsample_tail (d0, d1, s, i, n);
}

__attribute__ ((target ("avx2")))
static void
merge_avx2 (uint8_t *d, const uint8_t *u0, const uint8_t *u1, const uint8_t *c0,
    const uint8_t *c1, const uint8_t *d0, const uint8_t *d1, int n, int type,
    int pr, int pg, int pb)
{
  const __m256i even = _mm256_set1_epi16 (0x00ff);
  const __m256i ff = _mm256_set1_epi8 ((char) 0xff);
  const int pa = 6 - pr - pg - pb;
  int i = 0;
  for (; i + 16 <= n; i += 16) {
    const __m256i U0 = _mm256_loadu_si256 ((const __m256i *) (u0 + 2 * i));
    const __m256i U1 = _mm256_loadu_si256 ((const __m256i *) (u1 + 2 * i));
    const __m256i C0 = _mm256_loadu_si256 ((const __m256i *) (c0 + 2 * i));
    const __m256i C1 = _mm256_loadu_si256 ((const __m256i *) (c1 + 2 * i));
    const __m256i D0 = _mm256_loadu_si256 ((const __m256i *) (d0 + 2 * i));
    const __m256i D1 = _mm256_loadu_si256 ((const __m256i *) (d1 + 2 * i));
    __m256i r, g, b;
    if (type == 0) {
      r = _mm256_avg_epu8 (U1, D1);
      g = _mm256_avg_epu8 (_mm256_avg_epu8 (U0, D0), C1);
      g = _mm256_or_si256 (_mm256_and_si256 (g, even),
          _mm256_andnot_si256 (even, C1));
      b = C0;
    } else {
      b = _mm256_avg_epu8 (U0, D0);
      g = _mm256_avg_epu8 (_mm256_avg_epu8 (U1, D1), C0);
      g = _mm256_or_si256 (_mm256_andnot_si256 (even, g),
          _mm256_and_si256 (even, C0));
      r = C1;
    }
    __m256i p[4];
    p[pr] = r;
    p[pg] = g;
    p[pb] = b;
    p[pa] = ff;
    /* the 256-bit unpacks work per 128-bit half: pixels 0-7|16-23 and 8-15|24-31 */
    const __m256i a_lo = _mm256_unpacklo_epi8 (p[0], p[1]);
    const __m256i a_hi = _mm256_unpackhi_epi8 (p[0], p[1]);
    const __m256i b_lo = _mm256_unpacklo_epi8 (p[2], p[3]);
    const __m256i b_hi = _mm256_unpackhi_epi8 (p[2], p[3]);
    const __m256i q0 = _mm256_unpacklo_epi16 (a_lo, b_lo);      /* px 0-3 | 16-19 */
    const __m256i q1 = _mm256_unpackhi_epi16 (a_lo, b_lo);      /* px 4-7 | 20-23 */
    const __m256i q2 = _mm256_unpacklo_epi16 (a_hi, b_hi);      /* px 8-11 | 24-27 */
    const __m256i q3 = _mm256_unpackhi_epi16 (a_hi, b_hi);      /* px 12-15 | 28-31 */
    uint8_t *o = d + 8 * i;
    _mm256_storeu_si256 ((__m256i *) (o + 0), _mm256_permute2x128_si256 (q0, q1, 0x20));
    _mm256_storeu_si256 ((__m256i *) (o + 32), _mm256_permute2x128_si256 (q2, q3, 0x20));
    _mm256_storeu_si256 ((__m256i *) (o + 64), _mm256_permute2x128_si256 (q0, q1, 0x31));
    _mm256_storeu_si256 ((__m256i *) (o + 96), _mm256_permute2x128_si256 (q2, q3, 0x31));
  }
  merge_tail (d, u0, u1, c0, c1, d0, d1, i, n, type, pr, pg, pb, pa);
}

/* ---- the reference's row-kernel entry points, per ISA ------------------------ */
/* layout order bgra, abgr, argb, rgba (gstbayer2rgb.c:409-421) -> position of
 * the r-named, green and b-named byte in the output pixel */
static const int kLayoutPos[4][3] = {
  {2, 1, 0},                    /* bgra */
  {3, 2, 1},                    /* abgr */
  {1, 2, 3},                    /* argb */
  {0, 1, 2},                    /* rgba */
};

void
simd_merge (int isa, int layout, int type, uint8_t *d, const uint8_t *u0,
    const uint8_t *u1, const uint8_t *c0, const uint8_t *c1, const uint8_t *d0,
    const uint8_t *d1, int n)
{
  const int *p = kLayoutPos[layout];
  if (isa == 2)
    merge_avx2 (d, u0, u1, c0, c1, d0, d1, n, type, p[0], p[1], p[2]);
  else
    merge_sse2 (d, u0, u1, c0, c1, d0, d1, n, type, p[0], p[1], p[2]);
}

/* 1 = SSE2 (always there on x86-64), 2 = AVX2 */
int
oracle_simd_best_isa (void)
{
  return __builtin_cpu_supports ("avx2") ? 2 : 1;
}
