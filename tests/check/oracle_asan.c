/* TEST INFRASTRUCTURE.  Runs the CPU oracle over exact-size heap buffers for a sweep of geometries, all orders
 * and layouts, built by tests/test_oracle_sanitizers.py with -fsanitize=address,undefined: any out-of-bounds
 * access or UB in the restatement (which judges the HIP path) aborts the run.  The reference itself reads one
 * byte past short rows for W < 4 and uninitialised scratch for odd W (gstbayer2rgb.c:362, :365-380); the oracle
 * rejects that domain instead, which is asserted here too. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bayer2rgb_oracle.h"

static const int layouts[4][3] = { {0, 1, 2}, {2, 1, 0}, {1, 2, 3}, {3, 2, 1} };

int
main (void)
{
  int w, h, p, l, n = 0;
  unsigned seed = 12345;

  for (w = 4; w <= 78; w += 2) {
    for (h = 3; h <= 9; h++) {
      int sstride = (w + 3) & ~3;
      uint8_t *src = malloc ((size_t) sstride * h);
      uint8_t *dst = malloc ((size_t) 4 * w * h);
      uint8_t *back = malloc ((size_t) sstride * h);
      uint8_t *alt = malloc ((size_t) 4 * w * h);
      size_t i;

      for (i = 0; i < (size_t) sstride * h; i++) {
        seed = seed * 1664525u + 1013904223u;
        src[i] = (uint8_t) (seed >> 24);
      }
      for (p = 0; p < 4; p++) {
        for (l = 0; l < 4; l++) {
          if (oracle_bayer2rgb (dst, 4 * w, src, sstride, w, h, p, layouts[l][0], layouts[l][1],
                  layouts[l][2]) != 0)
            return 2;
          /* the SIMD restatements of the ORC programs and the band-parallel driver: same bytes, and no
           * access outside the exact-size buffers (vector loops + scalar tails at every width) */
          {
            int mode, cut = 1 + (w + h + p) % (h - 1);
            for (mode = ORACLE_ROWS_SSE2; mode <= ORACLE_ROWS_SSE2 + oracle_simd_best_isa () - 1; mode++) {
              memset (alt, 0x5a, (size_t) 4 * w * h);
              if (oracle_bayer2rgb_mode (alt, 4 * w, src, sstride, w, h, p, layouts[l][0], layouts[l][1],
                      layouts[l][2], mode, 0, cut) != 0
                  || oracle_bayer2rgb_mode (alt, 4 * w, src, sstride, w, h, p, layouts[l][0], layouts[l][1],
                      layouts[l][2], mode, cut, h) != 0)
                return 9;
              if (memcmp (alt, dst, (size_t) 4 * w * h) != 0)
                return 10;
            }
          }
          /* exact left inverse */
          if (oracle_rgb2bayer (back, sstride, dst, 4 * w, w, h, p, layouts[l][0], layouts[l][1],
                  layouts[l][2]) != 0)
            return 3;
          for (i = 0; i < (size_t) h; i++)
            if (memcmp (back + i * sstride, src + i * sstride, (size_t) w) != 0)
              return 4;
          n++;
        }
      }
      free (src);
      free (dst);
      free (back);
      free (alt);
    }
  }
  {
    uint8_t tiny[64] = { 0 }, out[256];
    if (oracle_bayer2rgb (out, 8, tiny, 4, 2, 4, 0, 0, 1, 2) == 0)
      return 5;                 /* W < 4 */
    if (oracle_bayer2rgb (out, 20, tiny, 8, 5, 4, 0, 0, 1, 2) == 0)
      return 6;                 /* odd W */
    if (oracle_bayer2rgb (out, 16, tiny, 4, 4, 2, 0, 0, 1, 2) == 0)
      return 7;                 /* H < 3 */
    if (oracle_bayer2rgb (out, 16, tiny, 4, 4, 4, 0, 0, 2, 1) == 0)
      return 8;                 /* unknown layout */
  }
  {
    uint8_t *frames = malloc ((size_t) 3 * 5 * 36);
    uint8_t *outs = malloc ((size_t) 3 * 5 * 4 * 34);
    oracle_fill_synthetic (frames, 34, 5, 36, (size_t) 5 * 36, 2, 3, 9);
    if (oracle_bayer2rgb_batch_bands (outs, (size_t) 5 * 4 * 34, 4 * 34, frames, (size_t) 5 * 36, 36, 34, 5, 1,
            2, 1, 0, 3, 4, 5, ORACLE_ROWS_SSE2) != 0)
      return 11;
    free (frames);
    free (outs);
  }
  printf ("sanitized oracle: %d conversions ok\n", n);
  return 0;
}
