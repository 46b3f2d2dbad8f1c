/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU restatement of gst-plugins-bad's bayer2rgb frame path, used only as the
 * parity checker (tests/, __graft_entry__.smoke()) and as the timed
 * `cpu_baseline` leg of bench.py.  Nothing under gst-plugins-bad_amd/ may
 * include, link or dlopen this.
 *
 * Reference (v1.19.2, /root/reference):
 *   gst/bayer/gstbayer2rgb.c:354-381   gst_bayer2rgb_split_and_upsample_horiz
 *   gst/bayer/gstbayer2rgb.c:387-451   gst_bayer2rgb_process
 *   gst/bayer/gstbayerorc.orc:3-19     bayer_orc_horiz_upsample_unaligned
 *   gst/bayer/gstbayerorc.orc:43-248   bayer_orc_merge_{bg,gr}_{bgra,abgr,rgba,argb}
 *
 * Parity pin: see oracle/README.md (the reference's own frame driver and row
 * kernels compiled into oracle/_ref, the md5 known answers of SURVEY.md
 * Appendix B.3 and the hand-checkable 4x4 frame of Appendix B.4).
 */
#ifndef BAYER2RGB_ORACLE_H
#define BAYER2RGB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* same numbering as the reference enum, gstbayer2rgb.c:95-101 */
enum {
  ORACLE_BAYER_BGGR = 0,
  ORACLE_BAYER_GBRG = 1,
  ORACLE_BAYER_GRBG = 2,
  ORACLE_BAYER_RGGB = 3
};

/* One frame.  Arguments mirror gst_bayer2rgb_process (gstbayer2rgb.c:387-389)
 * with the GstBayer2RGB fields it reads (width, height, format, r/g/b_off)
 * passed explicitly.  Returns 0, or -1 for geometry outside the domain in
 * which the reference is well defined (even width >= 4, height >= 3) or an
 * (r,g,b) offset triple the reference has no merge function for. */
int oracle_bayer2rgb (uint8_t *dst, int dst_stride,
    const uint8_t *src, int src_stride,
    int width, int height, int pattern, int r_off, int g_off, int b_off);

/* Same driver, but every row is computed by the REFERENCE's own compiled row
 * kernels (gstbayerorc-dist.c built with -DDISABLE_ORC into
 * oracle/_ref/libbayerorc_ref.so).  Call oracle_load_ref_rows() first. */
int oracle_load_ref_rows (const char *so_path);
int oracle_bayer2rgb_refrows (uint8_t *dst, int dst_stride,
    const uint8_t *src, int src_stride,
    int width, int height, int pattern, int r_off, int g_off, int b_off);

/* nframes independent frames, frame f on thread (f mod nthreads).
 * use_ref_rows != 0 selects the reference row kernels. */
int oracle_bayer2rgb_batch (uint8_t *dst, size_t dst_frame_bytes, int dst_stride,
    const uint8_t *src, size_t src_frame_bytes, int src_stride,
    int width, int height, int pattern, int r_off, int g_off, int b_off,
    int nframes, int nthreads, int use_ref_rows);

/* Row-kernel selection for the two entry points below: 0 = the scalar
 * restatement, 1 = the reference's compiled row kernels (oracle/_ref, load them
 * first), 2 = SSE2, 3 = AVX2 restatement of the ORC programs as the pavgb /
 * punpck sequences they compile to (bayer2rgb_simd.c; -1 if the CPU lacks the
 * ISA).  All four are byte-identical (tests/test_oracle.py). */
enum {
  ORACLE_ROWS_OWN = 0,
  ORACLE_ROWS_REF = 1,
  ORACLE_ROWS_SSE2 = 2,
  ORACLE_ROWS_AVX2 = 3
};
/* 1 = SSE2 only, 2 = AVX2 available */
int oracle_simd_best_isa (void);

/* One frame, output rows y0 <= j < y1 only (y1 < 0: height): the unit of the
 * band-parallel CPU baseline.  Rows outside the band are not touched. */
int oracle_bayer2rgb_mode (uint8_t *dst, int dst_stride, const uint8_t *src,
    int src_stride, int width, int height, int pattern, int r_off, int g_off,
    int b_off, int mode, int y0, int y1);

/* nframes x nbands independent jobs (frame f, rows h*b/nbands .. h*(b+1)/nbands)
 * spread over nthreads pthreads: lets the all-cores baseline use every core of
 * the host whatever the number of frames. */
int oracle_bayer2rgb_batch_bands (uint8_t *dst, size_t dst_frame_bytes,
    int dst_stride, const uint8_t *src, size_t src_frame_bytes, int src_stride,
    int width, int height, int pattern, int r_off, int g_off, int b_off,
    int nframes, int nbands, int nthreads, int mode);

/* the same, `repeat` passes inside the worker threads (timing on many cores) */
int oracle_bayer2rgb_batch_bands_repeat (uint8_t *dst, size_t dst_frame_bytes,
    int dst_stride, const uint8_t *src, size_t src_frame_bytes, int src_stride,
    int width, int height, int pattern, int r_off, int g_off, int b_off,
    int nframes, int nbands, int nthreads, int mode, int repeat);

/* Inverse element rgb2bayer, reference gst/bayer/gstrgb2bayer.c:254-268: output
 * byte (j,i) is byte r_off / g_off / b_off of input pixel (j,i) according to the
 * CFA site ((j&1)<<1)|(i&1); the reference hard-codes ARGB (r,g,b = 1,2,3).
 * Padding bytes of a destination row are left untouched, as in the reference.
 * PARITY PINNED (round 5): the reference's own gst_rgb2bayer_transform
 * (gstrgb2bayer.c:229-278) is compiled where it lies into
 * oracle/_ref/libbayer_frame_ref.so (oracle/Makefile: ref_frame; its lines are
 * extracted at build time, real GStreamer 1.14 headers, no stand-ins); this
 * function equals it byte for byte incl. odd sizes and padded source rows, and
 * tests/golden/rgb2bayer_small.npz + the rgb2bayer md5 table hold its outputs
 * (tests/test_oracle.py).  Second, independent cross-check of the site ->
 * channel mapping: gst-plugins-base's videotestsrc, the GStreamer 1.14 binary
 * of this image, writes video/x-bayer itself from the ARGB it paints; its
 * mosaic equals this function of its ARGB frame for all four orders. */
int oracle_rgb2bayer (uint8_t *dst, int dst_stride, const uint8_t *src,
    int src_stride, int width, int height, int pattern, int r_off, int g_off,
    int b_off);

/* Counter-based synthetic frames, SURVEY.md Appendix C:
 * byte(f,y,x) = fmix32((f*H*W + y*W + x) * 2654435761 + seed*0x9E3779B9) & 0xff.
 * Padding bytes (x >= W) are written as 0. */
void oracle_fill_synthetic (uint8_t *buf, int width, int height, int stride,
    size_t frame_bytes, uint32_t first_frame, int nframes, uint32_t seed);

#ifdef __cplusplus
}
#endif
#endif
