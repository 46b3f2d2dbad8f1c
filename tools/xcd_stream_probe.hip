/* Development tool: do 8 XCD-affine sequential write streams interfere depending on their relative placement?
 * Block b runs on XCD b % 8 (observed dispatch rule) and writes the (b / 8)-th 32 KiB piece of that XCD's own region
 * at base + xcd * stride.  Sweeps `stride` and prints the aggregate write bandwidth.
 *   hipcc --offload-arch=gfx950 -O3 tools/xcd_stream_probe.hip -o tools/xcd_stream_probe */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf (stderr, "%s: %s\n", #x, hipGetErrorString (e)); exit (1); } } while (0)

/* 512 threads, each block writes 32 KiB = 512 lanes x 16 B x 4 */
__global__ void __launch_bounds__ (512) k_xcd_fill (uint8_t *base, size_t stride, size_t pieces_per_xcd, int mode)
{
  const size_t b = blockIdx.x;
  size_t off;
  if (mode == 0) {                       /* one private region per XCD */
    const size_t xcd = b % 8, i = b / 8;
    if (i >= pieces_per_xcd) return;
    off = xcd * stride + i * 32768;
  } else {                               /* identity: all XCDs interleaved in one region */
    if (b >= 8 * pieces_per_xcd) return;
    off = b * 32768;
  }
  u32x4 v = { (uint32_t) b, 2u, 3u, (uint32_t) threadIdx.x };
  u32x4 *p = (u32x4 *) (base + off) + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 4; k++)
    __builtin_nontemporal_store (v, p + k * 512);
}

int main (int argc, char **argv)
{
  const size_t per_xcd = (size_t) 256 << 20;              /* 256 MiB written per XCD, 2 GiB total */
  const size_t max_stride = per_xcd + ((size_t) 64 << 20);
  uint8_t *buf;
  CK (hipMalloc (&buf, 8 * max_stride + per_xcd));
  CK (hipMemset (buf, 0, 8 * max_stride + per_xcd));
  hipEvent_t ev0, ev1;
  CK (hipEventCreate (&ev0)); CK (hipEventCreate (&ev1));
  const size_t pieces = per_xcd / 32768;
  auto run = [&](size_t stride, int mode) {
    for (int w = 0; w < 2; w++)
      hipLaunchKernelGGL (k_xcd_fill, dim3 ((unsigned) (8 * pieces)), dim3 (512), 0, 0, buf, stride, pieces, mode);
    CK (hipEventRecord (ev0, 0));
    const int reps = 10;
    for (int r = 0; r < reps; r++)
      hipLaunchKernelGGL (k_xcd_fill, dim3 ((unsigned) (8 * pieces)), dim3 (512), 0, 0, buf, stride, pieces, mode);
    CK (hipEventRecord (ev1, 0)); CK (hipEventSynchronize (ev1));
    float ms; CK (hipEventElapsedTime (&ms, ev0, ev1));
    return 8.0 * per_xcd / (ms / reps) / 1e6;
  };
  printf ("identity (one interleaved region)            %8.1f GB/s\n", run (0, 1));
  const size_t steps_kib[] = { 0, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 16384, 32768, 65536 };
  for (size_t s : steps_kib) {
    const size_t stride = per_xcd + (s << 10);
    printf ("per-XCD regions, stride 256 MiB + %6zu KiB   %8.1f GB/s\n", s, run (stride, 0));
  }
  printf ("identity again                                %8.1f GB/s\n", run (0, 1));
  return 0;
}
