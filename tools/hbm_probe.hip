/* Development tool (not product): empirical HBM ceilings on the box for the traffic mixes that
 * matter to bayer2rgb -- pure read, pure write, 1:1 copy, and the 1 B read : 4 B written mix of
 * the demosaic -- with plain and non-temporal stores.  Build: hipcc --offload-arch=gfx950 -O3
 * tools/hbm_probe.hip -o tools/hbm_probe */
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf (stderr, "%s: %s\n", #x, hipGetErrorString (e)); exit (1); } } while (0)

template <bool NT>
__global__ void __launch_bounds__ (256) k_fill (u32x4 *dst, size_t n16)
{
  const u32x4 v = { 1u, 2u, 3u, (uint32_t) threadIdx.x };
  for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n16; i += (size_t) gridDim.x * blockDim.x) {
    if (NT) __builtin_nontemporal_store (v, dst + i); else dst[i] = v;
  }
}

__global__ void __launch_bounds__ (256) k_read (const u32x4 *src, size_t n16, uint32_t *sink)
{
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n16; i += (size_t) gridDim.x * blockDim.x) {
    u32x4 v = src[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}

template <bool NT>
__global__ void __launch_bounds__ (256) k_copy (u32x4 *dst, const u32x4 *src, size_t n16)
{
  for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n16; i += (size_t) gridDim.x * blockDim.x) {
    u32x4 v = src[i];
    if (NT) __builtin_nontemporal_store (v, dst + i); else dst[i] = v;
  }
}

/* 1 dword read -> 4 dwords written per lane: same shape as the demosaic (1 KiB contiguous per wave store) */
template <bool NT>
__global__ void __launch_bounds__ (256) k_mix14 (u32x4 *dst, const uint32_t *src, size_t n4)
{
  for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n4; i += (size_t) gridDim.x * blockDim.x) {
    uint32_t c = src[i];
    u32x4 v = { c, c >> 8, c >> 16, c >> 24 };
    if (NT) __builtin_nontemporal_store (v, dst + i); else dst[i] = v;
  }
}

template <typename F>
static double time_ms (F launch, int reps)
{
  hipEvent_t a, b;
  CK (hipEventCreate (&a)); CK (hipEventCreate (&b));
  for (int i = 0; i < 3; i++) launch ();
  CK (hipEventRecord (a, 0));
  for (int i = 0; i < reps; i++) launch ();
  CK (hipEventRecord (b, 0));
  CK (hipEventSynchronize (b));
  float ms; CK (hipEventElapsedTime (&ms, a, b));
  return ms / reps;
}

int main (int argc, char **argv)
{
  const size_t out_bytes = (size_t) 3840 * 2160 * 4 * 64;     /* the 4K x 64 batch output: 2.12 GB */
  const size_t in_bytes = out_bytes / 4;
  const int reps = argc > 1 ? atoi (argv[1]) : 20;
  const int only_grid = argc > 2 ? atoi (argv[2]) : 0;
  const char *only = argc > 3 ? argv[3] : NULL;       /* one kernel only: read, fill, fill_nt, copy, ... */
  auto want = [&] (const char *name) { return only == NULL || strcmp (only, name) == 0; };
  uint8_t *a, *b; uint32_t *sink;
  CK (hipMalloc (&a, out_bytes)); CK (hipMalloc (&b, out_bytes)); CK (hipMalloc (&sink, 4));
  CK (hipMemset (a, 1, out_bytes)); CK (hipMemset (b, 2, out_bytes));
  const int grids[] = { 2048, 8192, 32768 };
  for (int g : grids) {
    if (only_grid && g != only_grid)
      continue;
    dim3 grid (g), blk (256);
    double t;
    if (want ("read")) {
      t = time_ms ([&] { hipLaunchKernelGGL (k_read, grid, blk, 0, 0, (const u32x4 *) a, out_bytes / 16, sink); }, reps);
      printf ("grid %6d  read        %8.1f GB/s\n", g, out_bytes / t / 1e6);
    }
    if (want ("fill")) {
      t = time_ms ([&] { hipLaunchKernelGGL (k_fill<false>, grid, blk, 0, 0, (u32x4 *) a, out_bytes / 16); }, reps);
      printf ("grid %6d  fill        %8.1f GB/s\n", g, out_bytes / t / 1e6);
    }
    if (want ("fill_nt")) {
      t = time_ms ([&] { hipLaunchKernelGGL (k_fill<true>, grid, blk, 0, 0, (u32x4 *) a, out_bytes / 16); }, reps);
      printf ("grid %6d  fill_nt     %8.1f GB/s\n", g, out_bytes / t / 1e6);
    }
    if (want ("copy")) {
      t = time_ms ([&] { hipLaunchKernelGGL (k_copy<false>, grid, blk, 0, 0, (u32x4 *) b, (const u32x4 *) a, out_bytes / 16); }, reps);
      printf ("grid %6d  copy        %8.1f GB/s (r+w)\n", g, 2.0 * out_bytes / t / 1e6);
    }
    if (want ("copy_nt")) {
      t = time_ms ([&] { hipLaunchKernelGGL (k_copy<true>, grid, blk, 0, 0, (u32x4 *) b, (const u32x4 *) a, out_bytes / 16); }, reps);
      printf ("grid %6d  copy_nt     %8.1f GB/s (r+w)\n", g, 2.0 * out_bytes / t / 1e6);
    }
    if (want ("mix1r4w")) {
      t = time_ms ([&] { hipLaunchKernelGGL (k_mix14<false>, grid, blk, 0, 0, (u32x4 *) b, (const uint32_t *) a, in_bytes / 4); }, reps);
      printf ("grid %6d  mix1r4w     %8.1f GB/s (r+w)   <- bayer2rgb traffic shape\n", g, (in_bytes + out_bytes) / t / 1e6);
    }
    if (want ("mix1r4w_nt")) {
      t = time_ms ([&] { hipLaunchKernelGGL (k_mix14<true>, grid, blk, 0, 0, (u32x4 *) b, (const uint32_t *) a, in_bytes / 4); }, reps);
      printf ("grid %6d  mix1r4w_nt  %8.1f GB/s (r+w)\n", g, (in_bytes + out_bytes) / t / 1e6);
    }
  }
  hipDeviceProp_t p; CK (hipGetDeviceProperties (&p, 0));
  printf ("device %s  CUs %d  memClk %d kHz  busWidth %d  -> %.0f GB/s nominal\n", p.name, p.multiProcessorCount,
      p.memoryClockRate, p.memoryBusWidth, 2.0 * p.memoryClockRate * 1e3 * p.memoryBusWidth / 8 / 1e9);
  return 0;
}
