// mall_probe.hip — does the 256 MiB Infinity Cache pay for the Poisson solve's middle passes?
//
// The single-context solve runs y-forward, z solve, y-inverse as three passes over the whole 1 GiB half spectrum
// (3 x (8 R + 8 W) B/node of HBM traffic).  All three work on kx COLUMNS (every ky, every z of a kx range): taken
// block by block - the three passes of one kx block back to back - a block of <= ~64 MiB could stay on-die between
// its passes.  This probe times that access shape with a kernel that only moves bytes (p[i] = p[i] * a + b, 16 B
// per lane, in place):
//   A  three launches over the whole buffer                                       (today's pass structure)
//   B  contiguous chunks of S MiB, three launches per chunk                       (upper bound of the gain)
//   C  the real footprint: the buffer as rows of 257 double2 (4112 B), a chunk = W adjacent columns of EVERY row,
//      three launches per chunk                                                   (what a kx block looks like)
// Build: hipcc --offload-arch=gfx950 -O3 -o mall_probe tools/mall_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(256) k_rmw(double2* __restrict__ p, size_t n, double a, double b) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double2 v = p[i];
  v.x = v.x * a + b;
  v.y = v.y * a + b;
  p[i] = v;
}

// rows of `rowlen` double2; columns [c0, c0 + w) of rows [0, rows): thread -> (row, col), w adjacent lanes share a row
__global__ void __launch_bounds__(256) k_rmw_cols(double2* __restrict__ p, size_t rows, int rowlen, int c0, int w, double a, double b) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t row = t / (size_t)w;
  const int col = (int)(t - row * (size_t)w);
  if (row >= rows) return;
  double2* q = p + row * (size_t)rowlen + c0 + col;
  double2 v = *q;
  v.x = v.x * a + b;
  v.y = v.y * a + b;
  *q = v;
}

static float timed(hipEvent_t e0, hipEvent_t e1) {
  float ms;
  hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main(int argc, char** argv) {
  const int rowlen = 257;
  const size_t rows = (size_t)512 * 510;      // ny x unknown planes of cfg3
  const size_t n = rows * rowlen;             // double2 elements: 1.07 GB
  const int passes = argc > 1 ? atoi(argv[1]) : 3;
  double2* p;
  if (hipMalloc(&p, n * sizeof(double2)) != hipSuccess) return 1;
  hipMemset(p, 0, n * sizeof(double2));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const double gb = (double)n * 16 * 2 * passes / 1e9;  // bytes the passes move, read + write
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    for (int k = 0; k < passes; ++k) hipLaunchKernelGGL(k_rmw, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, p, n, 1.0000001, 1e-9);
    hipEventRecord(e1);
    float ms = timed(e0, e1);
    printf("A whole buffer            %d passes: %.3f ms  (%.0f GB/s of pass bytes)\n", passes, ms, gb / ms * 1e3);
    for (int smib : {8, 16, 32, 64, 96, 128, 192, 256}) {
      const size_t ce = (size_t)smib << 16;  // double2 per chunk
      hipEventRecord(e0);
      for (size_t o = 0; o < n; o += ce) {
        const size_t m = n - o < ce ? n - o : ce;
        for (int k = 0; k < passes; ++k) hipLaunchKernelGGL(k_rmw, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, 0, p + o, m, 1.0000001, 1e-9);
      }
      hipEventRecord(e1);
      ms = timed(e0, e1);
      printf("B contiguous %3d MiB chunks %d passes: %.3f ms  (%.0f GB/s)\n", smib, passes, ms, gb / ms * 1e3);
    }
    for (int w : {4, 8, 16, 32, 64}) {
      hipEventRecord(e0);
      for (int c0 = 0; c0 < rowlen; c0 += w) {
        const int ww = rowlen - c0 < w ? rowlen - c0 : w;
        const size_t thr = rows * (size_t)ww;
        for (int k = 0; k < passes; ++k)
          hipLaunchKernelGGL(k_rmw_cols, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, 0, p, rows, rowlen, c0, ww, 1.0000001, 1e-9);
      }
      hipEventRecord(e1);
      ms = timed(e0, e1);
      printf("C column blocks of %2d (%5.1f MiB) %d passes: %.3f ms  (%.0f GB/s)\n", w, rows * (double)w * 16 / 1048576.0, passes, ms, gb / ms * 1e3);
    }
    // one pass over column blocks (no reuse): what the block footprint alone costs
    for (int w : {16, 32}) {
      hipEventRecord(e0);
      for (int c0 = 0; c0 < rowlen; c0 += w) {
        const int ww = rowlen - c0 < w ? rowlen - c0 : w;
        const size_t thr = rows * (size_t)ww;
        hipLaunchKernelGGL(k_rmw_cols, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, 0, p, rows, rowlen, c0, ww, 1.0000001, 1e-9);
      }
      hipEventRecord(e1);
      ms = timed(e0, e1);
      printf("C1 column blocks of %2d, ONE pass: %.3f ms  (%.0f GB/s)\n", w, ms, (double)n * 32 / 1e6 / ms);
    }
  }
  hipFree(p);
  return 0;
}
