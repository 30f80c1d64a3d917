// pmc_calib.hip — known-byte-count streams in the bulk kernel's access shape (8 B per lane,
// 512 B per wave instruction), to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
// (MI355X_MICROARCH.md §HBM: FETCH_SIZE reads 1/2 of a 16 B/lane stream; other widths are
// uncalibrated).  k_copy8: 1 read + 1 write of N doubles.  k_copy8_shift: the same with the
// read misaligned by one element, like the pull of a c_x = +-1 direction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void k_copy8(const double* __restrict__ a, double* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) b[i] = a[i];
}
__global__ void k_copy8_shift(const double* __restrict__ a, double* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) b[i] = a[i + 1];
}

int main(int argc, char** argv) {
  size_t n = (size_t)1 << 30;  // 8 GiB per array: far beyond the 256 MiB Infinity Cache
  if (argc > 1) n = strtoull(argv[1], nullptr, 10);
  double *a, *b;
  if (hipMalloc(&a, (n + 1) * 8) != hipSuccess || hipMalloc(&b, n * 8) != hipSuccess) return 1;
  hipMemset(a, 0, (n + 1) * 8);
  hipMemset(b, 0, n * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_copy8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a, b, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("k_copy8       %zu doubles: %.3f ms, %.1f GB/s (read+write)\n", n, ms, 16.0 * n / ms / 1e6);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_copy8_shift, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a, b, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("k_copy8_shift %zu doubles: %.3f ms, %.1f GB/s (read+write)\n", n, ms, 16.0 * n / ms / 1e6);
  }
  hipFree(a);
  hipFree(b);
  return 0;
}
