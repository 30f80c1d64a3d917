// diag.hip — on-device diagnostics.
//
// Replaces the reference's every-50-steps pattern "copy three full fields to the host, loop on
// one core" (main.cu:211-222) by reductions on the device:
//   double current(c, cn, ez)   LBM.cu:2674-2710  ->  k_wall_current + k_sum_partials
//   record_umax(...)            LBM.cu:2712-2753  ->  k_max_uz + k_max_partials
// Wave64 reductions use DPP/permute shuffles (__shfl_down), one LDS slot per wave, one partial
// per workgroup, and a second single-workgroup pass in fixed order: no atomics, so the result
// is deterministic run to run.
#include "ekpnp_internal.h"

namespace ekpnp {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return v;
}

template <bool IS_MAX>
__device__ __forceinline__ double block_reduce(double v, double* lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  v = IS_MAX ? wave_max(v) : wave_sum(v);
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  double r = IS_MAX ? -1.0e300 : 0.0;
  if (wave == 0) {
    r = lane < nw ? lds[lane] : (IS_MAX ? -1.0e300 : 0.0);
    r = IS_MAX ? wave_max(r) : wave_sum(r);
  }
  return r;  // valid in thread 0
}

// terms of LBM.cu:2704-2706 with the wall extrapolation of LBM.cu:2689-2690 applied on the fly:
// (2 c(NZ-2) - c(NZ-3)  -  (2 cn(NZ-2) - cn(NZ-3))) * Ez(NZ-1), summed over the top plane
__global__ void __launch_bounds__(256) k_wall_current(const double* __restrict__ c, const double* __restrict__ cn,
                                                      const double* __restrict__ ez, long long plane, int nzl, double* __restrict__ partial) {
  __shared__ double lds[4];
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (long long)gridDim.x * blockDim.x) {
    const long long t = (long long)(nzl - 1) * plane + i;
    const double ce = 2.0 * c[t - plane] - c[t - 2 * plane];
    const double cne = 2.0 * cn[t - plane] - cn[t - 2 * plane];
    acc += (ce - cne) * ez[t];
  }
  const double r = block_reduce<false>(acc, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

__global__ void __launch_bounds__(256) k_max_uz(const double* __restrict__ uz, long long n, double* __restrict__ partial) {
  __shared__ double lds[4];
  double m = 0.0;  // umax starts at 0 (LBM.cu:2718)
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) m = fmax(m, uz[i]);
  const double r = block_reduce<true>(m, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// max |a - b| (Picard residual of the Poisson-Boltzmann start)
__global__ void __launch_bounds__(256) k_max_abs_diff(const double* __restrict__ p, const double* __restrict__ q, long long n,
                                                      double* __restrict__ partial) {
  __shared__ double lds[4];
  double m = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) m = fmax(m, fabs(p[i] - q[i]));
  const double r = block_reduce<true>(m, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

template <bool IS_MAX>
__global__ void __launch_bounds__(256) k_final(const double* __restrict__ partial, int n, double* __restrict__ out) {
  __shared__ double lds[4];
  double a = IS_MAX ? 0.0 : 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) a = IS_MAX ? fmax(a, partial[i]) : a + partial[i];
  const double r = block_reduce<IS_MAX>(a, lds);
  if (threadIdx.x == 0) out[0] = r;
}

constexpr int DIAG_BLOCKS = 1024;

void launch_current(Ctx& c, double* scratch /* DIAG_BLOCKS + 1 doubles */) {
  const int nb = (int)((c.plane + 255) / 256 < DIAG_BLOCKS ? (c.plane + 255) / 256 : DIAG_BLOCKS);
  hipLaunchKernelGGL(k_wall_current, dim3(nb), dim3(256), 0, c.stream, c.fld[EKPNP_C], c.fld[EKPNP_CN], c.fld[EKPNP_EZ], (long long)c.plane,
                     c.nzl, scratch);
  note_launch(c, "k_wall_current");
  hipLaunchKernelGGL((k_final<false>), dim3(1), dim3(256), 0, c.stream, scratch, nb, scratch + DIAG_BLOCKS);
  note_launch(c, "k_final<false>");
}

void launch_umax(Ctx& c, double* scratch) {
  const long long n = (long long)c.nloc;
  const int nb = (int)((n + 255) / 256 < DIAG_BLOCKS ? (n + 255) / 256 : DIAG_BLOCKS);
  hipLaunchKernelGGL(k_max_uz, dim3(nb), dim3(256), 0, c.stream, c.fld[EKPNP_UZ], n, scratch);
  note_launch(c, "k_max_uz");
  hipLaunchKernelGGL((k_final<true>), dim3(1), dim3(256), 0, c.stream, scratch, nb, scratch + DIAG_BLOCKS);
  note_launch(c, "k_final<true>");
}

void launch_max_abs_diff(Ctx& c, const double* p, const double* q, double* scratch) {
  const long long n = (long long)c.nloc;
  const int nb = (int)((n + 255) / 256 < DIAG_BLOCKS ? (n + 255) / 256 : DIAG_BLOCKS);
  hipLaunchKernelGGL(k_max_abs_diff, dim3(nb), dim3(256), 0, c.stream, p, q, n, scratch);
  note_launch(c, "k_max_abs_diff");
  hipLaunchKernelGGL((k_final<true>), dim3(1), dim3(256), 0, c.stream, scratch, nb, scratch + DIAG_BLOCKS);
  note_launch(c, "k_final<true>");
}

// plain contiguous copy, 16 bytes per lane (the shape tools/stream_probe.hip calls "copy 16B/lane")
__global__ void k_copy16(const double2* __restrict__ src, double2* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

void launch_copy16(Ctx& c, const void* src, void* dst, size_t bytes) {
  const size_t n = bytes / sizeof(double2);
  hipLaunchKernelGGL(k_copy16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.stream, (const double2*)src, (double2*)dst, n);
  note_launch(c, "k_copy16");
}

}  // namespace ekpnp
