// fft_plane.h — the 2-D transforms of fast_Poisson hand-written for gfx950: the y pass (strided columns, 512 or 1024
// rows) and the x pass (contiguous rows of 1024 real values <-> 513 complex ones).
//
// Replaces part of cufftExecZ2Z's work on a plane (poisson.cu:86,92; the reference transforms the odd extension in
// 3-D, this library transforms the NZ-2 interior planes in 2-D and solves z by a tridiagonal system).
// rocFFT's batched 2-D plan has a strided-column kernel for 512-long columns (2 + 2 kernels per solve) but not for
// 1024-long ones: on cfg5's 1024 x 1024 planes it transposes instead, 4 + 4 kernels, 3.14 ms per solve on a
// 1024 x 1024 x 128 slab (profiles/r03_cfg5_rank_shape_slab_kernel_stats.csv).  rocFFT's batched 1-D row transforms are
// two kernels per direction as well (0.87 / 0.98 ms).  Here a plane's transform is TWO kernels per direction: one for the
// rows (real <-> half spectrum, the even/odd split of a real transform fused in), one for the columns.
//
//   A workgroup owns COLS adjacent kx columns of one plane - COLS x 16 bytes of every row - with 64 threads per column.
//   Stockham autosort through one LDS image [row][column]; stage 0 reads the rows straight from global memory, the
//   last stage writes them straight back (in place: a workgroup reads all of its elements before it writes any).
//     512 rows:  radix 8 x 8 x 8,  COLS = 8 (one full 128-byte line per row), 72 KB of LDS
//     1024 rows: radix 16 x 8 x 8, COLS = 4 (64 bytes per row; the workgroup that owns the other half of the line runs
//                on the same XCD right behind it, so the line is fetched from HBM once), 80 KB of LDS
//   Twiddles exp(-2 pi i k / N) from a table the host computed in long double, kept in LDS.
// Unnormalised in both directions, like cuFFT / rocFFT.  The spectrum row pitch nxh is a multiple of 8 (capi.hip); the
// padding columns beyond NX/2 are transformed along (never read by anyone).
//
// Planes of 512 rows / 512 columns are served too (256-point row transform, 512-point column transform) and are the
// default there since round 4 (poisson.hip: plane_fft_setup): kernel against kernel the gain over rocFFT's 2 + 2 kernels is
// small (0.88 + 0.84 against 0.90 + 0.86 ms on 512 x 512 x 512, profiles/r03_own_plane_fft_probe.log), inside the full cfg3
// step the Poisson phase is 2.12 against 2.25 ms (profiles/r04_ab_own_fft_512.log).  EKPNP_OWN_FFT=0 keeps rocFFT.
//
// Header-only so that tools/fft_y_probe.hip times exactly the code the library runs.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>

namespace ekpnp {

template <int SIGN>
__device__ __forceinline__ double2 fy_mul_i(double2 a) {  // a * exp(SIGN i pi/2)
  return SIGN > 0 ? make_double2(-a.y, a.x) : make_double2(a.y, -a.x);
}
template <int SIGN>
__device__ __forceinline__ double2 fy_mul_w8(double2 a) {  // a * exp(SIGN i pi/4)
  constexpr double h = 0.70710678118654752440;
  return SIGN > 0 ? make_double2((a.x - a.y) * h, (a.x + a.y) * h) : make_double2((a.x + a.y) * h, (a.y - a.x) * h);
}
// a * w for the forward transform (w = exp(-2 pi i ...)), a * conj(w) for the inverse
template <int SIGN>
__device__ __forceinline__ double2 fy_mul_tw(double2 a, double2 w) {
  return SIGN > 0 ? make_double2(a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y) : make_double2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
}
__device__ __forceinline__ void fy_bfly(double2& a, double2& b) {
  const double2 t = a;
  a = make_double2(t.x + b.x, t.y + b.y);
  b = make_double2(t.x - b.x, t.y - b.y);
}
// 8-point DFT in registers; output X[k] is left in v[fy_rev3(k)]
template <int SIGN>
__device__ __forceinline__ void fy_dft8(double2 (&v)[8]) {
  fy_bfly(v[0], v[4]); fy_bfly(v[1], v[5]); fy_bfly(v[2], v[6]); fy_bfly(v[3], v[7]);
  v[5] = fy_mul_w8<SIGN>(v[5]);
  v[6] = fy_mul_i<SIGN>(v[6]);
  v[7] = fy_mul_i<SIGN>(fy_mul_w8<SIGN>(v[7]));
  fy_bfly(v[0], v[2]); fy_bfly(v[1], v[3]); fy_bfly(v[4], v[6]); fy_bfly(v[5], v[7]);
  v[3] = fy_mul_i<SIGN>(v[3]);
  v[7] = fy_mul_i<SIGN>(v[7]);
  fy_bfly(v[0], v[1]); fy_bfly(v[2], v[3]); fy_bfly(v[4], v[5]); fy_bfly(v[6], v[7]);
}
__host__ __device__ constexpr int fy_rev3(int i) { return ((i & 1) << 2) | (i & 2) | ((i >> 2) & 1); }

// 16-point DFT of v[0..15] (natural order in, natural order out): two 8-point DFTs of the even and the odd inputs
// and one layer of butterflies with exp(-+2 pi i k / 16)
template <int SIGN>
__device__ __forceinline__ void fy_dft16(double2 (&v)[16]) {
  double2 e[8], o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
  fy_dft8<SIGN>(e);
  fy_dft8<SIGN>(o);
  // forward table exp(-2 pi i k / 16), k = 0..7
  constexpr double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173, h = 0.70710678118654752440;
  constexpr double wc[8] = {1.0, c1, h, s1, 0.0, -s1, -h, -c1};
  constexpr double ws[8] = {0.0, -s1, -h, -c1, -1.0, -c1, -h, -s1};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const double2 a = e[fy_rev3(k)];
    const double2 b = fy_mul_tw<SIGN>(o[fy_rev3(k)], make_double2(wc[k], ws[k]));
    v[k] = make_double2(a.x + b.x, a.y + b.y);
    v[k + 8] = make_double2(a.x - b.x, a.y - b.y);
  }
}

constexpr size_t fy_lds_bytes(int n, int cols) { return (size_t)(n * cols + n) * sizeof(double2); }

// ---- 512 rows: radix 8 x 8 x 8, 8 columns per workgroup ----------------------------------------------------------
template <int SIGN>
__global__ void __launch_bounds__(512) k_fft_y512(double2* __restrict__ spec, const double2* __restrict__ tw, int nxh, long long plane_stride) {
  constexpr int N = 512, COLS = 8;
  extern __shared__ double2 fy_lds[];
  double2* buf = fy_lds;            // [512 rows][8 columns]
  double2* w = fy_lds + N * COLS;   // exp(-2 pi i k / 512)
  const int c = threadIdx.x & 7, t = threadIdx.x >> 3;  // column, butterfly 0..63
  double2* base = spec + (long long)blockIdx.y * plane_stride + (long long)blockIdx.x * COLS + c;
  double2 v[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = base[(long long)(t + 64 * r) * nxh];
  w[threadIdx.x] = tw[threadIdx.x];
  // stage 0 (sub-transform length 1): no twiddles; X[k] -> row 8 t + k
  fy_dft8<SIGN>(v);
#pragma unroll
  for (int i = 0; i < 8; ++i) buf[(8 * t + fy_rev3(i)) * COLS + c] = v[i];
  __syncthreads();
  // stage 1 (length 8): twiddle exp(-+2 pi i (t mod 8) r / 64); X[k] -> row (t / 8) 64 + (t mod 8) + 8 k
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = buf[(t + 64 * r) * COLS + c];
  {
    const int k = t & 7;
#pragma unroll
    for (int r = 1; r < 8; ++r) v[r] = fy_mul_tw<SIGN>(v[r], w[k * r * 8]);
  }
  fy_dft8<SIGN>(v);
  __syncthreads();  // every thread holds its inputs: the image may be overwritten
  {
    const int j0 = (t >> 3) * 64 + (t & 7);
#pragma unroll
    for (int i = 0; i < 8; ++i) buf[(j0 + 8 * fy_rev3(i)) * COLS + c] = v[i];
  }
  __syncthreads();
  // stage 2 (length 64): twiddle exp(-+2 pi i t r / 512); X[k] -> row t + 64 k
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = buf[(t + 64 * r) * COLS + c];
#pragma unroll
  for (int r = 1; r < 8; ++r) v[r] = fy_mul_tw<SIGN>(v[r], w[t * r]);
  fy_dft8<SIGN>(v);
#pragma unroll
  for (int i = 0; i < 8; ++i) base[(long long)(t + 64 * fy_rev3(i)) * nxh] = v[i];
}

// LDS row of the 1024-row image: a row is only 64 bytes (a quarter of the 64 banks), so rows whose index differs by a
// multiple of 4 share their banks - and stage 0 writes rows 16 t + k for neighbouring t in the same instruction.  The low
// two bits of the row are XOR-ed with bits 4-5 (a bijection inside every aligned group of 4 rows).
#ifndef EKPNP_FFTY_NO_SWIZZLE
__device__ __forceinline__ int fy_row(int r) { return r ^ ((r >> 4) & 3); }
#else
__device__ __forceinline__ int fy_row(int r) { return r; }
#endif

// ---- 512 rows, 4 columns per workgroup (round 3): radix 8 x 8 x 8 like k_fft_y512, but 64 bytes per row and workgroup,
// 40 KB of LDS (4 workgroups per CU instead of 2), the two halves of a line paired on one XCD like k_fft_y1024
__device__ __forceinline__ int fy_row8(int r) { return r ^ ((r >> 3) & 3); }  // stage 0 writes rows 8 t + k for neighbouring t
template <int SIGN>
__global__ void __launch_bounds__(256) k_fft_y512c4(double2* __restrict__ spec, const double2* __restrict__ tw, int nxh, long long plane_stride, int npairs, int group0, int groups) {
  constexpr int N = 512, COLS = 4;
  extern __shared__ double2 fy_lds[];
  double2* buf = fy_lds;            // [512 rows][4 columns]
  double2* w = fy_lds + N * COLS;   // exp(-2 pi i k / 512)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int pair = (slot >> 1) * 8 + xcd, half = slot & 1;
  if (pair >= npairs) return;  // uniform over the workgroup
  const int plane = pair / groups, group = group0 + pair - plane * groups;  // groups [group0, group0 + groups) of every plane
  const int c = threadIdx.x & 3, t = threadIdx.x >> 2;  // column, butterfly 0..63
  double2* base = spec + (long long)plane * plane_stride + (long long)group * 8 + half * COLS + c;
  double2 v[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = base[(long long)(t + 64 * r) * nxh];
#pragma unroll
  for (int i = 0; i < 2; ++i) w[threadIdx.x + 256 * i] = tw[threadIdx.x + 256 * i];
  fy_dft8<SIGN>(v);
#pragma unroll
  for (int i = 0; i < 8; ++i) buf[fy_row8(8 * t + fy_rev3(i)) * COLS + c] = v[i];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = buf[fy_row8(t + 64 * r) * COLS + c];
  {
    const int k = t & 7;
#pragma unroll
    for (int r = 1; r < 8; ++r) v[r] = fy_mul_tw<SIGN>(v[r], w[k * r * 8]);
  }
  fy_dft8<SIGN>(v);
  __syncthreads();
  {
    const int j0 = (t >> 3) * 64 + (t & 7);
#pragma unroll
    for (int i = 0; i < 8; ++i) buf[fy_row8(j0 + 8 * fy_rev3(i)) * COLS + c] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = buf[fy_row8(t + 64 * r) * COLS + c];
#pragma unroll
  for (int r = 1; r < 8; ++r) v[r] = fy_mul_tw<SIGN>(v[r], w[t * r]);
  fy_dft8<SIGN>(v);
#pragma unroll
  for (int i = 0; i < 8; ++i) base[(long long)(t + 64 * fy_rev3(i)) * nxh] = v[i];
}

// ---- 1024 rows: radix 16 x 8 x 8, 4 columns per workgroup -----------------------------------------------------------
// Workgroup -> (plane, 8-column group, half): workgroups are dealt round-robin to the 8 XCDs (bid % 8), and the two
// halves of one 128-byte line group are consecutive slots of ONE XCD, so the second one finds the lines in that L2.
template <int SIGN>
__global__ void __launch_bounds__(256) k_fft_y1024(double2* __restrict__ spec, const double2* __restrict__ tw, int nxh, long long plane_stride, int npairs, int group0, int groups) {
  constexpr int N = 1024, COLS = 4;
  extern __shared__ double2 fy_lds[];
  double2* buf = fy_lds;            // [1024 rows][4 columns]
  double2* w = fy_lds + N * COLS;   // exp(-2 pi i k / 1024)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int pair = (slot >> 1) * 8 + xcd, half = slot & 1;
  if (pair >= npairs) return;  // uniform over the workgroup
  const int plane = pair / groups, group = group0 + pair - plane * groups;  // groups [group0, group0 + groups) of every plane
  const int c = threadIdx.x & 3, t = threadIdx.x >> 2;  // column, butterfly 0..63
  double2* base = spec + (long long)plane * plane_stride + (long long)group * 8 + half * COLS + c;
  double2 v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = base[(long long)(t + 64 * r) * nxh];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[threadIdx.x + 256 * i] = tw[threadIdx.x + 256 * i];
  // stage 0 (sub-transform length 1, radix 16): inputs x[t + 64 r]; X_t[k] -> row 16 t + k
  fy_dft16<SIGN>(v);
#pragma unroll
  for (int k = 0; k < 16; ++k) buf[fy_row(16 * t + k) * COLS + c] = v[k];
  __syncthreads();
  // stage 1 (length 16 -> 128, radix 8): butterfly b = 16 q + k (q = 0..7, k = 0..15) combines the sub-sequences
  // q + 8 r at frequency k: inputs rows b + 128 r, twiddle exp(-+2 pi i k r / 128), X[k + 16 k2] -> row 128 q + k + 16 k2.
  // Two butterflies per thread (b = t and t + 64); all inputs are read before any output is written.
  double2 a0[8], a1[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    a0[r] = buf[fy_row(t + 128 * r) * COLS + c];
    a1[r] = buf[fy_row(t + 64 + 128 * r) * COLS + c];
  }
  {
    const int k0 = t & 15, k1 = (t + 64) & 15;  // (equal: 64 is a multiple of 16)
#pragma unroll
    for (int r = 1; r < 8; ++r) {
      a0[r] = fy_mul_tw<SIGN>(a0[r], w[k0 * r * 8]);
      a1[r] = fy_mul_tw<SIGN>(a1[r], w[k1 * r * 8]);
    }
  }
  fy_dft8<SIGN>(a0);
  fy_dft8<SIGN>(a1);
  __syncthreads();
  {
    const int b0 = t, b1 = t + 64;
    const int o0 = (b0 >> 4) * 128 + (b0 & 15), o1 = (b1 >> 4) * 128 + (b1 & 15);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      buf[fy_row(o0 + 16 * fy_rev3(i)) * COLS + c] = a0[i];
      buf[fy_row(o1 + 16 * fy_rev3(i)) * COLS + c] = a1[i];
    }
  }
  __syncthreads();
  // stage 2 (length 128 -> 1024, radix 8): butterfly f = 0..127: inputs rows f + 128 r, twiddle exp(-+2 pi i f r / 1024),
  // X[f + 128 k2] -> row f + 128 k2 of global memory
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    a0[r] = buf[fy_row(t + 128 * r) * COLS + c];
    a1[r] = buf[fy_row(t + 64 + 128 * r) * COLS + c];
  }
#pragma unroll
  for (int r = 1; r < 8; ++r) {
    a0[r] = fy_mul_tw<SIGN>(a0[r], w[t * r]);
    a1[r] = fy_mul_tw<SIGN>(a1[r], w[(t + 64) * r]);
  }
  fy_dft8<SIGN>(a0);
  fy_dft8<SIGN>(a1);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    base[(long long)(t + 128 * fy_rev3(i)) * nxh] = a0[i];
    base[(long long)(t + 64 + 128 * fy_rev3(i)) * nxh] = a1[i];
  }
}

inline bool fft_y_supported(int ny, int nxh) { return (ny == 512 || ny == 1024) && nxh % 8 == 0; }

// host table exp(-2 pi i k / ny), k = 0..ny-1, computed in long double, exact where the value is exact
inline void fft_y_twiddles(int ny, double2* h) {
  for (int k = 0; k < ny; ++k) {
    const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)ny;
    h[k] = make_double2((double)cosl(a), (double)sinl(a));
  }
  h[0] = make_double2(1.0, 0.0);
  h[ny / 4] = make_double2(0.0, -1.0);
  h[ny / 2] = make_double2(-1.0, 0.0);
  h[3 * ny / 4] = make_double2(0.0, 1.0);
}

// per-device function attributes (more than 64 KB of dynamic LDS); false if the device refuses
inline bool fft_y_prepare(int ny) {
  hipError_t e = hipSuccess;
  if (ny == 512) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fft_y512<-1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fy_lds_bytes(512, 8));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fft_y512<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fy_lds_bytes(512, 8));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fft_y512c4<-1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fy_lds_bytes(512, 4));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fft_y512c4<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fy_lds_bytes(512, 4));
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fft_y1024<-1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fy_lds_bytes(1024, 4));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fft_y1024<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fy_lds_bytes(1024, 4));
  }
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  return true;
}

// in place on `nplanes` planes [ny][nxh] of the half spectrum; sign -1: forward, +1: inverse
// group0, ngroups: the 8-column groups [group0, group0 + ngroups) of every plane only (a mode block of the slab solve);
// ngroups < 0: all nxh / 8 of them
inline void fft_y_launch(double2* spec, const double2* tw, int ny, int nxh, int nplanes, int sign, hipStream_t stream, int group0 = 0, int ngroups = -1) {
  if (nplanes <= 0) return;
  if (ngroups < 0) { group0 = 0; ngroups = nxh / 8; }
  if (ngroups == 0) return;
  const long long ps = (long long)ny * nxh;
  if (ny == 512) {
    // EKPNP_FFTY512_COLS=8: the 8-column kernel of round 2 (the A/B partner)
    static const bool cols8 = std::getenv("EKPNP_FFTY512_COLS") != nullptr && std::atoi(std::getenv("EKPNP_FFTY512_COLS")) == 8;
    if (cols8 && group0 == 0 && ngroups == nxh / 8) {
      const dim3 grid(nxh / 8, nplanes);
      if (sign < 0)
        hipLaunchKernelGGL(k_fft_y512<-1>, grid, dim3(512), fy_lds_bytes(512, 8), stream, spec, tw, nxh, ps);
      else
        hipLaunchKernelGGL(k_fft_y512<1>, grid, dim3(512), fy_lds_bytes(512, 8), stream, spec, tw, nxh, ps);
    } else {
      const int npairs = nplanes * ngroups;
      const unsigned blocks = (unsigned)((npairs + 7) / 8) * 8 * 2;
      if (sign < 0)
        hipLaunchKernelGGL(k_fft_y512c4<-1>, dim3(blocks), dim3(256), fy_lds_bytes(512, 4), stream, spec, tw, nxh, ps, npairs, group0, ngroups);
      else
        hipLaunchKernelGGL(k_fft_y512c4<1>, dim3(blocks), dim3(256), fy_lds_bytes(512, 4), stream, spec, tw, nxh, ps, npairs, group0, ngroups);
    }
  } else {
    const int npairs = nplanes * ngroups;
    const unsigned blocks = (unsigned)((npairs + 7) / 8) * 8 * 2;
    if (sign < 0)
      hipLaunchKernelGGL(k_fft_y1024<-1>, dim3(blocks), dim3(256), fy_lds_bytes(1024, 4), stream, spec, tw, nxh, ps, npairs, group0, ngroups);
    else
      hipLaunchKernelGGL(k_fft_y1024<1>, dim3(blocks), dim3(256), fy_lds_bytes(1024, 4), stream, spec, tw, nxh, ps, npairs, group0, ngroups);
  }
}


// ---- x pass: rows of NX real values <-> NX/2 + 1 complex ones (pitch nxh), NX = 1024 or 512 ------------------------
// A real row x[0..NX-1] is read as N = NX/2 complex numbers z[n] = x[2n] + i x[2n+1]; one wavefront per row does the
// N-point complex transform (N = 512: radix 8 x 8 x 8, N = 256: radix 4 x 4 x 4 x 4; constant-geometry Stockham through
// the row's own padded LDS image - one pad element per 8, so that the strided accesses of the stages fall on different
// banks) and the even/odd split of the real transform, w_k = exp(-2 pi i k / NX):
//   forward  Z = FFT_N(z), E[k] = (Z[k] + conj Z[N-k]) / 2, O[k] = (Z[k] - conj Z[N-k]) / 2i:
//            X[k] = E[k] + w_k O[k],  X[N-k] = conj(E[k] - w_k O[k]),  k = 0..N/2
//   inverse  A = X[k], B = conj X[N-k]:  Z[k] = (A + B) + i (A - B) conj(w_k),  Z[N-k] = conj(A + B) + i conj(A - B) w_k,
//            z = IFFT_N(Z) (unnormalised), x[2n] + i x[2n+1] = z[n]
// Lanes hold consecutive elements, so every global access of a wave is one contiguous kilobyte.  FX_ROWS rows per
// workgroup share the twiddle table exp(-2 pi i k / NX) in LDS.
constexpr int FX_ROWS = 4;
__device__ __forceinline__ int fx_pad(int i) { return i + (i >> 3); }  // padded row image: index i -> i + i / 8
constexpr size_t fx_lds_bytes(int nx) { return (size_t)(FX_ROWS * (nx / 2 + nx / 16) + nx) * sizeof(double2); }

// 4-point DFT, natural order in and out
template <int SIGN>
__device__ __forceinline__ void fy_dft4(double2 (&v)[4]) {
  fy_bfly(v[0], v[2]);
  fy_bfly(v[1], v[3]);
  v[3] = fy_mul_i<SIGN>(v[3]);
  const double2 x0 = make_double2(v[0].x + v[1].x, v[0].y + v[1].y), x2 = make_double2(v[0].x - v[1].x, v[0].y - v[1].y);
  const double2 x1 = make_double2(v[2].x + v[3].x, v[2].y + v[3].y), x3 = make_double2(v[2].x - v[3].x, v[2].y - v[3].y);
  v[0] = x0; v[1] = x1; v[2] = x2; v[3] = x3;
}

// N-point complex transform of the wave's row image `img` (natural order in LDS in; result in registers: the value of
// frequency t + 64 f sits in v[slot(f)], slot = fy_rev3 for N = 512, identity for N = 256); wnx = exp(-2 pi i k / NX), NX = 2 N
template <int SIGN, int N>
__device__ __forceinline__ void fx_fft_from_lds(double2* img, const double2* wnx, int t, double2 (&v)[N / 64]) {
  if constexpr (N == 512) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = img[fx_pad(t + 64 * r)];
    __syncthreads();
    fy_dft8<SIGN>(v);
#pragma unroll
    for (int i = 0; i < 8; ++i) img[fx_pad(8 * t + fy_rev3(i))] = v[i];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = img[fx_pad(t + 64 * r)];
    {
      const int k = t & 7;
#pragma unroll
      for (int r = 1; r < 8; ++r) v[r] = fy_mul_tw<SIGN>(v[r], wnx[k * r * 16]);
    }
    fy_dft8<SIGN>(v);
    __syncthreads();
    {
      const int j0 = (t >> 3) * 64 + (t & 7);
#pragma unroll
      for (int i = 0; i < 8; ++i) img[fx_pad(j0 + 8 * fy_rev3(i))] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = img[fx_pad(t + 64 * r)];
#pragma unroll
    for (int r = 1; r < 8; ++r) v[r] = fy_mul_tw<SIGN>(v[r], wnx[t * r * 2]);
    fy_dft8<SIGN>(v);
  } else {
    // N = 256 = 4^4: every stage reads rows t + 64 r, multiplies by exp(-+2 pi i (t mod L) r / (4 L)) and writes the
    // 4-point DFT to rows (t / L) 4 L + (t mod L) + L k; L = 1, 4, 16, 64 (the last stage stays in registers)
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = img[fx_pad(t + 64 * r)];
    __syncthreads();
    fy_dft4<SIGN>(v);
#pragma unroll
    for (int L = 1; L <= 16; L *= 4) {
      const int base = (t / L) * 4 * L + (t % L);
#pragma unroll
      for (int k = 0; k < 4; ++k) img[fx_pad(base + L * k)] = v[k];
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = img[fx_pad(t + 64 * r)];
      __syncthreads();
      const int Ln = 4 * L, j = t % Ln;  // next stage: sub-length Ln, twiddle exp(-+2 pi i j r / (4 Ln)) = wnx[j r 512 / (4 Ln)]
#pragma unroll
      for (int r = 1; r < 4; ++r) v[r] = fy_mul_tw<SIGN>(v[r], wnx[j * r * (128 / Ln)]);
      fy_dft4<SIGN>(v);
    }
  }
}
template <int N>
__device__ __forceinline__ constexpr int fx_slot(int f) { return N == 512 ? fy_rev3(f) : f; }

// forward: real rows [nrows][NX] -> half spectrum rows [nrows][nxh] (NX/2 + 1 of them written)
template <int NX>
__global__ void __launch_bounds__(64 * FX_ROWS) k_fft_x_r2c(const double* __restrict__ in, double2* __restrict__ out, const double2* __restrict__ tw, int nxh, long long nrows) {
  constexpr int N = NX / 2, VPT = N / 64, PITCH = N + N / 8;
  extern __shared__ double2 fx_lds[];
  double2* w = fx_lds + FX_ROWS * PITCH;
  const int t = threadIdx.x & 63, rw = threadIdx.x >> 6;
  double2* img = fx_lds + rw * PITCH;
  const long long row = (long long)blockIdx.x * FX_ROWS + rw;
  const bool live = row < nrows;  // wave-uniform; dead waves still take part in the barriers
  const double2* src = reinterpret_cast<const double2*>(in + (live ? row : 0) * NX);
#pragma unroll
  for (int i = 0; i < NX / 256; ++i) w[threadIdx.x + 256 * i] = tw[threadIdx.x + 256 * i];
#pragma unroll
  for (int r = 0; r < VPT; ++r) img[fx_pad(t + 64 * r)] = src[t + 64 * r];
  __syncthreads();
  double2 v[VPT];
  fx_fft_from_lds<-1, N>(img, w, t, v);
  __syncthreads();
#pragma unroll
  for (int f = 0; f < VPT; ++f) img[fx_pad(t + 64 * f)] = v[fx_slot<N>(f)];  // Z in natural order
  __syncthreads();
  if (!live) return;
  double2* dst = out + row * nxh;
#pragma unroll
  for (int j = 0; j < VPT / 2; ++j) {
    const int k = t + 64 * j;  // 0 .. N/2 - 1
    const double2 zk = img[fx_pad(k)], zn = img[fx_pad((N - k) & (N - 1))];
    const double2 e = make_double2(0.5 * (zk.x + zn.x), 0.5 * (zk.y - zn.y));   // (Z[k] + conj Z[N-k]) / 2
    const double2 o = make_double2(0.5 * (zk.y + zn.y), -0.5 * (zk.x - zn.x));  // (Z[k] - conj Z[N-k]) / 2i
    const double2 a = fy_mul_tw<-1>(o, w[k]);
    dst[k] = make_double2(e.x + a.x, e.y + a.y);
    dst[N - k] = make_double2(e.x - a.x, -(e.y - a.y));
  }
  if (t == 0) {  // k = N/2: X[N/2] = conj Z[N/2]
    const double2 z = img[fx_pad(N / 2)];
    dst[N / 2] = make_double2(z.x, -z.y);
  }
}

// inverse: half spectrum rows [nrows][nxh] -> real rows [nrows][NX], unnormalised
template <int NX>
__global__ void __launch_bounds__(64 * FX_ROWS) k_fft_x_c2r(const double2* __restrict__ in, double* __restrict__ out, const double2* __restrict__ tw, int nxh, long long nrows) {
  constexpr int N = NX / 2, VPT = N / 64, PITCH = N + N / 8;
  extern __shared__ double2 fx_lds[];
  double2* w = fx_lds + FX_ROWS * PITCH;
  const int t = threadIdx.x & 63, rw = threadIdx.x >> 6;
  double2* img = fx_lds + rw * PITCH;
  const long long row = (long long)blockIdx.x * FX_ROWS + rw;
  const bool live = row < nrows;
  const double2* src = in + (live ? row : 0) * nxh;
#pragma unroll
  for (int i = 0; i < NX / 256; ++i) w[threadIdx.x + 256 * i] = tw[threadIdx.x + 256 * i];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < VPT / 2; ++j) {
    const int k = t + 64 * j;  // 0 .. N/2 - 1
    const double2 a = src[k], xb = src[N - k];
    const double2 b = make_double2(xb.x, -xb.y);                       // conj X[N-k]
    const double2 s = make_double2(a.x + b.x, a.y + b.y), d = make_double2(a.x - b.x, a.y - b.y);
    const double2 dw = fy_mul_tw<1>(d, w[k]);                          // (A - B) conj(w_k)
    img[fx_pad(k)] = make_double2(s.x - dw.y, s.y + dw.x);             // (A + B) + i (A - B) conj(w_k)
    if (k != 0) {
      const double2 dc = fy_mul_tw<-1>(make_double2(d.x, -d.y), w[k]);  // conj(A - B) w_k
      img[fx_pad(N - k)] = make_double2(s.x - dc.y, -s.y + dc.x);      // conj(A + B) + i conj(A - B) w_k
    }
  }
  if (t == 0) {  // k = N/2: Z[N/2] = 2 conj X[N/2]
    const double2 a = src[N / 2];
    img[fx_pad(N / 2)] = make_double2(2.0 * a.x, -2.0 * a.y);
  }
  __syncthreads();
  double2 v[VPT];
  fx_fft_from_lds<1, N>(img, w, t, v);
  if (!live) return;
  // the phi array may be caller-bound (ekpnp_bind_field) and then only 8-byte aligned: a 16-byte store through a type
  // that promises no more than that (global_store_dwordx4 itself needs dword alignment only)
  typedef double fx_pair8 __attribute__((ext_vector_type(2), aligned(8)));
  fx_pair8* dst = reinterpret_cast<fx_pair8*>(out + row * NX);
#pragma unroll
  for (int f = 0; f < VPT; ++f) {
    const double2 z = v[fx_slot<N>(f)];
    const fx_pair8 pv = {z.x, z.y};
    dst[t + 64 * f] = pv;
  }
}

inline bool fft_x_supported(int nx) { return nx == 1024 || nx == 512; }
inline bool fft_x_prepare(int nx) {
  hipError_t e;
  if (nx == 1024) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fft_x_r2c<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fx_lds_bytes(1024));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fft_x_c2r<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fx_lds_bytes(1024));
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fft_x_r2c<512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fx_lds_bytes(512));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fft_x_c2r<512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fx_lds_bytes(512));
  }
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  return true;
}
// tw = exp(-2 pi i k / nx), k = 0..nx-1 (fft_y_twiddles(nx, .)); nrows = rows of all planes
inline void fft_x_forward(const double* real, double2* spec, const double2* tw, int nx, int nxh, long long nrows, hipStream_t stream) {
  if (nrows <= 0) return;
  const dim3 grid((unsigned)((nrows + FX_ROWS - 1) / FX_ROWS)), block(64 * FX_ROWS);
  if (nx == 1024)
    hipLaunchKernelGGL(k_fft_x_r2c<1024>, grid, block, fx_lds_bytes(1024), stream, real, spec, tw, nxh, nrows);
  else
    hipLaunchKernelGGL(k_fft_x_r2c<512>, grid, block, fx_lds_bytes(512), stream, real, spec, tw, nxh, nrows);
}
inline void fft_x_inverse(const double2* spec, double* real, const double2* tw, int nx, int nxh, long long nrows, hipStream_t stream) {
  if (nrows <= 0) return;
  const dim3 grid((unsigned)((nrows + FX_ROWS - 1) / FX_ROWS)), block(64 * FX_ROWS);
  if (nx == 1024)
    hipLaunchKernelGGL(k_fft_x_c2r<1024>, grid, block, fx_lds_bytes(1024), stream, spec, real, tw, nxh, nrows);
  else
    hipLaunchKernelGGL(k_fft_x_c2r<512>, grid, block, fx_lds_bytes(512), stream, spec, real, tw, nxh, nrows);
}

}  // namespace ekpnp
