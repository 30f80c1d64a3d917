// fft_y.hip — the y pass of the 2-D transforms of fast_Poisson for planes of 512 rows.
//
// Replaces the second half of cufftExecZ2Z's work on a plane (poisson.cu:86,92; the reference
// transforms the odd extension in 3-D, this library transforms the NZ-2 interior planes in 2-D and
// solves z by a tridiagonal system).  rocFFT's strided column kernel for this shape
// (sbcc, 4 columns = 64 B per workgroup and row) fetches every 128-byte line twice: 3.25 GB moved
// for the 2.16 GB a pass has to move on 512^3 (profiles/r02_cfg3_pmc_summary.json).  Here a
// workgroup owns 8 adjacent kx columns - one full 128-byte line per row - of one plane:
//   512 threads = 8 columns x 64 butterflies; Stockham autosort, radix 8 x 8 x 8;
//   stage 0 reads the rows straight from global memory, stage 2 writes them straight back (in
//   place: a workgroup reads all of its 512 x 8 elements before it writes any);
//   the two exchanges in between go through one 64 KB LDS image [row][column] - every LDS access
//   is 8 lanes on 128 contiguous bytes, the conflict-free shape of ds_read/write_b128;
//   twiddles exp(-+2 pi i k / 512) from a table the host computed in long double, kept in LDS.
// Unnormalised in both directions, like cuFFT / rocFFT.  The spectrum row pitch nxh is a multiple
// of 8 (capi.hip), so the 8-column groups tile the rows exactly; the padding columns beyond NX/2
// are transformed along (never read by anyone).
#include <cmath>

#include "ekpnp_internal.h"

namespace ekpnp {

constexpr int FY_N = 512, FY_COLS = 8;
constexpr size_t FY_LDS = (size_t)(FY_N * FY_COLS + FY_N) * sizeof(double2);  // image + twiddles = 72 KB

template <int SIGN>
__device__ __forceinline__ double2 mul_i(double2 a) {  // a * exp(SIGN i pi/2)
  return SIGN > 0 ? make_double2(-a.y, a.x) : make_double2(a.y, -a.x);
}
template <int SIGN>
__device__ __forceinline__ double2 mul_w8(double2 a) {  // a * exp(SIGN i pi/4)
  constexpr double h = 0.70710678118654752440;
  return SIGN > 0 ? make_double2((a.x - a.y) * h, (a.x + a.y) * h) : make_double2((a.x + a.y) * h, (a.y - a.x) * h);
}
template <int SIGN>
__device__ __forceinline__ double2 mul_tw(double2 a, double2 w) {  // a * w (forward table) or a * conj(w)
  return SIGN > 0 ? make_double2(a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y) : make_double2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
}
__device__ __forceinline__ void bfly(double2& a, double2& b) {
  const double2 t = a;
  a = make_double2(t.x + b.x, t.y + b.y);
  b = make_double2(t.x - b.x, t.y - b.y);
}
// 8-point DFT in registers; output X[k] is left in v[rev3(k)] (k -> bit-reversed slot)
template <int SIGN>
__device__ __forceinline__ void dft8(double2 (&v)[8]) {
  bfly(v[0], v[4]); bfly(v[1], v[5]); bfly(v[2], v[6]); bfly(v[3], v[7]);
  v[5] = mul_w8<SIGN>(v[5]);
  v[6] = mul_i<SIGN>(v[6]);
  v[7] = mul_i<SIGN>(mul_w8<SIGN>(v[7]));
  bfly(v[0], v[2]); bfly(v[1], v[3]); bfly(v[4], v[6]); bfly(v[5], v[7]);
  v[3] = mul_i<SIGN>(v[3]);
  v[7] = mul_i<SIGN>(v[7]);
  bfly(v[0], v[1]); bfly(v[2], v[3]); bfly(v[4], v[5]); bfly(v[6], v[7]);
}
__device__ __forceinline__ constexpr int rev3(int i) { return ((i & 1) << 2) | (i & 2) | ((i >> 2) & 1); }

template <int SIGN>
__global__ void __launch_bounds__(FY_N) k_fft_y512(double2* __restrict__ spec, const double2* __restrict__ tw, int nxh, long long plane_stride) {
  extern __shared__ double2 fy_lds[];
  double2* buf = fy_lds;               // [512 rows][8 columns]
  double2* w = fy_lds + FY_N * FY_COLS;  // exp(-2 pi i k / 512)
  const int c = threadIdx.x & 7, t = threadIdx.x >> 3;  // column, butterfly 0..63
  double2* base = spec + (long long)blockIdx.y * plane_stride + (long long)blockIdx.x * FY_COLS + c;
  double2 v[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = base[(long long)(t + 64 * r) * nxh];
  w[threadIdx.x] = tw[threadIdx.x];
  // stage 0 (sub-transform length 1): no twiddles; X[k] -> row 8 t + k
  dft8<SIGN>(v);
#pragma unroll
  for (int i = 0; i < 8; ++i) buf[(8 * t + rev3(i)) * FY_COLS + c] = v[i];
  __syncthreads();
  // stage 1 (length 8): twiddle exp(-+2 pi i (t mod 8) r / 64); X[k] -> row (t / 8) 64 + (t mod 8) + 8 k
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = buf[(t + 64 * r) * FY_COLS + c];
  {
    const int k = t & 7;
#pragma unroll
    for (int r = 1; r < 8; ++r) v[r] = mul_tw<SIGN>(v[r], w[k * r * 8]);
  }
  dft8<SIGN>(v);
  __syncthreads();  // every thread holds its inputs: the image may be overwritten
  {
    const int j0 = (t >> 3) * 64 + (t & 7);
#pragma unroll
    for (int i = 0; i < 8; ++i) buf[(j0 + 8 * rev3(i)) * FY_COLS + c] = v[i];
  }
  __syncthreads();
  // stage 2 (length 64): twiddle exp(-+2 pi i t r / 512); X[k] -> row t + 64 k
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = buf[(t + 64 * r) * FY_COLS + c];
#pragma unroll
  for (int r = 1; r < 8; ++r) v[r] = mul_tw<SIGN>(v[r], w[t * r]);
  dft8<SIGN>(v);
#pragma unroll
  for (int i = 0; i < 8; ++i) base[(long long)(t + 64 * rev3(i)) * nxh] = v[i];
}

bool fft_y_supported(int ny, int nxh) { return ny == FY_N && nxh % FY_COLS == 0; }

// device table exp(-2 pi i k / 512), k = 0..511
int fft_y_make_table(Ctx& c) {
  std::vector<double2> h(FY_N);
  for (int k = 0; k < FY_N; ++k) {
    const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)FY_N;
    h[k] = make_double2((double)cosl(a), (double)sinl(a));
  }
  // exact values where they are exact
  h[0] = make_double2(1.0, 0.0);
  h[FY_N / 4] = make_double2(0.0, -1.0);
  h[FY_N / 2] = make_double2(-1.0, 0.0);
  h[3 * FY_N / 4] = make_double2(0.0, 1.0);
  if (hipMalloc((void**)&c.fft_tw, FY_N * sizeof(double2)) != hipSuccess) { c.fft_tw = nullptr; c.err = "hipMalloc (twiddles) failed"; return EKPNP_ERR_NOMEM; }
  c.bytes += FY_N * sizeof(double2);
  if (hipMemcpy(c.fft_tw, h.data(), FY_N * sizeof(double2), hipMemcpyHostToDevice) != hipSuccess) { c.err = "hipMemcpy (twiddles) failed"; return EKPNP_ERR_HIP; }
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fft_y512<-1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FY_LDS) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fft_y512<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FY_LDS) != hipSuccess) {
    c.err = "hipFuncSetAttribute (72 KB of LDS for the y transform) failed";
    return EKPNP_ERR_HIP;
  }
  return EKPNP_OK;
}

// in place on the owned interior planes of the spectrum; sign -1: forward, +1: inverse
void launch_fft_y(Ctx& c, int sign) {
  if (c.fft_nz <= 0) return;
  double2* s = c.spec + (size_t)c.fft_z0 * c.p.ny * c.nxh;
  const dim3 grid(c.nxh / FY_COLS, c.fft_nz);
  const long long ps = (long long)c.p.ny * c.nxh;
  if (sign < 0)
    hipLaunchKernelGGL(k_fft_y512<-1>, grid, dim3(FY_N), FY_LDS, c.stream, s, c.fft_tw, c.nxh, ps);
  else
    hipLaunchKernelGGL(k_fft_y512<1>, grid, dim3(FY_N), FY_LDS, c.stream, s, c.fft_tw, c.nxh, ps);
  note_launch(c, "k_fft_y512");
}

}  // namespace ekpnp
