// fft_y_probe.hip (round 3) - a hand-written strided y pass for planes of 1024 rows (cfg5), against rocFFT.
// rocFFT's batched 2-D plan takes 4 + 4 kernels on 1024 x 1024 planes (two of them transposes: it has no strided-column
// kernel for 1024-long columns), 1.56 + 1.58 ms on a 1024 x 1024 x 126 slab.  Candidate: rocFFT 1-D row transforms
// (D2Z / Z2D, 2 kernels each) + ONE own kernel per direction for the columns.
//   k_fft_y<1024>: a workgroup owns COLS adjacent kx columns of one plane (COLS x 16 B of every row), 64 threads per
//   column, Stockham radix 16 x 8 x 8 through LDS: stage 0 reads the rows straight from global memory (16 per thread),
//   stage 2 writes them back in place.
// Prints: time of the 2-D plans, of the 1-D row plans, of the own column pass; max error of (rows + own columns) vs 2-D plan.
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define FK(x) do { hipfftResult r = (x); if (r != HIPFFT_SUCCESS) { printf("hipFFT error %d at %d\n", (int)r, __LINE__); exit(1); } } while (0)

#include "../ek-pnp-3d_amd/csrc/fft_plane.h"

template <class F>
static float timeit(const char* name, F f) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); f();
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("%-72s %8.3f ms\n", name, best);
  fflush(stdout);
  return best;
}

int main(int argc, char** argv) {
  const int nx = argc > 1 ? atoi(argv[1]) : 1024, ny = argc > 2 ? atoi(argv[2]) : 1024, nz = argc > 3 ? atoi(argv[3]) : 126;
  const int nxc = nx / 2 + 1, nxh = (nxc + 7) / 8 * 8;
  double *real, *back;
  double2 *specA, *specB, *tw;
  const size_t nreal = (size_t)nx * ny * nz, nspec = (size_t)nxh * ny * nz;
  CK(hipMalloc(&real, nreal * sizeof(double)));
  CK(hipMalloc(&back, nreal * sizeof(double)));
  CK(hipMalloc(&specA, nspec * sizeof(double2)));
  CK(hipMalloc(&specB, nspec * sizeof(double2)));
  std::vector<double> h(nreal);
  srand(7);
  for (auto& v : h) v = rand() / (double)RAND_MAX - 0.5;
  CK(hipMemcpy(real, h.data(), nreal * sizeof(double), hipMemcpyHostToDevice));
  CK(hipMemset(specA, 0, nspec * sizeof(double2)));
  CK(hipMemset(specB, 0, nspec * sizeof(double2)));
  std::vector<double2> htw(ny);
  ekpnp::fft_y_twiddles(ny, htw.data());
  CK(hipMalloc(&tw, ny * sizeof(double2)));
  CK(hipMemcpy(tw, htw.data(), ny * sizeof(double2), hipMemcpyHostToDevice));
  if (!ekpnp::fft_y_supported(ny, nxh) || !ekpnp::fft_y_prepare(ny)) { printf("own y pass does not support ny = %d\n", ny); return 1; }

  hipfftHandle p2f, p2i, p1f, p1i;
  {
    int n[2] = {ny, nx}, rembed[2] = {ny, nx}, cembed[2] = {ny, nxh};
    FK(hipfftPlanMany(&p2f, 2, n, rembed, 1, ny * nx, cembed, 1, ny * nxh, HIPFFT_D2Z, nz));
    FK(hipfftPlanMany(&p2i, 2, n, cembed, 1, ny * nxh, rembed, 1, ny * nx, HIPFFT_Z2D, nz));
    int n1[1] = {nx}, re1[1] = {nx}, ce1[1] = {nxh};
    FK(hipfftPlanMany(&p1f, 1, n1, re1, 1, nx, ce1, 1, nxh, HIPFFT_D2Z, ny * nz));
    FK(hipfftPlanMany(&p1i, 1, n1, ce1, 1, nxh, re1, 1, nx, HIPFFT_Z2D, ny * nz));
  }
  printf("%d x %d planes, batch %d, spectrum pitch %d\n", nx, ny, nz, nxh);
  timeit("rocFFT 2-D forward", [&] { FK(hipfftExecD2Z(p2f, real, (hipfftDoubleComplex*)specA)); });
  timeit("rocFFT 1-D rows forward", [&] { FK(hipfftExecD2Z(p1f, real, (hipfftDoubleComplex*)specB)); });
  timeit("own columns forward (in place)", [&] { ekpnp::fft_y_launch(specB, tw, ny, nxh, nz, -1, 0); });
  timeit("rocFFT 1-D rows + own columns, forward", [&] { FK(hipfftExecD2Z(p1f, real, (hipfftDoubleComplex*)specB)); ekpnp::fft_y_launch(specB, tw, ny, nxh, nz, -1, 0); });
  // correctness of the forward pair
  FK(hipfftExecD2Z(p2f, real, (hipfftDoubleComplex*)specA));
  FK(hipfftExecD2Z(p1f, real, (hipfftDoubleComplex*)specB));
  ekpnp::fft_y_launch(specB, tw, ny, nxh, nz, -1, 0);
  CK(hipDeviceSynchronize());
  {
    std::vector<double2> a((size_t)nxh * ny), b((size_t)nxh * ny);
    double worst = 0, big = 0;
    for (int z : {0, nz / 2, nz - 1}) {
      CK(hipMemcpy(a.data(), specA + (size_t)z * nxh * ny, a.size() * sizeof(double2), hipMemcpyDeviceToHost));
      CK(hipMemcpy(b.data(), specB + (size_t)z * nxh * ny, b.size() * sizeof(double2), hipMemcpyDeviceToHost));
      for (int ky = 0; ky < ny; ++ky)
        for (int kx = 0; kx < nxc; ++kx) {
          const double2 u = a[(size_t)ky * nxh + kx], v = b[(size_t)ky * nxh + kx];
          worst = fmax(worst, hypot(u.x - v.x, u.y - v.y));
          big = fmax(big, hypot(u.x, u.y));
        }
    }
    printf("forward: max |own - rocFFT 2-D| = %.3e of max %.3e\n", worst, big);
  }
  if (ekpnp::fft_x_supported(nx) && nx == ny && ekpnp::fft_x_prepare(nx)) {
    const long long nrows = (long long)ny * nz;
    timeit("own rows forward (R2C)", [&] { ekpnp::fft_x_forward(real, specB, tw, nx, nxh, nrows, 0); });
    timeit("own rows + own columns, forward", [&] { ekpnp::fft_x_forward(real, specB, tw, nx, nxh, nrows, 0); ekpnp::fft_y_launch(specB, tw, ny, nxh, nz, -1, 0); });
    CK(hipMemset(specB, 0, nspec * sizeof(double2)));
    ekpnp::fft_x_forward(real, specB, tw, nx, nxh, nrows, 0);
    ekpnp::fft_y_launch(specB, tw, ny, nxh, nz, -1, 0);
    CK(hipDeviceSynchronize());
    std::vector<double2> a((size_t)nxh * ny), b((size_t)nxh * ny);
    double worst = 0, big = 0;
    for (int z : {0, nz / 2, nz - 1}) {
      CK(hipMemcpy(a.data(), specA + (size_t)z * nxh * ny, a.size() * sizeof(double2), hipMemcpyDeviceToHost));
      CK(hipMemcpy(b.data(), specB + (size_t)z * nxh * ny, b.size() * sizeof(double2), hipMemcpyDeviceToHost));
      for (int ky = 0; ky < ny; ++ky)
        for (int kx = 0; kx < nxc; ++kx) {
          const double2 u = a[(size_t)ky * nxh + kx], v = b[(size_t)ky * nxh + kx];
          worst = fmax(worst, hypot(u.x - v.x, u.y - v.y));
          big = fmax(big, hypot(u.x, u.y));
        }
    }
    printf("forward, own rows + own columns: max |own - rocFFT 2-D| = %.3e of max %.3e\n", worst, big);
    timeit("own columns + own rows, inverse (C2R)", [&] { ekpnp::fft_y_launch(specB, tw, ny, nxh, nz, +1, 0); ekpnp::fft_x_inverse(specB, back, tw, nx, nxh, nrows, 0); });
    timeit("own rows inverse alone", [&] { ekpnp::fft_x_inverse(specB, back, tw, nx, nxh, nrows, 0); });
    ekpnp::fft_x_forward(real, specB, tw, nx, nxh, nrows, 0);
    ekpnp::fft_y_launch(specB, tw, ny, nxh, nz, -1, 0);
    ekpnp::fft_y_launch(specB, tw, ny, nxh, nz, +1, 0);
    ekpnp::fft_x_inverse(specB, back, tw, nx, nxh, nrows, 0);
    CK(hipDeviceSynchronize());
    std::vector<double> r((size_t)nx * ny);
    double rt = 0;
    for (int z : {0, nz - 1}) {
      CK(hipMemcpy(r.data(), back + (size_t)z * nx * ny, r.size() * sizeof(double), hipMemcpyDeviceToHost));
      for (size_t i = 0; i < r.size(); ++i) rt = fmax(rt, fabs(r[i] / ((double)nx * ny) - h[(size_t)z * nx * ny + i]));
    }
    printf("round trip, all four own passes: max error %.3e\n", rt);
    // own inverse against rocFFT's inverse on the SAME spectrum (rocFFT's 2-D forward)
    FK(hipfftExecD2Z(p2f, real, (hipfftDoubleComplex*)specA));
    CK(hipMemcpy(specB, specA, nspec * sizeof(double2), hipMemcpyDeviceToDevice));
    ekpnp::fft_y_launch(specB, tw, ny, nxh, nz, +1, 0);
    ekpnp::fft_x_inverse(specB, back, tw, nx, nxh, nrows, 0);
    CK(hipDeviceSynchronize());
    rt = 0;
    for (int z : {0, nz - 1}) {
      CK(hipMemcpy(r.data(), back + (size_t)z * nx * ny, r.size() * sizeof(double), hipMemcpyDeviceToHost));
      for (size_t i = 0; i < r.size(); ++i) rt = fmax(rt, fabs(r[i] / ((double)nx * ny) - h[(size_t)z * nx * ny + i]));
    }
    printf("rocFFT forward, own inverse: max error %.3e\n", rt);
  }
  // inverse: the 2-D plan overwrites its input, so always refill first
  timeit("rocFFT 2-D forward + inverse", [&] { FK(hipfftExecD2Z(p2f, real, (hipfftDoubleComplex*)specA)); FK(hipfftExecZ2D(p2i, (hipfftDoubleComplex*)specA, back)); });
  timeit("rows + own columns forward, own columns + rows inverse", [&] {
    FK(hipfftExecD2Z(p1f, real, (hipfftDoubleComplex*)specB)); ekpnp::fft_y_launch(specB, tw, ny, nxh, nz, -1, 0);
    ekpnp::fft_y_launch(specB, tw, ny, nxh, nz, +1, 0); FK(hipfftExecZ2D(p1i, (hipfftDoubleComplex*)specB, back)); });
  timeit("own columns inverse alone", [&] { ekpnp::fft_y_launch(specB, tw, ny, nxh, nz, +1, 0); });
  timeit("rocFFT 1-D rows inverse alone", [&] { FK(hipfftExecZ2D(p1i, (hipfftDoubleComplex*)specB, back)); });
  FK(hipfftExecD2Z(p1f, real, (hipfftDoubleComplex*)specB)); ekpnp::fft_y_launch(specB, tw, ny, nxh, nz, -1, 0);
  ekpnp::fft_y_launch(specB, tw, ny, nxh, nz, +1, 0); FK(hipfftExecZ2D(p1i, (hipfftDoubleComplex*)specB, back));
  CK(hipDeviceSynchronize());
  {
    std::vector<double> r((size_t)nx * ny);
    double rt = 0;
    for (int z : {0, nz - 1}) {
      CK(hipMemcpy(r.data(), back + (size_t)z * nx * ny, r.size() * sizeof(double), hipMemcpyDeviceToHost));
      for (size_t i = 0; i < r.size(); ++i) rt = fmax(rt, fabs(r[i] / ((double)nx * ny) - h[(size_t)z * nx * ny + i]));
    }
    printf("round trip (own columns both ways): max error %.3e\n", rt);
  }
  return 0;
}
