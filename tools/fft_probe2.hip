// fft_probe2.hip - round 3: does a TRANSPOSED half spectrum [z][kx][ky] (ky contiguous) spare rocFFT the two extra
// transpose passes its batched 2-D plan takes on planes whose y length has no strided-column kernel (cfg5: 1024 x 1024,
// profiles/r03_cfg5_rank_shape_slab_kernel_stats.csv: 4 + 4 kernels per solve instead of 2 + 2 on 512 x 512)?
//   A: hipFFT-style layout [z][ky][kx] (row pitch nxh), what csrc/capi.hip uses
//   B: rocFFT native API, output strides {nyp, 1}: element (kx, ky) of a plane at kx * nyp + ky
// Prints the time of each direction and checks B against A on random data.
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define RK(x) do { rocfft_status r = (x); if (r != rocfft_status_success) { printf("rocFFT error %d at line %d: %s\n", (int)r, __LINE__, #x); return false; } } while (0)

struct Plan {
  rocfft_plan p = nullptr;
  rocfft_execution_info info = nullptr;
  void* work = nullptr;
};

static bool make(Plan& P, bool forward, size_t nx, size_t ny, size_t batch, const size_t* rs, size_t rdist, const size_t* cs, size_t cdist) {
  rocfft_plan_description d = nullptr;
  RK(rocfft_plan_description_create(&d));
  if (forward)
    RK(rocfft_plan_description_set_data_layout(d, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, nullptr, nullptr, 2, rs, rdist, 2, cs, cdist));
  else
    RK(rocfft_plan_description_set_data_layout(d, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, nullptr, nullptr, 2, cs, cdist, 2, rs, rdist));
  const size_t len[2] = {nx, ny};
  RK(rocfft_plan_create(&P.p, rocfft_placement_notinplace, forward ? rocfft_transform_type_real_forward : rocfft_transform_type_real_inverse, rocfft_precision_double, 2, len, batch, d));
  RK(rocfft_plan_description_destroy(d));
  size_t ws = 0;
  RK(rocfft_plan_get_work_buffer_size(P.p, &ws));
  RK(rocfft_execution_info_create(&P.info));
  if (ws) {
    CK(hipMalloc(&P.work, ws));
    RK(rocfft_execution_info_set_work_buffer(P.info, P.work, ws));
  }
  printf("    plan ok, work buffer %.1f MB\n", ws / 1e6);
  return true;
}

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
  const size_t nx = argc > 1 ? atoi(argv[1]) : 1024, ny = argc > 2 ? atoi(argv[2]) : 1024, nz = argc > 3 ? atoi(argv[3]) : 126;
  const size_t nxc = nx / 2 + 1, nxh = (nxc + 7) / 8 * 8, nyp = ny + 8;  // nyp: padded pitch of the transposed rows (avoid a power-of-two stride)
  if (rocfft_setup() != rocfft_status_success) return 1;
  double *real, *back;
  double2 *specA, *specB;
  CK(hipMalloc(&real, nx * ny * nz * sizeof(double)));
  CK(hipMalloc(&back, nx * ny * nz * sizeof(double)));
  CK(hipMalloc(&specA, nxh * ny * nz * sizeof(double2)));
  CK(hipMalloc(&specB, nxh * nyp * nz * sizeof(double2)));
  std::vector<double> h(nx * ny * nz);
  srand(7);
  for (auto& v : h) v = rand() / (double)RAND_MAX - 0.5;
  CK(hipMemcpy(real, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
  CK(hipMemset(specA, 0, nxh * ny * nz * sizeof(double2)));
  CK(hipMemset(specB, 0, nxh * nyp * nz * sizeof(double2)));
  const size_t rs[2] = {1, nx};
  const size_t csA[2] = {1, nxh}, csB[2] = {nyp, 1};
  Plan Af, Ai, Bf, Bi;
  printf("%zu x %zu planes, batch %zu\nA: [z][ky][kx], pitch %zu\n", nx, ny, nz, nxh);
  if (!make(Af, true, nx, ny, nz, rs, nx * ny, csA, nxh * ny) || !make(Ai, false, nx, ny, nz, rs, nx * ny, csA, nxh * ny)) return 1;
  void* in[1]; void* out[1];
  timeit("A forward", [&] { in[0] = real; out[0] = specA; rocfft_execute(Af.p, in, out, Af.info); });
  printf("B: [z][kx][ky], pitch %zu\n", nyp);
  const bool okB = make(Bf, true, nx, ny, nz, rs, nx * ny, csB, nxh * nyp) && make(Bi, false, nx, ny, nz, rs, nx * ny, csB, nxh * nyp);
  if (okB) timeit("B forward", [&] { in[0] = real; out[0] = specB; rocfft_execute(Bf.p, in, out, Bf.info); });
  // the inverse of a real transform may overwrite its input: time it on a scratch copy pattern (forward again first)
  timeit("A forward + inverse", [&] { in[0] = real; out[0] = specA; rocfft_execute(Af.p, in, out, Af.info); in[0] = specA; out[0] = back; rocfft_execute(Ai.p, in, out, Ai.info); });
  if (okB) timeit("B forward + inverse", [&] { in[0] = real; out[0] = specB; rocfft_execute(Bf.p, in, out, Bf.info); in[0] = specB; out[0] = back; rocfft_execute(Bi.p, in, out, Bi.info); });
  if (okB) {
    // B's spectrum against A's, and B's round trip against the input
    in[0] = real; out[0] = specA; rocfft_execute(Af.p, in, out, Af.info);
    in[0] = real; out[0] = specB; rocfft_execute(Bf.p, in, out, Bf.info);
    CK(hipDeviceSynchronize());
    std::vector<double2> a(nxh * ny), b(nxh * nyp);
    CK(hipMemcpy(a.data(), specA + (nz - 1) * nxh * ny, a.size() * sizeof(double2), hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), specB + (nz - 1) * nxh * nyp, b.size() * sizeof(double2), hipMemcpyDeviceToHost));
    double worst = 0, big = 0;
    for (size_t ky = 0; ky < ny; ++ky)
      for (size_t kx = 0; kx < nxc; ++kx) {
        const double2 u = a[ky * nxh + kx], v = b[kx * nyp + ky];
        worst = fmax(worst, hypot(u.x - v.x, u.y - v.y));
        big = fmax(big, hypot(u.x, u.y));
      }
    in[0] = specB; out[0] = back; rocfft_execute(Bi.p, in, out, Bi.info);
    CK(hipDeviceSynchronize());
    std::vector<double> r(nx * ny);
    CK(hipMemcpy(r.data(), back + (nz - 1) * nx * ny, r.size() * sizeof(double), hipMemcpyDeviceToHost));
    double rt = 0;
    for (size_t i = 0; i < r.size(); ++i) rt = fmax(rt, fabs(r[i] / (double)(nx * ny) - h[(nz - 1) * nx * ny + i]));
    printf("B vs A spectrum: max |diff| %.3e of max %.3e; B round trip max error %.3e\n", worst, big, rt);
  }
  rocfft_cleanup();
  return 0;
}
