// fft_probe.hip — would a TRANSPOSED half spectrum [kx][z][y] (y contiguous) make the 2-D transform
// of the Poisson solve cheaper than the library's current [z][ky][kx] with a strided y pass?
//   A: one batched 2-D D2Z / Z2D plan per direction (what csrc/capi.hip uses), row pitch 264
//   B: x pass as a batched 1-D D2Z whose output stride is nz*ny (writes the transpose), y pass as a
//      contiguous batched 1-D Z2Z in place, and the mirror image for the inverse
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define FK(x) do { hipfftResult r = (x); if (r != HIPFFT_SUCCESS) { printf("hipFFT error %d at %d\n", (int)r, __LINE__); exit(1); } } while (0)

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
  printf("%-64s %8.3f ms\n", name, best);
  fflush(stdout);
  return best;
}

int main(int argc, char** argv) {
  const int nx = argc > 1 ? atoi(argv[1]) : 512, ny = argc > 2 ? atoi(argv[2]) : 512, nz = argc > 3 ? atoi(argv[3]) : 512;
  const int nxc = nx / 2 + 1, nxh = (nxc + 7) / 8 * 8;
  double* real;
  hipfftDoubleComplex* spec;
  const size_t nreal = (size_t)nx * ny * nz, nspec = (size_t)nxh * ny * nz;
  CK(hipMalloc(&real, nreal * sizeof(double)));
  CK(hipMalloc(&spec, nspec * sizeof(hipfftDoubleComplex)));
  CK(hipMemset(real, 0, nreal * sizeof(double)));
  CK(hipMemset(spec, 0, nspec * sizeof(hipfftDoubleComplex)));
  float a_f, a_i, b_xf, b_y, b_yi, b_xi;
  {
    hipfftHandle pf, pi;
    int n[2] = {ny, nx}, rembed[2] = {ny, nx}, cembed[2] = {ny, nxh};
    FK(hipfftPlanMany(&pf, 2, n, rembed, 1, ny * nx, cembed, 1, ny * nxh, HIPFFT_D2Z, nz));
    FK(hipfftPlanMany(&pi, 2, n, cembed, 1, ny * nxh, rembed, 1, ny * nx, HIPFFT_Z2D, nz));
    a_f = timeit("A forward  2-D D2Z, [z][ky][kx] pitch padded to 8", [&] { FK(hipfftExecD2Z(pf, real, spec)); });
    a_i = timeit("A inverse  2-D Z2D", [&] { FK(hipfftExecZ2D(pi, spec, real)); });
    hipfftDestroy(pf); hipfftDestroy(pi);
  }
  {
    hipfftHandle xf, xi, yy;
    const int nb = nz * ny;  // batch of x transforms
    int n1[1] = {nx}, rembed[1] = {nx}, cembed[1] = {nxc};
    // output element kx of batch b at kx*nb + b
    FK(hipfftPlanMany(&xf, 1, n1, rembed, 1, nx, cembed, nb, 1, HIPFFT_D2Z, nb));
    FK(hipfftPlanMany(&xi, 1, n1, cembed, nb, 1, rembed, 1, nx, HIPFFT_Z2D, nb));
    int n2[1] = {ny}, e2[1] = {ny};
    FK(hipfftPlanMany(&yy, 1, n2, e2, 1, ny, e2, 1, ny, HIPFFT_Z2Z, nxc * nz));
    b_xf = timeit("B x forward  1-D D2Z, output stride nz*ny (transposing)", [&] { FK(hipfftExecD2Z(xf, real, spec)); });
    b_y = timeit("B y forward  1-D Z2Z contiguous, in place", [&] { FK(hipfftExecZ2Z(yy, spec, spec, HIPFFT_FORWARD)); });
    b_yi = timeit("B y inverse", [&] { FK(hipfftExecZ2Z(yy, spec, spec, HIPFFT_BACKWARD)); });
    b_xi = timeit("B x inverse  1-D Z2D, input stride nz*ny", [&] { FK(hipfftExecZ2D(xi, spec, real)); });
    hipfftDestroy(xf); hipfftDestroy(xi); hipfftDestroy(yy);
  }
  printf("A total %.3f ms   B total %.3f ms\n", a_f + a_i, b_xf + b_y + b_yi + b_xi);
  return 0;
}
