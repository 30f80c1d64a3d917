/*
 * ref_driver.cpp — runs the REFERENCE's own kernels to produce golden vectors.  TEST INFRA ONLY.
 *
 * This file is ours; the three files it includes (LBM.h, LBM.cu, poisson.cu) are the
 * reference's sources as translated on the fly by build_ref.sh (AMD's hipify-perl renames the
 * cuda* / cufft* API calls to hip* / hipfft*; nothing else is edited except the whitespace
 * inside the `<< <` `>> >` launch brackets).  They are included in the same order as
 * main.cu:13-15 (unity build).  The translated text lives in a temporary directory during the
 * build only; it is never written into the repository and never travels.
 *
 * The driver repeats main.cu:21-35 (symbol copies), main.cu:78-152 (allocations, FFT plan,
 * wavenumber tables), then calls the reference's host API exactly as main.cu:163-198 does and
 * dumps the 11 macroscopic fields as raw float64 (no lossy %10.6f ASCII, LBM.cu:2619).
 *
 * usage: ref_driver <outdir> init                       -> <outdir>/g1_init.bin, g1_step{1,5,20,100}.bin
 *        ref_driver <outdir> fields <in.bin> <tag> n... -> upload 11 fields, fast_Poisson,
 *                                                          init_equilibrium, dump after the listed steps
 *        ref_driver <outdir> poisson <in.bin> <tag>     -> upload fields, one fast_Poisson, dump
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "LBM.h"
#include "LBM.cu"
#include "poisson.cu"

static double* g_fields[11];
static const char* g_names[11] = {"rho", "c", "cn", "phi", "ux", "uy", "uz", "Ex", "Ey", "Ez", "T"};

static void dump(const std::string& path) {
  std::vector<double> h((size_t)NX * NY * NZ);
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
  for (int i = 0; i < 11; ++i) {
    CHECK(hipMemcpy(h.data(), g_fields[i], mem_size_scalar, hipMemcpyDeviceToHost));
    fwrite(h.data(), sizeof(double), h.size(), f);
  }
  fclose(f);
  printf("wrote %s\n", path.c_str());
}

static void upload(const char* path) {
  std::vector<double> h((size_t)NX * NY * NZ);
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  for (int i = 0; i < 11; ++i) {
    if (fread(h.data(), sizeof(double), h.size(), f) != h.size()) { fprintf(stderr, "short read %s\n", path); exit(2); }
    CHECK(hipMemcpy(g_fields[i], h.data(), mem_size_scalar, hipMemcpyHostToDevice));
  }
  fclose(f);
}

static void one_step() { /* main.cu:189-200 */
  stream_collide_save(f0_gpu, f1_gpu, f2_gpu, h0_gpu, h1_gpu, h2_gpu, hn0_gpu, hn1_gpu, hn2_gpu, temp0_gpu, temp1_gpu, temp2_gpu,
                      rho_gpu, charge_gpu, chargen_gpu, ux_gpu, uy_gpu, uz_gpu, Ex_gpu, Ey_gpu, Ez_gpu, T_gpu, t, f0bc);
  fast_Poisson(charge_gpu, chargen_gpu, kx, ky, kz, plan);
  t = t + dt_host;
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: see header\n"); return 2; }
  std::string out = argv[1];
  std::string mode = argv[2];

  /* main.cu:21-35 */
  hipMemcpyFromSymbol(&dt_host, HIP_SYMBOL(dt), sizeof(double), 0, hipMemcpyDeviceToHost);
  hipMemcpyFromSymbol(&Lx_host, HIP_SYMBOL(Lx), sizeof(double), 0, hipMemcpyDeviceToHost);
  hipMemcpyFromSymbol(&Ly_host, HIP_SYMBOL(Ly), sizeof(double), 0, hipMemcpyDeviceToHost);
  hipMemcpyFromSymbol(&dy_host, HIP_SYMBOL(dy), sizeof(double), 0, hipMemcpyDeviceToHost);
  hipMemcpyFromSymbol(&Lz_host, HIP_SYMBOL(Lz), sizeof(double), 0, hipMemcpyDeviceToHost);
  hipMemcpyFromSymbol(&dz_host, HIP_SYMBOL(dz), sizeof(double), 0, hipMemcpyDeviceToHost);
  hipMemcpyToSymbol(HIP_SYMBOL(nu), &nu_host, sizeof(double), 0, hipMemcpyHostToDevice);
  hipMemcpyToSymbol(HIP_SYMBOL(uw), &uw_host, sizeof(double), 0, hipMemcpyHostToDevice);
  hipMemcpyToSymbol(HIP_SYMBOL(exf), &exf_host, sizeof(double), 0, hipMemcpyHostToDevice);
  hipMemcpyToSymbol(HIP_SYMBOL(K), &K_host, sizeof(double), 0, hipMemcpyHostToDevice);
  hipMemcpyToSymbol(HIP_SYMBOL(Kn), &Kn_host, sizeof(double), 0, hipMemcpyHostToDevice);
  hipMemcpyToSymbol(HIP_SYMBOL(epsn), &epsn_host, sizeof(double), 0, hipMemcpyHostToDevice);

  checkCudaErrors(hipSetDevice(0));
  /* main.cu:78-109 */
  checkCudaErrors(hipMalloc((void**)&f0bc, sizeof(double) * NX * NY * 2));
  checkCudaErrors(hipMalloc((void**)&f0_gpu, mem_size_0dir));
  checkCudaErrors(hipMalloc((void**)&f1_gpu, mem_size_n0dir));
  checkCudaErrors(hipMalloc((void**)&f2_gpu, mem_size_n0dir));
  checkCudaErrors(hipMalloc((void**)&h0_gpu, mem_size_0dir));
  checkCudaErrors(hipMalloc((void**)&h1_gpu, mem_size_n0dir));
  checkCudaErrors(hipMalloc((void**)&h2_gpu, mem_size_n0dir));
  checkCudaErrors(hipMalloc((void**)&hn0_gpu, mem_size_0dir));
  checkCudaErrors(hipMalloc((void**)&hn1_gpu, mem_size_n0dir));
  checkCudaErrors(hipMalloc((void**)&hn2_gpu, mem_size_n0dir));
  checkCudaErrors(hipMalloc((void**)&temp0_gpu, mem_size_0dir));
  checkCudaErrors(hipMalloc((void**)&temp1_gpu, mem_size_n0dir));
  checkCudaErrors(hipMalloc((void**)&temp2_gpu, mem_size_n0dir));
  double** sc[11] = {&rho_gpu, &charge_gpu, &chargen_gpu, &phi_gpu, &ux_gpu, &uy_gpu, &uz_gpu, &Ex_gpu, &Ey_gpu, &Ez_gpu, &T_gpu};
  for (int i = 0; i < 11; ++i) {
    checkCudaErrors(hipMalloc((void**)sc[i], mem_size_scalar));
    checkCudaErrors(hipMemset(*sc[i], 0, mem_size_scalar));
    g_fields[i] = *sc[i];
  }
  checkCudaErrors(hipMalloc((void**)&kx, sizeof(double) * NX));
  checkCudaErrors(hipMalloc((void**)&ky, sizeof(double) * NY));
  checkCudaErrors(hipMalloc((void**)&kz, sizeof(double) * NE));
  /* main.cu:112 */
  CHECK_CUFFT(hipfftPlan3d(&plan, NE, NY, NX, HIPFFT_Z2Z));
  /* main.cu:119-152 */
  for (unsigned i = 0; i <= NX / 2; i++) kx_host[i] = (double)i * 2.0 * M_PI / Lx_host;
  for (unsigned i = NX / 2 + 1; i < NX; i++) kx_host[i] = ((double)i - NX) * 2.0 * M_PI / Lx_host;
  for (unsigned i = 0; i <= NY / 2; i++) ky_host[i] = (double)i * 2.0 * M_PI / Ly_host;
  for (unsigned i = NY / 2 + 1; i < NY; i++) ky_host[i] = ((double)i - NY) * 2.0 * M_PI / Ly_host;
  for (unsigned i = 0; i <= NE / 2; i++) kz_host[i] = (double)i * 2.0 * M_PI / (NE * dz_host);
  for (unsigned i = NE / 2 + 1; i < NE; i++) kz_host[i] = ((double)i - NE) * 2.0 * M_PI / (NE * dz_host);
  CHECK(hipMemcpy(kx, kx_host, sizeof(double) * NX, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(ky, ky_host, sizeof(double) * NY, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(kz, kz_host, sizeof(double) * NE, hipMemcpyHostToDevice));

  printf("reference grid %ux%ux%u (NE=%u) dt=%g\n", NX, NY, NZ, NE, dt_host);

  if (mode == "init") {
    /* main.cu:169-175 */
    initialization(rho_gpu, charge_gpu, chargen_gpu, phi_gpu, ux_gpu, uy_gpu, uz_gpu, Ex_gpu, Ey_gpu, Ez_gpu, T_gpu);
    t = 0;
    dump(out + "/g1_init.bin");
    init_equilibrium(f0_gpu, f1_gpu, h0_gpu, h1_gpu, hn0_gpu, hn1_gpu, temp0_gpu, temp1_gpu, rho_gpu, charge_gpu, chargen_gpu, ux_gpu,
                     uy_gpu, uz_gpu, Ex_gpu, Ey_gpu, Ez_gpu, T_gpu);
    const int marks[4] = {1, 5, 20, 100};
    int done = 0;
    for (int m = 0; m < 4; ++m) {
      for (; done < marks[m]; ++done) one_step();
      dump(out + "/g1_step" + std::to_string(marks[m]) + ".bin");
    }
  } else if (mode == "fields" && argc >= 6) {
    upload(argv[3]);
    std::string tag = argv[4];
    fast_Poisson(charge_gpu, chargen_gpu, kx, ky, kz, plan);
    dump(out + "/" + tag + "_step0.bin");
    init_equilibrium(f0_gpu, f1_gpu, h0_gpu, h1_gpu, hn0_gpu, hn1_gpu, temp0_gpu, temp1_gpu, rho_gpu, charge_gpu, chargen_gpu, ux_gpu,
                     uy_gpu, uz_gpu, Ex_gpu, Ey_gpu, Ez_gpu, T_gpu);
    t = 0;
    int done = 0;
    for (int a = 5; a < argc; ++a) {
      int mark = atoi(argv[a]);
      for (; done < mark; ++done) one_step();
      dump(out + "/" + tag + "_step" + std::to_string(mark) + ".bin");
    }
  } else if (mode == "poisson" && argc >= 5) {
    upload(argv[3]);
    fast_Poisson(charge_gpu, chargen_gpu, kx, ky, kz, plan);
    dump(out + "/" + std::string(argv[4]) + ".bin");
  } else {
    fprintf(stderr, "bad mode\n");
    return 2;
  }
  checkCudaErrors(hipDeviceSynchronize());
  return 0;
}
