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
 * usage: ref_driver <outdir> [--set name=value]... init <tag> n...
 *            initialization() + init_equilibrium + steps; dumps <tag>_init.bin, <tag>_step<n>.bin,
 *            <tag>_init_trace.bin ([sweeps][NZ] phi column after every PB sweep's fast_Poisson) and
 *            <tag>_step_trace.bin ([steps][2][NZ] phi columns after every step)
 *        ref_driver <outdir> [--set ...] fields <in.bin> <tag> n...
 *            upload 11 fields, fast_Poisson, init_equilibrium, steps; dumps <tag>_step0.bin,
 *            <tag>_step<n>.bin and <tag>_step_trace.bin ([1+steps][2][NZ])
 *        ref_driver <outdir> poisson <in.bin> <tag>     -> upload fields, one fast_Poisson, dump
 *        ref_driver <outdir> kernels <in.bin> <tag>     -> G4: populations after init_equilibrium, one
 *            full step, then after each of gpu_collide_save / gpu_boundary / gpu_stream / gpu_bc_charge
 *        ref_driver <outdir> time <nsteps>              -> wall time of the reference's own step
 *        ref_driver <outdir> time0 <nsteps>             -> the same from uniform fields (large grids)
 *        ref_driver <outdir> io <in.bin> <tag>          -> upload fields; the reference's own
 *            save_data_tecplot (2 zones), save_data_end, record_umax and current() on them
 *        --set name=value writes a __constant__/__device__ physics symbol of LBM.h at run time
 *        (exf uw chargeinf voltage voltage2 Ext Ra TH K Kn nu D diffu diffun eps rho0 V VC VCn VT PB_omega):
 *        no source edit, hipMemcpyToSymbol only.
 * The phi columns let the tests measure the reference's DC-mode leak (poisson.cu:177) of every
 * single Poisson solve: phi_ref - phi_exact is one constant per solve on the interior.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "LBM.h"
#include "LBM.cu"
#include "poisson.cu"

#include <chrono>
static double seconds_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
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
  /* the reference's wall-current diagnostic, exactly as main.cu:211-215 calls it */
  CHECK(hipMemcpy(charge_host, charge_gpu, mem_size_scalar, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(chargen_host, chargen_gpu, mem_size_scalar, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(Ez_host, Ez_gpu, mem_size_scalar, hipMemcpyDeviceToHost));
  double cur = current(charge_host, chargen_host, Ez_host);
  FILE* g = fopen((path + ".current").c_str(), "wb");
  if (g) { fwrite(&cur, sizeof(double), 1, g); fclose(g); }
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

static std::vector<double> g_trace;
static void trace_phi() { /* columns (0,0) and (NX/2,NY/2) of phi */
  static std::vector<double> h((size_t)NX * NY * NZ);
  CHECK(hipMemcpy(h.data(), phi_gpu, mem_size_scalar, hipMemcpyDeviceToHost));
  for (unsigned z = 0; z < NZ; ++z) g_trace.push_back(h[scalar_index(0, 0, z)]);
  for (unsigned z = 0; z < NZ; ++z) g_trace.push_back(h[scalar_index(NX / 2, NY / 2, z)]);
}
static void write_trace(const std::string& path) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
  fwrite(g_trace.data(), sizeof(double), g_trace.size(), f);
  fclose(f);
  printf("wrote %s (%zu doubles)\n", path.c_str(), g_trace.size());
  g_trace.clear();
}

static void one_step() { /* main.cu:189-200 */
  stream_collide_save(f0_gpu, f1_gpu, f2_gpu, h0_gpu, h1_gpu, h2_gpu, hn0_gpu, hn1_gpu, hn2_gpu, temp0_gpu, temp1_gpu, temp2_gpu,
                      rho_gpu, charge_gpu, chargen_gpu, ux_gpu, uy_gpu, uz_gpu, Ex_gpu, Ey_gpu, Ez_gpu, T_gpu, t, f0bc);
  fast_Poisson(charge_gpu, chargen_gpu, kx, ky, kz, plan);
  t = t + dt_host;
  trace_phi();
}

/* The body of the reference's initialization() (LBM.cu:68-109), same kernels, same order, same
 * host round trip of phi_old, with a phi trace after every fast_Poisson. */
static void traced_initialization() {
  dim3 grid(NX / nThreads, NY, NZ);
  dim3 threads(nThreads, 1, 1);
  gpu_initialization<<<grid, threads>>>(rho_gpu, charge_gpu, chargen_gpu, phi_gpu, ux_gpu, uy_gpu, uz_gpu, Ex_gpu, Ey_gpu, Ez_gpu, T_gpu);
  checkCudaErrors(hipMalloc((void**)&phi_old_gpu, mem_size_scalar));
  double* phi_old_host = (double*)malloc(mem_size_scalar);
  CHECK(hipMemcpy(phi_old_host, phi_gpu, mem_size_scalar, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(phi_old_gpu, phi_old_host, mem_size_scalar, hipMemcpyHostToDevice));
  for (unsigned int i = 0; i <= 500; ++i) {
    gpu_PBE<<<grid, threads>>>(charge_gpu, phi_gpu, chargen_gpu);
    fast_Poisson(charge_gpu, chargen_gpu, kx, ky, kz, plan);
    trace_phi();
    gpu_PBE_phi<<<grid, threads>>>(phi_gpu, phi_old_gpu);
    CHECK(hipMemcpy(phi_old_host, phi_gpu, mem_size_scalar, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(phi_old_gpu, phi_old_host, mem_size_scalar, hipMemcpyHostToDevice));
  }
  free(phi_old_host);
  checkCudaErrors(hipFree(phi_old_gpu));
}

/* populations of the four lattices on the y rows {0,3,5}: X0 then X1 (which == 1) or X2 (which == 2),
 * each as [27][NZ][3][NX] with d = 0 the rest population */
static void dump_pops(const std::string& path, int which) {
  const unsigned ys[3] = {0, 3, 5};
  double* x0[4] = {f0_gpu, h0_gpu, hn0_gpu, temp0_gpu};
  double* x1[4] = {f1_gpu, h1_gpu, hn1_gpu, temp1_gpu};
  double* x2[4] = {f2_gpu, h2_gpu, hn2_gpu, temp2_gpu};
  std::vector<double> h0((size_t)NX * NY * NZ), hn((size_t)NX * NY * NZ * 26), o;
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
  for (int l = 0; l < 4; ++l) {
    CHECK(hipMemcpy(h0.data(), x0[l], mem_size_0dir, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hn.data(), which == 1 ? x1[l] : x2[l], mem_size_n0dir, hipMemcpyDeviceToHost));
    o.clear();
    for (unsigned d = 0; d < 27; ++d)
      for (unsigned z = 0; z < NZ; ++z)
        for (unsigned k = 0; k < 3; ++k)
          for (unsigned x = 0; x < NX; ++x)
            o.push_back(d == 0 ? h0[scalar_index(x, ys[k], z)] : hn[(size_t)NX * (NY * (NZ * (d - 1) + z) + ys[k]) + x]);
    fwrite(o.data(), sizeof(double), o.size(), f);
  }
  fclose(f);
  printf("wrote %s\n", path.c_str());
}

static std::vector<double> snapshot() {
  std::vector<double> all;
  std::vector<double> h((size_t)NX * NY * NZ);
  for (int i = 0; i < 11; ++i) {
    CHECK(hipMemcpy(h.data(), g_fields[i], mem_size_scalar, hipMemcpyDeviceToHost));
    all.insert(all.end(), h.begin(), h.end());
  }
  return all;
}

static void set_symbol(const std::string& kv) {
  size_t eq = kv.find('=');
  if (eq == std::string::npos) { fprintf(stderr, "bad --set %s\n", kv.c_str()); exit(2); }
  std::string name = kv.substr(0, eq);
  double v = atof(kv.c_str() + eq + 1);
  hipError_t e = hipErrorInvalidValue;
#define SETSYM(sym) else if (name == #sym) e = hipMemcpyToSymbol(HIP_SYMBOL(sym), &v, sizeof(double))
  if (false) {}
  SETSYM(exf); SETSYM(uw); SETSYM(chargeinf); SETSYM(voltage); SETSYM(voltage2); SETSYM(Ext); SETSYM(Ra); SETSYM(TH);
  SETSYM(K); SETSYM(Kn); SETSYM(nu); SETSYM(D); SETSYM(diffu); SETSYM(diffun); SETSYM(eps); SETSYM(rho0);
  SETSYM(V); SETSYM(VC); SETSYM(VCn); SETSYM(VT); SETSYM(PB_omega);
#undef SETSYM
  if (e != hipSuccess) { fprintf(stderr, "cannot set %s\n", name.c_str()); exit(2); }
  printf("set %s = %.17g\n", name.c_str(), v);
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: see header\n"); return 2; }
  std::string out = argv[1];
  std::vector<std::string> sets, args;
  for (int a = 2; a < argc; ++a) {
    if (std::string(argv[a]) == "--set" && a + 1 < argc) sets.push_back(argv[++a]);
    else args.push_back(argv[a]);
  }
  if (args.empty()) { fprintf(stderr, "usage: see header\n"); return 2; }
  std::string mode = args[0];

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
  for (auto& kv : sets) set_symbol(kv);

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

  if (mode == "init" && args.size() >= 2) {
    std::string tag = args[1];
    /* main.cu:169-175: the reference's own initialization() first ... */
    initialization(rho_gpu, charge_gpu, chargen_gpu, phi_gpu, ux_gpu, uy_gpu, uz_gpu, Ex_gpu, Ey_gpu, Ez_gpu, T_gpu);
    std::vector<double> a = snapshot();
    /* ... then its body again with the phi trace; both must agree bit for bit */
    traced_initialization();
    std::vector<double> b = snapshot();
    printf("traced initialization bitwise identical to initialization(): %s\n",
           memcmp(a.data(), b.data(), a.size() * sizeof(double)) == 0 ? "yes" : "NO");
    write_trace(out + "/" + tag + "_init_trace.bin");
    t = 0;
    dump(out + "/" + tag + "_init.bin");
    init_equilibrium(f0_gpu, f1_gpu, h0_gpu, h1_gpu, hn0_gpu, hn1_gpu, temp0_gpu, temp1_gpu, rho_gpu, charge_gpu, chargen_gpu, ux_gpu,
                     uy_gpu, uz_gpu, Ex_gpu, Ey_gpu, Ez_gpu, T_gpu);
    int done = 0;
    for (size_t a2 = 2; a2 < args.size(); ++a2) {
      int mark = atoi(args[a2].c_str());
      for (; done < mark; ++done) one_step();
      dump(out + "/" + tag + "_step" + std::to_string(mark) + ".bin");
    }
    write_trace(out + "/" + tag + "_step_trace.bin");
  } else if (mode == "fields" && args.size() >= 3) {
    upload(args[1].c_str());
    std::string tag = args[2];
    fast_Poisson(charge_gpu, chargen_gpu, kx, ky, kz, plan);
    trace_phi();
    dump(out + "/" + tag + "_step0.bin");
    init_equilibrium(f0_gpu, f1_gpu, h0_gpu, h1_gpu, hn0_gpu, hn1_gpu, temp0_gpu, temp1_gpu, rho_gpu, charge_gpu, chargen_gpu, ux_gpu,
                     uy_gpu, uz_gpu, Ex_gpu, Ey_gpu, Ez_gpu, T_gpu);
    t = 0;
    int done = 0;
    for (size_t a2 = 3; a2 < args.size(); ++a2) {
      int mark = atoi(args[a2].c_str());
      for (; done < mark; ++done) one_step();
      dump(out + "/" + tag + "_step" + std::to_string(mark) + ".bin");
    }
    write_trace(out + "/" + tag + "_step_trace.bin");
  } else if (mode == "io" && args.size() >= 3) {
    /* the reference's writers on uploaded fields: two Tecplot zones (first = 1, then 0), the
     * restart file and one umax line, as main.cu:179,207,221,256 produce them */
    upload(args[1].c_str());
    std::string tag = args[2];
    FILE* f = fopen((out + "/" + tag + "_data.dat").c_str(), "wb+");
    save_data_tecplot(f, 1.25e-8, rho_gpu, charge_gpu, chargen_gpu, phi_gpu, ux_gpu, uy_gpu, uz_gpu, Ex_gpu, Ey_gpu, Ez_gpu, T_gpu, 1);
    save_data_tecplot(f, 2.5e-8, rho_gpu, charge_gpu, chargen_gpu, phi_gpu, ux_gpu, uy_gpu, uz_gpu, Ex_gpu, Ey_gpu, Ez_gpu, T_gpu, 0);
    fclose(f);
    FILE* e = fopen((out + "/" + tag + "_data_end.dat").c_str(), "wb+");
    save_data_end(e, 1.25e-8, rho_gpu, charge_gpu, chargen_gpu, phi_gpu, ux_gpu, uy_gpu, uz_gpu, Ex_gpu, Ey_gpu, Ez_gpu, T_gpu);
    fclose(e);
    FILE* u = fopen((out + "/" + tag + "_umax.dat").c_str(), "wb+");
    record_umax(u, 1.25e-8, ux_gpu, uy_gpu, uz_gpu);
    fclose(u);
    CHECK(hipMemcpy(charge_host, charge_gpu, mem_size_scalar, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(chargen_host, chargen_gpu, mem_size_scalar, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(Ez_host, Ez_gpu, mem_size_scalar, hipMemcpyDeviceToHost));
    double cur = current(charge_host, chargen_host, Ez_host);
    FILE* g = fopen((out + "/" + tag + "_current.bin").c_str(), "wb");
    fwrite(&cur, sizeof(double), 1, g);
    fclose(g);
    printf("wrote %s io files, current = %.17g\n", tag.c_str(), cur);
  } else if (mode == "kernels" && args.size() >= 3) {
    /* per-kernel vectors: the four launches of stream_collide_save
     * (LBM.cu:474-477) one by one, populations dumped after each */
    upload(args[1].c_str());
    std::string tag = args[2];
    fast_Poisson(charge_gpu, chargen_gpu, kx, ky, kz, plan);
    dump(out + "/" + tag + "_fields0.bin");
    init_equilibrium(f0_gpu, f1_gpu, h0_gpu, h1_gpu, hn0_gpu, hn1_gpu, temp0_gpu, temp1_gpu, rho_gpu, charge_gpu, chargen_gpu, ux_gpu,
                     uy_gpu, uz_gpu, Ex_gpu, Ey_gpu, Ez_gpu, T_gpu);
    dump_pops(out + "/" + tag + "_eq.bin", 1);
    /* one full step first, so that the vectors are not those of a pure equilibrium */
    stream_collide_save(f0_gpu, f1_gpu, f2_gpu, h0_gpu, h1_gpu, h2_gpu, hn0_gpu, hn1_gpu, hn2_gpu, temp0_gpu, temp1_gpu, temp2_gpu,
                        rho_gpu, charge_gpu, chargen_gpu, ux_gpu, uy_gpu, uz_gpu, Ex_gpu, Ey_gpu, Ez_gpu, T_gpu, t, f0bc);
    dump_pops(out + "/" + tag + "_step1.bin", 1);
    dim3 grid(NX / nThreads, NY, NZ);
    dim3 threads(nThreads, 1, 1);
    gpu_collide_save<<<grid, threads>>>(f0_gpu, f1_gpu, f2_gpu, h0_gpu, h1_gpu, h2_gpu, hn0_gpu, hn1_gpu, hn2_gpu, temp0_gpu, temp1_gpu,
                                        temp2_gpu, rho_gpu, charge_gpu, chargen_gpu, ux_gpu, uy_gpu, uz_gpu, Ex_gpu, Ey_gpu, Ez_gpu, T_gpu, t, f0bc);
    dump_pops(out + "/" + tag + "_collide.bin", 2);
    dump(out + "/" + tag + "_fields_collide.bin");
    gpu_boundary<<<grid, threads>>>(f0_gpu, f1_gpu, f2_gpu, h0_gpu, h1_gpu, h2_gpu, hn0_gpu, hn1_gpu, hn2_gpu, temp0_gpu, temp1_gpu, temp2_gpu, f0bc);
    dump_pops(out + "/" + tag + "_boundary.bin", 2);
    gpu_stream<<<grid, threads>>>(f0_gpu, f1_gpu, f2_gpu, h0_gpu, h1_gpu, h2_gpu, hn0_gpu, hn1_gpu, hn2_gpu, temp0_gpu, temp1_gpu, temp2_gpu);
    dump_pops(out + "/" + tag + "_stream.bin", 1);
    gpu_bc_charge<<<grid, threads>>>(h0_gpu, h1_gpu, h2_gpu, hn0_gpu, hn1_gpu, hn2_gpu, temp0_gpu, temp1_gpu, temp2_gpu);
    dump_pops(out + "/" + tag + "_bc_charge.bin", 1);
  } else if ((mode == "time" || mode == "time0") && args.size() >= 2) {
    /* throughput of the reference's own step (main.cu:189-200, no IO) on this GPU.  "time0" starts from the
     * uniform fields of gpu_initialization alone (LBM.cu:76) instead of the 501 PB sweeps: on large grids those
     * cost 2 x 501 full-field PCIe copies, and the sweep diverges beyond ~180 planes anyway; the time of a step
     * does not depend on the values */
    const int n = atoi(args[1].c_str());
    if (mode == "time0") {
      dim3 grid0(NX / nThreads, NY, NZ);
      dim3 threads0(nThreads, 1, 1);
      gpu_initialization<<<grid0, threads0>>>(rho_gpu, charge_gpu, chargen_gpu, phi_gpu, ux_gpu, uy_gpu, uz_gpu, Ex_gpu, Ey_gpu, Ez_gpu, T_gpu);
    } else
    initialization(rho_gpu, charge_gpu, chargen_gpu, phi_gpu, ux_gpu, uy_gpu, uz_gpu, Ex_gpu, Ey_gpu, Ez_gpu, T_gpu);
    init_equilibrium(f0_gpu, f1_gpu, h0_gpu, h1_gpu, hn0_gpu, hn1_gpu, temp0_gpu, temp1_gpu, rho_gpu, charge_gpu, chargen_gpu, ux_gpu,
                     uy_gpu, uz_gpu, Ex_gpu, Ey_gpu, Ez_gpu, T_gpu);
    auto step = [&]() {
      stream_collide_save(f0_gpu, f1_gpu, f2_gpu, h0_gpu, h1_gpu, h2_gpu, hn0_gpu, hn1_gpu, hn2_gpu, temp0_gpu, temp1_gpu, temp2_gpu,
                          rho_gpu, charge_gpu, chargen_gpu, ux_gpu, uy_gpu, uz_gpu, Ex_gpu, Ey_gpu, Ez_gpu, T_gpu, t, f0bc);
      fast_Poisson(charge_gpu, chargen_gpu, kx, ky, kz, plan);
    };
    for (int i = 0; i < (mode == "time0" ? 3 : 20); ++i) step();
    checkCudaErrors(hipDeviceSynchronize());
    const double t0 = seconds_now();
    for (int i = 0; i < n; ++i) step();
    checkCudaErrors(hipDeviceSynchronize());
    const double dtw = seconds_now() - t0;
    printf("reference step on this GPU: %d steps of %ux%ux%u in %.4f s = %.3f ms/step = %.2f MLUPS\n", n, NX, NY, NZ, dtw, 1e3 * dtw / n,
           (double)n * NX * NY * NZ / dtw / 1e6);
  } else if (mode == "poisson" && args.size() >= 3) {
    upload(args[1].c_str());
    fast_Poisson(charge_gpu, chargen_gpu, kx, ky, kz, plan);
    dump(out + "/" + args[2] + ".bin");
  } else {
    fprintf(stderr, "bad mode\n");
    return 2;
  }
  checkCudaErrors(hipDeviceSynchronize());
  return 0;
}
