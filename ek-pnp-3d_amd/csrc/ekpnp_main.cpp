// ekpnp_main.cpp — thin C++ host over the C ABI that keeps main.cu's driver / IO surface.
//
// Same sequence as main.cu:19-296 of the reference: parameters -> banner -> (read previous data |
// initialization) -> init_equilibrium -> first Tecplot zone -> time loop {stream_collide_save;
// fast_Poisson; t += dt; every NSAVE steps a Tecplot zone; every printCurrent steps the wall
// current and a umax line} -> performance banner -> last zone -> data_end.dat.  Same file names
// (data.dat, umax.dat, data_end.dat), same cadence (i % NSAVE == 1, i % printCurrent == 1,
// main.cu:206,211), same text formats (the writers are in io.hip).
//
// What is different on purpose: the grid, the step count and the physics knobs are run-time
// options instead of compile-time constants (LBM.h:29-125); the "read previous data" question
// is a flag instead of scanf (main.cu:158-159); no system("pause") (main.cu:294); errors are
// reported and returned, not exit()ed from inside the library.  Only the C ABI of
// include/ekpnp.h is used: this file is also the worked example of INTEGRATION.md.
//
// --gpus N (N > 1) slab-decomposes the lattice along z over N GPUs of the node in THIS process
// (ekpnp_group_*: RCCL ring + all-gather over xGMI on a high-priority comm stream per GPU, halo
// exchange overlapped with the collision of the interior planes); same files, same formats.
// --devices 0,0,1,1 places the slabs by hand (slabs sharing a device exchange by device copies).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/ekpnp.h"

// the lattice is either one context or a group of z slabs; the driver below does not care
static ekpnp_ctx* ctx = nullptr;
static ekpnp_group* grp = nullptr;

static bool group_path = false;  // --gpus N / --devices: errors live in the group's slot, also before the group exists

static int fail(const char* what, int rc) {
  std::fprintf(stderr, "ekpnp_main: %s failed (%d): %s\n", what, rc, group_path ? ekpnp_group_last_error(grp) : ekpnp_last_error(ctx));
  if (grp) ekpnp_group_destroy(grp);
  else if (ctx) ekpnp_destroy(ctx);
  return 1;
}
#define CK(call)                                   \
  do {                                             \
    int rc_ = (call);                              \
    if (rc_ != EKPNP_OK) return fail(#call, rc_);  \
  } while (0)
// one verb, two spellings
#define RUN(verb, ...) (grp ? ekpnp_group_##verb(grp, ##__VA_ARGS__) : ekpnp_##verb(ctx, ##__VA_ARGS__))

int main(int argc, char* argv[]) {
  // LBM.h:32-35,122-125
  int nx = 50, ny = 8, nz = 51;
  unsigned nsteps = 1000, nsave = 0, print_current = 50;
  int flag = 0;  // 1: read previous data (main.cu:161); 2: from the lossless data_end.bin
  int binary_state = 0;  // also write data_end.bin (ekpnp_save_state) at the end
  int lattices = 4;
  int gpus = 1, transport = EKPNP_TRANSPORT_AUTO;
  std::vector<int> devices;
  std::string out = ".";
  double exf = 0.0, uw = 0.0, chargeinf = -1.0, Ra = -1.0, TH = -1.0;
  double converged_tol = 0.0;  // > 0: ekpnp_initialization_converged instead of the reference's fixed 501 Picard sweeps
  int batch = 0;  // 1: ekpnp_step(n) from one output mark to the next instead of one stream_collide_save + fast_Poisson pair per iteration
  std::vector<std::pair<std::string, int>> tunes;  // --tune knob=value: ekpnp_tune / ekpnp_group_tune right after creation
  for (int i = 1; i < argc; ++i) {
    auto val = [&](const char* name) -> const char* {
      if (std::strcmp(argv[i], name) == 0 && i + 1 < argc) return argv[++i];
      return nullptr;
    };
    const char* v;
    if ((v = val("--nx"))) nx = std::atoi(v);
    else if ((v = val("--ny"))) ny = std::atoi(v);
    else if ((v = val("--nz"))) nz = std::atoi(v);
    else if ((v = val("--steps"))) nsteps = (unsigned)std::atoi(v);
    else if ((v = val("--nsave"))) nsave = (unsigned)std::atoi(v);
    else if ((v = val("--print-current"))) print_current = (unsigned)std::atoi(v);
    else if ((v = val("--read-previous"))) flag = std::atoi(v);
    else if ((v = val("--binary-state"))) binary_state = std::atoi(v);
    else if ((v = val("--lattices"))) lattices = std::atoi(v);
    else if ((v = val("--gpus"))) gpus = std::atoi(v);
    else if ((v = val("--transport"))) transport = !std::strcmp(v, "rccl") ? EKPNP_TRANSPORT_RCCL : !std::strcmp(v, "copy") ? EKPNP_TRANSPORT_COPY : EKPNP_TRANSPORT_AUTO;
    else if ((v = val("--devices"))) {
      devices.clear();
      for (const char* q = v; *q;) {
        devices.push_back(std::atoi(q));
        while (*q && *q != ',') ++q;
        if (*q == ',') ++q;
      }
    }
    else if ((v = val("--out"))) out = v;
    else if ((v = val("--exf"))) exf = std::atof(v);
    else if ((v = val("--uw"))) uw = std::atof(v);
    else if ((v = val("--chargeinf"))) chargeinf = std::atof(v);
    else if ((v = val("--Ra"))) Ra = std::atof(v);
    else if ((v = val("--TH"))) TH = std::atof(v);
    else if ((v = val("--converged-init"))) converged_tol = std::atof(v);
    else if ((v = val("--batch"))) batch = std::atoi(v);
    else if ((v = val("--tune"))) {
      const char* eq = std::strchr(v, '=');
      if (!eq || eq == v) { std::fprintf(stderr, "--tune wants knob=value, got %s\n", v); return 2; }
      tunes.emplace_back(std::string(v, eq), std::atoi(eq + 1));
    }
    else {
      std::fprintf(stderr,
                   "usage: ekpnp_main [--nx N --ny N --nz N] [--steps N] [--nsave N] [--print-current N] [--read-previous 0|1|2]\n"
                   "                  [--binary-state 0|1] [--gpus N [--transport auto|rccl|copy] [--devices d0,d1,...]]\n"
                   "                  [--lattices 1|3|4] [--exf F --uw U --chargeinf C --Ra R --TH T] [--out DIR] [--converged-init TOL]\n"
                   "                  [--tune knob=value ...] [--batch 0|1]\n"
                   "  --batch 1: the time loop advances with ONE ekpnp_step(ctx, n) call from each output mark (Tecplot zone, current / umax\n"
                   "  line) to the next, with the knob batch_moments on (only the last step of a call stores rho, u, c, cn, T: nothing looks at\n"
                   "  the steps in between); the same files, byte for byte, as the default loop, which mirrors main.cu:189-224 call by call.\n"
                   "  --tune knob=value: a launch-shape or transport knob of include/ekpnp.h's ekpnp_tune (with --gpus N: on every slab), e.g.\n"
                   "  edge_chunks=4 (the slab Poisson solve's all-gather in 4 pipelined blocks), lead_planes=0, inline_exchanges=0, comm_cus=8;\n"
                   "  the results are the same bits under every setting.\n"
                   "  --converged-init TOL: Poisson-Boltzmann start-up with a convergence test and a damping that cannot diverge\n"
                   "  (ekpnp_initialization_converged); the reference's 501 sweeps with PB_omega = 0.05 (LBM.cu:89-106) diverge to NaN on\n"
                   "  channels taller than about 180 planes at the default spacing.\n"
                   "  --read-previous 1: restart from data_end.dat (%%10.6f text, the reference's); 2: from data_end.bin (--binary-state 1,\n"
                   "  the 11 fields as raw FP64).  Both restarts are the reference's (main.cu:161-175): the populations are rebuilt as the\n"
                   "  EQUILIBRIUM of the fields, so a restarted run is not the bitwise continuation of the interrupted one.\n");
      return 2;
    }
  }
  if (nsave == 0) nsave = nsteps / 2 ? nsteps / 2 : 1;  // LBM.h:123
  if (print_current == 0) print_current = 1;

  if (!devices.empty()) gpus = (int)devices.size();
  if (gpus < 1) gpus = 1;
  ekpnp_params P;
  if (ekpnp_default_params(&P, nx, ny, nz) != EKPNP_OK) return fail("ekpnp_default_params", 1);
  if (nx == 50 && ny == 8 && nz == 51) { P.Lx = 0.5e-6; P.Ly = 0.08e-6; P.Lz = 0.5e-6; }  // literals of LBM.h:40-42
  P.n_lattices = lattices;
  P.exf = exf; P.uw = uw;  // main.cu:30-31
  if (chargeinf >= 0.0) P.chargeinf = chargeinf;
  if (Ra >= 0.0) P.Ra = Ra;
  if (TH >= 0.0) P.TH = TH;

  // main.cu:40-52
  std::printf("Simulating 3D electrokinetic flow with heat transfer vortices\n");
  std::printf("      domain size (NX x NY x NZ): %ux%ux%u\n", (unsigned)nx, (unsigned)ny, (unsigned)nz);
  std::printf("               Ra: %g\n", P.Ra);
  std::printf("               Pr: %g\n", P.nu / P.D);  // compute_parameters, LBM.cu:2445
  std::printf("            uwall: %g\n", P.uw);
  std::printf("   External force: %g\n", P.exf);
  std::printf("        timesteps: %u\n", nsteps);
  std::printf("       save every: %u\n", nsave);
  std::printf("    message every: %u\n", nsave);
  std::printf("\n");

  if (gpus > 1 || !devices.empty()) {
    group_path = true;
    int rc = ekpnp_group_create(&P, gpus, devices.empty() ? nullptr : devices.data(), transport, &grp);
    if (rc != EKPNP_OK) return fail("ekpnp_group_create", rc);
  } else {
    int rc = ekpnp_create(&P, &ctx);
    if (rc != EKPNP_OK) return fail("ekpnp_create", rc);
  }
  if (batch) tunes.insert(tunes.begin(), std::make_pair(std::string("batch_moments"), 1));  // (an explicit --tune batch_moments=0 comes later and wins)
  for (const auto& kv : tunes) {
    const int rc = grp ? ekpnp_group_tune(grp, kv.first.c_str(), kv.second) : ekpnp_tune(ctx, kv.first.c_str(), kv.second);
    if (rc != EKPNP_OK) return fail(("--tune " + kv.first).c_str(), rc);
  }
  std::printf("HIP information\n");
  if (grp)
    std::printf("      z slabs: %d, halo transport: %s\n", ekpnp_group_size(grp), ekpnp_group_transport(grp) == EKPNP_TRANSPORT_RCCL ? "RCCL" : "device copies");
  std::printf("      device memory held by the solver: %.1f MiB\n\n", (double)(grp ? ekpnp_group_device_bytes(grp) : ekpnp_device_bytes(ctx)) / (1024.0 * 1024.0));

  const std::string f_data = out + "/data.dat", f_umax = out + "/umax.dat", f_end = out + "/data_end.dat";
  const std::string f_bin = out + "/data_end.bin";
  double t = 0.0;
  if (flag == 1) {  // main.cu:161-164
    std::printf("Reading previous data...\n");
    CK(RUN(read_data, f_end.c_str(), &t));
  } else if (flag == 2) {  // the same restart from the lossless file
    std::printf("Reading previous data (binary)...\n");
    CK(RUN(read_state, f_bin.c_str(), &t));
  } else {  // main.cu:165-171
    std::printf("Initializing...\n");
    if (converged_tol > 0.0) {
      int sweeps = 0;
      double res = 0.0;
      CK(RUN(initialization_converged, converged_tol, 200000, &sweeps, &res));
      std::printf("      Poisson-Boltzmann start-up: %d sweeps, relative residual %.2e\n", sweeps, res);
    } else {
      CK(RUN(initialization));
    }
    t = 0.0;
  }
  CK(RUN(set_time, t));
  CK(RUN(init_equilibrium));                             // main.cu:174
  CK(RUN(save_data_tecplot, f_data.c_str(), 0, t, 1));   // main.cu:178-179 ("wb+")
  { FILE* f = std::fopen(f_umax.c_str(), "wb"); if (f) std::fclose(f); }  // main.cu:180

  CK(RUN(synchronize));
  const auto begin = std::chrono::steady_clock::now();  // main.cu:185-186
  for (unsigned i = 0; i < nsteps; i++) {               // main.cu:189-224
    if (batch) {
      // iterations i .. j in one call, j = the next iteration something looks at the fields (or the last one)
      unsigned j = i;
      while (j + 1 < nsteps && !(j % nsave == 1 || j % print_current == 1)) ++j;
      CK(RUN(step, (int)(j - i + 1)));
      for (unsigned k = i; k <= j; ++k) t = t + P.dt;  // the same additions as the loop below makes, so the files carry the same time
      i = j;
    } else {
      CK(RUN(stream_collide_save, t));
      CK(RUN(fast_poisson));
      t = t + P.dt;
    }
    if (i % nsave == 1) {
      CK(RUN(save_data_tecplot, f_data.c_str(), 1, t, 1));
      std::printf("Iteration: %u, physical time: %g.\n", i, t);
    }
    if (i % print_current == 1) {
      double I = 0.0;
      CK(RUN(current, &I));  // reduced on the device (main.cu:212-215 copies 3 fields to the host)
      std::printf("Iteration: %u, physical time: %g, Current = %g\n", i, t, I);
      CK(RUN(record_umax, f_umax.c_str(), 1, t));
    }
  }
  CK(RUN(synchronize));
  const double runtime = std::chrono::duration<double>(std::chrono::steady_clock::now() - begin).count();

  // main.cu:241-251
  const double nodes_updated = (double)nsteps * (double)nx * (double)ny * (double)nz;
  std::printf(" ----- performance information -----\n");
  std::printf("               timesteps: %u\n", nsteps);
  std::printf("           clock runtime: %.3f (s)\n", runtime);
  std::printf("                   speed: %.2f (Mlups)\n", nodes_updated / (1e6 * runtime));

  CK(RUN(save_data_tecplot, f_data.c_str(), 1, t, 1));  // main.cu:253
  CK(RUN(save_data_end, f_end.c_str(), 0, t));          // main.cu:256-257
  if (binary_state) CK(RUN(save_state, f_bin.c_str(), t));
  CK(grp ? ekpnp_group_destroy(grp) : ekpnp_destroy(ctx));    // main.cu:264-290
  return 0;
}
