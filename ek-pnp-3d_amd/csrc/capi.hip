// capi.hip — the C ABI of include/ekpnp.h: context, memory, plans, step orchestration.
// Host-side replacement of main.cu:58-152,189-200,264-290 and of the host wrappers
// initialization / init_equilibrium / stream_collide_save / fast_Poisson
// (LBM.cu:68-109,150-160,465-481; poisson.cu:75-103).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "ekpnp_internal.h"

using namespace ekpnp;

static thread_local std::string g_create_err;

#define HIPCHK(ctx, call)                                                                       \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      (ctx).err = std::string(#call) + ": " + hipGetErrorString(e_);                            \
      return e_ == hipErrorOutOfMemory ? EKPNP_ERR_NOMEM : EKPNP_ERR_HIP;                       \
    }                                                                                           \
  } while (0)
// after the launches of an entry point: the first rejected launch, by kernel name
#define LAUNCHCHK(ctx)                                                                          \
  do {                                                                                          \
    if (take_launch_error(ctx) != hipSuccess) {                                                 \
      if ((ctx).err.empty()) (ctx).err = "a kernel launch failed";                              \
      return EKPNP_ERR_HIP;                                                                     \
    }                                                                                           \
  } while (0)

#define FFTCHK(ctx, call)                                                                       \
  do {                                                                                          \
    hipfftResult r_ = (call);                                                                   \
    if (r_ != HIPFFT_SUCCESS) {                                                                 \
      (ctx).err = std::string(#call) + ": hipfft error " + std::to_string((int)r_);             \
      return EKPNP_ERR_FFT;                                                                     \
    }                                                                                           \
  } while (0)

#define NEEDCTX(ctx)                                                                            \
  if (!(ctx)) return EKPNP_ERR_INVALID;                                                         \
  Ctx& c = (ctx)->c

static int fail(Ctx& c, const char* msg) {
  c.err = msg;
  return EKPNP_ERR_INVALID;
}

namespace ekpnp {
void set_create_error(const std::string& msg) { g_create_err = msg; }
void note_launch(Ctx& c, const char* kernel) {
  static const bool debug_sync = std::getenv("EKPNP_DEBUG_SYNC") != nullptr;
  // fault injection for the tests of this very path: EKPNP_INJECT_LAUNCH_FAILURE=<kernel name>[@<slab rank>][#<n>]
  // (only on the context of that rank; only the n-th launch of that kernel on a context, counted from 1)
  struct Inject { std::string name; int rank = -1; long nth = 0; bool on = false; };
  static const Inject inject = [] {
    Inject j;
    if (const char* v = std::getenv("EKPNP_INJECT_LAUNCH_FAILURE")) {
      std::string t = v;
      const size_t h = t.rfind('#');
      if (h != std::string::npos) { j.nth = std::atol(t.c_str() + h + 1); t.erase(h); }
      const size_t r = t.rfind('@');
      if (r != std::string::npos) { j.rank = std::atoi(t.c_str() + r + 1); t.erase(r); }
      j.name = t;
      j.on = !t.empty();
    }
    return j;
  }();
  hipError_t e = hipGetLastError();
  if (inject.on && e == hipSuccess && inject.name == kernel && (inject.rank < 0 || inject.rank == c.rank)) {
    ++c.inject_seen;
    if (inject.nth <= 0 || c.inject_seen == inject.nth) e = hipErrorLaunchFailure;
  }
  if (e == hipSuccess && debug_sync) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(c.stream, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) e = hipStreamSynchronize(c.stream);
  }
  if (e != hipSuccess && c.launch_err == hipSuccess) {
    c.launch_err = e;
    c.launch_what = kernel;
  }
}
hipError_t take_launch_error(Ctx& c) {
  hipError_t e = c.launch_err;
  if (e != hipSuccess) {
    c.err = std::string("kernel ") + (c.launch_what ? c.launch_what : "?") + ": " + hipGetErrorString(e);
    c.launch_err = hipSuccess;
    c.launch_what = nullptr;
    (void)hipGetLastError();
    return e;
  }
  return hipGetLastError();
}
}  // namespace ekpnp

static void drop_graph(Ctx& c);

namespace ekpnp {
int ctx_tune(Ctx& c, const char* knob, int value) {
  if (std::strcmp(knob, "ab_zchunk") == 0 && value >= 0) { c.ab_zchunk = value; return EKPNP_OK; }
  if (std::strcmp(knob, "bulk_yband") == 0 && value >= -1 && value < 0x7fff) { c.bulk_yband = value; drop_graph(c); return EKPNP_OK; }
  // slab launch shapes (same bits either way): the lead-in launch of the interior sweep, one launch for both faces
  if (std::strcmp(knob, "lead_planes") == 0 && value >= 0 && c.slab) { c.lead_planes = value; return EKPNP_OK; }
  if (std::strcmp(knob, "merged_faces") == 0 && (value == 0 || value == 1) && c.slab) { c.merged_faces = value != 0; return EKPNP_OK; }
  // mode blocks of the slab z solve: the layout of the edge buffers changes with it, so only a context whose exchanges the
  // library moves itself may ask for more than one (an external transport gathers the whole buffer in one piece)
  if (std::strcmp(knob, "edge_chunks") == 0 && value >= 1 && value <= 16 && c.slab && (value == 1 || c.team)) { c.edge_chunks = value; return EKPNP_OK; }
  if (std::strcmp(knob, "poisson_blocks") == 0 && value >= 0 && value <= 256 && !c.slab) { c.poisson_blocks = value; drop_graph(c); return EKPNP_OK; }
  if (std::strcmp(knob, "poisson_zchunk") == 0 && value >= 0 && !c.slab) { c.poisson_zchunk = value; drop_graph(c); return EKPNP_OK; }
  if (std::strcmp(knob, "merged_walls") == 0) { c.merged_walls = value != 0; drop_graph(c); return EKPNP_OK; }
  if (std::strcmp(knob, "tri_partition") == 0 && value >= 0 && value <= 2) { c.tri_partition = value; drop_graph(c); return EKPNP_OK; }
  if (std::strcmp(knob, "tri_wide") == 0 && (value == 0 || value == 1)) { c.tri_wide = value != 0 && c.tri_lds_ok && tridiag_wide_prepare_device(); drop_graph(c); return EKPNP_OK; }
  if (std::strcmp(knob, "batch_moments") == 0 && (value == 0 || value == 1)) { c.batch_moments = value != 0; return EKPNP_OK; }
  if (std::strcmp(knob, "lazy_efield") == 0 && (value == 0 || value == 1)) {  // the A/B partner of the EPHI kernels: 0 = k_phi_efield in every solve
    const int rc = ensure_efield(c);
    c.lazy_efield = value;
    drop_graph(c);
    return rc;
  }
  c.err = std::string("ekpnp_tune: unknown knob or bad value: ") + knob + " = " + std::to_string(value);
  return EKPNP_ERR_INVALID;
}
}  // namespace ekpnp

// measurement hook (tools/sweep_zchunk.py, bench.py's comm_ab): change a launch-shape or transport knob of a live context
extern "C" int ekpnp_tune(ekpnp_ctx* ctx, const char* knob, int value) {
  if (!ctx || !knob) return EKPNP_ERR_INVALID;
  Ctx& c = ctx->c;
  // the transport's own knobs first (an attached slab; the members of an in-process group are tuned through ekpnp_group_tune)
  if (c.team && (std::strcmp(knob, "inline_exchanges") == 0 || std::strcmp(knob, "comm_cus") == 0 || std::strcmp(knob, "edge_p2p") == 0)) return team_ctx_tune(c, knob, value);
  return ctx_tune(c, knob, value);
}

extern "C" int ekpnp_debug_sync_enabled(void) { return std::getenv("EKPNP_DEBUG_SYNC") != nullptr; }

static void drop_graph(Ctx& c) {
  if (c.graph2) { (void)hipGraphExecDestroy(c.graph2); c.graph2 = nullptr; }
  c.graph_cur = -1;
}

// ---- lazy E (round 4; Ctx::e_stale) -------------------------------------------------------------------------------
namespace ekpnp {
bool lazy_efield_ok(const Ctx& c) {
  return c.lazy_efield != 0 && c.p.n_lattices > 1 && !c.e_exposed && c.fld_owned[EKPNP_PHI] && c.fld_owned[EKPNP_EX] &&
         c.fld_owned[EKPNP_EY] && c.fld_owned[EKPNP_EZ];
}
bool batch_moments_ok(const Ctx& c) {
  if (!c.batch_moments || c.mom_exposed || c.timing) return false;  // (measurement runs keep every step alike)
  for (int id : {EKPNP_RHO, EKPNP_UX, EKPNP_UY, EKPNP_UZ, EKPNP_C, EKPNP_CN, EKPNP_T})
    if (!c.fld_owned[id]) return false;  // a caller's own array (ekpnp_bind_field) may be read on the device at any time
  return true;
}
int ensure_efield(Ctx& c) {
  if (!c.e_stale) return EKPNP_OK;
  launch_phi_efield(c);  // odd_extract's plates + gpu_efield + gpu_bc (poisson.cu:40-69,198-203) from the phi the last solve left
  c.e_stale = false;
  LAUNCHCHK(c);
  return EKPNP_OK;
}
int efield_set_from_outside(Ctx& c) {
  int rc = ensure_efield(c);  // a caller that sets ONE of phi / Ex / Ey / Ez finds the other three as the reference would have them
  c.e_phi_valid = false;
  return rc;
}
}  // namespace ekpnp
static inline bool is_phi_or_e(int id) { return id == EKPNP_PHI || id == EKPNP_EX || id == EKPNP_EY || id == EKPNP_EZ; }
// the state a solve leaves behind: lazily (E lives in phi until somebody looks) or with the arrays written
static inline void mark_solved(Ctx& c, bool lazy) {
  c.e_stale = lazy;
  c.e_phi_valid = true;
}

// ------------------------------------------------------------------------------------------

extern "C" int ekpnp_default_params(ekpnp_params* p, int nx, int ny, int nz) {
  if (!p || nx < 1 || ny < 1 || nz < 4) return EKPNP_ERR_INVALID;
  std::memset(p, 0, sizeof(*p));
  // LBM.h:29-118, same literal expressions
  p->nx = nx; p->ny = ny; p->nz = nz;
  p->n_lattices = 4;
  p->pb_iterations = 501;
  p->dx = 1.0e-6 / 100.0; p->dy = 1.0e-6 / 100.0; p->dz = 1.0e-6 / 100.0;
  p->Lx = nx * p->dx; p->Ly = ny * p->dy; p->Lz = (nz - 1) * p->dz;
  p->CFL = 0.01;
  p->dt = 0.01 * 1.0e-6 / 100.0;
  p->cs_square = 1.0 / 3.0 / (0.01 * 0.01);
  p->rho0 = 1000.0;
  p->chargeinf = 0.01;
  p->voltage = -5.2574e-3; p->voltage2 = -5.2574e-3;
  p->Ext = 1.0e4;
  p->eps = 6.95e-10;
  p->diffu = 1.0e-8; p->diffun = 1.0e-8;
  p->nu = 0.889e-6;
  p->K = 4.245e-7; p->Kn = -4.245e-7;
  p->D = 0.889e-6; p->Ra = 1; p->TH = 1;
  p->uw = 0.0; p->exf = 0.0;
  p->kB = 1.38e-23; p->electron = 1.6e-19; p->roomT = 273.0;
  p->convertCtoCharge = 9.64e4;
  p->PB_omega = 0.05;
  p->V = 1.0 / 12.0; p->VC = 1.0e-6; p->VCn = 1.0e-6; p->VT = 1.0 / 12.0;
  return EKPNP_OK;
}

KArgs Ctx::kargs() const {
  KArgs a{};
  for (int l = 0; l < MAXL; ++l) {
    a.A[l] = cur_base(l);
    a.B[l] = next_base(l);
  }
  for (int i = 0; i < EKPNP_NFIELDS; ++i) a.fld[i] = fld[i];
  a.nx = p.nx; a.ny = p.ny; a.nz = p.nz;
  a.nzl = nzl; a.z0 = z0;
  // ghost planes are filled by the halo transport (slabs) or by k_ghost_wrap right after the sweep
  // (in-place mode: the opposite wall plane of the old state may be overwritten by then); a plain
  // two-buffer context needs neither: its wall nodes index the opposite wall plane
  a.zwrap = (!slab && !inplace) ? 1 : 0;
  a.plane = (long long)plane;
  a.rowstride = (long long)rowstride;
  const double cs2 = p.cs_square, dt = p.dt;
  // relaxation rates, LBM.cu:488-495 (same expression order)
  const double omega_plus = 1.0 / (p.nu / cs2 / dt + 1.0 / 2.0) / dt;
  const double omega_minus = 1.0 / (p.V / (p.nu / cs2 / dt) + 1.0 / 2.0) / dt;
  const double omega_c_minus = 1.0 / (p.diffu / cs2 / dt + 1.0 / 2.0) / dt;
  const double omega_c_plus = 1.0 / (p.VC / (p.diffu / cs2 / dt) + 1.0 / 2.0) / dt;
  const double omega_cn_minus = 1.0 / (p.diffun / cs2 / dt + 1.0 / 2.0) / dt;
  const double omega_cn_plus = 1.0 / (p.VCn / (p.diffun / cs2 / dt) + 1.0 / 2.0) / dt;
  const double omega_T_minus = 1.0 / (p.D / cs2 / dt + 1.0 / 2.0) / dt;
  const double omega_T_plus = 1.0 / (p.VT / (p.D / cs2 / dt) + 1.0 / 2.0) / dt;
  a.wp[0] = omega_plus * dt;    a.wm[0] = omega_minus * dt;     // LBM.cu:1700-1707
  a.wp[1] = omega_c_plus * dt;  a.wm[1] = omega_c_minus * dt;
  a.wp[2] = omega_cn_plus * dt; a.wm[2] = omega_cn_minus * dt;
  a.wp[3] = omega_T_plus * dt;  a.wm[3] = omega_T_minus * dt;
  a.mob[0] = 0.0; a.mob[1] = p.K; a.mob[2] = p.Kn; a.mob[3] = 0.0;
  a.sp = 1.0 - 0.5 * dt * omega_plus;   // LBM.cu:1660-1661
  a.sm = 1.0 - 0.5 * dt * omega_minus;
  a.cflinv = 1.0 / p.CFL;               // LBM.cu:1112
  a.inv_cs2 = 1.0 / cs2;
  a.cflinv2 = a.cflinv * a.cflinv / cs2;  // LBM.cu:1115
  a.dt = dt;
  a.F = p.convertCtoCharge; a.Ext = p.Ext; a.exf = p.exf;
  a.rho0 = p.rho0; a.Ra = p.Ra; a.nu = p.nu; a.D = p.D;
  a.TH = p.TH;
  a.uw_multi = 2.0 * p.rho0 * p.uw / cs2 / p.CFL;  // LBM.cu:1896-1898 without the weight
  a.wmom = skip_moments ? 0 : 1;
  a.rhs = p.n_lattices > 1 ? work : nullptr;
  a.eps = p.eps;
  a.rhs_wall_lo = p.voltage / p.dz / p.dz;    // poisson.cu:124
  a.rhs_wall_hi = p.voltage2 / p.dz / p.dz;   // poisson.cu:134
  a.phi_lo = phi_halo[2]; a.phi_hi = phi_halo[3];
  a.voltage = p.voltage; a.voltage2 = p.voltage2;
  a.dx = p.dx; a.dy = p.dy; a.dz = p.dz;
  return a;
}

PArgs Ctx::pargs() const {
  PArgs a{};
  for (int i = 0; i < EKPNP_NFIELDS; ++i) a.fld[i] = fld[i];
  a.work = work; a.spec = spec; a.cprime = cprime;
  a.phi_lo = phi_halo[2]; a.phi_hi = phi_halo[3];
  a.vwall = vwall;
  a.nx = p.nx; a.ny = p.ny; a.nz = p.nz; a.nxh = nxh; a.nzl = nzl; a.z0 = z0;
  a.plane = (long long)plane;
  a.F = p.convertCtoCharge; a.eps = p.eps; a.voltage = p.voltage; a.voltage2 = p.voltage2;
  a.rhs_wall_lo = p.voltage / p.dz / p.dz;
  a.rhs_wall_hi = p.voltage2 / p.dz / p.dz;
  a.dx = p.dx; a.dy = p.dy; a.dz = p.dz;
  a.Lx = p.Lx; a.Ly = p.Ly;
  a.inv_nxny = 1.0 / ((double)p.nx * (double)p.ny);
  a.bx0 = 0; a.bw = nxh;
  return a;
}

static int dev_alloc(Ctx& c, void** ptr, size_t bytes) {
  HIPCHK(c, hipMalloc(ptr, bytes));
  c.bytes += bytes;
  return EKPNP_OK;
}

static int validate(const ekpnp_params* p, int rank, int nranks, std::string& err) {
  if (!p) { err = "params is NULL"; return EKPNP_ERR_INVALID; }
  if (p->nx < 1 || p->ny < 1 || p->nz < 4) { err = "grid must be at least 1 x 1 x 4"; return EKPNP_ERR_INVALID; }
  if (p->ny > 65535) { err = "ny must be at most 65535 (a grid dimension of the wall and halo kernels)"; return EKPNP_ERR_INVALID; }
  if (p->n_lattices != 1 && p->n_lattices != 3 && p->n_lattices != 4) { err = "n_lattices must be 1, 3 or 4"; return EKPNP_ERR_INVALID; }
  if (p->n_lattices < 4 && p->Ra != 0.0) { err = "n_lattices < 4 requires Ra == 0 (temperature feeds the buoyancy force, LBM.cu:637)"; return EKPNP_ERR_INVALID; }
  if (p->n_lattices == 1 && p->chargeinf != 0.0) { err = "n_lattices == 1 requires chargeinf == 0"; return EKPNP_ERR_INVALID; }
  if (nranks < 1 || rank < 0 || rank >= nranks) { err = "bad rank/nranks"; return EKPNP_ERR_INVALID; }
  if (nranks > 1 && p->nz / nranks < 4) { err = "each z slab needs at least 4 planes"; return EKPNP_ERR_INVALID; }
  if (!(p->dx > 0 && p->dy > 0 && p->dz > 0 && p->dt > 0 && p->CFL > 0 && p->cs_square > 0 && p->Lx > 0 && p->Ly > 0 && p->Lz > 0)) {
    err = "dx, dy, dz, dt, CFL, cs_square, Lx, Ly, Lz must be positive"; return EKPNP_ERR_INVALID;
  }
  if (p->pb_iterations < 0) { err = "pb_iterations must be >= 0"; return EKPNP_ERR_INVALID; }
  if (p->in_place != 0 && p->in_place != 1) { err = "in_place must be 0 or 1"; return EKPNP_ERR_INVALID; }
  return EKPNP_OK;
}

static int placement_search(Ctx& c, size_t pitch, int nbuf);

namespace ekpnp {
// the rocFFT plans (hipFFT API) of the batched 2-D real transforms over the owned interior planes
int make_fft_plans(Ctx& c) {
  if (c.plans) return EKPNP_OK;
  const ekpnp_params* p = &c.p;
  int n[2] = {p->ny, p->nx};
  int rembed[2] = {p->ny, p->nx};      // real planes, dense
  int cembed[2] = {p->ny, c.nxh};      // half spectrum with the padded row pitch
  hipfftResult r = hipfftPlanMany(&c.plan_fwd, 2, n, rembed, 1, p->ny * p->nx, cembed, 1, p->ny * c.nxh, HIPFFT_D2Z, c.fft_nz);
  if (r != HIPFFT_SUCCESS) { c.plan_fwd = 0; c.err = "hipfftPlanMany (D2Z) failed: " + std::to_string((int)r); return EKPNP_ERR_FFT; }
  c.have_fwd = true;
  r = hipfftPlanMany(&c.plan_inv, 2, n, cembed, 1, p->ny * c.nxh, rembed, 1, p->ny * p->nx, HIPFFT_Z2D, c.fft_nz);
  if (r != HIPFFT_SUCCESS) { c.plan_inv = 0; c.err = "hipfftPlanMany (Z2D) failed: " + std::to_string((int)r); return EKPNP_ERR_FFT; }
  c.have_inv = true;
  c.plans = true;
  hipfftSetStream(c.plan_fwd, c.stream);
  hipfftSetStream(c.plan_inv, c.stream);
  // the work areas rocFFT allocated for the two plans are device memory held by the solver too
  size_t ws = 0;
  if (hipfftGetSize(c.plan_fwd, &ws) == HIPFFT_SUCCESS) c.bytes += ws;
  ws = 0;
  if (hipfftGetSize(c.plan_inv, &ws) == HIPFFT_SUCCESS) c.bytes += ws;
  return EKPNP_OK;
}

}  // namespace ekpnp

extern "C" int ekpnp_plane_transforms(const ekpnp_ctx* ctx, int* own_passes, int* ranks_on_device) {
  if (!ctx) return EKPNP_ERR_INVALID;
  if (own_passes) *own_passes = ctx->c.own_fft ? 1 : 0;
  if (ranks_on_device) *ranks_on_device = ctx->c.ranks_on_device;
  return EKPNP_OK;
}

extern "C" int ekpnp_pass_order(const ekpnp_ctx* ctx, int* band_rows, int* poisson_blocks, int* poisson_zchunk) {
  if (!ctx) return EKPNP_ERR_INVALID;
  const Ctx& c = ctx->c;
  if (band_rows) *band_rows = bulk_band_rows(c);
  if (poisson_blocks) *poisson_blocks = poisson_block_count(c);
  if (poisson_zchunk) *poisson_zchunk = !c.slab && c.own_fft && c.poisson_zchunk > 0 && c.poisson_zchunk < c.fft_nz ? c.poisson_zchunk : 0;
  return EKPNP_OK;
}

// the population buffers inside their one allocation: A0 A1 A2 A3 B0 B1 B2 B3 (buffer-major; EKPNP_POP_ORDER=1, an
// experiment of round 3: lattice-major A0 B0 A1 B1 ... - no difference, profiles/r03_direction_sweep.log)
static void carve_arena(Ctx& c, void* base, size_t pitch) {
  static const bool lattice_major = std::getenv("EKPNP_POP_ORDER") != nullptr && std::atoi(std::getenv("EKPNP_POP_ORDER")) == 1;
  const int nb = c.inplace ? 1 : 2, nl = c.p.n_lattices;
  for (int b = 0; b < nb; ++b)
    for (int l = 0; l < nl; ++l) {
      const int k = lattice_major ? l * nb + b : b * nl + l;
      c.pop[b][l] = (double*)((char*)base + pitch * k);
    }
}

static int create_impl(const ekpnp_params* p, int rank, int nranks, bool slab, ekpnp_ctx** out) {
  if (!out) { g_create_err = "out is NULL"; return EKPNP_ERR_INVALID; }
  *out = nullptr;
  int rc = validate(p, rank, nranks, g_create_err);
  if (rc) return rc;
  ekpnp_ctx* h = new (std::nothrow) ekpnp_ctx();
  if (!h) { g_create_err = "host allocation failed"; return EKPNP_ERR_NOMEM; }
  Ctx& c = h->c;
  c.p = *p;
  c.rank = rank; c.nranks = nranks;
  c.slab = slab;
  c.z0 = slab_begin(p->nz, nranks, rank);
  c.nzl = slab_begin(p->nz, nranks, rank + 1) - c.z0;
  // row pitch of the half spectrum: NX/2+1 complex, padded to a multiple of 8 (128 bytes) so that
  // rocFFT's strided y pass and the mode-parallel z sweeps work on aligned rows; the pad columns
  // are zero and stay zero (the transforms never touch them, the z solve maps 0 to 0)
  c.nxh = (p->nx / 2 + 1 + 7) / 8 * 8;
  c.plane = (size_t)p->nx * p->ny;
  c.nloc = c.plane * c.nzl;
  c.rowstride = (size_t)((p->nx + 63) / 64) * TILE;
  c.pplane = c.rowstride * p->ny;
  auto bail = [&](int code) {
    g_create_err = c.err;
    ekpnp_destroy(h);
    return code;
  };
  if (hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking) != hipSuccess) {
    c.err = "hipStreamCreate failed (no HIP device?)";
    return bail(EKPNP_ERR_HIP);
  }
  c.own_stream = true;
  if (hipGetDevice(&c.device) != hipSuccess) { c.err = "hipGetDevice failed"; return bail(EKPNP_ERR_HIP); }
  c.tri_lds_ok = tridiag_prepare_device();
  if (hipDeviceGetAttribute(&c.ncus, hipDeviceAttributeMultiprocessorCount, c.device) != hipSuccess || c.ncus < 1) { (void)hipGetLastError(); c.ncus = 256; }
  if (const char* e = std::getenv("EKPNP_BULK_ZCHUNK")) c.ab_zchunk = std::atoi(e) > 0 ? std::atoi(e) : 0;
  if (const char* e = std::getenv("EKPNP_BULK_YBAND")) c.bulk_yband = std::atoi(e) < -1 ? -1 : (std::atoi(e) > 0x7ffe ? 0x7ffe : std::atoi(e));
  c.merged_walls = std::getenv("EKPNP_NO_MERGED_WALLS") == nullptr;
  if (const char* e = std::getenv("EKPNP_LAZY_E")) c.lazy_efield = std::atoi(e) != 0 ? 1 : 0;
  if (const char* e = std::getenv("EKPNP_HALO_DIRECT")) c.halo_direct = std::atoi(e) != 0;
  if (const char* e = std::getenv("EKPNP_BATCH_MOMENTS")) c.batch_moments = std::atoi(e) != 0;
  if (const char* e = std::getenv("EKPNP_MERGED_FACES")) c.merged_faces = std::atoi(e) != 0;
  if (const char* e = std::getenv("EKPNP_SLAB_LEAD_PLANES")) c.lead_planes = std::atoi(e) < 0 ? 0 : std::atoi(e);
  if (const char* e = std::getenv("EKPNP_POISSON_BLOCKS")) c.poisson_blocks = std::atoi(e) < 0 ? 0 : (std::atoi(e) > 256 ? 256 : std::atoi(e));
  if (const char* e = std::getenv("EKPNP_POISSON_ZCHUNK")) c.poisson_zchunk = std::atoi(e) < 0 ? 0 : std::atoi(e);
  if (const char* e = std::getenv("EKPNP_EDGE_CHUNKS")) c.edge_chunks = std::atoi(e) < 1 ? 1 : (std::atoi(e) > 16 ? 16 : std::atoi(e));
  {
    const char* e = std::getenv("EKPNP_TRI_WIDE");
    c.tri_wide = c.tri_lds_ok && (e ? std::atoi(e) != 0 : false) && tridiag_wide_prepare_device();
  }
  if (const char* e = std::getenv("EKPNP_TRI_PARTITION")) c.tri_partition = std::atoi(e) < 0 ? 0 : (std::atoi(e) > 2 ? 2 : std::atoi(e));
  // In-place mode: one buffer per lattice with `shift` spare planes.  A sweep writes plane z of
  // the new state `shift` planes below (parity 0, bulk launches of `zchunk` planes in ascending z)
  // or above (parity 1, descending) where plane z of the old state lies; shift >= zchunk + 1
  // guarantees that a launch only overwrites planes every later launch is done with.
  c.inplace = p->in_place != 0;
  c.zchunk = c.inplace ? (c.nzl / 4 < 1 ? 1 : c.nzl / 4 > 64 ? 64 : c.nzl / 4) : 0;
  c.shift = c.inplace ? c.zchunk + 1 : 0;
  const size_t popbytes = (size_t)(c.nzl + 2 + c.shift) * c.pplane * sizeof(double);
  // (skewing the population arrays against each other like the macroscopic arrays below was
  // measured in round 1: -1.5 %, profiles/r01_sweep_skew.log)
  // All population buffers are carved out of ONE allocation.  With a hipMalloc per buffer (8 of 29 GB on
  // cfg3) the step time of otherwise identical contexts spread over 42.5 ... 44.2 ms depending on where
  // the driver happened to place them; out of one 233 GB allocation it is 42.5 ... 42.7 ms, every time
  // (profiles/r02_population_arena.log).  EKPNP_POP_ARENA=<bytes> puts a gap between the buffers (0 and
  // 4096 measured the same), EKPNP_POP_ARENA=-1 restores one allocation per buffer (the A/B partner).
  static const long long arena_gap = std::getenv("EKPNP_POP_ARENA") ? std::atoll(std::getenv("EKPNP_POP_ARENA")) : 0;
  bool arena = arena_gap >= 0;
  size_t pop_pitch = 0;
  int pop_nbuf = 0;
  if (arena) {
    const int nbuf = (c.inplace ? 1 : 2) * p->n_lattices;
    const size_t pitch = (popbytes + (size_t)arena_gap + 255) / 256 * 256;
    pop_pitch = pitch;
    pop_nbuf = nbuf;
    // EKPNP_POP_CONTIGUOUS=1 (experiment, round 3): ask the driver for physically contiguous VRAM (hipDeviceMallocContiguous)
    static const bool want_contig = std::getenv("EKPNP_POP_CONTIGUOUS") != nullptr && std::atoi(std::getenv("EKPNP_POP_CONTIGUOUS")) != 0;
    bool got = false;
    if (want_contig) {
      got = hipExtMallocWithFlags(&c.pop_alloc[0][0], pitch * nbuf, hipDeviceMallocContiguous) == hipSuccess;
      if (!got) { (void)hipGetLastError(); c.pop_alloc[0][0] = nullptr; }
      if (std::getenv("EKPNP_DEBUG_ARENA")) std::fprintf(stderr, "ekpnp: contiguous arena %s\n", got ? "granted" : "REFUSED, plain hipMalloc");
    }
    if (got || hipMalloc(&c.pop_alloc[0][0], pitch * nbuf) == hipSuccess) {
      c.bytes += pitch * nbuf;
      if (std::getenv("EKPNP_DEBUG_ARENA")) std::fprintf(stderr, "ekpnp: population arena %p, %zu buffers of %zu bytes\n", c.pop_alloc[0][0], (size_t)nbuf, pitch);
      carve_arena(c, c.pop_alloc[0][0], pitch);
    } else {  // no contiguous range of that size (a fragmented device): one allocation per buffer may still fit
      (void)hipGetLastError();
      c.pop_alloc[0][0] = nullptr;
      arena = false;
    }
  }
  if (!arena) {
    for (int b = 0; b < (c.inplace ? 1 : 2); ++b)
      for (int l = 0; l < p->n_lattices; ++l) {
        if ((rc = dev_alloc(c, &c.pop_alloc[b][l], popbytes))) return bail(rc);
        c.pop[b][l] = (double*)c.pop_alloc[b][l];
      }
  }
  if (c.inplace && slab)
    for (int l = 0; l < p->n_lattices; ++l)
      if ((rc = dev_alloc(c, (void**)&c.stage[l], 2 * c.pplane * sizeof(double)))) return bail(rc);
  // The 11 macroscopic arrays are equally sized (a power of two bytes on cfg2/cfg3) and are walked
  // in lockstep by the kernels; identical placement modulo the HBM channel interleave makes all of
  // their streams queue on the same channels.  Each owned array is therefore skewed by a different
  // multiple of `skew` bytes inside a slightly larger allocation (EKPNP_FIELD_SKEW: tuning knob).
  static const size_t skew = std::getenv("EKPNP_FIELD_SKEW") ? (size_t)std::atoll(std::getenv("EKPNP_FIELD_SKEW")) : 4096;
  // EKPNP_FIELD_ARENA=1 (tuning experiment): the 11 arrays out of one allocation, same skew
  static const bool field_arena = std::getenv("EKPNP_FIELD_ARENA") != nullptr && std::atoi(std::getenv("EKPNP_FIELD_ARENA")) != 0;
  if (field_arena) {
    const size_t pitch = (c.nloc * sizeof(double) + (size_t)EKPNP_NFIELDS * skew + 255) / 256 * 256;
    if ((rc = dev_alloc(c, &c.fld_arena, pitch * EKPNP_NFIELDS))) return bail(rc);
    for (int i = 0; i < EKPNP_NFIELDS; ++i) {
      c.fld[i] = (double*)((char*)c.fld_arena + pitch * i + (size_t)i * skew);
      c.fld_owned[i] = true;
      if (hipMemsetAsync(c.fld[i], 0, c.nloc * sizeof(double), c.stream) != hipSuccess) { c.err = "hipMemsetAsync failed"; return bail(EKPNP_ERR_HIP); }
    }
  } else
  for (int i = 0; i < EKPNP_NFIELDS; ++i) {
    c.fld_bytes[i] = c.nloc * sizeof(double) + (size_t)EKPNP_NFIELDS * skew;
    if ((rc = dev_alloc(c, &c.fld_alloc[i], c.fld_bytes[i]))) { c.fld_bytes[i] = 0; return bail(rc); }
    c.fld[i] = (double*)((char*)c.fld_alloc[i] + (size_t)i * skew);
    c.fld_owned[i] = true;
    if (hipMemsetAsync(c.fld[i], 0, c.nloc * sizeof(double), c.stream) != hipSuccess) { c.err = "hipMemsetAsync failed"; return bail(EKPNP_ERR_HIP); }
  }
  if ((rc = dev_alloc(c, (void**)&c.work, c.nloc * sizeof(double)))) return bail(rc);
  if ((rc = dev_alloc(c, (void**)&c.spec, (size_t)c.nzl * p->ny * c.nxh * sizeof(double2)))) return bail(rc);
  if (hipMemsetAsync(c.spec, 0, (size_t)c.nzl * p->ny * c.nxh * sizeof(double2), c.stream) != hipSuccess) { c.err = "hipMemsetAsync failed"; return bail(EKPNP_ERR_HIP); }
  const size_t nmodes = (size_t)p->ny * c.nxh;
  // slabs: the full table of their local rows; one context: only the rows its z solve restarts from
  int rows_max = 0;  // the longest block of unknown rows among the ranks (set-up of their (u_1, u_m))
  for (int r = 0; r < nranks; ++r) rows_max = slab_rows(p->nz, nranks, r) > rows_max ? slab_rows(p->nz, nranks, r) : rows_max;
  const size_t cprime_rows = !slab ? (size_t)(p->nz - 2) / TRI_CHECK + 1 : (size_t)rows_max + 2;
  if ((rc = dev_alloc(c, (void**)&c.cprime, cprime_rows * nmodes * sizeof(double)))) return bail(rc);
  {
    const double vw[4] = {p->voltage, p->voltage, p->voltage2, p->voltage2};  // pairs: the two-nodes-per-lane phi/E kernel loads 16 bytes
    if ((rc = dev_alloc(c, (void**)&c.vwall, sizeof(vw)))) return bail(rc);
    if (hipMemcpy(c.vwall, vw, sizeof(vw), hipMemcpyHostToDevice) != hipSuccess) { c.err = "hipMemcpy failed"; return bail(EKPNP_ERR_HIP); }
  }
  c.halo_doubles = (size_t)p->n_lattices * 9 * c.plane;
  if (slab) {
    if (nranks > 16) { c.err = "at most 16 z slabs"; return bail(EKPNP_ERR_INVALID); }
    for (int k = 0; k < 4; ++k) {
      if ((rc = dev_alloc(c, (void**)&c.halo[k], c.halo_doubles * sizeof(double)))) return bail(rc);
      if ((rc = dev_alloc(c, (void**)&c.phi_halo[k], c.plane * sizeof(double)))) return bail(rc);
      if (hipMemsetAsync(c.halo[k], 0, c.halo_doubles * sizeof(double), c.stream) != hipSuccess ||
          hipMemsetAsync(c.phi_halo[k], 0, c.plane * sizeof(double), c.stream) != hipSuccess) { c.err = "hipMemsetAsync failed"; return bail(EKPNP_ERR_HIP); }
    }
    // unknown rows of the global z system owned by this slab: interior planes only
    c.slab_row_a = (c.z0 == 0) ? 1 : 0;
    const int row_e = (c.z0 + c.nzl == p->nz) ? c.nzl - 2 : c.nzl - 1;
    c.slab_m = row_e - c.slab_row_a + 1;
    if ((rc = dev_alloc(c, (void**)&c.slab_u, (size_t)c.slab_m * nmodes * sizeof(double)))) return bail(rc);
    if ((rc = dev_alloc(c, (void**)&c.slab_w, (size_t)c.slab_m * nmodes * sizeof(double)))) return bail(rc);
    if ((rc = dev_alloc(c, (void**)&c.u1um, (size_t)nranks * 2 * nmodes * sizeof(double)))) return bail(rc);
    if ((rc = dev_alloc(c, (void**)&c.edge_local, 4 * nmodes * sizeof(double)))) return bail(rc);
    if ((rc = dev_alloc(c, (void**)&c.edge_all, (size_t)nranks * 4 * nmodes * sizeof(double)))) return bail(rc);
  }
  // batched 2-D real transforms over the owned INTERIOR planes (replaces cufftPlan3d, main.cu:112): the
  // plates carry no unknown (poisson.cu:116-119,136-139), so their rows are neither transformed nor
  // solved; the inverse transform writes phi itself (the z solve folds the 1/(NX NY) in) straight
  // into the phi array
  c.fft_z0 = (c.z0 == 0) ? 1 : 0;
  c.fft_nz = c.nzl - c.fft_z0 - ((c.z0 + c.nzl == p->nz) ? 1 : 0);
  // planes of 1024 x 1024 (cfg5) are transformed by the library's own row and column kernels (fft_plane.h: 2 + 2 kernels
  // per solve where rocFFT takes 4 + 4); everything else by rocFFT plans
  if ((rc = plane_fft_setup(c))) return bail(rc);
  if (!c.own_fft && (rc = make_fft_plans(c))) return bail(rc);
  // (experiment of round 5, profiles/r05_shared_device_experiments.log: EKPNP_ALSO_MAKE_PLANS made the unused rocFFT plans
  // beside the own passes - that alone cured the shared-device slowdown, which is how the HIP runtime's stream -> hardware
  // queue mapping was found to be the cause, not the transforms; see ekpnp_plane_transforms in include/ekpnp.h)
  if ((rc = build_cprime(c))) return bail(rc);
  if (hipStreamSynchronize(c.stream) != hipSuccess || hipGetLastError() != hipSuccess) { c.err = "device initialisation failed"; return bail(EKPNP_ERR_HIP); }
  if (arena && (rc = placement_search(c, pop_pitch, pop_nbuf))) return bail(rc);
  *out = h;
  return EKPNP_OK;
}

extern "C" int ekpnp_create(const ekpnp_params* p, ekpnp_ctx** out) { return create_impl(p, 0, 1, false, out); }
// nranks == 1 is allowed: one slab whose z neighbours are itself (the ring closes on the same
// rank), driven through the same split calls - the full multi-rank code path on a single GPU.
extern "C" int ekpnp_create_slab(const ekpnp_params* p, int rank, int nranks, ekpnp_ctx** out) { return create_impl(p, rank, nranks, true, out); }

extern "C" int ekpnp_destroy(ekpnp_ctx* ctx) {
  if (!ctx) return EKPNP_ERR_INVALID;
  Ctx& c = ctx->c;
  team_detach(c);  // an attached communicator / comm stream goes first (no-op otherwise)
  if (c.stream) (void)hipStreamSynchronize(c.stream);
  for (int b = 0; b < 2; ++b)
    for (int l = 0; l < MAXL; ++l)
      if (c.pop_alloc[b][l]) (void)hipFree(c.pop_alloc[b][l]);
  for (int l = 0; l < MAXL; ++l)
    if (c.stage[l]) (void)hipFree(c.stage[l]);
  for (int i = 0; i < EKPNP_NFIELDS; ++i)
    if (c.fld_alloc[i] && c.fld_owned[i]) (void)hipFree(c.fld_alloc[i]);
  if (c.fld_arena) (void)hipFree(c.fld_arena);
  if (c.work) (void)hipFree(c.work);
  if (c.spec) (void)hipFree(c.spec);
  if (c.cprime) (void)hipFree(c.cprime);
  drop_graph(c);
  if (c.slab_u) (void)hipFree(c.slab_u);
  if (c.slab_w) (void)hipFree(c.slab_w);
  if (c.u1um) (void)hipFree(c.u1um);
  if (c.edge_local) (void)hipFree(c.edge_local);
  if (c.edge_all) (void)hipFree(c.edge_all);
  if (c.phi_old) (void)hipFree(c.phi_old);
  if (c.diag) (void)hipFree(c.diag);
  if (c.vwall) (void)hipFree(c.vwall);
  for (int k = 0; k < 4; ++k) {
    if (c.halo[k]) (void)hipFree(c.halo[k]);
    if (c.phi_halo[k]) (void)hipFree(c.phi_halo[k]);
  }
  if (c.fft_tw) (void)hipFree(c.fft_tw);
  if (c.have_fwd) hipfftDestroy(c.plan_fwd);
  if (c.have_inv) hipfftDestroy(c.plan_inv);
  for (auto& e : c.ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  for (auto& e : c.ev_poisson) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  for (auto& e : c.ev_stage)
    for (hipEvent_t ev : e) (void)hipEventDestroy(ev);
  if (c.own_stream && c.stream) (void)hipStreamDestroy(c.stream);
  delete ctx;
  return EKPNP_OK;
}

extern "C" const char* ekpnp_last_error(const ekpnp_ctx* ctx) { return ctx ? ctx->c.err.c_str() : g_create_err.c_str(); }

extern "C" int ekpnp_set_stream(ekpnp_ctx* ctx, void* s) {
  NEEDCTX(ctx);
  HIPCHK(c, hipStreamSynchronize(c.stream));
  if (c.own_stream) { (void)hipStreamDestroy(c.stream); c.own_stream = false; }
  drop_graph(c);
  c.stream = (hipStream_t)s;
  if (c.have_fwd) FFTCHK(c, hipfftSetStream(c.plan_fwd, c.stream));
  if (c.have_inv) FFTCHK(c, hipfftSetStream(c.plan_inv, c.stream));
  return EKPNP_OK;
}

extern "C" int ekpnp_synchronize(ekpnp_ctx* ctx) {
  NEEDCTX(ctx);
  if (int rc = ensure_efield(c)) return rc;  // "complete" includes the Ex / Ey / Ez arrays a lazy solve left behind
  HIPCHK(c, hipStreamSynchronize(c.stream));
  LAUNCHCHK(c);
  return EKPNP_OK;
}

extern "C" int ekpnp_bind_field(ekpnp_ctx* ctx, int id, double* dptr) {
  NEEDCTX(ctx);
  if (id < 0 || id >= EKPNP_NFIELDS || !dptr) return fail(c, "bad field id or NULL pointer");
  if (is_phi_or_e(id)) {  // a caller-owned phi / E array is written by every solve, as the reference writes its own (eager path)
    if (int rc = ensure_efield(c)) return rc;
  }
  HIPCHK(c, hipStreamSynchronize(c.stream));
  HIPCHK(c, hipMemcpy(dptr, c.fld[id], c.nloc * sizeof(double), hipMemcpyDeviceToDevice));
  if (c.fld_owned[id]) {
    if (c.fld_alloc[id]) (void)hipFree(c.fld_alloc[id]);  // (an array inside a shared arena stays allocated until ekpnp_destroy)
    c.fld_alloc[id] = nullptr;
    c.bytes -= c.fld_bytes[id];  // the whole allocation, skew pad included
    c.fld_bytes[id] = 0;
  }
  c.fld[id] = dptr;
  c.fld_owned[id] = false;
  c.rhs_ready = false;
  drop_graph(c);
  return EKPNP_OK;
}

extern "C" int ekpnp_field_device_ptr(ekpnp_ctx* ctx, int id, double** dptr) {
  NEEDCTX(ctx);
  if (id < 0 || id >= EKPNP_NFIELDS || !dptr) return fail(c, "bad field id or NULL pointer");
  if (is_phi_or_e(id)) {
    // whoever holds this pointer may read - or write - phi / E on the device between any two calls: from here on every
    // solve writes them (the eager path of rounds 1-3), and the collide reads the arrays
    if (int rc = ensure_efield(c)) return rc;
    if (!c.e_exposed) { c.e_exposed = true; drop_graph(c); }
  } else {
    c.mom_exposed = true;  // ... and every step stores the moments ("batch_moments" is off for this context from here on)
  }
  *dptr = c.fld[id];
  return EKPNP_OK;
}

extern "C" int ekpnp_set_field(ekpnp_ctx* ctx, int id, const double* host) {
  NEEDCTX(ctx);
  if (id < 0 || id >= EKPNP_NFIELDS || !host) return fail(c, "bad field id or NULL pointer");
  if (is_phi_or_e(id)) {
    if (int rc = efield_set_from_outside(c)) return rc;
  }
  HIPCHK(c, hipStreamSynchronize(c.stream));
  HIPCHK(c, hipMemcpy(c.fld[id], host, c.nloc * sizeof(double), hipMemcpyHostToDevice));
  c.rhs_ready = false;
  return EKPNP_OK;
}

extern "C" int ekpnp_get_field(ekpnp_ctx* ctx, int id, double* host) {
  NEEDCTX(ctx);
  if (id < 0 || id >= EKPNP_NFIELDS || !host) return fail(c, "bad field id or NULL pointer");
  if (is_phi_or_e(id)) {
    if (int rc = ensure_efield(c)) return rc;
  }
  HIPCHK(c, hipStreamSynchronize(c.stream));
  HIPCHK(c, hipMemcpy(host, c.fld[id], c.nloc * sizeof(double), hipMemcpyDeviceToHost));
  return EKPNP_OK;
}

extern "C" int ekpnp_local_extent(ekpnp_ctx* ctx, int* z0, int* nzl) {
  NEEDCTX(ctx);
  if (z0) *z0 = c.z0;
  if (nzl) *nzl = c.nzl;
  return EKPNP_OK;
}

extern "C" int ekpnp_get_time(ekpnp_ctx* ctx, double* t) {
  NEEDCTX(ctx);
  if (!t) return fail(c, "NULL pointer");
  *t = c.t;
  return EKPNP_OK;
}
extern "C" int ekpnp_set_time(ekpnp_ctx* ctx, double t) {
  NEEDCTX(ctx);
  c.t = t;
  return EKPNP_OK;
}

extern "C" size_t ekpnp_device_bytes(const ekpnp_ctx* ctx) { return ctx ? ctx->c.bytes : 0; }
extern "C" int ekpnp_graph_state(const ekpnp_ctx* ctx) { return !ctx ? 0 : ctx->c.graph_failed ? -1 : ctx->c.graph2 ? 1 : 0; }

extern "C" int ekpnp_copy_bandwidth(ekpnp_ctx* ctx, size_t bytes, double* gb_per_s) {
  NEEDCTX(ctx);
  if (!gb_per_s || bytes < 16) return fail(c, "ekpnp_copy_bandwidth: bad arguments");
  bytes &= ~(size_t)15;
  void *src = nullptr, *dst = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipMalloc(&src, bytes);
  if (e == hipSuccess) e = hipMalloc(&dst, bytes);
  if (e == hipSuccess) e = hipMemsetAsync(src, 0, bytes, c.stream);
  if (e == hipSuccess) e = hipMemsetAsync(dst, 0, bytes, c.stream);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  float best = 0.0f;
  if (e == hipSuccess) {
    launch_copy16(c, src, dst, bytes);  // warm-up
    for (int rep = 0; rep < 5 && e == hipSuccess; ++rep) {
      e = hipEventRecord(e0, c.stream);
      launch_copy16(c, src, dst, bytes);
      if (e == hipSuccess) e = hipEventRecord(e1, c.stream);
      if (e == hipSuccess) e = hipEventSynchronize(e1);
      float ms = 0.0f;
      if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
      if (e == hipSuccess && (best == 0.0f || ms < best)) best = ms;
    }
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (src) (void)hipFree(src);
  if (dst) (void)hipFree(dst);
  HIPCHK(c, e);
  *gb_per_s = best > 0.0f ? 2.0 * (double)bytes / ((double)best * 1e6) : 0.0;
  return EKPNP_OK;
}

// ------------------------------------------------------------------------------------------
// Poisson

// measurement hook: HIP events around a Poisson solve (stage 1 .. stage 3 on slabs), summed by ekpnp_phase_timing_get
static int poisson_timing_mark(Ctx& c, bool begin) {
  if (!c.timing) return EKPNP_OK;
  if (begin) {
    if (c.evp_used == c.ev_poisson.size()) {
      hipEvent_t a, b;
      HIPCHK(c, hipEventCreate(&a));
      HIPCHK(c, hipEventCreate(&b));
      c.ev_poisson.emplace_back(a, b);
    }
    while (c.slab && c.ev_stage.size() < c.ev_poisson.size()) {
      std::array<hipEvent_t, 4> st{};
      for (hipEvent_t& ev : st) HIPCHK(c, hipEventCreate(&ev));
      c.ev_stage.push_back(st);
    }
    HIPCHK(c, hipEventRecord(c.ev_poisson[c.evp_used].first, c.stream));
  } else if (c.evp_used < c.ev_poisson.size()) {
    HIPCHK(c, hipEventRecord(c.ev_poisson[c.evp_used].second, c.stream));
    c.evp_used++;
  }
  return EKPNP_OK;
}

// the marks inside a slab's solve (Ctx::ev_stage): 0 end of stage 1, 1 start of stage 2, 2 end of stage 2, 3 start of stage 3
static int stage_mark(Ctx& c, int which) {
  if (!c.timing || !c.slab || c.evp_used >= c.ev_stage.size()) return EKPNP_OK;
  HIPCHK(c, hipEventRecord(c.ev_stage[c.evp_used][which], c.stream));
  return EKPNP_OK;
}

// allow_lazy: the solve may leave E in phi (time loop, ekpnp_fast_poisson).  The Poisson-Boltzmann start-up passes false:
// its kernels read and relax phi on the plates too (LBM.cu:98-104,131-146), so every sweep pins them like odd_extract does.
static int poisson_single(Ctx& c, bool allow_lazy) {
  int trc = poisson_timing_mark(c, true);
  if (trc) return trc;
  if (!c.rhs_ready) launch_poisson_rhs(c);
  c.rhs_ready = false;
  const int nb = poisson_block_count(c);
  const int zc = c.own_fft && c.poisson_zchunk > 0 && c.poisson_zchunk < c.fft_nz ? c.poisson_zchunk : 0;
  if (nb > 1 || zc) {
    // The passes in pieces that stay in the Infinity Cache from one pass to the next (poisson.hip: poisson_block; same kernels,
    // same bits).  Plane chunks: rows + columns of one run of planes back to back; column blocks: the middle passes of one
    // kx block back to back.  Both: forward by plane chunks, then z solve + y inverse by column blocks, then all rows.
    const ModeBlock all = poisson_block_whole(c);
    if (zc) {
      for (int z0 = 0; z0 < c.fft_nz; z0 += zc) {
        const int n = c.fft_nz - z0 < zc ? c.fft_nz - z0 : zc;
        if (int frc = plane_fft_forward_rows(c, z0, n)) return frc;
        plane_fft_forward_columns(c, all, z0, n);
      }
    } else if (int frc = plane_fft_forward_rows(c)) return frc;
    for (int k = 0; k < nb; ++k) {
      const ModeBlock blk = nb > 1 ? poisson_block(c, k) : all;
      if (!zc) plane_fft_forward_columns(c, blk);
      launch_tridiag(c, nb > 1 ? &blk : nullptr);
      if (nb > 1) plane_fft_inverse_columns(c, blk);
    }
    if (nb == 1) {  // (plane chunks only)
      for (int z0 = 0; z0 < c.fft_nz; z0 += zc) {
        const int n = c.fft_nz - z0 < zc ? c.fft_nz - z0 : zc;
        plane_fft_inverse_columns(c, all, z0, n);
        if (int frc = plane_fft_inverse_rows(c, z0, n)) return frc;
      }
    } else if (int frc = plane_fft_inverse_rows(c)) return frc;
  } else {
    if (int frc = plane_fft_forward(c)) return frc;
    launch_tridiag(c);
    if (int frc = plane_fft_inverse(c)) return frc;
  }
  const bool lazy = allow_lazy && lazy_efield_ok(c);
  if (!lazy) launch_phi_efield(c);
  mark_solved(c, lazy);
  LAUNCHCHK(c);
  return poisson_timing_mark(c, false);
}

// A right-hand side the collide wrote from its registers is only trusted when both concentration
// arrays are the library's own: caller-bound arrays (ekpnp_bind_field) may have been changed on the
// device between the two calls, and the reference's fast_Poisson reads charge_gpu / chargen_gpu at
// call time (poisson.cu:83,114-135).  Both producers evaluate poisson_rhs_value(): same bits.
static inline void distrust_bound_rhs(Ctx& c) {
  if (!c.fld_owned[EKPNP_C] || !c.fld_owned[EKPNP_CN]) c.rhs_ready = false;
}

extern "C" int ekpnp_fast_poisson(ekpnp_ctx* ctx) {
  NEEDCTX(ctx);
  distrust_bound_rhs(c);
  if (c.slab) return c.team ? team_ctx_fast_poisson(c) : fail(c, "ekpnp_fast_poisson on a slab context without a transport: attach one (ekpnp_slab_attach_comm), use ekpnp_group_*, or drive ekpnp_poisson_stage1/2/3 yourself");
  return poisson_single(c, true);
}

extern "C" int ekpnp_invalidate_rhs(ekpnp_ctx* ctx) {
  NEEDCTX(ctx);
  c.rhs_ready = false;
  return EKPNP_OK;
}

// ------------------------------------------------------------------------------------------
// initial state

extern "C" int ekpnp_init_fields(ekpnp_ctx* ctx) {  // gpu_initialization, LBM.cu:111-128
  NEEDCTX(ctx);
  c.rhs_ready = false;
  c.e_stale = false;      // every array is written here: phi = voltage, E = 0 (LBM.cu:111-128)
  c.e_phi_valid = false;
  launch_init_fields(c);
  LAUNCHCHK(c);
  return EKPNP_OK;
}

extern "C" int ekpnp_pbe_begin(ekpnp_ctx* ctx) {  // LBM.cu:79-86: phi_old <- phi
  NEEDCTX(ctx);
  if (int rc = ensure_efield(c)) return rc;  // phi_old takes the plates too
  if (!c.phi_old) HIPCHK(c, hipMalloc((void**)&c.phi_old, c.nloc * sizeof(double)));
  HIPCHK(c, hipMemcpyAsync(c.phi_old, c.fld[EKPNP_PHI], c.nloc * sizeof(double), hipMemcpyDeviceToDevice, c.stream));
  return EKPNP_OK;
}

extern "C" int ekpnp_pbe_concentrations(ekpnp_ctx* ctx) {  // gpu_PBE, LBM.cu:139-146
  NEEDCTX(ctx);
  if (int rc = ensure_efield(c)) return rc;  // reads phi on every plane, the plates included
  c.rhs_ready = false;
  launch_pbe(c);
  LAUNCHCHK(c);
  return EKPNP_OK;
}

extern "C" int ekpnp_pbe_relax(ekpnp_ctx* ctx) {  // gpu_PBE_phi + phi_old update, LBM.cu:98-104
  NEEDCTX(ctx);
  if (!c.phi_old) return fail(c, "ekpnp_pbe_relax without ekpnp_pbe_begin");
  if (int rc = efield_set_from_outside(c)) return rc;  // phi is relaxed on every plane: E no longer derives from it
  launch_pbe_relax(c, c.phi_old, c.p.PB_omega);
  LAUNCHCHK(c);
  return EKPNP_OK;
}

extern "C" int ekpnp_pbe_end(ekpnp_ctx* ctx) {  // LBM.cu:107-108
  NEEDCTX(ctx);
  HIPCHK(c, hipStreamSynchronize(c.stream));
  if (c.phi_old) { (void)hipFree(c.phi_old); c.phi_old = nullptr; }
  c.e_phi_valid = false;  // phi was relaxed after the last solve formed E (LBM.cu:96-104): E is what the arrays say
  return EKPNP_OK;
}

extern "C" int ekpnp_initialization(ekpnp_ctx* ctx) {
  NEEDCTX(ctx);
  if (c.slab) return c.team ? team_ctx_initialization(c) : fail(c, "ekpnp_initialization on a slab context without a transport: attach one, use ekpnp_group_*, or drive ekpnp_init_fields / ekpnp_pbe_* and the Poisson stages yourself");
  int rc = ekpnp_init_fields(ctx);
  if (rc == EKPNP_OK) rc = ekpnp_pbe_begin(ctx);
  for (int i = 0; rc == EKPNP_OK && i < c.p.pb_iterations; ++i) {  // LBM.cu:89-106
    c.rhs_ready = false;
    launch_pbe(c);
    rc = poisson_single(c, false);
    launch_pbe_relax(c, c.phi_old, c.p.PB_omega);
    c.e_phi_valid = false;  // the relaxed phi is not the phi E was taken from (LBM.cu:96-104: efield runs inside fast_Poisson, before gpu_PBE_phi)
  }
  int rc2 = ekpnp_pbe_end(ctx);
  if (rc == EKPNP_OK) rc = rc2;
  if (rc == EKPNP_OK) LAUNCHCHK(c);
  return rc;
}

// the same Picard sweeps as initialization() (LBM.cu:89-106), but
//  - stopped by a convergence test instead of the fixed 501 sweeps: the residual is
//    max |phi_solved - phi_old| / max(|voltage|, |voltage2|), reduced on the device;
//  - with a damping that cannot diverge: the lowest z mode of the linearised iteration is
//    amplified by A = (Lz/(pi lambda_D))^2 per sweep, so PB_omega = 0.05 diverges once A > 39
//    (NZ > ~180 at the default spacing); omega = min(PB_omega, 1.6/(1 + A)) is used.
extern "C" int ekpnp_initialization_converged(ekpnp_ctx* ctx, double rel_tol, int max_sweeps, int* sweeps, double* residual) {
  NEEDCTX(ctx);
  if (c.slab) return c.team ? team_ctx_initialization_converged(c, rel_tol, max_sweeps, sweeps, residual) : fail(c, "ekpnp_initialization_converged on a slab context without a transport: attach one or use ekpnp_group_*");
  if (max_sweeps < 0) return fail(c, "max_sweeps < 0");
  double omega = c.p.PB_omega;
  if (c.p.chargeinf > 0.0) {
    const double lam2 = c.p.eps * c.p.kB * c.p.roomT / c.p.electron / (2.0 * c.p.chargeinf * c.p.convertCtoCharge);
    const double A = c.p.Lz * c.p.Lz / (M_PI * M_PI * lam2);
    if (1.6 / (1.0 + A) < omega) omega = 1.6 / (1.0 + A);
  }
  double scale = std::fabs(c.p.voltage) > std::fabs(c.p.voltage2) ? std::fabs(c.p.voltage) : std::fabs(c.p.voltage2);
  if (scale == 0.0) scale = 1.0;
  if (!c.diag) HIPCHK(c, hipMalloc((void**)&c.diag, DIAG_SCRATCH * sizeof(double)));
  int rc = ekpnp_init_fields(ctx);
  if (rc == EKPNP_OK) rc = ekpnp_pbe_begin(ctx);
  int done = 0;
  double res = 0.0;
  const int check_every = 10;
  while (rc == EKPNP_OK && done < max_sweeps) {
    c.rhs_ready = false;
    launch_pbe(c);
    rc = poisson_single(c, false);
    if (rc) break;
    ++done;
    const bool check = (done % check_every == 0) || done == max_sweeps;
    if (check) launch_max_abs_diff(c, c.fld[EKPNP_PHI], c.phi_old, c.diag);
    launch_pbe_relax(c, c.phi_old, omega);
    c.e_phi_valid = false;
    if (check) {
      double r = 0.0;
      hipError_t e = hipMemcpyAsync(&r, c.diag + 1024, sizeof(double), hipMemcpyDeviceToHost, c.stream);
      if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
      if (e != hipSuccess) { c.err = "residual read-back failed"; rc = EKPNP_ERR_HIP; break; }
      res = r / scale;
      if (!(res == res)) { c.err = "Poisson-Boltzmann iteration produced NaN"; rc = EKPNP_ERR_INVALID; break; }
      if (res <= rel_tol) break;
    }
  }
  int rc2 = ekpnp_pbe_end(ctx);
  if (rc == EKPNP_OK) rc = rc2;
  if (sweeps) *sweeps = done;
  if (residual) *residual = res;
  return rc;
}

extern "C" int ekpnp_init_equilibrium(ekpnp_ctx* ctx) {
  NEEDCTX(ctx);
  if (int rc = ensure_efield(c)) return rc;  // the equilibria drift with u + K E (LBM.cu:207-462): the E arrays are read
  c.halo_recv_valid = false;  // the next step does not pull, and the halos of the old populations mean nothing to the new ones
  launch_init_equilibrium(c);
  c.streamed_state = true;
  LAUNCHCHK(c);
  return EKPNP_OK;
}

// ------------------------------------------------------------------------------------------
// time step

static int timing_begin(Ctx& c, hipEvent_t** stop) {
  *stop = nullptr;
  if (!c.timing) return EKPNP_OK;
  if (c.ev_used == c.ev.size()) {
    hipEvent_t a, b;
    HIPCHK(c, hipEventCreate(&a));
    HIPCHK(c, hipEventCreate(&b));
    c.ev.emplace_back(a, b);
  }
  HIPCHK(c, hipEventRecord(c.ev[c.ev_used].first, c.stream));
  *stop = &c.ev[c.ev_used].second;
  c.ev_used++;
  return EKPNP_OK;
}

// first/last owned plane handled by the bulk kernel (wall planes go to the wall kernel)
static inline int bulk_begin(const Ctx& c) { return c.z0 == 0 ? 1 : 0; }
static inline int bulk_end(const Ctx& c) { return (c.z0 + c.nzl == c.p.nz) ? c.nzl - 1 : c.nzl; }

static int collide_range(Ctx& c, int zb, int ze, bool timed) {
  hipEvent_t* stop = nullptr;
  if (timed) {
    int rc = timing_begin(c, &stop);
    if (rc) return rc;
  }
  // the two-buffer sweep in launches of c.ab_zchunk planes (0: one launch); ekpnp_tune / EKPNP_BULK_ZCHUNK
  const int zchunk = c.ab_zchunk;
  if (zchunk > 0)
    for (int z = zb; z < ze; z += zchunk) launch_collide_bulk(c, z, z + zchunk < ze ? z + zchunk : ze);
  else
    launch_collide_bulk(c, zb, ze);
  if (stop) {
    HIPCHK(c, hipEventRecord(*stop, c.stream));
    c.timed_nodes = (long long)(ze - zb) * (long long)c.plane;
  }
  return EKPNP_OK;
}

// bulk launches of planes [zb, ze) in the z order the in-place shift requires (one launch in A/B mode).
// lead > 0: the sweep starts with a launch of only `lead` planes (see ekpnp_collide_interior_planes); any launch of
// at most zchunk planes keeps the in-place invariant (shift >= planes per launch + 1).
static void ordered_bulk(Ctx& c, int zb, int ze, int lead = 0) {
  if (!c.inplace) { launch_collide_bulk(c, zb, ze); return; }
  if (lead > c.zchunk) lead = c.zchunk;
  if (c.cur == 0) {
    if (lead > 0 && zb + lead < ze) { launch_collide_bulk(c, zb, zb + lead); zb += lead; }
    for (int z = zb; z < ze; z += c.zchunk) launch_collide_bulk(c, z, z + c.zchunk < ze ? z + c.zchunk : ze);
  } else {
    if (lead > 0 && ze - lead > zb) { launch_collide_bulk(c, ze - lead, ze); ze -= lead; }
    for (int z = ze; z > zb; z -= c.zchunk) launch_collide_bulk(c, z - c.zchunk > zb ? z - c.zchunk : zb, z);
  }
}

// Where the population arena lies in HBM decides how fast the sweep runs on lattices that fill only part of the device:
// the SAME context re-created in one process sweeps a 512x512x128 lattice in 9.65 ... 11.0 ms (identical virtual addresses,
// physically contiguous or not: profiles/r03_placement_spread_thin_lattices.log), while a plain copy varies by 2 % and cfg3,
// whose arena is most of the device, does not vary at all.  Counters (round 4, profiles/r04_thin_slab_512x512x128_pmc_sq_lds_tcc_hbm.json):
// the slow placements move the SAME bytes (0.47697 GB per plane, cfg3: 0.47698) at the SAME L2 hit rate (0.3743 / 0.3742) -
// it is DRAM service time, i.e. which of the ~300 sequential streams of the sweep meet in the same banks, that differs.
// So when there is room, a context tries up to EKPNP_PLACEMENT_TRIES (default 5; 1 = off) arenas, times the real sweep on
// each - BOTH directions - and keeps the fastest (~0.1 s per try, once per context).  The arithmetic never sees the difference.
//
// Memory: a candidate is allocated while the best so far is still held (otherwise the allocator hands the same pages back),
// so creation transiently needs up to TWO arenas; rejected candidates are kept as well while they fit into 85 % of what is
// free (they keep the next candidate from landing where a slow one was) and are dropped first when they do not.  A lattice
// whose second arena does not fit (cfg3, cfg5's and cfg4@2's slabs) is not searched at all.  Ranks that share one device
// (rehearsals) should set EKPNP_PLACEMENT_TRIES=1: two searches at once can take each other's memory (bench.py does).
static int placement_search(Ctx& c, size_t pitch, int nbuf) {
  static const int tries_env = std::getenv("EKPNP_PLACEMENT_TRIES") ? std::atoi(std::getenv("EKPNP_PLACEMENT_TRIES")) : 5;
  const int zb = bulk_begin(c), ze = bulk_end(c);
  const size_t total = pitch * (size_t)nbuf;
  c.placement_tries = 0;
  c.placement_chosen = 0;
  if (tries_env <= 1 || ze - zb < 8 || c.nloc < (size_t)4 * 1024 * 1024) return EKPNP_OK;  // launch-bound lattices: nothing to gain
  const int tries = tries_env > 8 ? 8 : tries_env;
  auto fits = [&]() {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return false; }
    return (double)total <= 0.85 * (double)free_b;
  };
  if (!fits()) return EKPNP_OK;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  HIPCHK(c, hipEventCreate(&e0));
  {
    const hipError_t ee = hipEventCreate(&e1);
    if (ee != hipSuccess) { (void)hipEventDestroy(e0); HIPCHK(c, ee); }
  }
  auto point_at = [&](void* base) { carve_arena(c, base, pitch); };
  const bool streamed0 = c.streamed_state;
  c.streamed_state = false;  // time the kernel the steps run: pull + collide,
  c.e_phi_valid = lazy_efield_ok(c);  // E from phi (all zero here) where the steps will take it from phi
  void* cand[8] = {c.pop_alloc[0][0]};  // held allocations by try index (null: freed again or never made)
  double best_ms[8] = {};
  int n = 0, best = 0;
  hipError_t e = hipSuccess;
  for (; n < tries && e == hipSuccess; ++n) {
    if (n > 0) {
      if (!fits()) {  // drop the rejected candidates, keep the best so far
        for (int k = 0; k < n; ++k)
          if (cand[k] && k != best) { (void)hipFree(cand[k]); cand[k] = nullptr; }
        if (!fits()) break;
      }
      if (hipMalloc(&cand[n], total) != hipSuccess) { (void)hipGetLastError(); cand[n] = nullptr; break; }
    }
    point_at(cand[n]);
    e = hipMemsetAsync(cand[n], 0, total, c.stream);
    // a step reads the buffers the step before wrote: BOTH directions (A -> B, B -> A; in place: down, up) are timed, two
    // sweeps each after a warm-up pair - on some placements one direction is 8 % slower than the other
    float dir_ms[2] = {0.f, 0.f};
    c.cur = 0;
    for (int rep = 0; rep < 6 && e == hipSuccess; ++rep) {
      e = hipEventRecord(e0, c.stream);
      ordered_bulk(c, zb, ze);
      if (e == hipSuccess) e = hipEventRecord(e1, c.stream);
      if (e == hipSuccess) e = hipEventSynchronize(e1);
      float ms = 0.f;
      if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
      if (rep >= 2 && e == hipSuccess && (dir_ms[c.cur] == 0.f || ms < dir_ms[c.cur])) dir_ms[c.cur] = ms;
      c.cur ^= 1;
    }
    best_ms[n] = 0.5 * ((double)dir_ms[0] + (double)dir_ms[1]);
    if (e == hipSuccess && best_ms[n] > 0.0 && (best_ms[best] <= 0.0 || best_ms[n] < best_ms[best])) best = n;
    if (std::getenv("EKPNP_DEBUG_ARENA")) std::fprintf(stderr, "ekpnp: arena %d: %.3f / %.3f ms per sweep in the two directions\n", n, dir_ms[0], dir_ms[1]);
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  for (int k = 0; k < 8; ++k)
    if (cand[k] && k != best) (void)hipFree(cand[k]);
  c.pop_alloc[0][0] = cand[best];
  point_at(cand[best]);
  c.streamed_state = streamed0;
  c.e_phi_valid = false;
  c.cur = 0;
  c.rhs_ready = false;
  c.placement_tries = n;
  c.placement_chosen = best;
  for (int k = 0; k < 8; ++k) c.placement_ms[k] = k < n ? best_ms[k] : 0.0;
  // the probe sweeps wrote moments of an all-zero lattice (NaN) into the kept arena, the field arrays and the right-hand
  // side: all back to zero, which is what a context without a search starts from
  if (e == hipSuccess) e = hipMemsetAsync(cand[best], 0, total, c.stream);
  for (int i = 0; i < EKPNP_NFIELDS && e == hipSuccess; ++i) e = hipMemsetAsync(c.fld[i], 0, c.nloc * sizeof(double), c.stream);
  if (e == hipSuccess) e = hipMemsetAsync(c.work, 0, c.nloc * sizeof(double), c.stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
  if (take_launch_error(c) != hipSuccess) return EKPNP_ERR_HIP;
  HIPCHK(c, e);
  if (std::getenv("EKPNP_DEBUG_ARENA")) {
    std::fprintf(stderr, "ekpnp: placement search, %d arenas:", n);
    for (int k = 0; k < n; ++k) std::fprintf(stderr, " %.3f%s", best_ms[k], k == best ? "*" : "");
    std::fprintf(stderr, " ms per sweep\n");
  }
  return EKPNP_OK;
}

extern "C" int ekpnp_placement_report(ekpnp_ctx* ctx, int* n_tried, int* chosen, double* sweep_ms, int capacity) {
  NEEDCTX(ctx);
  if (n_tried) *n_tried = c.placement_tries;
  if (chosen) *chosen = c.placement_chosen;
  for (int k = 0; sweep_ms && k < capacity && k < 8; ++k) sweep_ms[k] = c.placement_ms[k];
  return EKPNP_OK;
}

static void finish_collide(Ctx& c) {
  c.cur ^= 1;
  c.streamed_state = false;
  c.rhs_ready = c.p.n_lattices > 1;  // every owned plane has written its Poisson rhs into work[]
}

extern "C" int ekpnp_stream_collide_save(ekpnp_ctx* ctx, double t) {
  NEEDCTX(ctx);
  (void)t;  // the reference passes t but never uses it (LBM.cu:483-1846)
  if (c.slab) return c.team ? team_ctx_stream_collide_save(c) : fail(c, "slab context without a transport: attach one, use ekpnp_group_*, or drive ekpnp_collide_boundary_planes/interior_planes + the halo exchange yourself");
  // (running the two wall planes on a second stream beside the bulk kernel was measured: no gain on
  // large lattices, the bulk kernel already saturates HBM and merely stretches -
  // profiles/r01_bench_after_tuning.log; on the launch-bound 50x8x51 lattice the fork / join events
  // cost more than the overlap saves: 0.0505 vs 0.0388 ms/step, profiles/r02_small_grid_timing.log)
  const int zb = bulk_begin(c), ze = bulk_end(c);
  // launch-bound lattices: plates and bulk in ONE launch (k_collide_all): one kernel of the dependent
  // chain less (the reference's 50x8x51: 0.0388 -> see profiles/r02_small_grid_timing.log); large lattices
  // keep the lean bulk kernel and a separate wall kernel
  if (!c.inplace && c.merged_walls && !c.timing && c.nloc <= (size_t)4 * 1024 * 1024 && c.nzl > 2) {
    launch_collide_all(c);
  } else if (!c.inplace) {
    int rc = collide_range(c, zb, ze, true);
    if (rc) return rc;
    launch_collide_walls(c, c.stream, true, true);
  } else {
    // ordered sweep: the wall plane on the side the lattice moves towards first, bulk launches of
    // zchunk planes towards the other side, the far wall plane last
    hipEvent_t* stop = nullptr;
    const bool up = c.cur == 0;  // ascending z, new state written `shift` planes below
    launch_collide_walls(c, c.stream, up, !up);
    int rc = timing_begin(c, &stop);
    if (rc) return rc;
    ordered_bulk(c, zb, ze);
    if (stop) {
      HIPCHK(c, hipEventRecord(*stop, c.stream));
      c.timed_nodes = (long long)(ze - zb) * (long long)c.plane;
    }
    launch_collide_walls(c, c.stream, !up, up);
  }
  finish_collide(c);
  if (c.inplace) launch_ghost_wrap(c);  // z-periodic ghost loop of gpu_stream, LBM.cu:1972,1975 (by index otherwise)
  LAUNCHCHK(c);
  return EKPNP_OK;
}

static int one_step(ekpnp_ctx* ctx) {  // main.cu:189-200
  Ctx& c = ctx->c;
  int rc = ekpnp_stream_collide_save(ctx, c.t);
  if (rc) return rc;
  rc = poisson_single(c, true);
  if (rc) return rc;
  c.t = c.t + c.p.dt;
  return EKPNP_OK;
}

// Small lattices are launch-bound (the reference's own 50x8x51 problem: ~10 launches of a few
// microseconds each per step).  Two consecutive steps (buffers A->B, then B->A) are captured once
// into a hipGraph and replayed; the arithmetic and its order are exactly the eager ones.
static bool graph_wanted(const Ctx& c, int nsteps) {
  static const bool off = std::getenv("EKPNP_NO_GRAPH") != nullptr;
  return !off && !c.graph_failed && !c.timing && !c.streamed_state && nsteps >= 4 && !c.slab &&
         c.nloc <= (size_t)4 * 1024 * 1024;  // beyond ~4 M nodes a step is >1 ms of kernels: nothing to gain
}

static int capture_two_steps(ekpnp_ctx* ctx) {
  Ctx& c = ctx->c;
  drop_graph(c);
  const int cur0 = c.cur;
  const double t0 = c.t;
  const bool stale0 = c.e_stale, phiv0 = c.e_phi_valid;
  if (hipStreamBeginCapture(c.stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); c.graph_failed = true; return EKPNP_OK; }
  int rc = one_step(ctx);
  if (rc == EKPNP_OK) rc = one_step(ctx);
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(c.stream, &g);
  // nothing has executed: restore the host-side state the two calls advanced
  c.cur = cur0;
  c.t = t0;
  c.rhs_ready = false;
  c.e_stale = stale0;
  c.e_phi_valid = phiv0;
  if (rc != EKPNP_OK || e != hipSuccess || !g) {
    (void)hipGetLastError();
    if (g) (void)hipGraphDestroy(g);
    c.graph_failed = true;
    c.err.clear();
    return EKPNP_OK;
  }
  e = hipGraphInstantiate(&c.graph2, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) { (void)hipGetLastError(); c.graph2 = nullptr; c.graph_failed = true; return EKPNP_OK; }
  c.graph_cur = cur0;
  return EKPNP_OK;
}

extern "C" int ekpnp_step(ekpnp_ctx* ctx, int nsteps) {
  NEEDCTX(ctx);
  if (nsteps < 0) return fail(c, "nsteps < 0");
  if (c.slab) return c.team ? team_ctx_step(c, nsteps) : fail(c, "slab context without a transport: attach one, use ekpnp_group_*, or drive the split calls yourself");
  int i = 0;
  // the first step after init_equilibrium does not pull; one whose E was set from outside reads the E arrays: neither is
  // the step the graph below captures
  if ((c.streamed_state || (lazy_efield_ok(c) && !c.e_phi_valid)) && nsteps > 0) {
    int rc = one_step(ctx);
    if (rc) return rc;
    ++i;
  }
  if (graph_wanted(c, nsteps - i)) {
    if (!c.graph2 || c.graph_cur != c.cur) {
      int rc = capture_two_steps(ctx);
      if (rc) return rc;
    }
    while (c.graph2 && nsteps - i >= 2) {
      HIPCHK(c, hipGraphLaunch(c.graph2, c.stream));
      mark_solved(c, lazy_efield_ok(c));
      c.t = c.t + c.p.dt;
      c.t = c.t + c.p.dt;
      i += 2;
    }
  }
  const bool batch = batch_moments_ok(c);
  for (; i < nsteps; ++i) {
    c.skip_moments = batch && i < nsteps - 1;  // only the call's LAST step can be looked at
    int rc = one_step(ctx);
    c.skip_moments = false;
    if (rc) return rc;
  }
  return EKPNP_OK;
}

// ------------------------------------------------------------------------------------------
// measurement hooks

extern "C" int ekpnp_kernel_timing_enable(ekpnp_ctx* ctx, int enable) {
  NEEDCTX(ctx);
  HIPCHK(c, hipStreamSynchronize(c.stream));
  c.timing = enable != 0;
  c.ev_used = 0;
  c.evp_used = 0;
  team_timing_reset(c);
  return EKPNP_OK;
}

extern "C" int ekpnp_kernel_timing_get(ekpnp_ctx* ctx, int* n_launches, double* total_ms, int64_t* nodes_per_launch) {
  NEEDCTX(ctx);
  HIPCHK(c, hipStreamSynchronize(c.stream));
  double tot = 0.0;
  for (size_t i = 0; i < c.ev_used; ++i) {
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c.ev[i].first, c.ev[i].second));
    tot += ms;
  }
  if (n_launches) *n_launches = (int)c.ev_used;
  if (total_ms) *total_ms = tot;
  if (nodes_per_launch) *nodes_per_launch = c.timed_nodes;
  c.ev_used = 0;
  return EKPNP_OK;
}

extern "C" int ekpnp_phase_timing_get(ekpnp_ctx* ctx, int* n_solves, double* poisson_ms) {
  NEEDCTX(ctx);
  HIPCHK(c, hipStreamSynchronize(c.stream));
  double tot = 0.0;
  for (size_t i = 0; i < c.evp_used; ++i) {
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c.ev_poisson[i].first, c.ev_poisson[i].second));
    tot += ms;
  }
  if (n_solves) *n_solves = (int)c.evp_used;
  if (poisson_ms) *poisson_ms = tot;
  c.evp_used = 0;
  return EKPNP_OK;
}

extern "C" int ekpnp_poisson_stage_timing_get(ekpnp_ctx* ctx, int* n_solves, double* stage_ms) {
  NEEDCTX(ctx);
  if (!stage_ms) return fail(c, "stage_ms is NULL (5 doubles: stage 1, EDGE exchange, stage 2, PHI exchange, stage 3)");
  for (int k = 0; k < 5; ++k) stage_ms[k] = 0.0;
  if (n_solves) *n_solves = 0;
  if (!c.slab) return EKPNP_OK;
  HIPCHK(c, hipStreamSynchronize(c.stream));
  const size_t n = c.evp_used < c.ev_stage.size() ? c.evp_used : c.ev_stage.size();
  for (size_t i = 0; i < n; ++i) {
    const hipEvent_t t[6] = {c.ev_poisson[i].first, c.ev_stage[i][0], c.ev_stage[i][1], c.ev_stage[i][2], c.ev_stage[i][3], c.ev_poisson[i].second};
    for (int k = 0; k < 5; ++k) {
      float ms = 0.f;
      HIPCHK(c, hipEventElapsedTime(&ms, t[k], t[k + 1]));
      stage_ms[k] += ms;
    }
  }
  if (n_solves) *n_solves = (int)n;
  return EKPNP_OK;
}

// ------------------------------------------------------------------------------------------
// z-slab interface; the transport between the calls is the caller's

extern "C" int ekpnp_halo_buffer(ekpnp_ctx* ctx, int which, double** dptr, size_t* n) {
  NEEDCTX(ctx);
  if (which < 0 || which > 3 || !dptr || !n) return fail(c, "bad halo buffer id");
  if (!c.slab) return fail(c, "no halo buffers on a single-slab context");
  *dptr = c.halo[which];
  *n = c.halo_doubles;
  return EKPNP_OK;
}

extern "C" int ekpnp_phi_halo_buffer(ekpnp_ctx* ctx, int which, double** dptr, size_t* n) {
  NEEDCTX(ctx);
  if (which < 0 || which > 3 || !dptr || !n) return fail(c, "bad halo buffer id");
  if (!c.slab) return fail(c, "no halo buffers on a single-slab context");
  *dptr = c.phi_halo[which];
  *n = c.plane;
  return EKPNP_OK;
}

extern "C" int ekpnp_halo_pack(ekpnp_ctx* ctx) {
  NEEDCTX(ctx);
  if (!c.slab) return fail(c, "no halo buffers on a single-slab context");
  if (c.inplace && c.collide_phase != 1) return fail(c, "in-place slab: ekpnp_halo_pack belongs between the boundary and the interior call");
  if (c.halo_sent) {  // the boundary-plane launches stored their outgoing directions straight into the send buffers
    c.halo_sent = false;
    return EKPNP_OK;
  }
  if (c.inplace) {
    launch_halo_pack_stage(c);
  } else {
    // between the boundary and the interior call the fresh planes are in the NEXT buffer
    launch_halo_pack(c, c.collide_phase == 1 ? (c.cur ^ 1) : c.cur);
  }
  LAUNCHCHK(c);
  return EKPNP_OK;
}

extern "C" int ekpnp_halo_unpack(ekpnp_ctx* ctx) {
  NEEDCTX(ctx);
  if (!c.slab) return fail(c, "no halo buffers on a single-slab context");
  if (c.collide_phase != 0) return fail(c, "ekpnp_halo_unpack before ekpnp_collide_interior_planes");
  if (c.halo_direct) {
    // nothing is copied: the edge planes of the NEXT step pull straight out of the receive buffers (which the next exchange
    // only overwrites after those launches); the ghost planes are filled on demand (ekpnp_save_checkpoint)
    c.halo_recv_valid = true;
    return EKPNP_OK;
  }
  launch_halo_unpack(c);
  LAUNCHCHK(c);
  return EKPNP_OK;
}

namespace ekpnp {
// the ghost planes of a slab, for whoever reads them instead of the receive buffers (the per-rank checkpoint)
int ensure_ghost_planes(Ctx& c) {
  if (!c.slab || !c.halo_recv_valid) return EKPNP_OK;
  launch_halo_unpack(c);
  LAUNCHCHK(c);
  return EKPNP_OK;
}
}  // namespace ekpnp

extern "C" int ekpnp_collide_boundary_planes(ekpnp_ctx* ctx) {
  NEEDCTX(ctx);
  if (!c.slab) return fail(c, "single-slab context: use ekpnp_stream_collide_save");
  if (c.collide_phase != 0) return fail(c, "ekpnp_collide_boundary_planes called twice");
  KArgs lo = c.kargs(), hi = lo;
  if (c.halo_direct) {
    // no pack / unpack copies: the edge launches write the send buffers and read the receive buffers themselves
    lo.halo_out_dn = c.halo[0];
    hi.halo_out_up = c.halo[1];
    if (c.halo_recv_valid) { lo.halo_in_lo = c.halo[2]; hi.halo_in_hi = c.halo[3]; }
  }
  if (c.inplace) {
    // redirect the stores of plane zg = 1 / zg = nzl into the staging planes 0 / 1
    for (int l = 0; l < c.p.n_lattices; ++l) {
      lo.B[l] = c.stage[l] - (ptrdiff_t)c.pplane;                              // plane zg = 1   -> staging plane 0
      hi.B[l] = c.stage[l] + (ptrdiff_t)c.pplane - (ptrdiff_t)c.nzl * (ptrdiff_t)c.pplane;  // plane zg = nzl -> staging plane 1
    }
  }
  if (c.halo_direct && c.merged_faces) {  // (A/B knob: EKPNP_MERGED_FACES / ekpnp_tune "merged_faces")
    launch_collide_faces(c, lo, hi);  // both faces, plate or not, in one launch
  } else {
    launch_collide_walls(c, lo, c.stream, true, false);
    launch_collide_walls(c, hi, c.stream, false, true);
    if (c.z0 != 0) { if (c.halo_direct) launch_collide_bulk_edge(c, lo, 0); else launch_collide_bulk(c, lo, 0, 1); }
    if (c.z0 + c.nzl != c.p.nz) { if (c.halo_direct) launch_collide_bulk_edge(c, hi, c.nzl - 1); else launch_collide_bulk(c, hi, c.nzl - 1, c.nzl); }
  }
  c.halo_sent = c.halo_direct;
  c.halo_recv_valid = false;  // consumed; the exchange that follows refills the buffers, ekpnp_halo_unpack says when
  c.collide_phase = 1;
  LAUNCHCHK(c);
  return EKPNP_OK;
}

extern "C" int ekpnp_collide_interior_planes(ekpnp_ctx* ctx) {
  NEEDCTX(ctx);
  if (c.collide_phase != 1) return fail(c, "ekpnp_collide_interior_planes without ekpnp_collide_boundary_planes");
  hipEvent_t* stop = nullptr;
  int rc = timing_begin(c, &stop);
  if (rc) return rc;
  // The halo transfer that was started before this call (comm stream) and the sweep below become
  // ready at the same moment, and a single launch of ~10^6 workgroups keeps every wave slot of the
  // chip refilled with its own workgroups until it drains: the workgroups of an RCCL kernel, which
  // need more registers than one retiring collide wave frees, were only placed when the sweep
  // ended (rocprofv3: the exchange kernel ended 0.1 ms AFTER a 39 ms sweep,
  // profiles/r02_slab_overlap_before.json).  A short lead-in launch drains after ~0.1 ms and lets
  // them in; the rest of the sweep then runs beside the transfer.
  const int lead_env = c.lead_planes;  // EKPNP_SLAB_LEAD_PLANES / ekpnp_tune "lead_planes", default 2
  // (in-place slabs sweep in launches of zchunk planes anyway, but the first of them is up to 64 planes = 5 ms
  // long on cfg3's planes: they get the same short lead-in, at the end of the lattice their ordered sweep starts from)
  const int lead = (lead_env > 0 && c.nzl - 2 > 4 * lead_env) ? lead_env : 0;
  if (lead && !c.inplace) {
    launch_collide_bulk(c, 1, 1 + lead);
    launch_collide_bulk(c, 1 + lead, c.nzl - 1);
  } else {
    ordered_bulk(c, 1, c.nzl - 1, lead);
  }
  if (stop) {
    HIPCHK(c, hipEventRecord(*stop, c.stream));
    c.timed_nodes = (long long)(c.nzl - 2) * (long long)c.plane;
  }
  finish_collide(c);
  if (c.inplace) launch_unstage(c);  // the two staged planes take their place in the shifted lattice
  c.collide_phase = 0;
  LAUNCHCHK(c);
  return EKPNP_OK;
}

extern "C" int ekpnp_advance_time(ekpnp_ctx* ctx) {
  NEEDCTX(ctx);
  c.t = c.t + c.p.dt;
  return EKPNP_OK;
}

namespace ekpnp {
int poisson_stage1_begin(Ctx& c) {
  distrust_bound_rhs(c);
  {
    int trc = poisson_timing_mark(c, true);
    if (trc) return trc;
  }
  if (!c.rhs_ready) launch_poisson_rhs(c);
  c.rhs_ready = false;
  if (int frc = plane_fft_forward_rows(c)) return frc;
  LAUNCHCHK(c);
  return EKPNP_OK;
}
int poisson_stage1_block(Ctx& c, int k) {
  plane_fft_forward_columns(c, mode_block(c, k));
  launch_slab_thomas_local(c, k);
  LAUNCHCHK(c);
  return EKPNP_OK;
}
int poisson_stage1_end(Ctx& c) { return stage_mark(c, 0); }
int poisson_stage2_block(Ctx& c, int k) {
  if (k == 0) {
    if (int mrc = stage_mark(c, 1)) return mrc;
  }
  launch_slab_reduce_correct(c, k);
  plane_fft_inverse_columns(c, mode_block(c, k));
  LAUNCHCHK(c);
  return EKPNP_OK;
}
int poisson_stage2_end(Ctx& c) {
  if (int frc = plane_fft_inverse_rows(c)) return frc;
  launch_phi_halo_pack(c);
  LAUNCHCHK(c);
  return stage_mark(c, 2);
}
}  // namespace ekpnp

// The exported stages are for a transport of the CALLER's between them, which gathers the edge buffer in one piece: they
// refuse a context that was tuned to several mode blocks (only the library's own transport moves those).
extern "C" int ekpnp_poisson_stage1(ekpnp_ctx* ctx) {
  NEEDCTX(ctx);
  if (!c.slab) return fail(c, "single-slab context: use ekpnp_fast_poisson");
  if (edge_chunk_count(c) != 1) return fail(c, "ekpnp_poisson_stage1: edge_chunks > 1 lays the edge buffers out in mode blocks, which only the library's own transport exchanges");
  if (int rc = poisson_stage1_begin(c)) return rc;
  if (int rc = poisson_stage1_block(c, 0)) return rc;
  return poisson_stage1_end(c);
}

extern "C" int ekpnp_poisson_edge_buffer(ekpnp_ctx* ctx, int gathered, double** dptr, size_t* n) {
  NEEDCTX(ctx);
  if (!c.slab || !dptr || !n) return fail(c, "no edge buffers on a single-slab context");
  const size_t per_rank = 4 * (size_t)c.p.ny * c.nxh;
  *dptr = gathered ? c.edge_all : c.edge_local;
  *n = gathered ? per_rank * c.nranks : per_rank;
  return EKPNP_OK;
}

extern "C" int ekpnp_poisson_stage2(ekpnp_ctx* ctx) {
  NEEDCTX(ctx);
  if (!c.slab) return fail(c, "single-slab context: use ekpnp_fast_poisson");
  if (edge_chunk_count(c) != 1) return fail(c, "ekpnp_poisson_stage2: edge_chunks > 1 needs the library's own transport");
  if (int rc = stage_mark(c, 1)) return rc;
  launch_slab_reduce_correct(c, 0);
  if (int frc = plane_fft_inverse(c)) return frc;
  LAUNCHCHK(c);
  return EKPNP_OK;
}

extern "C" int ekpnp_phi_halo_pack(ekpnp_ctx* ctx) {
  NEEDCTX(ctx);
  if (!c.slab) return fail(c, "no halo buffers on a single-slab context");
  launch_phi_halo_pack(c);
  LAUNCHCHK(c);
  return stage_mark(c, 2);
}

extern "C" int ekpnp_poisson_stage3(ekpnp_ctx* ctx) {
  NEEDCTX(ctx);
  if (!c.slab) return fail(c, "single-slab context: use ekpnp_fast_poisson");
  if (int mrc = stage_mark(c, 3)) return mrc;
  // the phi planes of the neighbouring slabs have arrived (PHI exchange): either E is written now, or the next collide
  // forms it from phi and these very halo planes (which the next PHI exchange only overwrites after that collide)
  const bool lazy = c.phi_old == nullptr && lazy_efield_ok(c);  // not inside the Poisson-Boltzmann sweeps (ekpnp_pbe_begin .. ekpnp_pbe_end), see poisson_single
  if (!lazy) launch_phi_efield(c);
  mark_solved(c, lazy);
  LAUNCHCHK(c);
  return poisson_timing_mark(c, false);
}
