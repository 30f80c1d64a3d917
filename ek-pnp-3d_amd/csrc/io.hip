// io.hip — diagnostics and the main.cu IO surface behind the C ABI.
//
//   double current(c, cn, ez)             LBM.cu:2674-2710, main.cu:211-216  -> ekpnp_current
//   record_umax(FILE*, t, ux, uy, uz)     LBM.cu:2712-2753, main.cu:221      -> ekpnp_umax, ekpnp_record_umax
//   save_data_tecplot(FILE*, t, ..., 1)   LBM.cu:2492-2565, main.cu:179,207  -> ekpnp_save_data_tecplot
//   save_data_end(FILE*, t, ...)          LBM.cu:2567-2630, main.cu:256      -> ekpnp_save_data_end
//   read_data(&t, ...)                    LBM.cu:2632-2671, main.cu:163      -> ekpnp_read_data
// The writers keep the reference's text formats byte for byte (Tecplot POINT zones, the lossy
// "%10.6f" restart file) and its wall extrapolation of rho, c, cn, u (LBM.cu:2527-2542); a FILE*
// cannot cross a C ABI, so they take a path and an append flag.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "ekpnp_internal.h"

using namespace ekpnp;

#define NEEDCTX(ctx)                    \
  if (!(ctx)) return EKPNP_ERR_INVALID; \
  Ctx& c = (ctx)->c
#define HIPCHK(ctx, call)                                                  \
  do {                                                                     \
    hipError_t e_ = (call);                                                \
    if (e_ != hipSuccess) {                                                \
      (ctx).err = std::string(#call) + ": " + hipGetErrorString(e_);       \
      return EKPNP_ERR_HIP;                                                \
    }                                                                      \
  } while (0)

static int fail(Ctx& c, const char* msg) {
  c.err = msg;
  return EKPNP_ERR_INVALID;
}

static int need_scratch(Ctx& c) {
  if (!c.diag) HIPCHK(c, hipMalloc((void**)&c.diag, DIAG_SCRATCH * sizeof(double)));
  return EKPNP_OK;
}

extern "C" int ekpnp_current(ekpnp_ctx* ctx, double* I) {
  NEEDCTX(ctx);
  if (!I) return fail(c, "NULL pointer");
  *I = 0.0;
  const bool combine = c.team && !team_is_group(c);  // attached transport: every rank returns the lattice's value
  if (c.z0 + c.nzl == c.p.nz) {  // a slab that does not hold the upper plate contributes 0
    if (c.nzl < 3) return fail(c, "the upper slab needs 3 planes for the wall extrapolation");
    int rc = need_scratch(c);
    if (rc) return rc;
    if ((rc = ensure_efield(c))) return rc;  // reads the Ez array
    launch_current(c, c.diag);
    if (take_launch_error(c) != hipSuccess) return EKPNP_ERR_HIP;
    double s = 0.0;
    HIPCHK(c, hipMemcpyAsync(&s, c.diag + 1024, sizeof(double), hipMemcpyDeviceToHost, c.stream));
    HIPCHK(c, hipStreamSynchronize(c.stream));
    *I = s * c.p.K * c.p.dz * c.p.dz;  // LBM.cu:2708
  }
  return combine ? team_ctx_reduce(c, I, false) : EKPNP_OK;
}

extern "C" int ekpnp_umax(ekpnp_ctx* ctx, double* umax) {
  NEEDCTX(ctx);
  if (!umax) return fail(c, "NULL pointer");
  int rc = need_scratch(c);
  if (rc) return rc;
  launch_umax(c, c.diag);
  if (take_launch_error(c) != hipSuccess) return EKPNP_ERR_HIP;
  double s = 0.0;
  HIPCHK(c, hipMemcpyAsync(&s, c.diag + 1024, sizeof(double), hipMemcpyDeviceToHost, c.stream));
  HIPCHK(c, hipStreamSynchronize(c.stream));
  *umax = s;
  return (c.team && !team_is_group(c)) ? team_ctx_reduce(c, umax, true) : EKPNP_OK;
}

extern "C" int ekpnp_record_umax(ekpnp_ctx* ctx, const char* path, int append, double time) {
  NEEDCTX(ctx);
  if (!path) return fail(c, "NULL path");
  double um = 0.0;
  int rc = ekpnp_umax(ctx, &um);
  if (rc) return rc;
  if (c.team && !team_is_group(c) && c.rank != 0) return EKPNP_OK;  // one line per call: rank 0 writes the combined value
  FILE* f = std::fopen(path, append ? "ab" : "wb");
  if (!f) return fail(c, "cannot open umax file");
  std::fprintf(f, "%10.6f %10.6f\n", time, um);  // LBM.cu:2748
  std::fclose(f);
  return EKPNP_OK;
}

// host copies of the 11 fields of this context's planes, with the reference's wall extrapolation
// of rho, c, cn, u on the plates it holds (LBM.cu:2527-2542 / 2596-2611); planes 0..2 and
// NZ-3..NZ-1 always lie inside one slab (a slab has at least 4 planes)
static int fetch_fields(Ctx& c, std::vector<std::vector<double>>& h, bool extrapolate) {
  if (int rc = ensure_efield(c)) return rc;
  HIPCHK(c, hipStreamSynchronize(c.stream));
  h.assign(EKPNP_NFIELDS, std::vector<double>(c.nloc));
  for (int i = 0; i < EKPNP_NFIELDS; ++i) HIPCHK(c, hipMemcpy(h[i].data(), c.fld[i], c.nloc * sizeof(double), hipMemcpyDeviceToHost));
  if (extrapolate) {
    const size_t pl = c.plane, nzl = c.nzl;
    const bool lower = c.z0 == 0, upper = c.z0 + c.nzl == c.p.nz;
    if ((lower || upper) && nzl < 3) return fail(c, "a slab holding a plate needs 3 planes for the wall extrapolation");
    const int ids[6] = {EKPNP_RHO, EKPNP_C, EKPNP_CN, EKPNP_UX, EKPNP_UY, EKPNP_UZ};
    for (int k = 0; k < 6; ++k) {
      double* a = h[ids[k]].data();
      for (size_t i = 0; i < pl; ++i) {
        if (lower) a[i] = 2.0 * a[pl + i] - a[2 * pl + i];
        if (upper) a[(nzl - 1) * pl + i] = 2.0 * a[(nzl - 2) * pl + i] - a[(nzl - 3) * pl + i];
      }
    }
  }
  return EKPNP_OK;
}

namespace ekpnp {
// This context's planes of a whole-lattice text file: kind 0 = Tecplot POINT zone (header lines
// when it holds plane 0), kind 1 = the 12-column restart file.  Slabs append in z order.
int io_write_text_part(Ctx& c, const TextIoArgs& a) {
  std::vector<std::vector<double>> h;
  int rc = fetch_fields(c, h, true);
  if (rc) return rc;
  FILE* f = std::fopen(a.path, a.append ? "ab" : "wb");
  if (!f) return fail(c, a.kind == 0 ? "cannot open Tecplot file" : "cannot open restart file");
  if (a.kind == 0 && c.z0 == 0) {
    if (a.first)  // LBM.cu:2546-2548
      std::fprintf(f, "%s\n", "VARIABLES=\"x\",\"y\",\"z\",\"u\",\"v\",\"w\",\"p\",\"charge\",\"neg charge\",\"phi\",\"Ex\",\"Ey\",\"Ez\",\"Temperature\"");
    std::fprintf(f, "\n");
    std::fprintf(f, "ZONE T=\"t=%g\", F=POINT, I = %d, J = %d, K = %d\n", a.time, c.p.nx, c.p.ny, c.p.nz);  // LBM.cu:2551
  }
  const double dx = c.p.dx, dy = c.p.dy, dz = c.p.dz;
  size_t i = 0;
  for (unsigned zl = 0; zl < (unsigned)c.nzl; ++zl)
    for (unsigned y = 0; y < (unsigned)c.p.ny; ++y)
      for (unsigned x = 0; x < (unsigned)c.p.nx; ++x, ++i) {
        if (a.kind == 0)  // LBM.cu:2559-2561
          std::fprintf(f, "%g %g %g %g %g %g %g %g %10.6f %10.6f %10.6f %10.6f %10.6f %10.6f\n", dx * x, dy * y, dz * (zl + (unsigned)c.z0),
                       h[EKPNP_UX][i], h[EKPNP_UY][i], h[EKPNP_UZ][i], h[EKPNP_RHO][i], h[EKPNP_C][i], h[EKPNP_CN][i], h[EKPNP_PHI][i],
                       h[EKPNP_EX][i], h[EKPNP_EY][i], h[EKPNP_EZ][i], h[EKPNP_T][i]);
        else  // LBM.cu:2619-2622
          std::fprintf(f, "%10.6f %10.6f %10.6f %10.6f %10.6f %10.6f %10.6f %10.6f %10.6f %10.6f %10.6f %10.6f\n", a.time, h[EKPNP_UX][i],
                       h[EKPNP_UY][i], h[EKPNP_UZ][i], h[EKPNP_RHO][i], h[EKPNP_C][i], h[EKPNP_CN][i], h[EKPNP_PHI][i], h[EKPNP_EX][i],
                       h[EKPNP_EY][i], h[EKPNP_EZ][i], h[EKPNP_T][i]);
      }
  const bool bad = std::ferror(f) != 0;
  return (std::fclose(f) != 0 || bad) ? fail(c, "write error on output file") : EKPNP_OK;
}

// This context's planes out of a save_data_end file of the whole lattice (LBM.cu:2645-2657)
int io_read_data_part(Ctx& c, const char* path, double* time) {
  FILE* f = std::fopen(path, "r");
  if (!f) return fail(c, "cannot open restart file");
  std::vector<std::vector<double>> h(EKPNP_NFIELDS, std::vector<double>(c.nloc));
  bool ok = true;
  double skip[12];
  for (size_t i = 0, n = (size_t)c.z0 * c.plane; ok && i < n; ++i)  // the planes below this slab
    ok = std::fscanf(f, "%lf %lf %lf %lf %lf %lf %lf %lf %lf %lf %lf %lf", skip, skip + 1, skip + 2, skip + 3, skip + 4, skip + 5, skip + 6,
                     skip + 7, skip + 8, skip + 9, skip + 10, skip + 11) == 12;
  for (size_t i = 0; ok && i < c.nloc; ++i)  // LBM.cu:2652-2655
    ok = std::fscanf(f, "%lf %lf %lf %lf %lf %lf %lf %lf %lf %lf %lf %lf", time, &h[EKPNP_UX][i], &h[EKPNP_UY][i], &h[EKPNP_UZ][i],
                     &h[EKPNP_RHO][i], &h[EKPNP_C][i], &h[EKPNP_CN][i], &h[EKPNP_PHI][i], &h[EKPNP_EX][i], &h[EKPNP_EY][i], &h[EKPNP_EZ][i],
                     &h[EKPNP_T][i]) == 12;
  std::fclose(f);
  if (!ok) return fail(c, "restart file is shorter than the lattice or malformed");
  if (int rc = efield_set_from_outside(c)) return rc;
  HIPCHK(c, hipStreamSynchronize(c.stream));
  for (int i = 0; i < EKPNP_NFIELDS; ++i) HIPCHK(c, hipMemcpy(c.fld[i], h[i].data(), c.nloc * sizeof(double), hipMemcpyHostToDevice));
  c.t = *time;
  c.rhs_ready = false;
  return EKPNP_OK;
}
}  // namespace ekpnp

namespace {
int write_part_turn(Ctx& c, void* arg) {
  TextIoArgs a = *static_cast<TextIoArgs*>(arg);
  if (c.rank != 0) a.append = 1;
  return io_write_text_part(c, a);
}
struct ReadTurn { const char* path; double* time; };
int read_part_turn(Ctx& c, void* arg) {
  ReadTurn* a = static_cast<ReadTurn*>(arg);
  return io_read_data_part(c, a->path, a->time);
}
const char* const kNoTransport = "whole-lattice file IO on a slab context without a transport of its own: attach one (ekpnp_slab_attach_comm) or use ekpnp_group_*";
}  // namespace

extern "C" int ekpnp_save_data_tecplot(ekpnp_ctx* ctx, const char* path, int append, double time, int first) {
  NEEDCTX(ctx);
  if (!path) return fail(c, "NULL path");
  TextIoArgs a{path, append, time, first, 0};
  if (c.team && !team_is_group(c)) return team_ctx_turns(c, write_part_turn, &a);  // the ranks take turns in z order
  if (c.slab && c.nranks > 1) return fail(c, kNoTransport);
  return io_write_text_part(c, a);
}

extern "C" int ekpnp_save_data_end(ekpnp_ctx* ctx, const char* path, int append, double time) {
  NEEDCTX(ctx);
  if (!path) return fail(c, "NULL path");
  TextIoArgs a{path, append, time, 0, 1};
  if (c.team && !team_is_group(c)) return team_ctx_turns(c, write_part_turn, &a);
  if (c.slab && c.nranks > 1) return fail(c, kNoTransport);
  return io_write_text_part(c, a);
}

extern "C" int ekpnp_read_data(ekpnp_ctx* ctx, const char* path, double* time) {
  NEEDCTX(ctx);
  if (!path || !time) return fail(c, "NULL pointer");
  if (c.team && !team_is_group(c)) {
    ReadTurn a{path, time};
    return team_ctx_turns(c, read_part_turn, &a);
  }
  if (c.slab && c.nranks > 1) return fail(c, kNoTransport);
  return io_read_data_part(c, path, time);
}

// ---- lossless binary state (no reference counterpart beyond the save_data_end / read_data pair) ----
extern "C" int ekpnp_save_state(ekpnp_ctx* ctx, const char* path, double time) {
  NEEDCTX(ctx);
  if (!path) return fail(c, "NULL path");
  FILE* f = std::fopen(path, "wb");
  if (!f) return fail(c, "cannot open state file");
  StateHeader h{};
  std::memcpy(h.magic, "EKPNPST1", 8);
  h.nx = c.p.nx; h.ny = c.p.ny; h.nz = c.p.nz; h.z0 = c.z0; h.nzl = c.nzl; h.nfields = EKPNP_NFIELDS;
  h.time = time;
  bool ok = std::fwrite(&h, sizeof h, 1, f) == 1;
  std::vector<double> buf(c.nloc < STATE_CHUNK ? c.nloc : STATE_CHUNK);
  if (ensure_efield(c) != EKPNP_OK) { std::fclose(f); return EKPNP_ERR_HIP; }
  hipError_t e = hipStreamSynchronize(c.stream);
  for (int i = 0; ok && e == hipSuccess && i < EKPNP_NFIELDS; ++i)
    for (size_t o = 0; ok && e == hipSuccess && o < c.nloc; o += buf.size()) {
      const size_t n = c.nloc - o < buf.size() ? c.nloc - o : buf.size();
      e = hipMemcpy(buf.data(), c.fld[i] + o, n * sizeof(double), hipMemcpyDeviceToHost);
      if (e == hipSuccess) ok = std::fwrite(buf.data(), sizeof(double), n, f) == n;
    }
  ok = (std::fclose(f) == 0) && ok;
  HIPCHK(c, e);
  return ok ? EKPNP_OK : fail(c, "write error on state file");
}

extern "C" int ekpnp_read_state(ekpnp_ctx* ctx, const char* path, double* time) {
  NEEDCTX(ctx);
  if (!path || !time) return fail(c, "NULL pointer");
  FILE* f = std::fopen(path, "rb");
  if (!f) return fail(c, "cannot open state file");
  StateHeader h{};
  if (std::fread(&h, sizeof h, 1, f) != 1 || std::memcmp(h.magic, "EKPNPST1", 8) != 0) {
    std::fclose(f);
    return fail(c, "not an EKPNPST1 state file");
  }
  if (h.nx != c.p.nx || h.ny != c.p.ny || h.nz != c.p.nz || h.z0 != c.z0 || h.nzl != c.nzl || h.nfields != EKPNP_NFIELDS) {
    std::fclose(f);
    return fail(c, "state file was written for a different lattice or slab");
  }
  std::vector<double> buf(c.nloc < STATE_CHUNK ? c.nloc : STATE_CHUNK);
  bool ok = true;
  if (efield_set_from_outside(c) != EKPNP_OK) { std::fclose(f); return EKPNP_ERR_HIP; }
  hipError_t e = hipStreamSynchronize(c.stream);
  for (int i = 0; ok && e == hipSuccess && i < EKPNP_NFIELDS; ++i)
    for (size_t o = 0; ok && e == hipSuccess && o < c.nloc; o += buf.size()) {
      const size_t n = c.nloc - o < buf.size() ? c.nloc - o : buf.size();
      ok = std::fread(buf.data(), sizeof(double), n, f) == n;
      if (ok) e = hipMemcpy(c.fld[i] + o, buf.data(), n * sizeof(double), hipMemcpyHostToDevice);
    }
  std::fclose(f);
  HIPCHK(c, e);
  if (!ok) return fail(c, "state file is shorter than the lattice");
  *time = h.time;
  c.t = h.time;
  c.rhs_ready = false;
  return EKPNP_OK;
}

// ---- full checkpoint: fields AND populations (no reference counterpart; SURVEY.md §5 "checkpoint /
// resume ... if built later, binary FP64").  Unlike the restart above, loading it continues the
// interrupted run bit for bit: the post-collision populations of all lattices travel too.
// Layout: 64-byte header, the 11 fields (owned planes), then per lattice the owned population
// planes in the library's tiled layout [z][y][x/64][27][64] (ekpnp_internal.h); a slab context on
// its own also stores its two ghost planes (flag with_ghosts) so that a per-rank file is
// self-contained.  A whole-lattice file (z0 = 0, nz_local = nz, no ghosts) is what a single context
// and ekpnp_group_save_checkpoint write; either can load it, in two-buffer or in-place mode.
namespace ekpnp {
static_assert(sizeof(CkptHeader) == 64, "checkpoint header layout");

int io_ckpt_write_header(Ctx& c, FILE* f, int z0, int nzl, int with_ghosts, double time) {
  CkptHeader h{};
  std::memcpy(h.magic, "EKPNPCK2", 8);
  h.nx = c.p.nx; h.ny = c.p.ny; h.nz = c.p.nz; h.z0 = z0; h.nzl = nzl; h.nfields = EKPNP_NFIELDS;
  h.nl = c.p.n_lattices; h.streamed_state = c.streamed_state ? 1 : 0; h.with_ghosts = with_ghosts;
  h.time = time;
  return std::fwrite(&h, sizeof h, 1, f) == 1 ? EKPNP_OK : fail(c, "write error on checkpoint file");
}

// device memory <-> file through a bounce buffer; dir 0: device -> file, 1: file -> device
int io_stream_device(Ctx& c, FILE* f, double* dev, size_t n, int dir) {
  static thread_local std::vector<double> buf;
  if (buf.size() < STATE_CHUNK) buf.resize(STATE_CHUNK);
  for (size_t o = 0; o < n; o += STATE_CHUNK) {
    const size_t m = n - o < STATE_CHUNK ? n - o : STATE_CHUNK;
    if (dir == 0) {
      HIPCHK(c, hipMemcpy(buf.data(), dev + o, m * sizeof(double), hipMemcpyDeviceToHost));
      if (std::fwrite(buf.data(), sizeof(double), m, f) != m) return fail(c, "write error on checkpoint file");
    } else {
      if (std::fread(buf.data(), sizeof(double), m, f) != m) return fail(c, "checkpoint file is shorter than the lattice");
      HIPCHK(c, hipMemcpy(dev + o, buf.data(), m * sizeof(double), hipMemcpyHostToDevice));
    }
  }
  return EKPNP_OK;
}

int io_ckpt_fields(Ctx& c, FILE* f, int field, int dir) {
  if (int rc = dir == 0 ? ensure_efield(c) : efield_set_from_outside(c)) return rc;
  if (dir == 0) HIPCHK(c, hipStreamSynchronize(c.stream));
  return io_stream_device(c, f, c.fld[field], c.nloc, dir);
}

// owned planes zg = 1..nzl of lattice l's current state (with_ghosts: zg = 0..nzl+1)
int io_ckpt_populations(Ctx& c, FILE* f, int l, int with_ghosts, int dir) {
  double* base = c.cur_base(l);
  if (!base) return fail(c, "lattice not allocated");
  return with_ghosts ? io_stream_device(c, f, base, (size_t)(c.nzl + 2) * c.pplane, dir)
                     : io_stream_device(c, f, base + c.pplane, (size_t)c.nzl * c.pplane, dir);
}

int io_ckpt_check_header(Ctx& c, const CkptHeader& h, int z0, int nzl) {
  if (std::memcmp(h.magic, "EKPNPCK2", 8) != 0) return fail(c, "not an EKPNPCK2 checkpoint file");
  if (h.nx != c.p.nx || h.ny != c.p.ny || h.nz != c.p.nz || h.nfields != EKPNP_NFIELDS || h.nl != c.p.n_lattices)
    return fail(c, "checkpoint was written for a different lattice");
  if (h.z0 != z0 || h.nzl != nzl) return fail(c, "checkpoint holds other planes than this context / group owns");
  return EKPNP_OK;
}

void io_ckpt_finish_load(Ctx& c, const CkptHeader& h) {
  c.streamed_state = h.streamed_state != 0;
  c.t = h.time;
  c.rhs_ready = false;
  c.collide_phase = 0;
  c.halo_sent = false;
  c.halo_recv_valid = false;  // the loaded ghost planes are what the next pull of a slab's edge planes reads
}
}  // namespace ekpnp

extern "C" int ekpnp_save_checkpoint(ekpnp_ctx* ctx, const char* path) {
  NEEDCTX(ctx);
  if (!path) return fail(c, "NULL path");
  if (c.collide_phase != 0) return fail(c, "ekpnp_save_checkpoint between the boundary and the interior collide call");
  if (int grc = ensure_ghost_planes(c)) return grc;  // a slab's file carries its ghost planes: the halos may still sit in the receive buffers
  HIPCHK(c, hipStreamSynchronize(c.stream));
  FILE* f = std::fopen(path, "wb");
  if (!f) return fail(c, "cannot open checkpoint file");
  const int ghosts = c.slab ? 1 : 0;
  int rc = io_ckpt_write_header(c, f, c.z0, c.nzl, ghosts, c.t);
  for (int i = 0; rc == EKPNP_OK && i < EKPNP_NFIELDS; ++i) rc = io_ckpt_fields(c, f, i, 0);
  for (int l = 0; rc == EKPNP_OK && l < c.p.n_lattices; ++l) rc = io_ckpt_populations(c, f, l, ghosts, 0);
  if (std::fclose(f) != 0 && rc == EKPNP_OK) rc = fail(c, "write error on checkpoint file");
  return rc;
}

extern "C" int ekpnp_load_checkpoint(ekpnp_ctx* ctx, const char* path, double* time) {
  NEEDCTX(ctx);
  if (!path) return fail(c, "NULL path");
  HIPCHK(c, hipStreamSynchronize(c.stream));
  FILE* f = std::fopen(path, "rb");
  if (!f) return fail(c, "cannot open checkpoint file");
  CkptHeader h{};
  int rc = std::fread(&h, sizeof h, 1, f) == 1 ? io_ckpt_check_header(c, h, c.z0, c.nzl) : fail(c, "not an EKPNPCK2 checkpoint file");
  if (rc == EKPNP_OK && c.slab && c.nranks > 1 && !h.with_ghosts)
    rc = fail(c, "a slab context loads its own per-rank checkpoint (with ghost planes); whole-lattice files go through ekpnp_group_load_checkpoint");
  if (rc == EKPNP_OK && !c.slab && h.with_ghosts) rc = fail(c, "this is a slab's per-rank checkpoint");
  for (int i = 0; rc == EKPNP_OK && i < EKPNP_NFIELDS; ++i) rc = io_ckpt_fields(c, f, i, 1);
  for (int l = 0; rc == EKPNP_OK && l < c.p.n_lattices; ++l) rc = io_ckpt_populations(c, f, l, h.with_ghosts, 1);
  std::fclose(f);
  if (rc) return rc;
  io_ckpt_finish_load(c, h);
  if (!h.with_ghosts) {  // gpu_stream's z wrap (LBM.cu:1972,1975): ghost planes <- opposite wall planes
    launch_ghost_wrap(c);
    if (take_launch_error(c) != hipSuccess) return EKPNP_ERR_HIP;
    HIPCHK(c, hipStreamSynchronize(c.stream));
  }
  if (time) *time = h.time;
  return EKPNP_OK;
}

extern "C" int ekpnp_compute_parameters(const ekpnp_params* p, double* T, double* M, double* C, double* Fe, double* Pr) {
  if (!p) return EKPNP_ERR_INVALID;
  // LBM.cu:2440-2444, same expressions (charge0_host there is chargeinf, LBM.cu:2436)
  if (M) *M = std::sqrt(p->eps / p->rho0) / p->K;
  if (T) *T = p->eps * p->voltage / p->K / p->nu / p->rho0;
  if (C) *C = p->chargeinf * p->Lz * p->Lz / (p->voltage * p->eps);
  if (Fe) *Fe = p->K * p->voltage / p->diffu;
  if (Pr) *Pr = p->nu / p->D;
  return EKPNP_OK;
}

extern "C" int ekpnp_save_scalar(ekpnp_ctx* ctx, const char* name, int field_id, unsigned n, unsigned nsteps) {
  NEEDCTX(ctx);
  if (!name || field_id < 0 || field_id >= EKPNP_NFIELDS) return fail(c, "bad name or field id");
  if (std::strlen(name) > 100) return fail(c, "file name too long");  // the reference assumes it fits 128 chars
  const int ndigits = (int)std::floor(std::log10((double)(nsteps ? nsteps : 1)) + 1.0);  // LBM.cu:2461
  char format[16], filename[160];
  std::snprintf(format, sizeof format, "%%s%%0%dd.bin", ndigits);                        // LBM.cu:2465
  std::snprintf(filename, sizeof filename, format, name, n);
  std::vector<double> h(c.nloc);
  if (int rc = ensure_efield(c)) return rc;
  HIPCHK(c, hipStreamSynchronize(c.stream));
  HIPCHK(c, hipMemcpy(h.data(), c.fld[field_id], c.nloc * sizeof(double), hipMemcpyDeviceToHost));
  FILE* f = std::fopen(filename, "wb");
  if (!f) return fail(c, "cannot open scalar file");
  const size_t w = std::fwrite(h.data(), 1, c.nloc * sizeof(double), f);  // LBM.cu:2475
  std::fclose(f);
  return w == c.nloc * sizeof(double) ? EKPNP_OK : fail(c, "short write on scalar file");
}
