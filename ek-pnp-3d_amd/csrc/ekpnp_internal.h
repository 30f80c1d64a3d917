// ekpnp_internal.h — shared between the HIP kernels and the C-ABI host (private).
#pragma once
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>

#include <array>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/ekpnp.h"

namespace ekpnp {

constexpr int Q = 27;
constexpr int MAXL = 4;  // f, h, hn, temp

// D3Q27 velocity set in the reference's numbering (gpu_stream, LBM.cu:1983-2008):
// population d at node x was at x - c_d before streaming; opposite of odd d is d+1.
__host__ __device__ constexpr int ex_of(int d) {
  constexpr int t[Q] = {0, 1, -1, 0, 0, 0, 0, 1, -1, 1, -1, 0, 0, 1, -1, 1, -1, 0, 0, 1, -1, 1, -1, 1, -1, -1, 1};
  return t[d];
}
__host__ __device__ constexpr int ey_of(int d) {
  constexpr int t[Q] = {0, 0, 0, 1, -1, 0, 0, 1, -1, 0, 0, 1, -1, -1, 1, 0, 0, 1, -1, 1, -1, 1, -1, -1, 1, 1, -1};
  return t[d];
}
__host__ __device__ constexpr int ez_of(int d) {
  constexpr int t[Q] = {0, 0, 0, 0, 0, 1, -1, 0, 0, 1, -1, 1, -1, 0, 0, -1, 1, -1, 1, 1, -1, -1, 1, 1, -1, 1, -1};
  return t[d];
}
__host__ __device__ constexpr int opp_of(int d) { return d == 0 ? 0 : ((d & 1) ? d + 1 : d - 1); }
// Slot of direction d inside a population tile ([Q][64] doubles), round 5: the 27 directions ordered by (c_z, c_y, c_x)
// instead of the reference's numbering, so that the three directions a destination row pulls from ONE source row (same c_y,
// c_z) are 1.5 KB of neighbours and the nine a destination plane pulls from one source plane are 4.6 KB - bulk kernel 38.69 -
// 38.87 -> 38.62 - 38.71 ms on cfg3 (alternating legs on one box, profiles/r05g_ab_grouped_slots.log; round 1's copy probe had
// priced a scattered numbering at 0.35 %).  The numbering of the ARITHMETIC (d, opp_of, the order of every sum) is the
// reference's as before: only where a population lies in its tile changed - and with it the population part of the
// checkpoint files, hence their new format id "EKPNPCK2".  EKPNP_SLOT_BY_DIRECTION builds the A/B partner (slot = d).
__host__ __device__ constexpr int slot_of(int d) {
#ifdef EKPNP_SLOT_BY_DIRECTION
  return d;
#else
  return (ez_of(d) + 1) * 9 + (ey_of(d) + 1) * 3 + (ex_of(d) + 1);
#endif
}
__host__ __device__ constexpr double w_of(int d) {
  return d == 0 ? 8.0 / 27.0 : d <= 6 ? 2.0 / 27.0 : d <= 18 ? 1.0 / 54.0 : 1.0 / 216.0;  // LBM.h:109-112
}

// The 9 directions that cross a z face upwards / downwards.
__host__ __device__ constexpr int up_dir(int k) {
  constexpr int t[9] = {5, 9, 11, 16, 18, 19, 22, 23, 25};
  return t[k];
}
__host__ __device__ constexpr int dn_dir(int k) {
  constexpr int t[9] = {6, 10, 12, 15, 17, 20, 21, 24, 26};
  return t[k];
}

// Population layout (one array per lattice): [zg][y][x/64][Q slots][64] - a TILE holds the 27 populations
// of 64 consecutive x nodes (13.8 KB; direction d in slot slot_of(d), above), a row is ceil(nx/64) tiles, plane zg = 0 / nzl+1 are the
// ghost planes.  The bulk kernel's wave writes ONE contiguous tile and pulls from the tiles of 9
// neighbour rows; with the direction-major layout [Q][zg][y][x] it touched 27 + 27 streams that
// lie gigabytes apart (pure-copy ceiling of the two shapes on one box: 5.95 vs 5.48 TB/s,
// profiles/r01_stream_probe_aosoa.log).  Lanes of the last tile beyond nx are never touched.
constexpr int TILE = Q * 64;
#ifndef EKPNP_TRI_CHECK
#define EKPNP_TRI_CHECK 16  // tuning knob (8 / 32 measured: profiles/r02_tridiag_variants.log)
#endif
constexpr int TRI_CHECK = EKPNP_TRI_CHECK;  // the z solves keep every TRI_CHECK-th row of c' and of d' (poisson.hip TRI_BS)
__host__ __device__ inline long long pop_xoff(int x) { return (long long)(x >> 6) * TILE + (x & 63); }

// z partition of the slabs: rank r owns planes [slab_begin(r), slab_begin(r+1)); sizes differ by at most one
// plane (NZ is typically 2^k + 1 in the reference's runs: LBM.h:35,37, main.cu:112), walls on ranks 0 and P-1
inline int slab_begin(int nz, int nranks, int r) { return (int)((long long)r * nz / nranks); }
// unknown rows of the global z system that rank r owns (the plates carry none)
inline int slab_rows(int nz, int nranks, int r) {
  const int z0 = slab_begin(nz, nranks, r), z1 = slab_begin(nz, nranks, r + 1);
  return (z1 - z0) - (z0 == 0 ? 1 : 0) - (z1 == nz ? 1 : 0);
}

// Everything a kernel needs, passed by value.
struct KArgs {
  // population buffers, tiled as above; B may also be the 2-plane staging buffer (same layout)
  const double* A[MAXL];
  double* B[MAXL];
  double* fld[EKPNP_NFIELDS];  // [nzl][ny][nx]
  int nx, ny, nz;              // global lattice
  int nzl, z0;                 // owned planes and the global index of the first one
  int zwrap;                   // 1: whole lattice in one two-buffer context, wall nodes wrap z by index
  long long plane;             // nx*ny (macroscopic fields)
  long long rowstride;         // doubles per population row: ceil(nx/64)*TILE
  // derived physics (host-side, in the reference's expression order, LBM.cu:488-495,1660-1661)
  double wp[MAXL], wm[MAXL];   // omega_plus*dt / omega_minus*dt per lattice
  double mob[MAXL];            // drift mobility: 0, K, Kn, 0
  double sp, sm;               // 1 - dt*omega/2
  double cflinv, inv_cs2, cflinv2, dt;
  double F, Ext, exf, Ra, nu, D;
  double TH, uw_multi;         // uw_multi = 2*rho0*uw/cs_square/CFL (times w_d in the kernel)
  double rho0;
  // Poisson right-hand side written by the collide itself (poisson.cu:114-135 fused in)
  double* rhs;                 // [nzl][ny][nx] or null
  double eps;
  double rhs_wall_lo, rhs_wall_hi;  // voltage/dz/dz (plane 1), voltage2/dz/dz (plane nz-2)
  // E taken from phi inside the collide (EPHI kernels, round 4): gpu_efield / gpu_bc (poisson.cu:40-69) evaluated
  // where E is consumed, with the expression of k_phi_efield, so the Ex / Ey / Ez arrays need not be written per step
  const double* phi_lo;        // phi plane below / above the slab (slab contexts), or null
  const double* phi_hi;
  double voltage, voltage2, dx, dy, dz;
  // Slab edge planes without pack / unpack copies (round 4): the launches that collide a slab's first and last plane
  // store their 9 outgoing directions ALSO straight into the send buffers (layout [lattice][9][ny][nx], k_halo_pack's),
  // and pull their 9 incoming directions straight out of the receive buffers instead of a ghost plane.  Null: not a
  // slab edge launch / the ghost planes are what is valid (after a checkpoint load, or EKPNP_HALO_DIRECT=0).
  double* halo_out_dn;
  double* halo_out_up;
  const double* halo_in_lo;
  const double* halo_in_hi;
  // 0: the bulk launch does not store rho, u, c, cn, T (an intermediate step of one ekpnp_step(n) call under the opt-in knob
  // "batch_moments": nothing can look at them before the call's last step has written them); everything else is the same
  int wmom;
};

// position of direction d among the 9 directions that cross a z face the way d does (up_dir / dn_dir): its slot in a halo buffer
__host__ __device__ constexpr int halo_slot(int d) {
  for (int k = 0; k < 9; ++k)
    if (up_dir(k) == d || dn_dir(k) == d) return k;
  return -1;
}

// Right-hand side of the Poisson equation on an interior plane, odd_extension's rows 1..NZ-2
// (poisson.cu:121-135) in the reference's expression order: -F (c - cn)/eps, minus voltage/dz/dz on
// the plane next to the lower plate, minus voltage2/dz/dz next to the upper one.  ONE function for
// the collide kernel (which writes it from registers) and for k_poisson_rhs, so that
// ekpnp_fast_poisson returns the same bits whichever of the two produced its input.
__host__ __device__ inline double poisson_rhs_value(double F, double eps, double c, double cn, int z, int nz, double wall_lo, double wall_hi) {
  double v = -F * (c - cn) / eps;
  if (z == 1) v = v - wall_lo;
  if (z == nz - 2) v = v - wall_hi;
  return v;
}

struct PArgs {
  double* fld[EKPNP_NFIELDS];
  double* work;                // real [nzl][ny][nx]
  double2* spec;               // complex [nzl][ny][nxh]
  const double* cprime;        // Thomas factors c': slabs [nzl+2][ny][nxh] (local rows); single context every 16th row only
  const double* phi_lo;        // phi plane below / above the slab (slab mode), may be null
  const double* phi_hi;
  const double* vwall;         // {voltage, voltage, voltage2, voltage2} in device memory
  int nx, ny, nz, nxh, nzl, z0;
  long long plane;
  double F, eps, voltage, voltage2, dx, dy, dz, inv_nxny, Lx, Ly;
  double rhs_wall_lo, rhs_wall_hi;  // as in KArgs
  int bx0, bw;                 // slab z solve: the mode block (kx in [bx0, bx0 + bw)) a launch works on; the whole spectrum: 0, nxh
};

// one mode block of a slab's z solve (poisson.hip: mode_block): its kx columns and its pieces of the edge buffers
struct ModeBlock {
  int x0, bw;
  size_t local_off, all_off, doubles;  // offsets (doubles) into Ctx::edge_local / Ctx::edge_all; doubles per rank (= 4 ny bw)
};

struct Ctx;
struct Team;

// lbm_kernels.hip
void launch_init_fields(Ctx&);
void launch_pbe(Ctx&);
void launch_pbe_relax(Ctx&, double* phi_old, double omega);
void launch_init_equilibrium(Ctx&);
int bulk_band_rows(const Ctx&, int rchunk = 64);  // rows per band of the interior sweep in effect (0: plane after plane)
void launch_collide_all(Ctx&);  // launch-bound lattices: plates and bulk in ONE launch (single two-buffer context)
void launch_collide_bulk(Ctx&, int zl_begin, int zl_end);
void launch_collide_bulk(Ctx&, const KArgs&, int zl_begin, int zl_end);
void launch_collide_faces(Ctx&, const KArgs& lo, const KArgs& hi);  // a slab's first and last plane (plate or interior face each) in ONE launch
void launch_collide_bulk_edge(Ctx&, const KArgs&, int zl);  // one slab edge plane: KArgs::halo_* honoured (EDGE kernels)
void launch_collide_walls(Ctx&, const KArgs&, hipStream_t stream, bool lower, bool upper);
void launch_halo_pack_stage(Ctx&);
void launch_unstage(Ctx&);
void launch_collide_walls(Ctx&, hipStream_t stream, bool lower, bool upper);
void launch_ghost_wrap(Ctx&);
void launch_halo_pack(Ctx&, int buffer);
void launch_halo_unpack(Ctx&);
// poisson.hip
int build_cprime(Ctx&);  // EKPNP_OK or a status with Ctx::err set
void launch_poisson_rhs(Ctx&);
int plane_fft_setup(Ctx&);      // decides Ctx::own_fft for this lattice and device, makes the twiddle table
int plane_fft_forward(Ctx&);    // fft_in() -> fft_spec(): the own kernels or the rocFFT plan; EKPNP_OK or a status with Ctx::err set
int plane_fft_inverse(Ctx&);    // fft_spec() -> fft_out()
void launch_tridiag(Ctx&, const ModeBlock* block = nullptr);  // the whole spectrum, or the kx columns of one block
int poisson_block_count(const Ctx&);                    // column blocks of a single context's solve (Ctx::poisson_blocks; 1 where they do not apply)
ModeBlock poisson_block(const Ctx&, int block);
ModeBlock poisson_block_whole(const Ctx&);              // all kx columns
bool tridiag_prepare_device();  // per-device function attributes of the partition z solves (current device)
bool tridiag_wide_prepare_device();  // ... of the 16-wavefront forms (128 KB of LDS)
void launch_phi_efield(Ctx&);
void launch_slab_thomas_local(Ctx&, int block = 0);    // stage 1 of the slab z solve on mode block `block`
void launch_slab_reduce_correct(Ctx&, int block = 0);  // stage 2 on mode block `block` (its edge values gathered)
int edge_chunk_count(const Ctx&);                       // mode blocks of this context's slab solve (Ctx::edge_chunks, at most nxh / 8)
ModeBlock mode_block(const Ctx&, int block);
int plane_fft_forward_rows(Ctx&, int z0 = 0, int nz = -1);                       // the transforms in pieces (rows of all planes / columns of one block)
void plane_fft_forward_columns(Ctx&, const ModeBlock&, int z0 = 0, int nz = -1);
void plane_fft_inverse_columns(Ctx&, const ModeBlock&, int z0 = 0, int nz = -1);
int plane_fft_inverse_rows(Ctx&, int z0 = 0, int nz = -1);
void launch_phi_halo_pack(Ctx&);
// diag.hip
constexpr int DIAG_SCRATCH = 1024 + 8;
void launch_current(Ctx&, double* scratch);
void launch_umax(Ctx&, double* scratch);
void launch_max_abs_diff(Ctx&, const double* p, const double* q, double* scratch);
void launch_copy16(Ctx&, const void* src, void* dst, size_t bytes);

struct Ctx {
  ekpnp_params p{};
  int device = 0;              // HIP device the context was created on (current device of every call into it)
  Team* team = nullptr;        // slab contexts: the exchange domain that moves its halos (slab_team.hip), or null
  int team_slot = 0;           // index of this context among the team's local slabs
  int rank = 0, nranks = 1;
  bool slab = false;           // created by ekpnp_create_slab: driven through the split calls + a transport
  int nzl = 0, z0 = 0, nxh = 0;
  size_t plane = 0, nloc = 0;
  size_t rowstride = 0, pplane = 0;  // doubles per population row / plane (tiled layout, ekpnp::TILE)
  hipStream_t stream = nullptr;
  bool own_stream = false;
  double* pop[2][MAXL] = {};   // [buffer][lattice]; in-place mode uses pop[0] only
  void* pop_alloc[2][MAXL] = {};  // what hipMalloc returned: [0][0] alone when all buffers share one allocation (default)
  int cur = 0;                 // A/B mode: buffer holding the current state.  In-place mode: the
                               // parity of the storage offset (0: lattice at +shift planes, the
                               // next sweep runs z-ascending and writes at offset 0; 1: mirror)
  bool inplace = false;
  int shift = 0;               // planes the lattice moves per sweep in in-place mode (0 in A/B mode)
  int zchunk = 0;              // planes per bulk launch in in-place mode (shift >= zchunk + 1)
  int ab_zchunk = 0;           // two-buffer mode: planes per bulk launch of the sweep (0: one launch)
  int bulk_yband = -1;         // the interior sweep (in place: each of its launches) takes bands of this many rows of EVERY plane, band after band, instead of
                               // plane after plane (lbm_kernels.hip: bulk_row_of_block): the three uses of a phi row lie within the Infinity
                               // Cache's reach.  -1: decided from the plane's traffic (bulk_dispatch), 0: plane after plane (ekpnp_tune
                               // "bulk_yband", EKPNP_BULK_YBAND)
  bool merged_walls = true;    // launch-bound two-buffer lattices: plates and bulk in one launch (k_collide_all)
  // base pointer of lattice l's CURRENT state (plane zg = 0 is the ghost plane below)
  double* cur_base(int l) const {
    if (!inplace) return pop[cur][l];
    return pop[0][l] ? pop[0][l] + (size_t)(cur == 0 ? shift : 0) * pplane : nullptr;
  }
  double* next_base(int l) const {
    if (!inplace) return pop[cur ^ 1][l];
    return pop[0][l] ? pop[0][l] + (size_t)(cur == 0 ? 0 : shift) * pplane : nullptr;
  }
  int placement_tries = 0, placement_chosen = 0;  // placement_search (capi.hip): arenas timed at creation, the one kept
  double placement_ms[8] = {};                    // their best sweep times
  bool rhs_ready = false;      // work[] holds the Poisson rhs of the current c, cn (written by the collide)
  // Lazy E (round 4).  A solve inside the time loop leaves phi's interior planes in the phi array and does NOT run
  // k_phi_efield: the next collide takes E = central differences of phi itself (EPHI kernels, same expression, same
  // bits), and the Ex / Ey / Ez arrays and the pinned plates of phi are brought up to date only when somebody looks
  // (ensure_efield: get_field, writers, diagnostics, ekpnp_synchronize ...).
  bool e_stale = false;        // the E arrays and phi's plates are older than phi's interior (implies e_phi_valid)
  bool e_phi_valid = false;    // E == central differences of the phi array (true after every solve, false once E or phi were set from outside)
  bool e_exposed = false;      // ekpnp_field_device_ptr handed out phi / E: the caller may read or write them at any time -> eager from then on
  int lazy_efield = 1;         // knob (EKPNP_LAZY_E, ekpnp_tune "lazy_efield"): 0 = k_phi_efield in every solve, as in rounds 1-3
  // opt-in knob (EKPNP_BATCH_MOMENTS, ekpnp_tune "batch_moments", default 0): inside ONE ekpnp_step(ctx, n) call only the LAST
  // step's bulk launch stores the seven moment arrays (the steps before it cannot be looked at); 56 of the 1 808 B/node
  bool batch_moments = false;
  bool skip_moments = false;   // the step being enqueued is such an intermediate step (set and cleared by the step loops)
  bool mom_exposed = false;    // ekpnp_field_device_ptr handed out one of rho, u, c, cn, T: every step stores them from then on
  bool streamed_state = true;  // true: pop[cur] holds X1 (post-stream, e.g. fresh equilibrium);
                               // false: pop[cur] holds post-collision populations (pull next)
  double* fld[EKPNP_NFIELDS] = {};
  bool fld_owned[EKPNP_NFIELDS] = {};
  void* fld_alloc[EKPNP_NFIELDS] = {};  // what hipMalloc returned for an owned field (fld[i] is skewed into it)
  size_t fld_bytes[EKPNP_NFIELDS] = {}; // size of that allocation (skew pad included)
  void* fld_arena = nullptr;            // EKPNP_FIELD_ARENA: the owned fields share this one allocation
  double* work = nullptr;
  double2* spec = nullptr;
  // the transforms run over the owned interior planes [fft_z0, fft_z0 + fft_nz): right-hand side in
  // work[], half spectrum in spec[], phi out into the phi array
  int fft_z0 = 0, fft_nz = 0;
  double* fft_in() const { return work + (size_t)fft_z0 * plane; }
  hipfftDoubleComplex* fft_spec() const { return (hipfftDoubleComplex*)(spec + (size_t)fft_z0 * p.ny * nxh); }
  double* fft_out() const { return fld[EKPNP_PHI] + (size_t)fft_z0 * plane; }
  double* cprime = nullptr;
  double* halo[4] = {};        // send-down, send-up, recv-from-below, recv-from-above
  size_t halo_doubles = 0;
  double* phi_halo[4] = {};
  double* stage[MAXL] = {};    // in-place slabs: first/last plane of the new state, 2 tiled planes
  // distributed tridiagonal (slab contexts)
  int slab_row_a = 0, slab_m = 0;  // first unknown row (local plane) and number of unknown rows
  double* slab_u = nullptr;        // u = A^-1 e_1, [slab_m][modes]
  double* slab_w = nullptr;        // forward elimination of e_1, [slab_m][modes]
  double* u1um = nullptr;          // (u_1, u_m) of EVERY rank's block, [nranks][2][modes] (slabs may differ by one plane)
  double* edge_local = nullptr;    // [4][modes]
  double* edge_all = nullptr;      // [nranks][4][modes]
  int edge_chunks = 1;             // mode blocks of the slab z solve (ekpnp_tune "edge_chunks" on a context with a transport, EKPNP_EDGE_CHUNKS):
                                   // > 1: the EDGE all-gather of block k runs on the comm stream beside the column pass of block k + 1
  int poisson_blocks = 0;          // single context (0: decided from the spectrum's size, poisson.hip: poisson_block_count): kx column blocks of the solve's middle passes (y forward, z solve, y inverse of one block back to
                                   // back, so that the block stays in the Infinity Cache between them; ekpnp_tune "poisson_blocks", EKPNP_POISSON_BLOCKS)
  int poisson_zchunk = 0;          // single context, own transforms: planes per chunk of the row + column passes (rows and columns of one chunk back
                                   // to back: the chunk is still in the Infinity Cache; 0 = whole passes; ekpnp_tune "poisson_zchunk", EKPNP_POISSON_ZCHUNK)
  double* phi_old = nullptr;       // PB relaxation state (ekpnp_pbe_begin/end)
  double* diag = nullptr;          // reduction scratch (DIAG_SCRATCH doubles)
  double* vwall = nullptr;         // {voltage, voltage, voltage2, voltage2}
  int collide_phase = 0;           // 0 idle, 1 boundary planes done
  // slab edge planes without pack / unpack copies (KArgs::halo_*): knob, and where the current halos are
  bool halo_direct = true;         // EKPNP_HALO_DIRECT=0: k_halo_pack / k_halo_unpack as in rounds 1-3 (the A/B partner)
  bool merged_faces = true;        // both faces of a slab in ONE launch (k_collide_faces); 0: a launch per face (EKPNP_MERGED_FACES, ekpnp_tune "merged_faces")
  int lead_planes = 2;             // planes of the short lead-in launch in front of the interior sweep (EKPNP_SLAB_LEAD_PLANES, ekpnp_tune "lead_planes"; 0: none)
  bool halo_sent = false;          // the boundary-plane launches of this step have filled the send buffers already
  bool halo_recv_valid = false;    // the receive buffers hold what the next pull of the edge planes needs (else: the ghost planes do)
  // hipGraph of two consecutive steps (A->B, B->A) for launch-bound lattices
  hipGraphExec_t graph2 = nullptr;
  int graph_cur = -1;              // value of `cur` the graph was captured at
  bool graph_failed = false;       // capture is not possible here: stay eager
  bool own_fft = false;            // the 2-D transforms are the library's own kernels (fft_plane.h), no rocFFT plans
  int ranks_on_device = 1;         // ranks of this lattice on this context's device, itself included (known once a communicator is attached)
  double2* fft_tw = nullptr;       // their twiddle table exp(-2 pi i k / 1024)
  hipfftHandle plan_fwd = 0, plan_inv = 0;
  bool tri_lds_ok = false;  // this context's device grants the partition z solves their dynamic LDS (tridiag_prepare_device)
  bool tri_wide = false;    // columns of more than 256 rows: 16 modes (wavefronts) per workgroup = 256-byte pieces of every row (EKPNP_TRI_WIDE, ekpnp_tune "tri_wide")
  int ncus = 0;             // compute units of the context's device (the grid of the pipelined z solves)
  int tri_partition = 1;  // z solve of a single context: 0 serial sweeps, 1 partition solve on large lattices, 2 wherever it applies
  bool have_fwd = false, have_inv = false;  // each handle is destroyed on its own (a failing second plan must not leak the first)
  bool plans = false;
  double t = 0.0;
  size_t bytes = 0;
  // first kernel launch that failed since the last check (note_launch / take_launch_error)
  hipError_t launch_err = hipSuccess;
  const char* launch_what = nullptr;
  long inject_seen = 0;  // launches of the kernel EKPNP_INJECT_LAUNCH_FAILURE names, on this context (tests)
  // kernel timing
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  size_t ev_used = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_poisson;  // the same around every Poisson solve (all its stages)
  size_t evp_used = 0;
  // slabs: four marks inside solve i (same index as ev_poisson): end of stage 1, start of stage 2 (the EDGE exchange lies between
  // the two), end of stage 2 + phi pack, start of stage 3 (the PHI exchange lies between) - ekpnp_poisson_stage_timing_get
  std::vector<std::array<hipEvent_t, 4>> ev_stage;
  long long timed_nodes = 0;
  std::string err;

  KArgs kargs() const;
  PArgs pargs() const;
};

// capi.hip: may an intermediate step of a batch leave the moment arrays alone (knob on, all seven arrays the library's own, unexposed)?
bool batch_moments_ok(const Ctx& c);
// capi.hip: may this context leave E in phi (every one of phi, Ex, Ey, Ez is the library's own, unexposed array)?
bool lazy_efield_ok(const Ctx& c);
// capi.hip: bring the E arrays and phi's plates up to date if a lazy solve left them behind (no-op otherwise)
int ensure_efield(Ctx& c);
// capi.hip: fill a slab's ghost planes from the receive buffers if that is where the current halos are (checkpoint writers)
int ensure_ghost_planes(Ctx& c);
// capi.hip: phi or E are about to be overwritten from outside (set_field, readers): up to date first, then E is what the arrays say
int efield_set_from_outside(Ctx& c);

// Called after EVERY kernel launch of the library: a launch that the runtime rejects (bad grid,
// missing code object ...) is recorded with the kernel's name instead of surfacing, nameless, at the
// end of the entry point.  With EKPNP_DEBUG_SYNC set in the environment the stream is also
// synchronised after each launch, so that a fault inside a kernel is reported against that kernel
// (debug runs only: it serialises host and device).
void note_launch(Ctx& c, const char* kernel);
// The status the entry points return after their launches: the recorded launch error (its message
// names the kernel) or whatever hipGetLastError() holds.  Clears both.
hipError_t take_launch_error(Ctx& c);
// the message of a failed call that has no context to hold it (what ekpnp_last_error(NULL) returns)
void set_create_error(const std::string& msg);

// slab_team.hip: the reference's verbs on a slab context whose team moves the halos itself
// (ekpnp_slab_attach_comm).  Each returns EKPNP_ERR_INVALID with a message if the context's team
// is an in-process group (those are driven through ekpnp_group_*).
int team_ctx_stream_collide_save(Ctx&);
int team_ctx_fast_poisson(Ctx&);
int team_ctx_step(Ctx&, int nsteps);
int team_ctx_initialization(Ctx&);
int team_ctx_initialization_converged(Ctx&, double rel_tol, int max_sweeps, int* sweeps, double* residual);
int team_ctx_reduce(Ctx&, double* value, bool is_max);  // combine a per-slab diagnostic over the ranks
int team_ctx_turns(Ctx&, int (*fn)(Ctx&, void*), void* arg);  // fn on every slab in rank order (file IO)
int team_ctx_tune(Ctx&, const char* knob, int value);  // the transport's knobs (ekpnp_tune on an attached slab): EKPNP_OK, or EKPNP_ERR_INVALID with a message for an unknown knob / bad value
// capi.hip: the slab solve in pieces, driven by slab_team.hip around the exchanges (the exported ekpnp_poisson_stage1/2/3 are built from them)
int poisson_stage1_begin(Ctx&);          // right-hand side + row pass (rocFFT plans: the whole forward transform)
int poisson_stage1_block(Ctx&, int k);   // column pass of mode block k + its edge values -> edge_local
int poisson_stage1_end(Ctx&);            // (measurement mark)
int poisson_stage2_block(Ctx&, int k);   // interface system + z solve of mode block k (its edge values gathered) + inverse column pass
int poisson_stage2_end(Ctx&);            // inverse row pass (rocFFT plans: the whole inverse transform) + phi halo pack
int ctx_tune(Ctx&, const char* knob, int value);  // ekpnp_tune's per-context knobs
int make_fft_plans(Ctx&);                          // capi.hip: the rocFFT plans of the plane transforms (idempotent)
void team_detach(Ctx&);  // called by ekpnp_destroy
bool team_is_group(const Ctx&);  // the context is a member of an in-process ekpnp_group
void team_timing_reset(Ctx&);    // ekpnp_kernel_timing_enable: forget the exchanges bracketed so far
int team_comm_timing_get(Ctx&, int kind, int* n, double* wait_ms, double* transfer_ms, size_t* bytes_sent);

// io.hip pieces shared with slab_team.hip (a slab writes / reads its own planes of a whole-lattice file)
struct TextIoArgs { const char* path; int append; double time; int first; int kind; };  // kind 0 Tecplot, 1 data_end
int io_write_text_part(Ctx&, const TextIoArgs&);
int io_read_data_part(Ctx&, const char* path, double* time);

// lossless state file "EKPNPST1" (fields only; io.hip and the group variant in slab_team.hip)
struct StateHeader {
  char magic[8];
  int32_t nx, ny, nz, z0, nzl, nfields;
  double time;
};
static_assert(sizeof(StateHeader) == 40, "state header layout");
constexpr size_t STATE_CHUNK = (size_t)4 << 20;  // doubles per bounce-buffer transfer (32 MiB)

// full checkpoint (io.hip), pieces shared with the group variant in slab_team.hip
struct CkptHeader {
  char magic[8];
  int32_t nx, ny, nz, z0, nzl, nfields, nl, streamed_state, with_ghosts, pad_[3];
  double time;
};
int io_ckpt_write_header(Ctx&, FILE*, int z0, int nzl, int with_ghosts, double time);
int io_ckpt_check_header(Ctx&, const CkptHeader&, int z0, int nzl);
int io_ckpt_fields(Ctx&, FILE*, int field, int dir);              // dir 0: device -> file, 1: file -> device
int io_ckpt_populations(Ctx&, FILE*, int lattice, int with_ghosts, int dir);
void io_ckpt_finish_load(Ctx&, const CkptHeader&);

}  // namespace ekpnp

// the opaque handle of include/ekpnp.h
struct ekpnp_ctx {
  ekpnp::Ctx c;
};
