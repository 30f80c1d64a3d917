// poisson.hip — the spectral Poisson half of the hot path on gfx950.
//
// Replaces fast_Poisson (poisson.cu:75-103): odd_extension (114-158) -> cuFFT Z2Z 3-D forward
// (poisson.cu:86, plan main.cu:112) -> gpu_derivative (169-180) -> Z2Z inverse (poisson.cu:92)
// -> odd_extract (191-204) -> gpu_efield / gpu_bc (40-69).
//
// The reference's 3-D complex FFT of the odd extension of length NE = 2(NZ-1) is a DST-I in z,
// i.e. the exact diagonalisation of the second-order finite-difference z operator with
// Dirichlet walls (its eigenvalue is the (4/dz^2) sin^2(kz dz/2) of poisson.cu:176).  The same
// linear system is solved here as: real 2-D FFT in x,y per plane (rocFFT through hipFFT, D2Z,
// batched over planes) + one constant-coefficient tridiagonal solve in z per (kx,ky) mode
// (Thomas algorithm, one thread per mode, coalesced over kx) + inverse 2-D FFT.  Mode
// (kx,ky)=(0,0) is a regular Dirichlet problem here, so the reference's DC-mode leak
// (poisson.cu:177, mu := 1) cannot occur: this is the canonical "DC = 0" result.
// It moves 4x fewer bytes than the 2N-point complex transform and has no constraint on NZ.
#include <cstdlib>

#include "ekpnp_internal.h"
#include "fft_plane.h"

namespace ekpnp {

// ---- the 2-D transforms of the interior planes ---------------------------------------------------------------------
// The library's own row and column passes (fft_plane.h) on planes of 512 / 1024 x 512 / 1024 nodes - cfg3, cfg4, cfg5 -,
// rocFFT plans (hipFFT API) everywhere else.  On 1024-long columns rocFFT has no strided-column kernel and transposes
// (4 + 4 kernels, 3.2 ms per solve on a 1024 x 1024 x 128 slab against 0.84 + 0.81 ms); on 512-long ones its 2 + 2 kernels
// are nearly as fast (1.66 against 1.60 ms per solve on cfg3).  EKPNP_OWN_FFT=0 at creation keeps rocFFT (the A/B partner).
int plane_fft_setup(Ctx& c) {
  c.own_fft = false;
  const char* e = std::getenv("EKPNP_OWN_FFT");
  // default: the own passes wherever they exist (rows of 512 or 1024 nodes, columns of 512 or 1024); EKPNP_OWN_FFT=0: never.
  // Rounds 2-3 kept rocFFT on 512-long columns (kernel against kernel the own passes gain only 0.06 ms per solve there);
  // inside the full cfg3 step the Poisson phase is 2.12 ms with them against 2.25 ms with rocFFT's plans - the z solve
  // between the passes runs faster too (490 vs 514 us) - and the plans' work areas go (profiles/r04_ab_own_fft_512.log).
  const int want = e ? std::atoi(e) : -1;
  if (want == 0) return EKPNP_OK;
  if (!fft_x_supported(c.p.nx) || !fft_y_supported(c.p.ny, c.nxh)) return EKPNP_OK;
  if (!fft_x_prepare(c.p.nx) || !fft_y_prepare(c.p.ny)) return EKPNP_OK;  // this device does not grant the LDS: rocFFT
  // one table serves both passes when the plane is square: exp(-2 pi i k / NX), then exp(-2 pi i k / NY)
  std::vector<double2> h((size_t)c.p.nx + c.p.ny);
  fft_y_twiddles(c.p.nx, h.data());
  fft_y_twiddles(c.p.ny, h.data() + c.p.nx);
  hipError_t he = hipMalloc((void**)&c.fft_tw, h.size() * sizeof(double2));
  if (he == hipSuccess) he = hipMemcpy(c.fft_tw, h.data(), h.size() * sizeof(double2), hipMemcpyHostToDevice);
  if (he != hipSuccess) {
    c.err = std::string("twiddle table of the plane transforms: ") + hipGetErrorString(he);
    return he == hipErrorOutOfMemory ? EKPNP_ERR_NOMEM : EKPNP_ERR_HIP;
  }
  c.bytes += h.size() * sizeof(double2);
  c.own_fft = true;
  return EKPNP_OK;
}

// The transforms in pieces, for the slab solve with mode blocks (own passes only; with rocFFT plans the 2-D transform is one
// call, issued with the first / last piece): forward = rows of all planes, then the columns of block k; inverse = the columns
// of block k, then the rows of all planes.
// (z0, nz: a run of the transform's planes, for the plane chunks of a single context's solve - "poisson_zchunk"; nz < 0: all)
int plane_fft_forward_rows(Ctx& c, int z0, int nz) {
  if (!c.own_fft) return plane_fft_forward(c);
  if (nz < 0) { z0 = 0; nz = c.fft_nz; }
  fft_x_forward(c.fft_in() + (size_t)z0 * c.plane, reinterpret_cast<double2*>(c.fft_spec()) + (size_t)z0 * c.p.ny * c.nxh, c.fft_tw, c.p.nx, c.nxh,
                (long long)c.p.ny * nz, c.stream);
  note_launch(c, "k_fft_x_r2c");
  return EKPNP_OK;
}
void plane_fft_forward_columns(Ctx& c, const ModeBlock& b, int z0, int nz) {
  if (!c.own_fft) return;
  if (nz < 0) { z0 = 0; nz = c.fft_nz; }
  fft_y_launch(reinterpret_cast<double2*>(c.fft_spec()) + (size_t)z0 * c.p.ny * c.nxh, c.fft_tw + c.p.nx, c.p.ny, c.nxh, nz, -1, c.stream, b.x0 / 8, b.bw / 8);
  note_launch(c, "k_fft_y<-1>");
}
void plane_fft_inverse_columns(Ctx& c, const ModeBlock& b, int z0, int nz) {
  if (!c.own_fft) return;
  if (nz < 0) { z0 = 0; nz = c.fft_nz; }
  fft_y_launch(reinterpret_cast<double2*>(c.fft_spec()) + (size_t)z0 * c.p.ny * c.nxh, c.fft_tw + c.p.nx, c.p.ny, c.nxh, nz, +1, c.stream, b.x0 / 8, b.bw / 8);
  note_launch(c, "k_fft_y<1>");
}
int plane_fft_inverse_rows(Ctx& c, int z0, int nz) {
  if (!c.own_fft) return plane_fft_inverse(c);
  if (nz < 0) { z0 = 0; nz = c.fft_nz; }
  fft_x_inverse(reinterpret_cast<const double2*>(c.fft_spec()) + (size_t)z0 * c.p.ny * c.nxh, c.fft_out() + (size_t)z0 * c.plane, c.fft_tw, c.p.nx, c.nxh,
                (long long)c.p.ny * nz, c.stream);
  note_launch(c, "k_fft_x_c2r");
  return EKPNP_OK;
}

int plane_fft_forward(Ctx& c) {
  if (c.own_fft) {
    fft_x_forward(c.fft_in(), reinterpret_cast<double2*>(c.fft_spec()), c.fft_tw, c.p.nx, c.nxh, (long long)c.p.ny * c.fft_nz, c.stream);
    note_launch(c, "k_fft_x_r2c");
    fft_y_launch(reinterpret_cast<double2*>(c.fft_spec()), c.fft_tw + c.p.nx, c.p.ny, c.nxh, c.fft_nz, -1, c.stream);
    note_launch(c, "k_fft_y<-1>");
    return EKPNP_OK;
  }
  const hipfftResult r = hipfftExecD2Z(c.plan_fwd, c.fft_in(), c.fft_spec());
  if (r != HIPFFT_SUCCESS) { c.err = "hipfftExecD2Z: hipfft error " + std::to_string((int)r); return EKPNP_ERR_FFT; }
  return EKPNP_OK;
}

int plane_fft_inverse(Ctx& c) {
  if (c.own_fft) {
    fft_y_launch(reinterpret_cast<double2*>(c.fft_spec()), c.fft_tw + c.p.nx, c.p.ny, c.nxh, c.fft_nz, +1, c.stream);
    note_launch(c, "k_fft_y<1>");
    fft_x_inverse(reinterpret_cast<const double2*>(c.fft_spec()), c.fft_out(), c.fft_tw, c.p.nx, c.nxh, (long long)c.p.ny * c.fft_nz, c.stream);
    note_launch(c, "k_fft_x_c2r");
    return EKPNP_OK;
  }
  const hipfftResult r = hipfftExecZ2D(c.plan_inv, c.fft_spec(), c.fft_out());
  if (r != HIPFFT_SUCCESS) { c.err = "hipfftExecZ2D: hipfft error " + std::to_string((int)r); return EKPNP_ERR_FFT; }
  return EKPNP_OK;
}

// rhs of the interior planes, poisson.cu:114-135: g = -F (c - cn)/eps, wall potentials folded
// into the planes next to the walls; wall planes themselves carry 0.
__global__ void k_poisson_rhs(PArgs a) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n = (long long)a.nzl * a.plane;
  if (i >= n) return;
  const int z = a.z0 + (int)(i / a.plane);
  double v = 0.0;
  if (z > 0 && z < a.nz - 1) v = poisson_rhs_value(a.F, a.eps, a.fld[EKPNP_C][i], a.fld[EKPNP_CN][i], z, a.nz, a.rhs_wall_lo, a.rhs_wall_hi);
  a.work[i] = v;
}

// A mode BLOCK of a slab's z solve: the (kx, ky) modes with bx0 <= kx < bx0 + bw, numbered j = ky bw + (kx - bx0), so that the
// edge values of a block are one contiguous piece [4][ny bw] of the exchange buffers and the all-gather of one block can run
// while the next block is still being transformed (slab_team.hip: "edge_chunks").  The whole half spectrum is the block
// bx0 = 0, bw = nxh, where j is the mode index itself.
__device__ __forceinline__ long long block_mode(const PArgs& a, long long j) {
  if (a.bw == a.nxh) return j;
  const long long ky = j / a.bw;
  return ky * a.nxh + a.bx0 + (j - ky * a.bw);
}

// Diagonal of the z system of one (kx,ky) mode:  phi[z-1] + b phi[z] + phi[z+1] = dz^2 g[z],
// b = -(2 + dz^2 (kx^2 + ky^2)), wavenumbers exactly as main.cu:119-136.  (Pad columns of the half
// spectrum get some b <= -2 too: their right-hand side is 0 and stays 0.)
__device__ __forceinline__ double mode_diag(int m, int ny, int nxh, double Lx, double Ly, double dz) {
  const int ix = m % nxh, iy = m / nxh;
  const double kx = (double)ix * 2.0 * M_PI / Lx;
  const double ky = (iy <= ny / 2) ? (double)iy * 2.0 * M_PI / Ly : ((double)iy - ny) * 2.0 * M_PI / Ly;
  return -(2.0 + dz * dz * (kx * kx + ky * ky));
}

// Thomas factorisation of that system for the rows z = 1 .. nz-2:  c'[0] = 0, c'[z] = 1/(b - c'[z-1]).
// Stored: every `every`-th row, table row k = c'[k * every], layout [rows][ny][nxh].  every = 1 is the
// full table of the slab kernels; the single-context solve keeps only the rows it restarts the
// recurrence from (every = TRI_BS) and recomputes the others - same operations, same bits, and the
// table is neither read in the forward sweep nor read twice (0.55 GB per read on 512^3).
__global__ void k_build_cprime(double* cprime, int nx, int ny, int nz, int nxh, double Lx, double Ly, double dz, int every) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= ny * nxh) return;
  const double b = mode_diag(m, ny, nxh, Lx, Ly, dz);
  const long long ms = (long long)ny * nxh;
  double cp = 0.0;
  cprime[m] = 0.0;
  for (int z = 1; z <= nz - 2; ++z) {
    cp = 1.0 / (b - cp);
    if (z % every == 0) cprime[(long long)(z / every) * ms + m] = cp;
  }
  if (every == 1) cprime[(long long)(nz - 1) * ms + m] = 0.0;
}

// Forward elimination + back substitution for one (kx,ky) mode, in place on the spectrum.
// Single-slab version (the whole z extent is local).
//
// The eliminated right-hand side d' is needed in reverse order by the back substitution.  Instead
// of storing all of it (a second write + read of the whole spectrum) the forward sweep keeps only
// every TRI_BS-th row (written over the spectrum row it belongs to) and the back substitution
// recomputes the rows in between, block by block, from the untouched right-hand side: the same
// operations in the same order, hence the same bits, for 4.4 instead of 5.5 GB on 512^3.  The
// factors c' are treated the same way: the forward sweep runs their recurrence in registers (one
// division per row, hidden behind the loads), the back substitution restarts it from the table row
// below each block: 3.4 GB, i.e. the spectrum read twice and written once and nothing else.
constexpr int TRI_BS = TRI_CHECK;
// The solution rows are not read again before the inverse transform has gone through all of them:
// non-temporal stores (0.744 -> 0.723 ms on 512^3; non-temporal LOADS in the forward sweep change
// nothing: profiles/r02_tridiag_variants.log).  EKPNP_TRI_PLAIN_STORE builds the A/B partner.
#ifndef EKPNP_TRI_PLAIN_STORE
#define TRI_STORE(ptr, v) do { double2 v_ = (v); __builtin_nontemporal_store(v_.x, &(ptr)->x); __builtin_nontemporal_store(v_.y, &(ptr)->y); } while (0)
#else
#define TRI_STORE(ptr, v) (*(ptr) = (v))
#endif
#define TRI_LOAD_FWD(ptr) (*(ptr))
#ifndef EKPNP_TRI_THREADS
#define EKPNP_TRI_THREADS 64  // tuning knob: modes (threads) per workgroup of the z solve
#endif
__global__ void __launch_bounds__(EKPNP_TRI_THREADS) k_tridiag(PArgs a) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  const long long ms = (long long)a.ny * a.nxh;
  if (m >= ms) return;
  const double dz2 = a.dz * a.dz;
  const double b = mode_diag(m, a.ny, a.nxh, a.Lx, a.Ly, a.dz);
  double2* s = a.spec + m;
  const double* ck = a.cprime + m;  // row k = c'[k * TRI_BS]
  const int n = a.nz;
  {
    double dr = 0.0, di = 0.0, c = 0.0;
#pragma unroll 8
    for (int z = 1; z <= n - 2; ++z) {
      const double2 r = TRI_LOAD_FWD(s + (long long)z * ms);
      c = 1.0 / (b - c);
      dr = (dz2 * r.x - dr) * c;
      di = (dz2 * r.y - di) * c;
      if ((z & (TRI_BS - 1)) == 0) s[(long long)z * ms] = make_double2(dr, di);  // checkpoint d'[z]
    }
  }
  // phi[n-2] = d'[n-2]; phi[z] = d'[z] - c'[z] phi[z+1]
  // (requesting the next block's rows before the current block's chain of 16 divisions - software
  // pipelining at 196 VGPRs - was measured slower: 0.82 vs 0.75 ms on 512^3, profiles/r02_tridiag_variants.log)
  double pr = 0.0, pi = 0.0;
  for (int zhi = n - 2; zhi >= 1;) {
    const int zlo = ((zhi - 1) / TRI_BS) * TRI_BS + 1;  // block zlo..zhi; row zlo-1 is a checkpoint (or the wall)
    double2 d[TRI_BS];
    double c[TRI_BS];
    double2 prev = make_double2(0.0, 0.0);
    double cprev = 0.0;
    if (zlo > 1) {
      prev = s[(long long)(zlo - 1) * ms];
      cprev = ck[(long long)((zlo - 1) / TRI_BS) * ms];
    }
#pragma unroll
    for (int i = 0; i < TRI_BS; ++i) {
      const int z = zlo + i;
      if (z <= zhi) d[i] = s[(long long)z * ms];
    }
#pragma unroll
    for (int i = 0; i < TRI_BS; ++i) {
      const int z = zlo + i;
      if (z <= zhi) {
        cprev = 1.0 / (b - cprev);
        c[i] = cprev;
        if (i != TRI_BS - 1) {  // not a checkpoint row: d[i] still holds the right-hand side
          d[i].x = (dz2 * d[i].x - prev.x) * c[i];
          d[i].y = (dz2 * d[i].y - prev.y) * c[i];
        }
        prev = d[i];
      }
    }
#pragma unroll
    for (int i = TRI_BS - 1; i >= 0; --i) {
      const int z = zlo + i;
      if (z <= zhi) {
        if (z == n - 2) {
          pr = d[i].x;
          pi = d[i].y;
        } else {
          pr = d[i].x - c[i] * pr;
          pi = d[i].y - c[i] * pi;
        }
        // stored with the 1/(NX NY) of the unnormalised transforms folded in (poisson.cu:196: / size):
        // the inverse FFT then delivers phi itself, straight into the phi array
        TRI_STORE(s + (long long)z * ms, make_double2(pr * a.inv_nxny, pi * a.inv_nxny));
      }
    }
    zhi = zlo - 1;
  }
}

// Channels of 66 < NZ <= 64 R + 2 planes on large lattices: the z solve with the spectrum read ONCE.
// k_tridiag above reads every row twice (3.4 GB on 512^3) because one thread cannot hold a column; here a
// whole wavefront holds it: lane l keeps R consecutive rows (slots R l .. R l + R - 1; row = slot + 1) of
// one (kx,ky) mode in registers, and the system  x[i-1] + b x[i] + x[i+1] = dz^2 r[i]  is solved by the
// partition method:
//   * every lane eliminates its R-1 interior rows (a Thomas sweep of R-1 rows, three right-hand sides:
//     the data g and the unit couplings v, w to the interface unknowns y[l-1], y[l] = the lane's last row);
//   * the interface rows form a tridiagonal system of 64 unknowns across the lanes, solved by parallel
//     cyclic reduction in 6 shuffle steps (as k_tridiag_pcr64);
//   * x = g - y[l-1] v - y[l] w.
// Rows beyond NZ-2 are identity rows.  A workgroup takes 8 adjacent modes (one 128-byte line per row): the
// rows come in coalesced (thread (c, t): rows t + 64 r of column c), change owner through a 64 R x 8 LDS
// image (column index XOR-swizzled by the owning lane: the solve-phase reads are 2-way, the rest
// conflict-free) and leave the same way.  Same system, different elimination order than k_tridiag:
// results agree to rounding (tests/test_parity_gpu.py::test_partition_z_solve...).
// 1/x for the pivots of the partition solve: hardware reciprocal + two Newton steps (full double precision to an
// ulp, not correctly rounded; the IEEE division sequence with its scaling and fix-up costs twice the instructions).
// Range: the pivots are NORMAL, finite and of one sign.  The local Thomas pivots are |b - c'| >= 1 (b <= -2, 0 >= c' > -1).
// The interface (Schur) pivots of a low mode b -> -2 start near 2/R and roughly halve at each of the 6 reduction
// levels (~4e-3 for R = 8), those of a high mode (b = -(2 + dz^2 k^2), up to ~ -1e5 for dz >> dx) stay near |b|; both
// are far inside the range where v_rcp_f64 needs no scaling (|x| in 2^-1022 .. 2^1022 would do).
// tests/test_parity_gpu.py::test_partition_z_solve_extreme_anisotropy compares it with the serial IEEE-division
// sweeps at dz/dx = 1e-3 and 1e3.
__device__ __forceinline__ double recip(double x) {
#ifdef EKPNP_TRI_IEEE_DIV
  return 1.0 / x;
#else
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
#endif
}
// The body serves the single context (rows = planes 1 .. NZ-2, zero beyond both ends) and, since round 3, a z slab
// (rows = the slab's own unknown rows; x just below the first and just above the last row are the interface values
// `bound` = (g_lo.re, g_lo.im, g_hi.re, g_hi.im)[modes] that k_slab_interface made from the gathered edge values:
// they move to the right-hand side of the first and the last row).
// LANES (round 3): lanes of a wavefront that share one mode.  64 = one wavefront per mode (8 modes per workgroup, columns
// of up to 64 R rows).  Short columns - cfg5's and cfg4@8's 128-plane slabs - would leave R = 2 rows per lane, and the
// cyclic reduction (6 levels of lane exchanges per 2 KB of column) becomes the whole cost: 0.77 ms for the bytes the
// R = 8 kernel moves in 0.56.  With LANES = 16 (32) a wavefront holds 4 (2) modes of 16 R (32 R) rows, the reduction has
// 4 (5) levels and serves all of them at once, and a workgroup covers 32 (16) adjacent modes: 512 (256) contiguous
// bytes per row.
// ---- the pieces of the partition solve (shared by the one-shot kernels and the pipelined ones below) ---------------------
template <int R, int LANES, int NW = 8>
struct TriPart {
  static constexpr int MPW = 64 / LANES;        // modes per wavefront
  static constexpr int MC = NW * MPW;           // modes (LDS columns) per workgroup of NW wavefronts (8; 16: twice as wide pieces of every row)
  static_assert(LANES * R <= 512 && 64 % LANES == 0, "a column has LANES x R row slots");
  static constexpr int TR = 64 * NW / MC;       // rows loaded per pass of the workgroup
  static constexpr int FM = (MC < 16 ? MC : 16) - 1;  // the column index is XOR-ed with the owning lane (mod 16 columns = 256 bytes of banks)
  static constexpr int IMAGE = LANES * R * MC;  // double2 elements of one LDS image [LANES R slots][MC columns] = R x 8 KB

  // the workgroup's rows of mode group m0 .. m0 + MC - 1, coalesced (thread (c, t): rows t + TR r of column c), into registers
  // (mcol: the mode of THIS thread's column c = tid % MC - m0 + c, or its image under block_mode() for a mode block of a slab)
  static __device__ __forceinline__ void load(const double2* __restrict__ rows, long long ms, long long mcol, int n, double2 (&v)[R], const int tid = threadIdx.x) {
    const int t = tid / MC;
    const double2* src = rows + mcol;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int s = t + TR * r;
      v[r] = make_double2(0.0, 0.0);
      if (s < n) v[r] = src[(long long)s * ms];
    }
  }
  // registers -> LDS image (the rows change owner there)
  static __device__ __forceinline__ void put(double2* __restrict__ img, const double2 (&v)[R], const int tid = threadIdx.x) {
    const int c = tid % MC, t = tid / MC;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int s = t + TR * r;
      img[s * MC + (c ^ ((s / R) & FM))] = v[r];
    }
  }
  // LDS image -> global, the way the rows came
  static __device__ __forceinline__ void store(double2* __restrict__ rows, long long ms, long long mcol, int n, const double2* __restrict__ img, const int tid = threadIdx.x) {
    const int c = tid % MC, t = tid / MC;
    double2* dst = rows + mcol;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int s = t + TR * r;
      if (s < n) TRI_STORE(dst + (long long)s * ms, img[s * MC + (c ^ ((s / R) & FM))]);
    }
  }
};

// the solve proper, on an LDS image: every lane takes its R rows out of the image, eliminates, takes part in the cyclic
// reduction of the interface rows and puts its R solution rows back
template <int R, int LANES, bool SLAB, int NW = 8>
__device__ __forceinline__ void tridiag_part_solve(const PArgs& a, double2* __restrict__ img, const long long m0, const int n, const double* __restrict__ bound) {
  using TP = TriPart<R, LANES, NW>;
  constexpr int MPW = TP::MPW, MC = TP::MC, FM = TP::FM;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l = lane % LANES;                 // lane within its mode: owns slots R l .. R l + R - 1
  const int col = w * MPW + lane / LANES;     // the mode among the workgroup's MC
  const double b = mode_diag((int)block_mode(a, m0 + col), a.ny, a.nxh, a.Lx, a.Ly, a.dz);  // (the whole spectrum is the block 0, nxh)
  const double dz2 = a.dz * a.dz;
  double2* mine = img + (R * l) * MC + (col ^ (l & FM));
  double2 g[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const double2 r = mine[k * MC];
    g[k] = make_double2(dz2 * r.x, dz2 * r.y);
  }
  if (SLAB) {
    const long long msl = (long long)a.ny * a.bw;  // the boundary values are stored by mode BLOCK: [4][ny bw], local numbering
    const double* bw = bound + m0 + col;
    const double glr = bw[0], gli = bw[msl], ghr = bw[2 * msl], ghi = bw[3 * msl];
    if (l == 0) g[0] = make_double2(g[0].x - glr, g[0].y - gli);
#pragma unroll
    for (int k = 0; k < R; ++k)
      if (R * l + k == n - 1) g[k] = make_double2(g[k].x - ghr, g[k].y - ghi);
  }
  // interior rows k = 0 .. R-2 of this lane
  double cp[R - 1], v[R - 1], wv[R - 1];
  {
    double cprev = 0.0, vprev = 0.0;
    double2 gprev = make_double2(0.0, 0.0);
#pragma unroll
    for (int k = 0; k < R - 1; ++k) {
      const int s = R * l + k;
      const bool real = s < n;
      const double Ain = (real && k > 0) ? 1.0 : 0.0;               // coupling to the row before, inside the block
      const double A0 = (real && k == 0 && s > 0) ? 1.0 : 0.0;      // coupling of the first row to y[l-1]
      const double Bk = real ? b : 1.0;
      const double Cany = (real && s < n - 1) ? 1.0 : 0.0;          // coupling to the row after
      const double inv = recip(Bk - Ain * cprev);
      cp[k] = (k < R - 2 ? Cany : 0.0) * inv;
      g[k] = make_double2((g[k].x - Ain * gprev.x) * inv, (g[k].y - Ain * gprev.y) * inv);
      v[k] = (A0 - Ain * vprev) * inv;
      wv[k] = (k == R - 2 ? Cany : 0.0) * inv;                      // rows before R-2 have no w right-hand side
      cprev = cp[k];
      vprev = v[k];
      gprev = g[k];
    }
#pragma unroll
    for (int k = R - 3; k >= 0; --k) {
      g[k] = make_double2(g[k].x - cp[k] * g[k + 1].x, g[k].y - cp[k] * g[k + 1].y);
      v[k] = v[k] - cp[k] * v[k + 1];
      wv[k] = -cp[k] * wv[k + 1];
    }
  }
  // the interface row (slot R l + R - 1) in terms of y[l-1], y[l], y[l+1]; lane exchanges stay inside the LANES lanes of a mode
  double lo, up, bd, rr, ri;
  {
    const int s = R * l + R - 1;
    const bool real = s < n;
    const double A7 = real ? 1.0 : 0.0, B7 = real ? b : 1.0, C7 = (real && s < n - 1) ? 1.0 : 0.0;
    const double g0x = __shfl_down(g[0].x, 1, LANES), g0y = __shfl_down(g[0].y, 1, LANES);
    const double v0n = __shfl_down(v[0], 1, LANES), w0n = __shfl_down(wv[0], 1, LANES);
    lo = -A7 * v[R - 2];
    bd = B7 - A7 * wv[R - 2] - C7 * v0n;
    up = -C7 * w0n;
    rr = g[R - 1].x - A7 * g[R - 2].x - C7 * g0x;
    ri = g[R - 1].y - A7 * g[R - 2].y - C7 * g0y;
    if (l == LANES - 1) up = 0.0;
    if (l == 0) lo = 0.0;
  }
#pragma unroll
  for (int st = 1; st < LANES; st <<= 1) {
    double lo_l = __shfl_up(lo, st, LANES), up_l = __shfl_up(up, st, LANES), bd_l = __shfl_up(bd, st, LANES);
    double rr_l = __shfl_up(rr, st, LANES), ri_l = __shfl_up(ri, st, LANES);
    double lo_u = __shfl_down(lo, st, LANES), up_u = __shfl_down(up, st, LANES), bd_u = __shfl_down(bd, st, LANES);
    double rr_u = __shfl_down(rr, st, LANES), ri_u = __shfl_down(ri, st, LANES);
    // a lane without a partner at this distance has lo (up) == 0 - an invariant of the reduction, true of the start
    // values - so whatever finite values its shuffle returned (its own) are multiplied by zero: no selects needed
    const double al = -lo * recip(bd_l), ga = -up * recip(bd_u);
    bd = bd + al * up_l + ga * lo_u;
    rr = rr + al * rr_l + ga * rr_u;
    ri = ri + al * ri_l + ga * ri_u;
    lo = al * lo_l;
    up = ga * up_u;
  }
  const double ibd = recip(bd);
  const double yx = rr * ibd, yy = ri * ibd;
  double ylx = __shfl_up(yx, 1, LANES), yly = __shfl_up(yy, 1, LANES);
  if (l == 0) { ylx = 0.0; yly = 0.0; }
  // 1/(NX NY) of the unnormalised transforms folded in, as in k_tridiag
#pragma unroll
  for (int k = 0; k < R - 1; ++k)
    mine[k * MC] = make_double2((g[k].x - ylx * v[k] - yx * wv[k]) * a.inv_nxny, (g[k].y - yly * v[k] - yy * wv[k]) * a.inv_nxny);
  mine[(R - 1) * MC] = make_double2(yx * a.inv_nxny, yy * a.inv_nxny);
}

template <int R, int LANES, bool SLAB, int NW = 8>
__device__ __forceinline__ void tridiag_part_body(const PArgs& a, double2* __restrict__ rows, const int n, const double* __restrict__ bound) {
  using TP = TriPart<R, LANES, NW>;
  extern __shared__ double2 tp_lds[];    // [LANES R slots][MC columns] = R x NW KB
  const long long ms = (long long)a.ny * a.nxh;
  const long long m0 = (long long)blockIdx.x * TP::MC;  // numbered within the kernel's mode block (block_mode)
  const long long mcol = block_mode(a, m0 + threadIdx.x % TP::MC);
  {
    double2 v[R];
    TP::load(rows, ms, mcol, n, v);
    TP::put(tp_lds, v);
  }
  __syncthreads();
  tridiag_part_solve<R, LANES, SLAB, NW>(a, tp_lds, m0, n, bound);
  __syncthreads();
  TP::store(rows, ms, mcol, n, tp_lds);
}

// What bounds this kernel, with counters, and the five restructurings that were built, measured and removed again in round 4
// (pipelined with register prefetch, wave-specialised, early-exit reduction, chain-free pivots; 16 modes per workgroup stays
// as the tri_wide knob): DESIGN.md section 4 - the access pattern (510 pieces of 128 bytes, a plane apart, per workgroup),
// 4.4 TB/s, which a plain copy of that shape does not beat either.
template <int R, int LANES = 64, int NW = 8>
__global__ void __launch_bounds__(64 * NW) k_tridiag_part(PArgs a) {
  tridiag_part_body<R, LANES, false, NW>(a, a.spec + (long long)a.ny * a.nxh, a.nz - 2, nullptr);
}

// z slab: the same solve on the slab's m unknown rows (first one on local plane row_a), spectrum read once
template <int R, int LANES = 64, int NW = 8>
__global__ void __launch_bounds__(64 * NW) k_slab_part(PArgs a, int row_a, int m, const double* __restrict__ bound) {
  tridiag_part_body<R, LANES, true, NW>(a, a.spec + (long long)row_a * a.ny * a.nxh, m, bound);
}

// Short channels (NZ - 2 <= 64 unknown rows, e.g. the reference's own 51 planes): the serial
// sweeps above are a chain of ~2 NZ dependent loads per mode and dominate a step that is only
// tens of microseconds long.  Here one wave64 owns one (kx,ky) mode, lane i holds row i + 1, and
// the constant-coefficient system  x[i-1] + b x[i] + x[i+1] = r[i]  is solved by parallel cyclic
// reduction in 6 shuffle steps (rows outside 1..NZ-2 act as identity rows).
__global__ void __launch_bounds__(256) k_tridiag_pcr64(PArgs a) {
  const int lane = threadIdx.x & 63;
  const int md = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long ms = (long long)a.ny * a.nxh;
  if (md >= ms) return;  // wave-uniform
  const int m = a.nz - 2;  // unknown rows, <= 64
  const bool live = lane < m;
  const double bdiag = mode_diag(md, a.ny, a.nxh, a.Lx, a.Ly, a.dz);
  double2* s = a.spec + md + (long long)(lane + 1) * ms;
  const double dz2 = a.dz * a.dz;
  double lo = live && lane > 0 ? 1.0 : 0.0;      // coefficient of x[i - stride]
  double up = live && lane < m - 1 ? 1.0 : 0.0;  // coefficient of x[i + stride]
  double bd = live ? bdiag : 1.0;
  double rr = 0.0, ri = 0.0;
  if (live) {
    const double2 r = *s;
    rr = dz2 * r.x;
    ri = dz2 * r.y;
  }
#pragma unroll
  for (int st = 1; st < 64; st <<= 1) {
    // neighbours' rows (identity rows outside the wave)
    const bool hl = lane - st >= 0, hu = lane + st < 64;
    double lo_l = __shfl_up(lo, st, 64), up_l = __shfl_up(up, st, 64), bd_l = __shfl_up(bd, st, 64);
    double rr_l = __shfl_up(rr, st, 64), ri_l = __shfl_up(ri, st, 64);
    double lo_u = __shfl_down(lo, st, 64), up_u = __shfl_down(up, st, 64), bd_u = __shfl_down(bd, st, 64);
    double rr_u = __shfl_down(rr, st, 64), ri_u = __shfl_down(ri, st, 64);
    if (!hl) { lo_l = 0.0; up_l = 0.0; bd_l = 1.0; rr_l = 0.0; ri_l = 0.0; }
    if (!hu) { lo_u = 0.0; up_u = 0.0; bd_u = 1.0; rr_u = 0.0; ri_u = 0.0; }
    const double al = -lo / bd_l, ga = -up / bd_u;
    bd = bd + al * up_l + ga * lo_u;
    rr = rr + al * rr_l + ga * rr_u;
    ri = ri + al * ri_l + ga * ri_u;
    lo = al * lo_l;
    up = ga * up_u;
  }
  if (live) *s = make_double2(rr / bd * a.inv_nxny, ri / bd * a.inv_nxny);  // 1/(NX NY) folded in, as in k_tridiag
}

// odd_extract + gpu_efield + gpu_bc fused (poisson.cu:191-204, 40-69): phi = ifft/(NX NY) on
// interior planes, wall planes pinned to voltage/voltage2; E = central differences of phi,
// periodic in x,y; Ez of a wall plane copies the neighbouring interior plane.
// planes marched per thread: 16 on large lattices (no measurable difference between 1 and 64
// there, profiles/r01_sweep_phi_zchunk.log), 1 on small ones where the serial chain of a column
// would be the whole run time of the kernel
#ifndef EKPNP_PHI_ZCHUNK
#define EKPNP_PHI_ZCHUNK 16  // A/B knob
#endif
constexpr int PHI_ZCHUNK_LARGE = EKPNP_PHI_ZCHUNK;

// One thread marches up a column of PHI_ZCHUNK planes with phi(z-1), phi(z), phi(z+1) in
// registers: every phi value is read once for the three z uses (the x+-1 / y+-1 neighbours come
// from the same or the adjacent row, i.e. from cache).
//
// z is uniform over the workgroup, so every case distinction (wall plane, slab halo) is a scalar
// select of a pointer / scale / value - no branch surrounds a load - and the five loads of plane
// z+1 are issued before the four stores of plane z (a first version with phi_at() per operand
// compiled to load - wait - store chains: 1.71 ms on 512^3 against 0.9 ms at copy speed).
//
// XCD-aware placement as in k_collide_bulk: the (y, z-chunk) rows are dealt to the 8 XCDs in
// runs of 64 consecutive y, so the y+-1 neighbour rows are found in the XCD's own L2 (with rows
// dealt one by one every L2 fetched all three rows: 4.1 GB fetched for 1.07 GB of phi).
// 16-byte store to an address that is only guaranteed to be 8-byte aligned (caller-bound field
// arrays, ekpnp_bind_field): global_store_dwordx4 needs dword alignment only
typedef double pair8 __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ void store_pair(double* p, double a, double b) {
  pair8 v = {a, b};
#ifndef EKPNP_PHI_PLAIN_STORE  // E is next read by the collide of the following step, tens of ms later:
  __builtin_nontemporal_store(v, reinterpret_cast<pair8*>(p));  // non-temporal (814 vs 823-848 us, profiles/r02_phi_variants.log)
#else
  *reinterpret_cast<pair8*>(p) = v;
#endif
}
__device__ __forceinline__ double2 load_pair(const double* p) {
  const pair8 v = *reinterpret_cast<const pair8*>(p);
  return make_double2(v.x, v.y);
}

struct PhiColumn {
  const double* __restrict__ work;   // the phi array: its interior planes were written by the inverse transform
  const double* __restrict__ lo;
  const double* __restrict__ hi;
  const double* __restrict__ vwall;  // {voltage, voltage, voltage2, voltage2} in device memory: the wall case is a pointer select too
  long long plane, oc;
  int z0, nzl, nz;
  // phi(x, y, z) for z0-1 <= z <= z0+nzl (phi_at() without branches: one unconditional load)
  __device__ __forceinline__ double center(int z) const {
    const int zl = z - z0;
    const bool wall_lo = z <= 0, wall_hi = z >= nz - 1;
    const int zc = zl < 0 ? 0 : (zl >= nzl ? nzl - 1 : zl);
    const double* p = work + (long long)zc * plane + oc;
    p = zl < 0 ? lo + oc : p;
    p = zl >= nzl ? hi + oc : p;
    p = wall_lo ? vwall : p;
    p = wall_hi ? vwall + 2 : p;
    return *p;
  }
  // the same for the node pair (x, x+1), x even: one 16-byte load
  __device__ __forceinline__ double2 center2(int z) const {
    const int zl = z - z0;
    const bool wall_lo = z <= 0, wall_hi = z >= nz - 1;
    const int zc = zl < 0 ? 0 : (zl >= nzl ? nzl - 1 : zl);
    const double* p = work + (long long)zc * plane + oc;
    p = zl < 0 ? lo + oc : p;
    p = zl >= nzl ? hi + oc : p;
    p = wall_lo ? vwall : p;
    p = wall_hi ? vwall + 2 : p;
    const pair8 v = *reinterpret_cast<const pair8*>(p);  // a caller-bound phi array may be 8-byte aligned only
    return make_double2(v.x, v.y);
  }
};



template <int PHI_ZCHUNK>
__global__ void __launch_bounds__(256) k_phi_efield(PArgs a, const int nxb, const int nrows) {
  constexpr int RCHUNK = 64;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int r = slot / nxb, xb = slot - r * nxb;
  const int row = ((r / RCHUNK) * 8 + xcd) * RCHUNK + r % RCHUNK;
  if (row >= nrows) return;
  const int x = xb * blockDim.x + threadIdx.x;
  if (x >= a.nx) return;
  const int y = row % a.ny;
  const int zl0 = (row / a.ny) * PHI_ZCHUNK;
  const int zl1 = min(zl0 + PHI_ZCHUNK, a.nzl);
  const int xp1 = x + 1 == a.nx ? 0 : x + 1, xm1 = x == 0 ? a.nx - 1 : x - 1;
  const int yp1 = y + 1 == a.ny ? 0 : y + 1, ym1 = y == 0 ? a.ny - 1 : y - 1;
  const long long oc = (long long)y * a.nx + x;
  const long long oxm = (long long)y * a.nx + xm1, oxp = (long long)y * a.nx + xp1;
  const long long oym = (long long)ym1 * a.nx + x, oyp = (long long)yp1 * a.nx + x;
  const PhiColumn col{a.fld[EKPNP_PHI], a.phi_lo, a.phi_hi, a.vwall, a.plane, oc, a.z0, a.nzl, a.nz};
  const double* w = a.fld[EKPNP_PHI];  // interior planes: phi as the inverse transform left it (wall planes: written below)
  double* o_phi = a.fld[EKPNP_PHI];
  double* __restrict__ o_ex = a.fld[EKPNP_EX];
  double* __restrict__ o_ey = a.fld[EKPNP_EY];
  double* __restrict__ o_ez = a.fld[EKPNP_EZ];

  double pm = col.center(a.z0 + zl0 - 1);
  double p0 = col.center(a.z0 + zl0);
  // operands of plane zl0
  const double* wz = w + (long long)zl0 * a.plane;
  double nxm = wz[oxm], nxp = wz[oxp], nym = wz[oym], nyp = wz[oyp];
  double pp = col.center(a.z0 + zl0 + 1);
#pragma unroll 4
  for (int zl = zl0; zl < zl1; ++zl) {
    const int z = a.z0 + zl;
    // next plane's operands first (clamped to the chunk: the last iteration re-reads its own plane)
    const int zn = zl + 1 < zl1 ? zl + 1 : zl;
    const double* wn = w + (long long)zn * a.plane;
    const double n_xm = wn[oxm], n_xp = wn[oxp], n_ym = wn[oym], n_yp = wn[oyp];
    const double n_pp = col.center(a.z0 + zn + 1);

    const bool wall = z == 0 || z == a.nz - 1;
    const double vw = z == 0 ? a.voltage : a.voltage2;
    const double exm = wall ? vw : nxm, exp_ = wall ? vw : nxp;
    const double eym = wall ? vw : nym, eyp = wall ? vw : nyp;
    const long long i = (long long)zl * a.plane + oc;
    // (plain stores: non-temporal ones, other chunk lengths and block widths measured the same,
    // profiles/r01_sweep_phi_variants.log)
    if (wall) o_phi[i] = p0;  // odd_extract pins the plates (poisson.cu:198-203); the interior is already there
    // the reference's expression 0.5*(a - b)/d (poisson.cu:53-55), kept so that E is the same
    // bits as a central difference of the returned phi
    o_ex[i] = 0.5 * (exm - exp_) / a.dx;
    o_ey[i] = 0.5 * (eym - eyp) / a.dy;
    const double ez = 0.5 * (pm - pp) / a.dz;
    // gpu_bc (poisson.cu:57-69): Ez(0) <- Ez(1), Ez(NZ-1) <- Ez(NZ-2); planes 0,1 and NZ-2,NZ-1 always
    // belong to the same slab, so the interior plane's thread writes its wall neighbour too
    if (!wall) o_ez[i] = ez;
    if (z == 1) o_ez[i - a.plane] = ez;
    if (z == a.nz - 2) o_ez[i + a.plane] = ez;
    pm = p0;
    p0 = pp;
    pp = n_pp;
    nxm = n_xm; nxp = n_xp; nym = n_ym; nyp = n_yp;
  }
}

// The same kernel with TWO x nodes per lane (rows of a multiple of 128 nodes, i.e. every wave is
// full): 16-byte loads and stores, and the x neighbours come out of the neighbouring lanes'
// registers (the centre values the march holds anyway) instead of two more loads per node; only
// lane 0 / lane 63 fetch the one value beyond the wave's 128 nodes.  Per 128 nodes and plane: 4 loads
// and 4 stores instead of 10 and 8.  Same expressions, same bits as k_phi_efield.
template <int PHI_ZCHUNK>
__global__ void __launch_bounds__(256) k_phi_efield_x2(PArgs a, const int nxb, const int nrows) {
  constexpr int RCHUNK = 64;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int r = slot / nxb, xb = slot - r * nxb;
  const int row = ((r / RCHUNK) * 8 + xcd) * RCHUNK + r % RCHUNK;
  if (row >= nrows) return;
  const int x0 = (xb * blockDim.x + threadIdx.x) * 2;
  if (x0 >= a.nx) return;  // whole waves only (nx % 128 == 0)
  const int lane = threadIdx.x & 63;
  const int y = row % a.ny;
  const int zl0 = (row / a.ny) * PHI_ZCHUNK;
  const int zl1 = min(zl0 + PHI_ZCHUNK, a.nzl);
  const int yp1 = y + 1 == a.ny ? 0 : y + 1, ym1 = y == 0 ? a.ny - 1 : y - 1;
  const long long oc = (long long)y * a.nx + x0;
  const long long oym = (long long)ym1 * a.nx + x0, oyp = (long long)yp1 * a.nx + x0;
  // the one node outside the wave's 128: to the left for lane 0, to the right for lane 63
  const bool edge = lane == 0 || lane == 63;
  const int xe = lane == 0 ? (x0 == 0 ? a.nx - 1 : x0 - 1) : (x0 + 2 == a.nx ? 0 : x0 + 2);
  const long long oe = (long long)y * a.nx + xe;
  const PhiColumn col{a.fld[EKPNP_PHI], a.phi_lo, a.phi_hi, a.vwall, a.plane, oc, a.z0, a.nzl, a.nz};
  const double* w = a.fld[EKPNP_PHI];  // interior planes: phi as the inverse transform left it (wall planes: written below)
  double* o_phi = a.fld[EKPNP_PHI];
  double* __restrict__ o_ex = a.fld[EKPNP_EX];
  double* __restrict__ o_ey = a.fld[EKPNP_EY];
  double* __restrict__ o_ez = a.fld[EKPNP_EZ];

  double2 pm = col.center2(a.z0 + zl0 - 1);
  double2 p0 = col.center2(a.z0 + zl0);
  const double* wz = w + (long long)zl0 * a.plane;
  double2 nym = load_pair(wz + oym), nyp = load_pair(wz + oyp);
  double ne = edge ? wz[oe] : 0.0;
  double2 pp = col.center2(a.z0 + zl0 + 1);
// (no unroll request: the lane exchange is a convergent operation, the optimizer declines)
  for (int zl = zl0; zl < zl1; ++zl) {
    const int z = a.z0 + zl;
    // next plane's operands first (clamped to the chunk: the last iteration re-reads its own plane)
    const int zn = zl + 1 < zl1 ? zl + 1 : zl;
    const double* wn = w + (long long)zn * a.plane;
    const double2 n_ym = load_pair(wn + oym), n_yp = load_pair(wn + oyp);
    const double n_e = edge ? wn[oe] : 0.0;
    const double2 n_pp = col.center2(a.z0 + zn + 1);

    const bool wall = z == 0 || z == a.nz - 1;
    const double vw = z == 0 ? a.voltage : a.voltage2;
    const double es = wall ? vw : ne;
    const double from_left = __shfl_up(p0.y, 1, 64), from_right = __shfl_down(p0.x, 1, 64);
    const double left = lane == 0 ? es : from_left;     // phi(x0 - 1)
    const double right = lane == 63 ? es : from_right;  // phi(x0 + 2)
    const double eym0 = wall ? vw : nym.x, eym1 = wall ? vw : nym.y;
    const double eyp0 = wall ? vw : nyp.x, eyp1 = wall ? vw : nyp.y;
    const long long i = (long long)zl * a.plane + oc;
    if (wall) store_pair(o_phi + i, p0.x, p0.y);  // the plates; the interior is already there
    // the reference's expression 0.5*(a - b)/d (poisson.cu:53-55)
    store_pair(o_ex + i, 0.5 * (left - p0.y) / a.dx, 0.5 * (p0.x - right) / a.dx);
    store_pair(o_ey + i, 0.5 * (eym0 - eyp0) / a.dy, 0.5 * (eym1 - eyp1) / a.dy);
    const double ez0 = 0.5 * (pm.x - pp.x) / a.dz, ez1 = 0.5 * (pm.y - pp.y) / a.dz;
    // gpu_bc (poisson.cu:57-69): Ez(0) <- Ez(1), Ez(NZ-1) <- Ez(NZ-2)
    if (!wall) store_pair(o_ez + i, ez0, ez1);
    if (z == 1) store_pair(o_ez + i - a.plane, ez0, ez1);
    if (z == a.nz - 2) store_pair(o_ez + i + a.plane, ez0, ez1);
    pm = p0;
    p0 = pp;
    pp = n_pp;
    nym = n_ym; nyp = n_yp; ne = n_e;
  }
}

// ------------------------------------------------------------------------------------------
// z-slab version of the tridiagonal solve (no reference counterpart).
// Rank r owns the unknown rows a..e of the global system (the interior planes of its slab,
// m = e-a+1 of them).  With g_lo = x[a-1] and g_hi = x[e+1] (0 at a wall, the neighbour's edge
// value at a slab interface) the local rows read  A x = r - g_lo e_1 - g_hi e_m,  A = tridiag(1,b,1),
// hence  x = p - g_lo u - g_hi v,  p = A^-1 r,  u = A^-1 e_1,  v_k = u_{m+1-k}.
// Only the edge values (p_1, p_m) are needed to couple the slabs, so the Thomas solve is SPLIT
// around the exchange instead of being followed by a correction pass over all rows:
//   stage 1  forward elimination d' of the local rows (every TRI_BS-th row of it kept in the
//            spectrum), p_m = d'_m and p_1 = u . r (A is symmetric, so e_1^T A^-1 r = u^T r);
//   all-gather of (p_1, p_m); every rank solves the same 2(P-1)-unknown interface system per mode;
//   stage 2  back substitution of the TRUE system A x = r - g_lo e_1 - g_hi e_m: by linearity its
//            eliminated right-hand side is d'_k - g_lo w_k - g_hi c'_m [k = m], where w is the
//            forward elimination of e_1 (w_1 = c'_1, w_k = -w_{k-1} c'_k, a table like u).
// Per solve this reads the spectrum twice and writes it once (like the single-context Thomas
// solve) instead of three times each.

// u = A^-1 e_1 for an m-row block; stores the full vector, (u_1, u_m) and, if asked for, the
// forward elimination w of e_1.
__global__ void k_slab_unit_response(const double* __restrict__ cprime, int m, int nmodes, double* __restrict__ u, double* __restrict__ u1um,
                                     double* __restrict__ w) {
  const int md = blockIdx.x * blockDim.x + threadIdx.x;
  if (md >= nmodes) return;
  const long long ms = nmodes;
  const double* cp = cprime + md;
  // forward: d'_1 = 1 * c'_1, d'_k = (0 - d'_{k-1}) c'_k ; the table row k of cprime is c'_k
  // back:    u_m = d'_m, u_k = d'_k - c'_k u_{k+1}
  // d' is kept in u between the two sweeps
  double d = 0.0;
  for (int k = 1; k <= m; ++k) {
    d = ((k == 1 ? 1.0 : 0.0) - d) * cp[(long long)k * ms];
    u[(long long)(k - 1) * ms + md] = d;
    if (w) w[(long long)(k - 1) * ms + md] = d;
  }
  double x = d;
  for (int k = m - 1; k >= 1; --k) {
    x = u[(long long)(k - 1) * ms + md] - cp[(long long)k * ms] * x;
    u[(long long)(k - 1) * ms + md] = x;
  }
  u1um[md] = x;       // u_1
  u1um[ms + md] = d;  // u_m = d'_m
}

// stage 1: forward elimination of the owned unknown rows, in place; edges -> edge buffer
// edge layout per rank: [4][nmodes] = p_first.re, p_first.im, p_last.re, p_last.im
__global__ void k_slab_thomas_local(PArgs a, int row_a, int m, const double* __restrict__ u, double* __restrict__ edge) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;  // mode within the block a.bx0, a.bw; edge: [4][ny bw]
  const long long ms = (long long)a.ny * a.nxh, msl = (long long)a.ny * a.bw;
  if (j >= msl) return;
  const int md = (int)block_mode(a, j);
  const double dz2 = a.dz * a.dz;
  double2* s = a.spec + md + (long long)row_a * ms;  // local plane of the first unknown row
  const double b = mode_diag(md, a.ny, a.nxh, a.Lx, a.Ly, a.dz);
  const double* up = u + md;
  double dr = 0.0, di = 0.0, p1r = 0.0, p1i = 0.0, c = 0.0;
#pragma unroll 8
  for (int k = 1; k <= m; ++k) {
    const double2 r = s[(long long)(k - 1) * ms];
    c = 1.0 / (b - c);  // c'_k: the table's recurrence, in registers (k_build_cprime)
    const double uk = up[(long long)(k - 1) * ms];
    const double rr = dz2 * r.x, ri = dz2 * r.y;
    p1r += uk * rr;
    p1i += uk * ri;
    dr = (rr - dr) * c;
    di = (ri - di) * c;
    if ((k & (TRI_BS - 1)) == 0) s[(long long)(k - 1) * ms] = make_double2(dr, di);  // checkpoint, as in k_tridiag
  }
  edge[j] = p1r;             // p_1 = u . r
  edge[msl + j] = p1i;
  edge[2 * msl + j] = dr;    // p_m = d'_m
  edge[3 * msl + j] = di;
}

// stage 1 of the read-once slab solve (round 3): ONLY the two edge values, as dot products.  A is symmetric and
// persymmetric, so  p_1 = e_1^T A^-1 r = u . r  and  p_m = e_m^T A^-1 r = sum_j u_j r_{m+1-j}:  no recurrence, no
// checkpoint rows, and - u_j decays like lambda^j, lambda = the small root of  lambda^2 + b lambda + 1 = 0  -
// only the K rows at either end of the block where u_j is not below 2^-66 of u_1.  The bound is ABSOLUTE: a dropped term
// is |u_j r_j| <= 2^-66 |u_1| max|r|, i.e. what is neglected in p_1 (p_m) is below 2^-66 m of |u_1| max|r| - a rounding
// error of the edge value whenever the edge value is of the size of u_1 max|r|, and in any case ~1e-20 of the field's scale
// (it is NOT relative to the kept sum: with charge confined to the planes next to one plate the kept terms of the far
// edge's p can be smaller than a dropped one; tests/test_group_gpu.py::test_slab_edge_values_with_charge_at_one_plate).  The lowest modes (b -> -2, lambda -> 1) need every row, a mid-range mode a few
// dozen: on a 512-plane slab the kernel touches ~1/6 of the spectrum instead of all of it.  Workgroup = 64 adjacent
// modes x EDGE_SEGS segments of the K rows, partial sums combined through LDS in segment order.
constexpr int EDGE_SEGS = 16;  // 4 measured first: 0.133 ms on a 512-plane slab, the low modes' 128 dependent rows per thread being the critical path
__global__ void __launch_bounds__(64 * EDGE_SEGS) k_slab_edges(PArgs a, int row_a, int m, const double* __restrict__ u, double* __restrict__ edge) {
  __shared__ double part[EDGE_SEGS][4][64];
  const int tx = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const long long ms = (long long)a.ny * a.nxh, msl = (long long)a.ny * a.bw;
  const long long jm = (long long)blockIdx.x * 64 + tx;  // mode within the block a.bx0, a.bw; edge: [4][ny bw]
  const bool live = jm < msl;
  const long long md = live ? block_mode(a, jm) : 0;
  int K = 0;
  if (live) {
    const double b = mode_diag((int)md, a.ny, a.nxh, a.Lx, a.Ly, a.dz);
    // lambda = 2 / (-b + sqrt(b^2 - 4)) in (0, 1];  K = rows until lambda^K < 2^-66
    const double disc = b * b - 4.0;
    const double lam = 2.0 / (-b + sqrt(disc > 0.0 ? disc : 0.0));
    const double nl = -log(lam);
    K = (nl * (double)m > 45.75) ? (int)(45.75 / nl) + 1 : m;
    if (K > m) K = m;
  }
  // The K rows of a mode are dealt to the EDGE_SEGS segments by the mode's OWN K, rounded up to a power of two (so that the
  // 64 adjacent modes of a wave, whose K differ little, mostly walk the same rows: coalesced) - NOT by the longest K of the
  // workgroup as in rounds 3-4: the order in which a mode's terms are added must not depend on which other modes happen to
  // share its workgroup, or the solve's bits would change with the mode blocks ("edge_chunks").
  int Kc = K;
  if (K > 1 && K < m) Kc = min(m, 1 << (32 - __clz(K - 1)));
  const int chunk = (Kc + EDGE_SEGS - 1) / EDGE_SEGS;
  const int j0 = seg * chunk + 1, j1 = min(j0 + chunk - 1, K);  // this thread's rows j0 .. j1 of 1 .. K
  double p1r = 0.0, p1i = 0.0, pmr = 0.0, pmi = 0.0;
  if (live) {
    const double2* s = a.spec + md + (long long)row_a * ms;
    const double* up = u + md;
#pragma unroll 4
    for (int j = j0; j <= j1; ++j) {
      const double uj = up[(long long)(j - 1) * ms];
      const double2 lo = s[(long long)(j - 1) * ms];
      const double2 hi = s[(long long)(m - j) * ms];
      p1r = fma(uj, lo.x, p1r);
      p1i = fma(uj, lo.y, p1i);
      pmr = fma(uj, hi.x, pmr);
      pmi = fma(uj, hi.y, pmi);
    }
  }
  part[seg][0][tx] = p1r; part[seg][1][tx] = p1i; part[seg][2][tx] = pmr; part[seg][3][tx] = pmi;
  __syncthreads();
  if (seg < 4 && live) {  // wave q sums quantity q over the segments, in segment order
    const double dz2 = a.dz * a.dz;
    double acc = part[0][seg][tx];
#pragma unroll
    for (int g = 1; g < EDGE_SEGS; ++g) acc += part[g][seg][tx];
    edge[(long long)seg * msl + jm] = dz2 * acc;
  }
}

// stage 2: interface system (block tridiagonal, 2x2 blocks, P-1 interfaces) + back substitution.
// Interface i sits between slab i and i+1: X_i = x_last(slab i), Y_i = x_first(slab i+1):
//   um_i X_{i-1} + X_i + u1_i Y_i             = p_last(i)
//   u1_{i+1} X_i + Y_i + um_{i+1} Y_{i+1}     = p_first(i+1)
// u1/um of slab j are those of ITS row count (slabs may differ by a plane; the edge slabs lose their plate).
constexpr int MAXR = 16;
// the interface system is tiny (2(P-1) unknowns per mode) but wants per-thread arrays; it has a
// kernel of its own so that the sweep below keeps its registers and needs no scratch memory
// (fused, the sweep carried 1.8 KB of scratch per lane: 0.650 ms on a 512^3 slab against 0.458 + 0.006 ms
// split, profiles/r02_slab_after_kernel_stats.csv)
__global__ void k_slab_interface(PArgs a, int rank, int nranks, const double* __restrict__ edges_all, const double* __restrict__ u1um_all,
                                 double* __restrict__ g) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;  // mode within the block a.bx0, a.bw; edges_all: [nranks][4][ny bw], g: [4][ny bw]
  const long long ms = (long long)a.ny * a.nxh, msl = (long long)a.ny * a.bw;
  if (j >= msl) return;
  const int md = (int)block_mode(a, j);
  const int ni = nranks - 1;
  double u1[MAXR], um[MAXR];
  for (int q = 0; q < nranks; ++q) {
    const double* t = u1um_all + (long long)q * 2 * ms;  // (u_1, u_m) of rank q's block (its own row count)
    u1[q] = t[md];
    um[q] = t[ms + md];
  }
  // block Thomas.  D_i = [[1, u1_i],[u1_{i+1}, 1]], L_i = [[um_i,0],[0,0]], U_i = [[0,0],[0,um_{i+1}]]
  // forward: D'_i = D_i - L_i D'^{-1}_{i-1} U_{i-1};  R'_i = R_i - L_i D'^{-1}_{i-1} R'_{i-1}
  double d11[MAXR], d12[MAXR], d21[MAXR], d22[MAXR];
  double rxr[MAXR], rxi[MAXR], ryr[MAXR], ryi[MAXR];
  for (int i = 0; i < ni; ++i) {
    const double* ei = edges_all + (long long)i * 4 * msl;        // slab i
    const double* ej = edges_all + (long long)(i + 1) * 4 * msl;  // slab i+1
    double a11 = 1.0, a12 = u1[i], a21 = u1[i + 1], a22 = 1.0;
    double bxr = ei[2 * msl + j], bxi = ei[3 * msl + j];  // p_last(i)
    double byr = ej[j], byi = ej[msl + j];                // p_first(i+1)
    if (i > 0) {
      // L_i D'^{-1}_{i-1}: only row 0, L = [[um_i,0],[0,0]] -> row0 = um_i * (first row of D'^{-1})
      const double det = d11[i - 1] * d22[i - 1] - d12[i - 1] * d21[i - 1];
      const double i11 = d22[i - 1] / det, i12 = -d12[i - 1] / det;  // first row of the inverse
      const double l1 = um[i] * i11, l2 = um[i] * i12;
      // U_{i-1} = [[0,0],[0,um_i]] -> (L D'^{-1} U) = [[0, l2*um_i],[0,0]]
      a12 -= l2 * um[i];
      bxr -= l1 * rxr[i - 1] + l2 * ryr[i - 1];
      bxi -= l1 * rxi[i - 1] + l2 * ryi[i - 1];
    }
    d11[i] = a11; d12[i] = a12; d21[i] = a21; d22[i] = a22;
    rxr[i] = bxr; rxi[i] = bxi; ryr[i] = byr; ryi[i] = byi;
  }
  // back substitution: Z_i = D'^{-1}_i (R'_i - U_i Z_{i+1}),  U_i Z_{i+1} = (0, um_{i+1} Y_{i+1})
  double Xr[MAXR], Xi[MAXR], Yr[MAXR], Yi[MAXR];
  for (int i = ni - 1; i >= 0; --i) {
    double bxr = rxr[i], bxi = rxi[i], byr = ryr[i], byi = ryi[i];
    if (i < ni - 1) {
      byr -= um[i + 1] * Yr[i + 1];
      byi -= um[i + 1] * Yi[i + 1];
    }
    const double det = d11[i] * d22[i] - d12[i] * d21[i];
    Xr[i] = (d22[i] * bxr - d12[i] * byr) / det;
    Xi[i] = (d22[i] * bxi - d12[i] * byi) / det;
    Yr[i] = (-d21[i] * bxr + d11[i] * byr) / det;
    Yi[i] = (-d21[i] * bxi + d11[i] * byi) / det;
  }
  // the two values this rank's rows see: x just below its first row, x just above its last row
  g[j] = rank > 0 ? Xr[rank - 1] : 0.0;
  g[msl + j] = rank > 0 ? Xi[rank - 1] : 0.0;
  g[2 * msl + j] = rank < nranks - 1 ? Yr[rank] : 0.0;
  g[3 * msl + j] = rank < nranks - 1 ? Yi[rank] : 0.0;
}

__global__ void __launch_bounds__(64) k_slab_reduce_correct(PArgs a, int row_a, int m, const double* __restrict__ g, const double* __restrict__ w) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;  // mode within the block a.bx0, a.bw; g: [4][ny bw]
  const long long ms = (long long)a.ny * a.nxh, msl = (long long)a.ny * a.bw;
  if (j >= msl) return;
  const int md = (int)block_mode(a, j);
  const double glr = g[j], gli = g[msl + j], ghr = g[2 * msl + j], ghi = g[3 * msl + j];
  // back substitution of A x = r - g_lo e_1 - g_hi e_m; d' is recomputed block by block from the
  // checkpoints stage 1 left in every TRI_BS-th row (as in k_tridiag)
  double2* s = a.spec + md + (long long)row_a * ms;
  const double* cp = a.cprime + md;  // table row k = c'_k; only the rows below the blocks are read
  const double* wp = w + md;         // table row k-1 = w_k; likewise
  const double b = mode_diag(md, a.ny, a.nxh, a.Lx, a.Ly, a.dz);
  const double dz2 = a.dz * a.dz;
  double xr = 0.0, xi = 0.0;
  for (int khi = m; khi >= 1;) {
    const int klo = ((khi - 1) / TRI_BS) * TRI_BS + 1;
    double2 d[TRI_BS];
    double c[TRI_BS], wv[TRI_BS];
    double2 prev = make_double2(0.0, 0.0);
    double cprev = 0.0, wprev = 0.0;
    if (klo > 1) {
      prev = s[(long long)(klo - 2) * ms];
      cprev = cp[(long long)(klo - 1) * ms];
      wprev = wp[(long long)(klo - 2) * ms];
    }
#pragma unroll
    for (int i = 0; i < TRI_BS; ++i) {
      const int k = klo + i;
      if (k <= khi) d[i] = s[(long long)(k - 1) * ms];
    }
#pragma unroll
    for (int i = 0; i < TRI_BS; ++i) {
      const int k = klo + i;
      if (k <= khi) {
        // c'_k and w_k by the recurrences that filled the tables (k_build_cprime, k_slab_unit_response)
        cprev = 1.0 / (b - cprev);
        c[i] = cprev;
        wprev = ((k == 1 ? 1.0 : 0.0) - wprev) * cprev;
        wv[i] = wprev;
        if (i != TRI_BS - 1) {
          d[i].x = (dz2 * d[i].x - prev.x) * c[i];
          d[i].y = (dz2 * d[i].y - prev.y) * c[i];
        }
        prev = d[i];
      }
    }
#pragma unroll
    for (int i = TRI_BS - 1; i >= 0; --i) {
      const int k = klo + i;
      if (k <= khi) {
        if (k == m) {
          xr = (d[i].x - glr * wv[i]) - ghr * c[i];
          xi = (d[i].y - gli * wv[i]) - ghi * c[i];
        } else {
          xr = (d[i].x - glr * wv[i]) - c[i] * xr;
          xi = (d[i].y - gli * wv[i]) - c[i] * xi;
        }
        TRI_STORE(s + (long long)(k - 1) * ms, make_double2(xr * a.inv_nxny, xi * a.inv_nxny));  // 1/(NX NY) folded in, as in k_tridiag
      }
    }
    khi = klo - 1;
  }
}

// phi of the slab's first and last plane (interior value or pinned wall value) for the neighbours
__global__ void k_phi_halo_pack(PArgs a, double* __restrict__ send_dn, double* __restrict__ send_up) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.plane) return;
  const int zb = a.z0, zt = a.z0 + a.nzl - 1;
  const double* phi = a.fld[EKPNP_PHI];  // interior planes hold the solution already (the plates may not yet)
  send_dn[i] = zb == 0 ? a.voltage : phi[i];
  send_up[i] = zt == a.nz - 1 ? a.voltage2 : phi[(long long)(a.nzl - 1) * a.plane + i];
}

int build_cprime(Ctx& c) {
  const int nm = c.p.ny * c.nxh;
  auto hipfail = [&](const char* what, hipError_t e) {
    c.err = std::string(what) + ": " + hipGetErrorString(e);
    return e == hipErrorOutOfMemory ? EKPNP_ERR_NOMEM : EKPNP_ERR_HIP;
  };
  // single context: rows 1..nz-2 of the global system; slab: rows 1..nzl of a local block
  int rows_nz = c.p.nz;
  if (c.slab) {  // rows 1..(longest block among the ranks): the set-up below needs every rank's (u_1, u_m)
    rows_nz = 0;
    for (int r = 0; r < c.nranks; ++r) rows_nz = slab_rows(c.p.nz, c.nranks, r) + 2 > rows_nz ? slab_rows(c.p.nz, c.nranks, r) + 2 : rows_nz;
  }
  hipLaunchKernelGGL(k_build_cprime, dim3((nm + 127) / 128), dim3(128), 0, c.stream, c.cprime, c.p.nx, c.p.ny, rows_nz, c.nxh, c.p.Lx,
                     c.p.Ly, c.p.dz, c.slab ? 1 : TRI_BS);
  note_launch(c, "k_build_cprime");
  if (take_launch_error(c) != hipSuccess) return EKPNP_ERR_HIP;  // message names the kernel
  if (c.slab) {
    // (u_1, u_m) of every rank's block - they depend on the block's row count only, so each distinct
    // count is computed once (set-up only); this rank's own run also fills its u and w tables
    const int np = c.nranks;
    int rows_max = 0;
    for (int r = 0; r < np; ++r) rows_max = slab_rows(c.p.nz, np, r) > rows_max ? slab_rows(c.p.nz, np, r) : rows_max;
    double* tmp = nullptr;
    hipError_t e = hipMalloc((void**)&tmp, (size_t)rows_max * nm * sizeof(double));
    if (e != hipSuccess) return hipfail("scratch allocation for the slab unit response", e);
    int rc = EKPNP_OK;
    for (int r = 0; r < np && rc == EKPNP_OK; ++r) {
      const int m = slab_rows(c.p.nz, np, r);
      double* slot = c.u1um + (size_t)r * 2 * nm;
      int same = -1;
      for (int q = 0; q < r; ++q)
        if (q != c.rank && slab_rows(c.p.nz, np, q) == m) { same = q; break; }
      if (r == c.rank) {
        hipLaunchKernelGGL(k_slab_unit_response, dim3((nm + 127) / 128), dim3(128), 0, c.stream, c.cprime, m, nm, c.slab_u, slot, c.slab_w);
        note_launch(c, "k_slab_unit_response");
      } else if (same >= 0) {
        e = hipMemcpyAsync(slot, c.u1um + (size_t)same * 2 * nm, (size_t)2 * nm * sizeof(double), hipMemcpyDeviceToDevice, c.stream);
        if (e != hipSuccess) rc = hipfail("copy of a unit response", e);
      } else {
        hipLaunchKernelGGL(k_slab_unit_response, dim3((nm + 127) / 128), dim3(128), 0, c.stream, c.cprime, m, nm, tmp, slot, (double*)nullptr);
        note_launch(c, "k_slab_unit_response");
      }
      if (rc == EKPNP_OK && take_launch_error(c) != hipSuccess) rc = EKPNP_ERR_HIP;
    }
    const hipError_t es = hipStreamSynchronize(c.stream);
    const hipError_t ef = hipFree(tmp);
    if (rc) return rc;
    if (es != hipSuccess) return hipfail("slab unit response", es);
    if (ef != hipSuccess) return hipfail("hipFree of the unit-response scratch", ef);
  }
  return EKPNP_OK;
}

// Slabs of up to 512 unknown rows take the read-once pair (k_slab_edges + k_slab_part: the spectrum is read once for
// the solve, plus the few rows at either end of the block for the edge values); taller slabs and
// ekpnp_tune(ctx, "tri_partition", 0) (the A/B partner) keep the serial pair k_slab_thomas_local + k_slab_reduce_correct.
// EKPNP_TRI_WIDE_MODES=0: one wavefront per mode also on short columns (the A/B partner of LANES = 16 / 32)
static inline bool wide_modes() {
  static const bool on = !(std::getenv("EKPNP_TRI_WIDE_MODES") && std::atoi(std::getenv("EKPNP_TRI_WIDE_MODES")) == 0);
  return on;
}
static inline bool slab_read_once(const Ctx& c) { return c.tri_partition > 0 && c.tri_lds_ok && c.nxh % 8 == 0 && c.slab_m >= 1 && c.slab_m <= 512; }

// ---- mode blocks of the slab z solve ("edge_chunks", ekpnp_internal.h: Ctx::edge_chunks) --------------------------------
// The half spectrum's nxh / 8 column groups (8 kx columns = one 128-byte line of every row) are dealt to the blocks as evenly
// as they go; block k covers kx in [x0, x0 + bw) and owns the piece [4][ny bw] at doubles offset 4 ny x0 of edge_local and
// [nranks][4][ny bw] at nranks 4 ny x0 of edge_all.  One block (the default) is the layout of rounds 1-4.
//
// The z-solve kernel (which instantiation of k_slab_part: how many modes share a workgroup) is chosen from the WHOLE
// spectrum, never from a block - a block solved by another instantiation would eliminate in another order and change the
// bits - so blocks are made of units of `gu` column groups such that every block holds whole workgroups of any of them
// (ny 8 gu a multiple of 32 modes; gu = 1 on every lattice with ny a multiple of 4), the same units on every rank.
static int slab_part_modes(const Ctx& c) {  // modes per workgroup of the k_slab_part instantiation this slab runs
  const int m = c.slab_m, nm = c.p.ny * c.nxh;
  if (m <= 128) return wide_modes() && nm % 32 == 0 ? 32 : 8;
  if (m <= 256) return wide_modes() && nm % 16 == 0 ? 16 : 8;
  return c.tri_wide && nm % 16 == 0 ? 16 : 8;
}
static int gcd_int(int a, int b) { return b == 0 ? a : gcd_int(b, a % b); }
static int block_unit_groups(const Ctx& c) {
  // The same on EVERY rank (the blocks are the pieces of a collective): it depends on the lattice only, not on this slab's row
  // count or knobs - units that hold whole workgroups of the widest instantiation (32 modes) hold whole ones of the others too.
  if (c.nxh % 8 != 0) return 1;
  return 4 / gcd_int(4, c.p.ny);
}
int edge_chunk_count(const Ctx& c) {
  if (c.nxh % 8 != 0) return 1;
  const int gu = block_unit_groups(c), units = (c.nxh / 8 + gu - 1) / gu;
  return c.edge_chunks < 1 ? 1 : (c.edge_chunks > units ? units : c.edge_chunks);
}
ModeBlock mode_block(const Ctx& c, int k) {
  ModeBlock b;
  if (c.nxh % 8 != 0) {
    b.x0 = 0;
    b.bw = c.nxh;
  } else {
    const int groups = c.nxh / 8, gu = block_unit_groups(c), units = (groups + gu - 1) / gu, nb = edge_chunk_count(c);
    const int u0 = (int)((long long)units * k / nb), u1 = (int)((long long)units * (k + 1) / nb);
    const int g0 = u0 * gu, g1 = u1 * gu < groups ? u1 * gu : groups;
    b.x0 = g0 * 8;
    b.bw = (g1 - g0) * 8;
  }
  b.local_off = (size_t)4 * c.p.ny * b.x0;
  b.all_off = (size_t)c.nranks * 4 * c.p.ny * b.x0;
  b.doubles = (size_t)4 * c.p.ny * b.bw;
  return b;
}
static inline PArgs block_args(const Ctx& c, const ModeBlock& b) {
  PArgs a = c.pargs();
  a.bx0 = b.x0;
  a.bw = b.bw;
  return a;
}

void launch_slab_thomas_local(Ctx& c, int k) {
  const ModeBlock blk = mode_block(c, k);
  PArgs a = block_args(c, blk);
  const int nm = c.p.ny * blk.bw;
  double* edge = c.edge_local + blk.local_off;
  if (slab_read_once(c)) {
    hipLaunchKernelGGL(k_slab_edges, dim3((nm + 63) / 64), dim3(64 * EDGE_SEGS), 0, c.stream, a, c.slab_row_a, c.slab_m, c.slab_u, edge);
    note_launch(c, "k_slab_edges");
    return;
  }
  hipLaunchKernelGGL(k_slab_thomas_local, dim3((nm + 63) / 64), dim3(64), 0, c.stream, a, c.slab_row_a, c.slab_m, c.slab_u, edge);
  note_launch(c, "k_slab_thomas_local");
}

void launch_slab_reduce_correct(Ctx& c, int k) {
  const ModeBlock blk = mode_block(c, k);
  PArgs a = block_args(c, blk);
  const int nm = c.p.ny * blk.bw;
  double* bound = c.edge_local + blk.local_off;
  // this block's piece of the rank's edge buffer has been gathered and is free again: it takes (g_lo, g_hi)
  hipLaunchKernelGGL(k_slab_interface, dim3((nm + 63) / 64), dim3(64), 0, c.stream, a, c.rank, c.nranks, c.edge_all + blk.all_off, c.u1um, bound);
  note_launch(c, "k_slab_interface");
  if (slab_read_once(c)) {
    const int m = c.slab_m;
    // (launch + name of one instantiation)
#define SLAB_PART(RR, LL, GROUP, NAME)                                                                                           \
    do {                                                                                                                         \
        hipLaunchKernelGGL((k_slab_part<RR, LL>), dim3(nm / (GROUP)), dim3(512), (RR) * 8192, c.stream, a, c.slab_row_a, m, bound); \
        note_launch(c, "k_slab_part<" NAME ">");                                                                                 \
    } while (0)
    // short columns: several modes per wavefront (LANES of the partition solve) where the mode count allows whole workgroups
    // - decided on the whole spectrum (slab_part_modes), the block holds whole workgroups of it by construction (mode_block)
    const int mc = slab_part_modes(c);
    if (m <= 128 && mc == 32) SLAB_PART(8, 16, 32, "8,16");
    else if (m <= 128) SLAB_PART(2, 64, 8, "2");
    else if (m <= 256 && mc == 16) SLAB_PART(8, 32, 16, "8,32");
    else if (m <= 256) SLAB_PART(4, 64, 8, "4");
    else if (mc == 16) {
      hipLaunchKernelGGL((k_slab_part<8, 64, 16>), dim3(nm / 16), dim3(1024), 16 * 8192, c.stream, a, c.slab_row_a, m, bound);
      note_launch(c, "k_slab_part<8,64,16>");
    } else SLAB_PART(8, 64, 8, "8");
#undef SLAB_PART
    return;
  }
  hipLaunchKernelGGL(k_slab_reduce_correct, dim3((nm + 63) / 64), dim3(64), 0, c.stream, a, c.slab_row_a, c.slab_m, bound, c.slab_w);
  note_launch(c, "k_slab_reduce_correct");
}

void launch_phi_halo_pack(Ctx& c) {
  PArgs a = c.pargs();
  hipLaunchKernelGGL(k_phi_halo_pack, dim3((unsigned)((c.plane + 255) / 256)), dim3(256), 0, c.stream, a, c.phi_halo[0], c.phi_halo[1]);
  note_launch(c, "k_phi_halo_pack");
}

void launch_poisson_rhs(Ctx& c) {
  PArgs a = c.pargs();
  hipLaunchKernelGGL(k_poisson_rhs, dim3((unsigned)((c.nloc + 255) / 256)), dim3(256), 0, c.stream, a);
  note_launch(c, "k_poisson_rhs");
}

// the dynamic-LDS limit of the partition solves is a per-DEVICE function attribute: set when a context is made, on
// the device it is made on (ekpnp_group_* drives contexts on several devices from one process)
bool tridiag_prepare_device() {
  hipError_t e = hipSuccess;
  auto lds = [&](const void* fn, int bytes) {
    if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  };
  lds(reinterpret_cast<const void*>(&k_tridiag_part<8, 64>), 8 * 8192);
  lds(reinterpret_cast<const void*>(&k_tridiag_part<4, 64>), 4 * 8192);
  lds(reinterpret_cast<const void*>(&k_tridiag_part<8, 32>), 8 * 8192);
  lds(reinterpret_cast<const void*>(&k_tridiag_part<8, 16>), 8 * 8192);
  lds(reinterpret_cast<const void*>(&k_slab_part<8, 64>), 8 * 8192);
  lds(reinterpret_cast<const void*>(&k_slab_part<4, 64>), 4 * 8192);
  lds(reinterpret_cast<const void*>(&k_slab_part<2, 64>), 2 * 8192);
  lds(reinterpret_cast<const void*>(&k_slab_part<8, 32>), 8 * 8192);
  lds(reinterpret_cast<const void*>(&k_slab_part<8, 16>), 8 * 8192);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  return true;
}
// the 16-wavefront forms (128 KB of LDS): a device that does not grant it keeps the 8-wavefront kernels
bool tridiag_wide_prepare_device() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tridiag_part<8, 64, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 8192);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_slab_part<8, 64, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 8192);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  return true;
}

// launch + name of one instantiation of the partition solve
#define TRI_PART(RR, LL, GROUP, NAME)                                                                                            \
    do {                                                                                                                         \
        hipLaunchKernelGGL((k_tridiag_part<RR, LL>), dim3(nb / (GROUP)), dim3(512), (RR) * 8192, c.stream, a);                   \
        note_launch(c, "k_tridiag_part<" NAME ">");                                                                              \
    } while (0)
// ---- column blocks of the single context's solve ("poisson_blocks", ekpnp_internal.h: Ctx::poisson_blocks) ----------------
// The three middle passes of a solve - y forward, z solve, y inverse - all work on kx COLUMNS of the half spectrum (every ky,
// every plane of a kx range).  Taken block by block, the three passes of one block back to back, a block that fits the
// 256 MiB Infinity Cache is read from HBM once and written once instead of three times each.  Same kernels, same
// instantiation (chosen on the whole spectrum, as for the slab's mode blocks), every mode solved by itself: same bits.
// Only where the library's own column passes and a partition solve run (cfg2 ... cfg5 plane shapes); else one block.
static bool tridiag_is_partition(const Ctx& c) {
  const int nm = c.p.ny * c.nxh, rows = c.p.nz - 2;
  const bool part = c.tri_partition > 0 && c.tri_lds_ok && c.nxh % 8 == 0;
  const bool large = c.tri_partition > 1 || (size_t)nm * (size_t)rows >= (size_t)4 * 1024 * 1024;
  return part && large && rows > 64 && rows <= 512;
}
// Measured (profiles/r05c_ab_poisson_blocks*.jsonl, MI355X): half spectra of 1.1 GB - cfg3's 512 x 512 x 512 and a 1024 x 1024 x
// 128 context - gain 0.13 ms per solve with THREE blocks (2.11 -> 1.97 ms inside cfg3's step; 2, 4, 6 blocks gain less, 11 and
// more lose: narrow blocks are read in short pieces, tools/mall_probe.hip prices that footprint at +0.2 ms per pass, more
// than the cache gives back), spectra of 0.13 - 0.27 GB (256^3, 512 x 512 x 128) gain nothing or lose 0.02 - 0.1 ms.  Hence the
// default (Ctx::poisson_blocks == 0): three blocks from 768 MiB of half spectrum on, one below.
int poisson_block_count(const Ctx& c) {
  if (c.slab || !c.own_fft || c.poisson_blocks == 1 || !tridiag_is_partition(c)) return 1;
  int want = c.poisson_blocks;
  if (want <= 0) want = (size_t)c.p.ny * c.nxh * (size_t)(c.p.nz - 2) * sizeof(double2) >= ((size_t)768 << 20) ? 3 : 1;
  const int gu = block_unit_groups(c), units = (c.nxh / 8 + gu - 1) / gu;
  return want > units ? units : want;
}
ModeBlock poisson_block_whole(const Ctx& c) {
  ModeBlock b{};
  b.x0 = 0;
  b.bw = c.nxh;
  return b;
}
ModeBlock poisson_block(const Ctx& c, int k) {
  ModeBlock b{};
  const int groups = c.nxh / 8, gu = block_unit_groups(c), units = (groups + gu - 1) / gu, nb = poisson_block_count(c);
  const int u0 = (int)((long long)units * k / nb), u1 = (int)((long long)units * (k + 1) / nb);
  const int g0 = u0 * gu, g1 = u1 * gu < groups ? u1 * gu : groups;
  b.x0 = nb == 1 ? 0 : g0 * 8;
  b.bw = nb == 1 ? c.nxh : (g1 - g0) * 8;
  return b;
}

void launch_tridiag(Ctx& c, const ModeBlock* blk) {
  PArgs a = c.pargs();
  if (blk) { a.bx0 = blk->x0; a.bw = blk->bw; }
  const int nm = c.p.ny * c.nxh;   // the kernel is chosen on the WHOLE spectrum
  const int nb = c.p.ny * a.bw;    // modes of this launch
  // c.tri_partition (ekpnp_tune "tri_partition" / EKPNP_TRI_PARTITION): 0 the serial sweeps everywhere (the A/B
  // partner of k_tridiag_part), 1 the partition solve on large lattices, 2 wherever it applies (tests)
  const bool part = c.tri_partition > 0 && c.tri_lds_ok && c.nxh % 8 == 0;
  const int rows = c.p.nz - 2;
  const bool large = c.tri_partition > 1 || (size_t)nm * (size_t)rows >= (size_t)4 * 1024 * 1024;  // enough modes to fill the chip 8 at a time
  // every branch notes the kernel it really launched: a rejected launch is reported by that name
  if (rows <= 64) {
    hipLaunchKernelGGL(k_tridiag_pcr64, dim3((nm + 3) / 4), dim3(256), 0, c.stream, a);
    note_launch(c, "k_tridiag_pcr64");
  } else if (part && large && rows <= 128 && wide_modes() && nm % 32 == 0) {
    TRI_PART(8, 16, 32, "8,16");
  } else if (part && large && rows <= 256 && wide_modes() && nm % 16 == 0) {
    TRI_PART(8, 32, 16, "8,32");
  } else if (part && large && rows <= 256) {
    TRI_PART(4, 64, 8, "4");
  } else if (part && large && rows <= 512 && c.tri_wide && nm % 16 == 0) {
    // 16 wavefronts = 16 adjacent modes per workgroup: every row is read and written in 256-byte pieces instead of 128-byte ones
    hipLaunchKernelGGL((k_tridiag_part<8, 64, 16>), dim3(nb / 16), dim3(1024), 16 * 8192, c.stream, a);
    note_launch(c, "k_tridiag_part<8,64,16>");
  } else if (part && large && rows <= 512) {
    TRI_PART(8, 64, 8, "8");
  } else {
    hipLaunchKernelGGL(k_tridiag, dim3((nm + EKPNP_TRI_THREADS - 1) / EKPNP_TRI_THREADS), dim3(EKPNP_TRI_THREADS), 0, c.stream, a);
    note_launch(c, "k_tridiag");
  }
}

#undef TRI_PART

void launch_phi_efield(Ctx& c) {
  PArgs a = c.pargs();
  const bool small = c.nloc < (size_t)2 * 1024 * 1024;
  const int zchunk = small ? 1 : PHI_ZCHUNK_LARGE;
  const int nrows = c.p.ny * ((c.nzl + zchunk - 1) / zchunk);
  const long long per_xcd = ((long long)nrows + 8 * 64 - 1) / (8 * 64) * 64;
  static const bool no_x2 = std::getenv("EKPNP_PHI_X1") != nullptr;  // A/B knob: the one-node-per-lane kernel everywhere
  if (!small && !no_x2 && c.p.nx % 128 == 0) {  // whole waves of node pairs
#ifndef EKPNP_PHI_X2_THREADS
#define EKPNP_PHI_X2_THREADS 256  // A/B knob
#endif
    const int bx = c.p.nx >= 2 * EKPNP_PHI_X2_THREADS ? EKPNP_PHI_X2_THREADS : c.p.nx / 2;  // threads per block, each two nodes
    const int nxb = (c.p.nx + 2 * bx - 1) / (2 * bx);  // the last block of a row may be partly idle (whole waves: nx % 128 == 0)
    hipLaunchKernelGGL(k_phi_efield_x2<PHI_ZCHUNK_LARGE>, dim3((unsigned)(8 * per_xcd * nxb)), dim3(bx), 0, c.stream, a, nxb, nrows);
    note_launch(c, "k_phi_efield_x2<PHI_ZCHUNK_LARGE>");
    return;
  }
  const int bx = c.p.nx >= 256 ? 256 : 64;
  const int nxb = (c.p.nx + bx - 1) / bx;
  if (small)
    hipLaunchKernelGGL(k_phi_efield<1>, dim3((unsigned)(8 * per_xcd * nxb)), dim3(bx), 0, c.stream, a, nxb, nrows);
  else
    hipLaunchKernelGGL(k_phi_efield<PHI_ZCHUNK_LARGE>, dim3((unsigned)(8 * per_xcd * nxb)), dim3(bx), 0, c.stream, a, nxb, nrows);
  note_launch(c, "k_phi_efield");
}

}  // namespace ekpnp
