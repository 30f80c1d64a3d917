// lbm_kernels.hip — hand-written gfx950 kernels for the LBM half of the hot path.
//
// Replaces the four launches of stream_collide_save (LBM.cu:465-481):
//   gpu_collide_save (LBM.cu:483-1846), gpu_boundary (1848-1961), gpu_stream (1963-2093),
//   gpu_bc_charge (2095-2416)
// by ONE pull-stream + collide pass over double-buffered populations, plus a small kernel for
// the two wall planes.  The state kept between steps is the POST-collision population set
// (what the reference holds in f2/h2/hn2/temp2 after gpu_boundary); the pull in the next step
// performs gpu_stream, and the wall kernel performs gpu_bc_charge on the fly.
//
// Mapping to CDNA4: a workgroup is NL wave64s over the same 64 consecutive x nodes; wave l
// owns lattice l (f, h, hn, temp), so each lane holds 27 FP64 populations (54 VGPRs) instead
// of 108.  The waves exchange their seven moments through 3.5 KB of LDS, then every wave
// collides its own lattice.  Populations are tiled [zg][y][x/64][27][64] (ekpnp_internal.h): the 27
// stores of a wave fill one contiguous 13.8 KB tile, the 27 loads are 512-byte segments of the
// tiles of the 9 neighbour rows; row bases are wave-uniform (SGPR), the direction is an immediate
// offset and only the three x offsets live in VGPRs.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "ekpnp_internal.h"

namespace ekpnp {

template <int I, int N, int S, class Fn>
__device__ __forceinline__ void static_for(Fn&& fn) {
  if constexpr (I < N) {
    fn(std::integral_constant<int, I>{});
    static_for<I + S, N, S>(fn);
  }
}

// ------------------------------------------------------------------------------------------
// moments, LBM.cu:621-644 (same summation order as the reference)

__device__ __forceinline__ double sum27(const double (&f)[Q]) {
  double s = f[0];
#pragma unroll
  for (int d = 1; d < Q; ++d) s = s + f[d];
  return s;
}
__device__ __forceinline__ void momentum(const double (&f)[Q], double& jx, double& jy, double& jz) {
  jx = (f[1] + f[7] + f[9] + f[13] + f[15] + f[19] + f[21] + f[23] + f[26]) -
       (f[2] + f[8] + f[10] + f[14] + f[16] + f[20] + f[22] + f[24] + f[25]);
  jy = (f[3] + f[7] + f[11] + f[14] + f[17] + f[19] + f[21] + f[24] + f[25]) -
       (f[4] + f[8] + f[12] + f[13] + f[18] + f[20] + f[22] + f[23] + f[26]);
  jz = (f[5] + f[9] + f[11] + f[16] + f[18] + f[19] + f[22] + f[23] + f[25]) -
       (f[6] + f[10] + f[12] + f[15] + f[17] + f[20] + f[21] + f[24] + f[26]);
}

struct Force {
  double x, y, z;
};

// u = (sum c_i f_i / CFL + F dt/2) / rho, LBM.cu:639-644.  The fused multiply-add is written out: with
// -ffp-contract=fast the compiler may fuse either product of "j*cflinv + F*hdt", and it chose differently
// in different kernels (1 ulp in u on the plates between k_collide_wall and k_collide_all); an explicit fma
// leaves it no choice, so a node gets the same bits from every launch shape.
__device__ __forceinline__ double velocity(double rhoinv, double j, double cflinv, double F, double hdt) {
  return rhoinv * fma(F, hdt, j * cflinv);
}
// the body force, LBM.cu:635-637, with its fused multiply-adds written out for the same reason
__device__ __forceinline__ Force body_force(const KArgs& a, double c, double cn, double T, double Ex, double Ey, double Ez) {
  Force F;
  const double q = a.F * (c - cn);
  F.x = fma(q, Ex + a.Ext, a.exf);
  F.y = q * Ey;
  F.z = fma(q, Ez, a.rho0 * T * a.Ra * a.nu * a.D);
  return F;
}

// ------------------------------------------------------------------------------------------
// TRT collision of the fluid lattice with Guo forcing, LBM.cu:830-1845 (f rows).
// Algebraically the reference's formulas, written per (d, opp d) pair:
//   eq+ = w rho (omusq + t^2/2), eq- = w rho t, t = c_d.u/cs^2
//   F+  = w/cs^2 (-u.F + (e.u)(e.F) cflinv2),  F- = w/cs^2 cflinv (e.F)
//   out_d = f_d - [wp (f+ - eq+) + wm (f- - eq-)] + dt [sp F+ + sm F-]
template <class Store>
__device__ __forceinline__ void collide_fluid(const KArgs& a, const double (&f)[Q], double rho, double ux, double uy, double uz,
                                              const Force& F, Store&& store) {
  const double wp = a.wp[0], wm = a.wm[0];
  const double omusq = 1.0 - 0.5 * (ux * ux + uy * uy + uz * uz) * a.inv_cs2;
  const double ts = a.inv_cs2 * a.cflinv;
  const double uF = ux * F.x + uy * F.y + uz * F.z;
  const double dsp = a.dt * a.sp, dsm = a.dt * a.sm;
  {
    const double e0 = w_of(0) * rho * omusq;
    const double F0 = -(w_of(0) * a.inv_cs2) * uF;
    store(std::integral_constant<int, 0>{}, f[0] - wp * (f[0] - e0) + dsp * F0);
  }
  static_for<1, Q, 2>([&](auto ic) {
    constexpr int d = decltype(ic)::value;
    constexpr int cx = ex_of(d), cy = ey_of(d), cz = ez_of(d);
    constexpr double w = w_of(d);
    const double eu = cx * ux + cy * uy + cz * uz;
    const double eF = cx * F.x + cy * F.y + cz * F.z;
    const double t = eu * ts;
    const double wr = w * rho;
    const double ep = wr * (omusq + 0.5 * t * t);
    const double em = wr * t;
    const double fp = 0.5 * (f[d] + f[d + 1]);
    const double fm = 0.5 * (f[d] - f[d + 1]);
    const double Fp = (w * a.inv_cs2) * (eu * eF * a.cflinv2 - uF);
    const double Fm = (w * a.inv_cs2) * a.cflinv * eF;
    const double sym = wp * (fp - ep) - dsp * Fp;
    const double asym = wm * (fm - em) - dsm * Fm;
    store(std::integral_constant<int, d>{}, f[d] - sym - asym);
    store(std::integral_constant<int, d + 1>{}, f[d + 1] - sym + asym);
  });
}

// TRT collision of an advected scalar (h, hn, temp rows of LBM.cu:830-1845): equilibrium
// velocity v = u + mobility*E (ions) or u (temperature); no source term.
template <class Store>
__device__ __forceinline__ void collide_scalar(const KArgs& a, const double (&f)[Q], double m, double vx, double vy, double vz,
                                               double wp, double wm, Store&& store) {
  const double omusq = 1.0 - 0.5 * (vx * vx + vy * vy + vz * vz) * a.inv_cs2;
  const double ts = a.inv_cs2 * a.cflinv;
  {
    const double e0 = w_of(0) * m * omusq;
    store(std::integral_constant<int, 0>{}, f[0] - wp * (f[0] - e0));
  }
  static_for<1, Q, 2>([&](auto ic) {
    constexpr int d = decltype(ic)::value;
    constexpr int cx = ex_of(d), cy = ey_of(d), cz = ez_of(d);
    constexpr double w = w_of(d);
    const double t = (cx * vx + cy * vy + cz * vz) * ts;
    const double wr = w * m;
    const double ep = wr * (omusq + 0.5 * t * t);
    const double em = wr * t;
    const double fp = 0.5 * (f[d] + f[d + 1]);
    const double fm = 0.5 * (f[d] - f[d + 1]);
    const double sym = wp * (fp - ep);
    const double asym = wm * (fm - em);
    store(std::integral_constant<int, d>{}, f[d] - sym - asym);
    store(std::integral_constant<int, d + 1>{}, f[d + 1] - sym + asym);
  });
}

// equilibrium populations, LBM.cu:207-462 / 850-1103
__device__ __forceinline__ void equilibrium(const KArgs& a, double m, double vx, double vy, double vz, double (&eq)[Q]) {
  const double omusq = 1.0 - 0.5 * (vx * vx + vy * vy + vz * vz) * a.inv_cs2;
  const double ts = a.inv_cs2 * a.cflinv;
  static_for<0, Q, 1>([&](auto ic) {
    constexpr int d = decltype(ic)::value;
    const double t = (ex_of(d) * vx + ey_of(d) * vy + ez_of(d) * vz) * ts;
    eq[d] = (w_of(d) * m) * (omusq + t * (1.0 + 0.5 * t));
  });
}

// ------------------------------------------------------------------------------------------
// bulk kernel: every owned plane that is not a wall plane

#ifndef EKPNP_BULK_MIN_WAVES
#define EKPNP_BULK_MIN_WAVES 1  // tuning knob: min waves per SIMD the register allocator must allow
#endif
// the work of one bulk workgroup: row `row` of the launch (y = row % ny, plane zl_begin + row / ny), x block xb
// EPHI: E is not read from the Ex / Ey / Ez arrays but formed from phi right here - gpu_efield's central differences
// (poisson.cu:45-55) in the expression of k_phi_efield, 0.5*(a - b)/d, hence the same bits -, one component per wave
// (h: Ex, hn: Ey, temp - or f when there is no temperature lattice - Ez), shared through three more rows of the moment
// image.  The kernel then reads phi(y-1), phi(y), phi(y+1), phi(z-1), phi(z+1): 24 bytes per node of HBM traffic like
// the three E arrays (the y neighbours are L2 hits), and k_phi_efield (8 R + 24 W per node) drops out of the step.
// EDGE: the launch collides ONE plane, a slab's first or last: see KArgs::halo_*.
template <int NL, bool PULL, bool EPHI, bool EDGE = false>
__device__ __forceinline__ void bulk_body(const KArgs& a, const int zl_begin, const int row, const int xb) {
  __shared__ double mom[EPHI ? 10 : 7][64];
  const int y = row % a.ny;
  const int zl = zl_begin + row / a.ny;
  const int zg = zl + 1;
  const int lane = threadIdx.x & 63;
  const int lat = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int x = xb * 64 + lane;
  const bool act = x < a.nx;
  const int xc = act ? x : a.nx - 1;
  // source x for c_x = -1, 0, +1 (pull: x - c_x, periodic, LBM.cu:1970-1975)
  const unsigned xo[3] = {(unsigned)pop_xoff(xc + 1 == a.nx ? 0 : xc + 1), (unsigned)pop_xoff(xc), (unsigned)pop_xoff(xc == 0 ? a.nx - 1 : xc - 1)};
  const int ys[3] = {y + 1 == a.ny ? 0 : y + 1, y, y == 0 ? a.ny - 1 : y - 1};
  // the same three source x as plain node indices (the halo buffers are dense [ny][nx] planes)
  const int xr[3] = {xc + 1 == a.nx ? 0 : xc + 1, xc, xc == 0 ? a.nx - 1 : xc - 1};

  const double* __restrict__ src = a.A[lat];
  double f[Q];
  static_for<0, Q, 1>([&](auto ic) {
    constexpr int d = decltype(ic)::value;
    constexpr int cx = PULL ? ex_of(d) : 0, cy = PULL ? ey_of(d) : 0, cz = PULL ? ez_of(d) : 0;
    if constexpr (EDGE && cz != 0) {
      // a population that crosses the slab's face comes from the neighbouring slab: straight out of the receive buffer
      // (the plane index is uniform over the launch, so this is a scalar branch)
      const double* __restrict__ hb = cz > 0 ? a.halo_in_lo : a.halo_in_hi;
      if (hb && zl == (cz > 0 ? 0 : a.nzl - 1)) {
        f[d] = hb[((long long)lat * 9 + halo_slot(d)) * a.plane + (long long)ys[cy + 1] * a.nx + xr[cx + 1]];
        return;
      }
    }
    const double* rowp = src + ((long long)(zg - cz) * a.ny + ys[cy + 1]) * a.rowstride + slot_of(d) * 64;
#ifdef EKPNP_NT_LOADS  // A/B partner: non-temporal loads (every population is pulled exactly once) LOSE 3 %, profiles/r02_sweep_nt_loads.log
    f[d] = __builtin_nontemporal_load(rowp + xo[cx + 1]);
#else
    f[d] = rowp[xo[cx + 1]];
#endif
  });

  if constexpr (NL > 1) {
    if constexpr (EPHI) {
      constexpr int ZLAT = NL > 3 ? 3 : 0;  // the wave that forms Ez
      const double* __restrict__ ph = a.fld[EKPNP_PHI] + (long long)zl * a.plane;
      if (lat == 1) {
        const double* __restrict__ r = ph + (long long)y * a.nx;
        const double pl = r[xc == 0 ? a.nx - 1 : xc - 1], pr = r[xc + 1 == a.nx ? 0 : xc + 1];
        mom[7][lane] = 0.5 * (pl - pr) / a.dx;
      } else if (lat == 2) {
        const double pl = ph[(long long)ys[2] * a.nx + xc], pr = ph[(long long)ys[0] * a.nx + xc];
        mom[8][lane] = 0.5 * (pl - pr) / a.dy;
      }
      if (lat == ZLAT) {
        // phi(z-1), phi(z+1): the plate's pinned value, the neighbouring slab's plane or the array (z is uniform over the workgroup)
        const int z = a.z0 + zl;
        const long long oc = (long long)y * a.nx + xc;
        const double pm = z - 1 <= 0 ? a.voltage : (zl == 0 ? a.phi_lo[oc] : ph[oc - a.plane]);
        const double pp = z + 1 >= a.nz - 1 ? a.voltage2 : (zl == a.nzl - 1 ? a.phi_hi[oc] : ph[oc + a.plane]);
        mom[9][lane] = 0.5 * (pm - pp) / a.dz;
      }
    }
    if (lat == 0) {
      double jx, jy, jz;
      momentum(f, jx, jy, jz);
      mom[0][lane] = sum27(f);
      mom[1][lane] = jx;
      mom[2][lane] = jy;
      mom[3][lane] = jz;
    } else {
      mom[3 + lat][lane] = sum27(f);
    }
    __syncthreads();
  }
  double rho, jx, jy, jz, c = 0.0, cn = 0.0, T = 0.0, Ex = 0.0, Ey = 0.0, Ez = 0.0;
  const long long sidx = ((long long)zl * a.ny + y) * (long long)a.nx + xc;
  if constexpr (NL > 1) {
    rho = mom[0][lane];
    jx = mom[1][lane];
    jy = mom[2][lane];
    jz = mom[3][lane];
    c = mom[4][lane];
    cn = mom[5][lane];
    if constexpr (NL > 3) T = mom[6][lane];
    if constexpr (EPHI) {
      Ex = mom[7][lane];
      Ey = mom[8][lane];
      Ez = mom[9][lane];
    } else {
      Ex = a.fld[EKPNP_EX][sidx];
      Ey = a.fld[EKPNP_EY][sidx];
      Ez = a.fld[EKPNP_EZ][sidx];
    }
  } else {
    rho = sum27(f);
    momentum(f, jx, jy, jz);
  }
  const Force F = body_force(a, c, cn, T, Ex, Ey, Ez);
  const double rhoinv = 1.0 / rho;
  const double hdt = a.dt * 0.5;
  const double ux = velocity(rhoinv, jx, a.cflinv, F.x, hdt);  // LBM.cu:639-644
  const double uy = velocity(rhoinv, jy, a.cflinv, F.y, hdt);
  const double uz = velocity(rhoinv, jz, a.cflinv, F.z, hdt);

  double* __restrict__ dst = a.B[lat] + ((long long)zg * a.ny + y) * a.rowstride + xo[1];
  auto store = [&](auto ic, double v) {
    constexpr int d = decltype(ic)::value;
    // non-temporal: the populations written here are not read again before the next step
    // (+1 % on cfg3, profiles/r01_sweep_libs.log); EKPNP_PLAIN_STORES builds the A/B partner
#ifdef EKPNP_PLAIN_STORES
    if (act) dst[slot_of(d) * 64] = v;
#else
    if (act) __builtin_nontemporal_store(v, dst + slot_of(d) * 64);
#endif
    if constexpr (EDGE && ez_of(d) != 0) {
      // ... and one that leaves through the face goes into the send buffer as well (what k_halo_pack would copy there)
      double* __restrict__ hb = ez_of(d) < 0 ? a.halo_out_dn : a.halo_out_up;
      if (hb && act && zl == (ez_of(d) < 0 ? 0 : a.nzl - 1)) hb[((long long)lat * 9 + halo_slot(d)) * a.plane + (long long)y * a.nx + x] = v;
    }
  };
  // (a.wmom == 0: an intermediate step of a batch, KArgs::wmom - uniform over the launch, a scalar branch)
  if (lat == 0) {
    if (act && a.wmom) {  // LBM.cu:807-810
      a.fld[EKPNP_RHO][sidx] = rho;
      a.fld[EKPNP_UX][sidx] = ux;
      a.fld[EKPNP_UY][sidx] = uy;
      a.fld[EKPNP_UZ][sidx] = uz;
    }
    collide_fluid(a, f, rho, ux, uy, uz, F, store);
  } else {
    const double m = lat == 1 ? c : lat == 2 ? cn : T;
    if (act && a.wmom) a.fld[lat == 1 ? EKPNP_C : lat == 2 ? EKPNP_CN : EKPNP_T][sidx] = m;  // LBM.cu:811-813
    if (lat == 1 && act && a.rhs) {  // odd_extension's interior rows, poisson.cu:121-135, from registers
      a.rhs[sidx] = poisson_rhs_value(a.F, a.eps, c, cn, a.z0 + zl, a.nz, a.rhs_wall_lo, a.rhs_wall_hi);
    }
    const double k = a.mob[lat];
    collide_scalar(a, f, m, ux + k * Ex, uy + k * Ey, uz + k * Ez, a.wp[lat], a.wm[lat], store);
  }
}

// row of the launch a workgroup works on (XCD-aware placement), or -1 when it is beyond the last row
__device__ __forceinline__ int bulk_row_of_block(const int nrows, const int nxb, const int rchunk_arg, int& xb, const int ny = 0) {
  // XCD-aware placement.  Workgroups are dealt round-robin over the 8 XCDs (bid % 8 picks the
  // XCD).  Each XCD is given runs of `rchunk` CONSECUTIVE x rows (and walks along x inside a row),
  // so that every one of the 27+27 direction streams is sequential per XCD instead of a 1-in-8
  // row comb.  Measured on cfg3 (512^3, 4 lattices): 45.0 ms with rows dealt one by one,
  // 43.0 ms with runs of >= 8 rows (profiles/r01_sweep_map.log); 64 is used.
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int r = slot / nxb;
  xb = slot - r * nxb;
  const int rchunk = rchunk_arg & 0xffff, yband = rchunk_arg >> 16;
  int row = ((r / rchunk) * 8 + xcd) * rchunk + r % rchunk;
  if (row >= nrows) return -1;
  if (yband > 0) {
    // y bands (k_collide_bulk only; bulk_band_rows() decides, ekpnp_tune "bulk_yband"): the sweep takes band b (yband rows) of EVERY plane, then band
    // b + 1, instead of plane after plane - between the three uses of a phi row (as z+1, z, z-1) lie one and two band-planes
    // of traffic (118 / 237 MB at yband = 128 on 512 x 512 planes) instead of one and two planes (474 / 948 MB)
    const int per_band = (nrows / ny) * yband, band = row / per_band, rem = row - band * per_band, z = rem / yband;
    row = z * ny + band * yband + (rem - z * yband);
  }
  return row;
}

template <int NL, bool PULL, bool EPHI>
__global__ void __launch_bounds__(64 * NL, EKPNP_BULK_MIN_WAVES) k_collide_bulk(const KArgs a, const int zl_begin, const int nrows, const int nxb, const int rchunk) {
  int xb;
  const int row = bulk_row_of_block(nrows, nxb, rchunk, xb, a.ny);
  if (row < 0) return;  // whole workgroup leaves together
  bulk_body<NL, PULL, EPHI>(a, zl_begin, row, xb);
}

// one plane, a slab's first or last (not a plate): halos straight to / from the exchange buffers
template <int NL, bool PULL, bool EPHI>
__global__ void __launch_bounds__(64 * NL, EKPNP_BULK_MIN_WAVES) k_collide_edge(const KArgs a, const int zl, const int nrows, const int nxb, const int rchunk) {
  int xb;
  const int row = bulk_row_of_block(nrows, nxb, rchunk, xb);
  if (row < 0) return;
  bulk_body<NL, PULL, EPHI, true>(a, zl, row, xb);
}

// ------------------------------------------------------------------------------------------
// wall planes (global z = 0 and z = NZ-1): one thread per wall node, lattices in sequence.

// pre-collision populations of lattice L at a node, as gpu_stream would have left them
template <bool PULL>
__device__ __forceinline__ void gather(const KArgs& a, const int lat, const double* __restrict__ src, int x, int y, int zg, double (&f)[Q]) {
  const int xs[3] = {x + 1 == a.nx ? 0 : x + 1, x, x == 0 ? a.nx - 1 : x - 1};
  const int ys[3] = {y + 1 == a.ny ? 0 : y + 1, y, y == 0 ? a.ny - 1 : y - 1};
  static_for<0, Q, 1>([&](auto ic) {
    constexpr int d = decltype(ic)::value;
    constexpr int cx = PULL ? ex_of(d) : 0, cy = PULL ? ey_of(d) : 0, cz = PULL ? ez_of(d) : 0;
    int zs = zg - cz;
    if constexpr (cz != 0) {
      // gpu_stream's z wrap (LBM.cu:1972,1975): with zwrap the wall nodes read the opposite wall
      // plane directly instead of a ghost-plane copy of it
      if (a.zwrap) zs = zs == 0 ? a.nzl : (zs == a.nzl + 1 ? 1 : zs);
      // slabs: what would be a ghost plane is the neighbouring slab's edge plane, read straight out of the receive buffer
      const double* __restrict__ hb = zs == 0 ? a.halo_in_lo : (zs == a.nzl + 1 ? a.halo_in_hi : nullptr);
      if (hb) {
        f[d] = hb[((long long)lat * 9 + halo_slot(d)) * a.plane + (long long)ys[cy + 1] * a.nx + xs[cx + 1]];
        return;
      }
    }
    f[d] = src[((long long)zs * a.ny + ys[cy + 1]) * a.rowstride + slot_of(d) * 64 + pop_xoff(xs[cx + 1])];
  });
}

// pre-collision populations of a scalar lattice AT A WALL NODE: what gpu_bc_charge
// (LBM.cu:2095-2416) leaves in h1/hn1/temp1 (+temp0) after gpu_stream: the node's own
// post-collision populations swapped (ions) or negated-and-swapped plus 2 TH w (temperature).
template <bool PULL>
__device__ __forceinline__ void wall_scalar_pops(const KArgs& a, int lat, const double* __restrict__ src, int x, int y, int zg,
                                                 double TH_wall, double (&f)[Q]) {
  if constexpr (!PULL) {
    gather<false>(a, lat, src, x, y, zg, f);
  } else {
    const long long o = ((long long)zg * a.ny + y) * a.rowstride + pop_xoff(x);
    static_for<0, Q, 1>([&](auto ic) {
      constexpr int d = decltype(ic)::value;
      const double v = src[slot_of(opp_of(d)) * 64 + o];
      f[d] = (lat == 3) ? (-v + 2.0 * TH_wall * w_of(d)) : v;
    });
  }
}

// One wall node per LANE, one lattice per WAVE (round 4; rounds 1-3: one thread did the lattices in sequence at 255 VGPRs,
// one wave per SIMD, and gathered every scalar lattice twice - 0.44 ms for a 1024 x 1024 plate, half the HBM rate).  A
// workgroup is NL wave64s over the same 64 wall nodes of plate `top` (0 lower, 1 upper; uniform over the workgroup), like
// the bulk kernel: wave l gathers lattice l's pre-collision populations - and, on the lower plate, those of node z = 1 for
// the velocity override -, the waves exchange their moments through LDS, then wave 0 reflects the fluid and waves 1.. collide
// their scalar lattice out of the registers they already hold.  Same arithmetic per lattice in the same order: same bits.
// EPHI (see bulk_body): on a plate phi is constant, so Ex = Ey = +0 (k_phi_efield forms 0.5*(v - v)/d there), and Ez is
// gpu_bc's copy of the neighbouring interior plane's Ez (poisson.cu:57-69), formed from phi like that plane forms it.
template <int NL, bool PULL, bool EPHI>
__device__ __forceinline__ void wall_body(const KArgs& a, const int top, const int xraw, const int y) {
  __shared__ double wmom[NL > 1 ? 13 : 1][64];  // rho, j(3), c, cn, T of the wall node; j(3), c, cn, T of node z = 1
  const int lane = threadIdx.x & 63;
  const int lat = NL > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
  const bool act = xraw < a.nx;
  const int x = act ? xraw : a.nx - 1;
  const int zl = top ? a.nzl - 1 : 0;
  const int zg = zl + 1;
  const double TH_wall = top ? 0.0 : a.TH;  // LBM.cu:2226-2229 vs 2357-2412
  const long long sidx = ((long long)zl * a.ny + y) * (long long)a.nx + x;
  const long long orow = ((long long)zg * a.ny + y) * a.rowstride + pop_xoff(x);

  // this wave's lattice at the wall node
  double f[Q];
  if (lat == 0) gather<PULL>(a, 0, a.A[0], x, y, zg, f);
  else wall_scalar_pops<PULL>(a, lat, a.A[lat], x, y, zg, TH_wall, f);
  double rho = 0.0, jx = 0.0, jy = 0.0, jz = 0.0;
  double ms[3] = {0.0, 0.0, 0.0};              // c, cn, T
  double j1x = 0.0, j1y = 0.0, j1z = 0.0;      // node z = 1 (lower plate only)
  double m1[3] = {0.0, 0.0, 0.0};
  if constexpr (NL > 1) {
    if (lat == 0) {
      momentum(f, jx, jy, jz);
      wmom[0][lane] = sum27(f);
      wmom[1][lane] = jx;
      wmom[2][lane] = jy;
      wmom[3][lane] = jz;
    } else {
      wmom[3 + lat][lane] = sum27(f);
    }
    if (!top) {
      // z == 0 override, LBM.cu:663-801: node z=1's pre-collision populations
      double g[Q];
      gather<PULL>(a, lat, a.A[lat], x, y, zg + 1, g);
      if (lat == 0) {
        momentum(g, j1x, j1y, j1z);
        wmom[7][lane] = j1x;
        wmom[8][lane] = j1y;
        wmom[9][lane] = j1z;
      } else {
        wmom[9 + lat][lane] = sum27(g);
      }
    }
    __syncthreads();
    rho = wmom[0][lane];
    jx = wmom[1][lane];
    jy = wmom[2][lane];
    jz = wmom[3][lane];
    ms[0] = wmom[4][lane];
    ms[1] = wmom[5][lane];
    if constexpr (NL > 3) ms[2] = wmom[6][lane];
    if (!top) {
      j1x = wmom[7][lane];
      j1y = wmom[8][lane];
      j1z = wmom[9][lane];
      m1[0] = wmom[10][lane];
      m1[1] = wmom[11][lane];
      if constexpr (NL > 3) m1[2] = wmom[12][lane];
    }
  } else {
    rho = sum27(f);
    momentum(f, jx, jy, jz);
    if (!top) {
      double g[Q];
      gather<PULL>(a, 0, a.A[0], x, y, zg + 1, g);
      momentum(g, j1x, j1y, j1z);
    }
  }
  double Ex = 0.0, Ey = 0.0, Ez = 0.0;
  double E1x = 0.0, E1y = 0.0, E1z = 0.0;
  if constexpr (NL > 1) {
    if constexpr (EPHI) {
      const double* __restrict__ ph = a.fld[EKPNP_PHI];
      const long long oc = (long long)y * a.nx + x;
      // plane 1 / NZ-2: 0.5*(phi(z-1) - phi(z+1))/dz with the plate's pinned value on one side
      Ez = top ? 0.5 * (ph[(long long)(a.nzl - 3) * a.plane + oc] - a.voltage2) / a.dz : 0.5 * (a.voltage - ph[2 * a.plane + oc]) / a.dz;
      if (!top) {
        const double* __restrict__ p1 = ph + a.plane;  // plane 1
        const int xm = x == 0 ? a.nx - 1 : x - 1, xp = x + 1 == a.nx ? 0 : x + 1;
        const int ym = y == 0 ? a.ny - 1 : y - 1, yp = y + 1 == a.ny ? 0 : y + 1;
        E1x = 0.5 * (p1[(long long)y * a.nx + xm] - p1[(long long)y * a.nx + xp]) / a.dx;
        E1y = 0.5 * (p1[(long long)ym * a.nx + x] - p1[(long long)yp * a.nx + x]) / a.dy;
        E1z = Ez;  // Ez(0) is the copy of Ez(1)
      }
    } else {
      Ex = a.fld[EKPNP_EX][sidx];
      Ey = a.fld[EKPNP_EY][sidx];
      Ez = a.fld[EKPNP_EZ][sidx];
      if (!top) {
        const long long s1 = sidx + a.plane;
        E1x = a.fld[EKPNP_EX][s1];
        E1y = a.fld[EKPNP_EY][s1];
        E1z = a.fld[EKPNP_EZ][s1];
      }
    }
  }
  const double rhoinv = 1.0 / rho;
  const double hdt = a.dt * 0.5;
  double ux, uy, uz;
  if (!top) {
    // minus the velocity formula evaluated with node z=1's populations, field and moments, but with 1/rho of node z=0 (LBM.cu:780)
    const Force F1 = body_force(a, m1[0], m1[1], m1[2], E1x, E1y, E1z);
    ux = -velocity(rhoinv, j1x, a.cflinv, F1.x, hdt);
    uy = -velocity(rhoinv, j1y, a.cflinv, F1.y, hdt);
    uz = -velocity(rhoinv, j1z, a.cflinv, F1.z, hdt);
  } else {
    const Force F = body_force(a, ms[0], ms[1], ms[2], Ex, Ey, Ez);
    ux = velocity(rhoinv, jx, a.cflinv, F.x, hdt);
    uy = velocity(rhoinv, jy, a.cflinv, F.y, hdt);
    uz = velocity(rhoinv, jz, a.cflinv, F.z, hdt);
  }
  // a plate that is a slab's edge plane sends its outgoing populations (the wall-to-wall ghost loop of gpu_stream,
  // LBM.cu:1972,1975) like any edge plane: also straight into the send buffer (KArgs::halo_out_*)
  double* __restrict__ const hout = top ? a.halo_out_up : a.halo_out_dn;
  const long long hnode = (long long)y * a.nx + x;
  if (lat == 0) {
    if (act) {
      a.fld[EKPNP_RHO][sidx] = rho;
      a.fld[EKPNP_UX][sidx] = ux;
      a.fld[EKPNP_UY][sidx] = uy;
      a.fld[EKPNP_UZ][sidx] = uz;
      if (a.rhs) a.rhs[sidx] = 0.0;  // wall planes carry no Poisson unknown (poisson.cu:116-119,136-139)
    }
    // fluid: gpu_boundary (LBM.cu:1848-1961) discards the wall collision: f0 <- pre-collision f0,
    // f2[d] <- pre-collision f1[opp d] (+ moving-wall terms on the upper plate).
    double* __restrict__ dst = a.B[0];
    static_for<0, Q, 1>([&](auto ic) {
      constexpr int d = decltype(ic)::value;
      double v = f[opp_of(d)];
      if (top && d != 0) {
        constexpr int sgn = (ex_of(d) > 0 || d == 3) ? 1 : (ex_of(d) < 0 ? -1 : 0);  // LBM.cu:1902-1927
        if constexpr (sgn != 0) v = v + sgn * (a.uw_multi * w_of(d));
      }
      if (act) dst[slot_of(d) * 64 + orow] = v;
      if constexpr (ez_of(d) != 0) {
        if (hout && act && (ez_of(d) > 0) == (top != 0)) hout[(long long)halo_slot(d) * a.plane + hnode] = v;
      }
    });
  } else {
    // ions and temperature collide on the wall like anywhere else (their post-collision values
    // are what gpu_bc_charge reflects in the next step)
    if (act) a.fld[lat == 1 ? EKPNP_C : lat == 2 ? EKPNP_CN : EKPNP_T][sidx] = ms[lat - 1];
    double* __restrict__ dst = a.B[lat];
    auto store = [&](auto ic, double v) {
      constexpr int d = decltype(ic)::value;
      if (act) dst[slot_of(d) * 64 + orow] = v;
      if constexpr (ez_of(d) != 0) {
        if (hout && act && (ez_of(d) > 0) == (top != 0)) hout[((long long)lat * 9 + halo_slot(d)) * a.plane + hnode] = v;
      }
    };
    const double k = a.mob[lat];
    collide_scalar(a, f, ms[lat - 1], ux + k * Ex, uy + k * Ey, uz + k * Ez, a.wp[lat], a.wm[lat], store);
  }
}

template <int NL, bool PULL, bool EPHI>
__global__ void __launch_bounds__(64 * NL) k_collide_wall(const KArgs a, const int first_wall) {
  // one launch covers the walls this context owns: blockIdx.z = 0 is wall `first_wall`
  // (0 lower plate, 1 upper plate), blockIdx.z = 1 the upper plate
  wall_body<NL, PULL, EPHI>(a, first_wall + (int)blockIdx.z, (int)(blockIdx.x * 64 + (threadIdx.x & 63)), (int)blockIdx.y);
}

// Both faces of a slab in ONE launch (round 4): rows [0, ny) are the slab's first plane, rows [ny, 2 ny) its last one; each is
// a plate (wall_body) or an interior face (the bulk body with EDGE), decided per face - uniform over a workgroup, like the
// row.  The two faces may differ in where their new populations go (in-place slabs stage them) and in their halo buffers:
// two argument blocks, selected by the row.  Replaces up to two plate launches and two edge launches per step; the bodies
// are the ones of k_collide_wall / k_collide_edge, so every node gets the same bits.
template <int NL, bool PULL, bool EPHI>
__global__ void __launch_bounds__(64 * NL) k_collide_faces(const KArgs lo, const KArgs hi, const int lo_plate, const int hi_plate, const int nxb, const int rchunk) {
  int xb;
  const int ny = lo.ny;
  const int row = bulk_row_of_block(2 * ny, nxb, rchunk, xb);
  if (row < 0) return;
  // (the argument block is chosen by BRANCHING, not by a pointer: taking the address of a by-value kernel argument sends
  // the whole block through scratch memory - 1 KB per lane in the first build of this kernel)
  if (row < ny) {
    if (lo_plate) wall_body<NL, PULL, EPHI>(lo, 0, xb * 64 + (int)(threadIdx.x & 63), row);
    else bulk_body<NL, PULL, EPHI, true>(lo, 0, row, xb);
  } else {
    if (hi_plate) wall_body<NL, PULL, EPHI>(hi, 1, xb * 64 + (int)(threadIdx.x & 63), row - ny);
    else bulk_body<NL, PULL, EPHI, true>(hi, hi.nzl - 1, row - ny, xb);
  }
}

// Launch-bound lattices (the reference's own 50x8x51: wall planes 12 us, bulk 10 us, both pure latency):
// ONE launch for the whole lattice.  The rows beyond the bulk rows are the plates' rows (the row, hence the branch, is
// uniform over the workgroup, so every barrier is reached by whole workgroups).
template <int NL, bool PULL, bool EPHI>
__global__ void __launch_bounds__(64 * NL) k_collide_all(const KArgs a, const int zl_begin, const int nrows_bulk, const int nxb, const int rchunk) {
  int xb;
  const int row = bulk_row_of_block(nrows_bulk + 2 * a.ny, nxb, rchunk, xb);
  if (row < 0) return;
  if (row < nrows_bulk) {
    bulk_body<NL, PULL, EPHI>(a, zl_begin, row, xb);
  } else {
    const int w = row - nrows_bulk;  // [0, ny): lower plate, [ny, 2 ny): upper plate
    wall_body<NL, PULL, EPHI>(a, w / a.ny, xb * 64 + (int)(threadIdx.x & 63), w % a.ny);
  }
}

// ------------------------------------------------------------------------------------------
// z-periodic ghost loop of gpu_stream (LBM.cu:1972,1975) for a single slab, and the slab halo
// pack/unpack.  Buffers: [lattice][9 dirs][ny][nx].

// tiled offset of node (x, y) of plane zg for direction d = 0 (add d*64)
struct PopGeom {
  int nx, ny, nzl;
  long long rowstride, pplane;
  __device__ long long at(int zg, int y, int x) const { return (long long)zg * pplane + (long long)y * rowstride + pop_xoff(x); }
};

__global__ void k_ghost_wrap(double* p0, double* p1, double* p2, double* p3, int nl, PopGeom g) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, k = blockIdx.z;
  if (x >= g.nx) return;
  double* pp[MAXL] = {p0, p1, p2, p3};
  const int du = slot_of(up_dir(k)) * 64, dd = slot_of(dn_dir(k)) * 64;
  for (int l = 0; l < nl; ++l) {
    double* p = pp[l];
    p[g.at(0, y, x) + du] = p[g.at(g.nzl, y, x) + du];      // ghost below <- top plane
    p[g.at(g.nzl + 1, y, x) + dd] = p[g.at(1, y, x) + dd];  // ghost above <- bottom plane
  }
}

__global__ void k_halo_pack(const double* p0, const double* p1, const double* p2, const double* p3, int nl, PopGeom g, double* send_dn,
                            double* send_up) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, k = blockIdx.z;
  if (x >= g.nx) return;
  const long long plane = (long long)g.nx * g.ny, i = (long long)y * g.nx + x;
  const double* pp[MAXL] = {p0, p1, p2, p3};
  for (int l = 0; l < nl; ++l) {
    const double* p = pp[l];
    send_up[((long long)l * 9 + k) * plane + i] = p[g.at(g.nzl, y, x) + slot_of(up_dir(k)) * 64];
    send_dn[((long long)l * 9 + k) * plane + i] = p[g.at(1, y, x) + slot_of(dn_dir(k)) * 64];
  }
}

// In-place slabs: the slab's first and last plane are collided into a 2-plane staging buffer (two
// tiled planes per lattice; they must exist before the ordered sweep of the planes in between may
// start, so that the halo exchange can overlap it); the halo is packed from there and the two
// planes are copied into the lattice after the sweep.
__global__ void k_halo_pack_stage(const double* s0, const double* s1, const double* s2, const double* s3, int nl, PopGeom g,
                                  double* send_dn, double* send_up) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, k = blockIdx.z;
  if (x >= g.nx) return;
  const long long plane = (long long)g.nx * g.ny, i = (long long)y * g.nx + x;
  const double* ss[MAXL] = {s0, s1, s2, s3};
  for (int l = 0; l < nl; ++l) {
    send_up[((long long)l * 9 + k) * plane + i] = ss[l][g.at(1, y, x) + slot_of(up_dir(k)) * 64];  // last plane
    send_dn[((long long)l * 9 + k) * plane + i] = ss[l][g.at(0, y, x) + slot_of(dn_dir(k)) * 64];  // first plane
  }
}

// staging planes 0 / 1 -> planes zg = 1 / zg = nzl (whole tiled planes, pad lanes included)
__global__ void k_unstage(double* p0, double* p1, double* p2, double* p3, const double* s0, const double* s1, const double* s2,
                          const double* s3, int nl, long long pplane, int nzl) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pplane) return;
  double* pp[MAXL] = {p0, p1, p2, p3};
  const double* ss[MAXL] = {s0, s1, s2, s3};
  for (int l = 0; l < nl; ++l) {
    pp[l][pplane + i] = ss[l][i];
    pp[l][(long long)nzl * pplane + i] = ss[l][pplane + i];
  }
}

__global__ void k_halo_unpack(double* p0, double* p1, double* p2, double* p3, int nl, PopGeom g, const double* recv_lo,
                              const double* recv_hi) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, k = blockIdx.z;
  if (x >= g.nx) return;
  const long long plane = (long long)g.nx * g.ny, i = (long long)y * g.nx + x;
  double* pp[MAXL] = {p0, p1, p2, p3};
  for (int l = 0; l < nl; ++l) {
    double* p = pp[l];
    p[g.at(0, y, x) + slot_of(up_dir(k)) * 64] = recv_lo[((long long)l * 9 + k) * plane + i];
    p[g.at(g.nzl + 1, y, x) + slot_of(dn_dir(k)) * 64] = recv_hi[((long long)l * 9 + k) * plane + i];
  }
}

// ------------------------------------------------------------------------------------------
// initial state: gpu_initialization (LBM.cu:111-128), gpu_PBE (139-146), gpu_PBE_phi (131-137),
// gpu_init_equilibrium (162-463)

__global__ void k_init_fields(KArgs a, double voltage, double Lz, double dz) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n = (long long)a.nzl * a.plane;
  if (i >= n) return;
  const int z = a.z0 + (int)(i / a.plane);
  a.fld[EKPNP_RHO][i] = a.rho0;
  a.fld[EKPNP_C][i] = 0.0;
  a.fld[EKPNP_CN][i] = 0.0;
  a.fld[EKPNP_PHI][i] = voltage;
  a.fld[EKPNP_UX][i] = 0.0;
  a.fld[EKPNP_UY][i] = 0.0;
  a.fld[EKPNP_UZ][i] = 0.0;
  a.fld[EKPNP_EX][i] = 0.0;
  a.fld[EKPNP_EY][i] = 0.0;
  a.fld[EKPNP_EZ][i] = 0.0;
  a.fld[EKPNP_T][i] = a.TH * (Lz - dz * z) / Lz;
}

__global__ void k_pbe(double* c, double* cn, const double* phi, long long n, double chargeinf, double electron, double kB, double roomT) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double fi = phi[i];
  c[i] = chargeinf * exp(-electron * fi / kB / roomT);
  cn[i] = chargeinf * exp(electron * fi / kB / roomT);
}

// phi <- omega*phi + (1-omega)*phi_old; phi_old <- phi   (LBM.cu:98-104 without the host round trip)
__global__ void k_pbe_relax(double* phi, double* phi_old, long long n, double omega) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = omega * phi[i] + (1.0 - omega) * phi_old[i];
  phi[i] = v;
  phi_old[i] = v;
}

template <int NL>
__global__ void k_init_equilibrium(KArgs a) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n = (long long)a.nzl * a.plane;
  if (i >= n) return;
  const int x = (int)(i % a.nx), y = (int)((i / a.nx) % a.ny), zg = (int)(i / a.plane) + 1;  // ghost plane below
  const long long o = ((long long)zg * a.ny + y) * a.rowstride + pop_xoff(x);
  const double rho = a.fld[EKPNP_RHO][i], ux = a.fld[EKPNP_UX][i], uy = a.fld[EKPNP_UY][i], uz = a.fld[EKPNP_UZ][i];
  const double Ex = a.fld[EKPNP_EX][i], Ey = a.fld[EKPNP_EY][i], Ez = a.fld[EKPNP_EZ][i];
  double eq[Q];
  static_for<0, NL, 1>([&](auto lc) {
    constexpr int lat = decltype(lc)::value;
    const double m = lat == 0 ? rho : a.fld[lat == 1 ? EKPNP_C : lat == 2 ? EKPNP_CN : EKPNP_T][i];
    const double k = a.mob[lat];
    equilibrium(a, m, ux + k * Ex, uy + k * Ey, uz + k * Ez, eq);
    double* dst = a.B[lat];
#pragma unroll
    for (int d = 0; d < Q; ++d) dst[slot_of(d) * 64 + o] = eq[d];
  });
}

// ------------------------------------------------------------------------------------------
// host-side launchers

static inline dim3 grid1d(long long n, int b) { return dim3((unsigned)((n + b - 1) / b)); }

// the collide forms E from phi (EPHI kernels) whenever phi is known to be what E derives from AND nobody outside can
// have touched the E arrays since; both sources hold the same bits when both are valid
static inline bool collide_takes_e_from_phi(const Ctx& c) { return c.p.n_lattices > 1 && c.e_phi_valid && lazy_efield_ok(c); }

void launch_init_fields(Ctx& c) {
  KArgs a = c.kargs();
  hipLaunchKernelGGL(k_init_fields, grid1d((long long)c.nloc, 256), dim3(256), 0, c.stream, a, c.p.voltage, c.p.Lz, c.p.dz);
  note_launch(c, "k_init_fields");
}

void launch_pbe(Ctx& c) {
  hipLaunchKernelGGL(k_pbe, grid1d((long long)c.nloc, 256), dim3(256), 0, c.stream, c.fld[EKPNP_C], c.fld[EKPNP_CN], c.fld[EKPNP_PHI],
                     (long long)c.nloc, c.p.chargeinf, c.p.electron, c.p.kB, c.p.roomT);
  note_launch(c, "k_pbe");
}

void launch_pbe_relax(Ctx& c, double* phi_old, double omega) {
  hipLaunchKernelGGL(k_pbe_relax, grid1d((long long)c.nloc, 256), dim3(256), 0, c.stream, c.fld[EKPNP_PHI], phi_old, (long long)c.nloc,
                     omega);
  note_launch(c, "k_pbe_relax");
}

void launch_init_equilibrium(Ctx& c) {
  KArgs a = c.kargs();
  // write into the CURRENT buffer: kargs() exposes it as A (const); B is the other one
  for (int l = 0; l < MAXL; ++l) a.B[l] = c.cur_base(l);
  dim3 g = grid1d((long long)c.nloc, 128), b(128);
  switch (c.p.n_lattices) {
    case 1: hipLaunchKernelGGL(k_init_equilibrium<1>, g, b, 0, c.stream, a); note_launch(c, "k_init_equilibrium<1>"); break;
    case 3: hipLaunchKernelGGL(k_init_equilibrium<3>, g, b, 0, c.stream, a); note_launch(c, "k_init_equilibrium<3>"); break;
    default: hipLaunchKernelGGL(k_init_equilibrium<4>, g, b, 0, c.stream, a); note_launch(c, "k_init_equilibrium<4>"); break;
  }
}

// rows per band of the interior sweep in effect (0: plane after plane): Ctx::bulk_yband, or the rule above bulk_dispatch's launch
int bulk_band_rows(const Ctx& c, int rchunk) {
  int yband = c.bulk_yband;
  if (yband < 0) yband = (size_t)c.p.nx * c.p.ny * (size_t)(c.p.n_lattices * 27 * 16 + 80) > ((size_t)192 << 20) ? 128 : 0;
  return yband > 0 && yband < 0x7fff && yband < c.p.ny && c.p.ny % yband == 0 && yband % rchunk == 0 ? yband : 0;
}

template <int NL>
static void bulk_dispatch(Ctx& c, const KArgs& a, int zl_begin, int zl_end) {
  const int nrows = (zl_end - zl_begin) * c.p.ny;
  if (nrows <= 0) return;
  const int nxb = (c.p.nx + 63) / 64;
  static const int rchunk_env = std::getenv("EKPNP_BULK_RCHUNK") ? std::atoi(std::getenv("EKPNP_BULK_RCHUNK")) : 64;  // tuning knob
  int rchunk = rchunk_env < 1 ? 1 : (rchunk_env > 0xffff ? 0xffff : rchunk_env);
  // rows per XCD, rounded up to whole runs: the 8 XCDs together cover [0, 8*per_xcd) >= nrows
  const long long per_xcd = ((long long)nrows + 8LL * rchunk - 1) / (8LL * rchunk) * rchunk;
  dim3 g((unsigned)(8 * per_xcd * nxb)), b(64 * NL);
  // y bands (bulk_row_of_block; in-place contexts: inside each of the sweep's launches of `zchunk` planes - a launch writes only
  // where no plane it reads lies, whatever the order of its workgroups).  The
  // sweep of one PLANE moves nx ny x 1 808 B = 474 MB on 512 x 512 planes, 1.9 GB on 1024 x 1024 ones, so the phi row a
  // workgroup reads as z + 1 has left the 256 MiB Infinity Cache long before it comes back as z and z - 1.  Taken in bands of
  // 128 rows the re-reads find it there: bulk kernel 38.68 -> 38.40 ms on cfg3, 38.90 -> 38.43 ms on cfg5's 1024 x 1024 x 128
  // slab (bands of 64 and 256 rows gain about half of that; profiles/r05c_ab_bulk_yband*.jsonl).  Same arithmetic per node,
  // another order of the workgroups: same bits.  Default (-1): bands of 128 rows where a plane's sweep moves more than
  // 192 MiB and NY is a multiple of 128; smaller planes (256 x 256: 90 - 118 MB) are within the cache's reach as they are.
  rchunk |= bulk_band_rows(c, rchunk) << 16;
  const bool ephi = collide_takes_e_from_phi(c);
  if (c.streamed_state) {
    if (ephi) hipLaunchKernelGGL((k_collide_bulk<NL, false, (NL > 1)>), g, b, 0, c.stream, a, zl_begin, nrows, nxb, rchunk);
    else hipLaunchKernelGGL((k_collide_bulk<NL, false, false>), g, b, 0, c.stream, a, zl_begin, nrows, nxb, rchunk);
  } else {
    if (ephi) hipLaunchKernelGGL((k_collide_bulk<NL, true, (NL > 1)>), g, b, 0, c.stream, a, zl_begin, nrows, nxb, rchunk);
    else hipLaunchKernelGGL((k_collide_bulk<NL, true, false>), g, b, 0, c.stream, a, zl_begin, nrows, nxb, rchunk);
  }
  note_launch(c, "k_collide_bulk");
}

template <int NL>
static void edge_dispatch(Ctx& c, const KArgs& a, int zl) {
  const int nrows = c.p.ny, nxb = (c.p.nx + 63) / 64, rchunk = 64;
  const long long per_xcd = ((long long)nrows + 8LL * rchunk - 1) / (8LL * rchunk) * rchunk;
  dim3 g((unsigned)(8 * per_xcd * nxb)), b(64 * NL);
  const bool ephi = collide_takes_e_from_phi(c);
  if (c.streamed_state) {
    if (ephi) hipLaunchKernelGGL((k_collide_edge<NL, false, (NL > 1)>), g, b, 0, c.stream, a, zl, nrows, nxb, rchunk);
    else hipLaunchKernelGGL((k_collide_edge<NL, false, false>), g, b, 0, c.stream, a, zl, nrows, nxb, rchunk);
  } else {
    if (ephi) hipLaunchKernelGGL((k_collide_edge<NL, true, (NL > 1)>), g, b, 0, c.stream, a, zl, nrows, nxb, rchunk);
    else hipLaunchKernelGGL((k_collide_edge<NL, true, false>), g, b, 0, c.stream, a, zl, nrows, nxb, rchunk);
  }
  note_launch(c, "k_collide_edge");
}

template <int NL>
static void faces_dispatch(Ctx& c, const KArgs& lo, const KArgs& hi, int lo_plate, int hi_plate) {
  const int nrows = 2 * c.p.ny, nxb = (c.p.nx + 63) / 64, rchunk = 64;
  const long long per_xcd = ((long long)nrows + 8LL * rchunk - 1) / (8LL * rchunk) * rchunk;
  dim3 g((unsigned)(8 * per_xcd * nxb)), b(64 * NL);
  const bool ephi = collide_takes_e_from_phi(c);
  if (c.streamed_state) {
    if (ephi) hipLaunchKernelGGL((k_collide_faces<NL, false, (NL > 1)>), g, b, 0, c.stream, lo, hi, lo_plate, hi_plate, nxb, rchunk);
    else hipLaunchKernelGGL((k_collide_faces<NL, false, false>), g, b, 0, c.stream, lo, hi, lo_plate, hi_plate, nxb, rchunk);
  } else {
    if (ephi) hipLaunchKernelGGL((k_collide_faces<NL, true, (NL > 1)>), g, b, 0, c.stream, lo, hi, lo_plate, hi_plate, nxb, rchunk);
    else hipLaunchKernelGGL((k_collide_faces<NL, true, false>), g, b, 0, c.stream, lo, hi, lo_plate, hi_plate, nxb, rchunk);
  }
  note_launch(c, "k_collide_faces");
}

// a slab's first and last plane in one launch (plates and / or interior faces; KArgs::halo_* honoured)
void launch_collide_faces(Ctx& c, const KArgs& lo, const KArgs& hi) {
  const int lo_plate = c.z0 == 0, hi_plate = c.z0 + c.nzl == c.p.nz;
  switch (c.p.n_lattices) {
    case 1: faces_dispatch<1>(c, lo, hi, lo_plate, hi_plate); break;
    case 3: faces_dispatch<3>(c, lo, hi, lo_plate, hi_plate); break;
    default: faces_dispatch<4>(c, lo, hi, lo_plate, hi_plate); break;
  }
}

void launch_collide_bulk_edge(Ctx& c, const KArgs& a, int zl) {
  switch (c.p.n_lattices) {
    case 1: edge_dispatch<1>(c, a, zl); break;
    case 3: edge_dispatch<3>(c, a, zl); break;
    default: edge_dispatch<4>(c, a, zl); break;
  }
}

template <int NL>
static void all_dispatch(Ctx& c, const KArgs& a) {
  const int nrows_bulk = (c.nzl - 2) * c.p.ny, nrows = nrows_bulk + 2 * c.p.ny;
  const int nxb = (c.p.nx + 63) / 64, rchunk = 64;
  const long long per_xcd = ((long long)nrows + 8LL * rchunk - 1) / (8LL * rchunk) * rchunk;
  dim3 g((unsigned)(8 * per_xcd * nxb)), b(64 * NL);
  const bool ephi = collide_takes_e_from_phi(c);
  if (c.streamed_state) {
    if (ephi) hipLaunchKernelGGL((k_collide_all<NL, false, (NL > 1)>), g, b, 0, c.stream, a, 1, nrows_bulk, nxb, rchunk);
    else hipLaunchKernelGGL((k_collide_all<NL, false, false>), g, b, 0, c.stream, a, 1, nrows_bulk, nxb, rchunk);
  } else {
    if (ephi) hipLaunchKernelGGL((k_collide_all<NL, true, (NL > 1)>), g, b, 0, c.stream, a, 1, nrows_bulk, nxb, rchunk);
    else hipLaunchKernelGGL((k_collide_all<NL, true, false>), g, b, 0, c.stream, a, 1, nrows_bulk, nxb, rchunk);
  }
  note_launch(c, "k_collide_all");
}

// the whole lattice of a single two-buffer context (both plates and everything in between) in one launch
void launch_collide_all(Ctx& c) {
  const KArgs a = c.kargs();
  switch (c.p.n_lattices) {
    case 1: all_dispatch<1>(c, a); break;
    case 3: all_dispatch<3>(c, a); break;
    default: all_dispatch<4>(c, a); break;
  }
}

void launch_collide_bulk(Ctx& c, int zl_begin, int zl_end) { launch_collide_bulk(c, c.kargs(), zl_begin, zl_end); }

void launch_collide_bulk(Ctx& c, const KArgs& a, int zl_begin, int zl_end) {
  switch (c.p.n_lattices) {
    case 1: bulk_dispatch<1>(c, a, zl_begin, zl_end); break;
    case 3: bulk_dispatch<3>(c, a, zl_begin, zl_end); break;
    default: bulk_dispatch<4>(c, a, zl_begin, zl_end); break;
  }
}

template <int NL>
static void wall_dispatch(Ctx& c, const KArgs& a, int first_wall, int nwalls, hipStream_t stream) {
  dim3 g((unsigned)((c.p.nx + 63) / 64), (unsigned)c.p.ny, (unsigned)nwalls), b(64 * NL);
  const bool ephi = collide_takes_e_from_phi(c);
  if (c.streamed_state) {
    if (ephi) hipLaunchKernelGGL((k_collide_wall<NL, false, (NL > 1)>), g, b, 0, stream, a, first_wall);
    else hipLaunchKernelGGL((k_collide_wall<NL, false, false>), g, b, 0, stream, a, first_wall);
  } else {
    if (ephi) hipLaunchKernelGGL((k_collide_wall<NL, true, (NL > 1)>), g, b, 0, stream, a, first_wall);
    else hipLaunchKernelGGL((k_collide_wall<NL, true, false>), g, b, 0, stream, a, first_wall);
  }
  note_launch(c, "k_collide_wall");
}

void launch_collide_walls(Ctx& c, hipStream_t stream, bool want_lower, bool want_upper) {
  launch_collide_walls(c, c.kargs(), stream, want_lower, want_upper);
}

void launch_collide_walls(Ctx& c, const KArgs& a, hipStream_t stream, bool want_lower, bool want_upper) {
  const bool lower = want_lower && c.z0 == 0, upper = want_upper && c.z0 + c.nzl == c.p.nz;
  if (!lower && !upper) return;
  const int first_wall = lower ? 0 : 1, nwalls = (lower && upper) ? 2 : 1;
  switch (c.p.n_lattices) {
    case 1: wall_dispatch<1>(c, a, first_wall, nwalls, stream); break;
    case 3: wall_dispatch<3>(c, a, first_wall, nwalls, stream); break;
    default: wall_dispatch<4>(c, a, first_wall, nwalls, stream); break;
  }
}

static PopGeom pop_geom(const Ctx& c) { return PopGeom{c.p.nx, c.p.ny, c.nzl, (long long)c.rowstride, (long long)c.pplane}; }
static dim3 halo_grid(const Ctx& c) { return dim3((unsigned)((c.p.nx + 255) / 256), (unsigned)c.p.ny, 9); }

void launch_ghost_wrap(Ctx& c) {
  double* p[MAXL] = {c.cur_base(0), c.cur_base(1), c.cur_base(2), c.cur_base(3)};
  hipLaunchKernelGGL(k_ghost_wrap, halo_grid(c), dim3(256), 0, c.stream, p[0], p[1], p[2], p[3], c.p.n_lattices, pop_geom(c));
  note_launch(c, "k_ghost_wrap");
}

void launch_halo_pack(Ctx& c, int buffer) {
  double** p = c.pop[buffer];
  hipLaunchKernelGGL(k_halo_pack, halo_grid(c), dim3(256), 0, c.stream, p[0], p[1], p[2], p[3], c.p.n_lattices, pop_geom(c), c.halo[0],
                     c.halo[1]);
  note_launch(c, "k_halo_pack");
}

void launch_halo_unpack(Ctx& c) {
  double* p[MAXL] = {c.cur_base(0), c.cur_base(1), c.cur_base(2), c.cur_base(3)};
  hipLaunchKernelGGL(k_halo_unpack, halo_grid(c), dim3(256), 0, c.stream, p[0], p[1], p[2], p[3], c.p.n_lattices, pop_geom(c), c.halo[2],
                     c.halo[3]);
  note_launch(c, "k_halo_unpack");
}

void launch_halo_pack_stage(Ctx& c) {
  hipLaunchKernelGGL(k_halo_pack_stage, halo_grid(c), dim3(256), 0, c.stream, c.stage[0], c.stage[1], c.stage[2], c.stage[3],
                     c.p.n_lattices, pop_geom(c), c.halo[0], c.halo[1]);
  note_launch(c, "k_halo_pack_stage");
}

void launch_unstage(Ctx& c) {
  double* p[MAXL] = {c.cur_base(0), c.cur_base(1), c.cur_base(2), c.cur_base(3)};
  dim3 g((unsigned)((c.pplane + 255) / 256)), b(256);
  hipLaunchKernelGGL(k_unstage, g, b, 0, c.stream, p[0], p[1], p[2], p[3], c.stage[0], c.stage[1], c.stage[2], c.stage[3], c.p.n_lattices,
                     (long long)c.pplane, c.nzl);
  note_launch(c, "k_unstage");
}

}  // namespace ekpnp
