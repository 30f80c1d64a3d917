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
#include "ekpnp_internal.h"

namespace ekpnp {

// rhs of the interior planes, poisson.cu:114-135: g = -F (c - cn)/eps, wall potentials folded
// into the planes next to the walls; wall planes themselves carry 0.
__global__ void k_poisson_rhs(PArgs a) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n = (long long)a.nzl * a.plane;
  if (i >= n) return;
  const int z = a.z0 + (int)(i / a.plane);
  double v = 0.0;
  if (z > 0 && z < a.nz - 1) {
    v = -a.F * (a.fld[EKPNP_C][i] - a.fld[EKPNP_CN][i]) / a.eps;
    if (z == 1) v = v - a.voltage * a.inv_dz2;
    if (z == a.nz - 2) v = v - a.voltage2 * a.inv_dz2;
  }
  a.work[i] = v;
}

// Thomas factorisation table of  phi[z-1] - (2 + kappa) phi[z] + phi[z+1] = dz^2 g[z],
// kappa = dz^2 (kx^2 + ky^2), for interior planes z = 1 .. nz-2 of the GLOBAL lattice:
// c'[1] = 1/b, c'[z] = 1/(b - c'[z-1]), b = -(2 + kappa).  Layout [nz][ny][nxh].
__global__ void k_build_cprime(double* cprime, int nx, int ny, int nz, int nxh, double Lx, double Ly, double dz) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= ny * nxh) return;
  const int ix = m % nxh, iy = m / nxh;
  // wavenumbers exactly as main.cu:119-136
  const double kx = (double)ix * 2.0 * M_PI / Lx;
  const double ky = (iy <= ny / 2) ? (double)iy * 2.0 * M_PI / Ly : ((double)iy - ny) * 2.0 * M_PI / Ly;
  const double b = -(2.0 + dz * dz * (kx * kx + ky * ky));
  const long long ms = (long long)ny * nxh;
  double cp = 0.0;
  cprime[m] = 0.0;
  for (int z = 1; z <= nz - 2; ++z) {
    cp = 1.0 / (b - cp);
    cprime[(long long)z * ms + m] = cp;
  }
  cprime[(long long)(nz - 1) * ms + m] = 0.0;
}

// Forward elimination + back substitution for one (kx,ky) mode, in place on the spectrum.
// Single-slab version (the whole z extent is local).
__global__ void k_tridiag(PArgs a) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  const long long ms = (long long)a.ny * a.nxh;
  if (m >= ms) return;
  const double dz2 = a.dz * a.dz;
  double2* s = a.spec + m;
  const double* cp = a.cprime + m;
  double dr = 0.0, di = 0.0;
  const int n = a.nz;
#pragma unroll 4
  for (int z = 1; z <= n - 2; ++z) {
    const double2 r = s[(long long)z * ms];
    const double c = cp[(long long)z * ms];
    dr = (dz2 * r.x - dr) * c;
    di = (dz2 * r.y - di) * c;
    s[(long long)z * ms] = make_double2(dr, di);
  }
  // phi[n-2] = d'[n-2]; phi[z] = d'[z] - c'[z] phi[z+1]
  double pr = dr, pi = di;
#pragma unroll 4
  for (int z = n - 3; z >= 1; --z) {
    const double2 d = s[(long long)z * ms];
    const double c = cp[(long long)z * ms];
    pr = d.x - c * pr;
    pi = d.y - c * pi;
    s[(long long)z * ms] = make_double2(pr, pi);
  }
}

// odd_extract + gpu_efield + gpu_bc fused (poisson.cu:191-204, 40-69): phi = ifft/(NX NY) on
// interior planes, wall planes pinned to voltage/voltage2; E = central differences of phi,
// periodic in x,y; Ez of a wall plane copies the neighbouring interior plane.
__device__ __forceinline__ double phi_at(const PArgs& a, int x, int y, int z /*global*/) {
  if (z <= 0) return a.voltage;  // z==0 wall (z==-1 is never used for a result that survives gpu_bc)
  if (z >= a.nz - 1) return a.voltage2;
  const int zl = z - a.z0;
  if (zl < 0) return a.phi_lo[(long long)y * a.nx + x];
  if (zl >= a.nzl) return a.phi_hi[(long long)y * a.nx + x];
  return a.work[((long long)zl * a.ny + y) * a.nx + x] * a.inv_nxny;
}

__global__ void k_phi_efield(PArgs a) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= a.nx) return;
  const int y = blockIdx.y;
  const int zl = blockIdx.z;
  const int z = a.z0 + zl;
  const long long i = ((long long)zl * a.ny + y) * a.nx + x;
  const int xp1 = x + 1 == a.nx ? 0 : x + 1, xm1 = x == 0 ? a.nx - 1 : x - 1;
  const int yp1 = y + 1 == a.ny ? 0 : y + 1, ym1 = y == 0 ? a.ny - 1 : y - 1;
  a.fld[EKPNP_PHI][i] = phi_at(a, x, y, z);
  a.fld[EKPNP_EX][i] = 0.5 * (phi_at(a, xm1, y, z) - phi_at(a, xp1, y, z)) / a.dx;
  a.fld[EKPNP_EY][i] = 0.5 * (phi_at(a, x, ym1, z) - phi_at(a, x, yp1, z)) / a.dy;
  const int ze = z == 0 ? 1 : (z == a.nz - 1 ? a.nz - 2 : z);  // gpu_bc: wall Ez <- neighbour's Ez
  a.fld[EKPNP_EZ][i] = 0.5 * (phi_at(a, x, y, ze - 1) - phi_at(a, x, y, ze + 1)) / a.dz;
}

void build_cprime(Ctx& c) {
  const int nm = c.p.ny * c.nxh;
  hipLaunchKernelGGL(k_build_cprime, dim3((nm + 127) / 128), dim3(128), 0, c.stream, c.cprime, c.p.nx, c.p.ny, c.p.nz, c.nxh, c.p.Lx,
                     c.p.Ly, c.p.dz);
}

void launch_poisson_rhs(Ctx& c) {
  PArgs a = c.pargs();
  hipLaunchKernelGGL(k_poisson_rhs, dim3((unsigned)((c.nloc + 255) / 256)), dim3(256), 0, c.stream, a);
}

void launch_tridiag(Ctx& c) {
  PArgs a = c.pargs();
  const int nm = c.p.ny * c.nxh;
  hipLaunchKernelGGL(k_tridiag, dim3((nm + 63) / 64), dim3(64), 0, c.stream, a);
}

void launch_phi_efield(Ctx& c) {
  PArgs a = c.pargs();
  const int bx = c.p.nx >= 256 ? 256 : 64;
  hipLaunchKernelGGL(k_phi_efield, dim3((c.p.nx + bx - 1) / bx, c.p.ny, c.nzl), dim3(bx), 0, c.stream, a);
}

}  // namespace ekpnp
