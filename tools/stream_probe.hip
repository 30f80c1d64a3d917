// stream_probe.hip — what streaming rate can the bulk kernel's access pattern reach on this GPU?
// Variants of a pure copy (no arithmetic): 8 vs 16 bytes per lane, aligned vs shifted by one
// double, plain vs non-temporal stores, and the LBM shape: 27 direction streams read and 27
// written per workgroup with the D3Q27 pull offsets.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k8(const double* __restrict__ a, double* __restrict__ b, size_t n, int sh) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) b[i] = a[i + sh];
}
__global__ void k8nt(const double* __restrict__ a, double* __restrict__ b, size_t n, int sh) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load(a + i + sh), b + i);
}
__global__ void k16(const double* __restrict__ a, double* __restrict__ b, size_t n, int sh) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (i + 1 < n) {
    double v0, v1;
    if (sh == 0) { double2 v = *(const double2*)(a + i); v0 = v.x; v1 = v.y; }
    else { v0 = a[i + sh]; v1 = a[i + sh + 1]; }
    *(double2*)(b + i) = make_double2(v0, v1);
  }
}
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ void k16u(const double* __restrict__ a, double* __restrict__ b, size_t n, int sh) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (i + 1 < n) {
    d2 v;
    __builtin_memcpy(&v, a + i + sh, 16);  // unaligned 16-byte load
    *(d2*)(b + i) = v;
  }
}
// LBM shape: nd direction streams, plane-strided, pull offsets in x (+-1), y and z rows
__global__ void klbm(const double* __restrict__ a, double* __restrict__ b, int nx, int ny, int nz, long long dstride, int nd, int nt) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, z = blockIdx.z + 1;
  if (x >= nx) return;
  const int xm = x == 0 ? nx - 1 : x - 1, xp = x + 1 == nx ? 0 : x + 1;
  const int ym = y == 0 ? ny - 1 : y - 1, yp = y + 1 == ny ? 0 : y + 1;
  const long long o = ((long long)z * ny + y) * nx + x;
  double acc[27];
#pragma unroll
  for (int d = 0; d < 27; ++d) {
    if (d < nd) {
      const int cx = (d % 3) - 1, cy = ((d / 3) % 3) - 1, cz = (d / 9) - 1;
      const int xs = cx < 0 ? xp : cx > 0 ? xm : x, ys = cy < 0 ? yp : cy > 0 ? ym : y;
      acc[d] = a[(long long)d * dstride + ((long long)(z - cz) * ny + ys) * nx + xs];
    }
  }
#pragma unroll
  for (int d = 0; d < 27; ++d)
    if (d < nd) {
      if (nt) __builtin_nontemporal_store(acc[d], b + (long long)d * dstride + o);
      else b[(long long)d * dstride + o] = acc[d];
    }
}

// LBM shape with the bulk kernel's XCD-chunked row mapping (1-D grid), nl lattices = nl waves per
// block on separate arrays (a + l*lstride), like k_collide_bulk
__global__ void klbm_chunk(const double* __restrict__ a, double* __restrict__ b, int nx, int ny, int nz, long long dstride, long long lstride,
                           int nxb, int rchunk, int nt) {
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int r = slot / nxb, xb = slot - r * nxb;
  const int row = ((r / rchunk) * 8 + xcd) * rchunk + r % rchunk;
  if (row >= ny * nz) return;
  const int lat = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int x = xb * 64 + lane, y = row % ny, z = row / ny + 1;
  const double* aa = a + lat * lstride;
  double* bb = b + lat * lstride;
  const int xm = x == 0 ? nx - 1 : x - 1, xp = x + 1 == nx ? 0 : x + 1;
  const int ym = y == 0 ? ny - 1 : y - 1, yp = y + 1 == ny ? 0 : y + 1;
  const long long o = ((long long)z * ny + y) * nx + x;
  double acc[27];
#pragma unroll
  for (int d = 0; d < 27; ++d) {
    const int cx = (d % 3) - 1, cy = ((d / 3) % 3) - 1, cz = (d / 9) - 1;
    const int xs = cx < 0 ? xp : cx > 0 ? xm : x, ys = cy < 0 ? yp : cy > 0 ? ym : y;
    acc[d] = aa[(long long)d * dstride + ((long long)(z - cz) * ny + ys) * nx + xs];
  }
#pragma unroll
  for (int d = 0; d < 27; ++d) {
    if (nt) __builtin_nontemporal_store(acc[d], bb + (long long)d * dstride + o);
    else bb[(long long)d * dstride + o] = acc[d];
  }
}

// as klbm_chunk, but every thread moves RPT consecutive y rows (more bytes per stream per visit)
template <int RPT>
__global__ void klbm_rpt(const double* __restrict__ a, double* __restrict__ b, int nx, int ny, int nz, long long dstride, long long lstride,
                         int nxb, int rchunk) {
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int r = slot / nxb, xb = slot - r * nxb;
  const int row0 = (((r / rchunk) * 8 + xcd) * rchunk + r % rchunk) * RPT;
  if (row0 >= ny * nz) return;
  const int lat = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int x = xb * 64 + lane;
  const double* aa = a + lat * lstride;
  double* bb = b + lat * lstride;
  const int xm = x == 0 ? nx - 1 : x - 1, xp = x + 1 == nx ? 0 : x + 1;
  double acc[RPT][27];
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int row = row0 + q, y = row % ny, z = row / ny + 1;
    const int ym = y == 0 ? ny - 1 : y - 1, yp = y + 1 == ny ? 0 : y + 1;
#pragma unroll
    for (int d = 0; d < 27; ++d) {
      const int cx = (d % 3) - 1, cy = ((d / 3) % 3) - 1, cz = (d / 9) - 1;
      const int xs = cx < 0 ? xp : cx > 0 ? xm : x, ys = cy < 0 ? yp : cy > 0 ? ym : y;
      acc[q][d] = aa[(long long)d * dstride + ((long long)(z - cz) * ny + ys) * nx + xs];
    }
  }
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int row = row0 + q, y = row % ny, z = row / ny + 1;
    const long long o = ((long long)z * ny + y) * nx + x;
#pragma unroll
    for (int d = 0; d < 27; ++d) bb[(long long)d * dstride + o] = acc[q][d];
  }
}

// lattice-interleaved layout [d][z][y][x][4]: one 32-byte element holds the four lattices'
// population of direction d at a node; wave w of the block owns a quarter of the directions.
struct __attribute__((aligned(32))) quad { double v[4]; };
__global__ void klbm_inter(const quad* __restrict__ a, quad* __restrict__ b, int nx, int ny, int nz, long long dstride, int nxb, int rchunk, int nt) {
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int r = slot / nxb, xb = slot - r * nxb;
  const int row = ((r / rchunk) * 8 + xcd) * rchunk + r % rchunk;
  if (row >= ny * nz) return;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int x = xb * 64 + lane, y = row % ny, z = row / ny + 1;
  const int xm = x == 0 ? nx - 1 : x - 1, xp = x + 1 == nx ? 0 : x + 1;
  const int ym = y == 0 ? ny - 1 : y - 1, yp = y + 1 == ny ? 0 : y + 1;
  const long long o = ((long long)z * ny + y) * nx + x;
  quad acc[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int d = wv * 7 + k;
    if (d < 27) {
      const int cx = (d % 3) - 1, cy = ((d / 3) % 3) - 1, cz = (d / 9) - 1;
      const int xs = cx < 0 ? xp : cx > 0 ? xm : x, ys = cy < 0 ? yp : cy > 0 ? ym : y;
      acc[k] = a[(long long)d * dstride + ((long long)(z - cz) * ny + ys) * nx + xs];
    }
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int d = wv * 7 + k;
    if (d < 27) {
      if (nt) {
        double* q = (double*)(b + (long long)d * dstride + o);
        __builtin_nontemporal_store(acc[k].v[0], q); __builtin_nontemporal_store(acc[k].v[1], q + 1);
        __builtin_nontemporal_store(acc[k].v[2], q + 2); __builtin_nontemporal_store(acc[k].v[3], q + 3);
      } else b[(long long)d * dstride + o] = acc[k];
    }
  }
}

// the same traffic with a tiled (AoSoA) layout: [z][y][x/64][27][64]
__global__ void klbm_tiled(const double* __restrict__ a, double* __restrict__ b, int nx, int ny, int nz, int nd) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, z = blockIdx.z + 1;
  if (x >= nx) return;
  const int nxt = nx / 64;
  const int xm = x == 0 ? nx - 1 : x - 1, xp = x + 1 == nx ? 0 : x + 1;
  const int ym = y == 0 ? ny - 1 : y - 1, yp = y + 1 == ny ? 0 : y + 1;
  const unsigned ox[3] = {(unsigned)((xp >> 6) * 27 * 64 + (xp & 63)), (unsigned)((x >> 6) * 27 * 64 + (x & 63)), (unsigned)((xm >> 6) * 27 * 64 + (xm & 63))};
  double acc[27];
#pragma unroll
  for (int d = 0; d < 27; ++d) {
    if (d < nd) {
      const int cx = (d % 3) - 1, cy = ((d / 3) % 3) - 1, cz = (d / 9) - 1;
      const int ys = cy < 0 ? yp : cy > 0 ? ym : y;
      const double* row = a + ((long long)(z - cz) * ny + ys) * (long long)nxt * 27 * 64 + d * 64;
      acc[d] = row[ox[cx + 1]];
    }
  }
  double* orow = b + ((long long)z * ny + y) * (long long)nxt * 27 * 64;
#pragma unroll
  for (int d = 0; d < 27; ++d)
    if (d < nd) orow[d * 64 + ox[1]] = acc[d];
}


// AoSoA layout [z][y][x/64][27][64] per lattice (lattices on separate arrays, a + l*lstride), with the
// bulk kernel's XCD row-run mapping: a workgroup writes 4 x 13.8 KB contiguous chunks; pulls come from 9
// neighbour rows' chunks (+ one element of the adjacent x block for the 18 directions with c_x != 0).
// merged != 0: lattices inside the tile, [z][y][x/64][4][27][64] (one 55 KB chunk per workgroup).
__global__ void klbm_aosoa(const double* __restrict__ a, double* __restrict__ b, int nx, int ny, int nz, long long lstride, int nxb, int rchunk, int merged, int mode) {
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int r = slot / nxb, xb = slot - r * nxb;
  const int row = ((r / rchunk) * 8 + xcd) * rchunk + r % rchunk;
  if (row >= ny * nz) return;
  const int lat = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int y = row % ny, z = row / ny + 1;
  const long long tile = merged ? 4LL * 27 * 64 : 27LL * 64;
  const long long rowstride = (long long)nxb * tile;
  const double* aa = merged ? a + lat * 27 * 64 : a + lat * lstride;
  double* bb = merged ? b + lat * 27 * 64 : b + lat * lstride;
  const int ym = y == 0 ? ny - 1 : y - 1, yp = y + 1 == ny ? 0 : y + 1;
  const int xbm = xb == 0 ? nxb - 1 : xb - 1, xbp = xb + 1 == nxb ? 0 : xb + 1;
  // element offsets inside a row for c_x = -1 (pull from x+1), 0, +1 (pull from x-1)
  const long long ox[3] = {lane == 63 ? xbp * tile : xb * tile + lane + 1, xb * tile + lane, lane == 0 ? xbm * tile + 63 : xb * tile + lane - 1};
  double acc[27];
#pragma unroll
  for (int d = 0; d < 27; ++d) {
    // mode bit 2: a direction numbering in which the three c_x of one (c_y, c_z) are NOT neighbours
    const int dd = (mode & 4) ? (d * 10) % 27 : d;
    int cx = (dd % 3) - 1, cy = ((dd / 3) % 3) - 1, cz = (dd / 9) - 1;
    if (mode & 1) cx = 0;
    if (mode & 2) cy = cz = 0;
    const int ys = cy < 0 ? yp : cy > 0 ? ym : y;
    acc[d] = aa[((long long)(z - cz) * ny + ys) * rowstride + d * 64 + ox[cx + 1]];
  }
  double* orow = bb + ((long long)z * ny + y) * rowstride + ox[1];
#pragma unroll
  for (int d = 0; d < 27; ++d) orow[d * 64] = acc[d];
}

__global__ void k_read_only(const double* __restrict__ a, double* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  double s = 0;
  for (; i < n; i += stride) s += a[i];
  if (s == 12345.678) out[0] = s;
}
__global__ void k_write_only(double* __restrict__ b, size_t n, double v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) b[i] = v;
}

template <class F>
static void timeit(const char* name, double bytes, F f) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f();
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("%-34s %8.3f ms  %7.1f GB/s\n", name, best, bytes / best / 1e6);
  fflush(stdout);
}

int main() {
  const size_t n = (size_t)1 << 29;  // 4 GiB per array
  double *a, *b;
  CK(hipMalloc(&a, (n + 16) * 8)); CK(hipMalloc(&b, (n + 16) * 8));
  CK(hipMemset(a, 0, (n + 16) * 8)); CK(hipMemset(b, 0, (n + 16) * 8));
  const double bytes = 16.0 * n;
  dim3 g8((unsigned)(n / 256)), g16((unsigned)(n / 512)), blk(256);
  timeit("copy 8B/lane aligned", bytes, [&] { hipLaunchKernelGGL(k8, g8, blk, 0, 0, a, b, n, 0); });
  timeit("copy 8B/lane shifted", bytes, [&] { hipLaunchKernelGGL(k8, g8, blk, 0, 0, a, b, n, 1); });
  timeit("copy 8B/lane nt ld+st", bytes, [&] { hipLaunchKernelGGL(k8nt, g8, blk, 0, 0, a, b, n, 0); });
  timeit("copy 8B/lane nt shifted", bytes, [&] { hipLaunchKernelGGL(k8nt, g8, blk, 0, 0, a, b, n, 1); });
  timeit("copy 16B/lane aligned", bytes, [&] { hipLaunchKernelGGL(k16, g16, blk, 0, 0, a, b, n, 0); });
  timeit("copy 16B/lane 2x8B shifted loads", bytes, [&] { hipLaunchKernelGGL(k16, g16, blk, 0, 0, a, b, n, 1); });
  timeit("copy 16B/lane unaligned x4 load", bytes, [&] { hipLaunchKernelGGL(k16u, g16, blk, 0, 0, a, b, n, 1); });
  timeit("read only (grid-stride sum)", 8.0 * n, [&] { hipLaunchKernelGGL(k_read_only, dim3(256 * 32), blk, 0, 0, a, b, n); });
  timeit("write only (fill)", 8.0 * n, [&] { hipLaunchKernelGGL(k_write_only, g8, blk, 0, 0, b, n, 1.0); });
  CK(hipFree(a)); CK(hipFree(b));
  // LBM-shaped: 512x512x(130) planes, 27 streams
  const int nx = 512, ny = 512, nz = 130;
  const long long dstride = (long long)nx * ny * (nz + 2);
  CK(hipMalloc(&a, dstride * 27 * 8)); CK(hipMalloc(&b, dstride * 27 * 8));
  CK(hipMemset(a, 0, dstride * 27 * 8)); CK(hipMemset(b, 0, dstride * 27 * 8));
  const double lb = 16.0 * 27 * nx * ny * (double)nz;
  for (int bx : {64, 128, 256}) {
    dim3 g(nx / bx, ny, nz), bb(bx);
    char nm[64];
    snprintf(nm, sizeof nm, "lbm-shape 27 streams, block %d", bx);
    timeit(nm, lb, [&] { hipLaunchKernelGGL(klbm, g, bb, 0, 0, a, b, nx, ny, nz, dstride, 27, 0); });
    snprintf(nm, sizeof nm, "lbm-shape 27 streams nt, block %d", bx);
    timeit(nm, lb, [&] { hipLaunchKernelGGL(klbm, g, bb, 0, 0, a, b, nx, ny, nz, dstride, 27, 1); });
  }
  CK(hipFree(a)); CK(hipFree(b));
  {
    // 4 lattices, 512x512x128 each: 4 x 27 x 130 planes
    const int nzc = 128;
    const long long ds = (long long)nx * ny * (nzc + 2), ls = ds * 27;
    CK(hipMalloc(&a, ls * 4 * 8)); CK(hipMalloc(&b, ls * 4 * 8));
    CK(hipMemset(a, 0, ls * 4 * 8)); CK(hipMemset(b, 0, ls * 4 * 8));
    const double lb4 = 16.0 * 27 * 4 * nx * ny * (double)nzc;
    const int nxb = nx / 64;
    {
      const long long nrows2 = (long long)ny * nzc / 2, per2 = (nrows2 + 8LL * 32 - 1) / (8LL * 32) * 32;
      timeit("lbm-shape 4 lattices x 27, 2 rows per thread", lb4, [&] { hipLaunchKernelGGL(klbm_rpt<2>, dim3((unsigned)(8 * per2 * nxb)), dim3(256), 0, 0, a, b, nx, ny, nzc, ds, ls, nxb, 32); });
      const long long nrows4 = (long long)ny * nzc / 4, per4 = (nrows4 + 8LL * 16 - 1) / (8LL * 16) * 16;
      timeit("lbm-shape 4 lattices x 27, 4 rows per thread", lb4, [&] { hipLaunchKernelGGL(klbm_rpt<4>, dim3((unsigned)(8 * per4 * nxb)), dim3(256), 0, 0, a, b, nx, ny, nzc, ds, ls, nxb, 16); });
    }
    for (int rc : {1, 64}) for (int nt : {0}) {
      const long long nrows = (long long)ny * nzc;
      const long long per = (nrows + 8LL * rc - 1) / (8LL * rc) * rc;
      char nm[80];
      snprintf(nm, sizeof nm, "lbm-shape 4 lattices x 27, rchunk %d%s", rc, nt ? " nt" : "");
      timeit(nm, lb4, [&] { hipLaunchKernelGGL(klbm_chunk, dim3((unsigned)(8 * per * nxb)), dim3(256), 0, 0, a, b, nx, ny, nzc, ds, ls, nxb, rc, nt); });
    }
    for (int merged : {0, 1}) for (int rc : {1, 8, 64}) {
      const long long nrows = (long long)ny * nzc;
      const long long per = (nrows + 8LL * rc - 1) / (8LL * rc) * rc;
      char nm[80];
      snprintf(nm, sizeof nm, "lbm-shape AoSoA %s, rchunk %d", merged ? "[z][y][xb][4][27][64]" : "4 x [z][y][xb][27][64]", rc);
      timeit(nm, lb4, [&] { hipLaunchKernelGGL(klbm_aosoa, dim3((unsigned)(8 * per * nxb)), dim3(256), 0, 0, a, b, nx, ny, nzc, ls, nxb, rc, merged, 0); });
    }
    for (int merged : {0, 1}) for (int mode : {1, 2, 3, 4}) {
      const int rc = 64;
      const long long nrows = (long long)ny * nzc;
      const long long per = (nrows + 8LL * rc - 1) / (8LL * rc) * rc;
      char nm[96];
      snprintf(nm, sizeof nm, "AoSoA merged=%d rchunk 64 %s%s%s", merged, mode & 1 ? "no-x-shift " : "", mode & 2 ? "no-yz-offset " : "", mode & 4 ? "scattered-dirs" : "");
      timeit(nm, lb4, [&] { hipLaunchKernelGGL(klbm_aosoa, dim3((unsigned)(8 * per * nxb)), dim3(256), 0, 0, a, b, nx, ny, nzc, ls, nxb, rc, merged, mode); });
    }
    CK(hipFree(a)); CK(hipFree(b));
  }
  return 0;
}
