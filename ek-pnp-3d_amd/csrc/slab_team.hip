// slab_team.hip — the z-slab path's own transport, behind the C ABI.
//
// No reference counterpart: the reference is single-GPU (cudaSetDevice(0), main.cu:58).  What is
// rebuilt here for P slabs is its time loop (main.cu:189-200), its initialization()
// (LBM.cu:68-109) and its IO / diagnostics surface (LBM.cu:2492-2753), with the three exchanges a
// z decomposition needs between the split entry points of capi.hip:
//   HALO  9 c_z=+1 populations of the top plane up, 9 c_z=-1 of the bottom plane down, per lattice;
//         a RING, because gpu_stream wraps z (LBM.cu:1972,1975);
//   EDGE  4 interface coefficients per (kx,ky) mode of the distributed tridiagonal, all-gathered;
//   PHI   one phi plane each way for Ez (poisson.cu:50-55), same ring.
//
// A Team is the set of slab contexts ONE process drives plus the means to reach the others:
//   * one process per GPU (bench.py under torch.distributed.run, an MPI host ...): one local slab,
//     ncclCommInitRank from an id the host distributes        -> ekpnp_slab_attach_comm;
//   * one process, N GPUs (ekpnp_main --gpus N): all slabs local, ncclCommInitAll, or plain
//     hipMemcpyPeerAsync between the slabs' buffers            -> ekpnp_group_create.
// Transfers run on a per-slab COMM STREAM of the highest priority the device offers; the compute
// stream and the comm stream are ordered by events only (ready: buffers packed / consumed, done:
// data landed), so the HALO exchange overlaps the collision of the slab's interior planes:
//   boundary planes -> pack -> [comm: ring] || interior planes -> wait(done) -> unpack.
// RCCL is bound lazily (dlopen of librccl.so.1) so that single-GPU hosts never load it.
#include <dlfcn.h>
#include <unistd.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "ekpnp_internal.h"

namespace ekpnp {

// ---- librccl, bound on first use ---------------------------------------------------------------
struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional: only used to tear down after a failed collective
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static Rccl* rccl(std::string& err) {
  static Rccl r;
  static bool tried = false;
  static std::string why;
  if (!tried) {
    tried = true;
    // an already loaded librccl.so.1 (e.g. the copy PyTorch-ROCm ships) is reused by soname.
    // EKPNP_RCCL_LIBRARY names another file to bind (and nothing else is tried then): a site's own build of
    // RCCL, or - tests/test_capi_cpu.py - a file that does not exist, to see this very failure reported.
    const char* named = std::getenv("EKPNP_RCCL_LIBRARY");
    const std::string lib = named && *named ? named : "librccl.so.1";
    r.handle = dlopen(lib.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!r.handle && !(named && *named)) r.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!r.handle) {
      const char* e = dlerror();  // ONE call: glibc clears the message when it is read
      why = lib + " cannot be loaded: " + (e ? e : "?");
    } else {
      bool ok = true;
      auto bind = [&](auto& fn, const char* name) {
        fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(r.handle, name));
        if (!fn) { ok = false; why = lib + " lacks " + name; }
      };
      bind(r.GetUniqueId, "ncclGetUniqueId");
      bind(r.CommInitRank, "ncclCommInitRank");
      bind(r.CommInitAll, "ncclCommInitAll");
      bind(r.CommDestroy, "ncclCommDestroy");
      r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.handle, "ncclCommAbort"));
      bind(r.GroupStart, "ncclGroupStart");
      bind(r.GroupEnd, "ncclGroupEnd");
      bind(r.Send, "ncclSend");
      bind(r.Recv, "ncclRecv");
      bind(r.AllGather, "ncclAllGather");
      bind(r.AllReduce, "ncclAllReduce");
      bind(r.GetErrorString, "ncclGetErrorString");
      if (!ok) r.handle = nullptr;
    }
  }
  if (!r.handle) { err = why; return nullptr; }
  return &r;
}

// ---- the team ------------------------------------------------------------------------------------
enum { X_HALO = 0, X_PHI = 1, X_EDGE = 2, X_KINDS = 3 };
constexpr int MAX_BLOCKS = 16;  // mode blocks of the EDGE exchange ("edge_chunks"): its ready / done events are per block, index blk * nslabs + slab

struct Team {
  std::vector<ekpnp_ctx*> m;  // the slabs this process drives, ascending rank
  int nranks = 1;             // slabs of the whole lattice
  bool all_local = true;      // every rank is in m (rank r is m[r])
  bool group = false;         // in-process group (ekpnp_group_*): the member contexts do not dispatch on their own
  int kind = EKPNP_TRANSPORT_COPY;
  Rccl* nc = nullptr;
  std::vector<ncclComm_t> comm;
  std::vector<hipStream_t> cs;                            // comm stream per local slab
  std::vector<hipEvent_t> ready[X_KINDS], done[X_KINDS];  // per local slab
  std::vector<double*> red;                               // 2 doubles of device scratch per local slab
  // compute streams made with a CU mask (EKPNP_COMM_CUS: some CUs kept free of the slab's own kernels so that the
  // exchange kernel finds its workgroup slots at once), per local slab, or null
  std::vector<hipStream_t> masked;
  // measurement (ekpnp_kernel_timing_enable on the slab): per exchange kind and local slab, timed events around the
  // transfer on the comm stream and around the compute stream's wait for it
  struct XEv { hipEvent_t xfer_begin, xfer_end, wait_begin, wait_end; };
  std::vector<std::vector<XEv>> xev[X_KINDS];
  std::vector<size_t> xev_begun[X_KINDS], xev_used[X_KINDS];  // occurrences started / finished since the last reset (several mode blocks of one EDGE exchange are in flight at once)
  // knobs (environment at creation, ekpnp_tune / ekpnp_group_tune later; the same on every rank)
  bool inline_x = true;   // EKPNP_INLINE_EXCHANGES: the EDGE and PHI exchanges of an RCCL team on the compute stream itself
  int comm_cus = 0;       // EKPNP_COMM_CUS: compute units kept free of the slab's own kernels
  bool edge_p2p = false;  // EKPNP_EDGE_P2P: the EDGE all-gather of an RCCL team as direct ncclSend / ncclRecv pairs with every peer
  bool comm_strict = false;
  double t = 0.0;
  std::string err;
  // in-process groups (ekpnp_group_*): a verb that failed on one slab has left the others one call behind or ahead -
  // the group is drained, marked, and answers every further verb with the first failure (group_fail)
  bool poisoned = false;
  std::string poison;
  bool comm_broken = false;  // an RCCL call failed inside a collective: its kernels may never finish, the communicators are aborted, not drained
};

static inline Ctx& S(Team& T, int i) { return T.m[i]->c; }

#define THIP(T, call)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) {                                                             \
      (T).err = std::string(#call) + ": " + hipGetErrorString(e_);                      \
      return EKPNP_ERR_HIP;                                                             \
    }                                                                                   \
  } while (0)
#define TNCCL(T, call)                                                                  \
  do {                                                                                  \
    ncclResult_t r_ = (call);                                                           \
    if (r_ != ncclSuccess) {                                                            \
      (T).err = std::string(#call) + ": " + (T).nc->GetErrorString(r_);                 \
      return EKPNP_ERR_HIP;                                                             \
    }                                                                                   \
  } while (0)
// a call into one slab's own entry points (capi.hip); its message becomes the team's
#define TSLAB(T, i, call)                                                               \
  do {                                                                                  \
    int rc_ = (call);                                                                   \
    if (rc_ != EKPNP_OK) {                                                              \
      (T).err = "slab " + std::to_string(S(T, i).rank) + ": " + S(T, i).err;            \
      return rc_;                                                                       \
    }                                                                                   \
  } while (0)

static int use(Team& T, int i) {
  THIP(T, hipSetDevice(S(T, i).device));
  return EKPNP_OK;
}

static hipError_t xcopy(void* dst, int ddev, const void* src, int sdev, size_t bytes, hipStream_t s) {
  return ddev == sdev ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s) : hipMemcpyPeerAsync(dst, ddev, src, sdev, bytes, s);
}

// the local slabs whose buffers slab i's data is written into (COPY) == whose data lands in slab i's
// buffers: the ring neighbours, or everybody for the all-gather
template <class Fn>
static void for_partners(Team& T, int i, int x, Fn&& fn) {
  const int n = (int)T.m.size();
  if (x == X_EDGE) {
    for (int j = 0; j < n; ++j)
      if (j != i) fn(j);
  } else {
    const int up = (i + 1) % n, dn = (i + n - 1) % n;
    if (up != i) fn(up);
    if (dn != i && dn != up) fn(dn);
  }
}

// the timed events of the next occurrence of exchange x on local slab i (made on first use, reused after a reset)
static int xev_slot(Team& T, int x, int i, Team::XEv** out) {
  std::vector<Team::XEv>& v = T.xev[x][i];
  if (T.xev_begun[x][i] == v.size()) {
    Team::XEv e{};
    THIP(T, hipEventCreate(&e.xfer_begin));
    THIP(T, hipEventCreate(&e.xfer_end));
    THIP(T, hipEventCreate(&e.wait_begin));
    THIP(T, hipEventCreate(&e.wait_end));
    v.push_back(e);
  }
  *out = &v[T.xev_begun[x][i]];
  return EKPNP_OK;
}

// The EDGE and PHI exchanges sit on the critical path of the Poisson solve: nothing runs beside them, so a hop to the comm
// stream and back (two event hand-overs, ~40 us each way) buys nothing.  Over RCCL they are issued on the slab's COMPUTE
// stream itself (stream order replaces the events); the population halo keeps its comm stream - that is where the overlap
// is.  Device-copy groups keep the comm stream for all three (their copies write into the partners' buffers, which the
// events guard).  EKPNP_INLINE_EXCHANGES=0 is the A/B partner.
// An EDGE exchange in several mode blocks is there to run BESIDE the transforms of the other blocks: always on the comm stream.
static inline int edge_blocks(Team& T) { return edge_chunk_count(S(T, 0)); }
static inline bool inline_exchange(Team& T, int x) {
  return T.inline_x && T.kind == EKPNP_TRANSPORT_RCCL && x != X_HALO && !(x == X_EDGE && edge_blocks(T) > 1);
}

// start exchange x (EDGE: of mode block blk): everything the slabs have enqueued so far on their compute streams precedes it
static int exchange_begin(Team& T, int x, int blk = 0) {
  const int n = (int)T.m.size();
  const bool inl = inline_exchange(T, x);
  const int eo = blk * n;  // this block's events
  auto xs = [&](int i) { return inl ? S(T, i).stream : T.cs[i]; };  // the stream slab i's part of the exchange runs on
  std::vector<Team::XEv*> tev((size_t)n, nullptr);
  int rc;
  for (int i = 0; i < n && !inl; ++i) {
    if ((rc = use(T, i))) return rc;
    THIP(T, hipEventRecord(T.ready[x][eo + i], S(T, i).stream));
  }
  for (int i = 0; i < n; ++i) {
    if ((rc = use(T, i))) return rc;
    if (!inl) THIP(T, hipStreamWaitEvent(T.cs[i], T.ready[x][eo + i], 0));
    if (T.kind == EKPNP_TRANSPORT_COPY) {  // slab i's copies write into its partners' receive buffers
      hipError_t e = hipSuccess;
      for_partners(T, i, x, [&](int j) { if (e == hipSuccess) e = hipStreamWaitEvent(T.cs[i], T.ready[x][eo + j], 0); });
      THIP(T, e);
    }
    if (S(T, i).timing) {  // the stream gets here when this slab's (and its partners') buffers are ready
      if ((rc = xev_slot(T, x, i, &tev[i]))) return rc;
      T.xev_begun[x][i]++;
      if (inl) THIP(T, hipEventRecord(tev[i]->wait_begin, xs(i)));  // an inline exchange holds the compute stream for its whole length
      THIP(T, hipEventRecord(tev[i]->xfer_begin, xs(i)));
    }
  }
  if (T.kind == EKPNP_TRANSPORT_RCCL) {
    if (x == X_EDGE && T.edge_p2p) {  // a rank's own piece of the gather: a device copy ahead of the sends, same stream
      for (int i = 0; i < n; ++i) {
        Ctx& c = S(T, i);
        if ((rc = use(T, i))) return rc;
        const ModeBlock mb = mode_block(c, blk);
        THIP(T, hipMemcpyAsync(c.edge_all + mb.all_off + (size_t)c.rank * mb.doubles, c.edge_local + mb.local_off, mb.doubles * sizeof(double), hipMemcpyDeviceToDevice, xs(i)));
      }
    }
    TNCCL(T, T.nc->GroupStart());
    for (int i = 0; i < n; ++i) {
      Ctx& c = S(T, i);
      if ((rc = use(T, i))) { (void)T.nc->GroupEnd(); return rc; }
      ncclResult_t r = ncclSuccess;
      if (x == X_EDGE && T.edge_p2p) {
        // the same gather as one direct send / receive pair per peer: on xGMI every peer is one hop away over a link of its
        // own, so a rank's piece goes out on all links at once instead of travelling round a ring (A/B partner of the
        // ncclAllGather below; the pieces land where it puts them - rank q's at [q] of the block -, hence the same bits)
        const ModeBlock mb = mode_block(c, blk);
        for (int q = 0; q < T.nranks && r == ncclSuccess; ++q) {
          if (q == c.rank) continue;
          r = T.nc->Send(c.edge_local + mb.local_off, mb.doubles, ncclDouble, q, T.comm[i], xs(i));
          if (r == ncclSuccess) r = T.nc->Recv(c.edge_all + mb.all_off + (size_t)q * mb.doubles, mb.doubles, ncclDouble, q, T.comm[i], xs(i));
        }
      } else if (x == X_EDGE) {
        const ModeBlock mb = mode_block(c, blk);
        r = T.nc->AllGather(c.edge_local + mb.local_off, c.edge_all + mb.all_off, mb.doubles, ncclDouble, T.comm[i], xs(i));
      } else {
        double** b = x == X_HALO ? c.halo : c.phi_halo;
        const size_t cnt = x == X_HALO ? c.halo_doubles : c.plane;
        const int up = (c.rank + 1) % T.nranks, dn = (c.rank + T.nranks - 1) % T.nranks;
        // order matters when both neighbours are the same peer (2 ranks) or the rank itself (1 rank):
        // [send up, send down] pairs with the peer's [recv from below, recv from above]
        r = T.nc->Send(b[1], cnt, ncclDouble, up, T.comm[i], xs(i));
        if (r == ncclSuccess) r = T.nc->Send(b[0], cnt, ncclDouble, dn, T.comm[i], xs(i));
        if (r == ncclSuccess) r = T.nc->Recv(b[2], cnt, ncclDouble, dn, T.comm[i], xs(i));
        if (r == ncclSuccess) r = T.nc->Recv(b[3], cnt, ncclDouble, up, T.comm[i], xs(i));
      }
      if (r != ncclSuccess) {
        (void)T.nc->GroupEnd();
        T.comm_broken = true;  // some ranks' operations may have been issued without their partners'
        T.err = std::string("RCCL exchange failed: ") + T.nc->GetErrorString(r);
        return EKPNP_ERR_HIP;
      }
    }
    {
      const ncclResult_t ge = T.nc->GroupEnd();
      if (ge != ncclSuccess) {
        T.comm_broken = true;
        T.err = std::string("ncclGroupEnd: ") + T.nc->GetErrorString(ge);
        return EKPNP_ERR_HIP;
      }
    }
  } else {
    for (int i = 0; i < n; ++i) {
      Ctx& c = S(T, i);
      if ((rc = use(T, i))) return rc;
      if (x == X_EDGE) {
        const ModeBlock mb = mode_block(c, blk);
        for (int j = 0; j < n; ++j)
          THIP(T, xcopy(S(T, j).edge_all + mb.all_off + (size_t)c.rank * mb.doubles, S(T, j).device, c.edge_local + mb.local_off, c.device, mb.doubles * sizeof(double), T.cs[i]));
      } else {
        const int up = (i + 1) % n, dn = (i + n - 1) % n;
        Ctx &cu = S(T, up), &cd = S(T, dn);
        double **b = x == X_HALO ? c.halo : c.phi_halo, **bu = x == X_HALO ? cu.halo : cu.phi_halo, **bd = x == X_HALO ? cd.halo : cd.phi_halo;
        const size_t bytes = (x == X_HALO ? c.halo_doubles : c.plane) * sizeof(double);
        THIP(T, xcopy(bu[2], cu.device, b[1], c.device, bytes, T.cs[i]));  // my send-up   -> upper neighbour's recv-from-below
        THIP(T, xcopy(bd[3], cd.device, b[0], c.device, bytes, T.cs[i]));  // my send-down -> lower neighbour's recv-from-above
      }
    }
  }
  for (int i = 0; i < n; ++i) {
    if ((rc = use(T, i))) return rc;
    if (!inl) THIP(T, hipEventRecord(T.done[x][eo + i], T.cs[i]));
    if (tev[i]) {
      THIP(T, hipEventRecord(tev[i]->xfer_end, xs(i)));
      if (inl) THIP(T, hipEventRecord(tev[i]->wait_end, xs(i)));
    }
  }
  return EKPNP_OK;
}

// what the slabs enqueue on their compute streams from here on sees the exchanged data (EDGE: of mode block blk; the blocks
// are finished in the order they were begun)
static int exchange_finish(Team& T, int x, int blk = 0) {
  const int n = (int)T.m.size();
  const int eo = blk * n;
  int rc;
  if (inline_exchange(T, x)) {  // already in stream order; only the measurement's bookkeeping is left
    for (int i = 0; i < n; ++i)
      if (S(T, i).timing && T.xev_used[x][i] < T.xev_begun[x][i]) T.xev_used[x][i]++;
    return EKPNP_OK;
  }
  for (int i = 0; i < n; ++i) {
    if ((rc = use(T, i))) return rc;
    // measurement: the time the compute stream spends between these two events is the time it had nothing to do but
    // wait for the exchange (0 when the transfer was hidden behind what the stream ran meanwhile)
    const bool timed = S(T, i).timing && T.xev_used[x][i] < T.xev_begun[x][i];
    if (timed) THIP(T, hipEventRecord(T.xev[x][i][T.xev_used[x][i]].wait_begin, S(T, i).stream));
    THIP(T, hipStreamWaitEvent(S(T, i).stream, T.done[x][eo + i], 0));
    if (T.kind == EKPNP_TRANSPORT_COPY) {
      hipError_t e = hipSuccess;
      for_partners(T, i, x, [&](int j) { if (e == hipSuccess) e = hipStreamWaitEvent(S(T, i).stream, T.done[x][eo + j], 0); });
      THIP(T, e);
    }
    if (timed) {
      THIP(T, hipEventRecord(T.xev[x][i][T.xev_used[x][i]].wait_end, S(T, i).stream));
      T.xev_used[x][i]++;
    }
  }
  return EKPNP_OK;
}

// stream_collide_save (LBM.cu:465-481) over the slabs, halo exchange hidden behind the interior planes
static int team_stream_collide_save(Team& T) {
  const int n = (int)T.m.size();
  int rc;
  for (int i = 0; i < n; ++i) {
    if ((rc = use(T, i))) return rc;
    TSLAB(T, i, ekpnp_collide_boundary_planes(T.m[i]));
    TSLAB(T, i, ekpnp_halo_pack(T.m[i]));
  }
  if ((rc = exchange_begin(T, X_HALO))) return rc;
  for (int i = 0; i < n; ++i) {
    if ((rc = use(T, i))) return rc;
    TSLAB(T, i, ekpnp_collide_interior_planes(T.m[i]));
  }
  if ((rc = exchange_finish(T, X_HALO))) return rc;
  for (int i = 0; i < n; ++i) {
    if ((rc = use(T, i))) return rc;
    TSLAB(T, i, ekpnp_halo_unpack(T.m[i]));
  }
  return EKPNP_OK;
}

// fast_Poisson (poisson.cu:75-103) over the slabs.  The z coupling of the reference's 3-D transform (poisson.cu:86-92) is here
// an all-gather of every slab's edge values (EDGE).  With "edge_chunks" = C > 1 the half spectrum is cut into C blocks of kx
// columns after the row pass: block k's column pass and edge values run on the compute stream while block k-1's all-gather
// is on the comm stream, and stage 2 (interface system, z solve, inverse column pass) starts on block 0 as soon as ITS
// edge values are there, with the later blocks' all-gathers still in flight.  Every mode sees the same operations in the same
// order whatever C is: phi is bit-identical.
static int team_fast_poisson(Team& T) {
  const int n = (int)T.m.size();
  const int nb = edge_blocks(T);
  int rc;
  for (int i = 0; i < n; ++i) {
    if ((rc = use(T, i))) return rc;
    TSLAB(T, i, poisson_stage1_begin(S(T, i)));
  }
  for (int k = 0; k < nb; ++k) {
    for (int i = 0; i < n; ++i) {
      if ((rc = use(T, i))) return rc;
      TSLAB(T, i, poisson_stage1_block(S(T, i), k));
      if (k == nb - 1) TSLAB(T, i, poisson_stage1_end(S(T, i)));
    }
    if ((rc = exchange_begin(T, X_EDGE, k))) return rc;
  }
  for (int k = 0; k < nb; ++k) {
    if ((rc = exchange_finish(T, X_EDGE, k))) return rc;
    for (int i = 0; i < n; ++i) {
      if ((rc = use(T, i))) return rc;
      TSLAB(T, i, poisson_stage2_block(S(T, i), k));
    }
  }
  for (int i = 0; i < n; ++i) {
    if ((rc = use(T, i))) return rc;
    TSLAB(T, i, poisson_stage2_end(S(T, i)));
  }
  if ((rc = exchange_begin(T, X_PHI)) || (rc = exchange_finish(T, X_PHI))) return rc;
  for (int i = 0; i < n; ++i) {
    if ((rc = use(T, i))) return rc;
    TSLAB(T, i, ekpnp_poisson_stage3(T.m[i]));
  }
  return EKPNP_OK;
}

static int team_step(Team& T, int nsteps) {  // main.cu:189-200
  if (nsteps < 0) { T.err = "nsteps < 0"; return EKPNP_ERR_INVALID; }
  for (int s = 0; s < nsteps; ++s) {
    // opt-in "batch_moments": only the call's last step stores the moment arrays of the interior planes (capi.hip: ekpnp_step)
    for (size_t i = 0; i < T.m.size(); ++i) S(T, (int)i).skip_moments = s < nsteps - 1 && batch_moments_ok(S(T, (int)i));
    int rc = team_stream_collide_save(T);
    for (size_t i = 0; i < T.m.size(); ++i) S(T, (int)i).skip_moments = false;
    if (rc == EKPNP_OK) rc = team_fast_poisson(T);
    if (rc) return rc;
    for (size_t i = 0; i < T.m.size(); ++i) TSLAB(T, (int)i, ekpnp_advance_time(T.m[i]));
  }
  return EKPNP_OK;
}

static int team_synchronize(Team& T) {
  int rc;
  for (size_t i = 0; i < T.m.size(); ++i) {
    if ((rc = use(T, (int)i))) return rc;
    THIP(T, hipStreamSynchronize(T.cs[i]));
    TSLAB(T, (int)i, ekpnp_synchronize(T.m[i]));
  }
  return EKPNP_OK;
}

// combine one number per local slab over all ranks: sum or max.  Every rank gets the result.
static int team_reduce(Team& T, const std::vector<double>& local, bool is_max, double* out) {
  double v = local.empty() ? 0.0 : local[0];
  for (size_t i = 1; i < local.size(); ++i) v = is_max ? std::fmax(v, local[i]) : v + local[i];
  if (!T.all_local) {  // one local slab, the others are other processes
    int rc = use(T, 0);
    if (rc) return rc;
    hipStream_t st = S(T, 0).stream;
    THIP(T, hipMemcpyAsync(T.red[0], &v, sizeof(double), hipMemcpyHostToDevice, st));
    TNCCL(T, T.nc->AllReduce(T.red[0], T.red[0] + 1, 1, ncclDouble, is_max ? ncclMax : ncclSum, T.comm[0], st));
    THIP(T, hipMemcpyAsync(&v, T.red[0] + 1, sizeof(double), hipMemcpyDeviceToHost, st));
    THIP(T, hipStreamSynchronize(st));
  }
  *out = v;
  return EKPNP_OK;
}

// initialization() (LBM.cu:68-109) with the slab Poisson solve inside the Picard loop
static int team_initialization(Team& T) {
  const int n = (int)T.m.size();
  int rc;
  for (int i = 0; i < n; ++i) {
    if ((rc = use(T, i))) return rc;
    TSLAB(T, i, ekpnp_init_fields(T.m[i]));  // gpu_initialization, LBM.cu:76
    TSLAB(T, i, ekpnp_pbe_begin(T.m[i]));    // LBM.cu:79-86
  }
  const int sweeps = S(T, 0).p.pb_iterations;
  for (int it = 0; it < sweeps; ++it) {      // LBM.cu:89-106
    for (int i = 0; i < n; ++i) {
      if ((rc = use(T, i))) return rc;
      TSLAB(T, i, ekpnp_pbe_concentrations(T.m[i]));
    }
    if ((rc = team_fast_poisson(T))) return rc;
    for (int i = 0; i < n; ++i) {
      if ((rc = use(T, i))) return rc;
      TSLAB(T, i, ekpnp_pbe_relax(T.m[i]));
    }
  }
  for (int i = 0; i < n; ++i) {
    if ((rc = use(T, i))) return rc;
    TSLAB(T, i, ekpnp_pbe_end(T.m[i]));
  }
  return EKPNP_OK;
}

// ekpnp_initialization_converged over the slabs: same damping rule, residual = max over all ranks
static int team_initialization_converged(Team& T, double rel_tol, int max_sweeps, int* sweeps, double* residual) {
  if (max_sweeps < 0) { T.err = "max_sweeps < 0"; return EKPNP_ERR_INVALID; }
  const int n = (int)T.m.size();
  const ekpnp_params& p = S(T, 0).p;
  double omega = p.PB_omega;
  if (p.chargeinf > 0.0) {
    const double lam2 = p.eps * p.kB * p.roomT / p.electron / (2.0 * p.chargeinf * p.convertCtoCharge);
    const double A = p.Lz * p.Lz / (M_PI * M_PI * lam2);
    if (1.6 / (1.0 + A) < omega) omega = 1.6 / (1.0 + A);
  }
  double scale = std::fabs(p.voltage) > std::fabs(p.voltage2) ? std::fabs(p.voltage) : std::fabs(p.voltage2);
  if (scale == 0.0) scale = 1.0;
  int rc;
  for (int i = 0; i < n; ++i) {
    Ctx& c = S(T, i);
    if ((rc = use(T, i))) return rc;
    if (!c.diag) THIP(T, hipMalloc((void**)&c.diag, DIAG_SCRATCH * sizeof(double)));
    TSLAB(T, i, ekpnp_init_fields(T.m[i]));
    TSLAB(T, i, ekpnp_pbe_begin(T.m[i]));
  }
  int done = 0;
  double res = 0.0;
  const int check_every = 10;
  rc = EKPNP_OK;
  while (done < max_sweeps) {
    for (int i = 0; i < n; ++i) {
      if ((rc = use(T, i))) return rc;
      TSLAB(T, i, ekpnp_pbe_concentrations(T.m[i]));
    }
    if ((rc = team_fast_poisson(T))) break;
    ++done;
    const bool check = (done % check_every == 0) || done == max_sweeps;
    for (int i = 0; i < n; ++i) {
      Ctx& c = S(T, i);
      if ((rc = use(T, i))) return rc;
      if (check) launch_max_abs_diff(c, c.fld[EKPNP_PHI], c.phi_old, c.diag);
      launch_pbe_relax(c, c.phi_old, omega);
    }
    if (check) {
      std::vector<double> r(n, 0.0);
      for (int i = 0; i < n; ++i) {
        Ctx& c = S(T, i);
        if ((rc = use(T, i))) return rc;
        THIP(T, hipMemcpyAsync(&r[i], c.diag + 1024, sizeof(double), hipMemcpyDeviceToHost, c.stream));
        THIP(T, hipStreamSynchronize(c.stream));
        if (!(r[i] == r[i])) r[i] = 1.0e300;  // NaN must win the max
      }
      double rmax = 0.0;
      if ((rc = team_reduce(T, r, true, &rmax))) break;
      res = rmax / scale;
      if (rmax >= 1.0e300) { T.err = "Poisson-Boltzmann iteration produced NaN"; rc = EKPNP_ERR_INVALID; break; }
      if (res <= rel_tol) break;
    }
  }
  for (int i = 0; i < n; ++i) {
    if (use(T, i) == EKPNP_OK) (void)ekpnp_pbe_end(T.m[i]);
  }
  if (sweeps) *sweeps = done;
  if (residual) *residual = res;
  return rc;
}

// fn on every slab of the lattice in rank order (whole-lattice text files are written plane by
// plane); between the turns of different processes the ranks agree on the status so far
static int team_turns(Team& T, int (*fn)(Ctx&, void*), void* arg) {
  int status = EKPNP_OK;
  for (int r = 0; r < T.nranks; ++r) {
    const int slot = T.all_local ? r : (S(T, 0).rank == r ? 0 : -1);
    if (slot >= 0 && status == EKPNP_OK) {
      int rc = use(T, slot);
      if (rc == EKPNP_OK) {
        rc = fn(S(T, slot), arg);
        if (rc) T.err = "slab " + std::to_string(r) + ": " + S(T, slot).err;
      }
      status = rc;
    }
    if (!T.all_local) {
      double worst = 0.0;
      int rc = team_reduce(T, {(double)status}, true, &worst);
      if (rc) return rc;
      if (worst != 0.0 && status == EKPNP_OK) {
        T.err = "file IO failed on another rank";
        status = (int)worst;
      }
    }
  }
  return status;
}

// ---- resources -----------------------------------------------------------------------------------
static int team_make_streams(Team& T) {
  const int n = (int)T.m.size();
  T.cs.assign(n, nullptr);
  T.red.assign(n, nullptr);
  T.masked.assign(n, nullptr);
  for (int x = 0; x < X_KINDS; ++x) {
    T.ready[x].assign((size_t)n * (x == X_EDGE ? MAX_BLOCKS : 1), nullptr);
    T.done[x].assign(T.ready[x].size(), nullptr);
    T.xev[x].assign(n, {});
    T.xev_begun[x].assign(n, 0);
    T.xev_used[x].assign(n, 0);
  }
  // EKPNP_COMM_CUS=<n> (A/B knob, default 0): keep n compute units free of the slab's OWN kernels - its compute
  // stream is re-made with a CU mask that leaves them out - so that the workgroups of the exchange kernel are placed
  // the moment it is dispatched instead of competing for wave slots with a sweep of ~10^6 workgroups (the sweep is
  // bandwidth-bound and does not miss 3 % of the CUs).  KFD deals the mask bits round-robin over the XCDs, so a
  // multiple of 8 takes the same number of CUs from every XCD and the kernels' bid % 8 XCD maps stay valid.
  // EKPNP_COMM_CUS_STRICT=1 also confines the comm stream to exactly those CUs.
  // Both, and EKPNP_INLINE_EXCHANGES, are read when the team is made; ekpnp_tune / ekpnp_group_tune change them on a live one.
  T.inline_x = !(std::getenv("EKPNP_INLINE_EXCHANGES") && std::atoi(std::getenv("EKPNP_INLINE_EXCHANGES")) == 0);
  T.edge_p2p = std::getenv("EKPNP_EDGE_P2P") && std::atoi(std::getenv("EKPNP_EDGE_P2P")) == 1;
  T.comm_cus = std::getenv("EKPNP_COMM_CUS") ? std::atoi(std::getenv("EKPNP_COMM_CUS")) : 0;
  T.comm_strict = std::getenv("EKPNP_COMM_CUS_STRICT") != nullptr && std::atoi(std::getenv("EKPNP_COMM_CUS_STRICT")) != 0;
  const int comm_cus = T.comm_cus;
  const bool comm_strict = T.comm_strict;
  for (int i = 0; i < n; ++i) {
    int rc = use(T, i);
    if (rc) return rc;
    int least = 0, greatest = 0;
    THIP(T, hipDeviceGetStreamPriorityRange(&least, &greatest));
    int ncu = 0;
    THIP(T, hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, S(T, i).device));
    const bool reserve = comm_cus > 0 && comm_cus < ncu;
    if (reserve) {
      std::vector<uint32_t> comp((size_t)(ncu + 31) / 32, 0u), comm(comp.size(), 0u);
      for (int b = 0; b < ncu; ++b) (b < comm_cus ? comm : comp)[(size_t)b / 32] |= 1u << (b % 32);
      THIP(T, hipExtStreamCreateWithCUMask(&T.masked[i], (uint32_t)comp.size(), comp.data()));
      TSLAB(T, i, ekpnp_set_stream(T.m[i], T.masked[i]));  // the slab's kernels and transforms now run beside the reserved CUs
      if (comm_strict) THIP(T, hipExtStreamCreateWithCUMask(&T.cs[i], (uint32_t)comm.size(), comm.data()));
    }
    // highest priority: a transfer enqueued behind tens of milliseconds of collision kernels must be
    // dispatched as soon as workgroup slots free up, not after the compute queue has drained
    if (!T.cs[i]) THIP(T, hipStreamCreateWithPriority(&T.cs[i], hipStreamNonBlocking, greatest));
    for (int x = 0; x < X_KINDS; ++x)
      for (size_t e = (size_t)i; e < T.ready[x].size(); e += (size_t)n) {  // (EDGE: one pair per mode block)
        THIP(T, hipEventCreateWithFlags(&T.ready[x][e], hipEventDisableTiming));
        THIP(T, hipEventCreateWithFlags(&T.done[x][e], hipEventDisableTiming));
      }
    THIP(T, hipMalloc((void**)&T.red[i], 2 * sizeof(double)));
  }
  return EKPNP_OK;
}

static void team_release(Team& T) {
  // A collective that failed half-issued has kernels waiting for partners that never come.  With ncclCommAbort they are
  // aborted first and everything below is safe; without it (an RCCL library that does not export it) nothing may wait for
  // those streams: the comm streams, the slabs' compute streams' pending work and the communicators are LEFT BEHIND (leaked)
  // rather than blocking ekpnp_group_destroy / ekpnp_destroy for ever.  Untested on hardware: no RCCL failure
  // could be provoked on the one-GPU boxes; the ordinary path is what the tests run.
  const bool stuck = T.comm_broken && !(T.nc && T.nc->CommAbort);
  if (T.comm_broken && T.nc && T.nc->CommAbort) {
    for (size_t i = 0; i < T.comm.size(); ++i)
      if (T.comm[i]) (void)T.nc->CommAbort(T.comm[i]);
    T.comm.clear();
  }
  if (stuck) {
    std::fprintf(stderr, "ekpnp: an RCCL collective failed and this RCCL library has no ncclCommAbort: %zu communicator(s) and their streams are leaked instead of waited for\n", T.comm.size());
    T.comm.clear();
    for (size_t i = 0; i < T.cs.size(); ++i) T.cs[i] = nullptr;  // not synchronised, not destroyed
  }
  for (size_t i = 0; i < T.m.size() && !stuck; ++i) {
    (void)hipSetDevice(S(T, (int)i).device);
    if (i < T.cs.size() && T.cs[i]) (void)hipStreamSynchronize(T.cs[i]);
    if (S(T, (int)i).stream) (void)hipStreamSynchronize(S(T, (int)i).stream);
  }
  for (size_t i = 0; i < T.comm.size(); ++i)
    if (T.comm[i] && T.nc) (void)T.nc->CommDestroy(T.comm[i]);
  T.comm.clear();
  for (size_t i = 0; i < T.m.size(); ++i) {
    (void)hipSetDevice(S(T, (int)i).device);
    for (int x = 0; x < X_KINDS; ++x) {
      for (size_t e = i; e < T.ready[x].size(); e += T.m.size()) {
        if (T.ready[x][e]) (void)hipEventDestroy(T.ready[x][e]);
        if (e < T.done[x].size() && T.done[x][e]) (void)hipEventDestroy(T.done[x][e]);
      }
      if (i < T.xev[x].size())
        for (Team::XEv& e : T.xev[x][i]) {
          (void)hipEventDestroy(e.xfer_begin);
          (void)hipEventDestroy(e.xfer_end);
          (void)hipEventDestroy(e.wait_begin);
          (void)hipEventDestroy(e.wait_end);
        }
    }
    if (i < T.cs.size() && T.cs[i]) (void)hipStreamDestroy(T.cs[i]);
    if (i < T.red.size() && T.red[i]) (void)hipFree(T.red[i]);
    if (i < T.masked.size() && T.masked[i]) {
      // the slab outlives its team (ekpnp_destroy comes after): it gets a plain stream of its own back
      Ctx& c = S(T, (int)i);
      hipStream_t plain = nullptr;
      if (hipStreamCreateWithFlags(&plain, hipStreamNonBlocking) == hipSuccess && ekpnp_set_stream(T.m[i], plain) == EKPNP_OK) c.own_stream = true;
      (void)hipStreamDestroy(T.masked[i]);
    }
  }
  T.cs.clear();
  T.red.clear();
  T.masked.clear();
  for (int x = 0; x < X_KINDS; ++x) { T.ready[x].clear(); T.done[x].clear(); T.xev[x].clear(); T.xev_begun[x].clear(); T.xev_used[x].clear(); }
}

// ---- knobs of a live team (ekpnp_tune on an attached slab, ekpnp_group_tune) -------------------------------------------------
// Every rank of the lattice must make the same calls in the same order (like every other verb): "inline_exchanges" decides
// on which stream a rank's half of a collective is issued, which is its own business, but a run is only an A/B leg when all
// ranks agree.  Each returns with the team's streams drained.
static int team_set_comm_cus(Team& T, int cus) {
  int rc = team_synchronize(T);
  if (rc) return rc;
  for (int i = 0; i < (int)T.m.size(); ++i) {
    if ((rc = use(T, i))) return rc;
    Ctx& c = S(T, i);
    int ncu = 0;
    THIP(T, hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c.device));
    const bool reserve = cus > 0 && cus < ncu;
    hipStream_t fresh = nullptr;
    if (reserve) {
      std::vector<uint32_t> comp((size_t)(ncu + 31) / 32, 0u);
      for (int b = cus; b < ncu; ++b) comp[(size_t)b / 32] |= 1u << (b % 32);
      THIP(T, hipExtStreamCreateWithCUMask(&fresh, (uint32_t)comp.size(), comp.data()));
    } else {
      THIP(T, hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking));
    }
    hipStream_t old_masked = T.masked[i];
    TSLAB(T, i, ekpnp_set_stream(T.m[i], fresh));  // (destroys the slab's previous stream if that was its own)
    if (old_masked) (void)hipStreamDestroy(old_masked);
    T.masked[i] = reserve ? fresh : nullptr;
    c.own_stream = !reserve;  // a plain stream belongs to the slab from here on, a masked one to the team (team_release)
  }
  T.comm_cus = cus;
  return EKPNP_OK;
}
static int team_tune(Team& T, const char* knob, int value) {
  if (std::strcmp(knob, "inline_exchanges") == 0 && (value == 0 || value == 1)) {
    const int rc = team_synchronize(T);
    if (rc) return rc;
    T.inline_x = value != 0;
    return EKPNP_OK;
  }
  if (std::strcmp(knob, "edge_p2p") == 0 && (value == 0 || value == 1)) {
    const int rc = team_synchronize(T);
    if (rc) return rc;
    T.edge_p2p = value != 0;
    return EKPNP_OK;
  }
  if (std::strcmp(knob, "comm_cus") == 0 && value >= 0 && value <= 64) return team_set_comm_cus(T, value);
  T.err = std::string("unknown transport knob or bad value: ") + knob + " = " + std::to_string(value);
  return EKPNP_ERR_INVALID;
}

// ---- the reference's verbs on ONE attached slab context (one process per GPU) --------------------
static int own_team(Ctx& c, Team** T) {
  if (!c.team) { c.err = "slab context without a transport: ekpnp_slab_attach_comm it, use ekpnp_group_*, or drive the split calls"; return EKPNP_ERR_INVALID; }
  if (c.team->group) { c.err = "this slab belongs to an ekpnp_group: use the ekpnp_group_* calls"; return EKPNP_ERR_INVALID; }
  *T = c.team;
  return EKPNP_OK;
}
#define OWN_TEAM(c)                  \
  Team* T = nullptr;                 \
  {                                  \
    int rc0_ = own_team(c, &T);      \
    if (rc0_) return rc0_;           \
  }
static int lift(Ctx& c, Team& T, int rc) {  // the attached context reports the team's message as its own
  if (rc && !T.err.empty()) c.err = T.err;
  return rc;
}

int team_ctx_stream_collide_save(Ctx& c) { OWN_TEAM(c); return lift(c, *T, team_stream_collide_save(*T)); }
int team_ctx_fast_poisson(Ctx& c) { OWN_TEAM(c); return lift(c, *T, team_fast_poisson(*T)); }
int team_ctx_step(Ctx& c, int n) { OWN_TEAM(c); return lift(c, *T, team_step(*T, n)); }
int team_ctx_initialization(Ctx& c) { OWN_TEAM(c); return lift(c, *T, team_initialization(*T)); }
int team_ctx_initialization_converged(Ctx& c, double tol, int maxs, int* sweeps, double* res) {
  OWN_TEAM(c);
  return lift(c, *T, team_initialization_converged(*T, tol, maxs, sweeps, res));
}
int team_ctx_reduce(Ctx& c, double* value, bool is_max) {
  OWN_TEAM(c);
  double out = 0.0;
  int rc = team_reduce(*T, {*value}, is_max, &out);
  if (rc == EKPNP_OK) *value = out;
  return lift(c, *T, rc);
}
int team_ctx_turns(Ctx& c, int (*fn)(Ctx&, void*), void* arg) { OWN_TEAM(c); return lift(c, *T, team_turns(*T, fn, arg)); }
int team_ctx_tune(Ctx& c, const char* knob, int value) { OWN_TEAM(c); return lift(c, *T, team_tune(*T, knob, value)); }

bool team_is_group(const Ctx& c) { return c.team && c.team->group; }

void team_timing_reset(Ctx& c) {
  if (!c.team) return;
  Team& T = *c.team;
  for (int x = 0; x < X_KINDS; ++x)
    if ((size_t)c.team_slot < T.xev_used[x].size()) T.xev_used[x][c.team_slot] = T.xev_begun[x][c.team_slot] = 0;
}

// sums over the exchanges of kind x bracketed since the last reset, for the slab in slot c.team_slot
int team_comm_timing_get(Ctx& c, int x, int* n, double* wait_ms, double* transfer_ms, size_t* bytes_sent) {
  if (!c.team) { c.err = "ekpnp_comm_timing_get: this context has no transport"; return EKPNP_ERR_INVALID; }
  if (x < 0 || x >= X_KINDS) { c.err = "ekpnp_comm_timing_get: unknown exchange kind"; return EKPNP_ERR_INVALID; }
  Team& T = *c.team;
  const int i = c.team_slot;
  hipError_t e = hipSetDevice(c.device);
  if (e == hipSuccess) e = hipStreamSynchronize(T.cs[i]);
  if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
  double w = 0.0, t = 0.0;
  const size_t used = T.xev_used[x][i];
  for (size_t k = 0; k < used && e == hipSuccess; ++k) {
    const Team::XEv& v = T.xev[x][i][k];
    float ms = 0.f;
    e = hipEventElapsedTime(&ms, v.wait_begin, v.wait_end);
    w += ms;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, v.xfer_begin, v.xfer_end);
    t += ms;
  }
  if (e != hipSuccess) { c.err = std::string("ekpnp_comm_timing_get: ") + hipGetErrorString(e); return EKPNP_ERR_HIP; }
  if (n) *n = (int)used;
  if (wait_ms) *wait_ms = w;
  if (transfer_ms) *transfer_ms = t;
  if (bytes_sent) {
    const size_t edge = 4 * (size_t)c.p.ny * c.nxh / (size_t)edge_chunk_count(c);  // per exchange: one mode block (their mean size)
    *bytes_sent = sizeof(double) * (x == X_HALO ? 2 * c.halo_doubles : x == X_PHI ? 2 * c.plane : edge);
  }
  T.xev_used[x][i] = T.xev_begun[x][i] = 0;
  return EKPNP_OK;
}

void team_detach(Ctx& c) {
  if (!c.team || c.team->group) return;  // a group releases its own team (ekpnp_group_destroy)
  Team* T = c.team;
  team_release(*T);
  c.team = nullptr;
  delete T;
}

}  // namespace ekpnp

using namespace ekpnp;

// ---- C ABI: one process per GPU --------------------------------------------------------------------

// a group call visits several devices; the caller gets its current device back
struct DeviceGuard {
  int prev = -1;
  DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};


extern "C" int ekpnp_comm_unique_id(void* id128) {
  if (!id128) return EKPNP_ERR_INVALID;
  // no context here: the message goes where ekpnp_create's goes, ekpnp_last_error(NULL)
  std::string err;
  Rccl* nc = rccl(err);
  if (!nc) { set_create_error("ekpnp_comm_unique_id: " + err); return EKPNP_ERR_HIP; }
  ncclUniqueId id;
  const ncclResult_t r = nc->GetUniqueId(&id);
  if (r != ncclSuccess) { set_create_error(std::string("ncclGetUniqueId: ") + nc->GetErrorString(r)); return EKPNP_ERR_HIP; }
  static_assert(sizeof(id) == EKPNP_UNIQUE_ID_BYTES, "ncclUniqueId size");
  std::memcpy(id128, &id, sizeof id);
  return EKPNP_OK;
}

extern "C" int ekpnp_rccl_available(void) {
  std::string err;
  if (rccl(err)) return EKPNP_OK;
  set_create_error("ekpnp_rccl_available: " + err);
  return EKPNP_ERR_HIP;
}

// How many ranks of the lattice run on this rank's device?  One all-gather of (host, device) identities over the fresh
// communicator: the host is the machine's boot id + host name (NCCL_HOSTID, which the one-GPU rehearsals set to make RCCL
// accept several ranks per device, is deliberately NOT what is compared), the device its PCI bus id.
static uint64_t fnv1a(const void* data, size_t n, uint64_t h = 1469598103934665603ull) {
  const unsigned char* b = static_cast<const unsigned char*>(data);
  for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}
static int count_ranks_on_my_device(Team& T, Ctx& c) {
  c.ranks_on_device = 1;
  if (c.nranks == 1) return EKPNP_OK;
  uint64_t me[2] = {0, 0};
  {
    char buf[256] = {0};
    if (gethostname(buf, sizeof buf - 1) == 0) me[0] = fnv1a(buf, std::strlen(buf));
    if (FILE* f = std::fopen("/proc/sys/kernel/random/boot_id", "r")) {
      char id[64] = {0};
      if (std::fgets(id, sizeof id, f)) me[0] = fnv1a(id, std::strlen(id), me[0] ? me[0] : 1469598103934665603ull);
      std::fclose(f);
    }
    char bus[64] = {0};
    THIP(T, hipDeviceGetPCIBusId(bus, sizeof bus, c.device));
    me[1] = fnv1a(bus, std::strlen(bus));
  }
  hipStream_t st = !T.cs.empty() && T.cs[0] ? T.cs[0] : nullptr;  // (a rank whose stream set-up failed still answers, on the null stream)
  uint64_t* d_all = nullptr;
  THIP(T, hipMalloc((void**)&d_all, (size_t)(c.nranks + 1) * sizeof me));
  std::vector<uint64_t> all((size_t)c.nranks * 2, 0);
  hipError_t e = hipMemcpyAsync(d_all + (size_t)c.nranks * 2, me, sizeof me, hipMemcpyHostToDevice, st);
  ncclResult_t r = ncclSuccess;
  if (e == hipSuccess) r = T.nc->AllGather(d_all + (size_t)c.nranks * 2, d_all, 2, ncclUint64, T.comm[0], st);
  if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(all.data(), d_all, all.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d_all);
  if (r != ncclSuccess) { T.err = std::string("ncclAllGather (device identities): ") + T.nc->GetErrorString(r); return EKPNP_ERR_HIP; }
  THIP(T, e);
  int same = 0;
  for (int k = 0; k < c.nranks; ++k) same += all[(size_t)2 * k] == me[0] && all[(size_t)2 * k + 1] == me[1];
  c.ranks_on_device = same < 1 ? 1 : same;
  return EKPNP_OK;
}

extern "C" int ekpnp_slab_attach_comm(ekpnp_ctx* ctx, const void* id128) {
  if (!ctx) return EKPNP_ERR_INVALID;
  Ctx& c = ctx->c;
  if (!c.slab) { c.err = "ekpnp_slab_attach_comm needs a context made by ekpnp_create_slab"; return EKPNP_ERR_INVALID; }
  if (c.team) { c.err = "this slab already has a transport"; return EKPNP_ERR_INVALID; }
  if (!id128) { c.err = "NULL id"; return EKPNP_ERR_INVALID; }
  Rccl* nc = rccl(c.err);
  if (!nc) return EKPNP_ERR_HIP;
  Team* T = new (std::nothrow) Team();
  if (!T) { c.err = "host allocation failed"; return EKPNP_ERR_NOMEM; }
  T->m = {ctx};
  T->nranks = c.nranks;
  // one rank: everything is local and host-side combining would do; EKPNP_TEAM_FORCE_COLLECTIVES makes the
  // reductions and the file-IO turns take the multi-process route anyway (ncclAllReduce over the one
  // rank), so that those code paths can be exercised on a one-GPU box (tests/test_group_gpu.py)
  T->all_local = c.nranks == 1 && std::getenv("EKPNP_TEAM_FORCE_COLLECTIVES") == nullptr;
  T->group = false;
  T->kind = EKPNP_TRANSPORT_RCCL;
  T->nc = nc;
  int rc = team_make_streams(*T);
  {
    // ncclCommInitRank is collective: a rank whose local set-up failed still enters it (and destroys the
    // communicator again below), otherwise its peers would wait in theirs for ever
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    T->comm.assign(1, nullptr);
    ncclResult_t r = nc->CommInitRank(&T->comm[0], c.nranks, id, c.rank);
    if (r != ncclSuccess) {
      if (rc == EKPNP_OK) { T->err = std::string("ncclCommInitRank: ") + nc->GetErrorString(r); rc = EKPNP_ERR_HIP; }
      T->comm.clear();
    }
  }
  // collective as well: every rank that HAS a communicator takes part, also one whose local set-up failed (its peers must return)
  if (!T->comm.empty() && T->comm[0]) {
    const int crc = count_ranks_on_my_device(*T, c);
    if (rc == EKPNP_OK) rc = crc;
  }
  if (rc) {
    c.err = T->err;
    team_release(*T);
    delete T;
    return rc;
  }
  c.team = T;
  c.team_slot = 0;
  if (c.ranks_on_device > 1 && std::getenv("GPU_MAX_HW_QUEUES") == nullptr) {
    // Several PROCESSES on one device (rehearsals, tests): each HIP process opens up to 4 hardware queues for its streams plus
    // one per priority level, RCCL adds its own, and four such processes oversubscribe the device's hardware queue slots - the
    // scheduler then time-slices QUEUES with a coarse quantum and every small kernel of a step waits tens of milliseconds for
    // its queue's turn (4 ranks on one MI355X: 0.75 - 1.3 s per step against 43 - 50 ms; the per-stage times of the solve show
    // the stretch in stage 1 and stage 2, which contain no exchange: profiles/r05_shared_device_experiments.log).
    // GPU_MAX_HW_QUEUES=1 in the ranks' environment (read by the HIP runtime when it starts) cures it.  The library can only say so.
    static bool said = false;
    if (!said) {
      said = true;
      std::fprintf(stderr, "ekpnp: %d ranks of this lattice share device %d: set GPU_MAX_HW_QUEUES=1 in the ranks' environment, or the "
                           "processes oversubscribe the device's hardware queues (steps 10 - 30x slower; include/ekpnp.h: ekpnp_plane_transforms)\n",
                   c.ranks_on_device, c.device);
    }
  }
  return EKPNP_OK;
}

extern "C" int ekpnp_comm_timing_get(ekpnp_ctx* ctx, int kind, int* n_exchanges, double* wait_ms, double* transfer_ms, size_t* bytes_sent) {
  if (!ctx) return EKPNP_ERR_INVALID;
  DeviceGuard device_guard_;
  return team_comm_timing_get(ctx->c, kind, n_exchanges, wait_ms, transfer_ms, bytes_sent);
}

// ---- C ABI: one process, N slabs (ekpnp_main --gpus N) -------------------------------------------

struct ekpnp_group {
  Team t;
};

static thread_local std::string g_group_err;

extern "C" const char* ekpnp_group_last_error(const ekpnp_group* g) { return g ? g->t.err.c_str() : g_group_err.c_str(); }

extern "C" int ekpnp_group_destroy(ekpnp_group* g) {
  if (!g) return EKPNP_ERR_INVALID;
  int prev = -1;
  if (hipGetDevice(&prev) != hipSuccess) prev = -1;
  Team& T = g->t;
  team_release(T);
  for (ekpnp_ctx* m : T.m) {
    (void)hipSetDevice(m->c.device);
    m->c.team = nullptr;
    (void)ekpnp_destroy(m);
  }
  delete g;
  if (prev >= 0) (void)hipSetDevice(prev);
  return EKPNP_OK;
}

extern "C" int ekpnp_group_create(const ekpnp_params* p, int nslabs, const int* devices, int transport, ekpnp_group** out) {
  if (!out) { g_group_err = "out is NULL"; return EKPNP_ERR_INVALID; }
  *out = nullptr;
  if (!p || nslabs < 1 || nslabs > 16) { g_group_err = "ekpnp_group_create: 1 to 16 slabs"; return EKPNP_ERR_INVALID; }
  if (transport != EKPNP_TRANSPORT_AUTO && transport != EKPNP_TRANSPORT_RCCL && transport != EKPNP_TRANSPORT_COPY) {
    g_group_err = "ekpnp_group_create: unknown transport";
    return EKPNP_ERR_INVALID;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_group_err = "no HIP device"; return EKPNP_ERR_HIP; }
  std::vector<int> dev(nslabs);
  bool distinct = true;
  for (int i = 0; i < nslabs; ++i) {
    dev[i] = devices ? devices[i] : i % ndev;
    if (dev[i] < 0 || dev[i] >= ndev) { g_group_err = "ekpnp_group_create: device index out of range"; return EKPNP_ERR_INVALID; }
    for (int j = 0; j < i; ++j) distinct = distinct && dev[j] != dev[i];
  }
  // RCCL refuses two ranks on one device: slabs that share a device move their halos by copies
  int kind = transport;
  if (kind == EKPNP_TRANSPORT_AUTO) kind = (distinct && nslabs > 1) ? EKPNP_TRANSPORT_RCCL : EKPNP_TRANSPORT_COPY;
  if (kind == EKPNP_TRANSPORT_RCCL && !distinct) {
    g_group_err = "EKPNP_TRANSPORT_RCCL needs one device per slab (RCCL refuses two ranks on one device): use EKPNP_TRANSPORT_COPY";
    return EKPNP_ERR_INVALID;
  }
  ekpnp_group* g = new (std::nothrow) ekpnp_group();
  if (!g) { g_group_err = "host allocation failed"; return EKPNP_ERR_NOMEM; }
  Team& T = g->t;
  T.nranks = nslabs;
  T.all_local = true;
  T.group = true;
  T.kind = kind;
  auto bail = [&](int code) {
    g_group_err = T.err;
    ekpnp_group_destroy(g);
    return code;
  };
  DeviceGuard device_guard_;
  for (int i = 0; i < nslabs; ++i) {
    if (hipSetDevice(dev[i]) != hipSuccess) { T.err = "hipSetDevice failed"; return bail(EKPNP_ERR_HIP); }
    ekpnp_ctx* c = nullptr;
    int rc = ekpnp_create_slab(p, i, nslabs, &c);
    if (rc) { T.err = std::string("slab ") + std::to_string(i) + ": " + ekpnp_last_error(nullptr); return bail(rc); }
    c->c.team = &T;
    c->c.team_slot = i;
    T.m.push_back(c);
  }
  int rc = team_make_streams(T);
  if (rc) return bail(rc);
  if (kind == EKPNP_TRANSPORT_RCCL) {
    T.nc = rccl(T.err);
    if (!T.nc) return bail(EKPNP_ERR_HIP);
    T.comm.assign(nslabs, nullptr);
    ncclResult_t r = T.nc->CommInitAll(T.comm.data(), nslabs, dev.data());
    if (r != ncclSuccess) { T.err = std::string("ncclCommInitAll: ") + T.nc->GetErrorString(r); T.comm.clear(); return bail(EKPNP_ERR_HIP); }
  } else if (distinct && nslabs > 1) {
    // direct peer copies over xGMI where the devices allow it (otherwise HIP stages through the host)
    for (int i = 0; i < nslabs; ++i)
      for (int j = 0; j < nslabs; ++j) {
        int can = 0;
        if (i != j && hipDeviceCanAccessPeer(&can, dev[i], dev[j]) == hipSuccess && can) {
          (void)hipSetDevice(dev[i]);
          hipError_t e = hipDeviceEnablePeerAccess(dev[j], 0);
          if (e != hipSuccess) (void)hipGetLastError();  // already enabled is fine
        }
      }
  }
  *out = g;
  return EKPNP_OK;
}

#define NEEDGROUP(g)                    \
  if (!(g)) return EKPNP_ERR_INVALID;   \
  DeviceGuard device_guard_;            \
  Team& T = (g)->t
// ... and the group has not been poisoned by an earlier failure (every verb that computes, exchanges or reads device state)
#define NEEDLIVEGROUP(g)                                                                              \
  NEEDGROUP(g);                                                                                       \
  if (T.poisoned) {                                                                                   \
    T.err = "this group is poisoned (destroy it; ekpnp_group_last_error keeps the cause): " + T.poison; \
    return EKPNP_ERR_INVALID;                                                                         \
  }

// Failure of a group verb.  In an in-process group every slab is driven by this one host thread, so
// a non-OK status from one slab is fully local: nobody is waiting inside a collective for a peer process.  But the other
// slabs have work enqueued that the failed slab's part never joined (one has collided, the next has not; an exchange was
// begun or not).  The call therefore returns only after EVERY slab's compute and comm stream has drained (no kernel of
// the group is running when the caller sees the error), the slabs' per-step state is reset, and the group is POISONED:
// its fields are from mixed steps, so every further verb answers EKPNP_ERR_INVALID with the first failure, and only
// ekpnp_group_destroy / _last_error / _size / _transport / _context remain.  Attached ranks (one process per GPU) keep
// the documented contract: the control plane ends all ranks.  The drain is SKIPPED when the failure was an RCCL call inside a
// collective (comm_broken): its kernels may wait for ever, the communicators are aborted at destroy instead (team_release).
static int group_fail(Team& T, int rc) {
  if (rc == EKPNP_OK || !T.group) return rc;
  const std::string first = T.err;
  for (size_t i = 0; i < T.m.size(); ++i) {
    Ctx& c = S(T, (int)i);
    if (hipSetDevice(c.device) != hipSuccess) continue;
    if (!T.comm_broken && i < T.cs.size() && T.cs[i]) (void)hipStreamSynchronize(T.cs[i]);
    if (!T.comm_broken && c.stream) (void)hipStreamSynchronize(c.stream);
    (void)hipGetLastError();
    c.launch_err = hipSuccess;
    c.launch_what = nullptr;
    c.collide_phase = 0;
  }
  if (!T.poisoned) {
    T.poisoned = true;
    T.poison = first;
  }
  T.err = first;
  return rc;
}

extern "C" int ekpnp_group_size(const ekpnp_group* g) { return g ? (int)g->t.m.size() : 0; }
extern "C" int ekpnp_group_transport(const ekpnp_group* g) { return g ? g->t.kind : 0; }

extern "C" int ekpnp_group_context(ekpnp_group* g, int slab, ekpnp_ctx** ctx) {
  NEEDGROUP(g);
  if (!ctx || slab < 0 || slab >= (int)T.m.size()) { T.err = "bad slab index"; return EKPNP_ERR_INVALID; }
  *ctx = T.m[slab];
  return EKPNP_OK;
}

extern "C" size_t ekpnp_group_device_bytes(const ekpnp_group* g) {
  size_t b = 0;
  if (g) for (ekpnp_ctx* m : g->t.m) b += ekpnp_device_bytes(m);
  return b;
}

extern "C" int ekpnp_group_synchronize(ekpnp_group* g) { NEEDLIVEGROUP(g); return group_fail(T, team_synchronize(T)); }
extern "C" int ekpnp_group_initialization(ekpnp_group* g) { NEEDLIVEGROUP(g); return group_fail(T, team_initialization(T)); }
extern "C" int ekpnp_group_initialization_converged(ekpnp_group* g, double rel_tol, int max_sweeps, int* sweeps, double* residual) {
  NEEDLIVEGROUP(g);
  return group_fail(T, team_initialization_converged(T, rel_tol, max_sweeps, sweeps, residual));
}
static int team_init_equilibrium(Team& T) {
  for (size_t i = 0; i < T.m.size(); ++i) {
    int rc = use(T, (int)i);
    if (rc) return rc;
    TSLAB(T, (int)i, ekpnp_init_equilibrium(T.m[i]));
  }
  return EKPNP_OK;
}
extern "C" int ekpnp_group_init_equilibrium(ekpnp_group* g) { NEEDLIVEGROUP(g); return group_fail(T, team_init_equilibrium(T)); }
extern "C" int ekpnp_group_stream_collide_save(ekpnp_group* g, double t) { NEEDLIVEGROUP(g); (void)t; return group_fail(T, team_stream_collide_save(T)); }
extern "C" int ekpnp_group_fast_poisson(ekpnp_group* g) { NEEDLIVEGROUP(g); return group_fail(T, team_fast_poisson(T)); }
extern "C" int ekpnp_group_step(ekpnp_group* g, int nsteps) { NEEDLIVEGROUP(g); return group_fail(T, team_step(T, nsteps)); }

extern "C" int ekpnp_group_tune(ekpnp_group* g, const char* knob, int value) {
  NEEDLIVEGROUP(g);
  if (!knob) { T.err = "knob is NULL"; return EKPNP_ERR_INVALID; }
  if (std::strcmp(knob, "inline_exchanges") == 0 || std::strcmp(knob, "comm_cus") == 0 || std::strcmp(knob, "edge_p2p") == 0) return group_fail(T, team_tune(T, knob, value));
  for (size_t i = 0; i < T.m.size(); ++i) {  // a per-slab knob: the same on every slab (an invalid one is refused by the first, nothing changed)
    int rc = use(T, (int)i);
    if (rc) return rc;
    if ((rc = ctx_tune(S(T, (int)i), knob, value))) { T.err = S(T, (int)i).err; return rc; }
  }
  return EKPNP_OK;
}

extern "C" int ekpnp_group_get_time(ekpnp_group* g, double* t) {
  NEEDGROUP(g);
  if (!t) { T.err = "NULL pointer"; return EKPNP_ERR_INVALID; }
  return ekpnp_get_time(T.m[0], t);
}
extern "C" int ekpnp_group_set_time(ekpnp_group* g, double t) {
  NEEDGROUP(g);
  for (ekpnp_ctx* m : T.m) (void)ekpnp_set_time(m, t);
  return EKPNP_OK;
}

// whole-lattice host arrays [NZ][NY][NX] <-> the slabs' planes
extern "C" int ekpnp_group_set_field(ekpnp_group* g, int field_id, const double* host) {
  NEEDLIVEGROUP(g);
  if (!host) { T.err = "NULL pointer"; return EKPNP_ERR_INVALID; }
  for (size_t i = 0; i < T.m.size(); ++i) {
    int rc = use(T, (int)i);
    if (rc) return rc;
    TSLAB(T, (int)i, ekpnp_set_field(T.m[i], field_id, host + (size_t)S(T, (int)i).z0 * S(T, (int)i).plane));
  }
  return EKPNP_OK;
}
extern "C" int ekpnp_group_get_field(ekpnp_group* g, int field_id, double* host) {
  NEEDLIVEGROUP(g);
  if (!host) { T.err = "NULL pointer"; return EKPNP_ERR_INVALID; }
  for (size_t i = 0; i < T.m.size(); ++i) {
    int rc = use(T, (int)i);
    if (rc) return rc;
    THIP(T, hipStreamSynchronize(T.cs[i]));
    TSLAB(T, (int)i, ekpnp_get_field(T.m[i], field_id, host + (size_t)S(T, (int)i).z0 * S(T, (int)i).plane));
  }
  return EKPNP_OK;
}

// diagnostics (main.cu:211-222): every slab reduces its own planes on its device, the host combines
static int group_diag(Team& T, bool is_max, double* out) {
  std::vector<double> v(T.m.size(), 0.0);
  for (size_t i = 0; i < T.m.size(); ++i) {
    int rc = use(T, (int)i);
    if (rc) return rc;
    // a group member's ekpnp_current / ekpnp_umax return the slab's own value (no team reduction)
    TSLAB(T, (int)i, is_max ? ekpnp_umax(T.m[i], &v[i]) : ekpnp_current(T.m[i], &v[i]));
  }
  return team_reduce(T, v, is_max, out);
}
extern "C" int ekpnp_group_current(ekpnp_group* g, double* I) {
  NEEDLIVEGROUP(g);
  if (!I) { T.err = "NULL pointer"; return EKPNP_ERR_INVALID; }
  return group_diag(T, false, I);
}
extern "C" int ekpnp_group_umax(ekpnp_group* g, double* umax) {
  NEEDLIVEGROUP(g);
  if (!umax) { T.err = "NULL pointer"; return EKPNP_ERR_INVALID; }
  return group_diag(T, true, umax);
}
extern "C" int ekpnp_group_record_umax(ekpnp_group* g, const char* path, int append, double time) {
  NEEDLIVEGROUP(g);
  if (!path) { T.err = "NULL path"; return EKPNP_ERR_INVALID; }
  double um = 0.0;
  int rc = group_diag(T, true, &um);
  if (rc) return rc;
  FILE* f = std::fopen(path, append ? "ab" : "wb");
  if (!f) { T.err = "cannot open umax file"; return EKPNP_ERR_INVALID; }
  std::fprintf(f, "%10.6f %10.6f\n", time, um);  // LBM.cu:2748
  std::fclose(f);
  return EKPNP_OK;
}

// whole-lattice text files: every slab appends its planes in rank order
static int write_part(Ctx& c, void* arg) {
  TextIoArgs a = *static_cast<TextIoArgs*>(arg);
  if (c.rank != 0) a.append = 1;
  return io_write_text_part(c, a);
}
extern "C" int ekpnp_group_save_data_tecplot(ekpnp_group* g, const char* path, int append, double time, int first) {
  NEEDLIVEGROUP(g);
  if (!path) { T.err = "NULL path"; return EKPNP_ERR_INVALID; }
  TextIoArgs a{path, append, time, first, 0};
  return team_turns(T, write_part, &a);
}
extern "C" int ekpnp_group_save_data_end(ekpnp_group* g, const char* path, int append, double time) {
  NEEDLIVEGROUP(g);
  if (!path) { T.err = "NULL path"; return EKPNP_ERR_INVALID; }
  TextIoArgs a{path, append, time, 0, 1};
  return team_turns(T, write_part, &a);
}
namespace {
struct ReadArgs { const char* path; double* time; };
int read_part(Ctx& c, void* arg) {
  ReadArgs* a = static_cast<ReadArgs*>(arg);
  return io_read_data_part(c, a->path, a->time);
}
}  // namespace
extern "C" int ekpnp_group_read_data(ekpnp_group* g, const char* path, double* time) {
  NEEDLIVEGROUP(g);
  if (!path || !time) { T.err = "NULL pointer"; return EKPNP_ERR_INVALID; }
  ReadArgs a{path, time};
  return team_turns(T, read_part, &a);
}

// whole-lattice EKPNPST1 file (same format a single context writes: z0 = 0, nz_local = nz)
extern "C" int ekpnp_group_save_state(ekpnp_group* g, const char* path, double time) {
  NEEDLIVEGROUP(g);
  if (!path) { T.err = "NULL path"; return EKPNP_ERR_INVALID; }
  int rc = team_synchronize(T);
  if (rc) return rc;
  FILE* f = std::fopen(path, "wb");
  if (!f) { T.err = "cannot open state file"; return EKPNP_ERR_INVALID; }
  const ekpnp_params& p = S(T, 0).p;
  StateHeader h{};
  std::memcpy(h.magic, "EKPNPST1", 8);
  h.nx = p.nx; h.ny = p.ny; h.nz = p.nz; h.z0 = 0; h.nzl = p.nz; h.nfields = EKPNP_NFIELDS;
  h.time = time;
  bool ok = std::fwrite(&h, sizeof h, 1, f) == 1;
  std::vector<double> buf(STATE_CHUNK);
  hipError_t e = hipSuccess;
  for (int id = 0; ok && e == hipSuccess && id < EKPNP_NFIELDS; ++id)
    for (size_t i = 0; ok && e == hipSuccess && i < T.m.size(); ++i) {
      Ctx& c = S(T, (int)i);
      e = hipSetDevice(c.device);
      for (size_t o = 0; ok && e == hipSuccess && o < c.nloc; o += buf.size()) {
        const size_t n = c.nloc - o < buf.size() ? c.nloc - o : buf.size();
        e = hipMemcpy(buf.data(), c.fld[id] + o, n * sizeof(double), hipMemcpyDeviceToHost);
        if (e == hipSuccess) ok = std::fwrite(buf.data(), sizeof(double), n, f) == n;
      }
    }
  ok = (std::fclose(f) == 0) && ok;
  THIP(T, e);
  if (!ok) { T.err = "write error on state file"; return EKPNP_ERR_INVALID; }
  return EKPNP_OK;
}

extern "C" int ekpnp_group_read_state(ekpnp_group* g, const char* path, double* time) {
  NEEDLIVEGROUP(g);
  if (!path || !time) { T.err = "NULL pointer"; return EKPNP_ERR_INVALID; }
  int rc = team_synchronize(T);
  if (rc) return rc;
  FILE* f = std::fopen(path, "rb");
  if (!f) { T.err = "cannot open state file"; return EKPNP_ERR_INVALID; }
  const ekpnp_params& p = S(T, 0).p;
  StateHeader h{};
  if (std::fread(&h, sizeof h, 1, f) != 1 || std::memcmp(h.magic, "EKPNPST1", 8) != 0) {
    std::fclose(f);
    T.err = "not an EKPNPST1 state file";
    return EKPNP_ERR_INVALID;
  }
  if (h.nx != p.nx || h.ny != p.ny || h.nz != p.nz || h.z0 != 0 || h.nzl != p.nz || h.nfields != EKPNP_NFIELDS) {
    std::fclose(f);
    T.err = "state file does not hold this whole lattice";
    return EKPNP_ERR_INVALID;
  }
  std::vector<double> buf(STATE_CHUNK);
  bool ok = true;
  hipError_t e = hipSuccess;
  for (int id = 0; ok && e == hipSuccess && id < EKPNP_NFIELDS; ++id)
    for (size_t i = 0; ok && e == hipSuccess && i < T.m.size(); ++i) {
      Ctx& c = S(T, (int)i);
      e = hipSetDevice(c.device);
      for (size_t o = 0; ok && e == hipSuccess && o < c.nloc; o += buf.size()) {
        const size_t n = c.nloc - o < buf.size() ? c.nloc - o : buf.size();
        ok = std::fread(buf.data(), sizeof(double), n, f) == n;
        if (ok) e = hipMemcpy(c.fld[id] + o, buf.data(), n * sizeof(double), hipMemcpyHostToDevice);
      }
      c.rhs_ready = false;
      c.e_phi_valid = false;  // phi and E are what the file said (team_synchronize above brought the arrays up to date first)
      c.t = h.time;
    }
  std::fclose(f);
  THIP(T, e);
  if (!ok) { T.err = "state file is shorter than the lattice"; return EKPNP_ERR_INVALID; }
  *time = h.time;
  return EKPNP_OK;
}

// whole-lattice EKPNPCK2 checkpoint (fields + post-collision populations): the file a single
// context writes with ekpnp_save_checkpoint; loading continues the run bit for bit
extern "C" int ekpnp_group_save_checkpoint(ekpnp_group* g, const char* path) {
  NEEDLIVEGROUP(g);
  if (!path) { T.err = "NULL path"; return EKPNP_ERR_INVALID; }
  int rc = team_synchronize(T);
  if (rc) return rc;
  FILE* f = std::fopen(path, "wb");
  if (!f) { T.err = "cannot open checkpoint file"; return EKPNP_ERR_INVALID; }
  const int n = (int)T.m.size();
  rc = io_ckpt_write_header(S(T, 0), f, 0, S(T, 0).p.nz, 0, S(T, 0).t);
  if (rc) T.err = S(T, 0).err;
  for (int id = 0; rc == EKPNP_OK && id < EKPNP_NFIELDS; ++id)
    for (int i = 0; rc == EKPNP_OK && i < n; ++i)
      if ((rc = use(T, i)) == EKPNP_OK && (rc = io_ckpt_fields(S(T, i), f, id, 0))) T.err = S(T, i).err;
  for (int l = 0; rc == EKPNP_OK && l < S(T, 0).p.n_lattices; ++l)
    for (int i = 0; rc == EKPNP_OK && i < n; ++i)
      if ((rc = use(T, i)) == EKPNP_OK && (rc = io_ckpt_populations(S(T, i), f, l, 0, 0))) T.err = S(T, i).err;
  if (std::fclose(f) != 0 && rc == EKPNP_OK) { T.err = "write error on checkpoint file"; rc = EKPNP_ERR_INVALID; }
  return rc;
}

extern "C" int ekpnp_group_load_checkpoint(ekpnp_group* g, const char* path, double* time) {
  NEEDLIVEGROUP(g);
  if (!path) { T.err = "NULL path"; return EKPNP_ERR_INVALID; }
  int rc = team_synchronize(T);
  if (rc) return rc;
  FILE* f = std::fopen(path, "rb");
  if (!f) { T.err = "cannot open checkpoint file"; return EKPNP_ERR_INVALID; }
  const int n = (int)T.m.size();
  CkptHeader h{};
  if (std::fread(&h, sizeof h, 1, f) != 1) { std::fclose(f); T.err = "not an EKPNPCK2 checkpoint file"; return EKPNP_ERR_INVALID; }
  rc = io_ckpt_check_header(S(T, 0), h, 0, S(T, 0).p.nz);
  if (rc) T.err = S(T, 0).err;
  if (rc == EKPNP_OK && h.with_ghosts) { T.err = "this is a slab's per-rank checkpoint, not a whole-lattice one"; rc = EKPNP_ERR_INVALID; }
  for (int id = 0; rc == EKPNP_OK && id < EKPNP_NFIELDS; ++id)
    for (int i = 0; rc == EKPNP_OK && i < n; ++i)
      if ((rc = use(T, i)) == EKPNP_OK && (rc = io_ckpt_fields(S(T, i), f, id, 1))) T.err = S(T, i).err;
  for (int l = 0; rc == EKPNP_OK && l < S(T, 0).p.n_lattices; ++l)
    for (int i = 0; rc == EKPNP_OK && i < n; ++i)
      if ((rc = use(T, i)) == EKPNP_OK && (rc = io_ckpt_populations(S(T, i), f, l, 0, 1))) T.err = S(T, i).err;
  std::fclose(f);
  if (rc) return rc;
  // ghost planes: what the halo exchange of the interrupted run had left there - the neighbouring
  // slab's edge plane (only its 9 z-crossing directions are ever read), ring-closed (LBM.cu:1972,1975)
  for (int i = 0; i < n; ++i) {
    Ctx& c = S(T, i);
    io_ckpt_finish_load(c, h);
    Ctx &below = S(T, (i + n - 1) % n), &above = S(T, (i + 1) % n);
    if ((rc = use(T, i))) return rc;
    for (int l = 0; l < c.p.n_lattices; ++l) {
      const size_t bytes = c.pplane * sizeof(double);
      THIP(T, xcopy(c.cur_base(l), c.device, below.cur_base(l) + (size_t)below.nzl * below.pplane, below.device, bytes, c.stream));
      THIP(T, xcopy(c.cur_base(l) + (size_t)(c.nzl + 1) * c.pplane, c.device, above.cur_base(l) + above.pplane, above.device, bytes, c.stream));
    }
  }
  if ((rc = team_synchronize(T))) return rc;
  if (time) *time = h.time;
  return EKPNP_OK;
}
