/*
 * ekpnp.h — C ABI of the MI355X-native EK-PNP hot path.
 *
 * This header is the drop-in boundary for the reference's host step API
 * (gyf135/EK-PNP-3D, prototypes LBM.h:159-180, called from main.cu:163-198).
 * Every entry point below names the reference interface it replaces.
 *
 * Conventions
 *   - plain C, no C++/torch types; all pointers are raw host or device pointers
 *   - every call returns an int status (EKPNP_OK == 0); the message of the last
 *     failure is available through ekpnp_last_error(); nothing ever calls exit()
 *     (the reference prints and exit()s: LBM.cu:35-53, LBM.h:187-208)
 *   - a context owns all device memory, FFT plans, streams and events; there are
 *     no globals (the reference keeps global device pointers, LBM.h:131-142)
 *   - macroscopic fields use the reference scalar layout [NZ][NY][NX], x fastest
 *     (LBM.cu:22-25); the population layout is private to the implementation
 *   - every call is complete-on-return with respect to the context's stream
 *     unless stated otherwise (ekpnp_step / ekpnp_stream_collide_save /
 *     ekpnp_fast_poisson only enqueue; ekpnp_synchronize waits)
 */
#ifndef EKPNP_H
#define EKPNP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EKPNP_OK 0
#define EKPNP_ERR_INVALID 1   /* bad argument / bad state                     */
#define EKPNP_ERR_HIP 2       /* a HIP runtime call failed                    */
#define EKPNP_ERR_FFT 3       /* a hipFFT call failed                         */
#define EKPNP_ERR_NOMEM 4     /* device allocation failed                     */

/* Macroscopic field ids, in the argument order of the reference's
 * initialization(r, c, cn, fi, u, v, w, ex, ey, ez, temp) (LBM.h:159). */
enum {
  EKPNP_RHO = 0,
  EKPNP_C = 1,   /* cation concentration  (charge_gpu)  */
  EKPNP_CN = 2,  /* anion concentration   (chargen_gpu) */
  EKPNP_PHI = 3,
  EKPNP_UX = 4,
  EKPNP_UY = 5,
  EKPNP_UZ = 6,
  EKPNP_EX = 7,
  EKPNP_EY = 8,
  EKPNP_EZ = 9,
  EKPNP_T = 10,
  EKPNP_NFIELDS = 11
};

/* Runtime replacement of the compile-time grid and the __constant__/__device__
 * physics globals of LBM.h:29-118 (set at main.cu:23-35). */
typedef struct ekpnp_params {
  int32_t nx, ny, nz;       /* global lattice, LBM.h:32-35                      */
  int32_t n_lattices;       /* 4: f,h,hn,temp (reference). 3: f,h,hn (parity-   */
                            /* safe iff Ra == 0). 1: f only (iff chargeinf==0   */
                            /* and Ra == 0)                                     */
  int32_t pb_iterations;    /* Poisson-Boltzmann sweeps in initialization();    */
                            /* the reference loops i = 0..500 -> 501            */
  int32_t in_place;         /* 0: two population buffers (A/B, fastest). 1: ONE buffer,   */
                            /* each sweep writes the lattice shifted by min(nz/4,64)+1      */
                            /* planes in z and the shift direction alternates: 2 x 216 N    */
                            /* -> ~1.13 x 216 N bytes of populations per lattice at nz=512, */
                            /* same results bit for bit                                     */
  double Lx, Ly, Lz;        /* LBM.h:40-42; Lx=nx*dx, Ly=ny*dy, Lz=(nz-1)*dz    */
  double dx, dy, dz;        /* LBM.h:43-45                                      */
  double CFL;               /* LBM.h:51                                         */
  double dt;                /* LBM.h:52                                         */
  double cs_square;         /* LBM.h:53  = 1/3/CFL^2                            */
  double rho0;              /* LBM.h:54                                         */
  double chargeinf;         /* LBM.h:56                                         */
  double voltage, voltage2; /* LBM.h:60,62                                      */
  double Ext;               /* LBM.h:64                                         */
  double eps;               /* LBM.h:65                                         */
  double diffu, diffun;     /* LBM.h:66,73                                      */
  double nu;                /* LBM.h:67-68                                      */
  double K, Kn;             /* LBM.h:69-70,75-76                                */
  double D, Ra, TH;         /* LBM.h:97-99                                      */
  double uw, exf;           /* LBM.h:47-50                                      */
  double kB, electron, roomT;   /* LBM.h:87-89                                  */
  double convertCtoCharge;  /* LBM.h:90                                         */
  double PB_omega;          /* LBM.h:91                                         */
  double V, VC, VCn, VT;    /* TRT magic numbers, LBM.h:115-118                 */
} ekpnp_params;

typedef struct ekpnp_ctx ekpnp_ctx;

/* Fill *p with the reference's defaults (LBM.h:29-118) for an nx*ny*nz lattice
 * with dx=dy=dz=1e-8, Lx=nx*dx, Ly=ny*dy, Lz=(nz-1)*dz. */
int ekpnp_default_params(ekpnp_params* p, int nx, int ny, int nz);

/* Replaces main.cu:58-152 (device selection, the 13+11+3 cudaMallocs, the cuFFT
 * plan and the wavenumber tables) for a whole lattice on the current device.
 * Memory: a lattice whose populations fill less than ~45 % of the device's free memory is timed on up to
 * EKPNP_PLACEMENT_TRIES (default 5, 1 = off) placements and the fastest is kept - creation then transiently
 * holds TWO population arenas (more while they fit into 85 % of the free memory).  Processes that share a
 * device should set EKPNP_PLACEMENT_TRIES=1. */
int ekpnp_create(const ekpnp_params* p, ekpnp_ctx** out);

/* z-slab variant (no reference counterpart): rank `rank` of `nranks` owns planes
 * [rank*nz/nranks, (rank+1)*nz/nranks) (integer division: the slabs differ by at most one plane, so
 * the reference's usual NZ = 2^k + 1 decomposes over any number of GPUs); every slab needs >= 4
 * planes, at most 16 slabs.  Halo transport: the library's own (ekpnp_slab_attach_comm,
 * ekpnp_group_* below) or the caller's, through ekpnp_halo_*.  nranks == 1 is allowed (the ring
 * closes on the same rank): the whole multi-rank call sequence on one GPU. */
int ekpnp_create_slab(const ekpnp_params* p, int rank, int nranks, ekpnp_ctx** out);

/* Replaces main.cu:264-290 (cudaFree / cufftDestroy / cudaDeviceReset). */
int ekpnp_destroy(ekpnp_ctx* ctx);

/* Message of the last failing call on this context (or of the last failing
 * ekpnp_create* when ctx == NULL). Never NULL. */
const char* ekpnp_last_error(const ekpnp_ctx* ctx);

/* Use the caller's HIP stream (a hipStream_t passed as void*) for all device
 * work of this context instead of the context's own stream. */
int ekpnp_set_stream(ekpnp_ctx* ctx, void* hip_stream);
/* Wait for everything enqueued on the context's stream; also brings Ex, Ey, Ez and phi's plates up to date
 * if a lazy solve left them behind (see ekpnp_fast_poisson). */
int ekpnp_synchronize(ekpnp_ctx* ctx);

/* Back macroscopic field `field_id` by caller-owned device memory (nz_local*ny*nx
 * doubles, reference layout) instead of the context's own allocation: this is how
 * main.cu's rho_gpu ... T_gpu (main.cu:96-106) stay the arrays it copies out. */
int ekpnp_bind_field(ekpnp_ctx* ctx, int field_id, double* device_ptr);
/* The device address of a field array.  Asking for phi, Ex, Ey or Ez (or binding one) makes the context EAGER
 * from then on: whoever holds the pointer may read or write the array between any two calls, so every solve
 * writes all four and the collide reads the arrays, as the reference does (see ekpnp_fast_poisson). */
int ekpnp_field_device_ptr(ekpnp_ctx* ctx, int field_id, double** device_ptr);

/* Host <-> device transfer of one macroscopic field of this context's planes
 * (nz_local*ny*nx doubles), replacing the cudaMemcpy calls of main.cu:211-213 and
 * LBM.cu:2511-2521 / 2645-2657. */
int ekpnp_set_field(ekpnp_ctx* ctx, int field_id, const double* host);
int ekpnp_get_field(ekpnp_ctx* ctx, int field_id, double* host);

/* void initialization(r,c,cn,fi,u,v,w,ex,ey,ez,temp)  — LBM.h:159, LBM.cu:68-109:
 * uniform fields + pb_iterations Poisson-Boltzmann sweeps, all on the device. */
int ekpnp_initialization(ekpnp_ctx* ctx);

/* initialization() with a convergence test instead of the fixed sweep count (SURVEY.md §8(f)
 * row 4; LBM.cu:89-106): stops when max|phi_solved - phi_old| <= rel_tol * max(|voltage|,
 * |voltage2|) or after max_sweeps; damping omega = min(PB_omega, 1.6/(1 + (Lz/(pi lambda_D))^2)),
 * which equals the reference's PB_omega on its own grid and does not diverge on tall channels
 * (the reference's 0.05 does beyond NZ ~ 180).  rel_tol = 0 and max_sweeps = pb_iterations
 * reproduce ekpnp_initialization bit for bit wherever the damping is not reduced. */
int ekpnp_initialization_converged(ekpnp_ctx* ctx, double rel_tol, int max_sweeps, int* sweeps_done, double* residual);

/* void init_equilibrium(f0,f1,h0,h1,hn0,hn1,temp0,temp1,r,c,cn,u,v,w,ex,ey,ez,temp)
 * — LBM.h:162-163, LBM.cu:150-160: populations <- equilibrium of the fields. */
int ekpnp_init_equilibrium(ekpnp_ctx* ctx);

/* void stream_collide_save(f0,f1,f2,h0,...,Temp,double t,double* f0bc)
 * — LBM.h:165-166, LBM.cu:465-481: one collide + wall + stream + ion/thermal wall
 * sweep; writes the pre-collision moments rho,u,c,cn,T like LBM.cu:807-813. */
int ekpnp_stream_collide_save(ekpnp_ctx* ctx, double t);

/* void fast_Poisson(charge, chargen, kx, ky, kz, plan) — LBM.h:176, poisson.cu:75-103,
 * including its efield() tail (poisson.cu:28-69): phi, Ex, Ey, Ez from c, cn.
 * Lazy E (round 4): while phi, Ex, Ey, Ez are all the library's own arrays and no device pointer to them has
 * been handed out, the solve leaves phi's interior planes in the phi array and does NOT write Ex, Ey, Ez or
 * re-pin phi's plates; the next ekpnp_stream_collide_save forms E = 0.5*(phi(-1) - phi(+1))/d itself - the
 * expression of gpu_efield / gpu_bc (poisson.cu:40-69), hence the same bits - and the arrays are written the
 * moment somebody looks (ekpnp_get_field, ekpnp_synchronize, writers, diagnostics, ekpnp_init_equilibrium,
 * checkpoints).  ekpnp_set_field of any of the four is honoured as in the reference (the collide reads E,
 * whatever wrote it, LBM.cu:632-637).  EKPNP_LAZY_E=0 or ekpnp_tune(ctx, "lazy_efield", 0): every solve
 * writes the arrays (rounds 1-3); results are bit-identical either way. */
int ekpnp_fast_poisson(ekpnp_ctx* ctx);
/* The collide writes the Poisson right-hand side -F(c - cn)/eps from its registers, and
 * ekpnp_fast_poisson uses it instead of re-reading c, cn when nothing has changed them since:
 * ekpnp_set_field / ekpnp_read_* / ekpnp_bind_field invalidate it, and caller-bound c or cn
 * arrays are ALWAYS re-read (like the reference's fast_Poisson reads charge_gpu at call time,
 * poisson.cu:83).  A host that changes the library's OWN c / cn arrays on the device through
 * ekpnp_field_device_ptr between the two calls says so with this call.  Either way the result is
 * the same bits (one shared expression, poisson.cu:121-135). */
int ekpnp_invalidate_rhs(ekpnp_ctx* ctx);

/* The time loop body of main.cu:189-200, n times:
 * stream_collide_save; fast_Poisson; t += dt. */
int ekpnp_step(ekpnp_ctx* ctx, int nsteps);
int ekpnp_get_time(ekpnp_ctx* ctx, double* t);
int ekpnp_set_time(ekpnp_ctx* ctx, double t);

/* Lattice geometry of this context: global nz, first owned plane, owned planes. */
int ekpnp_local_extent(ekpnp_ctx* ctx, int* z0, int* nz_local);

/* ---- diagnostics and the main.cu IO surface ------- */
/* double current(double* c, double* cn, double* ez) — LBM.h:179, LBM.cu:2674-2710, called at
 * main.cu:211-216: I = K dz^2 sum_{x,y} (c - cn) Ez on the upper plate after the linear wall
 * extrapolation of c, cn.  Reduced on the device (wavefront shuffles); a slab that does not
 * hold the upper plate returns 0, so the ranks' values add up. */
int ekpnp_current(ekpnp_ctx* ctx, double* I);
/* the number record_umax prints (LBM.h:180, LBM.cu:2712-2753): max(0, max uz) over this
 * context's planes; ekpnp_record_umax appends the reference's "%10.6f %10.6f\n" line. */
int ekpnp_umax(ekpnp_ctx* ctx, double* umax);
int ekpnp_record_umax(ekpnp_ctx* ctx, const char* path, int append, double time);
/* void save_data_tecplot(FILE*, double time, r, c, cn, fi, u, v, w, ex, ey, ez, temp, int first)
 * — LBM.h:169, LBM.cu:2492-2565: one Tecplot POINT zone (header when first != 0), byte-compatible. */
int ekpnp_save_data_tecplot(ekpnp_ctx* ctx, const char* path, int append, double time, int first);
/* void save_data_end(FILE*, double time, ...) — LBM.h:170, LBM.cu:2567-2630 (12 columns %10.6f). */
int ekpnp_save_data_end(ekpnp_ctx* ctx, const char* path, int append, double time);
/* void read_data(double* time, r, c, cn, fi, u, v, w, ex, ey, ez, temp) — LBM.h:160,
 * LBM.cu:2632-2671: fills the 11 fields from a save_data_end file (main.cu:161-164). */
int ekpnp_read_data(ekpnp_ctx* ctx, const char* path, double* time);
/* Lossless variant of the save_data_end / read_data pair (the reference's
 * restart file keeps 6 decimals, LBM.cu:2619-2622, and overwrites the wall values of rho, c, cn, u
 * by their extrapolation, LBM.cu:2596-2611): the 11 macroscopic fields of the owned planes as
 * raw little-endian FP64 behind a 40-byte header {"EKPNPST1", nx, ny, nz, z0, nz_local, 11,
 * time}.  Restart is the reference's: ekpnp_read_state, then ekpnp_init_equilibrium
 * (main.cu:161-175).  "Lossless" is about the FIELDS: like the reference's restart, this one
 * re-equilibrates the populations from them, i.e. the non-equilibrium part of all four lattices
 * is dropped and a restarted run is NOT the bitwise continuation of the interrupted one (it
 * rejoins it as the relaxation forgets the kick: tests/test_io_gpu.py documents the size).
 * Slab contexts write / read their own planes (one file per rank). */
int ekpnp_save_state(ekpnp_ctx* ctx, const char* path, double time);
int ekpnp_read_state(ekpnp_ctx* ctx, const char* path, double* time);

/* Full checkpoint (no reference counterpart; the reference can only restart from fields): the 11
 * fields AND the post-collision populations of every lattice, raw FP64 behind a 64-byte header
 * "EKPNPCK2" (round 5: the directions inside a population tile are ordered by (c_z, c_y, c_x), slot = 9 (c_z + 1) + 3 (c_y + 1)
 * + (c_x + 1); files of the earlier layout, "EKPNPCK1", are refused).  Loading it continues the interrupted run bit for bit, in a two-buffer or an in-place
 * context alike (512^3 x 4 lattices: 128 GB).  A single context writes / reads a whole-lattice file,
 * interchangeable with ekpnp_group_save_checkpoint / _load_checkpoint; a slab context on its own
 * writes / reads a per-rank file that includes its two ghost planes. */
int ekpnp_save_checkpoint(ekpnp_ctx* ctx, const char* path);
int ekpnp_load_checkpoint(ekpnp_ctx* ctx, const char* path, double* time);

/* void compute_parameters(double* T, double* M, double* C, double* Fe, double* Pr) — LBM.h:171,
 * LBM.cu:2419-2446: the dimensionless groups main.cu:38 computes for its banner.  Pure host
 * arithmetic on the parameter struct (needs no context and no device). */
int ekpnp_compute_parameters(const ekpnp_params* p, double* T, double* M, double* C, double* Fe, double* Pr);
/* void save_scalar(const char* name, double* scalar_gpu, double* scalar_host, unsigned n) —
 * LBM.h:168, LBM.cu:2454-2490: raw FP64 dump of one field into "<name><n zero-padded>.bin"; the
 * pad width is floor(log10(nsteps) + 1) like the reference derives from its NSTEPS. */
int ekpnp_save_scalar(ekpnp_ctx* ctx, const char* name, int field_id, unsigned n, unsigned nsteps);

/* ---- measurement hooks (bench.py; no reference counterpart) ------------------ */
/* When enabled, every launch of the bulk collide/stream kernel is bracketed by
 * HIP events on the context's stream; the sum is returned by ..._get. */
int ekpnp_kernel_timing_enable(ekpnp_ctx* ctx, int enable);
int ekpnp_kernel_timing_get(ekpnp_ctx* ctx, int* n_launches, double* total_ms,
                            int64_t* nodes_per_launch);
/* While timing is enabled every Poisson solve (all its stages, on slabs from stage 1 to stage 3
 * including the exchanges in between) is bracketed the same way; this returns and resets the sum. */
int ekpnp_phase_timing_get(ekpnp_ctx* ctx, int* n_solves, double* poisson_ms);
/* Slab contexts: where the time of those solves went, from four more HIP events inside each (stage_ms[5], summed over the
 * n_solves bracketed since timing was enabled or ekpnp_phase_timing_get was last called; call this one FIRST, it does not
 * reset): [0] stage 1 (right-hand side, forward transform, edge values), [1] from the end of stage 1 to the start of stage 2
 * = the EDGE all-gather as the compute stream saw it, [2] stage 2 (interface system, z solve, inverse transform, phi pack;
 * with "edge_chunks" > 1 the waits for the later mode blocks fall in here), [3] the PHI exchange, [4] stage 3.  The sum is
 * ekpnp_phase_timing_get's poisson_ms.  n_solves = 0 on a single context.  (poisson.cu:75-103) */
int ekpnp_poisson_stage_timing_get(ekpnp_ctx* ctx, int* n_solves, double* stage_ms);
size_t ekpnp_device_bytes(const ekpnp_ctx* ctx);
/* Placement search (no reference counterpart).  On lattices that fill only part of the device the speed of the sweep
 * depends on where the population arena lies in HBM (up to 13 % between placements of the same context); when there is
 * room, ekpnp_create / ekpnp_create_slab time the real sweep on up to EKPNP_PLACEMENT_TRIES (environment, default 5,
 * 1 = off) arenas and keep the fastest.  This reports how many were tried, which one was kept and their sweep times
 * in ms (n_tried == 0: no search - the arena is most of the device, or the lattice is launch-bound). */
int ekpnp_placement_report(ekpnp_ctx* ctx, int* n_tried, int* chosen, double* sweep_ms, int capacity);
/* Streaming-copy rate of this device in GB/s (read + write bytes / time) of a plain contiguous
 * copy of `bytes` bytes on the context's stream: the secondary denominator SURVEY.md 8(d) asks
 * for next to the 8 TB/s spec figure.  Allocates and frees 2 x `bytes` of scratch. */
int ekpnp_copy_bandwidth(ekpnp_ctx* ctx, size_t bytes, double* gb_per_s);
/* Launch-shape knobs of a live context, for tuning sweeps (tools/sweep_zchunk.py).  "ab_zchunk":
 * planes per launch of the two-buffer collide sweep (0 = the whole sweep in one launch).
 * "merged_walls": 1 (default) = lattices of up to 4 M nodes collide plates and bulk in ONE launch,
 * 0 = always separate launches (what large lattices, in-place contexts and slabs do anyway); same
 * results bit for bit.  "tri_partition": the z solve of a single context - 0 = the serial Thomas sweeps
 * everywhere, 1 (default) = the partition solve (spectrum read once) on large lattices of 67 to 514 planes,
 * 2 = wherever it applies; the two solve the same system in a different elimination order (equal to rounding).
 * "tri_wide": 1 = 16 modes (wavefronts) per workgroup on columns of more than 256 rows (256-byte pieces of every row; 3 %
 * faster in isolation, default 0), same bits.
 * "bulk_yband" (EKPNP_BULK_YBAND): the interior sweep takes bands of that many rows of EVERY plane, band after band, instead of
 * plane after plane, so that the phi rows the collide reads three times (as z+1, z, z-1: E is formed from phi) are still in
 * the 256 MiB Infinity Cache when they come back.  -1 (default) = bands of 128 rows where the sweep of one plane moves more
 * than 192 MiB (cfg3: bulk kernel 38.69 -> 38.45 ms) and plane order elsewhere, 0 = plane order, n = bands of n rows (a
 * multiple of 64 that divides NY; anything else is ignored).  Another order of the workgroups: same bits.
 * "poisson_zchunk" (EKPNP_POISSON_ZCHUNK, default 0 = off): rows + columns of runs of that many planes back to back - the
 * measured alternative to "poisson_blocks" (gains less); same bits.
 * "poisson_blocks" (EKPNP_POISSON_BLOCKS; single contexts on 512- / 1024-wide planes whose z solve is the partition solve):
 * the solve's three middle passes - y forward, z solve, y inverse - taken kx block by kx block, the three passes of one block
 * back to back, so that part of a block is still in the 256 MiB Infinity Cache when the next pass wants it.  0 (default) = the
 * library decides from the half spectrum's size (three blocks from 768 MiB on - cfg3: 2.11 -> 1.97 ms per solve -, one below),
 * 1 = one block (the A/B partner), n = n blocks.  Same kernels, every mode solved by itself: same bits for every count.
 * "lazy_efield": see ekpnp_fast_poisson; same bits.
 * "batch_moments" (EKPNP_BATCH_MOMENTS, default 0 = the reference's behaviour, LBM.cu:807-813: every step stores rho, u, c,
 * cn, T): 1 = inside ONE ekpnp_step(ctx, n) / ekpnp_group_step(g, n) call only the LAST step's sweep stores the seven moment
 * arrays of the interior planes - nothing can look at the steps before it, and every array, diagnostic and file the caller
 * can see after the call holds the same bits.  56 of the sweep's 1 808 B/node (3 %) for a host that steps in batches between
 * its outputs, as main.cu's loop does between its NSAVE / printCurrent marks.  Off for a context whose moment arrays are
 * caller-bound or exposed (ekpnp_bind_field, ekpnp_field_device_ptr) and while kernel timing is enabled; the split calls
 * (ekpnp_stream_collide_save ...) always store.  bench.py's headline never uses it; the N=1 line reports it beside the
 * headline as config.batch_moments_ab.
 * Slab contexts (same bits either way): "lead_planes" (EKPNP_SLAB_LEAD_PLANES, default 2): planes of the short launch
 * in front of the interior sweep that lets the exchange kernel in (0: none); "merged_faces" (EKPNP_MERGED_FACES, default 1):
 * both faces of a slab in one launch.
 * Slab contexts whose exchanges the library moves (ekpnp_slab_attach_comm; for groups: ekpnp_group_tune) - every rank makes
 * the same call, each returns with the streams drained:
 *   "inline_exchanges" (EKPNP_INLINE_EXCHANGES, default 1): the EDGE and PHI exchanges of an RCCL transport are issued on
 *     the compute stream itself; 0: on the comm stream, with an event hand-over each way.
 *   "comm_cus" (EKPNP_COMM_CUS, default 0): that many compute units (a multiple of 8: one per XCD) are kept free of the
 *     slab's own kernels, for the exchange kernel; the compute stream is re-made.
 *   "edge_chunks" (EKPNP_EDGE_CHUNKS, default 1, at most 16): the slab solve cuts the half spectrum into that many blocks
 *     of kx columns and pipelines them: column pass and edge values of block k on the compute stream beside the all-gather
 *     of block k-1 on the comm stream, stage 2 of block k as soon as its edge values are there.  The edge buffers are then
 *     laid out block by block, which only the library's own transport exchanges: ekpnp_poisson_stage1/2 refuse such a
 *     context.  phi is bit-identical for every value.
 *   "edge_p2p" (EKPNP_EDGE_P2P, default 0): the EDGE all-gather of an RCCL transport as one direct ncclSend / ncclRecv pair
 *     with every peer (plus a device copy of the rank's own piece) instead of ncclAllGather: on xGMI every peer is one hop
 *     away on a link of its own, so a rank's piece leaves on all links at once.  The pieces land where the all-gather puts
 *     them: same bits.  Which of the two is faster between devices only a multi-GPU run can say (a `comm_ab` leg).
 * bench.py runs a few steps under each of these after its timed region on N > 1 GPUs (`comm_ab`). */
int ekpnp_tune(ekpnp_ctx* ctx, const char* knob, int value);
/* Every kernel launch of the library is checked: a rejected launch makes the entry point return
 * EKPNP_ERR_HIP with the KERNEL's name in ekpnp_last_error.  With EKPNP_DEBUG_SYNC set in the
 * environment the stream is additionally synchronised after every launch, so a fault inside a
 * kernel is reported against that kernel too (debug runs; this returns 1 then). */
int ekpnp_debug_sync_enabled(void);
/* 1: ekpnp_step holds an instantiated 2-step hipGraph, 0: none yet, -1: capture failed (eager). */
int ekpnp_graph_state(const ekpnp_ctx* ctx);

/* ---- z-slab halo interface (no reference counterpart) ------- */
/* The populations a neighbour needs after a collide: the 9 c_z=+1 directions of
 * the top owned plane go up, the 9 c_z=-1 directions of the bottom plane go down,
 * for each active lattice; gpu_stream's z wrap (LBM.cu:1972,1975) closes the ring.
 * which: 0 = send-down, 1 = send-up, 2 = recv-from-below, 3 = recv-from-above.   */
int ekpnp_halo_buffer(ekpnp_ctx* ctx, int which, double** device_ptr, size_t* n_doubles);
/* Since round 4 neither call copies anything by default: ekpnp_collide_boundary_planes stores the outgoing directions
 * straight into the send buffers (ekpnp_halo_pack then has nothing left to do), and after ekpnp_halo_unpack - which now
 * only says "the receive buffers hold the current halos" - the NEXT ekpnp_collide_boundary_planes pulls straight out of
 * them.  The call sequence is unchanged, but two ordering rules come with it for a host that moves the buffers itself:
 *   receive side: the receive buffers must stay untouched between ekpnp_halo_unpack and the next
 *     ekpnp_collide_boundary_planes (that launch reads them);
 *   send side: the send buffers are written by the launch of ekpnp_collide_boundary_planes itself, so a transfer out of them
 *     must be ordered after THAT launch on the context's stream (an event recorded after ekpnp_halo_pack still is: the
 *     call sequence keeps it behind the boundary launch), and they must not be read before it.
 * ekpnp_halo_pack keeps its place in the sequence and its checks (in-place slabs: between the boundary and the interior
 * call) and launches nothing when the boundary launch has filled the buffers already.  EKPNP_HALO_DIRECT=0 at creation
 * restores the copies through the ghost planes. */
int ekpnp_halo_pack(ekpnp_ctx* ctx);    /* post-collision boundary planes -> send buffers */
int ekpnp_halo_unpack(ekpnp_ctx* ctx);  /* recv buffers -> (what the next pull of the edge planes reads) */
/* One phi plane each way for Ez (poisson.cu:50-55); same `which` numbering. */
int ekpnp_phi_halo_buffer(ekpnp_ctx* ctx, int which, double** device_ptr, size_t* n_doubles);
/* Distributed z-tridiagonal (replaces the z part of the 3-D cuFFT, poisson.cu:86-92):
 * stage 1 = rhs + 2-D FFT + local solve, leaves 2 interface coefficients per
 * (kx,ky) mode in the edge buffer; the caller all-gathers the edge buffers of all
 * ranks (rank-major) into the gathered buffer; stage 2 = reduced solve + correction
 * + inverse FFT + phi; stage 3 (after the phi halo exchange) = E field. */
int ekpnp_poisson_stage1(ekpnp_ctx* ctx);
int ekpnp_poisson_edge_buffer(ekpnp_ctx* ctx, int gathered, double** device_ptr, size_t* n_doubles);
int ekpnp_poisson_stage2(ekpnp_ctx* ctx);
int ekpnp_phi_halo_pack(ekpnp_ctx* ctx);
int ekpnp_poisson_stage3(ekpnp_ctx* ctx);
/* Split form of ekpnp_stream_collide_save for halo/compute overlap: the slab's first and last
 * plane first (then ekpnp_halo_pack + start the exchange), the planes in between afterwards
 * (overlapping the exchange), then ekpnp_halo_unpack once the exchange has landed. */
int ekpnp_collide_boundary_planes(ekpnp_ctx* ctx);
int ekpnp_collide_interior_planes(ekpnp_ctx* ctx);
/* The pieces of initialization() (LBM.cu:68-109) for a slab host, whose Poisson solve needs the
 * transport between the stages: gpu_initialization (LBM.cu:111-128); the phi_old copy
 * (LBM.cu:79-86); gpu_PBE (LBM.cu:139-146); gpu_PBE_phi + phi_old update (LBM.cu:98-104). */
int ekpnp_init_fields(ekpnp_ctx* ctx);
int ekpnp_pbe_begin(ekpnp_ctx* ctx);
int ekpnp_pbe_concentrations(ekpnp_ctx* ctx);
int ekpnp_pbe_relax(ekpnp_ctx* ctx);
int ekpnp_pbe_end(ekpnp_ctx* ctx);
/* t += dt (main.cu:200) for hosts that drive the split calls themselves. */
int ekpnp_advance_time(ekpnp_ctx* ctx);

/* ---- the z-slab path's own transport (no reference counterpart) ------------
 * The split entry points above leave the halo transport to the caller.  The library can also do
 * it itself, over RCCL (ncclSend/ncclRecv ring + ncclAllGather over xGMI, librccl.so.1 bound on
 * first use) or, between slabs of one process, by hipMemcpyPeerAsync.  Transfers run on a
 * per-slab comm stream of the highest priority the device offers and are tied to the compute
 * stream by events only, so the population halo exchange overlaps the collision of the slab's
 * interior planes.  With a transport the reference's verbs work on slabs like on a whole lattice. */
#define EKPNP_TRANSPORT_AUTO 0 /* RCCL when every slab has a device of its own, else COPY          */
#define EKPNP_TRANSPORT_RCCL 1
#define EKPNP_TRANSPORT_COPY 2 /* in-process groups only                                           */
#define EKPNP_UNIQUE_ID_BYTES 128

/* One process per GPU (bench.py under torch.distributed.run, an MPI host, ...): ONE rank makes an
 * id (ncclGetUniqueId), the host hands it to every rank by its own means, and every rank attaches
 * its slab context - collectively: this is ncclCommInitRank with the context's rank / nranks.
 * From then on ekpnp_initialization, ekpnp_initialization_converged, ekpnp_stream_collide_save,
 * ekpnp_fast_poisson, ekpnp_step, ekpnp_current / ekpnp_umax (combined over the ranks),
 * ekpnp_record_umax, ekpnp_save_data_tecplot, ekpnp_save_data_end and ekpnp_read_data (ONE
 * whole-lattice file, the ranks take turns in z order) work on the slab context; every rank must
 * make the same calls in the same order.  ekpnp_destroy releases the communicator. */
int ekpnp_comm_unique_id(void* id128);  /* on failure the message is in ekpnp_last_error(NULL) */
int ekpnp_slab_attach_comm(ekpnp_ctx* ctx, const void* id128);
/* Can this process bind the RCCL library (dlopen + every symbol the transport uses)?  EKPNP_OK, or EKPNP_ERR_HIP with
 * the reason in ekpnp_last_error(NULL).  Needs no device and makes no communicator.  A host calls it on EVERY rank and
 * agrees on the result over its control plane BEFORE anybody attaches: a rank that cannot bind RCCL returns from
 * ekpnp_slab_attach_comm before the collective, and its peers would wait in theirs (bench.py does exactly this). */
int ekpnp_rccl_available(void);
/* Which plane transforms does this context run (1: the library's own row / column passes, 0: rocFFT plans), and how many
 * ranks of the lattice share its device (itself included; known once ekpnp_slab_attach_comm has made the communicator - an
 * all-gather of boot id + host name + PCI bus id -, 1 before).
 * PROCESSES THAT SHARE A DEVICE (rehearsals and tests on a one-GPU box; never the production layout): set
 * GPU_MAX_HW_QUEUES=1 in their environment (the HIP runtime reads it when it starts).  Each HIP process opens up to 4
 * hardware queues for its streams, more for its priority streams and RCCL's; four such processes oversubscribe the device's
 * hardware queue slots, the scheduler time-slices the QUEUES, and every small kernel of a step waits for its queue's turn:
 * 4 ranks on one MI355X ran 0.75 - 1.3 s per step against 43 - 50 ms, with stage 1 of the slab solve (transforms and edge
 * values, no exchange inside) at 130 - 356 ms instead of 0.27 ms (profiles/r05_shared_device_experiments.log).  Round 4
 * had blamed the library's own plane transforms (EKPNP_OWN_FFT=0 also cured it); round 5's experiments show that merely
 * CREATING the rocFFT plans beside the own passes cures it too - plan creation shifts the runtime's stream -> queue mapping -
 * and that with GPU_MAX_HW_QUEUES=1 both transforms run at full speed (42.8 / 44.0 ms per step).  The library prints this
 * advice once when ranks_on_device > 1 and the variable is unset; bench.py --single-device and the test workers set it. */
int ekpnp_plane_transforms(const ekpnp_ctx* ctx, int* own_passes, int* ranks_on_device);
/* The orders in effect that keep re-used rows in the Infinity Cache (ekpnp_tune "bulk_yband", "poisson_blocks", "poisson_zchunk";
 * no reference counterpart: one thread per node in plane order, LBM.cu:474, and one 3-D transform, poisson.cu:86): *band_rows =
 * rows per band of the interior sweep (0: plane after plane), *poisson_blocks = kx column blocks of a single context's solve
 * (1: whole passes; always 1 on a slab), *poisson_zchunk = planes per chunk of its row + column passes (0: whole passes).
 * Any pointer may be NULL. */
int ekpnp_pass_order(const ekpnp_ctx* ctx, int* band_rows, int* poisson_blocks, int* poisson_zchunk);
/* Failure semantics of the collective calls (no reference counterpart: the reference exit()s on any error,
 * LBM.cu:35-53).  Once RCCL is bound and the (small, host-side) team object exists, ekpnp_slab_attach_comm always
 * enters ncclCommInitRank, also on a rank whose stream / event set-up failed, so its peers return; the returns BEFORE
 * that point - a NULL or non-slab context, a context that already has a transport, RCCL not bindable on this rank
 * (ekpnp_rccl_available), host allocation failure - leave the peers waiting, and the control plane must end all
 * ranks as below.  Between the turns of the whole-lattice file IO the ranks agree on a common
 * status (a file one rank cannot open fails the call on every rank), and the residual / NaN test of
 * ekpnp_initialization_converged is a maximum over the ranks (all stop in the same sweep).  Everywhere else a rank that returns
 * a non-OK status from a verb of an attached slab has NOT taken part in that verb's exchanges: its peers
 * are then waiting inside RCCL, and the host's control plane must end all ranks (as torch.distributed.run
 * and mpirun do when one rank exits non-zero) - the library cannot recall a collective its peers are in.
 * EKPNP_RCCL_LIBRARY in the environment names the RCCL library to bind instead of librccl.so.1. */

/* Measurement hook (bench.py's `comm` block; no reference counterpart): while ekpnp_kernel_timing_enable
 * is on, every exchange of a slab with a transport (attached, or a member of a group: pass the slab's own
 * context, ekpnp_group_context) is bracketed by HIP events.  kind: 0 the population halo ring, 1 the phi
 * planes, 2 the all-gather of the Poisson interface coefficients.  Returned and reset: how many exchanges,
 * the summed time the COMPUTE stream had to wait for them (0 when hidden behind the interior sweep), the
 * summed time from "buffers ready" to "data landed" on the comm stream, and the bytes this rank sends per
 * exchange. */
int ekpnp_comm_timing_get(ekpnp_ctx* ctx, int kind, int* n_exchanges, double* wait_ms,
                          double* transfer_ms, size_t* bytes_sent);

/* One process, several GPUs (ekpnp_main --gpus N): a group is nslabs slab contexts, slab i on HIP
 * device devices[i] (devices == NULL: i modulo the device count), created, stepped and destroyed
 * together by one host thread.  Devices may repeat (then the transport is COPY: RCCL refuses two
 * ranks on one device) - that is how the whole multi-slab path is tested on a one-GPU box.  The
 * calls mirror the single-context ones (same reference citations); fields cross the boundary as
 * whole-lattice host arrays [NZ][NY][NX]; files are the single-context formats byte for byte.
 * Failure of a group verb: every slab is driven by the calling host thread, so a non-OK status from one slab is local -
 * nobody waits inside a collective for another process.  The call returns only after every slab's compute and comm
 * stream has drained (no kernel of the group is running when the caller sees the error), and the group is then POISONED:
 * some slabs have taken part in the failed verb and some have not, so every further verb that computes, exchanges or
 * reads device state answers EKPNP_ERR_INVALID with the first failure in ekpnp_group_last_error; ekpnp_group_destroy,
 * _last_error, _size, _transport and _context remain.  One exception to "drained": when the failure was an RCCL call inside a
 * collective, kernels of that collective may be waiting for partners that never come; the streams are then NOT waited for,
 * the communicators are aborted (ncclCommAbort) by ekpnp_group_destroy, and if the RCCL library has no ncclCommAbort the
 * destroy leaves those streams and communicators behind (leaked, with a message on stderr) instead of blocking for ever. */
typedef struct ekpnp_group ekpnp_group;
int ekpnp_group_create(const ekpnp_params* p, int nslabs, const int* devices, int transport, ekpnp_group** out);
int ekpnp_group_destroy(ekpnp_group* g);
const char* ekpnp_group_last_error(const ekpnp_group* g); /* g == NULL: of the last failing ekpnp_group_create */
int ekpnp_group_size(const ekpnp_group* g);
int ekpnp_group_transport(const ekpnp_group* g);          /* EKPNP_TRANSPORT_RCCL or _COPY */
int ekpnp_group_context(ekpnp_group* g, int slab, ekpnp_ctx** ctx); /* borrowed: for ekpnp_local_extent, timing hooks ...; calls that
                                                                      touch the device need hipSetDevice(devices[slab]) first.
                                                                      The ekpnp_group_* calls themselves leave the caller's
                                                                      current device as they found it. */
size_t ekpnp_group_device_bytes(const ekpnp_group* g);
int ekpnp_group_synchronize(ekpnp_group* g);
int ekpnp_group_set_field(ekpnp_group* g, int field_id, const double* host);
int ekpnp_group_get_field(ekpnp_group* g, int field_id, double* host);
int ekpnp_group_initialization(ekpnp_group* g);                       /* LBM.cu:68-109  */
int ekpnp_group_initialization_converged(ekpnp_group* g, double rel_tol, int max_sweeps, int* sweeps_done, double* residual);
int ekpnp_group_init_equilibrium(ekpnp_group* g);                     /* LBM.cu:150-160 */
int ekpnp_group_stream_collide_save(ekpnp_group* g, double t);        /* LBM.cu:465-481 */
int ekpnp_group_fast_poisson(ekpnp_group* g);                         /* poisson.cu:75-103 */
int ekpnp_group_step(ekpnp_group* g, int nsteps);                     /* main.cu:189-200 */
int ekpnp_group_tune(ekpnp_group* g, const char* knob, int value);    /* ekpnp_tune's slab and transport knobs on every slab of the group */
int ekpnp_group_get_time(ekpnp_group* g, double* t);
int ekpnp_group_set_time(ekpnp_group* g, double t);
int ekpnp_group_current(ekpnp_group* g, double* I);                   /* LBM.cu:2674-2710 */
int ekpnp_group_umax(ekpnp_group* g, double* umax);                   /* LBM.cu:2712-2753 */
int ekpnp_group_record_umax(ekpnp_group* g, const char* path, int append, double time);
int ekpnp_group_save_data_tecplot(ekpnp_group* g, const char* path, int append, double time, int first); /* LBM.cu:2492-2565 */
int ekpnp_group_save_data_end(ekpnp_group* g, const char* path, int append, double time);                /* LBM.cu:2567-2630 */
int ekpnp_group_read_data(ekpnp_group* g, const char* path, double* time);                               /* LBM.cu:2632-2671 */
/* one EKPNPST1 file for the whole lattice: interchangeable with a single context's ekpnp_save_state */
int ekpnp_group_save_state(ekpnp_group* g, const char* path, double time);
int ekpnp_group_read_state(ekpnp_group* g, const char* path, double* time);
int ekpnp_group_save_checkpoint(ekpnp_group* g, const char* path);
int ekpnp_group_load_checkpoint(ekpnp_group* g, const char* path, double* time);

#ifdef __cplusplus
}
#endif
#endif /* EKPNP_H */
