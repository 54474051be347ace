/*
 * bh.h — C-ABI of the MI355X-native Barnes-Hut engine (libbh.so).
 *
 * Drop-in boundary for the headless per-step path of the reference
 * (bgcarmin/NBody-Barnes-Hut-CUDA, file `nbody_v5_bench.cu`; all "ref:" citations
 * below are lines of that file).  The reference has no FFI/plugin interface: its
 * boundary is a set of file-scope device pointers (ref:31-40), compile-time
 * constants (ref:13-18) and `void simulationStep()` (ref:255-283) driven by
 * `main()` (ref:285-390).  This header replaces that with one opaque context per
 * GPU plus one entry point per reference stage, plain pointers and sizes only.
 *
 * Conventions
 *   - every call returns BH_OK (0) or a negative bh_status; nothing throws or aborts;
 *   - host buffers are caller-owned SoA arrays of n floats (any alignment);
 *   - device buffers are owned by the context; a context is bound to one HIP device
 *     and one stream; calls on one context are not thread-safe;
 *   - stage calls and bh_step are asynchronous on the context's stream;
 *     bh_download*, bh_get_stats and bh_sync synchronise;
 *   - the library never falls back to a CPU path: with no usable GPU bh_create fails.
 */
#ifndef BH_H_
#define BH_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BH_ABI_VERSION 6 /* 2: bh_params grew key_curve (72 bytes); 3: bh_dd_phase_*, timing mode 3; 4: force_coop,
                             bh_dd_get_info, larger X1 payload (boundary proposals), bh_walk_stats fields; 5: bh_comm,
                             bh_rank_*, bh_group (the multi-GPU step behind the ABI), BH_ERR_COMM / _DOMAIN_LEFT; 6: bh_comm.all_to_all_v */

typedef struct bh_ctx bh_ctx; /* opaque; replaces the globals ref:31-40 */

typedef enum bh_status {
  BH_OK = 0,
  BH_ERR_BAD_ARG = -1,      /* null pointer, n < 1, bad parameter value            */
  BH_ERR_NO_DEVICE = -2,    /* no HIP device / device index out of range           */
  BH_ERR_HIP = -3,          /* a HIP runtime call failed (see bh_last_hip_error)   */
  BH_ERR_OOM = -4,          /* device or host allocation failed                    */
  BH_ERR_POOL_OVERFLOW = -5,/* octree record pool exhausted (cannot happen with
                               the 2n+8 pool; checked anyway, cf. ref:321 D8)     */
  BH_ERR_ORDER = -6,        /* stage called before the stage it depends on         */
  BH_ERR_SMALL_BUFFER = -7, /* caller buffer too small for a download              */
  BH_ERR_DEVICE_FLAG = -8,  /* a device-side sticky flag (BH_FLAG_*, bh_get_stats().status_flags)
                               is set: results since it was raised are invalid       */
  BH_ERR_COMM = -9,         /* a bh_comm transfer failed (RCCL error, a callback returned non-zero, hub aborted) */
  BH_ERR_DOMAIN_LEFT = -10  /* COLLECTIVE: every rank returns it from the same step of bh_rank_step — a rank
                               announced a local failure through its X4 header, or a LET outgrew let_cap (decided
                               from all-gathered counts).  The only multi-rank error a caller may answer
                               collectively (e.g. by moving every rank to another scheme); anything else is
                               rank-local                                                                     */
} bh_status;

/* Physical and structural parameters.  Defaults = the reference's #defines. */
typedef struct bh_params {
  float G;          /* G_CONST   0.5   ref:14 */
  float theta;      /* THETA     0.5   ref:15 */
  float dt;         /* DT        0.02  ref:16 */
  float eps2;       /* SOFTENING 50.0  ref:17 — added to r^2 (ref:207); must be > 0 */
  float max_speed;  /* MAX_SPEED 500.0 ref:18 */
  int32_t leaf_cap;   /* bodies per leaf before a cell is split; 1 = reference intent (ref:100-124) */
  int32_t max_depth;  /* deepest cell level; <= key_bits/3 (reference loop bound 25, ref:93) */
  int32_t key_bits;   /* 63 (21 bits/axis, default) or 30 (10 bits/axis: bit-exact
                         reference Morton code, ref:42-63)                         */
  int32_t strict_fp;  /* 0: fast force kernel (fma + v_rsq_f32);
                         1: force arithmetic exactly as the reference source text
                            in IEEE fp32 (sqrtf, '/', no contraction; ref:203-213) */
  int32_t force_variant; /* fast force kernel: 0 = hand-scheduled gfx950 walk (default), 1 = the same walk
                            as scheduled by the compiler (A/B and fallback)                */
  int32_t xcd_mode;      /* fast force kernels, block -> body-chunk placement (speed only):
                            0 = one contiguous eighth of the body order per XCD, 1 = identity,
                            2 = runs of 64 chunks per XCD dealt round-robin,
                            3 = automatic (default): 2 when the launch has more waves than the GPU holds
                                at once, else 0                                                  */
  int32_t sort_variant;  /* 0 = automatic (default): splitter sort (one partition pass + per-bucket LDS
                            sort) when the bodies are still in an earlier step's key order and fit,
                            else 2;  1 = histogram + scan + scatter kernels per radix pass (no
                            inter-workgroup hand-off at all; A/B and fallback);  2 = one kernel per
                            radix pass with decoupled look-back;  3 = splitter sort on any input (tests).
                            All are stable sorts: identical results                                 */
  int32_t literal_force; /* 1 = reproduce the reference BINARY instead of its intent: the force on
                            every body is the root monopole G*M*(COM-p)/(|COM-p|^2+eps2)^(3/2),
                            which is what computeForceKernel literally evaluates (`idx < n`
                            accepts the root, ref:198,208) — for diffing
                            trajectories against nbody_v5_bench.exe                              */
  int32_t force_block;   /* threads per workgroup of the default force kernel: 64, 128 or 256
                            (0 = library default); waves never cooperate, speed only           */
  int32_t step_graph;    /* bh_step: 0 = launch its kernels one by one (default), 1 = replay the step as a HIP
                            graph, one per ping-pong parity of the body arrays (measured slower on ROCm 7.2:
                            DESIGN.md)                                                                  */
  int32_t force_group;   /* bodies per group of the default force kernel: 64, 32 or 16 (upper lanes idle), or
                            0 = automatic: 64 (with force_coop = 1, the one-wave walk: 16 up to 20,480 bodies, 32 up to
                            57,344, else 64).  The one-wave walk's results do not depend on it; with several waves
                            per group a body's bits depend on its group's composition                       */
  int32_t key_curve;     /* numbering of the 2^21-per-axis cell grid behind the 63-bit keys: 0 = Morton / Z order
                            (the reference's, ref:42-63), 1 = Hilbert order (default).  Same cells, same tree up
                            to the order of the children inside a block and of the bodies in memory; the
                            64-body groups of the force walk are more compact (-5 % force time).  30-bit keys
                            are always Morton (reference-literal code)                                */
  int32_t force_coop;    /* waves that share the walk of one group of bodies (round 4): 0 = by context size — 8 while
                            eight per group fit the GPU at once (65,536 bodies: 8 waves per SIMD), 4 up to ~305,000 bodies, above that
                            one wave per group except for the last ~2,400 groups of the launch, which get four (the
                            short jobs fill the wave slots the long ones leave: DESIGN.md §4) —, 1 = one wave per
                            group throughout (the depth-first walk), 2..8 = that many for every group.  Results are
                            reproducible bit for bit for a given body count and value; different values differ in
                            the association of the fp32 sums only.  The passes of the domain-decomposed step
                            (bh_dd_*) follow the same rule with the rank's body count                        */
} bh_params;

/* One 32-byte octree record ("entry").  The tree is an array of entries:
 * entry 0 is the root; the children of an internal entry are the `count`
 * consecutive entries starting at `first` (ascending key digit — the octant digit
 * x<<2|y<<1|z with Morton keys, the Hilbert digit with key_curve = 1 — empty
 * octants omitted); `first` is always even (BH_KIND_PAD entries fill the gaps).
 * Replaces `struct OctreeNode` (ref:20-28, 76 B). */
#define BH_KIND_BODY 0     /* exactly one body: (x,y,z,m) is the body, s = 0, first = sorted body index */
#define BH_KIND_INTERNAL 1 /* subdivided cell: first/count = child block                               */
#define BH_KIND_MULTI 2    /* unsplit cell holding `count` > 1 bodies [first, first+count) of the
                              Morton-sorted body array (depth-capped or leaf_cap > 1)                  */
#define BH_KIND_PAD 3      /* padding entry (all fields zero): child blocks start at EVEN entries (64-byte
                              boundaries), so entry 1 and the entry after every block of an odd number of
                              children are padding; no record refers to them                           */
typedef struct bh_node {
  float x, y, z;  /* centre of mass (ref:22 comX..Z after finalizeCOM ref:175-189) */
  float m;        /* total mass (ref:21)                                           */
  float s;        /* cell edge length used by the MAC (ref:208 maxX-minX); -1 for a body, so
                     that `s/dist < theta` accepts it for every theta >= 0 (ref:208 `idx < n`) */
  int32_t first;
  int32_t count;
  int32_t kind;
} bh_node;

typedef struct bh_stats {
  int32_t n;               /* bodies                                               */
  int32_t n_internal;      /* subdivided cells ( = reference nodeCounter intent, ref:109) */
  int32_t n_entries;       /* tree entries in use (root + all children + padding)   */
  int32_t max_level;       /* deepest internal level + 1                           */
  int32_t status_flags;    /* device-side sticky error bits (BH_FLAG_*)            */
  int32_t steps;           /* bh_step calls since upload                           */
  /* per-stage device time of the most recent timed step, milliseconds (hipEvent);
     zero unless bh_set_timing(ctx,1).  Order = ref:259-282. */
  float ms_bbox, ms_morton, ms_sort, ms_build, ms_com, ms_force, ms_integrate, ms_step;
  /* interaction counters of the most recent bh_force_count call (whole system):
     V = cell MAC evaluations, O = opened cells, P = body-body interactions.
     SURVEY §8(d): B_alg = 24 V + 32 O + 16 P + 24 n bytes. */
  uint64_t count_V, count_O, count_P;
  int32_t force_redo_waves; /* fast force kernel: waves (64 bodies) since upload that redid their walk with the
                               generic loop (cross-lane stack deeper than 64 entries, or an unsplit cell of
                               more than 8 bodies); a speed matter only                                   */
  int32_t sort_slow_buckets; /* splitter sort: buckets since upload that did not fit LDS and were sorted by one
                                workgroup through global memory (many equal keys, or an order that drifted far);
                                when bh_get_stats finds that more than every fourth sort since its last call had
                                one, the context sorts with the radix passes until the next upload          */
  int32_t reserved[6];
} bh_stats;

#define BH_FLAG_POOL_OVERFLOW 1
#define BH_FLAG_STACK_OVERFLOW 2
#define BH_FLAG_SORT_TIMEOUT 4 /* a look-back spin of the radix sort hit its bound: order invalid */
#define BH_FLAG_TRAVERSAL_LIMIT 128 /* a wave popped > 2^22 child blocks: malformed record pool, forces invalid */
#define BH_FLAG_DD_LET_INVALID 64   /* a gathered LET record pointed outside its segment or had > 8 children: the
                                       record was closed (never opened), forces of this step are invalid      */

/* ---- lifecycle ( <-> cudaMalloc block ref:311-326, cudaFree ref:372-387 ) ---- */
int bh_abi_version(void);
int bh_default_params(bh_params* p);
int bh_create(bh_ctx** out, int n, const bh_params* p, int device);
/* same, but run on a caller-owned hipStream_t (e.g. torch's current stream) */
int bh_create_on_stream(bh_ctx** out, int n, const bh_params* p, int device, void* hip_stream);
void bh_destroy(bh_ctx* c);
const char* bh_strerror(int status);
int bh_last_hip_error(const bh_ctx* c); /* raw hipError_t of the last BH_ERR_HIP */

/* ---- data in ( <-> 7x cudaMemcpy H2D ref:329-335 ) ---- */
int bh_upload(bh_ctx* c, const float* x, const float* y, const float* z,
              const float* vx, const float* vy, const float* vz, const float* m);

/* ---- the step ( <-> simulationStep ref:255-283 ) and its stages, reference order ---- */
int bh_step(bh_ctx* c);
int bh_bbox(bh_ctx* c);      /* computeBoundingBoxKernel ref:134-156 / launch ref:259          */
int bh_morton(bh_ctx* c);    /* computeMortonCodesKernel ref:51-63  / launch ref:260          */
int bh_sort(bh_ctx* c);      /* thrust::sort_by_key ref:262-264 (+ physical gather) */
int bh_build(bh_ctx* c);     /* memset+initRoot+insertParticles ref:266-275                    */
int bh_com(bh_ctx* c);       /* computeCOM+finalizeCOM ref:279-280                             */
int bh_force(bh_ctx* c);     /* computeForceKernel ref:281                                     */
int bh_integrate(bh_ctx* c); /* integrateKernel ref:282                                        */

/* force for the Morton-sorted bodies [lo,hi) only (multi-rank sharding).  Every body gets the bits of the
   full launch.  Where the context walks groups with several waves (force_coop != 1: the whole context below ~305,000
   bodies, the last ~2,400 groups above) a body's bits depend on its 64-body group, so a slab that starts inside that
   part must start on one of its group boundaries — a multiple of 64 bodies (256 is always safe: dist.slab_bounds) —
   else BH_ERR_BAD_ARG */
int bh_force_range(bh_ctx* c, int lo, int hi);
/* same traversal with per-body V/O/P counters; totals land in bh_stats */
int bh_force_count(bh_ctx* c);
/* Measurement only: one launch of the default force walk with its event counters switched on (accelerations
   are not stored).  What the walk issued, summed over all waves — the quantities its issue-rate roofline is
   priced with (bench.py) — and the shader clock the waves saw while they ran. */
typedef struct bh_walk_stats {
  uint64_t waves;        /* 64-body groups                                                            */
  uint64_t pairs;        /* record pairs evaluated (two records per packed instruction; a block of an
                            odd number of children evaluates one null record)                         */
  uint64_t blocks;       /* child blocks popped (= pushes + 1 per wave)                                */
  uint64_t masked_pairs; /* pairs with an opened record (force half runs with the take masks)          */
  double clock_ghz;      /* median over the waves of shader cycles / constant-clock time               */
  double wave_cycles_max, wave_cycles_mean; /* wave lifetime, shader cycles                            */
  uint64_t lane_spills;  /* stack entries that went through the cross-lane stack (3 v_writelane + 3 v_readlane
                            each); the others stayed in scalar registers from push to pop                 */
  uint64_t no_taker_pairs; /* of the masked pairs: those in which every active lane opens both records */
  uint64_t fetch_wait_cycles; /* small-launch instance only: shader cycles, summed over waves and blocks, between the
                                 issue of a child block's scalar loads and the arrival of its records       */
  uint64_t reserved[1];
} bh_walk_stats;
int bh_force_walk_stats(bh_ctx* c, bh_walk_stats* out);
/* Measurement only: the force launch bh_step would make for all bodies (the same kernel, grid and placement; not
   fused with the integrate step) with one row of four words per wave: start and end of its walk on the chip-wide
   100 MHz clock, HW_ID, XCC_ID — the resident waves over time, when the last wave starts, how long every SIMD idles
   before the launch ends (bench.py roofline.issue.residency, tools/force_trace.py).  rows[4 * capacity_rows];
   *n_rows = rows written, 0 when the context's force walk has no traced instance (strict_fp, literal_force,
   force_variant 1, domain-decomposed) or the capacity is too small (a launch has at most 4 rows per 64 bodies). */
int bh_force_launch_trace(bh_ctx* c, uint32_t* rows, int capacity_rows, int* n_rows);

/* ---- data out ---- */
/* caller (upload) order; any pointer may be NULL */
int bh_download(bh_ctx* c, float* x, float* y, float* z, float* vx, float* vy, float* vz);
int bh_download_acc(bh_ctx* c, float* ax, float* ay, float* az);        /* caller order      */
int bh_download_bounds(bh_ctx* c, float bounds[6]);                      /* ref:150-155 layout */
int bh_download_keys(bh_ctx* c, uint64_t* keys);   /* current (sorted after bh_sort) order      */
int bh_download_order(bh_ctx* c, int32_t* ids);    /* caller index of the body at each sorted slot */
int bh_download_sorted_bodies(bh_ctx* c, float* xyzm /* 4n floats */);
/* the canonical tree records: valid after bh_com (stage calls) or after a bh_step of a strict_fp / literal_force
   context; a bh_step of the default engine writes only the force kernel's digests and bh_build alone leaves the
   centres of mass unset — BH_ERR_ORDER in both cases (the entry count is still returned when out == NULL) */
int bh_download_tree(bh_ctx* c, bh_node* out, int capacity, int* n_entries);
int bh_download_counters(bh_ctx* c, uint32_t* V, uint32_t* O, uint32_t* P); /* caller order */
int bh_download_mass(bh_ctx* c, float* m);                              /* caller order      */
/* interleaved xyz positions and speed-mapped RGB colours, caller order, 3n floats each
   (the reference viewer's updateVisualsKernel, nbody_v5.cu:278-292: t = min(|v|/150, 1),
   rgb = (0.4+0.6t, 0.3+0.4t, 1-0.7t)); written to host buffers instead of a GL VBO */
int bh_export_visual(bh_ctx* c, float* pos_xyz, float* col_rgb);
int bh_get_stats(bh_ctx* c, bh_stats* s);
/* on = 1: an event after every stage of bh_step (bh_stats.ms_*, bh_timing_history); on = 2: only the pair around
   the force launch (ms_force; the other times read 0) — each event record costs the stream several microseconds,
   eight of them ~60 us per 1M-body step; on = 3: that pair on every 4th step only (bh_timing_history then holds one
   entry per sampled step); 0 = off.  In modes 0, 2 and 3 the force launch of bh_step also integrates the bodies
   (same arithmetic, same results; ms_force of modes 2 / 3 = force + integrate); mode 1 times separate kernels */
int bh_set_timing(bh_ctx* c, int on);
/* waits for the context's stream; BH_ERR_DEVICE_FLAG if a sticky device flag is set (the step loop's check:
   stack / pool overflow, sort time-out, traversal limit — see BH_FLAG_*) */
int bh_sync(bh_ctx* c);

/* ---- multi-rank plumbing: raw device views (valid until bh_destroy) ---- */
/* acc is float4[n] (ax,ay,az,unused) in Morton-sorted order; an all-gather of the
   per-rank [lo,hi) slabs straight into this buffer completes the exchange step. */
int bh_device_acc(bh_ctx* c, void** dptr, int64_t* bytes);
/* make the engine write/read accelerations in a caller-owned device buffer of at least
   n float4 (e.g. a torch tensor that is also the all-gather output); NULL restores the
   engine's own buffer.  The caller keeps the buffer alive while it is bound. */
int bh_bind_acc(bh_ctx* c, void* device_float4_n);
int bh_n(const bh_ctx* c);

/* ---- domain-decomposed multi-GPU stepping, engine side (the reference is single-GPU) ----
 * One context per rank, each owning the bodies of one contiguous interval of the key curve.  These entry points pack
 * and consume plain device buffers and never communicate; bh_rank_step (below) is the per-step protocol around them:
 *
 *   [X1 all-gather] bh_dd_phase_migrate   global cube (exact), splitter keys, emigrants packed
 *   [X2 all-gather] bh_dd_phase_tree      immigrants absorbed; local sort / build / COM; piece descriptors
 *   [X3 all-gather] bh_dd_phase_let       own pieces walked on a side stream (two-pass steps); LET marked + exported
 *   [X4 all-to-all] bh_dd_phase_force     validation, top tree, remote (or whole) pass — which integrates
 *                   bh_dd_phase_end       the next step's X1 payload
 *
 * The local octree of a rank is the canonical octree of the global cube restricted to its bodies;
 * every cell that does not touch either end of the rank's body range is a complete global cell.
 * The children of the end-touching ("spine") cells are the rank's PIECES; the canonical tree above
 * the pieces of all ranks (the top tree) is rebuilt identically on every rank from the gathered
 * piece descriptors, so the stitched tree is the same octree a single GPU builds and forces agree
 * with the 1-rank run up to summation order (cell sums: fp64, differently associated).
 * A LET segment holds the child blocks of every local cell that some body of another rank could
 * open (conservative box test against the remote pieces).  Requires key_bits 63, leaf_cap 1 and max_depth 21
 * (the default: unsplit cells then hold coincident bodies only and are far too small for a remote body to open;
 * bh_dd_init returns BH_ERR_BAD_ARG otherwise).  The fine-grained calls (bh_dd_cube_pack .. bh_dd_let_check) are the
 * pieces the phase calls are made of; tests and tools drive them directly. */
#define BH_DD_PIECE_CAP 512          /* pieces per rank: <= 2 spines x 21 levels x 7 = 294      */
#define BH_FLAG_DD_BODIES 16         /* local body count exceeded the context's capacity         */
#define BH_FLAG_DD_PIECES 32         /* more than BH_DD_PIECE_CAP pieces                          */

typedef struct bh_dd_sizes {
  int64_t x1_bytes;     /* per-rank payload of exchange X1                                       */
  int64_t x2_bytes;     /* per-rank payload of exchange X2 (header + mig_cap bodies of 32 B)     */
  int64_t x3_bytes;     /* per-rank payload of exchange X3 (header + BH_DD_PIECE_CAP descriptors) */
  int64_t pool_records; /* 32-byte records the caller's pool must hold                           */
  int64_t seg_base;     /* pool index of LET segment 0: X4 gathers world x stride records here   */
  int64_t let_min;      /* smallest legal stride (header + piece slots)                          */
  int64_t let_cap;      /* largest legal stride                                                  */
  int64_t top_base;     /* pool index of the top-tree root                                       */
} bh_dd_sizes;

/* sizes for a context created with capacity n_cap bodies; world <= 13 (the top tree holds 4096 pieces and a rank
   contributes at most 294, so it cannot overflow) */
int bh_dd_query(int n_cap, int world, int mig_cap, int let_cap, bh_dd_sizes* out);
/* switch a context (created with n = body capacity) to domain-decomposed stepping; `pool` is a
   caller-owned device buffer of pool_records x 32 B that becomes the context's record pool */
int bh_dd_init(bh_ctx* c, int world, int rank, int64_t n_total, int mig_cap, int let_cap,
               void* pool, int64_t pool_records);
/* the bodies this rank starts with (any subset; a Morton slab of the global order avoids a large
   first migration) and their global ids, host pointers */
int bh_dd_upload(bh_ctx* c, int n_loc, const float* x, const float* y, const float* z,
                 const float* vx, const float* vy, const float* vz, const float* m,
                 const int32_t* ids);
int bh_dd_cube_pack(bh_ctx* c, void* send_x1);
int bh_dd_cube_apply(bh_ctx* c, const void* gathered_x1);
/* limit = emigrant slots per rank in this round's X2 (header 32 B + limit x 32 B per rank,
   1 <= limit <= mig_cap); emigrants beyond it wait for another round of the same step */
int bh_dd_migrate_pack(bh_ctx* c, void* send_x2, int limit);
/* synchronises; *n_loc = bodies this rank now holds, *more = 1 when some rank still holds
   emigrants (run another pack/gather/apply round), *most = most emigrants found on any rank */
int bh_dd_migrate_apply(bh_ctx* c, const void* gathered_x2, int limit, int* n_loc, int* more,
                        int* most);
int bh_dd_tree(bh_ctx* c, void* send_x3);
/* stride = records per LET segment in this step's X4 (let_min <= stride <= let_cap) */
int bh_dd_let_pack(bh_ctx* c, const void* gathered_x3, void* send_x4, int stride);
/* X4 flavour (default 0).  0: send_x4 is ONE segment of `stride` records — the union of what any other rank may
   open — and the caller ALL-GATHERS it into the pool's segment area.  1: send_x4 is `world` segments of `stride`
   records, segment q holding only what rank q may open (own slot: a closed header), and the caller exchanges them
   with an ALL-TO-ALL (rank q receives this rank's segment q at segment index `rank` of its pool): about a third of
   the received bytes at 8 ranks.  Every segment carries its sender's needs for all receivers (records 1..3), so all
   ranks still take the same stride decision in bh_dd_let_check. */
int bh_dd_set_let_mode(bh_ctx* c, int mode);
/* on = 1: the own-pieces pass of a two-pass step runs on the context's main stream instead of a side stream — for
   ranks that SHARE one GPU (one-GPU rehearsals: bh_create_group sets it when a device is listed twice), whose side
   streams would run into each other's work and measure nothing a GPU of their own would show */
int bh_dd_set_serial(bh_ctx* c, int on);
/* two-pass steps: the walk of only the first pct per cent of the rank's bodies is split into an own-pieces pass (beside
   the LET export and X4) and a remote pass; the other bodies are walked in ONE pass after X4, at the same time as that
   remote pass (1 <= pct <= 100; 100 = every body in two passes) */
int bh_dd_set_split_percent(bh_ctx* c, int pct);
/* optional, right after the X3 all-gather: walk the rank's OWN pieces on a side stream while the LET
   marking, export and the X4 all-gather run on the main stream.  bh_dd_top / bh_dd_force then cover
   only the other ranks' pieces and bh_integrate adds the two partial accelerations.  The split is
   exact: both passes apply the same MAC to the same cells, a top cell accepted as a whole contributes
   each pass's share of its mass at the cell's centre. */
int bh_dd_force_local(bh_ctx* c, const void* gathered_x3);
int bh_dd_top(bh_ctx* c, const void* gathered_x3, int stride);
int bh_dd_force(bh_ctx* c);
/* synchronises up to the end of X4 only; counts[world] = records each rank needed.  Returns
   BH_OK, or BH_ERR_SMALL_BUFFER when some count exceeds stride (repeat let_pack..force larger) */
int bh_dd_let_check(bh_ctx* c, int stride, int32_t* counts);
/* local bodies in local Morton order: posm[4 n_loc] = x,y,z,m; velid[4 n_loc] = vx,vy,vz,bits(id) */
/* One entry point per phase group (the same sequences as the calls above; what dist.DomainStepper uses):
     bh_dd_phase_migrate = bh_dd_cube_apply + bh_dd_migrate_pack
     bh_dd_phase_tree    = bh_dd_migrate_apply, then bh_dd_tree unless *more (another migration round comes first)
     bh_dd_phase_let     = bh_dd_force_local (if own_pass) + bh_dd_let_pack
     bh_dd_phase_force   = bh_dd_top + bh_dd_let_check, then — every LET fitted — bh_dd_force, whose launch also integrates
                           the bodies when the rank holds enough of them (*fits = 0: nothing walked, repeat X4 larger)
     bh_dd_phase_end     = bh_integrate + bh_dd_cube_pack (the next step's X1 payload: this rank's min / max are
                           folded by the integrate kernel, no bounding-box launches) */
int bh_dd_phase_migrate(bh_ctx* c, const void* gathered_x1, void* send_x2, int limit);
int bh_dd_phase_tree(bh_ctx* c, const void* gathered_x2, int limit, void* send_x3, int* n_loc, int* more, int* most);
int bh_dd_phase_let(bh_ctx* c, const void* gathered_x3, void* send_x4, int stride, int own_pass);
int bh_dd_phase_force(bh_ctx* c, const void* gathered_x3, int stride, int32_t* counts, int* fits);
int bh_dd_phase_end(bh_ctx* c, void* send_x1);
int bh_dd_download(bh_ctx* c, float* posm, float* velid, float* acc);
/* Measurement only: bh_force_walk_stats for a rank of the decomposed step — the counted walk of its bodies over the
   stitched pool as the last step left it; which = 0: the tree of the last (or only) force pass, 1: the own-pieces pass
   (two-pass steps) */
int bh_dd_walk_stats(bh_ctx* c, int which, bh_walk_stats* out);
/* Measurement only (two-pass steps): the rank's force passes re-launched over the pool as the last step left it, with
   nothing else on the GPU.  Full two-pass step: ms[0] own-pieces pass alone, ms[1] remote pass alone, ms[2] both at
   once on two streams, ms[3] 0.  Partial step (split_pct < 100): ms[0] / ms[1] own / remote pass of the split part,
   ms[2] the one pass of the other bodies, ms[3] that one pass and the remote pass at once, as the step runs them */
int bh_dd_pass_times(bh_ctx* c, float ms[4]);
/* Measurement only: the needs matrix of the last X4 — out[q * world + j] = records rank q's segment for receiver j
   needed (header + pieces + exported blocks; 0 on the diagonal) — as every rank sees it in the received headers */
int bh_dd_needs_matrix(bh_ctx* c, int32_t* out);
/* what the last step's migration did, for logs and tests (synchronises): out[0] bodies this rank holds, [1] emigrants
   it found in its last classification, [2] steps since bh_dd_init in which the domain boundaries moved, [3] what the
   last step did with them: 0 kept (a rank owns a fixed interval of the curve: the splitter keys persist from step
   to step), 1 moved to the exact quantiles (a rank's body count had left n / P by more than
   1.5 %), 2 drawn from position samples (first step, or a boundary would have had to cross a whole rank) */
int bh_dd_get_info(bh_ctx* c, int32_t out[8]);

/* ---- the multi-GPU step behind the C-ABI (round 5; SURVEY §8b bh_create_group / bh_step_group) ----
 * bh_rank  = one rank of the domain-decomposed step: a context, its exchange buffers and the per-step PROTOCOL
 *            (the order X1..X4, extra migration rounds, the negotiation of the X2 / X4 sizes, the LET retry, the rule
 *            that a failed rank keeps exchanging empty payloads so that nobody is stranded in a collective).  Bytes
 *            move through a bh_comm: three transports ship with the library (RCCL, in-process device copies, caller
 *            callbacks — through which dist.py keeps torch.distributed).  One process per GPU calls bh_rank_step.
 * bh_group = the ranks of ONE process, one host thread per rank: `int devices[n]` all different -> one rank per GPU
 *            over RCCL (ncclCommInitAll); a device listed several times -> in-process copies (one-GPU rehearsal and
 *            tests; RCCL refuses two ranks on one device).  bh_step_group is simulationStep() (ref:255-283) for the
 *            whole node, bh_group_upload / bh_group_download are main()'s copies (ref:329-335, 337-343).           */
typedef struct bh_comm { /* how a rank's buffers travel.  Functions return 0 or non-zero; they enqueue on hip_stream
                            (or complete before returning); pointers are device pointers (host pointers for a scripted
                            rank) */
  int32_t world, rank;
  void* user;
  /* recv[q * bytes .. (q + 1) * bytes) <- rank q's send[0 .. bytes) */
  int (*all_gather)(void* user, void* recv, const void* send, int64_t bytes, void* hip_stream);
  /* recv chunk q <- rank q's send chunk `rank`, chunks of bytes_per_peer */
  int (*all_to_all)(void* user, void* recv, const void* send, int64_t bytes_per_peer, void* hip_stream);
  void (*release)(void* user); /* called once by bh_rank_destroy (or by the caller if no rank took the comm); may be NULL */
  /* ABI 6, may be NULL (the library then uses all_to_all): the all-to-all with a size per pair.  Send slot q and receive
     slot q lie slot_bytes apart as in all_to_all; the first recv_bytes[q] bytes of receive slot q <- the first
     send_bytes[rank] bytes of rank q's send slot `rank` (equal by construction: every rank derives all sizes from the
     same all-gathered numbers).  X4's segments are two thirds padding under one size for all pairs. */
  int (*all_to_all_v)(void* user, void* recv, const void* send, int64_t slot_bytes, const int64_t* send_bytes,
                      const int64_t* recv_bytes, void* hip_stream);
} bh_comm;

/* RCCL transport (librccl.so.1 is opened at the first call, not linked: a process that already holds RCCL — torch —
   shares its copy).  _from: a communicator the caller owns.  _unique_id + _init_rank: one process per GPU
   (ncclGetUniqueId on one rank, the 128 bytes broadcast by the launcher, ncclCommInitRank everywhere); the
   communicator is destroyed by the comm's release. */
int bh_comm_rccl_from(bh_comm* out, void* nccl_comm, int world, int rank);
int bh_comm_rccl_unique_id(void* id128);
int bh_comm_rccl_init_rank(bh_comm* out, const void* id128, int world, int rank, int device);
/* Collective self-test of any transport on the calling thread's current device: one all-gather and one all-to-all
   of known words, checked on the host.  BH_OK, or BH_ERR_COMM if a call failed or a chunk arrived in the wrong
   place.  bench.py and bh_bench run it once after building the RCCL transport. */
int bh_comm_check(const bh_comm* comm);
/* in-process transport: `world` ranks of one process (host threads), device-to-device copies ordered by events */
typedef struct bh_hub bh_hub;
int bh_hub_create(bh_hub** out, int world);
int bh_comm_hub(bh_comm* out, bh_hub* hub, int rank);
void bh_hub_abort(bh_hub* hub);   /* releases every rank waiting in an exchange with BH_ERR_COMM (a rank died) */
void bh_hub_destroy(bh_hub* hub);

typedef struct bh_rank bh_rank;
typedef struct bh_rank_opts {
  int32_t n_cap;    /* body capacity of the rank; 0 = 1.3 x ceil(n_total / world) + 4096                     */
  int32_t mig_cap;  /* emigrant slots of the X2 buffers; 0 = min(max(4096, n_cap / 2), 4 n_cap / world)        */
  int32_t let_cap;  /* records per LET segment; 0 = 516 + n_cap                                                */
  int32_t let_mode; /* X4: 1 = per-destination segments, all-to-all (default); 0 = one union segment, all-gather */
  int32_t split;    /* force passes per step.  0 (and -1, the default): ONE pass after X4.  1: two passes for the first
                       split_pct per cent of the rank's bodies — own pieces on a side stream, launched behind the LET
                       export, while X4 is in flight; remote pieces after X4 — and one pass after X4 for the rest.
                       2: adaptive — one pass while the measured X4 (events around the exchange) is short, the split
                       form once it lasts ~0.2 ms and more (where it pays: DESIGN.md §6); the rank's own decision, so
                       results then depend on timing in the last bits.  Only with more than one rank and a capacity
                       of >= 400,000 bodies (launches that fill the GPU), else one pass                            */
  int32_t log;      /* 1 = keep (emigrants, boundary action) per step for bh_rank_read_log: synchronises, tests  */
  int32_t serial;   /* 1 = bh_dd_set_serial (ranks sharing one GPU)                                               */
  int32_t split_pct; /* two-pass steps: per cent of the bodies whose walk is split (bh_dd_set_split_percent); 0 = default
                        (30), 100 = every body in two passes                                                        */
  int32_t reserved[8];
} bh_rank_opts;
typedef struct bh_rank_plan { /* bh_rank_query: the resolved capacities and the byte sizes of the eight buffers */
  int32_t n_cap, mig_cap, let_cap, stride0;
  bh_dd_sizes sz;
  int64_t bytes[8]; /* x1s, x1r, x2s, x2r, x3s, x3r, x4s (send segments), pool */
} bh_rank_plan;
typedef struct bh_rank_buffers { void *x1s, *x1r, *x2s, *x2r, *x3s, *x3r, *x4s, *pool; } bh_rank_buffers;
typedef struct bh_rank_info {
  int32_t n_loc;        /* bodies this rank holds                                                    */
  int32_t stride;       /* records per LET segment of the next X4                                    */
  int32_t mig_stride;   /* emigrant slots per rank of the next step's first X2 round                 */
  int32_t mig_last;     /* most emigrants any rank found in the last step                            */
  int32_t mig_rounds;   /* extra migration rounds since creation                                     */
  int32_t let_retries;  /* X4 repeats since creation                                                 */
  int32_t left_rank;    /* BH_ERR_DOMAIN_LEFT: the rank that left (-1: a LET outgrew let_cap)        */
  int32_t left_status;  /* this rank's own failing status when it is the one that left, else 0       */
  int64_t steps;        /* completed steps                                                           */
  int32_t let_counts[64]; /* records every rank needed in the last X4                                */
  int32_t split_now;    /* 1: the next step walks part of the bodies in two passes (split 1, or adaptive and on) */
  int32_t x4_us;        /* adaptive form: running mean of the measured X4 duration, microseconds (-1: none yet) */
  int32_t x4_recv_kb;   /* KiB this rank received in the last step's (last) X4                                  */
  int32_t reserved[5];
} bh_rank_info;

int bh_rank_default_opts(bh_rank_opts* o);
int bh_rank_query(int64_t n_total, int world, const bh_rank_opts* o, bh_rank_plan* out);
/* hip_stream NULL: the rank makes its own; bufs NULL: the rank allocates (else eight caller-owned device buffers of
   at least plan.bytes[] each, zero-filled, alive until bh_rank_destroy).  The comm is copied. */
int bh_rank_create(bh_rank** out, const bh_comm* comm, int64_t n_total, const bh_params* p, const bh_rank_opts* o,
                   int device, void* hip_stream, const bh_rank_buffers* bufs);
/* the bodies this rank starts with and their global ids (bh_dd_upload) */
int bh_rank_upload(bh_rank* r, int n_loc, const float* x, const float* y, const float* z, const float* vx,
                   const float* vy, const float* vz, const float* m, const int32_t* ids);
/* `steps` steps of the protocol (collective: every rank of the comm calls it with the same count) */
int bh_rank_step(bh_rank* r, int steps);
int bh_rank_get_info(bh_rank* r, bh_rank_info* out);
bh_ctx* bh_rank_ctx(bh_rank* r); /* the rank's context: bh_dd_download, bh_get_stats, bh_sync, bh_dd_get_info */
int bh_rank_buffers_of(bh_rank* r, bh_rank_buffers* out, bh_rank_plan* plan);
/* device time per phase, events on the rank's stream: on = 1 starts (and clears), 0 stops */
#define BH_RANK_PHASES 6 /* x1 exchange | cube, splitters, X2 migration, local tree | x3 exchange | LET export + X4 |
                            top tree + force | integrate + next X1 payload */
int bh_rank_set_profile(bh_rank* r, int on);
int bh_rank_phase_ms(bh_rank* r, double mean_ms[BH_RANK_PHASES], int* steps);
int bh_rank_read_log(bh_rank* r, int32_t* pairs, int capacity_pairs, int* n_pairs);
/* Measurement only: the force phase of the LAST COMPLETED step run again on the rank's own two streams with nothing else
   on the GPU — LET marking and export, an idle wave of x4_us microseconds on the main stream in the place of X4 (the
   imported segments of the last step are still in the pool), validation, top trees, the force pass(es) without the
   integration — `reps` times after one untimed round; *ms = mean time from the first LET kernel to the end of the last
   force launch.  split = 0: one pass; 1: the two-pass form for split_pct per cent of the bodies.  What a rank-step
   spends between X3 and its end on a GPU of its own, for every form, from a one-GPU rehearsal of P ranks (the other
   ranks must be idle: call it for one rank at a time; `bh_bench --replay`).  The rank steps on afterwards as if it had
   not been called; only the acceleration arrays hold the replay's values until the next step. */
int bh_rank_replay_force_phase(bh_rank* r, int split, int split_pct, int x4_us, int reps, float* ms);
void bh_rank_destroy(bh_rank* r);
/* Test hook: the same protocol around a SCRIPTED engine and host buffers (no GPU): tests/test_dist_cpu.py drives
   it over gloo.  The callbacks have the meaning of bh_dd_cube_pack, bh_dd_phase_migrate, bh_dd_migrate_pack,
   bh_dd_phase_tree, bh_dd_phase_let, bh_dd_phase_force, bh_dd_phase_end with `user` for the context. */
typedef struct bh_rank_script {
  void* user;
  int (*cube_pack)(void* user, void* x1s);
  int (*phase_migrate)(void* user, const void* x1r, void* x2s, int limit);
  int (*migrate_pack)(void* user, void* x2s, int limit);
  int (*phase_tree)(void* user, const void* x2r, int limit, void* x3s, int* n_loc, int* more, int* most);
  int (*phase_let)(void* user, const void* x3r, void* x4s, int stride, int own_pass);
  int (*phase_force)(void* user, const void* x3r, int stride, int32_t* counts, int* fits);
  int (*phase_end)(void* user, void* x1s);
} bh_rank_script;
int bh_rank_create_scripted(bh_rank** out, const bh_comm* comm, const bh_rank_script* script, const bh_rank_plan* plan,
                            const bh_rank_opts* o);

typedef struct bh_group bh_group;
/* transport: 0 = automatic (RCCL when all devices differ, else in-process copies), 1 = RCCL, 2 = in-process copies */
int bh_create_group(bh_group** out, int ngpus, const int* devices, int64_t n_total, const bh_params* p,
                    const bh_rank_opts* o, int transport);
/* the whole system, caller order (as bh_upload): keyed and sorted once on devices[0], slab q of the curve -> rank q */
int bh_group_upload(bh_group* g, const float* x, const float* y, const float* z, const float* vx, const float* vy,
                    const float* vz, const float* m);
int bh_step_group(bh_group* g, int steps); /* returns when every rank has enqueued its steps; BH_ERR_DOMAIN_LEFT or the
                                              first rank-local error (the other ranks are released) */
int bh_group_sync(bh_group* g);            /* bh_sync of every rank */
int bh_group_download(bh_group* g, float* x, float* y, float* z, float* vx, float* vy, float* vz); /* caller order */
int bh_group_download_acc(bh_group* g, float* ax, float* ay, float* az);                           /* caller order */
int bh_group_size(const bh_group* g);
bh_rank* bh_group_rank(bh_group* g, int rank);
void bh_destroy_group(bh_group* g);

/* per-step device times (hipEvent pairs recorded on the context's stream while
   bh_set_timing is on) of the most recent steps, oldest first: ms_force[i], ms_step[i].
   Synchronises.  At most BH_TIMING_RING steps are kept. */
#define BH_TIMING_RING 256
int bh_timing_history(bh_ctx* c, float* ms_force, float* ms_step, int capacity, int* count);

/* ---- state dump / restart (host side, no GPU needed) ---- */
/* text format of the reference's older generation (output_bh.txt:1-4):
     # Barnes-Hut N-Body Simulation Results / # Final positions and velocities after K steps /
     # Bodies: N, Theta: T, dt: D / # Format: x y z vx vy vz / one "%f %f %f %f %f %f" row per body */
int bh_write_text(const char* path, int n, int steps, float theta, float dt, const float* x,
                  const float* y, const float* z, const float* vx, const float* vy, const float* vz);
/* reads at most capacity bodies; *n_out = bodies in the file (BH_ERR_SMALL_BUFFER if more) */
int bh_read_text(const char* path, int capacity, int* n_out, int* steps_out, float* x, float* y,
                 float* z, float* vx, float* vy, float* vz);
/* lossless binary snapshot: header {magic "BHSNAP01", n, steps, bh_params} + 7 SoA float arrays */
int bh_write_snapshot(const char* path, int n, int steps, const bh_params* p, const float* x,
                      const float* y, const float* z, const float* vx, const float* vy,
                      const float* vz, const float* m);
int bh_read_snapshot(const char* path, int capacity, int* n_out, int* steps_out, bh_params* p_out,
                     float* x, float* y, float* z, float* vx, float* vy, float* vz, float* m);

/* ---- synthetic initial conditions (host side; <-> IC loop ref:294-308) ---- */
/* Plummer sphere, scale radius a, masses U[2,7) (same law as ref:302), counter-based
   splitmix64 RNG keyed on (seed, body index): SURVEY §8(d). */
int bh_ic_plummer(int n, uint64_t seed, float a, float G,
                  float* x, float* y, float* z, float* vx, float* vy, float* vz, float* m);
/* the reference's rotating thin disc (ref:297-307) with the same RNG */
int bh_ic_disc(int n, uint64_t seed, float G,
               float* x, float* y, float* z, float* vx, float* vy, float* vz, float* m);
/* the same disc exactly as the reference BINARY draws it: srand(seed) and the Microsoft C runtime's rand()
   (the authors' nbody_v5_bench.exe is an MSVC build; RAND_MAX 32767) in the call order of ref:294-308, so
   that literal_force trajectories can be diffed against the CUDA program's output (seed 42 = ref:294) */
int bh_ic_disc_msvc(int n, uint32_t seed, float G,
                    float* x, float* y, float* z, float* vx, float* vy, float* vz, float* m);

#ifdef __cplusplus
}
#endif
#endif /* BH_H_ */
