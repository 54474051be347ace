// bh_internal.h — context layout and kernel-launcher prototypes shared by the
// translation units of libbh.so.  Not part of the public ABI (include/bh.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bh.h"

typedef unsigned long long u64;
typedef unsigned int u32;

// fast-kernel record ("digest"), written by the COM stage.  LOGICAL fields:
struct bh_frec {
  float x, y, z;  // centre of mass
  float gm;       // G*m, 0 when m <= 0
  float thr2;     // (s/theta)^2; -1 for a body or a mass<=0 record (always accepted)
  int first;      // child block; body index for a body; digest slot of the first body for an unsplit multi-body cell
  int meta;       // child count (bodies of an unsplit cell count as its children)
  int pad;        // LINK = (first << 5) | min(meta, 63): byte offset of the child block and its size in one dword —
                  // what the hand-scheduled walk keeps on its cross-lane stack (one lane write per push, one lane
                  // read per pop instead of two); derived by frec_put / the COM stage, never set by callers
};
// PHYSICAL layout of a digest pool: records 2p and 2p+1 share one 64-byte PAIR, fields interleaved
//   dword  0 1 | 2 3 | 4 5 | 6  7  | 8    9    | 10     11     | 12    13    | 14 15
//          x0 x1 y0 y1 z0 z1 gm0 gm1 thr0 thr1   first0 first1   meta0 meta1   pad
// so one s_load_dwordx16 brings two records with every field of the two in an adjacent, even-aligned SGPR
// pair — the operand shape of the packed fp32 VALU instructions (v_pk_add/mul/fma_f32), which cost the
// same issue time as ONE scalar-operand fp32 instruction but evaluate both records (tools/ubench_forms.hip).
// Child blocks start at even record indices, so a block is a whole number of pairs; a block with an odd
// number of children ends in a NULL record (gm = 0, thr2 = -1: accepted by every body with zero force).
// A pool pointer is typed bh_frec* (32 bytes per record) but must only be dereferenced through these accessors.
#define BH_FREC_DW(e, field) ((size_t)((e) >> 1) * 16 + (size_t)(field) * 2 + ((e) & 1))
enum { BH_FF_X = 0, BH_FF_Y, BH_FF_Z, BH_FF_GM, BH_FF_THR2, BH_FF_FIRST, BH_FF_META, BH_FF_PAD };
__host__ __device__ __forceinline__ bh_frec frec_get(const bh_frec* pool, long long e) {
  const float* f = reinterpret_cast<const float*>(pool) + (size_t)(e >> 1) * 16 + (e & 1);
  const int* i = reinterpret_cast<const int*>(f);
  bh_frec r;
  r.x = f[0]; r.y = f[2]; r.z = f[4]; r.gm = f[6]; r.thr2 = f[8];
  r.first = i[10]; r.meta = i[12]; r.pad = i[14];
  return r;
}
// child blocks start at even records, so bit 5 of first << 5 is free as well: 6 bits of count (63 = "63 or more":
// the walk then redoes the wave with the generic loop, which reads `meta`)
__host__ __device__ __forceinline__ int frec_link(int first, int meta) {
  const unsigned c = meta < 0 ? 0u : (meta > 63 ? 63u : (unsigned)meta);
  return (int)(((unsigned)first << 5) | c);
}
__host__ __device__ __forceinline__ void frec_put(bh_frec* pool, long long e, const bh_frec& r) {
  float* f = reinterpret_cast<float*>(pool) + (size_t)(e >> 1) * 16 + (e & 1);
  int* i = reinterpret_cast<int*>(f);
  f[0] = r.x; f[2] = r.y; f[4] = r.z; f[6] = r.gm; f[8] = r.thr2;
  i[10] = r.first; i[12] = r.meta; i[14] = frec_link(r.first, r.meta);
}
__host__ __device__ __forceinline__ bh_frec frec_null() {
  bh_frec z;
  z.x = z.y = z.z = z.gm = 0.0f;
  z.thr2 = -1.0f;
  z.first = 0; z.meta = 0; z.pad = 0;
  return z;
}
// digest slot of body b of the unsplit multi-body cell whose first body is lo: the cell's bodies form a child
// block, which must start at an even slot; the ranges [2 lo, 2 hi) of different cells do not overlap
#define BH_BODY_DIGEST(rec_cap, lo, b) ((rec_cap) + 2 * (lo) + ((b) - (lo)))
#define BH_FREC_POOL(rec_cap, n) ((size_t)(rec_cap) + 2 * (size_t)(n) + 8)  // tree digests + body digests + window pad

struct bh_d4 {  // fp64 prefix-sum element: (sum m, sum m*x, sum m*y, sum m*z)
  double m, x, y, z;
};

// stage bits for order checking
enum {
  BH_ST_UPLOADED = 1,
  BH_ST_BBOX = 2,
  BH_ST_MORTON = 4,
  BH_ST_SORT = 8,
  BH_ST_BUILD = 16,
  BH_ST_COM = 32,
  BH_ST_FORCE = 64,
};

// device-resident scalar block (no per-step host sync needed to read any of it)
struct bh_devinfo {
  int n_internal;  // M
  int n_entries;   // E
  int max_level;
  int flags;       // BH_FLAG_*
  int redo_waves;  // force waves that redid their walk with the generic loop (stack > 64 entries or a block of > 8 children)
  int slow_buckets;  // splitter-sort buckets that did not fit LDS (sorted by one workgroup through global memory)
  int dd_hold;     // domain-decomposed step: the X4 that just arrived does not fit or a rank has left (dd_validate_kernel):
                   // the force launches enqueued behind it return at once (bh_force.hip force_held) — they are enqueued
                   // before the host has looked at the headers, which it does while they run
  int pad[1];
};

struct bh_ctx {
  int n;
  int num_cus;  // compute units of the device (force-kernel placement heuristics)
  bh_params p;
  int B, D, cap;  // bits per axis, effective max depth, leaf cap
  int device;
  hipStream_t stream;
  bool own_stream;
  int last_hip;
  int stage;
  int ever;  // stages that have run at least once since upload (downloads of stale-but-present data)
  int steps;

  // particle state, Morton order after the first sort (HBM, float4 = one 16-B access per lane)
  float4* posm[2];   // (x,y,z,m)           ping-pong (gather target)
  float4* velid[2];  // (vx,vy,vz,bits(id))  ping-pong
  int cur;
  float4* acc;       // (ax,ay,az,0) — engine-owned or caller-bound (bh_bind_acc)
  float4* acc_own;
  float4* acc2;      // second addend of the accelerations (two-pass force of bh_dd.hip), or null
  float* stage_buf;  // 7n floats: SoA staging for upload/download

  // keys
  u64* keys[2];
  u32* vals[2];
  int key_buf;  // which keys[] holds the sorted keys
  u32* hist;    // 256 * ntiles
  // onesweep sort (bh_sort_onesweep.hip)
  u32* sw_hist;    // [8][256] global digit totals of every pass
  u64* sw_status;  // [passes][ntiles][256] look-back granules {tag|state|count}
  u32* sw_ticket;  // [0..7] tile tickets of the passes (cleared at the end of every sort), [8] sort calls so far,
                   // [9] finished blocks of pairs_kernel, [10] of integrate_kernel (cleared by their last block)
  u32 sort_calls;
  int sort_tiles;
  // splitter sort (bh_sort_onesweep.hip, bhk_sort_split)
  u64* sp_keys;     // [256] sorted splitters, padded with ~0
  u32* sp_count;    // [2][256] bucket sizes, double-buffered by call parity (the sort clears the other half)
  int sp_par;
  bool keys_split;  // keys[0] and sp_count[sp_par] come from keys_split_kernel and no sort has consumed them
  bool order_hint;  // the bodies are stored in the key order of an earlier sort (set by every sort, cleared by
                    // uploads): what makes evenly spaced bodies good splitters
  bool splitter_off;  // bh_get_stats saw LDS-overflowing buckets: radix passes until the next upload
  int slow_seen;
  u32 slow_seen_sorts;  // sort_calls at the last look

  // bbox
  float* bbox_partial;  // [BH_BBOX_BLOCKS][6]
  float* bounds;        // [8]: min xyz, min+size xyz, s0, pad
  float* bounds_next;   // [8]: the cube of the positions the last integrate wrote (valid iff bounds_next_ok)
  float* ibox_rows;     // [n / 256 + 1][6] per-block min / max of the integrate kernel
  bool bounds_next_ok;  // set by a step's integrate, cleared by anything else that writes positions

  // tree build temporaries
  signed char* d8;  // [n+1] leading octal digits shared by keys j-1, j; d8[0] = d8[n] = -1
  u64* ksamp;       // [<= 2048] every 2^ss-th sorted key (bisection seeds of the wide-cell searches)
  int* pa;        // [n] first body of the cell whose first child boundary is j
  int* pb;        // [n] end body of that cell
  int* pn;        // [n] its child count (0: j represents no emitted cell)
  int* cb;        // [n] offset of the cell's child block inside its 1024-pair tile (exclusive scan of the
                  // even-rounded pn within the tile)
  u32* blk_done;  // [2][n / 8192 + 4] block counters of bh_last_block: pairs_kernel, then (blk_done2) integrate_kernel
  u32* blk_done2;
  int* ttot;      // [2 * (n / 256 + 2)] child entries per pair tile (1024 or 256 pairs), then their exclusive
                  // prefix (tile bases)
  bh_node* rec;   // [rec_cap] tree records (canonical: ABI download, strict/counting kernels)
  bh_frec* frec;  // [BH_FREC_POOL] digests for the fast force kernel (written by COM, pair layout): tree records,
                  // then BH_BODY_DIGEST slots (used only for the bodies of unsplit multi-body cells)
  int* er_lo;     // [rec_cap] body range of each record (written by the canonical COM stage)
  int* er_hi;
  bool rec_proto; // rec / er_lo / er_hi were NOT made canonical by the last COM stage (bh_step of the default engine)
  bool com_digests;  // a COM stage has run on the current tree: digests and prefix sums P belong to rec (what
                     // bhk_canonical_records needs to make proto records canonical after the fact)
  int rec_cap;
  bh_d4* P;       // [n+1] fp64 exclusive prefix of (m, m x, m y, m z) over sorted bodies
  bh_devinfo* info;
  int* host_flags;  // pinned: bh_sync reads the sticky flags through it

  // scan scratch
  void* scan_tmp;
  size_t scan_cnt_off;  // byte offset of the bh_last_block counters inside scan_tmp / scan_tmp2
  size_t scan_tmp_bytes;
  void* scan_tmp2;  // scratch of the COM prefix scan that rides in the build's launches (bhk_build pm_scan)

  // domain-decomposed stepping (bh_dd.hip); null until bh_dd_init
  struct bh_dd_state* dd;
  bh_frec* frec_own;  // the context's own record pool while frec points into a caller pool
  // force + integrate in one launch (bh_step, force_fast_kernel FUSE): per-wave and per-group min / max rows, counters
  float* fuse_rows;   // [fuse_waves + fuse_waves / 32 + 2][6]
  u32* fuse_cnt;      // [fuse_waves / 32 + 3], zero between launches
  int fuse_waves;
  float* dd_minmax;   // [8] dd mode: this rank's min / max of the positions the last integrate wrote (X1 payload)
  bool dd_minmax_ok;  // set by bh_integrate in dd mode, cleared by anything else that writes positions
  bool dd_integrated; // the last force pass of the domain-decomposed step already integrated the bodies (bh_dd_force)

  // counters (bh_force_count)
  u32 *cV, *cO, *cP;
  u64 tV, tO, tP;

  // bh_step as a HIP graph, one per parity of `cur` at entry (bh_api.hip)
  hipGraphExec_t gexec[2];
  int g_keybuf[2];
  bool graph_failed;

  // timing
  bool timing;
  int timing_mode;  // 1: event after every stage; 2: only the pair around the force launch
  hipEvent_t* evring;  // [BH_TIMING_RING][8], created by the first bh_set_timing(1)
  long timed_steps;    // steps recorded into the ring since timing was switched on
};

// Record layout: entry 0 = root, entry 1 = padding, child blocks start at EVEN entries (a block of an odd
// number of children is followed by one padding entry), so a block never straddles more 64-byte lines than
// it has to: the force kernel's scalar loads are bound by the number of cache-line requests.
// Entries <= root + bodies + cells + one pad per cell <= 3n.
// EVEN: BH_BODY_DIGEST blocks start at rec_cap + 2 lo and a child block must start at an even record (64-byte pair)
#define BH_REC_CAP(n) ((3 * (n) + 8 + 1) & ~1)
#define BH_BLOCK0 2  // first child block
#define BH_FORCE_BLOCK_DEFAULT 64  // one wave per workgroup: a CU slot frees as soon as its wave retires (-2 % at 1M)
#define BH_BBOX_BLOCKS 1024
#define BH_INTEGRATE_TILE 4096  // bodies per integrate block (1024 threads x 4)
#define BH_PAIR_SMALL_N 163840  // bodies up to which the tree build uses 256-pair tiles (A/B per step: 16,384 -15 us,
                                // 65,536 -14 us, 125,000 -8 us, 262,144 +5 us)
#ifndef BH_KS_SMALL_N
#define BH_KS_SMALL_N 163840   // bodies up to which keys_split_kernel takes one key per thread (1024-key blocks)
#endif
#ifndef BH_INT_SMALL_N
#define BH_INT_SMALL_N 163840  // bodies up to which integrate_kernel<true> uses 1024-body blocks
#endif
#define BH_BLKDONE_STRIDE(n) ((size_t)(n) / 8192 + 4)  // one counter per 32 tiles of >= 256 pairs
#define BH_SCAN_TILE 2048  // 256 threads x 8 items
#ifndef BH_SORT_TILE
#define BH_SORT_TILE 4096  // keys per sort tile
#endif
#define BH_SORT_ITEMS (BH_SORT_TILE / 256)  // per thread in the 256-thread kernels

// "Last block finishes the job" hand-off without a release fence.  On a multi-XCD part __threadfence() writes
// back the XCD's whole L2 (measured: 3907 blocks doing one each took integrate_kernel from 14 to 150 us).
// Instead the per-block results are stored write-through at agent scope (bh_publish_*), the block waits until
// those stores are acknowledged (bh_published: then they are visible device-wide), only then counts itself
// done with a relaxed atomic; the block that sees the full count reads the results with agent-scope loads
// (bh_collect_*), which bypass the non-coherent caches.
// ISA ASSUMPTION (gfx942 / gfx950, the only targets this library is built for): an agent-scope relaxed atomic
// store compiles to a global_store with sc1 (write-through to the memory-side level shared by all XCDs) and
// increments vmcnt; `s_waitcnt vmcnt(0)` returns only when the write has been acknowledged there, i.e. when an
// agent-scope load of any other CU (sc1 load: misses the XCD-private L2) returns the new value.  The HIP / LLVM
// memory model does NOT promise a happens-before edge between these relaxed operations and the relaxed counter
// increment that follows — the ordering is this ISA behaviour plus the `asm volatile(... "memory")` compiler
// barrier.  Any other target gets the formally correct (slower) release / acquire pair below; tools/soak.py and
// tests/test_gpu_soak.py are the regression check for the fast form.
#if defined(__HIPCC__) && !defined(__HIP_DEVICE_COMPILE__)
#define BH_FENCE_FREE_HANDOFF 1  // host pass: the device pass below decides
#elif defined(__gfx942__) || defined(__gfx950__)
#define BH_FENCE_FREE_HANDOFF 1
#else
#define BH_FENCE_FREE_HANDOFF 0
#endif
#ifdef __HIPCC__
__device__ __forceinline__ void bh_publish_i32(int* p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void bh_publish_f32(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void bh_published() {
#if BH_FENCE_FREE_HANDOFF
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
  __atomic_thread_fence(__ATOMIC_RELEASE);  // formally ordered before the counter increment that follows
#endif
}
__device__ __forceinline__ int bh_collect_i32(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float bh_collect_f32(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// One thread of block b (of nb) calls this after bh_published(); true for exactly one block, the last.
// Two-level count — blocks in groups of 32, cnt[1 + group], then cnt[0] over the groups — because returning
// atomics on ONE address serialise at ~10 ns each (3907 blocks: +40 us).  The counters are left at zero.
__device__ __forceinline__ bool bh_last_block(u32* cnt, int b, int nb) {
  const int g = b >> 5;
  const int gsize = min(32, nb - (g << 5));
  if (__hip_atomic_fetch_add(cnt + 1 + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (u32)(gsize - 1))
    return false;
  __hip_atomic_store(cnt + 1 + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int ngroups = (nb + 31) >> 5;
  if (__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (u32)(ngroups - 1))
    return false;
  __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}
#endif

// ---- launchers (each enqueues on c->stream and returns hipGetLastError()) ----
hipError_t bhk_pack(bh_ctx* c);                       // stage_buf SoA -> posm/velid
hipError_t bhk_unpack(bh_ctx* c, int what);           // 0: pos+vel -> stage_buf (caller order); 1: acc
hipError_t bhk_bbox(bh_ctx* c);
hipError_t bhk_bbox_raw(bh_ctx* c, float* out6);  // local min/max only
hipError_t bhk_bounds_from_rows(bh_ctx* c, const float* rows, int nrows, int stride_floats);
hipError_t bhk_keys(bh_ctx* c, bool for_sort = true);
hipError_t bhk_sort(bh_ctx* c);  // sort + gather
hipError_t bhk_sort_onesweep(bh_ctx* c);              // radix implementation (bh_sort_onesweep.hip)
bool bhk_sort_split_eligible(const bh_ctx* c, int n_upper = 0);  // splitter sort: keys + bucket counts, then partition + local sort
hipError_t bhk_keys_split(bh_ctx* c, const int* n_dev = nullptr, int n_upper = 0);
hipError_t bhk_sort_split(bh_ctx* c);
hipError_t bhk_build(bh_ctx* c, bool pm_scan = false);  // pm_scan: + the COM prefix scan (small steps, bh_tree.hip)
hipError_t bhk_com(bh_ctx* c);
hipError_t bhk_force(bh_ctx* c, int lo, int hi, bool count, bool fuse_integrate = false, bool* fused = nullptr);
bool bhk_force_range_aligned(const bh_ctx* c, int lo);  // a slab starting at lo forms the full launch's groups
hipError_t bhk_force_root(bh_ctx* c, int lo, int hi, int root, hipStream_t stream, float4* acc,
                          const float4* fuse_add = nullptr, bool fuse = false,
                          bool* fused = nullptr, int fold_groups = 0);  // fast kernel from pool record `root`
hipError_t bhk_force_walk_stats(bh_ctx* c, u32* rows, int root = 0);  // measurement: per-wave event counters of the fast walk
int bh_walk_stats_from(bh_ctx* c, int root, bh_walk_stats* out);       // bh_api.hip: that launch + its reduction on the host
int bhk_force_walk_rows(const bh_ctx* c);               // waves (rows) of that launch
#define BH_WALK_ROW 16                                  // u32 words per row
hipError_t bhk_force_trace(bh_ctx* c, u32* trace, int cap_rows, int* rows);  // measurement: one row per wave
hipError_t bhk_integrate(bh_ctx* c, bool with_bbox);
void bh_dd_free(bh_ctx* c);  // bh_dd.hip
// measurement (bh_rank_replay_force_phase, bh_group.hip): the force phase of the last completed step run again
extern "C" int bh_dd_replay_begin(bh_ctx* c, int split, int split_pct, int saved[4]);
extern "C" int bh_dd_replay_end(bh_ctx* c, const int saved[4]);
extern "C" int bh_dd_idle_wave(bh_ctx* c, int us);
extern "C" int bh_dd_set_one_pass(bh_ctx* c);
extern "C" int bh_dd_x4_sizes(bh_ctx* c, int stride, int allow_prev, int64_t* send_bytes, int64_t* recv_bytes);

// device-wide scans (bh_scan.hip)
hipError_t bhk_scan_i32(bh_ctx* c, const int* in, int* out /* n+1 */, int n, const int* n_dev);
hipError_t bhk_scan_i32_even(bh_ctx* c, const int* in, int* out /* n+1 */, int n);  // of (in[i]+1)&~1
hipError_t bhk_scan_pm(bh_ctx* c, const float4* posm, bh_d4* out /* n+1 */, int n);
hipError_t bhk_com_records(bh_ctx* c, bool canonical, int* spine_pieces = nullptr, int* spine_count = nullptr);
                           // canonical = false: digests only; spine_pieces (digest-only pass of the decomposed step):
                           // also lists the rank's pieces (com_kernel)
hipError_t bhk_canonical_records(bh_ctx* c);  // proto records -> canonical, after a digest-only COM stage
size_t bhk_scan_tmp_bytes(int n);
size_t bhk_scan_cnt_offset(int n);
