// bh_dd.hip — the engine side of the domain-decomposed multi-GPU step (include/bh.h "bh_dd_*").
//
// The reference is single-GPU; nothing here has a reference counterpart.  One context per rank owns the bodies of ONE
// INTERVAL OF THE KEY CURVE.  This file packs and consumes plain device buffers and never communicates: the per-step
// protocol that moves them (three all-gathers and one all-to-all, size negotiation, retries, collective failure) is
// bh_rank_step in bh_group.hip; DESIGN.md §6 is the overview.
//
//  X1 cube + boundaries  each rank's min / max (exact, so the global cube — hence every key — is bit-identical to the
//                        single-GPU run), its body count, the bodies it proposes for its two domain boundaries and a
//                        regular sample of its positions.  The splitter KEYS PERSIST from step to step: a rank owns a
//                        fixed interval of the curve and a body changes owner when its key leaves it.  They move only
//                        when some rank's count leaves n / P by more than 1.5 % (decided from the gathered counts, so
//                        collectively), then to the exact quantiles the ranks propose — shifted against the drift of
//                        the counts (dd_want) — or, first step / far out of balance, to sample quantiles
//                        (dd_split_kernel; a position re-keyed under a new cube can jump octants, a key cannot).
//  X2 migration          owner(key) = #{splitter keys <= key}.  Bodies whose owner changed are compacted into the
//                        exchange buffer (its used size follows the observed count; a larger wave leaves in further
//                        rounds of the same step); every rank picks its immigrants out of the gathered buffers.
//                        Scan-based, so arrival order is deterministic.
//  X3 piece descriptors  after the local sort / build / COM.  A local cell that does not touch either end of the local
//                        body range is a complete global cell.  The end-touching cells form two root-to-leaf spines;
//                        their other children are the rank's PIECES (<= 2 x 21 x 7).  The canonical octree above all
//                        pieces (top tree) is a pure function of the piece keys: every rank rebuilds it identically
//                        (dd_top_kernel).
//  X4 LET segments       child blocks of every local cell some body of another rank could open (conservative test of
//                        the cell's MAC radius against the boxes of the remote pieces), one segment PER DESTINATION
//                        (all-to-all; let_mode 0: one union segment, all-gather), written with pool-relative child
//                        indices so that the received segments are traversable in place.  Every segment carries its
//                        sender's needs for all receivers: every rank holds the same needs matrix and takes the same
//                        decision on the next stride — and on what every PAIR moves in the next exchange (x4_chunk of
//                        what the pair needed: the slots are two thirds padding; bh_dd_x4_sizes).
//
// The stitched pool [local tree + body digests | two top trees | world x LET segment] is the same canonical octree a
// single GPU builds; the unchanged force walk traverses it from a top-tree root.
//
// Force passes: one pass over the stitched tree after X4 (the default), or — the split form, for the first split_pct
// per cent of the rank's bodies; it pays when X4 lasts ~0.1 ms and more: bh_group.hip rank_adapt — two: the own pieces (two thirds of the pair work; they need nothing from
// other ranks) on a side stream, launched behind the LET export, with the other ranks' pieces as null records and
// every top cell carrying this rank's share of its mass, while the main stream runs X4; then the mirror image (top
// tree re-emitted from the first one's structure), whose launch adds the own pass's accelerations — beside the ONE
// pass of the other bodies; both integrate and fold this rank's min / max.  Both passes apply the same MAC to the
// same cells, so the split is exact up to summation order.  The side stream is created with a CU mask
// (dd_make_side_stream: why, and what was measured).
//
// Safety: every walk over imported records is bounded; dd_validate_kernel closes malformed imported records before
// anything walks them (BH_FLAG_DD_LET_INVALID); an X4 segment that does not fit is sent closed (pieces unopenable)
// and the exchange is repeated larger; sticky flags of a step surface at the next step's bh_dd_migrate_apply.
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "bh_internal.h"
#include "bh_keys.h"

struct bh_dd_piece {  // X3 descriptor, 80 B
  u64 key;                // key of the piece's first body
  float bx, by, bz, bs;   // box: min corner and edge (0 for a single body)
  double sm, sx, sy, sz;  // fp64 sums (m, m x, m y, m z) of its bodies
  int rec_idx;            // local record index on the owner
  int owner;
  int count;              // bodies
  int kind;
  int pad[2];
};
static_assert(sizeof(bh_dd_piece) == 80, "descriptor layout");

// piece sums of the top tree: (m, m x, m y, m z) of all bodies and o = the mass owned by this rank
struct top5 {
  double m, x, y, z, o;
};

struct bh_dd_state {
  int world, rank;
  long long n_total;
  int mig_cap, let_cap;
  bh_frec* pool;
  long long pool_records;
  int top_base, top_base2, top_base3, seg_base;  // top tree of the remote pass (or the whole tree) / of the own pass /
                                                 // the whole tree beside a remote one (partial two-pass steps)
  int split_pct;   // two-pass steps: per cent of the rank's bodies whose walk is split into own + remote pass (100:
                   // all of them); the others are walked in one pass after X4
  int own_hi;      // bodies [0, own_hi) were walked by this step's own pass
  hipEvent_t ev_x4;
  int* w;          // [rec_cap + 1] records a cell exports (0: not needed by any other rank)
  int* dst;        // [rec_cap + 1] exclusive scan of w
  int* flag;       // [max(n_cap, world*mig_cap) + 1]
  int* fpos;       // same, exclusive scan of flag
  int* nloc;       // [world] body count of every rank after this step's migration
  int samp_cap;    // sample slots per rank in X1 (kSampTotal / world)
  u64* skeys;      // [world-1] splitter keys: they PERSIST from step to step (a rank owns a fixed interval of the
                   // curve) and move only when a rank's body count leaves the tolerance band (dd_split_kernel)
  int* drift;      // [0..63] per-rank drift of the body count, bodies per step x 16 (smoothed over the steps that kept the
                   // boundaries); [64..127] every rank's count at the previous step's X1; [128] counts valid, [129]
                   // estimate valid.  Written by dd_split_kernel from all-gathered counts: identical on every rank
  int* piece_tmp;  // [BH_DD_PIECE_CAP] unsorted piece records
  int* piece_idx;  // [BH_DD_PIECE_CAP] pieces in body order
  int* ddi;        // [16] device scalars: 0 piece counter, 1 remote boxes, 2 top pieces, 3 top children,
                   //      4..7 migration results, 8 pieces of this step, 9 steps that moved the splitters,
                   //      10 splitter keys valid, 11 what the last dd_split_kernel did (0 kept, 1 exact
                   //      quantiles from the ranks' candidates, 2 sample quantiles), 12 emigrants found on this rank
  float4* boxes;   // [world * BH_DD_PIECE_CAP] remote piece boxes (corner, edge), margin applied
  float4* rbox;    // [2 * world] bounding box of each remote rank's pieces + its range in boxes[]
  top5* top_ps;    // [2][kTopMax + 1] fp64 prefix of the piece sums (one per pass: the passes overlap)
  int* top_a;      // [2 kTopMax] first piece of every top-tree child
  int* top_b;      // [2 kTopMax] end piece of every top-tree child
  int4* top_ci;    // [2][2 kTopMax + 8] per top record: branching level (-1: piece), child offset / slot, count
  float4* acc2;    // [n_cap] accelerations of the remote pass (the own pass writes the context's acc)
  hipEvent_t ev_x3, ev_top1, ev_own;
  hipStream_t stream_own;  // the side stream of the two-pass forms: restricted to all but kReserveCus compute units, so that
                           // what the main stream launches while a pass holds the GPU (X4's kernel, the validation, the
                           // top trees) finds free units at once (dd_make_side_stream)
  bool split;      // two-pass force: own pieces while X4 is in flight, remote pieces after it
  bool serial;     // the own pass runs on the context's main stream instead of the side stream (bh_dd_set_serial: ranks
                   // that share one GPU in a rehearsal — their side streams would overlap each other's work)
  int* host;       // pinned: [world] LET counts, [64 .. 67] migration results, [68] their sequence number
  hipEvent_t ev_let;  // the LET export of this step has finished (main stream): the own pass may take the GPU
  bool replay;        // bh_dd_replay_begin .. _end (measurement): the force passes do not integrate
  bool top_early;     // one-pass step: the top tree's structure was built beside the LET kernels (dd_top_early)
  bool keys_spec;     // the key kernel of this step's sort was launched while the host waited for the body count
  // X4 with a size per pair (bh_comm.all_to_all_v): what rank q sends rank j this step is limited to
  // x4_chunk(what q needed for j in the LAST fitting X4) — every rank holds that matrix (the needs rows in the segment
  // headers) on the host (prev_rows) and on the device (prev_rows_dev, kept by dd_validate_kernel) and derives the
  // same sizes; a pair that outgrows its size makes the exchange "not fit": repeated with one size for all
  int* prev_rows;      // [world][32] host copy of the header rows of the last fitting X4
  int* prev_rows_dev;  // the same on the device
  bool prev_ok;        // prev_rows holds a matrix
  bool x4_v;           // this try's X4 moves a size per pair
  bool let_copy_pending;
  int* host_rows;  // pinned: [world][32] header + needs row (records 0..3) of every received X4 segment
  int let_mode;    // 0: X4 is an all-gather of the union every other rank may open; 1: per-destination segments
                   // (all-to-all)
  unsigned* wmask; // [rec_cap + 1] per-destination mode: ranks (bit q) that may open cell e
  int* list_e;     // [n_cap] exporting cells in record order
  int* list_w;     // [n_cap] their block sizes (child count rounded up to even)
  unsigned* list_m;  // [n_cap] their rank masks
  int* dstd;       // [world][n_cap] per destination: block offset of every exporting cell inside that segment
  int* dtot;       // [64] per destination: records of its blocks
  int* csum;       // [world][n_cap / 8192 + 1] chunk sums of the per-destination scan
  int* mark_cnt;   // [2][rec_cap / 1024 + 3] exporting cells per dd_mark_kernel block, then their exclusive bases
  u32* mark_done;  // bh_last_block counters of dd_mark_kernel
  u32* cls_done;   // bh_last_block counters of dd_classify_kernel
  u32* abs_done;   // ... of dd_absorb_flag_kernel
  int* arrive;     // [64] immigrants per rank of the current round (left at zero by the kernel)
  int absorb_seq;  // sequence number of the last dd_absorb_kernel launch
  int let_seq;     // ... of the last dd_validate_kernel launch (host[70]: the received X4 headers are in host_rows)
};

namespace {

constexpr int kB = 21;
constexpr int kTopMax = 4096;  // pieces in the whole system
constexpr int kDescPerRank = 1 + BH_DD_PIECE_CAP;
// X4 segment: record 0 header (first = records the sender needed for THIS receiver, meta = its pieces), records
// 1 .. 3 the sender's NEEDS ROW (24 ints: records it needed for receiver j — every rank thus sees the whole
// world x world matrix and takes the same decision on the next stride), records kSegPieces0 .. + PIECE_CAP the
// pieces' own records, then the exported child blocks from an EVEN record on (digest pairs: bh_internal.h)
constexpr int kSegPieces0 = 4;
constexpr int kSegBlocks0 = kSegPieces0 + BH_DD_PIECE_CAP;
static_assert(kSegBlocks0 % 2 == 0, "child blocks start at even records");
// dword of needs-row entry j inside a segment (pair layout: record r, field f -> (r >> 1) * 16 + 2 f + (r & 1))
__host__ __device__ inline int seg_row_dword(int j) {
  const int r = 1 + j / 8, f = j % 8;
  return (r >> 1) * 16 + 2 * f + (r & 1);
}
// records pair (q -> j) may move when sizes follow the last step's needs: a quarter more than it needed then + 4096
// (measured at 8 x 1M, tools/dd_needs.py: a pair's need grows 1-2 % per step and by up to 20,000 records — 22 % of
// a near pair's 90,000, several times a far pair's 1,000-10,000 — in a step that moves the domain boundaries), whole
// 256-record units, never more than the stride (the slot)
__host__ __device__ inline int x4_chunk(int need_prev, int stride) {
  const long long c = (((long long)need_prev * 5 / 4 + 4096 + 255) / 256) * 256;
  return (int)(c < stride ? c : stride);
}
constexpr int kNeedsRowMax = 24;  // ranks a segment's needs row has room for (records 1..3)
constexpr int kTopCap = 4 * 4096 + 8;  // records of one top tree incl. padding (kTopMax pieces)

constexpr int kSampTotal = 2048;  // position samples in the whole system (bitonic sort in LDS by one block, on the
                                  // critical path of every step: 49 us with 4096, quantile error 1/256 of a rank at 8 ranks)
__host__ __device__ inline int samp_cap_of(int world) { return kSampTotal / world; }
// X1 payload of a rank, floats: [0..5] min / max, [6] body count, [7] -, [8..11] / [12..15] the positions this rank
// proposes for its lower / upper domain boundary (w = 1: valid), [16 ..] position samples
constexpr int kX1Samples0 = 16;
__host__ __device__ inline int x1_floats(int world) { return kX1Samples0 + 4 * samp_cap_of(world); }
constexpr float kSplitTolerance = 0.015f;  // a rank's body count may leave n / P by this fraction before the
                                           // boundaries move
constexpr float kSplitHardBound = 0.10f;   // ... and beyond this fraction the proposals are not trusted: sample quantiles
constexpr int kDriftHorizon = 4;           // a rebalance aims at where the quantile will be this many steps ahead

// Where boundary q (between ranks q and q + 1) should stand in the global body order when the boundaries move: at the
// quantile (q + 1) n / P, shifted against the drift the ranks below it have shown — every rank sees every rank's count
// in X1, so the drift of a rank's count while the boundaries stood still is known everywhere (drift16: bodies per step
// x 16).  A rank that gains d bodies per step is given kDriftHorizon d bodies less than its share (at most 0.8 of the
// tolerance band, `cap`), so that its count crosses the band once instead of leaving it from the middle: the
// rebalances come about half as often for the same imbalance (the reference's per-step cube moves the key grid under
// the bodies, which is what makes the counts drift: DESIGN.md §6).  Pure integer arithmetic on all-gathered data:
// dd_x1_pack_kernel (which proposes) and dd_split_kernel (which checks) agree on every rank.
__device__ inline long long dd_want(int q, int world, long long n_total, const int* drift16, int est_valid,
                                    long long cap) {
  long long want = (long long)(q + 1) * n_total / world;
  if (!est_valid || cap <= 0) return want;
  long long sum = 0, below = 0;
  for (int r = 0; r < world; r++) {
    long long b = (long long)drift16[r] * kDriftHorizon / 16;
    b = b > cap ? cap : (b < -cap ? -cap : b);
    sum += b;
    if (r <= q) below += b;
  }
  const long long mean = sum / world;  // (the drifts add up to zero; what clamping leaves over is spread evenly)
  return want - (below - mean * (q + 1));
}

__device__ __forceinline__ int owner_of(u64 key, const u64* sk, int nsplit) {
  int o = 0;
  for (int q = 0; q < nsplit; q++) o += (sk[q] <= key) ? 1 : 0;
  return o;
}

// ------------------------------------------------------------------ X1
// samples k = 0 .. : the body at local index (k + 1/2) * g, g = n_total / kSampTotal — the same
// stride on every rank, so the merged samples weight every body equally.  send[0..5] = the rank's min / max
// (mm: folded by the previous step's integrate kernel, or by bbox_partial/final after an upload).
// Boundary candidates: every rank knows every rank's body count (nloc: written by the last migration of the previous
// step, identical everywhere), hence the index every boundary has in the global body order and the index it should
// have, (q + 1) n / P.  A boundary that has to move INTO this rank's range by d bodies goes to the position of this
// rank's d-th body from that end — the bodies are stored in the key order of the previous step's sort, which one
// integrate step disturbs only locally — so the rank on that side proposes it (w = 1), and dd_split_kernel takes the
// proposals when it decides to rebalance: exact quantiles, and only the bodies between the old and the new boundary
// change owner.
__global__ __launch_bounds__(256) void dd_x1_pack_kernel(float* __restrict__ send, const float* __restrict__ mm,
                                                         int n_loc, const float4* __restrict__ posm, double g,
                                                         int samp_cap, const int* __restrict__ nloc, int world,
                                                         int rank, long long n_total, const int* __restrict__ drift,
                                                         long long cap) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < 6) send[t] = mm[t];
  if (t == 0) {
    send[6] = __int_as_float(n_loc);
    send[7] = 0.0f;
    long long before = 0, total = 0;
    for (int q = 0; q < world; q++) {
      if (q < rank) before += nloc[q];
      total += nloc[q];
    }
    const bool known = total == n_total && nloc[rank] == n_loc;
    float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
    if (known && rank > 0) {
      // > 0: the lower boundary moves up into this range
      const long long d = dd_want(rank - 1, world, n_total, drift, drift[129], cap) - before;
      if (d > 0 && d < n_loc) {
        lo = posm[d];
        lo.w = 1.0f;
      }
    }
    if (known && rank < world - 1) {
      const long long d = before + n_loc - dd_want(rank, world, n_total, drift, drift[129], cap);  // > 0: the upper one moves down
      if (d > 0 && d < n_loc) {
        hi = posm[n_loc - d];
        hi.w = 1.0f;
      }
    }
    reinterpret_cast<float4*>(send + 8)[0] = lo;
    reinterpret_cast<float4*>(send + 8)[1] = hi;
  }
  if (t >= samp_cap) return;
  const long long idx = (long long)(((double)t + 0.5) * g);
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (idx < n_loc) {
    o = posm[idx];
    o.w = 1.0f;
  }
  reinterpret_cast<float4*>(send + kX1Samples0)[t] = o;
}

// The global cube from the gathered per-rank min / max (exact: min and max are associative; same arithmetic as
// write_cube of bh_tree.hip, ref:148-154), then this step's splitter keys.  The splitter KEYS persist: a rank owns a
// fixed interval of the curve, and only bodies whose key really left it change owner (round 3 re-drew the splitters
// from 2,048 position samples every step: the quantile noise alone moved 2-5 % of every rank's bodies per step).
// (Persisting splitter POSITIONS and re-keying them under every new cube measures the same emigrant counts at 8 x 1M,
// but a position that a coarse cell plane sweeps over jumps by octants along the curve and takes a rank's whole range
// with it: a 300-step soak ended in a body-capacity overflow that way.)  The keys move only when some rank's body count
// has left n / P by more than `tol`: then every boundary goes to the exact quantile its neighbours propose
// (dd_x1_pack_kernel), or — first step, or a proposal missing because a boundary would have to cross a whole rank —
// to the sample quantiles as in round 3.  Every rank runs this on the same gathered data and the same keys: same result.
__global__ __launch_bounds__(1024) void dd_split_kernel(const float* __restrict__ g, int world, int xf,
                                                        int samp_cap, float* __restrict__ bounds,
                                                        int curve, u64* __restrict__ skeys,
                                                        int* __restrict__ ddi, long long n_total, float tol,
                                                        int* __restrict__ drift, long long cap) {
  __shared__ int s_drift[64];  // the drift estimate the proposals of this step were made with (dd_x1_pack_kernel)
  __shared__ int s_est, s_prev_action;
  __shared__ u64 k[kSampTotal];
  __shared__ int nvalid;
  __shared__ float cube[8];
  __shared__ int s_mode;  // 0: keep, 1: the ranks' proposals, 2: sample quantiles
  __shared__ int s_need2;
  __shared__ float4 prop[64];
  const int tid = threadIdx.x;
  if (tid < 64) s_drift[tid] = drift[tid];
  if (tid == 0) {
    s_est = drift[129];
    s_prev_action = ddi[11];
  }
  __syncthreads();
  if (tid == 0) {
    // the drift estimate for the NEXT steps' proposals: the change of every rank's count over a step that kept the
    // boundaries (a step that moved them changed the counts by the rebalance itself), smoothed 3 : 1
    if (world > 1) {
      const bool sample = drift[128] && s_prev_action == 0;
      for (int r = 0; r < world; r++) {
        const int now = __float_as_int(g[(size_t)r * xf + 6]);
        if (sample) {
          const int dlt = now - drift[64 + r];
          drift[r] = s_est ? (3 * s_drift[r] + 16 * dlt) / 4 : 16 * dlt;
        }
        drift[64 + r] = now;
      }
      if (sample) drift[129] = 1;
      drift[128] = 1;
    }
    nvalid = 0;
    s_need2 = 0;
    float mn[3] = {1e10f, 1e10f, 1e10f}, mx[3] = {-1e10f, -1e10f, -1e10f};  // sentinels ref:138
    for (int r = 0; r < world; r++) {
      const float* o = g + (size_t)r * xf;
      mn[0] = fminf(mn[0], o[0]); mn[1] = fminf(mn[1], o[1]); mn[2] = fminf(mn[2], o[2]);
      mx[0] = fmaxf(mx[0], o[3]); mx[1] = fmaxf(mx[1], o[4]); mx[2] = fmaxf(mx[2], o[5]);
    }
    const float size = fmaxf(mx[0] - mn[0], fmaxf(mx[1] - mn[1], mx[2] - mn[2]));  // ref:148
    cube[0] = mn[0]; cube[1] = mn[1]; cube[2] = mn[2];
    cube[3] = mn[0] + size; cube[4] = mn[1] + size; cube[5] = mn[2] + size;  // ref:152-154
    cube[6] = fmaxf(cube[3] - cube[0], 1.0f);                                 // root edge s0, ref:55
    cube[7] = 0.0f;
    for (int q = 0; q < 8; q++) bounds[q] = cube[q];
    // what to do with the boundaries: keep them, unless some rank's count has left n / P by more than tol
    int mode = 0;
    if (world > 1) {
      if (!ddi[10]) {
        mode = 2;
      } else {
        const double fair = (double)n_total / (double)world;
        for (int r = 0; r < world; r++) {
          const double nr = (double)__float_as_int(g[(size_t)r * xf + 6]);
          if (fabs(nr - fair) > (double)tol * fair) mode = 1;
        }
        // far outside the band (proposals rejected step after step, or a step that scrambled the body order): the
        // proposals are bodies picked by their place in the previous key order and cannot be trusted to cross that much
        for (int r = 0; r < world; r++) {
          const double nr = (double)__float_as_int(g[(size_t)r * xf + 6]);
          if (fabs(nr - fair) > (double)kSplitHardBound * fair) mode = 2;
        }
      }
    }
    s_mode = mode;
  }
  __syncthreads();
  if (s_mode == 0) {
    if (tid == 0) {
      if (world > 1) ddi[10] = 1;
      ddi[11] = 0;
    }
    return;
  }
  // the keys of every rank's position samples under this step's cube: the sample quantiles (mode 2), and the yardstick
  // a proposed boundary is held against (mode 1)
  int mine = 0;
  for (int i = tid; i < kSampTotal; i += 1024) {
    u64 key = ~0ull;
    const int r = i / samp_cap, t = i - r * samp_cap;
    if (r < world) {
      const float4 p = reinterpret_cast<const float4*>(g + (size_t)r * xf + kX1Samples0)[t];
      if (p.w > 0.5f) {
        key = body_key<kB>(curve, p.x, p.y, p.z, cube[0], cube[1], cube[2], cube[6]);
        mine++;
      }
    }
    k[i] = key;
  }
  if (mine) atomicAdd(&nvalid, mine);
  __syncthreads();
  if (s_mode == 1 && tid < world - 1) {
    // Boundary q goes to the exact quantile: the rank into whose range it must move by d bodies proposed its d-th
    // body from that end (dd_x1_pack_kernel) — a body picked by its place in the PREVIOUS step's key order.  Under
    // the new cube its key must (a) still lie in the range of the rank that proposed it, and (b) stand among that
    // rank's position samples where its index says: about index / stride of them below it.  A body that one of the
    // cube's coarse planes has just passed jumps by octants along the curve; taking it hands a stretch of the curve
    // to the wrong rank.  (a) alone let a jump INSIDE the proposer's range through: a 1,000-step soak of 8 x 500k
    // overflowed a rank in its 35th rebalance, step 608.  With (b) a boundary can land at most a few sample strides
    // (a few per cent of a rank) from its quantile, far inside the capacity slack.  A boundary whose proposal fails
    // stays where it is for this step; the next step proposes another body.
    const int q = tid;
    long long before = 0;
    for (int r = 0; r <= q; r++) before += __float_as_int(g[(size_t)r * xf + 6]);
    const long long want = dd_want(q, world, n_total, s_drift, s_est, cap);
    const double stride = fmax(1.34 * (double)n_total / (double)kSampTotal, 1.0);  // (bh_dd_cube_pack)
    const float4 up = reinterpret_cast<const float4*>(g + (size_t)(q + 1) * xf + 8)[0];  // rank q+1, lower end
    const float4 dn = reinterpret_cast<const float4*>(g + (size_t)q * xf + 8)[1];        // rank q, upper end
    float4 take = make_float4(0.f, 0.f, 0.f, 0.f);  // (w = 0: this boundary stays)
    u64 kc = 0ull;
    int pr = -1;          // the proposing rank
    double index = 0.0;   // the proposed body's index among that rank's bodies
    if (want > before && up.w > 0.5f) {
      kc = body_key<kB>(curve, up.x, up.y, up.z, cube[0], cube[1], cube[2], cube[6]);
      if (kc >= skeys[q] && (q + 2 >= world || kc < skeys[q + 1])) {
        take = up;
        pr = q + 1;
        index = (double)(want - before);
      }
    } else if (want < before && dn.w > 0.5f) {
      kc = body_key<kB>(curve, dn.x, dn.y, dn.z, cube[0], cube[1], cube[2], cube[6]);
      if (kc < skeys[q] && (q == 0 || kc >= skeys[q - 1])) {
        take = dn;
        pr = q;
        index = (double)__float_as_int(g[(size_t)q * xf + 6]) - (double)(before - want);
      }
    } else if (want != before) {
      atomicOr(&s_need2, 1);  // a boundary would have to cross a whole rank: sample quantiles for all
    }
    if (pr >= 0) {
      int below = 0;
      for (int t = 0; t < samp_cap; t++) below += k[pr * samp_cap + t] < kc ? 1 : 0;
      const double expect = index / stride - 0.5;  // samples sit at (t + 0.5) stride
      if (fabs((double)below - expect) > 8.0 + expect / 32.0) take.w = 0.0f;
    }
    prop[q] = take;
  }
  __syncthreads();
  if (tid == 0) {
    const int mode = s_mode == 1 && !s_need2 ? 1 : 2;
    bool any = mode == 2;
    for (int q = 0; q < world - 1 && !any; q++) any = prop[q].w > 0.5f;
    if (any) ddi[9] += 1;  // steps in which a boundary really moved (every proposal may have been rejected)
    ddi[10] = 1;
    ddi[11] = any ? mode : 0;
    s_mode = mode;
  }
  __syncthreads();
  if (s_mode == 1) {  // the accepted proposals, keyed under this step's cube
    if (tid < world - 1 && prop[tid].w > 0.5f) {
      const float4 p = prop[tid];
      skeys[tid] = body_key<kB>(curve, p.x, p.y, p.z, cube[0], cube[1], cube[2], cube[6]);
    }
    return;
  }
  for (int size = 2; size <= kSampTotal; size <<= 1) {
    for (int j = size >> 1; j > 0; j >>= 1) {
      // kSampTotal / 2 compare-exchanges per stage, two per thread: pair p -> (i, i | j)
      for (int p = tid; p < kSampTotal / 2; p += 1024) {
        const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
        const int partner = i | j;
        const u64 a = k[i], b = k[partner];
        const bool up = (i & size) == 0;
        if ((a > b) == up) {
          k[i] = b;
          k[partner] = a;
        }
      }
      __syncthreads();
    }
  }
  const int M = nvalid;  // valid samples sort first (invalid = all ones)
  if (tid < world - 1) {
    if (M > 0) {
      skeys[tid] = k[(int)(((long long)(tid + 1) * M) / world)];
    } else {
      skeys[tid] = ~0ull;
    }
  }
}

// ------------------------------------------------------------------ X2
// Two kernels pack the emigrants (round 2: keys, classify, two scan launches, compact), one absorbs the
// immigrants (round 2: six launches).  All orders are by body index / (rank, slot): deterministic.
constexpr int kClsTile = 1024;  // bodies per block of the classify / compact kernels (256 threads x 4 rounds)

// Exclusive scan of nb per-block counts by ONE block of 256 threads (the block a bh_last_block hand-off elected):
// bbase[b] = counts of blocks before b; returns the total on every thread.  Every thread takes a run of consecutive
// counts — all loads in flight at once —, one shuffle scan per wave, the four wave sums through LDS (round 4: chunks
// of 256 through a Hillis-Steele scan in LDS, ~19 barriers per chunk, on the critical path of every rank-step).
__device__ __forceinline__ int last_block_scan(const int* bcnt, int* __restrict__ bbase, int nb) {
  __shared__ int wtot[4];
  constexpr int kRun = 8;
  int run_carry = 0;
  for (int c0 = 0; c0 < nb; c0 += 256 * kRun) {
    int v[kRun], sum = 0;
#pragma unroll
    for (int k = 0; k < kRun; k++) {
      const int b = c0 + (int)threadIdx.x * kRun + k;
      v[k] = b < nb ? bh_collect_i32(bcnt + b) : 0;
    }
#pragma unroll
    for (int k = 0; k < kRun; k++) sum += v[k];
    int inc = sum;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
      const int u = __shfl_up(inc, dd, 64);
      if (lane >= dd) inc += u;
    }
    if (lane == 63) wtot[wv] = inc;
    __syncthreads();
    int pre = run_carry + inc - sum;
    for (int w2 = 0; w2 < wv; w2++) pre += wtot[w2];
#pragma unroll
    for (int k = 0; k < kRun; k++) {
      const int b = c0 + (int)threadIdx.x * kRun + k;
      if (b < nb) bbase[b] = pre;
      pre += v[k];
    }
    run_carry += wtot[0] + wtot[1] + wtot[2] + wtot[3];
    __syncthreads();
  }
  return run_carry;
}

// kept[i] = 1 if body i stays (owner(key under the new cube) == me); bcnt[b] = emigrants of block b; the block
// that finishes last turns the counts into exclusive bases and writes the X2 header:
//   [0] emigrants found, [1] kept, [2] sent, [3] bodies still held  (+ 4 zero words)
__global__ __launch_bounds__(256) void dd_classify_kernel(const float4* __restrict__ posm, int n,
                                                          const float* __restrict__ bounds, int curve,
                                                          const u64* __restrict__ skeys, int nsplit, int me,
                                                          unsigned char* __restrict__ kept, int* __restrict__ bcnt,
                                                          int* __restrict__ bbase, u32* __restrict__ done,
                                                          int* __restrict__ header, int limit,
                                                          int* __restrict__ ddi12) {
  __shared__ u64 sk[64];
  __shared__ int wsum[4];
  __shared__ int s_last;
  __shared__ u32 htab[kHilbertTabWords];  // (bh_keys.h: the Hilbert keys from the state table)
  static_assert(kB == 21, "body_key21_fsm");
  if ((int)threadIdx.x < nsplit) sk[threadIdx.x] = skeys[threadIdx.x];
  hilbert_stage(htab);
  __syncthreads();
  const float b0 = bounds[0], b1 = bounds[1], b2 = bounds[2], size = bounds[6];
  int gone = 0;
#pragma unroll
  for (int r = 0; r < kClsTile / 256; r++) {
    const int i = blockIdx.x * kClsTile + r * 256 + (int)threadIdx.x;
    if (i < n) {
      const float4 p = posm[i];
      const u64 key = body_key21_fsm(htab, curve, p.x, p.y, p.z, b0, b1, b2, size);
      const int mine = owner_of(key, sk, nsplit) == me ? 1 : 0;
      kept[i] = (unsigned char)mine;
      gone += 1 - mine;
    }
  }
#pragma unroll
  for (int dd = 32; dd >= 1; dd >>= 1) gone += __shfl_xor(gone, dd, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = gone;
  __syncthreads();
  if (threadIdx.x == 0) {
    bh_publish_i32(bcnt + blockIdx.x, wsum[0] + wsum[1] + wsum[2] + wsum[3]);
    bh_published();  // (bh_internal.h: last-block hand-off without a fence)
    s_last = bh_last_block(done, (int)blockIdx.x, (int)gridDim.x) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  const int nb = (int)gridDim.x;
  const int run_carry = last_block_scan(bcnt, bbase, nb);
  if (threadIdx.x == 0) {
    const int found = run_carry;
    const int sent = min(found, limit);
    bbase[nb] = found;
    if (ddi12) *ddi12 = found;
    header[0] = found;
    header[1] = n - found;
    header[2] = sent;
    header[3] = n - sent;
    header[4] = header[5] = header[6] = header[7] = 0;
  }
}

// kept bodies -> the other ping-pong buffer (stable); the first `limit` emigrants -> X2 payload
// (header 2 x float4); emigrants beyond the limit stay behind the kept bodies and leave in a later
// round of the same step.
__global__ __launch_bounds__(256) void dd_compact_kernel(const float4* __restrict__ posm,
                                                         const float4* __restrict__ velid, int n,
                                                         const unsigned char* __restrict__ kept,
                                                         const int* __restrict__ bbase,
                                                         float4* __restrict__ posm2,
                                                         float4* __restrict__ velid2, float4* __restrict__ send,
                                                         int limit) {
  __shared__ int wcnt[kClsTile / 256][4];
  const int nb = (int)gridDim.x;
  const int total_gone = bbase[nb];
  const int n_kept = n - total_gone;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int f[kClsTile / 256];
  u64 bal[kClsTile / 256];
#pragma unroll
  for (int r = 0; r < kClsTile / 256; r++) {
    const int i = blockIdx.x * kClsTile + r * 256 + (int)threadIdx.x;
    f[r] = (i < n && !kept[i]) ? 1 : 0;  // emigrant
    bal[r] = __ballot(f[r]);
    if (lane == 0) wcnt[r][wv] = __popcll(bal[r]);
  }
  __syncthreads();
  int before = bbase[blockIdx.x];  // emigrants of earlier blocks
#pragma unroll
  for (int r = 0; r < kClsTile / 256; r++) {
    int pre = before;
    for (int w2 = 0; w2 < wv; w2++) pre += wcnt[r][w2];
    pre += __popcll(bal[r] & ((1ull << lane) - 1ull));
    const int i = blockIdx.x * kClsTile + r * 256 + (int)threadIdx.x;
    if (i < n) {
      const float4 p = posm[i], v = velid[i];
      if (!f[r]) {
        posm2[i - pre] = p;
        velid2[i - pre] = v;
      } else if (pre < limit) {
        send[2 + 2 * (size_t)pre] = p;
        send[3 + 2 * (size_t)pre] = v;
      } else {
        posm2[n_kept + pre - limit] = p;
        velid2[n_kept + pre - limit] = v;
      }
    }
    before += wcnt[r][0] + wcnt[r][1] + wcnt[r][2] + wcnt[r][3];
  }
}

// Absorbing the immigrants, two kernels.  Slot (q, k) = emigrant k of rank q's payload; block b covers 1024 slots
// of one rank.  dd_absorb_flag_kernel: owner of each emigrant under the current cube -> mine[slot], per-block
// count of mine, arrivals per owner (integer atomics: order-free); the block that finishes last turns the block
// counts into bases (arrival order = (rank, slot): deterministic), computes every rank's body count and hands the
// results to the device (out) and straight to pinned host memory, which the host polls (hres[4] = sequence number,
// written last with a system-scope release) instead of synchronising the stream.  dd_absorb_copy_kernel then
// appends the immigrants behind the bodies this rank still holds.
//   out / hres: [0] bodies this rank now holds, [1] flags, [2] most emigrants still waiting on any rank,
//               [3] most emigrants found on any rank (sizes the next exchange)
__global__ __launch_bounds__(256) void dd_absorb_flag_kernel(const float4* __restrict__ g, int world, int limit,
                                                             size_t f4, int cpr, const float* __restrict__ bounds,
                                                             int curve, const u64* __restrict__ skeys, int me,
                                                             int n_cap, unsigned char* __restrict__ mine_f,
                                                             int* __restrict__ bcnt, int* __restrict__ bbase,
                                                             int* __restrict__ arrive, u32* __restrict__ done,
                                                             bh_devinfo* __restrict__ info, int* __restrict__ nloc,
                                                             int* __restrict__ out, int* __restrict__ hres, int seq) {
  __shared__ u64 sk[64];
  __shared__ int cnt[64];
  __shared__ int wsum[4];
  __shared__ int s_last;
  __shared__ u32 htab[kHilbertTabWords];
  const int nsplit = world - 1;
  const int tid = threadIdx.x;
  if (tid < nsplit) sk[tid] = skeys[tid];
  if (tid < 64) cnt[tid] = 0;
  hilbert_stage(htab);
  __syncthreads();
  const int q = blockIdx.x / cpr, k0 = (blockIdx.x - q * cpr) * 1024;
  const int ne = min(reinterpret_cast<const int*>(g + (size_t)q * f4)[2], limit);
  const float b0 = bounds[0], b1 = bounds[1], b2 = bounds[2], size = bounds[6];
  int got = 0;
  if (k0 < ne) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int k = k0 + r * 256 + tid;
      int mine = 0;
      if (k < ne) {
        const float4 p = g[(size_t)q * f4 + 2 + 2 * (size_t)k];
        const u64 key = body_key21_fsm(htab, curve, p.x, p.y, p.z, b0, b1, b2, size);
        const int o = owner_of(key, sk, nsplit);
        atomicAdd(&cnt[o], 1);
        mine = (o == me && q != me) ? 1 : 0;
      }
      if (k < limit) mine_f[(size_t)q * limit + k] = (unsigned char)mine;
      got += mine;
    }
  }
#pragma unroll
  for (int dd = 32; dd >= 1; dd >>= 1) got += __shfl_xor(got, dd, 64);
  if ((tid & 63) == 0) wsum[tid >> 6] = got;
  __syncthreads();
  if (tid < world && cnt[tid]) atomicAdd(&arrive[tid], cnt[tid]);
  __syncthreads();
  if (tid == 0) {
    bh_publish_i32(bcnt + blockIdx.x, wsum[0] + wsum[1] + wsum[2] + wsum[3]);
    bh_published();  // also covers the arrive[] atomics above: issued by lanes of this same wave (tid < world <= 13)
    s_last = bh_last_block(done, (int)blockIdx.x, (int)gridDim.x) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  (void)last_block_scan(bcnt, bbase, (int)gridDim.x);
  if (tid < world) {
    const int held = reinterpret_cast<const int*>(g + (size_t)tid * f4)[3];
    const int a = __hip_atomic_load(arrive + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    cnt[tid] = held + a;
    nloc[tid] = held + a;
    __hip_atomic_store(arrive + tid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next call
  }
  __syncthreads();
  if (tid == 0) {
    int left = 0, most = 0;
    for (int r = 0; r < world; r++) {
      const int* h = reinterpret_cast<const int*>(g + (size_t)r * f4);
      left = max(left, h[0] - h[2]);
      most = max(most, h[0]);
    }
    if (cnt[me] > n_cap) atomicOr(&info->flags, BH_FLAG_DD_BODIES);  // the copy kernel drops what does not fit
    const int flags = __hip_atomic_load(&info->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out[0] = cnt[me]; out[1] = flags; out[2] = left; out[3] = most;
    hres[0] = cnt[me]; hres[1] = flags; hres[2] = left; hres[3] = most;
    __hip_atomic_store(hres + 4, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ __launch_bounds__(256) void dd_absorb_copy_kernel(const float4* __restrict__ g, int world, int limit,
                                                             size_t f4, int cpr, int me, int n_cap,
                                                             const unsigned char* __restrict__ mine_f,
                                                             const int* __restrict__ bbase,
                                                             float4* __restrict__ posm2,
                                                             float4* __restrict__ velid2) {
  __shared__ int wcnt[4][4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int q = blockIdx.x / cpr, k0 = (blockIdx.x - q * cpr) * 1024;
  const int ne = min(reinterpret_cast<const int*>(g + (size_t)q * f4)[2], limit);
  if (k0 >= ne) return;
  const int held = reinterpret_cast<const int*>(g + (size_t)me * f4)[3];
  int f[4];
  u64 bal[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int k = k0 + r * 256 + tid;
    f[r] = (k < ne && mine_f[(size_t)q * limit + k]) ? 1 : 0;
    bal[r] = __ballot(f[r]);
    if (lane == 0) wcnt[r][wv] = __popcll(bal[r]);
  }
  __syncthreads();
  int before = held + bbase[blockIdx.x];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    int pre = before;
    for (int w2 = 0; w2 < wv; w2++) pre += wcnt[r][w2];
    pre += __popcll(bal[r] & ((1ull << lane) - 1ull));
    const int k = k0 + r * 256 + tid;
    if (f[r] && pre < n_cap) {
      posm2[pre] = g[(size_t)q * f4 + 2 + 2 * (size_t)k];
      velid2[pre] = g[(size_t)q * f4 + 3 + 2 * (size_t)k];
    }
    before += wcnt[r][0] + wcnt[r][1] + wcnt[r][2] + wcnt[r][3];
  }
}

// ------------------------------------------------------------------ X3: pieces
// (the local tree's COM stage writes only the force kernel's digests, as in bh_step: a record's body range is still
// in the bit patterns of its x / y, where the build left it — put_rec, bh_tree.hip)
__device__ __forceinline__ int rec_lo(const bh_node& r) { return __float_as_int(r.x); }
__device__ __forceinline__ int rec_hi(const bh_node& r) { return __float_as_int(r.y); }

__global__ __launch_bounds__(BH_DD_PIECE_CAP) void dd_describe_kernel(
    const int* __restrict__ piece_tmp, const int* ddi, int* ddi_w, const bh_node* __restrict__ rec,
    const u64* __restrict__ keys,
    const bh_d4* __restrict__ P, const float4* __restrict__ posm, const float* __restrict__ bounds, int curve,
    int me, int n_loc, bh_dd_piece* __restrict__ out, int* __restrict__ piece_idx,
    bh_devinfo* __restrict__ info) {
  __shared__ int lo[BH_DD_PIECE_CAP], idx[BH_DD_PIECE_CAP];
  const int found = ddi[0];
  const int np = min(found, BH_DD_PIECE_CAP);
  const int t = threadIdx.x;
  if (t == 0) {
    if (found > BH_DD_PIECE_CAP) atomicOr(&info->flags, BH_FLAG_DD_PIECES);
    int* h = reinterpret_cast<int*>(out);
    for (int i = 0; i < 20; i++) h[i] = 0;
    h[0] = np;
    h[1] = n_loc;
  }
  if (t < np) {
    idx[t] = piece_tmp[t];
    lo[t] = rec_lo(rec[idx[t]]);
  }
  __syncthreads();
  if (t == 0) {
    ddi_w[8] = np;  // for the export kernel
    ddi_w[0] = 0;   // every thread has read the count: cleared for the next step's spine kernel
  }
  if (t >= np) return;
  int rank = 0;
  for (int u = 0; u < np; u++) rank += (lo[u] < lo[t]) ? 1 : 0;  // pieces are disjoint: distinct starts
  const int e = idx[t];
  piece_idx[rank] = e;
  const int a = lo[t], b = rec_hi(rec[e]);
  bh_dd_piece d;
  d.key = keys[a];
  d.rec_idx = e;
  d.owner = me;
  d.count = b - a;
  d.kind = rec[e].kind;
  d.pad[0] = d.pad[1] = 0;
  const float size = bounds[6];
  if (b - a == 1) {
    const float4 q = posm[a];
    const double m = (double)q.w;
    d.sm = m; d.sx = m * (double)q.x; d.sy = m * (double)q.y; d.sz = m * (double)q.z;
    d.bx = q.x; d.by = q.y; d.bz = q.z; d.bs = 0.0f;
  } else {
    const bh_d4 p1 = P[b], p0 = P[a];
    d.sm = p1.m - p0.m; d.sx = p1.x - p0.x; d.sy = p1.y - p0.y; d.sz = p1.z - p0.z;
    // box of the compressed cell: the Lb leading digits its bodies share
    const int Lb = common_digits(keys[a], keys[b - 1], kB);
    // cell coordinates of any body of the cell (the key de-interleaved, and for the Hilbert order run back
    // through the curve), low 21 - Lb bits cleared = the cell's minimum corner
    const u64 ka = keys[a];
    u32 ix = compact_bits21(ka >> 2), iy = compact_bits21(ka >> 1), iz = compact_bits21(ka);
    if (curve == 1) hilbert_transpose_to_axes(ix, iy, iz);
    const u32 keep = (Lb <= 0) ? 0u : ~((1u << (kB - Lb)) - 1u);
    ix &= keep; iy &= keep; iz &= keep;
    d.bx = bounds[0] + (float)ix / 2097152.0f * size;
    d.by = bounds[1] + (float)iy / 2097152.0f * size;
    d.bz = bounds[2] + (float)iz / 2097152.0f * size;
    d.bs = ldexpf(size, -Lb);
  }
  out[1 + rank] = d;
}

// ------------------------------------------------------------------ X4: LET
// piece boxes of the other ranks, rank by rank, grown by a margin that covers the fp32 fuzz of the
// key quantisation (<= 0.25 key units) and of the box arithmetic; plus each rank's bounding box.
// One block per rank.  rbox[2q], rbox[2q+1] = min / max corner (w of the min corner = first box,
// w of the max corner = end box of that rank).
__global__ __launch_bounds__(256) void dd_boxes_kernel(const bh_dd_piece* __restrict__ g, int world, int me,
                                                       const float* __restrict__ bounds,
                                                       float4* __restrict__ boxes, float4* __restrict__ rbox,
                                                       int* __restrict__ ddi) {
  __shared__ float red[4][6];
  __shared__ int s_off, s_np;
  const int q = blockIdx.x;
  if (threadIdx.x == 0) {
    int off = 0;
    for (int r = 0; r < q; r++)
      if (r != me) off += min(reinterpret_cast<const int*>(g + (size_t)r * kDescPerRank)[0], BH_DD_PIECE_CAP);
    s_off = off;
    s_np = (q == me) ? 0 : min(reinterpret_cast<const int*>(g + (size_t)q * kDescPerRank)[0], BH_DD_PIECE_CAP);
  }
  __syncthreads();
  const int off = s_off, np = s_np;
  const float mg = 1e-5f * bounds[6];
  float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
  for (int k = threadIdx.x; k < np; k += 256) {
    const bh_dd_piece d = g[(size_t)q * kDescPerRank + 1 + k];
    const float4 b = make_float4(d.bx - mg, d.by - mg, d.bz - mg, d.bs + 2.0f * mg);
    boxes[off + k] = b;
    mn[0] = fminf(mn[0], b.x); mn[1] = fminf(mn[1], b.y); mn[2] = fminf(mn[2], b.z);
    mx[0] = fmaxf(mx[0], b.x + b.w); mx[1] = fmaxf(mx[1], b.y + b.w); mx[2] = fmaxf(mx[2], b.z + b.w);
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], dd, 64));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], dd, 64));
    }
    if (lane == 0) {
      red[wv][a] = mn[a];
      red[wv][3 + a] = mx[a];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int u = 1; u < 4; u++)
      for (int a = 0; a < 3; a++) {
        red[0][a] = fminf(red[0][a], red[u][a]);
        red[0][3 + a] = fmaxf(red[0][3 + a], red[u][3 + a]);
      }
    rbox[2 * q] = make_float4(red[0][0], red[0][1], red[0][2], __int_as_float(off));
    rbox[2 * q + 1] = make_float4(red[0][3], red[0][4], red[0][5], __int_as_float(off + np));
    if (q == world - 1) ddi[1] = off + np;
  }
}

// w[e] = records cell e must export = its child count when some remote body could open it:
// min over the remote boxes of |com - box|^2 + eps2 <= (s/theta)^2, with slack on both sides.
// A cell with (s/theta)^2 < eps2 is accepted at any distance, so it is never a candidate.  A block takes chunks of 512
// records: every thread tests its records against the ranks' bounding boxes; the cells that are near some rank are
// compacted in LDS and tested against those ranks' piece boxes by eight lanes each.
constexpr int kMarkChunk = 512;
#ifndef BH_MARK_BOXES
#define BH_MARK_BOXES 512  // (a test build with 64 drives the memory path: tools/mkvariant.sh markboxes64 -DBH_MARK_BOXES=64)
#endif
constexpr int kMarkBoxes = BH_MARK_BOXES;  // remote boxes staged in LDS (more are read from memory)
constexpr int kMarkPer = kMarkChunk / 256;
// blocks per compute unit.  (Five — 27 KB of LDS each, five fit — and a prefetch of a block's next chunk looked like
// -10 us on a core rank's launch in the instrumented build, tools/dd_mark_trace.py; under rocprofv3 the product
// build ran 72.5 us against 68.3 with four and no prefetch, tools/mark_ab.sh.  Handing the chunks out through an atomic
// counter instead of round-robin: +8 us.)
#ifndef BH_MARK_GRID
#define BH_MARK_GRID 4
#endif
constexpr int kMarkGrid = BH_MARK_GRID;
// no point of the box [lo, hi] can open the candidate q = (com, threshold): |com - box|^2 + eps2 > thr2, with slack
__device__ __forceinline__ bool box_too_far(const float4 lo, const float4 hi, const float4 q, float eps2) {
  const float dx = fmaxf(fmaxf(lo.x - q.x, q.x - hi.x), 0.0f);
  const float dy = fmaxf(fmaxf(lo.y - q.y, q.y - hi.y), 0.0f);
  const float dz = fmaxf(fmaxf(lo.z - q.z, q.z - hi.z), 0.0f);
  return (dx * dx + dy * dy + dz * dz) * 0.9999f + eps2 > q.w;
}
#ifdef BH_DD_TRACE
// design-study instrumentation (tools/dd_mark_trace.py): 100 MHz stamps / sums per block of dd_mark_kernel
__device__ unsigned long long g_dd_trace[2048][8];
extern "C" int bh_debug_dd_trace(void* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dd_trace), sizeof(g_dd_trace));
}
#define DT_SET(k, v) if (threadIdx.x == 0 && blockIdx.x < 2048) g_dd_trace[blockIdx.x][k] = (v);
#define DT_NOW() __builtin_amdgcn_s_memrealtime()
#else
#define DT_SET(k, v)
#define DT_NOW() 0ull
#endif
// Round 4 (tools/dd_mark_trace.py): one block per 1,024 records of the POOL's capacity, one thread per candidate
// cell, was 94 us per rank-step at 8 x 1M bodies and 66 us at 8 x 125,000 — five rounds of blocks at 51 KB of LDS,
// most of them staging the boxes for chunks past the tree's last record; in the others a quarter of the records were
// candidates, most of them near no rank at all, and the few near several ranks ran hundreds of dependent LDS reads in
// one lane while the rest of their wave idled; the block that finished last then scanned 3,800 chunk counts with
// sixteen barriers per 256.  Now: blocks loop over the chunks that hold records, the rank-box test happens before a
// cell becomes a candidate, eight lanes share a candidate's piece tests (four pieces per lane and step), and the final
// scan covers the chunks with records, serially per thread plus one block-wide step.
__global__ __launch_bounds__(256) void dd_mark_kernel(const bh_frec* __restrict__ frec,
                                                      const bh_devinfo* __restrict__ info, int rec_cap,
                                                      const float4* __restrict__ boxes,
                                                      const float4* __restrict__ rbox, int world,
                                                      const int* __restrict__ ddi, float eps2,
                                                      int* __restrict__ w, unsigned* __restrict__ wmask,
                                                      int* __restrict__ lpos, int* __restrict__ bcnt,
                                                      int* __restrict__ bbase, u32* __restrict__ done, int nchunks) {
  __shared__ float4 sb[kMarkBoxes];
  __shared__ float4 clo[kMarkBoxes / 8], chi[kMarkBoxes / 8];  // bounding boxes of 8 consecutive piece boxes each
  __shared__ float4 srb[128];
  __shared__ float4 cxyz[kMarkChunk];     // candidate: com, threshold
  __shared__ int cidx[kMarkChunk];        // candidate: record index
  __shared__ int ccnt[kMarkChunk];        // candidate: child count
  __shared__ unsigned cnear[kMarkChunk];  // candidate: ranks whose bounding box is near enough
  __shared__ unsigned char oflag[kMarkChunk];  // per-destination mode: record of this chunk exports its children
  __shared__ int ncand;
  __shared__ int wsum2[4];
  __shared__ int s_last;
  const int E = min(info->n_entries, rec_cap);
  const int NB = ddi[1];
  // per-destination mode reads w / wmask only below E (dd_let_list_kernel); the union mode scans w up to rec_cap
  const int used = wmask ? min(nchunks, (E + kMarkChunk - 1) / kMarkChunk) : nchunks;
  for (int i = threadIdx.x; i < 2 * world; i += 256) srb[i] = rbox[i];
  for (int i = threadIdx.x; i < min(NB, kMarkBoxes); i += 256) sb[i] = boxes[i];
  __syncthreads();
  for (int cl = threadIdx.x; 8 * cl < min(NB, kMarkBoxes); cl += 256) {
    float4 lo = make_float4(1e30f, 1e30f, 1e30f, 0.f), hi = make_float4(-1e30f, -1e30f, -1e30f, 0.f);
    for (int i = 8 * cl; i < min(min(NB, kMarkBoxes), 8 * cl + 8); i++) {
      const float4 b = sb[i];
      lo.x = fminf(lo.x, b.x); lo.y = fminf(lo.y, b.y); lo.z = fminf(lo.z, b.z);
      hi.x = fmaxf(hi.x, b.x + b.w); hi.y = fmaxf(hi.y, b.y + b.w); hi.z = fmaxf(hi.z, b.z + b.w);
    }
    clo[cl] = lo;
    chi[cl] = hi;
  }
  DT_SET(0, DT_NOW())
  unsigned long long dt_a = 0, dt_b = 0, dt_c = 0, dt_n = 0, dt_t = 0;
  (void)dt_a; (void)dt_b; (void)dt_c; (void)dt_n; (void)dt_t;
  for (int chunk = blockIdx.x; chunk < used; chunk += gridDim.x) {
    const int e0 = chunk * kMarkChunk;
    if (e0 >= E) {  // block-uniform: no record here (union mode only)
      for (int i = threadIdx.x; i < kMarkChunk && e0 + i <= rec_cap; i += 256) w[e0 + i] = 0;
      continue;
    }
    __syncthreads();  // (the previous chunk's candidate arrays are free; first chunk: the boxes are staged)
    dt_t = DT_NOW();
    if (threadIdx.x == 0) ncand = 0;
    for (int i = threadIdx.x; i < kMarkChunk; i += 256) oflag[i] = 0;
    __syncthreads();
    {
      float4 q[kMarkPer];
      int meta[kMarkPer];
#pragma unroll
      for (int u = 0; u < kMarkPer; u++) {  // loads first
        const int e = e0 + u * 256 + (int)threadIdx.x;
        q[u] = make_float4(0.f, 0.f, 0.f, -1.0f);
        meta[u] = 0;
        if (e < E) {
          const float* f = reinterpret_cast<const float*>(frec) + (size_t)(e >> 1) * 16 + (e & 1);
          q[u] = make_float4(f[0], f[2], f[4], f[8] * 1.0001f);
          meta[u] = reinterpret_cast<const int*>(f)[12];
        }
      }
#pragma unroll
      for (int u = 0; u < kMarkPer; u++) {
        const int e = e0 + u * 256 + (int)threadIdx.x;
        if (e > rec_cap) break;
        unsigned near = 0u;
        if (e < E && q[u].w >= eps2)  // implies thr2 >= 0: an openable, massive cell
          for (int r = 0; r < world; r++) {
            const float4 lo = srb[2 * r], hi = srb[2 * r + 1];
            if (__float_as_int(hi.w) > __float_as_int(lo.w) && !box_too_far(lo, hi, q[u], eps2)) near |= 1u << r;
          }
        if (near) {
          const int k = atomicAdd(&ncand, 1);
          cxyz[k] = q[u];
          cidx[k] = e;
          ccnt[k] = meta[u] & 0x7fffffff;
          cnear[k] = near;
        } else {
          w[e] = 0;
          if (wmask) wmask[e] = 0u;
        }
      }
    }
    __syncthreads();
    dt_a += DT_NOW() - dt_t; dt_t = DT_NOW(); dt_n += (unsigned long long)ncand;
    // eight lanes per candidate (every branch below is uniform over a team: its lanes stay together)
    const int nc = ncand;
    const int team = threadIdx.x >> 3, tl = threadIdx.x & 7, tshift = (threadIdx.x & 63) & ~7;
    for (int c0 = 0; c0 < nc; c0 += 32) {
      const int c = c0 + team;
      if (c >= nc) continue;
      const float4 q = cxyz[c];
      unsigned mask = 0u;
      for (unsigned rm = cnear[c]; rm; rm &= rm - 1) {
        const int r = __ffs(rm) - 1;
        const int b0 = __float_as_int(srb[2 * r].w), b1 = __float_as_int(srb[2 * r + 1].w);
        bool hit = false;
        // clusters of eight consecutive boxes first (the pieces are in curve order: a cluster is compact), one per
        // lane and step; then the eight pieces of a cluster that is near, one per lane
        for (int k0 = b0 >> 3; k0 <= (b1 - 1) >> 3 && !hit; k0 += 8) {
          const int k = k0 + tl;
          bool nearc = false;
          if (k <= (b1 - 1) >> 3) nearc = 8 * k + 8 > kMarkBoxes || !box_too_far(clo[k], chi[k], q, eps2);
          for (unsigned cm = (unsigned)(__ballot(nearc) >> tshift) & 0xffu; cm && !hit; cm &= cm - 1) {
            const int i = 8 * (k0 + __ffs(cm) - 1) + tl;
            bool h = false;
            if (i >= b0 && i < b1) {  // (a cluster may reach into the neighbouring ranks' boxes)
              const float4 bx = i < kMarkBoxes ? sb[i] : boxes[i];
              h = !box_too_far(bx, make_float4(bx.x + bx.w, bx.y + bx.w, bx.z + bx.w, 0.f), q, eps2);
            }
            hit = ((unsigned)(__ballot(h) >> tshift) & 0xffu) != 0u;
          }
        }
        if (hit) mask |= 1u << r;
        if (hit && !wmask) break;  // union mode: the first rank that opens settles it
      }
      if (tl == 0) {
        const bool open = mask != 0u;
        w[cidx[c]] = open ? ccnt[c] : 0;
        if (wmask) {
          wmask[cidx[c]] = mask;
          if (open && ccnt[c] > 0) oflag[cidx[c] - e0] = 1;
        }
      }
    }
    __syncthreads();
    dt_b += DT_NOW() - dt_t; dt_t = DT_NOW();
    if (!wmask) continue;
    // per-destination mode: the exporting cells are compacted here (record order inside the chunk, chunks in order:
    // deterministic) instead of by a separate flag scan over the whole record pool.  lpos[e] = position inside the
    // chunk; the block that finishes last turns the chunk counts into bases (bbase, bbase[nchunks] = list length);
    // dd_let_list_kernel adds them up.
    {
      static_assert(kMarkPer == 2, "two flags per thread");
      const int t2 = 2 * (int)threadIdx.x;
      const int f0 = oflag[t2], f1 = oflag[t2 + 1];
      const int s2 = f0 + f1;
      int inc = s2;
      const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
      for (int dd = 1; dd < 64; dd <<= 1) {
        const int u = __shfl_up(inc, dd, 64);
        if (lane >= dd) inc += u;
      }
      if (lane == 63) wsum2[wv] = inc;
      __syncthreads();
      int pre = inc - s2;
      for (int k = 0; k < wv; k++) pre += wsum2[k];
      if (f0) lpos[e0 + t2] = pre;
      if (f1) lpos[e0 + t2 + 1] = pre + f0;
    }
    if (threadIdx.x == 0) bh_publish_i32(bcnt + chunk, wsum2[0] + wsum2[1] + wsum2[2] + wsum2[3]);
    dt_c += DT_NOW() - dt_t;
  }
  DT_SET(1, dt_a) DT_SET(2, dt_b) DT_SET(3, dt_c) DT_SET(4, dt_n) DT_SET(5, DT_NOW())
  if (!wmask) return;
  if (threadIdx.x == 0) {
    bh_published();
    s_last = bh_last_block(done, (int)blockIdx.x, (int)gridDim.x) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  // bases of the chunks that hold records: thread t adds up `per` consecutive counts, one block-wide exclusive scan
  // of the 256 sums, then every thread writes its chunks' bases
  const int per = (used + 255) / 256;
  const int c0 = (int)threadIdx.x * per;
  int sum = 0;
  for (int k = 0; k < per; k++)
    if (c0 + k < used) sum += bh_collect_i32(bcnt + c0 + k);
  int inc = sum;
  {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
      const int u = __shfl_up(inc, dd, 64);
      if (lane >= dd) inc += u;
    }
    __syncthreads();  // (wsum2 was last read before the bh_last_block barrier; kept for clarity)
    if (lane == 63) wsum2[wv] = inc;
    __syncthreads();
    for (int k = 0; k < wv; k++) inc += wsum2[k];
  }
  int run = inc - sum;
  for (int k = 0; k < per; k++)
    if (c0 + k < used) {
      bbase[c0 + k] = run;
      run += bh_collect_i32(bcnt + c0 + k);
    }
  if (threadIdx.x == 255) bbase[nchunks] = inc;  // list length
  DT_SET(6, DT_NOW())
}

__device__ __forceinline__ bh_frec reloc(bh_frec fr, int c, const int* __restrict__ w,
                                         const int* __restrict__ dst, int blocks0, int rec_cap) {
  if (c >= rec_cap) return fr;  // a body digest (child of an unsplit multi-body cell): no children
  const int wc = w[c];
  if (wc > 0) {
    fr.first = blocks0 + dst[c];
    fr.meta = wc;
  } else if (fr.thr2 >= 0.0f) {
    // no body of another rank can open this cell (conservative test): every one of them accepts it, so the
    // exported copy is closed outright — same forces, and the receiver never holds an openable record
    // whose child block it was not given
    fr.first = 0;
    fr.thr2 = -1.0f;
  }
  return fr;
}

// send[0] header, send[1 .. PIECE_CAP] the pieces' own records, one padding record, then the child blocks
// in scan order, each at an even record (dst = scan of the counts rounded up to even), an odd block
// followed by a null digest.  Child indices are pool indices of the receiving side: seg0 = pool index of
// this rank's segment (even).
__global__ __launch_bounds__(256) void dd_export_kernel(const bh_frec* __restrict__ frec,
                                                        int rec_cap,
                                                        const int* __restrict__ w, const int* __restrict__ dst,
                                                        const int* __restrict__ piece_idx,
                                                        const int* __restrict__ ddi, int seg0, int stride,
                                                        int world, bh_frec* __restrict__ send) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int blocks0 = seg0 + kSegBlocks0;
  const int np = ddi[8];
  if (e == 0) {
    bh_frec h = frec_null();
    h.thr2 = 0.0f;
    h.first = kSegBlocks0 + dst[rec_cap];  // records this rank needs (may exceed stride)
    h.meta = np;
    frec_put(send, 0, h);
    for (int j = 0; j < 24; j++)             // the union goes to every receiver: one need for all of them
      reinterpret_cast<int*>(send)[seg_row_dword(j)] = j < world ? h.first : 0;
  }
  // a segment that does not fit is sent closed: its pieces are made unopenable, so the (discarded)
  // force pass that runs before the host sees the header never walks unwritten records
  const bool fits = kSegBlocks0 + dst[rec_cap] <= stride;
  if (e < BH_DD_PIECE_CAP) {
    bh_frec fr = frec_null();
    if (e < np) fr = reloc(frec_get(frec, piece_idx[e]), piece_idx[e], w, dst, blocks0, rec_cap);
    if (!fits) fr.thr2 = -1.0f;
    frec_put(send, kSegPieces0 + e, fr);
  }
  if (e >= rec_cap || !fits) return;
  const int wv = w[e];
  if (wv == 0) return;
  const int off = kSegBlocks0 + dst[e];
  if (off + wv > stride) return;  // does not fit: the header tells the host to repeat with more room
  const bh_frec fr = frec_get(frec, e);
  // children of a cell, or the body digests of an unsplit multi-body cell: one kind of block
  for (int k = 0; k < wv; k++)
    frec_put(send, off + k, reloc(frec_get(frec, fr.first + k), fr.first + k, w, dst, blocks0, rec_cap));
  if ((wv & 1) && off + wv < stride) frec_put(send, off + wv, frec_null());
}

// ------------------------------------------------------------------ X4, per-destination segments (let_mode 1)
// The union above goes to every rank although a receiver only opens what ITS boxes can reach (about a third of
// it at 8 ranks).  Per destination: dd_mark_kernel records WHICH ranks may open a cell (wmask); the exporting cells
// are compacted into a list (lpos = exclusive scan of w > 0); one block per destination scans the list entries
// that carry its bit (dstd[q][i] = offset of cell i's child block inside the segment for q); the export kernel
// writes every exporting cell's block once per interested destination, child pointers relocated with THAT
// destination's offsets — a child cell this destination cannot open is closed in its copy.  The `world` segments
// (stride records each) are exchanged with an all-to-all; a receiver finds the sender's segment where the
// all-gather put it, so the top tree, the validation and the walk are unchanged.
__global__ __launch_bounds__(256) void dd_let_list_kernel(const int* __restrict__ w,
                                                          const unsigned* __restrict__ wmask,
                                                          int* __restrict__ lpos, const int* __restrict__ bbase,
                                                          int nblocks, int rec_cap, int lcap,
                                                          int* __restrict__ list_e, int* __restrict__ list_w,
                                                          unsigned* __restrict__ list_m,
                                                          const bh_devinfo* __restrict__ info) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e == 0) lpos[rec_cap] = bbase[nblocks];  // list length (read by the kernels that follow)
  if (e >= min(info->n_entries, rec_cap) || w[e] <= 0) return;  // (dd_mark_kernel wrote w below the last record only)
  const int i = bbase[e / kMarkChunk] + lpos[e];  // block base + position inside dd_mark_kernel's block
  lpos[e] = i;                                    // from here on: the cell's position in the list
  if (i >= lcap) return;
  list_e[i] = e;
  list_w[i] = (w[e] + 1) & ~1;  // blocks start at even records
  list_m[i] = wmask[e];
}

// dstd[q][i] = exclusive prefix, over the list, of the (even) block sizes of the entries that carry bit q.
// Two launches over (chunk of 8192 entries, destination): chunk sums, then every block adds up the sums of the
// chunks before it (a few dozen) and scans its own chunk.  (One block per destination looping over the list took
// 67-87 us per rank-step at 8 x 1M: each round is a dependent load -> scan -> store chain.)
constexpr int kLetChunk = 8192;
__global__ __launch_bounds__(1024) void dd_let_sums_kernel(const int* __restrict__ list_w,
                                                           const unsigned* __restrict__ list_m,
                                                           const int* __restrict__ lpos, int rec_cap, int lcap,
                                                           int nch, int* __restrict__ csum) {
  __shared__ int wsum[16];
  const int q = blockIdx.y, tid = threadIdx.x;
  const int L = min(lpos[rec_cap], lcap);
  // (the grid is a few blocks per destination, not one per chunk of the list's CAPACITY: the list length is only
  // known on the device, and 160 chunks x 8 destinations of 1024-thread blocks that mostly found nothing to do were
  // most of these two kernels' 25 us)
  for (int j = blockIdx.x; j < nch; j += gridDim.x) {
    const int i0 = j * kLetChunk;
    if (i0 >= L) {
      if (tid == 0) csum[q * nch + j] = 0;
      continue;
    }
    int s = 0;
#pragma unroll
    for (int k = 0; k < kLetChunk / 1024; k++) {
      const int i = i0 + k * 1024 + tid;
      if (i < L && ((list_m[i] >> q) & 1u)) s += list_w[i];
    }
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) s += __shfl_xor(s, dd, 64);
    __syncthreads();  // (wsum of the previous chunk has been read)
    if ((tid & 63) == 0) wsum[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int k = 0; k < 16; k++) t += wsum[k];
      csum[q * nch + j] = t;
    }
  }
}

__global__ __launch_bounds__(1024) void dd_let_scan_kernel(const int* __restrict__ list_w,
                                                           const unsigned* __restrict__ list_m,
                                                           const int* __restrict__ lpos, int rec_cap, int lcap,
                                                           int nch, const int* __restrict__ csum,
                                                           int* __restrict__ dstd, int* __restrict__ dtot) {
  __shared__ int wsum[16];
  __shared__ int s_base;
  constexpr int kPer = kLetChunk / 1024;
  const int q = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int L = min(lpos[rec_cap], lcap);
  for (int j = blockIdx.x; j < nch; j += gridDim.x) {  // (see dd_let_sums_kernel)
    const int i0 = j * kLetChunk;
    if (i0 >= L && j > 0) break;  // (chunk 0 always runs: it writes the total of an empty list)
    __syncthreads();              // (s_base / wsum of the previous chunk have been read)
    if (tid < 64) {               // base = chunk sums before this one
      int b = 0;
      for (int k = tid; k < j; k += 64) b += csum[q * nch + k];
#pragma unroll
      for (int dd = 32; dd >= 1; dd >>= 1) b += __shfl_xor(b, dd, 64);
      if (tid == 0) s_base = b;
    }
    int x[kPer], s = 0;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const int i = i0 + kPer * tid + k;
      x[k] = (i < L && ((list_m[i] >> q) & 1u)) ? list_w[i] : 0;
      s += x[k];
    }
    int inc = s;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
      const int u = __shfl_up(inc, dd, 64);
      if (lane >= dd) inc += u;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int pre = s_base;
    for (int k = 0; k < wv; k++) pre += wsum[k];
    int run = pre + inc - s;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const int i = i0 + kPer * tid + k;
      if (i < L) dstd[(size_t)q * lcap + i] = run;
      run += x[k];
    }
    // the block of the list's last chunk (or chunk 0 of an empty list) knows the destination's total
    if (tid == 1023 && (i0 + kLetChunk >= L)) dtot[q] = pre + inc;
  }
}

// the copy of record `fr` (pool index c on the sender) that goes to destination q
__device__ __forceinline__ bh_frec reloc_pd(bh_frec fr, int c, const int* __restrict__ w,
                                            const unsigned* __restrict__ wmask, const int* __restrict__ lpos,
                                            const int* __restrict__ dq /* dstd + q * lcap */, int lcap, int blocks0,
                                            int rec_cap, int q) {
  if (c >= rec_cap) return fr;  // a body digest (child of an unsplit multi-body cell): no children
  const int wc = w[c];
  const int i = wc > 0 ? lpos[c] : lcap;
  if (wc > 0 && i < lcap && ((wmask[c] >> q) & 1u)) {
    fr.first = blocks0 + dq[i];
    fr.meta = wc;
  } else if (fr.thr2 >= 0.0f) {  // no body of rank q can open this cell: its copy there is closed
    fr.first = 0;
    fr.thr2 = -1.0f;
  }
  return fr;
}

// header, needs row and pieces of every destination's segment: one block per destination
__global__ __launch_bounds__(BH_DD_PIECE_CAP) void dd_export_pd_head_kernel(
    const bh_frec* __restrict__ frec, int rec_cap, const int* __restrict__ w, const unsigned* __restrict__ wmask,
    const int* __restrict__ lpos, const int* __restrict__ dstd, const int* __restrict__ dtot, int lcap,
    const int* __restrict__ piece_idx, const int* __restrict__ ddi, int seg0, int stride, int world, int me,
    bh_frec* __restrict__ send) {
  const int q = blockIdx.x, t = threadIdx.x;
  bh_frec* seg = send + (size_t)q * stride;
  const int blocks0 = seg0 + kSegBlocks0;
  const int np = ddi[8];
  const bool over = lpos[rec_cap] > lcap;  // more exporting cells than the list holds: nothing fits
  const int need = (q == me) ? kSegBlocks0 : kSegBlocks0 + (over ? stride : dtot[q]);
  const bool fits = need <= stride && q != me;
  if (t == 0) {
    bh_frec h = frec_null();
    h.thr2 = 0.0f;
    h.first = need;
    h.meta = np;
    frec_put(seg, 0, h);
  }
  if (t < 24) {
    int v = 0;
    if (t < world && t != me) v = kSegBlocks0 + (over ? stride : dtot[t]);
    reinterpret_cast<int*>(seg)[seg_row_dword(t)] = v;
  }
  bh_frec fr = frec_null();
  if (t < np && q != me)
    fr = reloc_pd(frec_get(frec, piece_idx[t]), piece_idx[t], w, wmask, lpos, dstd + (size_t)q * lcap, lcap, blocks0,
                  rec_cap, q);
  // a segment that does not fit is sent closed: its pieces are made unopenable, so the (discarded) force pass
  // that runs before the host sees the header never walks unwritten records
  if (!fits) fr.thr2 = -1.0f;
  frec_put(seg, kSegPieces0 + t, fr);
}

__global__ __launch_bounds__(256) void dd_export_pd_kernel(const bh_frec* __restrict__ frec, int rec_cap,
                                                           const int* __restrict__ w,
                                                           const unsigned* __restrict__ wmask,
                                                           const int* __restrict__ lpos,
                                                           const int* __restrict__ list_e,
                                                           const int* __restrict__ dstd,
                                                           const int* __restrict__ dtot, int lcap, int seg0,
                                                           int stride, int world, int me,
                                                           bh_frec* __restrict__ send) {
  __shared__ int s_tot[64];
  if (threadIdx.x < 64) s_tot[threadIdx.x] = (int)threadIdx.x < world ? dtot[threadIdx.x] : 0;
  __syncthreads();
  const int Ln = lpos[rec_cap];
  if (Ln > lcap) return;
  const int blocks0 = seg0 + kSegBlocks0;
  // one work item per (exporting cell, child slot 0..7): a thread that wrote all eight children of its cell to up to
  // seven destinations was a chain of ~60 dependent loads and stores (48 us per rank-step at 8 x 1M; grid-stride:
  // the list length is only known on the device)
  for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < 8ll * Ln;
       it += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(it >> 3), k0 = (int)(it & 7);
    const int e = list_e[i];
    const int wv = w[e];  // (an unsplit cell of many bodies has more than eight: slot k0 takes k0, k0 + 8, ...)
    if (k0 > wv || (k0 == wv && !(wv & 1))) continue;  // slot wv of an odd block: the null digest that ends it
    // destinations that get this block: may open the cell, are not this rank, and their segment fits (else it is
    // sent closed) and has room for the block
    unsigned mask = wmask[e] & ~(1u << me);
    for (unsigned m = mask; m; m &= m - 1) {
      const int q = __ffs(m) - 1;
      if (kSegBlocks0 + s_tot[q] > stride || kSegBlocks0 + dstd[(size_t)q * lcap + i] + wv > stride) mask &= ~(1u << q);
    }
    if (!mask) continue;
    const int first = frec_get(frec, e).first;
    for (int k = k0; k <= wv; k += 8) {
      if (k == wv) {
        if (wv & 1)
          for (unsigned m = mask; m; m &= m - 1) {
            const int q = __ffs(m) - 1;
            const int off = kSegBlocks0 + dstd[(size_t)q * lcap + i];
            if (off + wv < stride) frec_put(send + (size_t)q * stride, off + wv, frec_null());
          }
        break;
      }
      // a child of the cell, or a body digest of an unsplit multi-body cell: one kind of block.  The child's own
      // export data is fetched once, then relocated per destination.
      const int c = first + k;
      const bh_frec cr = frec_get(frec, c);
      int wc = 0, ci = lcap;
      unsigned cm = 0u;
      if (c < rec_cap) {
        wc = w[c];
        if (wc > 0) {
          ci = lpos[c];
          cm = wmask[c];
        }
      }
      for (unsigned m = mask; m; m &= m - 1) {
        const int q = __ffs(m) - 1;
        bh_frec o = cr;
        if (c < rec_cap) {
          if (wc > 0 && ci < lcap && ((cm >> q) & 1u)) {
            o.first = blocks0 + dstd[(size_t)q * lcap + ci];
            o.meta = wc;
          } else if (o.thr2 >= 0.0f) {  // no body of rank q can open this child: its copy there is closed
            o.first = 0;
            o.thr2 = -1.0f;
          }
        }
        frec_put(send + (size_t)q * stride, kSegBlocks0 + dstd[(size_t)q * lcap + i] + k, o);
      }
    }
  }
}

// Gathered LET segments are records written by OTHER ranks: before any wave walks them, every openable record
// (thr2 >= 0) of every remote segment must keep its child block inside its own segment's block area, start it
// on an even record and have 1..8 children.  A record that does not is closed (thr2 = -1: accepted by every body,
// never opened) and BH_FLAG_DD_LET_INVALID is raised — a malformed or partially written segment then costs a
// wrong step that the caller is told about, never an out-of-bounds fetch.  (Cycles inside a segment are
// impossible to rule out locally; the walk's pop budget, BH_FLAG_TRAVERSAL_LIMIT, bounds them.)
// Block 0 also hands the host what it decides the step on: records 0 .. 3 (header + needs row, 128 bytes) of every
// received segment go to pinned memory with write-through stores, then a sequence number (bh_dd_let_check polls it).
// A device-to-host copy command for these 1 KB sat on the stream for 20 us between X4 and this kernel
// (profiles/r05_dd/step_timeline_world1_rccl.txt).
__global__ __launch_bounds__(256) void dd_validate_kernel(bh_frec* __restrict__ pool, int seg_base, int stride,
                                                          int world, int me, bh_devinfo* __restrict__ info,
                                                          int* __restrict__ host_rows, int* __restrict__ host_seq,
                                                          int seq, int* __restrict__ prev_rows, int use_prev) {
  // use_prev: this X4 moved a size per pair (x4_chunk of the needs in prev_rows) instead of the whole slot
  if (blockIdx.x == 0) {
    __shared__ int s_rows[64 * 32];
    __shared__ int s_hold;
    if (threadIdx.x == 0) s_hold = 0;
    for (int i = threadIdx.x; i < world * 32; i += 256) {
      const int q = i >> 5, dw = i & 31;
      const int v = reinterpret_cast<const int*>(pool)[((size_t)seg_base + (size_t)q * stride) * 8 + dw];
      s_rows[i] = v;
      __hip_atomic_store(host_rows + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's row words have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // the decision bh_dd_let_check takes on the host from these rows, taken here too: the force launches are enqueued
    // before the host has seen them and hold themselves back if the exchange has to be repeated (bh_devinfo.dd_hold)
    for (int q = threadIdx.x; q < world; q += 256) {
      const int* seg = s_rows + 32 * q;
      int need = seg[10];  // header record 0, field `first`; negative: that rank left the step
      bool over = false;
      if (need >= 0)
        for (int j = 0; j < world; j++) {
          const int nj = seg[seg_row_dword(j)];
          need = max(need, nj);
          if (use_prev && j != q) over = over || nj > x4_chunk(prev_rows[32 * q + seg_row_dword(j)], stride);
        }
      if (need < 0 || need > stride || over) atomicOr(&s_hold, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) info->dd_hold = s_hold;
    // a fitting exchange becomes the next step's yardstick (the host keeps the same copy: bh_dd_phase_force)
    if (!s_hold && prev_rows) {
      __syncthreads();  // (everybody has read the old rows)
      for (int i = threadIdx.x; i < world * 32; i += 256) prev_rows[i] = s_rows[i];
    }
  }
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)world * stride) return;
  const int q = (int)(t / stride), k = (int)(t - (long long)q * stride);
  if (q == me || k < kSegPieces0) return;  // own segment: written by this rank's export; records 0..3: header + needs row
  const long long seg0 = (long long)seg_base + (long long)q * stride;
  // the header's count = records the sender needed: beyond it (or in a segment sent closed because it did
  // not fit) nothing was written this step and nothing is reachable
  const int used = frec_get(pool, seg0).first;
  // (a size per pair: what lies beyond this pair's size did not travel — the exchange is then being repeated anyway)
  const int got = use_prev ? min(used, x4_chunk(prev_rows[32 * q + seg_row_dword(me)], stride)) : used;
  if (used > stride ? k >= kSegBlocks0 : k >= got) return;
  const long long e = (long long)seg_base + t;
  const bh_frec r = frec_get(pool, e);
  if (r.thr2 < 0.0f) return;  // closed (always accepted).  A NaN threshold fails this test and falls through to the
                              // range check: both fast walks open on `thr2 >= d2` (false for NaN: such a record is
                              // taken as a monopole, never opened), but the check does not rely on that
  const long long lo = seg0 + kSegBlocks0, hi = seg0 + stride;
  const bool ok = r.meta >= 1 && r.meta <= 8 && (r.first & 1) == 0 && (long long)r.first >= lo &&
                  (long long)r.first + r.meta <= hi && r.pad == frec_link(r.first, r.meta);  // (the walk follows the link)
  if (!ok) {
    reinterpret_cast<float*>(pool)[BH_FREC_DW(e, BH_FF_THR2)] = -1.0f;
    atomicOr(&info->flags, BH_FLAG_DD_LET_INVALID);
  }
}

// ------------------------------------------------------------------ top tree
// block-wide exclusive scans over up to kTopMax items, 4 consecutive items per thread
__device__ __forceinline__ int top_scan_i32(int* v /* LDS [kTopMax+1] */, int T, int* wsum /* [16] */) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int x[4], s = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int t = 4 * tid + i;
    x[i] = t < T ? v[t] : 0;
    s += x[i];
  }
  int inc = s;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int u = __shfl_up(inc, d, 64);
    if (lane >= d) inc += u;
  }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  int pre = 0, tot = 0;
  for (int i = 0; i < 16; i++) {
    if (i < wv) pre += wsum[i];
    tot += wsum[i];
  }
  int run = pre + inc - s;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int t = 4 * tid + i;
    if (t < T) v[t] = run;
    run += x[i];
  }
  if (tid == 0) v[T] = tot;
  __syncthreads();
  return tot;
}

__device__ __forceinline__ top5 t5_add(top5 a, top5 b) {
  return top5{a.m + b.m, a.x + b.x, a.y + b.y, a.z + b.z, a.o + b.o};
}
__device__ __forceinline__ top5 t5_shfl_up(top5 v, int d) {
  return top5{__shfl_up(v.m, d, 64), __shfl_up(v.x, d, 64), __shfl_up(v.y, d, 64), __shfl_up(v.z, d, 64),
              __shfl_up(v.o, d, 64)};
}

// side 0: every piece; side 1: only this rank's pieces, the others become null records (zero force,
// never opened); side 2: the reverse
__device__ __forceinline__ bh_frec top_piece_record(const bh_dd_piece* __restrict__ g, int slot, int me,
                                                    const bh_frec* __restrict__ pool, int seg_base, int stride,
                                                    int side) {
  const bh_dd_piece d = g[slot];
  if ((side == 1 && d.owner != me) || (side == 2 && d.owner == me)) return frec_null();
  if (d.owner == me) return frec_get(pool, d.rec_idx);
  const int k = slot - d.owner * kDescPerRank - 1;
  return frec_get(pool, (long long)seg_base + (long long)d.owner * stride + kSegPieces0 + k);
}

// Per-level bitmasks over the piece boundaries (row v+1: bit p set iff d[p] <= v): nearest-smaller
// queries become a masked word, a short word scan and clz / ctz / popcount instead of byte-by-byte LDS
// scans (same device as pairs_kernel of bh_tree.hip).  nw = words in use.
constexpr int kTopWords = kTopMax / 64;
__device__ __forceinline__ int tm_prev(const u64* row, int p) {  // highest set bit below p, or -1
  int w = p >> 6;
  u64 bits = row[w] & ((1ull << (p & 63)) - 1ull);
  while (bits == 0ull && w > 0) bits = row[--w];
  return bits ? w * 64 + 63 - __clzll((long long)bits) : -1;
}
__device__ __forceinline__ int tm_next(const u64* row, int p, int nw) {  // lowest set bit above p, or -1
  int w = p >> 6;
  u64 bits = row[w] & ~((2ull << (p & 63)) - 1ull);
  while (bits == 0ull && w < nw - 1) bits = row[++w];
  return bits ? w * 64 + __ffsll((long long)bits) - 1 : -1;
}
__device__ __forceinline__ int tm_count(const u64* row, int p, int q) {  // set bits in (p, q)
  const int w0 = p >> 6, w1 = q >> 6;
  const u64 above_p = ~((2ull << (p & 63)) - 1ull);
  const u64 below_q = (1ull << (q & 63)) - 1ull;
  if (w0 == w1) return __popcll(row[w0] & above_p & below_q);
  int c = __popcll(row[w0] & above_p) + __popcll(row[w1] & below_q);
  for (int w = w0 + 1; w < w1; w++) c += __popcll(row[w]);
  return c;
}

// record of the top cell covering pieces [c0, c1), branching at level Lb, children at pool index `first`
__device__ __forceinline__ bh_frec top_cell_record(const top5* __restrict__ ps, int c0, int c1, int Lb, int first,
                                                   int meta, float s0, float G, float theta, int side) {
  const top5 p1 = ps[c1], p0 = ps[c0];
  const double M = p1.m - p0.m;
  const double sx = p1.x - p0.x, sy = p1.y - p0.y, sz = p1.z - p0.z;
  bh_frec fr;
  const float mass = (float)M;
  if (mass > 1e-6f) {  // ref:180, as com_kernel
    fr.x = (float)(sx / M); fr.y = (float)(sy / M); fr.z = (float)(sz / M);
  } else {
    fr.x = (float)sx; fr.y = (float)sy; fr.z = (float)sz;
  }
  const bool massive = mass > 0.0f;
  // the MAC sees the whole cell (position, size); in the two-pass split an accepted cell contributes
  // the monopole of this pass's share of its mass at the same centre, so the two passes add up to it
  const double Mo = p1.o - p0.o;
  const float share = side == 0 ? mass : (float)(side == 1 ? Mo : M - Mo);
  fr.gm = (massive && share > 0.0f) ? G * share : 0.0f;
  if (massive) {
    const float t = ldexpf(s0, -Lb) / theta;
    fr.thr2 = t * t;
  } else {
    fr.thr2 = -1.0f;
  }
  fr.first = first;
  fr.meta = meta;
  fr.pad = 0;
  return fr;
}

// Top tree of the remote pass from the structure the own pass left in scratch (child ranges, branching
// levels, child-block offsets, prefix sums): one thread per record, no rebuild.
__global__ __launch_bounds__(256) void dd_top_emit_kernel(const bh_dd_piece* __restrict__ g, int me,
                                                          bh_frec* __restrict__ pool, int top_base, int seg_base,
                                                          int stride, const float* __restrict__ bounds, float G,
                                                          float theta, const top5* __restrict__ ps,
                                                          const int* __restrict__ cc0, const int* __restrict__ cc1,
                                                          const int4* __restrict__ ci, const int* __restrict__ ddi,
                                                          int side) {
  // e = position in the top tree: 0 the root, 1 padding, 2 + s the child slots (blocks at even positions)
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e > ddi[3]) return;
  const int4 c = ci[e];
  bh_frec fr;
  if (c.x == -2)
    fr = frec_null();
  else if (c.x < 0)
    fr = top_piece_record(g, c.y, me, pool, seg_base, stride, side);
  else
    fr = top_cell_record(ps, e == 0 ? 0 : cc0[e - 2], e == 0 ? ddi[2] : cc1[e - 2], c.x, top_base + c.y, c.z,
                         bounds[6], G, theta, side);
  frec_put(pool, top_base + e, fr);
}

__global__ __launch_bounds__(1024) void dd_top_kernel(const bh_dd_piece* __restrict__ g, int world, int me,
                                                      bh_frec* __restrict__ pool, int top_base, int seg_base,
                                                      int stride, const float* __restrict__ bounds, float G,
                                                      float theta, top5* __restrict__ ps, int* __restrict__ cc0,
                                                      int* __restrict__ cc1, int4* __restrict__ ci,
                                                      int* __restrict__ ddi, bh_devinfo* __restrict__ info,
                                                      int side) {
  __shared__ int offs[65];
  __shared__ int tslot[kTopMax];
  __shared__ signed char d[kTopMax + 64];
  __shared__ unsigned char pn[kTopMax + 1];  // children of the cell represented by boundary j (<= 8)
  __shared__ int cb[kTopMax + 1];
  __shared__ u64 lm[kB + 2][kTopWords];      // lm[v + 1]: boundaries with d <= v, v = -1 .. kB
  __shared__ int wsum[16];
  __shared__ top5 wsum4[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid < world)
    offs[tid + 1] = min(reinterpret_cast<const int*>(g + (size_t)tid * kDescPerRank)[0], BH_DD_PIECE_CAP);
  __syncthreads();
  if (tid == 0) {
    int o = 0;
    offs[0] = 0;
    for (int q = 0; q < world; q++) {
      o += offs[q + 1];
      offs[q + 1] = o;
    }
    if (o > kTopMax) atomicOr(&info->flags, BH_FLAG_DD_PIECES);
  }
  __syncthreads();
  const int T = min(offs[world], kTopMax);
  if (T < 1) return;
  for (int t = tid; t < T; t += 1024) {
    int q = 0;
    while (offs[q + 1] <= t) q++;
    tslot[t] = q * kDescPerRank + 1 + (t - offs[q]);
  }
  __syncthreads();
  for (int t = tid; t <= T; t += 1024)
    d[t] = (t == 0 || t == T) ? (signed char)-1
                              : (signed char)common_digits(g[tslot[t - 1]].key, g[tslot[t]].key, kB);
  // fp64 exclusive prefix of the piece sums (fixed association: identical on every rank)
  {
    const top5 zero = top5{0.0, 0.0, 0.0, 0.0, 0.0};
    top5 x[4], s = zero;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int t = 4 * tid + i;
      if (t < T) {
        const bh_dd_piece p = g[tslot[t]];
        x[i] = top5{p.sm, p.sx, p.sy, p.sz, p.owner == me ? p.sm : 0.0};
      } else {
        x[i] = zero;
      }
      s = t5_add(s, x[i]);
    }
    top5 inc = s;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
      const top5 u = t5_shfl_up(inc, dd);
      if (lane >= dd) inc = t5_add(u, inc);
    }
    if (lane == 63) wsum4[wv] = inc;
    __syncthreads();
    top5 pre = zero;
    for (int i = 0; i < wv; i++) pre = t5_add(pre, wsum4[i]);
    top5 excl = t5_shfl_up(inc, 1);
    if (lane == 0) excl = zero;
    top5 run = t5_add(pre, excl);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int t = 4 * tid + i;
      if (t < T) ps[t] = run;
      run = t5_add(run, x[i]);
      if (t == T - 1) ps[T] = run;
    }
  }
  __syncthreads();
  // level masks; positions past T count as sentinels (d = -1)
  const int nw = (T >> 6) + 1;
  for (int w = wv; w < nw; w += 16) {
    const int pos = w * 64 + lane;
    const int dv = pos < T ? (int)d[pos] : -1;
#pragma unroll
    for (int v = -1; v <= kB; v++) {
      const u64 bb = __ballot(dv <= v);
      if (lane == 0) lm[v + 1][w] = bb;
    }
  }
  __syncthreads();
  // cells: pair j at level L = d[j] names the cell [a, b); representative = its first boundary
  int ra[kTopMax / 1024], rb[kTopMax / 1024];
#pragma unroll
  for (int r = 0; r < kTopMax / 1024; r++) {
    const int j = tid + r * 1024;
    int nc = 0;
    ra[r] = rb[r] = 0;
    if (j >= 1 && j < T) {
      const int L = d[j];
      const u64* mle = lm[L + 1];
      const int q = tm_prev(mle, j);  // >= 0: d[0] = -1
      if (d[q] < L) {  // no earlier boundary of the same level inside the cell
        ra[r] = q;
        const int b = tm_next(lm[L], j, nw);  // exists: d[T] = -1
        rb[r] = b;
        nc = 2 + tm_count(mle, j, b);
      }
    }
    if (j < T) {
      pn[j] = (unsigned char)nc;
      cb[j] = (nc + 1) & ~1;  // child blocks start at even positions: an odd block is followed by padding
    }
  }
  __syncthreads();
  const int nslots = top_scan_i32(cb, T, wsum);
  // every representative lists the piece ranges of its children; then ONE THREAD PER CHILD builds the
  // record, so the dependent global loads (descriptor -> piece record) of all children overlap
#pragma unroll
  for (int r = 0; r < kTopMax / 1024; r++) {
    const int j = tid + r * 1024;
    if (j >= T || pn[j] == 0) continue;
    const int L = d[j], b = rb[r];
    const u64* mle = lm[L + 1];  // inside the cell every d >= L, so d <= L means d == L
    int e = cb[j];
    int c0 = ra[r], c1 = j;
    for (;;) {
      cc0[e] = c0;
      cc1[e] = c1;
      e++;
      if (c1 >= b) break;
      c0 = c1;
      const int nx = tm_next(mle, c0, nw);
      c1 = (nx < 0 || nx > b) ? b : nx;
    }
    if (pn[j] & 1) cc0[e] = cc1[e] = -1;  // padding slot
  }
  __syncthreads();
  const float s0 = bounds[6];
  if (tid == 0) {
    ddi[2] = T;
    ddi[3] = 1 + nslots;  // last position of the top tree
  }
  for (int e = tid; e <= 1 + nslots; e += 1024) {  // position 0 = the root, 1 = padding, 2 + s = child slot s
    const int c0 = e == 0 ? 0 : (e == 1 ? -1 : cc0[e - 2]);
    const int c1 = e == 0 ? T : (e == 1 ? -1 : cc1[e - 2]);
    bh_frec fr;
    int4 info4;
    if (c0 < 0) {
      fr = frec_null();
      info4 = make_int4(-2, 0, 0, 0);
    } else if (c1 - c0 == 1) {
      fr = top_piece_record(g, tslot[c0], me, pool, seg_base, stride, side);
      info4 = make_int4(-1, tslot[c0], 0, 0);
    } else {  // the cell branches at the first lowest boundary strictly inside the range: the smallest
      // level above the parent's (= the larger of the two end boundaries) with a boundary inside
      int l = c0 + 1, Lb = kB;
      for (int v = max((int)d[c0], (int)d[c1]) + 1; v < kB; v++) {
        const int sx = tm_next(lm[v + 1], c0, nw);
        if (sx >= 0 && sx < c1) {
          Lb = v;
          l = sx;
          break;
        }
      }
      fr = top_cell_record(ps, c0, c1, Lb, top_base + 2 + cb[l], (int)pn[l], s0, G, theta, side);
      info4 = make_int4(Lb, 2 + cb[l], (int)pn[l], 0);
    }
    frec_put(pool, top_base + e, fr);
    ci[e] = info4;
  }
}

__global__ __launch_bounds__(256) void dd_pack_ids_kernel(const float* __restrict__ s, const int* __restrict__ ids,
                                                          int n, float4* __restrict__ posm,
                                                          float4* __restrict__ velid) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t N = (size_t)n;
  posm[i] = make_float4(s[i], s[N + i], s[2 * N + i], s[6 * N + i]);
  velid[i] = make_float4(s[3 * N + i], s[4 * N + i], s[5 * N + i], __int_as_float(ids[i]));
}

}  // namespace

// ------------------------------------------------------------------ host side
#define BH_HIP(c, call)                       \
  do {                                        \
    hipError_t _e = (call);                   \
    if (_e != hipSuccess) {                   \
      (c)->last_hip = (int)_e;                \
      return BH_ERR_HIP;                      \
    }                                         \
  } while (0)

// the most a rebalance shifts a rank's share against its drift: 0.8 of the tolerance band (bodies)
// The side stream of the two-pass forms.  Measured on an MI355X (profiles/r05_dd/split_timeline_*): a kernel the
// main stream launches while a launch of the side stream holds every wave slot is not served before that launch has
// handed out its last workgroup, whatever the streams' priorities — dd_let_scan_kernel, 5 us of work, took 302 us
// beside the own pass.  So (a) the own pass is enqueued behind the LET export (bh_dd_phase_let) and hides X4 only,
// and (b) the side stream leaves kReserveCus compute units to the main stream (a CU mask; bits are dealt to the XCDs in
// turn, so clearing the highest ones takes the same number of units from every XCD): X4's kernel, the validation and
// the top-tree kernels start at once on them.  Without CU-mask support: a lowest-priority stream, as round 4.
// Contexts of one process on one device (the ranks of a one-GPU rehearsal) share ONE such stream: every masked stream
// is a hardware queue of its own, and with eight of them beside the main streams the queues are time-sliced — whole
// ranks' side-stream launches then start hundreds of microseconds late (seen in bh_rank_replay_force_phase tables
// as two or three ranks whose split forms cost 0.2-0.4 ms more, different ranks in every run).
constexpr int kReserveCus = 16;
namespace {
struct side_stream_slot {
  hipStream_t stream = nullptr;
  int users = 0;
};
std::mutex g_side_mutex;
side_stream_slot g_side[64];
}  // namespace
static bool dd_make_side_stream_raw(const bh_ctx* c, hipStream_t* out);
static bool dd_make_side_stream(const bh_ctx* c, hipStream_t* out) {
  if (c->device < 0 || c->device >= 64) return dd_make_side_stream_raw(c, out);
  std::lock_guard<std::mutex> lk(g_side_mutex);
  side_stream_slot& sl = g_side[c->device];
  if (sl.users == 0 && !dd_make_side_stream_raw(c, &sl.stream)) return false;
  sl.users++;
  *out = sl.stream;
  return true;
}
static void dd_release_side_stream(const bh_ctx* c, hipStream_t s) {
  if (c->device >= 0 && c->device < 64) {
    std::lock_guard<std::mutex> lk(g_side_mutex);
    side_stream_slot& sl = g_side[c->device];
    if (sl.stream == s) {
      if (--sl.users == 0) {
        (void)hipStreamDestroy(s);
        sl.stream = nullptr;
      }
      return;
    }
  }
  (void)hipStreamDestroy(s);
}
static bool dd_make_side_stream_raw(const bh_ctx* c, hipStream_t* out) {
  int reserve = kReserveCus;
#ifdef BH_STUDY
  if (getenv("BH_DD_RESERVE_CUS")) reserve = atoi(getenv("BH_DD_RESERVE_CUS"));
#endif
  const int cus = c->num_cus;
  if (reserve > 0 && cus > 2 * reserve) {
    uint32_t mask[32];
    const int words = (cus + 31) / 32;
    if (words <= 32) {
      for (int w = 0; w < words; w++) mask[w] = 0;
      for (int b = 0; b < cus - reserve; b++) mask[b >> 5] |= 1u << (b & 31);
      if (hipExtStreamCreateWithCUMask(out, (uint32_t)words, mask) == hipSuccess) return true;
      (void)hipGetLastError();
    }
  }
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return false;
  return hipStreamCreateWithPriority(out, hipStreamNonBlocking, least) == hipSuccess;
}

static long long dd_drift_cap(const bh_dd_state* d) {
  return (long long)(0.8 * (double)kSplitTolerance * (double)d->n_total / (double)d->world);
}

static void dd_set_n(bh_ctx* c, int n) {
  c->n = n;
  c->sort_tiles = (n + BH_SORT_TILE - 1) / BH_SORT_TILE;
  if (c->sort_tiles < 1) c->sort_tiles = 1;
}

void bh_dd_free(bh_ctx* c) {
  bh_dd_state* d = c->dd;
  if (!d) return;
  if (d->stream_own) {  // the own pass may still be reading the arrays freed below
    (void)hipStreamSynchronize(d->stream_own);
    dd_release_side_stream(c, d->stream_own);
  }
  void* ptrs[] = {d->w, d->dst, d->flag, d->fpos, d->nloc, d->skeys, d->drift, d->piece_tmp,
                  d->piece_idx, d->ddi, d->boxes, d->rbox, d->top_ps, d->top_a, d->top_b, d->top_ci, d->acc2,
                  d->cls_done, d->abs_done, d->arrive, c->dd_minmax, d->wmask, d->list_e, d->list_w, d->list_m, d->dstd,
                  d->dtot, d->csum, d->mark_cnt, d->mark_done};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (d->host) (void)hipHostFree(d->host);
  if (d->host_rows) (void)hipHostFree(d->host_rows);
  if (d->prev_rows_dev) (void)hipFree(d->prev_rows_dev);
  free(d->prev_rows);
  if (d->ev_let) (void)hipEventDestroy(d->ev_let);
  if (d->ev_x3) (void)hipEventDestroy(d->ev_x3);
  if (d->ev_own) (void)hipEventDestroy(d->ev_own);
  if (d->ev_top1) (void)hipEventDestroy(d->ev_top1);
  if (d->ev_x4) (void)hipEventDestroy(d->ev_x4);

  c->acc2 = nullptr;
  c->dd_minmax = nullptr;
  c->dd_minmax_ok = false;
  free(d);
  c->dd = nullptr;
  c->bounds_next_ok = false;  // the domain-decomposed steps moved and exchanged bodies
  c->order_hint = false;
}

extern "C" {

int bh_dd_query(int n_cap, int world, int mig_cap, int let_cap, bh_dd_sizes* o) {
  // a rank contributes at most 2 spines x 21 levels x 7 = 294 pieces, the top tree holds kTopMax = 4096 of them:
  // up to 13 ranks can never overflow it (one node has 8 GPUs); more are refused here rather than risking the
  // BH_FLAG_DD_PIECES error in the middle of a run
  if (!o || n_cap < 1 || world < 1 || world > kTopMax / 294 || mig_cap < 1) return BH_ERR_BAD_ARG;
  const long long let_min = kSegBlocks0;
  if (let_cap < let_min || (let_cap & 1)) return BH_ERR_BAD_ARG;  // segments hold whole digest pairs
  long long rec_cap = (long long)BH_FREC_POOL(BH_REC_CAP((long long)n_cap), n_cap);  // tree + body digests
  rec_cap += rec_cap & 1;
  const long long top_cap = kTopCap;
  o->x1_bytes = (int64_t)x1_floats(world) * 4;
  o->x2_bytes = 32 + 32LL * mig_cap;
  o->x3_bytes = (int64_t)sizeof(bh_dd_piece) * kDescPerRank;
  o->top_base = rec_cap;
  o->seg_base = rec_cap + 3 * top_cap;  // three top trees: remote pass (or the whole tree), own pass, whole tree
  o->pool_records = o->seg_base + (long long)world * let_cap + 8;
  o->let_min = let_min;
  o->let_cap = let_cap;
  if (o->pool_records >= (1LL << 27)) return BH_ERR_BAD_ARG;  // fast kernel: 32-bit byte offsets
  return BH_OK;
}

int bh_dd_init(bh_ctx* c, int world, int rank, int64_t n_total, int mig_cap, int let_cap, void* pool,
               int64_t pool_records) {
  if (!c || !pool || rank < 0 || rank >= world || c->dd) return BH_ERR_BAD_ARG;
  // (a depth cap leaves unsplit cells of many bodies that remote bodies can open: their child blocks have more than
  // the 8 records dd_validate_kernel accepts from another rank — the bound that keeps a walk over a malformed segment
  // short)
  if (c->p.key_bits != 63 || c->p.leaf_cap != 1 || c->D != 21 || c->p.strict_fp || c->p.literal_force)
    return BH_ERR_BAD_ARG;
  bh_dd_sizes sz;
  // the context was created with n = body capacity
  const int n_cap = c->n;
  int s = bh_dd_query(n_cap, world, mig_cap, let_cap, &sz);
  if (s) return s;
  if (pool_records < sz.pool_records || (long long)world * mig_cap > 4LL * n_cap || n_cap < 1024)
    return BH_ERR_BAD_ARG;
  BH_HIP(c, hipSetDevice(c->device));
  bh_dd_state* d = (bh_dd_state*)calloc(1, sizeof(bh_dd_state));
  if (!d) return BH_ERR_OOM;
  c->dd = d;
  c->bounds_next_ok = false;
  c->order_hint = false;
  d->world = world;
  d->rank = rank;
  d->n_total = n_total;
  d->mig_cap = mig_cap;
  d->let_cap = let_cap;
  d->pool = (bh_frec*)pool;
  d->pool_records = pool_records;
  d->top_base = (int)sz.top_base;
  d->top_base2 = (int)sz.top_base + kTopCap;
  d->top_base3 = (int)sz.top_base + 2 * kTopCap;
  d->split_pct = 100;
  d->seg_base = (int)sz.seg_base;
  size_t fl = (size_t)n_cap;
  if ((size_t)world * mig_cap > fl) fl = (size_t)world * mig_cap;
  bool ok = true;
  ok = ok && hipMalloc((void**)&d->w, ((size_t)c->rec_cap + 1 + 64) * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&d->dst, ((size_t)c->rec_cap + 1 + 64) * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&d->flag, (fl + 1 + 64) * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&d->fpos, (fl + 1 + 64) * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&d->nloc, 64 * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&d->skeys, 64 * 8) == hipSuccess;
  ok = ok && hipMalloc((void**)&d->drift, 192 * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&d->piece_tmp, BH_DD_PIECE_CAP * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&d->piece_idx, BH_DD_PIECE_CAP * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&d->ddi, 16 * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&d->boxes, (size_t)world * BH_DD_PIECE_CAP * sizeof(float4)) == hipSuccess;
  ok = ok && hipMalloc((void**)&d->rbox, (size_t)2 * 64 * sizeof(float4)) == hipSuccess;
  ok = ok && hipMalloc((void**)&d->top_ps, 2 * ((size_t)kTopMax + 1) * sizeof(top5)) == hipSuccess;
  ok = ok && hipMalloc((void**)&d->top_a, 2 * (size_t)kTopCap * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&d->top_b, 2 * (size_t)kTopCap * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&d->top_ci, 2 * (size_t)kTopCap * sizeof(int4)) == hipSuccess;
  ok = ok && hipMalloc((void**)&d->acc2, ((size_t)n_cap + 64) * sizeof(float4)) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&d->ev_x3, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&d->ev_own, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&d->ev_top1, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&d->ev_x4, hipEventDisableTiming) == hipSuccess;
  ok = ok && dd_make_side_stream(c, &d->stream_own);
  ok = ok && hipHostMalloc((void**)&d->host, (64 + 8) * sizeof(int)) == hipSuccess;
  if (ok) memset(d->host, 0, (64 + 8) * sizeof(int));  // [64..67] migration results, [68] their sequence number, [70] X4 headers'
  ok = ok && hipHostMalloc((void**)&d->host_rows, 64 * 32 * sizeof(int)) == hipSuccess;
  d->prev_rows = (int*)calloc(64 * 32, sizeof(int));
  ok = ok && d->prev_rows != nullptr;
  ok = ok && hipMalloc((void**)&d->prev_rows_dev, 64 * 32 * sizeof(int)) == hipSuccess;
  ok = ok && hipMemset(d->prev_rows_dev, 0, 64 * 32 * sizeof(int)) == hipSuccess;
  ok = ok && hipMalloc((void**)&d->wmask, ((size_t)c->rec_cap + 1 + 64) * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&d->list_e, ((size_t)n_cap + 64) * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&d->list_w, ((size_t)n_cap + 64) * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&d->list_m, ((size_t)n_cap + 64) * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&d->dstd, ((size_t)world * n_cap + 64) * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&d->dtot, 64 * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&d->csum, ((size_t)world * (n_cap / 8192 + 2) + 64) * 4) == hipSuccess;
  {
    const size_t mb = (size_t)c->rec_cap / kMarkChunk + 3;
    ok = ok && hipMalloc((void**)&d->mark_cnt, 2 * mb * 4 + 256) == hipSuccess;
    ok = ok && hipMalloc((void**)&d->mark_done, (mb / 32 + 8) * sizeof(u32)) == hipSuccess;
    ok = ok && hipMemset(d->mark_done, 0, (mb / 32 + 8) * sizeof(u32)) == hipSuccess;
  }
  d->let_mode = 0;
  ok = ok && hipMalloc((void**)&c->dd_minmax, 8 * sizeof(float)) == hipSuccess;
  c->dd_minmax_ok = false;
  {
    const size_t nc = (size_t)n_cap / kClsTile / 32 + 8;
    ok = ok && hipMalloc((void**)&d->cls_done, nc * sizeof(u32)) == hipSuccess;
    ok = ok && hipMemset(d->cls_done, 0, nc * sizeof(u32)) == hipSuccess;
    const size_t na = (size_t)world * ((size_t)mig_cap / 1024 + 1) / 32 + 8;
    ok = ok && hipMalloc((void**)&d->abs_done, na * sizeof(u32)) == hipSuccess;
    ok = ok && hipMemset(d->abs_done, 0, na * sizeof(u32)) == hipSuccess;
    ok = ok && hipMalloc((void**)&d->arrive, 64 * sizeof(int)) == hipSuccess;
    ok = ok && hipMemset(d->arrive, 0, 64 * sizeof(int)) == hipSuccess;
  }
  ok = ok && hipEventCreateWithFlags(&d->ev_let, hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    bh_dd_free(c);
    return BH_ERR_OOM;
  }
  d->samp_cap = samp_cap_of(world);
  BH_HIP(c, hipMemsetAsync(d->nloc, 0, 64 * 4, c->stream));
  BH_HIP(c, hipMemsetAsync(d->drift, 0, 192 * 4, c->stream));
  BH_HIP(c, hipMemsetAsync(d->ddi, 0, 16 * 4, c->stream));  // piece counter: re-cleared by dd_describe_kernel
  // the caller's pool becomes the record pool: the COM stage writes the local tree at [0, rec_cap)
  BH_HIP(c, hipMemsetAsync(pool, 0, (size_t)pool_records * sizeof(bh_frec), c->stream));
  c->frec_own = c->frec;
  c->frec = d->pool;
  BH_HIP(c, hipStreamSynchronize(c->stream));
  return BH_OK;
}

int bh_dd_upload(bh_ctx* c, int n_loc, const float* x, const float* y, const float* z, const float* vx,
                 const float* vy, const float* vz, const float* m, const int32_t* ids) {
  if (!c || !c->dd || !x || !y || !z || !vx || !vy || !vz || !m || !ids) return BH_ERR_BAD_ARG;
  const int n_cap = (c->rec_cap - 8) / 3;
  if (n_loc < 1 || n_loc > n_cap) return BH_ERR_BAD_ARG;
  BH_HIP(c, hipSetDevice(c->device));
  dd_set_n(c, n_loc);
  const size_t N = (size_t)n_loc, nb = N * sizeof(float);
  const float* src[7] = {x, y, z, vx, vy, vz, m};
  for (int k = 0; k < 7; k++)
    BH_HIP(c, hipMemcpyAsync(c->stage_buf + k * N, src[k], nb, hipMemcpyHostToDevice, c->stream));
  BH_HIP(c, hipMemcpyAsync(c->vals[0], ids, N * 4, hipMemcpyHostToDevice, c->stream));
  c->cur = 0;
  c->order_hint = false;  // caller order
  c->bounds_next_ok = false;
  c->dd_minmax_ok = false;
  dd_pack_ids_kernel<<<(n_loc + 255) / 256, 256, 0, c->stream>>>(c->stage_buf, (const int*)c->vals[0], n_loc,
                                                                  c->posm[0], c->velid[0]);
  BH_HIP(c, hipGetLastError());
  c->splitter_off = false;
  c->slow_seen = 0;
  c->slow_seen_sorts = c->sort_calls;
  BH_HIP(c, hipMemsetAsync(c->info, 0, sizeof(bh_devinfo), c->stream));
  BH_HIP(c, hipMemsetAsync(c->acc, 0, N * sizeof(float4), c->stream));
  // new bodies: the boundaries of an earlier run mean nothing for them (the first step draws sample quantiles:
  // ddi[10] "splitter keys valid"), nor do its counts, its boundary statistics and its drift estimate
  BH_HIP(c, hipMemsetAsync(c->dd->ddi + 9, 0, 4 * sizeof(int), c->stream));
  BH_HIP(c, hipMemsetAsync(c->dd->nloc, 0, 64 * sizeof(int), c->stream));
  BH_HIP(c, hipMemsetAsync(c->dd->drift, 0, 192 * sizeof(int), c->stream));
  c->dd->prev_ok = false;  // (sizes per pair start over: the first X4 moves whole slots)
  c->dd->x4_v = false;
  BH_HIP(c, hipStreamSynchronize(c->stream));
  c->stage = BH_ST_UPLOADED;
  c->ever = BH_ST_UPLOADED;
  c->steps = 0;
  return BH_OK;
}

int bh_dd_cube_pack(bh_ctx* c, void* send_x1) {
  if (!c || !c->dd || !send_x1) return BH_ERR_BAD_ARG;
  if (!(c->stage & BH_ST_UPLOADED)) return BH_ERR_ORDER;
  bh_dd_state* d = c->dd;
  // the previous step's integrate kernel already folded this rank's min / max (bh_integrate in dd mode);
  // after an upload two bbox kernels do
  if (!c->dd_minmax_ok) BH_HIP(c, bhk_bbox_raw(c, c->dd_minmax));
  c->dd_minmax_ok = false;
  // stride sized so that a rank holding up to 1.34x its fair share still fits its sample slots
  const double g = 1.34 * (double)d->n_total / (double)kSampTotal;
  const int th = d->samp_cap > 8 ? d->samp_cap : 8;
  dd_x1_pack_kernel<<<(th + 255) / 256, 256, 0, c->stream>>>((float*)send_x1, c->dd_minmax, c->n, c->posm[c->cur],
                                                             g > 1.0 ? g : 1.0, d->samp_cap, d->nloc, d->world,
                                                             d->rank, d->n_total, d->drift, dd_drift_cap(d));
  BH_HIP(c, hipGetLastError());
  return BH_OK;
}

int bh_dd_cube_apply(bh_ctx* c, const void* gathered_x1) {
  if (!c || !c->dd || !gathered_x1) return BH_ERR_BAD_ARG;
  bh_dd_state* d = c->dd;
  const int xf = x1_floats(d->world);
  dd_split_kernel<<<1, 1024, 0, c->stream>>>((const float*)gathered_x1, d->world, xf, d->samp_cap, c->bounds,
                                             c->p.key_curve, d->skeys, d->ddi, d->n_total, kSplitTolerance, d->drift,
                                             dd_drift_cap(d));
  BH_HIP(c, hipGetLastError());
  c->stage = BH_ST_UPLOADED | BH_ST_BBOX;
  c->ever |= BH_ST_BBOX;
  return BH_OK;
}

int bh_dd_migrate_pack(bh_ctx* c, void* send_x2, int limit) {
  if (!c || !c->dd || !send_x2) return BH_ERR_BAD_ARG;
  if (!(c->stage & BH_ST_BBOX)) return BH_ERR_ORDER;
  bh_dd_state* d = c->dd;
  if (limit < 1 || limit > d->mig_cap) return BH_ERR_BAD_ARG;
  const int n = c->n;
  const int blocks = (n + kClsTile - 1) / kClsTile > 0 ? (n + kClsTile - 1) / kClsTile : 1;
  unsigned char* kept = reinterpret_cast<unsigned char*>(d->flag);
  int* bcnt = d->fpos;
  int* bbase = d->fpos + blocks + 1;
  dd_classify_kernel<<<blocks, 256, 0, c->stream>>>(c->posm[c->cur], n, c->bounds, c->p.key_curve, d->skeys,
                                                    d->world - 1, d->rank, kept, bcnt, bbase, d->cls_done,
                                                    (int*)send_x2, limit, d->ddi + 12);
  dd_compact_kernel<<<blocks, 256, 0, c->stream>>>(c->posm[c->cur], c->velid[c->cur], n, kept, bbase,
                                                   c->posm[c->cur ^ 1], c->velid[c->cur ^ 1], (float4*)send_x2,
                                                   limit);
  BH_HIP(c, hipGetLastError());
  if (c->keys_split) {  // bucket counts of keys that no sort consumed (an earlier round of this step): void them
    BH_HIP(c, hipMemsetAsync(c->sp_count + 256 * (c->sp_par & 1), 0, 256 * sizeof(u32), c->stream));
    c->keys_split = false;
  }
  c->cur ^= 1;
  return BH_OK;
}

int bh_dd_migrate_apply(bh_ctx* c, const void* gathered_x2, int limit, int* n_loc, int* more, int* most) {
  if (!c || !c->dd || !gathered_x2) return BH_ERR_BAD_ARG;
  if (!(c->stage & BH_ST_BBOX)) return BH_ERR_ORDER;
  bh_dd_state* d = c->dd;
  if (limit < 1 || limit > d->mig_cap) return BH_ERR_BAD_ARG;
  const float4* g = (const float4*)gathered_x2;
  const size_t f4 = 2 + 2 * (size_t)limit;
  const int n_cap = (c->rec_cap - 8) / 3;
  const int seq = ++d->absorb_seq;
  const int cpr = (limit + 1023) / 1024;  // blocks per rank
  const int blocks = d->world * cpr;
  unsigned char* mine_f = reinterpret_cast<unsigned char*>(d->flag);
  int* bcnt = d->fpos;
  int* bbase = d->fpos + blocks + 1;
  dd_absorb_flag_kernel<<<blocks, 256, 0, c->stream>>>(g, d->world, limit, f4, cpr, c->bounds, c->p.key_curve, d->skeys,
                                                       d->rank, n_cap, mine_f, bcnt, bbase, d->arrive, d->abs_done,
                                                       c->info, d->nloc, d->ddi + 4, d->host + 64, seq);
  dd_absorb_copy_kernel<<<blocks, 256, 0, c->stream>>>(g, d->world, limit, f4, cpr, d->rank, n_cap, mine_f, bbase,
                                                       c->posm[c->cur], c->velid[c->cur]);
  BH_HIP(c, hipGetLastError());
  // The host needs the new body count before it can size the next launches — but not the first of them: the key
  // kernel of the splitter sort reads the count the absorb kernel left on the device (nloc) and is launched here on a
  // bound (what this rank held + every slot of the exchange), so the stream has ~18 us of work while the host polls.
  // If another migration round follows, its keys are never sorted: bh_dd_migrate_pack voids the bucket counts.
  d->keys_spec = false;
  {
    const long long bound = (long long)c->n + (long long)d->world * limit;
    const int n_upper = (int)(bound < n_cap ? bound : n_cap);
    if (c->B == 21 && bhk_sort_split_eligible(c, n_upper)) {
      BH_HIP(c, bhk_keys_split(c, d->nloc + d->rank, n_upper));
      d->keys_spec = true;
    }
  }
  // poll the pinned result words (a few microseconds after the kernel's store) instead of hipStreamSynchronize (tens
  // of microseconds of wake-up latency: a 40 us hole on the stream in round 2's trace); bounded, then the stream wait
  // decides
  {
    volatile int* hs = d->host + 68;
    bool seen = false;
    for (long spin = 0; spin < 400000000L; spin++) {
      if (__atomic_load_n(hs, __ATOMIC_ACQUIRE) == seq) {
        seen = true;
        break;
      }
      if ((spin & 0xfffff) == 0xfffff && hipStreamQuery(c->stream) != hipErrorNotReady) break;
    }
    if (!seen) {
      BH_HIP(c, hipStreamSynchronize(c->stream));
      if (__atomic_load_n(hs, __ATOMIC_ACQUIRE) != seq) return BH_ERR_HIP;
    }
  }
  const int nl = d->host[64], flags = d->host[65];
  if (n_loc) *n_loc = nl;
  if (more) *more = d->host[66] > 0 ? 1 : 0;
  if (most) *most = d->host[67];
  if (flags & BH_FLAG_DD_BODIES) return BH_ERR_POOL_OVERFLOW;
  if (nl < 2 || nl > n_cap) return BH_ERR_POOL_OVERFLOW;
  // any other sticky flag was raised by the PREVIOUS step (pieces / pool / stack / sort / traversal / LET):
  // its forces were invalid; the flags are already on the host here, so reporting them costs nothing
  if (flags) return BH_ERR_DEVICE_FLAG;
  dd_set_n(c, nl);
  return BH_OK;
}

int bh_dd_tree(bh_ctx* c, void* send_x3) {
  if (!c || !c->dd || !send_x3) return BH_ERR_BAD_ARG;
  if (!(c->stage & BH_ST_BBOX)) return BH_ERR_ORDER;
  bh_dd_state* d = c->dd;
  if (d->keys_spec && c->keys_split)  // the keys, splitters and bucket counts are on their way (bh_dd_migrate_apply)
    d->keys_spec = false;
  else
    BH_HIP(c, bhk_keys(c));
  c->key_buf = 0;
  BH_HIP(c, bhk_sort(c));
  BH_HIP(c, bhk_build(c, true));  // (+ the fp64 COM prefix scan, riding in the build's launches: as bh_step)
  // digests only (the walk's records) + the list of this rank's pieces (com_kernel); the piece kernels read the proto
  // records
  BH_HIP(c, bhk_com_records(c, false, d->piece_tmp, d->ddi));
  dd_describe_kernel<<<1, BH_DD_PIECE_CAP, 0, c->stream>>>(d->piece_tmp, d->ddi, d->ddi, c->rec,
                                                           c->keys[c->key_buf], c->P, c->posm[c->cur], c->bounds,
                                                           c->p.key_curve, d->rank, c->n, (bh_dd_piece*)send_x3,
                                                           d->piece_idx, c->info);
  BH_HIP(c, hipGetLastError());
  c->stage |= BH_ST_MORTON | BH_ST_SORT | BH_ST_BUILD | BH_ST_COM;
  c->ever |= BH_ST_MORTON | BH_ST_SORT | BH_ST_BUILD | BH_ST_COM;
  return BH_OK;
}

int bh_dd_let_pack(bh_ctx* c, const void* gathered_x3, void* send_x4, int stride) {
  if (!c || !c->dd || !gathered_x3 || !send_x4) return BH_ERR_BAD_ARG;
  if (!(c->stage & BH_ST_COM)) return BH_ERR_ORDER;
  bh_dd_state* d = c->dd;
  if (stride < kSegBlocks0 || stride > d->let_cap || (stride & 1)) return BH_ERR_BAD_ARG;
  dd_boxes_kernel<<<d->world, 256, 0, c->stream>>>((const bh_dd_piece*)gathered_x3, d->world, d->rank, c->bounds,
                                                   d->boxes, d->rbox, d->ddi);
  const int blocks = (c->rec_cap + 1 + 255) / 256;
  const int mark_blocks = (c->rec_cap + kMarkChunk) / kMarkChunk;
  dd_mark_kernel<<<mark_blocks < kMarkGrid * c->num_cus ? mark_blocks : kMarkGrid * c->num_cus, 256, 0, c->stream>>>(
      c->frec, c->info, c->rec_cap, d->boxes, d->rbox, d->world, d->ddi, c->p.eps2, d->w,
      d->let_mode == 1 ? d->wmask : nullptr, d->dst, d->mark_cnt, d->mark_cnt + mark_blocks + 1, d->mark_done, mark_blocks);
  BH_HIP(c, hipGetLastError());
  if (d->let_mode == 1) {  // `send_x4` holds world segments of `stride` records, exchanged with an all-to-all
    const int n_cap = (c->rec_cap - 8) / 3;
    // dst = lpos: list position of every exporting cell (dd_mark_kernel compacted them block by block)
    dd_let_list_kernel<<<blocks, 256, 0, c->stream>>>(d->w, d->wmask, d->dst, d->mark_cnt + mark_blocks + 1, mark_blocks,
                                                      c->rec_cap, n_cap, d->list_e, d->list_w, d->list_m, c->info);
    {
      const int nch = (n_cap + kLetChunk - 1) / kLetChunk;
      const dim3 grid((unsigned)(nch < 32 ? nch : 32), (unsigned)d->world);
      dd_let_sums_kernel<<<grid, 1024, 0, c->stream>>>(d->list_w, d->list_m, d->dst, c->rec_cap, n_cap, nch, d->csum);
      dd_let_scan_kernel<<<grid, 1024, 0, c->stream>>>(d->list_w, d->list_m, d->dst, c->rec_cap, n_cap, nch, d->csum,
                                                       d->dstd, d->dtot);
    }
    const int seg0 = d->seg_base + d->rank * stride;
    dd_export_pd_head_kernel<<<d->world, BH_DD_PIECE_CAP, 0, c->stream>>>(
        c->frec, c->rec_cap, d->w, d->wmask, d->dst, d->dstd, d->dtot, n_cap, d->piece_idx, d->ddi, seg0, stride,
        d->world, d->rank, (bh_frec*)send_x4);
    dd_export_pd_kernel<<<(int)((8ll * n_cap + 255) / 256 < 4096 ? (8ll * n_cap + 255) / 256 : 4096), 256, 0, c->stream>>>(c->frec, c->rec_cap, d->w, d->wmask, d->dst,
                                                                   d->list_e, d->dstd, d->dtot, n_cap, seg0, stride,
                                                                   d->world, d->rank, (bh_frec*)send_x4);
    BH_HIP(c, hipGetLastError());
    return BH_OK;
  }
  BH_HIP(c, bhk_scan_i32_even(c, d->w, d->dst, c->rec_cap));  // exported blocks start at even records
  dd_export_kernel<<<blocks, 256, 0, c->stream>>>(c->frec, c->rec_cap, d->w, d->dst,
                                                  d->piece_idx, d->ddi, d->seg_base + d->rank * stride, stride,
                                                  d->world, (bh_frec*)send_x4);
  BH_HIP(c, hipGetLastError());
  return BH_OK;
}

int bh_dd_set_split_percent(bh_ctx* c, int pct) {
  if (!c || !c->dd || pct < 1 || pct > 100) return BH_ERR_BAD_ARG;
  c->dd->split_pct = pct;
  return BH_OK;
}

int bh_dd_set_serial(bh_ctx* c, int on) {
  if (!c || !c->dd) return BH_ERR_BAD_ARG;
  c->dd->serial = on != 0;
  return BH_OK;
}

int bh_dd_set_let_mode(bh_ctx* c, int mode) {
  if (!c || !c->dd || (mode != 0 && mode != 1)) return BH_ERR_BAD_ARG;
  c->dd->let_mode = mode;
  return BH_OK;
}

// first half: what needs nothing but X3 — the own pass's top tree (one block) and the zeroes bh_dd_download adds for
// the bodies that get no remote pass — goes to the side stream at once, beside the LET kernels
static int dd_force_local_prepare(bh_ctx* c, const void* gathered_x3) {
  bh_dd_state* d = c->dd;
  d->split = true;
  hipStream_t so = d->serial ? c->stream : d->stream_own;
  BH_HIP(c, hipEventRecord(d->ev_x3, c->stream));  // the X3 gather and the local tree are complete here
  BH_HIP(c, hipStreamWaitEvent(so, d->ev_x3, 0));
  dd_top_kernel<<<1, 1024, 0, so>>>((const bh_dd_piece*)gathered_x3, d->world, d->rank, d->pool,
                                            d->top_base2, d->seg_base, kSegBlocks0, c->bounds, c->p.G,
                                            c->p.theta, d->top_ps + (kTopMax + 1), d->top_a + kTopCap,
                                            d->top_b + kTopCap, d->top_ci + kTopCap, d->ddi,
                                            c->info, 1);
  BH_HIP(c, hipGetLastError());
  BH_HIP(c, hipEventRecord(d->ev_top1, so));  // the remote pass re-emits from this tree's scratch
  // Partial two-pass step (split_pct < 100): only the first bodies' walk is split — their own pass lasts as long as
  // X4 takes —, the others wait for X4 and are walked in ONE pass: the exchange is hidden and the price of two
  // passes (a second drain, the top levels twice) is paid for a fraction of the bodies.
  d->own_hi = d->split_pct >= 100 ? c->n : (int)((long long)c->n * d->split_pct / 100) / 256 * 256;
  if (d->own_hi < c->n)  // (bh_dd_download adds the two partial accelerations: none of a remote pass beyond own_hi)
    BH_HIP(c, hipMemsetAsync(d->acc2 + d->own_hi, 0, (size_t)(c->n - d->own_hi) * sizeof(float4), so));
  return BH_OK;
}
// second half: the launch that fills the GPU.  after_let: behind what the main stream holds at this point (the LET
// export) — a saturating launch of the side stream starves whatever the main stream launches after it
// (dd_make_side_stream), so the own pass hides X4 and not the LET kernels.
static int dd_force_local_launch(bh_ctx* c, bool after_let) {
  bh_dd_state* d = c->dd;
  hipStream_t so = d->serial ? c->stream : d->stream_own;
  if (after_let && so != c->stream) {
    BH_HIP(c, hipEventRecord(d->ev_let, c->stream));
    BH_HIP(c, hipStreamWaitEvent(so, d->ev_let, 0));
  }
  if (d->own_hi > 0) BH_HIP(c, bhk_force_root(c, 0, d->own_hi, d->top_base2, so, c->acc));
  BH_HIP(c, hipEventRecord(d->ev_own, so));
  return BH_OK;
}

// One-pass step: the top tree's STRUCTURE (child ranges, branching levels, block offsets, prefix sums of the pieces'
// fp64 sums) needs X3 only — one block, 16 us — so it is built on the side stream beside the LET kernels, as the tree
// of an own pass would be (remote pieces null); after X4 bh_dd_top only re-emits the records with every piece real
// (dd_top_emit_kernel, 5 us): 11 us less between X4 and the force launch.
static int dd_top_early(bh_ctx* c, const void* gathered_x3) {
  bh_dd_state* d = c->dd;
  hipStream_t so = d->serial ? c->stream : d->stream_own;
  BH_HIP(c, hipEventRecord(d->ev_x3, c->stream));  // the X3 gather and the local tree are complete here
  BH_HIP(c, hipStreamWaitEvent(so, d->ev_x3, 0));
  dd_top_kernel<<<1, 1024, 0, so>>>((const bh_dd_piece*)gathered_x3, d->world, d->rank, d->pool, d->top_base2,
                                    d->seg_base, kSegBlocks0, c->bounds, c->p.G, c->p.theta,
                                    d->top_ps + (kTopMax + 1), d->top_a + kTopCap, d->top_b + kTopCap,
                                    d->top_ci + kTopCap, d->ddi, c->info, 1);
  BH_HIP(c, hipGetLastError());
  BH_HIP(c, hipEventRecord(d->ev_top1, so));
  d->top_early = true;
  return BH_OK;
}

// Own pass of the two-pass force, on the side stream: it needs only the gathered piece descriptors and
// the local tree, so it runs while the X4 exchange occupies the main stream.  Top tree of this pass: other
// ranks' pieces are null records, top cells carry this rank's share of their mass (bh_dd_top builds the mirror
// image for the remote pass).  Called on its own, the pass is launched at once; bh_dd_phase_let puts it behind
// the LET export.
int bh_dd_force_local(bh_ctx* c, const void* gathered_x3) {
  if (!c || !c->dd || !gathered_x3) return BH_ERR_BAD_ARG;
  if (!(c->stage & BH_ST_COM)) return BH_ERR_ORDER;
  const int s = dd_force_local_prepare(c, gathered_x3);
  return s ? s : dd_force_local_launch(c, false);
}

// measurement: one wave that does nothing for `ticks` of the 100 MHz wall clock — the main stream is busy and the GPU
// as free as during an exchange over the links (bh_dd_idle_wave; study builds of bh_group.hip: BH_DD_FAKE_X4_US=<us>
// puts it behind every X4, tools/r5_fake_x4.sh)
__global__ void dd_sleep_kernel(long long ticks) {
  const long long t0 = (long long)wall_clock64();
  while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

int bh_dd_top(bh_ctx* c, const void* gathered_x3, int stride) {
  if (!c || !c->dd || !gathered_x3) return BH_ERR_BAD_ARG;
  if (!(c->stage & BH_ST_COM)) return BH_ERR_ORDER;
  bh_dd_state* d = c->dd;
  if (stride < kSegBlocks0 || stride > d->let_cap || (stride & 1)) return BH_ERR_BAD_ARG;
  // the segment headers (records each rank needed) go to the host for bh_dd_let_check: written to pinned memory by
  // the validation kernel's first block, announced by a sequence number in d->host[70]
  {
    const long long recs = (long long)d->world * stride;
    d->let_seq++;
    dd_validate_kernel<<<(unsigned)((recs + 255) / 256), 256, 0, c->stream>>>(d->pool, d->seg_base, stride, d->world,
                                                                            d->rank, c->info, d->host_rows,
                                                                            d->host + 70, d->let_seq, d->prev_rows_dev,
                                                                            d->x4_v ? 1 : 0);
    BH_HIP(c, hipGetLastError());
  }
  d->let_copy_pending = true;
  if (d->split) {  // same structure as the own pass's tree: re-emit the records only
    BH_HIP(c, hipStreamWaitEvent(c->stream, d->ev_top1, 0));
    dd_top_emit_kernel<<<(kTopCap + 255) / 256, 256, 0, c->stream>>>(
        (const bh_dd_piece*)gathered_x3, d->rank, d->pool, d->top_base, d->seg_base, stride, c->bounds, c->p.G,
        c->p.theta, d->top_ps + (kTopMax + 1), d->top_a + kTopCap, d->top_b + kTopCap,
        d->top_ci + kTopCap, d->ddi, 2);
    if (d->own_hi < c->n)  // the bodies beyond own_hi walk the whole stitched tree: every piece real
      dd_top_emit_kernel<<<(kTopCap + 255) / 256, 256, 0, c->stream>>>(
          (const bh_dd_piece*)gathered_x3, d->rank, d->pool, d->top_base3, d->seg_base, stride, c->bounds, c->p.G,
          c->p.theta, d->top_ps + (kTopMax + 1), d->top_a + kTopCap, d->top_b + kTopCap,
          d->top_ci + kTopCap, d->ddi, 0);
  } else if (d->top_early) {  // one pass: structure built beside the LET kernels (dd_top_early), every piece real
    d->top_early = false;
    BH_HIP(c, hipStreamWaitEvent(c->stream, d->ev_top1, 0));
    dd_top_emit_kernel<<<(kTopCap + 255) / 256, 256, 0, c->stream>>>(
        (const bh_dd_piece*)gathered_x3, d->rank, d->pool, d->top_base, d->seg_base, stride, c->bounds, c->p.G,
        c->p.theta, d->top_ps + (kTopMax + 1), d->top_a + kTopCap, d->top_b + kTopCap,
        d->top_ci + kTopCap, d->ddi, 0);
  } else {
    dd_top_kernel<<<1, 1024, 0, c->stream>>>((const bh_dd_piece*)gathered_x3, d->world, d->rank, d->pool,
                                             d->top_base, d->seg_base, stride, c->bounds, c->p.G, c->p.theta,
                                             d->top_ps, d->top_a, d->top_b, d->top_ci, d->ddi, c->info, 0);
  }
  BH_HIP(c, hipGetLastError());
  return BH_OK;
}

// allow_fuse: the caller knows that this pass will not be repeated (bh_dd_phase_force: every rank's LET fitted)
static int dd_force_impl(bh_ctx* c, bool allow_fuse) {
  if (!c || !c->dd) return BH_ERR_BAD_ARG;
  if (!(c->stage & BH_ST_COM)) return BH_ERR_ORDER;
  bh_dd_state* d = c->dd;
  // The last pass of the step also integrates the bodies and folds this rank's min / max (large local body counts:
  // force_mixed_kernel FUSE) — in the two-pass form it then starts after the own pass, whose accelerations it adds.
  bool fused = false;
  if (d->split && d->own_hi < c->n) {
    // partial two-pass step: the remote pass of bodies [0, own_hi) follows their own pass on the side stream (low
    // priority: its workgroups fill what the other launch leaves), the one pass of the rest runs on the main stream;
    // both integrate their bodies and share one min / max fold (fold_groups)
    const int groups = (c->n + 63) / 64;
    hipStream_t so = d->serial ? c->stream : d->stream_own;
    bool f1 = false, f2 = false;
    BH_HIP(c, hipEventRecord(d->ev_x4, c->stream));  // X4, the validation and both top trees are complete here
    BH_HIP(c, hipStreamWaitEvent(so, d->ev_x4, 0));
    if (d->own_hi > 0)
      BH_HIP(c, bhk_force_root(c, 0, d->own_hi, d->top_base, so, d->acc2, c->acc, allow_fuse, &f1, groups));
    BH_HIP(c, hipEventRecord(d->ev_own, so));
    BH_HIP(c, bhk_force_root(c, d->own_hi, c->n, d->top_base3, c->stream, c->acc, nullptr,
                             allow_fuse && (f1 || d->own_hi == 0), &f2, groups));
    BH_HIP(c, hipStreamWaitEvent(c->stream, d->ev_own, 0));
    // (both launches take the fused instance or neither: the same conditions decide — a mixed outcome would leave
    // some bodies integrated and the others not)
    if (d->own_hi > 0 && f1 != f2) return BH_ERR_BAD_ARG;
    fused = f2;
    c->acc2 = d->acc2;
  } else if (d->split) {  // remote pass -> acc2; acc + acc2 is integrated once the own pass has finished too
    BH_HIP(c, hipStreamWaitEvent(c->stream, d->ev_own, 0));
    BH_HIP(c, bhk_force_root(c, 0, c->n, d->top_base, c->stream, d->acc2, c->acc, allow_fuse, &fused));
    c->acc2 = d->acc2;
  } else {
    BH_HIP(c, bhk_force_root(c, 0, c->n, d->top_base, c->stream, c->acc, nullptr, allow_fuse, &fused));
    c->acc2 = nullptr;
  }
  c->dd_integrated = fused;
  c->stage |= BH_ST_FORCE;
  c->ever |= BH_ST_FORCE;
  return BH_OK;
}

int bh_dd_force(bh_ctx* c) { return dd_force_impl(c, false); }

int bh_dd_let_check(bh_ctx* c, int stride, int32_t* counts) {
  if (!c || !c->dd) return BH_ERR_BAD_ARG;
  bh_dd_state* d = c->dd;
  if (!d->let_copy_pending) return BH_ERR_ORDER;
  // (polled pinned word, as the migration's result: a blocking wait wakes the host tens of microseconds late, and this
  // wait sits between X4 and the force launch; bounded, then the stream wait decides)
  {
    volatile int* hs = d->host + 70;
    bool seen = false;
    for (long spin = 0; spin < 400000000L; spin++) {
      if (__atomic_load_n(hs, __ATOMIC_ACQUIRE) == d->let_seq) {
        seen = true;
        break;
      }
      if ((spin & 0xfffff) == 0xfffff && hipStreamQuery(c->stream) != hipErrorNotReady) break;
    }
    if (!seen) {
      BH_HIP(c, hipStreamSynchronize(c->stream));
      if (__atomic_load_n(hs, __ATOMIC_ACQUIRE) != d->let_seq) return BH_ERR_HIP;
    }
  }
  d->let_copy_pending = false;
  // counts[q] = the most records rank q needed for any receiver (its needs row: every rank holds the same matrix,
  // so every rank takes the same decision); a negative header marks a rank that left the step
  int worst = 0;
  bool over = false;  // (a size per pair: some pair outgrew its size — dd_validate_kernel takes the same decision)
  for (int q = 0; q < d->world; q++) {
    const int* seg = d->host_rows + 32 * q;
    int need = seg[10];  // header record 0, field `first`
    if (need >= 0)
      for (int j = 0; j < d->world; j++) {
        const int nj = seg[seg_row_dword(j)];
        need = nj > need ? nj : need;
        if (d->x4_v && j != q) over = over || nj > x4_chunk(d->prev_rows[32 * q + seg_row_dword(j)], stride);
      }
    if (counts) counts[q] = need;
    if (need > worst) worst = need;
  }
  return (worst > stride || over) ? BH_ERR_SMALL_BUFFER : BH_OK;
}

// The sizes of this try's X4 (bh_group.hip, before the exchange).  allow_prev: a size per pair from the last fitting
// exchange's needs if there is one (first try of a step); else — a repeated exchange, the first step, more ranks than a
// needs row holds — the whole slot for every pair.  send_bytes[q] / recv_bytes[q]: what this rank sends to / receives
// from rank q.  The validation and the fit decision of this try follow what is chosen here.
int bh_dd_x4_sizes(bh_ctx* c, int stride, int allow_prev, int64_t* send_bytes, int64_t* recv_bytes) {
  if (!c || !c->dd || !send_bytes || !recv_bytes) return BH_ERR_BAD_ARG;
  bh_dd_state* d = c->dd;
  if (stride < kSegBlocks0 || stride > d->let_cap || (stride & 1)) return BH_ERR_BAD_ARG;
  d->x4_v = allow_prev && d->prev_ok && d->world <= kNeedsRowMax && d->let_mode == 1;
  for (int q = 0; q < d->world; q++) {
    int s = stride, r = stride;
    if (d->x4_v) {
      s = x4_chunk(d->prev_rows[32 * d->rank + seg_row_dword(q)], stride);
      r = x4_chunk(d->prev_rows[32 * q + seg_row_dword(d->rank)], stride);
    }
    send_bytes[q] = (int64_t)s * 32;
    recv_bytes[q] = (int64_t)r * 32;
  }
  return BH_OK;
}

// ---- one entry point per phase group: the per-step protocol of dist.DomainStepper in five calls (each is the
// plain sequence of the fine-grained entry points above, which stay for tests and for the failure paths)
int bh_dd_phase_migrate(bh_ctx* c, const void* gathered_x1, void* send_x2, int limit) {
  const int s = bh_dd_cube_apply(c, gathered_x1);
  return s ? s : bh_dd_migrate_pack(c, send_x2, limit);
}

int bh_dd_phase_tree(bh_ctx* c, const void* gathered_x2, int limit, void* send_x3, int* n_loc, int* more,
                     int* most) {
  int m = 0;
  const int s = bh_dd_migrate_apply(c, gathered_x2, limit, n_loc, &m, most);
  if (more) *more = m;
  if (s || m) return s;  // another migration round first
  return bh_dd_tree(c, send_x3);
}

int bh_dd_phase_let(bh_ctx* c, const void* gathered_x3, void* send_x4, int stride, int own_pass) {
  if (!c || !c->dd || !gathered_x3 || !send_x4) return BH_ERR_BAD_ARG;
  if (!(c->stage & BH_ST_COM)) return BH_ERR_ORDER;
  bool after_let = true;
#ifdef BH_STUDY
  if (getenv("BH_DD_OWN_AT_ONCE")) after_let = false;  // round 4 / early round 5: the own pass beside the LET kernels
#endif
  if (own_pass) {
    int s = dd_force_local_prepare(c, gathered_x3);
    if (!s && !after_let) s = dd_force_local_launch(c, false);
    if (s) return s;
  } else if (!c->dd->split) {
    const int s = dd_top_early(c, gathered_x3);
    if (s) return s;
  }
  int s = bh_dd_let_pack(c, gathered_x3, send_x4, stride);
  if (!s && own_pass && after_let) s = dd_force_local_launch(c, true);
  return s;
}

// The fit of every rank's LET is looked at BEFORE the last force pass is launched (the headers of the received
// segments are on the host as soon as X4 has arrived: the wait is hidden behind the own-pieces pass): a step whose X4
// has to be repeated larger launches nothing, and a pass that will not be repeated may integrate (dd_force_impl).
int bh_dd_phase_force(bh_ctx* c, const void* gathered_x3, int stride, int32_t* counts, int* fits) {
  int s = bh_dd_top(c, gathered_x3, stride);
  if (s) return s;
  int32_t own_counts[64];  // a caller that does not ask for the counts must still not walk a departed rank's LET
  if (!counts) counts = own_counts;
  // The force launches go out BEFORE the host has looked at the X4 headers: the validation kernel takes the same
  // decision on the device (bh_devinfo.dd_hold) and a launch behind an exchange that has to be repeated — or that a rank
  // has left — returns at once.  The host's look (a polled pinned word) then costs the step nothing.
  // (one rank in two passes: the remote pass walks nothing but the top record, and a launch that short integrates
  // slower than the streaming kernel does — 0.100 against 0.055 + 0.021 ms at 1M bodies; one pass integrates as
  // bh_step's launch does)
  const int stage0 = c->stage;
  s = dd_force_impl(c, !c->dd->replay && (c->dd->world > 1 || !c->dd->split));
  if (s) return s;
  s = bh_dd_let_check(c, stride, counts);
  bool left = false;
  if (s == BH_OK || s == BH_ERR_SMALL_BUFFER)
    for (int q = 0; q < c->dd->world; q++) left = left || counts[q] < 0;
  if (s != BH_OK || left) {  // the launches held themselves back: nothing was walked, nothing integrated
    c->dd_integrated = false;
    c->stage = stage0;
  } else if (!c->dd->replay) {  // a fitting exchange: its needs size the next step's (dd_validate_kernel kept the device's copy)
    memcpy(c->dd->prev_rows, c->dd->host_rows, (size_t)c->dd->world * 32 * sizeof(int));
    c->dd->prev_ok = true;
  }
  if (fits) *fits = (s == BH_OK && !left) ? 1 : 0;
  if (s == BH_ERR_SMALL_BUFFER && !left) return BH_OK;  // the caller repeats X4 with a larger stride
  if (s != BH_OK && s != BH_ERR_SMALL_BUFFER) return s;
  if (left) return counts == own_counts ? BH_ERR_DOMAIN_LEFT : BH_OK;  // the caller raises on every rank
  return BH_OK;
}

int bh_dd_phase_end(bh_ctx* c, void* send_x1) {
  const int s = bh_integrate(c);  // (a no-op on the device when the last force pass integrated: bh_dd_force)
  return s ? s : bh_dd_cube_pack(c, send_x1);
}

// what the step's migration did, for logs and tests (synchronises): out[0] bodies this rank holds, [1] emigrants it
// found in its last classification, [2] steps since bh_dd_init in which the domain boundaries moved, [3] what the last
// step did with them (0 kept, 1 exact quantiles proposed by the ranks, 2 sample quantiles), [4..7] reserved
int bh_dd_get_info(bh_ctx* c, int32_t out[8]) {
  if (!c || !c->dd || !out) return BH_ERR_BAD_ARG;
  int h[16];
  BH_HIP(c, hipMemcpyAsync(h, c->dd->ddi, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  BH_HIP(c, hipStreamSynchronize(c->stream));
  out[0] = c->n; out[1] = h[12]; out[2] = h[9]; out[3] = h[11];
  out[4] = out[5] = out[6] = out[7] = 0;
  return BH_OK;
}

// Measurement only: the counted walk (bh_force_walk_stats) of this rank's bodies over the stitched pool as the last
// step left it — which = 0: the top tree of the last (or only) force pass, 1: that of the own-pieces pass (two-pass
// steps only).  The bodies have moved by one step since that tree was built: counts, not forces.
int bh_dd_walk_stats(bh_ctx* c, int which, bh_walk_stats* out) {
  if (!c || !c->dd || !out || which < 0 || which > 1) return BH_ERR_BAD_ARG;
  if (!(c->ever & BH_ST_FORCE)) return BH_ERR_ORDER;
  if (which == 1 && !c->dd->split) return BH_ERR_ORDER;
  return bh_walk_stats_from(c, which ? c->dd->top_base2 : c->dd->top_base, out);
}

// Measurement only (two-pass steps): the force passes of this rank re-launched over the pool as the last step left it,
// with nothing else on the GPU (the second of two rounds: caches warm).  Full two-pass step: ms[0] the own-pieces pass
// alone, ms[1] the remote pass alone, ms[2] both at once (own pass on the side stream, remote pass on the main stream:
// first launch to last completion), ms[3] = 0.  Partial step (split_pct < 100): ms[0] own pass of the split part,
// ms[1] its remote pass, ms[2] the one pass of the other bodies, ms[3] those two at once on two streams, as the step
// runs them.  Overwrites the partial accelerations; the bodies are not touched.
int bh_dd_pass_times(bh_ctx* c, float ms[4]) {
  if (!c || !c->dd || !ms) return BH_ERR_BAD_ARG;
  bh_dd_state* d = c->dd;
  if (!(c->ever & BH_ST_FORCE) || !d->split) return BH_ERR_ORDER;
  BH_HIP(c, hipSetDevice(c->device));
  hipEvent_t e[5];
  for (int k = 0; k < 5; k++) BH_HIP(c, hipEventCreate(&e[k]));
  BH_HIP(c, hipStreamSynchronize(c->stream));
  BH_HIP(c, hipStreamSynchronize(d->stream_own));
  const bool partial = d->own_hi < c->n;
  const int oh = partial ? d->own_hi : c->n;
  ms[0] = ms[1] = ms[2] = ms[3] = 0.0f;
  for (int rep = 0; rep < 2; rep++) {
    BH_HIP(c, hipEventRecord(e[0], c->stream));
    if (oh > 0) BH_HIP(c, bhk_force_root(c, 0, oh, d->top_base2, c->stream, c->acc));
    BH_HIP(c, hipEventRecord(e[1], c->stream));
    if (oh > 0) BH_HIP(c, bhk_force_root(c, 0, oh, d->top_base, c->stream, d->acc2));
    BH_HIP(c, hipEventRecord(e[2], c->stream));
    if (partial) BH_HIP(c, bhk_force_root(c, oh, c->n, d->top_base3, c->stream, c->acc));
    BH_HIP(c, hipEventRecord(e[3], c->stream));
    BH_HIP(c, hipStreamSynchronize(c->stream));
  }
  BH_HIP(c, hipEventElapsedTime(&ms[0], e[0], e[1]));
  BH_HIP(c, hipEventElapsedTime(&ms[1], e[1], e[2]));
  if (partial) BH_HIP(c, hipEventElapsedTime(&ms[2], e[2], e[3]));
  // two launches at once: the first on the side stream, the second on the main stream
  BH_HIP(c, hipEventRecord(e[0], c->stream));
  BH_HIP(c, hipStreamWaitEvent(d->stream_own, e[0], 0));
  if (partial) {
    if (oh > 0) BH_HIP(c, bhk_force_root(c, 0, oh, d->top_base, d->stream_own, d->acc2));
    BH_HIP(c, hipEventRecord(e[4], d->stream_own));
    BH_HIP(c, bhk_force_root(c, oh, c->n, d->top_base3, c->stream, c->acc));
  } else {
    BH_HIP(c, bhk_force_root(c, 0, c->n, d->top_base2, d->stream_own, c->acc));
    BH_HIP(c, hipEventRecord(e[4], d->stream_own));
    BH_HIP(c, bhk_force_root(c, 0, c->n, d->top_base, c->stream, d->acc2));
  }
  BH_HIP(c, hipStreamWaitEvent(c->stream, e[4], 0));
  BH_HIP(c, hipEventRecord(e[1], c->stream));
  BH_HIP(c, hipStreamSynchronize(c->stream));
  BH_HIP(c, hipEventElapsedTime(&ms[partial ? 3 : 2], e[0], e[1]));
  for (int k = 0; k < 5; k++) (void)hipEventDestroy(e[k]);
  return BH_OK;
}

// ---- measurement: the force phase of the last completed step run again (bh_rank_replay_force_phase)
int bh_dd_replay_begin(bh_ctx* c, int split, int split_pct, int saved[4]) {
  if (!c || !c->dd || !saved || (split && (split_pct < 1 || split_pct > 100))) return BH_ERR_BAD_ARG;
  if (!(c->ever & BH_ST_FORCE)) return BH_ERR_ORDER;  // no step yet: no tree, no imported segments
  bh_dd_state* d = c->dd;
  BH_HIP(c, hipSetDevice(c->device));
  BH_HIP(c, hipStreamSynchronize(c->stream));
  BH_HIP(c, hipStreamSynchronize(d->stream_own));
  saved[0] = c->stage;
  saved[1] = (d->split ? 1 : 0) | (c->acc2 ? 2 : 0);
  saved[2] = d->split_pct;
  saved[3] = d->serial ? 1 : 0;
  // the tree, the digests and the imported segments of the last step are all still there; the bodies have been
  // integrated since (same order, positions one step on): the passes do the same work
  c->stage |= BH_ST_BBOX | BH_ST_MORTON | BH_ST_SORT | BH_ST_BUILD | BH_ST_COM;
  d->split = split != 0;
  if (split) d->split_pct = split_pct;
  d->serial = false;  // the rank's own two streams, as on a GPU of its own
  d->replay = true;
  d->x4_v = false;  // (nothing travels: what the real step received is in the pool, validated as whole slots)
  return BH_OK;
}
int bh_dd_replay_end(bh_ctx* c, const int saved[4]) {
  if (!c || !c->dd || !saved) return BH_ERR_BAD_ARG;
  bh_dd_state* d = c->dd;
  BH_HIP(c, hipStreamSynchronize(c->stream));
  BH_HIP(c, hipStreamSynchronize(d->stream_own));
  c->stage = saved[0];
  d->split = (saved[1] & 1) != 0;
  c->acc2 = (saved[1] & 2) ? d->acc2 : nullptr;
  d->split_pct = saved[2];
  d->serial = saved[3] != 0;
  d->replay = false;
  c->dd_integrated = false;
  return BH_OK;
}
// a rank that has stepped in two passes goes back to one (the adaptive form of bh_rank: bh_group.hip)
int bh_dd_set_one_pass(bh_ctx* c) {
  if (!c || !c->dd) return BH_ERR_BAD_ARG;
  c->dd->split = false;
  return BH_OK;
}
// measurement: the needs matrix of the last X4 as this rank saw it in the segment headers — out[q * world + j] =
// records rank q needed in its segment for receiver j (0 on the diagonal) — the same on every rank
int bh_dd_needs_matrix(bh_ctx* c, int32_t* out) {
  if (!c || !c->dd || !out) return BH_ERR_BAD_ARG;
  const bh_dd_state* d = c->dd;
  if (!(c->ever & BH_ST_FORCE)) return BH_ERR_ORDER;
  BH_HIP(c, hipStreamSynchronize(c->stream));
  for (int q = 0; q < d->world; q++)
    for (int j = 0; j < d->world; j++) out[q * d->world + j] = j < 24 ? d->host_rows[32 * q + seg_row_dword(j)] : 0;
  return BH_OK;
}
int bh_dd_idle_wave(bh_ctx* c, int us) {
  if (!c || us < 0) return BH_ERR_BAD_ARG;
  if (us > 0) dd_sleep_kernel<<<1, 64, 0, c->stream>>>(100ll * us);
  BH_HIP(c, hipGetLastError());
  return BH_OK;
}

int bh_dd_download(bh_ctx* c, float* posm, float* velid, float* acc) {
  if (!c || !c->dd) return BH_ERR_BAD_ARG;
  const size_t n = (size_t)c->n, nb = n * sizeof(float4);
  if (posm) BH_HIP(c, hipMemcpyAsync(posm, c->posm[c->cur], nb, hipMemcpyDeviceToHost, c->stream));
  if (velid) BH_HIP(c, hipMemcpyAsync(velid, c->velid[c->cur], nb, hipMemcpyDeviceToHost, c->stream));
  if (acc) BH_HIP(c, hipMemcpyAsync(acc, c->acc, nb, hipMemcpyDeviceToHost, c->stream));
  BH_HIP(c, hipStreamSynchronize(c->stream));
  if (acc && c->acc2) {  // two-pass force: own pass + remote pass
    float* tmp = (float*)malloc(nb);
    if (!tmp) return BH_ERR_OOM;
    hipError_t e = hipMemcpy(tmp, c->acc2, nb, hipMemcpyDeviceToHost);
    if (e == hipSuccess)
      for (size_t i = 0; i < 4 * n; i++) acc[i] += tmp[i];
    free(tmp);
    if (e != hipSuccess) {
      c->last_hip = (int)e;
      return BH_ERR_HIP;
    }
  }
  return BH_OK;
}

}  // extern "C"
