// bh_tree.hip — bounding cube, Morton keys, octree topology and centres of mass.
//
// Reference stages (nbody_v5_bench.cu): computeBoundingBoxKernel :134-156 (one GPU thread),
// computeMortonCodesKernel :51-63, memset + initRoot + insertParticles :266-275 (a host loop
// of N/1024 launches of a racy atomicCAS insertion), computeCOM + finalizeCOM :158-189
// (4N float atomics on the root).  None of that structure is kept:
//
//  * bbox   : grid-stride float4 loads, wave64 shuffle min/max, LDS across the 4 waves,
//             1024 partials folded by a second one-block launch.  min/max are exact, so the
//             cube equals the reference's serial loop bit for bit.
//  * build  : the (path-compressed) canonical octree is a pure function of the SORTED keys, so
//             every internal cell is found independently — no levels, no allocation atomics, no
//             races: one thread per adjacent key pair, binary searches on the key digits, one
//             exclusive scan of the child counts to place each cell's children in one contiguous
//             block of 32-byte records (details above pairs_kernel).
//  * COM    : one fp64 exclusive prefix scan of (m, m x, m y, m z) over the sorted bodies;
//             a cell's sums are P[hi] - P[lo], rounded once to fp32.  Deterministic, no atomics,
//             more accurate than any fp32 summation order (the reference's is non-deterministic).
#include "bh_internal.h"
#include "bh_scan_body.h"
#include "bh_keys.h"

namespace {

// ------------------------------------------------------------------ bbox
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
  return v;
}

__device__ __forceinline__ void block_minmax(float mn[3], float mx[3], float* lds /* [waves][6] */) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    mn[k] = wave_min(mn[k]);
    mx[k] = wave_max(mx[k]);
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      lds[w * 6 + k] = mn[k];
      lds[w * 6 + 3 + k] = mx[k];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = (int)(blockDim.x >> 6);
    for (int q = 1; q < nw; q++)
#pragma unroll
      for (int k = 0; k < 3; k++) {
        mn[k] = fminf(mn[k], lds[q * 6 + k]);
        mx[k] = fmaxf(mx[k], lds[q * 6 + 3 + k]);
      }
  }
}

// the cube of ref:148-154 from the folded min / max
__device__ __forceinline__ void write_cube(const float mn[3], const float mx[3], float* __restrict__ bounds) {
  const float size = fmaxf(mx[0] - mn[0], fmaxf(mx[1] - mn[1], mx[2] - mn[2]));  // ref:148
  bounds[0] = mn[0];
  bounds[1] = mn[1];
  bounds[2] = mn[2];
  bounds[3] = mn[0] + size;  // ref:152-154: cube anchored at the min corner
  bounds[4] = mn[1] + size;
  bounds[5] = mn[2] + size;
  bounds[6] = fmaxf(bounds[3] - bounds[0], 1.0f);  // root edge s0, ref:55
  bounds[7] = 0.0f;
}

__global__ __launch_bounds__(256) void bbox_partial_kernel(const float4* __restrict__ posm, int n,
                                                           float* __restrict__ partial) {
  __shared__ float lds[24];
  float mn[3] = {1e10f, 1e10f, 1e10f};  // sentinels ref:138
  float mx[3] = {-1e10f, -1e10f, -1e10f};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4 q = posm[i];
    mn[0] = fminf(mn[0], q.x);
    mn[1] = fminf(mn[1], q.y);
    mn[2] = fminf(mn[2], q.z);
    mx[0] = fmaxf(mx[0], q.x);
    mx[1] = fmaxf(mx[1], q.y);
    mx[2] = fmaxf(mx[2], q.z);
  }
  block_minmax(mn, mx, lds);
  if (threadIdx.x == 0) {
    float* o = partial + blockIdx.x * 6;
    o[0] = mn[0]; o[1] = mn[1]; o[2] = mn[2];
    o[3] = mx[0]; o[4] = mx[1]; o[5] = mx[2];
  }
}

// rows of 6 floats (min xyz, max xyz) `stride` floats apart; raw != 0: write the folded min/max
// only (6 floats), else the cube of ref:148-154
__global__ __launch_bounds__(256) void bbox_final_kernel(const float* __restrict__ partial, int nparts,
                                                         int stride, int raw,
                                                         float* __restrict__ bounds) {
  __shared__ float lds[24];
  float mn[3] = {1e10f, 1e10f, 1e10f};
  float mx[3] = {-1e10f, -1e10f, -1e10f};
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
    const float* o = partial + (size_t)i * stride;
    mn[0] = fminf(mn[0], o[0]); mn[1] = fminf(mn[1], o[1]); mn[2] = fminf(mn[2], o[2]);
    mx[0] = fmaxf(mx[0], o[3]); mx[1] = fmaxf(mx[1], o[4]); mx[2] = fmaxf(mx[2], o[5]);
  }
  block_minmax(mn, mx, lds);
  if (threadIdx.x == 0 && raw) {
    bounds[0] = mn[0]; bounds[1] = mn[1]; bounds[2] = mn[2];
    bounds[3] = mx[0]; bounds[4] = mx[1]; bounds[5] = mx[2];
  } else if (threadIdx.x == 0) {
    write_cube(mn, mx, bounds);
  }
}

// ------------------------------------------------------------------ integrate (+ the next step's bbox)
// ref:227-249, source text, no contraction: v += a dt; clamp |v| to max_speed; p += v dt.
// BBOX: the kernel also folds the min/max of the positions it writes — exactly the input of the next step's
// bounding cube (ref:134-156) — into one row per block, and the block that finishes last folds the rows into
// `bounds_next`: a step that follows a step needs no bbox kernels.  min/max are exact, so the cube is the
// same bit for bit.
// TILE = bodies per block: BH_INTEGRATE_TILE (4 per thread), or 1024 for small n (65,536 bodies: 64 blocks, not 16)
template <bool BBOX, int TILE = BH_INTEGRATE_TILE>
__global__ __launch_bounds__(1024) void integrate_kernel(float4* __restrict__ posm,
                                                         float4* __restrict__ velid,
                                                         const float4* __restrict__ acc,
                                                         const float4* __restrict__ acc2, int n, float DT,
                                                         float MAX_SPEED, float* __restrict__ rows,
                                                         u32* __restrict__ done_count,
                                                         float* __restrict__ bounds_next, int raw) {
  float mn[3] = {1e10f, 1e10f, 1e10f};  // sentinels ref:138
  float mx[3] = {-1e10f, -1e10f, -1e10f};
  constexpr int kPer = TILE / 1024;
  float4 p[kPer], v[kPer], a[kPer];
#pragma unroll
  for (int r = 0; r < kPer; r++) {
    const int i = blockIdx.x * TILE + r * 1024 + (int)threadIdx.x;
    if (i < n) {
      p[r] = posm[i];
      v[r] = velid[i];
      a[r] = acc[i];
      if (acc2) {  // two-pass force of the domain-decomposed step: own pass + remote pass
        const float4 b = acc2[i];
        a[r].x += b.x; a[r].y += b.y; a[r].z += b.z;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < kPer; r++) {
    const int i = blockIdx.x * TILE + r * 1024 + (int)threadIdx.x;
    if (i < n) {
      float vx = v[r].x + a[r].x * DT;
      float vy = v[r].y + a[r].y * DT;
      float vz = v[r].z + a[r].z * DT;
      const float speedSq = vx * vx + vy * vy + vz * vz;
      if (speedSq > MAX_SPEED * MAX_SPEED) {
        const float scale = MAX_SPEED / sqrtf(speedSq);
        vx *= scale;
        vy *= scale;
        vz *= scale;
      }
      v[r].x = vx; v[r].y = vy; v[r].z = vz;
      p[r].x += vx * DT;
      p[r].y += vy * DT;
      p[r].z += vz * DT;
      posm[i] = p[r];
      velid[i] = v[r];
      if (BBOX) {
        mn[0] = fminf(mn[0], p[r].x); mn[1] = fminf(mn[1], p[r].y); mn[2] = fminf(mn[2], p[r].z);
        mx[0] = fmaxf(mx[0], p[r].x); mx[1] = fmaxf(mx[1], p[r].y); mx[2] = fmaxf(mx[2], p[r].z);
      }
    }
  }
  if (!BBOX) return;
  __shared__ float lds[16 * 6];
  __shared__ int s_last;
  block_minmax(mn, mx, lds);
  if (threadIdx.x == 0) {
    float* o = rows + (size_t)blockIdx.x * 6;
#pragma unroll
    for (int q = 0; q < 3; q++) {
      bh_publish_f32(o + q, mn[q]);
      bh_publish_f32(o + 3 + q, mx[q]);
    }
    bh_published();  // (bh_internal.h: last-block hand-off without a fence)
    s_last = bh_last_block(done_count, (int)blockIdx.x, (int)gridDim.x) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  float fn[3] = {1e10f, 1e10f, 1e10f};
  float fx[3] = {-1e10f, -1e10f, -1e10f};
  for (int r = threadIdx.x; r < (int)gridDim.x; r += blockDim.x) {
    const float* o = rows + (size_t)r * 6;
#pragma unroll
    for (int q = 0; q < 3; q++) {
      fn[q] = fminf(fn[q], bh_collect_f32(o + q));
      fx[q] = fmaxf(fx[q], bh_collect_f32(o + 3 + q));
    }
  }
  __syncthreads();  // lds is reused
  block_minmax(fn, fx, lds);
  if (threadIdx.x == 0) {
    if (raw) {  // domain-decomposed step: this rank's min / max go into the X1 exchange, the cube comes after it
      bounds_next[0] = fn[0]; bounds_next[1] = fn[1]; bounds_next[2] = fn[2];
      bounds_next[3] = fx[0]; bounds_next[4] = fx[1]; bounds_next[5] = fx[2];
    } else {
      write_cube(fn, fx, bounds_next);
    }
  }
}

// ------------------------------------------------------------------ keys
template <int B>
__global__ __launch_bounds__(256) void keys_kernel(const float4* __restrict__ posm,
                                                   const float* __restrict__ bounds, int n, int curve,
                                                   u64* __restrict__ keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float minX = bounds[0], minY = bounds[1], minZ = bounds[2];
  const float size = bounds[6];  // fmaxf(bounds[3]-bounds[0], 1) ref:55
  const float4 q = posm[i];
  const u64 k = body_key<B>(curve, q.x, q.y, q.z, minX, minY, minZ, size);
  keys[i] = k;
}

// ------------------------------------------------------------------ build
// Path-compressed canonical octree from the sorted keys.
//
// Canonical rule (reference intent, nbody_v5_bench.cu:83-132 with D3-D5 removed): a cell is
// subdivided iff it holds more than leaf_cap bodies and its level < max_depth.  A cell with a
// single non-empty octant has the same (mass, COM) as that child and a larger edge, so under the
// MAC `s/dist < theta` it is accepted only if the child would be too: dropping such chain cells
// leaves every body's set of accepted cells/bodies unchanged (bit-identical sums in pre-order;
// tests/test_oracle.py checks it on the oracle) and bounds the tree: internal cells <= n-1,
// records <= 2n, whatever the input.  (The
// reference's 2N-node pool overflows silently on close pairs)
//
// Every emitted internal cell BRANCHES (>= 2 non-empty octants), so it has a first child
// boundary j: keys j-1 and j share exactly L = level digits.  Conversely each adjacent pair
// (j-1, j) with d = common digits < D names the cell at level d containing both.  Thread j
//   - finds that cell's start a (nearest i < j with d[i] < d[j]),
//   - is the cell's representative iff no boundary of the same level lies between a and j,
//   - if so finds the end b, counts the <= 8 children, and records nchild[j]
// (pairs_kernel below: LDS-window bitmask queries; key searches only for cells wider than the window).
// One exclusive scan of nchild[] places each cell's children in one contiguous block; the
// representative index j doubles as the cell id, so no compaction pass is needed.
// Sampled lower bound for the wide-cell searches: first index i in [0, n] with (k[i] >> sh) >= T.
// `samp` (in LDS) holds every (1 << ss)-th key, so the first ~11 bisection steps cost LDS reads and
// the dependent GLOBAL loads drop from ~log2(n) to ~ss, all inside one 2^ss-key (32 KiB) span.
constexpr int kSampMax = 2048;
__device__ __forceinline__ int key_lower_bound(const u64* __restrict__ k, int n, const u64* samp, int ns,
                                               int ss, int sh, u64 T) {
  int l = 0, h = ns;  // first sample whose prefix is >= T
  while (l < h) {
    const int mid = l + ((h - l) >> 1);
    if (key_prefix(samp[mid], sh) < T) l = mid + 1; else h = mid;
  }
  if (l == 0) return 0;
  int lo = ((l - 1) << ss) + 1;  // k[(l-1) << ss] < T
  int hi = min(l << ss, n);      // k[l << ss] >= T, or past the end
  // invariant: the answer is in [lo, hi].  16-ary rounds: 16 independent probes in flight per
  // lane, so a 4096-key span costs 3 memory round trips instead of 12 dependent ones (measured:
  // each dependent random probe of the key array costs ~2.5 us here).
  while (hi - lo > 16) {
    const int step = (hi - lo + 15) >> 4;
    u64 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = k[min(lo + i * step + step - 1, hi - 1)];
    int first = 16;
#pragma unroll
    for (int i = 15; i >= 0; i--)
      if (key_prefix(v[i], sh) >= T) first = i;
    if (first == 16) {  // every probe (the last one is hi-1) is below T
      lo = hi;
      break;
    }
    const int nlo = (first == 0) ? lo : min(lo + (first - 1) * step + step - 1, hi - 1) + 1;
    hi = min(lo + first * step + step - 1, hi - 1);
    lo = nlo;
  }
  if (hi > lo) {  // at most 16 keys left: one more round trip
    u64 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = k[min(lo + i, hi - 1)];
    int first = hi - lo;
#pragma unroll
    for (int i = 15; i >= 0; i--)
      if (i < hi - lo && key_prefix(v[i], sh) >= T) first = i;
    lo += first;
  }
  return lo;
}

// child boundaries of cell [a,b) at the level whose digit shift is dsh:
// pos[v] = first j in [a,b) with digit(k[j]) >= v
__device__ __forceinline__ void child_bounds(const u64* __restrict__ k, int a, int b, int dsh,
                                             int pos[9]) {
  pos[0] = a;
  pos[8] = b;
  if (b - a <= 16) {  // short range: one linear pass
    int v = 0;
    for (int j = a; j < b; j++) {
      const int g = (int)((k[j] >> dsh) & 7ull);
      while (v < g) pos[++v] = j;
    }
    while (v < 7) pos[++v] = b;
    return;
  }
  int lo = a;
  for (int v = 1; v < 8; v++) {
    int l = lo, h = b;
    while (l < h) {
      const int mid = l + ((h - l) >> 1);
      if ((int)((k[mid] >> dsh) & 7ull) < v) l = mid + 1; else h = mid;
    }
    pos[v] = l;
    lo = l;
  }
}

// d[j] = leading octal digits shared by keys j-1 and j (0..B); d[0] = d[n] = -1 (sentinels);
// samp[] = every 2^ss-th key (bisection seeds of the wide-cell searches)
__device__ __forceinline__ void lcp_body(const int bid, const u64* __restrict__ k, int n, int B,
                                         signed char* __restrict__ d, int ss, u64* __restrict__ samp,
                                         bh_devinfo* __restrict__ info) {
  const int j = bid * 256 + (int)threadIdx.x;
  if (j == 0) {  // first kernel of the build: tree statistics restart; the sticky flags (4th word) survive
    info->n_internal = 0;
    info->n_entries = 0;
    info->max_level = 0;
  }
  if (j > n) return;
  const u64 kj = (j < n) ? k[j] : 0ull;
  if (j < n && (j & ((1 << ss) - 1)) == 0) samp[j >> ss] = kj;
  d[j] = (j == 0 || j == n) ? (signed char)-1 : (signed char)common_digits(k[j - 1], kj, B);
}
__global__ __launch_bounds__(256) void lcp_kernel(const u64* __restrict__ k, int n, int B,
                                                  signed char* __restrict__ d, int ss, u64* __restrict__ samp,
                                                  bh_devinfo* __restrict__ info) {
  lcp_body((int)blockIdx.x, k, n, B, d, ss, samp, info);
}
// Cell of pair j at level L = d[j]:   start a = nearest i < j with d[i] < L,
//                                     end   b = nearest i > j with d[i] < L,
//   j is its first child boundary iff the nearest i < j with d[i] <= L already has d[i] < L,
//   children = 2 + #{ i in (j, b) : d[i] == L }.
// All four are nearest-smaller-value queries on the byte array d[].  A block resolves them for
// 1024 consecutive pairs inside an LDS window of 3072 positions (1024 of halo each side) using
// per-level bitmasks  m[v+1][w] = ballot(d <= v)  built with wave64 ballots: a query is a masked
// word, a short word scan, and clz / ctz / popcount.  Only cells that reach beyond the window
// (about one pair per thousand) are deferred to phase 2: sample-seeded 16-ary searches on the keys,
// 8 lanes per cell (key_lower_bound above).
#ifdef BH_TREE_TRACE
// design-study instrumentation (tools/tree_trace.py): 100 MHz wall-clock stamps per block
__device__ unsigned long long g_tree_trace[2][8192][12];
#define TT_STAMP(kern, k) \
  if (threadIdx.x == 0 && blockIdx.x < 8192) g_tree_trace[kern][blockIdx.x][k] = wall_clock64();
// after everything this thread has loaded so far has arrived
#define TT_STAMPW(kern, k)                                                    \
  if (threadIdx.x == 0 && blockIdx.x < 8192) {                                \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");               \
    g_tree_trace[kern][blockIdx.x][k] = wall_clock64();                       \
  }
#else
#define TT_STAMP(kern, k)
#define TT_STAMPW(kern, k)
#endif
constexpr int kPairTile = 1024;            // pairs per block (template parameter TILE: 1024, or 256 for small n)
constexpr int kHalo = 1024;                // window positions either side of the tile = the largest "narrow" cell
constexpr int kPairWin = 3 * kPairTile;    // LDS window slots (TILE 256 uses the first 256 + 2 * kHalo of them)
constexpr int kPairWords = kPairWin / 64;  // 48
constexpr int kPairLevels = 23;            // v = -1 .. 21

__device__ __forceinline__ int prev_set(const u64* row, int p) {  // highest set bit below p, or -1
  int w = p >> 6;
  u64 bits = row[w] & ((1ull << (p & 63)) - 1ull);
  while (bits == 0ull && w > 0) bits = row[--w];
  return bits ? w * 64 + 63 - __clzll((long long)bits) : -1;
}
__device__ __forceinline__ int next_set(const u64* row, int p) {  // lowest set bit above p, or -1
  int w = p >> 6;
  u64 bits = row[w] & ~((2ull << (p & 63)) - 1ull);
  while (bits == 0ull && w < kPairWords - 1) bits = row[++w];
  return bits ? w * 64 + __ffsll((long long)bits) - 1 : -1;
}
__device__ __forceinline__ int count_between(const u64* row, int p, int q) {  // set bits in (p, q)
  int w0 = p >> 6, w1 = q >> 6;
  const u64 above_p = ~((2ull << (p & 63)) - 1ull);
  const u64 below_q = (1ull << (q & 63)) - 1ull;
  if (w0 == w1) return __popcll(row[w0] & above_p & below_q);
  int c = __popcll(row[w0] & above_p) + __popcll(row[w1] & below_q);
  for (int w = w0 + 1; w < w1; w++) c += __popcll(row[w]);
  return c;
}

// Level masks of one window word: lane I of (lo, hi) <- ballot(d <= I - 1), I = 0 .. 22.  v_cmp into fixed SGPR
// pairs, then v_writelane_b32 with an inline-constant lane select; six compares are issued before the first
// write-lane of a group because a write-lane must not read an SGPR that the VALU wrote in the previous few
// cycles (the assembler does not see hazards inside inline asm; back to back the masks came out wrong).
__device__ __forceinline__ void mask_levels(int dv, u32& lo, u32& hi) {
  asm volatile(
      "v_cmp_ge_i32 s[20:21], -1, %2\n\t"
      "v_cmp_ge_i32 s[22:23], 0, %2\n\t"
      "v_cmp_ge_i32 s[24:25], 1, %2\n\t"
      "v_cmp_ge_i32 s[26:27], 2, %2\n\t"
      "v_cmp_ge_i32 s[28:29], 3, %2\n\t"
      "v_cmp_ge_i32 s[30:31], 4, %2\n\t"
      "v_writelane_b32 %0, s20, 0\n\t"
      "v_writelane_b32 %1, s21, 0\n\t"
      "v_writelane_b32 %0, s22, 1\n\t"
      "v_writelane_b32 %1, s23, 1\n\t"
      "v_writelane_b32 %0, s24, 2\n\t"
      "v_writelane_b32 %1, s25, 2\n\t"
      "v_writelane_b32 %0, s26, 3\n\t"
      "v_writelane_b32 %1, s27, 3\n\t"
      "v_writelane_b32 %0, s28, 4\n\t"
      "v_writelane_b32 %1, s29, 4\n\t"
      "v_writelane_b32 %0, s30, 5\n\t"
      "v_writelane_b32 %1, s31, 5\n\t"
      : "+v"(lo), "+v"(hi)
      : "v"(dv)
      : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31");
  asm volatile(
      "v_cmp_ge_i32 s[20:21], 5, %2\n\t"
      "v_cmp_ge_i32 s[22:23], 6, %2\n\t"
      "v_cmp_ge_i32 s[24:25], 7, %2\n\t"
      "v_cmp_ge_i32 s[26:27], 8, %2\n\t"
      "v_cmp_ge_i32 s[28:29], 9, %2\n\t"
      "v_cmp_ge_i32 s[30:31], 10, %2\n\t"
      "v_writelane_b32 %0, s20, 6\n\t"
      "v_writelane_b32 %1, s21, 6\n\t"
      "v_writelane_b32 %0, s22, 7\n\t"
      "v_writelane_b32 %1, s23, 7\n\t"
      "v_writelane_b32 %0, s24, 8\n\t"
      "v_writelane_b32 %1, s25, 8\n\t"
      "v_writelane_b32 %0, s26, 9\n\t"
      "v_writelane_b32 %1, s27, 9\n\t"
      "v_writelane_b32 %0, s28, 10\n\t"
      "v_writelane_b32 %1, s29, 10\n\t"
      "v_writelane_b32 %0, s30, 11\n\t"
      "v_writelane_b32 %1, s31, 11\n\t"
      : "+v"(lo), "+v"(hi)
      : "v"(dv)
      : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31");
  asm volatile(
      "v_cmp_ge_i32 s[20:21], 11, %2\n\t"
      "v_cmp_ge_i32 s[22:23], 12, %2\n\t"
      "v_cmp_ge_i32 s[24:25], 13, %2\n\t"
      "v_cmp_ge_i32 s[26:27], 14, %2\n\t"
      "v_cmp_ge_i32 s[28:29], 15, %2\n\t"
      "v_cmp_ge_i32 s[30:31], 16, %2\n\t"
      "v_writelane_b32 %0, s20, 12\n\t"
      "v_writelane_b32 %1, s21, 12\n\t"
      "v_writelane_b32 %0, s22, 13\n\t"
      "v_writelane_b32 %1, s23, 13\n\t"
      "v_writelane_b32 %0, s24, 14\n\t"
      "v_writelane_b32 %1, s25, 14\n\t"
      "v_writelane_b32 %0, s26, 15\n\t"
      "v_writelane_b32 %1, s27, 15\n\t"
      "v_writelane_b32 %0, s28, 16\n\t"
      "v_writelane_b32 %1, s29, 16\n\t"
      "v_writelane_b32 %0, s30, 17\n\t"
      "v_writelane_b32 %1, s31, 17\n\t"
      : "+v"(lo), "+v"(hi)
      : "v"(dv)
      : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31");
  asm volatile(
      "v_cmp_ge_i32 s[20:21], 17, %2\n\t"
      "v_cmp_ge_i32 s[22:23], 18, %2\n\t"
      "v_cmp_ge_i32 s[24:25], 19, %2\n\t"
      "v_cmp_ge_i32 s[26:27], 20, %2\n\t"
      "v_cmp_ge_i32 s[28:29], 21, %2\n\t"
      "s_nop 0\n\t"
      "v_writelane_b32 %0, s20, 18\n\t"
      "v_writelane_b32 %1, s21, 18\n\t"
      "v_writelane_b32 %0, s22, 19\n\t"
      "v_writelane_b32 %1, s23, 19\n\t"
      "v_writelane_b32 %0, s24, 20\n\t"
      "v_writelane_b32 %1, s25, 20\n\t"
      "v_writelane_b32 %0, s26, 21\n\t"
      "v_writelane_b32 %1, s27, 21\n\t"
      "v_writelane_b32 %0, s28, 22\n\t"
      "v_writelane_b32 %1, s29, 22\n\t"
      : "+v"(lo), "+v"(hi)
      : "v"(dv)
      : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31");
}

static_assert(kPairLevels == 23, "mask_levels writes exactly 23 levels");

// fill the LDS window: dl[s] = d[base + s] (-1 outside [0, n]) and the per-level masks.  TILE < kPairTile: only the
// first TILE + 2 kHalo slots belong to the window; the rest reads as d = 127 (no mask bit: a search that runs into
// it finds nothing, exactly like one that runs off the full window)
template <int TILE>
__device__ __forceinline__ void build_window(const signed char* __restrict__ d, int n, int base,
                                             u64 (*m)[kPairWords], signed char* dl) {
  constexpr int kUsed = TILE + 2 * kHalo;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // one 16-byte load per thread (192 threads cover the window); a byte-per-thread loop compiles
  // to 12 serialized global round trips per block and dominated this kernel
  for (int s = threadIdx.x * 16; s < kPairWin; s += 256 * 16) {
    const int g0 = base + s;
    if (s >= kUsed) {
      *reinterpret_cast<uint4*>(dl + s) = make_uint4(0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu);
    } else if (g0 >= 0 && g0 + 15 <= n) {
      *reinterpret_cast<uint4*>(dl + s) = *reinterpret_cast<const uint4*>(d + g0);
    } else {
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const int g = g0 + q;
        dl[s + q] = (g < 0 || g > n) ? (signed char)-1 : d[g];
      }
    }
  }
  __syncthreads();
  // level v's ballot goes into lane v + 1 of a register pair (v_writelane), then ONE store per word by lanes
  // 0..22 — the lane-0 store per level (exec save / move / store / restore) was most of the window build
  for (int w = wv; w < kPairWords; w += 4) {
    u32 lo = 0u, hi = 0u;
    if (w * 64 < kUsed) {  // wave-uniform
      const int dv = dl[w * 64 + lane];
      mask_levels(dv, lo, hi);
    }
    if (lane < kPairLevels) m[lane][w] = ((u64)hi << 32) | (u64)lo;
  }
  __syncthreads();
}

template <int TILE>
__device__ __forceinline__ void pairs_body(const int bid, const int nblk, const u64* __restrict__ k,
                                           const signed char* __restrict__ d, int n, int B, int D,
                                           int cap, const u64* __restrict__ ksamp, int ns, int ss,
                                           int* __restrict__ pa, int* __restrict__ pb,
                                           int* __restrict__ pn, int* __restrict__ cb,
                                           int* __restrict__ ttot, int tp_off,
                                           u32* __restrict__ done_count,
                                           bh_devinfo* __restrict__ info) {
  __shared__ u64 s_samp[kSampMax];
  __shared__ __attribute__((aligned(16))) int pnl[TILE];  // child counts of the tile's pairs
  __shared__ u64 m[kPairLevels][kPairWords];
  __shared__ __attribute__((aligned(16))) signed char dl[kPairWin];
  const int t0 = bid * TILE;
  const int base = t0 - kHalo;  // global position of window slot 0
  const int lane = threadIdx.x & 63;
  TT_STAMP(0, 0)
  build_window<TILE>(d, n, base, m, dl);
  TT_STAMP(0, 1)

  __shared__ int wide[TILE];  // window slots of the pairs whose cell leaves the window
  __shared__ int nwide;
  if (threadIdx.x == 0) nwide = 0;
  for (int q = threadIdx.x; q < TILE; q += 256) pnl[q] = 0;
  __syncthreads();

  int cells = 0, maxl = 0;
#pragma unroll 1
  for (int r = 0; r < TILE / 256; r++) {
    const int p = kHalo + r * 256 + (int)threadIdx.x;  // window slot
    const int j = base + p;
    if (j >= n) break;
    int nc = 0;
    bool deferred = false;
    const int L = dl[p];
    if (j >= 1 && L >= 0 && L < D) {
      int a = 0, b = 0;
      const u64* mle = m[L + 1];  // d <= L
      const u64* mlt = m[L];      // d <  L
      const int q = prev_set(mle, p);
      if (q < 0) {
        deferred = true;  // cell starts left of the window
      } else if (dl[q] < L) {  // j is the first child boundary
        const int q2 = next_set(mlt, p);
        if (q2 < 0) {
          deferred = true;  // cell ends right of the window
        } else {
          a = base + q;
          b = base + q2;
          if (b - a > cap) nc = 2 + count_between(mle, p, q2);
        }
      }
      if (nc) {
        pa[j] = a;
        pb[j] = b;
        cells++;
        maxl = max(maxl, L + 1);
      }
    }
    if (deferred) {
      wide[atomicAdd(&nwide, 1)] = p;  // list order is irrelevant: results are keyed by j
    } else {
      pn[j] = nc;
      pnl[p - kHalo] = nc;
    }
  }
  __syncthreads();
  TT_STAMP(0, 2)
#ifdef BH_TREE_TRACE
  if (threadIdx.x == 0 && bid < 8192) g_tree_trace[0][bid][6] = (unsigned long long)nwide;
#endif
  // phase 2: wide cells by key search, 16 lanes per pair: lane v = 0..8 finds the first key whose (L+1)-digit
  // prefix is >= pj*8 + v — v = 0 is the cell's start a, v = 8 its end b, 1..7 the octant boundaries — so all
  // nine searches of a pair (and all pairs of the block) run concurrently: one dependent-load chain instead of
  // three (start, then end, then octants: 17-27 us of the slowest blocks' 48).
  {
    const int sub = threadIdx.x & 15;
    const int nw = nwide;
    if (nw > 0) {  // block-uniform: stage the key samples for the bisection seeds
      for (int s = threadIdx.x; s < ns; s += 256) s_samp[s] = ksamp[s];
    }
    __syncthreads();
    TT_STAMP(0, 8)
    for (int idx = threadIdx.x >> 4; idx < nw; idx += 16) {
      const int p = wide[idx];
      const int j = base + p;
      const int L = dl[p];
      const int sh = 3 * (B - L), dsh = 3 * (B - 1 - L);
      const u64 pj = key_prefix(k[j], sh);
      const u64 kjm = k[j - 1];
      TT_STAMPW(0, 9)
      const int l = (sub <= 8) ? key_lower_bound(k, n, s_samp, ns, ss, dsh, (pj << 3) + (u64)sub) : 0;
      TT_STAMPW(0, 10)
      const int a = __shfl(l, 0, 16), b = __shfl(l, 8, 16);
      const int nxt = __shfl_down(l, 1, 16);
      const u64 bal = __ballot(sub < 8 && nxt > l);
      int nc = 0;
      // j is the cell's FIRST child boundary iff key j-1 lies in the cell's first non-empty octant, i.e. iff no key
      // of the cell has a smaller digit than key j-1's g: lower bound of (pj, g) == a (lane g's result; reading
      // k[a] instead put one more dependent global round trip, ~2.5 us, on the slowest blocks' chain)
      const int lg = __shfl(l, (int)((kjm >> dsh) & 7ull), 16);
      if (lg == a && b - a > cap)
        nc = __popcll((bal >> (lane & ~15)) & 0xffull);
      if (sub == 0) {
        pn[j] = nc;
        pnl[p - kHalo] = nc;
        if (nc) {
          pa[j] = a;
          pb[j] = b;
          cells++;
          maxl = max(maxl, L + 1);
        }
      }
    }
  }
  TT_STAMP(0, 3)
  // Child-block offsets: one contiguous block per cell, in pair order, every block starting at an even entry
  // (an odd block is followed by a padding entry).  The tile scans its own counts here and publishes its total;
  // emit_kernel turns the <= n/1024 totals into tile bases itself (no separate scan pass over n values).
  __shared__ int s_ws[4];
  __syncthreads();
  {
    constexpr int kPer = TILE / 256;  // consecutive pairs per thread: 4 or 1
    const int q0 = threadIdx.x * kPer;
    int v[kPer];
    int sum = 0;
#pragma unroll
    for (int q = 0; q < kPer; q++) {
      v[q] = (pnl[q0 + q] + 1) & ~1;
      sum += v[q];
    }
    int incl = sum;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
      const int u = __shfl_up(incl, dd, 64);
      if (lane >= dd) incl += u;
    }
    const int wv = threadIdx.x >> 6;
    if (lane == 63) s_ws[wv] = incl;
    __syncthreads();
    int ex = incl - sum;
    for (int q = 0; q < wv; q++) ex += s_ws[q];
    const int j0 = t0 + q0;
    int run = ex;
#pragma unroll
    for (int q = 0; q < kPer; q++) {
      if (j0 + q < n) cb[j0 + q] = run;
      run += v[q];
    }
    if (threadIdx.x == 255) bh_publish_i32(ttot + bid, ex + sum);
  }
  // tree statistics: reduce in LDS, then ONE pair of global atomics per block (a global atomic per
  // thread put ~31K same-address atomics in a row: 90 of this kernel's 116 us at 1M bodies).
  // Integer sums / maxima: order-independent.
  __shared__ int s_cells, s_maxl, s_last;
  if (threadIdx.x == 0) {
    s_cells = 0;
    s_maxl = 0;
  }
  __syncthreads();
  if (cells) {
    atomicAdd(&s_cells, cells);
    atomicMax(&s_maxl, maxl);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_cells) {
      atomicAdd(&info->n_internal, s_cells);
      atomicMax(&info->max_level, s_maxl);
    }
  }
  // The block that finishes LAST turns the per-tile totals into tile bases (exclusive scan, <= n/1024 values)
  // for emit_kernel (bh_internal.h: last-block hand-off without a fence).  Thread 255 published this tile's
  // total above; it waits for that store's acknowledgement, then counts the block done.
  TT_STAMP(0, 4)
  if (threadIdx.x == 255) {
    bh_published();
    s_last = bh_last_block(done_count, bid, nblk) ? 1 : 0;
  }
  __syncthreads();
  if (s_last) {
    const int ntiles = nblk;
    int* tpre = ttot + tp_off;
    int carry = 0;
    // this is a serial tail of the kernel: eight totals per thread, loaded together (one agent-scope round trip
    // per 2048 tiles), summed in registers, one block scan
    for (int c0 = 0; c0 < ntiles; c0 += 256 * 8) {
      const int i0 = c0 + (int)threadIdx.x * 8;
      int v[8];
#pragma unroll
      for (int q = 0; q < 8; q++) v[q] = (i0 + q < ntiles) ? bh_collect_i32(ttot + i0 + q) : 0;
      int sum = 0;
#pragma unroll
      for (int q = 0; q < 8; q++) sum += v[q];
      int incl = sum;
#pragma unroll
      for (int dd = 1; dd < 64; dd <<= 1) {
        const int u = __shfl_up(incl, dd, 64);
        if (lane >= dd) incl += u;
      }
      const int wv = threadIdx.x >> 6;
      __syncthreads();
      if (lane == 63) s_ws[wv] = incl;
      __syncthreads();
      int run = carry + incl - sum;
      for (int q = 0; q < wv; q++) run += s_ws[q];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        if (i0 + q < ntiles) tpre[i0 + q] = run;
        run += v[q];
      }
      carry += s_ws[0] + s_ws[1] + s_ws[2] + s_ws[3];
    }
    if (threadIdx.x == 0) tpre[ntiles] = carry;
  }
}

template <int TILE>
__global__ __launch_bounds__(256) void pairs_kernel(const u64* __restrict__ k,
                                                    const signed char* __restrict__ d, int n, int B, int D,
                                                    int cap, const u64* __restrict__ ksamp, int ns, int ss,
                                                    int* __restrict__ pa, int* __restrict__ pb,
                                                    int* __restrict__ pn, int* __restrict__ cb,
                                                    int* __restrict__ ttot, int tp_off,
                                                    u32* __restrict__ done_count,
                                                    bh_devinfo* __restrict__ info) {
  pairs_body<TILE>((int)blockIdx.x, (int)gridDim.x, k, d, n, B, D, cap, ksamp, ns, ss, pa, pb, pn, cb, ttot, tp_off,
                   done_count, info);
}
// The COM prefix scan of a step (fp64 sums of (m, m x, m y, m z) over the sorted bodies) needs nothing the build
// produces and the build nothing of it: the scan's tiles ride in the build's launches — tile sums in extra blocks of
// the pairs launch (blocks [0, pair_blocks): the pairs), the prefixes in extra blocks of the emit launch
// (emit_scan_apply_kernel).  Same tiles and association order as bhk_scan_pm: bit-identical sums.  Rounds 2-4 ran body
// gather + scan on a second stream instead (two event hand-overs, ~13 us of gaps per step); riding along, one stream
// measures faster at every size (65,536 bodies 0.217 -> 0.202 ms/step, 500,000 0.751 -> 0.737, 1M 1.294 -> 1.286).
template <int TILE>
__global__ __launch_bounds__(256) void pairs_scan_reduce_kernel(const u64* __restrict__ k,
                                                                const signed char* __restrict__ d, int n, int B, int D,
                                                                int cap, const u64* __restrict__ ksamp, int ns, int ss,
                                                                int* __restrict__ pa, int* __restrict__ pb,
                                                                int* __restrict__ pn, int* __restrict__ cb,
                                                                int* __restrict__ ttot, int tp_off,
                                                                u32* __restrict__ done_count,
                                                                bh_devinfo* __restrict__ info, int pair_blocks,
                                                                const float4* __restrict__ posm, int scan_tiles,
                                                                bh_d4* tile_sums, u32* __restrict__ scan_done) {
  if ((int)blockIdx.x < pair_blocks)
    pairs_body<TILE>((int)blockIdx.x, pair_blocks, k, d, n, B, D, cap, ksamp, ns, ss, pa, pb, pn, cb, ttot, tp_off,
                     done_count, info);
  else
    bhscan::reduce_body<bhscan::OpD4, bhscan::LoadPM>((int)blockIdx.x - pair_blocks, scan_tiles, bhscan::LoadPM{posm},
                                                      n, nullptr, tile_sums, scan_done);
}

// entry offset of the child block of the cell whose representative pair is j
#define BH_CB(j) (tpre[(j) >> tshift] + cb[j])  // tshift = log2(TILE) of the build

__device__ __forceinline__ bh_node make_child(const u64* __restrict__ k, int B, int D, int cap, float s0,
                                              const int* __restrict__ pn, const int* __restrict__ cb,
                                              const int* tpre, int tshift,
                                              int c0, int c1, int child_level, int n = 0,
                                              const u64* samp = nullptr, int ns = 0, int ss = 0) {
  bh_node r;
  r.x = r.y = r.z = r.m = 0.0f;
  const int m = c1 - c0;
  if (m == 1) {
    r.kind = BH_KIND_BODY;
    r.first = c0;
    r.count = 1;
    r.s = -1.0f;  // negative edge: accepted by every theta >= 0
    return r;
  }
  if (m <= cap) {
    r.kind = BH_KIND_MULTI;
    r.first = c0;
    r.count = m;
    r.s = ldexpf(s0, -child_level);
    return r;
  }
  const int Lb = common_digits(k[c0], k[c1 - 1], B);  // branching level of the (compressed) cell
  if (Lb >= D) {  // never branches above the depth cap: unsplit multi-body cell at level D
    r.kind = BH_KIND_MULTI;
    r.first = c0;
    r.count = m;
    r.s = ldexpf(s0, -D);
    return r;
  }
  // representative = first index whose digit at level Lb exceeds that of key c0
  const int dsh = 3 * (B - 1 - Lb);
  int l;
  if (samp != nullptr && c1 - c0 > 64) {
    // == first key whose (Lb+1)-digit prefix is >= prefix(k[c0]) + 1 (sample-seeded bisection)
    l = key_lower_bound(k, n, samp, ns, ss, dsh, (k[c0] >> dsh) + 1ull);
  } else {
    const int g0 = (int)((k[c0] >> dsh) & 7ull);
    int h = c1 - 1;  // digit(k[c1-1]) > g0, so the answer is in [c0+1, c1-1]
    l = c0 + 1;
    while (l < h) {
      const int mid = l + ((h - l) >> 1);
      if ((int)((k[mid] >> dsh) & 7ull) > g0) h = mid; else l = mid + 1;
    }
  }
  r.kind = BH_KIND_INTERNAL;
  r.first = BH_BLOCK0 + BH_CB(l);
  r.count = pn[l];
  r.s = ldexpf(s0, -Lb);
  return r;
}

// The build writes ONE 32-byte store per record: the record's body range [lo, hi) travels in the bit patterns of
// x and y (the centre of mass is not known yet).  The COM stage reads it from there; the separate er_lo / er_hi
// arrays and the record's x, y, z, m are written by the COM stage only in its canonical mode (com_kernel<true>).
__device__ __forceinline__ void put_rec(bh_node* __restrict__ rec, int e, bh_node r, int lo, int hi) {
  r.x = __int_as_float(lo);
  r.y = __int_as_float(hi);
  rec[e] = r;
}

__device__ __forceinline__ bh_node pad_entry() {
  bh_node r;
  r.x = r.y = r.z = r.m = r.s = 0.0f;
  r.first = r.count = 0;
  r.kind = BH_KIND_PAD;
  return r;
}

// Where, in its parent's child block, does the child [qa, qb) (window slots) go?  Parent = the smallest branching
// cell around it: level Lp = max(d[qa], d[qb]); it starts at the nearest position <= qa with d < Lp and ends at the
// nearest >= qb with d < Lp; its children are delimited by its positions with d == Lp, so the child's ordinal is
// the number of those in (start, qa], and the parent's representative pair (*slot) is the first of them.
// False: there is no parent (the child is the whole system) or the parent is wide (> kHalo bodies, which
// includes every parent that leaves the window) and emits the record itself.
__device__ __forceinline__ bool parent_slot(u64 (*m)[kPairWords], const signed char* dl, int qa, int qb, int* slot,
                                            int* ord) {
  const int da = dl[qa], db = dl[qb];
  const int Lp = max(da, db);
  if (Lp < 0) return false;
  const u64* mlt = m[Lp];       // d <  Lp
  const u64* mleq = m[Lp + 1];  // d <= Lp
  const int ps = (da < Lp) ? qa : prev_set(mlt, qa);
  const int pe = (db < Lp) ? qb : next_set(mlt, qb);
  if (ps < 0 || pe < 0 || pe - ps > kHalo) return false;
  *ord = (qa == ps) ? 0 : 1 + count_between(mleq, ps, qa);
  *slot = next_set(mleq, ps);
  return true;
}

// ttot[tp_off ..] = tile bases (exclusive scan of the per-tile child-entry totals, written by the last block of
// pairs_kernel); tpre[ntiles] = all entries
template <int TILE>
__device__ __forceinline__ void emit_body(const int bid, const u64* __restrict__ k,
                                          const signed char* __restrict__ d, int n, int B, int D,
                                          int cap, const u64* __restrict__ ksamp, int ns, int ss,
                                          const int* __restrict__ pa,
                                          const int* __restrict__ pb, const int* __restrict__ pn,
                                          const int* __restrict__ cb,
                                          const int* __restrict__ ttot, int ntiles, int tp_off,
                                          const float* __restrict__ bounds,
                                          bh_node* __restrict__ rec, int* __restrict__ er_lo,
                                          int* __restrict__ er_hi, int rec_cap,
                                          bh_devinfo* __restrict__ info) {
  __shared__ u64 s_samp[kSampMax];
  __shared__ u64 m[kPairLevels][kPairWords];
  __shared__ __attribute__((aligned(16))) signed char dl[kPairWin];
  const int* __restrict__ tpre = ttot + tp_off;
  constexpr int tshift = (TILE == 1024) ? 10 : 8;
  static_assert(TILE == 1024 || TILE == 256, "tile shift");
  const int t0 = bid * TILE;
  const int base = t0 - kHalo;
  TT_STAMP(1, 0)
  build_window<TILE>(d, n, base, m, dl);
  TT_STAMP(1, 1)
  __shared__ int wide[TILE];
  __shared__ int nwide;
  if (threadIdx.x == 0) nwide = 0;
  __syncthreads();
  const float s0 = bounds[6];
  if (t0 == 0 && threadIdx.x == 0) {  // root record (ref:65-81 initRootKernel)
    const int E = BH_BLOCK0 + tpre[ntiles];
    info->n_entries = E;
    if (E > rec_cap) atomicOr(&info->flags, BH_FLAG_POOL_OVERFLOW);
    put_rec(rec, 0, make_child(k, B, D, cap, s0, pn, cb, tpre, tshift, 0, n, 0), 0, n);
    put_rec(rec, 1, pad_entry(), 0, 0);  // child blocks start at even entries (BH_BLOCK0)
  }
  // Every record is written by a thread that knows it without searching: an emitted cell's thread (its
  // representative pair) writes the cell's OWN record into its parent's child block, and with leaf_cap = 1 the
  // thread of body j writes the leaf that starts at j (the body, or the group of bodies sharing all max_depth
  // digits with it).  Where the record goes follows from the d[] window alone (parent_slot).  (Round 2 began
  // with each parent resolving its <= 8 children one after the other — a level-by-level mask search per internal
  // child and a divergent loop, the wave waiting for its slowest lane: 29 of this kernel's 41 us per block.)
  // The thread's four pairs go through the passes together, so that the global loads of a pass are all in flight
  // at once.  Cells of more than kHalo bodies are "wide": their thread emits ALL their children in phase 2
  // (key searches); the rule depends on the cell alone, so the threads of its children — possibly in other blocks
  // — reach the same verdict.  A narrow cell lies inside this block's window: |a - j|, |b - j| <= kHalo.
  constexpr int kR = TILE / 256;
  int nc_[kR], e_[kR], a_[kR], b_[kR], jp_[kR], ord_[kR];
  int jl_[kR], ordl_[kR], cntl_[kR];  // the leaf starting at body j: parent's pair, ordinal, bodies
  {
    // all loads first, unconditionally (clamped index), then the arithmetic: with the loads inside `if (j < n)` and
    // the sum BH_CB right behind them the compiler waited for each pair's loads before issuing the next pair's —
    // four dependent round trips, 11 of this kernel's 34 us per block
    int tb_[kR], cb_[kR];
#pragma unroll
    for (int r = 0; r < kR; r++) {
      const int jc = min(t0 + r * 256 + (int)threadIdx.x, n - 1);
      nc_[r] = pn[jc];
      cb_[r] = cb[jc];
      tb_[r] = tpre[jc >> tshift];
      a_[r] = pa[jc];  // (defined only where nc > 0; unused otherwise)
      b_[r] = pb[jc];
    }
#pragma unroll
    for (int r = 0; r < kR; r++) {
      const int j = t0 + r * 256 + (int)threadIdx.x;
      const bool in = j > 0 && j < n;
      nc_[r] = in ? nc_[r] : 0;
      e_[r] = in ? BH_BLOCK0 + tb_[r] + cb_[r] : 0;
      a_[r] = in ? a_[r] : 0;
      b_[r] = in ? b_[r] : 0;
      jp_[r] = jl_[r] = -1;
      ord_[r] = ordl_[r] = cntl_[r] = 0;
    }
  }
  TT_STAMPW(1, 8)
#pragma unroll
  for (int r = 0; r < kR; r++) {
    const int p = kHalo + r * 256 + (int)threadIdx.x;
    const int j = base + p;
    // ---- the leaf that starts at body j (leaf_cap = 1)
    if (cap == 1 && j < n) {
      const int da = dl[p];
      if (da < D) {  // else: j shares all D digits with j-1 and sits inside the leaf that started earlier
        int qe = p + 1;
        if (dl[qe] >= D) qe = next_set(m[D], p);  // unsplit cell at the depth cap: up to the next d < D
        if (qe >= 0) {
          int slot, ord;
          if (parent_slot(m, dl, p, qe, &slot, &ord)) {
            jl_[r] = base + slot;
            ordl_[r] = ord;
            cntl_[r] = qe - p;
          }
        }
      }
    }
    const int nc = nc_[r], e = e_[r], a = a_[r], b = b_[r];
    if (nc == 0) continue;
    if (nc > 8 || e + nc > rec_cap) continue;  // cannot happen (records <= 2n); flagged by thread 0 if it did
    const int L = dl[p];
    const int qa = a - base, qb = b - base;
    if (b - a > kHalo) {
      wide[atomicAdd(&nwide, 1)] = p;
      continue;
    }
    if (nc & 1) {  // a block of an odd number of children is followed by one padding entry
      put_rec(rec, e + nc, pad_entry(), 0, 0);
    }
    // ---- leaf_cap > 1: the parent writes its leaf children (runs between the positions with d == L in (qa, qb),
    // p being the first); internal children still write themselves
    if (cap != 1) {
      const u64* mle = m[L + 1];
      int c0 = qa, c1 = p;
#pragma unroll 1
      for (int q = 0; q < nc; q++) {
        const int cnt = c1 - c0;
        bool leaf = true;
        bh_node r;
        r.x = r.y = r.z = r.m = 0.0f;
        r.first = base + c0;
        r.count = cnt;
        if (cnt == 1) {
          r.kind = BH_KIND_BODY;
          r.s = -1.0f;  // negative edge: accepted by every theta >= 0
        } else if (cnt <= cap) {
          r.kind = BH_KIND_MULTI;
          r.s = ldexpf(s0, -(L + 1));
        } else {
          // branches above the depth cap?  (a position with d < D strictly inside the child)
          const int nd = (D >= 1) ? next_set(m[D], c0) : -1;
          leaf = !(nd >= 0 && nd < c1);
          r.kind = BH_KIND_MULTI;  // never branches above the depth cap: unsplit multi-body cell at level D
          r.s = ldexpf(s0, -D);
        }
        if (leaf) {
          put_rec(rec, e + q, r, base + c0, base + c1);
        }
        c0 = c1;
        if (c1 < qb) {
          const int nx = next_set(mle, c0);
          c1 = (nx < 0 || nx > qb) ? qb : nx;
        }
      }
    }
    // ---- this cell's own record goes into its parent's child block
    if (a == 0 && b == n) continue;  // the root cell's record is entry 0
    {
      int slot, ord;
      if (parent_slot(m, dl, qa, qb, &slot, &ord)) {
        jp_[r] = base + slot;
        ord_[r] = ord;
      }
    }
  }
  TT_STAMPW(1, 9)
  int ep_[kR], el_[kR];
  {
    int pb_[kR], pc_[kR], lb_[kR], lc_[kR];  // (loads first, see above)
#pragma unroll
    for (int r = 0; r < kR; r++) {
      const int jp = max(jp_[r], 0), jl = max(jl_[r], 0);
      pb_[r] = tpre[jp >> tshift];
      pc_[r] = cb[jp];
      lb_[r] = tpre[jl >> tshift];
      lc_[r] = cb[jl];
    }
#pragma unroll
    for (int r = 0; r < kR; r++) {
      ep_[r] = (jp_[r] >= 0) ? BH_BLOCK0 + pb_[r] + pc_[r] + ord_[r] : rec_cap;
      el_[r] = (jl_[r] >= 0) ? BH_BLOCK0 + lb_[r] + lc_[r] + ordl_[r] : rec_cap;
    }
  }
  TT_STAMPW(1, 10)
#pragma unroll
  for (int r = 0; r < kR; r++) {
    const int p = kHalo + r * 256 + (int)threadIdx.x;
    if (ep_[r] < rec_cap) {
      bh_node rr;
      rr.x = rr.y = rr.z = rr.m = 0.0f;
      rr.s = ldexpf(s0, -(int)dl[p]);
      rr.first = e_[r];
      rr.count = nc_[r];
      rr.kind = BH_KIND_INTERNAL;
      put_rec(rec, ep_[r], rr, a_[r], b_[r]);
    }
    if (el_[r] < rec_cap) {
      bh_node rr;
      rr.x = rr.y = rr.z = rr.m = 0.0f;
      rr.first = base + p;
      rr.count = cntl_[r];
      if (cntl_[r] == 1) {
        rr.kind = BH_KIND_BODY;
        rr.s = -1.0f;  // negative edge: accepted by every theta >= 0
      } else {
        rr.kind = BH_KIND_MULTI;  // never branches above the depth cap: unsplit multi-body cell at level D
        rr.s = ldexpf(s0, -D);
      }
      put_rec(rec, el_[r], rr, base + p, base + p + cntl_[r]);
    }
  }
  __syncthreads();
  TT_STAMP(1, 2)
#ifdef BH_TREE_TRACE
  if (threadIdx.x == 0 && bid < 8192) g_tree_trace[1][bid][6] = (unsigned long long)nwide;
#endif
  // phase 2: wide cells, 8 lanes per cell, lane v emits the child in octant v (if non-empty)
  {
    const int lane = threadIdx.x & 63, sub = threadIdx.x & 7;
    const int nw = nwide;
    if (nw > 0) {  // block-uniform: stage the key samples for the bisection seeds
      for (int s = threadIdx.x; s < ns; s += 256) s_samp[s] = ksamp[s];
    }
    __syncthreads();
    for (int idx = threadIdx.x >> 3; idx < nw; idx += 32) {
      const int p = wide[idx];
      const int j = base + p;
      const int L = dl[p];
      const int a = pa[j], b = pb[j];
      const int dsh = 3 * (B - 1 - L);
      // lower bound of digit >= sub inside the cell = first key with (L+1)-digit prefix >= pj*8+sub.  The cell's
      // L-digit prefix is read off key j (a key of the cell whose address does not wait for pa[j])
      const u64 pj8 = (k[j] >> dsh) & ~7ull;
      const int l = (sub == 0) ? a : key_lower_bound(k, n, s_samp, ns, ss, dsh, pj8 | (u64)sub);
      int nxt = __shfl_down(l, 1, 8);
      if (sub == 7) nxt = b;
      const bool nonempty = nxt > l;
      const u64 grp = (__ballot(nonempty) >> (lane & ~7)) & 0xffull;
      if (nonempty) {
        const int e = BH_BLOCK0 + BH_CB(j) + __popcll(grp & ((1ull << sub) - 1ull));
        if (e < rec_cap) {
          put_rec(rec, e, make_child(k, B, D, cap, s0, pn, cb, tpre, tshift, l, nxt, L + 1, n, s_samp, ns, ss), l, nxt);
        }
      }
      if (sub == 0) {
        const int nc = __popcll(grp);
        const int e = BH_BLOCK0 + BH_CB(j) + nc;
        if ((nc & 1) && e < rec_cap) {
          put_rec(rec, e, pad_entry(), 0, 0);
        }
      }
    }
  }
  TT_STAMP(1, 3)
}
template <int TILE>
__global__ __launch_bounds__(256) void emit_kernel(const u64* __restrict__ k,
                                                   const signed char* __restrict__ d, int n, int B, int D,
                                                   int cap, const u64* __restrict__ ksamp, int ns, int ss,
                                                   const int* __restrict__ pa,
                                                   const int* __restrict__ pb, const int* __restrict__ pn,
                                                   const int* __restrict__ cb,
                                                   const int* __restrict__ ttot, int ntiles, int tp_off,
                                                   const float* __restrict__ bounds,
                                                   bh_node* __restrict__ rec, int* __restrict__ er_lo,
                                                   int* __restrict__ er_hi, int rec_cap,
                                                   bh_devinfo* __restrict__ info) {
  emit_body<TILE>((int)blockIdx.x, k, d, n, B, D, cap, ksamp, ns, ss, pa, pb, pn, cb, ttot, ntiles, tp_off, bounds, rec,
                  er_lo, er_hi, rec_cap, info);
}
// (see lcp_kernel: the COM prefix scan of a one-stream step) blocks [0, ntiles): emit; the rest: the scan's second pass
template <int TILE>
__global__ __launch_bounds__(256) void emit_scan_apply_kernel(const u64* __restrict__ k,
                                                              const signed char* __restrict__ d, int n, int B, int D,
                                                              int cap, const u64* __restrict__ ksamp, int ns, int ss,
                                                              const int* __restrict__ pa,
                                                              const int* __restrict__ pb, const int* __restrict__ pn,
                                                              const int* __restrict__ cb,
                                                              const int* __restrict__ ttot, int ntiles, int tp_off,
                                                              const float* __restrict__ bounds,
                                                              bh_node* __restrict__ rec, int* __restrict__ er_lo,
                                                              int* __restrict__ er_hi, int rec_cap,
                                                              bh_devinfo* __restrict__ info,
                                                              const float4* __restrict__ posm, int scan_tiles,
                                                              const bh_d4* __restrict__ tile_sums,
                                                              bh_d4* __restrict__ P) {
  if ((int)blockIdx.x < ntiles)
    emit_body<TILE>((int)blockIdx.x, k, d, n, B, D, cap, ksamp, ns, ss, pa, pb, pn, cb, ttot, ntiles, tp_off, bounds,
                    rec, er_lo, er_hi, rec_cap, info);
  else
    bhscan::apply_body<bhscan::OpD4, bhscan::LoadPM>((int)blockIdx.x - ntiles, bhscan::LoadPM{posm}, n, nullptr,
                                                     tile_sums, scan_tiles, P);
}

// ------------------------------------------------------------------ COM
// CANON: also write the canonical tree — the record's centre of mass / mass and the er_lo / er_hi arrays (what
// bh_download_tree, the strict and counting kernels and the domain-decomposed step read).  A whole step of the
// default engine needs only the digests (CANON = false): 24 bytes less traffic per record in an HBM-bound kernel.
template <bool CANON>
__global__ __launch_bounds__(256) void com_kernel(bh_node* __restrict__ rec, bh_frec* __restrict__ frec,
                                                  float G, float theta,
                                                  int* __restrict__ er_lo,
                                                  int* __restrict__ er_hi,
                                                  const bh_devinfo* __restrict__ info, int rec_cap,
                                                  const float4* __restrict__ posm,
                                                  const bh_d4* __restrict__ P, int proto,
                                                  int* __restrict__ spine_pieces, int* __restrict__ spine_count,
                                                  int n_bodies) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int E = min(info->n_entries, rec_cap);  // even: root + padding + blocks of even length
  if ((e & ~63) >= E) return;                   // whole wave beyond the tree
  // every lane builds the digest of its entry (a null digest for padding and beyond E); lanes 2p and 2p+1
  // then write the 64-byte pair of records 2p, 2p+1 with four 16-byte stores (pair layout: bh_internal.h)
  bh_frec fr = frec_null();
  if (e < E) {
    const bh_node r = rec[e];
    // the body range: in the bit patterns of x / y as the build left it (put_rec), or — a repeated bh_com on records
    // an earlier canonical pass already overwrote with the centre of mass — in er_lo / er_hi
    const int lo = proto ? __float_as_int(r.x) : er_lo[e], hi = proto ? __float_as_int(r.y) : er_hi[e];
    if (CANON) {
      er_lo[e] = lo;
      er_hi[e] = hi;
      if (r.kind == BH_KIND_PAD) *reinterpret_cast<float4*>(&rec[e]) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (!CANON && spine_pieces) {
      // Domain-decomposed step (bh_dd.hip): the PIECES of this rank's tree are the children of its two spines (the
      // cells that contain the rank's first or last body: their extent is shared with the neighbouring ranks) that
      // are not on a spine themselves.  The few dozen spine cells list them here — this pass has the body ranges in
      // hand (proto records are not modified by a digest-only pass) — instead of a pass of its own over all records.
      if (e == 0 && r.kind != BH_KIND_INTERNAL) {  // the whole rank is one body / one unsplit cell
        spine_pieces[atomicAdd(spine_count, 1)] = 0;
      } else if (r.kind == BH_KIND_INTERNAL && (lo == 0 || hi == n_bodies)) {
        for (int k = 0; k < r.count; k++) {
          const int c = r.first + k;
          const bh_node rc = rec[c];
          const int clo = proto ? __float_as_int(rc.x) : er_lo[c], chi = proto ? __float_as_int(rc.y) : er_hi[c];
          if (!(rc.kind == BH_KIND_INTERNAL && (clo == 0 || chi == n_bodies))) {
            const int idx = atomicAdd(spine_count, 1);  // (the order is fixed later, by body range: dd_describe_kernel)
            if (idx < BH_DD_PIECE_CAP) spine_pieces[idx] = c;
          }
        }
      }
    }
    if (r.kind != BH_KIND_PAD) {
      float4 o;
      if (r.kind == BH_KIND_BODY) {
        o = posm[lo];
      } else {
        const bh_d4 p1 = P[hi], p0 = P[lo];
        const double M = p1.m - p0.m;
        const double sx = p1.x - p0.x, sy = p1.y - p0.y, sz = p1.z - p0.z;
        const float mass = (float)M;
        o.w = mass;
        if (mass > 1e-6f) {  // ref:180
          o.x = (float)(sx / M);
          o.y = (float)(sy / M);
          o.z = (float)(sz / M);
        } else {
          o.x = (float)sx;
          o.y = (float)sy;
          o.z = (float)sz;
        }
      }
      // x,y,z,m are the first 16 bytes of the record
      if (CANON) *reinterpret_cast<float4*>(&rec[e]) = o;
      fr.x = o.x; fr.y = o.y; fr.z = o.z;
      const bool massive = o.w > 0.0f;  // ref:203: records with mass <= 0 are skipped
      fr.gm = massive ? G * o.w : 0.0f;
      if (!massive || r.kind == BH_KIND_BODY) {
        fr.thr2 = -1.0f;
      } else {
        const float t = r.s / theta;  // theta = 0 -> +inf: never accepted
        fr.thr2 = t * t;
      }
      fr.first = r.first;
      fr.meta = r.count;
      if (r.kind == BH_KIND_MULTI) {
        // an unsplit multi-body cell is, for the fast kernel, a cell whose children are its bodies: their
        // digests form a child block at BH_BODY_DIGEST (even start; a null digest follows an odd count);
        // only the slots of such bodies are ever written or read
        fr.first = BH_BODY_DIGEST(rec_cap, lo, lo);
        for (int b = lo; b < hi; b++) {
          const float4 q = posm[b];
          bh_frec br;
          br.x = q.x; br.y = q.y; br.z = q.z;
          br.gm = q.w > 0.0f ? G * q.w : 0.0f;
          br.thr2 = -1.0f;
          br.first = b;
          br.meta = 1;
          br.pad = 0;
          frec_put(frec, BH_BODY_DIGEST(rec_cap, lo, b), br);
        }
        if ((hi - lo) & 1) frec_put(frec, BH_BODY_DIGEST(rec_cap, lo, hi), frec_null());
      }
    }
  }
  // partner's digest (the other record of the pair)
  bh_frec q;
  q.x = __shfl_xor(fr.x, 1, 64); q.y = __shfl_xor(fr.y, 1, 64); q.z = __shfl_xor(fr.z, 1, 64);
  q.gm = __shfl_xor(fr.gm, 1, 64); q.thr2 = __shfl_xor(fr.thr2, 1, 64);
  q.first = __shfl_xor(fr.first, 1, 64); q.meta = __shfl_xor(fr.meta, 1, 64);
  if (e >= E) return;
  const int lk = frec_link(fr.first, fr.meta), lq = frec_link(q.first, q.meta);  // (bh_internal.h: the walk's stack word)
  float4* pair = reinterpret_cast<float4*>(frec) + (size_t)(e >> 1) * 4;
  if ((e & 1) == 0) {  // slot 0 writes x0 x1 y0 y1 | z0 z1 gm0 gm1
    pair[0] = make_float4(fr.x, q.x, fr.y, q.y);
    pair[1] = make_float4(fr.z, q.z, fr.gm, q.gm);
  } else {             // slot 1 writes thr0 thr1 first0 first1 | meta0 meta1 link0 link1
    pair[2] = make_float4(q.thr2, fr.thr2, __int_as_float(q.first), __int_as_float(fr.first));
    pair[3] = make_float4(__int_as_float(q.meta), __int_as_float(fr.meta), __int_as_float(lq), __int_as_float(lk));
  }
}

// The canonical records of a tree whose COM stage wrote only the digests (bh_step of the default engine), produced
// when somebody asks for them (bh_download_tree): the prefix sums P of that step are still there; a BODY record's
// position comes from its digest — the fused force launch has moved posm since — and its mass from posm (masses
// never change).  Writes exactly what com_kernel<true> would have written; touches no digest.
__global__ __launch_bounds__(256) void canon_kernel(bh_node* __restrict__ rec, const bh_frec* __restrict__ frec,
                                                    int* __restrict__ er_lo, int* __restrict__ er_hi,
                                                    const bh_devinfo* __restrict__ info, int rec_cap,
                                                    const float4* __restrict__ posm,
                                                    const bh_d4* __restrict__ P) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int E = min(info->n_entries, rec_cap);
  if (e >= E) return;
  const bh_node r = rec[e];
  const int lo = __float_as_int(r.x), hi = __float_as_int(r.y);
  er_lo[e] = lo;
  er_hi[e] = hi;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r.kind == BH_KIND_BODY) {
    const bh_frec d = frec_get(frec, e);
    o = make_float4(d.x, d.y, d.z, posm[lo].w);
  } else if (r.kind != BH_KIND_PAD) {
    const bh_d4 p1 = P[hi], p0 = P[lo];
    const double M = p1.m - p0.m;
    const double sx = p1.x - p0.x, sy = p1.y - p0.y, sz = p1.z - p0.z;
    const float mass = (float)M;
    o.w = mass;
    if (mass > 1e-6f) {  // ref:180
      o.x = (float)(sx / M);
      o.y = (float)(sy / M);
      o.z = (float)(sz / M);
    } else {
      o.x = (float)sx;
      o.y = (float)sy;
      o.z = (float)sz;
    }
  }
  *reinterpret_cast<float4*>(&rec[e]) = o;
}

}  // namespace

hipError_t bhk_bbox(bh_ctx* c) {
  const int n = c->n;
  int blocks = (n + 1023) / 1024;
  if (blocks > BH_BBOX_BLOCKS) blocks = BH_BBOX_BLOCKS;
  if (blocks < 1) blocks = 1;
  bbox_partial_kernel<<<blocks, 256, 0, c->stream>>>(c->posm[c->cur], n, c->bbox_partial);
  bbox_final_kernel<<<1, 256, 0, c->stream>>>(c->bbox_partial, blocks, 6, 0, c->bounds);
  return hipGetLastError();
}

// min/max of the local bodies only (6 floats to out6): the per-rank half of the global cube
hipError_t bhk_bbox_raw(bh_ctx* c, float* out6) {
  const int n = c->n;
  int blocks = (n + 1023) / 1024;
  if (blocks > BH_BBOX_BLOCKS) blocks = BH_BBOX_BLOCKS;
  if (blocks < 1) blocks = 1;
  bbox_partial_kernel<<<blocks, 256, 0, c->stream>>>(c->posm[c->cur], n, c->bbox_partial);
  bbox_final_kernel<<<1, 256, 0, c->stream>>>(c->bbox_partial, blocks, 6, 1, out6);
  return hipGetLastError();
}

// cube from `nrows` rows of per-rank min/max (exact: min/max are associative)
hipError_t bhk_bounds_from_rows(bh_ctx* c, const float* rows, int nrows, int stride_floats) {
  bbox_final_kernel<<<1, 256, 0, c->stream>>>(rows, nrows, stride_floats, 0, c->bounds);
  return hipGetLastError();
}

// with_bbox: also produce the next step's bounding cube in c->bounds_next (see integrate_kernel)
hipError_t bhk_integrate(bh_ctx* c, bool with_bbox) {
  const int n = c->n;
  const int blocks = (n + BH_INTEGRATE_TILE - 1) / BH_INTEGRATE_TILE;
  if (with_bbox && n <= BH_INT_SMALL_N)
    integrate_kernel<true, 1024><<<(n + 1023) / 1024, 1024, 0, c->stream>>>(
        c->posm[c->cur], c->velid[c->cur], c->acc, c->acc2, n, c->p.dt, c->p.max_speed, c->ibox_rows, c->blk_done2,
        c->dd ? c->dd_minmax : c->bounds_next, c->dd ? 1 : 0);
  else if (with_bbox)
    integrate_kernel<true><<<blocks, 1024, 0, c->stream>>>(c->posm[c->cur], c->velid[c->cur], c->acc, c->acc2, n,
                                                           c->p.dt, c->p.max_speed, c->ibox_rows,
                                                           c->blk_done2,
                                                           c->dd ? c->dd_minmax : c->bounds_next, c->dd ? 1 : 0);
  else
    integrate_kernel<false><<<blocks, 1024, 0, c->stream>>>(c->posm[c->cur], c->velid[c->cur], c->acc, c->acc2,
                                                            n, c->p.dt, c->p.max_speed, nullptr, nullptr,
                                                            nullptr, 0);
  return hipGetLastError();
}

// for_sort: a sort follows (keys + splitters + bucket counts in one kernel when the splitter sort applies)
hipError_t bhk_keys(bh_ctx* c, bool for_sort) {
  if (for_sort && bhk_sort_split_eligible(c)) return bhk_keys_split(c);
  if (c->keys_split) {  // bucket counts of keys that no sort consumed: void them
    const hipError_t e = hipMemsetAsync(c->sp_count + 256 * (c->sp_par & 1), 0, 256 * sizeof(u32), c->stream);
    if (e != hipSuccess) return e;
    c->keys_split = false;
  }
  const int n = c->n;
  const int blocks = (n + 255) / 256;
  if (c->B == 10)
    keys_kernel<10><<<blocks, 256, 0, c->stream>>>(c->posm[c->cur], c->bounds, n, 0, c->keys[0]);
  else
    keys_kernel<21><<<blocks, 256, 0, c->stream>>>(c->posm[c->cur], c->bounds, n, c->p.key_curve, c->keys[0]);
  return hipGetLastError();
}

#ifdef BH_TREE_TRACE
extern "C" int bh_debug_tree_trace(void* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tree_trace), sizeof(g_tree_trace));
}
#endif

// pm_scan: also the fp64 prefix sums of (m, m x, m y, m z) over the sorted bodies (what bhk_scan_pm writes into c->P
// for the COM stage), their tiles riding in the pairs / emit launches (one-stream steps: pairs_scan_reduce_kernel)
hipError_t bhk_build(bh_ctx* c, bool pm_scan) {
  const int n = c->n;

  c->rec_proto = true;  // records carry their body range in x / y until the (canonical) COM stage has run
  c->com_digests = false;
  const u64* k = c->keys[c->key_buf];
  // every 2^ss-th key, at most kSampMax of them: bisection seeds of the wide-cell searches
  int ss = 12;
  while (((n + (1 << ss) - 1) >> ss) > kSampMax) ss++;
  const int ns = (n + (1 << ss) - 1) >> ss;
  const int scan_tiles = (n + BH_SCAN_TILE - 1) / BH_SCAN_TILE;
  bh_d4* const tile_sums = reinterpret_cast<bh_d4*>(c->scan_tmp2);
  u32* const scan_done = reinterpret_cast<u32*>(reinterpret_cast<char*>(c->scan_tmp2) + c->scan_cnt_off);
  lcp_kernel<<<(n + 1 + 255) / 256, 256, 0, c->stream>>>(k, n, c->B, c->d8, ss, c->ksamp, c->info);
  // 1024 pairs per block; 256 up to 163,840 bodies, where n / 1024 blocks leave most of the 256 CUs idle and a
  // block's four rounds are pure latency (65,536 bodies: pairs 22 -> 13 us, emit 28 -> 15 us); the halo stays 1024
  const int tile = (n <= BH_PAIR_SMALL_N) ? 256 : kPairTile;
  const int ntiles = (n + tile - 1) / tile;
  const int tp_off = n / tile + 2;
  // child-block offsets = tile base (ttot[tp_off ..]) + offset in the tile (cb[]), both written by pairs_kernel
  if (tile == 256) {
    if (pm_scan)
      pairs_scan_reduce_kernel<256><<<ntiles + scan_tiles, 256, 0, c->stream>>>(
          k, c->d8, n, c->B, c->D, c->cap, c->ksamp, ns, ss, c->pa, c->pb, c->pn, c->cb, c->ttot, tp_off, c->blk_done,
          c->info, ntiles, c->posm[c->cur], scan_tiles, tile_sums, scan_done);
    else
      pairs_kernel<256><<<ntiles, 256, 0, c->stream>>>(k, c->d8, n, c->B, c->D, c->cap, c->ksamp, ns, ss, c->pa,
                                                       c->pb, c->pn, c->cb, c->ttot, tp_off, c->blk_done, c->info);
    if (pm_scan)
      emit_scan_apply_kernel<256><<<ntiles + scan_tiles, 256, 0, c->stream>>>(
          k, c->d8, n, c->B, c->D, c->cap, c->ksamp, ns, ss, c->pa, c->pb, c->pn, c->cb, c->ttot, ntiles, tp_off,
          c->bounds, c->rec, c->er_lo, c->er_hi, c->rec_cap, c->info, c->posm[c->cur], scan_tiles, tile_sums, c->P);
    else
      emit_kernel<256><<<ntiles, 256, 0, c->stream>>>(k, c->d8, n, c->B, c->D, c->cap, c->ksamp, ns, ss, c->pa, c->pb,
                                                      c->pn, c->cb, c->ttot, ntiles, tp_off, c->bounds, c->rec,
                                                      c->er_lo, c->er_hi, c->rec_cap, c->info);
  } else {
    if (pm_scan)
      pairs_scan_reduce_kernel<kPairTile><<<ntiles + scan_tiles, 256, 0, c->stream>>>(
          k, c->d8, n, c->B, c->D, c->cap, c->ksamp, ns, ss, c->pa, c->pb, c->pn, c->cb, c->ttot, tp_off, c->blk_done,
          c->info, ntiles, c->posm[c->cur], scan_tiles, tile_sums, scan_done);
    else
      pairs_kernel<kPairTile><<<ntiles, 256, 0, c->stream>>>(k, c->d8, n, c->B, c->D, c->cap, c->ksamp, ns, ss, c->pa,
                                                             c->pb, c->pn, c->cb, c->ttot, tp_off, c->blk_done,
                                                             c->info);
    if (pm_scan)
      emit_scan_apply_kernel<kPairTile><<<ntiles + scan_tiles, 256, 0, c->stream>>>(
          k, c->d8, n, c->B, c->D, c->cap, c->ksamp, ns, ss, c->pa, c->pb, c->pn, c->cb, c->ttot, ntiles, tp_off,
          c->bounds, c->rec, c->er_lo, c->er_hi, c->rec_cap, c->info, c->posm[c->cur], scan_tiles, tile_sums, c->P);
    else
      emit_kernel<kPairTile><<<ntiles, 256, 0, c->stream>>>(k, c->d8, n, c->B, c->D, c->cap, c->ksamp, ns, ss, c->pa,
                                                            c->pb, c->pn, c->cb, c->ttot, ntiles, tp_off, c->bounds,
                                                            c->rec, c->er_lo, c->er_hi, c->rec_cap, c->info);
  }
  return hipGetLastError();
}

hipError_t bhk_com(bh_ctx* c) {
  hipError_t e = bhk_scan_pm(c, c->posm[c->cur], c->P, c->n);
  if (e != hipSuccess) return e;
  return bhk_com_records(c, true);
}

// canonical = false: digests only (bh_step of the default engine); the canonical records then stay in the
// build's form (body range in x / y) until a stage call (bh_com) or a canonical step rewrites them
hipError_t bhk_com_records(bh_ctx* c, bool canonical, int* spine_pieces, int* spine_count) {
  const int blocks = (c->rec_cap + 255) / 256;
  // (a digest-only pass on records that are already canonical cannot happen: every caller builds first)
  const int proto = c->rec_proto ? 1 : 0;
  if (canonical)
    com_kernel<true><<<blocks, 256, 0, c->stream>>>(c->rec, c->frec, c->p.G, c->p.theta, c->er_lo, c->er_hi, c->info,
                                                    c->rec_cap, c->posm[c->cur], c->P, proto, nullptr, nullptr, 0);
  else
    com_kernel<false><<<blocks, 256, 0, c->stream>>>(c->rec, c->frec, c->p.G, c->p.theta, c->er_lo, c->er_hi, c->info,
                                                     c->rec_cap, c->posm[c->cur], c->P, proto, spine_pieces,
                                                     spine_count, c->n);
  if (canonical) c->rec_proto = false;
  c->com_digests = true;
  return hipGetLastError();
}

// bh_download_tree after a digest-only step: make the records canonical now (canon_kernel)
hipError_t bhk_canonical_records(bh_ctx* c) {
  if (!c->rec_proto) return hipSuccess;
  canon_kernel<<<(c->rec_cap + 255) / 256, 256, 0, c->stream>>>(c->rec, c->frec, c->er_lo, c->er_hi, c->info,
                                                                c->rec_cap, c->posm[c->cur], c->P);
  c->rec_proto = false;
  return hipGetLastError();
}
