// bh_sort_onesweep.hip — the step's stable sort by key, two implementations with identical results:
//
//  * splitter sort (bhk_sort_split, further down): every sort whose input is still in an earlier sort's key
//    order — keys + splitters + bucket counts in one kernel, ONE stable partition pass, one workgroup per bucket
//    sorting it in LDS and gathering the bodies (3 kernels);
//  * LSD radix sort, ONE kernel per 8-bit pass (bhk_sort_onesweep): the first sort after an upload, n > 1.5M,
//    bh_params.sort_variant 2.
//
// Both replace thrust::sort_by_key (ref nbody_v5_bench.cu:262-264); same contract as bh_sort.hip (stable
// ascending by key, value = slot index).  The radix pass:
//   * one up-front kernel builds the global digit histograms of ALL passes (global digit totals do
//     not depend on the order of the keys);
//   * each pass kernel counts its 4096-key tile's digits, publishes them, ranks the tile (wave64 match-any,
//     wave-private LDS counters combined in wave order -> stable) and resolves the tile's per-digit prefix by a
//     TWO-LEVEL DECOUPLED LOOK-BACK instead of a separate histogram + scan:
//       - tiles take a ticket (atomicAdd) when they start, so a tile's predecessors are already
//         running: look-back only ever waits on resident workgroups (no dispatch-order assumption);
//       - per (tile, digit) ONE 8-byte granule {tag:30 | state:2 | count:32} written and polled with
//         relaxed agent-scope atomics (the data-tagged single-granule hand-off of
//         cdna_hip_programming.md Guideline 16: flag and value travel in one word, so no
//         fence ordering between them is needed).  state 1 = this tile's count, 3 = sum over the tile's group
//         of 16 up to and including it, 2 = inclusive prefix.  The tag is the sort-call number, so the table is
//         never cleared and a stale entry from an earlier call reads as "not ready";
//       - thread d handles digit d (256 independent chains per block): own group first, then the last tile of
//         every earlier group (details at the kernel);
//       - every spin is bounded; on timeout the kernel sets BH_FLAG_SORT_TIMEOUT and goes on
//         (wrong order, loudly reported) instead of hanging the device.
#include "bh_internal.h"
#include "bh_keys.h"

namespace {

#ifndef BH_OS_THREADS
#define BH_OS_THREADS 512  // 8 keys per thread: the tile's serial ranking chain is half as long as with 256 x 16 (-2.2 us per pass at 1M keys)
#endif
#ifndef BH_OS_LOOK
#define BH_OS_LOOK 8
#endif
#ifndef BH_OS_GROUP
#define BH_OS_GROUP 16
#endif
constexpr int kTile = BH_SORT_TILE;
constexpr int kHistThreads = 256;
constexpr int kHistItems = kTile / kHistThreads;
constexpr int kThreads = BH_OS_THREADS;  // pass kernel: kWaves waves rank kItems x 64 keys each
constexpr int kWaves = kThreads / 64;
constexpr int kItems = kTile / kThreads;
constexpr u32 kSpinLimit = 1u << 22;
static_assert(kThreads >= 256 && kThreads % 256 == 0 && kTile % kThreads == 0, "sort tile shape");

#ifdef BH_OS_TRACE
// design-study instrumentation (tools/sort_trace.py): 100 MHz wall-clock stamps of every tile's phases
__device__ unsigned long long g_os_trace[8][4096][8];
#define OS_STAMP(k) \
  if (threadIdx.x == 0 && tile < 4096) g_os_trace[shift >> 3][tile][k] = wall_clock64();
__device__ unsigned long long g_ls_trace[256][16];
#define LS_STAMP(k, v) \
  if (threadIdx.x == 0) g_ls_trace[blockIdx.x][k] = (v);
__device__ unsigned long long g_ks_trace[1024][8];  // keys_split_kernel (tools/ks_trace.py)
#define KS_STAMP(k)                                             \
  if (threadIdx.x == 0 && blockIdx.x < 1024) {                  \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
    g_ks_trace[blockIdx.x][k] = wall_clock64();                 \
  }
#else
#define OS_STAMP(k)
#define LS_STAMP(k, v)
#define KS_STAMP(k)
#endif

__device__ __forceinline__ u64 pack_granule(u32 tag, u32 state, u32 value) {
  return ((u64)((tag << 2) | state) << 32) | (u64)value;
}

// global digit histograms of every pass in one sweep over the keys
__global__ __launch_bounds__(kHistThreads) void onesweep_hist_kernel(const u64* __restrict__ keys, int n,
                                                                 int passes, u32* __restrict__ ghist) {
  __shared__ u32 h[8][256];
  for (int p = 0; p < 8; p++) h[p][threadIdx.x] = 0;
  __syncthreads();
  const int base = blockIdx.x * kTile;
#pragma unroll 4
  for (int r = 0; r < kHistItems; r++) {
    const int i = base + r * kHistThreads + (int)threadIdx.x;
    // bodies are stored in the previous step's key order, so a wave's 64 keys usually share their upper digits:
    // one LDS atomic per wave then, instead of 64 conflicting ones on the same counter
    const bool valid = i < n;
    const u64 k = valid ? keys[i] : 0ull;
    const u64 act = __ballot(valid);
    if (act == 0ull) continue;
    const u64 k0 = __shfl(k, __ffsll((long long)act) - 1, 64);
    for (int p = 0; p < passes; p++) {
      const u32 dg = (u32)(k >> (8 * p)) & 255u, d0 = (u32)(k0 >> (8 * p)) & 255u;
      if (__ballot(valid && dg != d0) == 0ull) {
        if ((threadIdx.x & 63) == __ffsll((long long)act) - 1) atomicAdd(&h[p][d0], (u32)__popcll(act));
      } else if (valid) {
        atomicAdd(&h[p][dg], 1u);
      }
    }
  }
  __syncthreads();
  for (int p = 0; p < passes; p++) {
    const u32 v = h[p][threadIdx.x];
    if (v) atomicAdd(&ghist[p * 256 + threadIdx.x], v);
  }
}

// rank 64 keys of a wave by digit: returns the number of earlier keys (in this wave's earlier rows and in the
// lower lanes of this row) with the same digit, and bumps the wave-private counter.  Match-any over the 8
// digit bits; per bit: sign-extended bit field, ballot, two 3-input booleans = 4 VALU — the ranking is VALU-issue
// bound, so the instruction count is the cost.
__device__ __forceinline__ u32 wave_rank(u32 g, bool valid, u32* wc, u32 lt_lo, u32 lt_hi) {
  const u64 vb = __ballot(valid);
  u32 xlo = 0u, xhi = 0u;  // lanes whose digit differs from mine in some bit
#pragma unroll
  for (int bit = 0; bit < 8; bit++) {
    const u32 bm = (u32)__builtin_amdgcn_sbfe((int)g, bit, 1);  // all ones iff the bit is set
    const u64 bb = __ballot(bm != 0u);
    xlo = __builtin_amdgcn_bitop3_b32(xlo, (u32)bb, bm, 0xf6);          // x | (bb ^ bm): gfx950 3-input boolean
    xhi = __builtin_amdgcn_bitop3_b32(xhi, (u32)(bb >> 32), bm, 0xf6);
  }
  const u32 mlo = ~xlo & (u32)vb, mhi = ~xhi & (u32)(vb >> 32);
  const u32 rank = (u32)__popc(mlo & lt_lo) + (u32)__popc(mhi & lt_hi);
  u32 prev = 0;
  if (valid) prev = wc[g];
  if (valid && rank == 0) wc[g] = prev + (u32)__popc(mlo) + (u32)__popc(mhi);
  return prev + rank;
}

// One granule row of look-back: kN independent loads `stride` tiles apart starting at tile tt, consumed in
// order.  Accepts states >= min_state; returns how many granules were consumed and sets `hit` when one of
// them closed the walk (state 2 always; state 3 / tile <= floor when `close_on_3`).
template <int kN>
__device__ __forceinline__ int lookback_round(const u64* __restrict__ status, int t, int tt, int stride,
                                              int floor_tile, u32 tag, u32 min_state, bool close_on_3,
                                              u32& excl, bool& closed, bool& inclusive) {
  u64 e[kN];
#pragma unroll
  for (int q = 0; q < kN; q++)
    e[q] = (tt - q * stride >= floor_tile)
               ? __hip_atomic_load(status + (size_t)(tt - q * stride) * 256 + t, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT)
               : 0ull;
  int used = 0;
#pragma unroll
  for (int q = 0; q < kN; q++) {
    if (closed || used != q) continue;  // stop at the first granule that is not ready
    const u32 hi = (u32)(e[q] >> 32);
    const u32 st = hi & 3u;
    if (tt - q * stride < floor_tile || (hi >> 2) != tag || st < min_state) continue;
    excl += (u32)e[q];
    used = q + 1;
    if (st == 2u) closed = inclusive = true;
    if ((st == 3u && close_on_3) || tt - q * stride == floor_tile) closed = true;
  }
  return used;
}

// number of splitters <= key (sp[] ascending, padded with ~0): the bucket of the key, 0..255
__device__ __forceinline__ u32 splitter_bucket(const u64* sp, u64 key) {
  u32 lo = 0;
#pragma unroll
  for (u32 step = 128; step >= 1; step >>= 1)
    if (sp[lo + step - 1] <= key) lo += step;
  return lo;
}

// The splitters are the keys of evenly spaced bodies of the array being partitioned, so the body at position i
// almost always belongs to bucket ~ i nb / n: two independent LDS reads confirm the guess; the 8-step search
// runs only for the rows where some lane's guess fails (bucket boundaries, bodies that moved far).
__device__ __forceinline__ u32 splitter_bucket_guess(const u64* sp, u64 key, int i, float nb_over_n) {
  const u32 gq = min((u32)((float)i * nb_over_n), 255u);
  const u64 a = gq ? sp[gq - 1] : 0ull, b = sp[gq];
  if (a <= key && key < b) return gq;
  return splitter_bucket(sp, key);
}

// SPLIT = false: one LSD radix pass, digit = 8 key bits at `shift`.
// SPLIT = true : the partition pass of the splitter sort (bhk_sort_split below): "digit" = the key's bucket
//                among the <= 255 sorted splitters, ghist_pass = the bucket totals counted by keys_split_kernel.
template <bool SPLIT>
__global__ __launch_bounds__(kThreads) void onesweep_pass_kernel(
    const u64* __restrict__ kin, const u32* __restrict__ vin, u64* __restrict__ kout,
    u32* __restrict__ vout, int n, int shift, const u32* __restrict__ ghist_pass,
    u64* __restrict__ status, u32* __restrict__ ticket, const u32* __restrict__ call_ptr, int first_pass,
    bh_devinfo* __restrict__ info, const u64* __restrict__ splitters, float nb_over_n) {
  __shared__ u64 s_sp[SPLIT ? 256 : 1];
  __shared__ unsigned char sdig[SPLIT ? kTile : 1];  // the staged keys' buckets (cheaper than searching again)
  if (SPLIT && threadIdx.x < 256) s_sp[threadIdx.x] = splitters[threadIdx.x];
  u32 dg[SPLIT ? kItems : 1];
#define OS_DIGIT(r) (SPLIT ? dg[r] : ((u32)(key[r] >> shift) & 255u))
  // tag = number of this sort call, kept on the device (sw_ticket[8], advanced by the gather kernel that ends
  // every sort) so that the kernel arguments of a step never change: bh_step replays as a HIP graph
  const u32 tag = (*call_ptr + 1u) & 0x3fffffffu;
  __shared__ u32 wcnt[kWaves][256];
  __shared__ u32 th[256];
  __shared__ u32 gbase[256];
  __shared__ u32 toff[256];
  __shared__ u32 dsum[4];
  __shared__ u32 s_tile;
  __shared__ u64 stage[kTile];  // 32 KB: the tile in digit order (keys, then values as u32)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);  // tickets restart at 0: the gather kernel clears them
  for (int q = threadIdx.x; q < kWaves * 256; q += kThreads) (&wcnt[0][0])[q] = 0;
  if (threadIdx.x < 256) th[threadIdx.x] = 0;
  __syncthreads();
  const int tile = (int)s_tile;
  OS_STAMP(0)

  const int base = tile * kTile + w * (64 * kItems);
  u64 key[kItems];
  u32 val[kItems];
  u32 rk[kItems];
#pragma unroll
  for (int r = 0; r < kItems; r++) {
    const int i = base + r * 64 + lane;
    const bool valid = i < n;
    key[r] = valid ? kin[i] : ~0ull;
    val[r] = valid ? (first_pass ? (u32)i : vin[i]) : 0u;
  }
  if (SPLIT) {
    __syncthreads();  // s_sp
#pragma unroll
    for (int r = 0; r < kItems; r++)
      dg[SPLIT ? r : 0] = splitter_bucket_guess(s_sp, key[r], base + r * 64 + lane, nb_over_n);
  }
  // ---- 1. the tile's digit counts FIRST (one LDS atomic per key, or per wave when its 64 keys share the
  // digit), published before the ranking: by the time the ranking is done the neighbours' counts have
  // crossed the fabric (a publication takes ~1 us to become visible to another XCD) and the look-back below
  // reads them without spinning.
#pragma unroll
  for (int r = 0; r < kItems; r++) {
    const int i = base + r * 64 + lane;
    const bool valid = i < n;
    const u32 g = OS_DIGIT(r);
    const u64 act = __ballot(valid);
    if (act == 0ull) continue;
    const int l0 = __ffsll((long long)act) - 1;
    const u32 g0 = __shfl(g, l0, 64);
    if (__ballot(valid && g != g0) == 0ull) {
      if (lane == l0) atomicAdd(&th[g0], (u32)__popcll(act));
    } else if (valid) {
      atomicAdd(&th[g], 1u);
    }
  }
  __syncthreads();
  // threads 0..255 own one digit each (the first four waves); the other waves only keep the barriers company
  const bool dig = threadIdx.x < 256;
  const int t = threadIdx.x & 255;  // digit
  // Two-level look-back.  When n / tile ~ number of CUs every tile is resident at once and nobody has an
  // inclusive prefix early: a plain look-back then reads O(tiles) granule rows per tile (61 MB per pass at 1M
  // keys, more than the keys).  Instead tiles form groups of kGroup; a tile first sums the counts of its own
  // group's earlier tiles (consecutive rows, ONE round of independent loads) and publishes that as state 3
  // (group-inclusive), then walks back over the LAST tile of every earlier group only (rows kGroup apart),
  // which carry state 3 or 2.  States: 1 = count, 3 = group-inclusive, 2 = inclusive.
  constexpr int kGroup = BH_OS_GROUP, kLook = BH_OS_LOOK;
  const int gs = tile - tile % kGroup;
  const u32 h = dig ? th[t] : 0u;
  u64* mine = status + (size_t)tile * 256 + t;
  u32 excl = 0, spins = 0;
  bool done = (tile == 0), fail = false;
  if (dig)
    __hip_atomic_store(mine, pack_granule(tag, tile == 0 ? 2u : (tile == gs ? 3u : 1u), h), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);

  // ---- 2. rank the keys: wave64 match-any with 8 ballots, wave-private counters combined in wave order
#pragma unroll
  for (int r = 0; r < kItems; r++) {
    const int i = base + r * 64 + lane;
    rk[r] = wave_rank(OS_DIGIT(r), i < n, wcnt[w], (u32)lt, (u32)(lt >> 32));
  }
  __syncthreads();
  OS_STAMP(1)
  if (dig) {
    u32 run = 0;
#pragma unroll
    for (int q = 0; q < kWaves; q++) {
      const u32 cq = wcnt[q][t];
      wcnt[q][t] = run;  // keys of digit t in the earlier waves of this tile
      run += cq;
    }
    // ---- 3. own group: rows tile-1 .. gs
    if (tile != gs) {
      int tt = tile - 1;
      bool grp = false, incl = false;
      while (!grp && !fail) {
        const int used = lookback_round<kGroup - 1>(status, t, tt, 1, gs, tag, 1u, true, excl, grp, incl);
        tt -= used;
        if (!grp && used == 0) {
          if (++spins > kSpinLimit) fail = true;
          __builtin_amdgcn_s_sleep(1);
        }
      }
      done = incl;
      if (!done && !fail)
        __hip_atomic_store(mine, pack_granule(tag, 3u, excl + h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (gs == 0) done = true;  // group 0: group-inclusive is inclusive
  }
  OS_STAMP(2)
  // ---- 4. everything that does not need the prefix over the earlier groups, while the state-3 granules
  // travel: digit bases, tile-local offsets, the tile staged through LDS in digit order (so that each digit's
  // run leaves as one contiguous, coalesced global write instead of scattered 8-byte stores)
  const u32 dv = dig ? ghist_pass[t] : 0u;
  u32 incl = dv;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u32 u = __shfl_up(incl, d, 64);
    if (lane >= d) incl += u;
  }
  if (dig && lane == 63) dsum[w] = incl;
  __syncthreads();
  u32 wp = 0;
  for (int q = 0; q < (w & 3); q++) wp += dsum[q];
  const u32 gpos0 = wp + incl - dv;  // global position of the first key of digit t (all tiles)
  u32 inc2 = h;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u32 u = __shfl_up(inc2, d, 64);
    if (lane >= d) inc2 += u;
  }
  __syncthreads();  // dsum is reused
  if (dig && lane == 63) dsum[w] = inc2;
  __syncthreads();
  u32 wp2 = 0;
  for (int q = 0; q < (w & 3); q++) wp2 += dsum[q];
  const u32 lo = wp2 + inc2 - h;  // tile-local offset of digit t
  if (dig) toff[t] = lo;
  __syncthreads();
  const int nvalid = min(kTile, n - tile * kTile);
  u32 lp[kItems];
#pragma unroll
  for (int r = 0; r < kItems; r++) {
    const int i = base + r * 64 + lane;
    const u32 g = OS_DIGIT(r);
    lp[r] = toff[g] + wcnt[w][g] + rk[r];
    if (i < n) {
      stage[lp[r]] = key[r];
      if (SPLIT) sdig[lp[r]] = (unsigned char)g;
    }
  }
  __syncthreads();
  u64 k2[kItems];
#pragma unroll
  for (int r = 0; r < kItems; r++) {
    const int idx = r * kThreads + (int)threadIdx.x;
    k2[r] = (idx < nvalid) ? stage[idx] : ~0ull;
    if (SPLIT) dg[SPLIT ? r : 0] = (idx < nvalid) ? (u32)sdig[idx] : 0u;
  }
  __syncthreads();
  u32* stage32 = reinterpret_cast<u32*>(stage);
#pragma unroll
  for (int r = 0; r < kItems; r++) {
    const int i = base + r * 64 + lane;
    if (i < n) stage32[lp[r]] = val[r];
  }
  __syncthreads();
  u32 v2[kItems];
#pragma unroll
  for (int r = 0; r < kItems; r++) {
    const int idx = r * kThreads + (int)threadIdx.x;
    v2[r] = (idx < nvalid) ? stage32[idx] : 0u;
  }
  OS_STAMP(3)
  // ---- 5. earlier groups: their last tiles, kGroup rows apart
  if (dig) {
    int tt = gs - 1;
    while (!done && !fail) {
      bool closed = false, incl2 = false;
      const int used = lookback_round<kLook>(status, t, tt, kGroup, kGroup - 1, tag, 2u, false, excl, closed, incl2);
      tt -= used * kGroup;
      done = closed;
      if (!done && used == 0) {
        if (++spins > kSpinLimit) fail = true;
        __builtin_amdgcn_s_sleep(1);
      }
    }
    if (fail) atomicOr(&info->flags, BH_FLAG_SORT_TIMEOUT);
    if (tile != 0)
      __hip_atomic_store(mine, pack_granule(tag, 2u, excl + h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    gbase[t] = gpos0 + excl - lo;  // global position = gbase[digit] + index in the tile's digit-sorted order
  }
  __syncthreads();
  OS_STAMP(4)
  // ---- 6. the coalesced global writes
#pragma unroll
  for (int r = 0; r < kItems; r++) {
    const int idx = r * kThreads + (int)threadIdx.x;
    if (idx < nvalid) {
      const u32 pos = gbase[SPLIT ? dg[SPLIT ? r : 0] : ((u32)(k2[r] >> shift) & 255u)] + (u32)idx;
      if (pos < (u32)n) {  // always true unless a look-back timed out
        kout[pos] = k2[r];
        vout[pos] = v2[r];
      }
    }
  }
  OS_STAMP(5)
#undef OS_DIGIT
}

__global__ __launch_bounds__(256) void gather2_kernel(const u32* __restrict__ perm,
                                                      const float4* __restrict__ posm_in,
                                                      const float4* __restrict__ velid_in,
                                                      float4* __restrict__ posm_out,
                                                      float4* __restrict__ velid_out, int n,
                                                      u32* __restrict__ sw_hist, u32* __restrict__ sw_ticket) {
  // last kernel of the sort: leave the digit totals and the tile tickets cleared for the next call and
  // advance the call number (saves a memset launch, keeps every kernel argument of the step constant)
  if (blockIdx.x == 0) {
    for (int t = threadIdx.x; t < 8 * 256; t += 256) sw_hist[t] = 0u;
    if (threadIdx.x < 8) sw_ticket[threadIdx.x] = 0u;
    if (threadIdx.x == 8) sw_ticket[8] += 1u;
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32 j = perm[i];
  posm_out[i] = posm_in[j];
  velid_out[i] = velid_in[j];
}


// =====================================================================================================
// Splitter sort (bhk_sort_split): the sort of a STEP whose bodies are still stored in the previous step's key
// order.  At ~1M keys a radix pass is latency, not bandwidth (launch + look-back + the ranking chain: ~21 us
// of which ~3 us is data movement), so eight passes cost ~0.2 ms.  Instead:
//   1. keys_split_kernel  computes the keys AND picks <= 255 splitters — the keys of the bodies at evenly spaced
//      positions of the (nearly sorted) body array, so the buckets come out nearly equal — and counts the
//      bucket sizes;
//   2. ONE partition pass (onesweep_pass_kernel<true>) moves every key to its bucket, stably;
//   3. local_sort_kernel: one workgroup per bucket sorts it entirely in LDS (stable LSD radix passes over the top
//      32 of the bit positions that vary inside the bucket, runs of keys that agree on those ordered by neighbour
//      exchanges, the full set of passes if that does not finish: ls_sort_in_lds), writes the sorted keys and
//      gathers the bodies.
// Three kernels instead of eleven; result bit-identical to the radix sort (both are stable).  A bucket that does
// not fit LDS (order drifted a lot, or forced on random input) is sorted by its workgroup through global
// memory: slow, correct.  Not used for the first sort after an upload (no order to exploit).
constexpr int kLsThreads = 512;
constexpr int kLsWaves = kLsThreads / 64;
#ifndef BH_LS_CAP
#define BH_LS_CAP 12288  // 144 KB of the 160 KB LDS for keys + values
#endif
constexpr int kLsCap = BH_LS_CAP;  // keys per bucket sorted in LDS
#ifndef BH_LS_WINDOW
#define BH_LS_WINDOW 32  // key bits (below the highest one that varies in the bucket) sorted by radix passes
#endif
constexpr int kLsItems = kLsCap / kLsThreads;

constexpr int kKsThreads = 1024;  // x 4 keys: 4 waves per SIMD hide the LDS round trips of the bucket lookups
// TILE = keys per block: kTile (4 per thread), or 1024 (1 per thread) for small n, where kTile-sized blocks leave
// most of the 256 CUs idle (65,536 bodies: 16 blocks)
template <int B, int TILE>
__global__ __launch_bounds__(kKsThreads) void keys_split_kernel(const float4* __restrict__ posm,
                                                               const float* __restrict__ bounds, int n, int nb,
                                                               int curve, float nb_over_n,
                                                               u64* __restrict__ keys,
                                                               u64* __restrict__ splitters,
                                                               u32* __restrict__ bcount,
                                                               const int* __restrict__ n_dev = nullptr) {
  __shared__ u64 raw[256];
  __shared__ u64 sp[256];
  __shared__ u32 cnt[256];
  // n_dev: the body count is still on its way to the host (domain-decomposed step: bh_dd_migrate_apply launches this
  // kernel on an upper bound while it polls for the count the absorb kernel wrote) — taken from the device, with the
  // bucket count and the guess scale bhk_sort_split will derive from it on the host; blocks beyond it have nothing to do
  if (n_dev) {
    n = max(1, min(*n_dev, n));  // (n: the bound the grid was sized for — a count beyond the rank's capacity is an error
                                 // the host reports once it has seen it; nothing may be read out of bounds meanwhile)
    nb = (n + 511) / 512;
    nb = nb < 1 ? 1 : (nb > 256 ? 256 : nb);  // split_buckets
    nb_over_n = (float)nb / (float)n;
    if ((int)blockIdx.x * TILE >= n && blockIdx.x != 0) return;
  }
  const float minX = bounds[0], minY = bounds[1], minZ = bounds[2];
  const float size = bounds[6];  // fmaxf(bounds[3]-bounds[0], 1) ref:55
  const int tid = threadIdx.x, lane = tid & 63;
  KS_STAMP(0)
  // the tile's own bodies and the splitter candidates are loaded first; the Hilbert state table (bh_keys.h) is
  // staged in LDS meanwhile
  __shared__ u32 htab[B == 21 ? kHilbertTabWords : 1];
  const int base = blockIdx.x * TILE;
  float4 own[TILE / kKsThreads];
#pragma unroll
  for (int r = 0; r < TILE / kKsThreads; r++) {
    const int i = base + r * kKsThreads + tid;
    own[r] = posm[min(i, n - 1)];
  }
  float4 cand = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < 768 && (tid & 255) < nb - 1) {
    const int pos = (int)(((u64)((tid & 255) + 1) * (u64)n) / (u64)nb) + (tid >> 8) - 1;  // n / nb >= 2: inside [0, n)
    cand = posm[min(max(pos, 0), n - 1)];
  }
  if (B == 21) {
    hilbert_stage(htab);
    __syncthreads();
  }
  u64 k[TILE / kKsThreads];
#pragma unroll
  for (int r = 0; r < TILE / kKsThreads; r++) {
    const int i = base + r * kKsThreads + tid;
    k[r] = 0ull;
    if (i < n) {
      if constexpr (B == 21)
        k[r] = body_key21_fsm(htab, curve, own[r].x, own[r].y, own[r].z, minX, minY, minZ, size);
      else
        k[r] = body_key<B>(curve, own[r].x, own[r].y, own[r].z, minX, minY, minZ, size);
      keys[i] = k[r];
    }
  }
  KS_STAMP(1)
  // splitter t+1 = the MEDIAN of the keys of the three bodies stored around position (t+1) n / nb; unused slots
  // sort to the end.  (One body per splitter: a sample body that crossed a high-level cell plane last step — about
  // one in seventy does — takes its bucket boundary far away with it, and the two buckets beside it came out at
  // 0.25x and 2x the average: 938 / 3906 / 7791 keys at 1M, and the largest bucket is local_sort_kernel's length.)
  __shared__ u64 raw3[3][256];
  __shared__ u32 rnk[256];
  if (tid < 768) {
    const int t = tid & 255, j = tid >> 8;
    u64 sk3 = ~0ull;
    if (t < nb - 1) {
      if constexpr (B == 21)
        sk3 = body_key21_fsm(htab, curve, cand.x, cand.y, cand.z, minX, minY, minZ, size);
      else
        sk3 = body_key<B>(curve, cand.x, cand.y, cand.z, minX, minY, minZ, size);
    }
    raw3[j][t] = sk3;
  }
  __syncthreads();
  u64 sk = ~0ull;
  if (tid < 256) {
    const u64 a = raw3[0][tid], b = raw3[1][tid], c = raw3[2][tid];
    const u64 lo = a < b ? a : b, hi = a < b ? b : a;
    sk = c < lo ? lo : (c > hi ? hi : c);
    raw[tid] = sk;
    cnt[tid] = 0;
    rnk[tid] = 0;
  }
  __syncthreads();
  KS_STAMP(2)
  {  // rank sort of the <= 255 splitters, four threads per splitter (64 comparisons each: this serial loop is
     // most of the kernel at small n, where the grid is a handful of blocks)
    const int t = tid & 255, q = tid >> 8;
    const u64 mine = raw[t];
    int part = 0;
    for (int c = 64 * q; c < 64 * q + 64; c++) {
      const u64 o = raw[c];
      part += (o < mine || (o == mine && c < t)) ? 1 : 0;
    }
    if (part) atomicAdd(&rnk[t], (u32)part);
  }
  __syncthreads();
  if (tid < 256) sp[rnk[tid]] = sk;
  __syncthreads();
  if (blockIdx.x == 0 && tid < 256) splitters[tid] = sp[tid];
  KS_STAMP(3)
#pragma unroll
  for (int r = 0; r < TILE / kKsThreads; r++) {
    const int i = base + r * kKsThreads + tid;
    const bool valid = i < n;
    const u32 g = splitter_bucket_guess(sp, k[r], i, nb_over_n);
    const u64 act = __ballot(valid);
    if (act == 0ull) continue;
    const int l0 = __ffsll((long long)act) - 1;
    const u32 g0 = __shfl(g, l0, 64);
    if (__ballot(valid && g != g0) == 0ull) {
      if (lane == l0) atomicAdd(&cnt[g0], (u32)__popcll(act));
    } else if (valid) {
      atomicAdd(&cnt[g], 1u);
    }
  }
  __syncthreads();
  KS_STAMP(4)
  if (tid < 256 && cnt[tid]) atomicAdd(&bcount[tid], cnt[tid]);
  KS_STAMP(5)
}

// exclusive scan of one value per thread over the first 256 threads of the block (4 waves); all threads call
__device__ __forceinline__ u32 scan256(u32 v, u32* dsum, u32* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  u32 incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u32 u = __shfl_up(incl, d, 64);
    if (lane >= d) incl += u;
  }
  __syncthreads();  // dsum may still be in use
  if (w < 4 && lane == 63) dsum[w] = incl;
  __syncthreads();
  u32 wp = 0;
  for (int q = 0; q < (w & 3); q++) wp += dsum[q];
  if (total) *total = dsum[0] + dsum[1] + dsum[2] + dsum[3];
  return wp + incl - v;
}

// One bucket of <= ITEMS * kLsThreads keys sorted in LDS (keys in registers between the passes).  ITEMS = 8 covers a
// bucket of up to 4096 keys — the normal case: n / 256 is at most 6144 / 1.5 — with loops of 8; 16 and 24 take
// the buckets that came out up to 3x the average (neighbouring splitter bodies that both crossed a high-level
// cell plane; in the domain-decomposed step the immigrants, which cluster at the two ends of the rank's key range).
template <int ITEMS>
__device__ __forceinline__ void ls_sort_in_lds(u64* __restrict__ skey, u32* __restrict__ sval, u32 (*wcnt)[256],
                                               u32* __restrict__ toff, u32* __restrict__ dsum,
                                               u64* __restrict__ s_diff, const u64* kbuf, const u32* vbuf,
                                               u64* kout, u32* vout, const float4* __restrict__ posm_in,
                                               const float4* __restrict__ velid_in, float4* __restrict__ posm_out,
                                               float4* __restrict__ velid_out, int start, int size,
                                               int tid, int lane, int w, int b, u64 lt) {
    // ---- the bucket in registers, blocked by wave: wave w owns positions [w chunk, (w+1) chunk)
    const int chunk = ((size + kLsThreads - 1) / kLsThreads) * 64;
    const int nit = chunk / 64;
    const int wbase = w * chunk;
    u64 key[ITEMS];
    u32 val[ITEMS];
    u64 diff = 0ull;
    const u64 k0 = kbuf[start];
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
      key[r] = ~0ull;
      val[r] = 0u;
      if (r < nit) {
        const int idx = wbase + r * 64 + lane;
        if (idx < size) {
          key[r] = kbuf[start + idx];
          val[r] = vbuf[start + idx];
          diff |= key[r] ^ k0;
        }
      }
    }
    // digit positions that vary inside the bucket (the keys of a bucket share their leading bits)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) diff |= __shfl_xor(diff, d, 64);
    if (lane == 0) s_diff[w] = diff;
    __syncthreads();
    diff = 0ull;
#pragma unroll
    for (int q = 0; q < kLsWaves; q++) diff |= s_diff[q];
    LS_STAMP(1, wall_clock64())
    LS_STAMP(5, diff)

    // Which passes.  The keys of a bucket differ in ~55 bit positions (seven 8-bit passes), but ~4000 keys are
    // almost always told apart by a WINDOW of the top 32 of them: passes over the window only (4 instead of 7), then
    // one look at the neighbours — if no two adjacent keys agree on the window the order is final (the lower bits
    // cannot matter); runs of keys that do agree are ordered by neighbour exchanges; if those do not finish, the full
    // set of passes runs from the lowest varying byte, which is correct from any starting arrangement and keeps equal
    // keys in their original order because every pass is stable.  10 us less per sort at 1M.
    const int hb = (diff == 0ull) ? -1 : 63 - __clzll((long long)diff);  // highest varying bit
    // How wide the window has to be is read off the INPUT: the bucket arrives in the previous step's order, so its
    // adjacent keys are (nearly) the adjacent keys of the result, and a pair ties on a window exactly if its two keys
    // first differ below it.  Count, for three window widths, the adjacent input pairs that would tie; take the
    // narrowest window that leaves at most one pair in 128 tied (a few short runs for the exchanges below), none if
    // even the widest leaves more (a thin disc after some hundred steps, many coincident bodies): then the full
    // set of passes runs at once and no window pass is wasted.
    const int w0 = (size <= 1024) ? BH_LS_WINDOW - 8 : BH_LS_WINDOW;  // measured: 24 bits do for ~512-key buckets
    int lowbit = 0;
    if (hb >= w0) {  // block-uniform
      int c[3] = {0, 0, 0};
#pragma unroll
      for (int r = 0; r < ITEMS; r++) {
        if (r < nit) {
          const int idx = wbase + r * 64 + lane;
          const u64 nx = __shfl_down(key[r], 1, 64);
          const bool pair = lane < 63 && idx + 1 < size;
          const u64 x = key[r] ^ nx;
#pragma unroll
          for (int t = 0; t < 3; t++) {
            const int lb = hb - (w0 + 8 * t) + 1;
            c[t] += __popcll(__ballot(pair && lb > 0 && (x >> max(lb, 0)) == 0ull));
          }
        }
      }
      if (tid < 3) toff[tid] = 0;
      __syncthreads();
      if (lane == 0) {
#pragma unroll
        for (int t = 0; t < 3; t++)
          if (c[t]) atomicAdd(&toff[t], (u32)c[t]);
      }
      __syncthreads();
      const u32 few = (u32)(size >> 7);
#pragma unroll
      for (int t = 2; t >= 0; t--) {
        const int lb = hb - (w0 + 8 * t) + 1;
        if (lb >= 8 && toff[t] <= few) lowbit = lb;  // the narrowest acceptable window wins (t = 0 last)
      }
      __syncthreads();  // toff is a scratch of the passes
    }
    int phase = (lowbit > 0) ? 0 : 1;  // 0: the window [lowbit, hb]; 1: every varying byte from bit 0
    LS_STAMP(7, 0ull)
    int shift = lowbit;
    [[maybe_unused]] int npass = 0;
#pragma unroll 1
    for (;;) {
      while (shift <= hb && ((diff >> shift) & 255ull) == 0ull) shift += 8;  // a digit nobody differs in: block-uniform
      if (shift > hb) {
        if (phase == 1) break;
        int tie = 0;  // skey / key[] hold the bucket ordered by the window
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
          if (r < nit) {
            const int idx = wbase + r * 64 + lane;
            if (idx > 0 && idx < size) tie |= (((key[r] ^ skey[idx - 1]) >> lowbit) == 0ull) ? 1 : 0;
          }
        }
        if (!__syncthreads_or(tie)) {
          LS_STAMP(7, 1ull)
          break;
        }
        // Some neighbours agree on the whole window (a bucket that straddles a high-level cell plane has a high
        // top bit, and 32 bits below it do not reach the leaves of a dense region).  Their runs are short: order
        // each run by the full key with odd-even exchanges of neighbours inside a run (strict >: equal keys keep
        // their order).  A run longer than the round limit allows falls through to the full set of passes.
        bool done = false;
#pragma unroll 1
        for (int round = 0; round < 24 && !done; round++) {
          int swapped = 0;
#pragma unroll 1
          for (int par = 0; par < 2; par++) {
            for (int i = 2 * tid + par; i + 1 < size; i += 2 * kLsThreads) {
              const u64 ka = skey[i], kb = skey[i + 1];
              if (((ka ^ kb) >> lowbit) == 0ull && ka > kb) {
                const u32 va = sval[i], vb = sval[i + 1];
                skey[i] = kb; skey[i + 1] = ka;
                sval[i] = vb; sval[i + 1] = va;
                swapped = 1;
              }
            }
            __syncthreads();
          }
          done = !__syncthreads_or(swapped);
        }
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
          if (r < nit) {
            const int idx = wbase + r * 64 + lane;
            if (idx < size) {
              key[r] = skey[idx];
              val[r] = sval[idx];
            }
          }
        }
        LS_STAMP(7, done ? 2ull : 3ull)
        if (done) break;
        phase = 1;
        shift = 0;
        continue;
      }
#ifdef BH_OS_TRACE
      const bool tr = (npass == 1);
#define LS_PSTAMP(k) if (tr) { LS_STAMP(k, wall_clock64()) }
#else
#define LS_PSTAMP(k)
#endif
      npass++;
      LS_PSTAMP(8)
      for (int q = tid; q < kLsWaves * 256; q += kLsThreads) (&wcnt[0][0])[q] = 0;
      __syncthreads();
      LS_PSTAMP(9)
      u32 rk[ITEMS];
#pragma unroll
      for (int r = 0; r < ITEMS; r++) {
        rk[r] = 0;
        if (r < nit) {
          const int idx = wbase + r * 64 + lane;
          rk[r] = wave_rank((u32)(key[r] >> shift) & 255u, idx < size, wcnt[w], (u32)lt, (u32)(lt >> 32));
        }
      }
      __syncthreads();
      LS_PSTAMP(10)
      u32 h = 0;
      if (tid < 256) {
#pragma unroll
        for (int q = 0; q < kLsWaves; q++) {
          const u32 cq = wcnt[q][tid];
          wcnt[q][tid] = h;
          h += cq;
        }
      }
      const u32 lo = scan256(h, dsum, nullptr);
      if (tid < 256) toff[tid] = lo;
      __syncthreads();
      LS_PSTAMP(11)
#pragma unroll
      for (int r = 0; r < ITEMS; r++) {
        if (r < nit) {
          const int idx = wbase + r * 64 + lane;
          if (idx < size) {
            const u32 g = (u32)(key[r] >> shift) & 255u;
            const u32 lp = toff[g] + wcnt[w][g] + rk[r];
            skey[lp] = key[r];
            sval[lp] = val[r];
          }
        }
      }
      __syncthreads();
      LS_PSTAMP(12)
#pragma unroll
      for (int r = 0; r < ITEMS; r++) {
        if (r < nit) {
          const int idx = wbase + r * 64 + lane;
          if (idx < size) {
            key[r] = skey[idx];
            val[r] = sval[idx];
          }
        }
      }
      LS_PSTAMP(13)
      // (the barrier at the top of the next pass, or none needed after the last, orders these reads)
      shift += 8;
    }
    LS_STAMP(2, wall_clock64())
    LS_STAMP(6, (unsigned long long)npass)
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
      if (r < nit) {
        const int idx = wbase + r * 64 + lane;
        if (idx < size) {
          const u32 v = val[r];
          kout[start + idx] = key[r];
          vout[start + idx] = v;
          posm_out[start + idx] = posm_in[v];
          velid_out[start + idx] = velid_in[v];
        }
      }
    }
    LS_STAMP(3, wall_clock64())
}

__global__ __launch_bounds__(kLsThreads) void local_sort_kernel(
    u64* kbuf, u32* vbuf,    // the partitioned keys / values (scratch of the slow path)
    u64* kout, u32* vout,    // sorted keys / permutation
    const u32* __restrict__ bcount, u32* __restrict__ bcount_next,
    const float4* __restrict__ posm_in, const float4* __restrict__ velid_in, float4* __restrict__ posm_out,
    float4* __restrict__ velid_out, u32* __restrict__ sw_ticket, bh_devinfo* __restrict__ info) {
  __shared__ u64 skey[kLsCap];
  __shared__ u32 sval[kLsCap];
  __shared__ u32 wcnt[kLsWaves][256];
  __shared__ u32 toff[256];
  __shared__ u32 dsum[4];
  __shared__ u64 s_diff[kLsWaves];
  __shared__ u32 s_start, s_size;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int b = blockIdx.x;
  const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  {
    const u32 cv = (tid < 256) ? bcount[tid] : 0u;
    const u32 ex = scan256(cv, dsum, nullptr);
    if (tid == b) {
      s_start = ex;
      s_size = cv;
    }
  }
  if (b == 0) {
    // last kernel of the sort: leave the next call's bucket counters and the partition pass's tile ticket
    // cleared, advance the call number (tag of the look-back granules)
    if (tid < 256) bcount_next[tid] = 0u;
    if (tid == 0) {
      sw_ticket[0] = 0u;
      sw_ticket[8] += 1u;
    }
  }
  __syncthreads();
  const int start = (int)s_start, size = (int)s_size;
  LS_STAMP(0, wall_clock64())
  LS_STAMP(4, (unsigned long long)size)
  if (size == 0) return;

  if (size <= kLsCap) {  // block-uniform
    if (size <= 8 * kLsThreads)
      ls_sort_in_lds<8>(skey, sval, wcnt, toff, dsum, s_diff, kbuf, vbuf, kout, vout, posm_in, velid_in, posm_out,
                        velid_out, start, size, tid, lane, w, b, lt);
    else if (size <= 16 * kLsThreads)
      ls_sort_in_lds<16>(skey, sval, wcnt, toff, dsum, s_diff, kbuf, vbuf, kout, vout, posm_in, velid_in, posm_out,
                         velid_out, start, size, tid, lane, w, b, lt);
    else
      ls_sort_in_lds<kLsItems>(skey, sval, wcnt, toff, dsum, s_diff, kbuf, vbuf, kout, vout, posm_in, velid_in,
                               posm_out, velid_out, start, size, tid, lane, w, b, lt);
    return;
  }

  // ---- a bucket that does not fit LDS: stable LSD radix over all 8 digits by this one workgroup, ping-pong
  // between the bucket's own ranges of (kbuf, vbuf) and (kout, vout); 4096-key chunks in order, running
  // per-digit cursors in LDS.  Slow (one CU), only for buckets the splitters failed to balance.
  if (tid == 0) atomicAdd(&info->slow_buckets, 1);  // surfaces in bh_stats.sort_slow_buckets (bh_get_stats then
                                                     // sends this context's next sorts to the radix passes)
  u64* sk = kbuf + start;
  u32* sv = vbuf + start;
  u64* dk = kout + start;
  u32* dv = vout + start;
  u32* cur = toff;
  u32* cbase = reinterpret_cast<u32*>(skey);  // [256]
#pragma unroll 1
  for (int p = 0; p < 8; p++) {
    const int shift = 8 * p;
    if (tid < 256) cur[tid] = 0u;
    __syncthreads();
    for (int i = tid; i < size; i += kLsThreads) atomicAdd(&cur[(u32)(sk[i] >> shift) & 255u], 1u);
    __syncthreads();
    const u32 hv = (tid < 256) ? cur[tid] : 0u;
    const u32 ex = scan256(hv, dsum, nullptr);
    if (tid < 256) cur[tid] = ex;
    __syncthreads();
    constexpr int kChunk = kLsThreads * 8;
#pragma unroll 1
    for (int c0 = 0; c0 < size; c0 += kChunk) {
      for (int q = tid; q < kLsWaves * 256; q += kLsThreads) (&wcnt[0][0])[q] = 0;
      __syncthreads();
      u64 key[8];
      u32 val[8], rk[8];
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const int idx = c0 + w * 512 + r * 64 + lane;
        const bool valid = idx < size;
        key[r] = valid ? sk[idx] : ~0ull;
        val[r] = valid ? sv[idx] : 0u;
        rk[r] = wave_rank((u32)(key[r] >> shift) & 255u, valid, wcnt[w], (u32)lt, (u32)(lt >> 32));
      }
      __syncthreads();
      if (tid < 256) {
        u32 h = 0;
#pragma unroll
        for (int q = 0; q < kLsWaves; q++) {
          const u32 cq = wcnt[q][tid];
          wcnt[q][tid] = h;
          h += cq;
        }
        cbase[tid] = cur[tid];
        cur[tid] += h;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const int idx = c0 + w * 512 + r * 64 + lane;
        if (idx < size) {
          const u32 g = (u32)(key[r] >> shift) & 255u;
          const u32 pos = cbase[g] + wcnt[w][g] + rk[r];
          dk[pos] = key[r];
          dv[pos] = val[r];
        }
      }
      __syncthreads();
    }
    __threadfence();
    __syncthreads();
    u64* tk = sk; sk = dk; dk = tk;
    u32* tv = sv; sv = dv; dv = tv;
  }
  // eight passes: the sorted bucket is back in (kbuf, vbuf)
  for (int i = tid; i < size; i += kLsThreads) {
    const u32 v = sv[i];
    kout[start + i] = sk[i];
    vout[start + i] = v;
    posm_out[start + i] = posm_in[v];
    velid_out[start + i] = velid_in[v];
  }
}

}  // namespace

#ifdef BH_OS_TRACE
extern "C" int bh_debug_os_trace(void* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_os_trace), sizeof(g_os_trace));
}
extern "C" int bh_debug_ks_trace(void* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ks_trace), sizeof(g_ks_trace));
}
extern "C" int bh_debug_ls_trace(void* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ls_trace), sizeof(g_ls_trace));
}
#endif

// number of buckets of the splitter sort: ~512 keys each for small n, at most 256
static int split_buckets(int n) {
  int nb = (n + 511) / 512;
  return nb < 1 ? 1 : (nb > 256 ? 256 : nb);
}

// The splitter sort needs the bodies in (about) key order and buckets that fit LDS with room for drift.
bool bhk_sort_split_eligible(const bh_ctx* c, int n_upper) {
  const int v = c->p.sort_variant;
  if (v != 0 && v != 3) return false;
  if (c->p.step_graph == 1) return false;  // the call parity of the bucket counters is not a graph constant
  // (domain-decomposed contexts: n is the current local body count — or, n_upper, a bound on the count to come)
  if ((n_upper > 0 ? n_upper : c->n) > 256 * 6144) return false;
  // buckets that overflowed LDS were seen (bh_get_stats): many equal keys or an order that drifted too far — the
  // one-workgroup global-memory path is orders of magnitude slower than eight radix passes
  if (v != 3 && c->splitter_off) return false;
  return v == 3 || c->order_hint;
}

// n_dev / n_upper: launch before the host knows the body count (see keys_split_kernel): the grid covers n_upper bodies
hipError_t bhk_keys_split(bh_ctx* c, const int* n_dev, int n_upper) {
  const int n = n_dev ? n_upper : c->n;
  u32* bc = c->sp_count + 256 * (c->sp_par & 1);
  if (c->keys_split) {  // counted before and never consumed by a sort: start over
    const hipError_t e = hipMemsetAsync(bc, 0, 256 * sizeof(u32), c->stream);
    if (e != hipSuccess) return e;
  }
  const int nb = split_buckets(n);
  const bool small = n <= BH_KS_SMALL_N;  // one key per thread: four times the blocks (65,536 bodies: 64)
  const int grid = small ? (n + 1023) / 1024 : (n + kTile - 1) / kTile;
  if (c->B == 10) {
    if (small)
      keys_split_kernel<10, 1024><<<grid, kKsThreads, 0, c->stream>>>(
          c->posm[c->cur], c->bounds, n, nb, 0, (float)nb / (float)n, c->keys[0], c->sp_keys, bc, n_dev);
    else
      keys_split_kernel<10, kTile><<<grid, kKsThreads, 0, c->stream>>>(
          c->posm[c->cur], c->bounds, n, nb, 0, (float)nb / (float)n, c->keys[0], c->sp_keys, bc, n_dev);
  } else {
    if (small)
      keys_split_kernel<21, 1024><<<grid, kKsThreads, 0, c->stream>>>(
          c->posm[c->cur], c->bounds, n, nb, c->p.key_curve, (float)nb / (float)n, c->keys[0], c->sp_keys, bc, n_dev);
    else
      keys_split_kernel<21, kTile><<<grid, kKsThreads, 0, c->stream>>>(
          c->posm[c->cur], c->bounds, n, nb, c->p.key_curve, (float)nb / (float)n, c->keys[0], c->sp_keys, bc, n_dev);
  }
  c->keys_split = true;
  return hipGetLastError();
}

hipError_t bhk_sort_split(bh_ctx* c) {
  const int n = c->n;
  const int par = c->sp_par & 1;
  u32* bc = c->sp_count + 256 * par;
  c->sort_calls++;
  onesweep_pass_kernel<true><<<c->sort_tiles, kThreads, 0, c->stream>>>(
      c->keys[0], c->vals[0], c->keys[1], c->vals[1], n, 0, bc, c->sw_status, c->sw_ticket, c->sw_ticket + 8, 1,
      c->info, c->sp_keys, (float)split_buckets(n) / (float)n);
  local_sort_kernel<<<split_buckets(n), kLsThreads, 0, c->stream>>>(
      c->keys[1], c->vals[1], c->keys[0], c->vals[0], bc, c->sp_count + 256 * (par ^ 1), c->posm[c->cur],
      c->velid[c->cur], c->posm[c->cur ^ 1], c->velid[c->cur ^ 1], c->sw_ticket, c->info);
  c->key_buf = 0;
  c->cur ^= 1;
  c->sp_par ^= 1;
  c->keys_split = false;
  return hipGetLastError();
}

hipError_t bhk_sort_onesweep(bh_ctx* c) {
  const int n = c->n;
  const int ntiles = c->sort_tiles;
  const int passes = (c->p.key_bits + 7) / 8;
  // sw_hist is zero here: cleared at creation and by the gather kernel of the previous call
  onesweep_hist_kernel<<<ntiles, kHistThreads, 0, c->stream>>>(c->keys[0], n, passes, c->sw_hist);
  c->sort_calls++;
  int src = 0;
  for (int p = 0; p < passes; p++) {
    onesweep_pass_kernel<false><<<ntiles, kThreads, 0, c->stream>>>(
        c->keys[src], c->vals[src], c->keys[src ^ 1], c->vals[src ^ 1], n, 8 * p, c->sw_hist + p * 256,
        c->sw_status + (size_t)p * ntiles * 256, c->sw_ticket + p, c->sw_ticket + 8, p == 0, c->info, nullptr, 0.0f);
    src ^= 1;
  }
  c->key_buf = src;
  const int blocks = (n + 255) / 256;
  gather2_kernel<<<blocks, 256, 0, c->stream>>>(c->vals[src], c->posm[c->cur], c->velid[c->cur],
                                                 c->posm[c->cur ^ 1], c->velid[c->cur ^ 1], n, c->sw_hist, c->sw_ticket);
  c->cur ^= 1;
  return hipGetLastError();
}
