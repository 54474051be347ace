// bh_sort_onesweep.hip — stable LSD radix sort, ONE kernel per 8-bit pass (default sort path).
//
// Replaces thrust::sort_by_key (ref nbody_v5_bench.cu:262-264).  Same contract as bh_sort.hip
// (stable ascending by key, value = slot index), a third of the launches and two thirds of the
// traffic: keys and values are read once and written once per pass.
//   * one up-front kernel builds the global digit histograms of ALL passes (global digit totals do
//     not depend on the order of the keys);
//   * each pass kernel ranks its 4096-key tile exactly like sort_scatter_kernel (wave64 match-any
//     with 8 ballots, wave-private LDS counters combined in wave order -> stable), then resolves
//     the tile's per-digit prefix by DECOUPLED LOOK-BACK instead of a separate histogram + scan:
//       - tiles take a ticket (atomicAdd) when they start, so a tile's predecessors are already
//         running: look-back only ever waits on resident workgroups (no dispatch-order assumption);
//       - per (tile, digit) ONE 8-byte granule {tag:30 | state:2 | count:32} written and polled with
//         relaxed agent-scope atomics (the data-tagged single-granule hand-off of
//         cdna_hip_programming.md Guideline 16: flag and value travel in one word, so no
//         fence ordering between them is needed).  state 1 = this tile's count, 2 = inclusive
//         prefix.  The tag is the sort-call number, so the table is never cleared and a stale
//         entry from an earlier call reads as "not ready";
//       - thread d walks back over digit d's granules (256 independent chains per block);
//       - every spin is bounded; on timeout the kernel sets BH_FLAG_SORT_TIMEOUT and goes on
//         (wrong order, loudly reported) instead of hanging the device.
#include "bh_internal.h"

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
#else
#define OS_STAMP(k)
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

__global__ __launch_bounds__(kThreads) void onesweep_pass_kernel(
    const u64* __restrict__ kin, const u32* __restrict__ vin, u64* __restrict__ kout,
    u32* __restrict__ vout, int n, int shift, const u32* __restrict__ ghist_pass,
    u64* __restrict__ status, u32* __restrict__ ticket, const u32* __restrict__ call_ptr, int first_pass,
    bh_devinfo* __restrict__ info) {
  // tag = number of this sort call, kept on the device (sw_ticket[8], advanced by the gather kernel that ends
  // every sort) so that the kernel arguments of a step never change: bh_step replays as a HIP graph
  const u32 tag = (*call_ptr + 1u) & 0x3fffffffu;
  __shared__ u32 wcnt[kWaves][256];
  __shared__ u32 gbase[256];
  __shared__ u32 toff[256];
  __shared__ u32 dsum[4];
  __shared__ u32 s_tile;
  __shared__ u64 stage[kTile];  // 32 KB: the tile in digit order (keys, then values as u32)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);  // tickets restart at 0: the gather kernel clears them
  for (int q = threadIdx.x; q < kWaves * 256; q += kThreads) (&wcnt[0][0])[q] = 0;
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
#pragma unroll
  for (int r = 0; r < kItems; r++) {
    const int i = base + r * 64 + lane;
    const bool valid = i < n;
    const u32 g = (u32)(key[r] >> shift) & 255u;
    u64 mask = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; bit++) {
      const bool b = (g >> bit) & 1u;
      const u64 bb = __ballot(b);
      mask &= b ? bb : ~bb;
    }
    const u32 rank = (u32)__popcll(mask & lt);
    u32 prev = 0;
    if (valid) prev = wcnt[w][g];
    rk[r] = prev + rank;
    if (valid && rank == 0) wcnt[w][g] = prev + (u32)__popcll(mask);
  }
  __syncthreads();
  OS_STAMP(1)
  {
    // threads 0..255 own one digit each (the first four waves); the other waves only keep the barriers company
    const bool dig = threadIdx.x < 256;
    const int t = threadIdx.x & 255;  // digit
    u32 h = 0, excl = 0;
    if (dig) {
#pragma unroll
      for (int q = 0; q < kWaves; q++) {
        const u32 cq = wcnt[q][t];
        wcnt[q][t] = h;  // keys of digit t in the earlier waves of this tile
        h += cq;
      }
      // Publish this tile's count, then resolve the exclusive prefix over the earlier tiles.  When n / tile ~
      // number of CUs every tile is resident at once and nobody has an inclusive prefix early: a plain
      // look-back then reads O(tiles) granule rows per tile (61 MB per pass at 1M keys, more than the keys).
      // Two levels instead: tiles form groups of kGroup; a tile first sums the counts of its own group's
      // earlier tiles (consecutive rows) and publishes that as state 3 (group-inclusive), then walks back over
      // the LAST tile of every earlier group only (rows kGroup apart), which carry state 3 or 2.
      constexpr int kLook = BH_OS_LOOK, kGroup = BH_OS_GROUP;
      u64* mine = status + (size_t)tile * 256 + t;
      const int gs = tile - tile % kGroup;
      u32 spins = 0;
      bool done = false, fail = false;
      if (tile == 0) {
        done = true;
      } else if (tile == gs) {
        __hip_atomic_store(mine, pack_granule(tag, 3u, h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        __hip_atomic_store(mine, pack_granule(tag, 1u, h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int tt = tile - 1;
        bool grp = false;
        while (!grp && !fail) {
          u64 e[kLook];
#pragma unroll
          for (int q = 0; q < kLook; q++)
            e[q] = (tt - q >= gs) ? __hip_atomic_load(status + (size_t)(tt - q) * 256 + t, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT)
                                  : 0ull;
          int used = 0;
#pragma unroll
          for (int q = 0; q < kLook; q++) {
            if (grp || used != q) continue;  // stop at the first unpublished granule
            const u32 hi = (u32)(e[q] >> 32);
            if (tt - q < gs || (hi >> 2) != tag || (hi & 3u) == 0u) continue;
            excl += (u32)e[q];
            used = q + 1;
            if ((hi & 3u) == 2u) done = true;
            if ((hi & 3u) >= 2u || tt - q == gs) grp = true;
          }
          tt -= used;
          if (!grp && used == 0) {
            if (++spins > kSpinLimit) fail = true;
            __builtin_amdgcn_s_sleep(1);
          }
        }
        if (!done && !fail)
          __hip_atomic_store(mine, pack_granule(tag, 3u, excl + h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (gs == 0) done = true;  // group 0: group-inclusive is inclusive
      {
        int tt = gs - 1;  // last tile of the previous group
        while (!done && !fail) {
          u64 e[kLook];
#pragma unroll
          for (int q = 0; q < kLook; q++)
            e[q] = (tt - q * kGroup >= 0) ? __hip_atomic_load(status + (size_t)(tt - q * kGroup) * 256 + t,
                                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                          : 0ull;
          int used = 0;
#pragma unroll
          for (int q = 0; q < kLook; q++) {
            if (done || used != q) continue;  // stop at the first granule that is not (group-)inclusive yet
            const u32 hi = (u32)(e[q] >> 32);
            if (tt - q * kGroup < 0 || (hi >> 2) != tag || (hi & 3u) < 2u) continue;
            excl += (u32)e[q];
            used = q + 1;
            if ((hi & 3u) == 2u || tt - q * kGroup < kGroup) done = true;  // group 0's state 3 is inclusive too
          }
          tt -= used * kGroup;
          if (!done && used == 0) {
            if (++spins > kSpinLimit) fail = true;
            __builtin_amdgcn_s_sleep(1);
          }
        }
      }
      OS_STAMP(2)
      if (fail) atomicOr(&info->flags, BH_FLAG_SORT_TIMEOUT);
      __hip_atomic_store(mine, pack_granule(tag, 2u, excl + h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // digit base = exclusive scan of the 256 global digit totals of this pass
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
    const u32 gpos = wp + incl - dv + excl;  // global position of this tile's first key of digit t
    // tile-local digit offsets: exclusive scan of the tile's digit counts
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
    const u32 lo = wp2 + inc2 - h;
    if (dig) {
      toff[t] = lo;
      gbase[t] = gpos - lo;  // global position = gbase[digit] + index in the tile's digit-sorted order
    }
  }
  __syncthreads();
  OS_STAMP(3)
  // Stage the tile through LDS in digit order so that each digit's run leaves as one contiguous
  // (coalesced) global write instead of 16 scattered 8-byte stores; keys first, then the values
  // through the same buffer.
  const int nvalid = min(kTile, n - tile * kTile);
  u32 lp[kItems];
#pragma unroll
  for (int r = 0; r < kItems; r++) {
    const int i = base + r * 64 + lane;
    const u32 g = (u32)(key[r] >> shift) & 255u;
    lp[r] = toff[g] + wcnt[w][g] + rk[r];
    if (i < n) stage[lp[r]] = key[r];
  }
  __syncthreads();
  u32 gp[kItems];
#pragma unroll
  for (int r = 0; r < kItems; r++) {
    const int idx = r * kThreads + (int)threadIdx.x;
    gp[r] = 0xffffffffu;
    if (idx < nvalid) {
      const u64 k = stage[idx];
      const u32 pos = gbase[(u32)(k >> shift) & 255u] + (u32)idx;
      if (pos < (u32)n) {  // always true unless a look-back timed out
        kout[pos] = k;
        gp[r] = pos;
      }
    }
  }
  __syncthreads();
  OS_STAMP(4)
  u32* stage32 = reinterpret_cast<u32*>(stage);
#pragma unroll
  for (int r = 0; r < kItems; r++) {
    const int i = base + r * 64 + lane;
    if (i < n) stage32[lp[r]] = val[r];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kItems; r++) {
    const int idx = r * kThreads + (int)threadIdx.x;
    if (gp[r] != 0xffffffffu) vout[gp[r]] = stage32[idx];
  }
  OS_STAMP(5)
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

}  // namespace

#ifdef BH_OS_TRACE
extern "C" int bh_debug_os_trace(void* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_os_trace), sizeof(g_os_trace));
}
#endif

hipError_t bhk_sort_onesweep(bh_ctx* c) {
  const int n = c->n;
  const int ntiles = c->sort_tiles;
  const int passes = (c->p.key_bits + 7) / 8;
  // sw_hist is zero here: cleared at creation and by the gather kernel of the previous call
  onesweep_hist_kernel<<<ntiles, kHistThreads, 0, c->stream>>>(c->keys[0], n, passes, c->sw_hist);
  c->sort_calls++;
  int src = 0;
  for (int p = 0; p < passes; p++) {
    onesweep_pass_kernel<<<ntiles, kThreads, 0, c->stream>>>(
        c->keys[src], c->vals[src], c->keys[src ^ 1], c->vals[src ^ 1], n, 8 * p, c->sw_hist + p * 256,
        c->sw_status + (size_t)p * ntiles * 256, c->sw_ticket + p, c->sw_ticket + 8, p == 0, c->info);
    src ^= 1;
  }
  c->key_buf = src;
  const int blocks = (n + 255) / 256;
  gather2_kernel<<<blocks, 256, 0, c->stream>>>(c->vals[src], c->posm[c->cur], c->velid[c->cur],
                                                 c->posm[c->cur ^ 1], c->velid[c->cur ^ 1], n, c->sw_hist, c->sw_ticket);
  c->cur ^= 1;
  return hipGetLastError();
}
