// bh_sort.hip — stable LSD radix sort of (Morton key, slot index) pairs + physical gather.
//
// Replaces thrust::sort_by_key (ref nbody_v5_bench.cu:262-264; CUB onesweep, not vendored).
// Contract: result == stable ascending sort by key.  8-bit digits; per pass
//   1. tile histogram (LDS integer atomics, 256 bins, tile = 256 threads x 16 keys),
//   2. per-digit exclusive scan of the digit-major [256][ntiles] table (one block per digit);
//      the 256 digit totals are scanned again inside every scatter block (cheap, no launch),
//   3. stable scatter: each wave ranks its 64 keys per round with 8 wave64 ballots
//      (match-any on the digit bits) against a wave-private LDS digit counter; the four
//      waves' counters are then offset in wave order, which keeps the sort stable.
// After the last pass the particle state (posm, velid) is physically gathered into Morton
// order (the reference only permutes an index array).
#include "bh_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kItems = BH_SORT_ITEMS;
constexpr int kTile = BH_SORT_TILE;

__global__ __launch_bounds__(kThreads) void sort_hist_kernel(const u64* __restrict__ keys, int n,
                                                             int shift, u32* __restrict__ hist,
                                                             int ntiles) {
  __shared__ u32 h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int base = blockIdx.x * kTile;
#pragma unroll
  for (int r = 0; r < kItems; r++) {
    const int i = base + r * kThreads + (int)threadIdx.x;
    if (i < n) atomicAdd(&h[(u32)(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

// one block per digit: exclusive scan of that digit's row hist[digit][0..ntiles) in place;
// the row total goes to tot[digit] (the scatter kernel turns the 256 totals into digit bases)
__global__ __launch_bounds__(kThreads) void sort_rowscan_kernel(u32* __restrict__ hist, int ntiles,
                                                                u32* __restrict__ tot) {
  __shared__ u32 wsum[4];
  u32* row = hist + (size_t)blockIdx.x * ntiles;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  u32 carry = 0;
  for (int c0 = 0; c0 < ntiles; c0 += kThreads) {
    const int i = c0 + t;
    const u32 v = (i < ntiles) ? row[i] : 0u;
    u32 incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const u32 u = __shfl_up(incl, d, 64);
      if (lane >= d) incl += u;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    u32 wp = 0, total = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const u32 sv = wsum[q];
      if (q < w) wp += sv;
      total += sv;
    }
    __syncthreads();
    if (i < ntiles) row[i] = carry + wp + incl - v;
    carry += total;
  }
  if (t == 0) tot[blockIdx.x] = carry;
}

__global__ __launch_bounds__(kThreads) void sort_scatter_kernel(
    const u64* __restrict__ kin, const u32* __restrict__ vin, u64* __restrict__ kout,
    u32* __restrict__ vout, int n, int shift, const u32* __restrict__ hist,
    const u32* __restrict__ tot, int ntiles, int first_pass) {
  __shared__ u32 wcnt[4][256];
  __shared__ u32 gbase[256];
  __shared__ u32 dsum[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int q = 0; q < 4; q++) wcnt[q][threadIdx.x] = 0;
  __syncthreads();

  const int base = blockIdx.x * kTile + w * (64 * kItems);
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
  {
    const int t = threadIdx.x;
    const u32 c0 = wcnt[0][t], c1 = wcnt[1][t], c2 = wcnt[2][t];
    wcnt[0][t] = 0;
    wcnt[1][t] = c0;
    wcnt[2][t] = c0 + c1;
    wcnt[3][t] = c0 + c1 + c2;
    // digit base = exclusive scan of the 256 digit totals (block scan), + this tile's row prefix
    const u32 dv = tot[t];
    u32 incl = dv;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const u32 u = __shfl_up(incl, d, 64);
      if (lane >= d) incl += u;
    }
    if (lane == 63) dsum[w] = incl;
    __syncthreads();
    u32 wp = 0;
    for (int q = 0; q < w; q++) wp += dsum[q];
    gbase[t] = wp + incl - dv + hist[(size_t)t * ntiles + blockIdx.x];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kItems; r++) {
    const int i = base + r * 64 + lane;
    if (i < n) {
      const u32 g = (u32)(key[r] >> shift) & 255u;
      const u32 pos = gbase[g] + wcnt[w][g] + rk[r];
      kout[pos] = key[r];
      vout[pos] = val[r];
    }
  }
}

// physically reorder the particle state: slot i <- old slot perm[i]
__global__ __launch_bounds__(256) void gather_kernel(const u32* __restrict__ perm,
                                                     const float4* __restrict__ posm_in,
                                                     const float4* __restrict__ velid_in,
                                                     float4* __restrict__ posm_out,
                                                     float4* __restrict__ velid_out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32 j = perm[i];
  posm_out[i] = posm_in[j];
  velid_out[i] = velid_in[j];
}

}  // namespace

hipError_t bhk_sort(bh_ctx* c) {
  c->order_hint = true;  // the bodies leave every sort in key order
  if (c->keys_split) return bhk_sort_split(c);  // the keys came with splitters and bucket counts
  if (c->p.sort_variant != 1) return bhk_sort_onesweep(c);
  const int n = c->n;
  const int ntiles = c->sort_tiles;
  const int passes = (c->p.key_bits + 7) / 8;
  int src = 0;
  for (int p = 0; p < passes; p++) {
    const int shift = 8 * p;
    sort_hist_kernel<<<ntiles, kThreads, 0, c->stream>>>(c->keys[src], n, shift, c->hist, ntiles);
    u32* tot = c->hist + (size_t)256 * ntiles;  // 256 digit totals behind the table
    sort_rowscan_kernel<<<256, kThreads, 0, c->stream>>>(c->hist, ntiles, tot);
    sort_scatter_kernel<<<ntiles, kThreads, 0, c->stream>>>(c->keys[src], c->vals[src],
                                                            c->keys[src ^ 1], c->vals[src ^ 1], n,
                                                            shift, c->hist, tot, ntiles, p == 0);
    src ^= 1;
  }
  c->key_buf = src;
  const int blocks = (n + 255) / 256;
  gather_kernel<<<blocks, 256, 0, c->stream>>>(c->vals[src], c->posm[c->cur], c->velid[c->cur],
                                                c->posm[c->cur ^ 1], c->velid[c->cur ^ 1], n);
  c->cur ^= 1;
  return hipGetLastError();
}
