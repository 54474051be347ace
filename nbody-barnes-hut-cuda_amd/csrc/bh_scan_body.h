// bh_scan_body.h — the device side of the prefix sums of bh_scan.hip (reduce-then-scan over tiles of 256 threads x 8
// items, fixed association order: bit-reproducible), as functions of a block index so that a launch can do a scan's
// tiles next to other work (bh_tree.hip: the fp64 COM prefix scan of a small step rides in the launches of the tree
// build instead of two launches of its own).
#pragma once
#include "bh_internal.h"

namespace bhscan {


constexpr int kThreads = 256;
constexpr int kItems = 8;
constexpr int kTile = kThreads * kItems;  // == BH_SCAN_TILE

// ---- element ops ----
struct OpI32 {
  typedef int T;
  static __device__ __forceinline__ T zero() { return 0; }
  static __device__ __forceinline__ T add(T a, T b) { return a + b; }
  static __device__ __forceinline__ T shfl_up(T v, int d) { return __shfl_up(v, d, 64); }
  static __device__ __forceinline__ T shfl(T v, int l) { return __shfl(v, l, 64); }
  static __device__ __forceinline__ void publish(T* p, T v) { bh_publish_i32(p, v); }
  static __device__ __forceinline__ T collect(const T* p) { return bh_collect_i32(p); }
};
struct OpD4 {
  typedef bh_d4 T;
  static __device__ __forceinline__ T zero() { return bh_d4{0.0, 0.0, 0.0, 0.0}; }
  static __device__ __forceinline__ T add(T a, T b) {
    return bh_d4{a.m + b.m, a.x + b.x, a.y + b.y, a.z + b.z};
  }
  static __device__ __forceinline__ T shfl_up(T v, int d) {
    return bh_d4{__shfl_up(v.m, d, 64), __shfl_up(v.x, d, 64), __shfl_up(v.y, d, 64),
                 __shfl_up(v.z, d, 64)};
  }
  static __device__ __forceinline__ T shfl(T v, int l) {
    return bh_d4{__shfl(v.m, l, 64), __shfl(v.x, l, 64), __shfl(v.y, l, 64), __shfl(v.z, l, 64)};
  }
  static __device__ __forceinline__ void publish(T* p, T v) {
    double* q = reinterpret_cast<double*>(p);
    __hip_atomic_store(q + 0, v.m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 2, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 3, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  static __device__ __forceinline__ T collect(const T* p) {
    const double* q = reinterpret_cast<const double*>(p);
    return bh_d4{__hip_atomic_load(q + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                 __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                 __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                 __hip_atomic_load(q + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)};
  }
};

// ---- loaders ----
struct LoadI32 {
  const int* p;
  __device__ __forceinline__ int operator()(int i) const { return p[i]; }
};
struct LoadI32Even {  // child counts rounded up to even: child blocks start on 64-byte boundaries
  const int* p;
  __device__ __forceinline__ int operator()(int i) const { return (p[i] + 1) & ~1; }
};
struct LoadPM {  // body i -> (m, m x, m y, m z) in fp64; the products of two fp32 are exact in fp64
  const float4* posm;
  __device__ __forceinline__ bh_d4 operator()(int i) const {
    float4 q = posm[i];
    double m = (double)q.w;
    return bh_d4{m, m * (double)q.x, m * (double)q.y, m * (double)q.z};
  }
};

// inclusive scan across the 64 lanes of a wave
template <typename Op>
__device__ __forceinline__ typename Op::T wave_inclusive(typename Op::T v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    typename Op::T t = Op::shfl_up(v, d);
    if (lane >= d) v = Op::add(t, v);
  }
  return v;
}

// exclusive scan of one value per thread across a block of NT threads (NT/64 waves).
// Returns the exclusive prefix of this thread; *total receives the block total.
template <typename Op, int NT>
__device__ __forceinline__ typename Op::T block_exclusive(typename Op::T v, typename Op::T* lds,
                                                          typename Op::T* total) {
  typedef typename Op::T T;
  constexpr int NW = NT / 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  T incl = wave_inclusive<Op>(v, lane);
  T excl = Op::shfl_up(incl, 1);
  if (lane == 0) excl = Op::zero();
  if (lane == 63) lds[w] = incl;
  __syncthreads();
  T wprefix = Op::zero(), tot = Op::zero();
#pragma unroll
  for (int i = 0; i < NW; i++) {
    T s = lds[i];
    if (i < w) wprefix = Op::add(wprefix, s);
    tot = Op::add(tot, s);
  }
  __syncthreads();
  *total = tot;
  return Op::add(wprefix, excl);
}

// tile sums; the block that finishes last turns them into the exclusive tile prefixes in place
// (total -> tile_sums[ntiles]).  bid / nblk: this block's tile and the number of tiles — a kernel's block index and
// grid size, or a slice of a launch that also does something else (bh_tree.hip: lcp_scan_reduce_kernel)
template <typename Op, typename Load>
__device__ __forceinline__ void reduce_body(const int bid, const int nblk, Load load, int n_static,
                                            const int* __restrict__ n_dev, typename Op::T* tile_sums,
                                            u32* __restrict__ done) {
  typedef typename Op::T T;
  __shared__ T lds[kThreads / 64];
  __shared__ int s_last;
  const int n = n_dev ? *n_dev : n_static;
  const int i0 = bid * kTile + threadIdx.x * kItems;
  T s = Op::zero();
#pragma unroll
  for (int k = 0; k < kItems; k++)
    if (i0 + k < n) s = Op::add(s, load(i0 + k));
  T tot;
  (void)block_exclusive<Op, kThreads>(s, lds, &tot);
  if (threadIdx.x == 0) {
    Op::publish(tile_sums + bid, tot);
    bh_published();
    s_last = bh_last_block(done, bid, nblk) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  const int ntiles = nblk;
  T carry = Op::zero();
  // a serial tail of the kernel: four tile sums per thread, loaded together (one agent-scope round trip per
  // 1024 tiles), summed in registers, one block scan
  constexpr int kPer = 4;
  for (int c0 = 0; c0 < ntiles; c0 += kThreads * kPer) {
    const int i0 = c0 + (int)threadIdx.x * kPer;
    T v[kPer];
#pragma unroll
    for (int q = 0; q < kPer; q++) v[q] = (i0 + q < ntiles) ? Op::collect(tile_sums + i0 + q) : Op::zero();
    T sum = Op::zero();
#pragma unroll
    for (int q = 0; q < kPer; q++) sum = Op::add(sum, v[q]);
    T t2;
    T run = Op::add(carry, block_exclusive<Op, kThreads>(sum, lds, &t2));
#pragma unroll
    for (int q = 0; q < kPer; q++) {
      if (i0 + q < ntiles) tile_sums[i0 + q] = run;
      run = Op::add(run, v[q]);
    }
    carry = Op::add(carry, t2);
  }
  if (threadIdx.x == 0) tile_sums[ntiles] = carry;
}

template <typename Op, typename Load>
__device__ __forceinline__ void apply_body(const int bid, Load load, int n_static, const int* __restrict__ n_dev,
                                           const typename Op::T* __restrict__ tile_sums, int ntiles,
                                           typename Op::T* __restrict__ out) {
  typedef typename Op::T T;
  __shared__ T lds[kThreads / 64];
  const int n = n_dev ? *n_dev : n_static;
  const int i0 = bid * kTile + threadIdx.x * kItems;
  T v[kItems];
  T s = Op::zero();
#pragma unroll
  for (int k = 0; k < kItems; k++) {
    v[k] = (i0 + k < n) ? load(i0 + k) : Op::zero();
    s = Op::add(s, v[k]);
  }
  T tot;
  T ex = block_exclusive<Op, kThreads>(s, lds, &tot);
  T run = Op::add(tile_sums[bid], ex);
#pragma unroll
  for (int k = 0; k < kItems; k++) {
    if (i0 + k < n) out[i0 + k] = run;
    run = Op::add(run, v[k]);
  }
  if (bid == 0 && threadIdx.x == 0) out[n] = tile_sums[ntiles];
}


}  // namespace bhscan
