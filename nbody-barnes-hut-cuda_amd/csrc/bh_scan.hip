// bh_scan.hip — device-wide exclusive prefix sums (reduce-then-scan, 2 launches: the block of the reduce
// kernel that finishes last scans the tile sums — bh_last_block, bh_internal.h).
//
// Used by the octree build (cell counts -> cell ids, child counts -> child-block offsets)
// and by the centre-of-mass stage (fp64 prefix of (m, m*x, m*y, m*z) over the Morton-sorted
// bodies).  No atomics, no decoupled look-back: the association order is fixed by the tile
// structure, so results are bit-reproducible run to run.
// Tile = 256 threads x 8 items, blocked per thread (32 contiguous bytes per lane for int32:
// two dwordx4 accesses; 256 B per lane for the fp64 quadruple).  Wave-level steps are
// 64-lane __shfl_up; the four wave totals of a block go through LDS.
#include "bh_internal.h"
#include "bh_scan_body.h"

namespace {
using namespace bhscan;

template <typename Op, typename Load>
__global__ __launch_bounds__(kThreads) void scan_reduce_kernel(Load load, int n_static,
                                                               const int* __restrict__ n_dev,
                                                               typename Op::T* tile_sums, u32* __restrict__ done) {
  reduce_body<Op, Load>((int)blockIdx.x, (int)gridDim.x, load, n_static, n_dev, tile_sums, done);
}

template <typename Op, typename Load>
__global__ __launch_bounds__(kThreads) void scan_apply_kernel(Load load, int n_static,
                                                              const int* __restrict__ n_dev,
                                                              const typename Op::T* __restrict__ tile_sums,
                                                              int ntiles,
                                                              typename Op::T* __restrict__ out) {
  apply_body<Op, Load>((int)blockIdx.x, load, n_static, n_dev, tile_sums, ntiles, out);
}

template <typename Op, typename Load>
hipError_t run_scan(hipStream_t stream, void* tmp, size_t cnt_off, Load load, typename Op::T* out, int n,
                    const int* n_dev) {
  typedef typename Op::T T;
  const int ntiles = (n + kTile - 1) / kTile;
  T* sums = reinterpret_cast<T*>(tmp);
  u32* done = reinterpret_cast<u32*>(reinterpret_cast<char*>(tmp) + cnt_off);  // zero between scans
  scan_reduce_kernel<Op, Load><<<ntiles, kThreads, 0, stream>>>(load, n, n_dev, sums, done);
  scan_apply_kernel<Op, Load><<<ntiles, kThreads, 0, stream>>>(load, n, n_dev, sums, ntiles, out);
  return hipGetLastError();
}

}  // namespace

// scratch of one scan over up to n elements: [ntiles + 2] tile sums (largest element type), then the block
// counters of bh_last_block (must be zero at creation; every scan leaves them zero)
size_t bhk_scan_cnt_offset(int n) {
  const size_t ntiles = ((size_t)n + kTile - 1) / kTile;
  return (ntiles + 2) * sizeof(bh_d4);
}
size_t bhk_scan_tmp_bytes(int n) {
  const size_t ntiles = ((size_t)n + kTile - 1) / kTile;
  // int32 scans run over up to 3n + 8 elements (record pool) in the same scratch: 4x the groups
  return bhk_scan_cnt_offset(n) + (ntiles / 8 + 8) * sizeof(u32);
}

hipError_t bhk_scan_i32(bh_ctx* c, const int* in, int* out, int n, const int* n_dev) {
  return run_scan<OpI32>(c->stream, c->scan_tmp, c->scan_cnt_off, LoadI32{in}, out, n, n_dev);
}

hipError_t bhk_scan_i32_even(bh_ctx* c, const int* in, int* out, int n) {
  return run_scan<OpI32>(c->stream, c->scan_tmp, c->scan_cnt_off, LoadI32Even{in}, out, n, nullptr);
}

hipError_t bhk_scan_pm(bh_ctx* c, const float4* posm, bh_d4* out, int n) {
  return run_scan<OpD4>(c->stream, c->scan_tmp, c->scan_cnt_off, LoadPM{posm}, out, n, nullptr);
}
