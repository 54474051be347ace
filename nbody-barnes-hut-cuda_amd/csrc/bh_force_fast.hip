// bh_force_fast.hip — experimental variant of the default force kernel (force_variant = 3).
// MEASURED 4 % SLOWER than force_fast_kernel (1.89 vs 1.81 ms at 1M, theta 0.5) although it
// executes fewer scalar instructions and branches; kept for A/B and as a record of the experiment.
//
// Same algorithm and data as force_fast_kernel in bh_force.hip (see the header there: one wave64 =
// 64 Morton-consecutive bodies walking one shared depth-first traversal; wave-uniform records via
// scalar loads; per-lane MAC; lane masks in SGPR pairs; stack in registers across lanes; 15 VALU per
// (record, wave) on the pre-digested bh_frec records).  The kernel is instruction-issue bound
// (~31 instructions per (record, wave), half of them scalar/branch), so this version
// trims everything around the 15 VALU:
//   * a stack entry is 3 words (first | (count-1) << 28, mask lo, mask hi): 3 v_readlane per pop,
//     3 v_writelane per push; entries 0..63 (the common case) are reached without a set switch;
//   * record addresses are 32-bit offsets from the SGPR base (s_lshl + s_add/s_addc);
//   * a block of children is dispatched on its size (4 / 3 / 2 / 1 records per chunk) instead of
//     one guard per record;
//   * opened cells are handled once per chunk: the per-record open masks are OR-ed and tested once,
//     so the common "nothing opened" chunk pays one branch instead of four.
// Per-lane summation order = that lane's own depth-first order (chunk records ascending, then the
// chunk's multi-body leaves, children pushed ascending / popped descending): it does not depend on
// which other bodies share the wave, so any Morton-slab split reproduces the same bits.
// Reference: computeForceKernel nbody_v5_bench.cu:191-225 (intended recurrence).
#include "bh_internal.h"

namespace {

typedef __attribute__((address_space(4))) const float cfloat_t;
typedef float float8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(4))) const float8_t cfloat8_t;
typedef float float4v_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(4))) const float4v_t cfloat4_t;

constexpr int kStackCap = 192;  // 3 x 64 lanes >= 7*21+1
constexpr int kXcdRun = 16;

__device__ __forceinline__ int block_chunk(int mode) {  // see bh_force.hip
  const int nb = gridDim.x, b = blockIdx.x;
  if (mode == 1) return b;
  const int xcd = b & 7, p = b >> 3;
  if (mode == 2) return ((p / kXcdRun) * 8 + xcd) * kXcdRun + (p % kXcdRun);
  const int q = nb >> 3, r = nb & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + p;
}

// v_writelane_b32: value from an SGPR, lane select from M0 (one SGPR on the gfx9 constant bus);
// one wait state between the SALU write of M0 and its use as lane select.
__device__ __forceinline__ void writelane3(int& a, int& b, int& c, int ln, int va, int vb, int vc) {
  asm volatile(
      "s_mov_b32 m0, %3\n\ts_nop 0\n\t"
      "v_writelane_b32 %0, %4, m0\n\t"
      "v_writelane_b32 %1, %5, m0\n\t"
      "v_writelane_b32 %2, %6, m0"
      : "+v"(a), "+v"(b), "+v"(c)
      : "s"(ln), "s"(va), "s"(vb), "s"(vc)
      : "m0");
}

struct Stack3 {  // entry j lives in lane (j & 63) of set (j >> 6)
  int f0, l0, h0;
  int f1, l1, h1;
  int f2, l2, h2;
};

__device__ __forceinline__ void st_push(Stack3& s, int sp, int fc, u64 mask) {
  const int lo = (int)(u32)mask, hi = (int)(u32)(mask >> 32);
  if (__builtin_expect(sp < 64, 1))
    writelane3(s.f0, s.l0, s.h0, sp, fc, lo, hi);
  else if (sp < 128)
    writelane3(s.f1, s.l1, s.h1, sp - 64, fc, lo, hi);
  else
    writelane3(s.f2, s.l2, s.h2, sp - 128, fc, lo, hi);
}

__device__ __forceinline__ void st_pop(const Stack3& s, int sp, int& fc, u64& mask) {
  int lo, hi;
  if (__builtin_expect(sp < 64, 1)) {
    fc = __builtin_amdgcn_readlane(s.f0, sp);
    lo = __builtin_amdgcn_readlane(s.l0, sp);
    hi = __builtin_amdgcn_readlane(s.h0, sp);
  } else if (sp < 128) {
    fc = __builtin_amdgcn_readlane(s.f1, sp - 64);
    lo = __builtin_amdgcn_readlane(s.l1, sp - 64);
    hi = __builtin_amdgcn_readlane(s.h1, sp - 64);
  } else {
    fc = __builtin_amdgcn_readlane(s.f2, sp - 128);
    lo = __builtin_amdgcn_readlane(s.l2, sp - 128);
    hi = __builtin_amdgcn_readlane(s.h2, sp - 128);
  }
  mask = ((u64)(u32)hi << 32) | (u64)(u32)lo;
}

// (record, wave) evaluation: 15 VALU; the lanes that still need the cell opened come back in OPEN
#define BH_EVAL(R, OPEN)                                                           \
  {                                                                                \
    const float dx = (R)[0] - px, dy = (R)[1] - py, dz = (R)[2] - pz;              \
    const float d2 = fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, eps2)));               \
    const u64 accm = __builtin_amdgcn_ballot_w64(d2 > (R)[4]);                     \
    const u64 takem = mask & accm;                                                 \
    OPEN = mask & ~accm;                                                           \
    const float rinv = __builtin_amdgcn_rsqf(d2);                                  \
    const float f = ((R)[3] * rinv) * (rinv * rinv);                               \
    const float fm = __builtin_amdgcn_inverse_ballot_w64(takem) ? f : 0.0f;        \
    ax = fmaf(fm, dx, ax);                                                         \
    ay = fmaf(fm, dy, ay);                                                         \
    az = fmaf(fm, dz, az);                                                         \
  }

// an opened record: push its child block, or direct-sum an unsplit multi-body cell
#define BH_OPENED(R, OPEN)                                                         \
  if ((OPEN) != 0ull) {                                                            \
    const int cfirst = __float_as_int((R)[5]);                                     \
    const int cmeta = __float_as_int((R)[6]);                                      \
    if (cmeta >= 0) {                                                              \
      if (sp < kStackCap) {                                                        \
        st_push(st, sp, cfirst | ((cmeta - 1) << 28), (OPEN));                     \
        sp++;                                                                      \
      } else {                                                                     \
        overflow = 1;                                                              \
      }                                                                            \
    } else {                                                                       \
      const int b0 = cfirst, b1 = b0 + (cmeta & 0x7fffffff);                       \
      const bool wantl = __builtin_amdgcn_inverse_ballot_w64(OPEN);                \
      for (int bb = b0; bb < b1; bb++) {                                           \
        const float4v_t qb = *(cfloat4_t*)(bodies + (size_t)bb * 4);               \
        if (!(qb[3] > 0.0f)) continue; /* ref:203 */                               \
        const float ex = qb[0] - px, ey = qb[1] - py, ez = qb[2] - pz;             \
        const float e2 = fmaf(ez, ez, fmaf(ey, ey, fmaf(ex, ex, eps2)));           \
        const float ri = __builtin_amdgcn_rsqf(e2);                                \
        const float ff = wantl ? (Gv * qb[3]) * ri * (ri * ri) : 0.0f;             \
        ax = fmaf(ff, ex, ax);                                                     \
        ay = fmaf(ff, ey, ay);                                                     \
        az = fmaf(ff, ez, az);                                                     \
      }                                                                            \
    }                                                                              \
  }

__global__ __launch_bounds__(256) void force_fast2_kernel(const float* __restrict__ frec_g,
                                                          const float4* __restrict__ posm,
                                                          float4* __restrict__ acc, int lo, int hi, float G,
                                                          float eps2, int xcd_mode,
                                                          bh_devinfo* __restrict__ info) {
  cfloat_t* frec = (cfloat_t*)frec_g;
  cfloat_t* bodies = (cfloat_t*)posm;
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  const int chunk = block_chunk(xcd_mode);
  const int i = lo + (chunk * 4 + wib) * 64 + lane;
  const bool valid = i < hi;
  float px, py, pz;
  {
    const float4 p = valid ? posm[i] : make_float4(0.f, 0.f, 0.f, 0.f);  // ref:196
    px = p.x; py = p.y; pz = p.z;
  }
  float ax = 0.0f, ay = 0.0f, az = 0.0f;
  const float Gv = G;
  const u64 m0 = __builtin_amdgcn_ballot_w64(valid);
  if (m0 == 0) return;

  Stack3 st = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int sp = 0, overflow = 0;
  st_push(st, sp++, 0 /* first 0, count 1 */, m0);  // ref:198 stack = {root}

  while (sp > 0) {
    int fc;
    u64 mask;
    st_pop(st, --sp, fc, mask);
    const u32 first = (u32)fc & 0x0fffffffu;
    const int count = (int)(((u32)fc >> 28) & 7u) + 1;
    for (int k0 = 0; k0 < count; k0 += 4) {
      // the record pool is padded, so reading up to 3 records past the block is safe
      cfloat_t* rp = frec + (size_t)((first + (u32)k0) * 8u);  // 32-bit offset: rec_cap < 2^28
      const float8_t r0 = *(cfloat8_t*)(rp);
      const float8_t r1 = *(cfloat8_t*)(rp + 8);
      const float8_t r2 = *(cfloat8_t*)(rp + 16);
      const float8_t r3 = *(cfloat8_t*)(rp + 24);
      const int nk = count - k0;
      u64 o0 = 0, o1 = 0, o2 = 0, o3 = 0;
      if (nk >= 4) {
        BH_EVAL(r0, o0);
        BH_EVAL(r1, o1);
        BH_EVAL(r2, o2);
        BH_EVAL(r3, o3);
      } else if (nk == 3) {
        BH_EVAL(r0, o0);
        BH_EVAL(r1, o1);
        BH_EVAL(r2, o2);
      } else if (nk == 2) {
        BH_EVAL(r0, o0);
        BH_EVAL(r1, o1);
      } else {
        BH_EVAL(r0, o0);
      }
      if ((o0 | o1 | o2 | o3) != 0ull) {
        BH_OPENED(r0, o0);
        BH_OPENED(r1, o1);
        BH_OPENED(r2, o2);
        BH_OPENED(r3, o3);
      }
    }
  }
  if (valid) acc[i] = make_float4(ax, ay, az, 0.0f);  // ref:222-224
  if (overflow && lane == 0) atomicOr(&info->flags, BH_FLAG_STACK_OVERFLOW);
}

}  // namespace

hipError_t bhk_force_fast(bh_ctx* c, int lo, int hi) {
  const int blocks = (hi - lo + 255) / 256;
  const int mode = c->p.xcd_mode;
  int grid = blocks;
  if (mode == 2) grid = (blocks + 8 * kXcdRun - 1) / (8 * kXcdRun) * (8 * kXcdRun);
  force_fast2_kernel<<<grid, 256, 0, c->stream>>>((const float*)c->frec, c->posm[c->cur], c->acc, lo, hi,
                                                 c->p.G, c->p.eps2, mode, c->info);
  return hipGetLastError();
}
