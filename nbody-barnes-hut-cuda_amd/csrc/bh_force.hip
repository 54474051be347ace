// bh_force.hip — wave-cooperative Barnes-Hut tree traversal (the hot kernel), pack / unpack kernels.
//
// Reference: computeForceKernel nbody_v5_bench.cu:191-225 (one thread per body, private
// int stack[64] in scratch, AoS 76-byte nodes, bodies in random order inside a warp) and
// integrateKernel :227-249.  The reference's literal kernel only ever evaluates the root
// (SURVEY §0.1 D1-D3); this kernel implements the intended recurrence:
//   visit(entry): skip if mass <= 0 (:203); d = com - p; dist = sqrt(d.d + eps2) (:205-207);
//   a body, or a cell with s/dist < theta (:208), contributes G m d / dist^3 (:210-213);
//   any other cell is opened and its children are visited.
//
// MI355X design
//   * one wave64 owns 64 key-consecutive bodies (Hilbert order by default; one per lane) and walks ONE shared
//     traversal: the record being tested is wave-uniform, so it is fetched once per wave with scalar loads
//     (s_load_dwordx16 = one 64-byte digest pair, via the constant address space) instead of 64 times, and
//     feeds the VALU as SGPR operands;
//   * every lane applies the reference's per-body MAC exactly; a lane that accepted an ancestor is simply
//     masked off below it.  The set of lanes that still need a cell opened is the (EXEC-restricted) result of
//     the compare; if it is non-empty the (child block, lane mask) pair is pushed on the wave's stack.  Per-lane
//     results therefore equal the per-body recurrence of the CPU oracle (same interactions, fixed order);
//   * the wave's stack lives in registers ACROSS LANES: entry j is held by lane j of three VGPRs (block link, lane
//     mask), the top entry in scalar registers
//     (v_writelane / v_readlane) — no scratch, no LDS, no memory latency on push/pop.  64 entries cover every
//     tree seen in practice; a wave that needs more redoes its walk with the generic loop's 192 entries
//     (>= the 7*21+1 bound for 63-bit keys);
//   * children of a cell are one contiguous, 64-byte-aligned block of digest pairs, so an opened cell costs
//     two or four back-to-back scalar loads and up to 4 packed pair evaluations;
//   * blockIdx is remapped (block_chunk) so that the waves of an XCD share parts of the tree in its private L2;
//   * a launch is judged by what it does with the GPU's wave slots (bh_force_launch_trace).  The walk lives in 64
//     scalar registers — a window of three record pairs — so that a SIMD holds 8 waves of it (round 4: 78 / 7); K waves of one workgroup can share the walk of ONE group level by
//     level (coop_traverse_asm / force_coop_kernel: launches that do not fill the GPU), and a launch that does fill
//     it walks its last groups that way so that short jobs fill the slots the long ones leave (force_mixed_kernel).
// force_kernel<STRICT, COUNT> below is the plain per-record loop on the canonical 32-byte records (bit-exact
// reference arithmetic, V/O/P counters); force_fast_kernel / force_coop_kernel / force_mixed_kernel are the benchmarked
// ones (bhk_force picks by body count).  integrate lives in bh_tree.hip (and in the FUSE epilogue here).
#include <stdlib.h>

#include "bh_internal.h"

namespace {

// Wave-uniform reads go through the constant address space so the backend selects scalar loads
typedef __attribute__((address_space(4))) const float cfloat_t;

__device__ __forceinline__ bh_node load_rec(cfloat_t* base, int e) {
  cfloat_t* p = base + (size_t)e * 8;
  bh_node r;
  r.x = p[0]; r.y = p[1]; r.z = p[2]; r.m = p[3]; r.s = p[4];
  r.first = __float_as_int(p[5]);
  r.count = __float_as_int(p[6]);
  r.kind = __float_as_int(p[7]);
  return r;
}
__device__ __forceinline__ float4 load_body(cfloat_t* base, int b) {
  cfloat_t* p = base + (size_t)b * 4;
  return make_float4(p[0], p[1], p[2], p[3]);
}

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// blockIdx -> 256-body chunk of the Morton order.  Blocks b, b+8, b+16, ... share an XCD (and its
// private L2).  mode 0: each XCD gets one contiguous eighth of the order (best L2 locality, but a
// dense eighth finishes late); mode 1: identity (round-robin over XCDs, no locality);
// mode 2: runs of kXcdRun consecutive chunks per XCD, runs dealt round-robin (locality of a run,
// balance of round-robin; the launcher rounds the grid up to a multiple of 8*kXcdRun and surplus
// chunks fall out at the `valid` test).  Placement is a speed matter only, never correctness.
// Measured with Hilbert-ordered bodies (force ms, mode 0 / 1 / 2): 65,536: 0.224 / 0.242 / 0.227; 500,000:
// 0.705 / 0.777 / 0.734; 700,000: 0.961 / - / 0.941; 1M: 1.288 / 1.269 / 1.239; 1M theta 0.3: 3.99 / 3.83 / 3.75;
// 2M: 2.52 / - / 2.43; 8M: 9.86 / - / 9.60 — contiguous eighths win while the whole launch is resident at once,
// interleaved runs win once it is not (a dense eighth then finishes late): bh_params.xcd_mode 3 picks by that.
#ifndef BH_XCD_RUN
#define BH_XCD_RUN 64  // 16..64 measure alike at 1M bodies, 256 is 6 % slower, 2 is 2 % slower
#endif
constexpr int kXcdRun = BH_XCD_RUN;
template <int RUN = kXcdRun>
__device__ __forceinline__ int block_chunk_of(int mode, int b, int nb) {
  if (mode == 1) return b;
  const int xcd = b & 7, p = b >> 3;
  if (mode == 2) return ((p / RUN) * 8 + xcd) * RUN + (p % RUN);
  const int q = nb >> 3, r = nb & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + p;
}
__device__ __forceinline__ int block_chunk(int mode) { return block_chunk_of(mode, blockIdx.x, gridDim.x); }

struct WaveStack {  // entry j lives in lane (j & 63) of set (j >> 6)
  int f0, f1, f2;   // first child record
  int c0, c1, c2;   // child count
  int l0, l1, l2;   // lane mask low
  int h0, h1, h2;   // lane mask high
};

// v_writelane_b32 takes its value from an SGPR and its lane select from M0 (gfx9 allows one
// SGPR on the constant bus, M0 is exempt); clang has no writelane builtin, hence the asm.
// One wait state between the SALU write of M0 and its use as a lane select (s_nop 0).
__device__ __forceinline__ void writelane4(int& f, int& c, int& l, int& h, int ln, int vf, int vc,
                                           int vl, int vh) {
  asm volatile(
      "s_mov_b32 m0, %4\n\ts_nop 0\n\t"
      "v_writelane_b32 %0, %5, m0\n\t"
      "v_writelane_b32 %1, %6, m0\n\t"
      "v_writelane_b32 %2, %7, m0\n\t"
      "v_writelane_b32 %3, %8, m0"
      : "+v"(f), "+v"(c), "+v"(l), "+v"(h)
      : "s"(ln), "s"(vf), "s"(vc), "s"(vl), "s"(vh)
      : "m0");
}

template <int SETS = 3>
__device__ __forceinline__ void ws_push(WaveStack& s, int sp, int first, int count, u64 mask) {
  const int ln = sp & 63;
  const int lo = (int)(u32)mask, hi = (int)(u32)(mask >> 32);
  const int set = sp >> 6;
  if (SETS == 1 || set == 0)
    writelane4(s.f0, s.c0, s.l0, s.h0, ln, first, count, lo, hi);
  else if (set == 1)
    writelane4(s.f1, s.c1, s.l1, s.h1, ln, first, count, lo, hi);
  else
    writelane4(s.f2, s.c2, s.l2, s.h2, ln, first, count, lo, hi);
}

template <int SETS = 3>
__device__ __forceinline__ void ws_pop(const WaveStack& s, int sp, int& first, int& count, u64& mask) {
  const int ln = sp & 63;
  const int set = sp >> 6;
  int lo, hi;
  if (SETS == 1 || set == 0) {
    first = __builtin_amdgcn_readlane(s.f0, ln);
    count = __builtin_amdgcn_readlane(s.c0, ln);
    lo = __builtin_amdgcn_readlane(s.l0, ln);
    hi = __builtin_amdgcn_readlane(s.h0, ln);
  } else if (set == 1) {
    first = __builtin_amdgcn_readlane(s.f1, ln);
    count = __builtin_amdgcn_readlane(s.c1, ln);
    lo = __builtin_amdgcn_readlane(s.l1, ln);
    hi = __builtin_amdgcn_readlane(s.h1, ln);
  } else {
    first = __builtin_amdgcn_readlane(s.f2, ln);
    count = __builtin_amdgcn_readlane(s.c2, ln);
    lo = __builtin_amdgcn_readlane(s.l2, ln);
    hi = __builtin_amdgcn_readlane(s.h2, ln);
  }
  mask = ((u64)(u32)hi << 32) | (u64)(u32)lo;
}

constexpr int kStackCap = 192;
#ifndef BH_PF_MAX
#define BH_PF_MAX 200000
#endif
constexpr int kPrefetchMaxBodies = BH_PF_MAX;  // launches up to this size run the walk with its scalar-cache prefetch
constexpr int kTraversalBudget = 1 << 22;  // child blocks one wave may pop in the domain-decomposed entry

struct Lane {
  float px, py, pz;
  float ax, ay, az;
  u32 V, O, P;
};

// one (record-or-body, lane) interaction; returns true if this lane accepts
template <bool STRICT>
__device__ __forceinline__ bool interact(Lane& L, float cx, float cy, float cz, float cm, float cs,
                                         float G, float theta, float eps2, bool active) {
  const float dx = cx - L.px, dy = cy - L.py, dz = cz - L.pz;
  bool accept;
  float f;
  if (STRICT) {
    // reference source text, IEEE fp32, no contraction (library is built -ffp-contract=off)
    const float d2 = dx * dx + dy * dy + dz * dz;
    const float dist = sqrtf(d2 + eps2);
    accept = cs / dist < theta;
    f = G * cm / (dist * dist * dist);
    if (active && accept) {
      L.ax += f * dx;
      L.ay += f * dy;
      L.az += f * dz;
    }
  } else {
    const float d2 = fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, eps2)));
    const float rinv = __builtin_amdgcn_rsqf(d2);  // v_rsq_f32, 1 ulp
    accept = cs * rinv < theta;
    f = (G * cm) * (rinv * rinv * rinv);
    if (active && accept) {
      L.ax = fmaf(f, dx, L.ax);
      L.ay = fmaf(f, dy, L.ay);
      L.az = fmaf(f, dz, L.az);
    }
  }
  return accept;
}

template <bool STRICT, bool COUNT>
__global__ __launch_bounds__(256) void force_kernel(const bh_node* __restrict__ rec_g,
                                                    const float4* __restrict__ posm, float4* __restrict__ acc,
                                                    int lo, int hi, float G, float theta, float eps2,
                                                    u32* __restrict__ cV, u32* __restrict__ cO,
                                                    u32* __restrict__ cP, bh_devinfo* __restrict__ info) {
  cfloat_t* rec = (cfloat_t*)rec_g;
  cfloat_t* bodies = (cfloat_t*)posm;
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;

  const int chunk = block_chunk(0);

  const int i = lo + (chunk * 4 + wib) * 64 + lane;
  const bool valid = i < hi;
  Lane L;
  {
    const float4 p = valid ? posm[i] : make_float4(0.f, 0.f, 0.f, 0.f);  // ref:196
    L.px = p.x; L.py = p.y; L.pz = p.z;
  }
  L.ax = L.ay = L.az = 0.0f;
  L.V = L.O = L.P = 0;

  const u64 m0 = __ballot(valid);
  if (m0 == 0) return;

  WaveStack st = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int sp = 0;
  ws_push(st, sp++, 0, 1, m0);  // ref:198 stack = {root}

  while (sp > 0) {
    int first, count;
    u64 mask;
    ws_pop(st, --sp, first, count, mask);
    first = rfl(first);
    count = rfl(count);
    const bool active = (mask >> lane) & 1ull;

    for (int k0 = 0; k0 < count; k0 += 4) {
      // the record pool is padded, so reading up to 3 records past the block is safe
      const bh_node r0 = load_rec(rec, first + k0 + 0);
      const bh_node r1 = load_rec(rec, first + k0 + 1);
      const bh_node r2 = load_rec(rec, first + k0 + 2);
      const bh_node r3 = load_rec(rec, first + k0 + 3);
      const int nk = min(4, count - k0);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (k >= nk) break;
        const bh_node rr = (k == 0) ? r0 : (k == 1) ? r1 : (k == 2) ? r2 : r3;
        if (!(rr.m > 0.0f)) continue;  // ref:203 (wave-uniform)
        const bool accept = interact<STRICT>(L, rr.x, rr.y, rr.z, rr.m, rr.s, G, theta, eps2, active);
        const bool want = active && !accept;
        const u64 ob = __ballot(want);
        if (COUNT) {
          if (rr.kind == BH_KIND_BODY) {
            if (active) L.P++;
          } else {
            if (active) L.V++;
            if (want) L.O++;
          }
        }
        if (ob != 0ull) {
          if (rr.kind == BH_KIND_INTERNAL) {
            if (sp < kStackCap) {
              ws_push(st, sp++, rr.first, rr.count, ob);
            } else if (lane == 0) {
              atomicOr(&info->flags, BH_FLAG_STACK_OVERFLOW);
            }
          } else {
            // unsplit multi-body cell: its bodies interact directly (SURVEY D2/D5 intent)
            const int b1 = rr.first + rr.count;
            for (int bb = rr.first; bb < b1; bb++) {
              const float4 qb = load_body(bodies, bb);
              if (!(qb.w > 0.0f)) continue;
              (void)interact<STRICT>(L, qb.x, qb.y, qb.z, qb.w, -1.0f, G, theta, eps2, want);
              if (COUNT && want) L.P++;
            }
          }
        }
      }
    }
  }

  if (valid) {
    acc[i] = make_float4(L.ax, L.ay, L.az, 0.0f);  // ref:222-224
    if (COUNT) {
      cV[i] = L.V;
      cO[i] = L.O;
      cP[i] = L.P;
    }
  }
}

// ------------------------------------------------------------------ fast kernel
// Same traversal on the pre-digested records written by the COM stage (bh_frec, 32 B):
//   gm   = G*m (0 when m <= 0)           -> f = gm * rinv^3, no SGPR x SGPR product in the loop
//   thr2 = (s/theta)^2, -1 for a body or a mass<=0 record
//          -> accept  <=>  d2 + eps2 > thr2   (== s/dist < theta; one v_cmp, independent of the rsq)
//          a mass<=0 record is "accepted" with zero force, i.e. skipped like ref:203, branch-free
//   an unsplit multi-body cell is a cell whose children are its bodies' digests (COM stage), so the
//   loop has one kind of record and no body-array path
// Lane masks stay in SGPR pairs: accept = v_cmp into an SGPR pair, take/open = s_and/s_andn2
// with the stack entry's mask, the take mask drives one v_cndmask (inverse ballot).
// 15 VALU per (record, wave): 3 sub, 3 fma, cmp, rsq, 3 mul, cndmask, 3 fma.
struct FRec {
  float x, y, z, gm, thr2;
  int first, meta;  // child block and child count
};

typedef float float8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(4))) const float8_t cfloat8_t;

// one s_load_dwordx8 per record (a scalar field-by-field load lets the compiler defer `meta`
// into a dependent load on the open path)
__device__ __forceinline__ FRec load_frec(cfloat_t* base, int e) {
  const float8_t v = *(cfloat8_t*)(base + (size_t)e * 8);
  FRec r;
  r.x = v[0]; r.y = v[1]; r.z = v[2]; r.gm = v[3]; r.thr2 = v[4];
  r.first = __float_as_int(v[5]);
  r.meta = __float_as_int(v[6]);
  return r;
}

typedef float float16_t __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(4))) const float16_t cfloat16_t;

// two consecutive records with one s_load_dwordx16
typedef __attribute__((address_space(4))) const char cchar_t;
// records e (even) and e + 1 = one 64-byte pair of the digest pool (bh_internal.h: fields interleaved)
__device__ __forceinline__ void load_frec2(cfloat_t* base, int e, FRec& a, FRec& b) {
  // 32-bit unsigned byte offset: selects the SGPR-offset form of s_load (no 64-bit address
  // arithmetic per block); the launcher guarantees pool bytes < 4 GiB for this kernel
  const u32 off = (u32)e << 5;
  const float16_t v = *(cfloat16_t*)((cchar_t*)base + off);
  a.x = v[0]; a.y = v[2]; a.z = v[4]; a.gm = v[6]; a.thr2 = v[8];
  a.first = __float_as_int(v[10]);
  a.meta = __float_as_int(v[12]);
  b.x = v[1]; b.y = v[3]; b.z = v[5]; b.gm = v[7]; b.thr2 = v[9];
  b.first = __float_as_int(v[11]);
  b.meta = __float_as_int(v[13]);
}

#define BH_FAST_EVAL(R)                                                                  \
  {                                                                                      \
    const float dx = (R).x - px, dy = (R).y - py, dz = (R).z - pz;                       \
    const float d2 = fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, eps2)));                     \
    /* open test in the sense of the hand-scheduled walk (BH_M7): false for a NaN on either side */ \
    const u64 opnm = __builtin_amdgcn_ballot_w64((R).thr2 >= d2);                        \
    const u64 takem = mask & ~opnm, openm = mask & opnm;                                 \
    const float rinv = __builtin_amdgcn_rsqf(d2);                                        \
    const float f = ((R).gm * rinv) * (rinv * rinv);                                     \
    const float fm = __builtin_amdgcn_inverse_ballot_w64(takem) ? f : 0.0f;              \
    ax = fmaf(fm, dx, ax);                                                               \
    ay = fmaf(fm, dy, ay);                                                               \
    az = fmaf(fm, dz, az);                                                               \
    if (openm != 0ull) {                                                                 \
      {                                                                                  \
        if (sp < 64 * SETS) {                                                            \
          ws_push<SETS>(st, sp++, (R).first, (R).meta, openm);                           \
        } else {                                                                         \
          overflow = true; /* the caller redoes this wave with the large stack */        \
          sp = 0;          /* drain quickly: the loop tests sp only */                   \
        }                                                                                \
      }                                                                                  \
    }                                                                                    \
  }

// One wave's traversal from record `root` for the lanes of m0; returns false if the cross-lane stack
// (64 * SETS entries) overflowed, in which case ax..az are incomplete.
template <int SETS, bool BUDGET>
__device__ __forceinline__ bool fast_traverse(cfloat_t* frec, int root, u64 m0, float px, float py, float pz,
                                              float eps2, float& ax, float& ay, float& az, int budget,
                                              bool& limit_hit) {
  WaveStack st = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int sp = 0;
  bool overflow = false;
  ws_push<SETS>(st, sp++, root, 1, m0);  // ref:198 stack = {root}
  while (sp > 0) {
    int first, count;
    u64 mask;
    if (BUDGET && --budget < 0) {  // malformed pool (cycle / runaway counts): stop, the caller flags it
      limit_hit = true;
      return false;
    }
    ws_pop<SETS>(st, --sp, first, count, mask);
    int k0 = 0;
    do {  // count >= 1 for every stack entry
      // the record pool is padded, so reading up to 3 records past the block is safe
      FRec r0, r1, r2, r3;
      load_frec2(frec, first + k0, r0, r1);
      load_frec2(frec, first + k0 + 2, r2, r3);
      const int nk = count - k0;
      BH_FAST_EVAL(r0);
      if (nk > 1) {
        BH_FAST_EVAL(r1);
        if (nk > 2) {
          BH_FAST_EVAL(r2);
          if (nk > 3) BH_FAST_EVAL(r3);
        }
      }
      k0 += 4;
    } while (k0 < count);
  }
  return !overflow;
}

// ------------------------------------------------------------------ hand-scheduled traversal
// The same walk as fast_traverse<1>, written as one block of gfx950 assembly.  What shapes it (measured:
// tools/ubench_issue.hip, tools/ubench_forms.hip, tools/ubench_smem.hip, profiles/r02_ubench):
//   * the kernel is bound by VALU issue, and the issue cost of a VALU instruction depends on its FORM: only
//     two-source all-VGPR VOP2 runs at the full rate (2.3 cycles per wave64 instruction); anything with an
//     SGPR operand, three sources, an SGPR result (v_cmp) or a mask operand costs 3.5-4 cycles, v_rsq 6.9 —
//     and so does a PACKED fp32 instruction (v_pk_add/mul/fma_f32), which does the work of two.  The record
//     chain (x - px, ... with the record in SGPRs) is all slow forms: 58 cycles per record.  Evaluating the
//     two records of a digest pair (bh_internal.h: fields interleaved, so each field pair is an aligned SGPR
//     pair) with packed instructions costs 37 cycles per record;
//   * the entry's lane mask is loaded into EXEC once per child block, so inactive lanes need no take-mask and the
//     compare `v_cmp_ge thr2, d2` under that EXEC IS the open mask of the record; one `s_or_b64` + branch per
//     pair tests both records (scalar instructions cost ~1.25 cycles beside other waves' vector issue);
//   * lanes that open a record must not take its monopole: only the path that pushed (20 % of the pairs)
//     runs the variant of the force half with the two v_cndmask;
//   * a child block is fetched as up to 4 pairs (s_load_dwordx16 each, 64-byte aligned: blocks start at even
//     records) BEFORE anything else, two loads if it has <= 4 children, and evaluated LAST PAIR FIRST from the
//     entry a two-level branch tree on the child count selects: no per-record count test, no loop counter.  A
//     block of an odd number of children ends in a null record (gm 0, thr2 -1: it can never open);
//   * v_readlane / v_writelane cost ~5 cycles each in this loop: a stack entry is three dwords (the block LINK =
//     byte offset | child count, written into the digest by the COM stage, and the 64-bit lane mask), and the top
//     entry stays in scalar registers (BH_PUSH1);
//   * stack overflow and "child count > 8" are recorded with one s_max each and judged once, after the walk
//     (the caller then redoes the wave with the generic loop).
//   * the pair chain is software-pipelined: the force half of pair p (8 VALU) is interleaved, instruction by
//     instruction, with the MAC half of pair p-1 (10 VALU), so consecutive instructions of a wave are
//     independent (a wave's DEPENDENT VALU instructions issue only every ~4+ cycles whatever the occupancy).
// Per pair: 16 VALU + s_or_b64 + s_cbranch.  Per block: <= 4 s_load, ~13 SALU/branch, 3 v_readlane unless the
// entry was still in scalar registers.  Per push: ~9 SALU/branch, 3 v_writelane for the entry it displaces from
// the scalar registers.  v_readlane/v_writelane ignore EXEC.
// Code:  PRO(q) = MAC(q)              entry of a block of 2q+1 or 2q+2 children
//        SEG(q) = FORCE(q) || MAC(q-1)   q = 3..1;  SEG(0) = FORCE(0);  SEGm(q) = the same with the open masks
//        ARMS(q)= pushes of pair q, then SEGm(q)
// Fixed registers inside the block (everything below s74, see BH_WALK_SGPRS): s[24:71] record window of three pair
// slots, s[72:73] a mask pair, s10-s23 state; v[16:17] = (px,py), v[18:19] = (pz,-), v[20:21] = (eps2,eps2); pair sets
// (even / odd pairs) d = v[22:27] / v[30:35], rinv = v[28:29] / v[36:37]; the open masks live in the window's dead
// dwords (the `first` fields, which the walk never reads) and in s[72:73] — which pair uses which: BH_S0 .. BH_S3
// below —, written by v_cmp after the records have landed; v[38:39] d2, v[40:41] f, v[42:47] six
// partial accumulators, v48 / v50 / v51 cross-lane stack (link, mask lo, mask hi), s10 / s[22:23] its top entry;
// EXEC = the lane mask of the entry in hand, written when the entry is popped.
// Round 5: a window of THREE pairs, s[24:71] (pair slots A = s[24:39], C = s[40:55], B = s[56:71]) + one mask pair
// s[72:73]: the walk ends at s73 and a SIMD holds EIGHT waves of it (tools/ubench_occ.hip; one wave more measured -10 %
// force time between six and seven, profiles/r05_experiments/).  A block of 7-8 children loads its pairs 3, 2, 1 into
// A, B, C, evaluates pair 3 first as before, and fetches pair 0 into slot A once pair 3's registers are dead (after
// the position of its force half's mask instruction); the wait sits in front of pair 0's first use, a segment and
// a half later.  Each pair's two open masks live where nothing can overwrite them while they are alive:
//   pair 3: s[34:35] (slot A's dead `first` dwords while pair 3 sits there), s[72:73]
//   pair 2: s[66:67] (its own dead dwords), s[50:51] (pair 1's: written by pair 1's own compare only later)
//   pair 1: s[50:51], s[72:73]
//   pair 0: s[34:35] (after its records have landed), s[66:67]
#define BH_V0 "v[22:23]", "v[24:25]", "v[26:27]", "v[28:29]", "v28", "v29"
#define BH_V1 "v[30:31]", "v[32:33]", "v[34:35]", "v[36:37]", "v36", "v37"
#define BH_S0 BH_V0, "s[34:35]", "s[66:67]"  /* pair 0 */
#define BH_S2 BH_V0, "s[66:67]", "s[50:51]"  /* pair 2 */
#define BH_S1 BH_V1, "s[50:51]", "s[72:73]"  /* pair 1 */
#define BH_S3 BH_V1, "s[34:35]", "s[72:73]"  /* pair 3 */
// MAC of the pair (X, Y, Z, THR0, THR1) into set (DX, DY, DZ, R, R0, R1, MA, MB)
#define BH_M1(DX, X) "v_pk_add_f32 " DX ", " X ", v[16:17] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
#define BH_M2(DY, Y) "v_pk_add_f32 " DY ", " Y ", v[16:17] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n"
#define BH_M3(DZ, Z) "v_pk_add_f32 " DZ ", " Z ", v[18:19] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
#define BH_M4(DX) "v_pk_fma_f32 v[38:39], " DX ", " DX ", v[20:21]\n"
#define BH_M5(DY) "v_pk_fma_f32 v[38:39], " DY ", " DY ", v[38:39]\n"
#define BH_M7(MA, T0) "v_cmp_ge_f32_e64 " MA ", " T0 ", v38\n"
#define BH_M8(MB, T1) "v_cmp_ge_f32_e64 " MB ", " T1 ", v39\n"
#define BH_M9(R0, R1) "v_rsq_f32 " R0 ", v38\n v_rsq_f32 " R1 ", v39\n"
#define BH_CHK(q, MA, MB) "s_or_b64 s[12:13], " MA ", " MB "\n s_cbranch_scc1 L_push" #q "_%=\n"
#define BH_MAC_(q, DX, DY, DZ, R, R0, R1, MA, MB, X, Y, Z, T0, T1)                                       \
  BH_M1(DX, X) BH_M2(DY, Y) BH_M3(DZ, Z) BH_M4(DX) BH_M5(DY) BH_M5(DZ) BH_M7(MA, T0) BH_M8(MB, T1)        \
  BH_M9(R0, R1) BH_CHK(q, MA, MB)
// BH_S0 / BH_P0 ... are comma lists: BH_X expands them before the callee counts its arguments
#define BH_X(M, ...) M(__VA_ARGS__)
#define BH_MAC(...) BH_X(BH_MAC_, __VA_ARGS__)
#define BH_PRO(q, ...) "L_pro" #q "_%=:\n" BH_MAC(q, __VA_ARGS__) "s_branch L_seg" #q "_%=\n"
// force half on set F (pair GM), masked or not, interleaved with the MAC of pair qm (set M)
#define BH_F1(GM, R) "v_pk_mul_f32 v[40:41], " GM ", " R "\n"
#define BH_F2(R) "v_pk_mul_f32 " R ", " R ", " R "\n"
#define BH_F3(R) "v_pk_mul_f32 v[40:41], v[40:41], " R "\n"
#define BH_FA(D, A) "v_pk_fma_f32 " A ", v[40:41], " D ", " A "\n"
// (STATS: vcc_lo counts the pairs in which every active lane opens BOTH records — their force half is skipped, BH_ARMS_)
#define BH_FM(MA, MB) "v_cndmask_b32_e64 v40, v40, 0, " MA "\n v_cndmask_b32_e64 v41, v41, 0, " MB "\n"
// PRE: in front of the segment's first instruction; MID: behind the position of the mask instruction (the last read
// of pair q's scalar registers) — the late fetch of pair 0 (BH_LATE_LOAD) and the wait for it (BH_LATE_WAIT)
#define BH_SEG_(LBL, MASK, PRE, MID, qm, FDX, FDY, FDZ, FR, FR0, FR1, FMA, FMB, GM, DX, DY, DZ, R, R0, R1, MA, MB, X, Y, Z, T0, T1) \
  LBL ":\n" PRE                                                                                           \
  BH_M1(DX, X) BH_F1(GM, FR) BH_M2(DY, Y) BH_F2(FR) BH_M3(DZ, Z) BH_F3(FR) BH_M4(DX) MASK(FMA, FMB) MID   \
  BH_M5(DY) BH_FA(FDX, "v[42:43]") BH_M5(DZ) BH_FA(FDY, "v[44:45]") BH_M7(MA, T0) BH_FA(FDZ, "v[46:47]")  \
  BH_M8(MB, T1) BH_M9(R0, R1) BH_CHK(qm, MA, MB)
#define BH_NOMASK(MA, MB) ""
#define BH_LATE_LOAD "s_load_dwordx16 s[24:39], s[20:21], s18 offset:0\n"
#define BH_LATE_WAIT "s_waitcnt lgkmcnt(0)\n"
#define BH_SEG(q, qm, PRE, MID, ...) BH_X(BH_SEG_, "L_seg" #q "_%=", BH_NOMASK, PRE, MID, qm, __VA_ARGS__)
#define BH_STAT_MASKED ".if %c[stats]\n s_add_u32 s15, s15, 1\n .endif\n"
#define BH_SEGM(q, qm, PRE, MID, ...)                                                                    \
  "L_segm" #q "_%=:\n" BH_STAT_MASKED BH_X(BH_SEG_, "L_segmx" #q "_%=", BH_FM, PRE, MID, qm, __VA_ARGS__) \
  "s_branch L_seg" #qm "_%=\n"
#define BH_LAST_(LBL, MASK, FDX, FDY, FDZ, FR, FR0, FR1, FMA, FMB, GM)                                   \
  LBL ":\n" BH_F1(GM, FR) BH_F2(FR) BH_F3(FR) MASK(FMA, FMB) BH_FA(FDX, "v[42:43]") BH_FA(FDY, "v[44:45]") \
  BH_FA(FDZ, "v[46:47]")
#define BH_LAST(...) BH_X(BH_LAST_, __VA_ARGS__)
// Scalar-cache prefetch (template parameter PF of fast_traverse_asm): the walk is a chain pop -> fetch ->
// evaluate, and with few waves per SIMD the fetch latency (scalar loads served by the L2) is exposed.  A one-dword
// scalar load brings a 64-byte line into the scalar data cache without needing a register window: the first two
// lines of a child block (round 3, later: all four) are touched when the block is PUSHED (the last block pushed is the next one popped);
// s10 is a dummy target.  Measured (force ms without / with): 16,384 bodies 0.150 / 0.145, 65,536 0.219 / 0.214,
// 125,000 0.272 / 0.263, 250,000 0.442 / 0.441, 1M 1.217 / 1.226 — on for launches of <= kPrefetchMaxBodies.
// (Also touching the new stack top at every pop measured slower at every size: +0.4 % at 16k ... +5 % at 1M.)
// All four lines of the block are touched (a block of 5-8 children is fetched as four pairs; touching only the first
// two left half of the blocks waiting for the L2 at the pop): force ms, two lines / four: 16,384 bodies 0.143 /
// 0.135, 65,536 0.213 / 0.207, 125,000 0.261 / 0.250, 200,000 0.366 / 0.361; above the threshold it loses like
// the two-line form (300,000 +3 %, 1M +9 % against the walk without prefetch).  -DBH_PF2: the two-line form.
// (Touching the first line of EVERY child's block as soon as a block's records arrive: +8 ... +14 % at these sizes.)
#ifndef BH_PF2
#define BH_PF_MORE "s_load_dword s10, s[20:21], s11 offset:128\n s_load_dword s10, s[20:21], s11 offset:192\n"
#else
#define BH_PF_MORE ""
#endif
#define BH_PF_PUSH(LINK)                                                                                 \
  ".if %c[pf]\n"                                                                                          \
  "s_andn2_b32 s11, " LINK ", 63\n"                                                                        \
  "s_load_dword s10, s[20:21], s11 offset:0\n"                                                           \
  "s_load_dword s10, s[20:21], s11 offset:64\n"                                                          \
  BH_PF_MORE                                                                                              \
  ".endif\n"
// a stack entry = (link, lane mask): link = byte offset of the child block | its child count (bh_internal.h).
// The TOP of the stack is kept in scalar registers (s10 link — 0: none —, s[22:23] mask): a push first spills the
// previous top into lane s14 of v48 / v50 / v51, a pop takes the scalar copy if there is one.  The entry a block
// pushes last is the next one popped, so it never touches the lanes: v_writelane / v_readlane cost ~5 cycles each in
// this loop (6 per spilled entry), a scalar move ~1.25.  Same LIFO order, bit-identical results.  Measured (force
// ms, lanes only / scalar top): 1M 1.175 / 1.164, theta 0.3 3.567 / 3.528, 500,000 0.676 / 0.674, 65,536 0.209 /
// 0.211 — with one or two waves per SIMD the extra scalar instructions cost more than the lane instructions they
// save, so the small-launch instance (PF, <= kPrefetchMaxBodies) keeps every entry in the lanes.
#define BH_PUSH1(LINK, META, MLO, MHI)                                                                   \
  ".if %c[stats] == 0\n"                                                                                  \
  "s_max_u32 s16, s16, " META "\n"                                                                        \
  ".endif\n"                                                                                              \
  ".if %c[coop]\n"    /* cooperative walk: append to this wave's list of the next level (LDS, 16-byte entries) */ \
  "s_cmp_lt_u32 s22, s23\n"                                                                               \
  "s_cbranch_scc0 2f\n"                                                                                   \
  "v_mov_b32 v52, " LINK "\n"                                                                             \
  "v_mov_b32 v53, " MLO "\n"                                                                              \
  "v_mov_b32 v54, " MHI "\n"                                                                              \
  "v_mov_b32 v56, s22\n"                                                                                  \
  "ds_write_b96 v56, v[52:54]\n"                                                                          \
  "s_add_u32 s22, s22, 16\n"                                                                              \
  "s_branch 3f\n"                                                                                         \
  "2:\n"                /* list full: the entry goes onto this wave's own cross-lane stack and is walked  */ \
  "s_cmp_lt_u32 s14, 64\n" /* depth-first, children and all, before the wave takes its next list entry     */ \
  "s_cbranch_scc0 4f\n"                                                                                   \
  "s_mov_b32 s11, m0\n"                                                                                   \
  "s_mov_b32 m0, s14\n"                                                                                   \
  "s_add_u32 s14, s14, 1\n"                                                                               \
  "v_writelane_b32 v48, " LINK ", m0\n"                                                                   \
  "v_writelane_b32 v50, " MLO ", m0\n"                                                                    \
  "v_writelane_b32 v51, " MHI ", m0\n"                                                                    \
  "s_mov_b32 m0, s11\n"                                                                                   \
  "s_branch 3f\n"                                                                                         \
  "4:\n"                                                                                                  \
  "s_or_b32 s16, s16, 0x40000000\n" /* (64 entries there; beyond: the group is redone — a flag bit above any  */ \
                                    /* child count, which s16 otherwise holds the maximum of)                  */ \
  "3:\n"                                                                                                  \
  ".elseif %c[pf]\n"  /* small launches: straight into the lanes (see BH_POP_TAIL) */                      \
  "s_mov_b32 m0, s14\n"                                                                                   \
  "s_add_u32 s14, s14, 1\n"                                                                               \
  ".if %c[stats] == 0\n"                                                                                  \
  "s_max_u32 s15, s15, s14\n"                                                                             \
  ".else\n"                                                                                               \
  ".endif\n"                                                                                              \
  "v_writelane_b32 v48, " LINK ", m0\n"                                                                   \
  "v_writelane_b32 v50, " MLO ", m0\n"                                                                    \
  "v_writelane_b32 v51, " MHI ", m0\n"                                                                    \
  ".else\n"                                                                                               \
  "s_cmp_eq_u32 s10, 0\n"                                                                                \
  "s_cbranch_scc1 1f\n"                                                                                   \
  "s_mov_b32 m0, s14\n"                                                                                   \
  "s_add_u32 s14, s14, 1\n"                                                                               \
  ".if %c[stats] == 0\n"                                                                                  \
  "s_max_u32 s15, s15, s14\n"                                                                             \
  ".else\n"                                                                                               \
  "s_add_u32 vcc_hi, vcc_hi, 1\n"      /* entries spilled to the lanes (each is read back once) */        \
  ".endif\n"                                                                                              \
  "v_writelane_b32 v48, s10, m0\n"                                                                       \
  "v_writelane_b32 v50, s22, m0\n"                                                                        \
  "v_writelane_b32 v51, s23, m0\n"                                                                        \
  "1:\n"                                                                                                  \
  "s_mov_b32 s10, " LINK "\n"                                                                            \
  "s_mov_b32 s22, " MLO "\n"                                                                              \
  "s_mov_b32 s23, " MHI "\n"                                                                              \
  ".endif\n" BH_PF_PUSH(LINK)
// a pair with at least one opened record: push the opened one(s), then continue in the masked variant of its
// force half.  MA / MB are the pair's OPEN masks (v_cmp_nlt under EXEC = the block's lane mask).
// Both records opened: if every active lane opens both (a pair high above the group: 3.7 % of all pairs at 1M
// bodies, bh_walk_stats.no_taker_pairs), nobody takes a monopole and the pair's force half — 6 packed
// instructions and the two v_cndmask — is skipped: on to the MAC of the next pair (SKIP = its entry) or to the pop.
#define BH_ARMS_(q, SKIP, PRESKIP, MA, MALO, MAHI, MB, MBLO, MBHI, F0, M0, F1, M1)                        \
  "L_push" #q "_%=:\n"                                                                                    \
  "s_cmp_eq_u64 " MA ", 0\n"                                                                              \
  "s_cbranch_scc1 L_pushB" #q "_%=\n" BH_PUSH1(F0, M0, MALO, MAHI)                                         \
  "s_cmp_eq_u64 " MB ", 0\n"                                                                              \
  "s_cbranch_scc1 L_segm" #q "_%=\n" BH_PUSH1(F1, M1, MBLO, MBHI)                                          \
  "s_and_b64 s[12:13], " MA ", " MB "\n"                                                                  \
  "s_xor_b64 s[12:13], s[12:13], exec\n"                                                                  \
  "s_cbranch_scc1 L_segm" #q "_%=\n"                                                                      \
  ".if %c[stats]\n s_add_u32 vcc_lo, vcc_lo, 1\n .endif\n"                                                \
  PRESKIP "s_branch " SKIP "_%=\n"                                                                        \
  "L_pushB" #q "_%=:\n" BH_PUSH1(F1, M1, MBLO, MBHI)                                                      \
  "s_branch L_segm" #q "_%=\n"
// a pair slot: x at +0:1, y +2, z +4, gm +6, thr2 +8/+9, first +10/+11 (dead: open masks), meta +12/+13, link +14/+15.
// Slot A = s[24:39] holds pair 0 — or pair 3 of a block of 7-8 children until pair 0 is fetched behind it —, slot C =
// s[40:55] pair 1, slot B = s[56:71] pair 2.
#define BH_P0 "s[24:25]", "s[26:27]", "s[28:29]", "s32", "s33"
#define BH_P1 "s[40:41]", "s[42:43]", "s[44:45]", "s48", "s49"
#define BH_P2 "s[56:57]", "s[58:59]", "s[60:61]", "s64", "s65"
#define BH_P3 BH_P0
#define BH_PRO_SMALL BH_PRO(0, BH_S0, BH_P0) BH_PRO(1, BH_S1, BH_P1)
#define BH_PRO_BIG BH_PRO(2, BH_S2, BH_P2) "L_pro3_%=:\n" BH_MAC(3, BH_S3, BH_P3)
#define BH_SEG_ALL                                                                                       \
  BH_SEG(3, 2, "", BH_LATE_LOAD, BH_S3, "s[30:31]", BH_S2, BH_P2)                                         \
  BH_SEG(2, 1, "", "", BH_S2, "s[62:63]", BH_S1, BH_P1)                                                   \
  BH_SEG(1, 0, BH_LATE_WAIT, "", BH_S1, "s[46:47]", BH_S0, BH_P0)                                         \
  BH_LAST("L_seg0_%=", BH_NOMASK, BH_S0, "s[30:31]")
#define BH_SEGM_ALL                                                                                      \
  BH_SEGM(3, 2, "", BH_LATE_LOAD, BH_S3, "s[30:31]", BH_S2, BH_P2)                                        \
  BH_SEGM(2, 1, "", "", BH_S2, "s[62:63]", BH_S1, BH_P1)                                                  \
  BH_SEGM(1, 0, BH_LATE_WAIT, "", BH_S1, "s[46:47]", BH_S0, BH_P0)                                        \
  "L_segm0_%=:\n" BH_STAT_MASKED BH_LAST("L_segmx0_%=", BH_FM, BH_S0, "s[30:31]") BH_POP_TAIL
#define BH_MK0 "s[34:35]", "s34", "s35", "s[66:67]", "s66", "s67"  // the open masks of pair 0
#define BH_MK2 "s[66:67]", "s66", "s67", "s[50:51]", "s50", "s51"  // pair 2
#define BH_MK1 "s[50:51]", "s50", "s51", "s[72:73]", "s72", "s73"  // pair 1
#define BH_MK3 "s[34:35]", "s34", "s35", "s[72:73]", "s72", "s73"  // pair 3
#define BH_ARMS(...) BH_X(BH_ARMS_, __VA_ARGS__)
// (a pair nobody takes skips its force half: the skip of pair 3 issues the late fetch that half would have issued, the
// skip of pair 1 waits for it in front of pair 0's entry)
#define BH_ARMS_ALL                                                                                      \
  BH_ARMS(3, "L_pro2", BH_LATE_LOAD, BH_MK3, "s38", "s36", "s39", "s37")                                  \
  BH_ARMS(2, "L_pro1", "", BH_MK2, "s70", "s68", "s71", "s69")                                            \
  BH_ARMS(1, "L_pro0", BH_LATE_WAIT, BH_MK1, "s54", "s52", "s55", "s53")                                  \
  BH_ARMS(0, "L_tail", "", BH_MK0, "s38", "s36", "s39", "s37")
// end of a block: pop the next one (the test of L_pop folded into the loop-back branch)
#define BH_POP_TAIL                                                                                      \
  ".if %c[coop]\n s_cmp_eq_u32 s14, 0\n s_cbranch_scc1 L_centry_%=\n s_branch L_cpop_%=\n .else\n"           \
  ".if %c[pf] == 0\n s_cmp_lg_u32 s10, 0\n s_cbranch_scc1 L_take_%=\n .endif\n"                          \
  "s_sub_u32 s14, s14, 1\n s_cbranch_scc0 L_popb_%=\n s_branch L_done_%=\n .endif\n"
// Dispatch on the child count c (s19) by a two-level branch tree — no jump table, no computed jump:
//   c <= 4: two cache lines are fetched (always fetching four measured +1 %), entry PRO1 (c = 3, 4) or PRO0;
//   c >= 5: four lines, entry PRO3 (c >= 7; c > 8 also trips the "more than 8 children" redo) or PRO2.
// The second compare sits before the wait for the loads.  A block of 0 children is never
// built, and no record of such a block can be opened: the open test `thr2 >= d2` is false for a null record
// (thr2 = -1) whatever d2 is, NaN included.
#define BH_STAT_WAIT                                                                                     \
  ".if %c[stats] && %c[pf]\n s_waitcnt lgkmcnt(0)\n s_memtime s[12:13]\n s_waitcnt lgkmcnt(0)\n"            \
  "s_sub_u32 s11, s12, s22\n s_add_u32 vcc_hi, vcc_hi, s11\n .endif\n"
#define BH_DISPATCH_SMALL                                                                                \
  BH_STAT_WAIT                                                                                           \
  "s_cmp_gt_u32 s19, 2\n s_waitcnt lgkmcnt(0)\n s_cbranch_scc1 L_pro1_%=\n"

// One child block, from the stack entry in s18 (link) / EXEC (lane mask) to the jump back for the next entry:
// shared by the depth-first walk (fast_traverse_asm) and the cooperative level-by-level walk (coop_traverse_asm).
#define BH_WALK_BODY                                                                                     \
      "L_decode_%=:\n"                                                                                   \
      "s_and_b32 s19, s18, 63\n"         /* child count */                                               \
      "s_andn2_b32 s18, s18, 63\n"       /* byte offset of the block */                                  \
      "L_block_%=:\n"                                                                                    \
      ".if %c[stats] && %c[pf]\n"  /* measurement: shader clock from the fetch of a block to its arrival */ \
      "s_memtime s[22:23]\n"                                                                             \
      ".endif\n"                                                                                         \
      ".if %c[use_budget]\n"                                                                             \
      "s_sub_u32 s17, s17, 1\n"                                                                          \
      "s_cbranch_scc1 L_done_%=\n"                                                                       \
      ".endif\n"                                                                                         \
      "s_load_dwordx16 s[40:55], s[20:21], s18 offset:64\n"   /* slot C: pair 1 of every block */        \
      "s_cmp_gt_u32 s19, 6\n"                                                                            \
      "s_cbranch_scc1 L_four_%=\n"                                                                       \
      "s_load_dwordx16 s[24:39], s[20:21], s18 offset:0\n"    /* slot A: pair 0 */                       \
      ".if %c[stats]\n"                                                                                  \
      "s_add_u32 s17, s17, 1\n"   /* blocks popped */                                                    \
      "s_add_u32 s11, s19, 1\n"                                                                          \
      "s_lshr_b32 s11, s11, 1\n"                                                                         \
      "s_add_u32 s16, s16, s11\n"  /* pairs evaluated */                                                 \
      ".endif\n"                                                                                         \
      "s_cmp_gt_u32 s19, 4\n"                                                                            \
      "s_cbranch_scc1 L_big_%=\n"                                                                        \
      BH_DISPATCH_SMALL                                                                                  \
      BH_PRO_SMALL                                                                                       \
      "L_four_%=:\n"              /* 7-8 children: pair 3 into slot A, pair 0 follows it there (BH_LATE_LOAD) */ \
      "s_load_dwordx16 s[24:39], s[20:21], s18 offset:192\n"                                             \
      "s_load_dwordx16 s[56:71], s[20:21], s18 offset:128\n"                                             \
      ".if %c[stats]\n"                                                                                  \
      "s_add_u32 s17, s17, 1\n"                                                                          \
      "s_add_u32 s16, s16, 4\n"                                                                          \
      ".endif\n"                                                                                         \
      BH_STAT_WAIT                                                                                       \
      "s_waitcnt lgkmcnt(0)\n"                                                                           \
      "s_branch L_pro3_%=\n"                                                                             \
      "L_big_%=:\n"               /* 5-6 children */                                                     \
      "s_load_dwordx16 s[56:71], s[20:21], s18 offset:128\n"                                             \
      BH_STAT_WAIT                                                                                       \
      "s_waitcnt lgkmcnt(0)\n"                                                                           \
      BH_PRO_BIG                                                                                         \
      BH_SEG_ALL                                                                                         \
      "L_tail_%=:\n"                                                                                     \
      BH_POP_TAIL                                                                                        \
      BH_SEGM_ALL                                                                                        \
      BH_ARMS_ALL

// Returns false if the 64-entry cross-lane stack overflowed or a block with more than 8 children was met
// (unsplit cell of > 8 bodies); ax..az are then invalid and the caller redoes the wave.
// BUDGET: at most `budget` child blocks are popped (a malformed pool cannot hang the wave); the walk then
// stops and limit_hit is set.  `root` must be an even record index.
// STATS (measurement only, result discarded): s16 / s17 / s15 count pairs evaluated / blocks popped / pairs that
// took the masked path instead of tracking overflow, vcc_hi the stack entries spilled to the lanes, and the walk is
// stamped with s_memtime (shader clock) and s_memrealtime (100 MHz); st[0..5] = pairs, blocks, masked pairs,
// spilled entries, shader cycles, 10-ns ticks, masked pairs in which no active lane takes either record.
template <bool BUDGET, bool STATS = false, bool PF = false>
__device__ __forceinline__ bool fast_traverse_asm(const float* frec, int root, u64 m0, float px, float py,
                                                  float pz, float eps2, float& ax, float& ay, float& az,
                                                  int budget, bool& limit_hit, u32* st = nullptr) {
  int maxsp, maxc, left, notake = 0, waitcy = 0;
  u64 t0 = 0, r0 = 0;
  if (STATS) {
    t0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  asm volatile(
      "s_mov_b64 s[20:21], %[base]\n"
      "v_mov_b32 v16, %[px]\n"
      "v_mov_b32 v17, %[py]\n"
      "v_mov_b32 v18, %[pz]\n"
      "v_mov_b32 v19, 0\n"
      "v_mov_b32 v20, %[eps2]\n"
      "v_mov_b32 v21, %[eps2]\n"
      "v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n"
      "v_mov_b32 v48, 0\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n"
      "s_mov_b32 s14, 0\n"
      "s_mov_b32 s15, 0\n"
      "s_mov_b32 s16, 0\n"
      "s_mov_b32 s17, %[budget]\n"
      "s_lshl_b32 s18, %[root], 5\n"
      "s_mov_b32 s19, 1\n"
      "s_mov_b32 s10, 0\n"              // no top-of-stack entry in scalar registers yet
      ".if %c[stats]\n s_mov_b32 vcc_lo, 0\n s_mov_b32 vcc_hi, 0\n .endif\n"
      "s_mov_b64 exec, %[mask]\n"        // EXEC = the lane mask of the entry, from its pop to the next
      "s_branch L_block_%=\n"
      "L_take_%=:\n"                     // the entry pushed last is still in scalar registers
      "s_mov_b32 s18, s10\n"
      "s_mov_b64 exec, s[22:23]\n"
      "s_mov_b32 s10, 0\n"
      "s_branch L_decode_%=\n"
      "L_popb_%=:\n"
      "v_readlane_b32 s18, v48, s14\n"   // link
      "v_readlane_b32 s12, v50, s14\n"
      "v_readlane_b32 s13, v51, s14\n"
      "s_mov_b64 exec, s[12:13]\n"
      BH_WALK_BODY
      "L_done_%=:\n"
      "s_waitcnt lgkmcnt(0)\n"
      "s_mov_b64 exec, -1\n"            // (every caller enters with all 64 lanes on)
      "v_add_f32 %[ax], v42, v43\n"
      "v_add_f32 %[ay], v44, v45\n"
      "v_add_f32 %[az], v46, v47\n"
      "s_mov_b32 %[maxsp], s15\n"
      "s_mov_b32 %[maxc], s16\n"
      "s_mov_b32 %[left], s17\n"
      ".if %c[stats]\n s_mov_b32 %[ntk], vcc_lo\n s_mov_b32 %[wcy], vcc_hi\n .endif\n"
      : [ax] "=&v"(ax), [ay] "=&v"(ay), [az] "=&v"(az), [maxsp] "=s"(maxsp), [maxc] "=s"(maxc), [left] "=s"(left),
        [ntk] "=s"(notake), [wcy] "=s"(waitcy)
      : [base] "s"(frec), [root] "s"(root), [mask] "s"(m0), [px] "v"(px), [py] "v"(py), [pz] "v"(pz),
        [eps2] "s"(eps2), [budget] "s"(STATS ? 0 : budget), [use_budget] "n"(BUDGET && !STATS ? 1 : 0),
        [stats] "n"(STATS ? 1 : 0), [pf] "n"(PF ? 1 : 0), [coop] "n"(0)
      : "memory", "vcc", "scc", "m0",
        "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24",
        "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37", "s38", "s39",
        "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54",
        "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69",
        "s70", "s71", "s72", "s73", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27",
        "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42",
        "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
  if (STATS) {
    const u64 t1 = __builtin_amdgcn_s_memtime();
    const u64 r1 = __builtin_amdgcn_s_memrealtime();
    st[0] = (u32)maxc;   // s16: pairs
    st[1] = (u32)left;   // s17: blocks
    st[2] = (u32)maxsp;  // s15: masked pairs
    // vcc_hi: the stack entries that went through the lanes (every push in the small-launch instance, where the
    // register counts the fetch wait instead)
    st[3] = PF ? (u32)left - 1u : (u32)waitcy;
    st[4] = (u32)(t1 - t0);
    st[5] = (u32)(r1 - r0);
    st[6] = (u32)notake;  // vcc_lo: masked pairs without a taker
    st[7] = PF ? (u32)waitcy : 0u;  // vcc_hi (PF instance): shader cycles between the fetch of a block and its arrival
    limit_hit = false;
    return true;
  }
  limit_hit = BUDGET && left < 0;
  return maxsp <= 64 && maxc <= 8;
}

// VARIANT 0: hand-scheduled walk (fast_traverse_asm); 1: the compiler-scheduled walk (A/B, bh_params.force_variant).
// BUDGET: bound the number of child blocks a wave may pop (domain-decomposed entry: the pool holds records
// written by other ranks, and a malformed pool must end in BH_FLAG_TRAVERSAL_LIMIT, not in a hang).
// What a wave of the fused launch needs to integrate its own bodies and fold the next step's cube (FUSE below).
struct bh_fuse_args {
  float4* posm;     // the bodies, updated in place (each by the one wave that owns it)
  float4* velid;
  float dt, max_speed;
  float* rows;      // [waves][6] min / max of every wave's new positions
  float* grows;     // [waves / 32 + 1][6] the same per group of 32 waves
  u32* cnt;         // [1 + groups] counters of the two-level "last wave" hand-off (left at zero)
  float* bounds_next;
  int waves;        // waves of this launch that own bodies
  u32* trace;       // TRACE instances: [rows][4] (see trace_row); else unused
  const float4* acc_add;  // domain-decomposed step: the accelerations of the pass that ran before this one (own pieces),
                          // added first as integrate_kernel adds acc + acc2; null otherwise
  int raw;          // 1: bounds_next receives the raw min / max (this rank's share of the next global cube), not a cube
};

__device__ __forceinline__ float fuse_wave_min(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d, 64));
  return v;
}
__device__ __forceinline__ float fuse_wave_max(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
  return v;
}

// FUSE (bh_step of the default engine): the wave also integrates its bodies (ref:227-249, the arithmetic of
// integrate_kernel in bh_tree.hip, bit for bit) — nobody else reads a body's position during the launch: the walk
// reads the digests — and the launch folds the min / max of the new positions into the next step's cube the way
// integrate_kernel<true> does, two levels deep because a launch has up to 15,625 waves: the last wave of every
// group of 32 folds the group's rows, the last of those folds the groups (bh_internal.h: fence-free hand-off).
// One launch and 32 bytes per body less than force + integrate; the integrate kernel's 19 us (9 at 65,536 bodies)
// become a few hundred instructions at the end of waves that finish at different times anyway.
// FUSE epilogue of a force launch (force_fast_kernel, force_coop_kernel): the wave that holds the accelerations of
// group `w` (one body per lane, `valid` lanes) integrates them (ref:227-249, the arithmetic of integrate_kernel in
// bh_tree.hip, bit for bit) and the launch folds the min / max of the new positions into the next step's cube.
__device__ __forceinline__ void fuse_integrate_and_fold(const bh_fuse_args& fz, int w, int i, bool valid, int lane,
                                                        float px, float py, float pz, float pm, float ax, float ay,
                                                        float az) {
  float mn[3] = {1e10f, 1e10f, 1e10f};  // sentinels ref:138
  float mx[3] = {-1e10f, -1e10f, -1e10f};
  if (valid) {  // ref:227-249, source text, no contraction: v += a dt; clamp |v| to max_speed; p += v dt
    if (fz.acc_add) {  // own pass + remote pass, in integrate_kernel's order
      const float4 a0 = fz.acc_add[i];
      ax = a0.x + ax; ay = a0.y + ay; az = a0.z + az;
    }
    float4 v = fz.velid[i];
    const float DT = fz.dt, MAX_SPEED = fz.max_speed;
    float vx = v.x + ax * DT;
    float vy = v.y + ay * DT;
    float vz = v.z + az * DT;
    const float speedSq = vx * vx + vy * vy + vz * vz;
    if (speedSq > MAX_SPEED * MAX_SPEED) {
      const float scale = MAX_SPEED / sqrtf(speedSq);
      vx *= scale;
      vy *= scale;
      vz *= scale;
    }
    v.x = vx; v.y = vy; v.z = vz;
    px += vx * DT;
    py += vy * DT;
    pz += vz * DT;
    fz.posm[i] = make_float4(px, py, pz, pm);
    fz.velid[i] = v;
    mn[0] = mx[0] = px; mn[1] = mx[1] = py; mn[2] = mx[2] = pz;
  }
#pragma unroll
  for (int q = 0; q < 3; q++) {
    mn[q] = fuse_wave_min(mn[q]);
    mx[q] = fuse_wave_max(mx[q]);
  }
  const int g = w >> 5, gsize = min(32, fz.waves - (g << 5)), ngroups = (fz.waves + 31) >> 5;
  int last = 0;
  if (lane == 0) {
    float* o = fz.rows + (size_t)w * 6;
#pragma unroll
    for (int q = 0; q < 3; q++) {
      bh_publish_f32(o + q, mn[q]);
      bh_publish_f32(o + 3 + q, mx[q]);
    }
    bh_published();
    if (__hip_atomic_fetch_add(fz.cnt + 1 + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (u32)(gsize - 1)) {
      __hip_atomic_store(fz.cnt + 1 + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = 1;
    }
  }
  if (!__shfl(last, 0, 64)) return;
  {  // the group's rows (this wave's own among them)
    const float* o = fz.rows + ((size_t)(g << 5) + (size_t)min(lane, gsize - 1)) * 6;
#pragma unroll
    for (int q = 0; q < 3; q++) {
      mn[q] = fuse_wave_min(lane < gsize ? bh_collect_f32(o + q) : 1e10f);
      mx[q] = fuse_wave_max(lane < gsize ? bh_collect_f32(o + 3 + q) : -1e10f);
    }
  }
  last = 0;
  if (lane == 0) {
    float* o = fz.grows + (size_t)g * 6;
#pragma unroll
    for (int q = 0; q < 3; q++) {
      bh_publish_f32(o + q, mn[q]);
      bh_publish_f32(o + 3 + q, mx[q]);
    }
    bh_published();
    if (__hip_atomic_fetch_add(fz.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (u32)(ngroups - 1)) {
      __hip_atomic_store(fz.cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = 1;
    }
  }
  if (!__shfl(last, 0, 64)) return;
  float fn[3] = {1e10f, 1e10f, 1e10f}, fx[3] = {-1e10f, -1e10f, -1e10f};
  for (int r = lane; r < ngroups; r += 64) {
    const float* o = fz.grows + (size_t)r * 6;
#pragma unroll
    for (int q = 0; q < 3; q++) {
      fn[q] = fminf(fn[q], bh_collect_f32(o + q));
      fx[q] = fmaxf(fx[q], bh_collect_f32(o + 3 + q));
    }
  }
#pragma unroll
  for (int q = 0; q < 3; q++) {
    fn[q] = fuse_wave_min(fn[q]);
    fx[q] = fuse_wave_max(fx[q]);
  }
  if (lane == 0 && fz.raw) {  // domain-decomposed step: this rank's min / max (the X1 payload)
    float* b = fz.bounds_next;
    b[0] = fn[0]; b[1] = fn[1]; b[2] = fn[2];
    b[3] = fx[0]; b[4] = fx[1]; b[5] = fx[2];
    b[6] = b[7] = 0.0f;
  } else if (lane == 0) {  // the cube of ref:148-154 (write_cube, bh_tree.hip)
    float* b = fz.bounds_next;
    const float size = fmaxf(fx[0] - fn[0], fmaxf(fx[1] - fn[1], fx[2] - fn[2]));  // ref:148
    b[0] = fn[0]; b[1] = fn[1]; b[2] = fn[2];
    b[3] = fn[0] + size; b[4] = fn[1] + size; b[5] = fn[2] + size;  // ref:152-154: anchored at the min corner
    b[6] = fmaxf(b[3] - b[0], 1.0f);  // root edge s0, ref:55
    b[7] = 0.0f;
  }
}

// Measurement (bh_force_launch_trace, TRACE instances of the walk kernels): one row per wave — start / end of its walk on
// the chip-wide 100 MHz clock, HW_ID, XCC_ID — from which bench.py and tools/force_trace.py take the resident waves
// over time, the start of the last wave and the idle tail of every SIMD.
__device__ __forceinline__ void trace_row(u32* tr, int row, u32 t0) {
  u32* r = tr + (size_t)row * 4;
  r[0] = t0;
  r[1] = (u32)__builtin_amdgcn_s_memrealtime();
  r[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
  r[3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
}

// BH_WALK_SGPRS: a gfx950 SIMD holds floor(800 / (16-aligned SGPR count + 16)) waves (tools/ubench_occ.hip: highest
// register s72 -> 8 waves, s76 .. s88 -> 7, s92 and above -> 6).  Round 4: the walk in s10 .. s87 -> 7 waves.  Round 5:
// a three-pair record window, the walk in s10 .. s73, 80 scalar registers for the whole kernel -> 8 waves (the
// compiler's own values live in VGPR lanes across the walk: 62 spilled SGPRs, 64 VGPRs, no scratch)
#define BH_WALK_SGPRS __attribute__((amdgpu_num_sgpr(80)))
// Domain-decomposed step: the launch was enqueued behind an X4 whose fit the host had not yet looked at; the
// validation kernel has (bh_devinfo.dd_hold, bh_dd.hip): the exchange is repeated, or a rank has left — nothing to
// walk.
// `hold` is null for every launch outside the decomposed step (no load at all); the load is issued first thing and
// consumed after the wave's own position load has been issued, so that it adds no latency of its own to the wave's
// start (a launch of a few dozen microseconds per workgroup shows a dependent load at its head: +1.5 % at 65,536).
__device__ __forceinline__ int force_hold_load(const int* hold) {
  return hold ? __hip_atomic_load(hold, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
}
template <int VARIANT, bool BUDGET, bool PF = false, bool FUSE = false, bool TRACE = false>
__global__ __launch_bounds__(256) BH_WALK_SGPRS void force_fast_kernel(const float* __restrict__ frec_g,
                                                         const float4* posm,  // (FUSE: fz.posm is the same buffer)
                                                         float4* __restrict__ acc, int lo, int hi, float G,
                                                         float eps2, int xcd_mode,
                                                         bh_devinfo* __restrict__ info, int root, int budget,
                                                         int group, bh_fuse_args fz = bh_fuse_args{},
                                                         const int* __restrict__ hold = nullptr) {
  const int held = force_hold_load(hold);
  cfloat_t* frec = (cfloat_t*)frec_g;
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  const int chunk = block_chunk(xcd_mode);
  // waves never cooperate, so the workgroup may be 1, 2 or 4 waves (bh_params.force_block): a CU
  // slot is released when its LAST wave retires, and per-wave work varies by +-30 %.
  // group = bodies per wave: 64, or 32 (upper lanes idle) when the launch would not even put two waves on
  // a SIMD — half the bodies walk a smaller union of records and two waves per SIMD hide each other's latency
  const int i = lo + (chunk * (int)(blockDim.x >> 6) + wib) * group + lane;
  const bool valid = lane < group && i < hi;
  const u32 tr0 = TRACE ? (u32)__builtin_amdgcn_s_memrealtime() : 0u;
  float px, py, pz, pm;
  {
    const float4 p = valid ? posm[i] : make_float4(0.f, 0.f, 0.f, 0.f);  // ref:196
    px = p.x; py = p.y; pz = p.z; pm = p.w;
  }
  if (__builtin_amdgcn_readfirstlane(held)) return;
  float ax = 0.0f, ay = 0.0f, az = 0.0f;
  const u64 m0 = __builtin_amdgcn_ballot_w64(valid);
  if (m0 == 0) return;  // (a wave without bodies: not one of fz.waves)
  // 64 stack entries (one VGPR set, no set-select branches) cover every tree seen in practice; the
  // rare wave that needs more redoes its walk with the 192-entry stack (>= the 7*21+1 bound)
  bool ok, limit = false;
  if (VARIANT == 0)
    ok = fast_traverse_asm<BUDGET, false, PF>(frec_g, root, m0, px, py, pz, eps2, ax, ay, az, budget, limit);
  else
    ok = fast_traverse<1, BUDGET>(frec, root, m0, px, py, pz, eps2, ax, ay, az, budget, limit);
  if (!ok && !limit) {
    ax = ay = az = 0.0f;
    if (lane == 0) atomicAdd(&info->redo_waves, 1);
    if (!fast_traverse<3, BUDGET>(frec, root, m0, px, py, pz, eps2, ax, ay, az, budget, limit) && !limit &&
        lane == 0)
      atomicOr(&info->flags, BH_FLAG_STACK_OVERFLOW);
  }
  if (limit && lane == 0) atomicOr(&info->flags, BH_FLAG_TRAVERSAL_LIMIT);
  if (valid) acc[i] = make_float4(ax, ay, az, 0.0f);  // ref:222-224
  if (TRACE && lane == 0) trace_row(fz.trace, chunk * (int)(blockDim.x >> 6) + wib, tr0);
  if (!FUSE) return;

  fuse_integrate_and_fold(fz, chunk * (int)(blockDim.x >> 6) + wib, i, valid, lane, px, py, pz, pm, ax, ay, az);
}

// ------------------------------------------------------------------ cooperative walk (K waves per group)
// What bounds a launch that does not fill the GPU (round 4, tools/force_trace.py): a wave's walk is a chain pop ->
// fetch -> evaluate that runs at its own pace — ~1,500 cycles per child block, three quarters of them waiting for the
// scalar loads — whether it shares its SIMD with five other waves or with none (waves of a 1M-body launch started
// into the drain live as long as those started into a full machine); a SIMD saturates at about six waves.  So a
// launch of 1,024 groups on 1,024 SIMDs (65,536 bodies) runs at a sixth of the machine, and the last wave lifetime
// of ANY launch is spent with the machine emptying.  Here the K waves of one workgroup share the walk of ONE group
// of bodies, level by level: the entries (child block, lane mask) of a level sit in K lists in LDS — list w written
// by wave w while it evaluated the level above —, wave j evaluates entries (j - w) mod K, + K, + 2K ... of list w,
// accumulates the accepted records into its own partial accelerations and appends the opened children to its list
// of the next level; one s_barrier per level.  Which wave evaluates which block is a pure function of the tree, the
// group and K — no atomics, no arrival order —, and the K partial accelerations of a body are added in wave order
// (coop_kernel), so results are reproducible bit for bit for a given K; against the one-wave walk (K = 1 order) they
// differ by the association of the fp32 sums only (tests/test_gpu_parity.py::test_force_coop_*).
// Per block the same code as the depth-first walk (BH_WALK_BODY); a push is 4 v_mov + ds_write_b96 instead of
// 3 v_writelane, a pop ds_read_b96 + 3 v_readfirstlane instead of 3 v_readlane.
// LDS per workgroup: 2 levels x K lists x kCoopSub bytes: [count, -, -, -][kCoopCap entries of 16 bytes].
// (list size = 1 << SUBSH bytes, SUBSH 11 / 12 / 13: 127 / 255 / 511 entries per wave and level; the product uses 11 — the entries of a
// level grow like theta^-3, see force_coop_subsh; an entry a full list cannot take is walked depth-first by its wave)
constexpr int kCoopMaxK = 8;
__host__ __device__ constexpr int coop_sub(int subsh) { return 1 << subsh; }
__host__ __device__ constexpr int coop_cap(int subsh) { return (1 << subsh) / 16 - 1; }

// One wave's share of the cooperative walk.  cur / nxt: LDS byte addresses of the two level buffers (cur holds the
// root entry in list 0, every other count is zero; the caller has synchronised).  Returns false when a list
// overflowed or a block of more than 8 children was met (ax..az are then partial sums of an incomplete walk).
template <int SUBSH>
__device__ __forceinline__ bool coop_traverse_asm(const float* frec, u32 cur, u32 nxt, int K, int j, float px,
                                                  float py, float pz, float eps2, float& ax, float& ay, float& az,
                                                  bool& limit_hit) {
  int maxc;
  asm volatile(
      "s_mov_b64 s[20:21], %[base]\n"
      "v_mov_b32 v16, %[px]\n"
      "v_mov_b32 v17, %[py]\n"
      "v_mov_b32 v18, %[pz]\n"
      "v_mov_b32 v19, 0\n"
      "v_mov_b32 v20, %[eps2]\n"
      "v_mov_b32 v21, %[eps2]\n"
      "v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n"
      "v_mov_b32 v48, 0\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n"
      "v_mov_b32 v59, 0\n"                // levels walked (all lanes alike)
      "s_mov_b32 s16, 0\n"
      "s_mov_b32 s14, 0\n"                // entries on this wave's own stack (children a full list did not take)
      "s_mov_b32 s17, %[cur]\n"
      "s_mov_b32 vcc_lo, %[nxt]\n"
      "v_mbcnt_lo_u32_b32 v57, -1, 0\n"
      "v_mbcnt_hi_u32_b32 v57, -1, v57\n"
      "v_lshlrev_b32 v57, %c[subsh], v57\n" // lane l: byte offset of list l in a level buffer
      "L_level_%=:\n"
      "s_bfm_b64 exec, %[K], 0\n"         // lanes 0 .. K-1 fetch the K entry counts of this level: lane w = list w
      "v_add_u32 v56, s17, v57\n"
      "ds_read_b32 v58, v56\n"
      "s_mov_b64 exec, -1\n"
      "s_mov_b32 vcc_hi, 0\n"             // entries of this level, all lists
      "s_mov_b32 m0, 0\n"                 // list index w
      "s_mul_i32 s22, %[j], %c[sub]\n"    // this wave's list of the next level: s22 next free entry, s23 its end
      "s_add_u32 s22, s22, vcc_lo\n"
      "s_add_u32 s22, s22, 16\n"
      "s_add_u32 s23, s22, %c[capb]\n"
      "s_waitcnt lgkmcnt(0)\n"
      "L_sub_%=:\n"
      "s_cmp_ge_u32 m0, %[K]\n"
      "s_cbranch_scc1 L_lvlend_%=\n"
      "s_sub_i32 s11, %[j], m0\n"         // first entry of list w for wave j: (j - w) mod K
      "s_cmp_lt_i32 s11, 0\n"
      "s_cselect_b32 s10, %[K], 0\n"
      "s_add_u32 s11, s11, s10\n"
      "s_lshl_b32 s11, s11, 4\n"
      "s_lshl_b32 s10, m0, %c[subsh]\n"
      "s_add_u32 s10, s10, s17\n"      // header of list w
      "v_readlane_b32 s15, v58, m0\n"    // entries in list w
      "s_add_u32 vcc_hi, vcc_hi, s15\n"
      "s_lshl_b32 s15, s15, 4\n"
      "s_add_u32 s15, s15, s10\n"
      "s_add_u32 s15, s15, 16\n"        // end of list w
      "s_add_u32 s10, s10, s11\n"
      "s_add_u32 s10, s10, 16\n"        // this wave's first entry in it
      "s_add_u32 m0, m0, 1\n"
      "L_centry_%=:\n"
      "s_cmp_ge_u32 s10, s15\n"
      "s_cbranch_scc1 L_sub_%=\n"
      "v_mov_b32 v56, s10\n"
      "ds_read_b96 v[52:54], v56\n"
      "s_lshl_b32 s11, %[K], 4\n"
      "s_add_u32 s10, s10, s11\n"
      "s_waitcnt lgkmcnt(0)\n"
      "v_readfirstlane_b32 s18, v52\n"    // link
      "v_readfirstlane_b32 s12, v53\n"    // lane mask
      "v_readfirstlane_b32 s13, v54\n"
      "s_mov_b64 exec, s[12:13]\n"
      "s_branch L_decode_%=\n"
      "L_cpop_%=:\n"                     // an entry of this wave's own stack (its list was full)
      "s_add_u32 vcc_hi, vcc_hi, 0x10000\n"  // at most 65,535 of them per level (upper half of vcc_hi): a cycle in a
      "s_cbranch_scc0 5f\n"              // malformed pool (imported records) ends here, not in a hang
      "s_or_b32 s16, s16, 0x20000000\n"
      "s_mov_b32 s14, 0\n"
      "s_branch L_centry_%=\n"
      "5:\n"
      "s_sub_u32 s14, s14, 1\n"
      "v_readlane_b32 s18, v48, s14\n"
      "v_readlane_b32 s12, v50, s14\n"
      "v_readlane_b32 s13, v51, s14\n"
      "s_mov_b64 exec, s[12:13]\n"
      BH_WALK_BODY
      "L_lvlend_%=:\n"
      "s_mov_b64 exec, -1\n"
      "s_and_b32 s11, vcc_hi, 0xffff\n"   // (lower half: the level's entries, the same number on every wave)
      "s_cmp_eq_u32 s11, 0\n"            // an empty level: nobody pushed anything, every wave leaves here
      "s_cbranch_scc1 L_done_%=\n"
      "v_add_u32 v59, 1, v59\n"           // a tree has at most 21 levels below the root (+ the top tree's): a walk
      "s_nop 0\n"
      "v_readfirstlane_b32 s11, v59\n"    // that is still descending after 96 follows a cycle; every wave of the
      "s_cmp_gt_u32 s11, 96\n"            // workgroup sees the same count and leaves together
      "s_cbranch_scc0 6f\n"
      "s_or_b32 s16, s16, 0x20000000\n"
      "s_branch L_done_%=\n"
      "6:\n"
      "s_mul_i32 s11, %[j], %c[sub]\n"    // publish this wave's entry count of the next level
      "s_add_u32 s11, s11, vcc_lo\n"
      "s_sub_u32 s15, s22, s11\n"
      "s_sub_u32 s15, s15, 16\n"
      "s_lshr_b32 s15, s15, 4\n"

      "v_mov_b32 v56, s11\n"
      "v_mov_b32 v52, s15\n"
      "ds_write_b32 v56, v52\n"
      "s_waitcnt lgkmcnt(0)\n"
      "s_barrier\n"
      "s_mov_b32 s11, s17\n"              // swap the level buffers
      "s_mov_b32 s17, vcc_lo\n"
      "s_mov_b32 vcc_lo, s11\n"
      "s_branch L_level_%=\n"
      "L_done_%=:\n"
      "s_waitcnt lgkmcnt(0)\n"
      "v_add_f32 %[ax], v42, v43\n"
      "v_add_f32 %[ay], v44, v45\n"
      "v_add_f32 %[az], v46, v47\n"
      "s_mov_b32 %[maxc], s16\n"
      : [ax] "=&v"(ax), [ay] "=&v"(ay), [az] "=&v"(az), [maxc] "=s"(maxc)
      : [base] "s"(frec), [cur] "s"(cur), [nxt] "s"(nxt), [K] "s"(K), [j] "s"(j), [px] "v"(px), [py] "v"(py),
        [pz] "v"(pz), [eps2] "s"(eps2), [sub] "n"(coop_sub(SUBSH)), [subsh] "n"(SUBSH), [capb] "n"(coop_cap(SUBSH) * 16),
        [cap] "n"(coop_cap(SUBSH)),
        [use_budget] "n"(0), [stats] "n"(0), [pf] "n"(0), [coop] "n"(1)
      : "memory", "vcc", "scc", "m0",
        "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24",
        "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37", "s38", "s39",
        "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54",
        "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69",
        "s70", "s71", "s72", "s73", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27",
        "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42",
        "v43", "v44", "v45", "v46", "v47", "v48", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59");
  // s16 = the largest child count met (an unsplit cell of thousands of bodies: up to n < 2^29) | bit 30: the wave's
  // own stack overflowed | bit 29: level cap or spill budget, the pool is malformed (domain-decomposed pools)
  limit_hit = (maxc & 0x20000000) != 0;
  return (maxc & 0x1fffffff) <= 8 && (maxc & 0x40000000) == 0;
}

// One workgroup of K waves walks one group of `group` bodies (wave j of K; lds: coop_lds_bytes(K)).  Wave 0 adds
// the K partial accelerations in wave order, stores them and — FUSE — integrates the group
// (fuse_integrate_and_fold; g = the group's index in the launch).  A group whose walk overflowed a level list (or holds
// an unsplit cell of more than 8 bodies) is redone by wave 0 with the generic depth-first loop: the same decision on
// every run, so still reproducible.
__host__ __device__ constexpr size_t coop_lds_bytes(int K, int subsh) {
  return (size_t)K * (2 * coop_sub(subsh) + 64 * 3 * sizeof(float) + sizeof(u32));
}
template <bool FUSE, int SUBSH, bool TRACE = false>
__device__ __forceinline__ void coop_group(u32* coop_lds, int K, int j, int lane, const float* __restrict__ frec_g,
                                           const float4* posm, float4* __restrict__ acc, int lo, int hi, float eps2,
                                           bh_devinfo* __restrict__ info, int g, int group, const bh_fuse_args& fz,
                                           int trace_row0, int root = 0, int held = 0) {
  const int i = lo + g * group + lane;
  const bool valid = lane < group && i < hi;
  const u32 tr0 = TRACE ? (u32)__builtin_amdgcn_s_memrealtime() : 0u;
  float px, py, pz, pm;
  {
    const float4 p = valid ? posm[i] : make_float4(0.f, 0.f, 0.f, 0.f);  // ref:196
    px = p.x; py = p.y; pz = p.z; pm = p.w;
  }
  if (__builtin_amdgcn_readfirstlane(held)) return;  // (force_hold_load; the whole workgroup alike)
  const u64 m0 = __builtin_amdgcn_ballot_w64(valid);
  if (m0 == 0) return;  // the whole workgroup: every wave holds the same bodies
  // level buffers: [2][K][kCoopSub bytes]; then the partial accelerations [K][64][3] and K flags
  constexpr int kCoopSub = coop_sub(SUBSH);
  u32* const lvl = coop_lds;
  float* const part = reinterpret_cast<float*>(coop_lds + 2 * K * (kCoopSub / 4));
  u32* const bad = coop_lds + 2 * K * (kCoopSub / 4) + K * 192;
  if (threadIdx.x < (unsigned)K) {
    lvl[threadIdx.x * (kCoopSub / 4)] = threadIdx.x == 0 ? 1u : 0u;  // level 0: the root in list 0
    if (threadIdx.x == 0) {
      lvl[4] = ((u32)root << 5) | 1u;  // link of the root: its byte offset | one record
      lvl[5] = (u32)m0;
      lvl[6] = (u32)(m0 >> 32);
    }
  }
  __syncthreads();
  const u32 cur = (u32)(size_t)(__attribute__((address_space(3))) u32*)lvl;
  float ax, ay, az;
  bool lim = false;
  const bool ok = coop_traverse_asm<SUBSH>(frec_g, cur, cur + (u32)K * kCoopSub, K, j, px, py, pz, eps2, ax, ay, az,
                                           lim);
  if (TRACE && lane == 0) trace_row(fz.trace, trace_row0 + j, tr0);
  part[(j * 64 + lane) * 3 + 0] = ax;
  part[(j * 64 + lane) * 3 + 1] = ay;
  part[(j * 64 + lane) * 3 + 2] = az;
  if (lane == 0) bad[j] = lim ? 2u : (ok ? 0u : 1u);
  __syncthreads();
  if (j != 0) return;
  bool redo = false, limited = false;
  for (int w = 0; w < K; w++) {
    redo = redo || bad[w] != 0u;
    limited = limited || bad[w] == 2u;
  }
  if (limited) {  // forces of this step are invalid (BH_FLAG_TRAVERSAL_LIMIT); nothing is redone
    if (lane == 0) atomicOr(&info->flags, BH_FLAG_TRAVERSAL_LIMIT);
  } else if (!redo) {
    for (int w = 1; w < K; w++) {  // fixed order: wave 0 + wave 1 + ...
      ax += part[(w * 64 + lane) * 3 + 0];
      ay += part[(w * 64 + lane) * 3 + 1];
      az += part[(w * 64 + lane) * 3 + 2];
    }
  } else {
    bool limit = false;
    ax = ay = az = 0.0f;
    if (lane == 0) atomicAdd(&info->redo_waves, 1);
    if (!fast_traverse<3, true>((cfloat_t*)frec_g, root, m0, px, py, pz, eps2, ax, ay, az, kTraversalBudget, limit) &&
        lane == 0)
      atomicOr(&info->flags, limit ? BH_FLAG_TRAVERSAL_LIMIT : BH_FLAG_STACK_OVERFLOW);
  }
  if (valid) acc[i] = make_float4(ax, ay, az, 0.0f);  // ref:222-224
  if (FUSE) fuse_integrate_and_fold(fz, g, i, valid, lane, px, py, pz, pm, ax, ay, az);
}

// every group of the launch by K = blockDim.x / 64 waves (2..8): launches that would not fill the GPU otherwise
template <bool FUSE, int SUBSH, bool TRACE = false>
__global__ __launch_bounds__(512) BH_WALK_SGPRS void force_coop_kernel(const float* __restrict__ frec_g, const float4* posm,
                                                         float4* __restrict__ acc, int lo, int hi, float eps2,
                                                         int xcd_mode, bh_devinfo* __restrict__ info, int group,
                                                         bh_fuse_args fz, int root = 0,
                                                         const int* __restrict__ hold = nullptr) {
  extern __shared__ __attribute__((aligned(16))) u32 coop_lds[];
  const int held = force_hold_load(hold);
  const int K = (int)(blockDim.x >> 6);
  const int g = block_chunk(xcd_mode);  // one group per workgroup
  // (lo is a multiple of the group size: lo / group + g is the group's index among all groups of the context — what
  // the fused epilogue files its rows under when several launches share one fold, bhk_force_root)
  coop_group<FUSE, SUBSH, TRACE>(coop_lds, K, rfl((int)(threadIdx.x >> 6)), threadIdx.x & 63, frec_g, posm, acc, 0, hi,
                                 eps2, info, lo / group + g, group, fz, g * K, root, held);
}

// A launch that fills the GPU many times over still ends with one wave lifetime (~0.35 ms at 1M bodies) in which no
// new wave can start and the machine runs empty: the last wave started lives as long as the first, and a SIMD needs
// ~6 resident waves to be busy (tools/force_trace.py: the last wave of the 1M launch starts at 0.69 of its span).
// So the launch is ordered big jobs first: workgroups [0, nbulk) are four independent waves with a 64-body group
// each (the depth-first walk, groups [0, gb)), workgroups nbulk + t walk group gb + t with their four waves together
// (coop_group: a quarter of the lifetime) — the hardware dispatches workgroups in index order, so the short jobs fill
// the slots the long ones leave.  Which groups take which walk depends on the body count only.
constexpr int kMixedK = 4;
constexpr int kMixedRun = 16;  // chunks (workgroups of four groups) per XCD run: 64 groups, as in the one-wave launch —
                               // a run of 64 such chunks is 143 us of one XCD's time at 1M bodies, and the XCD that
                               // holds one run more than the others ends the launch that much later
#ifndef BH_TAIL_RUN
#define BH_TAIL_RUN 16
#endif
constexpr int kTailRun = BH_TAIL_RUN;  // groups of the cooperative tail per XCD run
template <bool FUSE, int SUBSH, bool BUDGET = false, bool TRACE = false>
__global__ __launch_bounds__(256) BH_WALK_SGPRS void force_mixed_kernel(const float* __restrict__ frec_g, const float4* posm,
                                                          float4* __restrict__ acc, int hi, float eps2, int xcd_mode,
                                                          bh_devinfo* __restrict__ info, int nbulk, int gb,
                                                          bh_fuse_args fz, int root = 0, int g0 = 0,
                                                          const int* __restrict__ hold = nullptr) {
  // g0: the launch covers the groups g0 .. of the context (bodies [64 g0, hi)), gb of them by one wave each
  __shared__ __attribute__((aligned(16))) u32 coop_lds[coop_lds_bytes(kMixedK, SUBSH) / 4];
  const int held = force_hold_load(hold);
  const int lane = threadIdx.x & 63;
  const int wib = rfl((int)(threadIdx.x >> 6));
  if ((int)blockIdx.x >= nbulk) {
    // (XCD placement of the tail as of the bulk: runs of kTailRun consecutive groups per XCD; nbulk is a multiple of
    // eight in mode 2, so a workgroup's XCD is also that of its index in the tail)
    const int t = block_chunk_of<kTailRun>(xcd_mode == 2 ? 2 : 1, (int)blockIdx.x - nbulk, 0);
    coop_group<FUSE, SUBSH, TRACE>(coop_lds, kMixedK, wib, lane, frec_g, posm, acc, 0, hi, eps2, info, g0 + gb + t, 64,
                                   fz, gb + t * kMixedK, root, held);
    return;
  }
  const int wl = block_chunk_of<kMixedRun>(xcd_mode, blockIdx.x, nbulk) * 4 + wib;  // the wave's group in the launch
  if (wl >= gb) return;
  const int w = g0 + wl;
  const int i = w * 64 + lane;  // (groups below gb are full: (g0 + gb) * 64 <= hi)
  const u32 tr0 = TRACE ? (u32)__builtin_amdgcn_s_memrealtime() : 0u;
  const float4 p = posm[i];  // ref:196
  if (__builtin_amdgcn_readfirstlane(held)) return;
  float px = p.x, py = p.y, pz = p.z;
  float ax = 0.0f, ay = 0.0f, az = 0.0f;
  const u64 m0 = ~0ull;
  bool limit = false;
  if (!fast_traverse_asm<BUDGET, false, false>(frec_g, root, m0, px, py, pz, eps2, ax, ay, az, kTraversalBudget, limit) &&
      !limit) {
    ax = ay = az = 0.0f;
    if (lane == 0) atomicAdd(&info->redo_waves, 1);
    if (!fast_traverse<3, BUDGET>((cfloat_t*)frec_g, root, m0, px, py, pz, eps2, ax, ay, az, kTraversalBudget, limit) &&
        !limit && lane == 0)
      atomicOr(&info->flags, BH_FLAG_STACK_OVERFLOW);
  }
  if (limit && lane == 0) atomicOr(&info->flags, BH_FLAG_TRAVERSAL_LIMIT);
  acc[i] = make_float4(ax, ay, az, 0.0f);  // ref:222-224
  if (TRACE && lane == 0) trace_row(fz.trace, wl, tr0);
  if (FUSE) fuse_integrate_and_fold(fz, w, i, true, lane, px, py, pz, p.w, ax, ay, az);
}

// measurement only (bh_force_walk_stats): the hand-scheduled walk with its event counters and clock stamps,
// one row of 8 words per wave; accelerations are not stored
constexpr int kWalkRow = BH_WALK_ROW;
template <bool PF>
__global__ __launch_bounds__(256) void force_walk_stats_kernel(const float* __restrict__ frec_g,
                                                               const float4* __restrict__ posm, int n, float eps2,
                                                               int xcd_mode, int group, u32* __restrict__ rows,
                                                               int root) {
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  const int chunk = block_chunk(xcd_mode);
  const int wave = chunk * (int)(blockDim.x >> 6) + wib;
  const int i = wave * group + lane;
  const bool valid = lane < group && i < n;
  float px, py, pz;
  {
    const float4 p = valid ? posm[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    px = p.x; py = p.y; pz = p.z;
  }
  float ax, ay, az;
  const u64 m0 = __builtin_amdgcn_ballot_w64(valid);
  if (m0 == 0) return;
  bool limit;
  u32 st[8];
  (void)fast_traverse_asm<false, true, PF>(frec_g, root, m0, px, py, pz, eps2, ax, ay, az, 0, limit, st);
  const bool odd = __float_as_uint(ax + ay + az) == 0x7fc12345u;  // never: keeps the walk's arithmetic alive
  if (lane == 0 || odd) {
    u32* r = rows + (size_t)wave * kWalkRow;
#pragma unroll
    for (int k = 0; k < 6; k++) r[k] = st[k];
    r[6] = (u32)__popcll(m0);
    r[7] = st[6];
    r[8] = st[7];
  }
}

// ------------------------------------------------------------------ pack / unpack
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ s, int n,
                                                   float4* __restrict__ posm, float4* __restrict__ velid) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t N = (size_t)n;
  posm[i] = make_float4(s[i], s[N + i], s[2 * N + i], s[6 * N + i]);
  velid[i] = make_float4(s[3 * N + i], s[4 * N + i], s[5 * N + i], __int_as_float(i));
}

// scatter back to caller order through the id carried in velid.w
__global__ __launch_bounds__(256) void unpack_state_kernel(const float4* __restrict__ posm,
                                                           const float4* __restrict__ velid, int n,
                                                           float* __restrict__ s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t N = (size_t)n;
  const float4 p = posm[i], v = velid[i];
  const size_t j = (size_t)__float_as_int(v.w);
  s[j] = p.x; s[N + j] = p.y; s[2 * N + j] = p.z;
  s[3 * N + j] = v.x; s[4 * N + j] = v.y; s[5 * N + j] = v.z;
  s[6 * N + j] = p.w;
}

__global__ __launch_bounds__(256) void unpack_acc_kernel(const float4* __restrict__ acc,
                                                         const float4* __restrict__ velid, int n,
                                                         float* __restrict__ s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t N = (size_t)n;
  const float4 a = acc[i];
  const size_t j = (size_t)__float_as_int(velid[i].w);
  s[j] = a.x; s[N + j] = a.y; s[2 * N + j] = a.z;
}

// the reference viewer's updateVisualsKernel (nbody_v5.cu:278-292) without the GL interop:
// s[0..3n) = interleaved xyz, s[3n..6n) = speed-mapped rgb, both in caller order
__global__ __launch_bounds__(256) void unpack_visual_kernel(const float4* __restrict__ posm,
                                                            const float4* __restrict__ velid, int n,
                                                            float* __restrict__ s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = posm[i], v = velid[i];
  const size_t j = (size_t)__float_as_int(v.w);
  float* vp = s + 3 * j;
  float* vc = s + 3 * (size_t)n + 3 * j;
  vp[0] = p.x; vp[1] = p.y; vp[2] = p.z;
  const float speed = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);  // v5:286
  const float t = fminf(speed / 150.0f, 1.0f);                   // v5:287
  vc[0] = 0.4f + t * 0.6f;                                       // v5:288-290
  vc[1] = 0.3f + t * 0.4f;
  vc[2] = 1.0f - t * 0.7f;
}

// what the reference binary's force kernel literally computes: the root is
// accepted for every body because `idx < n` holds for idx = 0 (ref:198,208), so the force is the
// root monopole; source-text arithmetic ref:205-213, accumulating from zero
__global__ __launch_bounds__(256) void force_literal_kernel(const bh_node* __restrict__ rec,
                                                            const float4* __restrict__ posm,
                                                            float4* __restrict__ acc, int lo, int hi, float G,
                                                            float eps2) {
  const int i = lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hi) return;
  const bh_node r = rec[0];
  const float4 p = posm[i];
  float ax = 0.0f, ay = 0.0f, az = 0.0f;
  if (r.m > 0.0f) {  // ref:203
    const float dx = r.x - p.x, dy = r.y - p.y, dz = r.z - p.z;
    const float d2 = dx * dx + dy * dy + dz * dz;
    const float dist = sqrtf(d2 + eps2);
    const float f = G * r.m / (dist * dist * dist);
    ax += f * dx;
    ay += f * dy;
    az += f * dz;
  }
  acc[i] = make_float4(ax, ay, az, 0.0f);
}

__global__ __launch_bounds__(256) void unpack_u32x3_kernel(const u32* __restrict__ a, const u32* __restrict__ b,
                                                           const u32* __restrict__ c3,
                                                           const float4* __restrict__ velid, int n,
                                                           u32* __restrict__ s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t N = (size_t)n;
  const size_t j = (size_t)__float_as_int(velid[i].w);
  s[j] = a[i]; s[N + j] = b[i]; s[2 * N + j] = c3[i];
}

}  // namespace

// bodies per wave of the fast walk: fewer than 64 (upper lanes idle) when 64 would leave the 1024 SIMDs with one
// or two waves each — smaller groups walk smaller unions of records and more waves hide each other's scalar-load
// latency.  Measured force ms with the four-line prefetch (16 / 32 / 64 bodies per wave): 8,192 bodies 0.150 /
// 0.144 / 0.146; 16,384: 0.136 / 0.146 / 0.160; 24,576: 0.170 / 0.160 / 0.176; 32,768: 0.172 / 0.166 / 0.184;
// 49,152: 0.221 / 0.201 / 0.198; 65,536: 0.268 / 0.204 / 0.203; 81,920: 0.319 / 0.248 / 0.240; 131,072: 0.486 /
// 0.317 / 0.251.  Results do not depend on the group size.
static int force_group(const bh_ctx* c, int bodies) {
  if (c->p.force_group == 16 || c->p.force_group == 32 || c->p.force_group == 64) return c->p.force_group;
  if (!c->dd && c->p.force_variant == 0 && c->p.force_coop != 1) return 64;  // cooperative walk: full groups, K waves
  return bodies <= 20 * 1024 ? 16 : (bodies <= 56 * 1024 ? 32 : 64);
}

constexpr int kWalkWaves = 8;  // resident waves per SIMD of the walk kernels (BH_WALK_SGPRS: 80 scalar registers)
constexpr int kTailWaves = 7;  // ... the cooperative tail of a mixed launch is sized with (round 4's figure; with eight
                               // the fused 1M launch measures 1.2 % slower, 500k and 2M alike: profiles/r05_experiments/)
// Groups at the end of a launch that are walked by four waves each (force_mixed_kernel; a launch of fewer groups is
// cooperative throughout): the short jobs have to refill what the long ones free while they drain — a third of the
// resident waves' worth of groups, whatever the launch size (same-box sweep at 1M bodies, force ms for 0 / 2,048 /
// 2,560 / 3,072 / 3,584 / 4,096 / 5,120 such groups: 1.166 / 1.091 / 1.091 / 1.099 / 1.097 / 1.110 / 1.114;
// 500,000 bodies: 0.692 / 0.601 / - / 0.604 / - / 0.605; 2M: 2.252 / 2.187 / - / 2.177 / - / 2.178; theta 0.3 at 1M:
// 3.533 / 3.245 / - / 3.220 / - / 3.232 — profiles/r04_drain/).
static long long force_tail_groups(const bh_ctx* c) {
  long long T = (long long)c->num_cus * 4 * kTailWaves / 3;
#ifdef BH_STUDY
  static const int env_tail = getenv("BH_FORCE_TAIL") ? atoi(getenv("BH_FORCE_TAIL")) : -1;
  if (env_tail >= 0) T = env_tail;
#endif
  return T;
}

// Waves per group (bh_params.force_coop: 0 = by context size, 1 = one wave per group: the depth-first walk, 2..8).
// Decided from the CONTEXT's body count, never from the range of one launch, so that a slab launch of a sharded step
// (bh_force_range) gives every body the bits of the full launch.  Automatic: as many waves per 64-body group as keep
// the launch within the ~6 waves per SIMD the walk saturates at.
static int force_coop(const bh_ctx* c, int group) {
  if (c->dd || c->p.force_variant != 0) return 1;
  if (c->p.force_coop >= 1 && c->p.force_coop <= kCoopMaxK) return c->p.force_coop;
  const long long groups = ((long long)c->n + group - 1) / group;
  // up to twice the tail of a mixed launch everything is walked cooperatively (same-box, force ms mixed / all by four
  // waves: 160,000 bodies 0.260 / 0.226, 200,000 0.303 / 0.280, 300,000 0.396 / 0.391; 500,000: 0.603 / 0.627)
  if (groups > 2 * force_tail_groups(c)) return 1;  // mixed launch: one wave per group first, the last groups by four
  // eight waves while eight per group still fit the GPU at once (16,384 bodies: 0.048 ms against 0.064 with four;
  // 32,768: 0.063 / 0.075), else four — one per SIMD of a CU; five to seven measure worse than either (65,536 bodies,
  // K = 4 / 5 / 6 / 7 / 8: 0.108 / 0.106 / 0.108 / 0.139 / 0.119)
  return groups * kCoopMaxK <= (long long)c->num_cus * 4 * kWalkWaves ? kCoopMaxK : 4;
}

// Level-list size of the cooperative walk (1 << subsh bytes per wave and level): 127 entries.  The entries of a level
// are the cells the group's bodies open on it — the longest list any wave writes grows like theta^-3 and hardly with
// the body count (K = 4, 1M Plummer: 76 at theta 0.5, 105 at 0.4, 170 at 0.3, 334 at 0.2; profiles/r04_coop/lists.txt);
// what a full list does not take goes onto the wave's own stack (BH_PUSH1).  Larger lists cost the mixed launch its
// occupancy (LDS is allocated per workgroup for the one-wave groups too: 35 KB instead of 19.5 -> 4 waves per SIMD).
static int force_coop_subsh(const bh_ctx* c) {
#ifdef BH_STUDY
  static const int env = getenv("BH_COOP_SUBSH") ? atoi(getenv("BH_COOP_SUBSH")) : 0;
  if (env >= 11 && env <= 13) return env;
#endif
  (void)c;
  return 11;
}

// Mixed launches (force_mixed_kernel): the bodies below this bound are walked one wave per 64-body group, the rest
// (force_tail_groups) four waves per group.
// 0: every group cooperatively (K > 1: the launch does not fill the GPU); n: none (force_coop = 1, other walks).
static int force_bulk_bodies(const bh_ctx* c, int group, int K) {
  if (K > 1) return 0;
  if (c->dd || c->p.force_variant != 0 || c->p.force_coop != 0 || group != 64) return c->n;
  const long long T = force_tail_groups(c);
  const long long G = ((long long)c->n + 63) / 64;
  const long long gb = (G - T > 0 ? G - T : 0) & ~3ll;  // whole workgroups of four one-wave groups
  return T == 0 ? c->n : (int)(gb * 64);
}

// bh_params.xcd_mode 3 (default): interleaved runs when the launch has more waves than the GPU holds at once
// (8 waves per SIMD: 54 VGPRs), contiguous eighths otherwise
static int resolve_xcd_mode(const bh_ctx* c, int bodies, int group) {
  const int mode = c->p.xcd_mode;
  if (mode != 3) return mode;
  const long long waves = ((long long)bodies + group - 1) / group;
  return waves > (long long)c->num_cus * 32 ? 2 : 0;
}

// workgroups of the cooperative tail of a mixed launch: whole XCD runs in mode 2 (the surplus groups hold no body)
static int mixed_tail_grid(int mmode, int tail) {
  return mmode == 2 ? (tail + 8 * kTailRun - 1) / (8 * kTailRun) * (8 * kTailRun) : tail;
}

// bh_force_range: a slab [lo, hi) gets the bits of the full launch only if the groups it forms are the full launch's
// groups — with several waves per group (cooperative walk) a body's bits depend on its group's composition, and the
// groups of the cooperative part are counted from `bulk`: lo must then sit on a group boundary.  (One wave per group:
// any lo — the result does not depend on the group.)
bool bhk_force_range_aligned(const bh_ctx* c, int lo) {
  if (c->p.strict_fp || c->p.literal_force || c->p.force_variant != 0) return true;
  const int group = force_group(c, c->n);
  const int K = force_coop(c, group);
  const int bulk = force_bulk_bodies(c, group, K);
  if (bulk >= c->n) return true;  // one wave per group throughout
  return lo <= bulk || (lo - bulk) % group == 0;
}

// fuse_integrate (bh_step): the launch may also integrate the bodies and fold the next step's cube into
// c->bounds_next (force_fast_kernel FUSE); *fused tells whether it did (only the hand-scheduled walk over all bodies)
hipError_t bhk_force(bh_ctx* c, int lo, int hi, bool count, bool fuse_integrate, bool* fused) {
  if (fused) *fused = false;
  if (hi <= lo) return hipSuccess;
  const int blocks = (hi - lo + 255) / 256;
  const bh_node* rec = c->rec;
  const float4* posm = c->posm[c->cur];
  const float G = c->p.G, th = c->p.theta, e2 = c->p.eps2;
  if (c->p.literal_force && !count) {
    force_literal_kernel<<<blocks, 256, 0, c->stream>>>(rec, posm, c->acc, lo, hi, G, e2);
    return hipGetLastError();
  }
  if (count) {
    if (c->p.strict_fp)
      force_kernel<true, true><<<blocks, 256, 0, c->stream>>>(rec, posm, c->acc, lo, hi, G, th, e2, c->cV, c->cO, c->cP, c->info);
    else
      force_kernel<false, true><<<blocks, 256, 0, c->stream>>>(rec, posm, c->acc, lo, hi, G, th, e2, c->cV, c->cO, c->cP, c->info);
  } else {
    if (c->p.strict_fp)
      force_kernel<true, false><<<blocks, 256, 0, c->stream>>>(rec, posm, c->acc, lo, hi, G, th, e2, nullptr, nullptr, nullptr, c->info);
    else {
      // the frec pool = tree digests + one digest slot per body (bh_internal.h)
      if ((long long)BH_FREC_POOL(c->rec_cap, c->n) >= (1ll << 27)) {
        // the fast kernel addresses records with 32-bit byte offsets: pool = 5n + 16 records of 32 B < 4 GiB,
        // i.e. up to ~26.8M bodies per context; beyond that the generic kernel does the same arithmetic on the
        // canonical records (INTEGRATION.md notes the threshold)
        force_kernel<false, false><<<blocks, 256, 0, c->stream>>>(rec, posm, c->acc, lo, hi, G, th, e2, nullptr, nullptr, nullptr, c->info);
        return hipGetLastError();
      }
      int tpb = c->p.force_block;  // 64, 128 or 256 threads per workgroup (0 = default)
      if (tpb != 64 && tpb != 128 && tpb != 256) tpb = BH_FORCE_BLOCK_DEFAULT;
      const int group = force_group(c, hi - lo);
      const int mode = resolve_xcd_mode(c, hi - lo, group);
      int g2 = ((hi - lo + group - 1) / group * 64 + tpb - 1) / tpb;
      if (mode == 2) g2 = (g2 + 8 * kXcdRun - 1) / (8 * kXcdRun) * (8 * kXcdRun);
#ifdef BH_STUDY  // design-study builds only (tools/mkvariant.sh study -DBH_STUDY): bounded walk / occupancy cap
      static const bool debug_budget = getenv("BH_FORCE_BUDGET") != nullptr;
      static const int lds_pad = getenv("BH_FORCE_LDS") ? atoi(getenv("BH_FORCE_LDS")) : 0;
#else
      constexpr bool debug_budget = false;
      constexpr int lds_pad = 0;
#endif
      const int waves = (hi - lo + group - 1) / group;
      // Which walk a group gets depends on the context's body count only (force_coop, force_bulk_bodies): bodies
      // [0, bulk) in 64-body groups by one wave each, bodies [bulk, n) by K waves per group.
      const int K = force_coop(c, group);
      const int bulk = debug_budget ? c->n : force_bulk_bodies(c, group, K);
      const int subsh = force_coop_subsh(c);
      const bool full = lo == 0 && hi == c->n;
      const bool fuse = fuse_integrate && fused && full && (c->n + group - 1) / group <= c->fuse_waves;
      const bh_fuse_args fz{c->posm[c->cur], c->velid[c->cur], c->p.dt,       c->p.max_speed, c->fuse_rows,
                            c->fuse_rows + (size_t)c->fuse_waves * 6, c->fuse_cnt, c->bounds_next, waves};
      if (bulk > 0 && bulk < c->n && full) {  // big jobs first, short jobs last: force_mixed_kernel
        const int gb = bulk / 64, tail = (c->n - bulk + 63) / 64;
        const int mmode = resolve_xcd_mode(c, bulk, 64);
        int nbulk = gb / 4;
        if (mmode == 2) nbulk = (nbulk + 8 * kMixedRun - 1) / (8 * kMixedRun) * (8 * kMixedRun);
#define BH_MIXED(F, S)                                                                                        \
  force_mixed_kernel<F, S><<<nbulk + mixed_tail_grid(mmode, tail), 256, 0, c->stream>>>((const float*)c->frec, posm, c->acc, hi, e2, mmode, \
                                                                c->info, nbulk, gb, F ? fz : bh_fuse_args{})
#ifdef BH_STUDY
        if (fuse) {
          if (subsh == 11) BH_MIXED(true, 11); else if (subsh == 12) BH_MIXED(true, 12); else BH_MIXED(true, 13);
        } else {
          if (subsh == 11) BH_MIXED(false, 11); else if (subsh == 12) BH_MIXED(false, 12); else BH_MIXED(false, 13);
        }
#else
        (void)subsh;
        if (fuse) BH_MIXED(true, 11); else BH_MIXED(false, 11);
#endif
#undef BH_MIXED
        if (fuse) *fused = true;
        return hipGetLastError();
      }
      if (hi > bulk) {  // groups of [max(lo, bulk), hi) by Kc waves each (force_coop_kernel): one workgroup per group
        const int clo = lo > bulk ? lo : bulk;
        const int Kc = K > 1 ? K : kMixedK;
        int gc = (hi - clo + group - 1) / group;
        const int cmode = resolve_xcd_mode(c, hi - clo, group);
        if (cmode == 2) gc = (gc + 8 * kXcdRun - 1) / (8 * kXcdRun) * (8 * kXcdRun);
#define BH_COOP(F, S)                                                                                         \
  force_coop_kernel<F, S><<<gc, Kc * 64, coop_lds_bytes(Kc, S), c->stream>>>(                                  \
      (const float*)c->frec, posm, c->acc, clo, hi, e2, cmode, c->info, group, F ? fz : bh_fuse_args{})
#ifdef BH_STUDY
        if (fuse && bulk == 0) {
          if (subsh == 11) BH_COOP(true, 11); else if (subsh == 12) BH_COOP(true, 12); else BH_COOP(true, 13);
        } else {
          if (subsh == 11) BH_COOP(false, 11); else if (subsh == 12) BH_COOP(false, 12); else BH_COOP(false, 13);
        }
#else
        if (fuse && bulk == 0) BH_COOP(true, 11); else BH_COOP(false, 11);
#endif
#undef BH_COOP
        if (fuse && bulk == 0) *fused = true;
        if (lo >= bulk) return hipGetLastError();
        hi = bulk;  // the rest by one wave per group, below
        g2 = ((hi - lo + group - 1) / group * 64 + tpb - 1) / tpb;
        if (mode == 2) g2 = (g2 + 8 * kXcdRun - 1) / (8 * kXcdRun) * (8 * kXcdRun);
        fuse_integrate = false;
      }
      if (fuse_integrate && fuse && !debug_budget && c->p.force_variant == 0) {
        if (hi - lo <= kPrefetchMaxBodies)
          force_fast_kernel<0, false, true, true><<<g2, tpb, 0, c->stream>>>(
              (const float*)c->frec, posm, c->acc, lo, hi, G, e2, mode, c->info, 0, 0, group, fz);
        else
          force_fast_kernel<0, false, false, true><<<g2, tpb, (size_t)lds_pad, c->stream>>>(
              (const float*)c->frec, posm, c->acc, lo, hi, G, e2, mode, c->info, 0, 0, group, fz);
        *fused = true;
        return hipGetLastError();
      }
      if (debug_budget)
        force_fast_kernel<0, true><<<g2, tpb, 0, c->stream>>>((const float*)c->frec, posm, c->acc, lo, hi, G, e2,
                                                              mode, c->info, 0, kTraversalBudget, group);
      else if (c->p.force_variant == 1)
        force_fast_kernel<1, false><<<g2, tpb, 0, c->stream>>>((const float*)c->frec, posm, c->acc, lo, hi, G, e2,
                                                               mode, c->info, 0, 0, group);
      else if (hi - lo <= kPrefetchMaxBodies)
        force_fast_kernel<0, false, true><<<g2, tpb, (size_t)lds_pad, c->stream>>>((const float*)c->frec, posm, c->acc,
                                                                                   lo, hi, G, e2, mode, c->info, 0, 0, group);
      else  // lds_pad (BH_STUDY builds, tools/occupancy_ab.sh): dynamic LDS bytes per workgroup cap the waves per CU
        force_fast_kernel<0, false><<<g2, tpb, (size_t)lds_pad, c->stream>>>((const float*)c->frec, posm, c->acc, lo,
                                                                             hi, G, e2, mode, c->info, 0, 0, group);
    }
  }
  return hipGetLastError();
}

// Measurement: the launch bh_step would make for all bodies (same kernel, same grid, same placement; not fused with
// the integrate step), with one trace row per wave (trace_row).  *rows = rows written (<= cap_rows, else nothing runs).
hipError_t bhk_force_trace(bh_ctx* c, u32* trace, int cap_rows, int* rows) {
  *rows = 0;
  if (c->p.strict_fp || c->p.literal_force || c->p.force_variant != 0 || c->dd ||
      (long long)BH_FREC_POOL(c->rec_cap, c->n) >= (1ll << 27))
    return hipSuccess;  // no trace for the other walks
  const int n = c->n, group = force_group(c, n);
  const int K = force_coop(c, group), bulk = force_bulk_bodies(c, group, K);
  const int waves = (n + group - 1) / group;
  const float4* posm = c->posm[c->cur];
  bh_fuse_args fz{};
  fz.trace = trace;
  if (bulk > 0 && bulk < n) {
    const int gb = bulk / 64, tail = (n - bulk + 63) / 64;
    if (gb + kMixedK * tail > cap_rows) return hipSuccess;
    const int mmode = resolve_xcd_mode(c, bulk, 64);
    int nbulk = gb / 4;
    if (mmode == 2) nbulk = (nbulk + 8 * kMixedRun - 1) / (8 * kMixedRun) * (8 * kMixedRun);
    force_mixed_kernel<false, 11, false, true><<<nbulk + mixed_tail_grid(mmode, tail), 256, 0, c->stream>>>(
        (const float*)c->frec, posm, c->acc, n, c->p.eps2, mmode, c->info, nbulk, gb, fz);
    *rows = gb + kMixedK * tail;
  } else if (bulk == 0) {
    if (waves * K > cap_rows) return hipSuccess;
    int gc = waves;
    const int cmode = resolve_xcd_mode(c, n, group);
    if (cmode == 2) gc = (gc + 8 * kXcdRun - 1) / (8 * kXcdRun) * (8 * kXcdRun);
    force_coop_kernel<false, 11, true><<<gc, K * 64, coop_lds_bytes(K, 11), c->stream>>>(
        (const float*)c->frec, posm, c->acc, 0, n, c->p.eps2, cmode, c->info, group, fz);
    *rows = waves * K;
  } else {
    if (waves > cap_rows) return hipSuccess;
    int tpb = c->p.force_block;
    if (tpb != 64 && tpb != 128 && tpb != 256) tpb = BH_FORCE_BLOCK_DEFAULT;
    const int mode = resolve_xcd_mode(c, n, group);
    int g2 = (waves * 64 + tpb - 1) / tpb;
    if (mode == 2) g2 = (g2 + 8 * kXcdRun - 1) / (8 * kXcdRun) * (8 * kXcdRun);
    if (n <= kPrefetchMaxBodies)
      force_fast_kernel<0, false, true, false, true><<<g2, tpb, 0, c->stream>>>(
          (const float*)c->frec, posm, c->acc, 0, n, c->p.G, c->p.eps2, mode, c->info, 0, 0, group, fz);
    else
      force_fast_kernel<0, false, false, false, true><<<g2, tpb, 0, c->stream>>>(
          (const float*)c->frec, posm, c->acc, 0, n, c->p.G, c->p.eps2, mode, c->info, 0, 0, group, fz);
    *rows = waves;
  }
  return hipGetLastError();
}

// the instruction stream of the launch bhk_force would make: the same bodies per wave (force_group) and the same
// instance of the walk — with the scalar-cache prefetch and every stack entry in the lanes up to kPrefetchMaxBodies
// bodies, with the stack top in scalar registers above (round-3 review: the small configurations were counted with
// the other instance)
int bhk_force_walk_rows(const bh_ctx* c) {
  const int group = force_group(c, c->n);
  return (c->n + group - 1) / group;
}
hipError_t bhk_force_walk_stats(bh_ctx* c, u32* rows /* [bhk_force_walk_rows][BH_WALK_ROW], device */, int root) {
  int tpb = c->p.force_block;
  if (tpb != 64 && tpb != 128 && tpb != 256) tpb = BH_FORCE_BLOCK_DEFAULT;
  const int mode = (c->p.xcd_mode == 1) ? 1 : 0;  // one row per wave in launch order: no grid padding here
  const int group = force_group(c, c->n);
  const int g2 = (bhk_force_walk_rows(c) * 64 + tpb - 1) / tpb;
  if (c->n <= kPrefetchMaxBodies)
    force_walk_stats_kernel<true><<<g2, tpb, 0, c->stream>>>((const float*)c->frec, c->posm[c->cur], c->n, c->p.eps2,
                                                             mode, group, rows, root);
  else
    force_walk_stats_kernel<false><<<g2, tpb, 0, c->stream>>>((const float*)c->frec, c->posm[c->cur], c->n, c->p.eps2,
                                                              mode, group, rows, root);
  return hipGetLastError();
}

// domain-decomposed stepping: the local bodies traverse the stitched pool (local tree + imported
// LET segments) from the top-tree root
// fuse_add: non-null = this is the LAST pass of the step (every earlier pass has finished): the launch adds those
// accelerations (null pointer value (float4*)1: there were none), integrates the bodies and folds this rank's min / max
// into c->dd_minmax (force_mixed_kernel FUSE); *fused tells whether it did (large local body counts only).
hipError_t bhk_force_root(bh_ctx* c, int lo, int hi, int root, hipStream_t stream, float4* acc, const float4* fuse_add,
                          bool fuse, bool* fused, int fold_groups) {
  // fold_groups: a step whose bodies are covered by SEVERAL fused launches (partial two-pass step: bh_dd.hip) shares
  // one min / max fold between them — the number of groups of all of them (0: this launch alone)
  if (fused) *fused = false;
  if (hi <= lo) return hipSuccess;
  int tpb = c->p.force_block;
  if (tpb != 64 && tpb != 128 && tpb != 256) tpb = BH_FORCE_BLOCK_DEFAULT;
  const int group = force_group(c, hi - lo);
  const int mode = resolve_xcd_mode(c, hi - lo, group);
  int g2 = ((hi - lo + group - 1) / group * 64 + tpb - 1) / tpb;
  if (mode == 2) g2 = (g2 + 8 * kXcdRun - 1) / (8 * kXcdRun) * (8 * kXcdRun);
  // a wave pops one child block per opened cell: no wave of a well-formed pool can pop more blocks than
  // the pool has records, so this bound never fires on valid data and always ends a walk over a cycle
  // (the decomposed step's launches are enqueued before the host knows whether the X4 behind them fits: bh_dd.hip)
  const int* hold = c->dd ? &c->info->dd_hold : nullptr;
  const int budget = kTraversalBudget;
  const bool aligned = lo % 256 == 0;  // the launch's groups are the context's groups
  const long long G = ((long long)(hi - lo) + 63) / 64;  // groups of this launch
  const long long Gall = fold_groups > 0 ? fold_groups : G;
  const bool can_fuse = fuse && fused && (fold_groups > 0 || (lo == 0 && hi == c->n)) && (int)Gall <= c->fuse_waves;
  bh_fuse_args fz{c->posm[c->cur], c->velid[c->cur], c->p.dt,       c->p.max_speed, c->fuse_rows,
                  c->fuse_rows + (size_t)c->fuse_waves * 6, c->fuse_cnt, c->dd_minmax, (int)Gall};
  fz.acc_add = fuse_add;
  fz.raw = 1;
  // big jobs first, short jobs last (force_mixed_kernel), as in bh_step: every pass of the domain-decomposed step is
  // a launch with its own drain
  if (c->p.force_variant == 0 && c->p.force_coop == 0 && aligned && group == 64 && G > 2 * force_tail_groups(c)) {
    const int gb = (int)((G - force_tail_groups(c)) & ~3ll), tail = (int)(G - gb);
    const int mmode = resolve_xcd_mode(c, gb * 64, 64);
    int nbulk = gb / 4;
    if (mmode == 2) nbulk = (nbulk + 8 * kMixedRun - 1) / (8 * kMixedRun) * (8 * kMixedRun);
    if (can_fuse) {
      force_mixed_kernel<true, 11, true><<<nbulk + mixed_tail_grid(mmode, tail), 256, 0, stream>>>(
          (const float*)c->frec, c->posm[c->cur], acc, hi, c->p.eps2, mmode, c->info, nbulk, gb, fz, root, lo / 64, hold);
      *fused = true;
      return hipGetLastError();
    }
    force_mixed_kernel<false, 11, true><<<nbulk + mixed_tail_grid(mmode, tail), 256, 0, stream>>>(
        (const float*)c->frec, c->posm[c->cur], acc, hi, c->p.eps2, mmode, c->info, nbulk, gb, bh_fuse_args{}, root,
        lo / 64, hold);
    return hipGetLastError();
  }
  // ... and a pass that would not fill the GPU is cooperative throughout (force_coop_kernel from the top-tree root:
  // the strong-scaling sizes, 1M bodies over 8 ranks = 125,000 per rank; the first part of a partial two-pass step),
  // K as in bh_step
  if (c->p.force_variant == 0 && c->p.force_coop != 1 && aligned && (c->p.force_group == 0 || c->p.force_group == 64)) {
    const int K = c->p.force_coop >= 2 && c->p.force_coop <= kCoopMaxK
                      ? c->p.force_coop
                      : (G * kCoopMaxK <= (long long)c->num_cus * 4 * kWalkWaves ? kCoopMaxK : 4);
    const int cmode = resolve_xcd_mode(c, hi - lo, 64);
    int gc = (int)G;
    if (cmode == 2) gc = (gc + 8 * kXcdRun - 1) / (8 * kXcdRun) * (8 * kXcdRun);
    if (can_fuse) {
      force_coop_kernel<true, 11><<<gc, K * 64, coop_lds_bytes(K, 11), stream>>>(
          (const float*)c->frec, c->posm[c->cur], acc, lo, hi, c->p.eps2, cmode, c->info, 64, fz, root, hold);
      *fused = true;
      return hipGetLastError();
    }
    force_coop_kernel<false, 11><<<gc, K * 64, coop_lds_bytes(K, 11), stream>>>(
        (const float*)c->frec, c->posm[c->cur], acc, lo, hi, c->p.eps2, cmode, c->info, 64, bh_fuse_args{}, root, hold);
    return hipGetLastError();
  }
  if (c->p.force_variant == 1)
    force_fast_kernel<1, true><<<g2, tpb, 0, stream>>>((const float*)c->frec, c->posm[c->cur], acc, lo, hi, c->p.G,
                                                       c->p.eps2, mode, c->info, root, budget, group, bh_fuse_args{}, hold);
  else if (hi - lo <= kPrefetchMaxBodies)
    force_fast_kernel<0, true, true><<<g2, tpb, 0, stream>>>((const float*)c->frec, c->posm[c->cur], acc, lo, hi,
                                                             c->p.G, c->p.eps2, mode, c->info, root, budget, group, bh_fuse_args{}, hold);
  else
    force_fast_kernel<0, true><<<g2, tpb, 0, stream>>>((const float*)c->frec, c->posm[c->cur], acc, lo, hi, c->p.G,
                                                       c->p.eps2, mode, c->info, root, budget, group, bh_fuse_args{}, hold);
  return hipGetLastError();
}


hipError_t bhk_pack(bh_ctx* c) {
  const int n = c->n;
  pack_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(c->stage_buf, n, c->posm[c->cur], c->velid[c->cur]);
  return hipGetLastError();
}

hipError_t bhk_unpack(bh_ctx* c, int what) {
  const int n = c->n;
  const int blocks = (n + 255) / 256;
  if (what == 0)
    unpack_state_kernel<<<blocks, 256, 0, c->stream>>>(c->posm[c->cur], c->velid[c->cur], n, c->stage_buf);
  else if (what == 1)
    unpack_acc_kernel<<<blocks, 256, 0, c->stream>>>(c->acc, c->velid[c->cur], n, c->stage_buf);
  else if (what == 2)
    unpack_u32x3_kernel<<<blocks, 256, 0, c->stream>>>(c->cV, c->cO, c->cP, c->velid[c->cur], n,
                                                       (u32*)c->stage_buf);
  else
    unpack_visual_kernel<<<blocks, 256, 0, c->stream>>>(c->posm[c->cur], c->velid[c->cur], n, c->stage_buf);
  return hipGetLastError();
}
