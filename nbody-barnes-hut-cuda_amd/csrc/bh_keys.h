// bh_keys.h — Morton-key device helpers shared by the tree stages (bh_tree.hip) and the
// domain-decomposed stepping (bh_dd.hip).  Reference: nbody_v5_bench.cu:42-63.
#pragma once
#include "bh_internal.h"

__device__ __forceinline__ u32 expand_bits10(u32 v) {  // ref:42-49
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}
__device__ __forceinline__ u64 expand_bits21(u32 q) {
  u64 x = q & 0x1fffffu;
  x = (x | x << 32) & 0x001f00000000ffffull;
  x = (x | x << 16) & 0x001f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}
// inverse of expand_bits21: every third bit of k, packed
__device__ __forceinline__ u32 compact_bits21(u64 x) {
  x &= 0x1249249249249249ull;
  x = (x | x >> 2) & 0x10c30c30c30c30c3ull;
  x = (x | x >> 4) & 0x100f00f00f00f00full;
  x = (x | x >> 8) & 0x001f0000ff0000ffull;
  x = (x | x >> 16) & 0x001f00000000ffffull;
  x = (x | x >> 32) & 0x1fffffull;
  return (u32)x;
}

// key of one position inside the cube `bounds` (min xyz at [0..2], edge at [6]); B = bits per axis.
// IEEE subtract, divide, multiply in the reference's order (ref:56-58); truncating conversion.
template <int B>
__device__ __forceinline__ u64 morton_key(float px, float py, float pz, float minX, float minY, float minZ,
                                          float size) {
  constexpr float scale = (B == 10) ? 1023.0f : 2097152.0f;  // ref:56-58 (x1023) / 2^21
  constexpr u32 qmax = (1u << B) - 1u;
  u32 x = (u32)((px - minX) / size * scale);
  u32 y = (u32)((py - minY) / size * scale);
  u32 z = (u32)((pz - minZ) / size * scale);
  x = min(x, qmax);
  y = min(y, qmax);
  z = min(z, qmax);
  if (B == 10) return (u64)((expand_bits10(x) << 2) | (expand_bits10(y) << 1) | expand_bits10(z));  // ref:61
  return (expand_bits21(x) << 2) | (expand_bits21(y) << 1) | expand_bits21(z);
}

// Hilbert order (bh_params.key_curve = 1): the same 3 x 21-bit cell coordinates, numbered along the Hilbert curve
// instead of the Z curve.  Skilling's "AxesToTranspose" (J. Skilling, Programming the Hilbert curve, AIP Conf.
// Proc. 707, 2004): undo the excess rotations level by level, then Gray-encode; the result is the "transposed"
// index whose bit-interleave is the Hilbert index.  Every 3L-bit key prefix still names one octree cell of level L
// (the cells are the same cubes, their numbering inside the parent changes), so sort, tree build, COM and the
// walk are untouched; consecutive bodies are always spatial neighbours (no Z-curve jumps), which makes the
// 64-body groups of the force walk more compact: -5 % force time at 1M bodies.
// Branch-free: written with `if (X1 & Q) ... else ...` (Skilling's text) the compiler built two divergent regions
// per level (EXEC save / restore, 40 v_cmp + 22 VCC-form v_cndmask per key — the slowest instruction form there
// is on gfx950); as masks every step is a couple of three-input boolean operations (v_bitop3_b32).
// bit_mask: bit Q of x spread over the word (0 or ~0) — inline asm so that the optimiser cannot see a select in it:
// given `0 - ((x >> q) & 1)` it rebuilds v_cmp + v_cndmask (79 VCC-form selects per key).
template <int Q>
__device__ __forceinline__ u32 bit_mask(u32 x) {
  u32 m;
  asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(x), "n"(Q));
  return m;
}
template <int Q>
__device__ __forceinline__ void hilbert_level(u32& X0, u32& X1, u32& X2) {
  constexpr u32 P = (1u << Q) - 1u;
  const u32 m0 = bit_mask<Q>(X0);  // all ones where the level's bit is set
  X0 ^= P & m0;
  const u32 m1 = bit_mask<Q>(X1);
  const u32 t1 = (X0 ^ X1) & P & ~m1;  // bit clear: exchange the low bits of X0 and X1; set: invert X0's
  X0 ^= (P & m1) ^ t1;
  X1 ^= t1;
  const u32 m2 = bit_mask<Q>(X2);
  const u32 t2 = (X0 ^ X2) & P & ~m2;
  X0 ^= (P & m2) ^ t2;
  X2 ^= t2;
  if constexpr (Q > 1) hilbert_level<Q - 1>(X0, X1, X2);
}
__device__ __forceinline__ void hilbert_axes_to_transpose(u32& X0, u32& X1, u32& X2) {
  hilbert_level<20>(X0, X1, X2);  // levels Q = 2^20 .. 2
  X1 ^= X0;
  X2 ^= X1;
  // Gray encode: bit b of t = parity of the bits of X2 above b (the loop `if (X2 & Q) t ^= Q - 1` over Q = 2^20 .. 2)
  u32 t = X2 >> 1;
  t ^= t >> 1;
  t ^= t >> 2;
  t ^= t >> 4;
  t ^= t >> 8;
  t ^= t >> 16;
  X0 ^= t; X1 ^= t; X2 ^= t;
}
// inverse ("TransposeToAxes"): transposed index -> cell coordinates
__device__ __forceinline__ void hilbert_transpose_to_axes(u32& X0, u32& X1, u32& X2) {
  u32 t = X2 >> 1;
  X2 ^= X1;
  X1 ^= X0;
  X0 ^= t;
  for (u32 Q = 2u; Q != (1u << 21); Q <<= 1) {
    const u32 P = Q - 1u;
    if (X2 & Q) X0 ^= P; else { const u32 u = (X0 ^ X2) & P; X0 ^= u; X2 ^= u; }
    if (X1 & Q) X0 ^= P; else { const u32 u = (X0 ^ X1) & P; X0 ^= u; X1 ^= u; }
    if (X0 & Q) X0 ^= P;
  }
}

// key of one position: CURVE 0 = Morton (the reference's order, ref:42-63), 1 = Hilbert (21 bits per axis only)
template <int B>
__device__ __forceinline__ u64 body_key(int curve, float px, float py, float pz, float minX, float minY,
                                        float minZ, float size) {
  if (B == 10 || curve == 0) return morton_key<B>(px, py, pz, minX, minY, minZ, size);
  constexpr u32 qmax = (1u << 21) - 1u;
  u32 x = min((u32)((px - minX) / size * 2097152.0f), qmax);  // the same quantisation as morton_key
  u32 y = min((u32)((py - minY) / size * 2097152.0f), qmax);
  u32 z = min((u32)((pz - minZ) / size * 2097152.0f), qmax);
  hilbert_axes_to_transpose(x, y, z);
  return (expand_bits21(x) << 2) | (expand_bits21(y) << 1) | expand_bits21(z);
}

// ---- the same Hilbert keys from a state table (round 4) ----
// The bit algorithm above costs ~500 vector instructions per key (20 levels of masks and exchanges, three 21-bit
// interleaves) and the kernels that key every body of a step are VALU-bound on it (keys_split_kernel 23 us at 1M
// bodies, dd_classify_kernel 26).  The curve is a finite-state machine: 24 orientations; kHilbertTab (generated from
// the bit algorithm by tools/hilbert_fsm.py, which also checks it on 2M coordinates at full depth) maps (state, two
// bits of x, y, z) to six key bits and the next state.  Ten lookups + one for the last level, ~10 instructions each;
// the table (3 KB) is staged in LDS by the block (hilbert_stage).  Bit-identical to hilbert_axes_to_transpose + the
// interleave: the golden-fixture and oracle tests compare keys bit for bit.
#include "bh_hilbert_tab.h"
constexpr int kHilbertTabWords = kHilbertStates * 64 / 2;  // 32-bit words
// stage the table: every thread of the block calls this, then __syncthreads()
__device__ __forceinline__ void hilbert_stage(u32* lds_tab /* [kHilbertTabWords] */) {
  const u32* g = reinterpret_cast<const u32*>(kHilbertTab);
  for (int i = threadIdx.x; i < kHilbertTabWords; i += blockDim.x) lds_tab[i] = g[i];
}
template <int SH>
__device__ __forceinline__ u32 hilbert_step(const unsigned char* tab, u32& st, u32 x, u32 y, u32 z) {
  // byte offset of the entry: state (already << 7) + (xx << 5 | yy << 3 | zz << 1)
  const u32 off = st + (((x >> SH) & 3u) << 5 | ((y >> SH) & 3u) << 3 | ((z >> SH) & 3u) << 1);
  const u32 e = *reinterpret_cast<const unsigned short*>(tab + off);
  st = e & 0xff80u;
  return e & 63u;
}
__device__ __forceinline__ u64 hilbert_key_fsm(const u32* lds_tab, u32 x, u32 y, u32 z) {
  const unsigned char* tab = reinterpret_cast<const unsigned char*>(lds_tab);
  u32 st = 0u;
  u32 hi = hilbert_step<19>(tab, st, x, y, z);
  hi = hi << 6 | hilbert_step<17>(tab, st, x, y, z);
  hi = hi << 6 | hilbert_step<15>(tab, st, x, y, z);
  hi = hi << 6 | hilbert_step<13>(tab, st, x, y, z);
  hi = hi << 6 | hilbert_step<11>(tab, st, x, y, z);
  u32 lo = hilbert_step<9>(tab, st, x, y, z);
  lo = lo << 6 | hilbert_step<7>(tab, st, x, y, z);
  lo = lo << 6 | hilbert_step<5>(tab, st, x, y, z);
  lo = lo << 6 | hilbert_step<3>(tab, st, x, y, z);
  lo = lo << 6 | hilbert_step<1>(tab, st, x, y, z);
  // the last level alone: its bit as the high bit of a pair, the upper three of the six key bits
  const u32 off = st + ((x & 1u) << 6 | (y & 1u) << 4 | (z & 1u) << 2);
  const u32 d = (*reinterpret_cast<const unsigned short*>(tab + off) >> 3) & 7u;
  return (u64)hi << 33 | (u64)lo << 3 | (u64)d;
}
// body_key with the table (21 bits per axis): the same quantisation, Morton unchanged
__device__ __forceinline__ u64 body_key21_fsm(const u32* lds_tab, int curve, float px, float py, float pz, float minX,
                                              float minY, float minZ, float size) {
  if (curve == 0) return morton_key<21>(px, py, pz, minX, minY, minZ, size);
  constexpr u32 qmax = (1u << 21) - 1u;
  const u32 x = min((u32)((px - minX) / size * 2097152.0f), qmax);
  const u32 y = min((u32)((py - minY) / size * 2097152.0f), qmax);
  const u32 z = min((u32)((pz - minZ) / size * 2097152.0f), qmax);
  return hilbert_key_fsm(lds_tab, x, y, z);
}

// leading octal digits shared by two keys of B digits
__device__ __forceinline__ int common_digits(u64 a, u64 b, int B) {
  const u64 x = a ^ b;
  if (x == 0) return B;
  const int hb = 63 - __clzll((long long)x);
  return B - 1 - hb / 3;
}

__device__ __forceinline__ u64 key_prefix(u64 k, int sh) { return (sh >= 64) ? 0ull : (k >> sh); }
