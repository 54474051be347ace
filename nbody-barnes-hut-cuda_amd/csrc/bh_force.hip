// bh_force.hip — wave-cooperative Barnes-Hut tree traversal (the hot kernel) + integrate.
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
//   * one wave64 owns 64 Morton-consecutive bodies (one per lane) and walks ONE shared
//     traversal: the record being tested is wave-uniform, so it is fetched once per wave
//     with scalar loads (s_load_dwordx8 into SGPRs, via the constant address space) instead
//     of 64 times, and feeds the VALU as SGPR operands;
//   * every lane applies the reference's per-body MAC exactly; a lane that accepted an
//     ancestor is simply masked off below it.  `__ballot(active && !accept)` is the set of
//     lanes that still need a cell opened; if it is non-empty the (child block, lane mask)
//     pair is pushed on the wave's stack.  Per-lane results therefore equal the per-body
//     recurrence of the CPU oracle (same interactions, deterministic order);
//   * the wave's stack lives in registers ACROSS LANES: entry j is held by lane j&63 of
//     VGPR set j>>6 (v_writelane / v_readlane) — no scratch, no LDS, no memory latency on
//     push/pop.  3 sets = 192 entries >= the 7*21+1 bound for 63-bit keys;
//   * children of a cell are one contiguous block of 32-byte records, so an opened cell costs
//     a few back-to-back scalar loads and up to 8 independent MAC evaluations (ILP);
//   * blockIdx is remapped so that each XCD walks a contiguous slab of the Morton order and
//     its private L2 keeps that slab's part of the tree.
#include "bh_internal.h"

namespace {

// Wave-uniform reads go through the constant address space so the backend selects scalar
// loads (s_load_dwordx4/x8 into SGPRs); the eight dwords of a record are merged into one.
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

__device__ __forceinline__ void ws_push(WaveStack& s, int sp, int first, int count, u64 mask) {
  const int ln = sp & 63;
  const int lo = (int)(u32)mask, hi = (int)(u32)(mask >> 32);
  const int set = sp >> 6;
  if (set == 0)
    writelane4(s.f0, s.c0, s.l0, s.h0, ln, first, count, lo, hi);
  else if (set == 1)
    writelane4(s.f1, s.c1, s.l1, s.h1, ln, first, count, lo, hi);
  else
    writelane4(s.f2, s.c2, s.l2, s.h2, ln, first, count, lo, hi);
}

__device__ __forceinline__ void ws_pop(const WaveStack& s, int sp, int& first, int& count, u64& mask) {
  const int ln = sp & 63;
  const int set = sp >> 6;
  int lo, hi;
  if (set == 0) {
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

  // XCD-aware bijective block remap (blocks b, b+8, ... share an XCD): give each XCD a
  // contiguous run of the Morton order
  const int nb = gridDim.x, b = blockIdx.x;
  const int xcd = b & 7, q = nb >> 3, r = nb & 7;
  const int chunk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);

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

// ------------------------------------------------------------------ integrate
// ref:227-249, source text, no contraction: v += a dt; clamp |v| to max_speed; p += v dt
__global__ __launch_bounds__(256) void integrate_kernel(float4* __restrict__ posm,
                                                        float4* __restrict__ velid,
                                                        const float4* __restrict__ acc, int n, float DT,
                                                        float MAX_SPEED) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = posm[i];
  float4 v = velid[i];
  const float4 a = acc[i];
  float vx = v.x + a.x * DT;
  float vy = v.y + a.y * DT;
  float vz = v.z + a.z * DT;
  const float speedSq = vx * vx + vy * vy + vz * vz;
  if (speedSq > MAX_SPEED * MAX_SPEED) {
    const float scale = MAX_SPEED / sqrtf(speedSq);
    vx *= scale;
    vy *= scale;
    vz *= scale;
  }
  v.x = vx; v.y = vy; v.z = vz;
  p.x += vx * DT;
  p.y += vy * DT;
  p.z += vz * DT;
  posm[i] = p;
  velid[i] = v;
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

hipError_t bhk_force(bh_ctx* c, int lo, int hi, bool count) {
  if (hi <= lo) return hipSuccess;
  const int blocks = (hi - lo + 255) / 256;
  const bh_node* rec = c->rec;
  const float4* posm = c->posm[c->cur];
  const float G = c->p.G, th = c->p.theta, e2 = c->p.eps2;
  if (count) {
    if (c->p.strict_fp)
      force_kernel<true, true><<<blocks, 256, 0, c->stream>>>(rec, posm, c->acc, lo, hi, G, th, e2, c->cV, c->cO, c->cP, c->info);
    else
      force_kernel<false, true><<<blocks, 256, 0, c->stream>>>(rec, posm, c->acc, lo, hi, G, th, e2, c->cV, c->cO, c->cP, c->info);
  } else {
    if (c->p.strict_fp)
      force_kernel<true, false><<<blocks, 256, 0, c->stream>>>(rec, posm, c->acc, lo, hi, G, th, e2, nullptr, nullptr, nullptr, c->info);
    else
      force_kernel<false, false><<<blocks, 256, 0, c->stream>>>(rec, posm, c->acc, lo, hi, G, th, e2, nullptr, nullptr, nullptr, c->info);
  }
  return hipGetLastError();
}

hipError_t bhk_integrate(bh_ctx* c) {
  const int n = c->n;
  integrate_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(c->posm[c->cur], c->velid[c->cur], c->acc, n,
                                                           c->p.dt, c->p.max_speed);
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
  else
    unpack_u32x3_kernel<<<blocks, 256, 0, c->stream>>>(c->cV, c->cO, c->cP, c->velid[c->cur], n,
                                                       (u32*)c->stage_buf);
  return hipGetLastError();
}
