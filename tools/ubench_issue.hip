// ubench_issue.hip — design-study microbenchmark (not product code): the gfx950 issue model the
// force kernel's roofline is priced against.  Every case is a loop of inline-asm instructions with no
// memory access, stamped in-kernel with s_memtime (shader clock) and s_memrealtime (100 MHz constant
// clock), so each line reports cycles per instruction per SIMD *and* the clock the chip actually ran
// at — at 1, 2, 4 and 8 waves per SIMD on every SIMD of the chip.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_issue.hip -o tools/bin/ubench_issue && tools/bin/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
typedef unsigned long long u64;

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

enum {
  C_FMA_INDEP = 0,   // 16 independent v_fma_f32 per iteration
  C_FMA_DEP,         // 16 v_fma_f32 in one dependency chain
  C_RSQ_INDEP,       // 16 independent v_rsq_f32
  C_MIX_14_1,        // 14 independent v_fma + 1 v_rsq, repeated (the record's VALU mix, no dependencies)
  C_SALU_INDEP,      // 16 independent s_add_u32
  C_VALU_SALU,       // 16 x (v_fma ; s_add) alternating, independent
  C_CMP_SALU,        // v_cmp -> s_andn2(vcc) -> s_cmp -> s_cbranch (not taken) + 1 v_fma, x4
  C_RECORD,          // the force kernel's exact per-record chain, 4 records, scalar part included
  C_RECORD_VALU,     // the same 15 VALU only (no s_andn2/s_cmp/s_cbranch/s_and)
  C_RECORD_ILP2,     // two records' chains interleaved instruction by instruction (VALU only)
  C_PKFMA_INDEP,     // 16 independent v_pk_fma_f32
  C_READLANE,        // 16 x v_readlane_b32 (independent)
  C_NCASES
};
static const char* kNames[C_NCASES] = {"v_fma_f32 indep", "v_fma_f32 dep chain", "v_rsq_f32 indep",
                                       "14 fma + 1 rsq indep", "s_add_u32 indep", "v_fma + s_add alternating",
                                       "cmp->andn2->cmp->cbranch + fma", "record chain (15 VALU + 6 scalar)",
                                       "record chain, VALU only", "two records interleaved, VALU only",
                                       "v_pk_fma_f32 indep", "v_readlane_b32 indep"};
// instructions per unrolled body (what "cycles per instruction" divides by)
static const int kInstr[C_NCASES] = {16, 16, 16, 15, 16, 32, 20, 84, 60, 60, 16, 16};
// VALU instructions per body
static const int kValu[C_NCASES] = {16, 16, 16, 15, 0, 16, 8, 60, 60, 60, 16, 16};

template <int CASE>
__global__ __launch_bounds__(256) void k(float* out, u64* stamps, int iters, float seed) {
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6,
        a7 = seed + 7;
  float b0 = seed, b1 = seed * 2, b2 = seed * 3, b3 = seed * 4, b4 = seed * 5, b5 = seed * 6, b6 = seed * 7,
        b7 = seed * 8;
  float px = threadIdx.x * 0.37f, py = seed, pz = seed * 0.5f;
  int s0 = 1, s1 = 2, s2 = 3, s3 = 4;
  u64 mask = ~0ull;
  const u64 t0 = __builtin_amdgcn_s_memtime();
  const u64 r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
    if (CASE == C_FMA_INDEP) {
      asm volatile(REP4("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    } else if (CASE == C_FMA_DEP) {
      asm volatile(REP16("v_fma_f32 %0, %0, %0, %0\n") : "+v"(a0));
    } else if (CASE == C_RSQ_INDEP) {
      asm volatile(REP4("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    } else if (CASE == C_MIX_14_1) {
      asm volatile(
          "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
          "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_rsq_f32 %7, %7\n"
          "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
          "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (CASE == C_SALU_INDEP) {
      asm volatile(REP4("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n")
                   : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)::"scc");
    } else if (CASE == C_VALU_SALU) {
      asm volatile(REP4("v_fma_f32 %0, %0, %0, %0\n s_add_u32 %4, %4, 1\n v_fma_f32 %1, %1, %1, %1\n s_add_u32 %5, %5, 1\n"
                        "v_fma_f32 %2, %2, %2, %2\n s_add_u32 %6, %6, 1\n v_fma_f32 %3, %3, %3, %3\n s_add_u32 %7, %7, 1\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)::"scc");
    } else if (CASE == C_CMP_SALU) {
      // mask stays ~0 and the compare is always true -> open mask empty -> branch never taken
      asm volatile(REP4("v_fma_f32 %0, %0, %0, %0\n v_cmp_lt_f32 vcc, %2, %1\n s_andn2_b64 s[20:21], %3, vcc\n"
                        "s_cmp_eq_u64 s[20:21], 0\n s_cbranch_scc0 1f\n")
                   "1:\n"
                   : "+v"(a0)
                   : "v"(b0), "s"(-1.0f), "s"(mask)
                   : "vcc", "scc", "s20", "s21");
    } else if (CASE == C_RECORD || CASE == C_RECORD_VALU) {
#define RECORD_BODY(SCALAR_A, SCALAR_B)                                                                  \
  "v_sub_f32 v40, %6, %3\n v_sub_f32 v41, %7, %4\n v_fma_f32 v43, v40, v40, %9\n v_sub_f32 v42, %8, %5\n" \
  "v_fmac_f32 v43, v41, v41\n v_fmac_f32 v43, v42, v42\n v_cmp_lt_f32 vcc, %10, v43\n" SCALAR_A           \
  "v_rsq_f32 v43, v43\n" SCALAR_B                                                                        \
  "v_mul_f32 v44, %11, v43\n v_mul_f32 v43, v43, v43\n v_mul_f32 v43, v44, v43\n"                        \
  "v_cndmask_b32 v43, 0, v43, vcc\n v_fmac_f32 %0, v43, v40\n v_fmac_f32 %1, v43, v41\n v_fmac_f32 %2, v43, v42\n"
      if (CASE == C_RECORD) {
        asm volatile(REP4(RECORD_BODY("s_andn2_b64 s[20:21], %12, vcc\n s_cmp_eq_u64 s[20:21], 0\n s_cbranch_scc0 9f\n",
                                      "s_and_b64 vcc, vcc, %12\n s_cmp_lt_i32 %13, 2\n s_cbranch_scc1 9f\n"))
                     "9:\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2)
                     : "v"(px), "v"(py), "v"(pz), "s"(seed * 3.f), "s"(seed * 5.f), "s"(seed * 7.f), "s"(50.0f),
                       "s"(-1.0f), "s"(seed), "s"(mask), "s"(s0 + 7)
                     : "vcc", "scc", "s20", "s21", "v40", "v41", "v42", "v43", "v44");
      } else {
        asm volatile(REP4(RECORD_BODY("", ""))
                     : "+v"(a0), "+v"(a1), "+v"(a2)
                     : "v"(px), "v"(py), "v"(pz), "s"(seed * 3.f), "s"(seed * 5.f), "s"(seed * 7.f), "s"(50.0f),
                       "s"(-1.0f), "s"(seed), "s"(mask), "s"(s0 + 7)
                     : "vcc", "v40", "v41", "v42", "v43", "v44");
      }
    } else if (CASE == C_RECORD_ILP2) {
      // records A (v40..44, vcc) and B (v45..49, s[22:23]) interleaved
#define ILP2                                                                                           \
  "v_sub_f32 v40, %6, %3\n v_sub_f32 v45, %7, %3\n v_sub_f32 v41, %7, %4\n v_sub_f32 v46, %8, %4\n"     \
  "v_fma_f32 v43, v40, v40, %9\n v_fma_f32 v48, v45, v45, %9\n v_sub_f32 v42, %8, %5\n v_sub_f32 v47, %6, %5\n" \
  "v_fmac_f32 v43, v41, v41\n v_fmac_f32 v48, v46, v46\n v_fmac_f32 v43, v42, v42\n v_fmac_f32 v48, v47, v47\n" \
  "v_cmp_lt_f32 vcc, %10, v43\n v_cmp_lt_f32 s[22:23], %10, v48\n v_rsq_f32 v43, v43\n v_rsq_f32 v48, v48\n"  \
  "v_mul_f32 v44, %11, v43\n v_mul_f32 v49, %11, v48\n v_mul_f32 v43, v43, v43\n v_mul_f32 v48, v48, v48\n" \
  "v_mul_f32 v43, v44, v43\n v_mul_f32 v48, v49, v48\n v_cndmask_b32 v43, 0, v43, vcc\n"                 \
  "v_cndmask_b32_e64 v48, 0, v48, s[22:23]\n"                                                          \
  "v_fmac_f32 %0, v43, v40\n v_fmac_f32 %1, v43, v41\n v_fmac_f32 %2, v43, v42\n"                        \
  "v_fmac_f32 %0, v48, v45\n v_fmac_f32 %1, v48, v46\n v_fmac_f32 %2, v48, v47\n"
      asm volatile(ILP2 ILP2
                   : "+v"(a0), "+v"(a1), "+v"(a2)
                   : "v"(px), "v"(py), "v"(pz), "s"(seed * 3.f), "s"(seed * 5.f), "s"(seed * 7.f), "s"(50.0f),
                     "s"(-1.0f), "s"(seed)
                   : "vcc", "s22", "s23", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49");
    } else if (CASE == C_PKFMA_INDEP) {
      asm volatile(REP4("v_pk_fma_f32 v[40:41], v[40:41], v[40:41], v[40:41]\n v_pk_fma_f32 v[42:43], v[42:43], v[42:43], v[42:43]\n"
                        "v_pk_fma_f32 v[44:45], v[44:45], v[44:45], v[44:45]\n v_pk_fma_f32 v[46:47], v[46:47], v[46:47], v[46:47]\n")
                   ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
    } else if (CASE == C_READLANE) {
      asm volatile(REP4("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9\n")
                   :: "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s20", "s21", "s22", "s23");
    }
  }
  const u64 t1 = __builtin_amdgcn_s_memtime();
  const u64 r1 = __builtin_amdgcn_s_memrealtime();
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if ((threadIdx.x & 63) == 0) {
    stamps[4 * wave] = t1 - t0;
    stamps[4 * wave + 1] = r1 - r0;
    stamps[4 * wave + 2] = r0;
    // HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh[12] se_id[15:13]; XCC_ID is its own register
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
    stamps[4 * wave + 3] = ((u64)(xcc & 0xf) << 32) | hw;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] =
      a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7 + (float)(s0 + s1 + s2 + s3);
}

typedef void (*kern_t)(float*, u64*, int, float);
template <int C>
static void fill(kern_t* t) {
  t[C] = k<C>;
  if constexpr (C + 1 < C_NCASES) fill<C + 1>(t);
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  printf("device: %s, %d CUs, clockRate %d kHz\n", prop.name, cus, prop.clockRate);
  float* out;
  u64* stamps;
  const int maxwaves = cus * 8 * 4;
  hipMalloc(&out, sizeof(float) * 64 * maxwaves);
  hipMalloc(&stamps, sizeof(u64) * 4 * maxwaves);
  std::vector<u64> h(4 * maxwaves);
  kern_t tab[C_NCASES];
  fill<0>(tab);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  const bool verbose = argc > 2;
  printf("%-36s %5s %9s %9s %9s %8s %8s\n", "case", "w/SIMD", "cyc/inst", "cyc/VALU", "cyc/body", "GHz", "ms");
  for (int c = 0; c < C_NCASES; c++)
    for (int bpc = 1; bpc <= 8; bpc *= 2) {  // 256-thread blocks per CU = waves per SIMD
      const int grid = cus * bpc;
      float ms = 0;
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        tab[c]<<<grid, 256>>>(out, stamps, iters, 1.25f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      const int waves = grid * 4;
      hipMemcpy(h.data(), stamps, sizeof(u64) * 4 * waves, hipMemcpyDeviceToHost);
      // median wave: cycles it was alive, and the clock it saw
      std::vector<double> cyc(waves), ghz(waves);
      u64 rmin = ~0ull, rend = 0;
      std::vector<int> per_simd(8 * 8 * 2 * 16 * 4, 0);
      for (int w = 0; w < waves; w++) {
        cyc[w] = (double)h[4 * w];
        ghz[w] = (double)h[4 * w] / ((double)h[4 * w + 1] * 10.0);  // 100 MHz realtime -> 10 ns per tick
        rmin = std::min(rmin, h[4 * w + 2]);
        rend = std::max(rend, h[4 * w + 2] + h[4 * w + 1]);
        const unsigned hw = (unsigned)h[4 * w + 3], xcc = (unsigned)(h[4 * w + 3] >> 32);
        const int simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_simd[((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4) + simd]++;
      }
      if (verbose) {
        std::vector<double> st(waves);
        for (int w = 0; w < waves; w++) st[w] = (double)(h[4 * w + 2] - rmin) * 0.01;  // us
        std::sort(st.begin(), st.end());
        int hist[40] = {0}, used = 0;
        for (int v : per_simd) if (v) { hist[std::min(v, 39)]++; used++; }
        printf("   wave start us: p50 %.1f p90 %.1f max %.1f; span %.1f us; lifetime min/p50/max %.0f/%.0f/%.0f kcyc; SIMDs used %d:",
               st[waves / 2], st[waves * 9 / 10], st[waves - 1], (double)(rend - rmin) * 0.01,
               *std::min_element(cyc.begin(), cyc.end()) * 1e-3, cyc[waves / 2] * 1e-3,
               *std::max_element(cyc.begin(), cyc.end()) * 1e-3, used);
        for (int v = 1; v < 40; v++) if (hist[v]) printf(" %dx%d", hist[v], v);
        printf(" (SIMDs x waves)\n");
      }
      std::sort(cyc.begin(), cyc.end());
      std::sort(ghz.begin(), ghz.end());
      const double wave_cyc = cyc[waves / 2];
      // bpc waves share one SIMD: SIMD cycles per body = wave lifetime / (iters * bpc)
      const double per_body = wave_cyc / ((double)iters * bpc);
      printf("%-36s %5d %9.2f %9.2f %9.1f %8.3f %8.3f\n", kNames[c], bpc, per_body / kInstr[c],
             kValu[c] ? per_body / kValu[c] : 0.0, per_body, ghz[waves / 2], ms);
    }
  return 0;
}
