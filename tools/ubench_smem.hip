// ubench_smem.hip — design-study microbenchmark (not product code): throughput of the scalar data cache
// (SQC) for the force kernel's access shape: 4 x s_load_dwordx16 (256 B) per "block", then a wait.
// Cases: footprint the random block addresses are drawn from (small = K$ hits, large = L2 / MALL hits),
// 32-B vs 64-B aligned blocks, 1..8 waves per SIMD, and vector-ALU work between the loads.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_smem.hip -o tools/bin/ubench_smem && tools/bin/ubench_smem
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef unsigned long long u64;

// LOADS = s_load_dwordx16 per block (1, 2 or 4); VALU = dependent-free v_fma per block
template <int LOADS, int VALU>
__global__ __launch_bounds__(256) void k(const float* base, unsigned mask, unsigned align_mask, int iters, float* out,
                                         u64* stamps) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  unsigned state = __builtin_amdgcn_readfirstlane(wave * 2654435761u + 12345u);
  float acc = 0.f, a0 = 1.f, a1 = 2.f, a2 = 3.f, a3 = 4.f;
  const u64 t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    state = state * 1664525u + 1013904223u;
    const unsigned off = (state >> 4) & mask & align_mask;  // byte offset of the block
    float s;
    if (LOADS == 4)
      asm volatile(
          "s_load_dwordx16 s[36:51], %1, %2 offset:0\n s_load_dwordx16 s[52:67], %1, %2 offset:64\n"
          "s_load_dwordx16 s[68:83], %1, %2 offset:128\n s_load_dwordx16 s[84:99], %1, %2 offset:192\n"
          "s_waitcnt lgkmcnt(0)\n s_add_u32 %0, s36, s52\n s_add_u32 %0, %0, s68\n s_add_u32 %0, %0, s84\n"
          : "=s"(s) : "s"(base), "s"(off)
          : "scc", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50",
            "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65",
            "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80",
            "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95",
            "s96", "s97", "s98", "s99");
    else if (LOADS == 2)
      asm volatile(
          "s_load_dwordx16 s[36:51], %1, %2 offset:0\n s_load_dwordx16 s[52:67], %1, %2 offset:64\n"
          "s_waitcnt lgkmcnt(0)\n s_add_u32 %0, s36, s52\n"
          : "=s"(s) : "s"(base), "s"(off)
          : "scc", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50",
            "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65",
            "s66", "s67");
    else
      asm volatile("s_load_dwordx16 s[36:51], %1, %2 offset:0\n s_waitcnt lgkmcnt(0)\n s_mov_b32 %0, s36\n"
                   : "=s"(s) : "s"(base), "s"(off)
                   : "scc", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49",
                     "s50", "s51");
    acc += s;
#pragma unroll
    for (int v = 0; v < VALU / 4; v++)
      asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
  }
  const u64 t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) stamps[wave] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + a0 + a1 + a2 + a3;
}

template <int L, int V>
static void run(const char* name, const float* buf, unsigned mask, unsigned amask, int cus, float* out, u64* stamps,
                std::vector<u64>& h) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 4000;
  for (int bpc = 1; bpc <= 8; bpc *= 2) {
    const int grid = cus * bpc;
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      k<L, V><<<grid, 256>>>(buf, mask, amask, iters, out, stamps);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const int waves = grid * 4;
    hipMemcpy(h.data(), stamps, sizeof(u64) * waves, hipMemcpyDeviceToHost);
    const double maxcyc = (double)*std::max_element(h.begin(), h.begin() + waves);
    // per CU: blocks per cycle = (waves per CU * iters) / max wave lifetime
    const double blocks_per_cu = (double)bpc * 4 * iters;
    printf("%-44s w/SIMD %d: %7.1f cyc per block per CU  (%5.1f B/cyc/CU, %6.2f TB/s chip)  %.3f ms\n", name, bpc,
           maxcyc / blocks_per_cu, L * 64.0 * blocks_per_cu / maxcyc,
           L * 64.0 * (double)waves * iters / (ms * 1e-3) * 1e-12, ms);
  }
}


// Half-window pipeline: a block = 2 x s_load_dwordx16 (128 B) + VALU fma work that READS the window.
// PREFETCH = 0: load -> wait -> compute.  PREFETCH = 1: wait -> issue the NEXT block's loads into the other
// 32-SGPR window -> compute on this one (SMEM returns out of order, so only lgkmcnt(0) is usable: the wait
// comes first, then the prefetch).
template <int PREFETCH, int VALU>
__global__ __launch_bounds__(256) void kp(const float* base, unsigned mask, int iters, float* out, u64* stamps) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  unsigned state = __builtin_amdgcn_readfirstlane(wave * 2654435761u + 12345u);
  float a0 = 1.f, a1 = 2.f, a2 = 3.f, a3 = 4.f;
  const u64 t0 = __builtin_amdgcn_s_memtime();
#define W0 "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67"
#define W1 "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99"
#define NEXT_OFF state = state * 1664525u + 1013904223u; off = (state >> 4) & mask & ~31u;
#define LOAD0 asm volatile("s_load_dwordx16 s[36:51], %0, %1 offset:0\n s_load_dwordx16 s[52:67], %0, %1 offset:64\n" ::"s"(base), "s"(off) : W0)
#define LOAD1 asm volatile("s_load_dwordx16 s[68:83], %0, %1 offset:0\n s_load_dwordx16 s[84:99], %0, %1 offset:64\n" ::"s"(base), "s"(off) : W1)
#define WAIT asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define WORK(A, B, C, D)                                                                                  \
  _Pragma("unroll") for (int v = 0; v < VALU / 4; v++)                                                    \
      asm volatile("v_fma_f32 %0, " A ", %0, %0\n v_fma_f32 %1, " B ", %1, %1\n v_fma_f32 %2, " C ", %2, %2\n v_fma_f32 %3, " D ", %3, %3\n" \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3))
  unsigned off;
  if (PREFETCH) {
    NEXT_OFF LOAD0;
    for (int it = 0; it < iters; it += 2) {
      WAIT; NEXT_OFF LOAD1; WORK("s36", "s44", "s52", "s60");
      WAIT; NEXT_OFF LOAD0; WORK("s68", "s76", "s84", "s92");
    }
    WAIT;
  } else {
    for (int it = 0; it < iters; it += 2) {
      NEXT_OFF LOAD0; WAIT; WORK("s36", "s44", "s52", "s60");
      NEXT_OFF LOAD1; WAIT; WORK("s68", "s76", "s84", "s92");
    }
  }
  const u64 t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) stamps[wave] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

template <int P, int V>
static void runp(const char* name, const float* buf, unsigned mask, int cus, float* out, u64* stamps, std::vector<u64>& h) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 4000;
  for (int bpc = 2; bpc <= 8; bpc *= 2) {
    const int grid = cus * bpc;
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      kp<P, V><<<grid, 256>>>(buf, mask, iters, out, stamps);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const int waves = grid * 4;
    hipMemcpy(h.data(), stamps, sizeof(u64) * waves, hipMemcpyDeviceToHost);
    const double maxcyc = (double)*std::max_element(h.begin(), h.begin() + waves);
    printf("%-52s w/SIMD %d: %7.1f cyc per 128-B block per CU   %.3f ms\n", name, bpc, maxcyc / ((double)bpc * 4 * iters), ms);
  }
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const size_t bytes = 256u << 20;
  float* buf;
  hipMalloc(&buf, bytes + 4096);
  hipMemset(buf, 0, bytes + 4096);
  float* out;
  u64* stamps;
  hipMalloc(&out, sizeof(float) * 256 * cus * 8);
  hipMalloc(&stamps, sizeof(u64) * cus * 32);
  std::vector<u64> h(cus * 32);
  const unsigned A32 = ~31u, A64 = ~63u, A256 = ~255u;
  run<4, 0>("4x16dw, 4 KB footprint (K$ hits), 32B-aligned", buf, (4u << 10) - 1, A32, cus, out, stamps, h);
  run<4, 0>("4x16dw, 2 MB footprint (L2), 32B-aligned", buf, (2u << 20) - 1, A32, cus, out, stamps, h);
  run<4, 0>("4x16dw, 2 MB footprint (L2), 256B-aligned", buf, (2u << 20) - 1, A256, cus, out, stamps, h);
  run<4, 0>("4x16dw, 64 MB footprint (MALL), 32B-aligned", buf, (64u << 20) - 1, A32, cus, out, stamps, h);
  run<4, 0>("4x16dw, 64 MB footprint (MALL), 256B-aligned", buf, (64u << 20) - 1, A256, cus, out, stamps, h);
  run<2, 0>("2x16dw, 64 MB footprint, 64B-aligned", buf, (64u << 20) - 1, A64, cus, out, stamps, h);
  run<1, 0>("1x16dw, 64 MB footprint, 64B-aligned", buf, (64u << 20) - 1, A64, cus, out, stamps, h);
  run<4, 96>("4x16dw + 96 v_fma, 64 MB, 32B-aligned", buf, (64u << 20) - 1, A32, cus, out, stamps, h);
  run<4, 96>("4x16dw + 96 v_fma, 4 KB (K$ hits)", buf, (4u << 10) - 1, A32, cus, out, stamps, h);
  runp<0, 48>("128B block + 48 fma, 64 MB, no prefetch", buf, (64u << 20) - 1, cus, out, stamps, h);
  runp<1, 48>("128B block + 48 fma, 64 MB, prefetch 1 ahead", buf, (64u << 20) - 1, cus, out, stamps, h);
  runp<0, 48>("128B block + 48 fma, 2 MB, no prefetch", buf, (2u << 20) - 1, cus, out, stamps, h);
  runp<1, 48>("128B block + 48 fma, 2 MB, prefetch 1 ahead", buf, (2u << 20) - 1, cus, out, stamps, h);
  runp<0, 96>("128B block + 96 fma, 64 MB, no prefetch", buf, (64u << 20) - 1, cus, out, stamps, h);
  runp<1, 96>("128B block + 96 fma, 64 MB, prefetch 1 ahead", buf, (64u << 20) - 1, cus, out, stamps, h);
  runp<0, 0>("128B block, no work, 64 MB, no prefetch", buf, (64u << 20) - 1, cus, out, stamps, h);
  return 0;
}
