// ubench_sstore.hip — design-study microbenchmark (not product code), round 3: could the force walk keep its
// spilled stack entries in MEMORY through the scalar data cache (s_store_dwordx4 / s_load_dwordx4: one SMEM issue
// each) instead of in the lanes of three VGPRs (3 v_writelane + 3 v_readlane, ~5 cycles each in that loop)?
//   (1) do scalar stores work on gfx950 at all: every wave writes 64 entries, reads them back, counts mismatches;
//   (2) what do they cost in context: the packed pair chain (16 VALU) + per iteration nothing / one lane-stack
//       push + pop / one memory-stack push + pop, 8 waves per SIMD, longest wave lifetime (s_memtime).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_sstore.hip -o tools/bin/ubench_sstore && timeout -k 5 60 tools/bin/ubench_sstore
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned long long u64;
typedef int v4i __attribute__((ext_vector_type(4)));
#define CH2 "v_pk_add_f32 v[40:41], s[20:21], v[48:49] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 v[42:43], s[22:23], v[50:51] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 v[44:45], s[24:25], v[52:53] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n" \
   "v_pk_fma_f32 v[46:47], v[40:41], v[40:41], v[60:61]\n v_pk_fma_f32 v[46:47], v[42:43], v[42:43], v[46:47]\n v_pk_fma_f32 v[46:47], v[44:45], v[44:45], v[46:47]\n" \
   "v_cmp_nlt_f32_e64 s[30:31], s28, v46\n v_cmp_nlt_f32_e64 s[32:33], s29, v47\n v_rsq_f32 v62, v46\n v_rsq_f32 v63, v47\n" \
   "v_pk_mul_f32 v[46:47], s[26:27], v[62:63]\n v_pk_mul_f32 v[62:63], v[62:63], v[62:63]\n v_pk_mul_f32 v[62:63], v[46:47], v[62:63]\n" \
   "v_pk_fma_f32 v[54:55], v[62:63], v[40:41], v[54:55]\n v_pk_fma_f32 v[56:57], v[62:63], v[42:43], v[56:57]\n v_pk_fma_f32 v[58:59], v[62:63], v[44:45], v[58:59]\n"
#define CLOB "scc","vcc","m0","s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","s32","s33","s34","s35","s36","s37","s40","s41","s42","s43","v36","v37","v38","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63"

// (1b) may the data registers of a scalar store be overwritten right behind it?  (the walk would: the spilled top
// entry's registers receive the new top at once).  dword + dwordx2 stores with soffset + immediate, as planned there.
__global__ __launch_bounds__(64) void check_overwrite(int* buf, int* bad) {
  const int wave = blockIdx.x;
  int* base = buf + (size_t)wave * 256;
  int wrong = 0;
  for (int d = 0; d < 64; d++) {
    const int off = d * 16, a = d * 7 + 1, b = 0x5a5a0000 | d, c = ~d;
    asm volatile("s_mov_b32 s40, %0\n s_mov_b32 s42, %1\n s_mov_b32 s43, %2\n"
                 "s_store_dword s40, %3, %4\n s_store_dwordx2 s[42:43], %3, %4 offset:8\n"
                 "s_mov_b32 s40, -1\n s_mov_b32 s42, -1\n s_mov_b32 s43, -1\n"
                 ::"s"(a), "s"(b), "s"(c), "s"(base), "s"(off) : "memory", "s40", "s42", "s43");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int d = 63; d >= 0; d--) {
    v4i r;
    const int off = d * 16;
    asm volatile("s_load_dwordx4 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(base), "s"(off) : "memory");
    wrong += (r.x != d * 7 + 1) + (r.z != (0x5a5a0000 | d)) + (r.w != ~d);
  }
  if (threadIdx.x == 0) bad[wave] = wrong;
}

__global__ __launch_bounds__(64) void check(int* buf, int* bad) {
  const int wave = blockIdx.x;
  int* base = buf + (size_t)wave * 256;  // 64 entries of 16 bytes
  int wrong = 0;
  for (int d = 0; d < 64; d++) {
    v4i e = {d * 7 + 1, wave, 0x5a5a0000 | d, ~d};
    const int off = d * 16;
    asm volatile("s_store_dwordx4 %0, %1, %2\n" ::"s"(e), "s"(base), "s"(off) : "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int d = 63; d >= 0; d--) {
    v4i r;
    const int off = d * 16;
    asm volatile("s_load_dwordx4 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(base), "s"(off) : "memory");
    wrong += (r.x != d * 7 + 1) + (r.y != wave) + (r.z != (0x5a5a0000 | d)) + (r.w != ~d);
  }
  asm volatile("s_dcache_wb" ::: "memory");
  if (threadIdx.x == 0) bad[wave] = wrong;
}

// MODE 0: pair chain only; 1: + push and pop of one entry through the lanes of v36-v38; 2: through memory
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, u64* stamps, int iters, int* buf) {
  float a0 = 1.0f;
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  int* base = buf + (size_t)wave * 256;
  const u64 t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    const int depth = it & 63;
    if (MODE == 0) asm volatile(CH2 CH2 : "+v"(a0)::CLOB);
    if (MODE == 1)
      asm volatile("s_mov_b32 m0, %1\n v_writelane_b32 v36, s30, m0\n v_writelane_b32 v37, s31, m0\n v_writelane_b32 v38, s32, m0\n"
                   CH2 "s_mov_b32 m0, %1\n s_nop 0\n v_readlane_b32 s40, v36, m0\n v_readlane_b32 s41, v37, m0\n v_readlane_b32 s42, v38, m0\n" CH2
                   : "+v"(a0) : "s"(depth) : CLOB);
    if (MODE == 2)
      asm volatile("s_lshl_b32 s34, %1, 4\n s_store_dwordx4 s[28:31], %2, s34\n" CH2
                   "s_load_dwordx4 s[40:43], %2, s34\n" CH2 "s_waitcnt lgkmcnt(0)\n"
                   : "+v"(a0) : "s"(depth), "s"(base) : CLOB, "memory");
  }
  const u64 t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) stamps[wave] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0;
}

int main() {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const int waves = cus * 32;
  int *buf, *bad; float* out; u64* stamps;
  hipMalloc(&buf, (size_t)waves * 1024); hipMalloc(&bad, waves * 4);
  hipMalloc(&out, sizeof(float) * 64 * waves); hipMalloc(&stamps, sizeof(u64) * waves);
  hipMemset(buf, 0, (size_t)waves * 1024); hipMemset(bad, 0xff, waves * 4);
  check<<<waves, 64>>>(buf, bad);
  hipError_t e = hipDeviceSynchronize();
  printf("scalar store check: %s\n", hipGetErrorString(e));
  if (e != hipSuccess) return 1;
  std::vector<int> hb(waves), hbuf((size_t)waves * 256);
  hipMemcpy(hb.data(), bad, waves * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hbuf.data(), buf, (size_t)waves * 1024, hipMemcpyDeviceToHost);
  long wrong = 0, wrong_mem = 0;
  for (int w = 0; w < waves; w++) {
    wrong += hb[w];
    for (int d = 0; d < 64; d++) wrong_mem += hbuf[(size_t)w * 256 + 4 * d] != d * 7 + 1;
  }
  printf("read back through the scalar cache: %ld mismatching dwords over %d waves x 64 entries; in memory after s_dcache_wb: %ld wrong entries\n", wrong, waves, wrong_mem);
  hipMemset(buf, 0, (size_t)waves * 1024); hipMemset(bad, 0xff, waves * 4);
  check_overwrite<<<waves, 64>>>(buf, bad);
  e = hipDeviceSynchronize();
  hipMemcpy(hb.data(), bad, waves * 4, hipMemcpyDeviceToHost);
  wrong = 0;
  for (int w = 0; w < waves; w++) wrong += hb[w];
  printf("store data registers overwritten right behind the store (dword + dwordx2, soffset + imm): %s, %ld mismatching dwords\n", hipGetErrorString(e), wrong);
  std::vector<u64> h(waves);
  const int iters = 20000;
  const char* names[3] = {"2 pair chains", "2 pair chains + lane push + pop (3 v_writelane, 3 v_readlane)", "2 pair chains + memory push + pop (s_store_dwordx4, s_load_dwordx4)"};
  for (int mode = 0; mode < 3; mode++) {
    for (int rep = 0; rep < 2; rep++) {
      if (mode == 0) k<0><<<cus * 8, 256>>>(out, stamps, iters, buf);
      if (mode == 1) k<1><<<cus * 8, 256>>>(out, stamps, iters, buf);
      if (mode == 2) k<2><<<cus * 8, 256>>>(out, stamps, iters, buf);
      e = hipDeviceSynchronize();
      if (e != hipSuccess) { printf("mode %d: %s\n", mode, hipGetErrorString(e)); return 1; }
    }
    hipMemcpy(h.data(), stamps, sizeof(u64) * waves, hipMemcpyDeviceToHost);
    const double mx = (double)*std::max_element(h.begin(), h.end());
    printf("%-70s w/SIMD 8: %7.1f cycles per iteration\n", names[mode], mx / ((double)iters * 8));
  }
  return 0;
}
