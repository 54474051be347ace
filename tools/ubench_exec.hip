// ubench_exec.hip — design-study microbenchmark (not product code), round 3:
//   (1) does a wave64 VALU instruction get cheaper when EXEC covers only an aligned half / quarter of the wave?
//       (the force walk evaluates 42 % of its lane slots with the lane masked off)
//   (2) what do the walk's SCALAR instructions cost in context: the packed pair chain (16 VALU) followed by 0 / 2 /
//       4 / 8 independent SALU instructions, 8 waves per SIMD
// Longest wave lifetime (s_memtime), as tools/ubench_forms.hip.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_exec.hip -o tools/bin/ubench_exec && tools/bin/ubench_exec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned long long u64;
#define CH2 "v_pk_add_f32 v[40:41], s[20:21], v[48:49] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 v[42:43], s[22:23], v[50:51] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 v[44:45], s[24:25], v[52:53] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n" \
   "v_pk_fma_f32 v[46:47], v[40:41], v[40:41], v[60:61]\n v_pk_fma_f32 v[46:47], v[42:43], v[42:43], v[46:47]\n v_pk_fma_f32 v[46:47], v[44:45], v[44:45], v[46:47]\n" \
   "v_cmp_nlt_f32_e64 s[30:31], s28, v46\n v_cmp_nlt_f32_e64 s[32:33], s29, v47\n v_rsq_f32 v62, v46\n v_rsq_f32 v63, v47\n" \
   "v_pk_mul_f32 v[46:47], s[26:27], v[62:63]\n v_pk_mul_f32 v[62:63], v[62:63], v[62:63]\n v_pk_mul_f32 v[62:63], v[46:47], v[62:63]\n" \
   "v_pk_fma_f32 v[54:55], v[62:63], v[40:41], v[54:55]\n v_pk_fma_f32 v[56:57], v[62:63], v[42:43], v[56:57]\n v_pk_fma_f32 v[58:59], v[62:63], v[44:45], v[58:59]\n"
#define S2 "s_or_b64 s[34:35], s[30:31], s[32:33]\n s_add_u32 s36, s36, 1\n"
#define CLOB "scc","vcc","s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","s32","s33","s34","s35","s36","s37","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63"
// NS = scalar instructions appended per pair chain; the EXEC mask is a kernel argument
// EXEC is narrowed and restored INSIDE each asm block (the compiler must never see a partial EXEC): 2 extra SALU
// per 4 pair chains in the masked runs
#define MIN "s_mov_b64 s[38:39], exec\n s_mov_b64 exec, %1\n"
#define MOUT "s_mov_b64 exec, s[38:39]\n"
template <int NS>
__global__ __launch_bounds__(256) void k(float* out, u64* stamps, int iters, u64 mask) {
  float a0 = 1.0f;
  const u64 t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    if (NS == -1) asm volatile(MIN CH2 CH2 CH2 CH2 MOUT : "+v"(a0) : "s"(mask) : "s38", "s39", CLOB);
    if (NS == 0) asm volatile(CH2 CH2 : "+v"(a0)::CLOB);
    if (NS == 2) asm volatile(CH2 S2 CH2 S2 : "+v"(a0)::CLOB);
    if (NS == 4) asm volatile(CH2 S2 S2 CH2 S2 S2 : "+v"(a0)::CLOB);
    if (NS == 8) asm volatile(CH2 S2 S2 S2 S2 CH2 S2 S2 S2 S2 : "+v"(a0)::CLOB);
  }
  const u64 t1 = __builtin_amdgcn_s_memtime();
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if ((threadIdx.x & 63) == 0) stamps[wave] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0;
}
typedef void (*kern_t)(float*, u64*, int, u64);
int main() {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  float* out; u64* stamps;
  hipMalloc(&out, sizeof(float) * 256 * cus * 8); hipMalloc(&stamps, sizeof(u64) * cus * 32);
  std::vector<u64> h(cus * 32);
  const int iters = 20000;
  struct { const char* name; u64 m; } masks[] = {{"all 64 lanes", ~0ull}, {"lanes 0-31", 0xffffffffull}, {"lanes 32-63", 0xffffffff00000000ull},
      {"lanes 0-15", 0xffffull}, {"lanes 16-31", 0xffff0000ull}, {"lane 0", 1ull}, {"even lanes", 0x5555555555555555ull}};
  kern_t ks[4] = {k<0>, k<2>, k<4>, k<8>};
  const int ns[4] = {0, 2, 4, 8};
  for (int bpc = 2; bpc <= 8; bpc *= 4) {
    const int grid = cus * bpc;
    for (auto& mk : masks) {
      for (int rep = 0; rep < 2; rep++) { k<-1><<<grid, 256>>>(out, stamps, iters / 2, mk.m); hipDeviceSynchronize(); }
      hipMemcpy(h.data(), stamps, sizeof(u64) * grid * 4, hipMemcpyDeviceToHost);
      const double mx = (double)*std::max_element(h.begin(), h.begin() + grid * 4);
      printf("pair chain (16 VALU), EXEC = %-12s w/SIMD %d: %6.1f cycles per pair\n", mk.name, bpc, mx / ((double)(iters / 2) * bpc * 4));
    }
    for (int v = 0; v < 4; v++) {
      for (int rep = 0; rep < 2; rep++) { ks[v]<<<grid, 256>>>(out, stamps, iters, ~0ull); hipDeviceSynchronize(); }
      hipMemcpy(h.data(), stamps, sizeof(u64) * grid * 4, hipMemcpyDeviceToHost);
      const double mx = (double)*std::max_element(h.begin(), h.begin() + grid * 4);
      printf("pair chain + %d SALU per pair                 w/SIMD %d: %6.1f cycles per pair\n", ns[v], bpc, mx / ((double)iters * bpc * 2));
    }
  }
  return 0;
}
