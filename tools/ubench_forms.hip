// ubench_forms.hip — design-study microbenchmark (not product code): issue cost of the exact VALU
// instruction FORMS of the force kernel's record chain (SGPR operands, VOP2 vs VOP3, v_cmp, v_cndmask),
// independent instructions, 8 waves per SIMD, measured by the longest wave lifetime (s_memtime).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_forms.hip -o tools/bin/ubench_forms && tools/bin/ubench_forms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned long long u64;
#define REP4(x) x x x x
enum { F_FMA_VVV, F_FMA_VVS, F_SUB_VV, F_SUB_SV, F_FMAC_VV, F_MUL_SV, F_MUL_VV, F_CMP_VV, F_CMP_SV, F_CMP64_SV, F_CNDMASK_VCC,
       F_CNDMASK_S, F_RSQ, F_MOV, F_ADD_U32, F_PKADD_SV, F_PKADD_VV, F_PKMUL_SV, F_PKMUL_VV, F_PKFMA_VVV, F_PKFMA_BCAST, F_CHAIN1, F_CHAIN2, F_NCASES };
static const char* kNames[F_NCASES] = {"v_fma_f32 v,v,v,v", "v_fma_f32 v,v,v,s", "v_sub_f32 v,v,v", "v_sub_f32 v,s,v",
  "v_fmac_f32 v,v,v", "v_mul_f32 v,s,v", "v_mul_f32 v,v,v", "v_cmp_lt_f32 vcc,v,v", "v_cmp_lt_f32 vcc,s,v",
  "v_cmp_lt_f32_e64 s[..],s,v", "v_cndmask_b32 v,0,v,vcc", "v_cndmask_b32_e64 v,0,v,s[..]", "v_rsq_f32", "v_mov_b32 v,v", "v_add_u32 v,v,v", "v_pk_add_f32 v2,s2,v2 (neg)", "v_pk_add_f32 v2,v2,v2", "v_pk_mul_f32 v2,s2,v2", "v_pk_mul_f32 v2,v2,v2", "v_pk_fma_f32 v2,v2,v2,v2", "v_pk_fma_f32 v2,v2,v(bcast),v2", "record chain, 1 record (15 VALU, current forms)", "pair chain, 2 records (packed forms)"};
template <int F>
__global__ __launch_bounds__(256) void k(float* out, u64* stamps, int iters, float seed) {
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, b0 = seed * 2, b1 = seed * 3;
  const u64 t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#define I4(S) asm volatile(REP4(S) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "s"(seed) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27")
    if (F == F_FMA_VVV) I4("v_fma_f32 %0,%4,%5,%0\n v_fma_f32 %1,%4,%5,%1\n v_fma_f32 %2,%4,%5,%2\n v_fma_f32 %3,%4,%5,%3\n");
    if (F == F_FMA_VVS) I4("v_fma_f32 %0,%4,%0,%6\n v_fma_f32 %1,%4,%1,%6\n v_fma_f32 %2,%4,%2,%6\n v_fma_f32 %3,%4,%3,%6\n");
    if (F == F_SUB_VV) I4("v_sub_f32 %0,%4,%0\n v_sub_f32 %1,%4,%1\n v_sub_f32 %2,%4,%2\n v_sub_f32 %3,%4,%3\n");
    if (F == F_SUB_SV) I4("v_sub_f32 %0,%6,%0\n v_sub_f32 %1,%6,%1\n v_sub_f32 %2,%6,%2\n v_sub_f32 %3,%6,%3\n");
    if (F == F_FMAC_VV) I4("v_fmac_f32 %0,%4,%5\n v_fmac_f32 %1,%4,%5\n v_fmac_f32 %2,%4,%5\n v_fmac_f32 %3,%4,%5\n");
    if (F == F_MUL_SV) I4("v_mul_f32 %0,%6,%0\n v_mul_f32 %1,%6,%1\n v_mul_f32 %2,%6,%2\n v_mul_f32 %3,%6,%3\n");
    if (F == F_MUL_VV) I4("v_mul_f32 %0,%4,%0\n v_mul_f32 %1,%4,%1\n v_mul_f32 %2,%4,%2\n v_mul_f32 %3,%4,%3\n");
    if (F == F_CMP_VV) I4("v_cmp_lt_f32 vcc,%4,%0\n v_cmp_lt_f32 vcc,%4,%1\n v_cmp_lt_f32 vcc,%4,%2\n v_cmp_lt_f32 vcc,%4,%3\n");
    if (F == F_CMP_SV) I4("v_cmp_lt_f32 vcc,%6,%0\n v_cmp_lt_f32 vcc,%6,%1\n v_cmp_lt_f32 vcc,%6,%2\n v_cmp_lt_f32 vcc,%6,%3\n");
    if (F == F_CMP64_SV) I4("v_cmp_lt_f32_e64 s[20:21],%6,%0\n v_cmp_lt_f32_e64 s[22:23],%6,%1\n v_cmp_lt_f32_e64 s[24:25],%6,%2\n v_cmp_lt_f32_e64 s[26:27],%6,%3\n");
    if (F == F_CNDMASK_VCC) I4("v_cndmask_b32 %0,0,%0,vcc\n v_cndmask_b32 %1,0,%1,vcc\n v_cndmask_b32 %2,0,%2,vcc\n v_cndmask_b32 %3,0,%3,vcc\n");
    if (F == F_CNDMASK_S) I4("v_cndmask_b32_e64 %0,0,%0,s[20:21]\n v_cndmask_b32_e64 %1,0,%1,s[22:23]\n v_cndmask_b32_e64 %2,0,%2,s[24:25]\n v_cndmask_b32_e64 %3,0,%3,s[26:27]\n");
    if (F == F_RSQ) I4("v_rsq_f32 %0,%0\n v_rsq_f32 %1,%1\n v_rsq_f32 %2,%2\n v_rsq_f32 %3,%3\n");
    if (F == F_MOV) I4("v_mov_b32 %0,%4\n v_mov_b32 %1,%4\n v_mov_b32 %2,%4\n v_mov_b32 %3,%4\n");
    if (F == F_ADD_U32) I4("v_add_u32 %0,%4,%0\n v_add_u32 %1,%4,%1\n v_add_u32 %2,%4,%2\n v_add_u32 %3,%4,%3\n");

    if (F == F_PKADD_SV) asm volatile(REP4("v_pk_add_f32 v[40:41], s[20:21], v[40:41] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 v[42:43], s[22:23], v[42:43] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 v[44:45], s[24:25], v[44:45] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 v[46:47], s[26:27], v[46:47] neg_lo:[0,1] neg_hi:[0,1]\n") ::: "v40","v41","v42","v43","v44","v45","v46","v47");
    if (F == F_PKADD_VV) asm volatile(REP4("v_pk_add_f32 v[40:41], v[48:49], v[40:41]\n v_pk_add_f32 v[42:43], v[48:49], v[42:43]\n v_pk_add_f32 v[44:45], v[48:49], v[44:45]\n v_pk_add_f32 v[46:47], v[48:49], v[46:47]\n") ::: "v40","v41","v42","v43","v44","v45","v46","v47");
    if (F == F_PKMUL_SV) asm volatile(REP4("v_pk_mul_f32 v[40:41], s[20:21], v[40:41]\n v_pk_mul_f32 v[42:43], s[22:23], v[42:43]\n v_pk_mul_f32 v[44:45], s[24:25], v[44:45]\n v_pk_mul_f32 v[46:47], s[26:27], v[46:47]\n") ::: "v40","v41","v42","v43","v44","v45","v46","v47");
    if (F == F_PKMUL_VV) asm volatile(REP4("v_pk_mul_f32 v[40:41], v[48:49], v[40:41]\n v_pk_mul_f32 v[42:43], v[48:49], v[42:43]\n v_pk_mul_f32 v[44:45], v[48:49], v[44:45]\n v_pk_mul_f32 v[46:47], v[48:49], v[46:47]\n") ::: "v40","v41","v42","v43","v44","v45","v46","v47");
    if (F == F_PKFMA_VVV) asm volatile(REP4("v_pk_fma_f32 v[40:41], v[48:49], v[50:51], v[40:41]\n v_pk_fma_f32 v[42:43], v[48:49], v[50:51], v[42:43]\n v_pk_fma_f32 v[44:45], v[48:49], v[50:51], v[44:45]\n v_pk_fma_f32 v[46:47], v[48:49], v[50:51], v[46:47]\n") ::: "v40","v41","v42","v43","v44","v45","v46","v47");
    if (F == F_PKFMA_BCAST) asm volatile(REP4("v_pk_fma_f32 v[40:41], v[48:49], v[50:51], v[40:41] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[42:43], v[48:49], v[50:51], v[42:43] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[44:45], v[48:49], v[50:51], v[44:45] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[46:47], v[48:49], v[50:51], v[46:47] op_sel_hi:[1,0,1]\n") ::: "v40","v41","v42","v43","v44","v45","v46","v47");
    // one record, the forms of the current kernel (e64 cndmask), x4 (dependent chain, as compiled)
#define CH1(X,Y,Z,GM,THR) "v_sub_f32 v40, " X ", %4\n v_sub_f32 v41, " Y ", %5\n v_sub_f32 v42, " Z ", %4\n v_fma_f32 v43, v40, v40, %6\n v_fmac_f32 v43, v41, v41\n v_fmac_f32 v43, v42, v42\n" \
   "v_cmp_lt_f32_e64 s[28:29], " THR ", v43\n v_rsq_f32 v44, v43\n v_mul_f32 v43, " GM ", v44\n v_mul_f32 v44, v44, v44\n v_mul_f32 v44, v43, v44\n v_cndmask_b32_e64 v44, 0, v44, s[28:29]\n" \
   "v_fmac_f32 %0, v44, v40\n v_fmac_f32 %1, v44, v41\n v_fmac_f32 %2, v44, v42\n"
    if (F == F_CHAIN1) asm volatile(CH1("s20","s21","s22","s23","s24") CH1("s21","s22","s23","s24","s25") CH1("s22","s23","s24","s25","s26") CH1("s23","s24","s25","s26","s27")
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "s"(seed) : "s28","s29","v40","v41","v42","v43","v44");
    // two records A,B as packed pairs: window x=s[20:21] y=s[22:23] z=s[24:25] gm=s[26:27] thr=s28,s29 (as laid out pairwise);
    // body position broadcast: v[48:49] = (px, -), v[50:51] = (py,-), v[52:53] = (pz,-) ; accumulators v[54:55], v[56:57], v[58:59]; eps2 pair v[60:61]
#define CH2 "v_pk_add_f32 v[40:41], s[20:21], v[48:49] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 v[42:43], s[22:23], v[50:51] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 v[44:45], s[24:25], v[52:53] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n" \
   "v_pk_fma_f32 v[46:47], v[40:41], v[40:41], v[60:61]\n v_pk_fma_f32 v[46:47], v[42:43], v[42:43], v[46:47]\n v_pk_fma_f32 v[46:47], v[44:45], v[44:45], v[46:47]\n" \
   "v_cmp_lt_f32_e64 s[30:31], s28, v46\n v_cmp_lt_f32_e64 s[32:33], s29, v47\n v_rsq_f32 v62, v46\n v_rsq_f32 v63, v47\n" \
   "v_pk_mul_f32 v[46:47], s[26:27], v[62:63]\n v_pk_mul_f32 v[62:63], v[62:63], v[62:63]\n v_pk_mul_f32 v[62:63], v[46:47], v[62:63]\n" \
   "v_pk_fma_f32 v[54:55], v[62:63], v[40:41], v[54:55]\n v_pk_fma_f32 v[56:57], v[62:63], v[42:43], v[56:57]\n v_pk_fma_f32 v[58:59], v[62:63], v[44:45], v[58:59]\n"
    if (F == F_CHAIN2) asm volatile(CH2 CH2
        : "+v"(a0) :: "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","s32","s33","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63");
  }
  const u64 t1 = __builtin_amdgcn_s_memtime();
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if ((threadIdx.x & 63) == 0) stamps[wave] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
typedef void (*kern_t)(float*, u64*, int, float);
template <int C> static void fill(kern_t* t) { t[C] = k<C>; if constexpr (C + 1 < F_NCASES) fill<C + 1>(t); }
int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  float* out; u64* stamps;
  hipMalloc(&out, sizeof(float) * 256 * cus * 8); hipMalloc(&stamps, sizeof(u64) * cus * 32);
  std::vector<u64> h(cus * 32);
  kern_t tab[F_NCASES]; fill<0>(tab);
  const int iters = 20000;
  for (int c = 0; c < F_NCASES; c++)
    for (int bpc = 2; bpc <= 8; bpc *= 4) {
      const int grid = cus * bpc;
      for (int rep = 0; rep < 2; rep++) { tab[c]<<<grid, 256>>>(out, stamps, iters, 1.25f); hipDeviceSynchronize(); }
      hipMemcpy(h.data(), stamps, sizeof(u64) * grid * 4, hipMemcpyDeviceToHost);
      const double mx = (double)*std::max_element(h.begin(), h.begin() + grid * 4);
      // 16 instructions + 3 loop instructions per iteration; report per instruction of the 19
      if (c == F_CHAIN1 || c == F_CHAIN2)
        printf("%-48s w/SIMD %d: %6.1f cycles per record (4 records per iteration)\n", kNames[c], bpc, mx / ((double)iters * bpc * 4));
      else
        printf("%-48s w/SIMD %d: %5.2f cycles per instruction (16 + 3 loop overhead per iteration)\n", kNames[c], bpc,
               mx / ((double)iters * bpc * 19));
    }
  return 0;
}
