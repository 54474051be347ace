// How many waves of a kernel does a gfx950 SIMD hold, as a function of the kernel's SGPR and VGPR counts?
// (round 4: the force walk uses 102 + 6 SGPRs and tools/force_trace.py found 6 resident waves per SIMD, not 8.)
// Each wave stamps its start on the 100 MHz clock and sleeps ~40 us; the waves that started within 10 us of the first
// are the resident set.  hipcc --offload-arch=gfx950 -O2 tools/ubench_occ.hip -o tools/bin/ubench_occ
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

template <int NS>
__global__ __launch_bounds__(64) void occ_kernel(unsigned* out) {
  const unsigned t0 = (unsigned)__builtin_amdgcn_s_memrealtime();
  // touch the highest SGPR so that the kernel's SGPR count is NS + 1 (+ VCC etc.)
  if (NS == 60) asm volatile("s_mov_b32 s60, 0" ::: "s60");
  if (NS == 72) asm volatile("s_mov_b32 s72, 0" ::: "s72");
  if (NS == 76) asm volatile("s_mov_b32 s76, 0" ::: "s76");
  if (NS == 80) asm volatile("s_mov_b32 s80, 0" ::: "s80");
  if (NS == 84) asm volatile("s_mov_b32 s84, 0" ::: "s84");
  if (NS == 88) asm volatile("s_mov_b32 s88, 0" ::: "s88");
  if (NS == 92) asm volatile("s_mov_b32 s92, 0" ::: "s92");
  if (NS == 96) asm volatile("s_mov_b32 s96, 0" ::: "s96");
  if (NS == 101) asm volatile("s_mov_b32 s101, 0" ::: "s101");
  for (int k = 0; k < 400; k++) __builtin_amdgcn_s_sleep(127);  // ~ 400 x 127 x 64 cycles
  if (threadIdx.x == 0) {
    out[blockIdx.x * 2] = t0;
    out[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
  }
}

template <int NS>
void run(unsigned* d, int blocks) {
  hipFuncAttributes fa;
  (void)hipFuncGetAttributes(&fa, (const void*)occ_kernel<NS>);
  hipMemset(d, 0, blocks * 8);
  occ_kernel<NS><<<blocks, 64>>>(d);
  hipDeviceSynchronize();
  std::vector<unsigned> h(blocks * 2);
  hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
  unsigned tmin = ~0u;
  for (int b = 0; b < blocks; b++) tmin = std::min(tmin, h[b * 2]);
  int early = 0, maxslot = 0;
  for (int b = 0; b < blocks; b++) {
    if (h[b * 2] - tmin < 1000) early++;  // 10 us
    maxslot = std::max(maxslot, (int)(h[b * 2 + 1] & 0xF));
  }
  printf("highest SGPR s%-3d  numRegs(sgpr) %3d  vgpr %3d : %5d waves resident at once = %.2f per SIMD (max wave slot id %d)\n",
         NS, fa.numRegs, 0, early, early / 1024.0, maxslot);
}

int main() {
  const int blocks = 1024 * 12;
  unsigned* d;
  hipMalloc(&d, blocks * 8);
  run<60>(d, blocks); run<72>(d, blocks); run<76>(d, blocks); run<80>(d, blocks); run<84>(d, blocks);
  run<88>(d, blocks); run<92>(d, blocks); run<96>(d, blocks); run<101>(d, blocks);
  return 0;
}
