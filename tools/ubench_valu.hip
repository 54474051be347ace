// ubench_valu.hip — design-study microbenchmark (not product code): how fast can a gfx950 SIMD
// issue the force kernel's per-record VALU sequence when nothing else is in the way?
// Each wave evaluates `iters` x 4 synthetic records held in SGPRs (no memory in the loop) with the
// same 15-VALU body as force_fast_kernel; full occupancy.  Prints cycles per (record, wave) per
// SIMD for several waves-per-SIMD settings, and a variant with the open-path branch per record.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/ubench_valu.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef unsigned long long u64;

template <int BRANCHY>
__global__ __launch_bounds__(256) void k(float* out, int iters, float eps2, int xbits, float gm, float thr2,
                                         u64 mask_in, int* sink) {
  const int lane = threadIdx.x & 63;
  float px = (float)lane * 0.37f, py = (float)(threadIdx.x >> 6), pz = 1.5f + blockIdx.x * 1e-3f;
  float ax = 0, ay = 0, az = 0;
  u64 mask = mask_in;
  int pushes = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      // wave-uniform record, varies per iteration on the scalar unit
      const float rx = __int_as_float(xbits + ((it * 4 + kk) & 1023));
      const float ry = __int_as_float(xbits + ((it * 7 + kk) & 1023));
      const float rz = __int_as_float(xbits + ((it * 3 + kk) & 1023));
      const float dx = rx - px, dy = ry - py, dz = rz - pz;
      const float d2 = fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, eps2)));
      const u64 accm = __builtin_amdgcn_ballot_w64(d2 > thr2);
      const u64 takem = mask & accm, openm = mask & ~accm;
      const float rinv = __builtin_amdgcn_rsqf(d2);
      const float f = (gm * rinv) * (rinv * rinv);
      const float fm = __builtin_amdgcn_inverse_ballot_w64(takem) ? f : 0.0f;
      ax = fmaf(fm, dx, ax);
      ay = fmaf(fm, dy, ay);
      az = fmaf(fm, dz, az);
      if (BRANCHY) {
        if (openm != 0ull) {  // uniform branch, as on the open path
          pushes++;
          mask ^= openm >> 1;
        }
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = ax + ay + az;
  if (pushes == 123456789) *sink = pushes;
}

// two bodies per lane: the scalar/branch overhead of a record is paid once per 128 bodies
__global__ __launch_bounds__(256) void k2(float* out, int iters, float eps2, int xbits, float gm, float thr2,
                                          u64 mask_in, int* sink) {
  const int lane = threadIdx.x & 63;
  float px[2], py[2], pz[2], ax[2] = {0, 0}, ay[2] = {0, 0}, az[2] = {0, 0};
  px[0] = (float)lane * 0.37f; px[1] = px[0] + 30.f;
  py[0] = (float)(threadIdx.x >> 6); py[1] = py[0] + 1.f;
  pz[0] = 1.5f + blockIdx.x * 1e-3f; pz[1] = pz[0] + 2.f;
  u64 mask[2] = {mask_in, mask_in};
  int pushes = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      const float rx = __int_as_float(xbits + ((it * 4 + kk) & 1023));
      const float ry = __int_as_float(xbits + ((it * 7 + kk) & 1023));
      const float rz = __int_as_float(xbits + ((it * 3 + kk) & 1023));
      u64 openm[2];
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const float dx = rx - px[j], dy = ry - py[j], dz = rz - pz[j];
        const float d2 = fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, eps2)));
        const u64 accm = __builtin_amdgcn_ballot_w64(d2 > thr2);
        const u64 takem = mask[j] & accm;
        openm[j] = mask[j] & ~accm;
        const float rinv = __builtin_amdgcn_rsqf(d2);
        const float f = (gm * rinv) * (rinv * rinv);
        const float fm = __builtin_amdgcn_inverse_ballot_w64(takem) ? f : 0.0f;
        ax[j] = fmaf(fm, dx, ax[j]);
        ay[j] = fmaf(fm, dy, ay[j]);
        az[j] = fmaf(fm, dz, az[j]);
      }
      if ((openm[0] | openm[1]) != 0ull) {
        pushes++;
        mask[0] ^= openm[0] >> 1;
        mask[1] ^= openm[1] >> 1;
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = ax[0] + ay[0] + az[0] + ax[1] + ay[1] + az[1];
  if (pushes == 123456789) *sink = pushes;
}

// packed fp32: two records per VALU instruction (v_pk_add/fma/mul_f32), 9 VALU per record
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void kpk(float* out, int iters, float eps2, int xbits, float gm, float thr2,
                                           u64 mask_in, int* sink) {
  const int lane = threadIdx.x & 63;
  const float px = (float)lane * 0.37f, py = (float)(threadIdx.x >> 6), pz = 1.5f + blockIdx.x * 1e-3f;
  const f2 px2 = {px, px}, py2 = {py, py}, pz2 = {pz, pz}, e2 = {eps2, eps2};
  f2 ax = {0, 0}, ay = {0, 0}, az = {0, 0};
  u64 mask = mask_in;
  int pushes = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int kk = 0; kk < 4; kk += 2) {
      f2 rx, ry, rz;
      rx[0] = __int_as_float(xbits + ((it * 4 + kk) & 1023));
      rx[1] = __int_as_float(xbits + ((it * 4 + kk + 1) & 1023));
      ry[0] = __int_as_float(xbits + ((it * 7 + kk) & 1023));
      ry[1] = __int_as_float(xbits + ((it * 7 + kk + 1) & 1023));
      rz[0] = __int_as_float(xbits + ((it * 3 + kk) & 1023));
      rz[1] = __int_as_float(xbits + ((it * 3 + kk + 1) & 1023));
      const f2 dx = rx - px2, dy = ry - py2, dz = rz - pz2;
      const f2 d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dx, dx, e2)));
      const u64 acc0 = __builtin_amdgcn_ballot_w64(d2[0] > thr2);
      const u64 acc1 = __builtin_amdgcn_ballot_w64(d2[1] > thr2);
      const u64 open = (mask & ~acc0) | (mask & ~acc1);
      f2 rinv;
      rinv[0] = __builtin_amdgcn_rsqf(d2[0]);
      rinv[1] = __builtin_amdgcn_rsqf(d2[1]);
      const f2 gm2 = {gm, gm};
      f2 f = (gm2 * rinv) * (rinv * rinv);
      f[0] = __builtin_amdgcn_inverse_ballot_w64(mask & acc0) ? f[0] : 0.0f;
      f[1] = __builtin_amdgcn_inverse_ballot_w64(mask & acc1) ? f[1] : 0.0f;
      ax = __builtin_elementwise_fma(f, dx, ax);
      ay = __builtin_elementwise_fma(f, dy, ay);
      az = __builtin_elementwise_fma(f, dz, az);
      if (open != 0ull) {
        pushes++;
        mask ^= open >> 1;
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = ax[0] + ax[1] + ay[0] + ay[1] + az[0] + az[1];
  if (pushes == 123456789) *sink = pushes;
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  float* out;
  int* sink;
  hipMalloc(&out, sizeof(float) * 256 * cus * 8);
  hipMalloc(&sink, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  const float x0 = 100.0f;
  int xbits;
  memcpy(&xbits, &x0, 4);
  for (int branchy = 0; branchy < 4; branchy++)
    for (int bpc = 1; bpc <= 8; bpc *= 2) {  // blocks per CU: 1,2,4,8 -> waves/SIMD 1,2,4,8
      const int grid = cus * bpc;
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (branchy == 3)  // packed fp32, two records per instruction, one branch per pair
          kpk<<<grid, 256>>>(out, iters, 50.0f, xbits, 2.0f, 1e9f, ~0ull, sink);
        else if (branchy == 2)  // two bodies per lane, one branch per record
          k2<<<grid, 256>>>(out, iters, 50.0f, xbits, 2.0f, 1e9f, ~0ull, sink);
        else if (branchy)
          k<1><<<grid, 256>>>(out, iters, 50.0f, xbits, 2.0f, 1e9f, ~0ull, sink);
        else
          k<0><<<grid, 256>>>(out, iters, 50.0f, xbits, 2.0f, 1e9f, ~0ull, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double wave_records = (double)grid * 4 * iters * 4;
      const double per_simd = wave_records / (cus * 4.0);
      printf("branchy=%d waves/SIMD=%d: %.3f ms, %.1f ns per (record,wave) per SIMD = %.1f cycles @2.4GHz\n",
             branchy, bpc, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    }
  return 0;
}
