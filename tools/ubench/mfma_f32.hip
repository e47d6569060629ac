// Issue rate of v_mfma_f32_32x32x2_f32 (the exact-fp32 matrix instruction of vq_nearest) on gfx950: cycles per instruction and SIMD
// with 1 / 2 / 4 independent accumulator chains, 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f32 tools/ubench/mfma_f32.hip && /tmp/mfma_f32
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int CH>
__global__ __launch_bounds__(512) void k(uint64_t* out, float* sink, int iters) {
  f32x16 acc[CH];
  for (int c = 0; c < CH; ++c)
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int c = 0; c < CH; ++c)
    for (int i = 0; i < 16; ++i) s += acc[c][i];
  if (s == 123.f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}
// the A operand of every MFMA is written by a VALU instruction right in front of it (as in vq_nearest: a select of the lane's half)
template <int CH, int GAP>
__global__ __launch_bounds__(512) void kdep(uint64_t* out, float* sink, int iters) {
  f32x16 acc[CH];
  for (int c = 0; c < CH; ++c)
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f, e = 0.f;
  const bool hh = (threadIdx.x & 32) != 0;
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        float av;
        asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(av) : "v"(a), "v"(b), "s"(__builtin_amdgcn_ballot_w64(hh)));
#pragma unroll
        for (int g = 0; g < GAP; ++g) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(e) : "v"(b));
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[c], 0, 0, 0);
      }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  float s = e;
  for (int c = 0; c < CH; ++c)
    for (int i = 0; i < 16; ++i) s += acc[c][i];
  if (s == 123.f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}
template <int CH, int GAP>
void rundep(uint64_t* d, float* sink, int wps) {
  uint64_t h[8];
  const int iters = 200;
  kdep<CH, GAP><<<1, 256 * wps>>>(d, sink, 10);
  kdep<CH, GAP><<<1, 256 * wps>>>(d, sink, iters);
  hipDeviceSynchronize();
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  uint64_t mx = 0;
  for (int w = 0; w < 4 * wps; ++w) mx = h[w] > mx ? h[w] : mx;
  printf("  A written by VALU in front of each MFMA, %d VALU in between, chains %d, waves/SIMD %d: %6.1f cycles per MFMA per SIMD\n", GAP, CH, wps,
         (double)mx / ((double)iters * 16 * CH * wps));
}
template <int CH>
void run(uint64_t* d, float* sink, int wps) {
  uint64_t h[8];
  const int iters = 200;
  k<CH><<<1, 256 * wps>>>(d, sink, 10);
  k<CH><<<1, 256 * wps>>>(d, sink, iters);
  hipDeviceSynchronize();
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  uint64_t mx = 0;
  for (int w = 0; w < 4 * wps; ++w) mx = h[w] > mx ? h[w] : mx;
  printf("  chains %d, waves/SIMD %d: %6.1f cycles per MFMA per SIMD\n", CH, wps, (double)mx / ((double)iters * 16 * CH * wps));
}
// the same chain on every CU at once, timed with HIP events: does the chip hold its clock under full fp32 / bf16 MFMA load?
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
template <int KIND>
__global__ __launch_bounds__(256) void kfull(float* sink, int iters) {
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  bf16x8_t fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(0.01f * i); fb[i] = (__bf16)(0.02f * i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (KIND == 0) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
      } else {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  if (s == 123.f) sink[0] = s;
}
template <int KIND>
void runfull(float* sink, int grid) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  kfull<KIND><<<grid, 256>>>(sink, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kfull<KIND><<<grid, 256>>>(sink, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double per = ms * 1e6 / ((double)iters * 32);   // ns per MFMA of one wave (= per SIMD: one wave per SIMD)
  const double cyc = KIND == 0 ? 64.0 : 32.0;
  printf("  %s on %3d workgroups (one wave per SIMD): %6.2f ns per MFMA per SIMD = %5.2f GHz at %g cycles each\n",
         KIND == 0 ? "f32 32x32x2 " : "bf16 32x32x16", grid, per, cyc / per, cyc);
}
int main() {
  uint64_t* d; float* sink;
  hipMalloc(&d, 64 * 8); hipMalloc(&sink, 64);
  runfull<0>(sink, 1); runfull<0>(sink, 32); runfull<0>(sink, 256);
  runfull<1>(sink, 1); runfull<1>(sink, 32); runfull<1>(sink, 256);
  for (int w = 1; w <= 2; ++w) { run<1>(d, sink, w); run<2>(d, sink, w); run<4>(d, sink, w); }
  for (int w = 1; w <= 2; ++w) { rundep<1, 0>(d, sink, w); rundep<2, 0>(d, sink, w); rundep<2, 2>(d, sink, w); rundep<2, 6>(d, sink, w); }
  return 0;
}
