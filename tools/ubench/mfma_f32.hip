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
int main() {
  uint64_t* d; float* sink;
  hipMalloc(&d, 64 * 8); hipMalloc(&sink, 64);
  for (int w = 1; w <= 2; ++w) { run<1>(d, sink, w); run<2>(d, sink, w); run<4>(d, sink, w); }
  return 0;
}
