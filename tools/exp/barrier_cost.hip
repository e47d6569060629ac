// micro-benchmark: cost of an s_barrier loop for 4- and 8-wave workgroups, with and without 160 KB of LDS
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ unsigned char smem[];
template <int WORK>
__global__ void k(int iters, float* out) {
  float acc = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int w = 0; w < WORK; ++w) acc = acc * 1.0001f + 0.5f;
  }
  if (acc == 12345.678f) out[0] = acc;
}
template <int WORK>
void run(int threads, size_t lds, int iters) {
  float* out; hipMalloc(&out, 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<WORK>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<WORK><<<256, threads, lds>>>(iters, out); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) k<WORK><<<256, threads, lds>>>(iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("threads %d lds %zu work %d iters %d: %.2f us per launch, %.1f ns per iteration\n", threads, lds, WORK, iters, ms * 100, ms * 1e5 / iters);
}
int main() {
  for (int threads : {256, 512, 1024}) for (size_t lds : {(size_t)0, (size_t)163840}) { run<0>(threads, lds, 1000); run<64>(threads, lds, 1000); }
  run<0>(512, 163840, 40); run<0>(512, 163840, 0);
  return 0;
}
