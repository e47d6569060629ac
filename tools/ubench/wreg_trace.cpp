// Cycle stamps of the weights-in-registers NT kernel (csrc/gemm.hip built with -DWR_TRACE=1): which part of a phase takes the time.
//   hipcc -O2 -o wreg_trace wreg_trace.cpp -ldl ; ./wreg_trace nt_trace.so [epi] [N]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef int (*nt_fn)(const void*, int64_t, const void*, int64_t, void*, int64_t, const float*, void*, int32_t, int32_t, int32_t,
                     int32_t, const float*, float, uint64_t, const uint32_t*, float*, void*);
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
int main(int argc, char** argv) {
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  nt_fn nt = (nt_fn)dlsym(h, "ttts_gemm_nt_bf16_ex");
  const int epi = argc > 2 ? atoi(argv[2]) : 0, N = argc > 3 ? atoi(argv[3]) : 2048, M = 9248, K = 512;
  void *dA, *dB, *dC, *dAux; float* dbias; unsigned long long* dtr;
  HIP(hipMalloc(&dA, (size_t)M * K * 2)); HIP(hipMalloc(&dB, (size_t)N * K * 2)); HIP(hipMalloc(&dC, (size_t)M * N * 4));
  HIP(hipMalloc(&dAux, (size_t)M * N * 2)); HIP(hipMalloc(&dbias, N * 4)); HIP(hipMalloc(&dtr, 4 * 128 * 8));
  HIP(hipMemset(dA, 0x3c, (size_t)M * K * 2)); HIP(hipMemset(dB, 0x3c, (size_t)N * K * 2)); HIP(hipMemset(dAux, 0x3c, (size_t)M * N * 2));
  HIP(hipMemset(dbias, 0, N * 4));
  for (int it = 0; it < 5; ++it) {
    HIP(hipMemset(dtr, 0, 4 * 128 * 8));
    int rc = nt(dA, K, dB, K, dC, N, dbias, dAux, M, N, K, epi, nullptr, 0.f, 0, nullptr, (float*)dtr, nullptr);
    if (rc) { fprintf(stderr, "rc %d\n", rc); return 3; }
    HIP(hipDeviceSynchronize());
  }
  std::vector<unsigned long long> t(4 * 128);
  HIP(hipMemcpy(t.data(), dtr, 4 * 128 * 8, hipMemcpyDeviceToHost));
  const char* names[4] = {"block 3 wave 0 (group 0)", "block 3 wave 4 (group 1)", "block 200 wave 0", "block 200 wave 4"};
  unsigned long long t0 = t[0];
  for (int w = 0; w < 4; ++w) {
    printf("%s: stamps relative to block 3 / wave 0 entry (cycles), then deltas\n ", names[w]);
    for (int i = 0; i < 128 && t[w * 128 + i]; ++i) printf(" %lld", (long long)(t[w * 128 + i] - t0));
    printf("\n  d:");
    for (int i = 1; i < 128 && t[w * 128 + i]; ++i) printf(" %lld", (long long)(t[w * 128 + i] - t[w * 128 + i - 1]));
    printf("\n");
  }
  return 0;
}
