// L2 -> CU fill-rate micro-benchmark for gfx950: bytes per clock and CU that a GEMM-like operand stream achieves through
//   mode 0  LDS-DMA (global_load_lds_dwordx4: 8 rows x 128 B per wave instruction, the tile-staging pattern of gemm.hip),
//   mode 1  global_load_dwordx4 into VGPRs, same 8 x 128 B pattern,
//   mode 2  global_load_dwordx4 into VGPRs in the MFMA B-fragment pattern (32 rows x 2 halves x 16 B per wave instruction),
//   mode 3  modes 0 and 2 interleaved one to one (do the two return paths add up?).
// Sources: a matrix of `rows` rows x 1024 B (K = 512 bf16); every workgroup walks its own 128-row panel (like an A tile) or
// all workgroups walk the same 128-row panel set of a small matrix (like a weight panel in L2).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fill_rate tools/ubench/fill_rate.hip && /tmp/fill_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void fill_kernel(const unsigned char* __restrict__ src, int64_t rows, int panels, int iters,
                                                   uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 64 KB ring for the DMA modes
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  // panel p of this workgroup: rows (p * gridDim.x + blockIdx.x) * 128 ... (mod rows)
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (int it = 0; it < iters; ++it) {
    const int64_t panel = ((int64_t)it * gridDim.x + blockIdx.x) % panels;
    const unsigned char* pbase = src + panel * 128 * 1024;            // 128 rows x 1024 B
    // one "k-step" = 128 rows x 128 B = 16 KB per workgroup = 4 wave instructions of 1 KB per wave; 8 k-steps per panel
#pragma unroll 1
    for (int ks = 0; ks < 8; ++ks) {
      if (MODE == 0 || MODE == 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = (wave * 4 + i) * 8 + (lane >> 3);
          const uint32_t voff = (uint32_t)(r * 1024 + ks * 128 + (lane & 7) * 16);
          const uint32_t lds = __builtin_amdgcn_readfirstlane(lds0 + ((ks & 3) * 16384) + (wave * 4 + i) * 1024);
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(pbase), "s"(lds) : "memory", "m0");
        }
      }
      if (MODE == 1) {
        u32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = (wave * 4 + i) * 8 + (lane >> 3);
          v[i] = *reinterpret_cast<const u32x4*>(pbase + r * 1024 + ks * 128 + (lane & 7) * 16);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc |= v[i];
      }
      if (MODE == 2 || MODE == 3) {
        // wave w owns rows 32 w .. 32 w + 31; per k16: lane -> row 32 w + (lane & 31), bytes ks * 128 + k16 * 32 + (lane >> 5) * 16
        u32x4 v[4];
#pragma unroll
        for (int k16 = 0; k16 < 4; ++k16)
          v[k16] = *reinterpret_cast<const u32x4*>(pbase + (wave * 32 + (lane & 31)) * 1024 + ks * 128 + k16 * 32 + (lane >> 5) * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc |= v[i];
      }
      if (MODE == 0 || MODE == 3) {
        if ((ks & 1) == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // keep ~1-2 k-steps of DMA in flight
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if ((acc[0] | acc[1] | acc[2] | acc[3]) == 0x12345u) sink[0] = 1;
}

template <int MODE>
int run(const unsigned char* src, int64_t rows, const char* what, int wg_per_cu, uint32_t* sink) {
  const int panels = (int)(rows / 128), iters = 64;
  const int grid = 256 * wg_per_cu;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  fill_kernel<MODE><<<grid, 256, 65536>>>(src, rows, panels, 4, sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  fill_kernel<MODE><<<grid, 256, 65536>>>(src, rows, panels, iters, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)grid * iters * 131072.0 * (MODE == 3 ? 2.0 : 1.0);
  const double clk = ms * 1e-3 * 2.4e9;
  printf("  mode %d %-34s %d WG/CU: %7.1f us  %6.2f TB/s  %5.1f B/clk/CU\n", MODE, what, wg_per_cu, ms * 1e3, bytes / (ms * 1e-3) / 1e12,
         bytes / clk / 256.0);
  return 0;
}

int main() {
  uint32_t* sink;
  CHECK(hipMalloc(&sink, 64));
  const int64_t sizes[3] = {2048, 9248 + 96, 1 << 20};   // rows of 1 KB: 2 MB (weights: L2), 9.5 MB (activations), 1 GB (HBM)
  const char* names[3] = {"2 MB matrix (L2-resident)", "9.5 MB matrix (L2 / MALL)", "1 GB matrix (HBM stream)"};
  for (int s = 0; s < 3; ++s) {
    unsigned char* src;
    CHECK(hipMalloc(&src, sizes[s] * 1024));
    CHECK(hipMemset(src, 1, sizes[s] * 1024));
    for (int w = 1; w <= 2; ++w) {
      if (run<0>(src, sizes[s], names[s], w, sink)) return 1;
      if (run<1>(src, sizes[s], names[s], w, sink)) return 1;
      if (run<2>(src, sizes[s], names[s], w, sink)) return 1;
      if (run<3>(src, sizes[s], names[s], w, sink)) return 1;
    }
    CHECK(hipFree(src));
  }
  return 0;
}
