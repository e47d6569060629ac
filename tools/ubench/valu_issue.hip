// Instruction-issue micro-benchmark for gfx950: how many cycles does one SIMD spend per wave-instruction of the kinds the
// attention inner loops are made of, alone and beside MFMAs, at 1 / 2 / 3 waves per SIMD?  Standalone (no torch):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue tools/ubench/valu_issue.hip && /tmp/valu_issue
// Every test is a loop of ITER iterations over a block of 64 copies of the pattern; registers are independent 8-deep chains, so
// issue rate, not dependency latency, is what is measured.  Values are garbage on purpose (asm VALU gets no hazard padding).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define MIX8X(i) MIX8
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

enum { OP_FMA, OP_EXP, OP_ADD, OP_MAD24, OP_CMPSEL, OP_CVTPK, OP_MAX3, OP_MIX, OP_MFMA, OP_MIX_MFMA, OP_MIX_NOEXP_MFMA, OP_EXP_FMA, OP_LDS128, OP_N };
static const char* NAMES[OP_N] = {"v_fma_f32", "v_exp_f32", "v_add_f32", "v_mad_u32_u24", "v_cmp_le_u32+v_cndmask", "v_cvt_pk_bf16_f32",
                                  "v_max3_f32", "softmax mix (fma exp add mad cmp sel; 6)", "mfma_32x32x16_bf16 (2 chains)",
                                  "8 scores of the mix (48) + 1 mfma", "8 scores without exp (40) + 1 mfma", "exp,fma alternating (2)", "ds_read_b128"};
// wave-instructions per loop iteration
static const int PER_IT[OP_N] = {64, 64, 64, 64, 128, 64, 64, 8 * 48, 64, 8 * 3 * 49, 8 * 3 * 41, 128, 64};

template <int OP>
__global__ __launch_bounds__(1024) void k(uint64_t* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  float a[8], b[8];
  uint32_t u[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = 1.0f + i; u[i] = threadIdx.x * 2654435761u + i; }
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(0.01f * i); fb[i] = (__bf16)(0.02f * i); }
  f32x16 acc0, acc1, acc2;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; acc2[i] = 0.f; }
  lds[threadIdx.x] = a[0];
  __syncthreads();
  const uint32_t la = (threadIdx.x & 63) * 16;
  float4 ld = {0, 0, 0, 0};
  const float c = 0.5f, m2 = 3.0f;
  const uint32_t thr = 0x19999999u;
  __builtin_amdgcn_s_barrier();
  const uint64_t t0 = __builtin_readcyclecounter();
  const uint64_t r0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#define MIX8 asm volatile("v_fma_f32 %0, %0, %24, %25\n\tv_fma_f32 %1, %1, %24, %25\n\tv_fma_f32 %2, %2, %24, %25\n\tv_fma_f32 %3, %3, %24, %25\n\tv_fma_f32 %4, %4, %24, %25\n\tv_fma_f32 %5, %5, %24, %25\n\tv_fma_f32 %6, %6, %24, %25\n\tv_fma_f32 %7, %7, %24, %25\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_add_f32 %8, %8, %0\n\tv_add_f32 %9, %9, %1\n\tv_add_f32 %10, %10, %2\n\tv_add_f32 %11, %11, %3\n\tv_add_f32 %12, %12, %4\n\tv_add_f32 %13, %13, %5\n\tv_add_f32 %14, %14, %6\n\tv_add_f32 %15, %15, %7\n\tv_mad_u32_u24 %16, %16, %26, %16\n\tv_mad_u32_u24 %17, %17, %26, %17\n\tv_mad_u32_u24 %18, %18, %26, %18\n\tv_mad_u32_u24 %19, %19, %26, %19\n\tv_mad_u32_u24 %20, %20, %26, %20\n\tv_mad_u32_u24 %21, %21, %26, %21\n\tv_mad_u32_u24 %22, %22, %26, %22\n\tv_mad_u32_u24 %23, %23, %26, %23\n\tv_cmp_le_u32 vcc, %26, %16\n\tv_cndmask_b32 %0, 0, %0, vcc\n\tv_cmp_le_u32 vcc, %26, %17\n\tv_cndmask_b32 %1, 0, %1, vcc\n\tv_cmp_le_u32 vcc, %26, %18\n\tv_cndmask_b32 %2, 0, %2, vcc\n\tv_cmp_le_u32 vcc, %26, %19\n\tv_cndmask_b32 %3, 0, %3, vcc\n\tv_cmp_le_u32 vcc, %26, %20\n\tv_cndmask_b32 %4, 0, %4, vcc\n\tv_cmp_le_u32 vcc, %26, %21\n\tv_cndmask_b32 %5, 0, %5, vcc\n\tv_cmp_le_u32 vcc, %26, %22\n\tv_cndmask_b32 %6, 0, %6, vcc\n\tv_cmp_le_u32 vcc, %26, %23\n\tv_cndmask_b32 %7, 0, %7, vcc" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]), "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "v"(c), "v"(m2), "v"(thr) : "vcc");
#define MIXNE8 asm volatile("v_fma_f32 %0, %0, %24, %25\n\tv_fma_f32 %1, %1, %24, %25\n\tv_fma_f32 %2, %2, %24, %25\n\tv_fma_f32 %3, %3, %24, %25\n\tv_fma_f32 %4, %4, %24, %25\n\tv_fma_f32 %5, %5, %24, %25\n\tv_fma_f32 %6, %6, %24, %25\n\tv_fma_f32 %7, %7, %24, %25\n\tv_add_f32 %8, %8, %0\n\tv_add_f32 %9, %9, %1\n\tv_add_f32 %10, %10, %2\n\tv_add_f32 %11, %11, %3\n\tv_add_f32 %12, %12, %4\n\tv_add_f32 %13, %13, %5\n\tv_add_f32 %14, %14, %6\n\tv_add_f32 %15, %15, %7\n\tv_mad_u32_u24 %16, %16, %26, %16\n\tv_mad_u32_u24 %17, %17, %26, %17\n\tv_mad_u32_u24 %18, %18, %26, %18\n\tv_mad_u32_u24 %19, %19, %26, %19\n\tv_mad_u32_u24 %20, %20, %26, %20\n\tv_mad_u32_u24 %21, %21, %26, %21\n\tv_mad_u32_u24 %22, %22, %26, %22\n\tv_mad_u32_u24 %23, %23, %26, %23\n\tv_cmp_le_u32 vcc, %26, %16\n\tv_cndmask_b32 %0, 0, %0, vcc\n\tv_cmp_le_u32 vcc, %26, %17\n\tv_cndmask_b32 %1, 0, %1, vcc\n\tv_cmp_le_u32 vcc, %26, %18\n\tv_cndmask_b32 %2, 0, %2, vcc\n\tv_cmp_le_u32 vcc, %26, %19\n\tv_cndmask_b32 %3, 0, %3, vcc\n\tv_cmp_le_u32 vcc, %26, %20\n\tv_cndmask_b32 %4, 0, %4, vcc\n\tv_cmp_le_u32 vcc, %26, %21\n\tv_cndmask_b32 %5, 0, %5, vcc\n\tv_cmp_le_u32 vcc, %26, %22\n\tv_cndmask_b32 %6, 0, %6, vcc\n\tv_cmp_le_u32 vcc, %26, %23\n\tv_cndmask_b32 %7, 0, %7, vcc" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]), "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "v"(c), "v"(m2), "v"(thr) : "vcc");
    if (OP == OP_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(m2));
      REP64(X)
#undef X
    } else if (OP == OP_EXP) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      REP64(X)
#undef X
    } else if (OP == OP_ADD) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      REP64(X)
#undef X
    } else if (OP == OP_MAD24) {
#define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(u[i]) : "v"(thr));
      REP64(X)
#undef X
    } else if (OP == OP_CMPSEL) {
#define X(i) asm volatile("v_cmp_le_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, 0, %0, vcc" : "+v"(a[i]) : "v"(thr), "v"(u[i]) : "vcc");
      REP64(X)
#undef X
    } else if (OP == OP_CVTPK) {
#define X(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(b[i]));
      REP64(X)
#undef X
    } else if (OP == OP_MAX3) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c));
      REP64(X)
#undef X
    } else if (OP == OP_MIX) {
      REP8(MIX8X)
    } else if (OP == OP_EXP_FMA) {
#define X(i) asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(a[i]), "+v"(b[i]) : "v"(c), "v"(m2));
      REP64(X)
#undef X
    } else if (OP == OP_MFMA) {
#define X(i)                                                                     \
  acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);         \
  acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if (OP == OP_MIX_MFMA) {
#define X(i)                                                                                   \
  acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);                       \
  __builtin_amdgcn_sched_barrier(0); MIX8 __builtin_amdgcn_sched_barrier(0);                   \
  acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);                       \
  __builtin_amdgcn_sched_barrier(0); MIX8 __builtin_amdgcn_sched_barrier(0);                   \
  acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc2, 0, 0, 0);                       \
  __builtin_amdgcn_sched_barrier(0); MIX8 __builtin_amdgcn_sched_barrier(0);
      REP8(X)
#undef X
    } else if (OP == OP_MIX_NOEXP_MFMA) {
#define X(i)                                                                                   \
  acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);                       \
  __builtin_amdgcn_sched_barrier(0); MIXNE8 __builtin_amdgcn_sched_barrier(0);                 \
  acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);                       \
  __builtin_amdgcn_sched_barrier(0); MIXNE8 __builtin_amdgcn_sched_barrier(0);                 \
  acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc2, 0, 0, 0);                       \
  __builtin_amdgcn_sched_barrier(0); MIXNE8 __builtin_amdgcn_sched_barrier(0);
      REP8(X)
#undef X
    } else if (OP == OP_LDS128) {
#define X(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld) : "v"(la), "n"(i * 1024));
      REP64(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  const uint64_t r1 = wall_clock64();
  float s = ld.x;
  for (int i = 0; i < 8; ++i) s += a[i] + b[i] + (float)u[i];
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i];
  if (s == 123.456f) out[1000] = 1;
  if ((threadIdx.x & 63) == 0) {
    out[2 * (threadIdx.x >> 6)] = t1 - t0;
    out[2 * (threadIdx.x >> 6) + 1] = r1 - r0;
  }
}

template <int OP>
void run(uint64_t* d, int waves_per_simd) {
  const int iters = 200, threads = 256 * waves_per_simd;
  uint64_t h[64];
  k<OP><<<1, threads>>>(d, 10);
  k<OP><<<1, threads>>>(d, iters);
  hipDeviceSynchronize();
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  uint64_t mx = 0, mr = 0;
  for (int w = 0; w < threads / 64; ++w) { if (h[2 * w] > mx) mx = h[2 * w]; if (h[2 * w + 1] > mr) mr = h[2 * w + 1]; }
  const double n_simd = (double)iters * PER_IT[OP] * waves_per_simd;   // wave-instructions one SIMD issued
  printf("  %-44s waves/SIMD %d: %7.2f memtime-ticks / instr / SIMD   (%7.2f ns realtime per instr)\n", NAMES[OP], waves_per_simd,
         (double)mx / n_simd, (double)mr * 10.0 / n_simd);
}

int main() {
  uint64_t* d;
  hipMalloc(&d, 8192 * 8);
  hipMemset(d, 0, 8192 * 8);
  int clk = 0;
  hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("device clock rate attribute: %d kHz; wall_clock64 is 100 MHz (10 ns per tick)\n", clk);
  for (int w = 1; w <= 3; ++w) {
    printf("-- %d wave(s) per SIMD (one workgroup of %d threads on one CU)\n", w, 256 * w);
    run<OP_FMA>(d, w); run<OP_EXP>(d, w); run<OP_ADD>(d, w); run<OP_MAD24>(d, w); run<OP_CMPSEL>(d, w); run<OP_CVTPK>(d, w);
    run<OP_MAX3>(d, w); run<OP_EXP_FMA>(d, w); run<OP_MIX>(d, w); run<OP_MFMA>(d, w); run<OP_MIX_MFMA>(d, w); run<OP_MIX_NOEXP_MFMA>(d, w);
    run<OP_LDS128>(d, w);
  }
  return 0;
}
