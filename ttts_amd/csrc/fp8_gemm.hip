// FP8 (OCP e4m3) matrix-core GEMMs for the 1 x 1 convolutions / linear layers of the diffusion mel-denoiser step
// (BASELINE config #5: "bf16 + fp8 MFMA GEMMs").  Replaces, in the fp8 precision mode of ttts_amd.diffusion, the
// nn.Conv1d(k = 1) / nn.Linear calls of AttentionBlock.qkv / .proj_out (ttts/utils/utils.py:172-215), ResBlock.in_layers[2]
// (ttts/diffusion/aa_model.py:70-131, efficient_config) and AA_diffusion.integrating_conv (:228) -- forward, data gradient and
// weight gradient, all three as ONE "NT" kernel over operands whose reduction axis is contiguous:
//
//     Y[go][m][n] (+)= alpha * sum_{gi} sum_k A[go][gi][m][k] * B[go][gi][n][k]  (+ bias[m]) (+ resid[go][m][n])
//
//   forward        y[b] = W x[b]          A = Wq [Cout][Cin],     B = xq^T  [b][T][Cin]      (go = b)
//   data gradient  dx[b] = W^T dy[b]      A = Wq^T [Cin][Cout],   B = dyq^T [b][T][Cout]     (go = b)
//   weight grad.   dW += sum_b dy[b] x[b]^T   A = dyq [b][Cout][Tp], B = xq [b][Cin][Tp]     (gi = b, K = Tp)
//
// Arithmetic: every operand tensor is scaled by 448 / amax (per-TENSOR "current scaling": the amax of the very tensor being
// quantised, measured by a pass in front of the quantisation), rounded to e4m3 (v_cvt_pk_fp8_f32, round-to-nearest-even, values
// clamped to +-448 first), multiplied on v_mfma_f32_32x32x16_fp8_fp8 with fp32 accumulation; alpha = amax_a amax_b / 448^2 is read
// from device memory by the kernel (no host round trip, capturable in a hipGraph).  Products of e4m3 values are exact in fp32, so
// the result differs from the oracle's (oracle/fp8_ref.py: same scales, same rounding, fp32 matmul) only by summation order.
//
// Kernel shape: 128 x 128 outputs per workgroup, four waves as 2 x 2, each 64 x 64 = 2 x 2 MFMA blocks (64 accumulator registers);
// the reduction axis advances 64 bytes per iteration: a lane (row r = lane & 31, half hh = lane >> 5) fetches bytes
// [32 j + 16 hh, + 16) of its row for j = 0, 1 with two 16-byte loads and feeds the four k-sub-steps from their 8-byte halves --
// A and B use the same byte -> sub-step map, which is all the contraction needs.  Fragments come straight from global memory
// (the operands are 1 byte per element and K-contiguous: a 128 x 128 tile re-reads 16 KB per iteration out of L2, no LDS staging),
// double-buffered in registers.  This is a first, correct, reasonably shaped kernel -- not yet a tuned one.
#include "common.hpp"

namespace ttts {

constexpr float FP8_MAX = 448.0f;

__device__ __forceinline__ float fp8_scale_of(const float* amax) {
  const float a = *amax;
  return a > 0.f ? FP8_MAX / a : 1.0f;
}
__device__ __forceinline__ uint32_t fp8_pack4(float a, float b, float c, float d) {
  a = fminf(fmaxf(a, -FP8_MAX), FP8_MAX); b = fminf(fmaxf(b, -FP8_MAX), FP8_MAX);
  c = fminf(fmaxf(c, -FP8_MAX), FP8_MAX); d = fminf(fmaxf(d, -FP8_MAX), FP8_MAX);
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (uint32_t)w;
}

// ---- amax --------------------------------------------------------------------------------------------------------------------
// |x| as an unsigned integer orders like the float: one atomicMax per workgroup into a word the launcher cleared.
__global__ __launch_bounds__(256) void fp8_amax_kernel(const float* __restrict__ x, int64_t n, uint32_t* __restrict__ out) {
  float m = 0.f;
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
      m = fmaxf(fmaxf(m, fabsf(v[0])), fmaxf(fmaxf(fabsf(v[1]), fabsf(v[2])), fabsf(v[3])));
    } else {
      for (int64_t j = i; j < n; ++j) m = fmaxf(m, fabsf(x[j]));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    atomicMax(out, __float_as_uint(m));
  }
}

// ---- quantisation, reduction axis already contiguous: q[r][c] = e4m3(x[r][c] * scale), zero for cols <= c < cols_pad ------------
__global__ __launch_bounds__(256) void fp8_quant_rows_kernel(const float* __restrict__ x, uint8_t* __restrict__ q,
                                                             const float* __restrict__ amax, int64_t rows, int cols, int cols_pad) {
  const float s = fp8_scale_of(amax);
  const int groups = cols_pad / 4;                        // one thread: four consecutive outputs of a row
  const int64_t total = rows * groups;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / groups;
    const int c = (int)(i - r * groups) * 4;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = c + e < cols ? x[r * cols + c + e] * s : 0.f;
    *reinterpret_cast<uint32_t*>(q + r * cols_pad + c) = fp8_pack4(v[0], v[1], v[2], v[3]);
  }
}

// ---- quantisation with transpose: x [B][C][T] -> q [B][T][Cp] (Cp >= C, zero padded): 64 x 64 tiles through LDS ----------------
__global__ __launch_bounds__(256) void fp8_quant_transpose_kernel(const float* __restrict__ x, uint8_t* __restrict__ q,
                                                                  const float* __restrict__ amax, int C, int T, int Cp) {
  __shared__ float tile[64][65];
  const float s = fp8_scale_of(amax);
  const int b = blockIdx.z, c0 = blockIdx.y * 64, t0 = blockIdx.x * 64;
  const float* xb = x + (int64_t)b * C * T;
  uint8_t* qb = q + (int64_t)b * T * Cp;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {                          // rows c0 + 4 i + ty, column t0 + tx: coalesced along t
    const int c = c0 + 4 * i + ty, t = t0 + tx;
    tile[4 * i + ty][tx] = (c < C && t < T) ? xb[(int64_t)c * T + t] * s : 0.f;
  }
  __syncthreads();
  const int cg = threadIdx.x & 15, tr = threadIdx.x >> 4;  // a thread: four consecutive channels of one output row
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = 16 * i + tr;
    if (t0 + t < T && c0 + 4 * cg < Cp)
      *reinterpret_cast<uint32_t*>(qb + (int64_t)(t0 + t) * Cp + c0 + 4 * cg) =
          fp8_pack4(tile[4 * cg][t], tile[4 * cg + 1][t], tile[4 * cg + 2][t], tile[4 * cg + 3][t]);
  }
}

// ---- the GEMM ----------------------------------------------------------------------------------------------------------------
struct Fp8GemmParams {
  const uint8_t* A; const uint8_t* B;
  float* Y; const float* bias; const float* resid;
  const float* amax_a; const float* amax_b;
  int M, N, K;                      // K in bytes (= elements), a multiple of 64
  int GI;                           // inner (accumulated) groups
  int64_t lda, ldb;                 // row pitches (bytes)
  int64_t a_so, a_si, b_so, b_si;   // outer / inner group strides (bytes)
  int64_t y_so, y_sm, y_sn;         // output strides (elements): Y[go * y_so + m * y_sm + n * y_sn]; resid shares them
  int accumulate;
};

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ f32x16 mfma_fp8(uint32_t a_lo, uint32_t a_hi, uint32_t b_lo, uint32_t b_hi, f32x16 c) {
  const long a = (long)(((uint64_t)a_hi << 32) | a_lo), b = (long)(((uint64_t)b_hi << 32) | b_lo);
  return __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a, b, c, 0, 0, 0);
}

__global__ __launch_bounds__(256, 2) void fp8_gemm_nt_kernel(Fp8GemmParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, hh = lane >> 5;
  const int go = blockIdx.z;
  const int m0 = blockIdx.y * 128 + wm * 64, n0 = blockIdx.x * 128 + wn * 64;
  const uint8_t* Ag = p.A + go * p.a_so;
  const uint8_t* Bg = p.B + go * p.b_so;
  // per-lane row pointers (rows past the edge repeat the last valid one: their products are never stored)
  const uint8_t* arow[2]; const uint8_t* brow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    arow[i] = Ag + (int64_t)min(m0 + 32 * i + r, p.M - 1) * p.lda + 16 * hh;
    brow[i] = Bg + (int64_t)min(n0 + 32 * i + r, p.N - 1) * p.ldb + 16 * hh;
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int kchunks = p.K / 64;
  const int64_t total = (int64_t)p.GI * kchunks;
  u32x4 fa[2][2][2], fb[2][2][2];                  // [buffer][block][j]
  auto load = [&](int buf, int64_t it) {
    const int gi = (int)(it / kchunks), kc = (int)(it - (int64_t)gi * kchunks);
    const int64_t ao = gi * p.a_si + (int64_t)kc * 64, bo = gi * p.b_si + (int64_t)kc * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        fa[buf][i][j] = *reinterpret_cast<const u32x4*>(arow[i] + ao + 32 * j);
        fb[buf][i][j] = *reinterpret_cast<const u32x4*>(brow[i] + bo + 32 * j);
      }
  };
  auto compute = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int bm = 0; bm < 2; ++bm)
#pragma unroll
          for (int bn = 0; bn < 2; ++bn)
            acc[bm][bn] = mfma_fp8(fa[buf][bm][j][2 * h2], fa[buf][bm][j][2 * h2 + 1], fb[buf][bn][j][2 * h2], fb[buf][bn][j][2 * h2 + 1],
                                   acc[bm][bn]);
  };
  load(0, 0);
  for (int64_t it = 0; it < total; it += 2) {
    if (it + 1 < total) load(1, it + 1);
    compute(0);
    if (it + 1 < total) {
      if (it + 2 < total) load(0, it + 2);
      compute(1);
    }
  }

  const float alpha = ((*p.amax_a > 0.f ? *p.amax_a : FP8_MAX) / FP8_MAX) * ((*p.amax_b > 0.f ? *p.amax_b : FP8_MAX) / FP8_MAX);
  float* Y = p.Y + go * p.y_so;
  const float* R = p.resid ? p.resid + go * p.y_so : nullptr;
#pragma unroll
  for (int bm = 0; bm < 2; ++bm)
#pragma unroll
    for (int bn = 0; bn < 2; ++bn) {
      const int n = n0 + 32 * bn + r;
      if (n >= p.N) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + 32 * bm + (e & 3) + 8 * (e >> 2) + 4 * hh;
        if (m >= p.M) continue;
        const int64_t o = (int64_t)m * p.y_sm + (int64_t)n * p.y_sn;
        float v = alpha * acc[bm][bn][e];
        if (p.bias) v += p.bias[m];
        if (R) v += R[o];
        Y[o] = p.accumulate ? Y[o] + v : v;
      }
    }
}

}  // namespace ttts

using namespace ttts;

extern "C" int ttts_fp8_amax_f32(const float* x, int64_t n, float* amax_out, void* stream) {
  TTTS_REQUIRE(x && amax_out && n > 0, "fp8_amax: null pointer / empty tensor");
  TTTS_REQUIRE(aligned16(x), "fp8_amax: 16-byte alignment required");
  hipStream_t s = as_stream(stream);
  if (hipMemsetAsync(amax_out, 0, sizeof(float), s) != hipSuccess) return fail(TTTS_EHIP, "fp8_amax: memset failed");
  const int grid = (int)std::min<int64_t>(cdiv(n, 256 * 4 * 4), 2048);
  fp8_amax_kernel<<<grid, 256, 0, s>>>(x, n, reinterpret_cast<uint32_t*>(amax_out));
  return check_launch("fp8_amax");
}

extern "C" int ttts_fp8_quant_f32(const float* x, void* q, const float* amax, int64_t rows, int32_t cols, int32_t cols_pad, void* stream) {
  TTTS_REQUIRE(x && q && amax && rows > 0 && cols > 0, "fp8_quant: null pointer / empty tensor");
  TTTS_REQUIRE(cols_pad >= cols && cols_pad % 4 == 0 && (reinterpret_cast<uintptr_t>(q) & 3u) == 0, "fp8_quant: cols_pad must be a multiple of 4 >= cols");
  const int64_t total = rows * (cols_pad / 4);
  fp8_quant_rows_kernel<<<(int)std::min<int64_t>(cdiv(total, 256), 8192), 256, 0, as_stream(stream)>>>(x, static_cast<uint8_t*>(q), amax, rows, cols, cols_pad);
  return check_launch("fp8_quant");
}

extern "C" int ttts_fp8_quant_transpose_f32(const float* x, void* q, const float* amax, int32_t B, int32_t C, int32_t T, int32_t Cp, void* stream) {
  TTTS_REQUIRE(x && q && amax && B > 0 && C > 0 && T > 0, "fp8_quant_transpose: null pointer / empty tensor");
  TTTS_REQUIRE(Cp >= C && Cp % 4 == 0 && (reinterpret_cast<uintptr_t>(q) & 3u) == 0, "fp8_quant_transpose: Cp must be a multiple of 4 >= C");
  const dim3 grid((unsigned)cdiv(T, 64), (unsigned)cdiv(Cp, 64), (unsigned)B);
  fp8_quant_transpose_kernel<<<grid, 256, 0, as_stream(stream)>>>(x, static_cast<uint8_t*>(q), amax, C, T, Cp);
  return check_launch("fp8_quant_transpose");
}

extern "C" int ttts_fp8_gemm_nt(const void* a, const void* b, float* y, const float* bias, const float* resid, const float* amax_a,
                                const float* amax_b, int32_t M, int32_t N, int32_t K, int32_t groups_outer, int32_t groups_inner,
                                int64_t lda, int64_t ldb, int64_t a_stride_outer, int64_t a_stride_inner, int64_t b_stride_outer,
                                int64_t b_stride_inner, int64_t y_stride_outer, int64_t y_stride_m, int64_t y_stride_n,
                                int32_t accumulate, void* stream) {
  TTTS_REQUIRE(a && b && y && amax_a && amax_b, "fp8_gemm_nt: null pointer");
  TTTS_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0 && groups_outer > 0 && groups_inner > 0, "fp8_gemm_nt: K must be a positive multiple of 64 (K=%d)", K);
  TTTS_REQUIRE(aligned16(a) && aligned16(b) && lda % 16 == 0 && ldb % 16 == 0 && a_stride_outer % 16 == 0 && a_stride_inner % 16 == 0 &&
               b_stride_outer % 16 == 0 && b_stride_inner % 16 == 0, "fp8_gemm_nt: operands, pitches and group strides must be 16-byte aligned");
  TTTS_REQUIRE(lda >= K && ldb >= K, "fp8_gemm_nt: row pitch smaller than K");
  Fp8GemmParams p;
  p.A = static_cast<const uint8_t*>(a); p.B = static_cast<const uint8_t*>(b); p.Y = y; p.bias = bias; p.resid = resid;
  p.amax_a = amax_a; p.amax_b = amax_b; p.M = M; p.N = N; p.K = K; p.GI = groups_inner; p.lda = lda; p.ldb = ldb;
  p.a_so = a_stride_outer; p.a_si = a_stride_inner; p.b_so = b_stride_outer; p.b_si = b_stride_inner;
  p.y_so = y_stride_outer; p.y_sm = y_stride_m; p.y_sn = y_stride_n; p.accumulate = accumulate;
  const dim3 grid((unsigned)cdiv(N, 128), (unsigned)cdiv(M, 128), (unsigned)groups_outer);
  fp8_gemm_nt_kernel<<<grid, 256, 0, as_stream(stream)>>>(p);
  return check_launch("fp8_gemm_nt");
}
