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
// from device memory by the kernel (no host round trip, capturable in a hipGraph).  Products of e4m3 values are exact in fp32; the
// result differs from the oracle's (oracle/fp8_ref.py: same scales, same rounding, exact sums) by how the sums are formed: the
// matrix core adds the 16 products of one instruction in its own internal format, not as an IEEE fp32 chain -- measured 1.6e-5 of
// the output range at K = 512 .. 1536 (tests/test_gpu_fp8.py holds 6e-5), two orders below e4m3's own 3e-2.
//
// Kernel shape: 128 x 128 (or 64 x 128, when that is what fills the chip) outputs per workgroup, four waves as 2 x 2; the reduction
// axis advances 64 bytes per step through a double-buffered LDS stage (72-byte row pitch: conflict-free ds_read_b64 fragments),
// global -> registers -> LDS with the next step's loads in flight under the current step's 16 (8) MFMAs per wave, one barrier per
// step.  Weight gradients (a few dozen output tiles, thousands of reduction steps) split the batch over `ksplit` workgroups per
// tile; the partial slabs are summed in a fixed order by a second launch (deterministic -- no atomics).  First round of this
// kernel: correct and reasonably shaped, not yet tuned (DESIGN.md section 11 has its measured rate).
#include "common.hpp"

namespace ttts {

constexpr float FP8_MAX = 448.0f;

// NaN is carried, not clamped away (fmaxf(NaN, x) == x): a NaN element stays NaN through the clamp and the e4m3 conversion, a NaN
// anywhere in a tensor makes its amax NaN (fp8_amax_kernel), and a NaN amax makes both the quantisation scale and the GEMM's alpha
// NaN -- a diverged tensor reaches the loss as NaN, as it does in the fp32 / split-bf16 paths.
__device__ __forceinline__ float fp8_scale_of(const float* amax) {
  const float a = *amax;
  return a != a ? a : (a > 0.f ? FP8_MAX / a : 1.0f);
}
__device__ __forceinline__ float fp8_alpha_of(const float* amax_a, const float* amax_b) {
  const float a = *amax_a, b = *amax_b;
  if (a != a || b != b) return a + b;
  return ((a > 0.f ? a : FP8_MAX) / FP8_MAX) * ((b > 0.f ? b : FP8_MAX) / FP8_MAX);
}
__device__ __forceinline__ float fp8_clamp(float v) {
  const float c = fminf(fmaxf(v, -FP8_MAX), FP8_MAX);
  return v != v ? v : c;
}
__device__ __forceinline__ uint32_t fp8_pack4(float a, float b, float c, float d) {
  a = fp8_clamp(a); b = fp8_clamp(b); c = fp8_clamp(c); d = fp8_clamp(d);
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (uint32_t)w;
}

// ---- amax --------------------------------------------------------------------------------------------------------------------
// |x| as an unsigned integer orders like the float: one atomicMax per workgroup into a word the launcher cleared.
__global__ __launch_bounds__(256) void fp8_amax_kernel(const float* __restrict__ x, int64_t n, uint32_t* __restrict__ out) {
  // the maximum is taken on the BIT PATTERNS of |x| (unsigned integers order like non-negative floats, and every NaN pattern is
  // above +inf's): a NaN element wins the maximum and the result word is a NaN
  uint32_t m = 0u;
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
      m = max(max(m, __float_as_uint(fabsf(v[0]))), max(max(__float_as_uint(fabsf(v[1])), __float_as_uint(fabsf(v[2]))), __float_as_uint(fabsf(v[3]))));
    } else {
      for (int64_t j = i; j < n; ++j) m = max(m, __float_as_uint(fabsf(x[j])));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  __shared__ uint32_t part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(part[0], part[1]), max(part[2], part[3]));
    atomicMax(out, m);
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

// ---- both layouts of one (B, C, T) tensor in ONE pass: rows q_r [B][C][Tp] (reduction over time: weight gradients) and transposed
// q_t [B][T][Cp] (reduction over channels: forward / data gradient) -- the backward of a 1 x 1 convolution needs dy in both, and the
// forward's activation is needed transposed now and row-wise by the backward (which then keeps 1 byte per element instead of 4)
__global__ __launch_bounds__(256) void fp8_quant_both_kernel(const float* __restrict__ x, uint8_t* __restrict__ qr, uint8_t* __restrict__ qt,
                                                             const float* __restrict__ amax, int C, int T, int Cp, int Tp) {
  __shared__ float tile[64][65];
  const float s = fp8_scale_of(amax);
  const int b = blockIdx.z, c0 = blockIdx.y * 64, t0 = blockIdx.x * 64;
  const float* xb = x + (int64_t)b * C * T;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + 4 * i + ty, t = t0 + tx;
    tile[4 * i + ty][tx] = (c < C && t < T) ? xb[(int64_t)c * T + t] * s : 0.f;
  }
  __syncthreads();
  const int g = threadIdx.x & 15, rr = threadIdx.x >> 4;
  uint8_t* qtb = qt + (int64_t)b * T * Cp;
  uint8_t* qrb = qr + (int64_t)b * C * Tp;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 16 * i + rr;
    if (t0 + r < T && c0 + 4 * g < Cp)                                   // transposed: row t, four consecutive channels
      *reinterpret_cast<uint32_t*>(qtb + (int64_t)(t0 + r) * Cp + c0 + 4 * g) =
          fp8_pack4(tile[4 * g][r], tile[4 * g + 1][r], tile[4 * g + 2][r], tile[4 * g + 3][r]);
    if (c0 + r < C && t0 + 4 * g < Tp)                                   // rows: row c, four consecutive positions (zeros past T)
      *reinterpret_cast<uint32_t*>(qrb + (int64_t)(c0 + r) * Tp + t0 + 4 * g) =
          fp8_pack4(tile[r][4 * g], tile[r][4 * g + 1], tile[r][4 * g + 2], tile[r][4 * g + 3]);
  }
}

// ---- the GEMM ----------------------------------------------------------------------------------------------------------------
struct Fp8GemmParams {
  const uint8_t* A; const uint8_t* B;
  float* Y; const float* bias; const float* resid;
  const float* amax_a; const float* amax_b;
  int M, N, K;                      // K in bytes (= elements), a multiple of 64
  int GI;                           // inner (accumulated) groups
  int ksplit;                       // > 1: the inner groups are dealt to `ksplit` workgroups per output tile, each writing its partial
  float* slab;                      //      sums to slab[(go * ksplit + part)][M][N] (plain layout); fp8_slab_reduce_kernel finishes
  int64_t lda, ldb;                 // row pitches (bytes)
  int64_t a_so, a_si, b_so, b_si;   // outer / inner group strides (bytes)
  int64_t y_so, y_sm, y_sn;         // output strides (elements): Y[go * y_so + m * y_sm + n * y_sn]; resid shares them
  int accumulate;
};

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

__device__ __forceinline__ f32x16 mfma_fp8(u32x2 a, u32x2 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(__builtin_bit_cast(long, a), __builtin_bit_cast(long, b), c, 0, 0, 0);
}

// Workgroup tile (64 BM) x 128: four waves as 2 x 2, a wave owns BM x 2 MFMA blocks.  Per 64-byte step of the reduction axis the
// workgroup stages (64 BM + 128) rows x 64 bytes in LDS (row pitch 72 bytes: the 32 lanes of a ds_read_b64 then hit 32 distinct
// bank pairs), fetched from global memory as 32 contiguous bytes per thread and double-buffered (registers -> the other LDS
// buffer while the current one feeds the MFMAs; one barrier per step).
constexpr int F8_PITCH = 72;
template <int BM>
__global__ __launch_bounds__(256, 2) void fp8_gemm_nt_kernel(Fp8GemmParams p) {
  constexpr int TM = 64 * BM, TN = 128;
  constexpr int A_BYTES = TM * F8_PITCH, B_BYTES = TN * F8_PITCH;
  __shared__ __attribute__((aligned(16))) uint8_t lds[2][A_BYTES + B_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, hh = lane >> 5;
  int bx_, by_, bz_;
  xcd_tile(bx_, by_, bz_);            // (an XCD walks a contiguous run of tiles: the A panel of a row of tiles is fetched into ONE L2)
  const int go = bz_ / p.ksplit, part = bz_ - go * p.ksplit;
  const int mt0 = by_ * TM, nt0 = bx_ * TN;
  const int gi_per = (p.GI + p.ksplit - 1) / p.ksplit;
  const int gi0 = part * gi_per, gi1 = min(p.GI, gi0 + gi_per);
  const uint8_t* Ag = p.A + go * p.a_so;
  const uint8_t* Bg = p.B + go * p.b_so;
  // staging role: thread -> (row, 32-byte half) of the A rows (tid < 2 TM) and of the B rows (all 256 threads: 128 rows x 2)
  const int srow = tid >> 1, shalf = tid & 1;
  const bool stage_a = srow < TM;
  const uint8_t* ga = Ag + (int64_t)min(mt0 + srow, p.M - 1) * p.lda + 32 * shalf;
  const uint8_t* gb = Bg + (int64_t)min(nt0 + srow, p.N - 1) * p.ldb + 32 * shalf;
  const int kchunks = p.K / 64;
  const int64_t total = (int64_t)max(gi1 - gi0, 0) * kchunks;

  f32x16 acc[BM][2];
#pragma unroll
  for (int i = 0; i < BM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  u32x4 ra[2], rb[2];
  auto load_g = [&](int64_t it) {
    const int gi = gi0 + (int)(it / kchunks), kc = (int)(it % kchunks);
    const int64_t ao = gi * p.a_si + (int64_t)kc * 64, bo = gi * p.b_si + (int64_t)kc * 64;
    if (stage_a) { ra[0] = *reinterpret_cast<const u32x4*>(ga + ao); ra[1] = *reinterpret_cast<const u32x4*>(ga + ao + 16); }
    rb[0] = *reinterpret_cast<const u32x4*>(gb + bo); rb[1] = *reinterpret_cast<const u32x4*>(gb + bo + 16);
  };
  auto store_l = [&](int buf) {
    uint8_t* la = lds[buf] + srow * F8_PITCH + 32 * shalf;
    uint8_t* lb = lds[buf] + A_BYTES + srow * F8_PITCH + 32 * shalf;
    if (stage_a) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        *reinterpret_cast<u32x2*>(la + 16 * q) = u32x2{ra[q][0], ra[q][1]};
        *reinterpret_cast<u32x2*>(la + 16 * q + 8) = u32x2{ra[q][2], ra[q][3]};
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      *reinterpret_cast<u32x2*>(lb + 16 * q) = u32x2{rb[q][0], rb[q][1]};
      *reinterpret_cast<u32x2*>(lb + 16 * q + 8) = u32x2{rb[q][2], rb[q][3]};
    }
  };
  auto compute = [&](int buf) {
    const uint8_t* la = lds[buf] + (wm * 32 * BM + r) * F8_PITCH + 8 * hh;
    const uint8_t* lb = lds[buf] + A_BYTES + (wn * 64 + r) * F8_PITCH + 8 * hh;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      u32x2 fa[BM], fb[2];
#pragma unroll
      for (int i = 0; i < BM; ++i) fa[i] = *reinterpret_cast<const u32x2*>(la + i * 32 * F8_PITCH + 16 * s4);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const u32x2*>(lb + j * 32 * F8_PITCH + 16 * s4);
#pragma unroll
      for (int i = 0; i < BM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma_fp8(fa[i], fb[j], acc[i][j]);
    }
  };
  if (total > 0) {
    load_g(0);
    store_l(0);
    __syncthreads();
    for (int64_t it = 0; it < total; ++it) {
      const int buf = (int)(it & 1);
      if (it + 1 < total) load_g(it + 1);
      compute(buf);
      if (it + 1 < total) store_l(buf ^ 1);
      __syncthreads();
    }
  }

  const int m0 = mt0 + wm * 32 * BM, n0 = nt0 + wn * 64;
  if (p.ksplit > 1) {                 // partial sums, unscaled, plain [M][N] layout
    float* S = p.slab + (int64_t)bz_ * p.M * p.N;
#pragma unroll
    for (int bm = 0; bm < BM; ++bm)
#pragma unroll
      for (int bn = 0; bn < 2; ++bn) {
        const int n = n0 + 32 * bn + r;
        if (n >= p.N) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + 32 * bm + (e & 3) + 8 * (e >> 2) + 4 * hh;
          if (m < p.M) S[(int64_t)m * p.N + n] = acc[bm][bn][e];
        }
      }
    return;
  }
  const float alpha = fp8_alpha_of(p.amax_a, p.amax_b);
  float* Y = p.Y + go * p.y_so;
  const float* R = p.resid ? p.resid + go * p.y_so : nullptr;
#pragma unroll
  for (int bm = 0; bm < BM; ++bm)
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

// sum of the `ksplit` partial slabs of every outer group, in slab order (deterministic), scaled and stored like the direct epilogue
__global__ __launch_bounds__(256) void fp8_slab_reduce_kernel(Fp8GemmParams p, int groups_outer) {
  const int64_t per = (int64_t)p.M * p.N, total = per * groups_outer;
  const float alpha = fp8_alpha_of(p.amax_a, p.amax_b);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int go = (int)(i / per);
    const int64_t e = i - go * per;
    const int m = (int)(e / p.N), n = (int)(e - (int64_t)m * p.N);
    float s = 0.f;
    for (int k = 0; k < p.ksplit; ++k) s += p.slab[((int64_t)go * p.ksplit + k) * per + e];
    const int64_t o = go * p.y_so + (int64_t)m * p.y_sm + (int64_t)n * p.y_sn;
    float v = alpha * s;
    if (p.bias) v += p.bias[m];
    if (p.resid) v += p.resid[o];
    p.Y[o] = p.accumulate ? p.Y[o] + v : v;
  }
}

}  // namespace ttts

using namespace ttts;

extern "C" int ttts_fp8_amax_f32(const float* x, int64_t n, float* amax_out, int32_t out_is_zero, void* stream) {
  TTTS_REQUIRE(x && amax_out && n > 0, "fp8_amax: null pointer / empty tensor");
  TTTS_REQUIRE(aligned16(x), "fp8_amax: 16-byte alignment required");
  hipStream_t s = as_stream(stream);
  // out_is_zero: the caller hands a word it knows to be zero (a fresh slot of a pre-cleared pool): no memset launch
  if (!out_is_zero && hipMemsetAsync(amax_out, 0, sizeof(float), s) != hipSuccess) return fail(TTTS_EHIP, "fp8_amax: memset failed");
  // at most 256 workgroups: each ends in one atomicMax on the same word (2048 of them took longer than the 14 MB read)
  const int grid = (int)std::min<int64_t>(cdiv(n, 256 * 4 * 4), 256);
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

extern "C" int ttts_fp8_quant_both_f32(const float* x, void* q_rows, void* q_t, const float* amax, int32_t B, int32_t C, int32_t T,
                                       int32_t Cp, int32_t Tp, void* stream) {
  TTTS_REQUIRE(x && q_rows && q_t && amax && B > 0 && C > 0 && T > 0, "fp8_quant_both: null pointer / empty tensor");
  TTTS_REQUIRE(Cp >= C && Cp % 64 == 0 && Tp >= T && Tp % 64 == 0, "fp8_quant_both: Cp / Tp must be multiples of 64 covering C / T");
  TTTS_REQUIRE((reinterpret_cast<uintptr_t>(q_rows) & 3u) == 0 && (reinterpret_cast<uintptr_t>(q_t) & 3u) == 0, "fp8_quant_both: 4-byte alignment required");
  const dim3 grid((unsigned)(Tp / 64), (unsigned)(Cp / 64), (unsigned)B);
  fp8_quant_both_kernel<<<grid, 256, 0, as_stream(stream)>>>(x, static_cast<uint8_t*>(q_rows), static_cast<uint8_t*>(q_t), amax, C, T, Cp, Tp);
  return check_launch("fp8_quant_both");
}

extern "C" int64_t ttts_fp8_gemm_nt_workspace_bytes(int32_t M, int32_t N, int32_t groups_outer, int32_t groups_inner) {
  // split over the inner groups only when the output alone cannot fill the chip (weight gradients: a few dozen tiles, thousands of
  // reduction steps); the launcher applies the same rule
  const int64_t tiles = cdiv(M, 64) * cdiv(N, 128) * groups_outer;
  if (groups_inner < 2 || tiles >= 256) return 0;
  const int ks = (int)std::min<int64_t>(groups_inner, std::max<int64_t>(1, 512 / tiles));
  return ks > 1 ? (int64_t)groups_outer * ks * M * N * (int64_t)sizeof(float) : 0;
}

extern "C" int ttts_fp8_gemm_nt(const void* a, const void* b, float* y, const float* bias, const float* resid, const float* amax_a,
                                const float* amax_b, int32_t M, int32_t N, int32_t K, int32_t groups_outer, int32_t groups_inner,
                                int64_t lda, int64_t ldb, int64_t a_stride_outer, int64_t a_stride_inner, int64_t b_stride_outer,
                                int64_t b_stride_inner, int64_t y_stride_outer, int64_t y_stride_m, int64_t y_stride_n,
                                int32_t accumulate, void* workspace, void* stream) {
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
  p.ksplit = 1; p.slab = nullptr;
  hipStream_t s = as_stream(stream);
  const int64_t ws = ttts_fp8_gemm_nt_workspace_bytes(M, N, groups_outer, groups_inner);
  if (ws > 0 && workspace) {
    p.ksplit = (int)(ws / ((int64_t)groups_outer * M * N * (int64_t)sizeof(float)));
    p.slab = static_cast<float*>(workspace);
  }
  // 128 x 128 tiles when they fill the chip twice over, else 64 x 128 (twice the workgroups)
  const bool big = cdiv(M, 128) * cdiv(N, 128) * groups_outer * p.ksplit >= 512;
  const dim3 grid((unsigned)cdiv(N, 128), (unsigned)cdiv(M, big ? 128 : 64), (unsigned)(groups_outer * p.ksplit));
  if (big) fp8_gemm_nt_kernel<2><<<grid, 256, 0, s>>>(p);
  else fp8_gemm_nt_kernel<1><<<grid, 256, 0, s>>>(p);
  int rc = check_launch("fp8_gemm_nt");
  if (rc || p.ksplit == 1) return rc;
  const int64_t total = (int64_t)groups_outer * M * N;
  fp8_slab_reduce_kernel<<<(int)std::min<int64_t>(cdiv(total, 256), 4096), 256, 0, s>>>(p, groups_outer);
  return check_launch("fp8_slab_reduce");
}
