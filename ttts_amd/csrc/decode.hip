// Autoregressive decoding of the GPT (ttts/gpt/model.py:34-184 GPT2InferenceModel, :533-562
// inference_speech, HF GenerationMixin._sample with its logits processors).
// One decode step = one query token per sequence: the GEMMs are the training path's NT kernel on M = #sequences rows (weight
// streaming, HBM-bound: 2 bytes per parameter per step), and the pieces that only exist at decode time live here:
//   decode_embed   : x[m] = mel_embedding[token[m]] + mel_pos[cur_len - Tt]
//   kv_cache_fill  : prompt K / V of the prefill pass -> cache [M][H][S_max][dh], replicated for num_return_sequences
//   attn_decode    : append this step's K / V, softmax(q . K^T) V over the cur_len + 1 cached keys, one workgroup per
//                    (sequence, head), keys spread over lane groups, online softmax merged through LDS
//   sample_logits  : repetition penalty -> typical -> temperature -> top-k -> top-p -> softmax -> inverse-CDF draw (or argmax),
//                    eos / pad bookkeeping; one workgroup per sequence, one bitonic sort in LDS
//   decode_advance : cur_len += 1, step += 1
// Every position / length is read from a device-side counter block, so ONE captured hipGraph replays every step and the
// host never synchronises inside the token loop.
#include <math.h>

#include <algorithm>

#include "common.hpp"

namespace ttts {

// counter block (int32, device memory owned by the caller)
enum { CTR_LEN = 0, CTR_STEP = 1, CTR_UNFINISHED = 2, CTR_WORDS = 4 };

__global__ __launch_bounds__(256) void decode_embed_kernel(const int64_t* __restrict__ tokens, const float* __restrict__ emb,
                                                           const float* __restrict__ pos, const int32_t* __restrict__ ctr,
                                                           int pos_offset, float* __restrict__ x, int D, int V, int P) {
  const int m = blockIdx.x;
  int64_t tok = tokens[m];
  tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
  int pp = ctr[CTR_LEN] + pos_offset;
  pp = pp < 0 ? 0 : (pp >= P ? P - 1 : pp);
  for (int d = threadIdx.x; d < D; d += 256) x[(int64_t)m * D + d] = emb[tok * D + d] + pos[(int64_t)pp * D + d];
}

// qkv bf16 [B*S, 3D] (row b*S + s) -> caches [B*rep][H][S_max][dh]
__global__ __launch_bounds__(256) void kv_cache_fill_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ kc,
                                                            bf16* __restrict__ vc, int B, int S, int H, int dh, int S_max,
                                                            int rep) {
  const int D = H * dh;
  const int chunks = dh / 8;
  const int64_t total = (int64_t)B * rep * H * S * chunks;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % chunks);
    const int s = (int)((i / chunks) % S);
    const int h = (int)((i / chunks / S) % H);
    const int m = (int)(i / chunks / S / H);
    const int b = m / rep;
    const bf16* src = qkv + ((int64_t)b * S + s) * 3 * D + h * dh + c * 8;
    const int64_t dst = (((int64_t)m * H + h) * S_max + s) * dh + c * 8;
    *reinterpret_cast<bf16x8*>(kc + dst) = *reinterpret_cast<const bf16x8*>(src + D);
    *reinterpret_cast<bf16x8*>(vc + dst) = *reinterpret_cast<const bf16x8*>(src + 2 * D);
  }
}

// one workgroup (16 waves) per (head, sequence).  A key is handled by dh/8 adjacent lanes (16-byte slices of the row), so a
// wave covers 64 / (dh/8) keys per pass.  Every lane group keeps an online-softmax state (m, l, acc[8]); states are merged
// across the groups of a wave by shuffles and across waves through LDS.
constexpr int AW = 16;   // waves per (sequence, head): 16 x (64 / (dh/8)) keys in flight per pass
template <int DH>
__global__ __launch_bounds__(AW * 64) void attn_decode_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ kc,
                                                          bf16* __restrict__ vc, const int32_t* __restrict__ ctr,
                                                          bf16* __restrict__ out, int H, int S_max, float scale_log2) {
  constexpr int LPK = DH / 8;          // lanes per key
  constexpr int KPW = 64 / LPK;        // keys per wave and pass
  __shared__ float sm_acc[AW][DH];
  __shared__ float sm_ml[AW][2];
  const int h = blockIdx.x, m = blockIdx.y;
  const int D = H * DH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane % LPK, grp = lane / LPK;
  const int t = min(ctr[CTR_LEN], S_max - 1);      // index of the new token = number of cached keys
  const bf16* row = qkv + (int64_t)m * 3 * D + h * DH;
  bf16* kbase = kc + ((int64_t)m * H + h) * S_max * DH;
  bf16* vbase = vc + ((int64_t)m * H + h) * S_max * DH;
  float q[8];
  {
    const bf16x8 qv = *reinterpret_cast<const bf16x8*>(row + sub * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) q[e] = (float)qv[e];
  }
  if (tid < LPK) {   // append this step's key / value (read back below from qkv, not from the cache)
    *reinterpret_cast<bf16x8*>(kbase + (int64_t)t * DH + tid * 8) = *reinterpret_cast<const bf16x8*>(row + D + tid * 8);
    *reinterpret_cast<bf16x8*>(vbase + (int64_t)t * DH + tid * 8) = *reinterpret_cast<const bf16x8*>(row + 2 * D + tid * 8);
  }
  float mx = -3.0e38f, l = 0.f, acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int k0 = wave * KPW; k0 <= t; k0 += AW * KPW) {
    const int key = k0 + grp;
    if (key <= t) {   // lanes of one group agree
      const bf16* kp = key == t ? row + D : kbase + (int64_t)key * DH;
      const bf16* vp = key == t ? row + 2 * D : vbase + (int64_t)key * DH;
      const bf16x8 kv = *reinterpret_cast<const bf16x8*>(kp + sub * 8);
      const bf16x8 vv = *reinterpret_cast<const bf16x8*>(vp + sub * 8);
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s = fmaf(q[e], (float)kv[e], s);
#pragma unroll
      for (int o = 1; o < LPK; o <<= 1) s += __shfl_xor(s, o, 64);
      s *= scale_log2;
      const float mn = fmaxf(mx, s);
      const float a = __builtin_amdgcn_exp2f(mx - mn), p = __builtin_amdgcn_exp2f(s - mn);
      l = l * a + p;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = acc[e] * a + p * (float)vv[e];
      mx = mn;
    }
  }
  // merge the lane groups of the wave (same sub, different grp): xor over the group index bits
#pragma unroll
  for (int o = LPK; o < 64; o <<= 1) {
    const float m2 = __shfl_xor(mx, o, 64), l2 = __shfl_xor(l, o, 64);
    const float mn = fmaxf(mx, m2);
    const float a = __builtin_amdgcn_exp2f(mx - mn), b2 = __builtin_amdgcn_exp2f(m2 - mn);
    l = l * a + l2 * b2;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = acc[e] * a + __shfl_xor(acc[e], o, 64) * b2;
    mx = mn;
  }
  if (grp == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) sm_acc[wave][sub * 8 + e] = acc[e];
    if (sub == 0) { sm_ml[wave][0] = mx; sm_ml[wave][1] = l; }
  }
  __syncthreads();
  if (tid < DH) {
    float M = sm_ml[0][0];
#pragma unroll
    for (int w = 1; w < AW; ++w) M = fmaxf(M, sm_ml[w][0]);
    float L = 0.f, o = 0.f;
#pragma unroll
    for (int w = 0; w < AW; ++w) {
      const float a = __builtin_amdgcn_exp2f(sm_ml[w][0] - M);
      L += sm_ml[w][1] * a;
      o += sm_acc[w][tid] * a;
    }
    out[(int64_t)m * D + h * DH + tid] = (bf16)(o / L);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// skinny linear layer for M <= 16 token rows: out[M, N] = epi(LN2(LN1(x))[M, K] . W[N, K]^T + bias).
// The training path's 256 x 128-tile GEMM needs ~9-16 us for these shapes (2-4 workgroups stream the whole weight matrix);
// here one workgroup owns 16 output columns: the (optionally layer-normed) rows are staged once in LDS as bf16, the 16
// waves split K in 32-wide steps of v_mfma_f32_16x16x32_bf16 (A = 16 weight rows x 32 k straight from HBM, B = the staged
// rows), partial tiles are summed through LDS and the epilogue rounds exactly like the GEMM kernel's
// (bf16(acc + bias), then GELU / fp32 residual add / store).
typedef float f32x4v __attribute__((ext_vector_type(4)));

struct LinearDecodeParams {
  const void* x; int64_t ldx; int x_is_f32;
  const float *g1, *b1, *g2, *b2;       // optional LayerNorms applied to x (f32 input only); LN1 output stays fp32 if LN2 follows
  const bf16* W; int64_t ldw;
  const float* bias;
  void* out; int64_t ldc;
  const float* resid;
  int M, N, K, epi;
  float eps;
};

constexpr int LW = 16;   // waves per workgroup: every wave has at most K / 512 weight loads, all in flight at once
__global__ __launch_bounds__(LW * 64) void linear_decode_kernel(LinearDecodeParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ld_smem[];
  const int KP = p.K + 8;                                 // row pitch in bf16 (16-byte skew against bank conflicts)
  bf16* xs = reinterpret_cast<bf16*>(ld_smem);            // [16][KP]
  float* red = reinterpret_cast<float*>(ld_smem + (size_t)16 * KP * sizeof(bf16));   // [LW][16][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * 16;
  // ---- stage the rows: one wave per row
  {
    const int r = tid >> 6, c = tid & 63;
    if (!p.x_is_f32) {
      const bf16* xr = static_cast<const bf16*>(p.x) + (int64_t)r * p.ldx;
      for (int k = c * 8; k < p.K; k += 512)
        *reinterpret_cast<bf16x8*>(xs + r * KP + k) = r < p.M ? *reinterpret_cast<const bf16x8*>(xr + k) : zero8();
    } else {
      const float* xr = static_cast<const float*>(p.x) + (int64_t)r * p.ldx;
      float mean = 0.f, rstd = 1.f, mean2 = 0.f, rstd2 = 1.f;
      if (p.g1) {   // two-pass statistics, as ln_fwd_kernel
        float s = 0.f;
        for (int k = c * 4; k < p.K; k += 256) {
          const float4 v = r < p.M ? *reinterpret_cast<const float4*>(xr + k) : make_float4(0.f, 0.f, 0.f, 0.f);
          s += (v.x + v.y) + (v.z + v.w);
        }
        for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o, 64);
        mean = s / (float)p.K;
        float q = 0.f;
        for (int k = c * 4; k < p.K; k += 256) {
          const float4 v = r < p.M ? *reinterpret_cast<const float4*>(xr + k) : make_float4(mean, mean, mean, mean);
          const float a = v.x - mean, b = v.y - mean, cc = v.z - mean, e = v.w - mean;
          q += (a * a + b * b) + (cc * cc + e * e);
        }
        for (int o = 1; o < 64; o <<= 1) q += __shfl_xor(q, o, 64);
        rstd = 1.0f / sqrtf(q / (float)p.K + p.eps);
        if (p.g2) {   // statistics of y1 = LN1(x) for the second norm
          float s2 = 0.f;
          for (int k = c * 4; k < p.K; k += 256) {
            const float4 v = r < p.M ? *reinterpret_cast<const float4*>(xr + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 g = *reinterpret_cast<const float4*>(p.g1 + k), b = *reinterpret_cast<const float4*>(p.b1 + k);
            s2 += (((v.x - mean) * rstd * g.x + b.x) + ((v.y - mean) * rstd * g.y + b.y)) +
                  (((v.z - mean) * rstd * g.z + b.z) + ((v.w - mean) * rstd * g.w + b.w));
          }
          for (int o = 1; o < 64; o <<= 1) s2 += __shfl_xor(s2, o, 64);
          mean2 = s2 / (float)p.K;
          float q2 = 0.f;
          for (int k = c * 4; k < p.K; k += 256) {
            const float4 v = r < p.M ? *reinterpret_cast<const float4*>(xr + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 g = *reinterpret_cast<const float4*>(p.g1 + k), b = *reinterpret_cast<const float4*>(p.b1 + k);
            const float a = (v.x - mean) * rstd * g.x + b.x - mean2, bb = (v.y - mean) * rstd * g.y + b.y - mean2;
            const float cc = (v.z - mean) * rstd * g.z + b.z - mean2, e = (v.w - mean) * rstd * g.w + b.w - mean2;
            q2 += (a * a + bb * bb) + (cc * cc + e * e);
          }
          for (int o = 1; o < 64; o <<= 1) q2 += __shfl_xor(q2, o, 64);
          rstd2 = 1.0f / sqrtf(q2 / (float)p.K + p.eps);
        }
      }
      for (int k = c * 4; k < p.K; k += 256) {
        float4 v = r < p.M ? *reinterpret_cast<const float4*>(xr + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.g1) {
          const float4 g = *reinterpret_cast<const float4*>(p.g1 + k), b = *reinterpret_cast<const float4*>(p.b1 + k);
          v.x = (v.x - mean) * rstd * g.x + b.x; v.y = (v.y - mean) * rstd * g.y + b.y;
          v.z = (v.z - mean) * rstd * g.z + b.z; v.w = (v.w - mean) * rstd * g.w + b.w;
          if (p.g2) {
            const float4 g2 = *reinterpret_cast<const float4*>(p.g2 + k), b2 = *reinterpret_cast<const float4*>(p.b2 + k);
            v.x = (v.x - mean2) * rstd2 * g2.x + b2.x; v.y = (v.y - mean2) * rstd2 * g2.y + b2.y;
            v.z = (v.z - mean2) * rstd2 * g2.z + b2.z; v.w = (v.w - mean2) * rstd2 * g2.w + b2.w;
          }
        }
        bf16x4 o;
        o[0] = (bf16)(r < p.M ? v.x : 0.f); o[1] = (bf16)(r < p.M ? v.y : 0.f);
        o[2] = (bf16)(r < p.M ? v.z : 0.f); o[3] = (bf16)(r < p.M ? v.w : 0.f);
        *reinterpret_cast<bf16x4*>(xs + r * KP + k) = o;
      }
    }
  }
  __syncthreads();
  // ---- K loop: wave w takes the 32-wide steps w, w + LW, ...
  const int arow = lane & 15, kg = lane >> 4;
  const int n = n0 + arow;
  const bf16* wr = p.W + (int64_t)min(n, p.N - 1) * p.ldw + kg * 8;
  const bf16* xr = xs + arow * KP + kg * 8;
  f32x4v acc = {0.f, 0.f, 0.f, 0.f};
  const int steps = p.K >> 5;
#pragma unroll 4
  for (int st = wave; st < steps; st += LW) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(wr + st * 32);
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(xr + st * 32);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  }
  // D[i = weight row (4 * (lane >> 4) + r)][j = token row (lane & 15)]
#pragma unroll
  for (int r = 0; r < 4; ++r) red[(wave * 16 + 4 * kg + r) * 16 + arow] = acc[r];
  __syncthreads();
  if (tid < 256) {
    const int m = tid >> 4, nl = tid & 15, nn = n0 + nl;
    if (m < p.M && nn < p.N) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < LW; ++w) v += red[(w * 16 + nl) * 16 + m];
      if (p.bias) v += p.bias[nn];
      const int64_t off = (int64_t)m * p.ldc + nn;
      if (p.epi == TTTS_EPI_STORE_F32) {
        static_cast<float*>(p.out)[off] = v;
      } else if (p.epi == TTTS_EPI_RESID_ADD_F32) {
        static_cast<float*>(p.out)[off] = p.resid[off] + (float)(bf16)v;
      } else if (p.epi == TTTS_EPI_GELU_BF16) {
        static_cast<bf16*>(p.out)[off] = (bf16)gelu_new_f((float)(bf16)v);
      } else {
        static_cast<bf16*>(p.out)[off] = (bf16)v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// sampling
constexpr int SAMPLE_MAXV = 2048;   // 4 float arrays + indices of this size live in LDS (36 KB)

struct SampleParams {
  const float* logits; int64_t ldl; int row_div;        // logits row of sequence m: m / row_div
  int V;
  int64_t* history; int64_t hist_stride; int hist_base;  // history[m][hist_base + step] receives the token; penalty over [0, hist_base + step)
  const int32_t* ctr;
  int64_t* tokens;                                       // [M] token for the next decode_embed
  int64_t* out; int64_t out_stride;                      // out[m][step]
  uint8_t* finished;                                     // [M]
  float repetition_penalty, typical_mass, inv_temperature, top_p;
  int top_k, do_sample, eos, pad;
  uint32_t seed_lo, seed_hi;
  float* probs_out;                                      // optional [M][V]: post-processor probabilities (tests)
  float* u_out;                                          // optional [M]: the uniform draw
};

constexpr int ST = 1024;   // sampler threads: one compare-exchange per thread and bitonic stage at V <= 2048
// bitonic sort of n (power of two) (key, idx) pairs in LDS, descending by key; ties by ascending idx (deterministic)
__device__ __forceinline__ void bitonic_desc(float* key, uint16_t* idx, int n, int tid) {
  for (int k = 2; k <= n; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int pr = tid; pr < (n >> 1); pr += ST) {                  // one compare-exchange per pair (i, i + j)
        const int i = ((pr & ~(j - 1)) << 1) | (pr & (j - 1));
        const int ixj = i + j;
        const float a = key[i], b = key[ixj];
        const uint16_t ia = idx[i], ib = idx[ixj];
        const bool a_first = (a > b) || (a == b && ia < ib);        // desired order: a before b
        const bool up = ((i & k) == 0);
        if (up ? !a_first : a_first) { key[i] = b; key[ixj] = a; idx[i] = ib; idx[ixj] = ia; }
      }
      __syncthreads();
    }
  }
}

__device__ __forceinline__ float block_reduce_sum(float v, float* sh, int tid) {
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) sh[tid >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < ST / 64; ++w) t += sh[w];
  return t;
}
__device__ __forceinline__ float block_reduce_max(float v, float* sh, int tid) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if ((tid & 63) == 0) sh[tid >> 6] = v;
  __syncthreads();
  float t = sh[0];
#pragma unroll
  for (int w = 1; w < ST / 64; ++w) t = fmaxf(t, sh[w]);
  return t;
}

// exclusive prefix sums of arr[0..n) in place order: returns through `pre` (LDS, n floats); ST threads, chunked
__device__ __forceinline__ void block_exclusive_scan(const float* arr, float* pre, int n, float* sh, int tid) {
  const int per = (n + ST - 1) / ST;
  const int lo = tid * per, hi = min(lo + per, n);
  float s = 0.f;
  for (int i = lo; i < hi; ++i) s += arr[i];
  // scan of the ST chunk sums: wave scan + wave offsets
  float inc = s;
  for (int o = 1; o < 64; o <<= 1) {
    const float v = __shfl_up(inc, o, 64);
    if ((tid & 63) >= o) inc += v;
  }
  __syncthreads();
  if ((tid & 63) == 63) sh[tid >> 6] = inc;
  __syncthreads();
  float base = 0.f;
  for (int w = 0; w < (tid >> 6); ++w) base += sh[w];
  float run = base + inc - s;
  for (int i = lo; i < hi; ++i) { pre[i] = run; run += arr[i]; }
  __syncthreads();
}

__global__ __launch_bounds__(ST) void sample_logits_kernel(SampleParams p) {
  __shared__ float sc[SAMPLE_MAXV];       // scores by token id
  __shared__ float key[SAMPLE_MAXV];      // sort keys / sorted scores / probabilities
  __shared__ float pre[SAMPLE_MAXV];      // probabilities in sort order / prefix sums
  __shared__ float cum[SAMPLE_MAXV];      // prefix sums
  __shared__ uint16_t idx[SAMPLE_MAXV];
  __shared__ uint32_t seen[SAMPLE_MAXV / 32];
  __shared__ float sh[ST / 64];
  __shared__ int sh_i[2];
  const int m = blockIdx.x, tid = threadIdx.x;
  const int V = p.V;
  int n2 = 1;
  while (n2 < V) n2 <<= 1;
  const int step = p.ctr[CTR_STEP];
  const float NEG_INF = -INFINITY;
  const float* lrow = p.logits + (int64_t)(m / p.row_div) * p.ldl;
  for (int i = tid; i < V; i += ST) sc[i] = lrow[i];
  for (int i = tid; i < SAMPLE_MAXV / 32; i += ST) seen[i] = 0u;
  __syncthreads();
  // 1. repetition penalty (RepetitionPenaltyLogitsProcessor): every token id that occurs in the row so far, once
  if (p.repetition_penalty != 1.0f) {
    const int hl = p.hist_base + step;
    for (int j = tid; j < hl; j += ST) {
      const int64_t tk = p.history[(int64_t)m * p.hist_stride + j];
      if (tk >= 0 && tk < V) atomicOr(&seen[tk >> 5], 1u << (tk & 31));
    }
    __syncthreads();
    for (int i = tid; i < V; i += ST)
      if (seen[i >> 5] & (1u << (i & 31))) { const float s = sc[i]; sc[i] = s < 0.f ? s * p.repetition_penalty : s / p.repetition_penalty; }
    __syncthreads();
  }
  // 2. typical filtering (ttts/utils/typical_sampling.py): keep the tokens whose surprise is closest to the entropy
  if (p.typical_mass > 0.f) {
    float mx = NEG_INF;
    for (int i = tid; i < V; i += ST) mx = fmaxf(mx, sc[i]);
    mx = block_reduce_max(mx, sh, tid);
    float se = 0.f;
    for (int i = tid; i < V; i += ST) se += expf(sc[i] - mx);
    se = block_reduce_sum(se, sh, tid);
    const float lse = mx + logf(se);
    float ent = 0.f;
    for (int i = tid; i < V; i += ST) { const float lp = sc[i] - lse; const float t = lp * expf(lp); ent -= (t == t) ? t : 0.f; }
    ent = block_reduce_sum(ent, sh, tid);
    // ascending by |(-logp) - ent|  ==  descending by its negative
    for (int i = tid; i < n2; i += ST) { key[i] = i < V ? -fabsf(-(sc[i] - lse) - ent) : NEG_INF; idx[i] = (uint16_t)i; }
    __syncthreads();
    bitonic_desc(key, idx, n2, tid);
    // cumulative softmax mass in that order
    for (int i = tid; i < V; i += ST) pre[i] = expf(sc[idx[i]] - lse);
    __syncthreads();
    // inclusive cumsum = exclusive scan + own value; count how many are < mass
    block_exclusive_scan(pre, cum, V, sh, tid);
    int cnt = 0;
    for (int i = tid; i < V; i += ST) cnt += (cum[i] + pre[i] < p.typical_mass) ? 1 : 0;
    float cntf = block_reduce_sum((float)cnt, sh, tid);
    int last = (int)(cntf + 0.5f);
    last = last < 0 ? 0 : (last > V - 1 ? V - 1 : last);
    const float thr = -key[last];                         // shifted score at the cut
    for (int i = tid; i < V; i += ST)
      if (-key[i] > thr) sc[idx[i]] = NEG_INF;
    __syncthreads();
  }
  // 3. temperature
  if (p.do_sample && p.inv_temperature != 1.0f) {   // the warpers (3-5) exist only when sampling, as in HF
    for (int i = tid; i < V; i += ST) sc[i] *= p.inv_temperature;
    __syncthreads();
  }
  // 4./5. top-k and top-p on one descending sort
  const bool need_sort = p.do_sample && ((p.top_k > 0 && p.top_k < V) || p.top_p < 1.0f);
  if (need_sort) {
    for (int i = tid; i < n2; i += ST) { key[i] = i < V ? sc[i] : NEG_INF; idx[i] = (uint16_t)i; }
    __syncthreads();
    bitonic_desc(key, idx, n2, tid);
    if (p.top_k > 0 && p.top_k < V) {
      const float kth = key[p.top_k - 1];
      for (int i = tid; i < V; i += ST)
        if (key[i] < kth) { sc[idx[i]] = NEG_INF; key[i] = NEG_INF; }
      __syncthreads();
    }
    if (p.top_p < 1.0f) {
      const float mx = key[0];
      float se = 0.f;
      for (int i = tid; i < V; i += ST) se += expf(key[i] - mx);
      se = block_reduce_sum(se, sh, tid);
      for (int i = tid; i < V; i += ST) pre[i] = expf(key[i] - mx) / se;      // probabilities, descending
      __syncthreads();
      block_exclusive_scan(pre, cum, V, sh, tid);                              // cum[r] = mass of the ranks before r
      // TopPLogitsWarper sorts ascending: cumulative mass up to and including rank r from the small end is 1 - cum[r];
      // remove while that is <= 1 - top_p, never the best one
      for (int r = tid; r < V; r += ST)
        if (r > 0 && (1.0f - cum[r]) <= (1.0f - p.top_p)) sc[idx[r]] = NEG_INF;
      __syncthreads();
    }
  }
  // 6. softmax over the survivors and the draw
  float mx = NEG_INF;
  for (int i = tid; i < V; i += ST) mx = fmaxf(mx, sc[i]);
  mx = block_reduce_max(mx, sh, tid);
  int token;
  if (!p.do_sample) {
    int best = V;
    for (int i = tid; i < V; i += ST)
      if (sc[i] == mx) { best = i; break; }
    __syncthreads();
    if (tid == 0) sh_i[0] = V;
    __syncthreads();
    atomicMin(&sh_i[0], best);
    __syncthreads();
    token = sh_i[0];
    if (p.probs_out)
      for (int i = tid; i < V; i += ST) p.probs_out[(int64_t)m * V + i] = sc[i];     // greedy: the processed scores
  } else {
    float se = 0.f;
    for (int i = tid; i < V; i += ST) { const float e = expf(sc[i] - mx); key[i] = e; se += e; }
    se = block_reduce_sum(se, sh, tid);
    block_exclusive_scan(key, pre, V, sh, tid);
    const uint32_t r = hash32((uint32_t)m * 0x9E3779B1u + (uint32_t)step, p.seed_lo, p.seed_hi);
    const float u = (float)(r >> 8) * (1.0f / 16777216.0f);               // [0, 1)
    const float target = u * se;
    // first token id whose inclusive cumulative weight exceeds the target
    int best = V;
    for (int i = tid; i < V; i += ST)
      if (key[i] > 0.f && pre[i] + key[i] > target) { best = i; break; }
    if (tid == 0) sh_i[0] = V;
    __syncthreads();
    atomicMin(&sh_i[0], best);
    __syncthreads();
    token = sh_i[0];
    if (token >= V) {   // rounding at the very top of the CDF: take the last survivor
      int lastv = -1;
      for (int i = tid; i < V; i += ST)
        if (key[i] > 0.f) lastv = max(lastv, i);
      if (tid == 0) sh_i[1] = -1;
      __syncthreads();
      atomicMax(&sh_i[1], lastv);
      __syncthreads();
      token = sh_i[1];
    }
    if (p.probs_out)
      for (int i = tid; i < V; i += ST) p.probs_out[(int64_t)m * V + i] = key[i] / se;
    if (p.u_out && tid == 0) p.u_out[m] = u;
  }
  // 7. eos / pad bookkeeping (GenerationMixin._sample: finished rows emit pad_token_id)
  if (tid == 0) {
    int64_t tk = token;
    if (p.finished[m]) tk = p.pad;
    else if (tk == p.eos) p.finished[m] = 1;
    p.tokens[m] = tk;
    if (p.history) p.history[(int64_t)m * p.hist_stride + p.hist_base + step] = tk;
    if (p.out) p.out[(int64_t)m * p.out_stride + step] = tk;
  }
}

__global__ void decode_advance_kernel(int32_t* ctr, const uint8_t* finished, int M) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    ctr[CTR_LEN] += 1;
    ctr[CTR_STEP] += 1;
    int u = 0;
    for (int i = 0; i < M; ++i) u += finished[i] ? 0 : 1;
    ctr[CTR_UNFINISHED] = u;
  }
}

}  // namespace ttts

using namespace ttts;

extern "C" int ttts_decode_embed_f32(const int64_t* tokens, const float* emb, const float* pos, const int32_t* ctr,
                                     int32_t pos_offset, float* x, int32_t M, int32_t D, int32_t V, int32_t P, void* stream) {
  TTTS_REQUIRE(tokens && emb && pos && ctr && x && M > 0 && D > 0 && V > 0 && P > 0, "decode_embed: bad arguments");
  decode_embed_kernel<<<M, 256, 0, as_stream(stream)>>>(tokens, emb, pos, ctr, pos_offset, x, D, V, P);
  return check_launch("decode_embed");
}

extern "C" int ttts_kv_cache_fill_bf16(const void* qkv, void* k_cache, void* v_cache, int32_t B, int32_t S, int32_t H,
                                       int32_t head_dim, int32_t S_max, int32_t rep, void* stream) {
  TTTS_REQUIRE(qkv && k_cache && v_cache && B > 0 && S > 0 && H > 0 && rep > 0, "kv_cache_fill: bad arguments");
  TTTS_REQUIRE(head_dim % 8 == 0 && S <= S_max, "kv_cache_fill: head_dim %% 8 != 0 or S > S_max");
  TTTS_REQUIRE(aligned16(qkv) && aligned16(k_cache) && aligned16(v_cache), "kv_cache_fill: 16-byte alignment");
  const int64_t total = (int64_t)B * rep * H * S * (head_dim / 8);
  kv_cache_fill_kernel<<<(int)std::min<int64_t>(cdiv(total, 256), 8192), 256, 0, as_stream(stream)>>>(
      static_cast<const bf16*>(qkv), static_cast<bf16*>(k_cache), static_cast<bf16*>(v_cache), B, S, H, head_dim, S_max, rep);
  return check_launch("kv_cache_fill");
}

extern "C" int ttts_attn_decode_bf16(const void* qkv, void* k_cache, void* v_cache, const int32_t* ctr, void* out, int32_t M,
                                     int32_t H, int32_t head_dim, int32_t S_max, float scale, void* stream) {
  TTTS_REQUIRE(qkv && k_cache && v_cache && ctr && out && M > 0 && H > 0 && S_max > 0, "attn_decode: bad arguments");
  TTTS_REQUIRE(aligned16(qkv) && aligned16(k_cache) && aligned16(v_cache), "attn_decode: 16-byte alignment");
  const float sl2 = scale * 1.4426950408889634f;
  dim3 grid((unsigned)H, (unsigned)M);
  switch (head_dim) {
    case 32: attn_decode_kernel<32><<<grid, AW * 64, 0, as_stream(stream)>>>(static_cast<const bf16*>(qkv), static_cast<bf16*>(k_cache), static_cast<bf16*>(v_cache), ctr, static_cast<bf16*>(out), H, S_max, sl2); break;
    case 64: attn_decode_kernel<64><<<grid, AW * 64, 0, as_stream(stream)>>>(static_cast<const bf16*>(qkv), static_cast<bf16*>(k_cache), static_cast<bf16*>(v_cache), ctr, static_cast<bf16*>(out), H, S_max, sl2); break;
    case 128: attn_decode_kernel<128><<<grid, AW * 64, 0, as_stream(stream)>>>(static_cast<const bf16*>(qkv), static_cast<bf16*>(k_cache), static_cast<bf16*>(v_cache), ctr, static_cast<bf16*>(out), H, S_max, sl2); break;
    default: return fail(TTTS_EUNSUPPORTED, "attn_decode: head_dim %d (32, 64, 128)", head_dim);
  }
  return check_launch("attn_decode");
}

extern "C" int ttts_linear_decode_bf16(const void* x, int64_t ldx, int32_t x_is_f32, const float* ln1_gamma, const float* ln1_beta,
                                       const float* ln2_gamma, const float* ln2_beta, const void* W, int64_t ldw,
                                       const float* bias, void* out, int64_t ldc, const float* resid, int32_t M, int32_t N,
                                       int32_t K, int32_t epilogue, void* stream) {
  TTTS_REQUIRE(x && W && out && M > 0 && M <= 16 && N > 0, "linear_decode: bad arguments (1 <= M <= 16)");
  TTTS_REQUIRE(K >= 32 && K % 32 == 0 && K <= 4096, "linear_decode: K must be a multiple of 32, <= 4096");
  TTTS_REQUIRE(ldw % 8 == 0 && aligned16(W) && aligned16(x) && ldx % (x_is_f32 ? 4 : 8) == 0, "linear_decode: alignment");
  TTTS_REQUIRE(x_is_f32 || !ln1_gamma, "linear_decode: LayerNorm fusion needs the fp32 rows");
  TTTS_REQUIRE((!ln1_gamma || ln1_beta) && (!ln2_gamma || (ln2_beta && ln1_gamma)), "linear_decode: bad LayerNorm arguments");
  TTTS_REQUIRE(epilogue == TTTS_EPI_STORE_BF16 || epilogue == TTTS_EPI_GELU_BF16 || epilogue == TTTS_EPI_RESID_ADD_F32 ||
               epilogue == TTTS_EPI_STORE_F32, "linear_decode: unsupported epilogue %d", epilogue);
  TTTS_REQUIRE(epilogue != TTTS_EPI_RESID_ADD_F32 || resid, "linear_decode: RESID_ADD needs resid");
  LinearDecodeParams p;
  p.x = x; p.ldx = ldx; p.x_is_f32 = x_is_f32; p.g1 = ln1_gamma; p.b1 = ln1_beta; p.g2 = ln2_gamma; p.b2 = ln2_beta;
  p.W = static_cast<const bf16*>(W); p.ldw = ldw; p.bias = bias; p.out = out; p.ldc = ldc; p.resid = resid;
  p.M = M; p.N = N; p.K = K; p.epi = epilogue; p.eps = 1e-5f;
  const size_t smem = (size_t)16 * (K + 8) * sizeof(bf16) + (size_t)LW * 16 * 16 * sizeof(float);
  static OnceFlag attr_set;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return fail(TTTS_EHIP, "linear_decode: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  linear_decode_kernel<<<(int)cdiv(N, 16), LW * 64, smem, as_stream(stream)>>>(p);
  return check_launch("linear_decode");
}

extern "C" int ttts_sample_logits_f32(const float* logits, int64_t ldl, int32_t row_div, int32_t M, int32_t V, int64_t* history,
                                      int64_t hist_stride, int32_t hist_base, const int32_t* ctr, int64_t* tokens, int64_t* out,
                                      int64_t out_stride, uint8_t* finished, float repetition_penalty, float typical_mass,
                                      float temperature, int32_t top_k, float top_p, int32_t do_sample, int32_t eos_token,
                                      int32_t pad_token, uint64_t seed, float* probs_out, float* u_out, void* stream) {
  TTTS_REQUIRE(logits && ctr && tokens && finished && M > 0 && row_div > 0, "sample_logits: bad arguments");
  TTTS_REQUIRE(V > 0 && V <= SAMPLE_MAXV, "sample_logits: vocabulary %d exceeds %d", V, SAMPLE_MAXV);
  TTTS_REQUIRE(temperature > 0.f && repetition_penalty > 0.f && top_p > 0.f, "sample_logits: temperature / penalty / top_p must be > 0");
  TTTS_REQUIRE(repetition_penalty == 1.0f || history, "sample_logits: repetition penalty needs the token history");
  SampleParams p;
  p.logits = logits; p.ldl = ldl; p.row_div = row_div; p.V = V;
  p.history = history; p.hist_stride = hist_stride; p.hist_base = hist_base; p.ctr = ctr; p.tokens = tokens;
  p.out = out; p.out_stride = out_stride; p.finished = finished;
  p.repetition_penalty = repetition_penalty; p.typical_mass = typical_mass; p.inv_temperature = 1.0f / temperature;
  p.top_p = top_p; p.top_k = top_k; p.do_sample = do_sample; p.eos = eos_token; p.pad = pad_token;
  p.seed_lo = (uint32_t)seed; p.seed_hi = (uint32_t)(seed >> 32);
  p.probs_out = probs_out; p.u_out = u_out;
  sample_logits_kernel<<<M, ST, 0, as_stream(stream)>>>(p);
  return check_launch("sample_logits");
}

extern "C" int ttts_decode_advance(int32_t* ctr, const uint8_t* finished, int32_t M, void* stream) {
  TTTS_REQUIRE(ctr && finished && M > 0, "decode_advance: bad arguments");
  decode_advance_kernel<<<1, 64, 0, as_stream(stream)>>>(ctr, finished, M);
  return check_launch("decode_advance");
}
