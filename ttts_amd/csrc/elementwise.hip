// HBM-bound kernels of the GPT train step: embeddings, LayerNorm, cross-entropy, bias-gradient column
// sums, fp32->bf16 shadow casts, gradient norm and AdamW.  gfx950: 64-wide waves, 16-byte vector
// accesses, one wave per row for the row-wise ops (no LDS needed at D <= 1024).
#include "common.hpp"

namespace ttts {

// ======================================================================================================
// embeddings (ttts/gpt/model.py:488,494-495,418)
// ======================================================================================================
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ text_inp,
                                                        const int64_t* __restrict__ mel_inp,
                                                        const float* __restrict__ text_emb,
                                                        const float* __restrict__ text_pos,
                                                        const float* __restrict__ mel_emb,
                                                        const float* __restrict__ mel_pos, float* __restrict__ x,
                                                        int B, int Tt, int Tm, int D, int n_text, int n_mel,
                                                        uint32_t thr, float inv_keep, uint32_t seed_lo,
                                                        uint32_t seed_hi, const uint32_t* ctr) {
  if (thr) seed_hi = seed_mix(seed_hi, ctr);
  const int S = Tt + Tm;
  const int D4 = D >> 2;
  const int64_t total = (int64_t)B * S * D4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int d4 = (int)(i % D4);
    const int64_t row = i / D4;
    const int t = (int)(row % S);
    const int b = (int)(row / S);
    float4 e, p;
    if (t < Tt) {
      int64_t tok = text_inp[(int64_t)b * Tt + t];
      tok = tok < 0 ? 0 : (tok >= n_text ? n_text - 1 : tok);
      e = reinterpret_cast<const float4*>(text_emb + tok * D)[d4];
      p = reinterpret_cast<const float4*>(text_pos + (int64_t)t * D)[d4];
    } else {
      int64_t tok = mel_inp[(int64_t)b * Tm + (t - Tt)];
      tok = tok < 0 ? 0 : (tok >= n_mel ? n_mel - 1 : tok);
      e = reinterpret_cast<const float4*>(mel_emb + tok * D)[d4];
      p = reinterpret_cast<const float4*>(mel_pos + (int64_t)(t - Tt) * D)[d4];
    }
    float4 o = make_float4(e.x + p.x, e.y + p.y, e.z + p.z, e.w + p.w);
    if (thr) {
      const uint32_t r0 = hash32((uint32_t)(i * 2), seed_lo, seed_hi);
      const uint32_t r1 = hash32((uint32_t)(i * 2 + 1), seed_lo, seed_hi);
      o.x = (r0 & 0xFFFFu) >= thr ? o.x * inv_keep : 0.f;
      o.y = (r0 >> 16) >= thr ? o.y * inv_keep : 0.f;
      o.z = (r1 & 0xFFFFu) >= thr ? o.z * inv_keep : 0.f;
      o.w = (r1 >> 16) >= thr ? o.w * inv_keep : 0.f;
    }
    reinterpret_cast<float4*>(x)[i] = o;
  }
}

// one thread per (t, d): deterministic batch sum for the positional tables, atomics for the token tables.  A wave's 64 atomics
// of one batch row cover 256 contiguous bytes of one table row (the float4-per-thread form spread them over 1 KB: 67 us -> 25 us).
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ text_inp,
                                                        const int64_t* __restrict__ mel_inp,
                                                        const float* __restrict__ dx, float* __restrict__ d_text_emb,
                                                        float* __restrict__ d_text_pos, float* __restrict__ d_mel_emb,
                                                        float* __restrict__ d_mel_pos, int B, int Tt, int Tm, int D,
                                                        uint32_t thr, float inv_keep, uint32_t seed_lo,
                                                        uint32_t seed_hi, const uint32_t* ctr) {
  if (thr) seed_hi = seed_mix(seed_hi, ctr);
  const int S = Tt + Tm;
  const int64_t total = (int64_t)S * D;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int d = (int)(i % D);
  const int t = (int)(i / D);
  float acc = 0.f;
#pragma unroll 4
  for (int b = 0; b < B; ++b) {
    const int64_t lin = ((int64_t)b * S + t) * D + d;
    float g = dx[lin];
    if (thr) {                                                  // the mask of embed_fwd: one hash word per element pair
      const uint32_t r = hash32((uint32_t)(lin >> 1), seed_lo, seed_hi);
      g = ((d & 1) ? (r >> 16) : (r & 0xFFFFu)) >= thr ? g * inv_keep : 0.f;
    }
    acc += g;
    float* dst = t < Tt ? d_text_emb + text_inp[(int64_t)b * Tt + t] * D + d : d_mel_emb + mel_inp[(int64_t)b * Tm + (t - Tt)] * D + d;
    atomicAdd(dst, g);
  }
  float* pos = (t < Tt ? d_text_pos + (int64_t)t * D : d_mel_pos + (int64_t)(t - Tt) * D) + d;
  *pos += acc;
}

// ======================================================================================================
// LayerNorm: one wave per row, VPL float4 chunks per lane (D <= VPL*256)
// ======================================================================================================
__device__ __forceinline__ int64_t split_row(int64_t row, int split_S, int split_T, int nB) {
  if (split_S <= 0) return row;
  const int64_t b = row / split_S;
  const int t = (int)(row % split_S);
  return t < split_T ? b * split_T + t : (int64_t)nB * split_T + b * (split_S - split_T) + (t - split_T);
}

constexpr int LN_FWD_RPW = 2;   // rows per wave: both rows' loads are in flight before the first reduction (no gain measured at M = 9248: 8.1 us either way, a latency-bound launch)
template <int VPL, bool OUT_BF16>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, void* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out, int M,
                                                     int D, float eps, int split_S, int split_T) {
  const int lane = threadIdx.x & 63;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * LN_FWD_RPW;
  if (row0 >= M) return;
  float4 v[LN_FWD_RPW][VPL];
  float s[LN_FWD_RPW];
#pragma unroll
  for (int r = 0; r < LN_FWD_RPW; ++r) {
    const float* xr = x + min(row0 + r, (int64_t)M - 1) * D;
    s[r] = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int d = (lane + i * 64) * 4;
      v[r][i] = d < D ? *reinterpret_cast<const float4*>(xr + d) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  float4 g[VPL], bt[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int d = (lane + i * 64) * 4;
    g[i] = d < D ? *reinterpret_cast<const float4*>(gamma + d) : make_float4(0.f, 0.f, 0.f, 0.f);
    bt[i] = d < D ? *reinterpret_cast<const float4*>(beta + d) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int nB = split_S > 0 ? M / split_S : 0;
#pragma unroll
  for (int r = 0; r < LN_FWD_RPW; ++r) {
    const int64_t row = row0 + r;
    if (row >= M) break;
    bool ok[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) ok[i] = (lane + i * 64) * 4 < D;
    float mean, rstd;
    ln_row_stats<VPL>(v[r], ok, D, eps, mean, rstd);   // (common.hpp: shared with the fused residual-GEMM + LayerNorm kernel)
    if (lane == 0) {
      mean_out[row] = mean;
      rstd_out[row] = rstd;
    }
    const int64_t orow = split_row(row, split_S, split_T, nB);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int d = (lane + i * 64) * 4;
      if (d < D) {
        const float4 o = ln_row_apply(v[r][i], mean, rstd, g[i], bt[i]);
        if (OUT_BF16) {
          bf16x4 ob;
          ob[0] = (bf16)o.x; ob[1] = (bf16)o.y; ob[2] = (bf16)o.z; ob[3] = (bf16)o.w;
          *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(y) + orow * D + d) = ob;
        } else {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(y) + orow * D + d) = o;
        }
      }
    }
  }
}

constexpr int LN_BWD_ROWS = 8;   // rows per block (2 per wave): 1156 workgroups at M = 9248 keep ~18 waves per CU in flight (16 rows: kernel 17 -> 21 us, more than the halved partial sums give back)

template <int VPL, bool DY_BF16>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* dx_in, float* dx,
                                                     bf16* __restrict__ dx_bf16, float* __restrict__ partial, int M,
                                                     int D, int split_S, int split_T, uint32_t thr, float inv_keep,
                                                     uint32_t seed_lo, uint32_t seed_hi, const uint32_t* ctr) {
  if (thr) seed_hi = seed_mix(seed_hi, ctr);
  extern __shared__ __attribute__((aligned(16))) float ln_smem[];  // [4 waves][3][D]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nB = split_S > 0 ? M / split_S : 0;
  float4 dg[VPL], db[VPL], gm[VPL], dc[VPL];  // dc: column sums of the bf16 copy (bias gradient of its consumer)
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    dc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int d = (lane + i * 64) * 4;
    gm[i] = d < D ? *reinterpret_cast<const float4*>(gamma + d) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int64_t row0 = (int64_t)blockIdx.x * LN_BWD_ROWS;
  for (int rr = wave; rr < LN_BWD_ROWS; rr += 4) {
    const int64_t row = row0 + rr;
    if (row >= M) break;
    const int64_t yrow = split_row(row, split_S, split_T, nB);
    const float mu = mean[row], rs = rstd[row];
    float4 g[VPL], xh[VPL];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int d = (lane + i * 64) * 4;
      if (d < D) {
        float4 dyv;
        if (DY_BF16) {
          const bf16x4 t = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(dy) + yrow * D + d);
          dyv = make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
        } else {
          dyv = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(dy) + yrow * D + d);
        }
        const float4 xv = *reinterpret_cast<const float4*>(x + row * D + d);
        xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
        g[i] = make_float4(dyv.x * gm[i].x, dyv.y * gm[i].y, dyv.z * gm[i].z, dyv.w * gm[i].w);
        c1 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
        c2 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
        dg[i].x += dyv.x * xh[i].x; dg[i].y += dyv.y * xh[i].y; dg[i].z += dyv.z * xh[i].z; dg[i].w += dyv.w * xh[i].w;
        db[i].x += dyv.x; db[i].y += dyv.y; db[i].z += dyv.z; db[i].w += dyv.w;
      } else {
        g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        xh[i] = g[i];
      }
    }
    c1 = wave_sum(c1) / (float)D;
    c2 = wave_sum(c2) / (float)D;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int d = (lane + i * 64) * 4;
      if (d < D) {
        float4 o;
        o.x = rs * (g[i].x - c2 - xh[i].x * c1);
        o.y = rs * (g[i].y - c2 - xh[i].y * c1);
        o.z = rs * (g[i].z - c2 - xh[i].z * c1);
        o.w = rs * (g[i].w - c2 - xh[i].w * c1);
        if (dx_in) {
          const float4 a = *reinterpret_cast<const float4*>(dx_in + row * D + d);
          o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        }
        *reinterpret_cast<float4*>(dx + row * D + d) = o;
        if (dx_bf16) {
          if (thr) {
            const uint32_t lin = (uint32_t)((row * D + d) >> 1);
            const uint32_t r0 = hash32(lin, seed_lo, seed_hi), r1 = hash32(lin + 1, seed_lo, seed_hi);
            o.x = (r0 & 0xFFFFu) >= thr ? o.x * inv_keep : 0.f;
            o.y = (r0 >> 16) >= thr ? o.y * inv_keep : 0.f;
            o.z = (r1 & 0xFFFFu) >= thr ? o.z * inv_keep : 0.f;
            o.w = (r1 >> 16) >= thr ? o.w * inv_keep : 0.f;
          }
          bf16x4 ob;
          ob[0] = (bf16)o.x; ob[1] = (bf16)o.y; ob[2] = (bf16)o.z; ob[3] = (bf16)o.w;
          *reinterpret_cast<bf16x4*>(dx_bf16 + row * D + d) = ob;
          dc[i].x += (float)ob[0]; dc[i].y += (float)ob[1]; dc[i].z += (float)ob[2]; dc[i].w += (float)ob[3];
        }
      }
    }
  }
  // cross-wave reduction of the per-lane column partials, then one partial row per block: [dgamma | dbeta | dcolsum]
  float* sg = ln_smem + (size_t)wave * 3 * D;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int d = (lane + i * 64) * 4;
    if (d < D) {
      *reinterpret_cast<float4*>(sg + d) = dg[i];
      *reinterpret_cast<float4*>(sg + D + d) = db[i];
      *reinterpret_cast<float4*>(sg + 2 * D + d) = dc[i];
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 3 * D; j += 256) {
    const float s = (ln_smem[j] + ln_smem[3 * D + j]) + (ln_smem[6 * D + j] + ln_smem[9 * D + j]);
    partial[(size_t)blockIdx.x * 3 * D + j] = s;
  }
}

// out[j] += sum over blocks of partial[b][j]: 16 columns (one 64-byte segment per partial row) x 64 row lanes per workgroup,
// fixed summation order.  96 workgroups at D = 512 (the 64-column form had 24: 8.3 us for 7 MB, bound by what 24 CUs can pull)
constexpr int LNF_COLS = 16, LNF_LANES = 64;
__global__ __launch_bounds__(1024) void ln_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int D,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               float* __restrict__ dcolsum) {
  __shared__ float sh[LNF_LANES][LNF_COLS + 1];
  const int tx = threadIdx.x % LNF_COLS, ty = threadIdx.x / LNF_COLS;
  const int j = blockIdx.x * LNF_COLS + tx;
  float s = 0.f;
  if (j < 3 * D) {
#pragma unroll 4
    for (int b = ty; b < nblk; b += LNF_LANES) s += partial[(size_t)b * 3 * D + j];
  }
  sh[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && j < 3 * D) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < LNF_LANES; ++r) t += sh[r][tx];
    if (j < D) dgamma[j] += t;
    else if (j < 2 * D) dbeta[j - D] += t;
    else if (dcolsum) dcolsum[j - 2 * D] += t;
  }
}

// the same for several deferred ttts_layernorm_bwd_ex calls in one launch: blockIdx.y picks the call.  With 14 calls' worth of
// workgroups the columns can be 32 wide (128-byte row segments instead of 64): each thread carries the two row classes ty and
// ty + 32 of the single-call kernel separately, so the sums are formed in exactly its order (bitwise equal results)
constexpr int LNB_COLS = 32;
__global__ __launch_bounds__(1024) void ln_bwd_finalize_batched_kernel(const ttts_ln_finalize_desc* __restrict__ desc, int nblk, int D) {
  __shared__ float sh[LNF_LANES][LNB_COLS + 1];
  const ttts_ln_finalize_desc dd = desc[blockIdx.y];
  const float* partial = reinterpret_cast<const float*>(dd.workspace);
  const int tx = threadIdx.x % LNB_COLS, ty = threadIdx.x / LNB_COLS;      // ty < 32
  const int j = blockIdx.x * LNB_COLS + tx;
  float s0 = 0.f, s1 = 0.f;
  if (j < 3 * D) {
#pragma unroll 4
    for (int b = ty; b < nblk; b += LNF_LANES) {
      s0 += partial[(size_t)b * 3 * D + j];
      if (b + 32 < nblk) s1 += partial[(size_t)(b + 32) * 3 * D + j];
    }
  }
  sh[ty][tx] = s0;
  sh[ty + 32][tx] = s1;
  __syncthreads();
  if (ty == 0 && j < 3 * D) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < LNF_LANES; ++r) t += sh[r][tx];
    if (j < D) dd.dgamma[j] += t;
    else if (j < 2 * D) dd.dbeta[j - D] += t;
    else if (dd.dcolsum) dd.dcolsum[j - 2 * D] += t;
  }
}

// ======================================================================================================
// cross-entropy: one wave per row of bf16 logits
// ======================================================================================================
__global__ __launch_bounds__(256) void ce_fwd_kernel(const bf16* __restrict__ logits, int64_t ldl,
                                                     const int64_t* __restrict__ targets, float* __restrict__ row_loss,
                                                     float* __restrict__ row_lse, int R, int C) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const bf16* lr = logits + row * ldl;
  float mx = -INFINITY;
  for (int c0 = lane * 8; c0 < C; c0 += 64 * 8) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(lr + c0);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (c0 + j < C) mx = fmaxf(mx, (float)v[j]);
  }
  mx = wave_max(mx);
  float s = 0.f;
  for (int c0 = lane * 8; c0 < C; c0 += 64 * 8) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(lr + c0);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (c0 + j < C) s += __expf((float)v[j] - mx);
  }
  s = wave_sum(s);
  if (lane == 0) {
    const float lse = mx + logf(s);
    int64_t t = targets[row];
    t = t < 0 ? 0 : (t >= C ? C - 1 : t);
    row_lse[row] = lse;
    row_loss[row] = lse - (float)lr[t];
  }
}

// deterministic mean of R per-row losses (single block)
__global__ __launch_bounds__(1024) void mean_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
  __shared__ double sh[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) s += (double)v[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 16; ++i) t += sh[i];
    *out = (float)(t / (double)n);
  }
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(const bf16* __restrict__ logits, int64_t ldl,
                                                     const int64_t* __restrict__ targets,
                                                     const float* __restrict__ row_lse, bf16* __restrict__ dlogits,
                                                     float scale, const float* __restrict__ scale_dev, int R, int C) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const bf16* lr = logits + row * ldl;
  bf16* dr = dlogits + row * ldl;
  const float lse = row_lse[row];
  const float sc = scale * (scale_dev ? *scale_dev : 1.0f) / (float)R;
  int64_t t = targets[row];
  t = t < 0 ? 0 : (t >= C ? C - 1 : t);
  for (int c0 = lane * 8; c0 < (int)ldl; c0 += 64 * 8) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(lr + c0);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j;
      float g = 0.f;
      if (c < C) g = (__expf((float)v[j] - lse) - (c == (int)t ? 1.0f : 0.0f)) * sc;
      o[j] = (bf16)g;
    }
    *reinterpret_cast<bf16x8*>(dr + c0) = o;  // columns in [C, ldl) are written as zeros (GEMM K padding)
  }
}

// ======================================================================================================
// bias gradients: out[n] += sum_m X[m][n]
// ======================================================================================================
// workgroup = 64 columns x 256 rows: 8 column chunks (16 B) x 32 row lanes, 8 rows per thread, LDS reduction over
// the row lanes, then one atomic per column and workgroup (M/256 atomics per column in total)
constexpr int COLSUM_ROWS = 256;
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* __restrict__ X, int64_t ldx, float* __restrict__ out,
                                                     int M, int N) {
  __shared__ float sh[32][65];
  const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
  const int n0 = blockIdx.x * 64 + tx * 8;
  const int m0 = blockIdx.y * COLSUM_ROWS;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (n0 < N) {
#pragma unroll
    for (int r = 0; r < COLSUM_ROWS / 32; ++r) {
      const int m = m0 + ty + 32 * r;
      if (m < M) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(X + (int64_t)m * ldx + n0);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) sh[ty][tx * 8 + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) s += sh[r][threadIdx.x];
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (n < N) atomicAdd(out + n, s);
  }
}

// several column sums in one launch: a workgroup finds its descriptor from the tile prefix sums (at most 64 descriptors)
__global__ __launch_bounds__(256) void colsum_batched_kernel(const ttts_colsum_desc* __restrict__ desc, int n_desc) {
  __shared__ float sh[32][65];
  int di = 0;
  for (int i = 1; i < n_desc; ++i)
    if ((int)blockIdx.x >= desc[i].tile_begin) di = i;
  const ttts_colsum_desc d = desc[di];
  const bf16* X = reinterpret_cast<const bf16*>(d.X);
  const int tiles_n = (d.N + 63) >> 6;
  const int tl = blockIdx.x - d.tile_begin;
  const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
  const int n0 = (tl % tiles_n) * 64 + tx * 8;
  const int m0 = (tl / tiles_n) * COLSUM_ROWS;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (n0 < d.N) {
#pragma unroll
    for (int r = 0; r < COLSUM_ROWS / 32; ++r) {
      const int m = m0 + ty + 32 * r;
      if (m < d.M) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(X + (int64_t)m * d.ldx + n0);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) sh[ty][tx * 8 + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) s += sh[r][threadIdx.x];
    const int n = (tl % tiles_n) * 64 + threadIdx.x;
    if (n < d.N) atomicAdd(d.out + n, s);
  }
}

// ======================================================================================================
// batched fp32 -> bf16 cast with optional transposed copy (32x32 tiles through LDS)
// ======================================================================================================
__global__ __launch_bounds__(256) void cast_batched_kernel(const ttts_cast_desc* __restrict__ desc, int n_desc) {
  __shared__ float tile[32][33];
  int di = 0;
  for (int i = 1; i < n_desc; ++i)
    if ((int)blockIdx.x >= desc[i].tile_begin) di = i;
  const ttts_cast_desc d = desc[di];
  const int tiles_c = (d.cols + 31) >> 5;
  const int tl = blockIdx.x - d.tile_begin;
  const int r0 = (tl / tiles_c) * 32, c0 = (tl % tiles_c) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  bf16* dst = reinterpret_cast<bf16*>(d.dst);
  bf16* dst_t = reinterpret_cast<bf16*>(d.dst_t);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    float v = 0.f;
    if (r < d.rows && c < d.cols) {
      v = d.src[(int64_t)r * d.cols + c];
      if (dst) dst[(int64_t)r * d.cols + c] = (bf16)v;
    }
    tile[ty + 8 * i][tx] = v;
  }
  if (!dst_t) return;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (r < d.rows && c < d.cols) dst_t[(int64_t)c * (d.ldt > 0 ? d.ldt : d.rows) + r] = (bf16)tile[tx][ty + 8 * i];
  }
}

// batched bf16 transpose, 64 x 64 tiles through LDS: 16-byte loads along the source rows, 16-byte stores along the destination
// rows (the transposed GEMM operand copies of the bf16 shadow weights AdamW has just written)
__global__ __launch_bounds__(256) void transpose_bf16_batched_kernel(const ttts_transpose_desc* __restrict__ desc, int n_desc) {
  __shared__ bf16 tile[64][66];
  int di = 0;
  for (int i = 1; i < n_desc; ++i)
    if ((int)blockIdx.x >= desc[i].tile_begin) di = i;
  const ttts_transpose_desc d = desc[di];
  const bf16* src = reinterpret_cast<const bf16*>(d.src);
  bf16* dst = reinterpret_cast<bf16*>(d.dst);
  const int tiles_c = (d.cols + 63) >> 6;
  const int tl = blockIdx.x - d.tile_begin;
  const int r0 = (tl / tiles_c) * 64, c0 = (tl % tiles_c) * 64;
  const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;           // 8 chunks of 8 elements x 32 rows
  const bool vec_in = (d.cols & 7) == 0 && ((uintptr_t)src & 15) == 0;
  const bool vec_out = (d.ldd & 7) == 0 && ((uintptr_t)dst & 15) == 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = r0 + ty + 32 * i, c = c0 + tx * 8;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
    if (r < d.rows) {
      if (vec_in && c + 8 <= d.cols) v = *reinterpret_cast<const bf16x8*>(src + (int64_t)r * d.cols + c);
      else
#pragma unroll
        for (int e = 0; e < 8; ++e) if (c + e < d.cols) v[e] = src[(int64_t)r * d.cols + c + e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[ty + 32 * i][tx * 8 + e] = v[e];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = c0 + ty + 32 * i, r = r0 + tx * 8;                // destination row c, columns r .. r + 7
    if (c >= d.cols) continue;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[tx * 8 + e][ty + 32 * i];
    if (vec_out && r + 8 <= d.rows) *reinterpret_cast<bf16x8*>(dst + (int64_t)c * d.ldd + r) = v;
    else
#pragma unroll
      for (int e = 0; e < 8; ++e) if (r + e < d.rows) dst[(int64_t)c * d.ldd + r + e] = v[e];
  }
}

// ======================================================================================================
// optimizer: schedule, gradient norm, AdamW on a flat arena
// ======================================================================================================
__global__ void adamw_schedule_kernel(float* state, float base_lr, float beta1, float beta2, int warmup_steps) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (state[6] != 0.f) return;                 // skipped step (ABI v11: set by ttts_loss_scale_check): the schedule does not advance
  const double step = (double)state[0] + 1.0;  // exact in fp32 up to 2^24 steps
  double factor = 1.0;
  if (warmup_steps > 0) {
    const double s = step - 1.0;  // LambdaLR value in effect for this optimizer step
    factor = s < (double)warmup_steps ? s / (double)warmup_steps : 1.0;
  }
  state[0] = (float)step;
  state[1] = (float)((double)base_lr * factor);
  state[2] = (float)(1.0 - pow((double)beta1, step));
  state[3] = (float)sqrt(1.0 - pow((double)beta2, step));
}

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, int64_t n4,
                                                            double* __restrict__ partial) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ((double)sh[0] + (double)sh[1]) + ((double)sh[2] + (double)sh[3]);
}

__global__ __launch_bounds__(256) void gradnorm_final_kernel(const double* __restrict__ partial, int nblk,
                                                             const float* __restrict__ g, int64_t n, int64_t tail_begin,
                                                             float max_norm, float* __restrict__ state) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) s += partial[i];
  for (int64_t i = tail_begin + threadIdx.x; i < n; i += 256) s += (double)g[i] * (double)g[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt((sh[0] + sh[1]) + (sh[2] + sh[3]));
    state[4] = norm;
    float coef = 1.0f;
    if (max_norm > 0.f) coef = fminf(1.0f, max_norm / (norm + 1e-6f));
    state[5] = coef;
  }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16* __restrict__ shadow, int64_t n4,
                                                    const float* __restrict__ state, float beta1, float beta2,
                                                    float eps, float wd, int zero_grad) {
  const float lr = state[1], bc1 = state[2], bc2s = state[3], coef = state[5];
  const float decay = 1.0f - lr * wd;
  const float step_size = lr / bc1;
  if (state[6] != 0.f) {                       // skipped step: parameters, moments and shadow stay; the gradient is still consumed
    if (zero_grad)
      for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
        reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    float4 gv = reinterpret_cast<float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float* pp = reinterpret_cast<float*>(&pv);
    float* gp = reinterpret_cast<float*>(&gv);
    float* mp = reinterpret_cast<float*>(&mv);
    float* vp = reinterpret_cast<float*>(&vv);
    bf16x4 sb;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gj = gp[j] * coef;
      float pj = pp[j] * decay;                              // param.mul_(1 - lr * wd)
      const float mj = mp[j] + (gj - mp[j]) * (1.0f - beta1);  // exp_avg.lerp_(grad, 1 - beta1)
      const float vj = vp[j] * beta2 + (1.0f - beta2) * gj * gj;  // mul_(beta2).addcmul_(g, g, 1 - beta2)
      const float denom = sqrtf(vj) / bc2s + eps;
      pj = pj - step_size * (mj / denom);                    // addcdiv_(exp_avg, denom, -step_size)
      pp[j] = pj; mp[j] = mj; vp[j] = vj;
      sb[j] = (bf16)pj;
    }
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(v)[i] = vv;
    if (zero_grad) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (shadow) reinterpret_cast<bf16x4*>(shadow)[i] = sb;
  }
}

}  // namespace ttts

// ======================================================================================================
// C ABI
// ======================================================================================================
using namespace ttts;

// ---- token plumbing of UnifiedVoice.forward in one launch ----------------------------------------------------------
// (the torch form is ~20 tiny launches per step: arange, where, six pads = fill + copy each, four buffer copies)
constexpr int PREP_MAX_B = 256;
struct PrepValid { int32_t v[PREP_MAX_B]; };   // per sample: first mel position rewritten to STOP (wav_len / compression + 1)

__global__ __launch_bounds__(256) void gpt_prepare_tokens_kernel(const int64_t* __restrict__ text, int64_t ldt,
                                                                 const int64_t* __restrict__ mel, int64_t ldm,
                                                                 int64_t* __restrict__ text_inp, int64_t* __restrict__ text_tar,
                                                                 int64_t* __restrict__ mel_inp, int64_t* __restrict__ mel_tar,
                                                                 int B, int Tt, int Tm, int64_t start_text, int64_t stop_text,
                                                                 int64_t start_mel, int64_t stop_mel, PrepValid valid) {
  // thread (b, i): i in [0, Tt + 2) handles the text pair, i in [Tt + 2, Tt + Tm + 4) the mel pair
  const int W = Tt + Tm + 4;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)B * W) return;
  const int b = (int)(t / W);
  int i = (int)(t % W);
  if (i < Tt + 2) {
    // t1 = [text[0..Tt), STOP];  inp = [START, t1];  tar = [t1, STOP]
    const int64_t inp = i == 0 ? start_text : (i - 1 < Tt ? text[(int64_t)b * ldt + i - 1] : stop_text);
    const int64_t tar = i < Tt ? text[(int64_t)b * ldt + i] : stop_text;
    text_inp[(int64_t)b * (Tt + 2) + i] = inp;
    text_tar[(int64_t)b * (Tt + 2) + i] = tar;
  } else {
    i -= Tt + 2;
    const int nv = valid.v[b];   // positions >= nv are padding: rewritten to STOP (set_mel_padding)
    auto m1 = [&](int j) -> int64_t { return (j < Tm && j < nv) ? mel[(int64_t)b * ldm + j] : stop_mel; };
    const int64_t inp = i == 0 ? start_mel : m1(i - 1);
    const int64_t tar = i <= Tm ? m1(i) : stop_mel;
    mel_inp[(int64_t)b * (Tm + 2) + i] = inp;
    mel_tar[(int64_t)b * (Tm + 2) + i] = tar;
  }
}

extern "C" int ttts_gpt_prepare_tokens(const int64_t* text, int64_t ld_text, const int64_t* mel, int64_t ld_mel,
                                       const int32_t* mel_valid_host, int32_t B, int32_t Tt, int32_t Tm,
                                       int32_t start_text, int32_t stop_text, int32_t start_mel, int32_t stop_mel,
                                       int64_t* text_inp, int64_t* text_tar, int64_t* mel_inp, int64_t* mel_tar, void* stream) {
  TTTS_REQUIRE(text && mel && mel_valid_host && text_inp && text_tar && mel_inp && mel_tar, "prepare_tokens: null pointer");
  TTTS_REQUIRE(B > 0 && B <= PREP_MAX_B && Tt >= 0 && Tm >= 0, "prepare_tokens: bad shape B=%d (max %d) Tt=%d Tm=%d", B, PREP_MAX_B, Tt, Tm);
  TTTS_REQUIRE(ld_text >= Tt && ld_mel >= Tm, "prepare_tokens: row pitch smaller than the clipped length");
  PrepValid v;
  for (int b = 0; b < B; ++b) {
    TTTS_REQUIRE(mel_valid_host[b] >= 0, "prepare_tokens: negative valid length");
    v.v[b] = mel_valid_host[b];
  }
  const int64_t total = (int64_t)B * (Tt + Tm + 4);
  gpt_prepare_tokens_kernel<<<(int)cdiv(total, 256), 256, 0, as_stream(stream)>>>(text, ld_text, mel, ld_mel, text_inp, text_tar,
                                                                                 mel_inp, mel_tar, B, Tt, Tm, start_text, stop_text,
                                                                                 start_mel, stop_mel, v);
  return check_launch("gpt_prepare_tokens");
}

extern "C" int ttts_gpt_embed_fwd(const int64_t* text_inp, const int64_t* mel_inp, const float* text_emb,
                                  const float* text_pos, const float* mel_emb, const float* mel_pos, float* x,
                                  int32_t B, int32_t Tt, int32_t Tm, int32_t D, int32_t n_text, int32_t n_mel,
                                  float dropout_p, uint64_t seed, const uint32_t* dropout_counter, void* stream) {
  TTTS_REQUIRE(text_inp && mel_inp && text_emb && text_pos && mel_emb && mel_pos && x, "embed_fwd: null pointer");
  TTTS_REQUIRE(B > 0 && Tt >= 0 && Tm >= 0 && Tt + Tm > 0 && D > 0 && D % 4 == 0, "embed_fwd: bad shape B=%d Tt=%d Tm=%d D=%d", B, Tt, Tm, D);
  TTTS_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "embed_fwd: dropout_p out of range");
  const int64_t total = (int64_t)B * (Tt + Tm) * (D / 4);
  const int grid = (int)std::min<int64_t>(cdiv(total, 256), 4096);
  const uint32_t thr = dropout_threshold(dropout_p);
  const float inv_keep = thr ? 65536.0f / (65536.0f - (float)thr) : 1.0f;
  embed_fwd_kernel<<<grid, 256, 0, as_stream(stream)>>>(text_inp, mel_inp, text_emb, text_pos, mel_emb, mel_pos, x, B,
                                                        Tt, Tm, D, n_text, n_mel, thr, inv_keep, (uint32_t)seed,
                                                        (uint32_t)(seed >> 32), dropout_counter);
  return check_launch("embed_fwd");
}

extern "C" int ttts_gpt_embed_bwd(const int64_t* text_inp, const int64_t* mel_inp, const float* dx, float* d_text_emb,
                                  float* d_text_pos, float* d_mel_emb, float* d_mel_pos, int32_t B, int32_t Tt,
                                  int32_t Tm, int32_t D, float dropout_p, uint64_t seed, const uint32_t* dropout_counter,
                                  void* stream) {
  TTTS_REQUIRE(text_inp && mel_inp && dx && d_text_emb && d_text_pos && d_mel_emb && d_mel_pos, "embed_bwd: null pointer");
  TTTS_REQUIRE(B > 0 && Tt + Tm > 0 && D > 0 && D % 4 == 0, "embed_bwd: bad shape");
  const int64_t total = (int64_t)(Tt + Tm) * D;
  const uint32_t thr = dropout_threshold(dropout_p);
  const float inv_keep = thr ? 65536.0f / (65536.0f - (float)thr) : 1.0f;
  embed_bwd_kernel<<<(int)cdiv(total, 256), 256, 0, as_stream(stream)>>>(text_inp, mel_inp, dx, d_text_emb, d_text_pos,
                                                                         d_mel_emb, d_mel_pos, B, Tt, Tm, D, thr,
                                                                         inv_keep, (uint32_t)seed, (uint32_t)(seed >> 32), dropout_counter);
  return check_launch("embed_bwd");
}

extern "C" int ttts_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, int32_t y_is_bf16,
                                  float* mean, float* rstd, int32_t M, int32_t D, float eps, int32_t split_S,
                                  int32_t split_T, void* stream) {
  TTTS_REQUIRE(x && gamma && beta && y && mean && rstd, "layernorm_fwd: null pointer");
  TTTS_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 1024, "layernorm_fwd: need 0 < D <= 1024, D %% 4 == 0 (D=%d)", D);
  TTTS_REQUIRE(split_S <= 0 || (M % split_S == 0 && split_T >= 0 && split_T <= split_S), "layernorm_fwd: bad split");
  const int grid = (int)cdiv(M, 4 * LN_FWD_RPW);
  hipStream_t s = as_stream(stream);
#define LN_FWD(V)                                                                                                     \
  if (y_is_bf16) ln_fwd_kernel<V, true><<<grid, 256, 0, s>>>(x, gamma, beta, y, mean, rstd, M, D, eps, split_S, split_T); \
  else ln_fwd_kernel<V, false><<<grid, 256, 0, s>>>(x, gamma, beta, y, mean, rstd, M, D, eps, split_S, split_T);
  if (D <= 256) { LN_FWD(1) } else if (D <= 512) { LN_FWD(2) } else { LN_FWD(4) }
#undef LN_FWD
  return check_launch("layernorm_fwd");
}

extern "C" int64_t ttts_layernorm_bwd_workspace_bytes(int32_t M, int32_t D) {
  return (int64_t)cdiv(M, LN_BWD_ROWS) * 3 * D * (int64_t)sizeof(float);
}

// internal entry with the dropout arguments for the bf16 copy (used by the GPT step)
static int layernorm_bwd_impl(const void* dy, int dy_is_bf16, const float* x, const float* gamma, const float* mean,
                              const float* rstd, const float* dx_in, float* dx, void* dx_bf16, float* dgamma,
                              float* dbeta, float* dcolsum, void* workspace, int M, int D, int split_S, int split_T,
                              float drop_p, uint64_t seed, const uint32_t* dropout_counter, hipStream_t s) {
  TTTS_REQUIRE(dy && x && gamma && mean && rstd && dx && workspace, "layernorm_bwd: null pointer");
  TTTS_REQUIRE((dgamma != nullptr) == (dbeta != nullptr), "layernorm_bwd: dgamma and dbeta are given or deferred together");
  TTTS_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 1024, "layernorm_bwd: need 0 < D <= 1024, D %% 4 == 0 (D=%d)", D);
  TTTS_REQUIRE(split_S <= 0 || (M % split_S == 0 && split_T >= 0 && split_T <= split_S), "layernorm_bwd: bad split");
  TTTS_REQUIRE(!dcolsum || dx_bf16, "layernorm_bwd: dcolsum needs the bf16 copy");
  const int nblk = (int)cdiv(M, LN_BWD_ROWS);
  const size_t smem = (size_t)4 * 3 * D * sizeof(float);
  float* partial = reinterpret_cast<float*>(workspace);
  const uint32_t thr = dropout_threshold(drop_p);
  const float inv_keep = thr ? 65536.0f / (65536.0f - (float)thr) : 1.0f;
#define LN_BWD(V)                                                                                                   \
  if (dy_is_bf16)                                                                                                   \
    ln_bwd_kernel<V, true><<<nblk, 256, smem, s>>>(dy, x, gamma, mean, rstd, dx_in, dx, (bf16*)dx_bf16, partial, M, D, \
                                                   split_S, split_T, thr, inv_keep, (uint32_t)seed, (uint32_t)(seed >> 32), dropout_counter); \
  else                                                                                                              \
    ln_bwd_kernel<V, false><<<nblk, 256, smem, s>>>(dy, x, gamma, mean, rstd, dx_in, dx, (bf16*)dx_bf16, partial, M, D, \
                                                    split_S, split_T, thr, inv_keep, (uint32_t)seed, (uint32_t)(seed >> 32), dropout_counter);
  if (D <= 256) { LN_BWD(1) } else if (D <= 512) { LN_BWD(2) } else { LN_BWD(4) }
#undef LN_BWD
  int rc = check_launch("layernorm_bwd");
  if (rc || !dgamma) return rc;           // deferred: the partial sums wait in the workspace for ttts_layernorm_bwd_finalize_batched
  ln_bwd_finalize_kernel<<<(int)cdiv(3 * D, LNF_COLS), 1024, 0, s>>>(partial, nblk, D, dgamma, dbeta, dcolsum);
  return check_launch("layernorm_bwd_finalize");
}

extern "C" int ttts_layernorm_bwd_finalize_batched(const ttts_ln_finalize_desc* desc, int32_t n_desc, int32_t M, int32_t D,
                                                   void* stream) {
  TTTS_REQUIRE(desc && n_desc > 0 && M > 0 && D > 0 && D % 4 == 0 && D <= 1024, "layernorm_bwd_finalize_batched: bad arguments");
  dim3 grid((unsigned)cdiv(3 * D, LNB_COLS), (unsigned)n_desc);
  ln_bwd_finalize_batched_kernel<<<grid, 1024, 0, as_stream(stream)>>>(desc, (int)cdiv(M, LN_BWD_ROWS), D);
  return check_launch("layernorm_bwd_finalize_batched");
}

extern "C" int ttts_layernorm_bwd(const void* dy, int32_t dy_is_bf16, const float* x, const float* gamma,
                                  const float* mean, const float* rstd, const float* dx_in, float* dx, void* dx_bf16,
                                  float* dgamma, float* dbeta, void* workspace, int32_t M, int32_t D, int32_t split_S,
                                  int32_t split_T, void* stream) {
  return layernorm_bwd_impl(dy, dy_is_bf16, x, gamma, mean, rstd, dx_in, dx, dx_bf16, dgamma, dbeta, nullptr, workspace,
                            M, D, split_S, split_T, 0.f, 0, nullptr, as_stream(stream));
}

extern "C" int ttts_layernorm_bwd_ex(const void* dy, int32_t dy_is_bf16, const float* x, const float* gamma,
                                     const float* mean, const float* rstd, const float* dx_in, float* dx, void* dx_bf16,
                                     float* dgamma, float* dbeta, float* dcolsum, void* workspace, int32_t M, int32_t D,
                                     int32_t split_S, int32_t split_T, float bf16_dropout_p, uint64_t bf16_dropout_seed,
                                     const uint32_t* dropout_counter, void* stream) {
  TTTS_REQUIRE(bf16_dropout_p >= 0.f && bf16_dropout_p < 1.f, "layernorm_bwd: dropout_p out of range");
  return layernorm_bwd_impl(dy, dy_is_bf16, x, gamma, mean, rstd, dx_in, dx, dx_bf16, dgamma, dbeta, dcolsum, workspace,
                            M, D, split_S, split_T, bf16_dropout_p, bf16_dropout_seed, dropout_counter, as_stream(stream));
}

extern "C" int ttts_ce_fwd_bf16(const void* logits, int64_t ldl, const int64_t* targets, float* row_loss,
                                float* row_lse, float* loss_mean, int32_t R, int32_t C, void* stream) {
  TTTS_REQUIRE(logits && targets && row_loss && row_lse && loss_mean, "ce_fwd: null pointer");
  TTTS_REQUIRE(R > 0 && C > 0 && ldl >= C && ldl % 8 == 0 && aligned16(logits), "ce_fwd: need ldl %% 8 == 0, ldl >= C");
  ce_fwd_kernel<<<(int)cdiv(R, 4), 256, 0, as_stream(stream)>>>((const bf16*)logits, ldl, targets, row_loss, row_lse, R, C);
  int rc = check_launch("ce_fwd");
  if (rc) return rc;
  mean_kernel<<<1, 1024, 0, as_stream(stream)>>>(row_loss, R, loss_mean);
  return check_launch("ce_mean");
}

extern "C" int ttts_ce_bwd_bf16(const void* logits, int64_t ldl, const int64_t* targets, const float* row_lse,
                                void* dlogits, float grad_scale, const float* grad_scale_dev, int32_t R, int32_t C,
                                void* stream) {
  TTTS_REQUIRE(logits && targets && row_lse && dlogits, "ce_bwd: null pointer");
  TTTS_REQUIRE(R > 0 && C > 0 && ldl >= C && ldl % 8 == 0 && aligned16(logits) && aligned16(dlogits), "ce_bwd: need ldl %% 8 == 0");
  ce_bwd_kernel<<<(int)cdiv(R, 4), 256, 0, as_stream(stream)>>>((const bf16*)logits, ldl, targets, row_lse,
                                                                (bf16*)dlogits, grad_scale, grad_scale_dev, R, C);
  return check_launch("ce_bwd");
}

extern "C" int ttts_colsum_bf16_accum_f32(const void* X, int64_t ldx, float* out, int32_t M, int32_t N, void* stream) {
  TTTS_REQUIRE(X && out && M > 0 && N > 0, "colsum: bad arguments");
  TTTS_REQUIRE(ldx % 8 == 0 && ldx >= ((N + 7) / 8) * 8 && aligned16(X), "colsum: need ldx %% 8 == 0 and ldx >= roundup8(N)");
  dim3 grid((unsigned)cdiv(N, 64), (unsigned)cdiv(M, COLSUM_ROWS));
  colsum_kernel<<<grid, 256, 0, as_stream(stream)>>>((const bf16*)X, ldx, out, M, N);
  return check_launch("colsum");
}

extern "C" int32_t ttts_colsum_desc_tiles(int32_t M, int32_t N) { return (int32_t)(cdiv(N, 64) * cdiv(M, COLSUM_ROWS)); }

extern "C" int ttts_colsum_bf16_accum_f32_batched(const ttts_colsum_desc* desc, int32_t n_desc, int32_t total_tiles, void* stream) {
  TTTS_REQUIRE(desc && n_desc > 0 && n_desc <= 64 && total_tiles > 0, "colsum_batched: need 1 .. 64 descriptors");
  colsum_batched_kernel<<<total_tiles, 256, 0, as_stream(stream)>>>(desc, n_desc);
  return check_launch("colsum_batched");
}

extern "C" int32_t ttts_transpose_desc_tiles(int32_t rows, int32_t cols) { return ((rows + 63) / 64) * ((cols + 63) / 64); }

extern "C" int ttts_transpose_bf16_batched(const ttts_transpose_desc* desc, int32_t n_desc, int32_t total_tiles, void* stream) {
  TTTS_REQUIRE(desc && n_desc > 0 && n_desc <= 64 && total_tiles > 0, "transpose_batched: need 1 .. 64 descriptors");
  transpose_bf16_batched_kernel<<<total_tiles, 256, 0, as_stream(stream)>>>(desc, n_desc);
  return check_launch("transpose_batched");
}

extern "C" int32_t ttts_cast_desc_tiles(int32_t rows, int32_t cols) { return ((rows + 31) / 32) * ((cols + 31) / 32); }

extern "C" int ttts_cast_bf16_batched(const ttts_cast_desc* desc, int32_t n_desc, int32_t total_tiles, void* stream) {
  TTTS_REQUIRE(desc && n_desc > 0 && total_tiles > 0, "cast_batched: bad arguments");
  cast_batched_kernel<<<total_tiles, 256, 0, as_stream(stream)>>>(desc, n_desc);
  return check_launch("cast_batched");
}

extern "C" int ttts_adamw_schedule(float* state, float base_lr, float beta1, float beta2, int32_t warmup_steps,
                                   void* stream) {
  TTTS_REQUIRE(state, "adamw_schedule: null state");
  adamw_schedule_kernel<<<1, 64, 0, as_stream(stream)>>>(state, base_lr, beta1, beta2, warmup_steps);
  return check_launch("adamw_schedule");
}

static int gradnorm_blocks(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>(1024, cdiv(n / 4, 256))); }
extern "C" int64_t ttts_gradnorm_workspace_bytes(int64_t n) { return (int64_t)gradnorm_blocks(n) * (int64_t)sizeof(double); }

extern "C" int ttts_gradnorm_f32(const float* g, int64_t n, float max_norm, float* state, void* workspace,
                                 void* stream) {
  TTTS_REQUIRE(g && state && workspace && n > 0 && aligned16(g), "gradnorm: bad arguments");
  const int nblk = gradnorm_blocks(n);
  sumsq_partial_kernel<<<nblk, 256, 0, as_stream(stream)>>>(g, n / 4, (double*)workspace);
  int rc = check_launch("gradnorm_partial");
  if (rc) return rc;
  gradnorm_final_kernel<<<1, 256, 0, as_stream(stream)>>>((const double*)workspace, nblk, g, n, (n / 4) * 4, max_norm, state);
  return check_launch("gradnorm_final");
}

extern "C" int ttts_adamw_f32(float* p, float* g, float* m, float* v, void* shadow_bf16, int64_t n, const float* state,
                              float beta1, float beta2, float eps, float weight_decay, int32_t zero_grad, void* stream) {
  TTTS_REQUIRE(p && g && m && v && state, "adamw: null pointer");
  TTTS_REQUIRE(n > 0 && n % 4 == 0 && aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), "adamw: n %% 4 == 0 and 16-byte alignment required");
  const int grid = (int)std::min<int64_t>(cdiv(n / 4, 256), 2048);
  adamw_kernel<<<grid, 256, 0, as_stream(stream)>>>(p, g, m, v, (bf16*)shadow_bf16, n / 4, state, beta1, beta2, eps,
                                                    weight_decay, zero_grad);
  return check_launch("adamw");
}
