// Pieces of the diffusion mel-denoiser step that the conv / attention families do not cover (SURVEY 8f row 3;
// ttts/diffusion/aa_model.py, ttts/utils/utils.py:113-215, ttts/utils/xtransformers.py:146-185, ttts/utils/diffusion.py):
//   groupnorm fwd/bwd  : GroupNorm32 (+ optional (1 + scale) * y + shift of the ResBlock's timestep embedding, + optional SiLU),
//                        one workgroup per (sample, group): a contiguous C/G x T block of the (B, C, T) tensor
//   relpos_bias        : T5-bucket relative position bias (H, Tq, Tk) from the (buckets, H) table, and its table gradient
//   softmax_bias       : in-place row softmax(S + bias) of the (B, H, Tq, Tk) score tensor
//   interp_nearest     : F.interpolate(mode='nearest') along time, forward and adjoint
//   timestep_embedding : sinusoidal embedding of the diffusion step
//   select_rows        : per-sample choice between the conditioning (B, C, T) and the learned unconditioned vector (C)
//   q_sample, diffusion_loss fwd/bwd : q(x_t | x_0) and the hybrid loss mse(eps) + learned-range variational bound
// All HBM-bound streaming kernels; reductions are fixed-order (no float atomics on results) except where noted.
#include <math.h>

#include <algorithm>

#include "common.hpp"

namespace ttts {

__device__ __forceinline__ float dsigmoid(float v) { return 1.f / (1.f + expf(-v)); }

__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// ---- GroupNorm ---------------------------------------------------------------------------------------------------------------
// y = act(((x - mean) rstd gamma + beta) * (1 + scale[b][c]) + shift[b][c]);  ss = [B][2C] (scale | shift) or NULL
__global__ __launch_bounds__(256) void groupnorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ ss,
                                                            float* __restrict__ y, float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out, int C, int T, int G, float eps, int act) {
  __shared__ float sh[4];
  const int b = blockIdx.x / G, g = blockIdx.x % G;
  const int Cg = C / G, n = Cg * T;
  const int64_t base = ((int64_t)b * C + g * Cg) * T;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += x[base + i];
  const float mean = block_sum(s, sh) / (float)n;
  float q = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) { const float d = x[base + i] - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(block_sum(q, sh) / (float)n + eps);
  if (threadIdx.x == 0) { mean_out[blockIdx.x] = mean; rstd_out[blockIdx.x] = rstd; }
  for (int i = threadIdx.x; i < n; i += 256) {
    const int c = g * Cg + i / T;
    float v = (x[base + i] - mean) * rstd * gamma[c] + beta[c];
    if (ss) v = v * (1.f + ss[(int64_t)b * 2 * C + c]) + ss[(int64_t)b * 2 * C + C + c];
    if (act) v = v * dsigmoid(v);
    y[base + i] = v;
  }
}

// dx, and per-(sample, channel) partial sums pg[b][c] (d gamma), pb[b][c] (d beta), dss[b][2C] (d scale | d shift)
__global__ __launch_bounds__(256) void groupnorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ ss, const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in, float* __restrict__ dx,
                                                            float* __restrict__ pg, float* __restrict__ pb,
                                                            float* __restrict__ dss, int C, int T, int G, int act) {
  __shared__ float sh[4];
  const int b = blockIdx.x / G, g = blockIdx.x % G;
  const int Cg = C / G, n = Cg * T;
  const int64_t base = ((int64_t)b * C + g * Cg) * T;
  const float mean = mean_in[blockIdx.x], rstd = rstd_in[blockIdx.x];
  float sum_dz = 0.f, sum_dzz = 0.f;      // group sums of dz and dz * z (thread-partial, reduced once at the end)
  // one channel per WAVE at a time: its four per-channel sums are wave reductions (the block-wide form cost four barrier pairs per
  // channel, 64+ per workgroup)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int cl = wave; cl < Cg; cl += 4) {
    const int c = g * Cg + cl;
    const float gm = gamma[c], bt = beta[c];
    const float sc = ss ? ss[(int64_t)b * 2 * C + c] : 0.f, sf = ss ? ss[(int64_t)b * 2 * C + C + c] : 0.f;
    float s_da = 0.f, s_daz = 0.f, s_dua = 0.f, s_du = 0.f;
    for (int t = lane; t < T; t += 64) {
      const int64_t o = base + (int64_t)cl * T + t;
      const float z = (x[o] - mean) * rstd;
      const float a = z * gm + bt;
      const float u = a * (1.f + sc) + sf;
      float du = dy[o];
      if (act) { const float sg = dsigmoid(u); du *= sg * (1.f + u * (1.f - sg)); }
      const float da = du * (1.f + sc);
      s_da += da; s_daz += da * z; s_dua += du * a; s_du += du;
      const float dz = da * gm;
      sum_dz += dz; sum_dzz += dz * z;
    }
    s_da = wave_sum(s_da); s_daz = wave_sum(s_daz);
    if (ss) { s_dua = wave_sum(s_dua); s_du = wave_sum(s_du); }
    if (lane == 0) {
      pg[(int64_t)b * C + c] = s_daz;
      pb[(int64_t)b * C + c] = s_da;
      if (ss) { dss[(int64_t)b * 2 * C + c] = s_dua; dss[(int64_t)b * 2 * C + C + c] = s_du; }
    }
  }
  const float m1 = block_sum(sum_dz, sh) / (float)n, m2 = block_sum(sum_dzz, sh) / (float)n;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int c = g * Cg + i / T;
    const float sc = ss ? ss[(int64_t)b * 2 * C + c] : 0.f, sf = ss ? ss[(int64_t)b * 2 * C + C + c] : 0.f;
    const float z = (x[base + i] - mean) * rstd;
    float du = dy[base + i];
    if (act) { const float u = (z * gamma[c] + beta[c]) * (1.f + sc) + sf; const float sg = dsigmoid(u); du *= sg * (1.f + u * (1.f - sg)); }
    const float dz = du * (1.f + sc) * gamma[c];
    dx[base + i] = rstd * (dz - m1 - z * m2);
  }
}

// (round 6) Register-resident forms for groups of up to 256 x 4 x GN_V elements (the diffusion model's 16 channels x <= 448 frames:
// 6 400).  The kernels above walk a group three times with 4-byte loads (sum, squared deviations, output) and wait for memory in each
// pass: 18 / 31 us forward / backward for 13 MB tensors whose one-pass traffic is 3.3 / 5 us.  Here a workgroup reads its group ONCE
// as float4s into registers, reduces twice (mean, then squared deviations of the held values -- the same two-pass arithmetic) and
// writes from registers.  Needs T % 4 == 0 (rows stay 16-byte aligned) and n / 4 <= 256 GN_V.
constexpr int GN_V = 8;
__global__ __launch_bounds__(256) void groupnorm_fwd_reg_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ ss,
                                                                float* __restrict__ y, float* __restrict__ mean_out,
                                                                float* __restrict__ rstd_out, int C, int T, int G, float eps, int act) {
  __shared__ float sh[4];
  const int b = blockIdx.x / G, g = blockIdx.x % G;
  const int Cg = C / G, n = Cg * T, n4 = n >> 2, T4 = T >> 2;
  const int64_t base = ((int64_t)b * C + g * Cg) * T;
  const float4* x4 = reinterpret_cast<const float4*>(x + base);
  float4 v[GN_V];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < GN_V; ++k) {
    const int i = threadIdx.x + 256 * k;
    v[k] = i < n4 ? x4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  }
  const float mean = block_sum(s, sh) / (float)n;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < GN_V; ++k) {
    if (threadIdx.x + 256 * k < n4) {
      const float d0 = v[k].x - mean, d1 = v[k].y - mean, d2 = v[k].z - mean, d3 = v[k].w - mean;
      q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  }
  const float rstd = 1.0f / sqrtf(block_sum(q, sh) / (float)n + eps);
  if (threadIdx.x == 0) { mean_out[blockIdx.x] = mean; rstd_out[blockIdx.x] = rstd; }
  float4* y4 = reinterpret_cast<float4*>(y + base);
#pragma unroll
  for (int k = 0; k < GN_V; ++k) {
    const int i = threadIdx.x + 256 * k;
    if (i >= n4) continue;
    const int c = g * Cg + i / T4;
    const float gmc = gamma[c], bt = beta[c];
    const float sc = ss ? 1.f + ss[(int64_t)b * 2 * C + c] : 1.f, sf = ss ? ss[(int64_t)b * 2 * C + C + c] : 0.f;
    float o[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float u = (o[e] - mean) * rstd * gmc + bt;
      if (ss) u = u * sc + sf;
      if (act) u = u * dsigmoid(u);
      o[e] = u;
    }
    y4[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// backward: x and dy held in registers (one channel per wave at a time, as above: the per-channel sums are wave reductions), dx
// written from them after the two group sums are known
constexpr int GN_BW = 7;       // elements per lane and channel held (T <= 64 * GN_BW = 448)
constexpr int GN_BC = 4;       // channels per wave held (Cg <= 4 * GN_BC = 16)
__global__ __launch_bounds__(256) void groupnorm_bwd_reg_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ ss, const float* __restrict__ mean_in,
                                                                const float* __restrict__ rstd_in, float* __restrict__ dx,
                                                                float* __restrict__ pg, float* __restrict__ pb,
                                                                float* __restrict__ dss, int C, int T, int G, int act) {
  __shared__ float sh[4];
  const int b = blockIdx.x / G, g = blockIdx.x % G;
  const int Cg = C / G, n = Cg * T;
  const int64_t base = ((int64_t)b * C + g * Cg) * T;
  const float mean = mean_in[blockIdx.x], rstd = rstd_in[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float zr[GN_BC][GN_BW], dzr[GN_BC][GN_BW];      // z and dz of this lane's elements
  float sum_dz = 0.f, sum_dzz = 0.f;
#pragma unroll
  for (int q = 0; q < GN_BC; ++q) {
    const int cl = wave + 4 * q;
    const bool cok = cl < Cg;
    const int c = g * Cg + min(cl, Cg - 1);
    const float gm = gamma[c], bt = beta[c];
    const float sc = ss ? ss[(int64_t)b * 2 * C + c] : 0.f, sf = ss ? ss[(int64_t)b * 2 * C + C + c] : 0.f;
    float xv[GN_BW], dv[GN_BW];
#pragma unroll
    for (int e = 0; e < GN_BW; ++e) {             // requests first (clamped addresses, selected below)
      const int64_t o = base + (int64_t)min(cl, Cg - 1) * T + min(lane + 64 * e, T - 1);
      xv[e] = x[o]; dv[e] = dy[o];
    }
    float s_da = 0.f, s_daz = 0.f, s_dua = 0.f, s_du = 0.f;
#pragma unroll
    for (int e = 0; e < GN_BW; ++e) {
      const bool ok = cok && lane + 64 * e < T;
      const float z = (xv[e] - mean) * rstd;
      const float a = z * gm + bt;
      const float u = a * (1.f + sc) + sf;
      float du = ok ? dv[e] : 0.f;
      if (act) { const float sg = dsigmoid(u); du *= sg * (1.f + u * (1.f - sg)); }
      const float da = du * (1.f + sc);
      s_da += da; s_daz += da * z; s_dua += du * a; s_du += du;
      const float dz = da * gm;
      sum_dz += dz; sum_dzz += dz * z;
      zr[q][e] = z; dzr[q][e] = dz;
    }
    s_da = wave_sum(s_da); s_daz = wave_sum(s_daz);
    if (ss) { s_dua = wave_sum(s_dua); s_du = wave_sum(s_du); }
    if (lane == 0 && cok) {
      pg[(int64_t)b * C + c] = s_daz;
      pb[(int64_t)b * C + c] = s_da;
      if (ss) { dss[(int64_t)b * 2 * C + c] = s_dua; dss[(int64_t)b * 2 * C + C + c] = s_du; }
    }
  }
  const float m1 = block_sum(sum_dz, sh) / (float)n, m2 = block_sum(sum_dzz, sh) / (float)n;
#pragma unroll
  for (int q = 0; q < GN_BC; ++q) {
    const int cl = wave + 4 * q;
    if (cl >= Cg) continue;
#pragma unroll
    for (int e = 0; e < GN_BW; ++e) {
      const int t = lane + 64 * e;
      if (t < T) dx[base + (int64_t)cl * T + t] = rstd * (dzr[q][e] - m1 - zr[q][e] * m2);
    }
  }
}

// out[c] (+)= sum_r in[r][c] for TWO arrays in one launch (the d gamma / d beta partials of a GroupNorm backward)
__global__ __launch_bounds__(256) void reduce_rows2_kernel(const float* __restrict__ in0, float* __restrict__ out0,
                                                           const float* __restrict__ in1, float* __restrict__ out1, int R, int C,
                                                           int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float* in = blockIdx.y ? in1 : in0;
  float* out = blockIdx.y ? out1 : out0;
  float s = 0.f;
  for (int r = 0; r < R; ++r) s += in[(int64_t)r * C + c];
  out[c] = accumulate ? out[c] + s : s;
}

// out[c] (+)= sum_r in[r][c]   (fixed order)
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C,
                                                          int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int r = 0; r < R; ++r) s += in[(int64_t)r * C + c];
  out[c] = accumulate ? out[c] + s : s;
}

// ---- relative position bias -----------------------------------------------------------------------------------------------------
// bias[h][i][j] = table[bucket[j - i + off]][h] * scale;  bucket: int32 [2 * off + 1] (host-built with the reference's formula)
__global__ __launch_bounds__(256) void relpos_bias_fwd_kernel(const float* __restrict__ table, const int32_t* __restrict__ bucket,
                                                              float* __restrict__ bias, int H, int Tq, int Tk, int off, float scale) {
  const int64_t n = (int64_t)H * Tq * Tk;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int j = (int)(e % Tk), i = (int)((e / Tk) % Tq), h = (int)(e / ((int64_t)Tk * Tq));
    bias[e] = table[bucket[j - i + off] * H + h] * scale;
  }
}
// d table[bucket][h] = scale * sum_{b, i, j} dS[b][h][i][j] [bucket(j - i) == bucket]: the bias only depends on r = j - i, so
//  1. relpos_diag_kernel: per (h, b, 64-row chunk) the diagonal sums D[r] = sum_i dS[i][i + r] -- thread-private accumulators,
//     coalesced reads, no atomics -> partial[blk][Tq + Tk - 1]
//  2. relpos_diag_reduce_kernel: sum the partials over (b, chunk) in fixed order -> D[h][r]
//  3. relpos_bucket_kernel: 32 bucket sums over r (LDS atomics inside one block: ~1 ulp ordering noise) -> dtable
constexpr int RP_ROWS = 64;
__global__ __launch_bounds__(256) void relpos_diag_kernel(const float* __restrict__ dS, float* __restrict__ partial, int B, int H,
                                                          int Tq, int Tk, int chunks) {
  const int ck = blockIdx.x % chunks, b = (blockIdx.x / chunks) % B, h = blockIdx.x / (chunks * B);
  const int nR = Tq + Tk - 1;
  const int i0 = ck * RP_ROWS, i1 = min(i0 + RP_ROWS, Tq);
  const float* base = dS + ((int64_t)b * H + h) * Tq * Tk;
  for (int ri = threadIdx.x; ri < nR; ri += 256) {
    const int r = ri - (Tq - 1);
    float acc = 0.f;
    const int lo = max(i0, -r), hi = min(i1, Tk - r);      // rows with 0 <= i + r < Tk
    for (int i = lo; i < hi; ++i) acc += base[(int64_t)i * Tk + i + r];
    partial[(int64_t)blockIdx.x * nR + ri] = acc;
  }
}
__global__ __launch_bounds__(256) void relpos_diag_reduce_kernel(const float* __restrict__ partial, float* __restrict__ D, int nR,
                                                                 int per_head) {
  const int h = blockIdx.y, ri = blockIdx.x * 256 + threadIdx.x;
  if (ri >= nR) return;
  float s = 0.f;
  for (int k = 0; k < per_head; ++k) s += partial[((int64_t)h * per_head + k) * nR + ri];
  D[(int64_t)h * nR + ri] = s;
}
__global__ __launch_bounds__(256) void relpos_bucket_kernel(const float* __restrict__ D, const int32_t* __restrict__ bucket,
                                                            float* __restrict__ dtable, int H, int Tq, int nR, int off, int NB,
                                                            float scale, int accumulate) {
  __shared__ float bins[64];
  const int h = blockIdx.x;
  if (threadIdx.x < 64) bins[threadIdx.x] = 0.f;
  __syncthreads();
  for (int ri = threadIdx.x; ri < nR; ri += 256) atomicAdd(&bins[bucket[ri - (Tq - 1) + off]], D[(int64_t)h * nR + ri]);
  __syncthreads();
  if (threadIdx.x < NB) dtable[threadIdx.x * H + h] = (accumulate ? dtable[threadIdx.x * H + h] : 0.f) + bins[threadIdx.x] * scale;
}

// S[b][h][i][:] = softmax(S + bias[h][i][:]) in place; one wave per row
__global__ __launch_bounds__(256) void softmax_bias_fwd_kernel(float* __restrict__ S, const float* __restrict__ bias, int B, int H,
                                                               int Tq, int Tk) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= (int64_t)B * H * Tq) return;
  const int i = (int)(row % Tq), h = (int)((row / Tq) % H);
  float* s = S + row * Tk;
  const float* bs = bias ? bias + ((int64_t)h * Tq + i) * Tk : nullptr;
  float mx = -INFINITY;
  for (int j = lane; j < Tk; j += 64) { const float v = s[j] + (bs ? bs[j] : 0.f); s[j] = v; mx = fmaxf(mx, v); }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < Tk; j += 64) { const float e = expf(s[j] - mx); s[j] = e; sum += e; }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  for (int j = lane; j < Tk; j += 64) s[j] *= inv;
}

// ---- nearest interpolation along time -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void interp_nearest_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t rows,
                                                                 int Tin, int Tout) {
  const int64_t n = rows * Tout;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int j = (int)(e % Tout);
    const int64_t r = e / Tout;
    const int src = min((int)(((int64_t)j * Tin) / Tout), Tin - 1);
    y[e] = x[r * Tin + src];
  }
}
__global__ __launch_bounds__(256) void interp_nearest_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int64_t rows,
                                                                 int Tin, int Tout) {
  const int64_t n = rows * Tin;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int i = (int)(e % Tin);
    const int64_t r = e / Tin;
    // outputs j with floor(j Tin / Tout) == i:  j in [ceil(i Tout / Tin), ceil((i + 1) Tout / Tin))
    const int j0 = (int)(((int64_t)i * Tout + Tin - 1) / Tin), j1 = min((int)(((int64_t)(i + 1) * Tout + Tin - 1) / Tin), Tout);
    float s = 0.f;
    for (int j = j0; j < j1; ++j) s += dy[r * Tout + j];
    dx[e] = s;
  }
}

// emb[n][k] = cos(t_n f_k), emb[n][half + k] = sin(t_n f_k) with the host-built frequency table f_k = exp(-ln(max_period) k / half)
// (aa_model.py:32-51; the table comes from the same fp32 torch expression the reference evaluates, so the arguments t f are
// bit-identical and only cos / sin themselves can differ)
__global__ __launch_bounds__(256) void timestep_embedding_kernel(const int64_t* __restrict__ t, const float* __restrict__ freqs,
                                                                 float* __restrict__ emb, int N, int dim) {
  const int half = dim / 2;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= N * dim) return;
  const int n = e / dim, k = e % dim;
  if (k >= 2 * half) { emb[e] = 0.f; return; }
  const float a = (float)t[n] * freqs[k < half ? k : k - half];
  emb[e] = k < half ? cosf(a) : sinf(a);
}

// out[b] = use[b] ? vec (broadcast over T) : a[b];  backward: da = use ? 0 : dout, dvec[c] += sum_{b: use, t} dout
__global__ __launch_bounds__(256) void select_rows_fwd_kernel(const uint8_t* __restrict__ use, const float* __restrict__ a,
                                                              const float* __restrict__ vec, float* __restrict__ out, int B, int C,
                                                              int T) {
  const int64_t n = (int64_t)B * C * T;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int c = (int)((e / T) % C), b = (int)(e / ((int64_t)T * C));
    out[e] = use[b] ? vec[c] : a[e];
  }
}
__global__ __launch_bounds__(256) void select_rows_bwd_kernel(const uint8_t* __restrict__ use, const float* __restrict__ dout,
                                                              float* __restrict__ da, float* __restrict__ dvec, int B, int C, int T,
                                                              int accumulate) {
  __shared__ float sh[4];
  const int c = blockIdx.x;
  float s = 0.f;
  for (int b = 0; b < B; ++b) {
    const int64_t o = ((int64_t)b * C + c) * T;
    for (int t = threadIdx.x; t < T; t += 256) {
      const float g = dout[o + t];
      if (da) da[o + t] = use[b] ? 0.f : g;
      if (use[b]) s += g;
    }
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0 && dvec) dvec[c] = (accumulate ? dvec[c] : 0.f) + s;
}

// ---- Gaussian diffusion ---------------------------------------------------------------------------------------------------------
// coefficient table tab[step][8] (fp32, host-built in float64 as the reference does):
//  0 sqrt_alphas_cumprod  1 sqrt_one_minus_alphas_cumprod  2 sqrt_recip_alphas_cumprod  3 sqrt_recipm1_alphas_cumprod
//  4 posterior_mean_coef1  5 posterior_mean_coef2  6 posterior_log_variance_clipped  7 log(betas)
__global__ __launch_bounds__(256) void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                                       const int64_t* __restrict__ t, const float* __restrict__ tab,
                                                       float* __restrict__ xt, int B, int64_t per) {
  const int64_t n = (int64_t)B * per;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int b = (int)(e / per);
    const float* cf = tab + t[b] * 8;
    xt[e] = cf[0] * x0[e] + cf[1] * noise[e];
  }
}

__device__ __forceinline__ float approx_cdf(float x) { return 0.5f * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x))); }
__device__ __forceinline__ float approx_cdf_grad(float x) {
  const float th = tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x));
  return 0.5f * (1.f - th * th) * 0.7978845608028654f * (1.f + 3.f * 0.044715f * x * x);
}

// model_out [B][2C][T] = (eps | v).  Per sample: mse = mean (noise - eps)^2;  vb = (t == 0 ? decoder NLL : KL) / ln 2 with the
// mean built from the DETACHED eps (diffusion.py:980).  Writes partial sums [B][blocks][2]; grad kernel below.
__global__ __launch_bounds__(256) void diffusion_loss_fwd_kernel(const float* __restrict__ mo, const float* __restrict__ x0,
                                                                 const float* __restrict__ xt, const float* __restrict__ noise,
                                                                 const int64_t* __restrict__ t, const float* __restrict__ tab,
                                                                 float* __restrict__ partial, int C, int T, int blocks) {
  __shared__ float sh[4];
  const int b = blockIdx.x / blocks, blk = blockIdx.x % blocks;
  const int64_t per = (int64_t)C * T;
  const float* cf = tab + t[b] * 8;
  const int tb = (int)t[b];
  float s_mse = 0.f, s_vb = 0.f;
  for (int64_t e = (int64_t)blk * 256 + threadIdx.x; e < per; e += (int64_t)blocks * 256) {
    const int64_t ox = (int64_t)b * per + e;
    const float eps = mo[(int64_t)b * 2 * per + e], v = mo[(int64_t)b * 2 * per + per + e];
    const float d = noise[ox] - eps;
    s_mse += d * d;
    const float frac = (v + 1.f) * 0.5f;
    const float logvar = frac * cf[7] + (1.f - frac) * cf[6];
    float px0 = cf[2] * xt[ox] - cf[3] * eps;
    px0 = fminf(fmaxf(px0, -1.f), 1.f);
    const float mean = cf[4] * px0 + cf[5] * xt[ox];
    const float x = x0[ox];
    if (tb == 0) {
      const float inv_std = expf(-0.5f * logvar), cen = x - mean;
      const float cp = approx_cdf(inv_std * (cen + 1.0f / 255.0f)), cm = approx_cdf(inv_std * (cen - 1.0f / 255.0f));
      float lp;
      if (x < -0.999f) lp = logf(fmaxf(cp, 1e-12f));
      else if (x > 0.999f) lp = logf(fmaxf(1.f - cm, 1e-12f));
      else lp = logf(fmaxf(cp - cm, 1e-12f));
      s_vb -= lp;
    } else {
      const float tm = cf[4] * x + cf[5] * xt[ox];
      const float dm = tm - mean;
      s_vb += 0.5f * (-1.0f + logvar - cf[6] + expf(cf[6] - logvar) + dm * dm * expf(-logvar));
    }
  }
  s_mse = block_sum(s_mse, sh); s_vb = block_sum(s_vb, sh);
  if (threadIdx.x == 0) { partial[(int64_t)blockIdx.x * 2] = s_mse; partial[(int64_t)blockIdx.x * 2 + 1] = s_vb; }
}
// terms[b] = (mse, vb, loss = mse + vb);  loss_mean (+)= mean_b loss
__global__ __launch_bounds__(64) void diffusion_loss_finish_kernel(const float* __restrict__ partial, float* __restrict__ terms,
                                                                   float* __restrict__ loss_mean, int B, int blocks, float inv_per) {
  float total = 0.f;
  if (threadIdx.x == 0) {
    for (int b = 0; b < B; ++b) {
      float m = 0.f, v = 0.f;
      for (int k = 0; k < blocks; ++k) { m += partial[((int64_t)b * blocks + k) * 2]; v += partial[((int64_t)b * blocks + k) * 2 + 1]; }
      m *= inv_per; v *= inv_per * 1.4426950408889634f;      // / ln 2
      terms[b * 3] = m; terms[b * 3 + 1] = v; terms[b * 3 + 2] = m + v;
      total += m + v;
    }
    loss_mean[0] = total / (float)B;
  }
}
// d loss_mean / d model_out (scaled by gout[0] if given)
__global__ __launch_bounds__(256) void diffusion_loss_bwd_kernel(const float* __restrict__ mo, const float* __restrict__ x0,
                                                                 const float* __restrict__ xt, const float* __restrict__ noise,
                                                                 const int64_t* __restrict__ t, const float* __restrict__ tab,
                                                                 const float* __restrict__ gout, float* __restrict__ dmo, int B, int C,
                                                                 int T) {
  const int64_t per = (int64_t)C * T, n = (int64_t)B * per;
  const float g = (gout ? gout[0] : 1.f) / ((float)B * (float)per);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int b = (int)(i / per);
    const int64_t e = i % per;
    const float* cf = tab + t[b] * 8;
    const float eps = mo[(int64_t)b * 2 * per + e], v = mo[(int64_t)b * 2 * per + per + e];
    dmo[(int64_t)b * 2 * per + e] = -2.f * (noise[i] - eps) * g;            // mse only: the vb term sees eps detached
    const float frac = (v + 1.f) * 0.5f;
    const float logvar = frac * cf[7] + (1.f - frac) * cf[6];
    float px0 = cf[2] * xt[i] - cf[3] * eps;
    px0 = fminf(fmaxf(px0, -1.f), 1.f);
    const float mean = cf[4] * px0 + cf[5] * xt[i];
    const float x = x0[i];
    float dlv;   // d (per-element vb term in nats) / d logvar
    if ((int)t[b] == 0) {
      const float inv_std = expf(-0.5f * logvar), cen = x - mean;
      const float ap = inv_std * (cen + 1.0f / 255.0f), am = inv_std * (cen - 1.0f / 255.0f);
      const float cp = approx_cdf(ap), cm = approx_cdf(am);
      // d a / d logvar = -a / 2
      const float dcp = approx_cdf_grad(ap) * (-0.5f * ap), dcm = approx_cdf_grad(am) * (-0.5f * am);
      if (x < -0.999f) dlv = cp > 1e-12f ? -dcp / cp : 0.f;
      else if (x > 0.999f) dlv = (1.f - cm) > 1e-12f ? dcm / (1.f - cm) : 0.f;
      else dlv = (cp - cm) > 1e-12f ? -(dcp - dcm) / (cp - cm) : 0.f;
    } else {
      const float tm = cf[4] * x + cf[5] * xt[i];
      const float dm = tm - mean;
      dlv = 0.5f * (1.0f - expf(cf[6] - logvar) - dm * dm * expf(-logvar));
    }
    dmo[(int64_t)b * 2 * per + per + e] = dlv * 0.5f * (cf[7] - cf[6]) * 1.4426950408889634f * g;
  }
}

}  // namespace ttts

using namespace ttts;

static inline int grid1d(int64_t n) { return (int)std::min<int64_t>(cdiv(n, 256), 8192); }

extern "C" int ttts_groupnorm_fwd_f32(const float* x, const float* gamma, const float* beta, const float* scale_shift, float* y,
                                      float* mean, float* rstd, int32_t B, int32_t C, int32_t T, int32_t groups, float eps,
                                      int32_t silu, void* stream) {
  TTTS_REQUIRE(x && gamma && beta && y && mean && rstd && B > 0 && C > 0 && T > 0, "groupnorm_fwd: bad arguments");
  TTTS_REQUIRE(groups > 0 && C % groups == 0, "groupnorm_fwd: C %% groups != 0");
  const int n = C / groups * T;
  if (T % 4 == 0 && n / 4 <= 256 * GN_V && (reinterpret_cast<uintptr_t>(x) & 15u) == 0 && (reinterpret_cast<uintptr_t>(y) & 15u) == 0)
    groupnorm_fwd_reg_kernel<<<B * groups, 256, 0, as_stream(stream)>>>(x, gamma, beta, scale_shift, y, mean, rstd, C, T, groups, eps, silu);
  else
    groupnorm_fwd_kernel<<<B * groups, 256, 0, as_stream(stream)>>>(x, gamma, beta, scale_shift, y, mean, rstd, C, T, groups, eps, silu);
  return check_launch("groupnorm_fwd");
}

extern "C" int ttts_groupnorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* beta,
                                      const float* scale_shift, const float* mean, const float* rstd, float* dx, float* dgamma,
                                      float* dbeta, float* d_scale_shift, float* workspace, int32_t B, int32_t C, int32_t T,
                                      int32_t groups, int32_t silu, int32_t accumulate, void* stream) {
  TTTS_REQUIRE(dy && x && gamma && beta && mean && rstd && dx && dgamma && dbeta && workspace, "groupnorm_bwd: null pointer");
  TTTS_REQUIRE(B > 0 && C > 0 && T > 0 && groups > 0 && C % groups == 0, "groupnorm_bwd: bad sizes");
  TTTS_REQUIRE(!scale_shift || d_scale_shift, "groupnorm_bwd: d_scale_shift missing");
  float* pg = workspace;                 // [B][C]
  float* pb = workspace + (int64_t)B * C;
  if (T <= 64 * GN_BW && C / groups <= 4 * GN_BC)
    groupnorm_bwd_reg_kernel<<<B * groups, 256, 0, as_stream(stream)>>>(dy, x, gamma, beta, scale_shift, mean, rstd, dx, pg, pb,
                                                                       d_scale_shift, C, T, groups, silu);
  else
    groupnorm_bwd_kernel<<<B * groups, 256, 0, as_stream(stream)>>>(dy, x, gamma, beta, scale_shift, mean, rstd, dx, pg, pb, d_scale_shift,
                                                                   C, T, groups, silu);
  reduce_rows2_kernel<<<dim3((unsigned)cdiv(C, 256), 2), 256, 0, as_stream(stream)>>>(pg, dgamma, pb, dbeta, B, C, accumulate);
  return check_launch("groupnorm_bwd");
}

extern "C" int ttts_relpos_bias_fwd_f32(const float* table, const int32_t* bucket, float* bias, int32_t H, int32_t Tq, int32_t Tk,
                                        int32_t bucket_offset, float scale, void* stream) {
  TTTS_REQUIRE(table && bucket && bias && H > 0 && Tq > 0 && Tk > 0, "relpos_bias_fwd: bad arguments");
  TTTS_REQUIRE(bucket_offset >= Tq - 1 && bucket_offset >= Tk - 1, "relpos_bias_fwd: bucket table too short");
  relpos_bias_fwd_kernel<<<grid1d((int64_t)H * Tq * Tk), 256, 0, as_stream(stream)>>>(table, bucket, bias, H, Tq, Tk, bucket_offset, scale);
  return check_launch("relpos_bias_fwd");
}

extern "C" int64_t ttts_relpos_bias_bwd_workspace_bytes(int32_t B, int32_t H, int32_t Tq, int32_t Tk) {
  const int64_t nR = (int64_t)Tq + Tk - 1;
  return ((int64_t)H * B * cdiv(Tq, RP_ROWS) + H) * nR * (int64_t)sizeof(float);
}

extern "C" int ttts_relpos_bias_bwd_f32(const float* dS, const int32_t* bucket, float* dtable, float* workspace, int32_t B, int32_t H,
                                        int32_t Tq, int32_t Tk, int32_t bucket_offset, int32_t num_buckets, float scale,
                                        int32_t accumulate, void* stream) {
  TTTS_REQUIRE(dS && bucket && dtable && workspace && B > 0 && H > 0 && Tq > 0 && Tk > 0, "relpos_bias_bwd: bad arguments");
  TTTS_REQUIRE(num_buckets > 0 && num_buckets <= 64, "relpos_bias_bwd: at most 64 buckets");
  TTTS_REQUIRE(bucket_offset >= Tq - 1 && bucket_offset >= Tk - 1, "relpos_bias_bwd: bucket table too short");
  const int chunks = (int)cdiv(Tq, RP_ROWS), nR = Tq + Tk - 1;
  float* D = workspace + (int64_t)H * B * chunks * nR;
  relpos_diag_kernel<<<H * B * chunks, 256, 0, as_stream(stream)>>>(dS, workspace, B, H, Tq, Tk, chunks);
  relpos_diag_reduce_kernel<<<dim3((unsigned)cdiv(nR, 256), (unsigned)H), 256, 0, as_stream(stream)>>>(workspace, D, nR, B * chunks);
  relpos_bucket_kernel<<<H, 256, 0, as_stream(stream)>>>(D, bucket, dtable, H, Tq, nR, bucket_offset, num_buckets, scale, accumulate);
  return check_launch("relpos_bias_bwd");
}

extern "C" int ttts_softmax_bias_fwd_f32(float* scores, const float* bias, int32_t B, int32_t H, int32_t Tq, int32_t Tk, void* stream) {
  TTTS_REQUIRE(scores && B > 0 && H > 0 && Tq > 0 && Tk > 0, "softmax_bias_fwd: bad arguments");
  softmax_bias_fwd_kernel<<<(int)cdiv((int64_t)B * H * Tq, 4), 256, 0, as_stream(stream)>>>(scores, bias, B, H, Tq, Tk);
  return check_launch("softmax_bias_fwd");
}

extern "C" int ttts_interp_nearest_fwd_f32(const float* x, float* y, int64_t rows, int32_t Tin, int32_t Tout, void* stream) {
  TTTS_REQUIRE(x && y && rows > 0 && Tin > 0 && Tout > 0, "interp_nearest_fwd: bad arguments");
  interp_nearest_fwd_kernel<<<grid1d(rows * Tout), 256, 0, as_stream(stream)>>>(x, y, rows, Tin, Tout);
  return check_launch("interp_nearest_fwd");
}
extern "C" int ttts_interp_nearest_bwd_f32(const float* dy, float* dx, int64_t rows, int32_t Tin, int32_t Tout, void* stream) {
  TTTS_REQUIRE(dy && dx && rows > 0 && Tin > 0 && Tout > 0, "interp_nearest_bwd: bad arguments");
  interp_nearest_bwd_kernel<<<grid1d(rows * Tin), 256, 0, as_stream(stream)>>>(dy, dx, rows, Tin, Tout);
  return check_launch("interp_nearest_bwd");
}

extern "C" int ttts_timestep_embedding_f32(const int64_t* t, const float* freqs, float* emb, int32_t N, int32_t dim, void* stream) {
  TTTS_REQUIRE(t && freqs && emb && N > 0 && dim > 1, "timestep_embedding: bad arguments");
  timestep_embedding_kernel<<<(int)cdiv((int64_t)N * dim, 256), 256, 0, as_stream(stream)>>>(t, freqs, emb, N, dim);
  return check_launch("timestep_embedding");
}

extern "C" int ttts_select_rows_fwd_f32(const uint8_t* use_vec, const float* a, const float* vec, float* out, int32_t B, int32_t C,
                                        int32_t T, void* stream) {
  TTTS_REQUIRE(use_vec && a && vec && out && B > 0 && C > 0 && T > 0, "select_rows_fwd: bad arguments");
  select_rows_fwd_kernel<<<grid1d((int64_t)B * C * T), 256, 0, as_stream(stream)>>>(use_vec, a, vec, out, B, C, T);
  return check_launch("select_rows_fwd");
}
extern "C" int ttts_select_rows_bwd_f32(const uint8_t* use_vec, const float* dout, float* da, float* dvec, int32_t B, int32_t C,
                                        int32_t T, int32_t accumulate, void* stream) {
  TTTS_REQUIRE(use_vec && dout && B > 0 && C > 0 && T > 0, "select_rows_bwd: bad arguments");
  select_rows_bwd_kernel<<<C, 256, 0, as_stream(stream)>>>(use_vec, dout, da, dvec, B, C, T, accumulate);
  return check_launch("select_rows_bwd");
}

extern "C" int ttts_q_sample_f32(const float* x_start, const float* noise, const int64_t* t, const float* table, float* x_t, int32_t B,
                                 int64_t per_sample, void* stream) {
  TTTS_REQUIRE(x_start && noise && t && table && x_t && B > 0 && per_sample > 0, "q_sample: bad arguments");
  q_sample_kernel<<<grid1d((int64_t)B * per_sample), 256, 0, as_stream(stream)>>>(x_start, noise, t, table, x_t, B, per_sample);
  return check_launch("q_sample");
}

constexpr int DLOSS_BLOCKS = 32;
extern "C" int64_t ttts_diffusion_loss_workspace_bytes(int32_t B) { return (int64_t)B * DLOSS_BLOCKS * 2 * (int64_t)sizeof(float); }

extern "C" int ttts_diffusion_loss_fwd_f32(const float* model_out, const float* x_start, const float* x_t, const float* noise,
                                           const int64_t* t, const float* table, float* terms, float* loss_mean, float* workspace,
                                           int32_t B, int32_t C, int32_t T, void* stream) {
  TTTS_REQUIRE(model_out && x_start && x_t && noise && t && table && terms && loss_mean && workspace, "diffusion_loss_fwd: null pointer");
  TTTS_REQUIRE(B > 0 && C > 0 && T > 0, "diffusion_loss_fwd: bad sizes");
  diffusion_loss_fwd_kernel<<<B * DLOSS_BLOCKS, 256, 0, as_stream(stream)>>>(model_out, x_start, x_t, noise, t, table, workspace, C, T, DLOSS_BLOCKS);
  diffusion_loss_finish_kernel<<<1, 64, 0, as_stream(stream)>>>(workspace, terms, loss_mean, B, DLOSS_BLOCKS, 1.0f / ((float)C * (float)T));
  return check_launch("diffusion_loss_fwd");
}

extern "C" int ttts_diffusion_loss_bwd_f32(const float* model_out, const float* x_start, const float* x_t, const float* noise,
                                           const int64_t* t, const float* table, const float* gout, float* d_model_out, int32_t B,
                                           int32_t C, int32_t T, void* stream) {
  TTTS_REQUIRE(model_out && x_start && x_t && noise && t && table && d_model_out && B > 0 && C > 0 && T > 0, "diffusion_loss_bwd: bad arguments");
  diffusion_loss_bwd_kernel<<<grid1d((int64_t)B * C * T), 256, 0, as_stream(stream)>>>(model_out, x_start, x_t, noise, t, table, gout, d_model_out, B, C, T);
  return check_launch("diffusion_loss_bwd");
}
