// Elementwise / small-reduction kernels of the VQ-VAE-GAN generator stacks (all fp32, (B, C, T) layout, HBM-bound):
//   * gated activations: WaveNet tanh*sigmoid (commons.fused_add_tanh_sigmoid_multiply, ttts/utils/commons.py:103-109)
//     and GLU (Conv1dGLU, ttts/vqvae/modules.py)
//   * sequence-mask multiply, reparameterised Gaussian sample (vq2.py:742-744), nearest x2 upsample (vq2.py:853-855)
//   * anti-aliased SnakeBeta: Activation1d(SnakeBeta) = kaiser-sinc x2 upsample -> x + sin^2(a x)/b -> low-pass x2
//     downsample (alias_free_torch/act.py:8-28, resample.py, filter.py; activations.py:62-119) as ONE kernel per row
//   * pointwise activations relu / mish, dropout with a regenerated hash mask
//   * channel LayerNorm of the attention stacks (modules.LayerNorm, ttts/vqvae/modules.py:19-31: normalise over C at
//     every (b, t))
#include <algorithm>

#include "common.hpp"

namespace ttts {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// ---- gates ----------------------------------------------------------------------------------------------------------
// x [B, 2H, T] -> y [B, H, T];  kind 0: tanh(a) * sigmoid(b),  kind 1: a * sigmoid(b)   (a = first H channels)
__global__ __launch_bounds__(256) void gate_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H,
                                                       int T, int kind) {
  const int64_t n = (int64_t)B * H * T, ht = (int64_t)H * T;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / ht, r = i % ht;
    const float a = x[b * 2 * ht + r], g = x[b * 2 * ht + ht + r];
    y[i] = (kind == 0 ? tanhf(a) : a) * sigmoidf_(g);
  }
}
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                       float* __restrict__ dx, int B, int H, int T, int kind) {
  const int64_t n = (int64_t)B * H * T, ht = (int64_t)H * T;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / ht, r = i % ht;
    const float a = x[b * 2 * ht + r], g = x[b * 2 * ht + ht + r], d = dy[i];
    const float s = sigmoidf_(g);
    if (kind == 0) {
      const float t = tanhf(a);
      dx[b * 2 * ht + r] = d * s * (1.f - t * t);
      dx[b * 2 * ht + ht + r] = d * t * s * (1.f - s);
    } else {
      dx[b * 2 * ht + r] = d * s;
      dx[b * 2 * ht + ht + r] = d * a * s * (1.f - s);
    }
  }
}

// gate_bwd + the per-(sample, channel) sums of dx over t (round 6: the WaveNet layers' conditioning gradient d g[b][c] = sum_t
// d x_in[b][c][t] was a separate pass over dx -- 12 288 one-row workgroups, 16 us -- behind every gate_bwd of a conditioned stack).
// One wave per (b, r): its a-row and g-row of dy / x, the two row sums by wave reduction -> rowsum[b][2H] (plain stores, fixed order).
__global__ __launch_bounds__(256) void gate_bwd_rowsum_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              float* __restrict__ dx, float* __restrict__ rowsum, int B, int H,
                                                              int T, int kind) {
  const int lane = threadIdx.x & 63;
  const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= (int64_t)B * H) return;                      // (whole waves)
  const int64_t b = pair / H, r = pair % H;
  const float* xa = x + (b * 2 * H + r) * T;
  const float* xg = xa + (int64_t)H * T;
  const float* d = dy + (b * H + r) * T;
  float* da = dx + (b * 2 * H + r) * T;
  float* dg = da + (int64_t)H * T;
  float sa = 0.f, sg = 0.f;
  for (int t = lane; t < T; t += 64) {
    const float a = xa[t], g = xg[t], dv = d[t];
    const float s = sigmoidf_(g);
    float va, vg;
    if (kind == 0) {
      const float th = tanhf(a);
      va = dv * s * (1.f - th * th);
      vg = dv * th * s * (1.f - s);
    } else {
      va = dv * s;
      vg = dv * a * s * (1.f - s);
    }
    da[t] = va; dg[t] = vg;
    sa += va; sg += vg;
  }
  sa = wave_sum(sa); sg = wave_sum(sg);
  if (lane == 0) { rowsum[b * 2 * H + r] = sa; rowsum[b * 2 * H + H + r] = sg; }
}

// y[b][c][t] = x[b][c][t] * m[b][t]
__global__ __launch_bounds__(256) void mul_mask_kernel(const float* __restrict__ x, const float* __restrict__ m,
                                                       float* __restrict__ y, int B, int C, int T) {
  const int64_t n = (int64_t)B * C * T;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int t = (int)(i % T);
    const int64_t b = i / ((int64_t)C * T);
    y[i] = x[i] * m[b * T + t];
  }
}

// stats [B, 2C, T] = (m | logs);  z = (m + eps * exp(logs)) * mask
__global__ __launch_bounds__(256) void gauss_sample_fwd_kernel(const float* __restrict__ stats, const float* __restrict__ eps,
                                                               const float* __restrict__ mask, float* __restrict__ z, int B,
                                                               int C, int T) {
  const int64_t n = (int64_t)B * C * T, ct = (int64_t)C * T;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / ct, r = i % ct;
    const float mk = mask ? mask[b * T + (r % T)] : 1.f;
    z[i] = (stats[b * 2 * ct + r] + eps[i] * expf(stats[b * 2 * ct + ct + r])) * mk;
  }
}
__global__ __launch_bounds__(256) void gauss_sample_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ stats,
                                                               const float* __restrict__ eps, const float* __restrict__ mask,
                                                               float* __restrict__ dstats, int B, int C, int T, int accumulate) {
  const int64_t n = (int64_t)B * C * T, ct = (int64_t)C * T;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / ct, r = i % ct;
    const float mk = mask ? mask[b * T + (r % T)] : 1.f;
    const float d = dz[i] * mk;
    const float dm = d, dl = d * eps[i] * expf(stats[b * 2 * ct + ct + r]);
    float* o = dstats + b * 2 * ct + r;
    o[0] = accumulate ? o[0] + dm : dm;
    o[ct] = accumulate ? o[ct] + dl : dl;
  }
}

// y[b][c][2t], y[b][c][2t+1] = x[b][c][t]
__global__ __launch_bounds__(256) void upsample2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    reinterpret_cast<float2*>(y)[i] = make_float2(x[i], x[i]);
}
__global__ __launch_bounds__(256) void upsample2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float2 v = reinterpret_cast<const float2*>(dy)[i];
    dx[i] = v.x + v.y;
  }
}

// ---- pointwise activations ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float softplusf_(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, int op) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = x[i];
    y[i] = op == TTTS_ACT_RELU ? fmaxf(v, 0.f) : (op == TTTS_ACT_SILU ? v * sigmoidf_(v) : v * tanhf(softplusf_(v)));
  }
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                      float* __restrict__ dx, int64_t n, int op) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = x[i];
    float g;
    if (op == TTTS_ACT_RELU) {
      g = v > 0.f ? 1.f : 0.f;
    } else if (op == TTTS_ACT_SILU) {   // d/dx x sigmoid(x) = s (1 + x (1 - s))
      const float sg = sigmoidf_(v);
      g = sg * (1.f + v * (1.f - sg));
    } else {  // mish: d/dx x tanh(sp(x)) = tanh(sp) + x (1 - tanh^2(sp)) sigmoid(x)
      const float t = tanhf(softplusf_(v));
      g = t + v * (1.f - t * t) * sigmoidf_(v);
    }
    dx[i] = dy[i] * g;
  }
}

// y = keep(e) ? x / (1 - p) : 0 with the keep rule of the GPT kernels (16 hash bits per element); the same call with
// dy as x is the backward.
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n,
                                                      uint32_t thr, float inv_keep, uint32_t s_lo, uint32_t s_hi,
                                                      const uint32_t* ctr) {
  s_hi = seed_mix(s_hi, ctr);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint32_t h = hash32((uint32_t)(i >> 1), s_lo, s_hi);
    const uint32_t bits = (h >> (16 * (uint32_t)(i & 1))) & 0xFFFFu;
    y[i] = bits >= thr ? x[i] * inv_keep : 0.f;
  }
}

// ---- anti-aliased SnakeBeta, one (b, c) row per workgroup --------------------------------------------------------------
// up:   u[n] = 2 * sum_j xp[j] * f[n + 15 - 2j],  xp = replicate-pad(x, 5, 5) (j = index into xp), n in [0, 2T)
// act:  v = u + sin^2(u * a) / (b + 1e-9),  a = exp(alpha[c]), b = exp(beta[c])
// down: y[t] = sum_k g[k] * vp[2t + k],  vp = replicate-pad(v, 5, 6)
constexpr int AA_K = 12;
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ __launch_bounds__(256) void snake_aa_fwd_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                                                           const float* __restrict__ beta, const float* __restrict__ fup,
                                                           const float* __restrict__ fdn, float* __restrict__ y, int C, int T) {
  extern __shared__ float aa_smem[];
  float* xs = aa_smem;          // [T]
  float* vs = aa_smem + T;      // [2T]
  __shared__ float fu[AA_K], fd[AA_K];
  const int row = blockIdx.x, c = row % C;
  if (threadIdx.x < AA_K) { fu[threadIdx.x] = fup[threadIdx.x]; fd[threadIdx.x] = fdn[threadIdx.x]; }
  for (int t = threadIdx.x; t < T; t += 256) xs[t] = x[(int64_t)row * T + t];
  __syncthreads();
  const float a = expf(alpha[c]), ib = 1.f / (expf(beta[c]) + 1e-9f);
  for (int n = threadIdx.x; n < 2 * T; n += 256) {
    // taps k = n + 15 - 2j in [0, 12)  <=>  j in ((n + 3) / 2, (n + 15) / 2], same parity pattern
    float u = 0.f;
    const int jhi = (n + 15) >> 1;
#pragma unroll
    for (int q = 0; q < AA_K / 2; ++q) {
      const int j = jhi - q, k = n + 15 - 2 * j;          // k = (n + 15) & 1 + 2q
      u = fmaf(xs[clampi(j - 5, 0, T - 1)], fu[k], u);
    }
    u *= 2.f;
    const float sn = sinf(u * a);
    vs[n] = u + ib * sn * sn;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < T; t += 256) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < AA_K; ++k) s = fmaf(fd[k], vs[clampi(2 * t + k - 5, 0, 2 * T - 1)], s);
    y[(int64_t)row * T + t] = s;
  }
}

// backward of the above; dalpha / dbeta accumulated with one atomic per row
__global__ __launch_bounds__(256) void snake_aa_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ alpha, const float* __restrict__ beta,
                                                           const float* __restrict__ fup, const float* __restrict__ fdn,
                                                           float* __restrict__ dx, float* __restrict__ dalpha,
                                                           float* __restrict__ dbeta, int C, int T) {
  extern __shared__ float aa_smem[];
  float* xs = aa_smem;              // [T]   x, later reused for dy
  float* dys = aa_smem + T;         // [T]
  float* dus = aa_smem + 2 * T;     // [2T]  du
  __shared__ float fu[AA_K], fd[AA_K], sh[4];
  const int row = blockIdx.x, c = row % C;
  if (threadIdx.x < AA_K) { fu[threadIdx.x] = fup[threadIdx.x]; fd[threadIdx.x] = fdn[threadIdx.x]; }
  for (int t = threadIdx.x; t < T; t += 256) { xs[t] = x[(int64_t)row * T + t]; dys[t] = dy[(int64_t)row * T + t]; }
  __syncthreads();
  const float a = expf(alpha[c]), eb = expf(beta[c]), ib = 1.f / (eb + 1e-9f);
  float ga = 0.f, gb = 0.f;
  for (int n = threadIdx.x; n < 2 * T; n += 256) {
    float u = 0.f;
    const int jhi = (n + 15) >> 1;
#pragma unroll
    for (int q = 0; q < AA_K / 2; ++q) {
      const int j = jhi - q, k = n + 15 - 2 * j;
      u = fmaf(xs[clampi(j - 5, 0, T - 1)], fu[k], u);
    }
    u *= 2.f;
    // dv[n] = sum over (t, k) with clamp(2t + k - 5) == n of g[k] dy[t]   (replicate-pad adjoint: edges collect)
    float dv = 0.f;
    if (n > 0 && n < 2 * T - 1) {
      // 2t + k - 5 = n, k in [0, 12): t from ceil((n - 6) / 2) to floor((n + 5) / 2)
      const int thi = (n + 5) >> 1;
#pragma unroll
      for (int q = 0; q < AA_K / 2; ++q) {
        const int t = thi - q, k = n + 5 - 2 * t;
        if (t >= 0 && t < T) dv = fmaf(fd[k], dys[t], dv);
      }
    } else {
      // edge sample: every padded position that clamps to it.  (Only the rows next to the edge can: 2 t + k - 5 <= 0 needs t <= 2,
      // >= 2 T - 1 needs t >= T - 3.  Round 6: the right edge used to walk ALL T rows x 12 taps in ONE thread -- 37 000 serial
      // iterations at T = 3072, 395 us for a kernel whose forward takes 14; same terms in the same order now.)
      for (int t = (n == 0 ? 0 : max(0, T - 4)); t < T; ++t) {
#pragma unroll
        for (int k = 0; k < AA_K; ++k) {
          const int pos = 2 * t + k - 5;
          const bool hit = (n == 0) ? (pos <= 0) : (pos >= 2 * T - 1);
          if (hit) dv = fmaf(fd[k], dys[t], dv);
        }
        if (n == 0 && 2 * t - 5 > 0) break;      // later t cannot reach the left edge
      }
    }
    const float sn = sinf(u * a), s2 = sinf(2.f * u * a);
    dus[n] = dv * (1.f + a * ib * s2);
    ga += dv * ib * s2 * u * a;                      // d/d(log-alpha): (1/b) sin(2ua) u * a
    gb += dv * sn * sn * (-eb * ib * ib);            // d/d(log-beta): sin^2 * d(1/(e^b + eps))/db
  }
  __syncthreads();
  // dx[i] = sum over j with clamp(j - 5) == i of 2 * sum_n du[n] f[n + 15 - 2j]
  for (int i = threadIdx.x; i < T; i += 256) {
    float s = 0.f;
    const int jlo = (i == 0) ? 0 : i + 5, jhi = (i == T - 1) ? T + 9 : i + 5;
    for (int j = jlo; j <= jhi; ++j) {
#pragma unroll
      for (int k = 0; k < AA_K; ++k) {
        const int n = 2 * j + k - 15;
        if (n >= 0 && n < 2 * T) s = fmaf(dus[n], fu[k], s);
      }
    }
    dx[(int64_t)row * T + i] = 2.f * s;
  }
  ga = wave_sum(ga); gb = wave_sum(gb);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = ga;
  __syncthreads();
  const float tga = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = gb;
  __syncthreads();
  const float tgb = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  if (threadIdx.x == 0) { atomicAdd(dalpha + c, tga); atomicAdd(dbeta + c, tgb); }
}

// ---- channel LayerNorm on (B, C, T): one thread per (b, t) column, coalesced along t --------------------------------
__global__ __launch_bounds__(256) void layernorm_ch_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ y,
                                                               float* __restrict__ mean, float* __restrict__ rstd, int B,
                                                               int C, int T, float eps) {
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (col >= (int64_t)B * T) return;
  const int64_t b = col / T, t = col % T;
  const float* xp = x + b * C * T + t;
  float s = 0.f;
  for (int c = 0; c < C; ++c) s += xp[(int64_t)c * T];
  const float mu = s / C;
  float v = 0.f;
  for (int c = 0; c < C; ++c) { const float d = xp[(int64_t)c * T] - mu; v = fmaf(d, d, v); }
  const float rs = rsqrtf(v / C + eps);
  mean[col] = mu; rstd[col] = rs;
  float* yp = y + b * C * T + t;
  for (int c = 0; c < C; ++c) yp[(int64_t)c * T] = (xp[(int64_t)c * T] - mu) * rs * gamma[c] + beta[c];
}
// dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)),  g = dy * gamma
__global__ __launch_bounds__(256) void layernorm_ch_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, float* __restrict__ dx,
                                                                  int B, int C, int T) {
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (col >= (int64_t)B * T) return;
  const int64_t b = col / T, t = col % T, base = b * C * T + t;
  const float mu = mean[col], rs = rstd[col];
  float s1 = 0.f, s2 = 0.f;
  for (int c = 0; c < C; ++c) {
    const float g = dy[base + (int64_t)c * T] * gamma[c], xh = (x[base + (int64_t)c * T] - mu) * rs;
    s1 += g; s2 = fmaf(g, xh, s2);
  }
  s1 /= C; s2 /= C;
  for (int c = 0; c < C; ++c) {
    const float g = dy[base + (int64_t)c * T] * gamma[c], xh = (x[base + (int64_t)c * T] - mu) * rs;
    dx[base + (int64_t)c * T] = rs * (g - s1 - xh * s2);
  }
}
// Tiled variants for C <= 256 (the 192-channel stacks at 32 x 256 frames are 8192 columns: one thread per column put 32
// workgroups on the chip and walked the channels serially, 90-107 us).  Workgroup = 32 columns x 8 channel groups, every value
// is read once and held in registers (<= 32 per thread), column statistics meet in LDS.
constexpr int LNC_R = 32;
__device__ __forceinline__ float lnc_group_sum(float v, float (*sh)[33], int cg, int tl) {
  __syncthreads();
  sh[cg][tl] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < 8; ++g) s += sh[g][tl];
  return s;
}
__global__ __launch_bounds__(256) void layernorm_ch_fwd_tiled_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, float* __restrict__ y,
                                                                     float* __restrict__ mean, float* __restrict__ rstd, int B,
                                                                     int C, int T, float eps) {
  __shared__ float sh[8][33];
  const int tl = threadIdx.x & 31, cg = threadIdx.x >> 5;
  const int64_t ncol = (int64_t)B * T, col = min((int64_t)blockIdx.x * 32 + tl, ncol - 1);
  const bool live = (int64_t)blockIdx.x * 32 + tl < ncol;
  const int64_t b = col / T, t = col % T, base = b * C * T + t;
  float xv[LNC_R];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LNC_R; ++i) {
    const int c = cg + 8 * i;
    xv[i] = c < C ? x[base + (int64_t)c * T] : 0.f;
    s += xv[i];
  }
  const float mu = lnc_group_sum(s, sh, cg, tl) / C;
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < LNC_R; ++i) {
    const float d = cg + 8 * i < C ? xv[i] - mu : 0.f;
    v = fmaf(d, d, v);
  }
  const float rs = rsqrtf(lnc_group_sum(v, sh, cg, tl) / C + eps);
  if (cg == 0 && live) { mean[col] = mu; rstd[col] = rs; }
#pragma unroll
  for (int i = 0; i < LNC_R; ++i) {
    const int c = cg + 8 * i;
    if (c < C && live) y[base + (int64_t)c * T] = (xv[i] - mu) * rs * gamma[c] + beta[c];
  }
}
__global__ __launch_bounds__(256) void layernorm_ch_bwd_dx_tiled_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                        const float* __restrict__ gamma,
                                                                        const float* __restrict__ mean,
                                                                        const float* __restrict__ rstd, float* __restrict__ dx,
                                                                        int B, int C, int T) {
  __shared__ float sh[8][33];
  const int tl = threadIdx.x & 31, cg = threadIdx.x >> 5;
  const int64_t ncol = (int64_t)B * T, col = min((int64_t)blockIdx.x * 32 + tl, ncol - 1);
  const bool live = (int64_t)blockIdx.x * 32 + tl < ncol;
  const int64_t b = col / T, t = col % T, base = b * C * T + t;
  const float mu = mean[col], rs = rstd[col];
  float gv[LNC_R], xh[LNC_R];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < LNC_R; ++i) {
    const int c = cg + 8 * i;
    const bool ok = c < C;
    gv[i] = ok ? dy[base + (int64_t)c * T] * gamma[c] : 0.f;
    xh[i] = ok ? (x[base + (int64_t)c * T] - mu) * rs : 0.f;
    s1 += gv[i]; s2 = fmaf(gv[i], xh[i], s2);
  }
  s1 = lnc_group_sum(s1, sh, cg, tl) / C;
  s2 = lnc_group_sum(s2, sh, cg, tl) / C;
#pragma unroll
  for (int i = 0; i < LNC_R; ++i) {
    const int c = cg + 8 * i;
    if (c < C && live) dx[base + (int64_t)c * T] = rs * (gv[i] - s1 - xh[i] * s2);
  }
}
// dgamma[c] += sum_{b,t} dy * xhat, dbeta[c] += sum dy: one workgroup per channel
__global__ __launch_bounds__(256) void layernorm_ch_bwd_param_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                     const float* __restrict__ mean,
                                                                     const float* __restrict__ rstd,
                                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                     int B, int C, int T) {
  __shared__ float sh[4];
  const int c = blockIdx.x;
  float sg = 0.f, sb = 0.f;
  for (int b = 0; b < B; ++b)
    for (int t = threadIdx.x; t < T; t += 256) {
      const int64_t o = ((int64_t)b * C + c) * T + t, col = (int64_t)b * T + t;
      const float d = dy[o];
      sg = fmaf(d, (x[o] - mean[col]) * rstd[col], sg);
      sb += d;
    }
  sg = wave_sum(sg); sb = wave_sum(sb);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sg;
  __syncthreads();
  const float tg = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sb;
  __syncthreads();
  if (threadIdx.x == 0) { dgamma[c] += tg; dbeta[c] += (sh[0] + sh[1]) + (sh[2] + sh[3]); }
}

// ---- token embedding straight into (B, C, T): y[b][c][t] = table[idx[b][t]][c]; bwd scatter-adds -------------------------
__global__ __launch_bounds__(256) void embedding_ct_fwd_kernel(const int64_t* __restrict__ idx, const float* __restrict__ table,
                                                               float* __restrict__ y, int B, int C, int T) {
  const int64_t n = (int64_t)B * C * T;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int t = (int)(i % T), c = (int)((i / T) % C);
    const int64_t b = i / ((int64_t)C * T);
    y[i] = table[idx[b * T + t] * C + c];
  }
}
__global__ __launch_bounds__(256) void embedding_ct_bwd_kernel(const int64_t* __restrict__ idx, const float* __restrict__ dy,
                                                               float* __restrict__ dtable, int B, int C, int T) {
  const int64_t n = (int64_t)B * C * T;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int t = (int)(i % T), c = (int)((i / T) % C);
    const int64_t b = i / ((int64_t)C * T);
    atomicAdd(dtable + idx[b * T + t] * C + c, dy[i]);
  }
}

// ---- masked temporal mean: y[b][c] = sum_t x[b][c][t] m[b][t] / sum_t m[b][t]  (one wave per (b, c)) ------------------------
__global__ __launch_bounds__(256) void masked_mean_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                              float* __restrict__ y, int B, int C, int T) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B * C) return;
  const int b = row / C;
  float s = 0.f, ms = 0.f;
  for (int t = lane; t < T; t += 64) {
    const float mk = mask ? mask[(int64_t)b * T + t] : 1.f;
    s = fmaf(x[(int64_t)row * T + t], mk, s);
    ms += mk;
  }
  s = wave_sum(s); ms = wave_sum(ms);
  if (lane == 0) y[row] = s / ms;
}
__global__ __launch_bounds__(256) void masked_mean_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ mask,
                                                              float* __restrict__ dx, int B, int C, int T) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B * C) return;
  const int b = row / C;
  float ms = 0.f;
  for (int t = lane; t < T; t += 64) ms += mask ? mask[(int64_t)b * T + t] : 1.f;
  ms = wave_sum(ms);
  const float g = dy[row] / ms;
  for (int t = lane; t < T; t += 64) dx[(int64_t)row * T + t] = g * (mask ? mask[(int64_t)b * T + t] : 1.f);
}

static inline int grid_for(int64_t n) { return (int)std::min<int64_t>(cdiv(n, 256), 4096); }

}  // namespace ttts

using namespace ttts;

extern "C" int ttts_gate_fwd_f32(const float* x, float* y, int32_t B, int32_t H, int32_t T, int32_t kind, void* stream) {
  TTTS_REQUIRE(x && y && B > 0 && H > 0 && T > 0 && (kind == 0 || kind == 1), "gate_fwd: bad arguments");
  gate_fwd_kernel<<<grid_for((int64_t)B * H * T), 256, 0, as_stream(stream)>>>(x, y, B, H, T, kind);
  return check_launch("gate_fwd");
}
extern "C" int ttts_gate_bwd_f32(const float* dy, const float* x, float* dx, int32_t B, int32_t H, int32_t T, int32_t kind,
                                 void* stream) {
  TTTS_REQUIRE(dy && x && dx && B > 0 && H > 0 && T > 0 && (kind == 0 || kind == 1), "gate_bwd: bad arguments");
  gate_bwd_kernel<<<grid_for((int64_t)B * H * T), 256, 0, as_stream(stream)>>>(dy, x, dx, B, H, T, kind);
  return check_launch("gate_bwd");
}
extern "C" int ttts_gate_bwd_rowsum_f32(const float* dy, const float* x, float* dx, float* rowsum, int32_t B, int32_t H, int32_t T,
                                        int32_t kind, void* stream) {
  TTTS_REQUIRE(dy && x && dx && rowsum && B > 0 && H > 0 && T > 0 && (kind == 0 || kind == 1), "gate_bwd_rowsum: bad arguments");
  gate_bwd_rowsum_kernel<<<(unsigned)cdiv((int64_t)B * H, 4), 256, 0, as_stream(stream)>>>(dy, x, dx, rowsum, B, H, T, kind);
  return check_launch("gate_bwd_rowsum");
}
extern "C" int ttts_mul_mask_f32(const float* x, const float* mask, float* y, int32_t B, int32_t C, int32_t T, void* stream) {
  TTTS_REQUIRE(x && mask && y && B > 0 && C > 0 && T > 0, "mul_mask: bad arguments");
  mul_mask_kernel<<<grid_for((int64_t)B * C * T), 256, 0, as_stream(stream)>>>(x, mask, y, B, C, T);
  return check_launch("mul_mask");
}
extern "C" int ttts_gauss_sample_fwd_f32(const float* stats, const float* eps, const float* mask, float* z, int32_t B,
                                         int32_t C, int32_t T, void* stream) {
  TTTS_REQUIRE(stats && eps && z && B > 0 && C > 0 && T > 0, "gauss_sample_fwd: bad arguments");
  gauss_sample_fwd_kernel<<<grid_for((int64_t)B * C * T), 256, 0, as_stream(stream)>>>(stats, eps, mask, z, B, C, T);
  return check_launch("gauss_sample_fwd");
}
extern "C" int ttts_gauss_sample_bwd_f32(const float* dz, const float* stats, const float* eps, const float* mask,
                                         float* dstats, int32_t B, int32_t C, int32_t T, int32_t accumulate, void* stream) {
  TTTS_REQUIRE(dz && stats && eps && dstats && B > 0 && C > 0 && T > 0, "gauss_sample_bwd: bad arguments");
  gauss_sample_bwd_kernel<<<grid_for((int64_t)B * C * T), 256, 0, as_stream(stream)>>>(dz, stats, eps, mask, dstats, B, C, T, accumulate);
  return check_launch("gauss_sample_bwd");
}
extern "C" int ttts_upsample2_fwd_f32(const float* x, float* y, int64_t n, void* stream) {
  TTTS_REQUIRE(x && y && n > 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0, "upsample2_fwd: bad arguments");
  upsample2_fwd_kernel<<<grid_for(n), 256, 0, as_stream(stream)>>>(x, y, n);
  return check_launch("upsample2_fwd");
}
extern "C" int ttts_upsample2_bwd_f32(const float* dy, float* dx, int64_t n, void* stream) {
  TTTS_REQUIRE(dy && dx && n > 0 && (reinterpret_cast<uintptr_t>(dy) & 7) == 0, "upsample2_bwd: bad arguments");
  upsample2_bwd_kernel<<<grid_for(n), 256, 0, as_stream(stream)>>>(dy, dx, n);
  return check_launch("upsample2_bwd");
}
extern "C" int ttts_act_fwd_f32(const float* x, float* y, int64_t n, int32_t op, void* stream) {
  TTTS_REQUIRE(x && y && n > 0 && (op >= TTTS_ACT_RELU && op <= TTTS_ACT_SILU), "act_fwd: bad arguments");
  act_fwd_kernel<<<grid_for(n), 256, 0, as_stream(stream)>>>(x, y, n, op);
  return check_launch("act_fwd");
}
extern "C" int ttts_act_bwd_f32(const float* dy, const float* x, float* dx, int64_t n, int32_t op, void* stream) {
  TTTS_REQUIRE(dy && x && dx && n > 0 && (op >= TTTS_ACT_RELU && op <= TTTS_ACT_SILU), "act_bwd: bad arguments");
  act_bwd_kernel<<<grid_for(n), 256, 0, as_stream(stream)>>>(dy, x, dx, n, op);
  return check_launch("act_bwd");
}
extern "C" int ttts_dropout_f32(const float* x, float* y, int64_t n, float p, uint64_t seed, const uint32_t* dropout_counter,
                                void* stream) {
  TTTS_REQUIRE(x && y && n > 0 && p >= 0.f && p < 1.f, "dropout: bad arguments");
  const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
  dropout_kernel<<<grid_for(n), 256, 0, as_stream(stream)>>>(x, y, n, dropout_threshold(p), 1.f / (1.f - p), lo, hi, dropout_counter);
  return check_launch("dropout");
}
extern "C" int ttts_snake_aa_fwd_f32(const float* x, const float* alpha, const float* beta, const float* up_filter,
                                     const float* down_filter, float* y, int32_t B, int32_t C, int32_t T, void* stream) {
  TTTS_REQUIRE(x && alpha && beta && up_filter && down_filter && y && B > 0 && C > 0, "snake_aa_fwd: bad arguments");
  TTTS_REQUIRE(T >= 8 && T <= 3072, "snake_aa_fwd: T = %d outside [8, 3072]", T);
  snake_aa_fwd_kernel<<<B * C, 256, (size_t)3 * T * sizeof(float), as_stream(stream)>>>(x, alpha, beta, up_filter, down_filter, y, C, T);
  return check_launch("snake_aa_fwd");
}
extern "C" int ttts_snake_aa_bwd_f32(const float* dy, const float* x, const float* alpha, const float* beta,
                                     const float* up_filter, const float* down_filter, float* dx, float* dalpha,
                                     float* dbeta, int32_t B, int32_t C, int32_t T, void* stream) {
  TTTS_REQUIRE(dy && x && alpha && beta && up_filter && down_filter && dx && dalpha && dbeta && B > 0 && C > 0, "snake_aa_bwd: bad arguments");
  TTTS_REQUIRE(T >= 8 && T <= 3072, "snake_aa_bwd: T = %d outside [8, 3072]", T);
  snake_aa_bwd_kernel<<<B * C, 256, (size_t)4 * T * sizeof(float), as_stream(stream)>>>(dy, x, alpha, beta, up_filter, down_filter, dx,
                                                                                      dalpha, dbeta, C, T);
  return check_launch("snake_aa_bwd");
}
extern "C" int ttts_layernorm_ch_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                         float* rstd, int32_t B, int32_t C, int32_t T, float eps, void* stream) {
  TTTS_REQUIRE(x && gamma && beta && y && mean && rstd && B > 0 && C > 0 && T > 0, "layernorm_ch_fwd: bad arguments");
  if (C <= 8 * LNC_R) layernorm_ch_fwd_tiled_kernel<<<(int)cdiv((int64_t)B * T, 32), 256, 0, as_stream(stream)>>>(x, gamma, beta, y, mean, rstd, B, C, T, eps);
  else layernorm_ch_fwd_kernel<<<(int)cdiv((int64_t)B * T, 256), 256, 0, as_stream(stream)>>>(x, gamma, beta, y, mean, rstd, B, C, T, eps);
  return check_launch("layernorm_ch_fwd");
}
extern "C" int ttts_layernorm_ch_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean,
                                         const float* rstd, float* dx, float* dgamma, float* dbeta, int32_t B, int32_t C,
                                         int32_t T, void* stream) {
  TTTS_REQUIRE(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && B > 0 && C > 0 && T > 0, "layernorm_ch_bwd: bad arguments");
  if (C <= 8 * LNC_R) layernorm_ch_bwd_dx_tiled_kernel<<<(int)cdiv((int64_t)B * T, 32), 256, 0, as_stream(stream)>>>(dy, x, gamma, mean, rstd, dx, B, C, T);
  else layernorm_ch_bwd_dx_kernel<<<(int)cdiv((int64_t)B * T, 256), 256, 0, as_stream(stream)>>>(dy, x, gamma, mean, rstd, dx, B, C, T);
  layernorm_ch_bwd_param_kernel<<<C, 256, 0, as_stream(stream)>>>(dy, x, mean, rstd, dgamma, dbeta, B, C, T);
  return check_launch("layernorm_ch_bwd");
}

extern "C" int ttts_embedding_ct_fwd_f32(const int64_t* idx, const float* table, float* y, int32_t B, int32_t C, int32_t T,
                                         void* stream) {
  TTTS_REQUIRE(idx && table && y && B > 0 && C > 0 && T > 0, "embedding_ct_fwd: bad arguments");
  embedding_ct_fwd_kernel<<<grid_for((int64_t)B * C * T), 256, 0, as_stream(stream)>>>(idx, table, y, B, C, T);
  return check_launch("embedding_ct_fwd");
}
extern "C" int ttts_embedding_ct_bwd_f32(const int64_t* idx, const float* dy, float* dtable, int32_t B, int32_t C, int32_t T,
                                         void* stream) {
  TTTS_REQUIRE(idx && dy && dtable && B > 0 && C > 0 && T > 0, "embedding_ct_bwd: bad arguments");
  embedding_ct_bwd_kernel<<<grid_for((int64_t)B * C * T), 256, 0, as_stream(stream)>>>(idx, dy, dtable, B, C, T);
  return check_launch("embedding_ct_bwd");
}
extern "C" int ttts_masked_mean_fwd_f32(const float* x, const float* mask, float* y, int32_t B, int32_t C, int32_t T,
                                        void* stream) {
  TTTS_REQUIRE(x && y && B > 0 && C > 0 && T > 0, "masked_mean_fwd: bad arguments");
  masked_mean_fwd_kernel<<<(int)cdiv((int64_t)B * C, 4), 256, 0, as_stream(stream)>>>(x, mask, y, B, C, T);
  return check_launch("masked_mean_fwd");
}
extern "C" int ttts_masked_mean_bwd_f32(const float* dy, const float* mask, float* dx, int32_t B, int32_t C, int32_t T,
                                        void* stream) {
  TTTS_REQUIRE(dy && dx && B > 0 && C > 0 && T > 0, "masked_mean_bwd: bad arguments");
  masked_mean_bwd_kernel<<<(int)cdiv((int64_t)B * C, 4), 256, 0, as_stream(stream)>>>(dy, mask, dx, B, C, T);
  return check_launch("masked_mean_bwd");
}
