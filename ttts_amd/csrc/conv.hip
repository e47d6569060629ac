// 1-D convolution family of the VQ-VAE-GAN path (fp32, (B, C, L) layout, groups = 1): the building block of
// ResBlock1 / WN / HiFi-GAN Generator / PosteriorAudioEncoder / DiscriminatorP (ttts/vqvae/modules.py:224-318,
// ttts/vqvae/vq2.py:341-415,667-745,418-494).
//
// Round-1 scope: CORRECT, reasonably tiled direct kernels that cover every (kernel, stride, dilation, padding)
// combination of the path -- forward, data gradient (which is also the ConvTranspose1d forward), weight gradient,
// bias gradient, weight-norm -- with the elementwise neighbours fused (leaky-relu on the input, bias, residual add,
// tanh on the output).  They run on the fp32 vector units; the MFMA implicit-GEMM version of the stride-1 case
// (channel-last tiles through the gemm.hip machinery) is the next step for this family.
//
// Tiling (fwd / dgrad): workgroup = 32 channels x 128 positions, thread = 4 channels x 4 positions (positions
// interleaved by 32 so that a wave reads consecutive LDS words), input channels staged 8 at a time through LDS together
// with the matching weight slab.
#include <algorithm>

#include "common.hpp"

namespace ttts {

// conv_mfma.hip: implicit-GEMM kernels on the f32 matrix cores; *handled = false leaves the launch to the kernels below
int conv1d_mfma_try(const float* x, const float* w, const float* bias, const float* bbias, const float* resid,
                    const float* gate, const float* omask, float* y, int B, int M, int N, int Lin, int Lout, int K,
                    int stride, int pad, int dil, int transposed, float in_slope, float gate_slope, int out_act,
                    float out_slope, float out_scale, int accumulate, const ConvCtx& cx, hipStream_t stream, bool* handled);
// db / db_done: bias gradient folded into the launch when the chosen kernel streams dy itself (*db_done set), else left to the caller
int conv1d_wgrad_mfma_try(const float* dy, const float* x, float* dw, float* db, bool* db_done, int B, int Cin, int Lin, int Cout,
                          int Lout, int K, int stride, int pad, int dil, float dy_slope, float x_slope, const ConvCtx& cx,
                          hipStream_t stream, bool* handled);
int conv1d_dgrad_strided_mfma_try(const float* dy, const float* w, const float* bias, const float* resid,
                                  const float* gate, const float* omask, float* dx, int B, int Cin, int Lin, int Cout,
                                  int Lout, int K, int stride, int pad, float in_slope, float gate_slope, float out_scale,
                                  int accumulate, const ConvCtx& cx, hipStream_t stream, bool* handled);

int conv1d_mfma_dual_try(const float* x, const float* w, const float* bias, const float* resid, const float* omask, float* y, float* y2,
                         int B, int M, int M1, int N, int Lin, int Lout, int K, int pad, int dil, float in_slope, int accumulate2,
                         const ConvCtx& cx, hipStream_t stream, bool* handled);

// conv_thin.hip: streaming kernels for one input channel / one output channel
int conv1d_thin_fwd_try(const float* x, const float* w, const float* bias, const float* bbias, const float* resid,
                        const float* gate, const float* omask, float* y, int B, int Cin, int Lin, int Cout, int Lout, int K,
                        int stride, int pad, int dil, float in_slope, int out_act, float out_slope, float out_scale,
                        int accumulate, hipStream_t stream, bool* handled);
int conv1d_thin_wgrad_try(const float* dy, const float* x, float* dw, int B, int Cin, int Lin, int Cout, int Lout, int K,
                          int stride, int pad, int dil, float dy_slope, float x_slope, const ConvCtx& cx, hipStream_t stream, bool* handled);
int conv1d_thin_dgrad_try(const float* dy, const float* w, const float* bias, const float* resid, const float* gate,
                          const float* omask, float* dx, int B, int Cin, int Lin, int Cout, int Lout, int K, int stride, int pad,
                          int dil, float in_slope, float out_scale, int accumulate, hipStream_t stream, bool* handled);

// conv_grouped.hip: few-channels-per-group convolutions on the f32-input matrix cores
int conv1d_grouped_fwd_mfma_try(const float* x, const float* w, const float* bias, const float* bbias, const float* resid,
                                const float* gate, const float* omask, float* y, int B, int Cin, int Lin, int Cout, int Lout,
                                int K, int stride, int pad, int dil, int groups, float in_slope, float gate_slope, int out_act,
                                float out_slope, float out_scale, int accumulate, hipStream_t stream, bool* handled);
int conv1d_grouped_dgrad_mfma_try(const float* dy, const float* w, const float* bias, const float* resid, const float* gate,
                                  const float* omask, float* dx, int B, int Cin, int Lin, int Cout, int Lout, int K, int stride,
                                  int pad, int dil, int groups, float in_slope, float gate_slope, float out_scale, int accumulate,
                                  hipStream_t stream, bool* handled);
int conv1d_grouped_wgrad_mfma_try(const float* dy, const float* x, float* dw, int B, int Cin, int Lin, int Cout, int Lout, int K,
                                  int stride, int pad, int dil, int groups, float dy_slope, float x_slope, const ConvCtx& cx,
                                  hipStream_t stream, bool* handled);

constexpr int CV_CT = 32;   // output-channel tile
constexpr int CV_LT = 128;  // position tile
constexpr int CV_CI = 8;    // input channels per LDS stage

struct ConvParams {
  const float* x;     // fwd: input [B, Cin, Lin]            dgrad: dy [B, Cout, Lout]
  const float* w;     // [Cout, Cin, K]
  const float* bias;  // per output channel of THIS kernel or NULL
  const float* bbias; // fwd: per (batch element, output channel) bias [B, Cout] or NULL (broadcast conditioning)
  const float* resid; // added to the output (same shape) or NULL
  const float* omask; // [B, L_out]: output multiplied by omask[b][l] (sequence mask x_mask of the VITS stacks) or NULL
  const float* gate;  // output multiplied by lrelu'(gate) = (gate > 0 ? 1 : gate_slope); same shape as the output; or NULL
  float* y;           // fwd: [B, Cout, Lout]                 dgrad: dx [B, Cin, Lin]
  int B, Cin, Lin, Cout, Lout, K, stride, pad, dil;   // Cin, Cout are PER GROUP
  int G;              // groups (channels of a group are contiguous): tensors hold G*Cin / G*Cout channels
  float in_slope;     // leaky-relu slope applied to the input on load (1 = identity)
  float gate_slope;
  int out_act;        // fwd: 0 none, 1 tanh, 2 leaky-relu(out_slope) on the output
  float out_slope;
  float out_scale;    // y = [y_old +] out_scale * act(gate * (conv + bias) + resid)
  int accumulate;     // 1: add to the existing output
};

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

// y[b][co][l] = bias[co] + sum_{ci,k} w[co][ci][k] * act(x[b][ci][l*stride - pad + k*dil])
__global__ __launch_bounds__(256) void conv1d_fwd_kernel(ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) float cv_smem[];
  const int lin_t = (CV_LT - 1) * p.stride + (p.K - 1) * p.dil + 1;
  float* xs = cv_smem;                      // [CV_CI][lin_t]
  float* ws = cv_smem + CV_CI * lin_t;      // [CV_CT][CV_CI][K]
  const int tid = threadIdx.x, tl = tid & 31, tc = tid >> 5;
  const int l0 = blockIdx.x * CV_LT, co0 = blockIdx.y * CV_CT, b = blockIdx.z / p.G, grp = blockIdx.z % p.G;
  const int in0 = l0 * p.stride - p.pad;    // input position of xs[.][0]
  const float* wg = p.w + (int64_t)grp * p.Cout * p.Cin * p.K;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  const float* xb = p.x + ((int64_t)b * p.G + grp) * p.Cin * p.Lin;
  for (int c0 = 0; c0 < p.Cin; c0 += CV_CI) {
    for (int i = tid; i < CV_CI * lin_t; i += 256) {
      const int ci = i / lin_t, pos = i % lin_t, gi = in0 + pos;
      float v = 0.f;
      if (c0 + ci < p.Cin && gi >= 0 && gi < p.Lin) v = lrelu(xb[(int64_t)(c0 + ci) * p.Lin + gi], p.in_slope);
      xs[i] = v;
    }
    for (int i = tid; i < CV_CT * CV_CI * p.K; i += 256) {
      const int co = i / (CV_CI * p.K), r = i % (CV_CI * p.K), ci = r / p.K, k = r % p.K;
      ws[i] = (co0 + co < p.Cout && c0 + ci < p.Cin) ? wg[((int64_t)(co0 + co) * p.Cin + c0 + ci) * p.K + k] : 0.f;
    }
    __syncthreads();
    for (int ci = 0; ci < CV_CI; ++ci) {
      const float* xr = xs + ci * lin_t;
      for (int k = 0; k < p.K; ++k) {
        float xv[4], wv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = xr[(tl + 32 * j) * p.stride + k * p.dil];
#pragma unroll
        for (int i = 0; i < 4; ++i) wv[i] = ws[((tc * 4 + i) * CV_CI + ci) * p.K + k];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(wv[i], xv[j], acc[i][j]);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int co = co0 + tc * 4 + i;
    if (co >= p.Cout) continue;
    const int cg = grp * p.Cout + co;         // channel index in the full tensor
    float bv = p.bias ? p.bias[cg] : 0.f;
    if (p.bbias) bv += p.bbias[(int64_t)b * p.G * p.Cout + cg];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int l = l0 + tl + 32 * j;
      if (l >= p.Lout) continue;
      const int64_t o = ((int64_t)b * p.G * p.Cout + cg) * p.Lout + l;
      float v = acc[i][j] + bv;
      if (p.gate) v *= (p.gate[o] > 0.f ? 1.f : p.gate_slope);
      if (p.resid) v += p.resid[o];
      if (p.out_act == 1) v = tanhf(v);
      else if (p.out_act == 2) v = lrelu(v, p.out_slope);
      if (p.omask) v *= p.omask[(int64_t)b * p.Lout + l];
      v *= p.out_scale;
      p.y[o] = p.accumulate ? p.y[o] + v : v;
    }
  }
}

// dx[b][ci][j] = act'(gate) * sum_{co,k : j + pad - k*dil = l*stride, 0 <= l < Lout} w[co][ci][k] * dy[b][co][l]
// (stride > 1 requires dil == 1).  With w = a ConvTranspose1d weight [Cin_t, Cout_t, K] this is its forward
// (x_t plays dy, "Cout" = Cin_t, "Cin" = Cout_t, "Lin" = the transposed conv's output length).
__global__ __launch_bounds__(256) void conv1d_dgrad_kernel(ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) float cv_smem[];
  const int j0 = blockIdx.x * CV_LT, ci0 = blockIdx.y * CV_CT, b = blockIdx.z / p.G, grp = blockIdx.z % p.G;
  const float* wg = p.w + (int64_t)grp * p.Cout * p.Cin * p.K;
  // dy positions that can touch dx[j0 .. j0+127]: l in [lmin, lmax]
  const int lo = j0 + p.pad - (p.K - 1) * p.dil;
  const int lmin = lo <= 0 ? 0 : (lo + p.stride - 1) / p.stride;
  const int lt = (CV_LT - 1 + (p.K - 1) * p.dil) / p.stride + 2;   // tile length in dy positions
  float* ds = cv_smem;                  // [CV_CI][lt]
  float* ws = cv_smem + CV_CI * lt;     // [CV_CI (co)][CV_CT (ci)][K]
  const int tid = threadIdx.x, tl = tid & 31, tc = tid >> 5;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  const float* dyb = p.x + ((int64_t)b * p.G + grp) * p.Cout * p.Lout;
  for (int c0 = 0; c0 < p.Cout; c0 += CV_CI) {
    for (int i = tid; i < CV_CI * lt; i += 256) {
      const int co = i / lt, pos = i % lt, l = lmin + pos;
      ds[i] = (c0 + co < p.Cout && l < p.Lout) ? lrelu(dyb[(int64_t)(c0 + co) * p.Lout + l], p.in_slope) : 0.f;
    }
    for (int i = tid; i < CV_CI * CV_CT * p.K; i += 256) {
      const int co = i / (CV_CT * p.K), r = i % (CV_CT * p.K), ci = r / p.K, k = r % p.K;
      ws[i] = (c0 + co < p.Cout && ci0 + ci < p.Cin) ? wg[((int64_t)(c0 + co) * p.Cin + ci0 + ci) * p.K + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = j0 + tl + 32 * jj;
      const int t0 = j + p.pad;
      // valid taps: k with (t0 - k*dil) % stride == 0; stride 1: all k; stride > 1 (dil == 1): k = t0 % stride + m*stride
      const int kfirst = p.stride == 1 ? 0 : t0 % p.stride;
      for (int k = kfirst; k < p.K; k += p.stride) {
        const int t = t0 - k * p.dil;
        if (t < 0) break;              // larger k only decreases t
        const int l = t / p.stride;
        if (l >= p.Lout) continue;
        const int pos = l - lmin;
        if (pos < 0 || pos >= lt) continue;
        for (int co = 0; co < CV_CI; ++co) {
          const float dv = ds[co * lt + pos];
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i][jj] = fmaf(ws[(co * CV_CT + tc * 4 + i) * p.K + k], dv, acc[i][jj]);
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ci = ci0 + tc * 4 + i;
    if (ci >= p.Cin) continue;
    const int cg = grp * p.Cin + ci;
    const float bv = p.bias ? p.bias[cg] : 0.f;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = j0 + tl + 32 * jj;
      if (j >= p.Lin) continue;
      const int64_t o = ((int64_t)b * p.G * p.Cin + cg) * p.Lin + j;
      float v = acc[i][jj] + bv;
      if (p.gate) v *= (p.gate[o] > 0.f ? 1.f : p.gate_slope);
      if (p.resid) v += p.resid[o];
      if (p.omask) v *= p.omask[(int64_t)b * p.Lin + j];
      v *= p.out_scale;
      p.y[o] = p.accumulate ? p.y[o] + v : v;
    }
  }
}

// dw[co][ci][k] += sum_{b,l} dy[b][co][l] * act(x[b][ci][l*stride - pad + k*dil]);  workgroup = 16 co x 16 ci, one
// (co, ci) pair per thread with K <= 16 accumulators, a chunk of 64 output positions of one batch element per stage
constexpr int WG_T = 16, WG_L = 64, WG_KMAX = 16;
__global__ __launch_bounds__(256) void conv1d_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           float* __restrict__ dw, int B, int Cin, int Lin, int Cout,
                                                           int Lout, int K, int stride, int pad, int dil, float dy_slope,
                                                           float x_slope, int chunks_per_block, int G) {
  extern __shared__ __attribute__((aligned(16))) float cv_smem[];
  const int lin_t = (WG_L - 1) * stride + (K - 1) * dil + 1;
  float* dys = cv_smem;               // [WG_T][WG_L + 1]
  float* xs = cv_smem + WG_T * (WG_L + 1);  // [WG_T][lin_t]
  const int tid = threadIdx.x, tco = tid >> 4, tci = tid & 15;
  const int tiles_ci = (Cin + WG_T - 1) / WG_T, tiles_co = (Cout + WG_T - 1) / WG_T;
  const int grp = blockIdx.x / (tiles_ci * tiles_co), tile = blockIdx.x % (tiles_ci * tiles_co);
  const int co0 = (tile / tiles_ci) * WG_T, ci0 = (tile % tiles_ci) * WG_T;
  const int nlc = (Lout + WG_L - 1) / WG_L;   // position chunks per batch element
  const int k0 = blockIdx.z * WG_KMAX;        // this block's slab of taps
  float acc[WG_KMAX];
#pragma unroll
  for (int k = 0; k < WG_KMAX; ++k) acc[k] = 0.f;
  for (int cc = 0; cc < chunks_per_block; ++cc) {
    const int chunk = blockIdx.y * chunks_per_block + cc;
    if (chunk >= B * nlc) break;
    const int b = chunk / nlc, l0 = (chunk % nlc) * WG_L, in0 = l0 * stride - pad;
    for (int i = tid; i < WG_T * WG_L; i += 256) {
      const int co = i / WG_L, l = i % WG_L;
      dys[co * (WG_L + 1) + l] = (co0 + co < Cout && l0 + l < Lout) ? lrelu(dy[(((int64_t)b * G + grp) * Cout + co0 + co) * Lout + l0 + l], dy_slope) : 0.f;
    }
    for (int i = tid; i < WG_T * lin_t; i += 256) {
      const int ci = i / lin_t, pos = i % lin_t, gi = in0 + pos;
      float v = 0.f;
      if (ci0 + ci < Cin && gi >= 0 && gi < Lin) v = lrelu(x[(((int64_t)b * G + grp) * Cin + ci0 + ci) * Lin + gi], x_slope);
      xs[i] = v;
    }
    __syncthreads();
    const float* dr = dys + tco * (WG_L + 1);
    const float* xr = xs + tci * lin_t;
    for (int l = 0; l < WG_L; ++l) {
      const float dv = dr[l];
#pragma unroll
      for (int k = 0; k < WG_KMAX; ++k)
        if (k0 + k < K) acc[k] = fmaf(dv, xr[l * stride + (k0 + k) * dil], acc[k]);
    }
    __syncthreads();
  }
  if (co0 + tco < Cout && ci0 + tci < Cin) {
    float* o = dw + (((int64_t)grp * Cout + co0 + tco) * Cin + ci0 + tci) * K;
#pragma unroll
    for (int k = 0; k < WG_KMAX; ++k)
      if (k0 + k < K) atomicAdd(o + k0 + k, acc[k]);
  }
}

// ---- grouped convolutions with few channels per group (DiscriminatorS: k41, stride 4, groups 4..256, 4 input channels
// per group): one workgroup covers 32 consecutive channels of the FULL tensor, i.e. several groups at once, so the lanes
// stay busy where the per-group kernels above would use 4 or 16 of their 32 channel rows. ------------------------------------
// forward: p.Cin / p.Cout are per-group counts (cig / cog); requires cog % 4 == 0 and (32 % cog == 0 or cog % 32 == 0)
__global__ __launch_bounds__(256) void conv1d_fwd_grouped_kernel(ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) float cv_smem[];
  const int cig = p.Cin, cog = p.Cout, CoutT = p.G * cog, CinT = p.G * cig;
  const int ngt = cog >= CV_CT ? 1 : CV_CT / cog;            // groups per tile
  const int nci = ngt * cig;                                 // input channels staged per tile
  const int lin_t = (CV_LT - 1) * p.stride + (p.K - 1) * p.dil + 1;
  float* xs = cv_smem;                      // [nci][lin_t]
  float* ws = cv_smem + nci * lin_t;        // [CV_CT][cig][K]
  const int tid = threadIdx.x, tl = tid & 31, tc = tid >> 5;
  const int l0 = blockIdx.x * CV_LT, co0 = blockIdx.y * CV_CT, b = blockIdx.z;
  const int g0 = co0 / cog, ci0 = g0 * cig;
  const int in0 = l0 * p.stride - p.pad;
  const float* xb = p.x + (int64_t)b * CinT * p.Lin;
  for (int c = tid >> 6; c < nci; c += 4) {
    const bool cok = ci0 + c < CinT;
    const float* xr = xb + (int64_t)(ci0 + c) * p.Lin;
    for (int pos = tid & 63; pos < lin_t; pos += 64) {
      const int gi = in0 + pos;
      xs[c * lin_t + pos] = (cok && gi >= 0 && gi < p.Lin) ? lrelu(xr[gi], p.in_slope) : 0.f;
    }
  }
  const int wk = cig * p.K;
  for (int i = tid; i < CV_CT * wk; i += 256) {
    const int co = i / wk;
    ws[i] = co0 + co < CoutT ? p.w[(int64_t)(co0 + co) * wk + (i - co * wk)] : 0.f;
  }
  __syncthreads();
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  const int gl = (co0 + tc * 4) / cog - g0;                  // this thread's group within the tile
  for (int ci = 0; ci < cig; ++ci) {
    const float* xr = xs + (gl * cig + ci) * lin_t;
    for (int k = 0; k < p.K; ++k) {
      float xv[4], wv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) xv[j] = xr[(tl + 32 * j) * p.stride + k * p.dil];
#pragma unroll
      for (int i = 0; i < 4; ++i) wv[i] = ws[(tc * 4 + i) * wk + ci * p.K + k];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(wv[i], xv[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cg = co0 + tc * 4 + i;
    if (cg >= CoutT) continue;
    float bv = p.bias ? p.bias[cg] : 0.f;
    if (p.bbias) bv += p.bbias[(int64_t)b * CoutT + cg];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int l = l0 + tl + 32 * j;
      if (l >= p.Lout) continue;
      const int64_t o = ((int64_t)b * CoutT + cg) * p.Lout + l;
      float v = acc[i][j] + bv;
      if (p.gate) v *= (p.gate[o] > 0.f ? 1.f : p.gate_slope);
      if (p.resid) v += p.resid[o];
      if (p.out_act == 1) v = tanhf(v);
      else if (p.out_act == 2) v = lrelu(v, p.out_slope);
      if (p.omask) v *= p.omask[(int64_t)b * p.Lout + l];
      v *= p.out_scale;
      p.y[o] = p.accumulate ? p.y[o] + v : v;
    }
  }
}

// data gradient, grouped: tile = 32 consecutive INPUT channels (32 / cig groups) x 128 positions; cig % 4 == 0, 32 % cig == 0
__global__ __launch_bounds__(256) void conv1d_dgrad_grouped_kernel(ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) float cv_smem[];
  const int cig = p.Cin, cog = p.Cout, CoutT = p.G * cog, CinT = p.G * cig;
  const int ngt = CV_CT / cig, nco = ngt * cog;              // groups / output channels feeding this tile
  const int j0 = blockIdx.x * CV_LT, ci0 = blockIdx.y * CV_CT, b = blockIdx.z;
  const int g0 = ci0 / cig, co0 = g0 * cog;
  const int lo = j0 + p.pad - (p.K - 1) * p.dil;
  const int lmin = lo <= 0 ? 0 : (lo + p.stride - 1) / p.stride;
  const int lt = (CV_LT - 1 + (p.K - 1) * p.dil) / p.stride + 2;
  float* ds = cv_smem;                  // [nco][lt]
  float* ws = cv_smem + nco * lt;       // [nco][cig][K]
  const int tid = threadIdx.x, tl = tid & 31, tc = tid >> 5;
  const float* dyb = p.x + (int64_t)b * CoutT * p.Lout;
  for (int c = tid >> 6; c < nco; c += 4) {
    const bool cok = co0 + c < CoutT;
    const float* dr = dyb + (int64_t)(co0 + c) * p.Lout;
    for (int pos = tid & 63; pos < lt; pos += 64) {
      const int l = lmin + pos;
      ds[c * lt + pos] = (cok && l < p.Lout) ? lrelu(dr[l], p.in_slope) : 0.f;
    }
  }
  const int wk = cig * p.K;
  for (int i = tid; i < nco * wk; i += 256) {
    const int co = i / wk;
    ws[i] = co0 + co < CoutT ? p.w[(int64_t)(co0 + co) * wk + (i - co * wk)] : 0.f;
  }
  __syncthreads();
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  const int gl = (tc * 4) / cig, cl = (tc * 4) % cig;       // group within the tile, first input channel within the group
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int t0 = j0 + tl + 32 * jj + p.pad;
    const int kfirst = p.stride == 1 ? 0 : t0 % p.stride;
    for (int k = kfirst; k < p.K; k += p.stride) {
      const int t = t0 - k * p.dil;
      if (t < 0) break;
      const int l = t / p.stride;
      if (l >= p.Lout) continue;
      const int pos = l - lmin;
      if (pos < 0 || pos >= lt) continue;
      for (int co = 0; co < cog; ++co) {
        const float dv = ds[(gl * cog + co) * lt + pos];
        const float* wr = ws + (gl * cog + co) * wk + cl * p.K + k;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][jj] = fmaf(wr[i * p.K], dv, acc[i][jj]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cg = ci0 + tc * 4 + i;
    if (cg >= CinT) continue;
    const float bv = p.bias ? p.bias[cg] : 0.f;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = j0 + tl + 32 * jj;
      if (j >= p.Lin) continue;
      const int64_t o = ((int64_t)b * CinT + cg) * p.Lin + j;
      float v = acc[i][jj] + bv;
      if (p.gate) v *= (p.gate[o] > 0.f ? 1.f : p.gate_slope);
      if (p.resid) v += p.resid[o];
      if (p.omask) v *= p.omask[(int64_t)b * p.Lin + j];
      v *= p.out_scale;
      p.y[o] = p.accumulate ? p.y[o] + v : v;
    }
  }
}

// weight gradient, grouped: workgroup = 16 consecutive output channels x all cig*K taps (<= 256), thread = one channel x
// up to 16 taps; 64 output positions of one batch element per stage
constexpr int WGG_ACC = 16;
__global__ __launch_bounds__(256) void conv1d_wgrad_grouped_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                   float* __restrict__ dw, int B, int cig, int Lin, int cog,
                                                                   int Lout, int K, int stride, int pad, int dil, int G,
                                                                   float dy_slope, float x_slope, int chunks_per_block) {
  extern __shared__ __attribute__((aligned(16))) float cv_smem[];
  const int CoutT = G * cog, CinT = G * cig, wk = cig * K;
  const int lin_t = (WG_L - 1) * stride + (K - 1) * dil + 1;
  const int ngt = cog >= 16 ? 1 : 16 / cog, nci = ngt * cig;
  float* dys = cv_smem;                     // [16][WG_L + 1]
  float* xs = cv_smem + 16 * (WG_L + 1);    // [nci][lin_t]
  const int tid = threadIdx.x, tco = tid >> 4, tq = tid & 15;
  const int co0 = blockIdx.x * 16, g0 = co0 / cog, ci0 = g0 * cig;
  const int gl = (co0 + tco) / cog - g0;
  int off[WGG_ACC];
#pragma unroll
  for (int t = 0; t < WGG_ACC; ++t) {
    const int f = tq + 16 * t;
    off[t] = f < wk ? (gl * cig + f / K) * lin_t + (f % K) * dil : -1;
  }
  float acc[WGG_ACC];
#pragma unroll
  for (int t = 0; t < WGG_ACC; ++t) acc[t] = 0.f;
  const int nlc = (Lout + WG_L - 1) / WG_L;
  for (int cc = 0; cc < chunks_per_block; ++cc) {
    const int chunk = blockIdx.y * chunks_per_block + cc;
    if (chunk >= B * nlc) break;
    const int b = chunk / nlc, l0 = (chunk % nlc) * WG_L, in0 = l0 * stride - pad;
    __syncthreads();
    for (int i = tid; i < 16 * WG_L; i += 256) {
      const int co = i / WG_L, l = i % WG_L;
      dys[co * (WG_L + 1) + l] = (co0 + co < CoutT && l0 + l < Lout) ? lrelu(dy[((int64_t)b * CoutT + co0 + co) * Lout + l0 + l], dy_slope) : 0.f;
    }
    for (int c = tid >> 6; c < nci; c += 4) {
      const bool cok = ci0 + c < CinT;
      const float* xr = x + ((int64_t)b * CinT + ci0 + c) * Lin;
      for (int pos = tid & 63; pos < lin_t; pos += 64) {
        const int gi = in0 + pos;
        xs[c * lin_t + pos] = (cok && gi >= 0 && gi < Lin) ? lrelu(xr[gi], x_slope) : 0.f;
      }
    }
    __syncthreads();
    const float* dr = dys + tco * (WG_L + 1);
    for (int l = 0; l < WG_L; ++l) {
      const float dv = dr[l];
#pragma unroll
      for (int t = 0; t < WGG_ACC; ++t)
        if (off[t] >= 0) acc[t] = fmaf(dv, xs[off[t] + l * stride], acc[t]);
    }
  }
  if (co0 + tco < CoutT) {
    float* o = dw + (int64_t)(co0 + tco) * wk;
#pragma unroll
    for (int t = 0; t < WGG_ACC; ++t)
      if (off[t] >= 0) atomicAdd(o + tq + 16 * t, acc[t]);
  }
}

// db[c] += sum_{b,l} dy[b][c][l];  grid (C, splits): a block sums rows b = blockIdx.y, blockIdx.y + gridDim.y, ...
__global__ __launch_bounds__(256) void conv1d_bias_grad_kernel(const float* __restrict__ dy, float* __restrict__ db, int B,
                                                               int C, int L, int lchunks) {
  __shared__ float sh[4];
  const int c = blockIdx.x;
  float s = 0.f;
  // work items = (b, chunk of L); chunk length = ceil(L / lchunks)
  const int clen = (L + lchunks - 1) / lchunks;
  for (int it = blockIdx.y; it < B * lchunks; it += gridDim.y) {
    const int b = it / lchunks, l0 = (it % lchunks) * clen, l1 = min(L, l0 + clen);
    const float* row = dy + ((int64_t)b * C + c) * L;
    for (int l = l0 + threadIdx.x; l < l1; l += 256) s += row[l];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(db + c, (sh[0] + sh[1]) + (sh[2] + sh[3]));
}

// weight norm over dim 0 (torch.nn.utils.weight_norm / parametrizations.weight_norm): w[r] = g[r] * v[r] / ||v[r]||
// zero (optional): a [rows][n] buffer cleared in the same pass -- the accumulate-into buffer of this layer's weight gradient
// (the backward used to launch one fill per weight-normed convolution for it: ~420 per VQ-VAE-GAN step)
__global__ __launch_bounds__(256) void weight_norm_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                              float* __restrict__ w, float* __restrict__ norm, float* __restrict__ zero,
                                                              int rows, int n) {
  __shared__ float sh[4];
  const int r = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) { const float t = v[(int64_t)r * n + i]; s += t * t; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  const float nr = sqrtf((sh[0] + sh[1]) + (sh[2] + sh[3]));
  if (threadIdx.x == 0) norm[r] = nr;
  const float sc = g[r] / nr;
  for (int i = threadIdx.x; i < n; i += 256) {
    w[(int64_t)r * n + i] = v[(int64_t)r * n + i] * sc;
    if (zero) zero[(int64_t)r * n + i] = 0.f;
  }
}
// dg[r] += <dw[r], v[r]> / ||v||;   dv[r] += g/||v|| * (dw[r] - v[r] * <dw[r], v[r]> / ||v||^2)
__global__ __launch_bounds__(256) void weight_norm_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ v,
                                                              const float* __restrict__ g, const float* __restrict__ norm,
                                                              float* __restrict__ dv, float* __restrict__ dg, int rows,
                                                              int n) {
  __shared__ float sh[4];
  const int r = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += dw[(int64_t)r * n + i] * v[(int64_t)r * n + i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  const float dot = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  const float nr = norm[r], gr = g[r];
  if (threadIdx.x == 0) dg[r] += dot / nr;
  const float a = gr / nr, bq = dot / (nr * nr);
  for (int i = threadIdx.x; i < n; i += 256) {
    const int64_t o = (int64_t)r * n + i;
    dv[o] += a * (dw[o] - v[o] * bq);
  }
}

// The same two kernels for ALL weight-normed layers of a network in one launch each (ABI v7): a VQ-VAE-GAN step has ~500 such
// layers and ran ~920 five-microsecond launches for them (profiles/r03_vqvae_kernel_stats.csv: 5.3 ms a step at the launch floor).
// One workgroup per weight row as before -- the per-row arithmetic and summation order are those of the single-layer kernels, so the
// results are bit-identical -- its layer found by binary search over the descriptors' first-row offsets.
__device__ __forceinline__ int wn_find(const ttts_wn_desc* __restrict__ d, int n_desc, int r) {
  int lo = 0, hi = n_desc - 1;
  while (lo < hi) {                                  // last descriptor with row_begin <= r (workgroup-uniform)
    const int mid = (lo + hi + 1) >> 1;
    if (d[mid].row_begin <= r) lo = mid; else hi = mid - 1;
  }
  return lo;
}
__global__ __launch_bounds__(256) void weight_norm_fwd_batched_kernel(const ttts_wn_desc* __restrict__ desc, int n_desc) {
  __shared__ float sh[4];
  const ttts_wn_desc d = desc[wn_find(desc, n_desc, blockIdx.x)];
  const int r = blockIdx.x - d.row_begin, n = d.n;
  const float* v = reinterpret_cast<const float*>(d.v) + (int64_t)r * n;
  float* w = reinterpret_cast<float*>(d.w) + (int64_t)r * n;
  float* zero = d.dw ? reinterpret_cast<float*>(d.dw) + (int64_t)r * n : nullptr;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) { const float t = v[i]; s += t * t; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  const float nr = sqrtf((sh[0] + sh[1]) + (sh[2] + sh[3]));
  if (threadIdx.x == 0) reinterpret_cast<float*>(d.norm)[r] = nr;
  const float sc = reinterpret_cast<const float*>(d.g)[r] / nr;
  for (int i = threadIdx.x; i < n; i += 256) {
    w[i] = v[i] * sc;
    if (zero) zero[i] = 0.f;
  }
}
__global__ __launch_bounds__(256) void weight_norm_bwd_batched_kernel(const ttts_wn_desc* __restrict__ desc, int n_desc) {
  __shared__ float sh[4];
  const ttts_wn_desc d = desc[wn_find(desc, n_desc, blockIdx.x)];
  const int r = blockIdx.x - d.row_begin, n = d.n;
  const float* v = reinterpret_cast<const float*>(d.v) + (int64_t)r * n;
  const float* dw = reinterpret_cast<const float*>(d.dw) + (int64_t)r * n;
  float* dv = reinterpret_cast<float*>(d.dv) + (int64_t)r * n;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += dw[i] * v[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  const float dot = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  const float nr = reinterpret_cast<const float*>(d.norm)[r], gr = reinterpret_cast<const float*>(d.g)[r];
  if (threadIdx.x == 0) reinterpret_cast<float*>(d.dg)[r] += dot / nr;
  const float a = gr / nr, bq = dot / (nr * nr);
  for (int i = threadIdx.x; i < n; i += 256) dv[i] += a * (dw[i] - v[i] * bq);
}

// elementwise helpers of the stack: dy_pre = dy * (1 - y^2)  (tanh output), y = lrelu(x) variants live in the convs
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                       float* __restrict__ dx, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float t = y[i];
    dx[i] = dy[i] * (1.f - t * t);
  }
}

// dx = dy * lrelu'(y)  (y = lrelu(pre) has the sign of pre for slope > 0)
__global__ __launch_bounds__(256) void lrelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                        float* __restrict__ dx, float slope, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    dx[i] = y[i] > 0.f ? dy[i] : dy[i] * slope;
}

// y = scale * (a + b + c + d) (b, c, d optional): the resblock-sum / num_kernels of the HiFi-GAN stacks and its gradient
__global__ __launch_bounds__(256) void add4_scale_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ c, const float* __restrict__ d,
                                                         float scale, float* __restrict__ y, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 v = reinterpret_cast<const float4*>(a)[i];
    if (b) { const float4 t = reinterpret_cast<const float4*>(b)[i]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    if (c) { const float4 t = reinterpret_cast<const float4*>(c)[i]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    if (d) { const float4 t = reinterpret_cast<const float4*>(d)[i]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
    reinterpret_cast<float4*>(y)[i] = v;
  }
}
__global__ __launch_bounds__(256) void add4_scale_tail_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              const float* __restrict__ c, const float* __restrict__ d,
                                                              float scale, float* __restrict__ y, int64_t begin, int64_t n) {
  const int64_t i = begin + threadIdx.x;
  if (i < n) y[i] = scale * (((a[i] + (b ? b[i] : 0.f)) + (c ? c[i] : 0.f)) + (d ? d[i] : 0.f));
}

static int set_smem_attr(const void* fn, OnceFlag& done) {
  if (done) return TTTS_OK;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return fail(TTTS_EHIP, "conv: hipFuncSetAttribute: %s", hipGetErrorString(e));
  done = true;
  return TTTS_OK;
}

}  // namespace ttts

using namespace ttts;

static int conv_check(int B, int Cin, int Lin, int Cout, int Lout, int K, int stride, int pad, int dil, int G) {
  TTTS_REQUIRE(G > 0 && Cin % G == 0 && Cout % G == 0, "conv1d: channels (%d -> %d) not divisible by groups %d", Cin, Cout, G);
  TTTS_REQUIRE(B > 0 && Cin > 0 && Lin > 0 && Cout > 0 && Lout > 0 && K > 0 && stride > 0 && dil > 0 && pad >= 0, "conv1d: bad shape");
  TTTS_REQUIRE(Lout == (Lin + 2 * pad - dil * (K - 1) - 1) / stride + 1, "conv1d: Lout %d inconsistent with Lin %d, K %d, stride %d, pad %d, dil %d", Lout, Lin, K, stride, pad, dil);
  return TTTS_OK;
}

extern "C" int ttts_conv1d_fwd_f32(const float* x, const float* w, const float* bias, const float* bbias,
                                   const float* resid, const float* gate, const float* omask, float* y, int32_t B, int32_t Cin, int32_t Lin,
                                   int32_t Cout, int32_t Lout, int32_t K, int32_t stride, int32_t pad, int32_t dil,
                                   int32_t groups, float in_slope, float gate_slope, int32_t out_act, float out_slope,
                                   float out_scale, int32_t accumulate, const ttts_conv_ctx* ctx, void* stream) {
  TTTS_REQUIRE(x && w && y, "conv1d_fwd: null pointer");
  TTTS_REQUIRE(!ctx || !ctx->workspace || aligned16(ctx->workspace), "conv1d_fwd: ctx workspace must be 16-byte aligned");
  const ConvCtx cx = conv_ctx_of(ctx);
  const DeviceHint device_hint_scope(cx.device);
  TTTS_REQUIRE(out_act >= 0 && out_act <= 2, "conv1d_fwd: out_act must be 0 (none), 1 (tanh) or 2 (leaky-relu)");
  int rc = conv_check(B, Cin, Lin, Cout, Lout, K, stride, pad, dil, groups);
  if (rc) return rc;
  if (groups == 1 && !(cx.flags & (256 | 8388608))) {   // (8388608: no thin-layer kernels, for comparison)
    bool handled = false;
    rc = conv1d_thin_fwd_try(x, w, bias, bbias, resid, gate, omask, y, B, Cin, Lin, Cout, Lout, K, stride, pad, dil, in_slope, out_act,
                             out_slope, out_scale, accumulate, as_stream(stream), &handled);
    if (rc || handled) return rc;
  }
  if (groups == 1 && !(cx.flags & 256)) {
    bool handled = false;
    rc = conv1d_mfma_try(x, w, bias, bbias, resid, gate, omask, y, B, Cout, Cin, Lin, Lout, K, stride, pad, dil, 0, in_slope,
                         gate_slope, out_act, out_slope, out_scale, accumulate, cx, as_stream(stream), &handled);
    if (rc || handled) return rc;
  }
  if (groups > 1 && !(cx.flags & (256 | 16777216))) {   // (16777216: the direct grouped kernels, for comparison)
    bool handled = false;
    rc = conv1d_grouped_fwd_mfma_try(x, w, bias, bbias, resid, gate, omask, y, B, Cin, Lin, Cout, Lout, K, stride, pad, dil, groups,
                                     in_slope, gate_slope, out_act, out_slope, out_scale, accumulate, as_stream(stream), &handled);
    if (rc || handled) return rc;
  }
  const int lin_t = (CV_LT - 1) * stride + (K - 1) * dil + 1;
  if (groups > 1 && !(cx.flags & 256)) {
    const int cig = Cin / groups, cog = Cout / groups;
    const int nci = (cog >= CV_CT ? 1 : CV_CT / cog) * cig;
    const size_t gsmem = ((size_t)nci * lin_t + (size_t)CV_CT * cig * K) * sizeof(float);
    if (cog % 4 == 0 && (CV_CT % cog == 0 || cog % CV_CT == 0) && nci <= 32 && gsmem <= 150 * 1024) {
      static OnceFlag gattr;
      rc = set_smem_attr(reinterpret_cast<const void*>(conv1d_fwd_grouped_kernel), gattr);
      if (rc) return rc;
      ConvParams p{x, w, bias, bbias, resid, omask, gate, y, B, cig, Lin, cog, Lout, K, stride, pad, dil, groups,
                   in_slope, gate_slope, out_act, out_slope, out_scale, accumulate};
      conv1d_fwd_grouped_kernel<<<dim3((unsigned)cdiv(Lout, CV_LT), (unsigned)cdiv(Cout, CV_CT), (unsigned)B), 256, gsmem, as_stream(stream)>>>(p);
      return check_launch("conv1d_fwd_grouped");
    }
  }
  const size_t smem = ((size_t)CV_CI * lin_t + (size_t)CV_CT * CV_CI * K) * sizeof(float);
  TTTS_REQUIRE(smem <= 160 * 1024, "conv1d_fwd: tile does not fit LDS (K=%d stride=%d dil=%d)", K, stride, dil);
  static OnceFlag attr;
  rc = set_smem_attr(reinterpret_cast<const void*>(conv1d_fwd_kernel), attr);
  if (rc) return rc;
  ConvParams p{x, w, bias, bbias, resid, omask, gate, y, B, Cin / groups, Lin, Cout / groups, Lout, K, stride, pad, dil, groups,
               in_slope, gate_slope, out_act, out_slope, out_scale, accumulate};
  dim3 grid((unsigned)cdiv(Lout, CV_LT), (unsigned)cdiv(Cout / groups, CV_CT), (unsigned)(B * groups));
  conv1d_fwd_kernel<<<grid, 256, smem, as_stream(stream)>>>(p);
  return check_launch("conv1d_fwd");
}

extern "C" int ttts_conv1d_fwd_dual_f32(const float* x, const float* w, const float* bias, const float* resid, const float* omask,
                                        float* y, float* y2, int32_t B, int32_t Cin, int32_t Lin, int32_t Cout, int32_t Cout1, int32_t Lout,
                                        int32_t K, int32_t pad, int32_t dil, float in_slope, int32_t accumulate2,
                                        const ttts_conv_ctx* ctx, void* stream) {
  TTTS_REQUIRE(x && w && y && y2, "conv1d_fwd_dual: null pointer");
  TTTS_REQUIRE(Cout1 > 0 && Cout1 < Cout, "conv1d_fwd_dual: need 0 < Cout1 < Cout");
  TTTS_REQUIRE(!ctx || !ctx->workspace || aligned16(ctx->workspace), "conv1d_fwd_dual: ctx workspace must be 16-byte aligned");
  const ConvCtx cx = conv_ctx_of(ctx);
  const DeviceHint device_hint_scope(cx.device);
  int rc = conv_check(B, Cin, Lin, Cout, Lout, K, 1, pad, dil, 1);
  if (rc) return rc;
  if (!(cx.flags & 256)) {
    bool handled = false;
    rc = conv1d_mfma_dual_try(x, w, bias, resid, omask, y, y2, B, Cout, Cout1, Cin, Lin, Lout, K, pad, dil, in_slope, accumulate2, cx,
                              as_stream(stream), &handled);
    if (rc || handled) return rc;
  }
  // the two single-destination launches it stands for (exact-fp32 mode, shapes the split-bf16 kernels do not take)
  rc = ttts_conv1d_fwd_f32(x, w, bias, nullptr, resid, nullptr, omask, y, B, Cin, Lin, Cout1, Lout, K, 1, pad, dil, 1, in_slope, 1.f, 0, 1.f,
                           1.f, 0, ctx, stream);
  if (rc) return rc;
  return ttts_conv1d_fwd_f32(x, w + (int64_t)Cout1 * Cin * K, bias ? bias + Cout1 : nullptr, nullptr, nullptr, nullptr, omask, y2, B, Cin, Lin,
                             Cout - Cout1, Lout, K, 1, pad, dil, 1, in_slope, 1.f, 0, 1.f, 1.f, accumulate2, ctx, stream);
}

extern "C" int ttts_conv1d_dgrad_f32(const float* dy, const float* w, const float* bias, const float* resid,
                                     const float* gate, const float* omask, float* dx, int32_t B, int32_t Cin, int32_t Lin, int32_t Cout,
                                     int32_t Lout, int32_t K, int32_t stride, int32_t pad, int32_t dil, int32_t groups,
                                     float in_slope, float gate_slope, float out_scale, int32_t accumulate,
                                     const ttts_conv_ctx* ctx, void* stream) {
  TTTS_REQUIRE(dy && w && dx, "conv1d_dgrad: null pointer");
  TTTS_REQUIRE(!ctx || !ctx->workspace || aligned16(ctx->workspace), "conv1d_dgrad: ctx workspace must be 16-byte aligned");
  const ConvCtx cx = conv_ctx_of(ctx);
  const DeviceHint device_hint_scope(cx.device);
  TTTS_REQUIRE(groups > 0 && Cin % groups == 0 && Cout % groups == 0, "conv1d_dgrad: channels not divisible by groups");
  TTTS_REQUIRE(B > 0 && Cin > 0 && Lin > 0 && Cout > 0 && Lout > 0 && K > 0 && stride > 0 && dil > 0 && pad >= 0, "conv1d_dgrad: bad shape");
  TTTS_REQUIRE(stride == 1 || dil == 1, "conv1d_dgrad: stride > 1 requires dilation 1");
  TTTS_REQUIRE((Lout - 1) * stride - 2 * pad + dil * (K - 1) + 1 <= Lin, "conv1d_dgrad: Lin too small for Lout");
  if (groups == 1 && !(cx.flags & (256 | 8388608))) {       // one output channel: a streaming kernel (conv_thin.hip)
    bool handled = false;
    int rc2 = conv1d_thin_dgrad_try(dy, w, bias, resid, gate, omask, dx, B, Cin, Lin, Cout, Lout, K, stride, pad, dil, in_slope, out_scale,
                                    accumulate, as_stream(stream), &handled);
    if (rc2 || handled) return rc2;
  }
  if (groups == 1 && stride == 1 && !(cx.flags & 256)) {
    // stride-1 data gradient == forward convolution of dy with the transposed, tap-flipped weights and pad' = dil (K-1) - pad
    bool handled = false;
    int rc2 = conv1d_mfma_try(dy, w, bias, nullptr, resid, gate, omask, dx, B, Cin, Cout, Lout, Lin, K, 1, dil * (K - 1) - pad,
                              dil, 1, in_slope, gate_slope, 0, 1.f, out_scale, accumulate, cx, as_stream(stream), &handled);
    if (rc2 || handled) return rc2;
  }
  if (groups == 1 && stride > 1 && dil == 1 && !(cx.flags & 256)) {
    bool handled = false;
    int rc2 = conv1d_dgrad_strided_mfma_try(dy, w, bias, resid, gate, omask, dx, B, Cin, Lin, Cout, Lout, K, stride, pad, in_slope,
                                            gate_slope, out_scale, accumulate, cx, as_stream(stream), &handled);
    if (rc2 || handled) return rc2;
  }
  if (groups > 1 && !(cx.flags & (256 | 16777216))) {
    bool handled = false;
    int rc2 = conv1d_grouped_dgrad_mfma_try(dy, w, bias, resid, gate, omask, dx, B, Cin, Lin, Cout, Lout, K, stride, pad, dil, groups,
                                            in_slope, gate_slope, out_scale, accumulate, as_stream(stream), &handled);
    if (rc2 || handled) return rc2;
  }
  const int lt = (CV_LT - 1 + (K - 1) * dil) / stride + 2;
  if (groups > 1 && !(cx.flags & 256)) {
    const int cig = Cin / groups, cog = Cout / groups;
    const int nco = cig <= CV_CT && cig % 4 == 0 && CV_CT % cig == 0 ? (CV_CT / cig) * cog : 0;
    const size_t gsmem = ((size_t)nco * lt + (size_t)nco * cig * K) * sizeof(float);
    if (nco > 0 && gsmem <= 150 * 1024) {
      static OnceFlag gattr;
      int rc = set_smem_attr(reinterpret_cast<const void*>(conv1d_dgrad_grouped_kernel), gattr);
      if (rc) return rc;
      ConvParams p{dy, w, bias, nullptr, resid, omask, gate, dx, B, cig, Lin, cog, Lout, K, stride, pad, dil, groups,
                   in_slope, gate_slope, 0, 1.f, out_scale, accumulate};
      conv1d_dgrad_grouped_kernel<<<dim3((unsigned)cdiv(Lin, CV_LT), (unsigned)cdiv(Cin, CV_CT), (unsigned)B), 256, gsmem, as_stream(stream)>>>(p);
      return check_launch("conv1d_dgrad_grouped");
    }
  }
  const size_t smem = ((size_t)CV_CI * lt + (size_t)CV_CI * CV_CT * K) * sizeof(float);
  TTTS_REQUIRE(smem <= 160 * 1024, "conv1d_dgrad: tile does not fit LDS");
  static OnceFlag attr;
  int rc = set_smem_attr(reinterpret_cast<const void*>(conv1d_dgrad_kernel), attr);
  if (rc) return rc;
  ConvParams p{dy, w, bias, nullptr, resid, omask, gate, dx, B, Cin / groups, Lin, Cout / groups, Lout, K, stride, pad, dil, groups,
               in_slope, gate_slope, 0, 1.f, out_scale, accumulate};
  dim3 grid((unsigned)cdiv(Lin, CV_LT), (unsigned)cdiv(Cin / groups, CV_CT), (unsigned)(B * groups));
  conv1d_dgrad_kernel<<<grid, 256, smem, as_stream(stream)>>>(p);
  return check_launch("conv1d_dgrad");
}

static int bias_grad_launch(const float* dy, float* db, int B, int C, int L, hipStream_t stream);

static int conv1d_wgrad_impl(const float* dy, const float* x, float* dw, float* db, bool* db_done, int32_t B, int32_t Cin, int32_t Lin,
                                     int32_t Cout, int32_t Lout, int32_t K, int32_t stride, int32_t pad, int32_t dil,
                                     int32_t groups, float dy_slope, float x_slope, const ttts_conv_ctx* ctx, void* stream) {
  TTTS_REQUIRE(dy && x && dw, "conv1d_wgrad: null pointer");
  TTTS_REQUIRE(!ctx || !ctx->workspace || aligned16(ctx->workspace), "conv1d_wgrad: ctx workspace must be 16-byte aligned");
  const ConvCtx cx = conv_ctx_of(ctx);
  const DeviceHint device_hint_scope(cx.device);
  TTTS_REQUIRE(groups > 0 && Cin % groups == 0 && Cout % groups == 0, "conv1d_wgrad: channels not divisible by groups");
  if (groups == 1 && !(cx.flags & (256 | 8388608))) {
    bool handled = false;
    int rc2 = conv1d_thin_wgrad_try(dy, x, dw, B, Cin, Lin, Cout, Lout, K, stride, pad, dil, dy_slope, x_slope, cx, as_stream(stream), &handled);
    if (rc2 || handled) return rc2;
  }
  if (groups == 1 && !(cx.flags & 256)) {
    bool handled = false;
    int rc2 = conv1d_wgrad_mfma_try(dy, x, dw, db, db_done, B, Cin, Lin, Cout, Lout, K, stride, pad, dil, dy_slope, x_slope,
                                    cx, as_stream(stream), &handled);
    if (rc2 || handled) return rc2;
  }
  if (groups > 1 && !(cx.flags & (256 | 16777216))) {
    bool handled = false;
    int rc2 = conv1d_grouped_wgrad_mfma_try(dy, x, dw, B, Cin, Lin, Cout, Lout, K, stride, pad, dil, groups, dy_slope, x_slope, cx,
                                            as_stream(stream), &handled);
    if (rc2 || handled) return rc2;
  }
  if (groups > 1 && !(cx.flags & 256)) {
    const int cig = Cin / groups, cog = Cout / groups;
    const int nci = (cog >= 16 ? 1 : 16 / cog) * cig;
    const int lin_g = (WG_L - 1) * stride + (K - 1) * dil + 1;
    const size_t gsmem = ((size_t)16 * (WG_L + 1) + (size_t)nci * lin_g) * sizeof(float);
    if (cig * K <= 16 * WGG_ACC && (16 % cog == 0 || cog % 16 == 0) && gsmem <= 150 * 1024) {
      static OnceFlag gattr;
      int rc = set_smem_attr(reinterpret_cast<const void*>(conv1d_wgrad_grouped_kernel), gattr);
      if (rc) return rc;
      const int chunks = B * (int)cdiv(Lout, WG_L);
      const int tiles = (int)cdiv(Cout, 16);
      const int by = (int)std::max<int64_t>(1, std::min<int64_t>(chunks, cdiv(1024, tiles)));
      const int cpb = (int)cdiv(chunks, by);
      conv1d_wgrad_grouped_kernel<<<dim3(tiles, (unsigned)cdiv(chunks, cpb)), 256, gsmem, as_stream(stream)>>>(
          dy, x, dw, B, cig, Lin, cog, Lout, K, stride, pad, dil, groups, dy_slope, x_slope, cpb);
      return check_launch("conv1d_wgrad_grouped");
    }
  }
  Cin /= groups; Cout /= groups;
  TTTS_REQUIRE(B > 0 && Cin > 0 && Lin > 0 && Cout > 0 && Lout > 0 && stride > 0 && dil > 0 && pad >= 0, "conv1d_wgrad: bad shape");
  TTTS_REQUIRE(K > 0, "conv1d_wgrad: bad K");
  const int lin_t = (WG_L - 1) * stride + (K - 1) * dil + 1;
  const size_t smem = ((size_t)WG_T * (WG_L + 1) + (size_t)WG_T * lin_t) * sizeof(float);
  static OnceFlag attr;
  int rc = set_smem_attr(reinterpret_cast<const void*>(conv1d_wgrad_kernel), attr);
  if (rc) return rc;
  const int tiles = (int)(cdiv(Cout, WG_T) * cdiv(Cin, WG_T)) * groups;
  const int chunks = B * (int)cdiv(Lout, WG_L);
  const int blocks_y = (int)std::max<int64_t>(1, std::min<int64_t>(chunks, cdiv(1024, tiles)));
  const int cpb = (int)cdiv(chunks, blocks_y);
  TTTS_REQUIRE(smem <= 160 * 1024, "conv1d_wgrad: tile does not fit LDS");
  conv1d_wgrad_kernel<<<dim3(tiles, (unsigned)cdiv(chunks, cpb), (unsigned)cdiv(K, WG_KMAX)), 256, smem, as_stream(stream)>>>(
      dy, x, dw, B, Cin, Lin, Cout, Lout, K, stride, pad, dil, dy_slope, x_slope, cpb, groups);
  return check_launch("conv1d_wgrad");
}

extern "C" int ttts_conv1d_wgrad_f32(const float* dy, const float* x, float* dw, float* db, int32_t B, int32_t Cin, int32_t Lin,
                                     int32_t Cout, int32_t Lout, int32_t K, int32_t stride, int32_t pad, int32_t dil,
                                     int32_t groups, float dy_slope, float x_slope, const ttts_conv_ctx* ctx, void* stream) {
  TTTS_REQUIRE(!db || dy_slope == 1.f, "conv1d_wgrad: the fused bias gradient needs dy_slope == 1");
  bool db_done = false;
  int rc = conv1d_wgrad_impl(dy, x, dw, db, &db_done, B, Cin, Lin, Cout, Lout, K, stride, pad, dil, groups, dy_slope, x_slope, ctx, stream);
  if (rc || !db || db_done) return rc;
  return bias_grad_launch(dy, db, B, Cout, Lout, as_stream(stream));     // the chosen kernel does not stream dy row-wise
}

static int bias_grad_launch(const float* dy, float* db, int B, int C, int L, hipStream_t stream) {
  const int lchunks = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(L, 2048), 64));
  const int splits = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)B * lchunks, cdiv(1024, C)));
  conv1d_bias_grad_kernel<<<dim3(C, splits), 256, 0, stream>>>(dy, db, B, C, L, lchunks);
  return check_launch("conv1d_bias_grad");
}

extern "C" int ttts_conv1d_bias_grad_f32(const float* dy, float* db, int32_t B, int32_t C, int32_t L, void* stream) {
  TTTS_REQUIRE(dy && db && B > 0 && C > 0 && L > 0, "conv1d_bias_grad: bad arguments");
  return bias_grad_launch(dy, db, B, C, L, as_stream(stream));
}

extern "C" int ttts_weight_norm_fwd_f32(const float* v, const float* g, float* w, float* norm, float* zero_out, int32_t rows,
                                        int32_t n, void* stream) {
  TTTS_REQUIRE(v && g && w && norm && rows > 0 && n > 0, "weight_norm_fwd: bad arguments");
  weight_norm_fwd_kernel<<<rows, 256, 0, as_stream(stream)>>>(v, g, w, norm, zero_out, rows, n);
  return check_launch("weight_norm_fwd");
}

extern "C" int ttts_weight_norm_bwd_f32(const float* dw, const float* v, const float* g, const float* norm, float* dv,
                                        float* dg, int32_t rows, int32_t n, void* stream) {
  TTTS_REQUIRE(dw && v && g && norm && dv && dg && rows > 0 && n > 0, "weight_norm_bwd: bad arguments");
  weight_norm_bwd_kernel<<<rows, 256, 0, as_stream(stream)>>>(dw, v, g, norm, dv, dg, rows, n);
  return check_launch("weight_norm_bwd");
}

extern "C" int ttts_weight_norm_fwd_batched_f32(const ttts_wn_desc* desc_dev, int32_t n_desc, int32_t total_rows, void* stream) {
  TTTS_REQUIRE(desc_dev && n_desc > 0 && total_rows > 0, "weight_norm_fwd_batched: bad arguments");
  weight_norm_fwd_batched_kernel<<<total_rows, 256, 0, as_stream(stream)>>>(desc_dev, n_desc);
  return check_launch("weight_norm_fwd_batched");
}
extern "C" int ttts_weight_norm_bwd_batched_f32(const ttts_wn_desc* desc_dev, int32_t n_desc, int32_t total_rows, void* stream) {
  TTTS_REQUIRE(desc_dev && n_desc > 0 && total_rows > 0, "weight_norm_bwd_batched: bad arguments");
  weight_norm_bwd_batched_kernel<<<total_rows, 256, 0, as_stream(stream)>>>(desc_dev, n_desc);
  return check_launch("weight_norm_bwd_batched");
}

extern "C" int ttts_tanh_bwd_f32(const float* dy, const float* y, float* dx, int64_t n, void* stream) {
  TTTS_REQUIRE(dy && y && dx && n > 0, "tanh_bwd: bad arguments");
  tanh_bwd_kernel<<<(int)std::min<int64_t>(cdiv(n, 256), 4096), 256, 0, as_stream(stream)>>>(dy, y, dx, n);
  return check_launch("tanh_bwd");
}

extern "C" int ttts_add4_scale_f32(const float* a, const float* b, const float* c, const float* d, float scale, float* y,
                                   int64_t n, void* stream) {
  TTTS_REQUIRE(a && y && n > 0, "add4_scale: bad arguments");
  TTTS_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)y) & 15) == 0, "add4_scale: pointers must be 16-byte aligned");
  const int64_t n4 = n / 4;
  if (n4) add4_scale_kernel<<<(int)std::min<int64_t>(cdiv(n4, 256), 4096), 256, 0, as_stream(stream)>>>(a, b, c, d, scale, y, n4);
  if (n4 * 4 < n) add4_scale_tail_kernel<<<1, 256, 0, as_stream(stream)>>>(a, b, c, d, scale, y, n4 * 4, n);
  return check_launch("add4_scale");
}

extern "C" int ttts_lrelu_bwd_f32(const float* dy, const float* y, float* dx, float slope, int64_t n, void* stream) {
  TTTS_REQUIRE(dy && y && dx && n > 0, "lrelu_bwd: bad arguments");
  lrelu_bwd_kernel<<<(int)std::min<int64_t>(cdiv(n, 256), 4096), 256, 0, as_stream(stream)>>>(dy, y, dx, slope, n);
  return check_launch("lrelu_bwd");
}
