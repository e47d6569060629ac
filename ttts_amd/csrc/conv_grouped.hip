// Grouped convolutions with few channels per group (DiscriminatorS: 41 taps, stride 4, groups 4 / 16 / 64 / 256 with 4 input
// channels per group) on the f32-input matrix cores -- v_mfma_f32_16x16x4_f32: exact fp32 products and accumulation, the same
// numerics as the direct kernels in conv.hip, which ran these layers at 2-9 TF/s (one workgroup per CU behind 100 KB of LDS,
// scalar inner loops).
//
// A "super-group" is 16 consecutive OUTPUT channels together with the input channels of the groups they belong to
// (nci = cig * 16 / cog, 4 or 16 here); weights of foreign groups inside a super-group are zero, so a group count that puts
// only 4 output channels in a group (the last layer) costs 4x redundant MFMA work on a layer that is small anyway.
//   forward : D[16 co][16 positions]           += A[co][r = (k, ci)]         * B[r][l]       = x[ci][l*S - pad + k]
//   dgrad   : D[16 = (ci, phase)][16 t]        += A[(ci, phase)][r = (q, co)] * B[r][t]       = dy[co][t + P - q]
//             (polyphase form: dx[ci][S t + phase] only sees the taps phase + S q; needs cig * S == 16 and pad % S == 0)
//   wgrad   : D[16 co][16 n = (ci, k)]         += A[co][l]                   * B[l][n]        = x[ci][l*S - pad + k]
// The input strip is staged in LDS de-interleaved by phase (position mod S), so that the lanes of a fragment read -- 16
// consecutive output positions -- hit consecutive addresses whatever the stride.
#include <algorithm>

#include "common.hpp"

namespace ttts {


struct GroupedParams {
  const float* x;     // fwd: input [B, G*cig, Lin]     dgrad: dy [B, G*cog, Lout]
  const float* w;     // [G*cog, cig, K]
  const float* bias; const float* bbias; const float* resid; const float* omask; const float* gate;
  float* y;           // fwd: [B, G*cog, Lout]           dgrad: dx [B, G*cig, Lin]
  int B, cig, Lin, cog, Lout, K, stride, pad, G;
  float in_slope, gate_slope; int out_act; float out_slope, out_scale; int accumulate;
};

__device__ __forceinline__ float g_lrelu(float v, float s) { return v > 0.f ? v : v * s; }
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int GP_NPT = 256;   // output positions per workgroup: 4 waves x 4 tiles of 16

// ---- forward -------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv1d_grouped_fwd_mfma_kernel(GroupedParams p) {
  extern __shared__ __attribute__((aligned(16))) float gp_smem[];
  const int S = p.stride, K = p.K, cig = p.cig, cog = p.cog;
  const int nci = cig * (16 / cog), R = nci * K, R4 = (R + 3) & ~3;
  const int IP = GP_NPT + (K - 1) / S + 1;                 // entries per (channel, phase) row
  float* xs = gp_smem;                                     // [nci][S][IP]
  float* af = xs + nci * S * IP;                           // [R4][16]
  int* boff = reinterpret_cast<int*>(af + R4 * 16);        // [R4]: r -> strip offset of (ci, k)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ln = lane & 15, lk = lane >> 4;
  const int l0 = blockIdx.x * GP_NPT, co0 = blockIdx.y * 16, b = blockIdx.z;
  const int CinT = p.G * cig, CoutT = p.G * cog;
  const int ci0 = (co0 / cog) * cig;
  const int in0 = l0 * S - p.pad;
  const float* xb = p.x + ((int64_t)b * CinT + ci0) * p.Lin;
  // (requests of a batch first, selects and LDS stores after: see the weight-gradient kernel below)
  for (int c = wave; c < nci; c += 4) {
    const float* xr = xb + (int64_t)c * p.Lin;
    const int nu = S * IP;
    for (int u0 = lane; u0 < nu; u0 += 64 * 8) {
      float t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = xr[min(max(in0 + u0 + 64 * j, 0), p.Lin - 1)];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int u = u0 + 64 * j, g = in0 + u;
        if (u < nu) xs[(c * S + u % S) * IP + u / S] = (g >= 0 && g < p.Lin) ? g_lrelu(t[j], p.in_slope) : 0.f;
      }
    }
  }
  for (int i = tid; i < R4 * 16; i += 256) {
    const int r = i >> 4, m = i & 15, k = r / nci, cl = r - k * nci;
    const int co = co0 + m, ci = ci0 + cl;
    af[i] = (r < R && co < CoutT && ci / cig == co / cog) ? p.w[((int64_t)co * cig + (ci - (co / cog) * cig)) * K + k] : 0.f;
  }
  for (int r = tid; r < R4; r += 256) {
    const int k = r / nci, cl = r - k * nci;
    boff[r] = r < R ? (cl * S + k % S) * IP + k / S : 0;
  }
  __syncthreads();
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* xw = xs + wave * 64 + ln;
#pragma unroll 2
  for (int r0 = 0; r0 < R4; r0 += 4) {
    const float a = af[(r0 + lk) * 16 + ln];
    const float* xp = xw + boff[r0 + lk];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = mfma16(a, xp[t * 16], acc[t]);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int l = l0 + wave * 64 + t * 16 + ln;
    if (l >= p.Lout) continue;
    const float om = p.omask ? p.omask[(int64_t)b * p.Lout + l] : 1.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + lk * 4 + r;
      if (co >= CoutT) continue;
      const int64_t o = ((int64_t)b * CoutT + co) * p.Lout + l;
      float v = acc[t][r] + (p.bias ? p.bias[co] : 0.f);
      if (p.bbias) v += p.bbias[(int64_t)b * CoutT + co];
      if (p.gate) v *= (p.gate[o] > 0.f ? 1.f : p.gate_slope);
      if (p.resid) v += p.resid[o];
      if (p.out_act == 1) v = tanhf(v);
      else if (p.out_act == 2) v = g_lrelu(v, p.out_slope);
      v *= om * p.out_scale;
      p.y[o] = p.accumulate ? p.y[o] + v : v;
    }
  }
}

// ---- data gradient (polyphase) ----------------------------------------------------------------------------------------------------
// one workgroup = one group, GP_NPT values of t (= GP_NPT * S output positions); rows m = ci * S + phase
__global__ __launch_bounds__(256) void conv1d_grouped_dgrad_mfma_kernel(GroupedParams p) {
  extern __shared__ __attribute__((aligned(16))) float gp_smem[];
  const int S = p.stride, K = p.K, cig = p.cig, cog = p.cog;
  const int Q = (K + S - 1) / S, P = p.pad / S, R = Q * cog, R4 = (R + 3) & ~3;
  const int TP = GP_NPT + Q - 1, TPp = TP | 1;
  float* ds = gp_smem;                                     // [cog][TPp]: dy[co][t0 + P - (Q-1) + v]
  float* af = ds + cog * TPp;                              // [R4][16]
  int* boff = reinterpret_cast<int*>(af + R4 * 16);        // [R4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ln = lane & 15, lk = lane >> 4;
  const int t0 = blockIdx.x * GP_NPT, grp = blockIdx.y, b = blockIdx.z;
  const int CinT = p.G * cig, CoutT = p.G * cog;
  const int lbase = t0 + P - (Q - 1);
  const float* dyb = p.x + ((int64_t)b * CoutT + grp * cog) * p.Lout;
  for (int c = wave; c < cog; c += 4) {
    const float* dr = dyb + (int64_t)c * p.Lout;
    for (int v0 = lane; v0 < TP; v0 += 64 * 8) {
      float t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = dr[min(max(lbase + v0 + 64 * j, 0), p.Lout - 1)];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int v = v0 + 64 * j, l = lbase + v;
        if (v < TP) ds[c * TPp + v] = (l >= 0 && l < p.Lout) ? g_lrelu(t[j], p.in_slope) : 0.f;
      }
    }
  }
  for (int i = tid; i < R4 * 16; i += 256) {
    const int r = i >> 4, m = i & 15, q = r / cog, cl = r - q * cog, ci = m / S, k = m % S + S * q;
    af[i] = (r < R && k < K) ? p.w[((int64_t)(grp * cog + cl) * cig + ci) * K + k] : 0.f;
  }
  for (int r = tid; r < R4; r += 256) {
    const int q = r / cog, cl = r - q * cog;
    boff[r] = r < R ? cl * TPp + (Q - 1 - q) : 0;
  }
  __syncthreads();
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* dw_ = ds + wave * 64 + ln;
#pragma unroll 2
  for (int r0 = 0; r0 < R4; r0 += 4) {
    const float a = af[(r0 + lk) * 16 + ln];
    const float* dp = dw_ + boff[r0 + lk];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = mfma16(a, dp[t * 16], acc[t]);
  }
  // lane (ln, lk), register r: row m = lk*4 + r -> channel m / S, phase m % S; position j = S * t + phase
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int tt = t0 + wave * 64 + t * 16 + ln;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = lk * 4 + r, ci = grp * cig + m / S, j = tt * S + m % S;
      if (j >= p.Lin) continue;
      const int64_t o = ((int64_t)b * CinT + ci) * p.Lin + j;
      float v = acc[t][r] + (p.bias ? p.bias[ci] : 0.f);
      if (p.gate) v *= (p.gate[o] > 0.f ? 1.f : p.gate_slope);
      if (p.resid) v += p.resid[o];
      if (p.omask) v *= p.omask[(int64_t)b * p.Lin + j];
      v *= p.out_scale;
      p.y[o] = p.accumulate ? p.y[o] + v : v;
    }
  }
}

// ---- weight gradient -------------------------------------------------------------------------------------------------------------
// workgroup = (super-group, split of the (batch element, 256-position chunk) list); NT = 16-column tiles of n = (ci, k);
// wave w owns positions w*64 .. +63 of every chunk; partial tiles of the four waves meet in LDS, one slab store per workgroup
constexpr int GW_LC = 256;
template <int NT>
__global__ __launch_bounds__(256) void conv1d_grouped_wgrad_mfma_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                        float* __restrict__ slab, int B, int cig, int Lin, int cog,
                                                                        int Lout, int K, int S, int pad, int G, float dy_slope,
                                                                        float x_slope, int chunks_per_block) {
  extern __shared__ __attribute__((aligned(16))) float gp_smem[];
  const int nci = cig * (16 / cog), N = nci * K;
  const int IP0 = GW_LC + (K - 1) / S + 1, IP = IP0 + ((8 - IP0 % 32) + 32) % 32;     // phase-row pitch = 8 mod 32: the 16 (ci, k)
  constexpr int DP = GW_LC + 1;                                                        // columns of a B read spread over the banks
  float* dys = gp_smem;                                    // [16][DP]
  float* xs = dys + 16 * DP;                               // [nci][S][IP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ln = lane & 15, lk = lane >> 4;
  const int co0 = blockIdx.x * 16, split = blockIdx.y;
  const int CinT = G * cig, CoutT = G * cog, ci0 = (co0 / cog) * cig;
  const int nlc = (Lout + GW_LC - 1) / GW_LC, nchunks = B * nlc;
  int noff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = t * 16 + ln, cl = min(n, N - 1) / K, k = min(n, N - 1) - cl * K;
    noff[t] = (cl * S + k % S) * IP + k / S;
  }
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int cc = 0; cc < chunks_per_block; ++cc) {
    const int chunk = split * chunks_per_block + cc;
    if (chunk >= nchunks) break;
    const int b = chunk / nlc, l0 = (chunk % nlc) * GW_LC, in0 = l0 * S - pad;
    __syncthreads();
    // staging: the requests of a batch first (unconditional, clamped), selects and LDS stores after.  Round 6 (tools/isa_scan.py): as
    // `lds[..] = ok ? lrelu(row[..]) : 0` per element the compiler waited for every single request before the next -- a wave's ~70
    // row pieces per chunk were 70 dependent round trips (333 us for the k41 layer).  Same values: bit-identical.
    for (int c = wave; c < 16; c += 4) {
      const bool cok = co0 + c < CoutT;
      const float* dr = dy + ((int64_t)b * CoutT + min(co0 + c, CoutT - 1)) * Lout;
      float t[GW_LC / 64];
#pragma unroll
      for (int j = 0; j < GW_LC / 64; ++j) t[j] = dr[min(l0 + lane + 64 * j, Lout - 1)];
#pragma unroll
      for (int j = 0; j < GW_LC / 64; ++j) {
        const int v = lane + 64 * j, l = l0 + v;
        dys[c * DP + v] = (cok && l < Lout) ? g_lrelu(t[j], dy_slope) : 0.f;
      }
    }
    for (int c = wave; c < nci; c += 4) {
      const float* xr = x + ((int64_t)b * CinT + ci0 + c) * Lin;
      const int nu = S * IP0;
      for (int u0 = lane; u0 < nu; u0 += 64 * 8) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = xr[min(max(in0 + u0 + 64 * j, 0), Lin - 1)];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int u = u0 + 64 * j, g = in0 + u;
          if (u < nu) xs[(c * S + u % S) * IP + u / S] = (g >= 0 && g < Lin) ? g_lrelu(t[j], x_slope) : 0.f;
        }
      }
    }
    __syncthreads();
    const float* ar = dys + ln * DP + wave * 64 + lk;
    const float* br = xs + wave * 64 + lk;
#pragma unroll 2
    for (int lp = 0; lp < 64; lp += 4) {
      const float a = ar[lp];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = mfma16(a, br[noff[t] + lp], acc[t]);
    }
  }
  // cross-wave sum in wave order through LDS (the staging buffers are free), then the slab store in dw's own layout
  float* red = gp_smem;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(t * 4 + r) * 64 + lane] = acc[t][r];
  }
  __syncthreads();
  if (wv == 1) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(t * 4 + r) * 64 + lane] += acc[t][r];
  }
  __syncthreads();
  if (wv == 2) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(t * 4 + r) * 64 + lane] += acc[t][r];
  }
  __syncthreads();
  if (wv != 3) return;
  float* sl = slab + (int64_t)split * CoutT * cig * K;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = t * 16 + ln;
    if (n >= N) continue;
    const int cl = n / K, k = n - cl * K, ci = ci0 + cl;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + lk * 4 + r;
      if (co < CoutT && ci / cig == co / cog)
        sl[((int64_t)co * cig + (ci - (co / cog) * cig)) * K + k] = acc[t][r] + red[(t * 4 + r) * 64 + lane];
    }
  }
}

// dw[i] += sum over splits, fixed order
__global__ __launch_bounds__(256) void grouped_slab_sum_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nsplit,
                                                               int64_t per) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int sp = 0;
    for (; sp + 4 <= nsplit; sp += 4) {
      const float v0 = slab[sp * per + i], v1 = slab[(sp + 1) * per + i], v2 = slab[(sp + 2) * per + i], v3 = slab[(sp + 3) * per + i];
      a0 += v0; a1 += v1; a2 += v2; a3 += v3;
    }
    for (; sp < nsplit; ++sp) a0 += slab[sp * per + i];
    dw[i] += (a0 + a1) + (a2 + a3);
  }
}

// ---- dispatch (conv.hip) -------------------------------------------------------------------------------------------------------
// ---- direct kernels for 4 input channels per group (round 4) ----------------------------------------------------------------------
// All four grouped DiscriminatorS layers have cig = 4.  The MFMA forms above spend their time on padding there: the 256-group
// layer (cog = 4, 80 output positions) fills a quarter of its 16 x 16 weight blocks and 80 of a workgroup's 256 positions -- 364 us
// for 0.9 GFLOP and 52 MB.  Here a thread owns a quad of output channels at one position: per (tap, input channel) one LDS read of
// x (phase-de-interleaved strip, consecutive lanes = consecutive positions), one 16-byte broadcast read of the quad's four weights
// and four fmaf.  QW = 256 / LT quads x LT positions per workgroup, LT chosen so that the row length wastes the least.
struct G4Params {
  const float* x; const float* w; const float* bias; float* y;
  int B, G, cog, Lin, Lout, K, S, pad;
  float in_slope; int out_act; float out_slope, out_scale;
};
template <int LT>
__global__ __launch_bounds__(256) void conv1d_g4_fwd_kernel(G4Params p) {
  constexpr int QW = 256 / LT;
  extern __shared__ __attribute__((aligned(16))) float g4_smem[];
  const int S = p.S, K = p.K, IP = LT + (K - 1) / S + 1;
  float* xs = g4_smem;                                      // [QW][4 ci][S][IP]
  float* ws = xs + (size_t)QW * 4 * S * IP;                 // [QW][K][4 ci][4 co]
  const int tid = threadIdx.x, q = tid / LT, l = tid % LT;
  const int l0 = blockIdx.x * LT, b = blockIdx.z, nquad = p.G * p.cog / 4;
  const int qg0 = blockIdx.y * QW;
  const int in0 = l0 * S - p.pad, strip = S * IP;
  // x strips, one per quad (quads of one group stage the same rows again: at most 16 x 4 x S x IP floats)
  for (int i = tid; i < QW * 4 * strip; i += 256) {
    const int u = i % strip, c = (i / strip) & 3, qq = i / (4 * strip);
    const int qg = min(qg0 + qq, nquad - 1), grp = (qg * 4) / p.cog, gpos = in0 + u;
    const float v = p.x[((int64_t)b * p.G * 4 + grp * 4 + c) * p.Lin + min(max(gpos, 0), p.Lin - 1)];
    xs[((qq * 4 + c) * S + u % S) * IP + u / S] = (gpos >= 0 && gpos < p.Lin) ? g_lrelu(v, p.in_slope) : 0.f;
  }
  for (int i = tid; i < QW * K * 16; i += 256) {
    const int j = i & 3, c = (i >> 2) & 3, k = (i >> 4) % K, qq = i / (16 * K);
    const int qg = min(qg0 + qq, nquad - 1);
    ws[i] = p.w[((int64_t)(qg * 4 + j) * 4 + c) * K + k];
  }
  __syncthreads();
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const float* xq = xs + (size_t)q * 4 * strip + l;
  const float* wq = ws + (size_t)q * K * 16;
  for (int ph = 0; ph < S; ++ph) {
    for (int k = ph, d = 0; k < K; k += S, ++d) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float xv = xq[(c * S + ph) * IP + d];
        const float4 w4 = *reinterpret_cast<const float4*>(wq + (k * 4 + c) * 4);
        acc[0] = fmaf(w4.x, xv, acc[0]); acc[1] = fmaf(w4.y, xv, acc[1]);
        acc[2] = fmaf(w4.z, xv, acc[2]); acc[3] = fmaf(w4.w, xv, acc[3]);
      }
    }
  }
  const int qg = qg0 + q, lo = l0 + l;
  if (qg >= nquad || lo >= p.Lout) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int co = qg * 4 + j;
    float v = acc[j] + (p.bias ? p.bias[co] : 0.f);
    if (p.out_act == 1) v = tanhf(v);
    else if (p.out_act == 2) v = g_lrelu(v, p.out_slope);
    p.y[((int64_t)b * p.G * p.cog + co) * p.Lout + lo] = v * p.out_scale;
  }
}

// weight gradient, cig = 4: one workgroup = one quad of output channels x one (batch element, 256-position chunk) slice range; thread
// = (co of the quad, ci, tap) for 16 K <= 768 combinations, looping over the positions of its slices with dy and the x strip in LDS.
// Partial sums per slice range go to a slab [range][Cout][4][K]; g4_slab_sum_kernel adds them in range order (deterministic).
constexpr int G4W_LC = 256;
__global__ __launch_bounds__(256) void conv1d_g4_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              float* __restrict__ slab, int B, int G, int cog, int Lin, int Lout,
                                                              int K, int S, int pad, float dy_slope, float x_slope, int cpb) {
  extern __shared__ __attribute__((aligned(16))) float g4_smem[];
  const int XW = (G4W_LC - 1) * S + K;                      // input positions a chunk of outputs touches
  float* ds = g4_smem;                                      // [4 co][G4W_LC]
  float* xs = ds + 4 * G4W_LC;                              // [4 ci][XW]
  const int tid = threadIdx.x, qg = blockIdx.x, rng = blockIdx.y;
  const int grp = (qg * 4) / cog, nlc = (Lout + G4W_LC - 1) / G4W_LC, nchunks = B * nlc;
  const int ncomb = 16 * K;
  float acc[3] = {0.f, 0.f, 0.f};                           // combos tid, tid + 256, tid + 512 (K <= 48)
  for (int ch = rng * cpb; ch < min(nchunks, (rng + 1) * cpb); ++ch) {
    const int b = ch / nlc, l0 = (ch - b * nlc) * G4W_LC, nl = min(G4W_LC, Lout - l0), in0 = l0 * S - pad;
    __syncthreads();
    for (int i = tid; i < 4 * G4W_LC; i += 256) {
      const int j = i / G4W_LC, ll = i % G4W_LC;
      ds[i] = ll < nl ? g_lrelu(dy[((int64_t)b * G * cog + qg * 4 + j) * Lout + l0 + ll], dy_slope) : 0.f;
    }
    for (int i = tid; i < 4 * XW; i += 256) {
      const int c = i / XW, u = i % XW, gpos = in0 + u;
      const float v = x[((int64_t)b * G * 4 + grp * 4 + c) * Lin + min(max(gpos, 0), Lin - 1)];
      xs[i] = (gpos >= 0 && gpos < Lin) ? g_lrelu(v, x_slope) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int cb = tid + 256 * r;
      if (cb >= ncomb) break;
      const int k = cb % K, c = (cb / K) & 3, j = cb / (4 * K);
      const float* dr = ds + j * G4W_LC;
      const float* xr = xs + c * XW + k;
      float a = acc[r];
      for (int ll = 0; ll < nl; ++ll) a = fmaf(dr[ll], xr[ll * S], a);
      acc[r] = a;
    }
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int cb = tid + 256 * r;
    if (cb >= ncomb) break;
    const int k = cb % K, c = (cb / K) & 3, j = cb / (4 * K);
    slab[((int64_t)rng * G * cog + qg * 4 + j) * 4 * K + c * K + k] = acc[r];
  }
}
__global__ __launch_bounds__(256) void g4_slab_sum_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nrng, int64_t per) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256) {
    float s = 0.f;
    for (int r = 0; r < nrng; ++r) s += slab[r * per + i];
    dw[i] += s;
  }
}

static int gp_attr(const void* fn, OnceFlag& done) {
  if (done) return TTTS_OK;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return fail(TTTS_EHIP, "conv_grouped: hipFuncSetAttribute: %s", hipGetErrorString(e));
  done = true;
  return TTTS_OK;
}

int conv1d_grouped_fwd_mfma_try(const float* x, const float* w, const float* bias, const float* bbias, const float* resid,
                                const float* gate, const float* omask, float* y, int B, int Cin, int Lin, int Cout, int Lout,
                                int K, int stride, int pad, int dil, int groups, float in_slope, float gate_slope, int out_act,
                                float out_slope, float out_scale, int accumulate, hipStream_t stream, bool* handled) {
  *handled = false;
  const int cig = Cin / groups, cog = Cout / groups;
  // (measured, B = 32: the 256-group layer 364 -> 120 us; with 16 output channels per group the MFMA kernel below stays ahead --
  // 105 vs 221 us on the 64-group layer -- so the quad kernel takes cog == 4 only)
  if (dil == 1 && cig == 4 && cog == 4 && K <= 64 && !bbias && !resid && !gate && !omask && !accumulate) {
    // direct quad kernel; position tile = the one of {16, 32, 64} that wastes the least of the row (ties: the larger)
    int LT = 64;
    for (int cand : {32, 16})
      if (cdiv(Lout, cand) * cand < cdiv(Lout, LT) * LT) LT = cand;
    const int QW = 256 / LT, IP = LT + (K - 1) / stride + 1;
    const size_t smem = ((size_t)QW * 4 * stride * IP + (size_t)QW * K * 16) * sizeof(float);
    if (smem <= 150 * 1024) {
      G4Params q{x, w, bias, y, B, groups, cog, Lin, Lout, K, stride, pad, in_slope, out_act, out_slope, out_scale};
      dim3 grid((unsigned)cdiv(Lout, LT), (unsigned)cdiv(Cout / 4, QW), (unsigned)B);
      int rc = TTTS_OK;
      if (LT == 64) { static OnceFlag a; rc = gp_attr(reinterpret_cast<const void*>(conv1d_g4_fwd_kernel<64>), a); if (!rc) conv1d_g4_fwd_kernel<64><<<grid, 256, smem, stream>>>(q); }
      else if (LT == 32) { static OnceFlag a; rc = gp_attr(reinterpret_cast<const void*>(conv1d_g4_fwd_kernel<32>), a); if (!rc) conv1d_g4_fwd_kernel<32><<<grid, 256, smem, stream>>>(q); }
      else { static OnceFlag a; rc = gp_attr(reinterpret_cast<const void*>(conv1d_g4_fwd_kernel<16>), a); if (!rc) conv1d_g4_fwd_kernel<16><<<grid, 256, smem, stream>>>(q); }
      if (rc) return rc;
      *handled = true;
      return check_launch("conv1d_g4_fwd");
    }
  }
  if (dil != 1 || cog > 16 || 16 % cog != 0 || cig * (16 / cog) > 16 || Cout % 16 != 0) return TTTS_OK;
  const int nci = cig * (16 / cog), R4 = (nci * K + 3) & ~3, IP = GP_NPT + (K - 1) / stride + 1;
  const size_t smem = ((size_t)nci * stride * IP + (size_t)R4 * 16 + R4) * sizeof(float);
  if (smem > 150 * 1024) return TTTS_OK;
  static OnceFlag attr;
  int rc = gp_attr(reinterpret_cast<const void*>(conv1d_grouped_fwd_mfma_kernel), attr);
  if (rc) return rc;
  GroupedParams p{x, w, bias, bbias, resid, omask, gate, y, B, cig, Lin, cog, Lout, K, stride, pad, groups,
                  in_slope, gate_slope, out_act, out_slope, out_scale, accumulate};
  conv1d_grouped_fwd_mfma_kernel<<<dim3((unsigned)cdiv(Lout, GP_NPT), (unsigned)(Cout / 16), (unsigned)B), 256, smem, stream>>>(p);
  *handled = true;
  return check_launch("conv1d_grouped_fwd_mfma");
}

int conv1d_grouped_dgrad_mfma_try(const float* dy, const float* w, const float* bias, const float* resid, const float* gate,
                                  const float* omask, float* dx, int B, int Cin, int Lin, int Cout, int Lout, int K, int stride,
                                  int pad, int dil, int groups, float in_slope, float gate_slope, float out_scale, int accumulate,
                                  hipStream_t stream, bool* handled) {
  *handled = false;
  const int cig = Cin / groups, cog = Cout / groups;
  if (dil != 1 || cig * stride != 16 || pad % stride != 0 || cog > 64) return TTTS_OK;
  const int Q = (K + stride - 1) / stride, R4 = (Q * cog + 3) & ~3, TPp = (GP_NPT + Q - 1) | 1;
  const size_t smem = ((size_t)cog * TPp + (size_t)R4 * 16 + R4) * sizeof(float);
  if (smem > 150 * 1024) return TTTS_OK;
  static OnceFlag attr;
  int rc = gp_attr(reinterpret_cast<const void*>(conv1d_grouped_dgrad_mfma_kernel), attr);
  if (rc) return rc;
  GroupedParams p{dy, w, bias, nullptr, resid, omask, gate, dx, B, cig, Lin, cog, Lout, K, stride, pad, groups,
                  in_slope, gate_slope, 0, 1.f, out_scale, accumulate};
  const int T = (int)cdiv(Lin, stride);
  conv1d_grouped_dgrad_mfma_kernel<<<dim3((unsigned)cdiv(T, GP_NPT), (unsigned)groups, (unsigned)B), 256, smem, stream>>>(p);
  *handled = true;
  return check_launch("conv1d_grouped_dgrad_mfma");
}

int conv1d_grouped_wgrad_mfma_try(const float* dy, const float* x, float* dw, int B, int Cin, int Lin, int Cout, int Lout, int K,
                                  int stride, int pad, int dil, int groups, float dy_slope, float x_slope, const ConvCtx& cx,
                                  hipStream_t stream, bool* handled) {
  *handled = false;
  const int cig = Cin / groups, cog = Cout / groups;
  if (dil == 1 && cig == 4 && cog == 4 && K <= 48 && cx.ws) {       // (256-group layer, B = 64: 670 -> 171 us; cog = 16: MFMA kernel)
    const int nchunks = B * (int)cdiv(Lout, G4W_LC), nquad = Cout / 4;
    const int ranges = (int)std::max<int64_t>(1, std::min<int64_t>(nchunks, cdiv(2048, nquad)));
    const int cpb = (int)cdiv(nchunks, ranges), nrng = (int)cdiv(nchunks, cpb);
    const int64_t per = (int64_t)Cout * 4 * K;
    const size_t smem = ((size_t)4 * G4W_LC + (size_t)4 * ((G4W_LC - 1) * stride + K)) * sizeof(float);
    if ((int64_t)nrng * per * (int64_t)sizeof(float) <= cx.ws_bytes && smem <= 150 * 1024) {
      static OnceFlag a;
      int rc = gp_attr(reinterpret_cast<const void*>(conv1d_g4_wgrad_kernel), a);
      if (rc) return rc;
      float* slab = static_cast<float*>(cx.ws);
      conv1d_g4_wgrad_kernel<<<dim3((unsigned)nquad, (unsigned)nrng), 256, smem, stream>>>(dy, x, slab, B, groups, cog, Lin, Lout, K, stride,
                                                                                        pad, dy_slope, x_slope, cpb);
      g4_slab_sum_kernel<<<(unsigned)std::min<int64_t>(cdiv(per, 256), 4096), 256, 0, stream>>>(slab, dw, nrng, per);
      *handled = true;
      return check_launch("conv1d_g4_wgrad");
    }
  }
  if (dil != 1 || cog > 16 || 16 % cog != 0 || cig * (16 / cog) > 16 || Cout % 16 != 0 || !cx.ws) return TTTS_OK;
  const int nci = cig * (16 / cog), NT = (int)cdiv(nci * K, 16);
  if (NT != 11 && NT != 41) return TTTS_OK;
  const int IP0 = GW_LC + (K - 1) / stride + 1, IP = IP0 + ((8 - IP0 % 32) + 32) % 32;
  const size_t stage = ((size_t)16 * (GW_LC + 1) + (size_t)nci * stride * IP) * sizeof(float), red = (size_t)NT * 4 * 64 * sizeof(float);
  const size_t smem = std::max(stage, red);
  if (smem > 150 * 1024) return TTTS_OK;
  const int chunks = B * (int)cdiv(Lout, GW_LC), sgs = Cout / 16;
  const int splits = (int)std::max<int64_t>(1, std::min<int64_t>(chunks, cdiv(768, sgs)));
  const int cpb = (int)cdiv(chunks, splits), nsplit = (int)cdiv(chunks, cpb);
  const int64_t per = (int64_t)Cout * cig * K;
  if ((int64_t)nsplit * per * (int64_t)sizeof(float) > cx.ws_bytes) return TTTS_OK;
  float* slab = static_cast<float*>(cx.ws);
  dim3 grid((unsigned)sgs, (unsigned)nsplit);
  if (NT == 11) {
    static OnceFlag a;
    int rc = gp_attr(reinterpret_cast<const void*>(conv1d_grouped_wgrad_mfma_kernel<11>), a);
    if (rc) return rc;
    conv1d_grouped_wgrad_mfma_kernel<11><<<grid, 256, smem, stream>>>(dy, x, slab, B, cig, Lin, cog, Lout, K, stride, pad, groups, dy_slope, x_slope, cpb);
  } else {
    static OnceFlag a;
    int rc = gp_attr(reinterpret_cast<const void*>(conv1d_grouped_wgrad_mfma_kernel<41>), a);
    if (rc) return rc;
    conv1d_grouped_wgrad_mfma_kernel<41><<<grid, 256, smem, stream>>>(dy, x, slab, B, cig, Lin, cog, Lout, K, stride, pad, groups, dy_slope, x_slope, cpb);
  }
  grouped_slab_sum_kernel<<<(unsigned)std::min<int64_t>(cdiv(per, 256), 4096), 256, 0, stream>>>(slab, dw, nsplit, per);
  *handled = true;
  return check_launch("conv1d_grouped_wgrad_mfma");
}

}  // namespace ttts
