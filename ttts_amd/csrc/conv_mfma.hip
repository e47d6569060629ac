// Implicit-GEMM 1-D convolution on the f32-input matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products and
// accumulation, bit-for-bit an fmaf chain -- the same numerics as the direct kernels in conv.hip at ~20x their rate).
//
//   forward / stride-1 data gradient:   out[b][m][j] = sum_{n,k} A[m][n][k] * in[b][n][j*stride - pad + k*dil]
//     GEMM view per batch element: M = output channels, N = output positions, K = input channels x taps.  The weight
//     slab [64 m][8 n x K taps] and the matching input strip [8 n][positions] are staged in LDS; the im2col matrix is never
//     materialised -- a lane reads in_s[n][col*stride + k*dil] through a small (n, k) -> offset table.
//     dgrad (stride 1) is the same kernel with A[m = ci][n = co][k'] = w[co][ci][K-1-k'] and pad' = dil (K-1) - pad.
//   weight gradient:  dw[co][ci*K + k] += sum_{b,l} dy[b][co][l] * x[b][ci][l*stride - pad + k*dil]
//     GEMM with the reduction over positions: A = dy strip [64 co][64 l], B = x strip gathered per (ci, k) column.
//
// Workgroup = 4 waves; wave tile 32 x 64 (two 32x32 accumulators).  Accumulator layout: lane l holds column (l & 31) and
// rows (r & 3) + 8 (r >> 2) + 4 (l >> 5) -- so a register row is 32 consecutive output positions: coalesced stores and the
// same fused epilogue (bias, per-sample bias, leaky-relu gate, residual, tanh / leaky-relu, sequence mask, scale,
// accumulate) as the direct kernels.
#include <algorithm>

#include "common.hpp"

namespace ttts {

struct ConvMfmaParams {
  const float* x;      // input  [B, N, Lin]
  const float* w;      // weights [Cout, Cin, K] (forward: M = Cout, N = Cin; transposed: M = Cin, N = Cout)
  const float* bias; const float* bbias; const float* resid; const float* omask; const float* gate;
  float* y;            // output [B, M, Lout]
  int B, M, N, Lin, Lout, K, stride, pad, dil;
  int transposed;      // 1: A[m][n][k] = w[n][m][K-1-k]  (stride-1 data gradient)
  int NT;              // input channels per LDS stage (multiple of 8)
  // polyphase view used by strided data gradients / transposed convolutions (identity: Kmem = K, 0, 1, 1, 0, Lout)
  int Kmem;            // taps per (m, n) pair in memory
  int tap_off, tap_stride;   // effective tap q reads memory tap tap_off + tap_stride * q
  int out_stride, out_off;   // local output index t writes position out_off + t * out_stride
  int LoutTotal;       // row length of the output tensor
  int SEG;             // positions per batch segment of a tile: LT (one batch element per tile) or 32 / 64 -- short rows
                       // (DiscriminatorP's late layers: 23..127 positions) fold several batch elements into one tile
  float in_slope, gate_slope;
  int out_act; float out_slope, out_scale; int accumulate;
};


__device__ __forceinline__ float lrelu_f(float v, float s) { return v > 0.f ? v : v * s; }

// WCO = waves along the output-channel axis (1 or 2); the other 4 / WCO waves tile positions.
template <int WCO>
__global__ __launch_bounds__(256) void conv1d_mfma_kernel(ConvMfmaParams p) {
  constexpr int MT = 32 * WCO;                 // output channels per workgroup
  constexpr int WL = 4 / WCO;                  // waves along positions
  constexpr int LT = 64 * WL;                  // positions per workgroup
  extern __shared__ __attribute__((aligned(16))) float cm_smem[];
  const int NT = p.NT;
  const int KK = NT * p.K;                     // reduction indices per stage (even)
  const int wpitch = KK | 1;                   // odd pitch: conflict-free A-fragment reads
  const int SEG = p.SEG, nseg = LT / SEG;
  const int lin_s = (SEG - 1) * p.stride + (p.K - 1) * p.dil + 1;      // input strip of one segment
  const int lin_t = nseg * lin_s;
  float* xs = cm_smem;                         // [NT][nseg][lin_s]
  float* ws = xs + NT * lin_t;                 // [MT][wpitch]
  int* foff = reinterpret_cast<int*>(ws + MT * wpitch);   // [KK]: (n_local, k) -> n_local * lin_t + k * dil
  int* tmap = foff + KK;                                  // transposed loader: i = m*K + k -> m * wpitch + (K-1-k)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, col = lane & 31;
  const int wco = wave % WCO, wl = wave / WCO;
  const int j0 = blockIdx.x * LT, m0 = blockIdx.y * MT, b0 = blockIdx.z * nseg;   // folded tiles: gridDim.x == 1, j0 == 0
  const int in0 = j0 * p.stride - p.pad;
  for (int f = tid; f < KK; f += 256) foff[f] = (f / p.K) * lin_t + (f % p.K) * p.dil;
  if (p.transposed)
    for (int i = tid; i < MT * p.Kmem; i += 256) {
      const int km = i % p.Kmem - p.tap_off;          // memory tap -> effective tap q (if on this phase) -> flipped slot
      tmap[i] = (km >= 0 && km % p.tap_stride == 0 && km / p.tap_stride < p.K) ? (i / p.Kmem) * wpitch + (p.K - 1 - km / p.tap_stride) : -1;
    }
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  const float* xb = p.x + (int64_t)b0 * p.N * p.Lin;
  const float* arow = ws + (wco * 32 + col) * wpitch + hh;
  // column c of the tile -> segment c / SEG (a 32-column MFMA tile never straddles segments), position c % SEG
  const int c0 = wl * 64 + col, c1 = c0 + 32;
  const int bpos0 = (c0 / SEG) * lin_s + (c0 % SEG) * p.stride, bpos1 = (c1 / SEG) * lin_s + (c1 % SEG) * p.stride;
  const int lrow = tid / (256 / MT), lq = tid % (256 / MT);     // weight loader: 256 / MT threads per output-channel row
  for (int n0 = 0; n0 < p.N; n0 += NT) {
    __syncthreads();
    // input strip: one wave per channel row at a time, lanes along positions (coalesced, no index arithmetic)
    for (int n = wave; n < NT; n += 4) {
      const bool nok = n0 + n < p.N;
      for (int sg = 0; sg < nseg; ++sg) {
        const bool bok = nok && b0 + sg < p.B;
        const float* xr = xb + ((int64_t)sg * p.N + n0 + n) * p.Lin;
        float* xd = xs + n * lin_t + sg * lin_s;
        for (int pos = lane; pos < lin_s; pos += 64) {
          const int gi = in0 + pos;
          xd[pos] = (bok && gi >= 0 && gi < p.Lin) ? lrelu_f(xr[gi], p.in_slope) : 0.f;
        }
      }
    }
    {
      const bool mok = m0 + lrow < p.M;
      float* wd = ws + lrow * wpitch;
      if (!p.transposed) {
        // A[m][n][k] = w[m][n][k]: the stage's KK values of a row are contiguous in memory
        const float* wr = p.w + ((int64_t)(m0 + lrow) * p.N + n0) * p.K;
        const int flim = mok ? min(KK, (p.N - n0) * p.K) : 0;
        for (int f = lq; f < KK; f += 256 / MT) wd[f] = f < flim ? wr[f] : 0.f;
      } else {
        // A[m][n][k] = w[n][m][K-1-k]  (w is [N][M][K] here): for one n the MT*K values w[n][m0 .. m0+MT)[:] are contiguous
        const int cnt = min(MT, p.M - m0) * p.Kmem;
        for (int n = wave; n < NT; n += 4) {
          const bool nok = n0 + n < p.N;
          const float* wr = p.w + ((int64_t)(n0 + n) * p.M + m0) * p.Kmem;
          for (int i = lane; i < MT * p.Kmem; i += 64) {
            const int slot = tmap[i];
            if (slot >= 0) ws[slot + n * p.K] = (nok && i < cnt) ? wr[i] : 0.f;
          }
        }
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int f = 0; f < KK; f += 2) {
      const float a = arow[f];
      const int o = foff[f + hh];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xs[o + bpos0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xs[o + bpos1], acc1, 0, 0, 0);
    }
  }
  // ---- epilogue
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int ct = wl * 64 + t * 32 + col;
    const int b = b0 + ct / SEG, jt = j0 + ct % SEG;
    if (jt >= p.Lout || b >= p.B) continue;
    const int j = p.out_off + jt * p.out_stride;
    const float om = p.omask ? p.omask[(int64_t)b * p.LoutTotal + j] : 1.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wco * 32 + acc_row(r, hh);
      if (m >= p.M) continue;
      const int64_t o = ((int64_t)b * p.M + m) * p.LoutTotal + j;
      float v = (t == 0 ? acc0[r] : acc1[r]) + (p.bias ? p.bias[m] : 0.f);
      if (p.bbias) v += p.bbias[(int64_t)b * p.M + m];
      if (p.gate) v *= (p.gate[o] > 0.f ? 1.f : p.gate_slope);
      if (p.resid) v += p.resid[o];
      if (p.out_act == 1) v = tanhf(v);
      else if (p.out_act == 2) v = lrelu_f(v, p.out_slope);
      v *= om * p.out_scale;
      p.y[o] = p.accumulate ? p.y[o] + v : v;
    }
  }
}

// ---- weight gradient ---------------------------------------------------------------------------------------------------
// dw[co][n] += sum_{b, l} lrelu(dy[b][co][l]) * lrelu(x[b][n / K][l*stride - pad + (n % K)*dil]),  n in [0, Cin*K)
// workgroup tile 64 co x 64 n; waves 2 (co) x 2 (n); reduction chunk = 64 positions of one batch element per stage.
struct WgradMfmaParams {
  const float* dy; const float* x; float* dw;
  int B, Cin, Lin, Cout, Lout, K, stride, pad, dil;
  float dy_slope, x_slope;
  int chunks_per_block;
};
constexpr int WM_L = 64;

__global__ __launch_bounds__(256) void conv1d_wgrad_mfma_kernel(WgradMfmaParams p) {
  extern __shared__ __attribute__((aligned(16))) float cm_smem[];
  const int NK = p.Cin * p.K;
  const int n0 = blockIdx.x * 64, co0 = blockIdx.y * 64;
  const int ci_first = n0 / p.K, ci_last = min(p.Cin - 1, (n0 + 63) / p.K), nch = ci_last - ci_first + 1;
  const int lin_t = (WM_L - 1) * p.stride + (p.K - 1) * p.dil + 1;
  const int xpitch = lin_t | 1;
  float* dys = cm_smem;                        // [64 co][WM_L + 1]
  float* xs = cm_smem + 64 * (WM_L + 1);       // [nch][xpitch]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, col = lane & 31;
  const int wco = wave & 1, wn = wave >> 1;
  const int n = n0 + wn * 32 + col;            // this lane's output column (ci, k)
  const int ci = n / p.K, k = n - ci * p.K;
  const bool ncol_ok = n < NK;
  const int boff = ncol_ok ? (ci - ci_first) * xpitch + k * p.dil : 0;
  const float* arow = dys + (wco * 32 + col) * (WM_L + 1);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int nlc = (p.Lout + WM_L - 1) / WM_L;
  for (int cc = 0; cc < p.chunks_per_block; ++cc) {
    const int chunk = blockIdx.z * p.chunks_per_block + cc;
    if (chunk >= p.B * nlc) break;
    const int b = chunk / nlc, l0 = (chunk % nlc) * WM_L, in0 = l0 * p.stride - p.pad;
    __syncthreads();
    for (int i = tid; i < 64 * WM_L; i += 256) {
      const int co = i / WM_L, l = i % WM_L;
      dys[co * (WM_L + 1) + l] = (co0 + co < p.Cout && l0 + l < p.Lout)
                                     ? lrelu_f(p.dy[((int64_t)b * p.Cout + co0 + co) * p.Lout + l0 + l], p.dy_slope) : 0.f;
    }
    for (int c = wave; c < nch; c += 4) {
      const float* xr = p.x + ((int64_t)b * p.Cin + ci_first + c) * p.Lin;
      for (int pos = lane; pos < lin_t; pos += 64) {
        const int gi = in0 + pos;
        xs[c * xpitch + pos] = (gi >= 0 && gi < p.Lin) ? lrelu_f(xr[gi], p.x_slope) : 0.f;
      }
    }
    __syncthreads();
    const float* bcol = xs + boff + hh * p.stride;
#pragma unroll 4
    for (int l = 0; l < WM_L; l += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[l + hh], ncol_ok ? bcol[l * p.stride] : 0.f, acc, 0, 0, 0);
  }
  if (ncol_ok) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wco * 32 + acc_row(r, hh);
      if (co < p.Cout) atomicAdd(p.dw + (int64_t)co * NK + n, acc[r]);
    }
  }
}

static int set_attr_once(const void* fn, bool& done) {
  if (done) return TTTS_OK;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return fail(TTTS_EHIP, "conv_mfma: hipFuncSetAttribute: %s", hipGetErrorString(e));
  done = true;
  return TTTS_OK;
}

// ---- dispatch helpers used by conv.hip ------------------------------------------------------------------------------------
// Returns TTTS_OK and sets *handled when the MFMA path took the launch.
static int conv1d_mfma_launch(ConvMfmaParams p, hipStream_t stream, bool* handled) {
  *handled = false;
  const bool narrow = p.M <= 32;
  const int LT = narrow ? 256 : 128, MT = narrow ? 32 : 64;
  const int K = p.K;
  const int SEG = (p.Lout <= 32 && p.B > 1) ? 32 : ((p.Lout <= 64 && p.B > 1) ? 64 : LT);
  p.SEG = SEG;
  const int lin_t = (LT / SEG) * ((SEG - 1) * p.stride + (K - 1) * p.dil + 1);
  auto smem_for = [&](int nt) {
    return ((size_t)nt * lin_t + (size_t)MT * ((nt * K) | 1)) * sizeof(float) + ((size_t)nt * K + (size_t)MT * p.Kmem) * sizeof(int);
  };
  // aim at ~96 reduction indices per stage (amortises the barrier pair), within 48 KB so that >= 3 workgroups share a CU
  int NT = std::min(64, std::max(8, (int)cdiv(96, K) / 8 * 8));
  while (NT > 8 && (NT > (p.N + 7) / 8 * 8 || smem_for(NT) > 48 * 1024)) NT -= 8;
  const size_t smem = smem_for(NT);
  if (smem > 96 * 1024) return TTTS_OK;
  p.NT = NT;
  dim3 grid((unsigned)cdiv(p.Lout, SEG == LT ? LT : SEG), (unsigned)cdiv(p.M, MT), (unsigned)cdiv(p.B, LT / SEG));
  static bool a1 = false, a2 = false;
  if (narrow) {
    int rc = set_attr_once(reinterpret_cast<const void*>(conv1d_mfma_kernel<1>), a1);
    if (rc) return rc;
    conv1d_mfma_kernel<1><<<grid, 256, smem, stream>>>(p);
  } else {
    int rc = set_attr_once(reinterpret_cast<const void*>(conv1d_mfma_kernel<2>), a2);
    if (rc) return rc;
    conv1d_mfma_kernel<2><<<grid, 256, smem, stream>>>(p);
  }
  *handled = true;
  return check_launch("conv1d_mfma");
}

// Returns TTTS_OK and sets *handled when the MFMA path took the launch.
int conv1d_mfma_try(const float* x, const float* w, const float* bias, const float* bbias, const float* resid,
                    const float* gate, const float* omask, float* y, int B, int M, int N, int Lin, int Lout, int K,
                    int stride, int pad, int dil, int transposed, float in_slope, float gate_slope, int out_act,
                    float out_slope, float out_scale, int accumulate, hipStream_t stream, bool* handled) {
  *handled = false;
  if (N < 8 || M < 8 || K > 16) return TTTS_OK;            // thin layers / long taps stay on the direct kernels
  ConvMfmaParams p{x, w, bias, bbias, resid, omask, gate, y, B, M, N, Lin, Lout, K, stride, pad, dil, transposed, 0,
                   K, 0, 1, 1, 0, Lout, 0, in_slope, gate_slope, out_act, out_slope, out_scale, accumulate};
  return conv1d_mfma_launch(p, stream, handled);
}

// Data gradient of a stride-s convolution (== forward of a ConvTranspose1d), dilation 1, as s stride-1 sub-convolutions:
// output phase phi = (j + pad) mod s only sees the taps k = phi + s q, so
//   dx[s t' + phi - pad] = sum_{co, q} w[co][ci][phi + s q] * dy[co][t' - q].
int conv1d_dgrad_strided_mfma_try(const float* dy, const float* w, const float* bias, const float* resid,
                                  const float* gate, const float* omask, float* dx, int B, int Cin, int Lin, int Cout,
                                  int Lout, int K, int stride, int pad, float in_slope, float gate_slope, float out_scale,
                                  int accumulate, hipStream_t stream, bool* handled) {
  *handled = false;
  if (Cout < 8 || Cin < 8 || K > 16 * stride || K < stride) return TTTS_OK;
  for (int phi = 0; phi < stride; ++phi) {
    const int Kp = (K - phi + stride - 1) / stride;                       // taps of this phase (>= 1 since K >= stride)
    const int tmin = phi >= pad ? 0 : (pad - phi + stride - 1) / stride;  // first t' with a non-negative output position
    const int off = stride * tmin + phi - pad;
    if (off >= Lin) continue;
    const int T = (Lin - 1 - off) / stride + 1;
    ConvMfmaParams p{dy, w, bias, nullptr, resid, omask, gate, dx, B, Cin, Cout, Lout, T, Kp, 1, (Kp - 1) - tmin, 1, 1, 0,
                     K, phi, stride, stride, off, Lin, 0, in_slope, gate_slope, 0, 1.f, out_scale, accumulate};
    bool h = false;
    int rc = conv1d_mfma_launch(p, stream, &h);
    if (rc) return rc;
    if (!h) return phi == 0 ? TTTS_OK : fail(TTTS_EUNSUPPORTED, "conv1d_dgrad: polyphase tile does not fit LDS");
  }
  *handled = true;
  return TTTS_OK;
}

int conv1d_wgrad_mfma_try(const float* dy, const float* x, float* dw, int B, int Cin, int Lin, int Cout, int Lout, int K,
                          int stride, int pad, int dil, float dy_slope, float x_slope, hipStream_t stream, bool* handled) {
  *handled = false;
  if (Cin * K < 32 || Cout < 16) return TTTS_OK;
  const int lin_t = (WM_L - 1) * stride + (K - 1) * dil + 1;
  const int nch_max = 64 / K + 2;
  const size_t smem = ((size_t)64 * (WM_L + 1) + (size_t)nch_max * (lin_t | 1)) * sizeof(float);
  if (smem > 96 * 1024) return TTTS_OK;
  const int chunks = B * (int)cdiv(Lout, WM_L);
  const int tiles = (int)(cdiv(Cin * K, 64) * cdiv(Cout, 64));
  const int splits = (int)std::max<int64_t>(1, std::min<int64_t>(chunks, cdiv(2048, tiles)));
  const int cpb = (int)cdiv(chunks, splits);
  WgradMfmaParams p{dy, x, dw, B, Cin, Lin, Cout, Lout, K, stride, pad, dil, dy_slope, x_slope, cpb};
  static bool a = false;
  int rc = set_attr_once(reinterpret_cast<const void*>(conv1d_wgrad_mfma_kernel), a);
  if (rc) return rc;
  dim3 grid((unsigned)cdiv(Cin * K, 64), (unsigned)cdiv(Cout, 64), (unsigned)cdiv(chunks, cpb));
  conv1d_wgrad_mfma_kernel<<<grid, 256, smem, stream>>>(p);
  *handled = true;
  return check_launch("conv1d_wgrad_mfma");
}

}  // namespace ttts
