// Thin convolutions: one input channel (the waveform-facing first layers: 1 -> 16 k7 / k15 over 163 840 samples, DiscriminatorP's
// 1 -> 32 k5 stride 3) or one output channel (the discriminators' 1024 -> 1 k3 heads).  There is no GEMM in them -- a tile
// kernel spends its time on padding (measured: 0.3 - 1.2 TF/s, 20-50x their HBM time) -- so these are plain streaming kernels:
// every activation byte is read or written once, coalesced, the few weights sit in LDS / registers.
#include <algorithm>

#include "common.hpp"

namespace ttts {

__device__ __forceinline__ float thin_lrelu(float v, float s) { return v > 0.f ? v : v * s; }

struct ThinFwdParams {
  const float* x; const float* w; const float* bias; float* y;
  int B, Lin, Cout, Lout, K, stride, pad, dil;
  float in_slope; int out_act; float out_slope, out_scale;
};

// ---- forward, Cin = 1:  y[b][co][l] = act(bias[co] + sum_k w[co][k] * lrelu(x[b][l*stride - pad + k*dil])) * out_scale --------
// workgroup = 256 threads x VEC positions (position j*256 + tid: dword loads / stores of a wave are one 256-byte run)
constexpr int T1_VEC = 4, T1_KMAX = 16;
__global__ __launch_bounds__(256) void conv1d_cin1_fwd_kernel(ThinFwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float thin_smem[];
  const int NP = 256 * T1_VEC;
  const int win = (NP - 1) * p.stride + (p.K - 1) * p.dil + 1;
  float* xs = thin_smem;                 // [win]
  float* ws = thin_smem + win;           // [Cout][K]
  const int tid = threadIdx.x, b = blockIdx.y, l0 = blockIdx.x * NP;
  const int in0 = l0 * p.stride - p.pad;
  const float* xr = p.x + (int64_t)b * p.Lin;
  for (int i = tid; i < win; i += 256) {
    const int g = in0 + i;
    xs[i] = (g >= 0 && g < p.Lin) ? thin_lrelu(xr[g], p.in_slope) : 0.f;
  }
  for (int i = tid; i < p.Cout * p.K; i += 256) ws[i] = p.w[i];
  __syncthreads();
  float xv[T1_VEC][T1_KMAX];
#pragma unroll
  for (int j = 0; j < T1_VEC; ++j)
#pragma unroll
    for (int k = 0; k < T1_KMAX; ++k) xv[j][k] = k < p.K ? xs[(j * 256 + tid) * p.stride + k * p.dil] : 0.f;
  for (int co = 0; co < p.Cout; ++co) {
    const float bv = p.bias ? p.bias[co] : 0.f;
    float acc[T1_VEC];
#pragma unroll
    for (int j = 0; j < T1_VEC; ++j) acc[j] = bv;
#pragma unroll
    for (int k = 0; k < T1_KMAX; ++k) {
      if (k < p.K) {
        const float wv = ws[co * p.K + k];             // same address in every lane: an LDS broadcast
#pragma unroll
        for (int j = 0; j < T1_VEC; ++j) acc[j] = fmaf(wv, xv[j][k], acc[j]);
      }
    }
    float* yr = p.y + ((int64_t)b * p.Cout + co) * p.Lout;
#pragma unroll
    for (int j = 0; j < T1_VEC; ++j) {
      const int l = l0 + j * 256 + tid;
      float v = acc[j];
      if (p.out_act == 1) v = tanhf(v);
      else if (p.out_act == 2) v = thin_lrelu(v, p.out_slope);
      if (l < p.Lout) yr[l] = v * p.out_scale;
    }
  }
}

// ---- weight gradient, Cin = 1:  dw[co][k] += sum_{b,l} lrelu(dy[b][co][l]) * lrelu(x[b][l*stride - pad + k*dil]) ---------------
// workgroup = (batch element, chunk of T1_CH positions); wave w owns the rows co = w, w + 4, ...: RW x K accumulators per lane,
// the x window of a 64-position block is read from LDS once and reused by all rows; one atomic per (co, k) and workgroup.
constexpr int T1_CH = 2048;
template <int K, int RW>
__global__ __launch_bounds__(256) void conv1d_cin1_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                float* __restrict__ dw, float* __restrict__ slab, int Lin, int Cout,
                                                                int Lout, int stride, int pad, int dil, float dy_slope, float x_slope) {
  extern __shared__ __attribute__((aligned(16))) float thin_smem[];
  const int win = (T1_CH - 1) * stride + (K - 1) * dil + 1;
  float* xs = thin_smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y, l0 = blockIdx.x * T1_CH;
  const int in0 = l0 * stride - pad;
  const float* xr = x + (int64_t)b * Lin;
  for (int i = tid; i < win; i += 256) {
    const int g = in0 + i;
    xs[i] = (g >= 0 && g < Lin) ? thin_lrelu(xr[min(max(g, 0), Lin - 1)], x_slope) : 0.f;
  }
  __syncthreads();
  float acc[RW][K];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int k = 0; k < K; ++k) acc[r][k] = 0.f;
  const int nblk = min(T1_CH, Lout - l0);                // positions of this chunk
  const float* dyb = dy + (int64_t)b * Cout * Lout + l0;
  for (int pb = 0; pb < nblk; pb += 64) {
    const int li = pb + lane;
    const bool ok = li < nblk;
    float xv[K];
#pragma unroll
    for (int k = 0; k < K; ++k) xv[k] = xs[min(li, T1_CH - 1) * stride + k * dil];
    float dv[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int co = min(wave + 4 * r, Cout - 1);
      dv[r] = dyb[(int64_t)co * Lout + min(li, nblk - 1)];
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const float d = (ok && wave + 4 * r < Cout) ? thin_lrelu(dv[r], dy_slope) : 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) acc[r][k] = fmaf(d, xv[k], acc[r][k]);
    }
  }
  // one partial per (workgroup, co, k): into the caller's slab when there is one (summed in a fixed order by
  // thin_slab_sum_kernel) -- thousands of workgroups adding to the same 112 addresses serialise in L2 (measured: 1.05 ms of a
  // 1.1 ms launch) -- else atomics
  float* part = slab ? slab + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * Cout * K : nullptr;
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int co = wave + 4 * r;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float s = wave_sum(acc[r][k]);
      if (lane == 0 && co < Cout) {
        if (part) part[co * K + k] = s;
        else atomicAdd(dw + co * K + k, s);
      }
    }
  }
}

// dw[i] += sum_p slab[p][i]: one workgroup per output element group, fixed summation order
__global__ __launch_bounds__(256) void thin_slab_sum_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nparts, int per) {
  __shared__ float sh[4];
  const int i = blockIdx.x;
  float s = 0.f;
  for (int p = threadIdx.x; p < nparts; p += 256) s += slab[(int64_t)p * per + i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) dw[i] += (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// ---- forward, Cout = 1, stride 1:  y[b][0][l] = act(bias + sum_{ci,k} w[ci][k] * lrelu(x[b][ci][l - pad + k*dil])) -----------
// positions of all batch elements flattened; workgroup = 16 positions x 16 channel groups, partial sums meet in LDS
constexpr int TO_KMAX = 8;
__global__ __launch_bounds__(256) void conv1d_cout1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y, int B,
                                                               int Cin, int Lin, int Lout, int K, int pad, int dil,
                                                               float in_slope, int out_act, float out_slope, float out_scale) {
  __shared__ float red[16][17];
  const int tid = threadIdx.x, pi = tid & 15, cg = tid >> 4;
  const int64_t P = (int64_t)B * Lout, pos = (int64_t)blockIdx.x * 16 + pi;
  const bool pok = pos < P;
  const int b = (int)(min(pos, P - 1) / Lout), l = (int)(min(pos, P - 1) % Lout);
  int gi[TO_KMAX]; bool gok[TO_KMAX];
#pragma unroll
  for (int k = 0; k < TO_KMAX; ++k) {
    const int g = l - pad + k * dil;
    gok[k] = k < K && g >= 0 && g < Lin;
    gi[k] = min(max(g, 0), Lin - 1);
  }
  float acc = 0.f;
  const float* xb = x + (int64_t)b * Cin * Lin;
  // Requests of FOUR channels (4 K taps + 4 K weights) first, arithmetic after.  Round 6 (tools/isa_scan.py): as `v = xr[gi[k]];
  // acc = fmaf(w, ok ? lrelu(v) : 0, acc)` the compiler waited for every pair of requests before issuing the next -- a thread's 64
  // channels x K taps were 100+ dependent round trips (62 us for a 66-MB read).  Same fmaf chain in the same order: bit-identical.
  constexpr int CU = 4;
  int ci = cg;
  for (; ci + 16 * (CU - 1) < Cin; ci += 16 * CU) {
    float xv[CU][TO_KMAX], wv[CU][TO_KMAX];
#pragma unroll
    for (int u = 0; u < CU; ++u) {
      const float* xr = xb + (int64_t)(ci + 16 * u) * Lin;
      const float* wr = w + (int64_t)(ci + 16 * u) * K;
#pragma unroll
      for (int k = 0; k < TO_KMAX; ++k) {
        if (k < K) { xv[u][k] = xr[gi[k]]; wv[u][k] = wr[k]; }
      }
    }
#pragma unroll
    for (int u = 0; u < CU; ++u) {
#pragma unroll
      for (int k = 0; k < TO_KMAX; ++k) {
        if (k < K) acc = fmaf(wv[u][k], gok[k] ? thin_lrelu(xv[u][k], in_slope) : 0.f, acc);
      }
    }
  }
  for (; ci < Cin; ci += 16) {
    const float* xr = xb + (int64_t)ci * Lin;
    const float* wr = w + (int64_t)ci * K;
#pragma unroll
    for (int k = 0; k < TO_KMAX; ++k) {
      if (k < K) {
        const float v = xr[gi[k]];
        acc = fmaf(wr[k], gok[k] ? thin_lrelu(v, in_slope) : 0.f, acc);
      }
    }
  }
  red[cg][pi] = acc;
  __syncthreads();
  if (tid < 16) {
    float s = bias ? bias[0] : 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) s += red[g][tid];
    if (out_act == 1) s = tanhf(s);
    else if (out_act == 2) s = thin_lrelu(s, out_slope);
    if (pok) y[pos] = s * out_scale;
  }
}

// ---- data gradient, Cout = 1, stride 1:  dx[b][ci][i] = out_scale * sum_k w[ci][k] * dy[b][i + pad - k*dil] -------------------------
// (round 4: DiscriminatorP's 1024 -> 1 heads ran the tile kernels at 0.4 TF/s, 190-250 us for 66 MB of output.)  One output element
// per thread, flat over (b, ci, i): stores are fully coalesced, the 3 weights and the 23..127-sample dy row come from L1.
__global__ __launch_bounds__(256) void conv1d_cout1_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                 float* __restrict__ dx, int64_t total, int Cin, int Lin, int Lout,
                                                                 int K, int pad, int dil, float out_scale) {
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int i = (int)(idx % Lin);
    const int64_t r = idx / Lin;
    const int ci = (int)(r % Cin);
    const int64_t b = r / Cin;
    const float* dyr = dy + b * Lout;
    const float* wr = w + (int64_t)ci * K;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
      const int j = i + pad - k * dil;
      if (j >= 0 && j < Lout) acc = fmaf(wr[k], dyr[j], acc);
    }
    dx[idx] = acc * out_scale;
  }
}

// ---- weight gradient, Cout = 1, stride 1:  dw[ci][k] += sum_{b,l} dy[b][l] * x[b][ci][l - pad + k*dil] ---------------------------
// One wave per (input channel, batch slice): lane = position (rows of 23..127 samples: one or two passes of 64), the row of x is one
// coalesced load, tap k's operand is the same row read at a shifted position (L1).  Eight rows in flight per wave; the per-slice
// partial sums go to a slab [slice][ci][k] and are added in slice order (thin_slab_sum_kernel): deterministic.
constexpr int TO_WG_SLICES = 16;
__global__ __launch_bounds__(256) void conv1d_cout1_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                 float* __restrict__ slab, int B, int Cin, int Lin, int Lout, int K,
                                                                 int pad, int dil) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ci = blockIdx.x * 4 + wave, slice = blockIdx.y;
  if (ci >= Cin) return;
  const int bper = (B + TO_WG_SLICES - 1) / TO_WG_SLICES, b0 = slice * bper, b1 = min(B, b0 + bper);
  float acc[TO_KMAX];
#pragma unroll
  for (int k = 0; k < TO_KMAX; ++k) acc[k] = 0.f;
  for (int l0 = 0; l0 < Lout; l0 += 64) {
    const int l = l0 + lane;
    const bool lok = l < Lout;
#pragma unroll 4
    for (int b = b0; b < b1; ++b) {
      const float d = lok ? dy[(int64_t)b * Lout + l] : 0.f;
      const float* xr = x + ((int64_t)b * Cin + ci) * Lin;
#pragma unroll
      for (int k = 0; k < TO_KMAX; ++k) {
        if (k < K) {
          const int g = l - pad + k * dil;
          const float v = (lok && g >= 0 && g < Lin) ? xr[g] : 0.f;
          acc[k] = fmaf(d, v, acc[k]);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < TO_KMAX; ++k) {
    if (k < K) {
      const float s = wave_sum(acc[k]);
      if (lane == 0) slab[((int64_t)slice * Cin + ci) * K + k] = s;
    }
  }
}

// ---- dispatch (conv.hip) ------------------------------------------------------------------------------------------------------
int conv1d_thin_fwd_try(const float* x, const float* w, const float* bias, const float* bbias, const float* resid,
                        const float* gate, const float* omask, float* y, int B, int Cin, int Lin, int Cout, int Lout, int K,
                        int stride, int pad, int dil, float in_slope, int out_act, float out_slope, float out_scale,
                        int accumulate, hipStream_t stream, bool* handled) {
  *handled = false;
  if (bbias || resid || gate || omask || accumulate) return TTTS_OK;
  if (Cin == 1 && K <= T1_KMAX && Cout <= 64) {
    const int NP = 256 * T1_VEC;
    const size_t smem = ((size_t)(NP - 1) * stride + (size_t)(K - 1) * dil + 1 + (size_t)Cout * K) * sizeof(float);
    if (smem > 64 * 1024) return TTTS_OK;
    ThinFwdParams p{x, w, bias, y, B, Lin, Cout, Lout, K, stride, pad, dil, in_slope, out_act, out_slope, out_scale};
    conv1d_cin1_fwd_kernel<<<dim3((unsigned)cdiv(Lout, NP), (unsigned)B), 256, smem, stream>>>(p);
    *handled = true;
    return check_launch("conv1d_cin1_fwd");
  }
  if (Cout == 1 && stride == 1 && K <= TO_KMAX && Cin >= 64) {
    conv1d_cout1_fwd_kernel<<<(unsigned)cdiv((int64_t)B * Lout, 16), 256, 0, stream>>>(x, w, bias, y, B, Cin, Lin, Lout, K, pad, dil, in_slope,
                                                                                       out_act, out_slope, out_scale);
    *handled = true;
    return check_launch("conv1d_cout1_fwd");
  }
  return TTTS_OK;
}

int conv1d_thin_dgrad_try(const float* dy, const float* w, const float* bias, const float* resid, const float* gate,
                          const float* omask, float* dx, int B, int Cin, int Lin, int Cout, int Lout, int K, int stride, int pad,
                          int dil, float in_slope, float out_scale, int accumulate, hipStream_t stream, bool* handled) {
  *handled = false;
  if (bias || resid || gate || omask || accumulate || in_slope != 1.f) return TTTS_OK;
  if (Cout == 1 && stride == 1 && K <= TO_KMAX && Cin >= 64) {
    const int64_t total = (int64_t)B * Cin * Lin;
    conv1d_cout1_dgrad_kernel<<<(unsigned)std::min<int64_t>(cdiv(total, 256), 65536), 256, 0, stream>>>(dy, w, dx, total, Cin, Lin, Lout, K, pad,
                                                                                                     dil, out_scale);
    *handled = true;
    return check_launch("conv1d_cout1_dgrad");
  }
  return TTTS_OK;
}

template <int K, int RW>
static int cin1_wgrad_launch(const float* dy, const float* x, float* dw, int B, int Lin, int Cout, int Lout, int stride, int pad,
                             int dil, float dy_slope, float x_slope, const ConvCtx& cx, hipStream_t stream) {
  const size_t smem = ((size_t)(T1_CH - 1) * stride + (size_t)(K - 1) * dil + 1) * sizeof(float);
  dim3 grid((unsigned)cdiv(Lout, T1_CH), (unsigned)B);
  const int nparts = (int)(grid.x * grid.y), per = Cout * K;
  float* slab = (cx.ws && nparts > 8 && (int64_t)nparts * per * (int64_t)sizeof(float) <= cx.ws_bytes) ? static_cast<float*>(cx.ws) : nullptr;
  conv1d_cin1_wgrad_kernel<K, RW><<<grid, 256, smem, stream>>>(dy, x, dw, slab, Lin, Cout, Lout, stride, pad, dil, dy_slope, x_slope);
  if (slab) thin_slab_sum_kernel<<<per, 256, 0, stream>>>(slab, dw, nparts, per);
  return check_launch("conv1d_cin1_wgrad");
}

int conv1d_thin_wgrad_try(const float* dy, const float* x, float* dw, int B, int Cin, int Lin, int Cout, int Lout, int K,
                          int stride, int pad, int dil, float dy_slope, float x_slope, const ConvCtx& cx, hipStream_t stream,
                          bool* handled) {
  *handled = false;
  if (Cout == 1 && stride == 1 && K <= TO_KMAX && Cin >= 64 && dy_slope == 1.f && x_slope == 1.f && cx.ws &&
      (int64_t)TO_WG_SLICES * Cin * K * (int64_t)sizeof(float) <= cx.ws_bytes) {
    float* slab = static_cast<float*>(cx.ws);
    conv1d_cout1_wgrad_kernel<<<dim3((unsigned)cdiv(Cin, 4), TO_WG_SLICES), 256, 0, stream>>>(dy, x, slab, B, Cin, Lin, Lout, K, pad, dil);
    thin_slab_sum_kernel<<<Cin * K, 256, 0, stream>>>(slab, dw, TO_WG_SLICES, Cin * K);
    *handled = true;
    return check_launch("conv1d_cout1_wgrad");
  }
  if (Cin != 1 || (size_t)((T1_CH - 1) * stride + (K - 1) * dil + 1) * sizeof(float) > 60 * 1024) return TTTS_OK;
  int rc = TTTS_OK;
  if (K == 7 && Cout <= 16) rc = cin1_wgrad_launch<7, 4>(dy, x, dw, B, Lin, Cout, Lout, stride, pad, dil, dy_slope, x_slope, cx, stream);
  else if (K == 15 && Cout <= 16) rc = cin1_wgrad_launch<15, 4>(dy, x, dw, B, Lin, Cout, Lout, stride, pad, dil, dy_slope, x_slope, cx, stream);
  else if (K == 5 && Cout <= 32) rc = cin1_wgrad_launch<5, 8>(dy, x, dw, B, Lin, Cout, Lout, stride, pad, dil, dy_slope, x_slope, cx, stream);
  else if (K == 3 && Cout <= 32) rc = cin1_wgrad_launch<3, 8>(dy, x, dw, B, Lin, Cout, Lout, stride, pad, dil, dy_slope, x_slope, cx, stream);
  else return TTTS_OK;
  *handled = rc == TTTS_OK;
  return rc;
}

}  // namespace ttts
