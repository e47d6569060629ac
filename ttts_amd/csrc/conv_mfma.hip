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
#include <mutex>
#include <unordered_map>
#include <vector>

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
  // split-bf16 path: weights pre-split into hi / lo bf16 in [N/16 blocks][Mpad][K][16] order (conv_weight_split_kernel)
  const bf16* a_hi; const bf16* a_lo; int Mpad;
  const bf16* x_hi; const bf16* x_lo;   // pre-split, zero-padded input [B][N/16 blocks][2 channel halves][Lp][8] (conv_input_split_kernel)
  int Lp, PADL;        // padded row length of the pre-split input; element i of a row holds position i - PADL
  int NBS;             // DMA kernel: 16-channel blocks per pipeline stage (1, 2 or 4: 1 x 1 convolutions have 6 MFMAs per wave and
                       // block -- a stage that short is all DMA latency)
  int catLg, catLout;  // > 0: the batch is laid end to end as ONE virtual row (B == 1 here), output position v = b * catLg + j,
                       // j < catLout real positions per batch element (short rows: DiscriminatorP's 23..127-position layers)
  int rowS, rpad;      // rowS > 0: PHASE-MERGED strided data gradient (round 4).  Row m of the GEMM is (channel m / rowS, phase
                       // m % rowS) of dx: dx[ci][rowS j + r] = sum_{n, e} w[n][ci][k(r, e)] dy[n][j + e], one stride-1 convolution
                       // with rowS x the rows and ceil(K / rowS) + 1 taps instead of rowS launches with M = Cin rows each
                       // (conv1d_dgrad_strided_mfma_try); rpad = the strided convolution's own padding, p.pad = -min e
                       // rowS < 0: PHASE-MERGED strided FORWARD with S = -rowS: input channel n of the GEMM is (channel n / S, phase
                       // n % S) of x, x'[(ci, r)][j] = x[ci][S j + r] (the pre-split pass de-interleaves), y[co][l] = sum w[co][ci][S t
                       // + r + rpad] x'[(ci, r)][l + t]: a stride-1 convolution over Cin x S channels with ~K / S + 1 taps (DMA kernel only)
  int Lreal;           // rowS < 0: row length of the real input x;  rowS > 0: bytes of dynamic LDS of the launch (DMA kernel epilogue)
  float* y2; int M1, acc2;   // y2 != NULL: DUAL-destination forward (round 4).  Output rows m < M1 go to y as usual (bias, resid, mask);
                       // rows m >= M1 go to y2[b][m - M1][j] = [y2 +] (acc + bias[m]) * mask * out_scale (acc2: accumulate).  The WN
                       // res/skip 1 x 1 convolution is ONE launch that way instead of two over the same input
};


__device__ __forceinline__ float lrelu_f(float v, float s) { return v > 0.f ? v : v * s; }

// ---- range bookkeeping of the single-pass fp16 ("TF32-class") mode (ABI v11) --------------------------------------------------
// fp16 has TF32's 11 significant bits but 5 exponent bits instead of 8: an operand above 65504 saturates, one at or below 2^-25
// becomes zero (one below 2^-14 keeps fewer bits: not counted, see common.hpp).  Every fp32 -> fp16 conversion of the mode (operand pre-passes, on-the-fly
// staging, weight splits) ORs what it saw into a per-thread word and adds it here once -- a module global, i.e. one copy per
// device -- so that a trainer can see, without a host sync inside the step, that its loss scale left the range
// (ttts_conv_f16_events; ttts_loss_scale_update turns it into GradScaler's halve / skip / grow rule).
// Counts are THREADS that saw the event (a thread converts 8..16 neighbouring elements), not elements.
__device__ unsigned int g_f16_events[4];      // {saturated, flushed to zero, reserved (0), -}
__device__ __forceinline__ void f16_events_commit(unsigned ev) {
  // one atomic per WAVE and event kind (flushes are not rare: ~3 M threads per VQ-VAE-GAN step see one), counts stay per thread
  const uint64_t m1 = __builtin_amdgcn_ballot_w64((ev & 1u) != 0), m2 = __builtin_amdgcn_ballot_w64((ev & 2u) != 0);
  if ((m1 | m2) != 0) {
    const uint64_t act = __builtin_amdgcn_ballot_w64(true);
    if ((int)(threadIdx.x & 63) == __builtin_ctzll(act)) {      // first active lane of the wave
      if (m1) atomicAdd(&g_f16_events[0], (unsigned)__builtin_popcountll(m1));
      if (m2) atomicAdd(&g_f16_events[1], (unsigned)__builtin_popcountll(m2));
    }
  }
}
__global__ void f16_events_fetch_kernel(int32_t* __restrict__ out, int reset) {
  const int i = threadIdx.x;
  if (i < 3) {
    out[i] += (int32_t)min(g_f16_events[i], 0x3fffffffu);
    if (reset) g_f16_events[i] = 0u;
  }
}
// GradScaler's rule on device words.  ls = {scale, 1 / scale, clean steps in a row, overflow seen since the last update,
// total saturation events, total flush events, skipped steps, reserved}.
//   check  (one per backward): events[0] != 0 -> *skip = 1 (the optimizer step that follows leaves parameters and moments alone) and
//          ls[3] = 1; totals accumulate; events are cleared for the next backward.
//   update (one per step): overflow -> scale *= backoff (not below 1), streak = 0; else streak += 1 and after `interval` clean steps
//          scale *= growth (not above 2^24).  1 / scale follows (both exact: powers of two for the default factors).
__global__ void loss_scale_check_kernel(float* __restrict__ ls, int32_t* __restrict__ events, float* __restrict__ skip) {
  if (threadIdx.x != 0) return;
  const bool over = events[0] != 0;
  if (skip) *skip = over ? 1.f : 0.f;
  if (over) { ls[3] = 1.f; ls[6] += 1.f; }
  ls[4] += (float)events[0]; ls[5] += (float)events[1]; ls[7] += (float)events[2];
  events[0] = 0; events[1] = 0; events[2] = 0;
}
__global__ void loss_scale_update_kernel(float* __restrict__ ls, int interval, float backoff, float growth) {
  if (threadIdx.x != 0) return;
  float scale = ls[0], streak = ls[2];
  if (ls[3] != 0.f) { scale = fmaxf(scale * backoff, 1.f); streak = 0.f; }
  else {
    streak += 1.f;
    if (interval > 0 && streak >= (float)interval) { scale = fminf(scale * growth, 16777216.f); streak = 0.f; }
  }
  ls[0] = scale; ls[1] = 1.f / scale; ls[2] = streak; ls[3] = 0.f;
}

// WCO = waves along the output-channel axis (1 or 2), NW = waves per workgroup (1, 2 or 4); the NW / WCO remaining waves tile
// positions.  Tile shapes (channels x positions): <2,4> 64x128, <1,4> 32x256, <2,2> 64x64, <1,2> 32x128, <1,1> 32x64 -- the
// small ones exist so that short / narrow layers (192 channels x 256 frames x batch 32 = 192 big tiles on 256 CUs) still put
// several workgroups on every CU, which is what hides the global -> LDS staging latency of a stage behind another
// workgroup's MFMAs.
template <int WCO, int NW>
__global__ __launch_bounds__(64 * NW) void conv1d_mfma_kernel(ConvMfmaParams p) {
  constexpr int NTHR = 64 * NW;
  constexpr int MT = 32 * WCO;                 // output channels per workgroup
  constexpr int WL = NW / WCO;                 // waves along positions
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
  for (int f = tid; f < KK; f += NTHR) foff[f] = (f / p.K) * lin_t + (f % p.K) * p.dil;
  if (p.transposed)
    for (int i = tid; i < MT * p.Kmem; i += NTHR) {
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
  const int lrow = tid / (NTHR / MT), lq = tid % (NTHR / MT);   // weight loader: NTHR / MT threads per output-channel row
  for (int n0 = 0; n0 < p.N; n0 += NT) {
    __syncthreads();
    // input strip: one wave per channel row at a time, lanes along positions (coalesced, no index arithmetic)
    for (int n = wave; n < NT; n += NW) {
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
        for (int f = lq; f < KK; f += NTHR / MT) wd[f] = f < flim ? wr[f] : 0.f;
      } else {
        // A[m][n][k] = w[n][m][K-1-k]  (w is [N][M][K] here): for one n the MT*K values w[n][m0 .. m0+MT)[:] are contiguous
        const int cnt = min(MT, p.M - m0) * p.Kmem;
        for (int n = wave; n < NT; n += NW) {
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
    int b = b0 + ct / SEG, jt = j0 + ct % SEG;
    if (jt >= p.Lout || b >= p.B) continue;
    if (p.catLg) {                                         // virtual row -> (batch element, position)
      b = jt / p.catLg; jt -= b * p.catLg;
      if (jt >= p.catLout) continue;
    }
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

// ---- split-bf16 ("bf16 x 3") implicit GEMM on the bf16 matrix cores ----------------------------------------------------------
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi); the product is accumulated in fp32 as hi*hi + hi*lo + lo*hi (the dropped
// lo*lo term and the representation residue are ~2^-16 relative: ~100x tighter than the TF32 the reference's cuDNN
// convolutions run with).  v_mfma_f32_32x32x16_bf16 is 16x the rate of the f32-input MFMA, so three of them per product still
// leave a 5x head-room.  Reduction order: (tap, channel) with the 16 channels of a stage minor, so that a lane's 8 k-values are
// one 16-byte LDS read: the input strip is staged position-major [pos][16 ch] (hi and lo), the weights arrive pre-split and
// pre-ordered [n/16][m][tap][16] from conv_weight_split_kernel -- which also bakes in the transposition / tap flip / polyphase
// tap selection of the data-gradient forms, so this kernel only ever sees a plain forward convolution.
// input pre-pass: x fp32 [B][N][L] -> leaky-relu -> hi / lo bf16 in [B][N/16][L][16] order, so that the strip a workgroup
// stages (16 channels x a window of positions) is ONE contiguous run of 32-byte records: the main kernel's staging becomes
// plain 16-byte copies (PMC before this pass: 12 VALU instructions per MFMA, mostly fp32 -> bf16 splitting, and VALU time
// above MFMA time)
__global__ __launch_bounds__(256) void conv_input_split_kernel(const float* __restrict__ x, bf16* __restrict__ hi,
                                                               bf16* __restrict__ lo, int B, int N, int L, int nblk, float slope,
                                                               int Lp, int PADL, int catW, int catB, int catL, int inS = 0, int Lreal = 0,
                                                               int f16 = 0) {
  // one thread per (batch element, 16-channel block, padded position): 16 coalesced row reads, four 16-byte stores.
  // catW > 0: ONE destination row (B == 1) holding the catB source rows of length catL end to end, catW positions apart
  const int64_t total = (int64_t)B * nblk * Lp;
  unsigned ev = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int pp = (int)(i % Lp), nb = (int)((i / Lp) % nblk);
    int pos = pp - PADL;
    int64_t b = i / Lp / nblk;
    int Lr = L;
    bool inside = pos >= 0 && pos < L;
    if (catW) {
      const int bb = pos >= 0 ? pos / catW : 0;
      pos -= bb * catW;
      inside = pos >= 0 && pos < catL && bb < catB;
      b = min(bb, catB - 1); Lr = catL;
    }
    float raw[16];
    bool okc[16];
    if (inS > 0) {       // phase-de-interleaved view: channel n = (ci, r), position pos  <-  x[b][ci][inS pos + r]  (catW == 0 here)
      const int Cin = N / inS;
      const float* xb = x + (b * Cin) * Lreal;
      const int64_t base = (int64_t)inS * max(pos, 0);
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int n = min(nb * 16 + c, N - 1), ci = n / inS, r = n - ci * inS;
        const int64_t q = base + r;
        okc[c] = q < Lreal;
        raw[c] = xb[(int64_t)ci * Lreal + min(q, (int64_t)Lreal - 1)];
      }
    } else {
      const float* xr = x + (b * N) * Lr + min(max(pos, 0), Lr - 1);
#pragma unroll
      for (int c = 0; c < 16; ++c) { raw[c] = xr[(int64_t)min(nb * 16 + c, N - 1) * Lr]; okc[c] = true; }   // unconditional (clamped) loads, then select
    }
    bf16x8 h0, h1, l0, l1;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      float v = (inside && okc[c] && nb * 16 + c < N) ? raw[c] : 0.f;
      v = lrelu_f(v, slope);
      const bf16 hv = f16 ? f16_slot_ev(v, ev) : (bf16)v;   // (f16: the single-pass mode's fp16 copy; the lo array is not written)
      const bf16 lv = (bf16)(v - (float)hv);
      if (c < 8) { h0[c] = hv; l0[c] = lv; } else { h1[c - 8] = hv; l1[c - 8] = lv; }
    }
    const int64_t o = ((((i / Lp / nblk) * nblk + nb) * 2) * Lp + pp) * 8;        // half 0; half 1 is Lp * 8 elements further
    *reinterpret_cast<bf16x8*>(hi + o) = h0;
    *reinterpret_cast<bf16x8*>(hi + o + (int64_t)Lp * 8) = h1;
    if (!f16) {
      *reinterpret_cast<bf16x8*>(lo + o) = l0;
      *reinterpret_cast<bf16x8*>(lo + o + (int64_t)Lp * 8) = l1;
    }
  }
  f16_events_commit(ev);
}

// the merged-phase weight of row m, virtual tap k: w[n][ci][kk] with ci = m / S, r = m % S, t = (r + rpad) / S - (k - vpad),
// kk = (r + rpad) % S + S t -- or zero when that tap does not exist
__device__ __forceinline__ float merged_phase_weight(const float* __restrict__ w, int m, int n, int k, int Cin, int Kmem, int S,
                                                     int rpad, int vpad) {
  const int ci = m / S, r = m - ci * S;
  const int t = (r + rpad) / S - (k - vpad), kk = (r + rpad) % S + S * t;
  return (t >= 0 && kk < Kmem) ? w[((int64_t)n * Cin + ci) * Kmem + kk] : 0.f;
}

// the merged-phase FORWARD weight of output row m, virtual input channel n = (ci, r), virtual tap k: w[m][ci][S (k - vpad) + r + rpad]
__device__ __forceinline__ float merged_fwd_weight(const float* __restrict__ w, int m, int n, int k, int Cin, int Kmem, int S,
                                                   int rpad, int vpad) {
  const int ci = n / S, r = n - ci * S;
  const int kk = S * (k - vpad) + r + rpad;
  return (kk >= 0 && kk < Kmem) ? w[((int64_t)m * Cin + ci) * Kmem + kk] : 0.f;
}

__global__ __launch_bounds__(256) void conv_weight_split_kernel(const float* __restrict__ w, bf16* __restrict__ a_hi,
                                                                bf16* __restrict__ a_lo, int M, int N, int Mpad, int nblk,
                                                                int K, int Kmem, int transposed, int tap_off, int tap_stride, int AP,
                                                                int rowS = 0, int rpad = 0, int vpad = 0, int f16 = 0) {
  // rows of AP >= K*16 elements ([tap][16 channels], zero tail): AP = K*16 + 8 is the LDS row pitch of the DMA-fed kernel,
  // whose stages are verbatim copies of [MT rows][AP] runs of these arrays
  const int64_t total = (int64_t)nblk * Mpad * AP;
  unsigned ev = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int e = (int)(i % AP), m = (int)(i / AP % Mpad), nb = (int)(i / AP / Mpad);
    const int c = e & 15, k = e >> 4;
    const int n = nb * 16 + c;
    float v = 0.f;
    if (k < K && m < M && n < N) {
      // forward: A[m][k][n] = w[m][n][k];  data gradient: A[m][k][n] = w[n][m][tap_off + tap_stride * (K - 1 - k)]
      if (rowS > 0) v = merged_phase_weight(w, m, n, k, M / rowS, Kmem, rowS, rpad, vpad);
      else if (rowS < 0) v = merged_fwd_weight(w, m, n, k, N / -rowS, Kmem, -rowS, rpad, vpad);
      else v = transposed ? w[((int64_t)n * M + m) * Kmem + tap_off + tap_stride * (K - 1 - k)] : w[((int64_t)m * N + n) * Kmem + k];
    }
    const bf16 h = f16 ? f16_slot_ev(v, ev) : (bf16)v;
    a_hi[i] = h;
    if (!f16) a_lo[i] = (bf16)(v - (float)h);
  }
  f16_events_commit(ev);
}

// ---- weight-split cache (ABI v8) -----------------------------------------------------------------------------------------
// A training step calls every convolution's forward once and its data gradient once, each preceded by a ~5 us
// conv_weight_split_kernel launch (1300 launches, 6.6 ms of a 160 ms VQ-VAE-GAN step).  The weights only change at known
// points (the optimizer step, WeightNormBank.refresh), so the host registers the arrays they live in; the first call with a
// given (weight pointer, layout) records a descriptor and a persistent destination, and from then on ONE launch at the phase
// start (ttts_conv_wsplit_cache_refresh) rewrites every recorded split.  A disarmed cache (outside a step: the arrays may have
// changed without a refresh) and any miss during stream capture fall back to the per-call split in scratch.
struct WsplitDesc {
  const float* w; bf16* hi; bf16* lo;
  int M, N, Mpad, nblk, K, Kmem, transposed, tap_off, tap_stride, AP;
  int block_begin, rowS, rpad, vpad;
  int f16, pad2_;                     // f16: single-pass mode (fp16 copy in `hi`, `lo` unused)
};
constexpr int WSPLIT_EPB = 2048;      // elements per workgroup of the batched launch (8 per thread)

__global__ __launch_bounds__(256) void conv_weight_split_batched_kernel(const WsplitDesc* __restrict__ table, int n_desc) {
  int lo = 0, hi = n_desc - 1;        // last descriptor with block_begin <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].block_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const WsplitDesc d = table[lo];
  const int64_t total = (int64_t)d.nblk * d.Mpad * d.AP;
  const int64_t i = ((int64_t)(blockIdx.x - d.block_begin) * 256 + threadIdx.x) * 8;     // AP % 8 == 0: one (row, tap, half) run
  if (i >= total) return;
  const int e = (int)(i % d.AP), m = (int)(i / d.AP % d.Mpad), nb = (int)(i / d.AP / d.Mpad);
  const int c0 = e & 15, k = e >> 4;
  bf16x8 h, l;
  unsigned ev = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int n = nb * 16 + c0 + c;
    float v = 0.f;
    if (k < d.K && m < d.M && n < d.N) {
      if (d.rowS > 0) v = merged_phase_weight(d.w, m, n, k, d.M / d.rowS, d.Kmem, d.rowS, d.rpad, d.vpad);
      else if (d.rowS < 0) v = merged_fwd_weight(d.w, m, n, k, d.N / -d.rowS, d.Kmem, -d.rowS, d.rpad, d.vpad);
      else v = d.transposed ? d.w[((int64_t)n * d.M + m) * d.Kmem + d.tap_off + d.tap_stride * (d.K - 1 - k)]
                            : d.w[((int64_t)m * d.N + n) * d.Kmem + k];
    }
    const bf16 hv = d.f16 ? f16_slot_ev(v, ev) : (bf16)v;
    h[c] = hv;
    l[c] = (bf16)(v - (float)hv);
  }
  *reinterpret_cast<bf16x8*>(d.hi + i) = h;
  if (!d.f16) *reinterpret_cast<bf16x8*>(d.lo + i) = l;
  f16_events_commit(ev);
}

struct WsplitKey {
  const float* w; int M, N, Mpad, K, Kmem, transposed, tap_off, tap_stride, AP, rowS, rpad, vpad, f16;
  bool operator==(const WsplitKey& o) const {
    return w == o.w && M == o.M && N == o.N && Mpad == o.Mpad && K == o.K && Kmem == o.Kmem && transposed == o.transposed &&
           tap_off == o.tap_off && tap_stride == o.tap_stride && AP == o.AP && rowS == o.rowS && rpad == o.rpad && vpad == o.vpad &&
           f16 == o.f16;
  }
};
struct WsplitKeyHash {
  size_t operator()(const WsplitKey& k) const {
    uint64_t h = reinterpret_cast<uint64_t>(k.w) * 0x9E3779B97F4A7C15ull;
    for (int v : {k.M, k.N, k.Mpad, k.K, k.Kmem, k.transposed, k.tap_off, k.tap_stride, k.AP, k.rowS, k.rpad, k.vpad, k.f16})
      h = (h ^ (uint64_t)(uint32_t)v) * 0x100000001B3ull;
    return (size_t)h;
  }
};
struct WsplitCache {
  uint32_t magic = TTTS_HANDLE_WSPLIT;      // (first member: how a ttts_conv_ctx::handles entry is told apart)
  std::mutex mu;
  const char* w_lo; const char* w_hi;
  char* storage; int64_t bytes, used;       // [descriptor table: max_entries][256-byte aligned split arrays ...]
  int max_entries;
  bool armed = false;
  std::vector<WsplitDesc> host;
  // the stream that wrote an entry's FIRST split (by its own launch, since the last refresh), else null: until the next refresh
  // rewrites the entry, only that stream is ordered behind the split -- a lookup from any other stream is answered "not cached"
  std::vector<hipStream_t> first_writer;
  std::unordered_map<WsplitKey, int, WsplitKeyHash> index;
  int64_t blocks = 0, hits = 0, misses = 0, launches_saved = 0;
};

// A descriptor goes to its slot of the device-side table as a KERNEL ARGUMENT (by value): no host memory has to outlive the call.
template <typename Desc>
__global__ void desc_store_kernel(Desc* dst, Desc d) { *dst = d; }

// (hi, lo) of the persistent split of this call's weights, or false: split into scratch as before.  *need_split: the caller
// launches the split itself (a descriptor recorded by this very call; the next refresh covers it).
static bool wsplit_lookup(const ConvMfmaParams& p, const ConvCtx& cx, int nblk, int AP, int64_t elems_alloc, hipStream_t stream,
                          bf16** hi, bf16** lo, bool* need_split) {
  const char* wp = reinterpret_cast<const char*>(p.w);
  for (int hidx = 0; hidx < cx.n_handles; ++hidx) {
    WsplitCache* c = static_cast<WsplitCache*>(cx.handles[hidx]);
    if (!c || c->magic != TTTS_HANDLE_WSPLIT || wp < c->w_lo || wp >= c->w_hi) continue;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->armed) continue;            // (a later handle over the same range may be the armed one)
    const int f16 = (cx.flags & TTTS_CONV_F16X1) ? 1 : 0;
    const WsplitKey key{p.w, p.M, p.N, p.Mpad, p.K, p.Kmem, p.transposed, p.tap_off, p.tap_stride, AP, p.rowS, p.rpad, p.pad, f16};
    auto it = c->index.find(key);
    if (it != c->index.end()) {
      if (c->first_writer[it->second] != nullptr && c->first_writer[it->second] != stream) { ++c->misses; return false; }
      const WsplitDesc& d = c->host[it->second];
      *hi = d.hi; *lo = d.lo; *need_split = false;
      ++c->hits;
      return true;
    }
    ++c->misses;
    const int64_t need = (((f16 ? 1 : 2) * elems_alloc * (int64_t)sizeof(bf16) + 255) / 256) * 256;    // (single-pass mode: no lo array)
    if ((int)c->host.size() >= c->max_entries || c->used + need > c->bytes) return false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return false;
    WsplitDesc d;
    d.w = p.w; d.hi = reinterpret_cast<bf16*>(c->storage + c->used); d.lo = f16 ? d.hi : d.hi + elems_alloc;
    d.M = p.M; d.N = p.N; d.Mpad = p.Mpad; d.nblk = nblk; d.K = p.K; d.Kmem = p.Kmem; d.transposed = p.transposed;
    d.tap_off = p.tap_off; d.tap_stride = p.tap_stride; d.AP = AP; d.rowS = p.rowS; d.rpad = p.rpad; d.vpad = p.pad;
    d.f16 = f16; d.pad2_ = 0;
    d.block_begin = (int)c->blocks;
    const int idx = (int)c->host.size();
    desc_store_kernel<WsplitDesc><<<1, 1, 0, stream>>>(reinterpret_cast<WsplitDesc*>(c->storage) + idx, d);
    if (hipGetLastError() != hipSuccess) return false;
    c->host.push_back(d);
    c->first_writer.push_back(stream);
    c->blocks += cdiv((int64_t)nblk * p.Mpad * AP, WSPLIT_EPB);
    c->used += need;
    c->index.emplace(key, idx);
    *hi = d.hi; *lo = d.lo; *need_split = true;
    return true;
  }
  return false;
}

// fused epilogue of the split-bf16 kernels: two 32 x 32 accumulators of a wave (positions wl*64 + {0, 32} + col)
// One lane's 16 accumulator rows (row of register r: mb + (r & 3) + 8 (r >> 2)) of ONE output column (b, j).
// Round 6: as a per-element loop `v += resid[o]; ...; y[o] = v` each element's operand read sat behind the previous element's
// store (y may alias resid / gate for all the compiler knows), i.e. 16-32 DEPENDENT memory round trips per lane: a workgroup of the
// ResBlock convolutions lived ~25 us around ~2 us of MFMAs (PMC: 21 VALU instructions per MFMA, 45 % of the wave cycles in
// s_waitcnt; profiles/r06_pmc_conv_onthefly.txt).  The launches of the training steps read at most ONE such operand (the lrelu gate
// of a data gradient, the residual of a ResBlock's second convolution, the output itself when accumulating, a per-sample bias):
// that case requests it for all 16 rows BEFORE the first store.  Same operations on the same values in the same order per element:
// bit-identical.  Everything else (dual destination, several operands) keeps the per-element loop.
__device__ __forceinline__ void conv_store_col16(const ConvMfmaParams& p, const f32x16& acc, int mb, int b, int j, float om) {
  // (row / column indices made opaque here: everything below depends only on kernel arguments and the thread index, and the compiler
  // computed all of it -- 16 rows x two code paths of addresses and selects -- in the kernel's PROLOGUE, live across the main loop:
  // 55 spilled VGPRs in the DMA kernel)
  asm volatile("" : "+v"(mb), "+v"(j));
  const int nops = (p.gate ? 1 : 0) + (p.resid ? 1 : 0) + (p.accumulate ? 1 : 0) + (p.bbias ? 1 : 0);
  if (p.y2 == nullptr && nops == 2 && p.gate && p.resid) {
    // gate AND residual: the first convolution's data gradient of a ResBlock pair that runs as one autograd node (TTTS_RESPAIR)
    const int64_t rs = p.LoutTotal, o0 = (int64_t)b * p.M * rs + j;
    const float sc = om * p.out_scale;
    float bia[16], gv[16], rv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = min(mb + (r & 3) + 8 * (r >> 2), p.M - 1);
      bia[r] = p.bias ? p.bias[m] : 0.f;
      gv[r] = p.gate[o0 + m * rs];
      rv[r] = p.resid[o0 + m * rs];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mb + (r & 3) + 8 * (r >> 2);
      if (m >= p.M) continue;
      float v = acc[r] + bia[r];
      v *= (gv[r] > 0.f ? 1.f : p.gate_slope);
      v += rv[r];
      if (p.out_act == 1) v = tanhf(v);
      else if (p.out_act == 2) v = lrelu_f(v, p.out_slope);
      v *= sc;
      p.y[o0 + m * rs] = v;
    }
    return;
  }
  if (p.y2 == nullptr && nops <= 1) {
    const int mode = p.gate ? 1 : p.resid ? 2 : p.accumulate ? 3 : p.bbias ? 4 : 0;      // (uniform)
    const float* ap = p.gate ? p.gate : p.resid ? p.resid : p.y;
    const int64_t rs = p.LoutTotal, o0 = (int64_t)b * p.M * rs + j;
    const float sc = om * p.out_scale;
    float bia[16], aux[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = min(mb + (r & 3) + 8 * (r >> 2), p.M - 1);          // (clamped: reads of rows past M are in range and unused)
      bia[r] = p.bias ? p.bias[m] : 0.f;
      aux[r] = mode == 0 ? 0.f : mode == 4 ? p.bbias[(int64_t)b * p.M + m] : ap[o0 + m * rs];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mb + (r & 3) + 8 * (r >> 2);
      if (m >= p.M) continue;
      float v = acc[r] + bia[r];
      if (mode == 4) v += aux[r];
      if (mode == 1) v *= (aux[r] > 0.f ? 1.f : p.gate_slope);
      if (mode == 2) v += aux[r];
      if (p.out_act == 1) v = tanhf(v);
      else if (p.out_act == 2) v = lrelu_f(v, p.out_slope);
      v *= sc;
      p.y[o0 + m * rs] = mode == 3 ? aux[r] + v : v;
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = mb + (r & 3) + 8 * (r >> 2);
    if (m >= p.M) continue;
    float v = acc[r] + (p.bias ? p.bias[m] : 0.f);
    // (dual destination, p.y2: rows >= M1 go to y2 with their own accumulate flag and without the residual -- selected, not
    // branched: a second store path here sent the 64 x 64 wave tile's accumulators through scratch)
    const bool second = p.y2 != nullptr && m >= p.M1;
    const int Mo = p.y2 ? (second ? p.M - p.M1 : p.M1) : p.M, mm = second ? m - p.M1 : m;
    float* dst = second ? p.y2 : p.y;
    const int accf = second ? p.acc2 : p.accumulate;
    const int64_t o = ((int64_t)b * Mo + mm) * p.LoutTotal + j;
    if (p.bbias) v += p.bbias[(int64_t)b * p.M + m];
    if (p.gate) v *= (p.gate[o] > 0.f ? 1.f : p.gate_slope);
    if (p.resid && !second) v += p.resid[o];
    if (p.out_act == 1) v = tanhf(v);
    else if (p.out_act == 2) v = lrelu_f(v, p.out_slope);
    v *= om * p.out_scale;
    dst[o] = accf ? dst[o] + v : v;
  }
}

// one 32-position block (t = 0 / 1) of a wave's accumulator pair.  (`acc` is the block's own accumulator, never a reference picked by
// `t == 0 ? acc0 : acc1`: that select became a pointer select in the DMA kernel and sent its accumulators to scratch -- 320 bytes
// per lane, DiscriminatorP's 1024-channel layers 307 -> 1665 us)
template <int WCO>
__device__ __forceinline__ void conv_tile_col(const ConvMfmaParams& p, const f32x16& acc, int t, int wl, int wco, int col, int hh,
                                              int j0, int m0, int b0, int SEG) {
  const int ct = wl * 64 + t * 32 + col;
  int b = b0 + ct / SEG, jt = j0 + ct % SEG;
  if (jt >= p.Lout || b >= p.B) return;
  if (p.catLg) {                                         // virtual row -> (batch element, position)
    b = jt / p.catLg; jt -= b * p.catLg;
    if (jt >= p.catLout) return;
  }
  const int j = p.out_off + jt * p.out_stride;
  const float om = p.omask ? p.omask[(int64_t)b * p.LoutTotal + j] : 1.f;
  if (p.rowS <= 0) {
    conv_store_col16(p, acc, m0 + wco * 32 + 4 * hh, b, j, om);
    return;
  }
  // phase-merged data gradient: row (ci, r) -> dx[b][ci][rowS jt + r]
  const int nch = p.M / p.rowS;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wco * 32 + acc_row(r, hh);
    if (m >= p.M) continue;
    const int ci = m / p.rowS, jj = jt * p.rowS + (m - ci * p.rowS);
    if (jj >= p.LoutTotal) continue;
    const int64_t o = ((int64_t)b * nch + ci) * p.LoutTotal + jj;
    float v = acc[r] + (p.bias ? p.bias[ci] : 0.f);
    if (p.gate) v *= (p.gate[o] > 0.f ? 1.f : p.gate_slope);
    if (p.resid) v += p.resid[o];
    v *= (p.omask ? p.omask[(int64_t)b * p.LoutTotal + jj] : 1.f) * p.out_scale;
    p.y[o] = p.accumulate ? p.y[o] + v : v;
  }
}
template <int WCO>
__device__ __forceinline__ void conv_tile_epilogue(const ConvMfmaParams& p, const f32x16& acc0, const f32x16& acc1, int wl,
                                                   int wco, int col, int hh, int j0, int m0, int b0, int SEG) {
  conv_tile_col<WCO>(p, acc0, 0, wl, wco, col, hh, j0, m0, b0, SEG);
  conv_tile_col<WCO>(p, acc1, 1, wl, wco, col, hh, j0, m0, b0, SEG);
}

// ---- inner loop shared by the split-bf16 kernels: CW 32-row blocks of output channels x two 32-position blocks per wave,
// fragments, then 6 CW MFMAs per tap
// F16 (last template argument of everything below, default false): the single-pass mode (TTTS_CONV_F16X1) -- the `hi` arrays hold
// fp16 values, the `lo` arrays are neither read nor multiplied, one MFMA group per tap instead of three.
template <int CW>
struct B3Frag { bf16x8 ah[CW], al[CW], bh[2], bl[2]; };
template <int CW, bool F16 = false>
__device__ __forceinline__ void b3_load(B3Frag<CW>& f, const bf16* xh, const bf16* xl, const bf16* ah, const bf16* al,
                                        const int (&arow)[CW], int bpos0, int bpos1, int k, int dil8) {
#pragma unroll
  for (int i = 0; i < CW; ++i) {
    f.ah[i] = *reinterpret_cast<const bf16x8*>(ah + arow[i] + k * 16);
    if (!F16) f.al[i] = *reinterpret_cast<const bf16x8*>(al + arow[i] + k * 16);
  }
  const int ko = k * dil8;
  f.bh[0] = *reinterpret_cast<const bf16x8*>(xh + bpos0 + ko);
  if (!F16) f.bl[0] = *reinterpret_cast<const bf16x8*>(xl + bpos0 + ko);
  f.bh[1] = *reinterpret_cast<const bf16x8*>(xh + bpos1 + ko);
  if (!F16) f.bl[1] = *reinterpret_cast<const bf16x8*>(xl + bpos1 + ko);
}
template <int CW, bool F16 = false>
__device__ __forceinline__ void b3_mma(const B3Frag<CW>& f, f32x16 (&acc)[CW][2]) {
  if constexpr (F16) {
#pragma unroll
    for (int i = 0; i < CW; ++i)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[i][t] = mfma32_f16(f.ah[i], f.bh[t], acc[i][t]);
    return;
  }
#pragma unroll
  for (int i = 0; i < CW; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[i][t] = mfma32(f.al[i], f.bh[t], acc[i][t]);
#pragma unroll
  for (int i = 0; i < CW; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[i][t] = mfma32(f.ah[i], f.bl[t], acc[i][t]);
#pragma unroll
  for (int i = 0; i < CW; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[i][t] = mfma32(f.ah[i], f.bh[t], acc[i][t]);
}
// Software-pipelined form of a stage for compile-time tap counts (round 4).  The plain unrolled loop below compiles to
// "read six fragments, s_waitcnt lgkmcnt(0), a few MFMAs" twelve times per 60 MFMAs: every wait drains ALL reads in flight, so
// the matrix pipe idles for an LDS latency a dozen times per 16-channel block.  Here tap k + 1's fragments are requested before
// tap k's MFMAs (a second fragment set, +16 CW VGPRs; the compiler's counted lgkmcnt then only waits for the older set), the
// schedule is pinned by sched_barrier, and the three MFMA groups of a tap are separated by `piece(slot)` -- the DMA kernel
// issues the NEXT stage's global_load_lds there, one or two at a time in the shadow of the running MFMAs, instead of a burst
// of ~26 in front of the stage during which the wave feeds nothing to the matrix cores.  Same MFMA order as b3_mma: results
// are bit-identical.
template <int CW, int KT, bool F16 = false, typename Piece>
__device__ __forceinline__ void b3_stage_pipe(const bf16* xh, const bf16* xl, const bf16* ah, const bf16* al, const int (&arow)[CW],
                                              int bpos0, int bpos1, int dil8, f32x16 (&acc)[CW][2], Piece&& piece) {
#ifdef TTTS_EXP_NO_MFMA                                     // (what-bounds-it build: the stage's DMA slots without its arithmetic)
#pragma unroll
  for (int s = 0; s < 3 * KT; ++s) piece(s);
  return;
#endif
  B3Frag<CW> f[2];
  b3_load<CW, F16>(f[0], xh, xl, ah, al, arow, bpos0, bpos1, 0, dil8);
  __builtin_amdgcn_sched_barrier(0);       // (tap 0's reads stay ahead of tap 1's: the first wait is then counted too)
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    const B3Frag<CW>& c = f[k & 1];
    if (k + 1 < KT) b3_load<CW, F16>(f[(k + 1) & 1], xh, xl, ah, al, arow, bpos0, bpos1, k + 1, dil8);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (F16) {                   // one MFMA group per tap; the tap's three DMA slots follow it
#pragma unroll
      for (int i = 0; i < CW; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[i][t] = mfma32_f16(c.ah[i], c.bh[t], acc[i][t]);
      __builtin_amdgcn_sched_barrier(0);
      piece(3 * k); piece(3 * k + 1); piece(3 * k + 2);
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int i = 0; i < CW; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[i][t] = mfma32(c.al[i], c.bh[t], acc[i][t]);
      __builtin_amdgcn_sched_barrier(0);
      piece(3 * k);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < CW; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[i][t] = mfma32(c.ah[i], c.bl[t], acc[i][t]);
      __builtin_amdgcn_sched_barrier(0);
      piece(3 * k + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < CW; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[i][t] = mfma32(c.ah[i], c.bh[t], acc[i][t]);
      __builtin_amdgcn_sched_barrier(0);
      piece(3 * k + 2);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}
struct B3NoPiece { __device__ __forceinline__ void operator()(int) const {} };
// where the pipelined form is used: compile-time taps, except the 64 x 64 wave tile with 11 taps (its per-tap fragment
// addresses + the second fragment set need 289 VGPRs: one wave per SIMD).  -DTTTS_CONV_NO_PIPE: nowhere (A/B builds).
template <int CW, int KT>
__host__ __device__ constexpr bool b3_pipe() {
#ifdef TTTS_CONV_NO_PIPE
  return false;
#else
  return KT > 0 && !(CW == 2 && KT > 7);
#endif
}

// KT = compile-time tap count (0: runtime K).  With the taps unrolled the scheduler hoists the LDS fragment reads of later
// taps above the MFMAs of earlier ones; the runtime loop waits for its six reads before every group of six MFMAs.
template <int CW, int KT, bool F16 = false>
__device__ __forceinline__ void b3_stage(const bf16* xh, const bf16* xl, const bf16* ah, const bf16* al, const int (&arow)[CW],
                                         int bpos0, int bpos1, int K, int dil8, f32x16 (&acc)[CW][2]) {
  // (an explicit two-deep fragment prefetch across taps with runtime K measured 10-14 % SLOWER than the plain loop: +44 VGPRs
  // and branchy control)
  if (KT > 0) {
    if constexpr (b3_pipe<CW, KT>()) {
      b3_stage_pipe<CW, KT, F16>(xh, xl, ah, al, arow, bpos0, bpos1, dil8, acc, B3NoPiece());
    } else {
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        B3Frag<CW> f;
        b3_load<CW, F16>(f, xh, xl, ah, al, arow, bpos0, bpos1, k, dil8);
        b3_mma<CW, F16>(f, acc);
      }
    }
  } else {
#pragma unroll 2
    for (int k = 0; k < K; ++k) {
      B3Frag<CW> f;
      b3_load<CW, F16>(f, xh, xl, ah, al, arow, bpos0, bpos1, k, dil8);
      b3_mma<CW, F16>(f, acc);
    }
  }
}

// Variant 1: the input is split ON THE FLY while it is staged (few output-channel tiles re-read it: the 16..128-channel
// long-row layers, where a separate split pass would cost more HBM traffic than it saves).  Single LDS stage; overlap comes
// from several workgroups per CU.
// waves per SIMD the register allocation has to leave room for: what the LDS footprint allows anyway (the round-6 staging keeps
// more loads in flight; without the bound <2, 7> and <2, 3> each lost a resident wave to it)
__host__ __device__ constexpr int b3_min_waves(int WCO, int KT) {
#ifdef TTTS_B3_WPF_ALL
  if (WCO == 2) return (KT == 11 || KT == 7 || KT == 5) ? 2 : 3;
#endif
  if (WCO == 2) return KT == 11 ? 2 : 3;    // (<2, 11> stages 58 KB: two workgroups per CU whatever the registers allow)
  return KT == 0 ? 3 : KT <= 3 ? 4 : 3;
}
template <int WCO, int KT, bool F16 = false>
__global__ __launch_bounds__(256, b3_min_waves(WCO, KT)) void conv1d_bf16x3_kernel(ConvMfmaParams p) {
  constexpr int MT = 32 * WCO, WL = 4 / WCO, LT = 64 * WL;
  extern __shared__ __attribute__((aligned(16))) float cm_smem[];
  const int K = p.K, SEG = p.SEG, nseg = LT / SEG;
  const int lin_s = (SEG - 1) * p.stride + (K - 1) * p.dil + 1, lin_t = nseg * lin_s;
  const int apitch = K * 16 + 8;                      // bf16 elements per weight row (+8: spreads rows over the banks)
  // input strip, hi and lo: [2 channel halves][lin_t positions][8 channels].  A lane's 16-byte B fragment is (its position,
  // its channel half = lane >> 5), so the 16 lanes of a ds_read_b128 group read 16 consecutive positions of ONE half: 16
  // distinct 16-byte slots, conflict-free for stride 1 and 3 (the position-major [pos][16 ch] records this replaces put a
  // group on 8 slots: 2-way conflicts on four of the six fragment reads per k-step, PMC: SQ_LDS_BANK_CONFLICT 1.56x
  // SQ_ACTIVE_INST_LDS, and the loop is LDS-bound)
  bf16* xh = reinterpret_cast<bf16*>(cm_smem);        // [2][lin_t][8]
  bf16* xl = xh + lin_t * 16;
  const int xhalf = lin_t * 8;
  bf16* ah = xl + lin_t * 16;                         // [MT][apitch]
  bf16* al = ah + MT * apitch;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, col = lane & 31;
  const int wco = wave % WCO, wl = wave / WCO;
  // (output-channel tile fastest, XCD-contiguous: the one or two workgroups over the same input strip share an L2, and so do
  // neighbouring strips' halos)
#ifdef TTTS_EXP_NO_XCD_OTF
  const int j0 = blockIdx.x * LT, m0 = blockIdx.y * MT, b0 = blockIdx.z * nseg;
#else
  const int lin_b = xcd_order(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);
  const int j0 = ((lin_b / gridDim.y) % gridDim.x) * LT, m0 = (lin_b % gridDim.y) * MT, b0 = (lin_b / (gridDim.y * gridDim.x)) * nseg;
#endif
  const int in0 = j0 * p.stride - p.pad;
  f32x16 acc[1][2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[0][0][r] = 0.f; acc[0][1][r] = 0.f; }
  const int c0 = wl * 64 + col, c1 = c0 + 32;
  const int bpos0 = ((c0 / SEG) * lin_s + (c0 % SEG) * p.stride) * 8 + hh * xhalf;
  const int bpos1 = ((c1 / SEG) * lin_s + (c1 % SEG) * p.stride) * 8 + hh * xhalf;
  const int arow[1] = {(wco * 32 + col) * apitch + hh * 8};
  const int nblk = (p.N + 15) / 16;
  const int64_t slab = (int64_t)p.Mpad * K * 16;      // elements per channel block of the split weights
  // weight staging map, the same in every stage: chunk tid + 256 i of the [MT][K][16] slab -> LDS element offset (PMC: the
  // per-chunk `ch / (2K)` made this loop 400 of the 512 VALU instructions a wave issued per stage -- as many issue cycles
  // as its MFMAs)
  constexpr int WCH = KT > 0 ? (MT * KT * 2 + 255) / 256 : MT * 16 * 2 / 256;   // chunks per thread (runtime K <= 16)
  const int wchunks = MT * K * 2, nw = (wchunks + 255) >> 8;      // nw: wave-uniform trip count (no exec-masked branches:
  int wlds[WCH];                                       // lanes beyond the slab load a clamped chunk and store it to a dump slot)
#pragma unroll
  for (int i = 0; i < WCH; ++i) {
    const int ch = tid + 256 * i, m = ch / (K * 2), r = ch - m * (K * 2);
    wlds[i] = ch < wchunks ? m * apitch + r * 8 : 2 * MT * apitch;            // dump slots: 2 x 8 elements behind `al`
  }
  const int wlast = (wchunks - 1 - tid) >> 8;          // last in-range chunk index of this thread (may be -1 -> clamp to 0)
  // One thread per strip position, 16 channels each (coalesced row segments); loads are unconditional from clamped addresses, then
  // selected (a guarded load is an exec-masked branch with its own wait -- sixteen of them per position serialised the staging of
  // every 16-channel block).  Round 6, two changes to the staging:
  //  * the strip of 256-position tiles is 256 positions + a halo ((K - 1) dil <= 50 more): as a loop `pp = tid; pp < lin_t; pp += 256`
  //    the halo was a second full memory round trip that only wave 0 made, with the other three waves parked at the barrier.  Every
  //    thread now also requests a QUARTER (4 channels) of one halo position together with its main position;
  //  * the NEXT stage's strip is requested right behind the barrier that publishes the current one and stays in flight under the
  //    current stage's MFMAs (16-20 VGPRs).  A stage was: barrier, request, ~2.5 k cycles of HBM latency, convert, weights
  //    (another round trip), barrier, ~2 k cycles of MFMAs -- a wave fed the matrix cores for 14 % of its life
  //    (profiles/r06_pmc_conv_onthefly.txt).  (Round 2 measured a prefetch that cost a resident wave and lost; the resident-wave
  //    count is pinned by b3_min_waves now.)
  unsigned ev = 0;
  bool ok_m, ok_q = false;
  const int cq = (tid & 3) * 4;
  auto strip_src = [&](int pp, bool& ok) {
    const int sg = nseg == 1 ? 0 : pp / lin_s, pos = pp - sg * lin_s, gi = in0 + pos;
    ok = b0 + sg < p.B && gi >= 0 && gi < p.Lin;
    return p.x + ((int64_t)min(b0 + sg, p.B - 1) * p.N) * p.Lin + min(max(gi, 0), p.Lin - 1);
  };
  const float* xm = strip_src(min(tid, lin_t - 1), ok_m);
  const float* xq = xm;
  if constexpr (WCO == 1) xq = strip_src(min(256 + (tid >> 2), lin_t - 1), ok_q);
  float raw[16], rq[WCO == 1 ? 4 : 1];
  auto request = [&](int nb) {
#pragma unroll
    for (int c = 0; c < 16; ++c) raw[c] = xm[(int64_t)min(nb * 16 + c, p.N - 1) * p.Lin];
    if constexpr (WCO == 1) {
#pragma unroll
      for (int c = 0; c < 4; ++c) rq[c] = xq[(int64_t)min(nb * 16 + cq + c, p.N - 1) * p.Lin];
    }
  };
  auto split1 = [&](float v, bf16& hv, bf16& lv) {
    v = lrelu_f(v, p.in_slope);
    hv = F16 ? f16_slot_ev(v, ev) : (bf16)v;
    lv = (bf16)(v - (float)hv);
  };
  auto put16 = [&](int nb, int pp, const float (&rw)[16], bool ok) {
    bf16x8 h0, h1, l0, l1;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      bf16 hv, lv;
      split1((ok && nb * 16 + c < p.N) ? rw[c] : 0.f, hv, lv);
      if (c < 8) { h0[c] = hv; l0[c] = lv; } else { h1[c - 8] = hv; l1[c - 8] = lv; }
    }
    *reinterpret_cast<bf16x8*>(xh + pp * 8) = h0;
    *reinterpret_cast<bf16x8*>(xh + xhalf + pp * 8) = h1;
    if (!F16) {
      *reinterpret_cast<bf16x8*>(xl + pp * 8) = l0;
      *reinterpret_cast<bf16x8*>(xl + xhalf + pp * 8) = l1;
    }
  };
  // (the 256-position tiles of the 16 / 32-channel layers keep the synchronous form: with the strip held across the MFMAs they ran
  // 10 % SLOWER -- RB1(32) k11 46 -> 51 us, k7 d3 57 -> 64 us: one resident wave less at K <= 3, spills at K = 11 -- while the
  // 64-row tiles gained: RB1(128) k11 166 -> 150 us, RB1(64) k7 62.5 -> 57.5 us; tools/gpu_r6_x.sh)
  constexpr bool PF = WCO == 2;
  // weight chunks requested ahead of the strip's conversion (16 VGPRs per chunk in flight).  256-position tiles: the first three, in
  // front of the conversion.  64-row tiles: ALL of a stage's chunks, one stage ahead with the strip (TTTS_B3_WPF chunks; the weights
  // are re-read by every workgroup -- 45 KB per stage at K = 11, 460 MB of L2 -> LDS traffic per RB1(128) launch against 42 MB of
  // input -- and were two more synchronous round trips per stage)
  // (K = 5 / 7 at 64 rows: no register room at three resident waves -- -DTTTS_B3_WPF_ALL gives them the prefetch at two waves)
  // K = 5 at 64 rows (no register room to hold chunks across the MFMAs at three resident waves): its first chunks are requested at
  // the TOP of the stage, in front of the barrier -- they fly under the barrier wait and the strip's conversion, when the fragment
  // registers are dead (W_TOP)
#ifdef TTTS_B3_WPF_ALL
  constexpr int WG0 = KT == 0 ? 0 : WCO == 2 ? WCH : WCH < 3 ? WCH : 3;
  constexpr bool W_AHEAD = true;
#else
  constexpr bool W_AHEAD = WCO == 2 && (KT == 11 || KT <= 3);
  constexpr int WG0 = KT == 0 ? 0 : WCO == 2 ? (W_AHEAD ? WCH : KT == 5 ? 3 : 0) : WCH < 3 ? WCH : 3;
#endif
  constexpr bool W_TOP = PF && !W_AHEAD && KT == 5;   // (K = 7: 13 spilled registers with the requests above the barrier)
  bf16x8 wh0[WG0 > 0 ? WG0 : 1], wl0[WG0 > 0 ? WG0 : 1];
  auto request_w = [&](int nb) {
    const bf16* gh = p.a_hi + nb * slab + (int64_t)m0 * K * 16 + tid * 8;
    const bf16* gl = p.a_lo + nb * slab + (int64_t)m0 * K * 16 + tid * 8;
#pragma unroll
    for (int j = 0; j < WG0; ++j) {
      if (j < nw) {
        const int ii = max(min(j, wlast), 0) * 2048;
        wh0[j] = *reinterpret_cast<const bf16x8*>(gh + ii);
        if (!F16) wl0[j] = *reinterpret_cast<const bf16x8*>(gl + ii);
      }
    }
  };
  if (PF) { request(0); if (W_AHEAD) request_w(0); }
  for (int nb = 0; nb < nblk; ++nb) {
    if (W_TOP) request_w(nb);
    __syncthreads();
    if (!PF) request(nb);
    if (!W_AHEAD && !W_TOP) request_w(nb);
    const bf16* gh = p.a_hi + nb * slab + (int64_t)m0 * K * 16 + tid * 8;
    const bf16* gl = p.a_lo + nb * slab + (int64_t)m0 * K * 16 + tid * 8;
    // this stage's strip: requested one stage ago
    if (tid < lin_t) put16(nb, tid, raw, ok_m);
    if constexpr (WCO == 1) {
      const int pq = 256 + (tid >> 2);
      if (pq < lin_t) {
        bf16x4 hq, lq;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          bf16 hv, lv;
          split1((ok_q && nb * 16 + cq + c < p.N) ? rq[c] : 0.f, hv, lv);
          hq[c] = hv; lq[c] = lv;
        }
        const int o = (cq >> 3) * xhalf + pq * 8 + (cq & 7);
        *reinterpret_cast<bf16x4*>(xh + o) = hq;
        if (!F16) *reinterpret_cast<bf16x4*>(xl + o) = lq;
      }
    }
    for (int pp = (WCO == 2 ? 256 : 320) + tid; pp < lin_t; pp += 256) {   // (beyond that: several short rows in one tile)
      bool ok;
      const float* xr = strip_src(pp, ok);
      float rw[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) rw[c] = xr[(int64_t)min(nb * 16 + c, p.N - 1) * p.Lin];
      put16(nb, pp, rw, ok);
    }
    // weights: the stage's [MT][K][16] slab is contiguous in the split arrays; 16-byte chunks, LDS slots precomputed (wlds)
    {
#pragma unroll
      for (int j = 0; j < WG0; ++j) {
        if (j < nw) {
          *reinterpret_cast<bf16x8*>(ah + wlds[j]) = wh0[j];
          if (!F16) *reinterpret_cast<bf16x8*>(ah + (wlds[j] == 2 * MT * apitch ? 2 * MT * apitch + 8 : wlds[j] + MT * apitch)) = wl0[j];
        }
      }
      // groups of three chunks in flight (all of them at once costs 16 VGPRs per chunk: <2, 7> went from four waves per SIMD
      // to three and lost 15 %)
      constexpr int GS = (WCO == 2 && KT == 7) ? 2 : 3;     // (<2, 7> with the prefetched strip: three chunks in flight spilled 3 VGPRs)
#pragma unroll
      for (int g = WG0; g < WCH; g += GS) {
        bf16x8 vh[GS], vl[GS];
#pragma unroll
        for (int j = 0; j < GS; ++j) {
          const int i = g + j;
          if (i < WCH && i < nw) {
            const int ii = max(min(i, wlast), 0) * 2048;
            vh[j] = *reinterpret_cast<const bf16x8*>(gh + ii);
            if (!F16) vl[j] = *reinterpret_cast<const bf16x8*>(gl + ii);
          }
        }
#pragma unroll
        for (int j = 0; j < GS; ++j) {
          const int i = g + j;
          if (i < WCH && i < nw) {
            *reinterpret_cast<bf16x8*>(ah + wlds[i]) = vh[j];
            if (!F16) *reinterpret_cast<bf16x8*>(ah + (wlds[i] == 2 * MT * apitch ? 2 * MT * apitch + 8 : wlds[i] + MT * apitch)) = vl[j];
          }
        }
      }
    }
    __syncthreads();
    if (PF && nb + 1 < nblk) { request(nb + 1); if (W_AHEAD) request_w(nb + 1); }   // in flight under this stage's MFMAs
    b3_stage<1, KT, F16>(xh, xl, ah, al, arow, bpos0, bpos1, K, p.dil * 8, acc);
  }
  if (F16) f16_events_commit(ev);
  conv_tile_epilogue<WCO>(p, acc[0][0], acc[0][1], wl, wco, col, hh, j0, m0, b0, SEG);
}

// Variant 2: PRE-SPLIT operands, LDS-DMA double buffer (layers whose input is re-read by >= 3 output-channel tiles).
// Both operands arrive as bf16 hi / lo arrays laid out so that an LDS stage is a verbatim copy of global memory:
//   input   [B][N/16][2 halves][Lp][8]  zero-padded rows (no bounds tests; conv_input_split_kernel writes the pads)
//   weights [N/16][Mpad][AP = K*16 + 8] rows already carry the LDS bank padding (conv_weight_split_kernel)
// so every stage is filled by global_load_lds_dwordx4 (16 bytes per lane straight into LDS, lane-linear destination: no VGPR
// round trip, no ds_write_b128 at 13 cycles each) while the previous stage feeds the matrix cores.  Tile 64 co x 128
// positions, waves 2 x 2, 16 input channels x K taps per stage.
// CW = 1: waves 2 (co) x 2 (positions), wave tile 32 x 64, workgroup tile 64 x 128.
// CW = 2: waves 1 x 4, wave tile 64 x 64, workgroup tile 64 x 256: 8 fragment reads feed 12 MFMAs (6 : 6 above) and a weight
//         stage is amortised over twice the positions -- for launches that still fill the chip with the larger tile.
constexpr int V2_WC = 6;   // most 64-slot DMA chunks of one weight array a wave issues per stage (K <= 11)
template <int CW, int KT, bool F16 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void conv1d_bf16x3_dma_kernel(ConvMfmaParams p) {
  constexpr int MT = 64, LT = 128 * CW, WCO = 2 / CW;
  constexpr int XC = CW == 1 ? 4 : 7;                       // most DMA chunks of one input array per wave and stage
  extern __shared__ __attribute__((aligned(16))) float cm_smem[];
  const int K = p.K, SEG = p.SEG, nseg = LT / SEG;
  const int lin_s = (SEG - 1) * p.stride + (K - 1) * p.dil + 1, lin_t = nseg * lin_s;
  const int AP = K * 16 + 8;
  const int nxs = 2 * lin_t, nws = MT * (2 * K + 1);         // 16-byte slots per input / weight array and stage
  const int nxc = (nxs + 63) >> 6, nwc = (nws + 63) >> 6;    // 64-slot chunks (the tail chunk over-writes into padding)
  const int XS = nxc * 512, WS = nwc * 512;                  // elements per array and stage
  const int NBS = p.NBS;                                     // channel blocks per stage
  const int STAGE1 = 2 * (XS + WS);                          // one block: [x hi][x lo][w hi][w lo]
  const int STAGE = NBS * STAGE1;
  bf16* smem = reinterpret_cast<bf16*>(cm_smem);
  const int xhalf = lin_t * 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, col = lane & 31;
  const int wco = wave % WCO, wl = wave / WCO;
  // logical tile order: output-channel tile fastest, then position tile, then batch fold -- and an XCD walks a contiguous
  // range of it, so the (few) position tiles an XCD works on are fetched into ITS L2 once and serve all their channel tiles
  // (PMC, round-robin order: 482 MB of fabric reads per DiscriminatorP 1024x1024 launch against 55 MB of operands)
  const int lin = xcd_order(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);
  const int by = lin % gridDim.y, bx = (lin / gridDim.y) % gridDim.x, bz = lin / (gridDim.y * gridDim.x);
  const int j0 = bx * LT, m0 = by * MT, b0 = bz * nseg;
  const int in0 = j0 * p.stride - p.pad;
  const int nblk = (p.N + 15) / 16;
  const int64_t xlo_d = p.x_lo - p.x_hi, wlo_d = p.a_lo - p.a_hi;
  const int64_t xstep = (int64_t)2 * p.Lp * 8, wstep = (int64_t)p.Mpad * AP;
  // this lane's DMA sources for stage 0 (stage s adds s * NBS blocks); chunk c = wave + 4 i of the stage's NBS * nxc (nwc)
  // chunks: block c / nxc, chunk-in-block c % nxc
  const bf16* xsrc[XC];
  const bf16* wsrc[V2_WC];
  int xdst[XC], wdst[V2_WC], xblk[XC], wblk[V2_WC];
#pragma unroll
  for (int i = 0; i < XC; ++i) {
    const int c = wave + 4 * i, blk = c / nxc, cc = c - blk * nxc;
    const int q = min(cc * 64 + lane, nxs - 1);
    const int half = q >= lin_t, pp = q - half * lin_t;
    const int sg = pp / lin_s, pos = pp - sg * lin_s;
    const int b = min(b0 + sg, p.B - 1);
    xsrc[i] = p.x_hi + ((((int64_t)b * nblk) * 2 + half) * p.Lp + (p.PADL + in0 + pos)) * 8 + blk * xstep;
    xdst[i] = blk * STAGE1 + cc * 512;
    xblk[i] = c < NBS * nxc ? blk : 1 << 20;
  }
#pragma unroll
  for (int i = 0; i < V2_WC; ++i) {
    const int c = wave + 4 * i, blk = c / nwc, cc = c - blk * nwc;
    const int q = min(cc * 64 + lane, nws - 1);
    wsrc[i] = p.a_hi + (int64_t)m0 * AP + (int64_t)q * 8 + blk * wstep;
    wdst[i] = blk * STAGE1 + 2 * XS + cc * 512;
    wblk[i] = c < NBS * nwc ? blk : 1 << 20;
  }
  // piece i of a stage's DMA work for this wave: i < XC an input chunk, else a weight chunk (hi and lo array each).  Inline-asm
  // DMA (lds_dma16_untracked): with the builtin in the kernel hipcc drains ALL LDS reads (lgkmcnt(0)) in front of every MFMA
  // group instead of counting, which defeats the fragment pipeline; the protocol is the stage-top vmcnt(0) + barrier below.
  const uint32_t lds0 = lds_byte_addr(smem);
  auto issue_piece = [&](int st_i, int buf, int i) {
#ifdef TTTS_EXP_NO_DMA                                      // (what-bounds-it build: only stage 0 is ever filled; results are wrong)
    if (st_i > 0) return;
#endif
    const uint32_t st = lds0 + (uint32_t)(buf * STAGE) * (uint32_t)sizeof(bf16);
    const int nb0 = st_i * NBS;
    if (i < XC) {
      const int j = i < XC ? i : 0;
      if (nb0 + __builtin_amdgcn_readfirstlane(xblk[j]) < nblk) {      // wave-uniform: a scalar branch
        const bf16* g = xsrc[j] + nb0 * xstep;
        lds_dma16_untracked(g, st + (uint32_t)xdst[j] * 2u);
        if (!F16) lds_dma16_untracked(g + xlo_d, st + (uint32_t)(xdst[j] + XS) * 2u);     // (F16: the lo arrays are never read)
      }
    } else {
      const int j = i >= XC ? i - XC : 0;
      if (nb0 + __builtin_amdgcn_readfirstlane(wblk[j]) < nblk) {
        const bf16* g = wsrc[j] + nb0 * wstep;
        lds_dma16_untracked(g, st + (uint32_t)wdst[j] * 2u);
        if (!F16) lds_dma16_untracked(g + wlo_d, st + (uint32_t)(wdst[j] + WS) * 2u);
      }
    }
  };
  auto issue = [&](int st_i, int buf) {
#pragma unroll
    for (int i = 0; i < XC + V2_WC; ++i) issue_piece(st_i, buf, i);
  };
  f32x16 acc[CW][2];
#pragma unroll
  for (int i = 0; i < CW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[i][0][r] = 0.f; acc[i][1][r] = 0.f; }
  const int c0 = wl * 64 + col, c1 = c0 + 32;
  const int bpos0 = ((c0 / SEG) * lin_s + (c0 % SEG) * p.stride) * 8 + hh * xhalf;
  const int bpos1 = ((c1 / SEG) * lin_s + (c1 % SEG) * p.stride) * 8 + hh * xhalf;
  int arow[CW];
#pragma unroll
  for (int i = 0; i < CW; ++i) arow[i] = ((wco * CW + i) * 32 + col) * AP + hh * 8;
  // (A ring of 3-5 stage buffers with DEPTH - 1 stages in flight and counted vmcnt waits -- one block per stage -- was measured
  // for the 1-tap / 3-tap layers and LOST: 1 x 1 convolution 33 -> 52 us.  The per-stage cost is the barrier + DMA issue, not the
  // DMA latency, so FEWER, LARGER stages (NBS) is the lever, not deeper look-ahead.)
  const int nstage = (nblk + NBS - 1) / NBS;
  issue(0, 0);
  for (int si = 0; si < nstage; ++si) {
    // stage si has landed (every wave drains its own DMA before the barrier) and nobody still reads the other buffer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef TTTS_EXP_NO_BARRIER
    if (si == 0)
#endif
    __syncthreads();
    const bf16* sb = smem + (si & 1) * STAGE;
    constexpr bool PIPE = b3_pipe<CW, KT>();
    if constexpr (PIPE) {
      // the next stage's DMA is issued piecewise between the MFMA groups of this stage's first block (3 KT slots)
      const bool more = si + 1 < nstage;
      constexpr int NSLOT = 3 * (KT > 0 ? KT : 1), NPIECE = XC + V2_WC;
      b3_stage_pipe<CW, (KT > 0 ? KT : 1), F16>(sb, sb + XS, sb + 2 * XS, sb + 2 * XS + WS, arow, bpos0, bpos1, p.dil * 8, acc, [&](int slot) {
        if (more) {
#pragma unroll
          for (int i = 0; i < NPIECE; ++i)
#ifdef TTTS_EXP_FRONT
            if (i / TTTS_EXP_FRONT == slot) issue_piece(si + 1, (si + 1) & 1, i);
#else
            if (i % NSLOT == slot) issue_piece(si + 1, (si + 1) & 1, i);
#endif
        }
      });
      for (int blk = 1; blk < NBS; ++blk) {
        if (si * NBS + blk >= nblk) break;
        const bf16* xh = sb + blk * STAGE1;
        b3_stage<CW, KT, F16>(xh, xh + XS, xh + 2 * XS, xh + 2 * XS + WS, arow, bpos0, bpos1, K, p.dil * 8, acc);
      }
    } else {
      if (si + 1 < nstage) issue(si + 1, (si + 1) & 1);
      for (int blk = 0; blk < NBS; ++blk) {
        if (si * NBS + blk >= nblk) break;
        const bf16* xh = sb + blk * STAGE1;
        b3_stage<CW, KT, F16>(xh, xh + XS, xh + 2 * XS, xh + 2 * XS + WS, arow, bpos0, bpos1, K, p.dil * 8, acc);
      }
    }
  }
  if (p.rowS > 0 && !(p.accumulate | (p.gate != nullptr) | (p.resid != nullptr) | (p.omask != nullptr)) &&
      (size_t)MT * (LT + 1) * sizeof(float) <= (size_t)p.Lreal) {       // (rowS > 0: Lreal = bytes of dynamic LDS the launch got)
    // Phase-merged data gradient: the rows of one channel are its `rowS` phases, so a direct store writes 4-byte elements rowS
    // apart (3x-10x the L2 write requests of a coalesced row).  The tile goes through LDS instead (the stage buffers are free now)
    // and is written back in dx order: consecutive lanes = consecutive dx positions S j + r of one channel.
    __syncthreads();                                         // every wave is done with the last stage
    float* tile = reinterpret_cast<float*>(smem);            // [MT][LT + 1]
    constexpr int LTP = LT + 1;
#pragma unroll
    for (int i = 0; i < CW; ++i)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          tile[((wco * CW + i) * 32 + acc_row(r, hh)) * LTP + wl * 64 + t * 32 + col] = acc[i][t][r];
    __syncthreads();
    const int S = p.rowS, nch = p.M / S;
    const int c_first = m0 / S, c_last = min((m0 + MT - 1) / S, nch - 1), span = S * LT;
    for (int c = c_first; c <= c_last; ++c) {
      for (int e = tid; e < span; e += 256) {
        const int jl = e / S, r = e - jl * S, ml = c * S + r - m0;
        if (ml < 0 || ml >= MT) continue;                    // (this channel's other phases belong to the neighbouring row tile)
        int b = b0 + jl / SEG, jt = j0 + jl % SEG;
        if (b >= p.B || jt >= p.Lout) continue;
        if (p.catLg) {                                       // virtual row -> (batch element, position)
          b = jt / p.catLg; jt -= b * p.catLg;
          if (jt >= p.catLout) continue;
        }
        const int jj = jt * S + r;
        if (jj >= p.LoutTotal) continue;
        const float v = (tile[ml * LTP + jl] + (p.bias ? p.bias[c] : 0.f)) * p.out_scale;
        p.y[((int64_t)b * nch + c) * p.LoutTotal + jj] = v;
      }
    }
    return;
  }
  // (written out, not a loop over i: with the larger round-6 epilogue body the loop was no longer unrolled before the accumulator
  // array was scalarised -- `acc[i]` with a run-time i put all four accumulators into scratch)
  conv_tile_epilogue<1>(p, acc[0][0], acc[0][1], wl, 0, col, hh, j0, m0 + (wco * CW) * 32, b0, SEG);
  if constexpr (CW == 2) conv_tile_epilogue<1>(p, acc[CW - 1][0], acc[CW - 1][1], wl, 0, col, hh, j0, m0 + (wco * CW + 1) * 32, b0, SEG);
}

// dw[i] += sum_split slab[split][i];  blockIdx.y sums a group of SLAB_G splits and adds its partial with one atomic (a single
// thread walking 1000 splits of a small weight tensor would be latency-bound; nsplit / SLAB_G atomics per element are few)
constexpr int SLAB_G = 32;
__global__ __launch_bounds__(256) void wgrad_slab_sum_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nsplit,
                                                             int64_t per) {
  const int s0 = blockIdx.y * SLAB_G, s1 = min(nsplit, s0 + SLAB_G);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;        // four loads in flight per step
    int sp = s0;
    for (; sp + 4 <= s1; sp += 4) {
      const float v0 = slab[sp * per + i], v1 = slab[(sp + 1) * per + i], v2 = slab[(sp + 2) * per + i], v3 = slab[(sp + 3) * per + i];
      a0 += v0; a1 += v1; a2 += v2; a3 += v3;
    }
    for (; sp < s1; ++sp) a0 += slab[sp * per + i];
    const float s = (a0 + a1) + (a2 + a3);
    if (gridDim.y == 1) dw[i] += s;
    else atomicAdd(dw + i, s);
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
  float* slab;         // non-NULL: per-split partial sums [split][Cout][Cin*K] (plain stores) instead of atomics into dw
  int SEGW;            // positions per segment of a 64-position reduction chunk: 64, or 16 for short rows (a chunk then holds
                       // 4 (batch element, 16-position window) segments: DiscriminatorP rows are 23..127 positions long)
  float* bslab;        // fused-taps kernel: per-split row sums of dy [split][Cout] (the bias gradient), or NULL
};
constexpr int WM_L = 64;

__global__ __launch_bounds__(256) void conv1d_wgrad_mfma_kernel(WgradMfmaParams p) {
  extern __shared__ __attribute__((aligned(16))) float cm_smem[];
  const int NK = p.Cin * p.K;
  int bx_, by_, bz_;
  xcd_tile(bx_, by_, bz_);
  const int n0 = bx_ * 64, co0 = by_ * 64;
  const int ci_first = n0 / p.K, ci_last = min(p.Cin - 1, (n0 + 63) / p.K), nch = ci_last - ci_first + 1;
  const int SEGW = p.SEGW, nsg = WM_L / SEGW;
  const int lin_s = (SEGW - 1) * p.stride + (p.K - 1) * p.dil + 1;      // input strip of one segment
  const int xpitch = (nsg * lin_s) | 1;
  float* dys = cm_smem;                        // [64 co][WM_L + 1]
  float* xs = cm_smem + 64 * (WM_L + 1);       // [nch][nsg][lin_s]
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
  const int nsl = (p.Lout + SEGW - 1) / SEGW;          // segments per batch element
  const int nsegs = p.B * nsl;                          // segments in all
  for (int cc = 0; cc < p.chunks_per_block; ++cc) {
    const int chunk = bz_ * p.chunks_per_block + cc;
    if (chunk * nsg >= nsegs) break;
    __syncthreads();
    {
      // thread -> fixed chunk position lc (256 % 64 == 0), rows co = wave, wave + 4, ...: the segment arithmetic is per chunk
      const int lc = lane, sgi = chunk * nsg + lc / SEGW;  // global segment -> (batch element, window)
      const int b = sgi / nsl, l = (sgi % nsl) * SEGW + lc % SEGW;
      const bool ok = sgi < nsegs && l < p.Lout;
      const float* src = p.dy + ((int64_t)b * p.Cout + co0) * p.Lout + l;
      for (int co = wave; co < 64; co += 4)
        dys[co * (WM_L + 1) + lc] = (ok && co0 + co < p.Cout) ? lrelu_f(src[(int64_t)co * p.Lout], p.dy_slope) : 0.f;
    }
    for (int c = wave; c < nch; c += 4)
      for (int sg = 0; sg < nsg; ++sg) {
        const int sgi = chunk * nsg + sg;
        const int b = sgi / nsl, in0 = (sgi % nsl) * SEGW * p.stride - p.pad;
        const float* xr = p.x + ((int64_t)b * p.Cin + ci_first + c) * p.Lin;
        for (int pos = lane; pos < lin_s; pos += 64) {
          const int gi = in0 + pos;
          xs[c * xpitch + sg * lin_s + pos] = (sgi < nsegs && gi >= 0 && gi < p.Lin) ? lrelu_f(xr[gi], p.x_slope) : 0.f;
        }
      }
    __syncthreads();
    for (int sg = 0; sg < nsg; ++sg) {
      const float* ar = arow + sg * SEGW;
      const float* bcol = xs + boff + sg * lin_s + hh * p.stride;
#pragma unroll 4
      for (int l = 0; l < SEGW; l += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[l + hh], ncol_ok ? bcol[l * p.stride] : 0.f, acc, 0, 0, 0);
    }
  }
  if (ncol_ok) {
    float* sl = p.slab ? p.slab + (int64_t)bz_ * p.Cout * NK : nullptr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wco * 32 + acc_row(r, hh);
      if (co < p.Cout) {
        if (sl) sl[(int64_t)co * NK + n] = acc[r];
        else atomicAdd(p.dw + (int64_t)co * NK + n, acc[r]);
      }
    }
  }
}


// ---- 1 x 1 convolutions without the operand pre-pass (round 6) ------------------------------------------------------------------
// The 192-channel 1 x 1 layers of the WaveNet stacks, the attention projections and the diffusion model's 512-channel linear layers
// took the general path: conv_input_split_kernel (a full pass over x: ~10 us) + the 64 x 256-tile DMA kernel (23-37 us for 1-4 GFLOP:
// six MFMAs per wave and 16-channel stage, all DMA latency).  ~350 such calls per VQ-VAE-GAN step, ~125 per diffusion step.
// Here a 1 x 1 convolution is what it is -- Y[b] = W X[b], a GEMM whose reduction is the channel axis:
//   * tile 64 output channels x 64 positions, four waves of 32 x 32, so a 192 x 256-frame x 32 layer launches 768 / 384 workgroups
//     of ~48 KB (3 per CU) instead of 96 / 192 large ones;
//   * x is read as fp32 straight from the tensor (coalesced 256-byte row pieces), leaky-relu'd and split into bf16 hi / lo ON THE WAY
//     into LDS, 128 channels per chunk, the next chunk's loads in flight under the current chunk's MFMAs -- no scratch copy;
//   * the weights' A fragments come straight from the pre-split [channel block][row][16] arrays (weight-split cache layout of the
//     on-the-fly kernel): the 32 rows x 32 bytes a wave needs per k-step are ONE contiguous 1-KB piece, i.e. a fully coalesced
//     global_load_dwordx4 per fragment -- no LDS for the weights at all.
// Same products in the same order as b3_mma (lo*hi, hi*lo, hi*hi per 16-channel block, blocks ascending): bit-identical to the
// path it replaces.  Epilogue: the non-phase-merged branch of conv_tile_epilogue (bias, per-sample bias, gate, residual, tanh /
// leaky-relu, mask, scale, accumulate, dual destination).  Flag 512: off (A/B switch).
// TALL (round 6, wide layers: M % 128 == 0, M >= 512 -- the diffusion model's 512 / 1024 / 1536-row projections): 128 x 64 tiles, the
// four waves stacked along M, each 32 rows x BOTH 32-position blocks.  Every output-channel tile re-splits the x tile it shares with the
// other M / tile-rows workgroups (VALU : MFMA cycles ~1.7 : 1 at 64 rows, whatever M is): twice the rows halve that, and a weight
// fragment feeds two MFMA triples instead of one.  Same products in the same order per output element: bit-identical to the 64-row form.
// FULLK: N % 128 == 0 (every chunk has all eight k-steps): the k-steps of a chunk are ONE straight-line block.
template <bool F16, bool TALL = false, bool FULLK = false>
__global__ __launch_bounds__(256, 3) void conv1x1_b3_kernel(ConvMfmaParams p) {
  constexpr int KC = 128, NTL = 64, NCB = TALL ? 2 : 1;
  __shared__ __attribute__((aligned(16))) bf16 xs[2][KC / 8][NTL][8];          // [hi | lo][8-channel group][position][8]: 32 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, col = lane & 31;
  const int wm = TALL ? wave : (wave & 1), wn = TALL ? 0 : (wave >> 1);
  // tile order: output-channel tile fastest, dealt to the XCDs in contiguous runs (xcd_order) -- the M / 64 (128) workgroups that read
  // one input tile then share an L2.  (Round-robin order: every XCD fetched every input tile -- the diffusion step's 208 launches
  // read 18.6 GB from the fabric for ~3 GB of operands, profiles/r06_diffusion_pmc_traffic.json)
  const int lin_b = xcd_order(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);
  const int by_ = lin_b % gridDim.y, bx_ = (lin_b / gridDim.y) % gridDim.x, bz_ = lin_b / (gridDim.y * gridDim.x);
  const int j0 = bx_ * NTL, m0 = by_ * (TALL ? 128 : 64), b = bz_;
  const int nblk = (p.N + 15) / 16, nchunk = (p.N + KC - 1) / KC;
  const float* xb = p.x + (int64_t)b * p.N * p.Lin;
  // staging: position sn, first 8-channel group sk (+4 per round).  sk is the WAVE index, made scalar: a row's base address
  // (channel x Lin) is then SALU arithmetic and the 32 requests of a chunk share ONE vector offset (the position).  PMC at the
  // diffusion qkv shape before (r6aj): 2 782 VALU instructions per wave around 192 MFMAs -- 14.5 per MFMA, a SIMD's VALU + MFMA issue
  // time 93 % of the waves' life -- most of them per-request 64-bit address arithmetic, clamps and selects of the staging below.
  const int sn = tid & 63, sk = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int spos = min(j0 + sn, p.Lin - 1);
  const bool sok = j0 + sn < p.Lin;
  const bool tile_in = j0 + NTL <= p.Lin;                                       // workgroup-uniform: every position of the tile exists
  const bool plain_in = p.in_slope == 1.f;                                      // uniform: no leaky-relu on the input
  float raw[4][8];
  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int ch0 = c0 + (it * 4 + sk) * 8;                                   // scalar
#pragma unroll
      for (int e = 0; e < 8; ++e)                                               // clamped (scalar) row, selected below
        raw[it][e] = (xb + (int64_t)min(ch0 + e, p.N - 1) * p.Lin)[spos];
    }
  };
  unsigned ev = 0;
  auto store_chunk = [&](int c0) {
    const bool all_in = tile_in && c0 + KC <= p.N;                              // (uniform) nothing to select
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int k8 = it * 4 + sk, ch0 = c0 + k8 * 8;
      bf16x8 h, l;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = raw[it][e];
      if (!all_in) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (sok && ch0 + e < p.N) ? v[e] : 0.f;
      }
      if (!plain_in) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = lrelu_f(v[e], p.in_slope);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bf16 hv = F16 ? f16_slot_ev(v[e], ev) : (bf16)v[e];
        h[e] = hv;
        l[e] = (bf16)(v[e] - (float)hv);
      }
      *reinterpret_cast<bf16x8*>(&xs[0][k8][sn][0]) = h;
      if (!F16) *reinterpret_cast<bf16x8*>(&xs[1][k8][sn][0]) = l;
    }
  };
  f32x16 acc[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
  const bf16* ahp = p.a_hi + ((int64_t)(m0 + wm * 32 + col)) * 16 + hh * 8;     // + kb * Mpad * 16 per 16-channel block
  const bf16* alp = p.a_lo + ((int64_t)(m0 + wm * 32 + col)) * 16 + hh * 8;
  const int64_t astep = (int64_t)p.Mpad * 16;
  load_chunk(0);
  for (int c = 0; c < nchunk; ++c) {
    bf16x8 ah[KC / 16], al[KC / 16];
#pragma unroll
    for (int ks = 0; ks < KC / 16; ++ks) {
      const int kb = min(c * (KC / 16) + ks, nblk - 1);                         // (blocks past the end: loaded clamped, never multiplied)
      ah[ks] = *reinterpret_cast<const bf16x8*>(ahp + kb * astep);
      if (!F16) al[ks] = *reinterpret_cast<const bf16x8*>(alp + kb * astep);
    }
    // (pinned here: the compiler sank these 16 requests to the front of the MFMA loop -- vmcnt(14), (12), ... (0) in front of every
    // k-step, i.e. the whole L2 round trip exposed once per chunk instead of flying under the conversion and the two barriers)
    __builtin_amdgcn_sched_barrier(0);
    if (c) __syncthreads();                                                     // the previous chunk's fragments have been read
    store_chunk(c * KC);
    __syncthreads();
    // UNCONDITIONAL (the last chunk re-requests itself, unused): behind `if (c + 1 < nchunk)` the two paths met in front of the MFMAs
    // and the compiler's wait-count merge took the stricter one -- s_waitcnt vmcnt(14) for the first weight fragment, which drains
    // the 32 requests just issued for the next chunk: nothing flew under the MFMAs (ISA read, round 6)
    load_chunk(min(c + 1, nchunk - 1) * KC);
    __builtin_amdgcn_sched_barrier(0);      // (and they stay in front of the MFMAs: in the 128-row form they had been sunk behind them)
    auto kstep = [&](int ks) {
      if constexpr (TALL && !F16) {
        // the two position blocks' products interleaved (consecutive MFMAs on different accumulators, each accumulator's own order
        // unchanged -- as b3_mma does)
        const bf16x8 bh0 = *reinterpret_cast<const bf16x8*>(&xs[0][2 * ks + hh][col][0]);
        const bf16x8 bh1 = *reinterpret_cast<const bf16x8*>(&xs[0][2 * ks + hh][32 + col][0]);
        const bf16x8 bl0 = *reinterpret_cast<const bf16x8*>(&xs[1][2 * ks + hh][col][0]);
        const bf16x8 bl1 = *reinterpret_cast<const bf16x8*>(&xs[1][2 * ks + hh][32 + col][0]);
        acc[0] = mfma32(al[ks], bh0, acc[0]);
        acc[NCB - 1] = mfma32(al[ks], bh1, acc[NCB - 1]);
        acc[0] = mfma32(ah[ks], bl0, acc[0]);
        acc[NCB - 1] = mfma32(ah[ks], bl1, acc[NCB - 1]);
        acc[0] = mfma32(ah[ks], bh0, acc[0]);
        acc[NCB - 1] = mfma32(ah[ks], bh1, acc[NCB - 1]);
      } else {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          const int pc = (TALL ? cb : wn) * 32 + col;
          const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&xs[0][2 * ks + hh][pc][0]);
          if constexpr (F16) {
            acc[cb] = mfma32_f16(ah[ks], bh, acc[cb]);
          } else {
            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&xs[1][2 * ks + hh][pc][0]);
            acc[cb] = mfma32(al[ks], bh, acc[cb]);
            acc[cb] = mfma32(ah[ks], bl, acc[cb]);
            acc[cb] = mfma32(ah[ks], bh, acc[cb]);
          }
        }
      }
    };
    // A whole chunk runs as ONE straight-line block.  With a (workgroup-uniform) `if (block < nblk)` around every k-step each k-step
    // was its own basic block and the compiler's wait-count bookkeeping merged conservatively at every block entry: s_waitcnt
    // vmcnt(14) in front of the first MFMA of a chunk -- i.e. "all but 14 requests done", which drains the NEXT chunk's 32 input
    // requests that had just been issued to fly under these MFMAs (ISA read, round 6).
    if constexpr (FULLK) {
#pragma unroll
      for (int ks = 0; ks < KC / 16; ++ks) kstep(ks);
    } else {
#pragma unroll
      for (int ks = 0; ks < KC / 16; ++ks)
        if (c * (KC / 16) + ks < nblk) kstep(ks);                                // workgroup-uniform
    }
  }
  if (F16) f16_events_commit(ev);
  {
    const int j = j0 + wn * 32 + col;
    if (j < p.Lout) conv_store_col16(p, acc[0], m0 + wm * 32 + 4 * hh, b, j, p.omask ? p.omask[(int64_t)b * p.LoutTotal + j] : 1.f);
  }
  if constexpr (TALL) {
    const int j = j0 + 32 + col;
    if (j < p.Lout) conv_store_col16(p, acc[NCB - 1], m0 + wm * 32 + 4 * hh, b, j, p.omask ? p.omask[(int64_t)b * p.LoutTotal + j] : 1.f);
  }
}

static bool conv1x1_b3_fits(const ConvMfmaParams& p, const ConvCtx& cx) {
  return p.K == 1 && p.stride == 1 && p.dil == 1 && p.pad == 0 && p.rowS == 0 && p.catLg == 0 && !p.x_hi && p.out_stride == 1 &&
         p.out_off == 0 && p.Lout == p.Lin && p.LoutTotal == p.Lout && p.Lout >= 48 && p.N >= 32 && !(cx.flags & 512);
}
static int conv1x1_b3_launch(ConvMfmaParams p, const ConvCtx& cx, hipStream_t stream, bool* handled) {
  const int nblk = (p.N + 15) / 16;
  p.Mpad = (int)(cdiv(p.M, 64) * 64);
  const int64_t elems = (int64_t)nblk * p.Mpad * 16;
  if (2 * elems * (int64_t)sizeof(bf16) > cx.ws_bytes) return TTTS_OK;
  bf16* hi = static_cast<bf16*>(cx.ws);
  bf16* lo = hi + elems;
  bool need_split = true;
  wsplit_lookup(p, cx, nblk, 16, elems, stream, &hi, &lo, &need_split);       // (the on-the-fly kernel's K * 16 layout: entries are shared)
  p.a_hi = hi; p.a_lo = lo;
  const int f16 = (cx.flags & TTTS_CONV_F16X1) ? 1 : 0;
  if (need_split)
    conv_weight_split_kernel<<<(int)std::min<int64_t>(cdiv(elems, 256), 2048), 256, 0, stream>>>(p.w, hi, lo, p.M, p.N, p.Mpad, nblk, 1, p.Kmem,
                                                                                              p.transposed, p.tap_off, p.tap_stride, 16, 0, 0, 0, f16);
  // 128-row tiles for the wide layers (flag 1024: off); the 192 / 384-row WaveNet layers keep 64 rows -- they need the workgroups
  const bool tall = p.M % 128 == 0 && p.M >= 512 && !(cx.flags & 1024);
  const dim3 grid((unsigned)cdiv(p.Lout, 64), (unsigned)(p.Mpad / (tall ? 128 : 64)), (unsigned)p.B);
  const bool fullk = p.N % 128 == 0;
  if (tall) {
    if (f16) conv1x1_b3_kernel<true, true><<<grid, 256, 0, stream>>>(p);
    else if (fullk) conv1x1_b3_kernel<false, true, true><<<grid, 256, 0, stream>>>(p);
    else conv1x1_b3_kernel<false, true><<<grid, 256, 0, stream>>>(p);
  } else if (f16) conv1x1_b3_kernel<true><<<grid, 256, 0, stream>>>(p);
  else if (fullk) conv1x1_b3_kernel<false, false, true><<<grid, 256, 0, stream>>>(p);
  else conv1x1_b3_kernel<false><<<grid, 256, 0, stream>>>(p);
  *handled = true;
  return check_launch("conv1x1_b3");
}

static int set_attr_once(const void* fn, OnceFlag& done) {
  if (done) return TTTS_OK;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return fail(TTTS_EHIP, "conv_mfma: hipFuncSetAttribute: %s", hipGetErrorString(e));
  done = true;
  return TTTS_OK;
}

// ---- dispatch helpers used by conv.hip ------------------------------------------------------------------------------------
// Returns TTTS_OK and sets *handled when the MFMA path took the launch.
template <int WCO, int NW>
static int conv1d_mfma_launch_t(ConvMfmaParams p, const ConvCtx& cx, hipStream_t stream, bool* handled) {
  constexpr int MT = 32 * WCO, LT = 64 * (NW / WCO);
  const int K = p.K, SEG = p.SEG > LT ? LT : p.SEG;
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
  static OnceFlag attr;
  int rc = set_attr_once(reinterpret_cast<const void*>(conv1d_mfma_kernel<WCO, NW>), attr);
  if (rc) return rc;
  conv1d_mfma_kernel<WCO, NW><<<grid, 64 * NW, smem, stream>>>(p);
  *handled = true;
  return check_launch("conv1d_mfma");
}


// ---- split-bf16 launchers ------------------------------------------------------------------------------------------------
// requested segment length (batch folding for short rows), clipped to the tile by the launchers
static int conv_seg_request(int Lout, int B) {
  if (B <= 1) return 1 << 20;
  return Lout <= 32 ? 32 : (Lout <= 64 ? 64 : (Lout <= 128 ? 128 : 1 << 20));
}

// geometry of the DMA-fed kernel for one launch: LDS bytes, chunk counts, and the zero pads its pre-split input rows need
struct DmaGeom { size_t smem; int nxc, nwc, padl, padr, nbs; int64_t wgs; bool ok; };
static DmaGeom conv_dma_geom(const ConvMfmaParams& p, int CW) {
  const int MT = 64, LT = 128 * CW;
  const int SEG = p.SEG > LT ? LT : p.SEG, K = p.K;
  const int lin_t = (LT / SEG) * ((SEG - 1) * p.stride + (K - 1) * p.dil + 1);
  DmaGeom g;
  g.nxc = (2 * lin_t + 63) / 64;
  g.nwc = (MT * (2 * K + 1) + 63) / 64;
  const int last = (SEG == LT ? (int)cdiv(p.Lout, LT) * LT : SEG) - 1;   // last row-relative output position a tile touches
  g.padl = std::max(0, p.pad);
  g.padr = std::max(0, last * p.stride + (K - 1) * p.dil - p.pad + 1 - p.Lin);
  g.wgs = cdiv(p.Lout, SEG == LT ? LT : SEG) * cdiv(p.M, MT) * cdiv(p.B, LT / SEG);
  // channel blocks per stage: as many as the chunk tables and LDS allow -- two workgroups per CU when the launch has more
  // than one per CU, the whole LDS otherwise (a launch of <= 256 workgroups only has its own stages to hide latency behind)
  const int xcmax = 4 * (CW == 1 ? 4 : 7), nblk = (p.N + 15) / 16;
  const size_t stage1 = (size_t)2 * (g.nxc + g.nwc) * 512 * sizeof(bf16);
  const size_t lds_cap = g.wgs > 256 ? 78 * 1024 : 150 * 1024;
  g.nbs = 1;
  for (int cand = 4; cand >= 2; cand >>= 1)
    if (cand * g.nxc <= xcmax && cand * g.nwc <= 4 * V2_WC && cand <= nblk && 2 * cand * stage1 <= lds_cap) { g.nbs = cand; break; }
  g.smem = 2 * g.nbs * stage1;
  g.ok = g.nxc <= xcmax && g.nwc <= 4 * V2_WC && g.smem <= 150 * 1024 && (p.M > 32 || (p.rowS < 0 && p.M >= 16)) && p.N >= 16;
  return g;
}
// the 64 x 256 tile when it still yields >= 2 workgroups per CU (flag 131072: never, flag 262144: whenever it fits)
static int conv_dma_pick(const ConvMfmaParams& p, const ConvCtx& cx) {
  const DmaGeom w = conv_dma_geom(p, 2);
  if ((cx.flags & 131072) || !w.ok) return 1;
  return (w.wgs >= 512 || (cx.flags & 262144)) ? 2 : 1;
}

static int conv1d_bf16x3_dma_launch(ConvMfmaParams p, const ConvCtx& cx, hipStream_t stream, bool* handled, int CW) {
  // short rows: instead of folding power-of-two segments of several batch elements into a tile (a 37-position row uses 58 %
  // of a 64-position segment), lay the whole batch end to end as one virtual row with zero gaps wide enough for the taps
  // (flag 134217728: keep the segment folding)
  int cat_w = 0, cat_b = 0, cat_l = 0;
  // (round 6: the phase-merged strided data gradients too -- their rows are the ceil(L / stride) positions of a phase, 85 of a
  // 128-position segment for DiscriminatorP's period 3; flag 524288: segments for them as before)
  const bool cat_rows = p.rowS == 0 ? p.LoutTotal == p.Lout : (p.rowS > 0 && p.stride == 1 && !(cx.flags & 524288));
  if (!p.x_hi && p.B > 1 && p.SEG <= 128 && p.out_stride == 1 && p.out_off == 0 && cat_rows && !(cx.flags & 134217728)) {
    const int S = p.stride;
    const int reach = std::max(std::max(p.pad, S * (p.Lout - 1) - p.pad + (p.K - 1) * p.dil - (p.Lin - 1)), 0);
    const int Lg = std::max(p.Lout, (int)cdiv(p.Lin + reach, S));
    if ((int64_t)p.SEG * 100 > (int64_t)Lg * 115) {
      cat_w = S * Lg; cat_b = p.B; cat_l = p.Lin;
      p.catLg = Lg; p.catLout = p.Lout;
      p.Lout = p.B * Lg; p.Lin = p.B * S * Lg; p.B = 1; p.SEG = 1 << 20;
      CW = conv_dma_pick(p, cx);
    }
  }
  const int MT = 64, LT = 128 * CW;
  const int K = p.K, SEG = p.SEG > LT ? LT : p.SEG;
  const DmaGeom g = conv_dma_geom(p, CW);
  p.SEG = SEG;
  if (!g.ok) return TTTS_OK;
  p.NBS = (cx.flags & 268435456) ? 1 : g.nbs;     // (flag 268435456: one channel block per stage, for comparison)
  size_t dma_smem = (size_t)2 * p.NBS * 2 * (g.nxc + g.nwc) * 512 * sizeof(bf16);
  if (p.rowS > 0) {      // the phase-merged epilogue transposes the 64 x LT tile through LDS (room for it, two workgroups per CU still fit)
    dma_smem = std::max(dma_smem, (size_t)MT * (LT + 1) * sizeof(float));
    p.Lreal = (int)dma_smem;
  }
  const int nblk = (p.N + 15) / 16, AP = K * 16 + 8;
  p.Mpad = (int)(cdiv(p.M, MT) * MT);
  if (!p.x_hi) { p.PADL = g.padl; p.Lp = (int)cdiv(g.padl + p.Lin + g.padr, 8) * 8; }
  else if (p.PADL < g.padl || p.Lp - p.PADL - p.Lin < g.padr) return fail(TTTS_EINVAL, "conv1d: shared input split lacks padding");
  const int64_t welems = (int64_t)nblk * p.Mpad * AP + 512;                 // +512: the clamped tail chunk stays inside
  const int64_t xel = (int64_t)p.B * nblk * 2 * p.Lp * 8;
  if (2 * (welems + xel) * (int64_t)sizeof(bf16) > cx.ws_bytes) return TTTS_OK;
  bf16* xhi = static_cast<bf16*>(cx.ws);       // scratch layout: [x hi][x lo][w hi][w lo] (the input split comes first so
  bf16* xlo = xhi + xel;                       // that the phases of a strided data gradient can share it)
  bf16* hi = xlo + xel;
  bf16* lo = hi + welems;
  bool need_split = true;
  wsplit_lookup(p, cx, nblk, AP, welems, stream, &hi, &lo, &need_split);      // (persistent copies when the weights are cached)
  p.a_hi = hi; p.a_lo = lo;
  const int f16 = (cx.flags & TTTS_CONV_F16X1) ? 1 : 0;
  if (need_split)
    conv_weight_split_kernel<<<(int)std::min<int64_t>(cdiv(welems, 256), 2048), 256, 0, stream>>>(p.w, hi, lo, p.M, p.N, p.Mpad, nblk, K, p.Kmem,
                                                                                               p.transposed, p.tap_off, p.tap_stride, AP, p.rowS, p.rpad, p.pad, f16);
  if (!p.x_hi) {   // (polyphase data gradients share one split of dy across their phase launches)
    conv_input_split_kernel<<<(int)std::min<int64_t>(cdiv(xel / 16, 256), 8192), 256, 0, stream>>>(p.x, xhi, xlo, p.B, p.N, p.Lin, nblk, p.in_slope,
                                                                                                   p.Lp, p.PADL, cat_w, cat_b, cat_l, p.rowS < 0 ? -p.rowS : 0, p.Lreal, f16);
    p.x_hi = xhi; p.x_lo = xlo;
  }
  dim3 grid((unsigned)cdiv(p.Lout, SEG == LT ? LT : SEG), (unsigned)cdiv(p.M, MT), (unsigned)cdiv(p.B, LT / SEG));
  int rc = TTTS_OK;
#define TTTS_DMA1(CW_, KT_, F_)                                                                                  \
  {                                                                                                               \
    static OnceFlag attr_;                                                                                    \
    rc = set_attr_once(reinterpret_cast<const void*>(conv1d_bf16x3_dma_kernel<CW_, KT_, F_>), attr_);             \
    if (rc) return rc;                                                                                            \
    conv1d_bf16x3_dma_kernel<CW_, KT_, F_><<<grid, 256, dma_smem, stream>>>(p);                                    \
  }
#define TTTS_DMA(CW_, KT_) if (f16) TTTS_DMA1(CW_, KT_, true) else TTTS_DMA1(CW_, KT_, false)
#define TTTS_DMA_K(CW_)                                                                                          \
  switch ((cx.flags & 33554432) ? 0 : K) {   /* flag 33554432: runtime tap loop everywhere */                    \
    case 1: TTTS_DMA(CW_, 1) break; case 2: TTTS_DMA(CW_, 2) break; case 3: TTTS_DMA(CW_, 3) break; case 5: TTTS_DMA(CW_, 5) break; \
    case 7: TTTS_DMA(CW_, 7) break; case 11: TTTS_DMA(CW_, 11) break; default: TTTS_DMA(CW_, 0) break;           \
  }
  if (CW == 1) TTTS_DMA_K(1) else TTTS_DMA_K(2)
#undef TTTS_DMA_K
#undef TTTS_DMA
#undef TTTS_DMA1
  *handled = true;
  return check_launch("conv1d_bf16x3_dma");
}

template <int WCO>
static int conv1d_bf16x3_launch_t(ConvMfmaParams p, const ConvCtx& cx, hipStream_t stream, bool* handled) {
  constexpr int MT = 32 * WCO, LT = 64 * (4 / WCO);
  // pre-split + DMA kernel when the input is re-read by >= 3 output-channel tiles (measured: a full extra pass over x costs
  // more than on-the-fly splitting for the 16..64-channel long-row layers, and wins from 192 channels up); flag 32768:
  // from one tile, flag 65536: never (tools/conv_bench.py)
  // (flag 512, experiment: 1 x 1 convolutions of up to 256 output channels on the on-the-fly kernel -- one launch instead of pre-pass + DMA kernel)
  const bool k1_direct = false;   // (round 4's flag-512 experiment -- 1 x 1 layers on the on-the-fly kernel -- lost; the flag now switches conv1x1_b3_kernel off)
  if (WCO == 2 && !(cx.flags & 65536) && !k1_direct && (p.x_hi != nullptr || p.rowS < 0 || cdiv(p.M, MT) >= ((cx.flags & 32768) ? 1 : 3))) {
    int rc = conv1d_bf16x3_dma_launch(p, cx, stream, handled, conv_dma_pick(p, cx));
    if (rc || *handled) return rc;
    if (p.x_hi) return fail(TTTS_EUNSUPPORTED, "conv1d: shared input split without the DMA kernel");
  }
  if (p.rowS < 0) return TTTS_OK;          // (the de-interleaved input only exists as the DMA kernel's pre-split copy)
  const int K = p.K, SEG = p.SEG > LT ? LT : p.SEG;
  p.SEG = SEG;
  const int lin_t = (LT / SEG) * ((SEG - 1) * p.stride + (K - 1) * p.dil + 1);
  const size_t smem = ((size_t)2 * lin_t * 16 + (size_t)2 * MT * (K * 16 + 8) + 16) * sizeof(bf16);   // (+16: the dump slots)
  if (smem > 100 * 1024) return TTTS_OK;   // (long strips of the stride-8 / 10 resampling layers: one workgroup per CU still beats the direct kernel 5x)
  const int nblk = (p.N + 15) / 16;
  p.Mpad = (int)(cdiv(p.M, MT) * MT);
  const int64_t elems = (int64_t)nblk * p.Mpad * K * 16;
  if (2 * elems * (int64_t)sizeof(bf16) > cx.ws_bytes) return TTTS_OK;
  bf16* hi = static_cast<bf16*>(cx.ws);
  bf16* lo = hi + elems;
  bool need_split = true;
  wsplit_lookup(p, cx, nblk, K * 16, elems, stream, &hi, &lo, &need_split);
  p.a_hi = hi; p.a_lo = lo;
  const int f16 = (cx.flags & TTTS_CONV_F16X1) ? 1 : 0;
  if (need_split)
    conv_weight_split_kernel<<<(int)std::min<int64_t>(cdiv(elems, 256), 2048), 256, 0, stream>>>(p.w, hi, lo, p.M, p.N, p.Mpad, nblk, K, p.Kmem,
                                                                                              p.transposed, p.tap_off, p.tap_stride, K * 16, p.rowS, p.rpad, p.pad, f16);
  dim3 grid((unsigned)cdiv(p.Lout, SEG == LT ? LT : SEG), (unsigned)cdiv(p.M, MT), (unsigned)cdiv(p.B, LT / SEG));
  int rc = TTTS_OK;
#define TTTS_V1F(KT_, F_)                                                                            \
  {                                                                                                   \
    static OnceFlag attr_;                                                                        \
    rc = set_attr_once(reinterpret_cast<const void*>(conv1d_bf16x3_kernel<WCO, KT_, F_>), attr_);     \
    if (rc) return rc;                                                                                \
    conv1d_bf16x3_kernel<WCO, KT_, F_><<<grid, 256, smem, stream>>>(p);                              \
  }
#define TTTS_V1(KT_) if (f16) TTTS_V1F(KT_, true) else TTTS_V1F(KT_, false)
  switch ((cx.flags & 33554432) ? 0 : K) {
    case 1: TTTS_V1(1) break; case 2: TTTS_V1(2) break; case 3: TTTS_V1(3) break; case 5: TTTS_V1(5) break;   // (2: the phase-merged strided layers)
    case 7: TTTS_V1(7) break; case 11: TTTS_V1(11) break; default: TTTS_V1(0) break;
  }
#undef TTTS_V1
#undef TTTS_V1F
  *handled = true;
  return check_launch("conv1d_bf16x3");
}

static int conv1d_mfma_launch(ConvMfmaParams p, const ConvCtx& cx, hipStream_t stream, bool* handled) {
  *handled = false;
  // batch folding for short rows (see SEG); requested segment length, clipped to the tile by the launcher
  p.SEG = conv_seg_request(p.Lout, p.B);
  if (cx.ws && p.N >= 16 && !(cx.flags & 4096)) {
    if (conv1x1_b3_fits(p, cx)) {
      int rc = conv1x1_b3_launch(p, cx, stream, handled);
      if (rc || *handled) return rc;
    }
    if (p.rowS < 0) return conv1d_bf16x3_launch_t<2>(p, cx, stream, handled);
    int rc = p.M <= 32 ? conv1d_bf16x3_launch_t<1>(p, cx, stream, handled) : conv1d_bf16x3_launch_t<2>(p, cx, stream, handled);
    if (rc || *handled) return rc;
    if (p.M <= 32) {   // the 32 x 256 tile's input strip did not fit LDS (large stride): 64 x 128, half of its rows idle
      rc = conv1d_bf16x3_launch_t<2>(p, cx, stream, handled);
      if (rc || *handled) return rc;
    }
  }
  if (p.rowS != 0 || p.y2) return TTTS_OK;   // (the phase-merged / dual-destination forms exist in the split-bf16 kernels only: the caller falls back)
  // tile choice: always the largest tile.  Measured (tools/conv_bench.py, B = 32): the smaller tiles <2,2>, <1,2>, <1,1> --
  // meant to put more workgroups on a CU for the 192-channel x 256-frame layers -- are 2-2.5x SLOWER there (WN in_layer
  // dgrad 346 -> 738 us): the kernel is bound by the global -> LDS staging work per MFMA, so less reuse per staged slab
  // costs more than the extra overlap gains.  They stay instantiable for experiments (flag 2048).
  if (cx.flags & 2048) {
    auto wgs = [&](int MT, int LT) {
      const int seg = p.SEG > LT ? LT : p.SEG;
      return cdiv(p.Lout, seg == LT ? LT : seg) * cdiv(p.M, MT) * cdiv(p.B, LT / seg);
    };
    if (p.M > 32 && wgs(64, 128) < 512) {
      if (wgs(64, 64) >= 512) return conv1d_mfma_launch_t<2, 2>(p, cx, stream, handled);
      if (wgs(32, 128) >= 512) return conv1d_mfma_launch_t<1, 2>(p, cx, stream, handled);
      return conv1d_mfma_launch_t<1, 1>(p, cx, stream, handled);
    }
  }
  if (p.M <= 32) return conv1d_mfma_launch_t<1, 4>(p, cx, stream, handled);
  return conv1d_mfma_launch_t<2, 4>(p, cx, stream, handled);
}

// Returns TTTS_OK and sets *handled when the MFMA path took the launch.
// dual-destination forward (stride 1): see ConvMfmaParams::y2
int conv1d_mfma_dual_try(const float* x, const float* w, const float* bias, const float* resid, const float* omask, float* y, float* y2,
                         int B, int M, int M1, int N, int Lin, int Lout, int K, int pad, int dil, float in_slope, int accumulate2,
                         const ConvCtx& cx, hipStream_t stream, bool* handled) {
  *handled = false;
  if (N < 16 || K > 16 || !cx.ws || (cx.flags & 4096)) return TTTS_OK;
  ConvMfmaParams p{x, w, bias, nullptr, resid, omask, nullptr, y, B, M, N, Lin, Lout, K, 1, pad, dil, 0, 0,
                   K, 0, 1, 1, 0, Lout, 0, in_slope, 1.f, 0, 1.f, 1.f, 0, nullptr, nullptr, 0, nullptr, nullptr};
  p.y2 = y2; p.M1 = M1; p.acc2 = accumulate2;
  return conv1d_mfma_launch(p, cx, stream, handled);
}

int conv1d_mfma_try(const float* x, const float* w, const float* bias, const float* bbias, const float* resid,
                    const float* gate, const float* omask, float* y, int B, int M, int N, int Lin, int Lout, int K,
                    int stride, int pad, int dil, int transposed, float in_slope, float gate_slope, int out_act,
                    float out_slope, float out_scale, int accumulate, const ConvCtx& cx, hipStream_t stream, bool* handled) {
  *handled = false;
  if (N < 8 || K > 16) return TTTS_OK;       // thin inputs / long taps stay on the direct kernels (M = 1 heads are fine:
                                             // a 32-row tile with one live row beats looping 1024 channels on the vector ALUs)
  // Phase-merged strided forward (round 4; see ConvMfmaParams::rowS < 0): the k16 stride-8 / stride-10 resampling layers staged a
  // 2576-position input strip per 256 outputs and ran at 5-25 TF/s; as a stride-1 convolution over Cin x stride de-interleaved
  // channels they take the DMA kernel: 668 -> 322 us (16 -> 32 k16 s10), 772 -> 150 us (256 -> 512 k16 s10); at stride 2 the plain
  // strided staging is still faster (32.6 vs 41.7 us), hence stride >= 4.  Flag 4194304: off (A/B switch, shared with the merged
  // data gradient).
  if ((stride >= 4 || (stride == 3 && (cx.flags & 2))) && dil == 1 && !transposed && K >= stride && cx.ws && N >= 2 && (int64_t)N * stride >= 16 && (int64_t)N * stride <= 8192 &&
      !(cx.flags & (4096 | 4194304 | 65536))) {
    const int tmin = pad > 0 ? -(int)cdiv(pad, stride) : 0;              // floor((0 - pad) / stride)
    const int tmax = (K - 1 - pad) >= 0 ? (K - 1 - pad) / stride : -(int)cdiv(pad - (K - 1), stride);
    const int Kv = tmax - tmin + 1;
    ConvMfmaParams q{x, w, bias, bbias, resid, omask, gate, y, B, M, N * stride, (int)cdiv(Lin, stride), Lout, Kv, 1, -tmin, 1, 0, 0,
                     K, 0, 1, 1, 0, Lout, 0, in_slope, gate_slope, out_act, out_slope, out_scale, accumulate, nullptr, nullptr, 0, nullptr, nullptr};
    q.rowS = -stride; q.rpad = pad; q.Lreal = Lin;
    bool h = false;
    int rc = conv1d_mfma_launch(q, cx, stream, &h);
    if (rc) return rc;
    if (h) { *handled = true; return TTTS_OK; }
  }
  ConvMfmaParams p{x, w, bias, bbias, resid, omask, gate, y, B, M, N, Lin, Lout, K, stride, pad, dil, transposed, 0,
                   K, 0, 1, 1, 0, Lout, 0, in_slope, gate_slope, out_act, out_slope, out_scale, accumulate, nullptr, nullptr, 0, nullptr, nullptr};
  return conv1d_mfma_launch(p, cx, stream, handled);
}

// Data gradient of a stride-s convolution (== forward of a ConvTranspose1d), dilation 1, as s stride-1 sub-convolutions:
// output phase phi = (j + pad) mod s only sees the taps k = phi + s q, so
//   dx[s t' + phi - pad] = sum_{co, q} w[co][ci][phi + s q] * dy[co][t' - q].
int conv1d_dgrad_strided_mfma_try(const float* dy, const float* w, const float* bias, const float* resid,
                                  const float* gate, const float* omask, float* dx, int B, int Cin, int Lin, int Cout,
                                  int Lout, int K, int stride, int pad, float in_slope, float gate_slope, float out_scale,
                                  int accumulate, const ConvCtx& cx, hipStream_t stream, bool* handled) {
  *handled = false;
  if (Cout < 8 || K > 16 * stride || K < stride) return TTTS_OK;
  // Phase-merged form (round 4): ONE stride-1 convolution whose rows are (channel, phase) pairs.  The per-phase launches below run
  // with M = Cin rows each -- 16 of a 64-row tile for the encoders' 16 -> 32 k16 stride-10 layer: 1600 us for 0.3 GB of output --
  // re-read dy `stride` times and reduce over one or two taps; merged, the launch has Cin x stride rows, reads dy once and reduces
  // over Cout x (ceil(K / stride) + 1) taps.  Split-bf16 kernels only: flag 4096 (exact fp32) and flag 4194304 (A/B switch) keep the
  // per-phase launches.
  if (cx.ws && Cout >= 16 && (int64_t)Cin * stride <= 8192 && !(cx.flags & 4096) && !(cx.flags & 4194304)) {
    int emax = (stride - 1 + pad) / stride, emin = emax;
    for (int r = 0; r < stride; ++r) {
      const int qr = (r + pad) / stride, rem = (r + pad) % stride;
      if (rem >= K) continue;                              // (cannot happen with K >= stride)
      emin = std::min(emin, qr - (K - 1 - rem) / stride);
    }
    const int Kv = emax - emin + 1, T = (int)cdiv(Lin, stride);
    ConvMfmaParams p{dy, w, bias, nullptr, resid, omask, gate, dx, B, Cin * stride, Cout, Lout, T, Kv, 1, -emin, 1, 1, 0,
                     K, 0, 1, 1, 0, Lin, 0, in_slope, gate_slope, 0, 1.f, out_scale, accumulate, nullptr, nullptr, 0, nullptr, nullptr, 0, 0};
    p.rowS = stride; p.rpad = pad;
    if (-emin >= 0) {
      bool h = false;
      int rc = conv1d_mfma_launch(p, cx, stream, &h);
      if (rc) return rc;
      if (h) { *handled = true; return TTTS_OK; }
    }
  }
  // split-bf16 path: split dy ONCE for all phases (same place conv1d_bf16x3_dma_launch would put it), padded for the
  // phase that reaches furthest; only if every phase fits the DMA kernel
  const bf16* xs_hi = nullptr;
  const bf16* xs_lo = nullptr;
  int sh_Lp = 0, sh_padl = 0;
  if (cx.ws && Cout >= 16 && Cin >= 192 && !(cx.flags & (4096 | 65536))) {
    bool all_ok = true;
    int padl = 0, padr = 0;
    for (int phi = 0; phi < stride && all_ok; ++phi) {
      const int Kp = (K - phi + stride - 1) / stride;
      const int tmin = phi >= pad ? 0 : (pad - phi + stride - 1) / stride;
      const int off = stride * tmin + phi - pad;
      if (off >= Lin) continue;
      ConvMfmaParams q{};
      q.B = B; q.M = Cin; q.N = Cout; q.Lin = Lout; q.Lout = (Lin - 1 - off) / stride + 1; q.K = Kp; q.stride = 1;
      q.pad = (Kp - 1) - tmin; q.dil = 1; q.SEG = conv_seg_request(q.Lout, B);
      const DmaGeom g = conv_dma_geom(q, conv_dma_pick(q, cx));
      all_ok = g.ok;
      padl = std::max(padl, g.padl); padr = std::max(padr, g.padr);
    }
    const int nblk = (Cout + 15) / 16;
    const int Lp = (int)cdiv(padl + Lout + padr, 8) * 8;
    const int64_t xel = (int64_t)B * nblk * 2 * Lp * 8;
    if (all_ok && 2 * xel * (int64_t)sizeof(bf16) + (32 << 20) <= cx.ws_bytes) {
      bf16* xh = static_cast<bf16*>(cx.ws);
      conv_input_split_kernel<<<(int)std::min<int64_t>(cdiv(xel / 16, 256), 8192), 256, 0, stream>>>(dy, xh, xh + xel, B, Cout, Lout, nblk, in_slope, Lp, padl, 0, 0, 0, 0, 0,
                                                                                                     (cx.flags & TTTS_CONV_F16X1) ? 1 : 0);
      xs_hi = xh; xs_lo = xh + xel; sh_Lp = Lp; sh_padl = padl;
    }
  }
  for (int phi = 0; phi < stride; ++phi) {
    const int Kp = (K - phi + stride - 1) / stride;                       // taps of this phase (>= 1 since K >= stride)
    const int tmin = phi >= pad ? 0 : (pad - phi + stride - 1) / stride;  // first t' with a non-negative output position
    const int off = stride * tmin + phi - pad;
    if (off >= Lin) continue;
    const int T = (Lin - 1 - off) / stride + 1;
    ConvMfmaParams p{dy, w, bias, nullptr, resid, omask, gate, dx, B, Cin, Cout, Lout, T, Kp, 1, (Kp - 1) - tmin, 1, 1, 0,
                     K, phi, stride, stride, off, Lin, 0, in_slope, gate_slope, 0, 1.f, out_scale, accumulate, nullptr, nullptr, 0, xs_hi, xs_lo, sh_Lp, sh_padl};
    bool h = false;
    int rc = conv1d_mfma_launch(p, cx, stream, &h);
    if (rc) return rc;
    if (!h) return phi == 0 ? TTTS_OK : fail(TTTS_EUNSUPPORTED, "conv1d_dgrad: polyphase tile does not fit LDS");
  }
  *handled = true;
  return TTTS_OK;
}

// ---- weight gradient on the bf16 matrix cores (split-bf16) ----------------------------------------------------------------
// dw[co][ci][k] += sum_{b,l} dy[b][co][l] * x[b][ci][l*stride - pad + k*dil].  One workgroup owns ONE tap k, so that both
// operands become plain row segments: dy[co][l0 .. l0+63] and, after a pre-pass that splits x into hi / lo bf16 and
// de-interleaves it by phase (position mod stride), x'[ci][phase(k)][l0 + d(k) .. +63].  The pre-passes also apply the fused
// leaky-relus and zero-pad the rows, so the main loop is: 16-byte loads -> LDS -> 3 MFMAs per k-step, no bounds tests.
//   dy split:  DY[b][co][Lq]           Lq = roundup64(Lout), zero beyond Lout
//   x split:   XS[par][b][ci][r][Li]   element i of phase r of parity copy `par` holds x[(i + par)*stride + r - PL]
//              (PL = roundup_stride(pad)); the odd copy lets every row segment start on a 4-byte boundary
// (2-D grids: blockIdx.y walks rows, a thread converts two adjacent elements -- the flat-index form spent most of its time in
// 64-bit divisions)
// (round 5: EIGHT adjacent elements per thread and 16-byte stores -- two per thread made 20 000 workgroups of 2 KB each for one
// layer's dy: 26 us for 31 MB, dispatch-bound; rows shorter than 256 x 8 elements are walked several at a time by one workgroup)
__device__ __forceinline__ void wgrad_split_store8(const float (&v)[8], bf16* __restrict__ hi, bf16* __restrict__ lo) {
  bf16x8 h, w;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    h[e] = (bf16)v[e];
    w[e] = (bf16)(v[e] - (float)h[e]);
  }
  *reinterpret_cast<bf16x8*>(hi) = h;
  *reinterpret_cast<bf16x8*>(lo) = w;
}
// rows-per-workgroup geometry shared by the launcher and the bodies: items = 16-byte groups per row
__host__ __device__ __forceinline__ int wgrad_split_rpb(int items) { return items >= 256 ? 1 : 256 / items; }

__device__ __forceinline__ void wgrad_split_dy_body(const float* __restrict__ dy, bf16* __restrict__ hi,
                                                    bf16* __restrict__ lo, int64_t rows, int Lout, int Lq, float slope,
                                                    float* __restrict__ db, int Cout, int bx, int by, int gy, float* sh) {
  // db != NULL (slope == 1): the layer's bias gradient, db[row % Cout] += sum of the row -- this pass reads all of dy anyway.
  // gy is a multiple of Cout then, so every row a workgroup walks (by, by + gy, by + 2 gy, ...) belongs to the same channel.
  const int items = Lq >> 3, rpb = wgrad_split_rpb(items);
  const int sr = rpb == 1 ? 0 : (int)threadIdx.x / items;
  const int it = rpb == 1 ? bx * 256 + (int)threadIdx.x : (int)threadIdx.x - sr * items;
  float bs = 0.f;
  if (it < items && sr < rpb) {
    const int l = it * 8;
    for (int64_t r = by + (int64_t)sr * gy; r < rows; r += (int64_t)gy * rpb) {
      const float* src = dy + r * Lout + l;
      float v[8];
      if (l + 8 <= Lout && (reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(src), c = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = c[e]; }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = l + e < Lout ? src[e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[e] = lrelu_f(v[e], slope); bs += v[e]; }
      wgrad_split_store8(v, hi + r * Lq + l, lo + r * Lq + l);
    }
  }
  if (db) {
    bs = wave_sum(bs);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = bs;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(db + by % Cout, (sh[0] + sh[1]) + (sh[2] + sh[3]));
  }
}
// bz = parity copy * stride + phase
__device__ __forceinline__ void wgrad_split_x_body(const float* __restrict__ x, bf16* __restrict__ hi,
                                                   bf16* __restrict__ lo, int64_t rows, int Lin, int stride, int Li,
                                                   int PL, float slope, int bx, int by, int bz, int gy) {
  const int items = Li >> 3, rpb = wgrad_split_rpb(items);
  const int sr = rpb == 1 ? 0 : (int)threadIdx.x / items;
  const int it = rpb == 1 ? bx * 256 + (int)threadIdx.x : (int)threadIdx.x - sr * items;
  if (it >= items || sr >= rpb) return;
  const int ii = it * 8;
  const int par = bz / stride, r = bz % stride;
  const int64_t per = rows * stride * Li;
  const int64_t q0 = (int64_t)(ii + par) * stride + r - PL;
  for (int64_t row = by + (int64_t)sr * gy; row < rows; row += (int64_t)gy * rpb) {
    const float* src = x + row * Lin;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int64_t q = q0 + (int64_t)e * stride;
      v[e] = (q >= 0 && q < Lin) ? lrelu_f(src[q], slope) : 0.f;
    }
    const int64_t o = par * per + (row * stride + r) * Li + ii;
    wgrad_split_store8(v, hi + o, lo + o);
  }
}
// Both operand splits of a weight gradient in ONE launch (they were two ~5 us launches per layer): workgroups [0, nb_dy) split
// dy, the rest x; each role decodes its own (x, y, z) block coordinates from the flat index.
struct WgradSplitPair {
  const float* dy; bf16* dyh; bf16* dyl; int64_t rows_dy; int Lout, Lq; float dy_slope; float* db; int Cout;
  int nbx_dy, gy_dy, nb_dy;
  const float* x; bf16* xh; bf16* xl; int64_t rows_x; int Lin, stride, Li, PL; float x_slope;
  int nbx_x, gy_x;
};
__global__ __launch_bounds__(256) void wgrad_split_pair_kernel(WgradSplitPair p) {
  __shared__ float sh[4];
  const int id = blockIdx.x;
  if (id < p.nb_dy) {
    wgrad_split_dy_body(p.dy, p.dyh, p.dyl, p.rows_dy, p.Lout, p.Lq, p.dy_slope, p.db, p.Cout, id % p.nbx_dy, id / p.nbx_dy, p.gy_dy, sh);
  } else {
    const int j = id - p.nb_dy, bx = j % p.nbx_x, t = j / p.nbx_x;
    wgrad_split_x_body(p.x, p.xh, p.xl, p.rows_x, p.Lin, p.stride, p.Li, p.PL, p.x_slope, bx, t % p.gy_x, t / p.gy_x, p.gy_x);
  }
}
// Short rows (DiscriminatorP: 23..127 positions, hundreds of rows): the batch elements of a channel are laid end to end as ONE
// virtual row, Lg = Lout + (K-1)*dil positions apart (zeros in between, so taps never reach the next element):
//   dy'[co][b*Lg + j] = dy[b][co][j],   x'[ci][b*Lg + j] = x[b][ci][j - pad]
// and sum_v dy'[v] x'[v + k*dil] is exactly the batch-summed weight gradient.  The all-taps kernel then runs with B = 1 and
// 64-position chunks that are ~full instead of one mostly-padding chunk per row (L = 23: 36 % -> 85 % useful positions).
__device__ __forceinline__ void wgrad_split_cat_body(const float* __restrict__ src, bf16* __restrict__ hi, bf16* __restrict__ lo,
                                                     int B, int C, int L, int Lg, int shift, int Lrow, float slope,
                                                     float* __restrict__ db, int S, int bx, int by, float* sh) {
  // by = channel * S + phase; element v of the virtual row: b = v / Lg, j = v % Lg, source position j*S + phase - shift
  const int c = by / S, ph = by % S, v0 = (bx * 256 + threadIdx.x) * 2;
  float bs = 0.f;
  if (v0 < Lrow) {
    float val[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int v = v0 + e, b = v / Lg, j = (v - b * Lg) * S + ph - shift;
      const bool ok = b < B && j >= 0 && j < L;
      const float t = src[((int64_t)min(b, B - 1) * C + c) * L + min(max(j, 0), L - 1)];
      val[e] = ok ? lrelu_f(t, slope) : 0.f;
    }
    bs = val[0] + val[1];
    bf16x2 h, w;
    h[0] = (bf16)val[0]; h[1] = (bf16)val[1];
    w[0] = (bf16)(val[0] - (float)h[0]); w[1] = (bf16)(val[1] - (float)h[1]);
    *reinterpret_cast<bf16x2*>(hi + (int64_t)by * Lrow + v0) = h;
    *reinterpret_cast<bf16x2*>(lo + (int64_t)by * Lrow + v0) = w;
  }
  if (db) {
    bs = wave_sum(bs);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = bs;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(db + c, (sh[0] + sh[1]) + (sh[2] + sh[3]));
  }
}

struct WgradCatSide { const float* src; bf16* hi; bf16* lo; int C, L, shift, Lrow; float slope; float* db; int S, nbx, nb; };
__global__ __launch_bounds__(256) void wgrad_split_cat_pair_kernel(WgradCatSide a, WgradCatSide b, int B, int Lg) {
  __shared__ float sh[4];          // (both sides of a weight gradient in one launch: workgroups [0, a.nb) lay out dy, the rest x)
  const int id = blockIdx.x;
  if (id < a.nb) wgrad_split_cat_body(a.src, a.hi, a.lo, B, a.C, a.L, Lg, a.shift, a.Lrow, a.slope, a.db, a.S, id % a.nbx, id / a.nbx, sh);
  else wgrad_split_cat_body(b.src, b.hi, b.lo, B, b.C, b.L, Lg, b.shift, b.Lrow, b.slope, b.db, b.S, (id - a.nb) % b.nbx, (id - a.nb) / b.nbx, sh);
}
static void launch_wgrad_cat_pair(const float* dy, bf16* dyh, bf16* dyl, int Cout, int Lout, int Lq, float dy_slope, float* db,
                                  const float* x, bf16* xh, bf16* xl, int Cin, int Lin, int shift, int Li, float x_slope, int S, int B,
                                  int Lg, hipStream_t stream) {
  WgradCatSide a{dy, dyh, dyl, Cout, Lout, 0, Lq, dy_slope, db, 1, (int)cdiv(Lq / 2, 256), 0};
  a.nb = a.nbx * Cout;
  WgradCatSide b{x, xh, xl, Cin, Lin, shift, Li, x_slope, nullptr, S, (int)cdiv(Li / 2, 256), 0};
  b.nb = b.nbx * Cin * S;
  wgrad_split_cat_pair_kernel<<<(unsigned)(a.nb + b.nb), 256, 0, stream>>>(a, b, B, Lg);
}

static void launch_wgrad_splits(const float* dy, const float* x, bf16* dyh, bf16* dyl, bf16* xh, bf16* xl, int64_t rows_dy, int Lout,
                                int Lq, float dy_slope, int64_t rows_x, int Lin, int stride, int Li, int PL, float x_slope, int npar,
                                float* db, int Cout, hipStream_t stream) {
  // grid.y walks rows; a workgroup takes rpb rows per trip (short rows), so cdiv(rows, rpb) workgroups cover a pass over the rows
  const int rpb_dy = wgrad_split_rpb(Lq / 8), rpb_x = wgrad_split_rpb(Li / 8);
  int64_t gy = std::min<int64_t>(cdiv(rows_dy, rpb_dy), 32768);
  if (db) gy = std::min<int64_t>(rows_dy, std::max<int64_t>(Cout, cdiv(gy, Cout) * Cout));   // rows r, r + gy, ... share a channel
  WgradSplitPair p;
  p.dy = dy; p.dyh = dyh; p.dyl = dyl; p.rows_dy = rows_dy; p.Lout = Lout; p.Lq = Lq; p.dy_slope = dy_slope; p.db = db; p.Cout = Cout;
  p.nbx_dy = (int)cdiv(Lq / 8, 256); p.gy_dy = (int)gy; p.nb_dy = p.nbx_dy * p.gy_dy;
  p.x = x; p.xh = xh; p.xl = xl; p.rows_x = rows_x; p.Lin = Lin; p.stride = stride; p.Li = Li; p.PL = PL; p.x_slope = x_slope;
  p.nbx_x = (int)cdiv(Li / 8, 256); p.gy_x = (int)std::min<int64_t>(cdiv(rows_x, rpb_x), 32768);
  const int64_t nb = (int64_t)p.nb_dy + (int64_t)p.nbx_x * p.gy_x * npar * stride;
  wgrad_split_pair_kernel<<<(unsigned)nb, 256, 0, stream>>>(p);
}

struct WgradB3Params {
  const bf16* dyh; const bf16* dyl; const bf16* xh; const bf16* xl; float* dw;
  int B, Cin, Cout, K, stride, dil, Lq, Li, PLmPad;   // PLmPad = PL - pad >= 0
  int64_t xpar;        // elements per parity copy of XS
  int chunks_per_block, nchunks, nlc;                 // chunks = (b, 64-position window) pairs; nlc = Lq / 64
  float* slab;         // all-taps kernel: per-split partial sums [split][k][co][ci] (plain coalesced stores; device-scope fp32
                       // atomics onto a 128x128x11 weight from 128 splits were the dominant cost), summed by wgrad_slab_reduce
};

// TILE = 64: workgroup tile 64 co x 64 ci, waves 2 x 2, each wave all four 16-position k-steps of a chunk
// TILE = 32: workgroup tile 32 co x 32 ci, the four waves split the k-steps of a chunk (partial sums meet in the atomics)
template <int TILE>
__global__ __launch_bounds__(256) void conv1d_wgrad_bf16x3_kernel(WgradB3Params p) {
  constexpr int PITCH = 72;                           // 64 positions + 8: 144-byte rows keep 16-byte alignment, spread banks
  __shared__ __attribute__((aligned(16))) bf16 sm[4 * TILE * PITCH];
  bf16* ah = sm; bf16* al = sm + TILE * PITCH; bf16* bh = sm + 2 * TILE * PITCH; bf16* bl = sm + 3 * TILE * PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, col = lane & 31;
  const int ci0 = blockIdx.x * TILE, co0 = blockIdx.y * TILE;
  const int k = blockIdx.z % p.K, split = blockIdx.z / p.K;
  const int c = k * p.dil + p.PLmPad, r = c % p.stride, d = c / p.stride;   // tap k reads phase r at l + d
  const int par = d & 1, dd = d - par;                                      // parity copy `par` starts at an even element
  const int wco = TILE == 64 ? (wave & 1) : 0, wci = TILE == 64 ? (wave >> 1) : 0;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const bf16* xh = p.xh + par * p.xpar;
  const bf16* xl = p.xl + par * p.xpar;
  for (int cc = 0; cc < p.chunks_per_block; ++cc) {
    const int chunk = split * p.chunks_per_block + cc;
    if (chunk >= p.nchunks) break;
    const int b = chunk / p.nlc, l0 = (chunk % p.nlc) * 64;
    __syncthreads();
    // TILE rows x 8 sixteen-byte pieces per operand half
    for (int i = tid; i < TILE * 8; i += 256) {
      const int row = i >> 3, pc = i & 7;
      bf16x8 vh = zero8(), vl = zero8(), wh = zero8(), wl = zero8();
      if (co0 + row < p.Cout) {
        const int64_t o = ((int64_t)b * p.Cout + co0 + row) * p.Lq + l0 + pc * 8;
        vh = *reinterpret_cast<const bf16x8*>(p.dyh + o);
        vl = *reinterpret_cast<const bf16x8*>(p.dyl + o);
      }
      if (ci0 + row < p.Cin) {
        const int64_t o = (((int64_t)b * p.Cin + ci0 + row) * p.stride + r) * p.Li + l0 + dd + pc * 8;   // even: 4-byte aligned
        __builtin_memcpy(&wh, __builtin_assume_aligned(xh + o, 4), 16);
        __builtin_memcpy(&wl, __builtin_assume_aligned(xl + o, 4), 16);
      }
      *reinterpret_cast<bf16x8*>(ah + row * PITCH + pc * 8) = vh;
      *reinterpret_cast<bf16x8*>(al + row * PITCH + pc * 8) = vl;
      *reinterpret_cast<bf16x8*>(bh + row * PITCH + pc * 8) = wh;
      *reinterpret_cast<bf16x8*>(bl + row * PITCH + pc * 8) = wl;
    }
    __syncthreads();
    const int arow = (wco * 32 + col) * PITCH + hh * 8, brow = (wci * 32 + col) * PITCH + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (TILE == 32 && ks != wave) continue;
      const bf16x8 a_h = *reinterpret_cast<const bf16x8*>(ah + arow + ks * 16);
      const bf16x8 a_l = *reinterpret_cast<const bf16x8*>(al + arow + ks * 16);
      const bf16x8 b_h = *reinterpret_cast<const bf16x8*>(bh + brow + ks * 16);
      const bf16x8 b_l = *reinterpret_cast<const bf16x8*>(bl + brow + ks * 16);
      acc = mfma32(a_l, b_h, acc);
      acc = mfma32(a_h, b_l, acc);
      acc = mfma32(a_h, b_h, acc);
    }
  }
  const int ci = ci0 + wci * 32 + col;
  if (ci < p.Cin) {
    // TILE 64: one wave owns an output element of its split -> plain slab stores; TILE 32: four waves share it -> atomics
    float* sl = (p.slab && TILE == 64) ? p.slab + ((int64_t)split * p.K + k) * p.Cout * p.Cin : nullptr;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = co0 + wco * 32 + acc_row(i, hh);
      if (co < p.Cout) {
        if (sl) sl[(int64_t)co * p.Cin + ci] = acc[i];
        else atomicAdd(p.dw + ((int64_t)co * p.Cin + ci) * p.K + k, acc[i]);
      }
    }
  }
}

// ---- split-bf16 weight gradient, ALL taps per workgroup (stride 1; the 3/7/11-tap dilated ResBlock1 convolutions) ---------
// Same pre-split operands as above, but the workgroup stages one x window of 64 + (K-1)*DIL positions and every tap reads it at
// its own offset k*DIL, so the operand traffic no longer grows with K.  An offset that is not a multiple of 8 elements is
// served by two aligned 16-byte LDS reads and a compile-time element shuffle (K and DIL are template parameters), K
// accumulators per wave (K = 11: 176 VGPRs, two waves per SIMD).  Per 64-position chunk a wave issues 12 K MFMAs against
// ~12 sixteen-byte global loads per thread: matrix-core bound.
typedef __attribute__((ext_vector_type(16))) __bf16 bf16x16;

template <int S>
__device__ __forceinline__ bf16x8 window8(bf16x8 lo, bf16x8 hi) {
  const bf16x16 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
  return __builtin_shufflevector(v, v, S, S + 1, S + 2, S + 3, S + 4, S + 5, S + 6, S + 7);
}

// taps [K0, K0 + KN) of a K-tap convolution (K = 11 runs as two launches of 6 + 5 taps: 11 accumulators would spill)
// Workgroup tile 64 co x 64 ci (TILE), waves 2 x 2, every wave all four 16-position k-steps of a chunk.  (A 32 x 32 tile with the
// k-steps split over the waves was tried for narrow layers and lost to the fused single-pass kernel further down.)
// Staging: LDS-DMA double buffer.  A stage is [dy hi][dy lo][x hi][x lo] with padded rows (PITCH / WP elements); the DMA
// destination is lane-linear (16-byte slot s = row * slots_per_row + piece), the SOURCE address is per lane, so the row padding
// costs one junk slot per row and nothing else; rows beyond Cout / Cin are clamped (their products are never stored).
// S > 1 (stride, DIL == 1): the input arrives de-interleaved by phase, XS[b][ci][r][i] = x[(i*S + r) - PL]; tap k reads phase
// (k + PO) % S at shift (k + PO) / S with PO = PL - pad, so a row of the LDS tile is S phase windows of WP elements.
template <int K, int DIL, int K0, int KN, int TILE, int S = 1, int PO = 0>
__global__ __launch_bounds__(256, 2) void conv1d_wgrad_bf16x3_taps_kernel(WgradB3Params p) {
  constexpr int PITCH = 72;
  constexpr int BASE = S > 1 ? 0 : (K0 * DIL) / 8 * 8;    // aligned start of the staged window (tap K0 begins at K0*DIL)
  constexpr int WIN = S > 1 ? 64 + (K0 + KN - 1 + PO) / S : 64 + (K0 + KN - 1) * DIL - BASE;    // window positions (per phase)
  constexpr int WP0 = (WIN + 7) / 8 * 8 + ((S > 1 && (WIN - 64) % 8 != 0) ? 0 : 8);   // (the second 16-byte piece of the last tap)
  constexpr int WP = S > 1 ? ((S * (WP0 / 8)) % 2 ? WP0 : WP0 + 8) : (WP0 | 8);   // row pitch (S * WP): an ODD number of 16-byte
                                                          // pieces (a 256-byte pitch would put all 32 rows of a fragment read on the same banks)
  constexpr int RP = S * WP;                              // elements per LDS row
  constexpr int AS = TILE * (PITCH / 8), BS = TILE * (RP / 8);            // 16-byte slots per dy / x array
  constexpr int AC = (AS + 63) / 64, BC = (BS + 63) / 64;                 // 64-slot DMA chunks
  constexpr int AI = (AC + 3) / 4, BI = (BC + 3) / 4;                     // ... per wave
  constexpr int STAGE_EL = 2 * (AC + BC) * 512;
  extern __shared__ __attribute__((aligned(16))) float cm_smem[];
  bf16* sm = reinterpret_cast<bf16*>(cm_smem);             // 2 stages
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, col = lane & 31;
  int bx_, by_, bz_;
  xcd_tile(bx_, by_, bz_);      // (the tiles of one split read the same positions of dy and x: one XCD walks them)
  const int ci0 = bx_ * TILE, co0 = by_ * TILE, split = bz_;
  const int wco = TILE == 64 ? (wave & 1) : 0, wci = TILE == 64 ? (wave >> 1) : 0;
  f32x16 acc[KN];
#pragma unroll
  for (int k = 0; k < KN; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  // this lane's DMA sources relative to (batch element 0, position 0)
  int aoff[AI], boff[BI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int sl = min((wave + 4 * i) * 64 + lane, AS - 1), row = sl / (PITCH / 8), pc = min(sl % (PITCH / 8), 7);
    aoff[i] = min(co0 + row, p.Cout - 1) * p.Lq + pc * 8;
  }
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int sl = min((wave + 4 * i) * 64 + lane, BS - 1), row = sl / (RP / 8), rem = sl % (RP / 8), ph = rem / (WP / 8), pc = rem % (WP / 8);
    boff[i] = (min(ci0 + row, p.Cin - 1) * S + ph) * p.Li + BASE + pc * 8;      // Li % 8 == 0, l0 % 64 == 0: 16-byte aligned, inside the row
  }
  const int64_t dyl_d = p.dyl - p.dyh, xl_d = p.xl - p.xh;
  auto issue = [&](int chunk, int buf) {
    const int b = chunk / p.nlc, l0 = (chunk % p.nlc) * 64;
    const bf16* ga = p.dyh + (int64_t)b * p.Cout * p.Lq + l0;
    const bf16* gb = p.xh + (int64_t)b * p.Cin * S * p.Li + l0;
    bf16* st = sm + buf * STAGE_EL;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int c = wave + 4 * i;
      if (c < AC) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga + aoff[i]),
                                         (__attribute__((address_space(3))) void*)(st + c * 512), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga + aoff[i] + dyl_d),
                                         (__attribute__((address_space(3))) void*)(st + (AC + c) * 512), 16, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int c = wave + 4 * i;
      if (c < BC) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + boff[i]),
                                         (__attribute__((address_space(3))) void*)(st + (2 * AC + c) * 512), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + boff[i] + xl_d),
                                         (__attribute__((address_space(3))) void*)(st + (2 * AC + BC + c) * 512), 16, 0, 0);
      }
    }
  };
  // stride 1: tap k reads x'[l + k*DIL + (PL - pad)] with PL == pad, parity copy 0 (element i holds x[i - pad])
  const int chunk0 = split * p.chunks_per_block;
  const int nmine = max(0, min(p.chunks_per_block, p.nchunks - chunk0));
  if (nmine > 0) issue(chunk0, 0);
  for (int cc = 0; cc < nmine; ++cc) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                       // stage cc landed; the other buffer is free
    if (cc + 1 < nmine) issue(chunk0 + cc + 1, (cc + 1) & 1);
    const bf16* ah = sm + (cc & 1) * STAGE_EL;
    const bf16* al = ah + AC * 512;
    const bf16* bh = al + AC * 512;
    const bf16* bl = bh + BC * 512;
    const bf16* arow_h = ah + (wco * 32 + col) * PITCH + hh * 8;
    const bf16* arow_l = al + (wco * 32 + col) * PITCH + hh * 8;
    const bf16* brow_h = bh + (wci * 32 + col) * RP + hh * 8;
    const bf16* brow_l = bl + (wci * 32 + col) * RP + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 a_h = *reinterpret_cast<const bf16x8*>(arow_h + ks * 16);
      const bf16x8 a_l = *reinterpret_cast<const bf16x8*>(arow_l + ks * 16);
#pragma unroll
      for (int k = 0; k < KN; ++k) {
        const int toff = S > 1 ? (K0 + k + PO) / S : (K0 + k) * DIL - BASE;   // tap offset inside the staged window (compile-time)
        const int off = ks * 16 + (toff / 8) * 8 + (S > 1 ? ((K0 + k + PO) % S) * WP : 0);   // its aligned part (+ phase window)
        const bf16x8 h0 = *reinterpret_cast<const bf16x8*>(brow_h + off), h1 = *reinterpret_cast<const bf16x8*>(brow_h + off + 8);
        const bf16x8 l0v = *reinterpret_cast<const bf16x8*>(brow_l + off), l1v = *reinterpret_cast<const bf16x8*>(brow_l + off + 8);
        bf16x8 b_h, b_l;
        switch (toff & 7) {                                  // compile-time after unrolling
          case 0: b_h = h0; b_l = l0v; break;
          case 1: b_h = window8<1>(h0, h1); b_l = window8<1>(l0v, l1v); break;
          case 2: b_h = window8<2>(h0, h1); b_l = window8<2>(l0v, l1v); break;
          case 3: b_h = window8<3>(h0, h1); b_l = window8<3>(l0v, l1v); break;
          case 4: b_h = window8<4>(h0, h1); b_l = window8<4>(l0v, l1v); break;
          case 5: b_h = window8<5>(h0, h1); b_l = window8<5>(l0v, l1v); break;
          case 6: b_h = window8<6>(h0, h1); b_l = window8<6>(l0v, l1v); break;
          default: b_h = window8<7>(h0, h1); b_l = window8<7>(l0v, l1v); break;
        }
        acc[k] = mfma32(a_l, b_h, acc[k]);
        acc[k] = mfma32(a_h, b_l, acc[k]);
        acc[k] = mfma32(a_h, b_h, acc[k]);
      }
    }
  }
  const int ci = ci0 + wci * 32 + col;
  if (ci < p.Cin) {
    float* sl = p.slab + (int64_t)split * K * p.Cout * p.Cin;
#pragma unroll
    for (int k = 0; k < KN; ++k)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = co0 + wco * 32 + acc_row(i, hh);
        if (co < p.Cout) sl[((int64_t)(K0 + k) * p.Cout + co) * p.Cin + ci] = acc[k][i];
      }
  }
}

// ---- narrow layers (16..96 channels, long rows): ONE pass over the fp32 operands ------------------------------------------
// The pre-split kernels above move every operand byte four times (fp32 read, bf16 hi/lo write, hi/lo read -- twice for K = 11);
// with <= 96 channels there is almost no reuse to pay for that and the layer is HBM-bound.  Here a workgroup owns a 32 co x 32
// ci tile and ALL K taps, reads the fp32 rows once, splits them into bf16 hi / lo on the way into LDS (two positions per
// thread -> 4-byte LDS stores), and prefetches the next chunk into registers while the matrix cores work.  Wave w owns the
// taps w, w + 4, w + 8 for all four 16-position k-steps of a chunk: <= 3 accumulators per wave (a wave holding all 11 taps
// needs 392 registers: one workgroup per CU, nothing to hide the loads behind) and no cross-wave reduction.
template <int K, int DIL, int W, int CH>
__device__ __forceinline__ void fused_taps_compute(const bf16* ah, const bf16* al, const bf16* bh, const bf16* bl, int col, int hh,
                                                   f32x16 (&acc)[3]) {
  constexpr int PITCH = CH + 8;
  constexpr int WIN = CH + (K - 1) * DIL, WP = ((WIN + 7) / 8 * 8 + 8) | 8;
  constexpr int NT = W < K ? (K - W + 3) / 4 : 0;           // taps of this wave
  const bf16* arow_h = ah + col * PITCH + hh * 8;
  const bf16* arow_l = al + col * PITCH + hh * 8;
  const bf16* brow_h = bh + col * WP + hh * 8;
  const bf16* brow_l = bl + col * WP + hh * 8;
#pragma unroll
  for (int ks = 0; ks < CH / 16; ++ks) {
    if (NT == 0) break;
    const bf16x8 a_h = *reinterpret_cast<const bf16x8*>(arow_h + ks * 16);
    const bf16x8 a_l = *reinterpret_cast<const bf16x8*>(arow_l + ks * 16);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int toff = (W + 4 * j) * DIL;                  // tap offset inside the window (compile-time after unrolling)
      const int off = ks * 16 + (toff / 8) * 8;
      const bf16x8 h0 = *reinterpret_cast<const bf16x8*>(brow_h + off), h1 = *reinterpret_cast<const bf16x8*>(brow_h + off + 8);
      const bf16x8 l0v = *reinterpret_cast<const bf16x8*>(brow_l + off), l1v = *reinterpret_cast<const bf16x8*>(brow_l + off + 8);
      bf16x8 b_h, b_l;
      switch (toff & 7) {
        case 0: b_h = h0; b_l = l0v; break;
        case 1: b_h = window8<1>(h0, h1); b_l = window8<1>(l0v, l1v); break;
        case 2: b_h = window8<2>(h0, h1); b_l = window8<2>(l0v, l1v); break;
        case 3: b_h = window8<3>(h0, h1); b_l = window8<3>(l0v, l1v); break;
        case 4: b_h = window8<4>(h0, h1); b_l = window8<4>(l0v, l1v); break;
        case 5: b_h = window8<5>(h0, h1); b_l = window8<5>(l0v, l1v); break;
        case 6: b_h = window8<6>(h0, h1); b_l = window8<6>(l0v, l1v); break;
        default: b_h = window8<7>(h0, h1); b_l = window8<7>(l0v, l1v); break;
      }
      acc[j] = mfma32(a_l, b_h, acc[j]);
      acc[j] = mfma32(a_h, b_l, acc[j]);
      acc[j] = mfma32(a_h, b_h, acc[j]);
    }
  }
}

// CH = positions per chunk (64 or 128: the loop is bound by the load -> split -> LDS -> barrier round trip per chunk, so long
// rows take 128)
// (Round 3 measured a variant that masks the prefetched values at the LDS store instead of at the load: 168.7 vs 165.9 ms per
// VQ-VAE-GAN step -- no gain, removed.)
template <int K, int DIL, int CH>
__global__ __launch_bounds__(256, 2) void conv1d_wgrad_fused_taps_kernel(WgradMfmaParams p) {
  constexpr int PITCH = CH + 8;
  constexpr int WIN = CH + (K - 1) * DIL;                 // staged window positions
  constexpr int WP = ((WIN + 7) / 8 * 8 + 8) | 8;         // row pitch (odd number of 16-byte pieces)
  constexpr int WPR = (WIN + 1) / 2;                      // position pairs per window row
  constexpr int NA = CH / 16, NB = (32 * WPR + 255) / 256;   // pairs per thread: dy tile (32 rows x CH/2 pairs), x window
  __shared__ __attribute__((aligned(16))) bf16 sm[2 * 32 * PITCH + 2 * 32 * WP];
  bf16* ah = sm; bf16* al = sm + 32 * PITCH; bf16* bh = sm + 2 * 32 * PITCH; bf16* bl = bh + 32 * WP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, col = lane & 31;
  int bx_, by_, bz_;
  xcd_tile(bx_, by_, bz_);
  const int ci0 = bx_ * 32, co0 = by_ * 32, split = bz_;
  const int nlc = (p.Lout + CH - 1) / CH, nchunks = p.B * nlc;
  f32x16 acc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  float ra[NA][2], rb[NB][2], bsum[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) bsum[i] = 0.f;
  // (unconditional loads from clamped addresses + a select: a guarded load compiles to an exec-masked branch with its own
  // s_waitcnt, which serialised the 48 loads of a chunk -- 11 us per chunk)
  // Requests only: the values are masked when they are deposited (mask_regs, one chunk later).  Round 6, ISA read (tools/isa_scan.py):
  // with the select `(rok && l < Lout) ? v0 : 0` right behind each request pair the compiler waited for every PAIR before issuing
  // the next in six instantiations (<7, 1, 128>, <7, 3, 128>, <3, *, 64>, <11, 1, 64>): 17 dependent round trips per chunk where one
  // was meant to fly under the MFMAs -- the RB1(128) k7 weight gradient spent ~50 of its 62 us in them.
  auto load_regs = [&](int chunk) {
    const int b = chunk / nlc, l0 = (chunk % nlc) * CH;
    // (<11, 1, 128> and <11, 5, 64> spill 8 / 9 VGPRs: the per-thread row / offset tables of the 17 request pairs are computed once
    // and held across the chunk loop.  Round 6 measured the alternative -- the thread index made opaque per chunk, so the tables are
    // recomputed (a division by the odd window pitch per pair): no spill, 238 VGPRs, and 157 us (all pairs) / 206 us (x window only)
    // against 147 us for the RB1(128) k11 weight gradient.  The spills are the cheaper form.)
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int q = tid + 256 * i, row = q / (CH / 2), l = l0 + (q % (CH / 2)) * 2;
      const float* src = p.dy + ((int64_t)b * p.Cout + min(co0 + row, p.Cout - 1)) * p.Lout;
      ra[i][0] = src[min(l, p.Lout - 1)];
      ra[i][1] = src[min(l + 1, p.Lout - 1)];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int q = tid + 256 * i, row = q / WPR, g = l0 - p.pad + (q - row * WPR) * 2;
      const float* src = p.x + ((int64_t)b * p.Cin + min(ci0 + min(row, 31), p.Cin - 1)) * p.Lin;
      rb[i][0] = src[min(max(g, 0), p.Lin - 1)];
      rb[i][1] = src[min(max(g + 1, 0), p.Lin - 1)];
    }
  };
  auto mask_regs = [&](int chunk) {                           // the selects of the chunk whose requests load_regs issued
    const int l0 = (chunk % nlc) * CH;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int q = tid + 256 * i, row = q / (CH / 2), l = l0 + (q % (CH / 2)) * 2;
      const bool rok = co0 + row < p.Cout;
      ra[i][0] = (rok && l < p.Lout) ? ra[i][0] : 0.f;
      ra[i][1] = (rok && l + 1 < p.Lout) ? ra[i][1] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int q = tid + 256 * i, row = q / WPR, g = l0 - p.pad + (q - row * WPR) * 2;
      const bool rok = row < 32 && ci0 + row < p.Cin;
      rb[i][0] = (rok && g >= 0 && g < p.Lin) ? rb[i][0] : 0.f;
      rb[i][1] = (rok && g + 1 >= 0 && g + 1 < p.Lin) ? rb[i][1] : 0.f;
    }
  };
  auto split2 = [](float v0, float v1, bf16x2& h, bf16x2& l) {
    h[0] = (bf16)v0; h[1] = (bf16)v1;
    l[0] = (bf16)(v0 - (float)h[0]); l[1] = (bf16)(v1 - (float)h[1]);
  };
  auto store_lds = [&](int chunk) {     // chunk: the one whose values load_regs left in ra / rb
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int q = tid + 256 * i, row = q / (CH / 2), pj = q % (CH / 2);
      bf16x2 h, l;
      bsum[i] += ra[i][0] + ra[i][1];
      split2(lrelu_f(ra[i][0], p.dy_slope), lrelu_f(ra[i][1], p.dy_slope), h, l);
      *reinterpret_cast<bf16x2*>(ah + row * PITCH + pj * 2) = h;
      *reinterpret_cast<bf16x2*>(al + row * PITCH + pj * 2) = l;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int q = tid + 256 * i, row = q / WPR, pj = q - row * WPR;
      if (row < 32) {
        bf16x2 h, l;
        split2(lrelu_f(rb[i][0], p.x_slope), lrelu_f(rb[i][1], p.x_slope), h, l);
        *reinterpret_cast<bf16x2*>(bh + row * WP + pj * 2) = h;
        *reinterpret_cast<bf16x2*>(bl + row * WP + pj * 2) = l;
      }
    }
  };
  const int chunk0 = split * p.chunks_per_block;
  const int nmine = max(0, min(p.chunks_per_block, nchunks - chunk0));
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  if (nmine > 0) load_regs(chunk0);
  for (int cc = 0; cc < nmine; ++cc) {
    __syncthreads();                                       // everyone is done reading the previous chunk
    mask_regs(chunk0 + cc);
    store_lds(chunk0 + cc);
    __syncthreads();
    if (cc + 1 < nmine) load_regs(chunk0 + cc + 1);        // in flight while the matrix cores run
    __builtin_amdgcn_sched_barrier(0);                     // (the requests stay in front of the MFMAs)
    if (wv == 0) fused_taps_compute<K, DIL, 0, CH>(ah, al, bh, bl, col, hh, acc);
    else if (wv == 1) fused_taps_compute<K, DIL, 1, CH>(ah, al, bh, bl, col, hh, acc);
    else if (wv == 2) fused_taps_compute<K, DIL, 2, CH>(ah, al, bh, bl, col, hh, acc);
    else fused_taps_compute<K, DIL, 3, CH>(ah, al, bh, bl, col, hh, acc);
  }
  if (p.bslab && bx_ == 0) {
    // bias gradient: row sums of dy seen by this split (a row of the tile belongs to one wave, or half-wave at CH = 64)
    constexpr int RL = CH == 128 ? 64 : 32;                // lanes per row
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      float v = bsum[i];
#pragma unroll
      for (int o = RL / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      const int row = (tid + 256 * i) / (CH / 2);
      if ((lane & (RL - 1)) == 0 && co0 + row < p.Cout) p.bslab[(int64_t)split * p.Cout + co0 + row] = v;
    }
  }
  const int ci = ci0 + col;
  if (ci < p.Cin) {
    float* sl = p.slab + (int64_t)split * K * p.Cout * p.Cin;      // [split][k][co][ci], summed by wgrad_slab_reduce_kernel
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int k = wv + 4 * j;
      if (k >= K) break;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = co0 + acc_row(i, hh);
        if (co < p.Cout) sl[((int64_t)k * p.Cout + co) * p.Cin + ci] = acc[j][i];
      }
    }
  }
}

// ---- 1 x 1 weight gradients of wide layers in ONE pass (round 6) ------------------------------------------------------------------
// dw[co][ci] = sum_{b, t} dy[b][co][t] x[b][ci][t]: a GEMM whose reduction runs along the rows' contiguous axis.  Wide layers (more than
// 128 x 128 channels: the WaveNet res / skip layers, the attention projections, the diffusion model's 512 / 1536-row linear layers;
// ~170 calls per VQ-VAE-GAN step, ~65 per diffusion step) took wgrad_split_pair_kernel (a full pass over both operands: fp32 read,
// bf16 hi / lo write) + the pre-split all-taps kernel with one tap (hi / lo read): 13 + 10 us at the WN shape, 18 + 26 us for the
// diffusion qkv layer.  Here a workgroup owns a 64 x 64 tile of dw, reads the fp32 rows of both operands once per 128-position chunk
// (float4 requests, the next chunk in flight under the current one's MFMAs), applies the fused leaky-relus, splits into bf16 hi / lo on
// the way into LDS and accumulates lo.hi + hi.lo + hi.hi per 16-position k-step (2 x 2 waves of 32 x 32).  Split-K over chunks with
// per-split slabs exactly like the other weight-gradient kernels (deferred, ordered reduce); the bias gradient rides along as row
// sums of dy.  Needs Lout % 4 == 0 (16-byte row pieces).
constexpr int W1_CH = 128, W1_PITCH = W1_CH + 8;
__global__ __launch_bounds__(256, 2) void conv1x1_wgrad_fused_kernel(WgradMfmaParams p) {
  __shared__ __attribute__((aligned(16))) bf16 sm[4 * 64 * W1_PITCH];
  bf16* ah = sm; bf16* al = sm + 64 * W1_PITCH; bf16* bh = sm + 2 * 64 * W1_PITCH; bf16* bl = sm + 3 * 64 * W1_PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, col = lane & 31;
  const int wm = wave & 1, wn = wave >> 1;
  int bx_, by_, bz_;
  xcd_tile(bx_, by_, bz_);
  const int ci0 = bx_ * 64, co0 = by_ * 64, split = bz_;
  const int L = p.Lout, nlc = (L + W1_CH - 1) / W1_CH, nchunks = p.B * nlc;
  // staging map: 8 requests per operand and thread; request i covers row (tid >> 5) + 8 i, positions 4 (tid & 31) .. + 3 of the chunk
  const int srow = tid >> 5, st = (tid & 31) * 4;
  float4 ra[8], rb[8];
  float bsum[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) bsum[i] = 0.f;
  auto request = [&](int chunk) {
    const int b = chunk / nlc, l = (chunk - b * nlc) * W1_CH + st;
    const int lc = min(l, L - 4);                                   // (clamped: a piece past the row end is requested in range, zeroed below)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = srow + 8 * i;
      ra[i] = *reinterpret_cast<const float4*>(p.dy + ((int64_t)b * p.Cout + min(co0 + r, p.Cout - 1)) * L + lc);
      rb[i] = *reinterpret_cast<const float4*>(p.x + ((int64_t)b * p.Cin + min(ci0 + r, p.Cin - 1)) * L + lc);
    }
  };
  auto split4 = [](const float4 v, float slope, bf16* hi, bf16* lo) {
    const float e0 = lrelu_f(v.x, slope), e1 = lrelu_f(v.y, slope), e2 = lrelu_f(v.z, slope), e3 = lrelu_f(v.w, slope);
    bf16x4 h, w;
    h[0] = (bf16)e0; h[1] = (bf16)e1; h[2] = (bf16)e2; h[3] = (bf16)e3;
    w[0] = (bf16)(e0 - (float)h[0]); w[1] = (bf16)(e1 - (float)h[1]); w[2] = (bf16)(e2 - (float)h[2]); w[3] = (bf16)(e3 - (float)h[3]);
    *reinterpret_cast<bf16x4*>(hi) = h;
    *reinterpret_cast<bf16x4*>(lo) = w;
  };
  auto deposit = [&](int chunk) {
    const int b = chunk / nlc, l = (chunk - b * nlc) * W1_CH + st;
    const bool tok = l < L;                                          // (L % 4 == 0: a piece is inside the row or past its end)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = srow + 8 * i;
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 a = (tok && co0 + r < p.Cout) ? ra[i] : z, bq = (tok && ci0 + r < p.Cin) ? rb[i] : z;
      bsum[i] += (a.x + a.y) + (a.z + a.w);
      split4(a, p.dy_slope, ah + r * W1_PITCH + st, al + r * W1_PITCH + st);
      split4(bq, p.x_slope, bh + r * W1_PITCH + st, bl + r * W1_PITCH + st);
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int chunk0 = split * p.chunks_per_block;
  const int nmine = max(0, min(p.chunks_per_block, nchunks - chunk0));
  const bf16* arow_h = ah + (wm * 32 + col) * W1_PITCH + hh * 8;
  const bf16* arow_l = al + (wm * 32 + col) * W1_PITCH + hh * 8;
  const bf16* brow_h = bh + (wn * 32 + col) * W1_PITCH + hh * 8;
  const bf16* brow_l = bl + (wn * 32 + col) * W1_PITCH + hh * 8;
  if (nmine > 0) request(chunk0);
  for (int cc = 0; cc < nmine; ++cc) {
    __syncthreads();                                       // everyone is done reading the previous chunk
    deposit(chunk0 + cc);
    __syncthreads();
    if (cc + 1 < nmine) request(chunk0 + cc + 1);          // in flight while the matrix cores run
#pragma unroll
    for (int ks = 0; ks < W1_CH / 16; ++ks) {
      const bf16x8 a_h = *reinterpret_cast<const bf16x8*>(arow_h + ks * 16), a_l = *reinterpret_cast<const bf16x8*>(arow_l + ks * 16);
      const bf16x8 b_h = *reinterpret_cast<const bf16x8*>(brow_h + ks * 16), b_l = *reinterpret_cast<const bf16x8*>(brow_l + ks * 16);
      acc = mfma32(a_l, b_h, acc);
      acc = mfma32(a_h, b_l, acc);
      acc = mfma32(a_h, b_h, acc);
    }
  }
  if (p.bslab && bx_ == 0) {
    // bias gradient: row sums of dy seen by this split (a row's 32 staging lanes are one half-wave)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v = bsum[i];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      const int r = srow + 8 * i;
      if ((lane & 31) == 0 && co0 + r < p.Cout) p.bslab[(int64_t)split * p.Cout + co0 + r] = v;
    }
  }
  const int ci = ci0 + wn * 32 + col;
  if (ci < p.Cin) {
    float* sl = p.slab + (int64_t)split * p.Cout * p.Cin;          // [split][co][ci], summed by the slab reduce
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = co0 + wm * 32 + acc_row(i, hh);
      if (co < p.Cout) sl[(int64_t)co * p.Cin + ci] = acc[i];
    }
  }
}

// dw[co][ci][k] += sum_split slab[split][k][co][ci]
__global__ __launch_bounds__(256) void wgrad_slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nsplit,
                                                                int K, int Cout, int Cin, const float* __restrict__ bslab,
                                                                float* __restrict__ db) {
  const int64_t per = (int64_t)K * Cout * Cin;
  const int s0 = blockIdx.y * SLAB_G, s1 = min(nsplit, s0 + SLAB_G);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256) {
    // four independent partial sums, four loads in flight per step (a plain `s += slab[...]` loop is one latency per split)
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int sp = s0;
    for (; sp + 4 <= s1; sp += 4) {
      const float v0 = slab[sp * per + i], v1 = slab[(sp + 1) * per + i], v2 = slab[(sp + 2) * per + i], v3 = slab[(sp + 3) * per + i];
      a0 += v0; a1 += v1; a2 += v2; a3 += v3;
    }
    for (; sp < s1; ++sp) a0 += slab[sp * per + i];
    const float s = (a0 + a1) + (a2 + a3);
    const int ci = (int)(i % Cin), co = (int)((i / Cin) % Cout), k = (int)(i / Cin / Cout);
    float* o = dw + ((int64_t)co * Cin + ci) * K + k;
    if (gridDim.y == 1) *o += s;
    else atomicAdd(o, s);
  }
  if (bslab) {       // the fused kernel's per-split row sums of dy -> bias gradient
    for (int c = blockIdx.x * 256 + threadIdx.x; c < Cout; c += gridDim.x * 256) {
      float s = 0.f;
      for (int sp = s0; sp < s1; ++sp) s += bslab[(int64_t)sp * Cout + c];
      if (gridDim.y == 1) db[c] += s;
      else atomicAdd(db + c, s);
    }
  }
}

// ---- deferred slab reduction (ABI v8) -------------------------------------------------------------------------------------
// Every split-bf16 weight-gradient launch used to be followed by its own wgrad_slab_reduce launch over slabs in the shared
// scratch (543 launches, 5.9 ms of a VQ-VAE-GAN step, most of them far too small to use the memory system).  The host
// registers the array the gradients accumulate into (a WeightNormBank's flat dW, the optimizer's gradient arena); while the
// arena is ARMED a weight-gradient call writes its slabs to a PERSISTENT slot (recorded on first sight, keyed by (dw, shape,
// splits)) and skips the reduce, and ttts_conv_wgrad_arena_reduce sums every layer touched since _begin in one launch.
struct SlabDesc {
  const float* slab; float* dw; const float* bslab; float* db;
  int nsplit, K, Cout, Cin;
  int block_begin, pad_;
};
constexpr int SLABB_EPB = 256;       // elements per workgroup: wave g sums splits g, g + 4, ... of 256 consecutive elements, FOUR per lane
                                     // (round 5: it was 64 with one element per lane -- 82 000 workgroups of a few hundred bytes for one
                                     // 1024 x 1024 x 5 gradient: dispatch-bound at 1.8 TB/s; per-element summation order unchanged)

__global__ __launch_bounds__(256) void wgrad_slab_reduce_batched_kernel(const SlabDesc* __restrict__ table, int n_desc, int block_base) {
  __shared__ float part[4][SLABB_EPB];
  const int gb = (int)blockIdx.x + block_base;       // (block_base: a launch over a single descriptor of the table)
  int lo = 0, hi = n_desc - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].block_begin <= gb) lo = mid; else hi = mid - 1;
  }
  const SlabDesc d = table[lo];
  const int64_t per = (int64_t)d.K * d.Cout * d.Cin;
  const int nblk_w = (int)((per + SLABB_EPB - 1) / SLABB_EPB);
  const int lb = gb - d.block_begin;
  const int e4 = (threadIdx.x & 63) * 4, g = threadIdx.x >> 6;
  const bool bias_blk = lb >= nblk_w;          // trailing workgroups of a layer: its bias gradient [split][Cout] -> db
  const float* src = bias_blk ? d.bslab : d.slab;
  const int64_t len = bias_blk ? d.Cout : per;
  const int64_t i0 = (int64_t)(bias_blk ? lb - nblk_w : lb) * SLABB_EPB + e4;
  float acc[4][4];                             // [independent chain][element]: four loads in flight per element, as before
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[c][j] = 0.f;
  const bool vec = (len & 3) == 0 && i0 + 3 < len && (reinterpret_cast<uintptr_t>(src) & 15u) == 0;
  auto load4 = [&](int sp, float (&v)[4]) {
    const float* q = src + sp * len + i0;
    if (vec) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(q);
      v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = i0 + j < len ? q[j] : 0.f;
    }
  };
  if (i0 < len) {
    int sp = g;
    for (; sp + 12 < d.nsplit; sp += 16) {
      float v0[4], v1[4], v2[4], v3[4];
      load4(sp, v0); load4(sp + 4, v1); load4(sp + 8, v2); load4(sp + 12, v3);
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc[0][j] += v0[j]; acc[1][j] += v1[j]; acc[2][j] += v2[j]; acc[3][j] += v3[j]; }
    }
    for (; sp < d.nsplit; sp += 4) {
      float v0[4];
      load4(sp, v0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[0][j] += v0[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) part[g][e4 + j] = (acc[0][j] + acc[1][j]) + (acc[2][j] + acc[3][j]);
  __syncthreads();
  if (g == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t i = i0 + j;
      if (i >= len) continue;
      const float sum = (part[0][e4 + j] + part[1][e4 + j]) + (part[2][e4 + j] + part[3][e4 + j]);
      if (bias_blk) {
        d.db[i] += sum;
      } else {
        const int ci = (int)(i % d.Cin), co = (int)((i / d.Cin) % d.Cout), k = (int)(i / d.Cin / d.Cout);
        d.dw[((int64_t)co * d.Cin + ci) * d.K + k] += sum;
      }
    }
  }
}

struct SlabKey {              // (the split count is NOT part of the key: it follows the batch's row lengths, the slot is per layer)
  const float* dw; const float* db; int K, Cout, Cin;
  bool operator==(const SlabKey& o) const { return dw == o.dw && db == o.db && K == o.K && Cout == o.Cout && Cin == o.Cin; }
};
struct SlabKeyHash {
  size_t operator()(const SlabKey& k) const {
    uint64_t h = reinterpret_cast<uint64_t>(k.dw) * 0x9E3779B97F4A7C15ull ^ reinterpret_cast<uint64_t>(k.db);
    for (int v : {k.K, k.Cout, k.Cin}) h = (h ^ (uint64_t)(uint32_t)v) * 0x100000001B3ull;
    return (size_t)h;
  }
};
struct SlabArena {
  uint32_t magic = TTTS_HANDLE_SLAB;
  std::mutex mu;
  const char* dw_lo; const char* dw_hi;
  char* storage; int64_t bytes, used;       // [descriptor table: max_entries][slabs ...]
  int max_entries;
  bool armed = false;
  std::vector<SlabDesc> host;
  std::vector<int> cap;                     // splits a slot has room for (a later batch with longer rows gets a new, larger slot)
  std::vector<uint8_t> touched;
  // an entry that was handed out during a stream capture is part of a recorded graph (the weight-gradient launch holds its slab
  // pointer and split count, the reduce launch reads them from the table at replay): it stays as it is -- calls that would need
  // another split count or a larger slot reduce immediately instead -- until ttts_conv_wgrad_arena_release_graphs
  std::vector<uint8_t> in_graph;
  int n_touched = 0;
  std::unordered_map<SlabKey, int, SlabKeyHash> index;
  int64_t blocks = 0, deferred = 0, fallbacks = 0, partial_reduces = 0, generation = 0;
};

static int slab_desc_blocks(const SlabDesc& d) {
  return (int)(cdiv((int64_t)d.K * d.Cout * d.Cin, SLABB_EPB) + (d.bslab ? cdiv(d.Cout, SLABB_EPB) : 0));
}

// The persistent slab of this weight-gradient call when its reduce can be deferred (then *bslab_out is the matching bias slab,
// or NULL without `db`), else NULL: slabs in scratch and an immediate reduce, as before.
static float* slab_defer(const ConvCtx& cx, float* dw, float* db, bool want_bslab, int nsplit, int K, int Cout, int Cin,
                         hipStream_t stream, float** bslab_out) {
  const char* wp = reinterpret_cast<const char*>(dw);
  if (want_bslab) {          // the bias gradient is summed later too: only into a persistent (registered) gradient array
    const char* bp = reinterpret_cast<const char*>(db);
    bool ok = false;
    for (int h = 0; h < cx.n_handles; ++h) {
      const SlabArena* c = static_cast<const SlabArena*>(cx.handles[h]);
      ok = ok || (c && c->magic == TTTS_HANDLE_SLAB && bp >= c->dw_lo && bp < c->dw_hi);
    }
    if (!ok) return nullptr;
  }
  for (int h = 0; h < cx.n_handles; ++h) {
    SlabArena* c = static_cast<SlabArena*>(cx.handles[h]);
    if (!c || c->magic != TTTS_HANDLE_SLAB || wp < c->dw_lo || wp >= c->dw_hi) continue;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->armed) continue;            // (a later handle over the same range may be the armed one)
    const SlabKey key{dw, want_bslab ? db : nullptr, K, Cout, Cin};
    const int64_t per = (int64_t)K * Cout * Cin;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
    auto it = c->index.find(key);
    int idx = it != c->index.end() ? it->second : -1;
    if (idx >= 0 && c->touched[idx]) { ++c->fallbacks; return nullptr; }      // a second gradient into the same dw in this phase
    if (idx >= 0 && nsplit <= c->cap[idx]) {
      if (c->host[idx].nsplit != nsplit) {                  // other row lengths than last time: the table entry follows
        if (capturing || c->in_graph[idx]) { ++c->fallbacks; return nullptr; }
        SlabDesc d = c->host[idx];
        d.nsplit = nsplit;
        desc_store_kernel<SlabDesc><<<1, 1, 0, stream>>>(reinterpret_cast<SlabDesc*>(c->storage) + idx, d);
        if (hipGetLastError() != hipSuccess) { ++c->fallbacks; return nullptr; }
        c->host[idx] = d;
        ++c->generation;
      }
    } else {                                                // first sighting, or the slot is too small for this split count
      const int capn = idx >= 0 ? std::max(nsplit, 2 * c->cap[idx]) : nsplit;
      const int64_t need = (((int64_t)capn * (per + (want_bslab ? Cout : 0)) * (int64_t)sizeof(float) + 255) / 256) * 256;
      if (capturing || (idx >= 0 && c->in_graph[idx]) || (idx < 0 && (int)c->host.size() >= c->max_entries) ||
          c->used + need > c->bytes) {
        ++c->fallbacks;
        return nullptr;
      }
      SlabDesc d;
      float* slab = reinterpret_cast<float*>(c->storage + c->used);
      d.slab = slab; d.dw = dw; d.bslab = want_bslab ? slab + (int64_t)capn * per : nullptr; d.db = want_bslab ? db : nullptr;
      d.nsplit = nsplit; d.K = K; d.Cout = Cout; d.Cin = Cin; d.pad_ = 0;
      const bool fresh = idx < 0;
      d.block_begin = fresh ? (int)c->blocks : c->host[idx].block_begin;
      const int slot = fresh ? (int)c->host.size() : idx;
      desc_store_kernel<SlabDesc><<<1, 1, 0, stream>>>(reinterpret_cast<SlabDesc*>(c->storage) + slot, d);
      if (hipGetLastError() != hipSuccess) { ++c->fallbacks; return nullptr; }
      if (fresh) {
        idx = slot;
        c->host.push_back(d);
        c->touched.push_back(0); c->in_graph.push_back(0); c->cap.push_back(capn);
        c->blocks += slab_desc_blocks(d);
        c->index.emplace(key, idx);
      } else {
        c->host[idx] = d;
        c->cap[idx] = capn;
      }
      c->used += need;
      ++c->generation;
    }
    c->touched[idx] = 1;
    if (capturing) c->in_graph[idx] = 1;
    ++c->n_touched;
    ++c->deferred;
    *bslab_out = const_cast<float*>(c->host[idx].bslab);
    return const_cast<float*>(c->host[idx].slab);
  }
  return nullptr;
}

template <int K, int DIL, int K0, int KN, int TILE, int S = 1, int PO = 0>
static void launch_wgrad_taps_one(const WgradB3Params& p, dim3 grid, hipStream_t stream) {
  constexpr int BASE = S > 1 ? 0 : (K0 * DIL) / 8 * 8, WIN = S > 1 ? 64 + (K0 + KN - 1 + PO) / S : 64 + (K0 + KN - 1) * DIL - BASE;
  constexpr int WP0 = (WIN + 7) / 8 * 8 + ((S > 1 && (WIN - 64) % 8 != 0) ? 0 : 8), WP = S > 1 ? ((S * (WP0 / 8)) % 2 ? WP0 : WP0 + 8) : (WP0 | 8);
  constexpr int AC = (TILE * 9 + 63) / 64, BC = (TILE * (S * WP / 8) + 63) / 64;
  constexpr int STAGE_EL = 2 * (AC + BC) * 512;
  constexpr size_t smem = (size_t)2 * STAGE_EL * sizeof(bf16);
  static_assert(smem <= 160 * 1024, "taps stage too large");
  static OnceFlag attr;
  if (set_attr_once(reinterpret_cast<const void*>(conv1d_wgrad_bf16x3_taps_kernel<K, DIL, K0, KN, TILE, S, PO>), attr)) return;
  conv1d_wgrad_bf16x3_taps_kernel<K, DIL, K0, KN, TILE, S, PO><<<grid, 256, smem, stream>>>(p);
}
template <int K, int DIL, int TILE>
static void launch_wgrad_taps_t(const WgradB3Params& p, dim3 grid, hipStream_t stream) {
  if (K <= 7) {
    launch_wgrad_taps_one<K, DIL, 0, (K <= 7 ? K : 1), TILE>(p, grid, stream);
  } else {
    launch_wgrad_taps_one<K, DIL, 0, 6, TILE>(p, grid, stream);
    launch_wgrad_taps_one<K, DIL, 6, (K > 6 ? K - 6 : 1), TILE>(p, grid, stream);
  }
}
template <int K, int DIL>
static void launch_wgrad_taps(const WgradB3Params& p, dim3 grid, int tile, hipStream_t stream) {
  (void)tile;      // (a 32 x 32 variant of this pre-split kernel lost to the fused single-pass kernel below 128 channels)
  launch_wgrad_taps_t<K, DIL, 64>(p, grid, stream);
}

#ifndef TTTS_W1_TILES
#define TTTS_W1_TILES 56      // most (ci, co) tiles for the one-pass 1 x 1 weight gradient (A/B builds: -DTTTS_W1_TILES=n)
#endif

static int conv1d_wgrad_bf16x3_try(const float* dy, const float* x, float* dw, float* db, bool* db_done, int B, int Cin, int Lin,
                                   int Cout, int Lout, int K, int stride, int pad, int dil, float dy_slope, float x_slope,
                                   const ConvCtx& cx, hipStream_t stream, bool* handled) {
  *handled = false;
  if (!cx.ws || (cx.flags & 4096)) return TTTS_OK;
  // (flag 1, experiment: small 1 x 1 weight gradients -- the WN res/skip layers, 192 x 192 over 8192 positions -- on the exact-f32
  // kernel: one launch instead of pre-pass + split-bf16 kernel)
  if ((cx.flags & 1) && K == 1 && (int64_t)Cin * Cout <= 256 * 256 && (int64_t)B * Lout <= 32768) return TTTS_OK;
  // workgroups a launch aims for when it splits the reduction: 512 = two per CU; flag 536870912: one per CU (half the slab
  // traffic, less latency hiding -- tools/gpu_ab_vq.sh)
  const int wg_target = (cx.flags & 536870912) ? 256 : (cx.flags & 1073741824) ? 1024 : 512;   // (flag 1073741824: four per CU)
  // all-taps kernel: stride 1, the ResBlock1 family (K in {3, 7, 11}, dilation in {1, 3, 5}).  64 x 64 tiles from 128
  // channels (measured 1.9x over one tap per workgroup at C = 128 / 256); 32 x 32 tiles below (16..96-channel long rows: the
  // exact-fp32 MFMA kernel they used to take ran at 15-60 TF/s).  Flag 16384: never; flag 1048576: not below 128 channels.
  const bool wide_c = (int64_t)Cin * Cout > 128 * 128;     // pre-split 64 x 64 tiles above, the fused 32 x 32 kernel up to 128 x 128
                                                           // channels (measured at 128: 153-167 us against 190; a tie at 256)
  const bool k5 = (K == 5 || K == 1) && dil == 1 && wide_c;   // (DiscriminatorP / S, the WN in-layers, 1 x 1 convolutions; pre-split kernel only)
  const bool taps = stride == 1 && (((K == 3 || K == 7 || K == 11) && (dil == 1 || dil == 3 || dil == 5)) || k5) &&
                    (wide_c || (Cin >= 16 && Cout >= 16 && !(cx.flags & 1048576))) && !(cx.flags & 16384);
  const int TT = wide_c ? 64 : 32;
  if (taps && !wide_c && !(cx.flags & 2097152)) {   // (flag 2097152: the pre-split kernels instead, for comparison)
    const int CH = (Lout > 64 && !(K == 11 && dil >= 3)) ? 128 : 64;   // (K = 11 with dilation 3 / 5 would spill at 128)
    const int nlc = (int)cdiv(Lout, CH), nchunks = B * nlc;
    const int tiles = (int)(cdiv(Cin, 32) * cdiv(Cout, 32));
    const int splits = (int)std::max<int64_t>(1, std::min<int64_t>(nchunks, cdiv(wg_target, tiles)));
    const int cpb = (int)cdiv(nchunks, splits), nsplit = (int)cdiv(nchunks, cpb);
    const int64_t slab_bytes = (int64_t)nsplit * (K * Cout * Cin + Cout) * (int64_t)sizeof(float);
    if (slab_bytes <= cx.ws_bytes) {
      float* slab = static_cast<float*>(cx.ws);
      float* bslab = db ? slab + (int64_t)nsplit * K * Cout * Cin : nullptr;
      float* pb = nullptr;
      float* ps = slab_defer(cx, dw, db, db != nullptr, nsplit, K, Cout, Cin, stream, &pb);     // (persistent slabs: reduce deferred)
      if (ps) { slab = ps; bslab = pb; }
      WgradMfmaParams p{dy, x, dw, B, Cin, Lin, Cout, Lout, K, 1, pad, dil, dy_slope, x_slope, cpb, slab, 64, bslab};
      dim3 grid((unsigned)cdiv(Cin, 32), (unsigned)cdiv(Cout, 32), (unsigned)nsplit);
#define TTTS_FUSED(KK, DD)                                                                               \
  if (K == KK && dil == DD) {                                                                            \
    if (CH == 128) conv1d_wgrad_fused_taps_kernel<KK, DD, (KK == 11 && DD >= 3) ? 64 : 128><<<grid, 256, 0, stream>>>(p); \
    else conv1d_wgrad_fused_taps_kernel<KK, DD, 64><<<grid, 256, 0, stream>>>(p);                        \
  }
      TTTS_FUSED(3, 1) TTTS_FUSED(3, 3) TTTS_FUSED(3, 5) TTTS_FUSED(7, 1) TTTS_FUSED(7, 3) TTTS_FUSED(7, 5)
      TTTS_FUSED(11, 1) TTTS_FUSED(11, 3) TTTS_FUSED(11, 5)
#undef TTTS_FUSED
      if (!ps) wgrad_slab_reduce_kernel<<<dim3((unsigned)std::min<int64_t>(cdiv((int64_t)K * Cout * Cin, 256), 4096), (unsigned)cdiv(nsplit, SLAB_G)), 256, 0, stream>>>(slab, dw, nsplit, K, Cout, Cin, bslab, db);
      if (db) *db_done = true;
      *handled = true;
      return check_launch("conv1d_wgrad_fused_taps");
    }
  }
  if (K == 1 && stride == 1 && pad == 0 && wide_c && Lout == Lin && Lout % 4 == 0 && Lout >= 48 && !(cx.flags & (16384 | 2048)) &&
      cdiv(Cin, 64) * cdiv(Cout, 64) <= TTTS_W1_TILES &&
      (reinterpret_cast<uintptr_t>(dy) & 15u) == 0 && (reinterpret_cast<uintptr_t>(x) & 15u) == 0) {
    // one pass over the fp32 operands (conv1x1_wgrad_fused_kernel; flag 2048: the pre-pass + pre-split kernel instead).  Up to 56
    // tiles: the WaveNet / attention-projection / 1025 -> 192 layers (WN res|skip 34.0 -> 24.8 us, 1025 -> 192 62.6 -> 46.9 us); the
    // diffusion model's 512 .. 1536-row layers (64 .. 192 tiles, each operand row re-read by 8 .. 24 tiles) measured level with the
    // pre-split path, whose LDS-DMA kernel re-reads half-size copies: 16.1 vs 15.9 ms per step, left there (re-measured after the
    // XCD-contiguous tile order, tools/gpu_r6_au.sh: <= 64 tiles 15.08 vs 15.14 ms, all layers 15.45 ms)
    const int nlc = (int)cdiv(Lout, W1_CH), nchunks = B * nlc;
    const int tiles = (int)(cdiv(Cin, 64) * cdiv(Cout, 64));
    const int splits = (int)std::max<int64_t>(1, std::min<int64_t>(nchunks, cdiv(wg_target, tiles)));
    const int cpb = (int)cdiv(nchunks, splits), nsplit = (int)cdiv(nchunks, cpb);
    const int64_t slab_bytes = (int64_t)nsplit * ((int64_t)Cout * Cin + Cout) * (int64_t)sizeof(float);
    if (slab_bytes <= cx.ws_bytes) {
      float* slab = static_cast<float*>(cx.ws);
      float* bslab = db ? slab + (int64_t)nsplit * Cout * Cin : nullptr;
      float* pb = nullptr;
      float* ps = slab_defer(cx, dw, db, db != nullptr, nsplit, 1, Cout, Cin, stream, &pb);     // (persistent slabs: reduce deferred)
      if (ps) { slab = ps; bslab = pb; }
      WgradMfmaParams p{dy, x, dw, B, Cin, Lin, Cout, Lout, 1, 1, 0, 1, dy_slope, x_slope, cpb, slab, 64, bslab};
      conv1x1_wgrad_fused_kernel<<<dim3((unsigned)cdiv(Cin, 64), (unsigned)cdiv(Cout, 64), (unsigned)nsplit), 256, 0, stream>>>(p);
      if (!ps) wgrad_slab_reduce_kernel<<<dim3((unsigned)std::min<int64_t>(cdiv((int64_t)Cout * Cin, 256), 4096), (unsigned)cdiv(nsplit, SLAB_G)), 256, 0, stream>>>(slab, dw, nsplit, 1, Cout, Cin, bslab, db);
      if (db) *db_done = true;
      *handled = true;
      return check_launch("conv1x1_wgrad_fused");
    }
  }
  if (taps) {
    // short rows: all batch elements of a channel as one virtual row (wgrad_split_cat_kernel)
    const int Lg = Lout + (K - 1) * dil;
    const bool cat = B > 1 && cdiv(Lout, 64) * 64 * 100 > (int64_t)Lg * 115;
    const int Bk = cat ? 1 : B;                                                  // batch elements as the kernel sees them
    const int Lq = (int)(cdiv(cat ? (int64_t)B * Lg : Lout, 64) * 64);
    const int WPmax = ((64 + (K - 1) * dil + 7) / 8 * 8 + 8) | 8;
    const int Li = (int)(((int64_t)Lq + (K - 1) * dil + WPmax + 8 + 7) / 8 * 8);   // every staged 16-byte piece stays inside the row
    const int64_t dy_el = (int64_t)Bk * Cout * Lq, x_par = (int64_t)Bk * Cin * Li;
    const int nlc0 = Lq / 64, nchunks0 = Bk * nlc0;
    const int tiles0 = (int)(cdiv(Cin, 64) * cdiv(Cout, 64));
    const int splits0 = (int)std::max<int64_t>(1, std::min<int64_t>(nchunks0, cdiv(wg_target, tiles0)));
    const int cpb0 = (int)cdiv(nchunks0, splits0);
    const int nsplit = (int)cdiv(nchunks0, cpb0);
    const int64_t slab_el = (int64_t)nsplit * K * Cout * Cin;
    const int64_t need = (2 * dy_el + 4 * x_par) * (int64_t)sizeof(bf16) + slab_el * (int64_t)sizeof(float) + 128;
    if (need <= cx.ws_bytes) {
      bf16* dyh = static_cast<bf16*>(cx.ws);
      bf16* dyl = dyh + (dy_el + 7) / 8 * 8;
      bf16* xh = dyl + (dy_el + 7) / 8 * 8;
      bf16* xl = xh + 2 * x_par;
      if (cat) {
        launch_wgrad_cat_pair(dy, dyh, dyl, Cout, Lout, Lq, dy_slope, db, x, xh, xl, Cin, Lin, pad, Li, x_slope, 1, B, Lg, stream);
      } else {
        launch_wgrad_splits(dy, x, dyh, dyl, xh, xl, (int64_t)B * Cout, Lout, Lq, dy_slope, (int64_t)B * Cin, Lin, 1, Li, pad, x_slope, 1, db, Cout, stream);
      }
      if (db) *db_done = true;
      const int nlc = nlc0, nchunks = nchunks0, cpb = cpb0;
      float* slab = reinterpret_cast<float*>(reinterpret_cast<char*>(xl + 2 * x_par) + ((16 - (reinterpret_cast<uintptr_t>(xl + 2 * x_par) & 15)) & 15));
      float* pb = nullptr;
      float* ps = slab_defer(cx, dw, nullptr, false, nsplit, K, Cout, Cin, stream, &pb);
      if (ps) slab = ps;
      WgradB3Params p{dyh, dyl, xh, xl, dw, Bk, Cin, Cout, K, 1, dil, Lq, Li, 0, x_par, cpb, nchunks, nlc, slab};
      dim3 grid((unsigned)cdiv(Cin, 64), (unsigned)cdiv(Cout, 64), (unsigned)nsplit);
#define TTTS_TAPS(KK, DD) if (K == KK && dil == DD) launch_wgrad_taps<KK, DD>(p, grid, 64, stream);
      TTTS_TAPS(1, 1) TTTS_TAPS(3, 1) TTTS_TAPS(3, 3) TTTS_TAPS(3, 5) TTTS_TAPS(5, 1) TTTS_TAPS(7, 1) TTTS_TAPS(7, 3) TTTS_TAPS(7, 5)
      TTTS_TAPS(11, 1) TTTS_TAPS(11, 3) TTTS_TAPS(11, 5)
#undef TTTS_TAPS
      if (!ps) wgrad_slab_reduce_kernel<<<dim3((unsigned)std::min<int64_t>(cdiv((int64_t)K * Cout * Cin, 256), 4096), (unsigned)cdiv(nsplit, SLAB_G)), 256, 0, stream>>>(slab, dw, nsplit, K, Cout, Cin, nullptr, nullptr);
      *handled = true;
      return check_launch("conv1d_wgrad_bf16x3_taps");
    }
  }
  // stride-3 five-tap layers (DiscriminatorP's 32 -> 128 -> 512 -> 1024 chain): all taps per workgroup over the phase-
  // de-interleaved input, short rows laid end to end like the stride-1 case (flag 67108864: the one-tap kernel instead)
  if (stride == 3 && K == 5 && dil == 1 && pad == 2 && (wide_c || (Cin >= 32 && Cout >= 64)) && !(cx.flags & (16384 | 67108864))) {
    // (from 32 input channels: a half-empty 64-channel tile on the bf16 matrix cores still beats the exact-f32 MFMA kernel)
    constexpr int S = 3, PO = 1, PL = 3, DMAX = (5 - 1 + PO) / S, WPK = 72;     // WPK: the kernel's phase-window pitch
    const int Lg = Lout + DMAX;
    const bool cat = B > 1 && cdiv(Lout, 64) * 64 * 100 > (int64_t)Lg * 115;
    const int Bk = cat ? 1 : B;
    const int Lq = (int)(cdiv(cat ? (int64_t)B * Lg : Lout, 64) * 64);
    const int Li = (int)cdiv((int64_t)Lq + WPK + 8, 8) * 8;
    const int64_t dy_el = (int64_t)Bk * Cout * Lq, x_el = (int64_t)Bk * Cin * S * Li;
    const int nlc = Lq / 64, nchunks = Bk * nlc;
    const int tiles = (int)(cdiv(Cin, 64) * cdiv(Cout, 64));
    const int splits = (int)std::max<int64_t>(1, std::min<int64_t>(nchunks, cdiv(wg_target, tiles)));
    const int cpb = (int)cdiv(nchunks, splits), nsplit = (int)cdiv(nchunks, cpb);
    const int64_t slab_el = (int64_t)nsplit * K * Cout * Cin;
    const int64_t need = (2 * dy_el + 2 * x_el + 64) * (int64_t)sizeof(bf16) + slab_el * (int64_t)sizeof(float) + 128;
    if (need <= cx.ws_bytes) {
      bf16* dyh = static_cast<bf16*>(cx.ws);
      bf16* dyl = dyh + (dy_el + 7) / 8 * 8;
      bf16* xh = dyl + (dy_el + 7) / 8 * 8;
      bf16* xl = xh + (x_el + 7) / 8 * 8;
      if (cat) {
        launch_wgrad_cat_pair(dy, dyh, dyl, Cout, Lout, Lq, dy_slope, db, x, xh, xl, Cin, Lin, PL, Li, x_slope, S, B, Lg, stream);
      } else {
        launch_wgrad_splits(dy, x, dyh, dyl, xh, xl, (int64_t)B * Cout, Lout, Lq, dy_slope, (int64_t)B * Cin, Lin, S, Li, PL, x_slope, 1, db, Cout, stream);
      }
      if (db) *db_done = true;
      char* end = reinterpret_cast<char*>(xl + (x_el + 7) / 8 * 8);
      float* slab = reinterpret_cast<float*>(end + ((16 - (reinterpret_cast<uintptr_t>(end) & 15)) & 15));
      float* pb = nullptr;
      float* ps = slab_defer(cx, dw, nullptr, false, nsplit, K, Cout, Cin, stream, &pb);
      if (ps) slab = ps;
      WgradB3Params p{dyh, dyl, xh, xl, dw, Bk, Cin, Cout, K, S, dil, Lq, Li, PO, x_el, cpb, nchunks, nlc, slab};
      dim3 grid((unsigned)cdiv(Cin, 64), (unsigned)cdiv(Cout, 64), (unsigned)nsplit);
      launch_wgrad_taps_one<5, 1, 0, 5, 64, S, PO>(p, grid, stream);
      if (!ps) wgrad_slab_reduce_kernel<<<dim3((unsigned)std::min<int64_t>(cdiv((int64_t)K * Cout * Cin, 256), 4096), (unsigned)cdiv(nsplit, SLAB_G)), 256, 0, stream>>>(slab, dw, nsplit, K, Cout, Cin, nullptr, nullptr);
      *handled = true;
      return check_launch("conv1d_wgrad_bf16x3_taps_s3");
    }
  }
  // measured (tools/conv_bench.py, B = 32): one-tap-per-workgroup staging costs K x the operand traffic of the fp32 kernel,
  // so this path wins for few taps and wide layers (DiscriminatorP/S 1024-channel k5: 1.9-2.8x, FFN k3 1.6x, 1x1 1.4x) and
  // loses for the 7/11-tap ResBlock convolutions and for narrow long rows (pre-pass bytes); flag 8192 forces it (tests)
  if (!(cx.flags & 8192) && !(K <= 5 && Cin >= 128 && Cout >= 128)) return TTTS_OK;
  if (Cin < 16 || Cout < 16) return TTTS_OK;
  const int Lq = (int)(cdiv(Lout, 64) * 64);
  const int PL = (int)(cdiv(pad, stride) * stride);
  const int64_t qmax = (int64_t)(Lq - 1) * stride + (int64_t)(K - 1) * dil + PL - pad;
  const int Li = (int)((qmax / stride + 1 + 1 + 8 + 7) / 8 * 8);     // +1: odd parity copy, +8: 16-byte over-read slack
  const int64_t dy_el = (int64_t)B * Cout * Lq, x_par = (int64_t)B * Cin * stride * Li, x_el = 2 * x_par;
  const int64_t need = (2 * dy_el + 2 * x_el) * (int64_t)sizeof(bf16) + 64;
  if (need > cx.ws_bytes) return TTTS_OK;
  bf16* dyh = static_cast<bf16*>(cx.ws);
  bf16* dyl = dyh + (dy_el + 7) / 8 * 8;
  bf16* xh = dyl + (dy_el + 7) / 8 * 8;
  bf16* xl = xh + x_el;
  launch_wgrad_splits(dy, x, dyh, dyl, xh, xl, (int64_t)B * Cout, Lout, Lq, dy_slope, (int64_t)B * Cin, Lin, stride, Li, PL, x_slope, 2, db, Cout, stream);
  if (db) *db_done = true;
  const bool small = Cin <= 32 && Cout <= 32;
  const int TILE = small ? 32 : 64;
  const int nlc = Lq / 64, nchunks = B * nlc;
  const int tiles = (int)(cdiv(Cin, TILE) * cdiv(Cout, TILE)) * K;
  const int splits = (int)std::max<int64_t>(1, std::min<int64_t>(nchunks, cdiv(3 * wg_target, tiles)));
  const int cpb = (int)cdiv(nchunks, splits);
  const int nsp = (int)cdiv(nchunks, cpb);
  float* slab = nullptr;
  {
    char* end = reinterpret_cast<char*>(xl + x_el);
    end += (16 - (reinterpret_cast<uintptr_t>(end) & 15)) & 15;
    const int64_t slab_bytes = (int64_t)nsp * K * Cout * Cin * (int64_t)sizeof(float);
    if (!small && nsp > 1 && (end - static_cast<char*>(cx.ws)) + slab_bytes <= cx.ws_bytes) slab = reinterpret_cast<float*>(end);
  }
  float* pb = nullptr;
  float* ps = slab ? slab_defer(cx, dw, nullptr, false, nsp, K, Cout, Cin, stream, &pb) : nullptr;
  if (ps) slab = ps;
  WgradB3Params p{dyh, dyl, xh, xl, dw, B, Cin, Cout, K, stride, dil, Lq, Li, PL - pad, x_par, cpb, nchunks, nlc, slab};
  dim3 grid((unsigned)cdiv(Cin, TILE), (unsigned)cdiv(Cout, TILE), (unsigned)(K * nsp));
  if (small) conv1d_wgrad_bf16x3_kernel<32><<<grid, 256, 0, stream>>>(p);
  else conv1d_wgrad_bf16x3_kernel<64><<<grid, 256, 0, stream>>>(p);
  if (slab && !ps) wgrad_slab_reduce_kernel<<<dim3((unsigned)std::min<int64_t>(cdiv((int64_t)K * Cout * Cin, 256), 4096), (unsigned)cdiv(nsp, SLAB_G)), 256, 0, stream>>>(slab, dw, nsp, K, Cout, Cin, nullptr, nullptr);
  *handled = true;
  return check_launch("conv1d_wgrad_bf16x3");
}

int conv1d_wgrad_mfma_try(const float* dy, const float* x, float* dw, float* db, bool* db_done, int B, int Cin, int Lin, int Cout,
                          int Lout, int K, int stride, int pad, int dil, float dy_slope, float x_slope, const ConvCtx& cx,
                          hipStream_t stream, bool* handled) {
  *handled = false;
  {
    int rc = conv1d_wgrad_bf16x3_try(dy, x, dw, db, db_done, B, Cin, Lin, Cout, Lout, K, stride, pad, dil, dy_slope, x_slope, cx, stream, handled);
    if (rc || *handled) return rc;
  }
  if (Cin * K < 32) return TTTS_OK;
  // short rows: 16-position segments when whole 64-position chunks would be mostly padding
  const double waste64 = (double)cdiv(Lout, 64) * 64 / Lout, waste16 = (double)cdiv(Lout, 16) * 16 / Lout;
  const int SEGW = (waste64 > 1.15 * waste16) ? 16 : 64;
  const int lin_s = (SEGW - 1) * stride + (K - 1) * dil + 1;
  const int nch_max = 64 / K + 2;
  const size_t smem = ((size_t)64 * (WM_L + 1) + (size_t)nch_max * (((WM_L / SEGW) * lin_s) | 1)) * sizeof(float);
  if (smem > 96 * 1024) return TTTS_OK;
  const int chunks = (int)cdiv((int64_t)B * cdiv(Lout, SEGW), WM_L / SEGW);
  const int tiles = (int)(cdiv(Cin * K, 64) * cdiv(Cout, 64));
  const int splits = (int)std::max<int64_t>(1, std::min<int64_t>(chunks, cdiv(2048, tiles)));
  const int cpb = (int)cdiv(chunks, splits);
  const int nz = (int)cdiv(chunks, cpb);
  // partial sums go to per-split slabs + one reduce pass when the caller-owned scratch is available (deterministic, and
  // device-scope fp32 atomics from hundreds of splits onto a small weight tensor cost more than the GEMM itself)
  float* slab = nullptr;
  const int64_t per = (int64_t)Cout * Cin * K;
  if (cx.ws && nz > 1 && (int64_t)nz * per * (int64_t)sizeof(float) <= cx.ws_bytes) slab = static_cast<float*>(cx.ws);
  WgradMfmaParams p{dy, x, dw, B, Cin, Lin, Cout, Lout, K, stride, pad, dil, dy_slope, x_slope, cpb, slab, SEGW};
  static OnceFlag a;
  int rc = set_attr_once(reinterpret_cast<const void*>(conv1d_wgrad_mfma_kernel), a);
  if (rc) return rc;
  dim3 grid((unsigned)cdiv(Cin * K, 64), (unsigned)cdiv(Cout, 64), (unsigned)nz);
  conv1d_wgrad_mfma_kernel<<<grid, 256, smem, stream>>>(p);
  if (slab) wgrad_slab_sum_kernel<<<dim3((unsigned)std::min<int64_t>(cdiv(per, 256), 4096), (unsigned)cdiv(nz, SLAB_G)), 256, 0, stream>>>(slab, dw, nz, per);
  *handled = true;
  return check_launch("conv1d_wgrad_mfma");
}

}  // namespace ttts

// C entry points of the weight-split cache and the weight-gradient arena (include/ttts_hip.h).  Both are caller-owned objects:
// the library keeps no list of them -- a convolution call sees the ones named in its ttts_conv_ctx::handles.
extern "C" {

int ttts_conv_wsplit_cache_create(const void* w_base, int64_t w_bytes, void* storage, int64_t storage_bytes, int32_t max_entries,
                                  void** cache_out) {
  using namespace ttts;
  if (!w_base || w_bytes <= 0 || !storage || !cache_out || max_entries <= 0) return fail(TTTS_EINVAL, "wsplit cache: null / empty argument");
  if (reinterpret_cast<uintptr_t>(storage) % 256) return fail(TTTS_EINVAL, "wsplit cache: storage must be 256-byte aligned");
  const int64_t table = (((int64_t)max_entries * (int64_t)sizeof(WsplitDesc) + 255) / 256) * 256;
  if (storage_bytes <= table) return fail(TTTS_EINVAL, "wsplit cache: storage smaller than its descriptor table");
  WsplitCache* c = new WsplitCache();
  c->w_lo = static_cast<const char*>(w_base); c->w_hi = c->w_lo + w_bytes;
  c->storage = static_cast<char*>(storage); c->bytes = storage_bytes; c->used = table;
  c->max_entries = max_entries;
  c->host.reserve(max_entries); c->first_writer.reserve(max_entries);
  *cache_out = c;
  return TTTS_OK;
}

static ttts::WsplitCache* as_wsplit(void* h) {
  ttts::WsplitCache* c = static_cast<ttts::WsplitCache*>(h);
  return c && c->magic == ttts::TTTS_HANDLE_WSPLIT ? c : nullptr;
}
static ttts::SlabArena* as_slab(void* h) {
  ttts::SlabArena* c = static_cast<ttts::SlabArena*>(h);
  return c && c->magic == ttts::TTTS_HANDLE_SLAB ? c : nullptr;
}

int ttts_conv_wsplit_cache_refresh(void* cache, void* stream) {
  using namespace ttts;
  WsplitCache* c = as_wsplit(cache);
  if (!c) return fail(TTTS_EINVAL, "wsplit cache: not a cache handle");
  int n; int64_t blocks;
  {
    std::lock_guard<std::mutex> g(c->mu);
    c->armed = true;
    n = (int)c->host.size(); blocks = c->blocks;
    c->launches_saved += n;
    std::fill(c->first_writer.begin(), c->first_writer.end(), nullptr);     // rewritten below: every stream forked after this call may read
  }
  if (n == 0) return TTTS_OK;
  conv_weight_split_batched_kernel<<<(unsigned)blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(
      reinterpret_cast<const WsplitDesc*>(c->storage), n);
  return check_launch("conv_weight_split_batched");
}

int ttts_conv_wsplit_cache_disarm(void* cache) {
  using namespace ttts;
  WsplitCache* c = as_wsplit(cache);
  if (!c) return fail(TTTS_EINVAL, "wsplit cache: not a cache handle");
  std::lock_guard<std::mutex> g(c->mu);
  c->armed = false;
  return TTTS_OK;
}

int ttts_conv_wsplit_cache_stats(void* cache, int64_t* out4) {
  using namespace ttts;
  WsplitCache* c = as_wsplit(cache);
  if (!c || !out4) return fail(TTTS_EINVAL, "wsplit cache: null argument");
  std::lock_guard<std::mutex> g(c->mu);
  out4[0] = (int64_t)c->host.size(); out4[1] = c->used; out4[2] = c->hits; out4[3] = c->misses;
  return TTTS_OK;
}

int ttts_conv_wsplit_cache_destroy(void* cache) {
  using namespace ttts;
  if (!cache) return TTTS_OK;
  WsplitCache* c = as_wsplit(cache);
  if (!c) return fail(TTTS_EINVAL, "wsplit cache: not a cache handle");
  c->magic = 0;                       // (a context that still lists the handle skips it; the caller removes it from its contexts first)
  delete c;
  return TTTS_OK;
}

int ttts_conv_wgrad_arena_create(const void* dw_base, int64_t dw_bytes, void* storage, int64_t storage_bytes, int32_t max_entries,
                                 void** arena_out) {
  using namespace ttts;
  if (!dw_base || dw_bytes <= 0 || !storage || !arena_out || max_entries <= 0) return fail(TTTS_EINVAL, "wgrad arena: null / empty argument");
  if (reinterpret_cast<uintptr_t>(storage) % 256) return fail(TTTS_EINVAL, "wgrad arena: storage must be 256-byte aligned");
  const int64_t table = (((int64_t)max_entries * (int64_t)sizeof(SlabDesc) + 255) / 256) * 256;
  if (storage_bytes <= table) return fail(TTTS_EINVAL, "wgrad arena: storage smaller than its descriptor table");
  SlabArena* c = new SlabArena();
  c->dw_lo = static_cast<const char*>(dw_base); c->dw_hi = c->dw_lo + dw_bytes;
  c->storage = static_cast<char*>(storage); c->bytes = storage_bytes; c->used = table;
  c->max_entries = max_entries;
  c->host.reserve(max_entries); c->touched.reserve(max_entries); c->in_graph.reserve(max_entries); c->cap.reserve(max_entries);
  *arena_out = c;
  return TTTS_OK;
}

int ttts_conv_wgrad_arena_begin(void* arena) {
  using namespace ttts;
  SlabArena* c = as_slab(arena);
  if (!c) return fail(TTTS_EINVAL, "wgrad arena: not an arena handle");
  std::lock_guard<std::mutex> g(c->mu);
  std::fill(c->touched.begin(), c->touched.end(), 0);
  c->n_touched = 0;
  c->armed = true;
  return TTTS_OK;
}

int ttts_conv_wgrad_arena_reduce(void* arena, void* stream_) {
  using namespace ttts;
  SlabArena* c = as_slab(arena);
  if (!c) return fail(TTTS_EINVAL, "wgrad arena: not an arena handle");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  std::lock_guard<std::mutex> g(c->mu);
  const bool was_armed = c->armed;
  c->armed = false;
  if (!was_armed || c->n_touched == 0) return TTTS_OK;
  const int n = (int)c->host.size();
  if (c->n_touched == n) {          // the usual case: every recorded layer ran exactly once -> one launch over the whole table
    wgrad_slab_reduce_batched_kernel<<<(unsigned)c->blocks, 256, 0, stream>>>(reinterpret_cast<const SlabDesc*>(c->storage), n, 0);
  } else {                          // some recorded layers did not run in this phase: their slabs are stale -> per-layer launches
    ++c->partial_reduces;
    for (int i = 0; i < n; ++i) {
      if (!c->touched[i]) continue;
      wgrad_slab_reduce_batched_kernel<<<(unsigned)slab_desc_blocks(c->host[i]), 256, 0, stream>>>(
          reinterpret_cast<const SlabDesc*>(c->storage) + i, 1, c->host[i].block_begin);
    }
  }
  std::fill(c->touched.begin(), c->touched.end(), 0);
  c->n_touched = 0;
  return check_launch("wgrad_slab_reduce_batched");
}

int ttts_conv_wgrad_arena_disarm(void* arena) {
  using namespace ttts;
  SlabArena* c = as_slab(arena);
  if (!c) return fail(TTTS_EINVAL, "wgrad arena: not an arena handle");
  std::lock_guard<std::mutex> g(c->mu);
  c->armed = false;
  std::fill(c->touched.begin(), c->touched.end(), 0);
  c->n_touched = 0;
  return TTTS_OK;
}

int ttts_conv_wgrad_arena_release_graphs(void* arena) {
  using namespace ttts;
  SlabArena* c = as_slab(arena);
  if (!c) return fail(TTTS_EINVAL, "wgrad arena: not an arena handle");
  std::lock_guard<std::mutex> g(c->mu);
  std::fill(c->in_graph.begin(), c->in_graph.end(), 0);
  return TTTS_OK;
}

int ttts_conv_wgrad_arena_stats(void* arena, int64_t* out6) {
  using namespace ttts;
  SlabArena* c = as_slab(arena);
  if (!c || !out6) return fail(TTTS_EINVAL, "wgrad arena: null argument");
  std::lock_guard<std::mutex> g(c->mu);
  out6[0] = (int64_t)c->host.size(); out6[1] = c->used; out6[2] = c->deferred; out6[3] = c->fallbacks; out6[4] = c->partial_reduces;
  out6[5] = c->generation;
  return TTTS_OK;
}

int ttts_conv_wgrad_arena_destroy(void* arena) {
  using namespace ttts;
  if (!arena) return TTTS_OK;
  SlabArena* c = as_slab(arena);
  if (!c) return fail(TTTS_EINVAL, "wgrad arena: not an arena handle");
  c->magic = 0;
  delete c;
  return TTTS_OK;
}

// ---- ABI v11: range events of the fp16 conversions + the dynamic loss scale built on them ---------------------------------------
int ttts_conv_f16_events(int32_t* events3, int32_t reset, void* stream) {
  using namespace ttts;
  if (!events3) return fail(TTTS_EINVAL, "f16 events: null destination");
  f16_events_fetch_kernel<<<1, 64, 0, static_cast<hipStream_t>(stream)>>>(events3, reset);
  return check_launch("f16_events_fetch");
}

int ttts_loss_scale_check(float* ls8, int32_t* events3, float* skip, void* stream) {
  using namespace ttts;
  if (!ls8 || !events3) return fail(TTTS_EINVAL, "loss scale check: null argument");
  loss_scale_check_kernel<<<1, 64, 0, static_cast<hipStream_t>(stream)>>>(ls8, events3, skip);
  return check_launch("loss_scale_check");
}

int ttts_loss_scale_update(float* ls8, int32_t growth_interval, float backoff, float growth, void* stream) {
  using namespace ttts;
  if (!ls8) return fail(TTTS_EINVAL, "loss scale update: null state");
  if (!(backoff > 0.f && backoff <= 1.f) || !(growth >= 1.f)) return fail(TTTS_EINVAL, "loss scale update: backoff in (0, 1], growth >= 1");
  loss_scale_update_kernel<<<1, 64, 0, static_cast<hipStream_t>(stream)>>>(ls8, growth_interval, backoff, growth);
  return check_launch("loss_scale_update");
}

}  // extern "C"
