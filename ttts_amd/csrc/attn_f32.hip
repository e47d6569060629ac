// fp32 attention of the VQ-VAE text/style encoders (T <= ~1k, heads 2-4, d_k 64-128): relative-position self-attention
// (attentions.MultiHeadAttention, ttts/vqvae/attentions.py:177-375, window 4, heads_share), the MRTE cross-attention
// (ttts/utils/vc_utils.py:514-627 via vq2.py:28-46) and the MelStyleEncoder self-attention (modules.py:606-683).
//
// These are ~1 % of the step's FLOPs, so the design is the simple materialised one: scores and probabilities live in
// HBM as [B, H, Tq, Tk] (16 MB per layer at the training config), the two big contractions are strided batched fp32
// GEMMs on (B, C, T) tensors in place (no head transposes: a head is a contiguous block of d_k channel rows), and the
// relative-position terms (9 diagonals) are small row kernels fused with the masked softmax and its gradient.
#include <algorithm>

#include "common.hpp"

namespace ttts {

// ---- strided batched GEMM: C[m][n] = alpha * sum_k A[m][k] * B[k][n] (+ beta * C) with arbitrary element strides ------
struct BgemmParams {
  const float* A; const float* B; float* C;
  int M, N, K;
  int64_t a_sm, a_sk, b_sk, b_sn, c_sm, c_sn;
  int inner;                                  // batch index z -> (z / inner, z % inner)
  int64_t a_so, a_si, b_so, b_si, c_so, c_si; // outer / inner batch strides
  float alpha, beta;
};
constexpr int BG_T = 64, BG_K = 16;

// (round 2) The 4 x 4-per-thread fmaf loop is replaced by v_mfma_f32_32x32x2_f32: the same k-ordered exact-fp32 chain per
// output element (bit-identical results), one MFMA + two 4-byte LDS reads per 2 k where the VALU form needed 8 LDS values and
// 32 issue slots per k -- this kernel was 32 % of the diffusion step and the attention of every VITS stack.
// Waves 2 x 2 over the 64 x 64 tile, wave tile 32 x 32.
__global__ __launch_bounds__(256) void bgemm_kernel(BgemmParams p) {
  __shared__ float As[BG_K][BG_T + 4];
  __shared__ float Bs[BG_K][BG_T + 4];
  int bx_, by_, bz_;
  xcd_tile(bx_, by_, bz_);            // (the tiles of one batch element share its A / B panels: an XCD walks a contiguous run of them)
  const int z = bz_, zo = z / p.inner, zi = z % p.inner;
  const float* A = p.A + zo * p.a_so + zi * p.a_si;
  const float* B = p.B + zo * p.b_so + zi * p.b_si;
  float* C = p.C + zo * p.c_so + zi * p.c_si;
  const int m0 = by_ * BG_T, n0 = bx_ * BG_T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, hh = lane >> 5, col = lane & 31;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const bool a_m_fast = p.a_sm == 1, b_n_fast = p.b_sn == 1;
  for (int k0 = 0; k0 < p.K; k0 += BG_K) {
    // unconditional loads from clamped addresses, then a select (guarded loads compile to one branch each)
    float av[4], bv[4];
    int am[4], ak[4], bn[4], bk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (a_m_fast) { am[i] = tid & 63; ak[i] = (tid >> 6) + 4 * i; } else { ak[i] = tid & 15; am[i] = (tid >> 4) + 16 * i; }
      if (b_n_fast) { bn[i] = tid & 63; bk[i] = (tid >> 6) + 4 * i; } else { bk[i] = tid & 15; bn[i] = (tid >> 4) + 16 * i; }
      av[i] = A[(int64_t)min(m0 + am[i], p.M - 1) * p.a_sm + (int64_t)min(k0 + ak[i], p.K - 1) * p.a_sk];
      bv[i] = B[(int64_t)min(k0 + bk[i], p.K - 1) * p.b_sk + (int64_t)min(n0 + bn[i], p.N - 1) * p.b_sn];
    }
    __syncthreads();                                   // the previous stage's fragments have been read
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      As[ak[i]][am[i]] = (m0 + am[i] < p.M && k0 + ak[i] < p.K) ? av[i] : 0.f;
      Bs[bk[i]][bn[i]] = (n0 + bn[i] < p.N && k0 + bk[i] < p.K) ? bv[i] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BG_K; k += 2)                  // A[m = lane & 31][k + (lane >> 5)], B[k + (lane >> 5)][n = lane & 31]
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[k + hh][wm * 32 + col], Bs[k + hh][wn * 32 + col], acc, 0, 0, 0);
  }
  const int n = n0 + wn * 32 + col;
  if (n < p.N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 32 + acc_row(r, hh);
      if (m >= p.M) continue;
      float* c = C + (int64_t)m * p.c_sm + (int64_t)n * p.c_sn;
      const float v = p.alpha * acc[r];
      *c = p.beta != 0.f ? v + p.beta * *c : v;
    }
  }
}

// ---- masked softmax with the relative-key logits, one wave per (b, h, i) row ----------------------------------------------
// s[j] = scores[i][j] + [|j-i| <= w] * scale * <q[:, i], Ek[j-i+w]>;  masked (qmask[i] * kmask[j] == 0) -> fill;
// P = softmax_j(s) written in place.  q: [B, H*dk, Tq] (head h = rows h*dk ..), Ek: [Hrel, 2w+1, dk].
__global__ __launch_bounds__(256) void attn_softmax_fwd_kernel(float* __restrict__ S, const float* __restrict__ q,
                                                               const float* __restrict__ Ek, const float* __restrict__ qmask,
                                                               const float* __restrict__ kmask, int B, int H, int Tq, int Tk,
                                                               int dk, int w, int hrel, float scale, float fill) {
  __shared__ float rels[4][64];
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B * H * Tq) return;           // whole waves exit together (one row per wave)
  const int i = row % Tq, h = (row / Tq) % H, b = row / (Tq * H);
  float* s = S + (int64_t)row * Tk;
  float* relw = rels[threadIdx.x >> 6];
  float rel = 0.f;   // lane r < 2w+1 computes the logit of diagonal r - w
  if (w > 0 && lane < 2 * w + 1) {
    const float* qc = q + ((int64_t)b * H + h) * dk * Tq + i;
    const float* e = Ek + ((int64_t)(hrel > 1 ? h : 0) * (2 * w + 1) + lane) * dk;
    for (int d = 0; d < dk; ++d) rel = fmaf(qc[(int64_t)d * Tq], e[d], rel);
    rel *= scale;
  }
  relw[lane] = rel;    // same-wave LDS exchange: visible after the wave's own writes complete
  __builtin_amdgcn_wave_barrier();
  const float qm = qmask ? qmask[(int64_t)b * Tq + i] : 1.f;
  float mx = -INFINITY;
  for (int j = lane; j < Tk; j += 64) {
    float v = s[j];
    const int r = j - i + w;
    if (w > 0 && r >= 0 && r <= 2 * w) v += relw[r];
    if (qm * (kmask ? kmask[(int64_t)b * Tk + j] : 1.f) == 0.f) v = fill;
    s[j] = v;
    mx = fmaxf(mx, v);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < Tk; j += 64) { const float e = expf(s[j] - mx); s[j] = e; sum += e; }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  for (int j = lane; j < Tk; j += 64) s[j] *= inv;
}

// dS = mask * P * (dP - sum_j dP P), in place on dP
__global__ __launch_bounds__(256) void attn_softmax_bwd_kernel(float* __restrict__ dP, const float* __restrict__ P,
                                                               const float* __restrict__ qmask, const float* __restrict__ kmask,
                                                               int B, int H, int Tq, int Tk) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B * H * Tq) return;
  const int i = row % Tq, b = row / (Tq * H);
  float* d = dP + (int64_t)row * Tk;
  const float* p = P + (int64_t)row * Tk;
  float dot = 0.f;
  for (int j = lane; j < Tk; j += 64) dot = fmaf(d[j], p[j], dot);
  dot = wave_sum(dot);
  const float qm = qmask ? qmask[(int64_t)b * Tq + i] : 1.f;
  for (int j = lane; j < Tk; j += 64) {
    const float mk = qm * (kmask ? kmask[(int64_t)b * Tk + j] : 1.f);
    d[j] = mk == 0.f ? 0.f : p[j] * (d[j] - dot);
  }
}

// ---- relative-position diagonals --------------------------------------------------------------------------------------
// mode 0 (value fwd):   out[b,h,d,i] += sum_r W[i][i+r] * E[r+w][d]                      (W = dropped probabilities)
// mode 1 (score grad):  W[i][i+r]   += sum_d X[b,h,d,i] * E[r+w][d]                       (X = dOut, W = dP)
// mode 2 (emb grad):    dE[r+w][d]  += scale * sum_{b,h,i} W[i][i+r] * X[b,h,d,i]          (value: W = P, X = dOut;
//                                                                                          key:   W = dS, X = q)
// one workgroup per (b, h, block of 64 positions i)
__global__ __launch_bounds__(256) void attn_rel_kernel(float* __restrict__ W, float* __restrict__ X, float* __restrict__ E,
                                                       int B, int H, int T, int dk, int w, int hrel, float scale, int mode) {
  extern __shared__ float rel_smem[];
  const int nr = 2 * w + 1;
  float* Es = rel_smem;              // [nr][dk]
  float* Ws = rel_smem + nr * dk;    // [64][nr]
  const int i0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  float* e = E + (int64_t)(hrel > 1 ? h : 0) * nr * dk;
  float* xb = X + ((int64_t)b * H + h) * dk * T;
  float* wb = W + ((int64_t)b * H + h) * T * T;
  if (mode != 2)
    for (int t = tid; t < nr * dk; t += 256) Es[t] = e[t];
  if (mode != 1)
    for (int t = tid; t < 64 * nr; t += 256) {
      const int ii = t / nr, r = t % nr, i = i0 + ii, j = i + r - w;
      Ws[t] = (i < T && j >= 0 && j < T) ? wb[(int64_t)i * T + j] : 0.f;
    }
  __syncthreads();
  if (mode == 0) {
    for (int t = tid; t < dk * 64; t += 256) {
      const int d = t >> 6, ii = t & 63, i = i0 + ii;
      if (i >= T) continue;
      float s = 0.f;
      for (int r = 0; r < nr; ++r) s = fmaf(Ws[ii * nr + r], Es[r * dk + d], s);
      xb[(int64_t)d * T + i] += s * scale;
    }
  } else if (mode == 1) {
    for (int t = tid; t < 64 * nr; t += 256) {
      const int r = t >> 6, ii = t & 63, i = i0 + ii, j = i + r - w;
      if (i >= T || j < 0 || j >= T) continue;
      float s = 0.f;
      for (int d = 0; d < dk; ++d) s = fmaf(xb[(int64_t)d * T + i], Es[r * dk + d], s);
      wb[(int64_t)i * T + j] += s * scale;
    }
  } else {
    for (int t = tid; t < nr * dk; t += 256) {
      const int r = t / dk, d = t % dk;
      float s = 0.f;
      for (int ii = 0; ii < 64 && i0 + ii < T; ++ii) s = fmaf(Ws[ii * nr + r], xb[(int64_t)d * T + i0 + ii], s);
      atomicAdd(e + t, s * scale);
    }
  }
}

}  // namespace ttts

using namespace ttts;

extern "C" int ttts_bgemm_f32(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K, int64_t a_sm,
                              int64_t a_sk, int64_t b_sk, int64_t b_sn, int64_t c_sm, int64_t c_sn, int32_t batch_outer,
                              int32_t batch_inner, int64_t a_so, int64_t a_si, int64_t b_so, int64_t b_si, int64_t c_so,
                              int64_t c_si, float alpha, float beta, void* stream) {
  TTTS_REQUIRE(A && B && C, "bgemm: null pointer");
  TTTS_REQUIRE(M > 0 && N > 0 && K > 0 && batch_outer > 0 && batch_inner > 0, "bgemm: bad shape");
  TTTS_REQUIRE((int64_t)batch_outer * batch_inner <= 65535, "bgemm: batch too large");
  BgemmParams p{A, B, C, M, N, K, a_sm, a_sk, b_sk, b_sn, c_sm, c_sn, batch_inner, a_so, a_si, b_so, b_si, c_so, c_si, alpha, beta};
  dim3 grid((unsigned)cdiv(N, BG_T), (unsigned)cdiv(M, BG_T), (unsigned)(batch_outer * batch_inner));
  bgemm_kernel<<<grid, 256, 0, as_stream(stream)>>>(p);
  return check_launch("bgemm");
}

extern "C" int ttts_attn_softmax_fwd_f32(float* scores, const float* q, const float* emb_rel_k, const float* qmask,
                                         const float* kmask, int32_t B, int32_t H, int32_t Tq, int32_t Tk, int32_t dk,
                                         int32_t window, int32_t heads_rel, float scale, float fill, void* stream) {
  TTTS_REQUIRE(scores && B > 0 && H > 0 && Tq > 0 && Tk > 0, "attn_softmax_fwd: bad arguments");
  TTTS_REQUIRE(window == 0 || (q && emb_rel_k && Tq == Tk && 2 * window + 1 <= 64 && dk > 0), "attn_softmax_fwd: relative attention needs q, emb_rel_k, Tq == Tk and window <= 31");
  attn_softmax_fwd_kernel<<<(int)cdiv((int64_t)B * H * Tq, 4), 256, 0, as_stream(stream)>>>(scores, q, emb_rel_k, qmask, kmask, B, H, Tq, Tk,
                                                                                          dk, window, heads_rel, scale, fill);
  return check_launch("attn_softmax_fwd");
}

extern "C" int ttts_attn_softmax_bwd_f32(float* dP, const float* P, const float* qmask, const float* kmask, int32_t B,
                                         int32_t H, int32_t Tq, int32_t Tk, void* stream) {
  TTTS_REQUIRE(dP && P && B > 0 && H > 0 && Tq > 0 && Tk > 0, "attn_softmax_bwd: bad arguments");
  attn_softmax_bwd_kernel<<<(int)cdiv((int64_t)B * H * Tq, 4), 256, 0, as_stream(stream)>>>(dP, P, qmask, kmask, B, H, Tq, Tk);
  return check_launch("attn_softmax_bwd");
}

extern "C" int ttts_attn_rel_f32(float* W, float* X, float* E, int32_t B, int32_t H, int32_t T, int32_t dk,
                                 int32_t window, int32_t heads_rel, float scale, int32_t mode, void* stream) {
  TTTS_REQUIRE(W && X && E && B > 0 && H > 0 && T > 0 && dk > 0 && window > 0 && mode >= 0 && mode <= 2, "attn_rel: bad arguments");
  const int nr = 2 * window + 1;
  const size_t smem = ((size_t)nr * dk + 64 * nr) * sizeof(float);
  TTTS_REQUIRE(smem <= 64 * 1024, "attn_rel: window * d_k too large");
  attn_rel_kernel<<<dim3((unsigned)cdiv(T, 64), H, B), 256, smem, as_stream(stream)>>>(W, X, E, B, H, T, dk, window, heads_rel, scale, mode);
  return check_launch("attn_rel");
}
