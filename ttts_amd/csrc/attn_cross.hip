// Fused fp32 cross-attention (round 3): the text<->audio attention of MRTE (ttts/utils/vc_utils.py:568-627 via
// ttts/vqvae/vq2.py:41-43) and every other non-windowed, dropout-free use of attentions.MultiHeadAttention / vc_utils.MultiHeadAttention
// on (B, C, T) tensors -- forward and backward without the [B, H, Tq, Tk] score / probability tensors of the bgemm + softmax path
// (attn_f32.hip, which stays for the relative-position windows and for attention dropout).
//
// Exact-fp32 products on v_mfma_f32_32x32x2_f32, no LDS, no barriers: one wave per 32 queries (forward, dQ) or per 32 keys
// (dK / dV).  The (B, C, T) layout makes a head's q / k / v a [d_k][T] matrix with T contiguous, which is exactly the MFMA
// operand layout for the contractions over d (lane = (position lane & 31, d parity lane >> 5): 128-byte coalesced loads); the
// contractions over positions (P.V, dS.K, ...) read their A operand as one dword per lane from 32 different channel rows -- small
// L1 / L2-resident gathers (a head's K or V is d_k x Tk x 4 bytes <= 64 KB).
//   S^T[j][t] = scale * sum_d K[d][j] Q[d][t]      masked (qmask[t] * kmask[j] == 0) -> fill      P = softmax_j
//   O^T[d][t] = sum_j V[d][j] P^T[j][t]
// Online softmax over 32-key blocks (running max / sum per query lane); (max, 1 / sum) saved for the backward.
#include <algorithm>

#include "common.hpp"

namespace ttts {

struct CrossParams {
  const float *q, *k, *v, *qmask, *kmask;
  float* out;
  float* lse;            // [B, H, Tq][2]: running max m and sum l of a query's (masked) scores: P = exp(s - m) / l exactly as
                         // the forward formed it (m + log l in one float loses the 1 / Tk of a fully masked -1e4 row)
  const float* dout;
  float *dq, *dk, *dv;
  float* delta;          // [B, H, Tq] workspace: rowsum(dO o O), written by the dQ kernel, read by the dK / dV kernel
  int B, H, Tq, Tk;
  float scale, fill;
};

__device__ __forceinline__ float xhalf_f(float v) { return __shfl_xor(v, 32, 64); }

// forward: wave = (b, h, 32 queries)
template <int DK>
__global__ __launch_bounds__(256) void attn_cross_fwd_kernel(CrossParams p) {
  constexpr int NM = DK / 2, NDB = DK / 32;
  const int lane = threadIdx.x & 63, tl = lane & 31, hh = lane >> 5;
  const int qb = blockIdx.x * 4 + (threadIdx.x >> 6), h = blockIdx.y, b = blockIdx.z;
  if (qb * 32 >= p.Tq) return;                       // (no barriers in this kernel)
  const int t = qb * 32 + tl, tc = min(t, p.Tq - 1);
  const int64_t C = (int64_t)p.H * DK;
  const float* qh = p.q + ((int64_t)b * C + (int64_t)h * DK) * p.Tq;
  const float* kh = p.k + ((int64_t)b * C + (int64_t)h * DK) * p.Tk;
  const float* vh = p.v + ((int64_t)b * C + (int64_t)h * DK) * p.Tk;
  float qf[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m) qf[m] = qh[(int64_t)(2 * m + hh) * p.Tq + tc];
  const float qm = p.qmask ? p.qmask[(int64_t)b * p.Tq + tc] : 1.f;
  f32x16 o[NDB];
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float mrun = -INFINITY, lrun = 0.f;
  for (int j0 = 0; j0 < p.Tk; j0 += 32) {
    const int jc = min(j0 + tl, p.Tk - 1);
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    // operand batches one ahead of the MFMAs that consume them (one wave per SIMD: nothing else covers the L2 round trips)
    float kf[2][8], vg[2][16];
    auto ld_k = [&](int m0, int bsel) {
#pragma unroll
      for (int u = 0; u < 8; ++u) kf[bsel][u] = kh[(int64_t)(2 * (m0 + u) + hh) * p.Tk + jc];
    };
    auto ld_v = [&](int db, int bsel) {
#pragma unroll
      for (int r = 0; r < 16; ++r) vg[bsel][r] = vh[(int64_t)(db * 32 + tl) * p.Tk + min(j0 + acc_row(r, hh), p.Tk - 1)];
    };
    ld_k(0, 0);
#pragma unroll
    for (int m0 = 0; m0 < NM; m0 += 8) {
      const int cur = (m0 >> 3) & 1;
      if (m0 + 8 < NM) ld_k(m0 + 8, cur ^ 1);
      else ld_v(0, 0);
#pragma unroll
      for (int u = 0; u < 8; ++u) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[cur][u], qf[m0 + u], s, 0, 0, 0);
    }
    float sc[16], bm = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = j0 + acc_row(r, hh);
      const float km = (p.kmask && j < p.Tk) ? p.kmask[(int64_t)b * p.Tk + j] : 1.f;
      float v = s[r] * p.scale;
      if (qm * km == 0.f) v = p.fill;
      if (j >= p.Tk) v = -INFINITY;
      sc[r] = v;
      bm = fmaxf(bm, v);
    }
    bm = fmaxf(bm, xhalf_f(bm));
    const float mnew = fmaxf(mrun, bm);
    // (0 on the first block: mrun = -inf.  With a fill of -inf a whole block can be masked: mnew stays -inf, nothing is added)
    const float alpha = mnew == -INFINITY ? 1.f : expf(mrun - mnew);
    mrun = mnew;
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sc[r] = sc[r] == -INFINITY ? 0.f : expf(sc[r] - mnew); psum += sc[r]; }
    lrun = lrun * alpha + psum;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
      if (db + 1 < NDB) ld_v(db + 1, (db + 1) & 1);
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(vg[db & 1][r], sc[r], o[db], 0, 0, 0);
    }
  }
  lrun += xhalf_f(lrun);
  const float inv = 1.f / lrun;
  if (t < p.Tq) {
    float* oh = p.out + ((int64_t)b * C + (int64_t)h * DK) * p.Tq;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) oh[(int64_t)(db * 32 + acc_row(r, hh)) * p.Tq + t] = o[db][r] * inv;
    if (hh == 0 && p.lse) *reinterpret_cast<float2*>(p.lse + 2 * (((int64_t)b * p.H + h) * p.Tq + t)) = make_float2(mrun, inv);
  }
}

// backward, dQ: wave = (b, h, 32 queries).  Also writes delta[t] = sum_d dO[d][t] O[d][t] for the dK / dV kernel.
template <int DK>
__global__ __launch_bounds__(256) void attn_cross_bwd_dq_kernel(CrossParams p) {
  constexpr int NM = DK / 2, NDB = DK / 32;
  const int lane = threadIdx.x & 63, tl = lane & 31, hh = lane >> 5;
  const int qb = blockIdx.x * 4 + (threadIdx.x >> 6), h = blockIdx.y, b = blockIdx.z;
  if (qb * 32 >= p.Tq) return;
  const int t = qb * 32 + tl, tc = min(t, p.Tq - 1);
  const int64_t C = (int64_t)p.H * DK;
  const int64_t hq = ((int64_t)b * C + (int64_t)h * DK) * p.Tq, hk = ((int64_t)b * C + (int64_t)h * DK) * p.Tk;
  const float* qh = p.q + hq;
  const float* kh = p.k + hk;
  const float* vh = p.v + hk;
  float qf[NM], dof[NM];
  float dl = 0.f;
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const int64_t off = (int64_t)(2 * m + hh) * p.Tq + tc;
    qf[m] = qh[off];
    dof[m] = p.dout[hq + off];
    dl = fmaf(dof[m], p.out[hq + off], dl);
  }
  dl += xhalf_f(dl);
  const int64_t st = ((int64_t)b * p.H + h) * p.Tq + tc;
  const float2 ml = *reinterpret_cast<const float2*>(p.lse + 2 * st);   // (max, 1 / sum)
  if (hh == 0 && t < p.Tq) p.delta[st] = dl;
  const float qm = p.qmask ? p.qmask[(int64_t)b * p.Tq + tc] : 1.f;
  f32x16 dq[NDB];
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;
  for (int j0 = 0; j0 < p.Tk; j0 += 32) {
    const int jc = min(j0 + tl, p.Tk - 1);
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    // operand batches one ahead of the MFMAs that consume them (one wave per SIMD: nothing else covers the L2 round trips)
    float kf[2][8], vf[2][8];
    auto ld_kv = [&](int m0, int bsel) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t off = (int64_t)(2 * (m0 + u) + hh) * p.Tk + jc;
        kf[bsel][u] = kh[off];
        vf[bsel][u] = vh[off];
      }
    };
    float kg[2][16];
    auto ld_kg = [&](int db, int bsel) {
#pragma unroll
      for (int r = 0; r < 16; ++r) kg[bsel][r] = kh[(int64_t)(db * 32 + tl) * p.Tk + min(j0 + acc_row(r, hh), p.Tk - 1)];
    };
    ld_kv(0, 0);
#pragma unroll
    for (int m0 = 0; m0 < NM; m0 += 8) {
      const int cur = (m0 >> 3) & 1;
      if (m0 + 8 < NM) ld_kv(m0 + 8, cur ^ 1);
      else ld_kg(0, 0);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[cur][u], qf[m0 + u], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[cur][u], dof[m0 + u], dp, 0, 0, 0);
      }
    }
    float ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = j0 + acc_row(r, hh);
      const float km = (p.kmask && j < p.Tk) ? p.kmask[(int64_t)b * p.Tk + j] : 1.f;
      const bool masked = qm * km == 0.f;
      const float v = masked ? p.fill : s[r] * p.scale;
      const float pr = j < p.Tk ? expf(v - ml.x) * ml.y : 0.f;
      ds[r] = masked ? 0.f : pr * (dp[r] - dl);      // masked_fill: no gradient reaches a masked score
    }
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
      if (db + 1 < NDB) ld_kg(db + 1, (db + 1) & 1);
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(kg[db & 1][r], ds[r], dq[db], 0, 0, 0);
    }
  }
  if (t < p.Tq) {
    float* dqh = p.dq + hq;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) dqh[(int64_t)(db * 32 + acc_row(r, hh)) * p.Tq + t] = dq[db][r] * p.scale;
  }
}

// backward, dK / dV: workgroup = (b, h, 32 keys); S form (lane = key, accumulator rows = queries).  Its four waves take every
// fourth 32-query block and add their partial dK / dV through LDS in a fixed order (wave 3 -> 1, 2 -> 0, then 1 -> 0): four times
// the waves of a wave-per-key-block form, whose 8 x 256 dependent MFMAs per wave were the backward's long pole.
template <int DK>
__global__ __launch_bounds__(256) void attn_cross_bwd_dkdv_kernel(CrossParams p) {
  constexpr int NM = DK / 2, NDB = DK / 32;
  __shared__ float red[2][2 * NDB * 16][64];           // two slots of [dk | dv][register][lane]
  const int lane = threadIdx.x & 63, tl = lane & 31, hh = lane >> 5, wave = threadIdx.x >> 6;
  const int kb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int j = kb * 32 + tl, jc = min(j, p.Tk - 1);
  const int64_t C = (int64_t)p.H * DK;
  const int64_t hq = ((int64_t)b * C + (int64_t)h * DK) * p.Tq, hk = ((int64_t)b * C + (int64_t)h * DK) * p.Tk;
  const float* qh = p.q + hq;
  const float* doh = p.dout + hq;
  float kf[NM], vf[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const int64_t off = (int64_t)(2 * m + hh) * p.Tk + jc;
    kf[m] = p.k[hk + off];
    vf[m] = p.v[hk + off];
  }
  const float km = p.kmask ? p.kmask[(int64_t)b * p.Tk + jc] : 1.f;
  f32x16 dkk[NDB], dvv[NDB];
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkk[db][r] = 0.f; dvv[db][r] = 0.f; }
  const int64_t stat = ((int64_t)b * p.H + h) * p.Tq;
  for (int t0 = wave * 32; t0 < p.Tq; t0 += 128) {
    const int tcl = min(t0 + tl, p.Tq - 1);
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int m0 = 0; m0 < NM; m0 += 8) {           // (a one-ahead prefetch as in the dQ kernel spills here: k, v and both
      float qa[8], da[8];                          // gradients are 256 live registers already)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t off = (int64_t)(2 * (m0 + u) + hh) * p.Tq + tcl;
        qa[u] = qh[off];
        da[u] = doh[off];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[u], kf[m0 + u], s, 0, 0, 0);      // S[t][j]
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(da[u], vf[m0 + u], dp, 0, 0, 0);    // dP[t][j]
      }
    }
    float pr[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = t0 + acc_row(r, hh), tcc = min(t, p.Tq - 1);
      const float qm = p.qmask ? p.qmask[(int64_t)b * p.Tq + tcc] : 1.f;
      const bool masked = qm * km == 0.f;
      const float v = masked ? p.fill : s[r] * p.scale;
      const float2 ml = *reinterpret_cast<const float2*>(p.lse + 2 * (stat + tcc));
      const float pe = (t < p.Tq && j < p.Tk) ? expf(v - ml.x) * ml.y : 0.f;
      pr[r] = pe;
      ds[r] = masked ? 0.f : pe * (dp[r] - p.delta[stat + tcc]);
    }
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
      float qg[16], dg[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t off = (int64_t)(db * 32 + tl) * p.Tq + min(t0 + acc_row(r, hh), p.Tq - 1);
        qg[r] = qh[off];
        dg[r] = doh[off];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        dvv[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(dg[r], pr[r], dvv[db], 0, 0, 0);   // dV^T[d][j] += dO[d][t] P[t][j]
        dkk[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(qg[r], ds[r], dkk[db], 0, 0, 0);   // dK^T[d][j] += Q[d][t] dS[t][j]
      }
    }
  }
  auto put = [&](int slot) {
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) { red[slot][db * 16 + r][lane] = dkk[db][r]; red[slot][(NDB + db) * 16 + r][lane] = dvv[db][r]; }
  };
  auto get = [&](int slot) {
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dkk[db][r] += red[slot][db * 16 + r][lane]; dvv[db][r] += red[slot][(NDB + db) * 16 + r][lane]; }
  };
  if (wave >= 2) put(wave - 2);
  __syncthreads();
  if (wave < 2) get(wave);
  __syncthreads();
  if (wave == 1) put(0);
  __syncthreads();
  if (wave == 0) {
    get(0);
    if (j < p.Tk) {
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t off = hk + (int64_t)(db * 32 + acc_row(r, hh)) * p.Tk + j;
          p.dk[off] = dkk[db][r] * p.scale;
          p.dv[off] = dvv[db][r];
        }
    }
  }
}

}  // namespace ttts

using namespace ttts;

static int cross_check(const CrossParams& p, int dk, const char* who) {
  TTTS_REQUIRE(p.q && p.k && p.v, "%s: null pointer", who);
  TTTS_REQUIRE(p.B > 0 && p.H > 0 && p.Tq > 0 && p.Tk > 0, "%s: bad shape", who);
  if (dk != 64 && dk != 96 && dk != 128) return fail(TTTS_EUNSUPPORTED, "%s: d_k must be 64, 96 or 128 (got %d)", who, dk);
  return TTTS_OK;
}

extern "C" int ttts_attn_cross_fwd_f32(const float* q, const float* k, const float* v, const float* qmask, const float* kmask,
                                       float* out, float* lse, int32_t B, int32_t H, int32_t dk, int32_t Tq, int32_t Tk,
                                       float scale, float fill, void* stream) {
  CrossParams p{};
  p.q = q; p.k = k; p.v = v; p.qmask = qmask; p.kmask = kmask; p.out = out; p.lse = lse;
  p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk; p.scale = scale; p.fill = fill;
  int rc = cross_check(p, dk, "attn_cross_fwd");
  if (rc) return rc;
  TTTS_REQUIRE(out, "attn_cross_fwd: null output");
  const dim3 grid((unsigned)cdiv(Tq, 128), (unsigned)H, (unsigned)B);
  hipStream_t s = as_stream(stream);
  if (dk == 128) attn_cross_fwd_kernel<128><<<grid, 256, 0, s>>>(p);
  else if (dk == 96) attn_cross_fwd_kernel<96><<<grid, 256, 0, s>>>(p);
  else attn_cross_fwd_kernel<64><<<grid, 256, 0, s>>>(p);
  return check_launch("attn_cross_fwd");
}

// (max, 1 / sum) pairs
extern "C" int64_t ttts_attn_cross_stats_bytes(int32_t B, int32_t H, int32_t Tq) { return (int64_t)B * H * Tq * 2 * (int64_t)sizeof(float); }

extern "C" int64_t ttts_attn_cross_bwd_workspace_bytes(int32_t B, int32_t H, int32_t Tq) { return (int64_t)B * H * Tq * (int64_t)sizeof(float); }

extern "C" int ttts_attn_cross_bwd_f32(const float* q, const float* k, const float* v, const float* qmask, const float* kmask,
                                       const float* out, const float* dout, const float* lse, float* dq, float* dk_out, float* dv,
                                       void* workspace, int32_t B, int32_t H, int32_t dk, int32_t Tq, int32_t Tk, float scale,
                                       float fill, void* stream) {
  CrossParams p{};
  p.q = q; p.k = k; p.v = v; p.qmask = qmask; p.kmask = kmask; p.out = const_cast<float*>(out); p.lse = const_cast<float*>(lse);
  p.dout = dout; p.dq = dq; p.dk = dk_out; p.dv = dv; p.delta = reinterpret_cast<float*>(workspace);
  p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk; p.scale = scale; p.fill = fill;
  int rc = cross_check(p, dk, "attn_cross_bwd");
  if (rc) return rc;
  TTTS_REQUIRE(out && dout && lse && dq && dk_out && dv && workspace, "attn_cross_bwd: null pointer");
  hipStream_t s = as_stream(stream);
  const dim3 gq((unsigned)cdiv(Tq, 128), (unsigned)H, (unsigned)B), gk((unsigned)cdiv(Tk, 32), (unsigned)H, (unsigned)B);
#define CROSS_BWD(D_)                                             \
  attn_cross_bwd_dq_kernel<D_><<<gq, 256, 0, s>>>(p);            \
  rc = check_launch("attn_cross_bwd_dq");                        \
  if (rc) return rc;                                              \
  attn_cross_bwd_dkdv_kernel<D_><<<gk, 256, 0, s>>>(p);
  if (dk == 128) { CROSS_BWD(128) } else if (dk == 96) { CROSS_BWD(96) } else { CROSS_BWD(64) }
#undef CROSS_BWD
  return check_launch("attn_cross_bwd_dkdv");
}
