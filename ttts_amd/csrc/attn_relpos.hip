// Non-causal self-attention with a T5-bucket relative-position bias, fused (flash-style: the (B, H, T, T) scores never reach HBM).
// Replaces, for the diffusion step (BASELINE config #5): QKVAttentionLegacy.forward + RelativePositionBias.forward
// (ttts/utils/utils.py:136-169,204-213; xtransformers.py:146-185) as called from AttentionBlock.forward (utils.py:204-215):
//     w = softmax((q k^T) / sqrt(ch) + bias[h][bucket(j - i)] * scale),   a = w v
// on qkv laid out (B, H, 3, ch, T) fp32 (the 1 x 1 convolution's output viewed per head; time contiguous), ch = 32.
// Before this file the step ran 2 + 4 batched GEMMs over MATERIALISED scores, a softmax pair and three bias kernels per
// attention block: 9.5 of its 25.5 ms and 28 of its 65 GB.
//
// Arithmetic: the contractions run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  NP = 3: every operand is split
// x = hi + lo (bf16 each) and a product is hi*hi + hi*lo + lo*hi -- fp32-equivalent (~2^-16 relative), what the rest of the
// default path uses; NP = 1: plain bf16 operands, the reference's autocast arithmetic (ttts/diffusion/train.py:171), used by the
// "fp8" mode of the step.  Softmax statistics, the bias, dS and all accumulators are fp32.
//
// Shape of the kernels (the S^T form of attn.hip): a workgroup (8 waves) owns (b, h, 256 queries), a wave 32 of them; ALL T keys of the
// head are staged once in LDS as bf16 [key][32 ch] rows (T <= 448 with NP = 3); S^T = K Q^T puts one query per lane, so the row
// maximum / sum are in-register reductions plus one lane <-> lane + 32 exchange and the exponentiated registers ARE the B operand
// of the next MFMA; V^T / K^T come out of the same row-major tiles through ds_read_b64_tr_b16.  The bias is a per-head table
// rel[d = key - query] in LDS (2 T floats), added to the score registers.  The backward is two kernels in the same style:
// dQ (+ delta, + the bias gradient: dS summed per diagonal -- the 32 x 32 blocks skewed in registers, one LDS atomic per lane and
// block --, one partial row per workgroup, reduced in a fixed order by a finishing kernel; the in-workgroup atomics' order is not
// fixed, so the bias gradient reproduces to fp32 summation noise, everything else bit for bit) and dK / dV ("S form": one key per lane, Q / dO of the head staged in LDS).
#include <algorithm>

#include "attn_common.hpp"

namespace ttts {

constexpr int RSTR = 40;           // bf16 elements per staged row: 32 channels + 8 (16-byte natural reads hit distinct slots)
constexpr float POS_BIG = 1.0e30f;

struct RelAttnParams {
  const float* qkv;      // (B, H, 3, 32, T)
  const float* table;    // (num_buckets, H)
  const int32_t* bucket; // [2 boff + 1]: bucket of d = key - query at d + boff
  const float* o;        // (B, H, 32, T) forward output (backward: delta)
  const float* d_o;      // (B, H, 32, T)
  float* out;            // forward
  float* lse;            // (B, H, T), log2 domain
  float* delta;          // (B, H, T)
  float* dqkv;           // (B, H, 3, 32, T)
  float* part;           // bias-gradient partials: [B H nqt][2 Tp]
  int B, H, T, Tp, nqt, boff;
  float c;               // log2(e) / sqrt(ch)
  float a;               // 1 / sqrt(ch)
  float bscale;          // RelativePositionBias.scale
  int want_dbias;
};

template <int NP>
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& h, bf16x8& l) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    h[e] = (bf16)v[e];
    l[e] = NP == 3 ? (bf16)(v[e] - (float)h[e]) : h[e];      // (NP = 1: never read)
  }
}
// hi*hi (+ hi*lo + lo*hi)
template <int NP>
__device__ __forceinline__ f32x16 mma_split(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x16 acc) {
  if (NP == 3) {
    acc = mfma32(al, bh, acc);
    acc = mfma32(ah, bl, acc);
  }
  return mfma32(ah, bh, acc);
}

// stage src (32 channels x T fp32, time contiguous) as bf16 rows [s][RSTR] (hi and, NP = 3, lo); rows T <= s < Tp are zero.
// All of a thread's loads are issued before the first conversion (a load-convert-store loop ran one ~1 us memory round trip per
// iteration: 13 of them per operand, 25 us of a workgroup's 30): MAXIT x 8 independent loads in flight per thread.
constexpr int WG_THREADS = 512;                    // 8 waves: 256 queries (keys) per workgroup
template <int NP>
__device__ __forceinline__ void stage_rows(const float* __restrict__ src, int T, int Tp, bf16* hi, bf16* lo, int tid) {
  constexpr int MAXIT = ((NP == 3 ? 448 : 896) * 4 + WG_THREADS - 1) / WG_THREADS;                 // items (8 channels x 1 position) per thread
  float v[MAXIT][8];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int i = min(tid + it * WG_THREADS, 4 * Tp - 1);
    const int cb = i / Tp, s = i - cb * Tp;
    const float* sp = src + (int64_t)(cb * 8) * T + min(s, T - 1);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[it][e] = sp[(int64_t)e * T];
  }
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int i = tid + it * WG_THREADS;
    if (i < 4 * Tp) {
      const int cb = i / Tp, s = i - cb * Tp;
      float w[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] = s < T ? v[it][e] : 0.f;
      bf16x8 h, l;
      split8<NP>(w, h, l);
      *reinterpret_cast<bf16x8*>(hi + s * RSTR + cb * 8) = h;
      if (NP == 3) *reinterpret_cast<bf16x8*>(lo + s * RSTR + cb * 8) = l;
    }
  }
}
// the per-head bias row in the exp2 domain: rel[d + Tp] = table[bucket(d)][h] * scale * log2(e), d = key - query
__device__ __forceinline__ void stage_rel(const RelAttnParams& p, int h, float* rel, int tid) {
  for (int i = tid; i < 2 * p.Tp; i += WG_THREADS) {
    const int idx = min(max(i - p.Tp + p.boff, 0), 2 * p.boff);
    rel[i] = p.table[(int64_t)p.bucket[idx] * p.H + h] * p.bscale * LOG2E;
  }
}
// B-operand fragments of a [32 ch][T] fp32 matrix at one time index per lane: frag j holds channels 16 j + 8 hh + 0..7
template <int NP>
__device__ __forceinline__ void load_col_frags(const float* __restrict__ src, int T, int t, int hh, float mul, bool valid,
                                               bf16x8 (&h)[2], bf16x8 (&l)[2]) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = valid ? src[(int64_t)(16 * j + 8 * hh + e) * T + t] * mul : 0.f;
    split8<NP>(v, h[j], l[j]);
  }
}

// -------------------------------------------------------------------------------------------------------------------------------
// forward
// -------------------------------------------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(WG_THREADS) void relattn_fwd_kernel(RelAttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char ra_smem[];
  const int Tp = p.Tp, T = p.T;
  bf16* Kh = reinterpret_cast<bf16*>(ra_smem);
  bf16* Vh = Kh + Tp * RSTR;
  bf16* Kl = Vh + Tp * RSTR;                  // NP == 3 only
  bf16* Vl = Kl + Tp * RSTR;
  float* rel = reinterpret_cast<float*>(ra_smem + (size_t)(NP == 3 ? 4 : 2) * Tp * RSTR * sizeof(bf16));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, col = lane & 31, g = lane >> 4, ip = lane & 15;
  const int bh = blockIdx.x / p.nqt, qt = blockIdx.x - bh * p.nqt, h = bh % p.H;
  const float* qb = p.qkv + (int64_t)bh * 96 * T;
  stage_rows<NP>(qb + (int64_t)32 * T, T, Tp, Kh, Kl, tid);
  stage_rows<NP>(qb + (int64_t)64 * T, T, Tp, Vh, Vl, tid);
  stage_rel(p, h, rel, tid);
  const int query = qt * 256 + wave * 32 + col, qq = min(query, T - 1);
  const bool valid = query < T;
  bf16x8 qh[2], ql[2];
  load_col_frags<NP>(qb, T, qq, hh, p.c, valid, qh, ql);
  __syncthreads();
  if (qt * 256 + wave * 32 >= T) return;       // (wave-uniform; no barrier follows)

  f32x16 ot;
#pragma unroll
  for (int r = 0; r < 16; ++r) ot[r] = 0.f;
  float m = NEG_BIG, l = 0.f;
  const int nkb = Tp / 32;
  const int k_nat = col * RSTR + hh * 8;
  const int v_tr = (4 * hh + (ip >> 2)) * RSTR + 16 * (g & 1) + 4 * (ip & 3);
  const float* relq = rel + Tp - qq;           // relq[key] = rel[key - query + Tp]
  for (int kb = 0; kb < nkb; ++kb) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bf16x8 kh = *reinterpret_cast<const bf16x8*>(&Kh[k_nat + kb * 32 * RSTR + 16 * j]);
      bf16x8 kl = kh;
      if (NP == 3) kl = *reinterpret_cast<const bf16x8*>(&Kl[k_nat + kb * 32 * RSTR + 16 * j]);
      s = mma_split<NP>(kh, kl, qh[j], ql[j], s);
    }
    const bool ragged = kb * 32 + 32 > T;
    float mx = NEG_BIG;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kb * 32 + acc_row(r, hh);
      s[r] += relq[key];
      if (ragged && key >= T) s[r] = NEG_BIG;
      mx = fmaxf(mx, s[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m, mx);
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
    m = m_new;
    l *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[r] *= alpha;
    float pv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      pv[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
      l += pv[r];
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      float pe[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) pe[e] = pv[8 * jj + e];
      bf16x8 ph, pl;
      split8<NP>(pe, ph, pl);
      const bf16* vt = &Vh[v_tr + (kb * 32 + 16 * jj) * RSTR];
      const bf16x8 vh = cat4(lds_tr_b64(vt), lds_tr_b64(vt + 8 * RSTR));
      bf16x8 vl = vh;
      if (NP == 3) {
        const bf16* vtl = &Vl[v_tr + (kb * 32 + 16 * jj) * RSTR];
        vl = cat4(lds_tr_b64(vtl), lds_tr_b64(vtl + 8 * RSTR));
      }
      ot = mma_split<NP>(vh, vl, ph, pl, ot);
    }
  }
  l += __shfl_xor(l, 32, 64);
  if (valid) {
    const float inv = 1.0f / l;
    float* op = p.out + (int64_t)bh * 32 * T + query;
#pragma unroll
    for (int r = 0; r < 16; ++r) op[(int64_t)acc_row(r, hh) * T] = ot[r] * inv;
    if (hh == 0) p.lse[(int64_t)bh * T + query] = m + __log2f(l);
  }
}

// -------------------------------------------------------------------------------------------------------------------------------
// backward, dQ (one query per lane) + delta + bias-gradient partials
// -------------------------------------------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(WG_THREADS) void relattn_bwd_dq_kernel(RelAttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char ra_smem[];
  const int Tp = p.Tp, T = p.T;
  bf16* Kh = reinterpret_cast<bf16*>(ra_smem);
  bf16* Vh = Kh + Tp * RSTR;
  bf16* Kl = Vh + Tp * RSTR;
  bf16* Vl = Kl + Tp * RSTR;
  float* rel = reinterpret_cast<float*>(ra_smem + (size_t)(NP == 3 ? 4 : 2) * Tp * RSTR * sizeof(bf16));
  float* dsum = rel + 2 * Tp;                  // [2 Tp]: sum of dS over this workgroup's (query, key) pairs with key - query = d
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, col = lane & 31, g = lane >> 4, ip = lane & 15;
  const int bh = blockIdx.x / p.nqt, qt = blockIdx.x - bh * p.nqt, h = bh % p.H;
  const float* qb = p.qkv + (int64_t)bh * 96 * T;
  stage_rows<NP>(qb + (int64_t)32 * T, T, Tp, Kh, Kl, tid);
  stage_rows<NP>(qb + (int64_t)64 * T, T, Tp, Vh, Vl, tid);
  stage_rel(p, h, rel, tid);
  for (int i = tid; i < 2 * Tp; i += WG_THREADS) dsum[i] = 0.f;
  const int query = qt * 256 + wave * 32 + col, qq = min(query, T - 1);
  const bool valid = query < T;
  bf16x8 qh[2], ql[2], gh[2], gl[2];
  load_col_frags<NP>(qb, T, qq, hh, p.c, valid, qh, ql);
  const float* dob = p.d_o + (int64_t)bh * 32 * T;
  const float* ob = p.o + (int64_t)bh * 32 * T;
  // delta = sum_c dO[c][t] O[c][t]: this lane's 16 channels, then the other half's
  float dl = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int64_t off = (int64_t)(16 * j + 8 * hh + e) * T + qq;
      v[e] = valid ? dob[off] : 0.f;
      dl += v[e] * (valid ? ob[off] : 0.f);
    }
    split8<NP>(v, gh[j], gl[j]);
  }
  dl += __shfl_xor(dl, 32, 64);
  const float lse = valid ? p.lse[(int64_t)bh * T + qq] : POS_BIG;     // invalid lanes: P = exp2(.. - BIG) = 0
  if (valid && hh == 0) p.delta[(int64_t)bh * T + query] = dl;
  __syncthreads();
  const bool active = qt * 256 + wave * 32 < T;                        // wave-uniform
  if (active) {
    f32x16 dq;
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[r] = 0.f;
    const int nkb = Tp / 32;
    const int k_nat = col * RSTR + hh * 8;
    const int k_tr = (4 * hh + (ip >> 2)) * RSTR + 16 * (g & 1) + 4 * (ip & 3);
    const float* relq = rel + Tp - qq;
    // Bias gradient: dsum[d] += dS over the diagonals d = key - query.  One LDS atomic per score (16 per lane and block) made this
    // kernel 2.5x slower than dK / dV; instead the 32 x 32 block is SKEWED in registers: for score register r the lanes of a half
    // hold 32 consecutive diagonals (k_r - t), so pulling from lane (k_r - lambda) mod 32 (ds_bpermute) puts diagonal
    // d_local = lambda (k_r >= lambda) or lambda - 32 (k_r < lambda) into lane lambda for EVERY r -- two running sums per lane.  The
    // negative half of block kb and the positive half of block kb - 1 are the same absolute diagonals, so one value per lane and
    // block goes to LDS (32 consecutive addresses from lane half 0: no conflicts), 13 atomics per wave instead of 208.
    int pull[16];
    uint32_t posmask = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kr = acc_row(r, hh);
      pull[r] = ((((kr - col) & 31) | (hh << 5)) << 2);
      posmask |= (kr >= col ? 1u : 0u) << r;
    }
    float carry = 0.f;
    float* dsw = dsum + Tp - (qt * 256 + wave * 32) + col;      // + 32 kb: diagonal (32 kb + lambda) - (wave's first query)
    for (int kb = 0; kb < nkb; ++kb) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bf16x8 kh = *reinterpret_cast<const bf16x8*>(&Kh[k_nat + kb * 32 * RSTR + 16 * j]);
        const bf16x8 vh = *reinterpret_cast<const bf16x8*>(&Vh[k_nat + kb * 32 * RSTR + 16 * j]);
        bf16x8 kl = kh, vl = vh;
        if (NP == 3) {
          kl = *reinterpret_cast<const bf16x8*>(&Kl[k_nat + kb * 32 * RSTR + 16 * j]);
          vl = *reinterpret_cast<const bf16x8*>(&Vl[k_nat + kb * 32 * RSTR + 16 * j]);
        }
        s = mma_split<NP>(kh, kl, qh[j], ql[j], s);           // S^T  = K Q^T (exp2 domain: Q carries log2(e) / sqrt(ch))
        dp = mma_split<NP>(vh, vl, gh[j], gl[j], dp);         // dP^T = V dO^T
      }
      const bool ragged = kb * 32 + 32 > T;
      float ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb * 32 + acc_row(r, hh);
        float sv = s[r] + relq[key];
        if (ragged && key >= T) sv = NEG_BIG;
        const float pr = __builtin_amdgcn_exp2f(sv - lse);
        ds[r] = pr * (dp[r] - dl);
      }
      if (p.want_dbias) {
        float pos = 0.f, neg = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(pull[r], __builtin_bit_cast(int, ds[r])));
          const bool isp = (posmask >> r) & 1u;
          pos += isp ? v : 0.f;
          neg += isp ? 0.f : v;
        }
        pos += __shfl_xor(pos, 32, 64);
        neg += __shfl_xor(neg, 32, 64);
        if (hh == 0) atomicAdd(&dsw[32 * (kb - 1)], carry + neg);   // diagonals 32 (kb - 1) + lambda - q0: complete now
        carry = pos;
      }
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        float de[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) de[e] = ds[8 * jj + e];
        bf16x8 dh_, dl_;
        split8<NP>(de, dh_, dl_);
        const bf16* kt = &Kh[k_tr + (kb * 32 + 16 * jj) * RSTR];
        const bf16x8 kth = cat4(lds_tr_b64(kt), lds_tr_b64(kt + 8 * RSTR));
        bf16x8 ktl = kth;
        if (NP == 3) {
          const bf16* ktl_p = &Kl[k_tr + (kb * 32 + 16 * jj) * RSTR];
          ktl = cat4(lds_tr_b64(ktl_p), lds_tr_b64(ktl_p + 8 * RSTR));
        }
        dq = mma_split<NP>(kth, ktl, dh_, dl_, dq);           // dQ^T += K^T dS^T
      }
    }
    if (p.want_dbias && hh == 0) atomicAdd(&dsw[32 * (nkb - 1)], carry);
    if (valid) {
      float* dqp = p.dqkv + (int64_t)bh * 96 * T + query;
#pragma unroll
      for (int r = 0; r < 16; ++r) dqp[(int64_t)acc_row(r, hh) * T] = dq[r] * p.a;
    }
  }
  if (p.want_dbias) {
    __syncthreads();
    float* pp = p.part + (int64_t)blockIdx.x * 2 * Tp;
    for (int i = tid; i < 2 * Tp; i += WG_THREADS) pp[i] = dsum[i];
  }
}

// bias-gradient finish: one workgroup per head.  tot[d] = sum over the head's workgroups (fixed order, eight loads in flight), then
// eight threads per bucket walk d in strides of eight and are combined by a fixed shuffle tree:
// dtable[bucket][h] (+)= scale * sum_{d in bucket} tot[d].  (This part is order-fixed.)
__global__ __launch_bounds__(256) void relattn_dbias_finish_kernel(RelAttnParams p, float* __restrict__ dtable, int num_buckets,
                                                                   int accumulate) {
  extern __shared__ __attribute__((aligned(16))) char ra_smem[];
  float* tot = reinterpret_cast<float*>(ra_smem);     // [2 Tp]
  const int h = blockIdx.x, Tp = p.Tp, tid = threadIdx.x;
  const int rows = p.B * p.nqt;
  for (int i = tid; i < 2 * Tp; i += 256) {
    float s = 0.f;
    int r = 0;
    // (32 requests in flight, then 8: with 16 workgroups on the chip the kernel is a chain of dependent round trips -- 14 per column
    // group at 8 in flight, 23 us for a 0.7-MB reduction; the sums are formed in the same order)
    for (; r + 32 <= rows; r += 32) {
      float v[32];
#pragma unroll
      for (int e = 0; e < 32; ++e) {
        const int rr = r + e, b = rr / p.nqt, qt = rr - b * p.nqt;
        v[e] = p.part[((int64_t)(b * p.H + h) * p.nqt + qt) * 2 * Tp + i];
      }
#pragma unroll
      for (int e = 0; e < 32; ++e) s += v[e];
    }
    for (; r + 8 <= rows; r += 8) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int rr = r + e, b = rr / p.nqt, qt = rr - b * p.nqt;
        v[e] = p.part[((int64_t)(b * p.H + h) * p.nqt + qt) * 2 * Tp + i];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[e];
    }
    for (; r < rows; ++r) {
      const int b = r / p.nqt, qt = r - b * p.nqt;
      s += p.part[((int64_t)(b * p.H + h) * p.nqt + qt) * 2 * Tp + i];
    }
    tot[i] = s;
  }
  __syncthreads();
  for (int b0 = 0; b0 < num_buckets; b0 += 32) {       // 32 buckets x 8 threads per pass
    const int bk = b0 + (tid >> 3), sub = tid & 7;
    float s = 0.f;
    for (int i = sub; i < 2 * Tp; i += 8) {
      const int d = i - Tp;
      if (d > -p.T && d < p.T && p.bucket[d + p.boff] == bk) s += tot[i];
    }
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if (sub == 0 && bk < num_buckets) {
      float* dst = dtable + (int64_t)bk * p.H + h;
      *dst = (accumulate ? *dst : 0.f) + s * p.bscale;
    }
  }
}

// -------------------------------------------------------------------------------------------------------------------------------
// backward, dK / dV (one key per lane; Q and dO of the head staged in LDS)
// -------------------------------------------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(WG_THREADS) void relattn_bwd_dkdv_kernel(RelAttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char ra_smem[];
  const int Tp = p.Tp, T = p.T;
  bf16* Qh = reinterpret_cast<bf16*>(ra_smem);
  bf16* Gh = Qh + Tp * RSTR;
  bf16* Ql = Gh + Tp * RSTR;
  bf16* Gl = Ql + Tp * RSTR;
  float* rel = reinterpret_cast<float*>(ra_smem + (size_t)(NP == 3 ? 4 : 2) * Tp * RSTR * sizeof(bf16));
  float* lse_s = rel + 2 * Tp;                 // [Tp]
  float* del_s = lse_s + Tp;                   // [Tp]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, col = lane & 31, g = lane >> 4, ip = lane & 15;
  const int bh = blockIdx.x / p.nqt, kt = blockIdx.x - bh * p.nqt, h = bh % p.H;
  const float* qb = p.qkv + (int64_t)bh * 96 * T;
  stage_rows<NP>(qb, T, Tp, Qh, Ql, tid);
  stage_rows<NP>(p.d_o + (int64_t)bh * 32 * T, T, Tp, Gh, Gl, tid);
  stage_rel(p, h, rel, tid);
  for (int i = tid; i < Tp; i += WG_THREADS) {
    lse_s[i] = i < T ? p.lse[(int64_t)bh * T + i] : POS_BIG;          // queries beyond T: P = 0
    del_s[i] = i < T ? p.delta[(int64_t)bh * T + i] : 0.f;
  }
  const int key = kt * 256 + wave * 32 + col, kk = min(key, T - 1);
  const bool valid = key < T;
  bf16x8 kh[2], kl[2], vh[2], vl[2];
  load_col_frags<NP>(qb + (int64_t)32 * T, T, kk, hh, p.c, valid, kh, kl);       // K carries log2(e) / sqrt(ch)
  load_col_frags<NP>(qb + (int64_t)64 * T, T, kk, hh, 1.0f, valid, vh, vl);
  __syncthreads();
  if (kt * 256 + wave * 32 >= T) return;
  f32x16 dk, dv;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }
  const int nqb = Tp / 32;
  const int q_nat = col * RSTR + hh * 8;
  const int q_tr = (4 * hh + (ip >> 2)) * RSTR + 16 * (g & 1) + 4 * (ip & 3);
  const float* relk = rel + Tp + kk;           // relk[-query] = rel[key - query + Tp]
  for (int qb_ = 0; qb_ < nqb; ++qb_) {
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bf16x8 qh = *reinterpret_cast<const bf16x8*>(&Qh[q_nat + qb_ * 32 * RSTR + 16 * j]);
      const bf16x8 gh = *reinterpret_cast<const bf16x8*>(&Gh[q_nat + qb_ * 32 * RSTR + 16 * j]);
      bf16x8 ql = qh, gl = gh;
      if (NP == 3) {
        ql = *reinterpret_cast<const bf16x8*>(&Ql[q_nat + qb_ * 32 * RSTR + 16 * j]);
        gl = *reinterpret_cast<const bf16x8*>(&Gl[q_nat + qb_ * 32 * RSTR + 16 * j]);
      }
      s = mma_split<NP>(qh, ql, kh[j], kl[j], s);             // S  = Q K^T   (rows: queries, column: this lane's key)
      dp = mma_split<NP>(gh, gl, vh[j], vl[j], dp);           // dP = dO V^T
    }
    float pr[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int q = qb_ * 32 + acc_row(r, hh);
      pr[r] = valid ? __builtin_amdgcn_exp2f(s[r] + relk[-q] - lse_s[q]) : 0.f;
      ds[r] = pr[r] * (dp[r] - del_s[q]);
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      float pe[8], de[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { pe[e] = pr[8 * jj + e]; de[e] = ds[8 * jj + e]; }
      bf16x8 ph, pl, dh_, dl_;
      split8<NP>(pe, ph, pl);
      split8<NP>(de, dh_, dl_);
      const bf16* gt = &Gh[q_tr + (qb_ * 32 + 16 * jj) * RSTR];
      const bf16* qt_ = &Qh[q_tr + (qb_ * 32 + 16 * jj) * RSTR];
      const bf16x8 gth = cat4(lds_tr_b64(gt), lds_tr_b64(gt + 8 * RSTR));
      const bf16x8 qth = cat4(lds_tr_b64(qt_), lds_tr_b64(qt_ + 8 * RSTR));
      bf16x8 gtl = gth, qtl = qth;
      if (NP == 3) {
        const bf16* gtp = &Gl[q_tr + (qb_ * 32 + 16 * jj) * RSTR];
        const bf16* qtp = &Ql[q_tr + (qb_ * 32 + 16 * jj) * RSTR];
        gtl = cat4(lds_tr_b64(gtp), lds_tr_b64(gtp + 8 * RSTR));
        qtl = cat4(lds_tr_b64(qtp), lds_tr_b64(qtp + 8 * RSTR));
      }
      dv = mma_split<NP>(gth, gtl, ph, pl, dv);               // dV^T += dO^T P
      dk = mma_split<NP>(qth, qtl, dh_, dl_, dk);             // dK^T += Q^T dS
    }
  }
  if (valid) {
    float* dkp = p.dqkv + (int64_t)bh * 96 * T + (int64_t)32 * T + key;
    float* dvp = dkp + (int64_t)32 * T;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      dkp[(int64_t)acc_row(r, hh) * T] = dk[r] * p.a;
      dvp[(int64_t)acc_row(r, hh) * T] = dv[r];
    }
  }
}

static int relattn_lds_bytes(int NP, int Tp, int extra_floats) {
  return (NP == 3 ? 4 : 2) * Tp * RSTR * (int)sizeof(bf16) + (2 * Tp + extra_floats) * (int)sizeof(float);
}

}  // namespace ttts

using namespace ttts;

/* T the fused kernels take with `products` split products per operand pair (LDS: all keys of a head as bf16 rows), 0 if none */
extern "C" int32_t ttts_attn_relpos_max_t(int32_t products) { return products == 3 ? 448 : (products == 1 ? 896 : 0); }

extern "C" int64_t ttts_attn_relpos_workspace_bytes(int32_t B, int32_t H, int32_t T) {
  if (B <= 0 || H <= 0 || T <= 0) return 0;
  const int64_t Tp = (T + 31) / 32 * 32, nqt = (T + 255) / 256;
  return ((int64_t)B * H * T + (int64_t)B * H * nqt * 2 * Tp) * (int64_t)sizeof(float);     // delta + bias-gradient partials
}

static int relattn_fill(RelAttnParams& p, const float* qkv, const float* table, const int32_t* bucket, int32_t bucket_off, int32_t B,
                        int32_t H, int32_t T, int32_t ch, float bias_scale, int32_t products) {
  TTTS_REQUIRE(qkv && table && bucket, "attn_relpos: null pointer");
  TTTS_REQUIRE(ch == 32, "attn_relpos: head channels must be 32 (got %d)", ch);
  TTTS_REQUIRE(products == 1 || products == 3, "attn_relpos: products must be 1 (bf16) or 3 (split bf16)");
  TTTS_REQUIRE(B > 0 && H > 0 && T > 0 && T <= ttts_attn_relpos_max_t(products), "attn_relpos: T = %d outside (0, %d]", T,
               ttts_attn_relpos_max_t(products));
  TTTS_REQUIRE(bucket_off >= T - 1, "attn_relpos: bucket table covers |d| <= %d, T = %d", bucket_off, T);
  p = RelAttnParams{};
  p.qkv = qkv; p.table = table; p.bucket = bucket; p.B = B; p.H = H; p.T = T; p.Tp = (T + 31) / 32 * 32; p.nqt = (T + 255) / 256;
  p.boff = bucket_off; p.a = 1.0f / sqrtf((float)ch); p.c = p.a * LOG2E; p.bscale = bias_scale;
  return TTTS_OK;
}

template <typename Kern>
static int relattn_opt_in(OnceFlag& f, Kern k, int bytes) {
  if (bytes <= 64 * 1024) return TTTS_OK;
  if (lds_opt_in(f, reinterpret_cast<const void*>(k)) != hipSuccess) return fail(TTTS_EHIP, "attn_relpos: dynamic LDS opt-in failed");
  return TTTS_OK;
}

extern "C" int ttts_attn_relpos_fwd_f32(const float* qkv, const float* table, const int32_t* bucket, int32_t bucket_off, float* out,
                                        float* lse, int32_t B, int32_t H, int32_t T, int32_t ch, float bias_scale, int32_t products,
                                        void* stream) {
  RelAttnParams p;
  if (int rc = relattn_fill(p, qkv, table, bucket, bucket_off, B, H, T, ch, bias_scale, products)) return rc;
  TTTS_REQUIRE(out && lse, "attn_relpos_fwd: null output");
  p.out = out; p.lse = lse;
  const int lds = relattn_lds_bytes(products, p.Tp, 0);
  static OnceFlag f1, f3;
  const dim3 grid((unsigned)(B * H * p.nqt));
  if (products == 3) {
    if (int rc = relattn_opt_in(f3, relattn_fwd_kernel<3>, lds)) return rc;
    relattn_fwd_kernel<3><<<grid, WG_THREADS, lds, as_stream(stream)>>>(p);
  } else {
    if (int rc = relattn_opt_in(f1, relattn_fwd_kernel<1>, lds)) return rc;
    relattn_fwd_kernel<1><<<grid, WG_THREADS, lds, as_stream(stream)>>>(p);
  }
  return check_launch("attn_relpos_fwd");
}

extern "C" int ttts_attn_relpos_bwd_f32(const float* qkv, const float* table, const int32_t* bucket, int32_t bucket_off, const float* out,
                                        const float* dout, const float* lse, float* dqkv, float* dtable, int32_t accumulate_dtable,
                                        void* workspace, int32_t B, int32_t H, int32_t T, int32_t ch, int32_t num_buckets,
                                        float bias_scale, int32_t products, void* stream) {
  RelAttnParams p;
  if (int rc = relattn_fill(p, qkv, table, bucket, bucket_off, B, H, T, ch, bias_scale, products)) return rc;
  TTTS_REQUIRE(out && dout && lse && dqkv && workspace, "attn_relpos_bwd: null pointer");
  TTTS_REQUIRE(num_buckets > 0 && num_buckets <= 256, "attn_relpos_bwd: 1..256 buckets");
  p.o = out; p.d_o = dout; p.lse = const_cast<float*>(lse); p.dqkv = dqkv;
  p.delta = static_cast<float*>(workspace); p.part = p.delta + (int64_t)B * H * T;
  p.want_dbias = dtable != nullptr;
  hipStream_t s = as_stream(stream);
  const dim3 grid((unsigned)(B * H * p.nqt));
  const int lds_q = relattn_lds_bytes(products, p.Tp, 2 * p.Tp), lds_k = relattn_lds_bytes(products, p.Tp, 2 * p.Tp);
  static OnceFlag fq1, fq3, fk1, fk3;
  if (products == 3) {
    if (int rc = relattn_opt_in(fq3, relattn_bwd_dq_kernel<3>, lds_q)) return rc;
    relattn_bwd_dq_kernel<3><<<grid, WG_THREADS, lds_q, s>>>(p);
  } else {
    if (int rc = relattn_opt_in(fq1, relattn_bwd_dq_kernel<1>, lds_q)) return rc;
    relattn_bwd_dq_kernel<1><<<grid, WG_THREADS, lds_q, s>>>(p);
  }
  if (int rc = check_launch("attn_relpos_bwd_dq")) return rc;
  if (products == 3) {
    if (int rc = relattn_opt_in(fk3, relattn_bwd_dkdv_kernel<3>, lds_k)) return rc;
    relattn_bwd_dkdv_kernel<3><<<grid, WG_THREADS, lds_k, s>>>(p);
  } else {
    if (int rc = relattn_opt_in(fk1, relattn_bwd_dkdv_kernel<1>, lds_k)) return rc;
    relattn_bwd_dkdv_kernel<1><<<grid, WG_THREADS, lds_k, s>>>(p);
  }
  if (int rc = check_launch("attn_relpos_bwd_dkdv")) return rc;
  if (dtable) {
    relattn_dbias_finish_kernel<<<H, 256, 2 * p.Tp * sizeof(float), s>>>(p, dtable, num_buckets, accumulate_dtable);
    return check_launch("attn_relpos_dbias_finish");
  }
  return TTTS_OK;
}
