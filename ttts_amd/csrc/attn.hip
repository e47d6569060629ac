// Causal self-attention forward / backward for gfx950 (flash-style: the S x S score matrix never reaches HBM): host entry
// points and the generic kernels (head_dim 32 / 128).  head_dim 64 -- the GPT train step -- is served by attn_dh64.hip.
// Replaces GPT2Attention's core (transformers modeling_gpt2.py:53-72) as reached from ttts/gpt/model.py:422.
//
// MFMA: v_mfma_f32_32x32x16_bf16.  All three kernels keep the softmax statistics lane-local by choosing which
// operand is "transposed":
//   forward / dQ kernels ("S^T form"): S^T = K.Q^T puts ONE QUERY PER LANE (column) and 16 keys of each 32-key
//     block in that lane's accumulator registers, so row max / row sum are in-register reductions plus one
//     lane<->lane+32 exchange, and the exponentiated registers ARE the B operand of the next MFMA
//     (O^T = V^T.P^T, dQ^T = K^T.dS^T) -- P never touches LDS.  The A operand of that MFMA (V^T / K^T) is fetched
//     from the row-major [key][dh] LDS tile with the hardware transpose read ds_read_b64_tr_b16.
//   dK/dV kernel ("S form"): S = Q.K^T puts ONE KEY PER LANE, so dV^T = dO^T.P and dK^T = Q^T.dS again consume
//     the accumulator registers directly as B operands; the per-query lse / delta come from a small LDS array.
// Work split: 4 waves x 32 rows (queries resp. keys) per workgroup, 64-row tiles of the other sequence axis
// staged global -> registers -> LDS (double buffered, next tile's loads in flight under the MFMAs).
// Causality: tiles strictly above the diagonal are never loaded; workgroups are launched heaviest-first.
#include <algorithm>
#include <type_traits>

#include "attn_common.hpp"

namespace ttts {

int attn_fwd_dh64(const AttnParams& p, hipStream_t s);   // attn_dh64.hip
int attn_bwd_dkdv_dh64(const AttnParams& p, hipStream_t s);
int attn_bwd_dq_dh64(const AttnParams& p, hipStream_t s);

// load / store a 64 x DH bf16 tile (rows row0.. of a [*, stride] matrix).  Rows beyond nrows: the load is unconditional from the
// last valid row (a guarded load compiles to an exec-masked branch per 16-byte piece -- 8 branches per loop iteration).  With
// ZERO = false the duplicate row is left in place -- K / V rows beyond the sequence only meet probabilities that are masked to
// exactly 0.  ZERO = true selects zeros at the load (the dK / dV kernel's Q / dO tiles: their rows beyond the sequence must be 0).
template <int DH, bool ZERO = true>
__device__ __forceinline__ void tile_load(bf16x8 (&r)[AttnCfg<DH>::CPT], const bf16* base, int64_t stride, int row0,
                                          int nrows, int tid) {
  if (ZERO && row0 + 64 <= nrows) {   // whole tile inside the sequence (workgroup-uniform): no clamp, no zero select
#pragma unroll
    for (int i = 0; i < AttnCfg<DH>::CPT; ++i) {
      const int c = tid + i * 256, row = c / AttnCfg<DH>::CPR, dc = (c % AttnCfg<DH>::CPR) * 8;
      r[i] = *reinterpret_cast<const bf16x8*>(base + (int64_t)(row0 + row) * stride + dc);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < AttnCfg<DH>::CPT; ++i) {
    const int c = tid + i * 256, row = c / AttnCfg<DH>::CPR, dc = (c % AttnCfg<DH>::CPR) * 8;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(base + (int64_t)min(row0 + row, nrows - 1) * stride + dc);
    r[i] = (!ZERO || row0 + row < nrows) ? v : zero8();
  }
}
template <int DH, int STR>
__device__ __forceinline__ void tile_store(const bf16x8 (&r)[AttnCfg<DH>::CPT], bf16* lds, int tid) {
#pragma unroll
  for (int i = 0; i < AttnCfg<DH>::CPT; ++i) {
    const int c = tid + i * 256, row = c / AttnCfg<DH>::CPR, dc = (c % AttnCfg<DH>::CPR) * 8;
    *reinterpret_cast<bf16x8*>(lds + row * STR + dc) = r[i];
  }
}

// -------------------------------------------------------------------------------------------------------
// forward
// -------------------------------------------------------------------------------------------------------
template <int DH, bool DROPOUT>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnParams p) {
  using C = AttnCfg<DH>;
  __shared__ __attribute__((aligned(16))) bf16 Ks[2][64 * C::KSTR];
  __shared__ __attribute__((aligned(16))) bf16 Vs[2][64 * C::VSTR];
  __shared__ __attribute__((aligned(16))) uint32_t Bm[2][64];   // dropout column multipliers of the staged key tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, g = lane >> 4, ip = lane & 15;
  const int nqb = (p.S + 127) / 128;
  // global longest-first order: ALL (b, h) pairs' heaviest (latest) query blocks are dispatched first, the 2-tile blocks
  // last, so the tail of the launch is made of short blocks (causal work per block: 2 .. 19 KV tiles)
  const int nbh = gridDim.x / nqb;
  const int qb = nqb - 1 - (int)(blockIdx.x / nbh);
  const int bh = blockIdx.x % nbh, h = bh % p.H, b = bh / p.H;
  const int q_base = qb * 128 + wave * 32;
  const int query = q_base + (lane & 31);
  const bf16* qp = p.q + (int64_t)b * p.sb + h * DH;
  const bf16* kp = p.k + (int64_t)b * p.sb + h * DH;
  const bf16* vp = p.v + (int64_t)b * p.sb + h * DH;

  bf16x8 qf[C::KS];
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks)
    qf[ks] = query < p.S ? *reinterpret_cast<const bf16x8*>(qp + (int64_t)query * p.ss + ks * 16 + hh * 8) : zero8();

  f32x16 ot[C::NB];
#pragma unroll
  for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[nb][r] = 0.f;
  float m = NEG_BIG, l = 0.f;

  const int kv_end = min(p.S, qb * 128 + 128);
  const int nt = (kv_end + 63) / 64;
  bf16x8 rk[C::CPT], rv[C::CPT];
  tile_load<DH, false>(rk, kp, p.ss, 0, p.S, tid);
  tile_load<DH, false>(rv, vp, p.ss, 0, p.S, tid);
  tile_store<DH, C::KSTR>(rk, Ks[0], tid);
  tile_store<DH, C::VSTR>(rv, Vs[0], tid);
  const uint32_t shi = DROPOUT ? seed_mix(p.seed_hi, p.ctr) : 0u;
  const uint32_t id_bh = (uint32_t)((b * p.H + h) * p.S), thr32 = p.thr << 16;
  const uint32_t rowh = DROPOUT ? drop_row_hash(id_bh + (uint32_t)query, p.seed_lo, shi) : 0u;
  if (DROPOUT && tid < 64) Bm[0][tid] = drop_col_mult(id_bh + (uint32_t)tid, p.seed_lo, shi);
  __syncthreads();
  const int k_nat = (lane & 31) * C::KSTR + hh * 8;
  const int v_tr = (4 * hh + (ip >> 2)) * C::VSTR + 16 * (g & 1) + 4 * (ip & 3);

  for (int jt = 0; jt < nt; ++jt) {
    const int buf = jt & 1, kv0 = jt * 64;
    if (jt + 1 < nt) {
      tile_load<DH, false>(rk, kp, p.ss, kv0 + 64, p.S, tid);
      tile_load<DH, false>(rv, vp, p.ss, kv0 + 64, p.S, tid);
    }
    if (kv0 <= q_base + 31) {  // wave-uniform: at least one (query, key) pair of this wave is unmasked
      f32x16 s[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&Ks[buf][k_nat + kb * 32 * C::KSTR + ks * 16]);
          s[kb] = mfma32(kf, qf[ks], s[kb]);
        }
      }
      if (kv0 + 63 > q_base || kv0 + 63 >= p.S) {  // diagonal / ragged tile: mask
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kv0 + kb * 32 + acc_row(r, hh);
            if (key > query || key >= p.S) s[kb][r] = NEG_BIG;
          }
      }
      float mx = NEG_BIG;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m, mx);
      const float alpha = __builtin_amdgcn_exp2f((m - m_new) * p.c);
      m = m_new;
      const float m2 = m_new * p.c;
      l *= alpha;
#pragma unroll
      for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[nb][r] *= alpha;
      bf16x8 pf[2][2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        u32x4_t pw[2];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          bool keep[4] = {true, true, true, true};
          if (DROPOUT) drop_keep4(rowh, &Bm[buf][kb * 32 + 8 * qd + 4 * hh], thr32, keep);
          float pv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            pv[e] = __builtin_amdgcn_exp2f(fmaf(s[kb][4 * qd + e], p.c, -m2));
            l += pv[e];
            if (DROPOUT) pv[e] = keep[e] ? pv[e] : 0.f;    // the 1 / keep factor is applied once, with 1 / l, at the end
          }
          pw[qd >> 1][2 * (qd & 1)] = pack_bf16x2(pv[0], pv[1]);
          pw[qd >> 1][2 * (qd & 1) + 1] = pack_bf16x2(pv[2], pv[3]);
        }
        pf[kb][0] = __builtin_bit_cast(bf16x8, pw[0]);
        pf[kb][1] = __builtin_bit_cast(bf16x8, pw[1]);
      }
#pragma unroll
      for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int cs = 0; cs < 2; ++cs) {
            const bf16* vt = &Vs[buf][v_tr + (kb * 32 + 16 * cs) * C::VSTR + nb * 32];
            const bf16x8 vf = cat4(lds_tr_b64(vt), lds_tr_b64(vt + 8 * C::VSTR));
            ot[nb] = mfma32(vf, pf[kb][cs], ot[nb]);
          }
    }
    if (jt + 1 < nt) {
      tile_store<DH, C::KSTR>(rk, Ks[buf ^ 1], tid);
      tile_store<DH, C::VSTR>(rv, Vs[buf ^ 1], tid);
      if (DROPOUT && tid < 64) Bm[buf ^ 1][tid] = drop_col_mult(id_bh + (uint32_t)(kv0 + 64 + tid), p.seed_lo, shi);
    }
    __syncthreads();
  }
  l += __shfl_xor(l, 32, 64);
  if (query < p.S) {
    const float inv = (DROPOUT ? p.inv_keep : 1.0f) / l;
    bf16* op = p.out + (int64_t)b * p.osb + (int64_t)query * p.oss + h * DH;
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)(ot[nb][4 * qd + e] * inv);
        *reinterpret_cast<bf16x4*>(op + nb * 32 + 8 * qd + 4 * hh) = o;
      }
    if (hh == 0) p.lse[(int64_t)(b * p.H + h) * p.S + query] = (m * p.c + __log2f(l)) * LN2;
  }
}

// -------------------------------------------------------------------------------------------------------
// forward, second work split: the one launched when dropout is on.
// Why: with 128-query workgroups the launch is 640 blocks of 2..19 KV tiles and is bound by its longest blocks (19 x ~3.9 k
// cycles), not by the matrix cores.  Here a workgroup owns 64 queries and walks 128 keys per iteration: wave (qh, kh) takes
// query half qh and the kh-th 64-key tile with its own online-softmax state; the two key halves of a query half are merged
// through LDS once at the end.  1216 blocks of <= 10 iterations instead of 640 of <= 19; the per-wave tile code is the
// kernel above.  K / V super tiles are single-buffered in LDS (43 KB at d_h 64) behind a register prefetch.
// -------------------------------------------------------------------------------------------------------
template <int DH, int ROWS>
__device__ __forceinline__ void tile_load_rows(bf16x8 (&r)[ROWS * (DH / 8) / 256], const bf16* base, int64_t stride, int row0,
                                               int nrows, int tid) {
  // K / V super tile: unconditional loads, rows beyond the sequence repeat the last one (see tile_load)
#pragma unroll
  for (int i = 0; i < ROWS * (DH / 8) / 256; ++i) {
    const int c = tid + i * 256, row = c / (DH / 8), dc = (c % (DH / 8)) * 8;
    r[i] = *reinterpret_cast<const bf16x8*>(base + (int64_t)min(row0 + row, nrows - 1) * stride + dc);
  }
}
template <int DH, int ROWS, int STR>
__device__ __forceinline__ void tile_store_rows(const bf16x8 (&r)[ROWS * (DH / 8) / 256], bf16* lds, int tid) {
#pragma unroll
  for (int i = 0; i < ROWS * (DH / 8) / 256; ++i) {
    const int c = tid + i * 256, row = c / (DH / 8), dc = (c % (DH / 8)) * 8;
    *reinterpret_cast<bf16x8*>(lds + row * STR + dc) = r[i];
  }
}

template <int DH, bool DROPOUT>
__global__ __launch_bounds__(256, (DH <= 64 ? 3 : 1)) void attn_fwd_kv2_kernel(AttnParams p) {
  using C = AttnCfg<DH>;
  constexpr int CPT2 = 128 * (DH / 8) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char kv2_smem[];
  bf16* Ks = reinterpret_cast<bf16*>(kv2_smem);                 // [128][KSTR]
  bf16* Vs = Ks + 128 * C::KSTR;                                // [128][VSTR]
  uint32_t* Bm = reinterpret_cast<uint32_t*>(Vs + 128 * C::VSTR);   // [128] dropout column multipliers of the super tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qh = wave & 1, kh = wave >> 1;
  const int hh = lane >> 5, g = lane >> 4, ip = lane & 15;
  const int nqb = (p.S + 63) / 64;
  const int nbh = gridDim.x / nqb;
  const int qb = nqb - 1 - (int)(blockIdx.x / nbh);             // longest blocks first
  const int bh = blockIdx.x % nbh, h = bh % p.H, b = bh / p.H;
  const int q0 = qb * 64;
  const int q_base = q0 + qh * 32;
  const int query = q_base + (lane & 31);
  const bf16* qp = p.q + (int64_t)b * p.sb + h * DH;
  const bf16* kp = p.k + (int64_t)b * p.sb + h * DH;
  const bf16* vp = p.v + (int64_t)b * p.sb + h * DH;

  bf16x8 qf[C::KS];
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks)
    qf[ks] = query < p.S ? *reinterpret_cast<const bf16x8*>(qp + (int64_t)query * p.ss + ks * 16 + hh * 8) : zero8();
  f32x16 ot[C::NB];
#pragma unroll
  for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[nb][r] = 0.f;
  float m = NEG_BIG, l = 0.f;

  const int kv_end = min(p.S, q0 + 64);
  const int nsup = (kv_end + 127) / 128;
  bf16x8 rk[CPT2], rv[CPT2];
  tile_load_rows<DH, 128>(rk, kp, p.ss, 0, p.S, tid);
  tile_load_rows<DH, 128>(rv, vp, p.ss, 0, p.S, tid);
  const int k_nat = (lane & 31) * C::KSTR + hh * 8;
  const int v_tr = (4 * hh + (ip >> 2)) * C::VSTR + 16 * (g & 1) + 4 * (ip & 3);
  const uint32_t shi = DROPOUT ? seed_mix(p.seed_hi, p.ctr) : 0u;
  const uint32_t id_bh = (uint32_t)((b * p.H + h) * p.S), thr32 = p.thr << 16;
  const uint32_t rowh = DROPOUT ? drop_row_hash(id_bh + (uint32_t)query, p.seed_lo, shi) : 0u;

  for (int js = 0; js < nsup; ++js) {
    __syncthreads();                                            // every wave is done reading the previous super tile
    tile_store_rows<DH, 128, C::KSTR>(rk, Ks, tid);
    tile_store_rows<DH, 128, C::VSTR>(rv, Vs, tid);
    if (DROPOUT && tid < 128) Bm[tid] = drop_col_mult(id_bh + (uint32_t)(js * 128 + tid), p.seed_lo, shi);
    __syncthreads();
    if (js + 1 < nsup) {                                        // next super tile's loads fly under this one's MFMAs
      tile_load_rows<DH, 128>(rk, kp, p.ss, (js + 1) * 128, p.S, tid);
      tile_load_rows<DH, 128>(rv, vp, p.ss, (js + 1) * 128, p.S, tid);
    }
    const int kv0 = js * 128 + kh * 64;                         // this wave's 64-key tile
    if (kv0 <= q_base + 31 && kv0 < p.S) {                      // wave-uniform: some (query, key) pair is unmasked
      const bf16* Kt = Ks + kh * 64 * C::KSTR;
      const bf16* Vt = Vs + kh * 64 * C::VSTR;
      f32x16 s[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&Kt[k_nat + kb * 32 * C::KSTR + ks * 16]);
          s[kb] = mfma32(kf, qf[ks], s[kb]);
        }
      }
      if (kv0 + 63 > q_base || kv0 + 63 >= p.S) {               // diagonal / ragged tile: mask
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kv0 + kb * 32 + acc_row(r, hh);
            if (key > query || key >= p.S) s[kb][r] = NEG_BIG;
          }
      }
      float mx = NEG_BIG;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m, mx);
      const float alpha = __builtin_amdgcn_exp2f((m - m_new) * p.c);
      m = m_new;
      const float m2 = m_new * p.c;
      l *= alpha;
#pragma unroll
      for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[nb][r] *= alpha;
      bf16x8 pf[2][2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        u32x4_t pw[2];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          bool keep[4] = {true, true, true, true};
          if (DROPOUT) drop_keep4(rowh, &Bm[kh * 64 + kb * 32 + 8 * qd + 4 * hh], thr32, keep);
          float pv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            pv[e] = __builtin_amdgcn_exp2f(fmaf(s[kb][4 * qd + e], p.c, -m2));
            l += pv[e];
            if (DROPOUT) pv[e] = keep[e] ? pv[e] : 0.f;    // the 1 / keep factor is applied once, with 1 / l, at the end
          }
          pw[qd >> 1][2 * (qd & 1)] = pack_bf16x2(pv[0], pv[1]);
          pw[qd >> 1][2 * (qd & 1) + 1] = pack_bf16x2(pv[2], pv[3]);
        }
        pf[kb][0] = __builtin_bit_cast(bf16x8, pw[0]);
        pf[kb][1] = __builtin_bit_cast(bf16x8, pw[1]);
      }
#pragma unroll
      for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int cs = 0; cs < 2; ++cs) {
            const bf16* vt = &Vt[v_tr + (kb * 32 + 16 * cs) * C::VSTR + nb * 32];
            const bf16x8 vf = cat4(lds_tr_b64(vt), lds_tr_b64(vt + 8 * C::VSTR));
            ot[nb] = mfma32(vf, pf[kb][cs], ot[nb]);
          }
    }
  }
  // merge the two key halves of each query half: wave (qh, 1) hands (m, l, O) to wave (qh, 0), lane for lane
  __syncthreads();
  float* mb = reinterpret_cast<float*>(kv2_smem) + (size_t)(qh * 64 + lane) * (2 + C::NB * 16);
  if (kh == 1) {
    mb[0] = m; mb[1] = l;
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mb[2 + nb * 16 + r] = ot[nb][r];
  }
  __syncthreads();
  if (kh == 1) return;
  {
    const float m_o = mb[0], l_o = mb[1];
    const float M = fmaxf(m, m_o);
    const float a1 = __builtin_amdgcn_exp2f((m - M) * p.c), a2 = __builtin_amdgcn_exp2f((m_o - M) * p.c);
    l = l * a1 + l_o * a2;
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[nb][r] = ot[nb][r] * a1 + mb[2 + nb * 16 + r] * a2;
    m = M;
  }
  l += __shfl_xor(l, 32, 64);
  if (query < p.S) {
    const float inv = (DROPOUT ? p.inv_keep : 1.0f) / l;
    bf16* op = p.out + (int64_t)b * p.osb + (int64_t)query * p.oss + h * DH;
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)(ot[nb][4 * qd + e] * inv);
        *reinterpret_cast<bf16x4*>(op + nb * 32 + 8 * qd + 4 * hh) = o;
      }
    if (hh == 0) p.lse[(int64_t)(b * p.H + h) * p.S + query] = (m * p.c + __log2f(l)) * LN2;
  }
}

// -------------------------------------------------------------------------------------------------------
// backward: delta = rowsum(dO * O)
// -------------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnParams p) {
  // DH / 8 lanes per (b, s, h) row, 16 bytes each: a wave instruction reads 1 KB of consecutive heads / positions (one thread
  // per row read 128-byte rows 128 bytes apart per lane -- 9 us for 19 MB); the lane group is summed with xor shuffles
  constexpr int LPR = DH / 8;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t i = t / LPR;                                  // (b, s, h)
  const int c = (int)(t % LPR);
  const int64_t total = (int64_t)p.B * p.S * p.H;
  const bool live = i < total;
  const int64_t ic = live ? i : total - 1;                    // whole waves stay converged for the shuffles
  const int h = (int)(ic % p.H);
  const int s = (int)((ic / p.H) % p.S);
  const int b = (int)(ic / ((int64_t)p.H * p.S));
  const bf16* o = p.o + (int64_t)b * p.osb + (int64_t)s * p.oss + h * DH + c * 8;
  const bf16* d = p.d_o + (int64_t)b * p.osb + (int64_t)s * p.oss + h * DH + c * 8;
  const bf16x8 ov = *reinterpret_cast<const bf16x8*>(o);
  const bf16x8 dv = *reinterpret_cast<const bf16x8*>(d);
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) acc += (float)ov[j] * (float)dv[j];
#pragma unroll
  for (int m = 1; m < LPR; m <<= 1) acc += __shfl_xor(acc, m, 64);
  if (live && c == 0) p.delta[(int64_t)(b * p.H + h) * p.S + s] = acc;
}

// -------------------------------------------------------------------------------------------------------
// backward: dQ (S^T form, one query per lane; loops over key tiles up to the diagonal)
// -------------------------------------------------------------------------------------------------------
template <int DH, bool DROPOUT>
__global__ __launch_bounds__(256, (DH <= 64 ? 3 : 2)) void attn_bwd_dq_kernel(AttnParams p) {
  using C = AttnCfg<DH>;
  __shared__ __attribute__((aligned(16))) bf16 Ks[2][64 * C::KSTR];
  __shared__ __attribute__((aligned(16))) bf16 Vs[2][64 * C::KSTR];
  __shared__ __attribute__((aligned(16))) uint32_t Bm[2][64];   // dropout column multipliers of the staged key tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, g = lane >> 4, ip = lane & 15;
  const int nqb = (p.S + 127) / 128;
  const int nbh = gridDim.x / nqb;
  const int qb = nqb - 1 - (int)(blockIdx.x / nbh);   // global longest-first order (see attn_fwd_kernel)
  const int bh = blockIdx.x % nbh, h = bh % p.H, b = bh / p.H;
  const int q_base = qb * 128 + wave * 32;
  const int query = q_base + (lane & 31);
  const int q_first = __builtin_amdgcn_readfirstlane(q_base);   // scalar copy for wave-uniform branches
  const bf16* qp = p.q + (int64_t)b * p.sb + h * DH;
  const bf16* kp = p.k + (int64_t)b * p.sb + h * DH;
  const bf16* vp = p.v + (int64_t)b * p.sb + h * DH;
  const bf16* dop = p.d_o + (int64_t)b * p.osb + h * DH;

  bf16x8 qf[C::KS], dof[C::KS];
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) {
    qf[ks] = query < p.S ? *reinterpret_cast<const bf16x8*>(qp + (int64_t)query * p.ss + ks * 16 + hh * 8) : zero8();
    dof[ks] = query < p.S ? *reinterpret_cast<const bf16x8*>(dop + (int64_t)query * p.oss + ks * 16 + hh * 8) : zero8();
  }
  const int64_t stat = (int64_t)(b * p.H + h) * p.S + query;
  const float lse2 = query < p.S ? p.lse_in[stat] * LOG2E : 0.f;
  const float delta = query < p.S ? p.delta[stat] : 0.f;

  f32x16 dqt[C::NB];
#pragma unroll
  for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqt[nb][r] = 0.f;

  const int kv_end = min(p.S, qb * 128 + 128);
  const int nt = (kv_end + 63) / 64;
  bf16x8 rk[C::CPT], rv[C::CPT];
  tile_load<DH, false>(rk, kp, p.ss, 0, p.S, tid);
  tile_load<DH, false>(rv, vp, p.ss, 0, p.S, tid);
  tile_store<DH, C::KSTR>(rk, Ks[0], tid);
  tile_store<DH, C::KSTR>(rv, Vs[0], tid);
  const uint32_t shi = DROPOUT ? seed_mix(p.seed_hi, p.ctr) : 0u;
  const uint32_t id_bh = (uint32_t)((b * p.H + h) * p.S), thr32 = p.thr << 16;
  const uint32_t rowh = DROPOUT ? drop_row_hash(id_bh + (uint32_t)query, p.seed_lo, shi) : 0u;
  if (DROPOUT && tid < 64) Bm[0][tid] = drop_col_mult(id_bh + (uint32_t)tid, p.seed_lo, shi);
  __syncthreads();
  const int k_nat = (lane & 31) * C::KSTR + hh * 8;
  const int k_tr = (4 * hh + (ip >> 2)) * C::KSTR + 16 * (g & 1) + 4 * (ip & 3);

  for (int jt = 0; jt < nt; ++jt) {
    const int buf = jt & 1, kv0 = jt * 64;
    if (jt + 1 < nt) {
      tile_load<DH, false>(rk, kp, p.ss, kv0 + 64, p.S, tid);
      tile_load<DH, false>(rv, vp, p.ss, kv0 + 64, p.S, tid);
    }
    if (kv0 <= q_base + 31) {
      bf16x8 dsf[2][2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&Ks[buf][k_nat + kb * 32 * C::KSTR + ks * 16]);
          const bf16x8 vf = *reinterpret_cast<const bf16x8*>(&Vs[buf][k_nat + kb * 32 * C::KSTR + ks * 16]);
          s = mfma32(kf, qf[ks], s);
          dp = mfma32(vf, dof[ks], dp);
        }
        u32x4_t dsw[2];
        // the causal / ragged test costs 3 VALU per score: only 32-key blocks that reach past the wave's first query or the
        // end of the sequence need it (wave-uniform choice; both bodies compute identical values where no pair is masked)
        auto scores_to_ds = [&](auto masked_c) {
          constexpr bool MASKED = decltype(masked_c)::value;
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            bool keep[4] = {true, true, true, true};
            if (DROPOUT) drop_keep4(rowh, &Bm[buf][kb * 32 + 8 * qd + 4 * hh], thr32, keep);
            float ds[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = 4 * qd + e;
              const int key = kv0 + kb * 32 + 8 * qd + 4 * hh + e;
              const float ex = __builtin_amdgcn_exp2f(fmaf(s[r], p.c, -lse2));
              const float pv = (MASKED && (key > query || key >= p.S)) ? 0.f : ex;
              // dS = P o (dropout'(dP) - delta): select first, then ONE fma for the 1 / keep scale and the subtraction
              const float dpv = DROPOUT ? (keep[e] ? dp[r] : 0.f) : dp[r];
              ds[e] = pv * (DROPOUT ? fmaf(dpv, p.inv_keep, -delta) : dpv - delta);
            }
            dsw[qd >> 1][2 * (qd & 1)] = pack_bf16x2(ds[0], ds[1]);
            dsw[qd >> 1][2 * (qd & 1) + 1] = pack_bf16x2(ds[2], ds[3]);
          }
        };
        const int last_key = kv0 + kb * 32 + 31;
        if (last_key > q_first || last_key >= p.S) scores_to_ds(std::true_type{});
        else scores_to_ds(std::false_type{});
        dsf[kb][0] = __builtin_bit_cast(bf16x8, dsw[0]);
        dsf[kb][1] = __builtin_bit_cast(bf16x8, dsw[1]);
      }
#pragma unroll
      for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int cs = 0; cs < 2; ++cs) {
            const bf16* kt = &Ks[buf][k_tr + (kb * 32 + 16 * cs) * C::KSTR + nb * 32];
            const bf16x8 kf = cat4(lds_tr_b64(kt), lds_tr_b64(kt + 8 * C::KSTR));
            dqt[nb] = mfma32(kf, dsf[kb][cs], dqt[nb]);
          }
    }
    if (jt + 1 < nt) {
      tile_store<DH, C::KSTR>(rk, Ks[buf ^ 1], tid);
      tile_store<DH, C::KSTR>(rv, Vs[buf ^ 1], tid);
      if (DROPOUT && tid < 64) Bm[buf ^ 1][tid] = drop_col_mult(id_bh + (uint32_t)(kv0 + 64 + tid), p.seed_lo, shi);
    }
    __syncthreads();
  }
  if (query < p.S) {
    bf16* dq = p.dq + (int64_t)b * p.sb + (int64_t)query * p.ss + h * DH;
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)(dqt[nb][4 * qd + e] * p.scale);
        *reinterpret_cast<bf16x4*>(dq + nb * 32 + 8 * qd + 4 * hh) = o;
      }
  }
}

// -------------------------------------------------------------------------------------------------------
// backward: dK, dV (S form, one key per lane; loops over query tiles from the diagonal down)
// -------------------------------------------------------------------------------------------------------
template <int DH, bool DROPOUT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_kernel(AttnParams p) {
  using C = AttnCfg<DH>;
  __shared__ __attribute__((aligned(16))) bf16 Qs[2][64 * C::KSTR];
  __shared__ __attribute__((aligned(16))) bf16 Ds[2][64 * C::KSTR];
  __shared__ __attribute__((aligned(16))) float Ls[2][64];
  __shared__ __attribute__((aligned(16))) float Dl[2][64];
  __shared__ __attribute__((aligned(16))) uint32_t Am[2][64];   // dropout row hashes of the staged query tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, g = lane >> 4, ip = lane & 15;
  const int nkb = (p.S + 127) / 128;
  const int nbh = gridDim.x / nkb;
  const int kblk = (int)(blockIdx.x / nbh);  // earliest key blocks see the most queries: all of them come first
  const int bh = blockIdx.x % nbh, h = bh % p.H, b = bh / p.H;
  const int k_base = kblk * 128 + wave * 32;
  const int key = k_base + (lane & 31);
  const int k_last = __builtin_amdgcn_readfirstlane(k_base) + 31;   // scalar: the wave's last key, for wave-uniform branches
  const bf16* qp = p.q + (int64_t)b * p.sb + h * DH;
  const bf16* kp = p.k + (int64_t)b * p.sb + h * DH;
  const bf16* vp = p.v + (int64_t)b * p.sb + h * DH;
  const bf16* dop = p.d_o + (int64_t)b * p.osb + h * DH;
  const float* lsep = p.lse_in + (int64_t)(b * p.H + h) * p.S;
  const float* delp = p.delta + (int64_t)(b * p.H + h) * p.S;

  bf16x8 kf[C::KS], vf[C::KS];
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) {
    kf[ks] = key < p.S ? *reinterpret_cast<const bf16x8*>(kp + (int64_t)key * p.ss + ks * 16 + hh * 8) : zero8();
    vf[ks] = key < p.S ? *reinterpret_cast<const bf16x8*>(vp + (int64_t)key * p.ss + ks * 16 + hh * 8) : zero8();
  }
  f32x16 dkt[C::NB], dvt[C::NB];
#pragma unroll
  for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkt[nb][r] = 0.f; dvt[nb][r] = 0.f; }

  const int qt0 = (kblk * 128) / 64;
  const int nqt = (p.S + 63) / 64;
  bf16x8 rq[C::CPT], rd[C::CPT];
  // per-tile statistics: threads 0..63 stage lse (in log2 units), threads 64..127 stage delta
  // the loaded value is not touched before STORE_STATS: any arithmetic on it up here would put an s_waitcnt vmcnt(0) -- for
  // the whole prefetch -- in front of the tile's compute
  float rstat = 0.f;
#define LOAD_STATS(q0)                                                                   \
  {                                                                                      \
    const int qi = (q0) + (tid & 63);                                                    \
    if (tid < 128) rstat = qi < p.S ? (tid < 64 ? lsep[qi] * LOG2E : delp[qi]) : 0.f;    \
  }
#define STORE_STATS(buf, q0_)                                                                                         \
  {                                                                                                                  \
    if (tid < 64) Ls[buf][tid] = rstat;                                                                              \
    else if (tid < 128) Dl[buf][tid - 64] = rstat;                                                                   \
    else if (DROPOUT && tid < 192) Am[buf][tid - 128] = drop_row_hash(id_bh + (uint32_t)((q0_) + tid - 128), p.seed_lo, shi); \
  }
  const uint32_t shi = DROPOUT ? seed_mix(p.seed_hi, p.ctr) : 0u;
  const uint32_t id_bh = (uint32_t)((b * p.H + h) * p.S), thr32 = p.thr << 16;
  const uint32_t colm = DROPOUT ? drop_col_mult(id_bh + (uint32_t)key, p.seed_lo, shi) : 0u;
  tile_load<DH, true>(rq, qp, p.ss, qt0 * 64, p.S, tid);
  tile_load<DH, true>(rd, dop, p.oss, qt0 * 64, p.S, tid);
  LOAD_STATS(qt0 * 64)
  tile_store<DH, C::KSTR>(rq, Qs[0], tid);
  tile_store<DH, C::KSTR>(rd, Ds[0], tid);
  STORE_STATS(0, qt0 * 64)
  __syncthreads();
  const int q_nat = (lane & 31) * C::KSTR + hh * 8;
  const int q_tr = (4 * hh + (ip >> 2)) * C::KSTR + 16 * (g & 1) + 4 * (ip & 3);
  for (int qt = qt0; qt < nqt; ++qt) {
    const int buf = (qt - qt0) & 1, q0 = qt * 64;
    if (qt + 1 < nqt) {
      tile_load<DH, true>(rq, qp, p.ss, q0 + 64, p.S, tid);
      tile_load<DH, true>(rd, dop, p.oss, q0 + 64, p.S, tid);
      LOAD_STATS(q0 + 64)
    }
    if (q0 + 63 >= k_base) {  // wave-uniform: some query of this tile can see some key of this wave
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
          const bf16x8 qa = *reinterpret_cast<const bf16x8*>(&Qs[buf][q_nat + qs * 32 * C::KSTR + ks * 16]);
          const bf16x8 da = *reinterpret_cast<const bf16x8*>(&Ds[buf][q_nat + qs * 32 * C::KSTR + ks * 16]);
          s = mfma32(qa, kf[ks], s);
          dp = mfma32(da, vf[ks], dp);
        }
        bf16x8 pf[2], dsf[2];
        u32x4_t pw[2], dsw[2];
        // causal / ragged test (3 VALU per score) only for 32-query blocks that start before the wave's last key or reach past
        // the sequence (wave-uniform choice; identical values where no pair is masked)
        auto scores_to_p_ds = [&](auto masked_c) {
          constexpr bool MASKED = decltype(masked_c)::value;
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(&Ls[buf][qs * 32 + 8 * qd + 4 * hh]);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(&Dl[buf][qs * 32 + 8 * qd + 4 * hh]);
            bool keep4[4] = {true, true, true, true};
            if (DROPOUT) {                 // product scheme: this lane's column multiplier x the four rows' hashes from the tile table
              const u32x4_t rh = *reinterpret_cast<const u32x4_t*>(&Am[buf][qs * 32 + 8 * qd + 4 * hh]);
#pragma unroll
              for (int e = 0; e < 4; ++e) keep4[e] = drop_keep(rh[e], colm, thr32);
            }
            float pd[4], ds[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = 4 * qd + e;
              const int query = q0 + qs * 32 + 8 * qd + 4 * hh + e;
              const float ex = __builtin_amdgcn_exp2f(fmaf(s[r], p.c, -l4[e]));
              const float pv = (MASKED && (key > query || query >= p.S)) ? 0.f : ex;
              float dpv = dp[r];
              pd[e] = pv;
              if (DROPOUT) {
                const bool keep = keep4[e];
                dpv = keep ? dpv : 0.f;
                pd[e] = keep ? pv : 0.f;                   // dV's 1 / keep factor is applied once, when dV is stored
              }
              // dS = P o (dropout'(dP) - delta): ONE fma for the 1 / keep scale and the subtraction
              ds[e] = pv * (DROPOUT ? fmaf(dpv, p.inv_keep, -d4[e]) : dpv - d4[e]);
            }
            pw[qd >> 1][2 * (qd & 1)] = pack_bf16x2(pd[0], pd[1]);
            pw[qd >> 1][2 * (qd & 1) + 1] = pack_bf16x2(pd[2], pd[3]);
            dsw[qd >> 1][2 * (qd & 1)] = pack_bf16x2(ds[0], ds[1]);
            dsw[qd >> 1][2 * (qd & 1) + 1] = pack_bf16x2(ds[2], ds[3]);
          }
        };
        const int first_query = q0 + qs * 32;
        if (k_last > first_query || first_query + 31 >= p.S) scores_to_p_ds(std::true_type{});
        else scores_to_p_ds(std::false_type{});
        pf[0] = __builtin_bit_cast(bf16x8, pw[0]); pf[1] = __builtin_bit_cast(bf16x8, pw[1]);
        dsf[0] = __builtin_bit_cast(bf16x8, dsw[0]); dsf[1] = __builtin_bit_cast(bf16x8, dsw[1]);
#pragma unroll
        for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
          for (int cs = 0; cs < 2; ++cs) {
            const bf16* dt = &Ds[buf][q_tr + (qs * 32 + 16 * cs) * C::KSTR + nb * 32];
            const bf16* qt_ = &Qs[buf][q_tr + (qs * 32 + 16 * cs) * C::KSTR + nb * 32];
            const bf16x8 dof = cat4(lds_tr_b64(dt), lds_tr_b64(dt + 8 * C::KSTR));
            const bf16x8 qf = cat4(lds_tr_b64(qt_), lds_tr_b64(qt_ + 8 * C::KSTR));
            dvt[nb] = mfma32(dof, pf[cs], dvt[nb]);
            dkt[nb] = mfma32(qf, dsf[cs], dkt[nb]);
          }
      }
    }
    if (qt + 1 < nqt) {
      tile_store<DH, C::KSTR>(rq, Qs[buf ^ 1], tid);
      tile_store<DH, C::KSTR>(rd, Ds[buf ^ 1], tid);
      STORE_STATS(buf ^ 1, q0 + 64)
    }
    __syncthreads();
  }
  if (key < p.S) {
    bf16* dk = p.dk + (int64_t)b * p.sb + (int64_t)key * p.ss + h * DH;
    bf16* dv = p.dv + (int64_t)b * p.sb + (int64_t)key * p.ss + h * DH;
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        bf16x4 ok, ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ok[e] = (bf16)(dkt[nb][4 * qd + e] * p.scale);
          ov[e] = (bf16)(DROPOUT ? dvt[nb][4 * qd + e] * p.inv_keep : dvt[nb][4 * qd + e]);
        }
        *reinterpret_cast<bf16x4*>(dk + nb * 32 + 8 * qd + 4 * hh) = ok;
        *reinterpret_cast<bf16x4*>(dv + nb * 32 + 8 * qd + 4 * hh) = ov;
      }
  }
}

__global__ __launch_bounds__(256) void attn_dropout_mask_kernel(uint8_t* mask, int64_t total, int S, int Sp,
                                                                uint32_t thr, uint32_t slo, uint32_t shi,
                                                                const uint32_t* ctr) {
  shi = seed_mix(shi, ctr);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / S;  // (b*H + h)*S + query
    const int key = (int)(i % S);
    const uint32_t colid = (uint32_t)((row / S) * S + key);   // (b*H + h)*S + key
    mask[i] = thr == 0 ? 1 : (drop_keep(drop_row_hash((uint32_t)row, slo, shi), drop_col_mult(colid, slo, shi), thr << 16) ? 1 : 0);
  }
}

static int fill_params(AttnParams& p, int B, int H, int S, int head_dim, int64_t sb, int64_t ss, int64_t osb,
                       int64_t oss, float scale, float dropout_p, uint64_t seed, const uint32_t* dropout_counter) {
  TTTS_REQUIRE(B > 0 && H > 0 && S > 0, "attn: bad shape B=%d H=%d S=%d", B, H, S);
  TTTS_REQUIRE(head_dim == 32 || head_dim == 64 || head_dim == 128, "attn: head_dim %d not in {32,64,128}", head_dim);
  TTTS_REQUIRE(ss % 8 == 0 && sb % 8 == 0 && oss % 8 == 0 && osb % 8 == 0, "attn: strides must be multiples of 8 elements");
  TTTS_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "attn: dropout_p out of range");
  p.Sp = (S + 3) & ~3;
  TTTS_REQUIRE((int64_t)B * H * S < (int64_t)1 << 32 || dropout_p == 0.f, "attn: dropout row-id space exceeds 2^32");
  p.B = B; p.H = H; p.S = S;
  p.sb = sb; p.ss = ss; p.osb = osb; p.oss = oss;
  p.scale = scale;
  p.c = scale * LOG2E;
  p.thr = dropout_threshold(dropout_p);
  p.inv_keep = p.thr ? 65536.0f / (65536.0f - (float)p.thr) : 1.0f;
  p.seed_lo = (uint32_t)seed;
  p.seed_hi = (uint32_t)(seed >> 32);
  p.ctr = dropout_counter;
  return TTTS_OK;
}

}  // namespace ttts

using namespace ttts;

extern "C" int ttts_attn_causal_fwd_bf16(const void* q, const void* k, const void* v, void* o, float* lse, int32_t B,
                                         int32_t H, int32_t S, int32_t head_dim, int64_t qkv_stride_b,
                                         int64_t qkv_stride_s, int64_t o_stride_b, int64_t o_stride_s, float scale,
                                         float dropout_p, uint64_t seed, const uint32_t* dropout_counter, void* stream) {
  TTTS_REQUIRE(q && k && v && o && lse, "attn_fwd: null pointer");
  TTTS_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(o), "attn_fwd: 16-byte alignment required");
  AttnParams p{};
  int rc = fill_params(p, B, H, S, head_dim, qkv_stride_b, qkv_stride_s, o_stride_b, o_stride_s, scale, dropout_p, seed, dropout_counter);
  if (rc) return rc;
  p.q = (const bf16*)q; p.k = (const bf16*)k; p.v = (const bf16*)v; p.out = (bf16*)o; p.lse = lse;
  const int grid = ((S + 127) / 128) * H * B;
  hipStream_t s = as_stream(stream);
  if (head_dim == 64) return attn_fwd_dh64(p, s);   // attn_dh64.hip; the kernels below serve head_dim 32 / 128
  // work split: with dropout the 64-query x 128-key kernel is used (round 1, pair-hash mask: 36.1 vs 43.8 us at the BASELINE shape;
  // round 2, product-scheme mask: 35.5 vs 36.0 us -- the cheaper mask closed the gap); without dropout the 128-query kernel
  // (30.3 vs 33.8 us).
  const bool use_kv2 = p.thr != 0;
  if (use_kv2) {
    const int grid2 = ((S + 63) / 64) * H * B;
#define FWD2(DH)                                                                                                         \
    {                                                                                                                    \
      using C2 = AttnCfg<DH>;                                                                                            \
      const size_t smem = (size_t)128 * (C2::KSTR + C2::VSTR) * sizeof(bf16) + 128 * sizeof(uint32_t);                    \
      static OnceFlag attr_set;                                                                                      \
      if (!attr_set) {                                                                                                   \
        hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kv2_kernel<DH, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  \
        hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kv2_kernel<DH, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        attr_set = true;                                                                                                 \
      }                                                                                                                  \
      if (p.thr) attn_fwd_kv2_kernel<DH, true><<<grid2, 256, smem, s>>>(p);                                              \
      else attn_fwd_kv2_kernel<DH, false><<<grid2, 256, smem, s>>>(p);                                                   \
    }
    if (head_dim == 32) FWD2(32) else if (head_dim == 64) FWD2(64) else FWD2(128)
#undef FWD2
    return check_launch("attn_fwd_kv2");
  }
#define FWD(DH)                                                        \
  if (p.thr) attn_fwd_kernel<DH, true><<<grid, 256, 0, s>>>(p);        \
  else attn_fwd_kernel<DH, false><<<grid, 256, 0, s>>>(p);
  if (head_dim == 32) { FWD(32) } else if (head_dim == 64) { FWD(64) } else { FWD(128) }
#undef FWD
  return check_launch("attn_fwd");
}

extern "C" int64_t ttts_attn_bwd_workspace_bytes(int32_t B, int32_t H, int32_t S) {
  return (int64_t)B * H * S * (int64_t)sizeof(float);
}

extern "C" int ttts_attn_causal_bwd_bf16(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                                         const float* lse, void* dq, void* dk, void* dv, void* workspace, int32_t B,
                                         int32_t H, int32_t S, int32_t head_dim, int64_t qkv_stride_b,
                                         int64_t qkv_stride_s, int64_t o_stride_b, int64_t o_stride_s, float scale,
                                         float dropout_p, uint64_t seed, const uint32_t* dropout_counter, void* stream) {
  TTTS_REQUIRE(q && k && v && o && d_o && lse && dq && dk && dv && workspace, "attn_bwd: null pointer");
  TTTS_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(o) && aligned16(d_o) && aligned16(dq) &&
               aligned16(dk) && aligned16(dv), "attn_bwd: 16-byte alignment required");
  AttnParams p{};
  int rc = fill_params(p, B, H, S, head_dim, qkv_stride_b, qkv_stride_s, o_stride_b, o_stride_s, scale, dropout_p, seed, dropout_counter);
  if (rc) return rc;
  p.q = (const bf16*)q; p.k = (const bf16*)k; p.v = (const bf16*)v; p.o = (const bf16*)o; p.d_o = (const bf16*)d_o;
  p.lse_in = lse; p.dq = (bf16*)dq; p.dk = (bf16*)dk; p.dv = (bf16*)dv; p.delta = (float*)workspace;
  hipStream_t s = as_stream(stream);
  const int grid = ((S + 127) / 128) * H * B;
  const int dgrid = (int)cdiv((int64_t)B * S * H * (head_dim / 8), 256);
#define BWD(DH)                                                             \
  attn_delta_kernel<DH><<<dgrid, 256, 0, s>>>(p);                          \
  if (p.thr) {                                                              \
    attn_bwd_dkdv_kernel<DH, true><<<grid, 256, 0, s>>>(p);                 \
    attn_bwd_dq_kernel<DH, true><<<grid, 256, 0, s>>>(p);                   \
  } else {                                                                  \
    attn_bwd_dkdv_kernel<DH, false><<<grid, 256, 0, s>>>(p);                \
    attn_bwd_dq_kernel<DH, false><<<grid, 256, 0, s>>>(p);                  \
  }
  if (head_dim == 64) {                               // attn_dh64.hip
    attn_delta_kernel<64><<<dgrid, 256, 0, s>>>(p);
    int rc2 = attn_bwd_dkdv_dh64(p, s);
    if (rc2 == TTTS_OK) rc2 = attn_bwd_dq_dh64(p, s);
    return rc2;
  }
  if (head_dim == 32) { BWD(32) } else { BWD(128) }
#undef BWD
  return check_launch("attn_bwd");
}

extern "C" int ttts_attn_dropout_mask_u8(uint8_t* mask, int32_t B, int32_t H, int32_t S, float dropout_p, uint64_t seed,
                                         const uint32_t* dropout_counter, void* stream) {
  TTTS_REQUIRE(mask && B > 0 && H > 0 && S > 0 && dropout_p >= 0.f && dropout_p < 1.f, "dropout_mask: bad arguments");
  const int64_t total = (int64_t)B * H * S * S;
  const int Sp = (S + 3) & ~3;
  TTTS_REQUIRE((int64_t)B * H * S * Sp < (int64_t)1 << 32, "dropout_mask: index space exceeds 2^32");
  attn_dropout_mask_kernel<<<(int)std::min<int64_t>(cdiv(total, 256), 8192), 256, 0, as_stream(stream)>>>(
      mask, total, S, Sp, dropout_threshold(dropout_p), (uint32_t)seed, (uint32_t)(seed >> 32), dropout_counter);
  return check_launch("dropout_mask");
}
