// Causal self-attention for head_dim 64 (the GPT train step's shape: B 8, H 8, S 1156, dropout 0.1) -- round-3 kernels.
// Replaces GPT2Attention's core (transformers modeling_gpt2.py:53-72) as reached from ttts/gpt/model.py:422.
//
// What is different from the generic kernels of attn.hip (which still serve head_dim 32 / 128):
//  * K / V (Q / dO) tiles reach LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no address VALU, no ds_write.
//    The DMA image is lane-linear, so rows are 128 bytes apart and UNPADDED; bank conflicts are removed by an XOR swizzle of
//    the 16-byte chunk index, chunk' = chunk ^ bitrev3(row[3:1]), applied on the DMA's per-lane SOURCE address and on every
//    read.  One swizzle serves both readers: ds_read_b128 of a row fragment (16 lanes = 16 rows distinct in bits 0..3 ->
//    16 distinct 16-byte slots of the 256-byte bank span) and ds_read_b64_tr_b16 (32 lanes = 4 rows x 64 bytes; row bit 1
//    flips chunk bit 2, so rows r and r + 2 land in different 64-byte quarters).
//  * Every wave is software-pipelined over its tiles: the score MFMAs of tile t + 1 are issued in front of tile t's
//    softmax VALU work and tile t's P.V MFMAs behind it, all in one straight-line block, so a wave keeps the matrix pipe and
//    the VALU busy at the same time (the round-2 kernels ran MFMA -> VALU -> MFMA strictly in turn: 13 % MFMA-busy).
//  * Lazy rescale: the running maximum is only raised (and O / l rescaled) when a tile's maximum exceeds it by more than
//    2^8; probabilities are then bounded by 2^8 instead of 1, which bf16 / f32 represent with the same relative precision.
//    The decision for tile t + 1 is taken after tile t's P.V MFMAs are issued, and rescales O, l together.
//  * Work split as before: a workgroup owns 64 queries (64 keys in the dK/dV kernel) and walks 128 rows of the other axis
//    per iteration; wave (a, b) takes 32-row half a and the b-th 64-row tile of the super tile with its own accumulators,
//    merged through LDS once at the end.  Longest workgroups are dispatched first.
// Operand layouts (v_mfma_f32_32x32x16_bf16) are those of attn.hip: S^T = K.Q^T keeps one query per lane, the exponentiated
// accumulator registers are the B operand of O^T += V^T.P^T with V^T read by the hardware transpose read.
#include <type_traits>

#include "attn_common.hpp"

namespace ttts {
namespace dh64 {

constexpr int DH = 64;
constexpr int ROW_B = DH * 2;                    // bytes per tile row
constexpr int TILE64_B = 64 * ROW_B;             // 64-row tile: 8 KB
constexpr int SUPER_B = 2 * TILE64_B;            // 128-row super tile: 16 KB
constexpr int STAGE_B = 2 * SUPER_B;             // two matrices per stage: 32 KB
constexpr int TAB_OFF = 2 * STAGE_B;             // per-stage tables behind the two stages
constexpr float RESCALE_LOG2 = 8.0f;             // lazy rescale threshold, log2 units

// chunk swizzle: bit-reversed row bits 3..1
__device__ __forceinline__ int swz3(int row) { return (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1); }

// LDS-DMA of one 1-KiB chunk (8 rows x 128 bytes): lane i lands at lds + 16 i.  Untracked inline asm (see common.hpp).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma16(uint32_t voff, const void* sbase, uint32_t lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds) : "memory", "m0");
}
#pragma clang diagnostic pop
__device__ __forceinline__ const bf16* uniform_ptr(const bf16* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<const bf16*>(((uint64_t)hi << 32) | lo);
}

// Per-lane source offsets (bytes) of this wave's four chunks of a 128-row super tile: chunk c = 4 wave + i covers rows
// 8 c .. 8 c + 7; lane = (row r8 = lane >> 3, physical chunk pc = lane & 7) fetches logical chunk pc ^ swz3(row).
// rows_valid: rows of the super tile inside the sequence (>= 1); rows beyond repeat the last valid one.
__device__ __forceinline__ void dma_offsets(uint32_t (&voff)[4], int lane, int wave, int64_t ss, int rows_valid) {
  const int r8 = lane >> 3, pc = lane & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * (4 * wave + i) + r8;
    voff[i] = (uint32_t)(min(row, rows_valid - 1) * (int)ss * 2 + ((pc ^ swz3(row)) << 4));
  }
}
__device__ __forceinline__ void dma_super(const uint32_t (&voff)[4], const bf16* sbase, uint32_t lds_super, int wave) {
#pragma unroll
  for (int i = 0; i < 4; ++i) dma16(voff[i], sbase, __builtin_amdgcn_readfirstlane(lds_super + (4 * wave + i) * 1024));
}

// row fragment (A operand of S^T = K.Q^T and friends): row r = lane & 31 of a 32-row block, head-dim slice 16 ks + 8 hh
__device__ __forceinline__ int nat_off(int lane, int ks) {
  const int r = lane & 31, hh = lane >> 5;
  return r * ROW_B + (((2 * ks + hh) ^ swz3(r)) << 4);
}
// transposed fragment (A operand of O^T += V^T.P^T): 4 rows x 16 columns per 16-lane group, see attn.hip
__device__ __forceinline__ int tr_off(int lane, int nb, int j) {
  const int hh = lane >> 5, g = lane >> 4, ip = lane & 15;
  const int row = 4 * hh + (ip >> 2) + 8 * j;
  const int chunk = nb * 4 + 2 * (g & 1) + ((ip & 3) >> 1);
  return row * ROW_B + ((chunk ^ swz3(row)) << 4) + 8 * (ip & 1);
}

__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32, 64); }
// single-instruction forms: hipcc canonicalises MFMA outputs in front of fmaxf (one extra v_max each) and SLP-packs adjacent
// f32 adds into v_pk_add_f32, which issues slower than the two adds it replaces beside MFMAs (MI355X_MICROARCH price list)
__device__ __forceinline__ float max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float add1(float a, float b) {
  float r;
  asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// max over a 2 x 16 score block pair and the other lane half (same query, other keys)
__device__ __forceinline__ float tile_max(const f32x16 (&s)[2]) {
  float a = max3(s[0][0], s[1][0], s[0][1]), b = max3(s[1][1], s[0][2], s[1][2]);
#pragma unroll
  for (int r = 3; r < 15; r += 2) {
    a = max3(a, s[0][r], s[1][r]);
    b = max3(b, s[0][r + 1], s[1][r + 1]);
  }
  a = max3(a, s[0][15], s[1][15]);
  a = fmaxf(a, b);
  return fmaxf(a, xhalf(a));
}

// causal mask of the diagonal tile, applied to the raw scores: key = key0 + kb * 32 + acc_row(r, hh) must be <= query
__device__ __forceinline__ void mask_diag(f32x16 (&s)[2], int qrel /* query - key0 - 4 hh */) {
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (kb * 32 + (r & 3) + 8 * (r >> 2) > qrel) s[kb][r] = NEG_BIG;
  asm volatile("" ::: "memory");   // keeps the caller's wave-uniform branch a branch (if-converted it costs 64 selects per tile)
}

// two scores -> one packed bf16 probability pair; row-sum partials in l[0], l[1]
template <bool DROPOUT>
__device__ __forceinline__ uint32_t sm2(float s0, float s1, float c, float m2, float (&l)[2], uint32_t rowh, uint32_t thr32,
                                        uint32_t cm0, uint32_t cm1) {
  float p0 = __builtin_amdgcn_exp2f(fmaf(s0, c, -m2)), p1 = __builtin_amdgcn_exp2f(fmaf(s1, c, -m2));
  l[0] = add1(l[0], p0);
  l[1] = add1(l[1], p1);
  if (DROPOUT) {                                    // 1 / keep is applied once, with 1 / l, at the end
    p0 = drop_keep(rowh, cm0, thr32) ? p0 : 0.f;
    p1 = drop_keep(rowh, cm1, thr32) ? p1 : 0.f;
  }
  return pack_bf16x2(p0, p1);
}
#define SB() __builtin_amdgcn_sched_barrier(0)

// -------------------------------------------------------------------------------------------------------
// forward
// -------------------------------------------------------------------------------------------------------
template <bool DROPOUT>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // stage st: K super tile at st * STAGE_B, V super tile behind it; Bm[2][128] dropout column multipliers at TAB_OFF
  const uint32_t lds0 = lds_byte_addr(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qh = wave & 1, kh = wave >> 1;
  const int hh = lane >> 5;
  const int nqb = (p.S + 63) / 64;
  const int nbh = gridDim.x / nqb;
  const int qb = nqb - 1 - (int)(blockIdx.x / nbh);             // longest workgroups first
  const int bh = blockIdx.x % nbh, h = bh % p.H, b = bh / p.H;
  const int q0 = qb * 64, q_base = q0 + qh * 32;
  const int query = q_base + (lane & 31);
  const bf16* qp = p.q + (int64_t)b * p.sb + h * DH;
  const bf16* kp = uniform_ptr(p.k + (int64_t)b * p.sb + h * DH);
  const bf16* vp = uniform_ptr(p.v + (int64_t)b * p.sb + h * DH);

  const int nsup = (qb >> 1) + 1;                               // 128-key super tiles up to the diagonal
  const int nt_w = nsup - ((kh == 1 && !(qb & 1)) ? 1 : 0);     // this wave's 64-key tiles: T = 2 js + kh <= qb
  const int64_t super_el = (int64_t)128 * p.ss;

  uint32_t voff[4];
  dma_offsets(voff, lane, wave, p.ss, 128);
  const bool ragged = nsup * 128 > p.S;                          // the last super tile reaches past the sequence (last query block only)
  auto issue = [&](int js, const bf16* base, int mat) {
    const uint32_t dst = lds0 + (js & 1) * STAGE_B + mat * SUPER_B;
    if (ragged && js == nsup - 1) {                              // rows beyond the sequence repeat the last one (only masked scores meet them)
      uint32_t vl[4];
      dma_offsets(vl, lane, wave, p.ss, p.S - (nsup - 1) * 128);
      dma_super(vl, base + js * super_el, dst, wave);
    } else {
      dma_super(voff, base + js * super_el, dst, wave);
    }
  };
  issue(0, kp, 0);
  issue(0, vp, 1);
  if (nsup > 1) issue(1, kp, 0);

  bf16x8 qf[4];
  {
    const bf16* qrow = qp + (int64_t)min(query, p.S - 1) * p.ss + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qrow + ks * 16);
  }
  const uint32_t shi = DROPOUT ? seed_mix(p.seed_hi, p.ctr) : 0u;
  const uint32_t id_bh = (uint32_t)((b * p.H + h) * p.S), thr32 = p.thr << 16;
  const uint32_t rowh = DROPOUT ? drop_row_hash(id_bh + (uint32_t)query, p.seed_lo, shi) : 0u;
  uint32_t* Bm = reinterpret_cast<uint32_t*>(smem + TAB_OFF);
  if (DROPOUT && tid < 128) Bm[tid] = drop_col_mult(id_bh + (uint32_t)tid, p.seed_lo, shi);

  int knat[4], vtr[2][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) knat[ks] = kh * TILE64_B + nat_off(lane, ks);
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int j = 0; j < 2; ++j) vtr[nb][j] = SUPER_B + kh * TILE64_B + tr_off(lane, nb, j);
  const int bm_lane = TAB_OFF + (kh * 64 + 4 * hh) * 4;
  const int qrel = query - (qb * 64) - 4 * hh;                   // the diagonal tile starts at key 64 qb = q0

  f32x16 ot[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[nb][r] = 0.f;
  float m = NEG_BIG, m2 = NEG_BIG, l[2] = {0.f, 0.f};

  auto qk = [&](f32x16 (&s)[2], int st) {                        // S^T of this wave's tile in stage st
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(smem + knat[ks] + st * STAGE_B + kb * 32 * ROW_B);
        s[kb] = mfma32(kf, qf[ks], s[kb]);
      }
    }
  };
  // raise the running max if the tile's max exceeds it by more than the threshold (wave-uniform branch, rare after the
  // first tiles): everything accumulated at the old max -- O and l -- is rescaled together
  auto decide = [&](const f32x16 (&s)[2]) {
    const float mx = tile_max(s);
    if (__builtin_amdgcn_ballot_w64((mx - m) * p.c > RESCALE_LOG2) != 0) {
      const float m_new = fmaxf(m, mx);
      const float alpha = __builtin_amdgcn_exp2f((m - m_new) * p.c);
      m = m_new;
      m2 = m_new * p.c;
      l[0] *= alpha; l[1] *= alpha;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[nb][r] *= alpha;
    }
  };

  lds_dma_wait_all();
  // the compiler must see Q consumed here: with its loads still pending on some path into the loop it waits for them by
  // count (vmcnt(3) .. vmcnt(0)) in front of the first MFMAs of every step -- and those counts also cover the step's DMA
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) asm volatile("" :: "v"(qf[ks]));
  __syncthreads();

  f32x16 sa[2], sb[2];
  if (nt_w > 0) {
    qk(sa, 0);
    if (nt_w == 1 && kh == (qb & 1)) mask_diag(sa, qrel);       // the wave's only tile is the diagonal one
    decide(sa);
  }
  __syncthreads();                                              // K(0) is overwritten by step 0's K(2)

  // One pipeline step: tile t's scores are in `cur`; with NEXT tile t + 1's are computed into `nxt`.  The issue order is
  // pinned slot by slot (sched_barrier between slots): 16 half-groups H0..H15 of softmax work (2 scores per lane each:
  // ~13 VALU), one MFMA behind each of the first twelve -- the 8 score MFMAs of the next tile (Q0..Q7), then the first
  // key block's 4 P.V MFMAs (P0..P3) -- and the second block's P.V MFMAs (P4..P7) in front of the running-max update.
  // LDS fragment reads are issued one to two slots ahead of the MFMA that consumes them.
  auto tile_body = [&](auto next_c, auto st_c, f32x16 (&cur)[2], f32x16 (&nxt)[2]) {
    constexpr bool NEXT = decltype(next_c)::value;
    constexpr int ST = decltype(st_c)::value;
    const unsigned char* kbase = smem + (ST ^ 1) * STAGE_B;      // next tile's K lives in the other stage
    const unsigned char* vbase = smem + ST * STAGE_B;
    const unsigned char* bmb = smem + bm_lane + ST * 512;
    bf16x8 kf[2][4], vf[8];
    u32x4_t cm[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    u32x4_t pw[2][2];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto load_k = [&](int kb) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kf[kb][ks] = *reinterpret_cast<const bf16x8*>(kbase + knat[ks] + kb * 32 * ROW_B);
    };
    auto load_v = [&](int n) {                                   // fragment of P.V MFMA n: kb = n >> 2, cs = (n >> 1) & 1, nb = n & 1
      const int off = ((n >> 2) * 32 + 16 * ((n >> 1) & 1)) * ROW_B;
      vf[n] = cat4(lds_tr_b64(reinterpret_cast<const bf16*>(vbase + vtr[n & 1][0] + off)),
                   lds_tr_b64(reinterpret_cast<const bf16*>(vbase + vtr[n & 1][1] + off)));
    };
    auto load_cm = [&](int G) { if (DROPOUT) cm[G & 1] = *reinterpret_cast<const u32x4_t*>(bmb + (G >> 2) * 128 + (G & 3) * 32); };
    if (NEXT) load_k(0);
    load_cm(0);
    SB();
#pragma unroll
    for (int G = 0; G < 8; ++G) {                                // G = 4 kb + qd: four scores per lane
      const int kb = G >> 2, qd = G & 3;
      if (G < 7) load_cm(G + 1);
      if (NEXT && G == 1) load_k(1);
      if (G == 3) { load_v(0); load_v(1); }
      if (G == 4) { load_v(2); load_v(3); }
      if (G == 6) { load_v(4); load_v(5); }
      if (G == 7) { load_v(6); load_v(7); }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int n = 2 * G + hf;
        pw[kb][qd >> 1][2 * (qd & 1) + hf] = sm2<DROPOUT>(cur[kb][4 * qd + 2 * hf], cur[kb][4 * qd + 2 * hf + 1], p.c, m2, l, rowh, thr32,
                                                          cm[G & 1][2 * hf], cm[G & 1][2 * hf + 1]);
        SB();
        if (NEXT && n < 8) {
          const int qb_ = n >> 2, ks = n & 3;
          nxt[qb_] = mfma32(kf[qb_][ks], qf[ks], ks == 0 ? zero16 : nxt[qb_]);
          SB();
        } else if (n >= 8 && n < 12) {
          const int m_ = n - 8;                                  // P0..P3: kb 0, cs = m_ >> 1, nb = m_ & 1
          ot[m_ & 1] = mfma32(vf[m_], __builtin_bit_cast(bf16x8, pw[0][m_ >> 1]), ot[m_ & 1]);
          SB();
        }
      }
    }
#pragma unroll
    for (int m_ = 4; m_ < 8; ++m_) ot[m_ & 1] = mfma32(vf[m_], __builtin_bit_cast(bf16x8, pw[1][(m_ >> 1) & 1]), ot[m_ & 1]);
  };
  auto step = [&](auto st_c, f32x16 (&cur)[2], f32x16 (&nxt)[2], int js) {
    constexpr int ST = decltype(st_c)::value;                   // js & 1
    if (js + 2 < nsup) issue(js + 2, kp, 0);                    // K(js + 2) replaces K(js): read during step js - 1
    if (js + 1 < nsup) {
      issue(js + 1, vp, 1);                                     // V(js + 1) replaces V(js - 1)
      if (DROPOUT && tid < 128) Bm[(ST ^ 1) * 128 + tid] = drop_col_mult(id_bh + (uint32_t)((js + 1) * 128 + tid), p.seed_lo, shi);
    }
    if (js + 1 < nt_w) {
      tile_body(std::true_type{}, st_c, cur, nxt);
      if (js + 2 == nt_w && kh == (qb & 1)) mask_diag(nxt, qrel);   // the wave's last tile is the diagonal one
      decide(nxt);
    } else if (js < nt_w) {
      tile_body(std::false_type{}, st_c, cur, nxt);
    }
    lds_dma_wait_all();
    __syncthreads();
  };
  for (int js = 0; js < nsup; js += 2) {
    step(std::integral_constant<int, 0>{}, sa, sb, js);
    if (js + 1 < nsup) step(std::integral_constant<int, 1>{}, sb, sa, js + 1);
  }

  // merge the two key halves of each query half: wave (qh, 1) hands (m, l, O) to wave (qh, 0), lane for lane
  float lt = l[0] + l[1];
  float* mb = reinterpret_cast<float*>(smem);                    // [34][128] words, k-major: conflict-free
  const int ml = qh * 64 + lane;
  if (kh == 1) {
    mb[ml] = m; mb[128 + ml] = lt;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mb[(2 + nb * 16 + r) * 128 + ml] = ot[nb][r];
  }
  __syncthreads();
  unsigned char* ost = smem + 34 * 128 * 4 + qh * (32 * 144);    // output staging: [32 queries][144-byte rows] per query half
  if (kh == 0) {
    const float m_o = mb[ml], l_o = mb[128 + ml];
    const float M = fmaxf(m, m_o);
    float a1 = __builtin_amdgcn_exp2f((m - M) * p.c), a2 = __builtin_amdgcn_exp2f((m_o - M) * p.c);
    lt = lt * a1 + l_o * a2;
    lt += xhalf(lt);
    const float inv = (DROPOUT ? p.inv_keep : 1.0f) / lt;
    a2 *= inv; a1 *= inv;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * qd + e;
          o[e] = (bf16)(ot[nb][r] * a1 + mb[(2 + nb * 16 + r) * 128 + ml] * a2);
        }
        *reinterpret_cast<bf16x4*>(ost + (lane & 31) * 144 + (nb * 32 + 8 * qd + 4 * hh) * 2) = o;
      }
    if (hh == 0 && query < p.S) p.lse[(int64_t)(b * p.H + h) * p.S + query] = (M * p.c + __log2f(lt)) * LN2;
  }
  __syncthreads();
  if (kh == 0) {                                                 // whole 128-byte rows: 8 lanes per row, 8 rows per store
    bf16* op = p.out + (int64_t)b * p.osb + h * DH;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = i * 8 + (lane >> 3), ch = lane & 7;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(ost + row * 144 + ch * 16);
      if (q_base + row < p.S) *reinterpret_cast<bf16x8*>(op + (int64_t)(q_base + row) * p.oss + ch * 8) = v;
    }
  }
}


// -------------------------------------------------------------------------------------------------------
// backward: dQ (S^T form, one query per lane; same work split, staging and swizzle as the forward kernel)
//   per 32-key block:  S^T = K.Q^T, dP^T = V.dO^T (8 MFMAs), dS = P o (dropout'(dP) - delta), dQ^T += K^T.dS^T (4 MFMAs).
// Pipelined per 32-key BLOCK in two phases (48 accumulator registers in flight instead of 128 for a tile-deep pipeline):
//   phase 1  P = exp2(S c - lse) in place (32 VALU)              beside the block's own 4 dP MFMAs,
//   phase 2  dS from P, dP, the dropout mask (~90 VALU)           beside the NEXT block's 4 S MFMAs and this block's dQ MFMAs.
// Tile js + 1 (K and V) is DMA'd at the top of step js into the other stage and published by a barrier in the middle of the
// step -- the first block's "next" is the same tile's second block; K(js) stays to the end of the step for the dQ reads.
// -------------------------------------------------------------------------------------------------------
// single-instruction forms (see max3 / add1): the compiler would pack these into v_pk_fma_f32 / v_pk_mul_f32
__device__ __forceinline__ float fma1(float a, float b, float c) {
  float r;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float mul1(float a, float b) {
  float r;
  asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// two probabilities + two dP values -> one packed bf16 dS pair: dS = P o (dropout'(dP) - delta)
template <bool DROPOUT>
__device__ __forceinline__ uint32_t ds2(float p0, float p1, float dp0, float dp1, float ndelta, float inv_keep, uint32_t rowh,
                                        uint32_t thr32, uint32_t cm0, uint32_t cm1) {
  float t0, t1;
  if (DROPOUT) {                                    // select first, then ONE fma for the 1 / keep scale and the subtraction
    t0 = fma1(drop_keep(rowh, cm0, thr32) ? dp0 : 0.f, inv_keep, ndelta);
    t1 = fma1(drop_keep(rowh, cm1, thr32) ? dp1 : 0.f, inv_keep, ndelta);
  } else {
    t0 = add1(dp0, ndelta);
    t1 = add1(dp1, ndelta);
  }
  return pack_bf16x2(mul1(p0, t0), mul1(p1, t1));
}

template <bool DROPOUT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lds0 = lds_byte_addr(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qh = wave & 1, kh = wave >> 1;
  const int hh = lane >> 5;
  const int nqb = (p.S + 63) / 64;
  const int nbh = gridDim.x / nqb;
  const int qb = nqb - 1 - (int)(blockIdx.x / nbh);             // longest workgroups first
  const int bh = blockIdx.x % nbh, h = bh % p.H, b = bh / p.H;
  const int q0 = qb * 64, q_base = q0 + qh * 32;
  const int query = q_base + (lane & 31), queryc = min(query, p.S - 1);
  const bf16* kp = uniform_ptr(p.k + (int64_t)b * p.sb + h * DH);
  const bf16* vp = uniform_ptr(p.v + (int64_t)b * p.sb + h * DH);
  const int nsup = (qb >> 1) + 1;
  const int nt_w = nsup - ((kh == 1 && !(qb & 1)) ? 1 : 0);
  const bool diag_last = kh == (qb & 1);                        // this wave's last tile is the diagonal one
  const int64_t super_el = (int64_t)128 * p.ss;

  uint32_t voff[4];
  dma_offsets(voff, lane, wave, p.ss, 128);
  const bool ragged = nsup * 128 > p.S;
  auto issue = [&](int js) {                                    // K and V of super tile js -> stage js & 1
    const uint32_t dst = lds0 + (js & 1) * STAGE_B;
    if (ragged && js == nsup - 1) {
      uint32_t vl[4];
      dma_offsets(vl, lane, wave, p.ss, p.S - (nsup - 1) * 128);
      dma_super(vl, kp + js * super_el, dst, wave);
      dma_super(vl, vp + js * super_el, dst + SUPER_B, wave);
    } else {
      dma_super(voff, kp + js * super_el, dst, wave);
      dma_super(voff, vp + js * super_el, dst + SUPER_B, wave);
    }
  };
  issue(0);

  bf16x8 qf[4], dof[4];
  {
    const bf16* qrow = p.q + (int64_t)b * p.sb + h * DH + (int64_t)queryc * p.ss + hh * 8;
    const bf16* drow = p.d_o + (int64_t)b * p.osb + h * DH + (int64_t)queryc * p.oss + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = *reinterpret_cast<const bf16x8*>(qrow + ks * 16);
      dof[ks] = *reinterpret_cast<const bf16x8*>(drow + ks * 16);
    }
  }
  const int64_t stat = (int64_t)(b * p.H + h) * p.S + queryc;
  const float lse2 = p.lse_in[stat] * LOG2E, delta = p.delta[stat];
  const uint32_t shi = DROPOUT ? seed_mix(p.seed_hi, p.ctr) : 0u;
  const uint32_t id_bh = (uint32_t)((b * p.H + h) * p.S), thr32 = p.thr << 16;
  const uint32_t rowh = DROPOUT ? drop_row_hash(id_bh + (uint32_t)query, p.seed_lo, shi) : 0u;
  uint32_t* Bm = reinterpret_cast<uint32_t*>(smem + TAB_OFF);
  if (DROPOUT && tid < 128) Bm[tid] = drop_col_mult(id_bh + (uint32_t)tid, p.seed_lo, shi);

  int knat[4], ktr[2][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) knat[ks] = kh * TILE64_B + nat_off(lane, ks);
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int j = 0; j < 2; ++j) ktr[nb][j] = kh * TILE64_B + tr_off(lane, nb, j);
  const int bm_lane = TAB_OFF + (kh * 64 + 4 * hh) * 4;
  const int qrel = query - (qb * 64) - 4 * hh;

  f32x16 dqt[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqt[nb][r] = 0.f;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  const float ndelta = -delta;
  // S^T of one 32-key block (stage st, block kb of this wave's tile), unpipelined: the wave's very first block
  auto scores = [&](f32x16& s, int st, int kb) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      s = mfma32(*reinterpret_cast<const bf16x8*>(smem + knat[ks] + st * STAGE_B + kb * 32 * ROW_B), qf[ks], ks == 0 ? zero16 : s);
  };
  auto mask_block = [&](f32x16& s, int kb) {                    // diagonal tile: key = 64 qb + 32 kb + acc_row must be <= query
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (kb * 32 + (r & 3) + 8 * (r >> 2) > qrel) s[r] = NEG_BIG;
    asm volatile("" ::: "memory");
  };

  lds_dma_wait_all();
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) asm volatile("" :: "v"(qf[ks]), "v"(dof[ks]));   // see the forward kernel
  asm volatile("" :: "v"(lse2), "v"(delta));
  __syncthreads();

  f32x16 sa, sb, dp;
  if (nt_w > 0) {
    scores(sa, 0, 0);
    if (nt_w == 1 && diag_last) mask_block(sa, 0);
  }

  // One pipelined block (ST, KB): `cs` holds its scores; with NEXT the following block's scores go to `ns` (same tile's
  // second block, or the next tile's first block in the other stage).  Issue order pinned slot by slot.
  auto block_body = [&](auto next_c, auto st_c, auto kb_c, f32x16& cs, f32x16& ns) {
    constexpr bool NEXT = decltype(next_c)::value;
    constexpr int ST = decltype(st_c)::value, KB = decltype(kb_c)::value;
    constexpr int NST = KB == 0 ? ST : (ST ^ 1), NKB = KB ^ 1;
    const unsigned char* nbase = smem + NST * STAGE_B + NKB * 32 * ROW_B;
    const unsigned char* tbase = smem + ST * STAGE_B + KB * 32 * ROW_B;
    const unsigned char* bmb = smem + bm_lane + ST * 512 + KB * 128;
    bf16x8 kf[4], vf[4], tf[4];
    u32x4_t cm[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    u32x4_t dsw[2];
    auto load_cm = [&](int G) { if (DROPOUT) cm[G & 1] = *reinterpret_cast<const u32x4_t*>(bmb + G * 32); };
    auto load_t = [&](int n) {                                   // K^T fragment of dQ MFMA n: cs = n >> 1, nb = n & 1
      const int off = 16 * (n >> 1) * ROW_B;
      tf[n] = cat4(lds_tr_b64(reinterpret_cast<const bf16*>(tbase + ktr[n & 1][0] + off)),
                   lds_tr_b64(reinterpret_cast<const bf16*>(tbase + ktr[n & 1][1] + off)));
    };
    // phase 1
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) vf[ks] = *reinterpret_cast<const bf16x8*>(tbase + SUPER_B + knat[ks]);
    SB();
#pragma unroll
    for (int G = 0; G < 4; ++G) {
      if (NEXT && G == 2) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[ks] = *reinterpret_cast<const bf16x8*>(nbase + knat[ks]);
      }
      if (G == 3) load_cm(0);
#pragma unroll
      for (int e = 0; e < 4; ++e) cs[4 * G + e] = __builtin_amdgcn_exp2f(fmaf(cs[4 * G + e], p.c, -lse2));
      SB();
      dp = mfma32(vf[G], dof[G], G == 0 ? zero16 : dp);
      SB();
    }
    // phase 2
#pragma unroll
    for (int G = 0; G < 4; ++G) {
      if (G < 3) load_cm(G + 1);
      if (G == 0) { load_t(0); load_t(1); }
      if (G == 2) { load_t(2); load_t(3); }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int n = 2 * G + hf, r = 4 * G + 2 * hf;
        dsw[G >> 1][2 * (G & 1) + hf] = ds2<DROPOUT>(cs[r], cs[r + 1], dp[r], dp[r + 1], ndelta, p.inv_keep, rowh, thr32,
                                                     cm[G & 1][2 * hf], cm[G & 1][2 * hf + 1]);
        SB();
        if (NEXT && n < 4) ns = mfma32(kf[n], qf[n], n == 0 ? zero16 : ns);
        if (n == 4 || n == 5) dqt[n - 4] = mfma32(tf[n - 4], __builtin_bit_cast(bf16x8, dsw[0]), dqt[n - 4]);
        SB();
      }
    }
    dqt[0] = mfma32(tf[2], __builtin_bit_cast(bf16x8, dsw[1]), dqt[0]);
    dqt[1] = mfma32(tf[3], __builtin_bit_cast(bf16x8, dsw[1]), dqt[1]);
  };

  auto step = [&](auto st_c, int js) {
    constexpr int ST = decltype(st_c)::value;                   // js & 1
    if (js + 1 < nsup) {
      issue(js + 1);
      if (DROPOUT && tid < 128) Bm[(ST ^ 1) * 128 + tid] = drop_col_mult(id_bh + (uint32_t)((js + 1) * 128 + tid), p.seed_lo, shi);
    }
    const bool active = js < nt_w, more = js + 1 < nt_w;
    if (active) {
      block_body(std::true_type{}, st_c, std::integral_constant<int, 0>{}, sa, sb);
      if (!more && diag_last) mask_block(sb, 1);
    }
    lds_dma_wait_all();
    __syncthreads();                                            // tile js + 1 is in LDS
    if (active) {
      if (more) {
        block_body(std::true_type{}, st_c, std::integral_constant<int, 1>{}, sb, sa);
        if (js + 2 == nt_w && diag_last) mask_block(sa, 0);
      } else {
        block_body(std::false_type{}, st_c, std::integral_constant<int, 1>{}, sb, sa);
      }
    }
    __syncthreads();                                            // everyone is done with stage ST
  };
  for (int js = 0; js < nsup; js += 2) {
    step(std::integral_constant<int, 0>{}, js);
    if (js + 1 < nsup) step(std::integral_constant<int, 1>{}, js + 1);
  }

  // merge the two key halves (sum), scale, convert; whole rows leave through an LDS staging tile
  float* mb = reinterpret_cast<float*>(smem);                    // [32][128] words
  const int ml = qh * 64 + lane;
  if (kh == 1) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mb[(nb * 16 + r) * 128 + ml] = dqt[nb][r];
  }
  __syncthreads();
  unsigned char* ost = smem + 32 * 128 * 4 + qh * (32 * 144);
  if (kh == 0) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * qd + e;
          o[e] = (bf16)((dqt[nb][r] + mb[(nb * 16 + r) * 128 + ml]) * p.scale);
        }
        *reinterpret_cast<bf16x4*>(ost + (lane & 31) * 144 + (nb * 32 + 8 * qd + 4 * hh) * 2) = o;
      }
  }
  __syncthreads();
  if (kh == 0) {
    bf16* dq = p.dq + (int64_t)b * p.sb + h * DH;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = i * 8 + (lane >> 3), ch = lane & 7;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(ost + row * 144 + ch * 16);
      if (q_base + row < p.S) *reinterpret_cast<bf16x8*>(dq + (int64_t)(q_base + row) * p.ss + ch * 8) = v;
    }
  }
}


// -------------------------------------------------------------------------------------------------------
// backward: dK, dV (S form, one key per lane; a workgroup owns 64 keys and walks 128-query super tiles from the diagonal
// down; wave (kh, qp) takes key half kh and the qp-th 64-query tile of each super tile, the two query parts are summed
// through LDS at the end).  Per 32-query block:
//   S = Q.K^T, dP = dO.V^T (8 MFMAs), P and dS as in the dQ kernel, dV^T += dO^T.P, dK^T += Q^T.dS (8 MFMAs).
// Two-phase block pipeline as in the dQ kernel: phase 1 exponentiates beside the block's dP MFMAs, phase 2 forms P / dS
// beside the next block's S MFMAs and this block's dV / dK MFMAs.  The per-query lse, delta and dropout row hash come
// from per-stage LDS tables (staged through registers: 128 + 128 + 128 values per super tile).
// -------------------------------------------------------------------------------------------------------
constexpr int STAT_B = 3 * 128 * 4;              // lse2[128], -delta[128], row hash[128] per stage

template <bool DROPOUT>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkdv_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // stage st: Q super tile at st * STAGE_B, dO super tile behind it; tables at TAB_OFF + st * STAT_B
  const uint32_t lds0 = lds_byte_addr(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = wave & 1, qp = wave >> 1;
  const int hh = lane >> 5;
  const int nkb = (p.S + 63) / 64;
  const int nbh = gridDim.x / nkb;
  const int kblk = (int)(blockIdx.x / nbh);                     // earliest key blocks see the most queries: they come first
  const int bh = blockIdx.x % nbh, h = bh % p.H, b = bh / p.H;
  const int k_base = kblk * 64 + kh * 32;
  const int key = k_base + (lane & 31), keyc = min(key, p.S - 1);
  const bf16* qp_ = uniform_ptr(p.q + (int64_t)b * p.sb + h * DH);
  const bf16* dop = uniform_ptr(p.d_o + (int64_t)b * p.osb + h * DH);
  const float* lsep = p.lse_in + (int64_t)(b * p.H + h) * p.S;
  const float* delp = p.delta + (int64_t)(b * p.H + h) * p.S;

  const int js0 = kblk >> 1;                                    // first 128-query super tile (holds the diagonal)
  const int nsup = (p.S + 127) / 128 - js0;                     // super tiles this workgroup walks
  const int tq_max = (p.S - 1) / 64;                            // last 64-query tile with a valid query
  const int i_first = qp < (kblk & 1) ? 1 : 0;                  // local index of this wave's first / last tile: Tq = 2 (js0 + i) + qp
  const int i_last = tq_max - qp >= 0 ? (tq_max - qp) / 2 - js0 : -1;
  const bool rag = (p.S & 63) != 0;
  const int64_t qsuper_el = (int64_t)128 * p.ss, dsuper_el = (int64_t)128 * p.oss;

  uint32_t voffq[4], voffd[4];
  dma_offsets(voffq, lane, wave, p.ss, 128);
  dma_offsets(voffd, lane, wave, p.oss, 128);
  auto issue = [&](int i) {                                     // Q and dO of local super tile i -> stage i & 1
    const int js = js0 + i;
    const uint32_t dst = lds0 + (i & 1) * STAGE_B;
    if (js * 128 + 128 > p.S) {                                 // rows beyond the sequence repeat the last one; their scores are masked
      uint32_t vq[4], vd[4];
      dma_offsets(vq, lane, wave, p.ss, p.S - js * 128);
      dma_offsets(vd, lane, wave, p.oss, p.S - js * 128);
      dma_super(vq, qp_ + js * qsuper_el, dst, wave);
      dma_super(vd, dop + js * dsuper_el, dst + SUPER_B, wave);
    } else {
      dma_super(voffq, qp_ + js * qsuper_el, dst, wave);
      dma_super(voffd, dop + js * dsuper_el, dst + SUPER_B, wave);
    }
  };
  issue(0);

  const uint32_t shi = DROPOUT ? seed_mix(p.seed_hi, p.ctr) : 0u;
  const uint32_t id_bh = (uint32_t)((b * p.H + h) * p.S), thr32 = p.thr << 16;
  const uint32_t colm = DROPOUT ? drop_col_mult(id_bh + (uint32_t)key, p.seed_lo, shi) : 0u;
  // per-query statistics of a super tile: threads 0..127 stage lse (log2 units) and the dropout row hash, 128..255 stage -delta
  const float* statp = tid < 128 ? lsep : delp;
  const float statm = tid < 128 ? LOG2E : -1.0f;
  auto load_stat = [&](int i) { return statp[min((js0 + i) * 128 + (tid & 127), p.S - 1)]; };
  auto store_stat = [&](int i, float v) {
    float* tab = reinterpret_cast<float*>(smem + TAB_OFF + (i & 1) * STAT_B);
    tab[tid] = v * statm;                                       // lse2 at [0, 128), -delta at [128, 256)
    if (DROPOUT && tid < 128)
      reinterpret_cast<uint32_t*>(tab)[256 + tid] = drop_row_hash(id_bh + (uint32_t)((js0 + i) * 128 + tid), p.seed_lo, shi);
  };
  float rstat = load_stat(0);

  bf16x8 kf[4], vf[4];
  {
    const bf16* krow = p.k + (int64_t)b * p.sb + h * DH + (int64_t)keyc * p.ss + hh * 8;
    const bf16* vrow = p.v + (int64_t)b * p.sb + h * DH + (int64_t)keyc * p.ss + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = *reinterpret_cast<const bf16x8*>(krow + ks * 16);
      vf[ks] = *reinterpret_cast<const bf16x8*>(vrow + ks * 16);
    }
  }
  int qnat[4], qtr[2][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qnat[ks] = qp * TILE64_B + nat_off(lane, ks);
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int j = 0; j < 2; ++j) qtr[nb][j] = qp * TILE64_B + tr_off(lane, nb, j);
  const int tab_lane = TAB_OFF + (qp * 64 + 4 * hh) * 4;

  f32x16 dkt[2], dvt[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkt[nb][r] = 0.f; dvt[nb][r] = 0.f; }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // S of one 32-query block (stage st, block qs of this wave's tile), unpipelined: the wave's very first block
  auto scores = [&](f32x16& s, int st, int qs) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      s = mfma32(*reinterpret_cast<const bf16x8*>(smem + qnat[ks] + st * STAGE_B + qs * 32 * ROW_B), kf[ks], ks == 0 ? zero16 : s);
  };
  // diagonal / ragged tile i: query = 64 Tq + 32 qs + acc_row must be >= key and < S
  auto mask_block = [&](f32x16& s, int i, int qs) {
    const int tq0 = (2 * (js0 + i) + qp) * 64 + 4 * hh;
    const int krel = key - tq0, qlim = p.S - tq0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cr = qs * 32 + (r & 3) + 8 * (r >> 2);
      if (cr < krel || cr >= qlim) s[r] = NEG_BIG;
    }
    asm volatile("" ::: "memory");
  };
  auto needs_mask = [&](int i) { return 2 * (js0 + i) + qp == kblk || (rag && 2 * (js0 + i) + qp == tq_max); };

  lds_dma_wait_all();
  store_stat(0, rstat);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) asm volatile("" :: "v"(kf[ks]), "v"(vf[ks]));   // see the forward kernel
  __syncthreads();

  f32x16 sa, sb, dp;
  if (i_first == 0 && i_last >= 0) {
    scores(sa, 0, 0);
    if (needs_mask(0)) mask_block(sa, 0, 0);
  }

  // One pipelined block (ST, QS): `cs` holds its scores; with NEXT the following block's scores go to `ns`.
  auto block_body = [&](auto next_c, auto st_c, auto qs_c, f32x16& cs, f32x16& ns) {
    constexpr bool NEXT = decltype(next_c)::value;
    constexpr int ST = decltype(st_c)::value, QS = decltype(qs_c)::value;
    constexpr int NST = QS == 0 ? ST : (ST ^ 1), NQS = QS ^ 1;
    const unsigned char* nbase = smem + NST * STAGE_B + NQS * 32 * ROW_B;
    const unsigned char* tbase = smem + ST * STAGE_B + QS * 32 * ROW_B;
    const unsigned char* tab = smem + tab_lane + ST * STAT_B + QS * 128;
    bf16x8 qa[4], da[4], tfd[4], tfq[4];
    f32x4 ls, nd[2];
    u32x4_t rh[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    u32x4_t pw[2], dsw[2];
    auto load_t = [&](int n) {                                   // dO^T and Q^T fragments of MFMA pair n: cs = n >> 1, nb = n & 1
      const int off = 16 * (n >> 1) * ROW_B;
      tfq[n] = cat4(lds_tr_b64(reinterpret_cast<const bf16*>(tbase + qtr[n & 1][0] + off)),
                    lds_tr_b64(reinterpret_cast<const bf16*>(tbase + qtr[n & 1][1] + off)));
      tfd[n] = cat4(lds_tr_b64(reinterpret_cast<const bf16*>(tbase + SUPER_B + qtr[n & 1][0] + off)),
                    lds_tr_b64(reinterpret_cast<const bf16*>(tbase + SUPER_B + qtr[n & 1][1] + off)));
    };
    auto load_g = [&](int G) {                                   // -delta and row hashes of group G (queries 8 G + 4 hh + 0..3)
      nd[G & 1] = *reinterpret_cast<const f32x4*>(tab + 512 + G * 32);
      if (DROPOUT) rh[G & 1] = *reinterpret_cast<const u32x4_t*>(tab + 1024 + G * 32);
    };
    // phase 1
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) da[ks] = *reinterpret_cast<const bf16x8*>(tbase + SUPER_B + qnat[ks]);
    ls = *reinterpret_cast<const f32x4*>(tab);
    SB();
#pragma unroll
    for (int G = 0; G < 4; ++G) {
      f32x4 ln = ls;
      if (G < 3) ln = *reinterpret_cast<const f32x4*>(tab + (G + 1) * 32);
      if (NEXT && G == 2) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qa[ks] = *reinterpret_cast<const bf16x8*>(nbase + qnat[ks]);
      }
      if (G == 3) load_g(0);
#pragma unroll
      for (int e = 0; e < 4; ++e) cs[4 * G + e] = __builtin_amdgcn_exp2f(fmaf(cs[4 * G + e], p.c, -ls[e]));
      ls = ln;
      SB();
      dp = mfma32(da[G], vf[G], G == 0 ? zero16 : dp);
      SB();
    }
    // phase 2
#pragma unroll
    for (int G = 0; G < 4; ++G) {
      if (G < 3) load_g(G + 1);
      if (G == 0) { load_t(0); load_t(1); }
      if (G == 2) { load_t(2); load_t(3); }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int n = 2 * G + hf, r = 4 * G + 2 * hf;
        float p0 = cs[r], p1 = cs[r + 1], d0 = dp[r], d1 = dp[r + 1], pd0 = p0, pd1 = p1;
        if (DROPOUT) {                                           // product scheme: this lane's column multiplier x the rows' hashes
          const bool k0 = drop_keep(rh[G & 1][2 * hf], colm, thr32), k1 = drop_keep(rh[G & 1][2 * hf + 1], colm, thr32);
          pd0 = k0 ? p0 : 0.f; pd1 = k1 ? p1 : 0.f;             // dV's 1 / keep factor is applied once, when dV is stored
          d0 = fma1(k0 ? d0 : 0.f, p.inv_keep, nd[G & 1][2 * hf]);
          d1 = fma1(k1 ? d1 : 0.f, p.inv_keep, nd[G & 1][2 * hf + 1]);
        } else {
          d0 = add1(d0, nd[G & 1][2 * hf]);
          d1 = add1(d1, nd[G & 1][2 * hf + 1]);
        }
        pw[G >> 1][2 * (G & 1) + hf] = pack_bf16x2(pd0, pd1);
        dsw[G >> 1][2 * (G & 1) + hf] = pack_bf16x2(mul1(p0, d0), mul1(p1, d1));
        SB();
        if (NEXT && n < 4) ns = mfma32(qa[n], kf[n], n == 0 ? zero16 : ns);
        if (n == 4 || n == 5) {
          dvt[n - 4] = mfma32(tfd[n - 4], __builtin_bit_cast(bf16x8, pw[0]), dvt[n - 4]);
          dkt[n - 4] = mfma32(tfq[n - 4], __builtin_bit_cast(bf16x8, dsw[0]), dkt[n - 4]);
        }
        SB();
      }
    }
    dvt[0] = mfma32(tfd[2], __builtin_bit_cast(bf16x8, pw[1]), dvt[0]);
    dkt[0] = mfma32(tfq[2], __builtin_bit_cast(bf16x8, dsw[1]), dkt[0]);
    dvt[1] = mfma32(tfd[3], __builtin_bit_cast(bf16x8, pw[1]), dvt[1]);
    dkt[1] = mfma32(tfq[3], __builtin_bit_cast(bf16x8, dsw[1]), dkt[1]);
  };

  auto step = [&](auto st_c, int i) {
    constexpr int ST = decltype(st_c)::value;                   // i & 1
    const bool have_next = i + 1 < nsup;
    if (have_next) {
      issue(i + 1);
      rstat = load_stat(i + 1);
    }
    const bool active = i >= i_first && i <= i_last, more = active && i + 1 <= i_last;
    if (active) {
      block_body(std::true_type{}, st_c, std::integral_constant<int, 0>{}, sa, sb);
      if (needs_mask(i)) mask_block(sb, i, 1);
    }
    lds_dma_wait_all();
    if (have_next) store_stat(i + 1, rstat);
    __syncthreads();                                            // super tile i + 1 and its tables are in LDS
    if (active) {
      if (more) {
        block_body(std::true_type{}, st_c, std::integral_constant<int, 1>{}, sb, sa);
        if (needs_mask(i + 1)) mask_block(sa, i + 1, 0);
      } else {
        block_body(std::false_type{}, st_c, std::integral_constant<int, 1>{}, sb, sa);
      }
    } else if (i + 1 == i_first && i + 1 <= i_last) {           // the wave's first tile is the next one: its first block, unpipelined
      scores(sa, ST ^ 1, 0);
      if (needs_mask(i + 1)) mask_block(sa, i + 1, 0);
    }
    __syncthreads();                                            // everyone is done with stage ST
  };
  for (int i = 0; i < nsup; i += 2) {
    step(std::integral_constant<int, 0>{}, i);
    if (i + 1 < nsup) step(std::integral_constant<int, 1>{}, i + 1);
  }

  // sum the two query parts, scale, convert; whole rows leave through LDS staging tiles
  float* mb = reinterpret_cast<float*>(smem);                    // [64][128] words
  const int ml = kh * 64 + lane;
  if (qp == 1) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        mb[(nb * 16 + r) * 128 + ml] = dkt[nb][r];
        mb[(32 + nb * 16 + r) * 128 + ml] = dvt[nb][r];
      }
  }
  __syncthreads();
  unsigned char* ost = smem + 64 * 128 * 4 + kh * (2 * 32 * 144);  // [dK | dV][32 keys][144-byte rows] per key half
  if (qp == 0) {
    const float vscale = DROPOUT ? p.inv_keep : 1.0f;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        bf16x4 ok, ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * qd + e;
          ok[e] = (bf16)((dkt[nb][r] + mb[(nb * 16 + r) * 128 + ml]) * p.scale);
          ov[e] = (bf16)((dvt[nb][r] + mb[(32 + nb * 16 + r) * 128 + ml]) * vscale);
        }
        const int off = (lane & 31) * 144 + (nb * 32 + 8 * qd + 4 * hh) * 2;
        *reinterpret_cast<bf16x4*>(ost + off) = ok;
        *reinterpret_cast<bf16x4*>(ost + 32 * 144 + off) = ov;
      }
  }
  __syncthreads();
  if (qp == 0) {
    bf16* dk = p.dk + (int64_t)b * p.sb + h * DH;
    bf16* dv = p.dv + (int64_t)b * p.sb + h * DH;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = i * 8 + (lane >> 3), ch = lane & 7;
      const bf16x8 vk = *reinterpret_cast<const bf16x8*>(ost + row * 144 + ch * 16);
      const bf16x8 vv = *reinterpret_cast<const bf16x8*>(ost + 32 * 144 + row * 144 + ch * 16);
      if (k_base + row < p.S) {
        *reinterpret_cast<bf16x8*>(dk + (int64_t)(k_base + row) * p.ss + ch * 8) = vk;
        *reinterpret_cast<bf16x8*>(dv + (int64_t)(k_base + row) * p.ss + ch * 8) = vv;
      }
    }
  }
}

constexpr size_t DKDV_SMEM = TAB_OFF + 2 * STAT_B;
constexpr size_t FWD_SMEM = TAB_OFF + 2 * 128 * sizeof(uint32_t);

}  // namespace dh64

int attn_fwd_dh64(const AttnParams& p, hipStream_t s) {
  const int grid = ((p.S + 63) / 64) * p.H * p.B;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(dh64::attn_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(dh64::attn_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  if (p.thr) dh64::attn_fwd_kernel<true><<<grid, 256, dh64::FWD_SMEM, s>>>(p);
  else dh64::attn_fwd_kernel<false><<<grid, 256, dh64::FWD_SMEM, s>>>(p);
  return check_launch("attn_fwd_dh64");
}

int attn_bwd_dkdv_dh64(const AttnParams& p, hipStream_t s) {
  const int grid = ((p.S + 63) / 64) * p.H * p.B;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(dh64::attn_bwd_dkdv_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(dh64::attn_bwd_dkdv_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  if (p.thr) dh64::attn_bwd_dkdv_kernel<true><<<grid, 256, dh64::DKDV_SMEM, s>>>(p);
  else dh64::attn_bwd_dkdv_kernel<false><<<grid, 256, dh64::DKDV_SMEM, s>>>(p);
  return check_launch("attn_bwd_dkdv_dh64");
}

int attn_bwd_dq_dh64(const AttnParams& p, hipStream_t s) {
  const int grid = ((p.S + 63) / 64) * p.H * p.B;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(dh64::attn_bwd_dq_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(dh64::attn_bwd_dq_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  if (p.thr) dh64::attn_bwd_dq_kernel<true><<<grid, 256, dh64::FWD_SMEM, s>>>(p);
  else dh64::attn_bwd_dq_kernel<false><<<grid, 256, dh64::FWD_SMEM, s>>>(p);
  return check_launch("attn_bwd_dq_dh64");
}

}  // namespace ttts
