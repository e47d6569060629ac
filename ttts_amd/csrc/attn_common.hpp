// Shared declarations of the causal self-attention kernels (attn.hip: every head_dim; attn_dh64.hip: the head_dim 64 fast path).
#pragma once
#include "common.hpp"

namespace ttts {

struct AttnParams {
  const bf16 *q, *k, *v;
  const bf16 *o, *d_o;
  bf16 *out, *dq, *dk, *dv;
  float* lse;
  const float* lse_in;
  float* delta;
  int B, H, S;
  int Sp;            // S rounded up to 4: row pitch of the dropout-mask index space
  int64_t sb, ss;    // q/k/v (and dq/dk/dv) strides: batch, sequence
  int64_t osb, oss;  // o / dO strides
  float scale, c;    // softmax scale, scale * log2(e)
  uint32_t thr;      // dropout threshold on 16 random bits (0 = no dropout)
  float inv_keep;
  uint32_t seed_lo, seed_hi;
  const uint32_t* ctr;  // caller-owned dropout stream counter (device) or NULL
};

constexpr float NEG_BIG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

template <int DH> struct AttnCfg {
  static constexpr int KSTR = DH + 8;                      // natural (ds_read_b128) tiles, elements
  static constexpr int VSTR = (DH == 32) ? 32 : DH + 32;   // transposed-read-only tiles
  static constexpr int KS = DH / 16;                       // MFMA k-steps across the head dim
  static constexpr int NB = DH / 32;                       // 32-wide output blocks across the head dim
  static constexpr int CPT = DH / 32;                      // 16-byte chunks per thread for a 64 x DH tile
  static constexpr int CPR = DH / 8;                       // 16-byte chunks per row
};

// two fp32 -> one packed bf16 pair (v_cvt_pk_bf16_f32); the probability fragments are assembled from these words
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// Attention dropout keep-mask (round 2): PRODUCT scheme.  Every query row gets one strong 32-bit hash R with bit 23 set (of
// (b*H + h)*S + query), every key column one odd 24-bit multiplier M with its top bit set (a hash of (b*H + h)*S + key under a
// tweaked seed); the pair (query, key) is kept iff  (R[23:0] * M + R) mod 2^32  >=  thr << 16  -- one v_mad_u32_u24, one
// compare, one select per score (the pair-shared counter hash of round 1 cost ~9 VALU per score: more cycles than the
// tile's MFMAs).  For a fixed column the map R -> R*M is a bijection of the low 24 bits and wraps >= 2^15 times around 2^32, for
// a fixed row the M are independent: keep rate, 256-bin chi-square, lag-1..8 row / column correlations, the 2 x 2
// interaction and per-row / per-column rates are indistinguishable from independent Bernoulli draws in a numpy emulation
// (tests/test_host_cpu.py::test_attention_dropout_product_scheme_statistics).  Both layouts evaluate it cheaply: the
// lane-owned operand is hashed once per kernel, the other comes from a 64-entry LDS table filled once per tile.
// (bit 23 of the row word is forced, like the multiplier's: with a small R[23:0] -- one row in 2^15 -- R[23:0] * M spans less than
// one wrap of 2^32 and the whole row would be kept or dropped together)
__device__ __forceinline__ uint32_t drop_row_hash(uint32_t rowid, uint32_t slo, uint32_t shi) { return hash32(rowid, slo, shi) | 0x800000u; }
__device__ __forceinline__ uint32_t drop_col_mult(uint32_t colid, uint32_t slo, uint32_t shi) {
  return (hash32(colid, slo ^ 0x5BD1E995u, shi) & 0xFFFFFFu) | 0x800001u;
}
__device__ __forceinline__ bool drop_keep(uint32_t rowh, uint32_t colm, uint32_t thr32) {
  return __umul24(rowh, colm) + rowh >= thr32;      // v_mad_u32_u24 uses the low 24 bits of both factors
}
// lane = query layouts: four consecutive keys' multipliers from the tile table
__device__ __forceinline__ void drop_keep4(uint32_t rowh, const uint32_t* colm4, uint32_t thr32, bool (&k)[4]) {
  const u32x4_t m = *reinterpret_cast<const u32x4_t*>(colm4);
#pragma unroll
  for (int e = 0; e < 4; ++e) k[e] = drop_keep(rowh, m[e], thr32);
}

}  // namespace ttts
