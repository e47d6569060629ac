// Causal self-attention for head_dim 64 (the GPT train step's shape: B 8, H 8, S 1156, dropout 0.1) -- round-3 kernels.
// Replaces GPT2Attention's core (transformers modeling_gpt2.py:53-72) as reached from ttts/gpt/model.py:422.
//
// What is different from the generic kernels of attn.hip (which still serve head_dim 32 / 128):
//  * K / V (Q / dO) tiles reach LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no address VALU, no ds_write.
//    The DMA image is lane-linear, so rows are 128 bytes apart and UNPADDED; bank conflicts are removed by an XOR swizzle of
//    the 16-byte chunk index, chunk' = chunk ^ bitrev3(row[3:1]), applied on the DMA's per-lane SOURCE address and on every
//    read.  One swizzle serves both readers: ds_read_b128 of a row fragment (16 lanes = 16 rows distinct in bits 0..3 ->
//    16 distinct 16-byte slots of the 256-byte bank span) and ds_read_b64_tr_b16 (32 lanes = 4 rows x 64 bytes; row bit 1
//    flips chunk bit 2, so rows r and r + 2 land in different 64-byte quarters).  Tiles are 64 rows (8 KB) in 3-slot rings.
//  * Every wave is software-pipelined per 32-row BLOCK of the streamed axis: the score MFMAs of block b + 1 and block b's
//    own output MFMAs are issued between the half-groups of block b's softmax / dS VALU work, slot by slot
//    (sched_barrier-pinned), so a wave keeps the matrix pipe and the VALU busy at the same time.
//  * Lazy rescale (forward): the running maximum is only raised (and O / l rescaled) when a block's maximum exceeds it by
//    more than 2^8; probabilities are then bounded by 2^8 instead of 1, which bf16 / f32 represent with the same relative
//    precision.  The decision for block b + 1 is taken after block b's P.V MFMAs are issued, and rescales O, l together.
//  * What bounds these kernels (measured): a wave issues one instruction per ~4 cycles whatever its
//    kind, the heaviest workgroup walks 19-20 tiles, and the VALU work of softmax + dropout (7.4 instructions per score) is
//    ~3x the MFMA time at head_dim 64 -- so the step loops are written for instruction count: straight-line phases instead of
//    per-step conditionals, ring addresses advanced in place, DMA sources kept as running pointers.
// Operand layouts (v_mfma_f32_32x32x16_bf16) are those of attn.hip: S^T = K.Q^T keeps one query per lane, the exponentiated
// accumulator registers are the B operand of O^T += V^T.P^T with V^T read by the hardware transpose read.
#include <stdlib.h>

#include <type_traits>

#include "attn_common.hpp"

namespace ttts {
namespace dh64 {

constexpr int DH = 64;
constexpr int ROW_B = DH * 2;                    // bytes per tile row
constexpr int TILE64_B = 64 * ROW_B;             // 64-row tile: 8 KB
constexpr int RING_B = 3 * TILE64_B;             // a ring of three tiles: 24 KB
constexpr int LDS_A = 0, LDS_B = RING_B, LDS_TAB = 2 * RING_B;   // ring A (K / Q), ring B (V / dO), tables
constexpr float RESCALE_LOG2 = 8.0f;             // lazy rescale threshold, log2 units

// chunk swizzle: bit-reversed row bits 3..1
__device__ __forceinline__ int swz3(int row) { return (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1); }

// LDS-DMA of one 1-KiB chunk (8 rows x 128 bytes): lane i lands at lds + 16 i.  Untracked inline asm (see common.hpp).  The
// base must come from SALU arithmetic on values made uniform at kernel entry (a v_readfirstlane right in front of the
// statement would need 5 wait states before the load reads the SGPR pair; hipcc pads nothing inside or around asm).
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

// Per-lane source offsets (bytes) of this wave's two chunks of a 64-row tile: chunk c = 2 wave + i covers rows 8 c .. 8 c + 7;
// lane = (row r8 = lane >> 3, physical chunk pc = lane & 7) fetches logical chunk pc ^ swz3(row).  rows_valid: rows of the
// tile inside the sequence (>= 1); rows beyond repeat the last valid one (only masked scores ever meet them).
__device__ __forceinline__ void dma_offsets(uint32_t (&voff)[2], int lane, int wave, int64_t stride, int rows_valid) {
  const int r8 = lane >> 3, pc = lane & 7;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 8 * (2 * wave + i) + r8;
    voff[i] = (uint32_t)(min(row, rows_valid - 1) * (int)stride * 2 + ((pc ^ swz3(row)) << 4));
  }
}
// this wave's quarter of one tile: src = the tile's first row, lds_tile = the ring slot's LDS byte address
__device__ __forceinline__ void dma_tile(const uint32_t (&voff)[2], const bf16* src, uint32_t lds_tile, int wave) {
#pragma unroll
  for (int i = 0; i < 2; ++i) dma16(voff[i], src, __builtin_amdgcn_readfirstlane(lds_tile + (2 * wave + i) * 1024));
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
// max / sum of a value with the other lane half's (lane ^ 32): one v_permlane32_swap instead of a ds_bpermute round trip.
// swap(A = v, B = v) leaves A = [v.lo, v.lo], B = [v.hi, v.hi] (the builtin pads the VALU -> permlane hazard itself).
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
__device__ __forceinline__ float xhalf_max(float v) {
  const uint32_t u = __builtin_bit_cast(uint32_t, v);
  const u32x2_t r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const uint32_t lo = r[0], hi = r[1];   // (bit_cast applied to r[1] directly reads element 0: a hipcc quirk)
  return fmaxf(__builtin_bit_cast(float, lo), __builtin_bit_cast(float, hi));
}
// This file is built with -fno-honor-nans (no v_max canonicalisation in front of fmaxf on MFMA outputs: nothing here produces
// or consumes a NaN -- masked scores are the finite NEG_BIG) and -fno-slp-vectorize (hipcc would pack adjacent f32 adds /
// multiplies into v_pk_*_f32, which issue slower beside MFMAs than the scalar pair: MI355X_MICROARCH price list).  VALU work is
// NOT written as inline asm: hipcc pads no hazards for an asm statement (v_exp -> consumer, MFMA result -> VALU reader), which
// showed up as stale exponentials on lanes 0-3 of every 8 in one build.
__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
// two scores -> one packed bf16 probability pair; row-sum partials in l[0], l[1]
template <bool DROPOUT>
__device__ __forceinline__ uint32_t sm2(float s0, float s1, float c, float m2, float (&l)[2], uint32_t rowh, uint32_t thr32,
                                        uint32_t cm0, uint32_t cm1) {
  float p0 = __builtin_amdgcn_exp2f(fmaf(s0, c, -m2)), p1 = __builtin_amdgcn_exp2f(fmaf(s1, c, -m2));
  l[0] += p0;
  l[1] += p1;
  if (DROPOUT) {                                    // 1 / keep is applied once, with 1 / l, at the end
    p0 = drop_keep(rowh, cm0, thr32) ? p0 : 0.f;
    p1 = drop_keep(rowh, cm1, thr32) ? p1 : 0.f;
  }
  return pack_bf16x2(p0, p1);
}
#define SB() __builtin_amdgcn_sched_barrier(0)

// two probabilities + two dP values -> one packed bf16 dS pair: dS = P o (dropout'(dP) - delta)
template <bool DROPOUT>
__device__ __forceinline__ uint32_t ds2(float p0, float p1, float dp0, float dp1, float ndelta, float inv_keep, uint32_t rowh,
                                        uint32_t thr32, uint32_t cm0, uint32_t cm1) {
  float t0, t1;
  if (DROPOUT) {                                    // select first, then ONE fma for the 1 / keep scale and the subtraction
    t0 = fmaf(drop_keep(rowh, cm0, thr32) ? dp0 : 0.f, inv_keep, ndelta);
    t1 = fmaf(drop_keep(rowh, cm1, thr32) ? dp1 : 0.f, inv_keep, ndelta);
  } else {
    t0 = (dp0 + ndelta);
    t1 = (dp1 + ndelta);
  }
  return pack_bf16x2((p0 * t0), (p1 * t1));
}

// -------------------------------------------------------------------------------------------------------
// forward: 128-query workgroups; wave w owns queries q0 + 32 w .. + 31 and walks all its 64-key tiles (a split of the key
// range over wave pairs was measured: the per-wave prologue / merge / epilogue made up ~40 % of that kernel's VALU work; so was
// a 64-query workgroup whose wave pairs split every tile's two 32-key blocks, one block per step: half the serial chain, +25 %
// instructions, 34 us against this kernel's 30 -- DESIGN section 16 has the issue-rate model that explains both).
// Pipelined per 32-key BLOCK: block b's softmax runs beside block b + 1's 4 score MFMAs and block b's own P.V MFMAs --
// 32 score registers in flight, 144 VGPRs, three workgroups per CU.
// K / V tiles live in two 3-slot rings.  Step j handles blocks 2j - 1 and 2j, which read K(j) only (for the next blocks'
// scores) and V(j - 1), V(j); at its top the wave issues V(j + 1) and then K(j + 2), at its end it waits with vmcnt(2):
// everything but the two K(j + 2) pieces has landed, i.e. K has two steps and V one step of flight.
// -------------------------------------------------------------------------------------------------------
template <bool DROPOUT>
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lds0 = lds_byte_addr(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5;
  const int nqb = (p.S + 127) / 128;
  const int nbh = gridDim.x / nqb;
  const int qb = nqb - 1 - (int)(blockIdx.x / nbh);             // longest workgroups first
  const int bh = blockIdx.x % nbh, h = bh % p.H, b = bh / p.H;
  const int q0 = qb * 128, q_base = q0 + wave * 32;
  const int query = q_base + (lane & 31);
  const bf16* kp = uniform_ptr(p.k + (int64_t)b * p.sb + h * DH);
  const bf16* vp = uniform_ptr(p.v + (int64_t)b * p.sb + h * DH);
  const int nt = min(2 * qb + 2, (p.S + 63) / 64);              // 64-key tiles this workgroup streams
  const int nt_w = q_base < p.S ? (q_base >> 6) + 1 : 0;        // this wave's tiles; the last one holds its diagonal
  const int nblk = 2 * nt_w;                                    // 32-key blocks
  const int64_t tile_el = (int64_t)64 * p.ss;

  uint32_t voff[2];
  dma_offsets(voff, lane, wave, p.ss, 64);
  const bool ragged = nt * 64 > p.S;                             // the last tile reaches past the sequence
  // tile js of K or V -> ring slot (byte offset `slot`); src points at the tile's first row
  auto issue = [&](int js, const bf16* src, int ring0, int slot) {
    if (ragged && js == nt - 1) {
      uint32_t vl[2];
      dma_offsets(vl, lane, wave, p.ss, p.S - js * 64);
      dma_tile(vl, src, lds0 + ring0 + slot, wave);
    } else {
      dma_tile(voff, src, lds0 + ring0 + slot, wave);
    }
  };
  issue(0, vp, LDS_B, 0);
  issue(0, kp, LDS_A, 0);
  if (nt > 1) issue(1, kp + tile_el, LDS_A, TILE64_B);

  bf16x8 qf[4];
  {
    const bf16* qrow = p.q + (int64_t)b * p.sb + h * DH + (int64_t)min(query, p.S - 1) * p.ss + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qrow + ks * 16);
  }
  const uint32_t shi = DROPOUT ? seed_mix(p.seed_hi, p.ctr) : 0u;
  const uint32_t id_bh = (uint32_t)((b * p.H + h) * p.S), thr32 = p.thr << 16;
  const uint32_t rowh = DROPOUT ? drop_row_hash(id_bh + (uint32_t)query, p.seed_lo, shi) : 0u;
  uint32_t* Bm = reinterpret_cast<uint32_t*>(smem + LDS_TAB);
  if (DROPOUT && tid < 64) Bm[tid] = drop_col_mult(id_bh + (uint32_t)tid, p.seed_lo, shi);

  // LDS addresses of the current ring slots (advanced in place every step): K(j) row fragments, V(j) / V(j - 1) transposed fragments
  int ka[4], va[2][2], vb[2][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) ka[ks] = LDS_A + nat_off(lane, ks);
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int j = 0; j < 2; ++j) va[nb][j] = vb[nb][j] = LDS_B + tr_off(lane, nb, j);
  const int bm_lane = LDS_TAB + 4 * hh * 4;
  const int qrel = (q_base & 32) + (lane & 31) - 4 * hh;         // query - first key of the diagonal tile - 4 hh

  f32x16 ot[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[nb][r] = 0.f;
  float m = NEG_BIG, m2 = NEG_BIG, l[2] = {0.f, 0.f};
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  auto mask_blk = [&](f32x16& s, int kb) {                      // diagonal tile, block kb
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (kb * 32 + (r & 3) + 8 * (r >> 2) > qrel) s[r] = NEG_BIG;
    asm volatile("" ::: "memory");
  };
  // raise the running max if this block's max exceeds it by more than the threshold (wave-uniform, rare after the first
  // blocks): O and l, everything accumulated at the old max, are rescaled together
  auto decide = [&](const f32x16& s) {
    float a = max3(s[0], s[1], s[2]), c2 = max3(s[3], s[4], s[5]);
    a = max3(a, s[6], s[7]); c2 = max3(c2, s[8], s[9]);
    a = max3(a, s[10], s[11]); c2 = max3(c2, s[12], s[13]);
    a = max3(a, s[14], s[15]);
    a = fmaxf(a, c2);
    const float mx = xhalf_max(a);
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
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) asm volatile("" :: "v"(qf[ks]));   // see attn_fwd_kernel
  __syncthreads();

  f32x16 sa, sb;
  if (nblk > 0) {                                               // block 0, unpipelined
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) sa = mfma32(*reinterpret_cast<const bf16x8*>(smem + ka[ks]), qf[ks], ks == 0 ? zero16 : sa);
    if (nt_w == 1) mask_blk(sa, 0);
    decide(sa);
  }

  // One pipelined block: softmax of `cur` (block kb = 1 - KBN of the tile whose V / Bm addresses are given) beside the 4
  // score MFMAs of the next block (block KBN of the tile at kaddr) and beside its own P.V MFMAs.  8 half-groups of 2 scores.
  // The first fragments a body needs (two K row fragments, the first four column multipliers) are fetched by its predecessor
  // -- or at the top of the step, behind the barrier that publishes K(j) -- so that no body opens with an LDS round trip.
  bf16x8 kfp[2];
  u32x4_t cmp0 = {0u, 0u, 0u, 0u};
  auto prefetch = [&](int kbn, const int (&ka)[4], int bma_blk) {
    kfp[0] = *reinterpret_cast<const bf16x8*>(smem + ka[0] + kbn * 32 * ROW_B);
    kfp[1] = *reinterpret_cast<const bf16x8*>(smem + ka[1] + kbn * 32 * ROW_B);
    if (DROPOUT) cmp0 = *reinterpret_cast<const u32x4_t*>(smem + bma_blk);
  };
  auto body = [&](auto next_c, auto kbn_c, auto pre_c, const f32x16& cur, f32x16& nxt, const int (&ka)[4], const int (&va)[2][2], int bma,
                  int bma_pre) {
    constexpr bool NEXT = decltype(next_c)::value, PRE = decltype(pre_c)::value;
    constexpr int KBN = decltype(kbn_c)::value, KBC = KBN ^ 1;
    bf16x8 kf[4], vf[4];
    u32x4_t cm[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    u32x4_t pw[2];
    auto load_k = [&](int ks) { kf[ks] = *reinterpret_cast<const bf16x8*>(smem + ka[ks] + KBN * 32 * ROW_B); };
    auto load_v = [&](int n) {                                   // fragment of P.V MFMA n: cs = n >> 1, nb = n & 1
      const int off = (KBC * 32 + 16 * (n >> 1)) * ROW_B;
      vf[n] = cat4(lds_tr_b64(reinterpret_cast<const bf16*>(smem + va[n & 1][0] + off)),
                   lds_tr_b64(reinterpret_cast<const bf16*>(smem + va[n & 1][1] + off)));
    };
    auto load_cm = [&](int G) { if (DROPOUT) cm[G & 1] = *reinterpret_cast<const u32x4_t*>(smem + bma + KBC * 128 + G * 32); };
    kf[0] = kfp[0]; kf[1] = kfp[1];
    cm[0] = cmp0;
#pragma unroll
    for (int G = 0; G < 4; ++G) {
      if (G < 3) load_cm(G + 1);
      if (NEXT && G == 0) { load_k(2); load_k(3); }              // fragments are fetched two slots ahead of their MFMA
      if (G == 1) load_v(0);
      if (G == 2) load_v(1);
      if (G == 3) { load_v(2); load_v(3); }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int n = 2 * G + hf;
        pw[G >> 1][2 * (G & 1) + hf] = sm2<DROPOUT>(cur[4 * G + 2 * hf], cur[4 * G + 2 * hf + 1], p.c, m2, l, rowh, thr32,
                                                    cm[G & 1][2 * hf], cm[G & 1][2 * hf + 1]);
        SB();
        if (NEXT && n < 4) nxt = mfma32(kf[n], qf[n], n == 0 ? zero16 : nxt);
        if (n == 4 || n == 5) ot[n - 4] = mfma32(vf[n - 4], __builtin_bit_cast(bf16x8, pw[0]), ot[n - 4]);
        SB();
      }
    }
    if (PRE) prefetch(1, ka, bma_pre);                           // for the step's second body: K(j) block 1, Bm(j) block 0
    ot[0] = mfma32(vf[2], __builtin_bit_cast(bf16x8, pw[1]), ot[0]);
    ot[1] = mfma32(vf[3], __builtin_bit_cast(bf16x8, pw[1]), ot[1]);
  };

  // Steps.  For this wave: step 0 runs body B only, steps 1 .. nt_w - 1 run A then B, step nt_w runs the last A, later steps
  // (waves whose diagonal comes earlier than the workgroup's last tile) only keep the DMA and the barriers going.  Blocks
  // 2 nt_w - 2 and 2 nt_w - 1 form the diagonal tile: they are masked as they come out of step nt_w - 1's two bodies.
  const bf16* vsrc = vp + tile_el;                               // V(j + 1), K(j + 2): sources and ring slots of this step's DMA
  const bf16* ksrc = kp + 2 * tile_el;
  int vs_next = TILE64_B, ks_next = 2 * TILE64_B;
  int j = 0;
  auto step_open = [&]() {
    if (j + 1 < nt) {
      issue(j + 1, vsrc, LDS_B, vs_next);                     // V(j + 1) replaces V(j - 2): last read in step j - 1
      if (DROPOUT && wave == ((j + 1) & 3)) Bm[((j + 1) & 3) * 64 + lane] = drop_col_mult(id_bh + (uint32_t)((j + 1) * 64 + lane), p.seed_lo, shi);
    }
    if (j + 2 < nt) issue(j + 2, ksrc, LDS_A, ks_next);       // K(j + 2) replaces K(j - 1): read in step j - 1
    vsrc += tile_el; ksrc += tile_el;
    vs_next = ks_next;
    ks_next = ks_next == 2 * TILE64_B ? 0 : ks_next + TILE64_B;
  };
  auto step_close = [&]() {
    // all but the two K(j + 2) pieces -- when this step issued them: near the end of the walk the newest pieces are V(j + 1),
    // which the next step reads (found as a run-to-run difference of a few thousand output elements: tools/exp/fwd_determinism.py)
    if (j + 2 < nt) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else lds_dma_wait_all();
    __syncthreads();
    // step_open has rotated vs_next to the slot of tile j + 2: it is slot 1 exactly when tile j sits in slot 2 (wave-uniform)
    const int adv = vs_next == TILE64_B ? -2 * TILE64_B : TILE64_B;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ka[ks] += adv;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) { vb[nb][jj] = va[nb][jj]; va[nb][jj] += adv; }
    ++j;
  };
  const auto T = std::true_type{};
  const auto F = std::false_type{};
  const auto I0 = std::integral_constant<int, 0>{};
  const auto I1 = std::integral_constant<int, 1>{};
  // step 0
  if (nt_w > 0) prefetch(1, ka, bm_lane);
  step_open();
  if (nt_w > 0) {
    body(T, I1, F, sa, sb, ka, va, bm_lane, 0);
    if (nt_w == 1) mask_blk(sb, 1);
    decide(sb);
  }
  step_close();
  // steps 1 .. nt_w - 2: no masks
  for (; j + 1 < nt_w;) {
    const int bma = bm_lane + ((j - 1) & 3) * 256, bmb = bm_lane + (j & 3) * 256;
    prefetch(0, ka, bma + 128);
    step_open();
    body(T, I0, T, sb, sa, ka, vb, bma, bmb);
    decide(sa);
    body(T, I1, F, sa, sb, ka, va, bmb, 0);
    decide(sb);
    step_close();
  }
  // step nt_w - 1 (if it is not step 0): its bodies produce the diagonal tile's scores
  if (nt_w >= 2) {
    const int bma = bm_lane + ((j - 1) & 3) * 256, bmb = bm_lane + (j & 3) * 256;
    prefetch(0, ka, bma + 128);
    step_open();
    body(T, I0, T, sb, sa, ka, vb, bma, bmb);
    mask_blk(sa, 0);
    decide(sa);
    body(T, I1, F, sa, sb, ka, va, bmb, 0);
    mask_blk(sb, 1);
    decide(sb);
    step_close();
  }
  // step nt_w: the wave's last block
  if (nt_w > 0) {
    const int bma = bm_lane + ((j - 1) & 3) * 256;
    cmp0 = *reinterpret_cast<const u32x4_t*>(smem + bma + 128);
    step_open();
    body(F, I0, F, sb, sa, ka, vb, bma, 0);
    step_close();
  }
  while (j <= nt) {                                             // the rest of the workgroup is still walking
    step_open();
    step_close();
  }

  // normalise, convert; whole 128-byte rows leave through a per-wave LDS staging tile (the rings are idle now)
  float lt = l[0] + l[1];
  lt += xhalf(lt);
  unsigned char* ost = smem + wave * (32 * 144);
  {
    const float inv = (DROPOUT ? p.inv_keep : 1.0f) / lt;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)(ot[nb][4 * qd + e] * inv);
        *reinterpret_cast<bf16x4*>(ost + (lane & 31) * 144 + (nb * 32 + 8 * qd + 4 * hh) * 2) = o;
      }
    if (hh == 0 && query < p.S) p.lse[(int64_t)(b * p.H + h) * p.S + query] = (m * p.c + __log2f(lt)) * LN2;
  }
  __syncthreads();
  {
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
// backward: dQ (S^T form, one query per lane; the forward kernel's work split, rings and step structure)
//   per 32-key block:  S^T = K.Q^T, dP^T = V.dO^T (8 MFMAs), dS = P o (dropout'(dP) - delta), dQ^T += K^T.dS^T (4 MFMAs).
// Two-phase block pipeline (48 accumulator registers in flight):
//   phase 1  P = exp2(S c - lse) in place (32 VALU)             beside the block's own 4 dP MFMAs,
//   phase 2  dS from P, dP, the dropout mask (~90 VALU)          beside the NEXT block's 4 S MFMAs and this block's dQ MFMAs.
// Step j handles blocks 2j - 1 (tile j - 1) and 2j (tile j): it reads K / V of tiles j - 1 and j, while tile j + 1 (issued at
// the top of the step into the third ring slot) is in flight; one vmcnt(0) + barrier per step.
// -------------------------------------------------------------------------------------------------------
template <bool DROPOUT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lds0 = lds_byte_addr(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5;
  const int nqb = (p.S + 127) / 128;
  const int nbh = gridDim.x / nqb;
  const int qb = nqb - 1 - (int)(blockIdx.x / nbh);             // longest workgroups first
  const int bh = blockIdx.x % nbh, h = bh % p.H, b = bh / p.H;
  const int q_base = qb * 128 + wave * 32;
  const int query = q_base + (lane & 31), queryc = min(query, p.S - 1);
  const bf16* kp = uniform_ptr(p.k + (int64_t)b * p.sb + h * DH);
  const bf16* vp = uniform_ptr(p.v + (int64_t)b * p.sb + h * DH);
  const int nt = min(2 * qb + 2, (p.S + 63) / 64);              // 64-key tiles this workgroup streams
  const int nt_w = q_base < p.S ? (q_base >> 6) + 1 : 0;        // this wave's tiles; the last one holds its diagonal
  const int64_t tile_el = (int64_t)64 * p.ss;

  uint32_t voff[2];
  dma_offsets(voff, lane, wave, p.ss, 64);
  const bool ragged = nt * 64 > p.S;
  auto issue = [&](int js, const bf16* ksrc, const bf16* vsrc, int slot) {   // K(js), V(js) -> ring slot
    if (ragged && js == nt - 1) {
      uint32_t vl[2];
      dma_offsets(vl, lane, wave, p.ss, p.S - js * 64);
      dma_tile(vl, ksrc, lds0 + LDS_A + slot, wave);
      dma_tile(vl, vsrc, lds0 + LDS_B + slot, wave);
    } else {
      dma_tile(voff, ksrc, lds0 + LDS_A + slot, wave);
      dma_tile(voff, vsrc, lds0 + LDS_B + slot, wave);
    }
  };
  issue(0, kp, vp, 0);

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
  const float lse2 = p.lse_in[stat] * LOG2E, ndelta = -p.delta[stat];
  const uint32_t shi = DROPOUT ? seed_mix(p.seed_hi, p.ctr) : 0u;
  const uint32_t id_bh = (uint32_t)((b * p.H + h) * p.S), thr32 = p.thr << 16;
  const uint32_t rowh = DROPOUT ? drop_row_hash(id_bh + (uint32_t)query, p.seed_lo, shi) : 0u;
  uint32_t* Bm = reinterpret_cast<uint32_t*>(smem + LDS_TAB);    // [4][64] dropout column multipliers, slot = tile & 3
  if (DROPOUT && tid < 64) Bm[tid] = drop_col_mult(id_bh + (uint32_t)tid, p.seed_lo, shi);

  // ring-slot addresses, advanced in place every step: row fragments of tile j / j - 1 (ring A; ring B = + LDS_B),
  // transposed fragments of K(j) / K(j - 1)
  int na[4], nb_[4], ta[2][2], tb[2][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) na[ks] = nb_[ks] = LDS_A + nat_off(lane, ks);
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int j = 0; j < 2; ++j) ta[nb][j] = tb[nb][j] = LDS_A + tr_off(lane, nb, j);
  const int bm_lane = LDS_TAB + 4 * hh * 4;
  const int qrel = (q_base & 32) + (lane & 31) - 4 * hh;         // query - first key of the diagonal tile - 4 hh

  f32x16 dqt[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqt[nb][r] = 0.f;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  auto mask_blk = [&](f32x16& s, int kb) {                      // diagonal tile, block kb
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (kb * 32 + (r & 3) + 8 * (r >> 2) > qrel) s[r] = NEG_BIG;
    asm volatile("" ::: "memory");
  };

  lds_dma_wait_all();
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) asm volatile("" :: "v"(qf[ks]), "v"(dof[ks]));   // see attn_fwd_kernel
  asm volatile("" :: "v"(lse2), "v"(ndelta));
  __syncthreads();

  f32x16 sa, sb, dp;
  if (nt_w > 0) {                                               // block 0, unpipelined
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) sa = mfma32(*reinterpret_cast<const bf16x8*>(smem + na[ks]), qf[ks], ks == 0 ? zero16 : sa);
    if (nt_w == 1) mask_blk(sa, 0);
  }

  // One pipelined block: `cur` holds the scores of block KBC = 1 - KBN of the tile at (nc, tc, bma); with NEXT the scores of
  // block KBN of the tile at nn go to `nxt`.  Issue order pinned slot by slot.
  auto body = [&](auto next_c, auto kbn_c, f32x16& cur, f32x16& nxt, const int (&nn)[4], const int (&nc)[4], const int (&tc)[2][2], int bma) {
    constexpr bool NEXT = decltype(next_c)::value;
    constexpr int KBN = decltype(kbn_c)::value, KBC = KBN ^ 1;
    bf16x8 kf[4], vf[4], tf[4];
    u32x4_t cm[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    u32x4_t dsw[2];
    auto load_cm = [&](int G) { if (DROPOUT) cm[G & 1] = *reinterpret_cast<const u32x4_t*>(smem + bma + KBC * 128 + G * 32); };
    auto load_t = [&](int n) {                                   // K^T fragment of dQ MFMA n: cs = n >> 1, nb = n & 1
      const int off = (KBC * 32 + 16 * (n >> 1)) * ROW_B;
      tf[n] = cat4(lds_tr_b64(reinterpret_cast<const bf16*>(smem + tc[n & 1][0] + off)),
                   lds_tr_b64(reinterpret_cast<const bf16*>(smem + tc[n & 1][1] + off)));
    };
    // phase 1
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) vf[ks] = *reinterpret_cast<const bf16x8*>(smem + nc[ks] + LDS_B + KBC * 32 * ROW_B);
    SB();
#pragma unroll
    for (int G = 0; G < 4; ++G) {
      if (NEXT && G == 2) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[ks] = *reinterpret_cast<const bf16x8*>(smem + nn[ks] + KBN * 32 * ROW_B);
      }
      if (G == 3) load_cm(0);
#pragma unroll
      for (int e = 0; e < 4; ++e) cur[4 * G + e] = __builtin_amdgcn_exp2f(fmaf(cur[4 * G + e], p.c, -lse2));
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
        dsw[G >> 1][2 * (G & 1) + hf] = ds2<DROPOUT>(cur[r], cur[r + 1], dp[r], dp[r + 1], ndelta, p.inv_keep, rowh, thr32,
                                                     cm[G & 1][2 * hf], cm[G & 1][2 * hf + 1]);
        SB();
        if (NEXT && n < 4) nxt = mfma32(kf[n], qf[n], n == 0 ? zero16 : nxt);
        if (n == 4 || n == 5) dqt[n - 4] = mfma32(tf[n - 4], __builtin_bit_cast(bf16x8, dsw[0]), dqt[n - 4]);
        SB();
      }
    }
    dqt[0] = mfma32(tf[2], __builtin_bit_cast(bf16x8, dsw[1]), dqt[0]);
    dqt[1] = mfma32(tf[3], __builtin_bit_cast(bf16x8, dsw[1]), dqt[1]);
  };

  // steps: see attn_fwd_kernel
  const bf16* ksrc = kp + tile_el;
  const bf16* vsrc = vp + tile_el;
  int slot_next = TILE64_B;                                      // ring slot of tile j + 1
  int j = 0;
  auto step_open = [&]() {
    if (j + 1 < nt) {
      issue(j + 1, ksrc, vsrc, slot_next);                      // tile j + 1 replaces tile j - 2: last read in step j - 1
      if (DROPOUT && wave == ((j + 1) & 3)) Bm[((j + 1) & 3) * 64 + lane] = drop_col_mult(id_bh + (uint32_t)((j + 1) * 64 + lane), p.seed_lo, shi);
    }
    ksrc += tile_el; vsrc += tile_el;
  };
  auto step_close = [&]() {
    lds_dma_wait_all();
    __syncthreads();
    const int adv = slot_next == 0 ? -2 * TILE64_B : TILE64_B;   // tile j + 1 sits in slot 0 exactly when tile j sits in slot 2
    slot_next = slot_next == 2 * TILE64_B ? 0 : slot_next + TILE64_B;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { nb_[ks] = na[ks]; na[ks] += adv; }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) { tb[nb][jj] = ta[nb][jj]; ta[nb][jj] += adv; }
    ++j;
  };
  const auto T = std::true_type{};
  const auto F = std::false_type{};
  const auto I0 = std::integral_constant<int, 0>{};
  const auto I1 = std::integral_constant<int, 1>{};
  step_open();                                                  // step 0
  if (nt_w > 0) {
    body(T, I1, sa, sb, na, na, ta, bm_lane);
    if (nt_w == 1) mask_blk(sb, 1);
  }
  step_close();
  for (; j + 1 < nt_w;) {                                       // steps 1 .. nt_w - 2: no masks
    const int bma = bm_lane + ((j - 1) & 3) * 256, bmb = bm_lane + (j & 3) * 256;
    step_open();
    body(T, I0, sb, sa, na, nb_, tb, bma);
    body(T, I1, sa, sb, na, na, ta, bmb);
    step_close();
  }
  if (nt_w >= 2) {                                              // step nt_w - 1: its bodies produce the diagonal tile's scores
    const int bma = bm_lane + ((j - 1) & 3) * 256, bmb = bm_lane + (j & 3) * 256;
    step_open();
    body(T, I0, sb, sa, na, nb_, tb, bma);
    mask_blk(sa, 0);
    body(T, I1, sa, sb, na, na, ta, bmb);
    mask_blk(sb, 1);
    step_close();
  }
  if (nt_w > 0) {                                               // step nt_w: the wave's last block
    step_open();
    body(F, I0, sb, sa, na, nb_, tb, bm_lane + ((j - 1) & 3) * 256);
    step_close();
  }
  while (j <= nt) {                                             // the rest of the workgroup is still walking
    step_open();
    step_close();
  }

  // scale, convert; whole 128-byte rows leave through a per-wave LDS staging tile (the rings are idle now)
  unsigned char* ost = smem + wave * (32 * 144);
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16)(dqt[nb][4 * qd + e] * p.scale);
      *reinterpret_cast<bf16x4*>(ost + (lane & 31) * 144 + (nb * 32 + 8 * qd + 4 * hh) * 2) = o;
    }
  __syncthreads();
  bf16* dq = p.dq + (int64_t)b * p.sb + h * DH;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = i * 8 + (lane >> 3), ch = lane & 7;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(ost + row * 144 + ch * 16);
    if (q_base + row < p.S) *reinterpret_cast<bf16x8*>(dq + (int64_t)(q_base + row) * p.ss + ch * 8) = v;
  }
}

// -------------------------------------------------------------------------------------------------------
// backward: dK, dV (S form, one key per lane).  A workgroup owns 128 keys (wave w: keys k0 + 32 w .. + 31) and walks the
// 64-query tiles from its diagonal down to the end of the sequence; Q / dO tiles live in two 3-slot rings, the per-query
// lse, -delta and dropout row hash of a tile in a 4-slot table ring (staged through registers, 64 + 64 + 64 values).
// Per 32-query block:  S = Q.K^T, dP = dO.V^T (8 MFMAs), P and dS as in the dQ kernel, dV^T += dO^T.P, dK^T += Q^T.dS (8 MFMAs);
// the dQ kernel's two-phase block pipeline and step structure.  All four waves walk the same tiles: for waves 2 and 3 (keys
// 64 .. 127 of the block) the workgroup's first tile is fully masked and the second one holds their diagonal, so the scores of
// the first two tiles and of the last (ragged) tile are masked (query >= key, query < S) as they come out of the MFMAs.
// -------------------------------------------------------------------------------------------------------
constexpr int STAT_TILE_B = 3 * 64 * 4;          // lse2[64], -delta[64], row hash[64] of one query tile

template <bool DROPOUT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lds0 = lds_byte_addr(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5;
  const int nkb = (p.S + 127) / 128;
  const int nbh = gridDim.x / nkb;
  const int kblk = (int)(blockIdx.x / nbh);                     // earliest key blocks see the most queries: they come first
  const int bh = blockIdx.x % nbh, h = bh % p.H, b = bh / p.H;
  const int k_base = kblk * 128 + wave * 32;
  const int key = k_base + (lane & 31), keyc = min(key, p.S - 1);
  const bf16* qp = uniform_ptr(p.q + (int64_t)b * p.sb + h * DH);
  const bf16* dop = uniform_ptr(p.d_o + (int64_t)b * p.osb + h * DH);
  const float* lsep = p.lse_in + (int64_t)(b * p.H + h) * p.S;
  const float* delp = p.delta + (int64_t)(b * p.H + h) * p.S;

  const int t0 = 2 * kblk;                                      // first 64-query tile (holds waves 0, 1's diagonal)
  const int nt = (p.S + 63) / 64 - t0;                          // tiles this workgroup streams (relative index j: tile t0 + j)
  const bool valid = k_base < p.S;                              // false: a wave past the sequence only keeps DMA and barriers going
  const bool rag = (p.S & 63) != 0;                             // the last tile holds queries beyond the sequence
  const int64_t qtile_el = (int64_t)64 * p.ss, dtile_el = (int64_t)64 * p.oss;

  uint32_t voffq[2], voffd[2];
  dma_offsets(voffq, lane, wave, p.ss, 64);
  dma_offsets(voffd, lane, wave, p.oss, 64);
  auto issue = [&](int js, const bf16* qsrc, const bf16* dsrc, int slot) {   // Q, dO of relative tile js -> ring slot
    if (rag && js == nt - 1) {
      uint32_t vq[2], vd[2];
      const int valid = p.S - (t0 + js) * 64;
      dma_offsets(vq, lane, wave, p.ss, valid);
      dma_offsets(vd, lane, wave, p.oss, valid);
      dma_tile(vq, qsrc, lds0 + LDS_A + slot, wave);
      dma_tile(vd, dsrc, lds0 + LDS_B + slot, wave);
    } else {
      dma_tile(voffq, qsrc, lds0 + LDS_A + slot, wave);
      dma_tile(voffd, dsrc, lds0 + LDS_B + slot, wave);
    }
  };
  issue(0, qp + t0 * qtile_el, dop + t0 * dtile_el, 0);

  const uint32_t shi = DROPOUT ? seed_mix(p.seed_hi, p.ctr) : 0u;
  const uint32_t id_bh = (uint32_t)((b * p.H + h) * p.S), thr32 = p.thr << 16;
  const uint32_t colm = DROPOUT ? drop_col_mult(id_bh + (uint32_t)key, p.seed_lo, shi) : 0u;
  // per-query statistics of a tile: threads 0..63 stage lse (log2 units), 64..127 stage -delta, 128..191 the dropout row hash
  const float* statp = tid < 64 ? lsep : delp;
  const float statm = tid < 64 ? LOG2E : -1.0f;
  auto load_stat = [&](int js) { return tid < 128 ? statp[min((t0 + js) * 64 + (tid & 63), p.S - 1)] : 0.f; };
  auto store_stat = [&](int js, float v) {
    float* tab = reinterpret_cast<float*>(smem + LDS_TAB + (js & 3) * STAT_TILE_B);
    if (tid < 128) tab[tid] = v * statm;                        // lse2 at [0, 64), -delta at [64, 128)
    else if (DROPOUT && tid < 192)
      reinterpret_cast<uint32_t*>(tab)[tid] = drop_row_hash(id_bh + (uint32_t)((t0 + js) * 64 + tid - 128), p.seed_lo, shi);
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
  // ring-slot addresses, advanced in place every step: row fragments of tile j / j - 1 (ring A = Q; ring B = dO at + LDS_B),
  // transposed fragments likewise
  int na[4], nb_[4], ta[2][2], tb[2][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) na[ks] = nb_[ks] = LDS_A + nat_off(lane, ks);
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int j = 0; j < 2; ++j) ta[nb][j] = tb[nb][j] = LDS_A + tr_off(lane, nb, j);
  const int tab_lane = LDS_TAB + 4 * hh * 4;

  f32x16 dkt[2], dvt[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkt[nb][r] = 0.f; dvt[nb][r] = 0.f; }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // block qs of relative tile js: query = 64 (t0 + js) + 32 qs + acc_row must be >= key and < S
  auto mask_blk = [&](f32x16& s, int js, int qs) {
    const int tq0 = (t0 + js) * 64 + 4 * hh;
    const int krel = key - tq0, qlim = p.S - tq0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cr = qs * 32 + (r & 3) + 8 * (r >> 2);
      if (cr < krel || cr >= qlim) s[r] = NEG_BIG;
    }
    asm volatile("" ::: "memory");
  };

  lds_dma_wait_all();
  store_stat(0, rstat);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) asm volatile("" :: "v"(kf[ks]), "v"(vf[ks]));   // see attn_fwd_kernel
  __syncthreads();

  f32x16 sa, sb, dp;
  if (valid) {                                                  // block 0 of tile 0, unpipelined
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) sa = mfma32(*reinterpret_cast<const bf16x8*>(smem + na[ks]), kf[ks], ks == 0 ? zero16 : sa);
    mask_blk(sa, 0, 0);
  }

  // One pipelined block: `cur` holds the scores of block QSC = 1 - QSN of the tile at (nc, tc, tab); with NEXT the scores of
  // block QSN of the tile at nn go to `nxt`.
  auto body = [&](auto next_c, auto qsn_c, f32x16& cur, f32x16& nxt, const int (&nn)[4], const int (&nc)[4], const int (&tc)[2][2], int tab) {
    constexpr bool NEXT = decltype(next_c)::value;
    constexpr int QSN = decltype(qsn_c)::value, QSC = QSN ^ 1;
    const unsigned char* tb_ = smem + tab + QSC * 128;           // this block's 32 table entries (+ 4 hh)
    bf16x8 qa[4], da[4], tfd[4], tfq[4];
    f32x4 ls[2], nd[2];
    u32x4_t rh[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    u32x4_t pw[2], dsw[2];
    auto load_t = [&](int n) {                                   // dO^T and Q^T fragments of MFMA pair n: cs = n >> 1, nb = n & 1
      const int off = (QSC * 32 + 16 * (n >> 1)) * ROW_B;
      tfq[n] = cat4(lds_tr_b64(reinterpret_cast<const bf16*>(smem + tc[n & 1][0] + off)),
                    lds_tr_b64(reinterpret_cast<const bf16*>(smem + tc[n & 1][1] + off)));
      tfd[n] = cat4(lds_tr_b64(reinterpret_cast<const bf16*>(smem + tc[n & 1][0] + LDS_B + off)),
                    lds_tr_b64(reinterpret_cast<const bf16*>(smem + tc[n & 1][1] + LDS_B + off)));
    };
    auto load_g = [&](int G) {                                   // -delta and row hashes of group G (queries 8 G + 4 hh + 0..3)
      nd[G & 1] = *reinterpret_cast<const f32x4*>(tb_ + 256 + G * 32);
      if (DROPOUT) rh[G & 1] = *reinterpret_cast<const u32x4_t*>(tb_ + 512 + G * 32);
    };
    // phase 1
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) da[ks] = *reinterpret_cast<const bf16x8*>(smem + nc[ks] + LDS_B + QSC * 32 * ROW_B);
    ls[0] = *reinterpret_cast<const f32x4*>(tb_);
    SB();
#pragma unroll
    for (int G = 0; G < 4; ++G) {
      if (G < 3) ls[(G + 1) & 1] = *reinterpret_cast<const f32x4*>(tb_ + (G + 1) * 32);
      if (NEXT && G == 2) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qa[ks] = *reinterpret_cast<const bf16x8*>(smem + nn[ks] + QSN * 32 * ROW_B);
      }
      if (G == 3) load_g(0);
#pragma unroll
      for (int e = 0; e < 4; ++e) cur[4 * G + e] = __builtin_amdgcn_exp2f(fmaf(cur[4 * G + e], p.c, -ls[G & 1][e]));
      SB();
      dp = mfma32(da[G], vf[G], G == 0 ? zero16 : dp);
      SB();
    }
    // phase 2
#pragma unroll
    for (int G = 0; G < 4; ++G) {
      if (G < 3) load_g(G + 1);
      if (G == 1) { load_t(0); load_t(1); }
      if (G == 3) { load_t(2); load_t(3); }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int n = 2 * G + hf, r = 4 * G + 2 * hf;
        float p0 = cur[r], p1 = cur[r + 1], d0 = dp[r], d1 = dp[r + 1], pd0 = p0, pd1 = p1;
        if (DROPOUT) {                                           // product scheme: this lane's column multiplier x the rows' hashes
          const bool k0 = drop_keep(rh[G & 1][2 * hf], colm, thr32), k1 = drop_keep(rh[G & 1][2 * hf + 1], colm, thr32);
          pd0 = k0 ? p0 : 0.f; pd1 = k1 ? p1 : 0.f;             // dV's 1 / keep factor is applied once, when dV is stored
          d0 = fmaf(k0 ? d0 : 0.f, p.inv_keep, nd[G & 1][2 * hf]);
          d1 = fmaf(k1 ? d1 : 0.f, p.inv_keep, nd[G & 1][2 * hf + 1]);
        } else {
          d0 += nd[G & 1][2 * hf];
          d1 += nd[G & 1][2 * hf + 1];
        }
        pw[G >> 1][2 * (G & 1) + hf] = pack_bf16x2(pd0, pd1);
        dsw[G >> 1][2 * (G & 1) + hf] = pack_bf16x2(p0 * d0, p1 * d1);
        SB();
        if (NEXT && n < 4) nxt = mfma32(qa[n], kf[n], n == 0 ? zero16 : nxt);
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

  // steps: step j streams tile j + 1 and handles blocks 2j - 1 (tile j - 1) and 2j (tile j); nt + 1 steps for every wave
  const bf16* qsrc = qp + (t0 + 1) * qtile_el;
  const bf16* dsrc = dop + (t0 + 1) * dtile_el;
  int slot_next = TILE64_B;
  int j = 0;
  auto step_open = [&]() {
    if (j + 1 < nt) {
      issue(j + 1, qsrc, dsrc, slot_next);                      // tile j + 1 replaces tile j - 2: last read in step j - 1
      rstat = load_stat(j + 1);
    }
    qsrc += qtile_el; dsrc += dtile_el;
  };
  auto step_close = [&]() {
    lds_dma_wait_all();
    if (j + 1 < nt) store_stat(j + 1, rstat);
    __syncthreads();
    const int adv = slot_next == 0 ? -2 * TILE64_B : TILE64_B;   // tile j + 1 sits in slot 0 exactly when tile j sits in slot 2
    slot_next = slot_next == 2 * TILE64_B ? 0 : slot_next + TILE64_B;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { nb_[ks] = na[ks]; na[ks] += adv; }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) { tb[nb][jj] = ta[nb][jj]; ta[nb][jj] += adv; }
    ++j;
  };
  const auto T = std::true_type{};
  const auto F = std::false_type{};
  const auto I0 = std::integral_constant<int, 0>{};
  const auto I1 = std::integral_constant<int, 1>{};
  auto tab_of = [&](int js) { return tab_lane + (js & 3) * STAT_TILE_B; };
  step_open();                                                  // step 0: block 0 -> block 1 of tile 0
  if (valid) {
    body(T, I1, sa, sb, na, na, ta, tab_of(0));
    mask_blk(sb, 0, 1);
  }
  step_close();
  if (nt >= 3) {                                                // step 1: the second tile's scores (waves 2, 3: their diagonal)
    step_open();
    if (valid) {
      body(T, I0, sb, sa, na, nb_, tb, tab_of(0));
      mask_blk(sa, 1, 0);
      body(T, I1, sa, sb, na, na, ta, tab_of(1));
      mask_blk(sb, 1, 1);
    }
    step_close();
  }
  while (j + 1 < nt) {                                          // steps 2 .. nt - 2: no masks
    step_open();
    if (valid) {
      body(T, I0, sb, sa, na, nb_, tb, tab_of(j - 1));
      body(T, I1, sa, sb, na, na, ta, tab_of(j));
    }
    step_close();
  }
  if (nt >= 2) {                                                // step nt - 1: the last tile's scores (ragged; diagonal too if nt = 2)
    step_open();
    if (valid) {
      body(T, I0, sb, sa, na, nb_, tb, tab_of(j - 1));
      mask_blk(sa, j, 0);
      body(T, I1, sa, sb, na, na, ta, tab_of(j));
      mask_blk(sb, j, 1);
    }
    step_close();
  }
  step_open();                                                  // step nt: the last block
  if (valid) body(F, I0, sb, sa, na, nb_, tb, tab_of(j - 1));
  step_close();

  // scale, convert; whole 128-byte rows leave through per-wave LDS staging tiles (the rings are idle now)
  unsigned char* ost = smem + wave * (2 * 32 * 144);
  const float vscale = DROPOUT ? p.inv_keep : 1.0f;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      bf16x4 ok, ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ok[e] = (bf16)(dkt[nb][4 * qd + e] * p.scale);
        ov[e] = (bf16)(dvt[nb][4 * qd + e] * vscale);
      }
      const int off = (lane & 31) * 144 + (nb * 32 + 8 * qd + 4 * hh) * 2;
      *reinterpret_cast<bf16x4*>(ost + off) = ok;
      *reinterpret_cast<bf16x4*>(ost + 32 * 144 + off) = ov;
    }
  __syncthreads();
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

constexpr size_t FWD_SMEM = LDS_TAB + 4 * 64 * sizeof(uint32_t);       // Bm[4][64]
constexpr size_t DKDV_SMEM = LDS_TAB + 4 * STAT_TILE_B;

}  // namespace dh64

// 49 - 51 KB of dynamic LDS per workgroup: below the 64 KB a kernel may use without a function attribute
int attn_fwd_dh64(const AttnParams& p, hipStream_t s) {
  const int grid = ((p.S + 127) / 128) * p.H * p.B;
  if (p.thr) dh64::attn_fwd_kernel<true><<<grid, 256, dh64::FWD_SMEM, s>>>(p);
  else dh64::attn_fwd_kernel<false><<<grid, 256, dh64::FWD_SMEM, s>>>(p);
  return check_launch("attn_fwd_dh64");
}

int attn_bwd_dkdv_dh64(const AttnParams& p, hipStream_t s) {
  const int grid = ((p.S + 127) / 128) * p.H * p.B;
  if (p.thr) dh64::attn_bwd_dkdv_kernel<true><<<grid, 256, dh64::DKDV_SMEM, s>>>(p);
  else dh64::attn_bwd_dkdv_kernel<false><<<grid, 256, dh64::DKDV_SMEM, s>>>(p);
  return check_launch("attn_bwd_dkdv_dh64");
}

int attn_bwd_dq_dh64(const AttnParams& p, hipStream_t s) {
  const int grid = ((p.S + 127) / 128) * p.H * p.B;
  if (p.thr) dh64::attn_bwd_dq_kernel<true><<<grid, 256, dh64::FWD_SMEM, s>>>(p);
  else dh64::attn_bwd_dq_kernel<false><<<grid, 256, dh64::FWD_SMEM, s>>>(p);
  return check_launch("attn_bwd_dq_dh64");
}

}  // namespace ttts
