// bf16 MFMA GEMMs of the GPT train step (gfx950, v_mfma_f32_32x32x16_bf16, fp32 accumulation).
//
//  * gemm_nt: C[M,N] = epi(A[M,K] . B[N,K]^T)   -- forward projections and dX (both operands K-contiguous;
//    the bf16 "shadow" weights are kept in both layouts so that no operand ever needs a transposed read).
//  * gemm_tn: C[Mo,No] += At[Kr,Mo]^T . Bt[Kr,No] -- weight gradients; the reduction runs over ROWS of both
//    operands, so fragments are fetched with the LDS transpose read (ds_read_b64_tr_b16); the reduction is split
//    across workgroups into fp32 slabs that a second kernel sums in a fixed order (no atomics).
//
// Tiling: 128x128 block tile, 4 waves (2x2), each wave a 64x64 sub-tile = 2x2 MFMA 32x32 tiles (64 fp32
// accumulator VGPRs), double-buffered LDS operand tiles, ONE barrier per K-tile.  The NT family has three more shapes of
// the same wave tile -- 256x128 on eight waves (with a split grid for dGELU), 160x128 on a 4-slot ring for narrow N -- and
// plan_nt() further down is the measured table of which shape / grid a problem gets (ttts_gemm_nt_plan_query).
// What the first profiles showed and this file answers (tools/kernel_bench.py ablations, MI355X):
//  - the register-staged main loop was LDS-WRITE bound (ds_write_b128 ~ 79 B/clk/CU): the NT main loop now fills LDS
//    with global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip, no ds_write); the DMA image is lane-linear, so the
//    bank-conflict-free layout comes from XOR-swizzling the SOURCE chunk (chunk ^ ((row >> 1) & 7)) and applying the
//    same involution on the ds_read_b128 side;
//  - per-lane epilogue stores touched 32 rows x 16 B per instruction (partial cache lines, half of the kernel time):
//    accumulators are now staged through LDS and leave as whole 256/512-byte rows, 16 bytes per lane.
// The MFMA is issued "swapped" (weights as the A operand) so every lane owns 4 consecutive output columns per
// accumulator quad (float4 staging writes).
#include <algorithm>

#include "common.hpp"

namespace ttts {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NT_LDS_STRIDE = BK + 8;    // register-staged fallback: 144-byte rows
constexpr int TN_LDS_STRIDE = 128 + 32;  // elements (320 B)
constexpr int ST_LD = 132;               // fp32 staging row pitch (floats)
constexpr int EPI_ACCUM_F32 = 16;        // internal: C += acc      (gemm_tn, single split)
constexpr int EPI_SLAB_F32 = 17;         // internal: slab = acc    (gemm_tn, several splits)

struct GemmEpi {
  void* C; int64_t ldc;
  const float* bias;
  bf16* aux;
  const float* resid_in;  // RESID_ADD: C = resid_in + dropout(bf16(acc + bias)); NULL -> in place (C += ...)
  int M, N;
  uint32_t thr; float inv_keep; uint32_t seed_lo, seed_hi;  // residual dropout (RESID_ADD only; thr = 0: off)
  const uint32_t* ctr;  // caller-owned dropout stream counter (device) or NULL
  float* colsum;        // STORE_BF16 / DGELU_BF16: colsum[n] += sum over rows of the bf16 output (the bias gradient of the layer
                        // whose dY this GEMM produces); NULL: off
};

struct GemmNtParams {
  const bf16* A; int64_t lda;
  const bf16* B; int64_t ldb;
  int K;
  GemmEpi e;
  int phase;   // gemm_nt_glds_kernel: start-up delay per co-resident workgroup slot, in units of 1024 cycles (0: none)
  int main_rt, tail_h;   // eight-wave kernel, split grid (tail_h > 0): main_rt 256-row tile rows, then tail_h-row tiles (see plan_nt)
};

// one output row piece: 8 consecutive columns n .. n + 7 of row m, v = acc + bias (fp32), through epilogue EPI
// Epilogue traffic is streamed once (outputs feed the NEXT kernel; the residual and the saved pre-activation are read once): it
// goes around the L2 with non-temporal accesses so that the operand panels, which every column / row tile of the launch
// re-reads, stay resident (tools/ubench/fill_rate.hip: LDS-DMA fills a CU at ~50 B/clk from an L2-resident matrix, ~20 B/clk
// once the matrix falls back to the MALL -- the rate these kernels' k-steps were running at).
typedef __attribute__((ext_vector_type(4))) uint32_t nt_u32x4;
template <typename T>
__device__ __forceinline__ void nt_store16(void* ptr, const T& val) {
  static_assert(sizeof(T) == 16, "16-byte values");
  __builtin_nontemporal_store(__builtin_bit_cast(nt_u32x4, val), reinterpret_cast<nt_u32x4*>(ptr));
}
template <typename T>
__device__ __forceinline__ T nt_load16(const void* ptr) {
  static_assert(sizeof(T) == 16, "16-byte values");
  return __builtin_bit_cast(T, __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4*>(ptr)));
}
#define NT_STORE(T, ptr, val) nt_store16<T>((ptr), (val))
#define NT_LOAD(T, ptr) nt_load16<T>(ptr)
template <int EPI>
__device__ __forceinline__ void epi_row8(const GemmEpi& e, int m, int n, float (&v)[8], bool vec_ok, float (&cs)[8]) {
  const int64_t off = (int64_t)m * e.ldc + n;
  const bool full = vec_ok && (n + 8 <= e.N);
  if (EPI == TTTS_EPI_STORE_BF16) {
    bf16* c = reinterpret_cast<bf16*>(e.C) + off;
    if (full) {
      bf16x8 o;
#pragma unroll
      for (int t = 0; t < 8; ++t) o[t] = (bf16)v[t];
      NT_STORE(bf16x8, c, o);
      if (e.colsum) {
#pragma unroll
        for (int t = 0; t < 8; ++t) cs[t] += (float)o[t];
      }
    } else {
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (n + t < e.N) { c[t] = (bf16)v[t]; cs[t] += (float)(bf16)v[t]; }
    }
  } else if (EPI == TTTS_EPI_GELU_BF16) {
    bf16* c = reinterpret_cast<bf16*>(e.C) + off;
    bf16* ax = e.aux + off;
    bf16x8 pre, act;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      pre[t] = (bf16)v[t];
      act[t] = (bf16)gelu_new_f((float)pre[t]);
    }
    if (full) {
      NT_STORE(bf16x8, ax, pre);   // (plain stores, for either output: GELU +8 ... 10 %, and the pair GELU -> c_proj +3 ... 4 %)
      NT_STORE(bf16x8, c, act);
    } else {
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (n + t < e.N) { ax[t] = pre[t]; c[t] = act[t]; }
    }
  } else if (EPI == TTTS_EPI_RESID_ADD_F32) {
    float* c = reinterpret_cast<float*>(e.C) + off;
    const float* rin = e.resid_in ? e.resid_in + off : c;
    float y[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) y[t] = (float)(bf16)v[t];
    if (e.thr) {  // resid_pdrop: element index m*N + n, 16 random bits per element (two elements per hash)
      const uint32_t lin = (uint32_t)(((int64_t)m * e.N + n) >> 1);
      const uint32_t shi = seed_mix(e.seed_hi, e.ctr);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const uint32_t r = hash32(lin + t, e.seed_lo, shi);
        y[2 * t] = (r & 0xFFFFu) >= e.thr ? y[2 * t] * e.inv_keep : 0.f;
        y[2 * t + 1] = (r >> 16) >= e.thr ? y[2 * t + 1] * e.inv_keep : 0.f;
      }
    }
    if (full) {
      const float4 r0 = NT_LOAD(float4, rin), r1 = NT_LOAD(float4, rin + 4);
      NT_STORE(float4, c, make_float4(r0.x + y[0], r0.y + y[1], r0.z + y[2], r0.w + y[3]));
      NT_STORE(float4, c + 4, make_float4(r1.x + y[4], r1.y + y[5], r1.z + y[6], r1.w + y[7]));
    } else {
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (n + t < e.N) c[t] = rin[t] + y[t];
    }
  } else if (EPI == TTTS_EPI_DGELU_BF16) {
    bf16* c = reinterpret_cast<bf16*>(e.C) + off;
    const bf16* ax = e.aux + off;
    if (full) {
      const bf16x8 pre = NT_LOAD(bf16x8, ax);
      bf16x8 o;
#pragma unroll
      for (int t = 0; t < 8; ++t) o[t] = (bf16)(v[t] * gelu_new_grad_f((float)pre[t]));
      NT_STORE(bf16x8, c, o);
      if (e.colsum) {
#pragma unroll
        for (int t = 0; t < 8; ++t) cs[t] += (float)o[t];
      }
    } else {
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (n + t < e.N) { c[t] = (bf16)(v[t] * gelu_new_grad_f((float)ax[t])); cs[t] += (float)c[t]; }
    }
  } else {  // STORE_F32 / ACCUM_F32 / SLAB_F32
    float* c = reinterpret_cast<float*>(e.C) + off;
    if (full) {
      float4 o0 = make_float4(v[0], v[1], v[2], v[3]), o1 = make_float4(v[4], v[5], v[6], v[7]);
      if (EPI == EPI_ACCUM_F32) {
        const float4 c0 = *reinterpret_cast<const float4*>(c), c1 = *reinterpret_cast<const float4*>(c + 4);
        o0 = make_float4(c0.x + o0.x, c0.y + o0.y, c0.z + o0.z, c0.w + o0.w);
        o1 = make_float4(c1.x + o1.x, c1.y + o1.y, c1.z + o1.z, c1.w + o1.w);
      }
      *reinterpret_cast<float4*>(c) = o0;
      *reinterpret_cast<float4*>(c + 4) = o1;
    } else {
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (n + t < e.N) c[t] = (EPI == EPI_ACCUM_F32 ? c[t] : 0.f) + v[t];
    }
  }
}

// Column sums of a tile's bf16 output (GemmEpi::colsum): every thread has summed its rows of ONE 8-column group (threads
// tid % CG share a group); lanes of a wave that share it, then the waves through the idle stage, then one atomic per column.
template <int CG, int NWAVES>
__device__ __forceinline__ void colsum_flush(const GemmEpi& e, float (&cs)[8], float* stage, int n0, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int o = CG; o < 64; o <<= 1)
#pragma unroll
    for (int t = 0; t < 8; ++t) cs[t] += __shfl_xor(cs[t], o, 64);
  if (lane < CG) {
#pragma unroll
    for (int t = 0; t < 8; ++t) stage[wave * (CG * 8) + lane * 8 + t] = cs[t];
  }
  __syncthreads();
  if (tid < CG * 8) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) s += stage[w * (CG * 8) + tid];
    if (n0 + tid < e.N) atomicAdd(e.colsum + n0 + tid, s);
  }
}

// ---- shared epilogue: accumulators -> fp32 LDS stage (64 rows at a time) -> coalesced global access ---------------
// stage: >= 64 * ST_LD floats of LDS that no wave reads any more (callers end their main loop with a barrier).
// NJ = 32-column blocks per wave: 2 for the 128 x 128 tile, 1 for the 128 x 64 tile (N = 512 GEMMs: 292 -> 584 tiles)
template <int EPI, int NJ = 2, int NWM = 2>
__device__ __forceinline__ void tile_epilogue(const GemmEpi& e, f32x16 (&acc)[2][2], float* stage, int m0, int n0,
                                              int tid, int nhalf = NWM) {   // nhalf: 64-row groups of the tile that hold rows (uniform)
  constexpr int TPR = 8 * NJ;              // threads per 64*NJ-column row (8 columns each)
  constexpr int RPP = 128 * NWM / TPR;     // rows per read-back pass (NWM = 64-row wave groups of the workgroup: 2 or 4)
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, h = lane >> 5;
  const bool bf16_out = (EPI == TTTS_EPI_STORE_BF16 || EPI == TTTS_EPI_GELU_BF16 || EPI == TTTS_EPI_DGELU_BF16);
  const bool vec_ok = bf16_out ? ((e.ldc & 7) == 0) : ((e.ldc & 3) == 0);
  // this thread stores columns n0 + (tid % TPR)*8 .. +8 of every row it touches: fetch their bias once
  float bias8[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int n = n0 + (tid % TPR) * 8 + t;
    const float b = (e.bias && n < e.N) ? e.bias[n] : 0.f;
    bias8[t] = (EPI == TTTS_EPI_STORE_F32) ? b : (float)(bf16)b;  // autocast rounds the bias to bf16
  }
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int half = 0; half < NWM; ++half) {
    if (half >= nhalf) break;
    if (wm == half) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = wn * (32 * NJ) + j * 32 + 8 * q + 4 * h;
            const float4 v = make_float4(acc[j][i][4 * q], acc[j][i][4 * q + 1], acc[j][i][4 * q + 2], acc[j][i][4 * q + 3]);
            *reinterpret_cast<float4*>(&stage[(i * 32 + (lane & 31)) * ST_LD + col]) = v;
          }
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 64 / RPP; ++pass) {
      const int row_l = pass * RPP + tid / TPR;
      const int m = m0 + half * 64 + row_l;
      const int n = n0 + (tid % TPR) * 8;
      if (m >= e.M || n >= e.N) continue;
      float v[8];
      {
        const float4 a = *reinterpret_cast<const float4*>(&stage[row_l * ST_LD + (tid % TPR) * 8]);
        const float4 b = *reinterpret_cast<const float4*>(&stage[row_l * ST_LD + (tid % TPR) * 8 + 4]);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] += bias8[t];
      epi_row8<EPI>(e, m, n, v, vec_ok, cs);
    }
    __syncthreads();
  }
  if ((EPI == TTTS_EPI_STORE_BF16 || EPI == TTTS_EPI_DGELU_BF16) && e.colsum) colsum_flush<TPR, 2 * NWM>(e, cs, stage, n0, tid);
}

// XCD-aware tile order: the dispatcher places workgroup b on XCD b % 8 (observed, speed only).  Remap so that each XCD
// walks a CONTIGUOUS range of tile ids: tiles that share an operand panel then hit the same (private) L2.  Bijective
// for any grid size.
__device__ __forceinline__ int xcd_tile(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

#define ZERO_ACC(acc)                                 \
  _Pragma("unroll") for (int j = 0; j < 2; ++j)       \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)       \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

// ---- NT, LDS-DMA main loop (K % BKT == 0) --------------------------------------------------------------------------
// LDS image per operand and buffer: [128 rows][BKT k] bf16, NO padding (the DMA writes lane-linearly: one wave
// instruction = 1 KB = 64*8/BKT... rows; lane l -> row += l / CPR, 16-byte slot l % CPR, CPR = BKT/8 chunks per row).
// Slot s of row r holds the logical k-chunk s ^ f(r), f(r) = (r >> 1) & 7 for 128-byte rows (BKT = 64) and
// (r >> 2) & 3 for 64-byte rows (BKT = 32): ds_read_b128 of a 16-lane service group then hits 64 distinct banks.
// BKT = 64: 64 KB LDS, 2 workgroups per CU.  BKT = 32: 33.8 KB, 3 workgroups per CU -- their store phases and main
// loops interleave instead of running in lock-step.  NWM = 4: eight waves on a 256 x 128 tile (see plan_nt).
// Round-2 measurements on this kernel (c_attn shape 9248 x 1536 x 512, 28.3 us = 514 TF/s; all variants parity-tested):
//  * PMC: 25 % MFMA-busy, 44 % of wave cycles parked in s_waitcnt / barrier, 30 % issue stalls, 5.4 VALU per MFMA;
//  * an LDS-DMA RING (3 or 4 stages, loads 2-3 k-steps ahead, counted vmcnt across raw s_barriers) does NOT help: 64-deep
//    stages x 3 (96 KB, one workgroup per CU) 37.9 us; 32-deep x 3 (3 per CU) 30.3 us; 32-deep x 4 31.1 us -- deeper prefetch
//    is not the limit, co-resident workgroups already cover the load latency;
//  * ablation of the 2-stage loop: fill only (no ds_read / MFMA) + epilogue 22.7 us, MFMA only (no DMA) + epilogue 21.6 us,
//    both 28.9 us, epilogue ~8 us: LDS fill (230 MB at ~15 TB/s) and compute (1.07 PF/s with its LDS reads) each need
//    ~14 us and overlap only half.  Two follow-ups were built, measured and removed: a persistent 8-wave kernel with store
//    waves (1.6x slower: its tile hand-off serialised on LDS slots) and a K-split of the surplus tiles of the 292-tile
//    launches (round 3: GPT step 4.01 vs 3.61 ms -- the fix-up launch and the fp32 slabs cost more than the tail they fill).
template <int EPI, int BKT, int NJ = 2, int NWM = 2, bool SPLIT = false>
__global__ __launch_bounds__(128 * NWM, (NWM == 4 ? 4 : BKT == 64 ? 2 : 3)) void gemm_nt_glds_kernel(GemmNtParams p) {
  static_assert(!SPLIT || NWM == 4, "the split grid belongs to the eight-wave tile");
  constexpr int BNT = 64 * NJ;           // tile columns: 128, or 64 for the narrow-N GEMMs
  constexpr int BMT = 64 * NWM;          // tile rows: 128 (four waves) or 256 (eight waves sharing the B stage)
  constexpr int NWAVES = 2 * NWM;
  constexpr int CPR = BKT / 8;           // 16-byte chunks per row
  constexpr int RPI = 64 / CPR;          // rows per wave instruction
  constexpr int IPW = BMT / RPI / NWAVES;   // instructions per wave for the A tile
  constexpr int IPWB = BNT / RPI / NWAVES;  // ... and for the B tile
  static_assert(IPWB >= 1, "tile too narrow for this many waves");
  constexpr int TA = BMT * BKT, TB = BM * BKT;   // elements per A / B buffer slot (the B slot is only partly used when BNT = 64)
  constexpr int SMEM = (2 * (TA + TB) * 2 > 64 * ST_LD * 4) ? 2 * (TA + TB) : 64 * ST_LD * 2;  // elements
  __shared__ __attribute__((aligned(16))) bf16 smem[SMEM];  // [buf][A | B]; reused as the epilogue stage
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (p.e.N + BNT - 1) / BNT;
  int tile, m0, rows = BMT;
  if (SPLIT) {
    // split grid: the first main_rt * tiles_n workgroups (one full round of the chip's slots) take 256-row tiles, the rest the
    // remaining rows in tail_h-row tiles, which follow them onto the slots as these free up (XCD-contiguous within each part)
    const int main_tiles = p.main_rt * tiles_n;
    if ((int)blockIdx.x < main_tiles) {
      tile = xcd_tile(blockIdx.x, main_tiles);
      m0 = (tile / tiles_n) * BMT;
    } else {
      tile = xcd_tile(blockIdx.x - main_tiles, gridDim.x - main_tiles);
      m0 = p.main_rt * BMT + (tile / tiles_n) * p.tail_h;
      rows = p.tail_h;
    }
  } else {
    tile = xcd_tile(blockIdx.x, gridDim.x);
    m0 = (tile / tiles_n) * BMT;
  }
  const int n0 = (tile % tiles_n) * BNT;
  const int m_end = min(p.e.M, m0 + rows);
  const int nact = SPLIT ? (m_end - m0 + 63) >> 6 : NWM;   // 64-row wave groups with rows to compute (workgroup-uniform)
  const int nk = p.K / BKT;
  // The workgroups that share a CU start together and have equal lives, so their main loops (matrix cores + LDS fill) and
  // their epilogues (VALU + HBM stores) fall on top of each other.  Where the launcher asks for it (p.phase: launches of many
  // rounds, and the VALU-heavy dGELU epilogue) the first-round workgroup of wave slot s starts s * phase * 1024 cycles late, and
  // the offset then carries through the rounds: the mel-head GEMM (8 rounds) -4 %, dGELU -4 % on top of their tile shapes
  // (tools/ubench/nt_phase.cpp; the slot is read from HW_ID.wave_id -- if the hardware hands slots out differently the delay
  // is merely useless).  Output bits do not depend on it.
  constexpr int WGS_PER_CU = NWM == 4 ? 2 : BKT == 64 ? 2 : 3;
  if (p.phase > 0 && blockIdx.x < 256 * WGS_PER_CU) {
    if (wave == 0) {
      const int slot = (int)((__builtin_amdgcn_s_getreg(6148) & 15u) / (NWM / 2)) % WGS_PER_CU;   // hwreg(HW_REG_HW_ID, 0, 4)
      for (int i = 0; i < slot * p.phase; ++i) __builtin_amdgcn_s_sleep(16);                       // 16 x 64 cycles
    }
    __syncthreads();
  }
  auto fsw = [](int r) { return BKT == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3); };

  // this lane's DMA source rows (clamped: rows beyond M / N are loaded from the last valid row and never stored)
  const bf16* ga[IPW];
  const bf16* gb[IPWB];
#pragma unroll
  for (int i = 0; i < IPW; ++i) {
    const int r = wave * (IPW * RPI) + i * RPI + lane / CPR;
    const int chunk = (lane % CPR) ^ fsw(r);
    ga[i] = p.A + (int64_t)min(m0 + r, p.e.M - 1) * p.lda + chunk * 8;
  }
#pragma unroll
  for (int i = 0; i < IPWB; ++i) {
    const int r = wave * (IPWB * RPI) + i * RPI + lane / CPR;
    const int chunk = (lane % CPR) ^ fsw(r);
    gb[i] = p.B + (int64_t)min(n0 + r, p.e.N - 1) * p.ldb + chunk * 8;
  }
  const bool a_on = !SPLIT || wave * (IPW * RPI) < nact * 64;   // this wave's A rows belong to a wave group with work
  auto issue = [&](int kt, int buf) {
    bf16* as = smem + buf * (TA + TB) + wave * (IPW * RPI) * BKT;
    bf16* bs = smem + buf * (TA + TB) + TA + wave * (IPWB * RPI) * BKT;
    if (a_on) {
#pragma unroll
      for (int i = 0; i < IPW; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga[i] + kt * BKT),
                                         (__attribute__((address_space(3))) void*)(as + i * RPI * BKT), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < IPWB; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb[i] + kt * BKT),
                                       (__attribute__((address_space(3))) void*)(bs + i * RPI * BKT), 16, 0, 0);
  };

  f32x16 acc[2][2];  // [j: 32-col block of N][i: 32-row block of M]; D = Btile . Atile^T (rows = n, cols = m)
  ZERO_ACC(acc)
  issue(0, 0);
  __syncthreads();  // (hipcc drains the LDS-DMA -- vmcnt(0) -- in front of the barrier)
  const int hh = lane >> 5;
  int aoff[2], boff[2], sw[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = wm * 64 + i * 32 + (lane & 31), rb = wn * (32 * NJ) + (i % NJ) * 32 + (lane & 31);
    aoff[i] = ra * BKT;
    boff[i] = rb * BKT;
    sw[0][i] = fsw(ra);
    sw[1][i] = fsw(rb);
  }
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
    const bf16* as = smem + buf * (TA + TB);
    const bf16* bs = as + TA;
    if (!SPLIT || wm < nact) {
#pragma unroll
      for (int ks = 0; ks < BKT / 16; ++ks) {
        bf16x8 af[2], bfr[2];
        const int lc = ks * 2 + hh;  // logical 16-byte chunk of this lane's 8 k-values
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          af[i] = *reinterpret_cast<const bf16x8*>(as + aoff[i] + ((lc ^ sw[0][i]) << 3));
          if (i < NJ) bfr[i] = *reinterpret_cast<const bf16x8*>(bs + boff[i] + ((lc ^ sw[1][i]) << 3));
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[j][i] = mfma32(bfr[j], af[i], acc[j][i]);
      }
    }
    __syncthreads();  // next tile landed (vmcnt(0)) and everyone is done reading this one
  }
  if (SPLIT) {
    GemmEpi el = p.e;
    el.M = m_end;      // rows past this tile's own belong to another workgroup
    tile_epilogue<EPI, NJ, NWM>(el, acc, reinterpret_cast<float*>(smem), m0, n0, tid, nact);
  } else {
    tile_epilogue<EPI, NJ, NWM>(p.e, acc, reinterpret_cast<float*>(smem), m0, n0, tid);
  }
}

// ---- NT, tall tile + deep ring for the narrow-N GEMMs (N <= 512: attn / mlp c_proj, dX of c_attn / c_fc) ------------------
// At M = 9248 a 128 x 128 tiling of an N = 512 GEMM has 292 tiles for 256 CUs: about one workgroup per CU, i.e. nothing
// to overlap a workgroup's own waits with.  Measured (round 3): 1.24 us per k-step of 64 -- 2 800 cycles around 512 cycles of
// MFMA work -- whatever the tile count per CU (a 232-tile single-round tiling alone gained 4 %): every k-step waits one full
// L2 / HBM round trip for the tile it issued one step earlier.  So this kernel keeps THREE k-tiles in flight behind the one
// being consumed: a 4-slot LDS ring (4 x 36 KB: one workgroup per CU, which is all these launches have anyway), LDS-DMA issued
// through the untracked asm form (hipcc would drain it with vmcnt(0) at every barrier) and counted waits -- vmcnt(18) leaves
// the two newest k-tiles (9 pieces each per wave) in flight.  One barrier per k-step: it publishes tile kt and, because every
// wave has finished tile kt - 1 when it arrives, frees slot (kt + 3) % 4 for the next issue.
// Tile: (32 MI) x (32 NW), NW waves: wave w owns output columns 32 w .. 32 w + 31 of all MI row blocks (MI accumulators; per
// k-step of 16 it reads MI A fragments and one B fragment).  Instance in use: MI 5, 4 waves, 160 x 128, 4-slot ring -- 58 x 4 =
// 232 tiles, one round (mlp c_proj 41.2 -> 36.3 us, dX c_fc 35.9 -> 30.4, attn c_proj 19.1 -> 17.5; GPT step -0.10 ms).
// Measured and not kept: an L2 prefetch of the A panel 2 ... 12 k-tiles beyond the ring (one LDS-DMA dword per 128-byte line
// into a dummy: +3 ... 8 % -- these launches are not waiting for HBM latency; tools/ubench/nt_phase.cpp);
// MI 8, 8 waves, 256 x 256, 2-slot ring for the N >= 1536 GEMMs (half the operand bytes per flop):
// c_attn 30.4 vs 28.8 us, c_fc 58.0 vs 43.0 us (296 tiles: two rounds) -- with one workgroup per CU the main loops and the
// output stores of all CUs run in phases instead of interleaving.
template <int EPI, int MI, int NW, int NST>
__global__ __launch_bounds__(64 * NW, 1) void gemm_nt_tall_kernel(GemmNtParams p) {
  constexpr int BKT = 64, TM = 32 * MI, TN = 32 * NW, NT = 64 * NW;   // NW waves, one 32-column strip each
  constexpr int ACH = TM / 8, BCH = TN / 8;        // 1-KB DMA chunks (8 rows) per operand tile
  constexpr int A_EL = TM * BKT, B_EL = TN * BKT, STAGE_EL = A_EL + B_EL;
  constexpr int PPW = (ACH + BCH) / NW;            // DMA pieces per wave and k-tile
  static_assert(ACH % NW == 0 && BCH % NW == 0, "chunks are dealt evenly to the waves");
  constexpr int ST_W = TN + 4;                     // fp32 staging row pitch (floats)
  extern __shared__ __attribute__((aligned(16))) unsigned char tall_smem[];
  bf16* smem = reinterpret_cast<bf16*>(tall_smem);  // [NST][A | B]; reused as the fp32 epilogue stage
  const uint32_t lds0 = lds_byte_addr(tall_smem);
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (p.e.N + TN - 1) / TN;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int m0 = (tile / tiles_n) * TM, n0 = (tile % tiles_n) * TN;
  const int nk = p.K / BKT;
  auto fsw = [](int r) { return (r >> 1) & 7; };

  // this lane's DMA source rows (clamped: rows beyond M / N are loaded from the last valid row and never stored)
  const bf16* ga[ACH / NW];
  const bf16* gb[BCH / NW];
#pragma unroll
  for (int i = 0; i < ACH / NW; ++i) {
    const int r = (wave * (ACH / NW) + i) * 8 + (lane >> 3);
    ga[i] = p.A + (int64_t)min(m0 + r, p.e.M - 1) * p.lda + (((lane & 7) ^ fsw(r)) << 3);
  }
#pragma unroll
  for (int i = 0; i < BCH / NW; ++i) {
    const int r = (wave * (BCH / NW) + i) * 8 + (lane >> 3);
    gb[i] = p.B + (int64_t)min(n0 + r, p.e.N - 1) * p.ldb + (((lane & 7) ^ fsw(r)) << 3);
  }
  auto issue = [&](int kt) {                       // k-tile kt -> ring slot kt % NST (PPW untracked DMA pieces)
    const uint32_t as = lds0 + ((kt & (NST - 1)) * STAGE_EL + wave * (ACH / NW) * 8 * BKT) * 2;
    const uint32_t bs = lds0 + ((kt & (NST - 1)) * STAGE_EL + A_EL + wave * (BCH / NW) * 8 * BKT) * 2;
#pragma unroll
    for (int i = 0; i < ACH / NW; ++i) lds_dma16_untracked(ga[i] + kt * BKT, as + i * 1024);
#pragma unroll
    for (int i = 0; i < BCH / NW; ++i) lds_dma16_untracked(gb[i] + kt * BKT, bs + i * 1024);
  };

  f32x16 acc[MI];                                   // [i: 32-row block of M]; D = Btile . Atile^T (rows = n, cols = m)
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
  for (int t = 0; t < NST - 1; ++t)
    if (t < nk) issue(t);
  const int rl = lane & 31, swl = fsw(rl);          // fsw(32 i + rl) = fsw(rl)
  const int boff = (wave * 32 + rl) * BKT;
  // tile `kt` has landed when at most the pieces of the (up to two) newer tiles are outstanding
  auto wait_tile = [&](int kt) {
    const int newer = min(NST - 2, nk - 1 - kt);
    // (lgkmcnt(0): this wave's fragment reads of the slot that is about to be refilled have really left the LDS)
    if (newer >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(2 * PPW) : "memory");
    else if (newer == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  // fragments of k-step ks (16 k-values) of tile kt: one B fragment, MI A fragments
  bf16x8 fb[2], fa[2][MI];
  auto load_frags = [&](int set, int kt, int ks) {
    const bf16* as = smem + (kt & (NST - 1)) * STAGE_EL;
    const int ch = ((ks * 2 + hh) ^ swl) << 3;       // this lane's 8 k-values: physical 16-byte chunk
    fb[set] = *reinterpret_cast<const bf16x8*>(as + A_EL + boff + ch);
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[set][i] = *reinterpret_cast<const bf16x8*>(as + (i * 32 + rl) * BKT + ch);
  };
  // Software pipeline, order pinned: the fragments of the NEXT k-step are requested in front of the current k-step's MI MFMAs
  // (with one workgroup per CU nothing else hides the LDS latency: the compiler's own order waited 20 times per tile).  The
  // barrier that publishes tile kt + 1 sits in front of tile kt's last k-step: by then every wave holds that k-step's
  // fragments in registers, so slot kt % 4 is free for tile kt + 4.
  wait_tile(0);
  if (NST - 1 < nk) issue(NST - 1);
  load_frags(0, 0, 0);
  for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
    for (int ks = 0; ks < BKT / 16; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks + 1 < BKT / 16) {
        load_frags(nxt, kt, ks + 1);
      } else if (kt + 1 < nk) {
        wait_tile(kt + 1);
        if (kt + NST < nk) issue(kt + NST);
        load_frags(nxt, kt + 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[i] = mfma32(fb[cur], fa[cur][i], acc[i]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();                                  // every wave is done reading the ring: it becomes the epilogue stage
  // epilogue: two row blocks (64 rows) at a time through the fp32 stage, whole rows out (TN / 8 threads x 8 columns per row)
  float* stage = reinterpret_cast<float*>(tall_smem);
  const bool bf16_out = (EPI == TTTS_EPI_STORE_BF16 || EPI == TTTS_EPI_GELU_BF16 || EPI == TTTS_EPI_DGELU_BF16);
  const bool vec_ok = bf16_out ? ((p.e.ldc & 7) == 0) : ((p.e.ldc & 3) == 0);
  float bias8[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int n = n0 + (tid % (TN / 8)) * 8 + t;
    const float b = (p.e.bias && n < p.e.N) ? p.e.bias[n] : 0.f;
    bias8[t] = (EPI == TTTS_EPI_STORE_F32) ? b : (float)(bf16)b;
  }
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int pass = 0; pass < (MI + 1) / 2; ++pass) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      const int i = 2 * pass + ii;
      if (i < MI) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = make_float4(acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]);
          *reinterpret_cast<float4*>(&stage[(ii * 32 + rl) * ST_W + wave * 32 + 8 * q + 4 * hh]) = v;
        }
      }
    }
    __syncthreads();
    const int rows = (2 * pass + 1 < MI) ? 64 : 32;
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      const int row_l = sub * 16 + tid / (TN / 8);
      const int m = m0 + pass * 64 + row_l, n = n0 + (tid % (TN / 8)) * 8;
      if (row_l >= rows || m >= p.e.M || n >= p.e.N) continue;
      float v[8];
      const float4 a = *reinterpret_cast<const float4*>(&stage[row_l * ST_W + (tid % (TN / 8)) * 8]);
      const float4 b = *reinterpret_cast<const float4*>(&stage[row_l * ST_W + (tid % (TN / 8)) * 8 + 4]);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] += bias8[t];
      epi_row8<EPI>(p.e, m, n, v, vec_ok, cs);
    }
    __syncthreads();
  }
  if ((EPI == TTTS_EPI_STORE_BF16 || EPI == TTTS_EPI_DGELU_BF16) && p.e.colsum) colsum_flush<TN / 8, NW>(p.e, cs, stage, n0, tid);
}

// ---- NT, weights in REGISTERS, persistent over row tiles (K = 512, wide N: c_attn, c_fc + GELU) -------------------------------
// What rounds 2 and 3 measured on the tiled kernels above for the K = 512, N >= 1536 GEMMs (c_attn 25 us, c_fc + GELU 42 us, dGELU
// 51 us against 6 / 8 / 8 us of matrix-core time): a k-step waits for its 32 KB of L2 -> LDS fill (the W panel is re-fetched
// for every row tile: 300 MB of fill per launch), and the epilogue of a tile runs with the matrix cores idle.  At K = 512 a
// wave's share of a weight panel fits its REGISTER FILE: 32 output columns x 512 k = 32 KB per wave = 128 VGPRs per lane, as the
// 32 MFMA A-operand fragments of the whole reduction.  So:
//   * one workgroup of eight waves per CU owns a 256-column panel of W (wave w: columns 32 w .. 32 w + 31), loaded once --
//     through the LDS in whole 1-KB rows (fragment-shaped global loads of it, 32 rows x 32 B per instruction, were L2-request
//     bound: 8 us of a 29-us launch);
//   * it walks its share of the rows in 64-row tiles; the ONLY operand that moves per tile is the activation tile (64 x 512 bf16 =
//     64 KB by LDS-DMA, two buffers), which every wave reads whole: per 16-deep k-step two ds_read_b128 and two MFMAs, no B
//     fragment reads, no k-loop barriers -- one tile = 64 MFMAs per wave straight through;
//   * PING-PONG phases: waves 0-3 (columns 0-127) and waves 4-7 (columns 128-255) sit pairwise on the four SIMDs and alternate
//     roles every phase, one barrier per phase: while one group runs the MFMAs of tile t, the other runs the epilogue of the
//     tile it has just finished (VALU, LDS, stores) -- the matrix pipe of a SIMD always has exactly one wave feeding it and the
//     epilogue never runs with the matrix cores idle.  (Two free-running workgroups per CU did not arrange themselves this way:
//     measured 25 / 30 / 40 us for store / GELU / dGELU at N = 2048, against 19.5 us without any epilogue.)
//   * every wave stages its own 64 x 32 accumulator block through a PRIVATE 4-KB LDS stage (no barrier inside a phase) and stores
//     16 bytes per lane, through the same epi_row8 epilogues as every other kernel of the family: one bf16 round where the stored
//     value is bf16(acc + bias) (store, GELU), two fp32 rounds of 32 rows otherwise.
// Memory-counter protocol (LDS-DMA is issued untracked; vmcnt retires in order, stores included): every wave issues its eight
// pieces of tile t + 2 at the start of the even phase in which the buffer is free.  Group 0 is computing then: it waits for them
// at the start of its next (epilogue) phase, before its first store -- everything it has in flight is a phase old.  Group 1 is
// in its epilogue then, so its pieces sit in front of that phase's stores: it waits at the end of its next (MFMA) phase.  The
// barrier that ends odd phase 2 t + 3 publishes tile t + 2 to both groups.
// Measured (round 4, MI355X; tools/ubench/nt_phase.cpp stand-alone, then rocprofv3 inside the train step, same box):
//   c_attn 9248 x 1536:  23.6 -> 20.9 us stand-alone, 24.4 -> 22.8 us in the step;  c_fc + GELU 9248 x 2048: 39.3 -> 30.9 / 42.0 -> 33.4.
//   Cycle stamps (tools/ubench/wreg_trace.cpp): prologue 12 k cycles (four weight passes, then the first tile: two DMA latencies),
//   an MFMA phase 2.7 k cycles (64 MFMAs = 2.05 k) and 3.2 k with the eight DMA issues in it (~60-160 cycles each: a tile is
//   64 wave-instructions whoever issues them), a store epilogue 1.8-2.0 k, +0.3 k per phase of barrier and memory waits.
//   Stores stay non-temporal like the tiled kernels' (plain write-back stores: c_fc + GELU 34.0 vs 30.9 us stand-alone).
//   dGELU is NOT taken: stand-alone 45.8 -> 39.2 us (pre-activation block by DMA into the private stage, replaced in place), but
//   in the step, where the saved pre-activation comes from HBM, 45.6 us = the split-grid tiled kernel's time -- one workgroup
//   per CU has nothing to cover that latency with, and no LDS is left to fetch it a phase earlier; an L2 prefetch queued in
//   front of the pieces made it 47.3 us (in-order vmcnt).  The path was removed again.
// Work split: ceil(N / 256) panels x (CUs / panels) row groups, rows dealt to the groups in 32-row units; the XCD-contiguous order
// keeps the panels of a row group on one L2 (the activation tile is fetched from HBM once per group).  Output bits equal the tiled
// kernels' (each element is the same k-ordered MFMA chain).
constexpr int WR_K = 512, WR_KS = WR_K / 16, WR_TM = 64, WR_PN = 256;
constexpr int WR_ABUF = WR_TM * WR_K * 2, WR_STAGE = 4096, WR_LDS = 2 * WR_ABUF + 8 * WR_STAGE;   // 2 x 64 KB tiles + 8 x 4 KB stages
#ifndef WR_TRACE
#define WR_TRACE 0
#endif
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_wreg_kernel(GemmNtParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wreg_smem[];   // [2][64 rows][64 x 16 B], slot = chunk ^ (row & 15) | stages
  const uint32_t lds0 = lds_byte_addr(wreg_smem);
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, rl = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp2 = wave >> 2;                          // ping-pong group: 0 computes in even phases, 1 in odd phases
  const int M = p.e.M, N = p.e.N;
  const int np = (N + WR_PN - 1) / WR_PN, n_groups = p.main_rt;
  const int id = xcd_tile(blockIdx.x, gridDim.x);
  const int grp = id / np, n0 = (id % np) * WR_PN, nw = n0 + wave * 32;   // nw: this wave's first column
  const int units = (M + 31) >> 5;
  const int r0 = (int)((uint32_t)grp * (uint32_t)units / (uint32_t)n_groups) * 32;            // (grp < CUs, units < 2^22)
  const int r1 = min(M, (int)((uint32_t)(grp + 1) * (uint32_t)units / (uint32_t)n_groups) * 32);
  if (r0 >= r1) return;
  const int ntile = (r1 - r0 + WR_TM - 1) / WR_TM;
#if WR_TRACE
  // debug build (tools/ubench/wreg_trace.cpp): cycle stamps of waves 0 and 4 of two workgroups into the colsum buffer
  unsigned long long* trc = reinterpret_cast<unsigned long long*>(p.e.colsum);
  int trc_n = 0;
  const bool trc_on = trc && lane == 0 && (wave & 3) == 0 && (blockIdx.x == 3 || blockIdx.x == 200);
  unsigned long long* trc_w = trc + ((blockIdx.x == 3 ? 0 : 2) + (wave >> 2)) * 128;
#define WR_STAMP() do { if (trc_on && trc_n < 128) trc_w[trc_n++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WR_STAMP() do {} while (0)
#endif
  WR_STAMP();

  // 64 rows x 1 KB of a row-major matrix -> tile buffer `buf`: eight 1-KB rows per wave; lane l fills slot l of its row with the
  // row's logical 16-byte chunk l ^ (row & 15) (rows beyond row_max: clamped, never stored)
  auto dma64 = [&](const bf16* base, int64_t ld, int row0, int row_max, int buf) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int R = wave * 8 + j;
      lds_dma16_untracked(base + (int64_t)min(row0 + R, row_max) * ld + ((lane ^ (R & 15)) << 3), lds0 + buf * WR_ABUF + R * 1024);
    }
  };
  auto bar = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  // fragment of row 32 i + rl of a tile buffer, k = 16 ks + 8 hh ..: logical chunk 2 ks + hh at slot (2 ks + hh) ^ (rl & 15)
  const uint32_t fbase = rl * 1024 + ((hh ^ (rl & 15)) << 4);
  auto frag = [&](const unsigned char* buf, int i, int ks) {
    return *reinterpret_cast<const bf16x8*>(buf + i * 32768 + (ks >> 3) * 256 + (fbase ^ ((ks & 7) << 5)));
  };
  // this wave's 32 weight rows as the 32 A-operand fragments of the whole reduction, read out of 1-KB-row DMA images of the panel
  bf16x8 w[WR_KS];
#pragma unroll
  for (int ks = 0; ks < WR_KS; ++ks) w[ks] = zero8();
  // (four passes of 64 panel rows, alternating the two tile buffers, each pass's DMA issued two passes ahead; the first two
  // activation tiles follow them into the buffers as these are released)
  dma64(p.B, p.ldb, n0, N - 1, 0);
  dma64(p.B, p.ldb, n0 + 64, N - 1, 1);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // pass r has landed (this wave's eight pieces of the next one may be in flight)
    bar();
    if ((wave >> 1) == r) {
#pragma unroll
      for (int ks = 0; ks < WR_KS; ++ks) w[ks] = frag(wreg_smem + (r & 1) * WR_ABUF, wave & 1, ks);
    }
    bar();
    if (r < 2) dma64(p.B, p.ldb, n0 + 64 * (r + 2), N - 1, r & 1);
    else dma64(p.A, p.lda, r0 + (r - 2) * WR_TM, M - 1, r & 1);   // tiles 0 and 1 (tile 1 of a one-tile group: clamped rows, never used)
  }

  constexpr bool STAGE_BF16 = (EPI == TTTS_EPI_STORE_BF16 || EPI == TTTS_EPI_GELU_BF16);   // the stored value IS bf16(acc + bias)
  const bool bf16_out = (EPI == TTTS_EPI_STORE_BF16 || EPI == TTTS_EPI_GELU_BF16 || EPI == TTTS_EPI_DGELU_BF16);
  const bool vec_ok = bf16_out ? ((p.e.ldc & 7) == 0) : ((p.e.ldc & 3) == 0);
  // bias: added in the accumulator layout, where a lane's 16 columns depend only on its wave and its half -- the wave's 32
  // values live in SGPRs (autocast-rounded once), a lane picks its half per use; no vector register is held for it.
  float bias_s[32];
#pragma unroll
  for (int t = 0; t < 32; ++t) bias_s[t] = 0.f;
  if (p.e.bias) {
#pragma unroll
    for (int t = 0; t < 32; ++t) {
      const float b0 = p.e.bias[min(nw + t, N - 1)];
      const float b = (EPI == TTTS_EPI_STORE_F32) ? b0 : (float)(bf16)b0;
      bias_s[t] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b)));
    }
  }
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  unsigned char* stage = wreg_smem + 2 * WR_ABUF + wave * WR_STAGE;   // this wave's own 4 KB
  const uint32_t stage_lds = lds0 + 2 * WR_ABUF + wave * WR_STAGE;
  const int er = lane >> 2, ec = lane & 3;             // read-back: lane -> (row er of a 16-row pass, columns nw + 8 ec .. + 7)

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tiles 0 and 1 (this wave's pieces)
  bar();

  WR_STAMP();                                        // [1]: prologue done
  for (int ph = 0; ph <= 2 * ntile; ++ph) {
    WR_STAMP();                                      // phase start
    // even phases: the tile buffer both groups finished with at the end of the previous phase is refilled two tiles ahead
    if (!(ph & 1) && ph >= 2 && (ph >> 1) + 1 < ntile) dma64(p.A, p.lda, r0 + ((ph >> 1) + 1) * WR_TM, M - 1, ((ph >> 1) + 1) & 1);
    const int t = (ph - grp2) >> 1;                    // the tile this wave computes (its MFMA phases) or has just computed
    if ((ph & 1) == grp2) {
      // ---------------- MFMA phase: tile t, 64 MFMAs ----------------
      if (t < ntile) {
        const int m0 = r0 + t * WR_TM;
        const bool two = m0 + 32 < r1;                // the second 32-row block holds rows (workgroup-uniform)
        const unsigned char* ab = wreg_smem + (t & 1) * WR_ABUF;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        // fragment pipeline, order pinned: the reads of k-step ks + 2 are issued in front of the MFMAs of k-step ks (this wave is
        // the only one feeding its SIMD's matrix pipe: two k-steps = 128 cycles cover the LDS latency); it runs at raised priority --
        // the partner wave's epilogue instructions take the issue slots the MFMA stream leaves
        __builtin_amdgcn_s_setprio(2);
        if (two) {
          bf16x8 fa[3][2];
          fa[0][0] = frag(ab, 0, 0); fa[0][1] = frag(ab, 1, 0);
          fa[1][0] = frag(ab, 0, 1); fa[1][1] = frag(ab, 1, 1);
#pragma unroll
          for (int ks = 0; ks < WR_KS; ++ks) {
            if (ks + 2 < WR_KS) { fa[(ks + 2) % 3][0] = frag(ab, 0, ks + 2); fa[(ks + 2) % 3][1] = frag(ab, 1, ks + 2); }
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = mfma32(w[ks], fa[ks % 3][0], acc[0]);
            acc[1] = mfma32(w[ks], fa[ks % 3][1], acc[1]);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          bf16x8 fa[4];
          fa[0] = frag(ab, 0, 0); fa[1] = frag(ab, 0, 1); fa[2] = frag(ab, 0, 2);
#pragma unroll
          for (int ks = 0; ks < WR_KS; ++ks) {
            if (ks + 3 < WR_KS) fa[(ks + 3) & 3] = frag(ab, 0, ks + 3);
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = mfma32(w[ks], fa[ks & 3], acc[0]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      WR_STAMP();                                    // MFMAs issued
      __builtin_amdgcn_s_setprio(0);
      // group 1 issued its tile pieces at the start of its previous (epilogue) phase and the tile is read two phases after that:
      // it has to publish them here, behind that phase's stores.  Group 0 (pieces issued at the start of THIS phase) waits at
      // the start of its epilogue phase instead, before it issues any store -- nothing young is in flight there.
      if (grp2 == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (t >= 0 && t < ntile) {
      // ---------------- epilogue phase: the tile computed in the previous phase ----------------
      if (grp2 == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // a phase old: its tile pieces
      const int m0 = r0 + t * WR_TM;
      const bool two = m0 + 32 < r1;
      if (STAGE_BF16) {
        // private bf16 stage [64 rows][4 slots of 16 B], slot s of row m at s ^ ((m >> 2) & 3): conflict-free 8-byte writes
        // (16 lanes = 16 rows of one slot) and 16-byte reads (16 lanes = 4 rows x 4 slots)
        bf16* sb = reinterpret_cast<bf16*>(stage);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (i == 1 && !two) break;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16)(acc[i][4 * q + e] + (hh ? bias_s[8 * q + 4 + e] : bias_s[8 * q + e]));
            *reinterpret_cast<bf16x4*>(sb + (i * 32 + rl) * 32 + ((q ^ ((rl >> 2) & 3)) << 3) + 4 * hh) = o;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        constexpr int RB = 4;                         // all four read-back chunks in flight at once
#pragma unroll
        for (int pb = 0; pb < 4; pb += RB) {
          bf16x8 ob[RB];
#pragma unroll
          for (int ps = 0; ps < RB; ++ps) ob[ps] = *reinterpret_cast<const bf16x8*>(sb + ((pb + ps) * 16 + er) * 32 + ((ec ^ ((((pb + ps) * 16 + er) >> 2) & 3)) << 3));
          __builtin_amdgcn_sched_barrier(0);          // the reads are in flight before the first pass waits
#pragma unroll
          for (int ps = 0; ps < RB; ++ps) {
            const int row = (pb + ps) * 16 + er, m = m0 + row, n = nw + ec * 8;
            const bf16x8 o = ob[ps];
            if (m >= r1 || n >= N) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)o[e];   // exact: epi_row8 rounds it back to the same bf16
            epi_row8<EPI>(p.e, m, n, v, vec_ok, cs);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // stage drained (the next MFMA phase's DMA may overwrite it)
      } else {
        // private fp32 stage, one 32-row block per round: [32 rows][8 slots of 16 B], slot s of row m at s ^ ((m >> 1) & 7)
        float* sf = reinterpret_cast<float*>(stage);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (i == 1 && !two) break;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(sf + rl * 32 + (((2 * q + hh) ^ ((rl >> 1) & 7)) << 2)) =
                make_float4(acc[i][4 * q] + (hh ? bias_s[8 * q + 4] : bias_s[8 * q]), acc[i][4 * q + 1] + (hh ? bias_s[8 * q + 5] : bias_s[8 * q + 1]),
                            acc[i][4 * q + 2] + (hh ? bias_s[8 * q + 6] : bias_s[8 * q + 2]), acc[i][4 * q + 3] + (hh ? bias_s[8 * q + 7] : bias_s[8 * q + 3]));
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int ps = 0; ps < 2; ++ps) {
            const int row = ps * 16 + er, m = m0 + i * 32 + row, n = nw + ec * 8;
            const int x = (row >> 1) & 7;
            const float4 a = *reinterpret_cast<const float4*>(sf + row * 32 + (((2 * ec) ^ x) << 2));
            const float4 b = *reinterpret_cast<const float4*>(sf + row * 32 + (((2 * ec + 1) ^ x) << 2));
            if (m >= r1 || n >= N) continue;
            float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            epi_row8<EPI>(p.e, m, n, v, vec_ok, cs);
            __builtin_amdgcn_sched_barrier(0);
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // round drained before the next one overwrites the stage
        }
      }
    }
    WR_STAMP();                                      // role done, before the barrier
    bar();
  }
  WR_STAMP();
  if (!WR_TRACE && EPI == TTTS_EPI_STORE_BF16 && p.e.colsum) {
    // column sums: lanes l, l + 4, ... of a wave hold the same 8 columns; one atomic per column and wave (32 row groups: as
    // contended as the tiled kernels' one atomic per column and tile)
#pragma unroll
    for (int o = 4; o < 64; o <<= 1)
#pragma unroll
      for (int e = 0; e < 8; ++e) cs[e] += __shfl_xor(cs[e], o, 64);
    if (lane < 4) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (nw + lane * 8 + e < N) atomicAdd(p.e.colsum + nw + lane * 8 + e, cs[e]);
    }
  }
}

// ---- NT over WHOLE ROWS (N = 512 = the model width): residual GEMM + the LayerNorm that follows it, one launch -------------------
// x_out = resid_in + dropout(bf16(A . W^T + bias)) is the fp32 residual stream; the next thing the model does with it is
// LayerNorm (GPT2Block: ln_2 after the attention projection, the next block's ln_1 -- or ln_f -- after the MLP projection,
// modeling_gpt2.py:229-309).  A 160 x 128 tile cannot normalise -- a row's statistics need all 512 columns -- so the LayerNorm was
// its own launch re-reading x_out (8.7 us x 14 per step, a latency-bound kernel).  Here a workgroup owns 64 WHOLE rows: eight
// waves x 64 columns, the 128 x 128 kernels' wave tile, two 72-KB LDS-DMA stages (64 rows of A + all 512 rows of W per 64-deep
// k-step: the weight is re-streamed from the L2 by every workgroup -- the same fill bytes per flop as a 128 x 128 tile), and the
// epilogue, through an fp32 stage of the whole 64 x 512 tile, does per row what epi_row8<RESID_ADD> and ln_fwd_kernel do: one
// wave per row, lane l on columns 4 l .. 4 l + 3 and 256 + 4 l .. + 3 -- ln_fwd_kernel's own element -> lane map, so the sums
// are formed in its order and x_out, mean, rstd and the normalised copy are BIT-IDENTICAL to the two-launch path
// (tests/test_gpu_kernels.py).  Measured (round 4, profiles/r04_ab_fused_ln.txt): stand-alone at K = 512 25.1 us against 32.0 for
// the two launches, at K = 2048 51.5 against 50.3 (145 workgroups = 57 % of the CUs stream 64 KB of weights per k-step each);
// inside the train step the K = 512 form LOSES 3.5 us per layer (same-box A/B 3.393 vs 3.372 ms): there the fp32 residual and
// the attention output come from HBM and 145 CUs cannot pull them as fast as 256.  The GPT engine therefore keeps the two
// launches by default (TTTS_FUSED_LN=1 opts in); the entry point stays for shapes where a launch costs more than it does here.
constexpr int RL_N = 512, RL_TM = 64, RL_BK = 64, RL_STAGE_EL = (RL_TM + RL_N) * RL_BK, RL_LDS = 2 * RL_STAGE_EL * 2;   // 2 x 72 KB
struct RowLnParams {
  const bf16* A; int64_t lda;
  const bf16* B; int64_t ldb;
  int M, K;
  const float* bias; const float* resid_in; float* x_out;
  uint32_t thr; float inv_keep; uint32_t seed_lo, seed_hi; const uint32_t* ctr;
  const float* gamma; const float* beta; float eps;
  void* y; int y_is_bf16; float* mean; float* rstd;
};
__global__ __launch_bounds__(512, 2) void gemm_nt_rowln_kernel(RowLnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rl_smem[];   // [2][A 64 x 64 | W 512 x 64] bf16; reused as the fp32 stage
  bf16* smem = reinterpret_cast<bf16*>(rl_smem);
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, rl = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = xcd_tile(blockIdx.x, gridDim.x) * RL_TM;
  const int nk = p.K / RL_BK;
  auto fsw = [](int r) { return (r >> 1) & 7; };
  // DMA: one wave instruction = 8 rows x 128 B; lane -> row l >> 3, slot l & 7 holding logical chunk (l & 7) ^ fsw(row).
  // A: 8 pieces (one per wave), W: 64 pieces (eight per wave).
  const bf16* ga;
  const bf16* gb[8];
  {
    const int r = wave * 8 + (lane >> 3);
    ga = p.A + (int64_t)min(m0 + r, p.M - 1) * p.lda + (((lane & 7) ^ fsw(r)) << 3);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int n = (wave * 8 + i) * 8 + (lane >> 3);
      gb[i] = p.B + (int64_t)n * p.ldb + (((lane & 7) ^ fsw(n)) << 3);
    }
  }
  auto issue = [&](int kt, int buf) {
    bf16* st = smem + buf * RL_STAGE_EL;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga + kt * RL_BK),
                                     (__attribute__((address_space(3))) void*)(st + wave * 8 * RL_BK), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb[i] + kt * RL_BK),
                                       (__attribute__((address_space(3))) void*)(st + RL_TM * RL_BK + (wave * 8 + i) * 8 * RL_BK), 16, 0, 0);
  };
  f32x16 acc[2][2];   // [j: 32-col block of this wave's 64 columns][i: 32-row block]; D = Wtile . Atile^T (rows = n, cols = m)
  ZERO_ACC(acc)
  issue(0, 0);
  __syncthreads();    // (hipcc drains the LDS-DMA -- vmcnt(0) -- in front of the barrier)
  int aoff[2], boff[2], sw[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = i * 32 + rl, rb = wave * 64 + i * 32 + rl;
    aoff[i] = ra * RL_BK;
    boff[i] = RL_TM * RL_BK + rb * RL_BK;
    sw[0][i] = fsw(ra);
    sw[1][i] = fsw(rb);
  }
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
    const bf16* st = smem + buf * RL_STAGE_EL;
#pragma unroll
    for (int ks = 0; ks < RL_BK / 16; ++ks) {
      bf16x8 af[2], bfr[2];
      const int lc = ks * 2 + hh;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = *reinterpret_cast<const bf16x8*>(st + aoff[i] + ((lc ^ sw[0][i]) << 3));
        bfr[i] = *reinterpret_cast<const bf16x8*>(st + boff[i] + ((lc ^ sw[1][i]) << 3));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[j][i] = mfma32(bfr[j], af[i], acc[j][i]);
    }
    __syncthreads();  // next stage landed and everyone is done reading this one
  }
  // ---- epilogue: fp32 stage [64 rows][128 slots of 16 B], slot s of row m at s ^ (m & 15) ------------------------------------
  float* stage = reinterpret_cast<float*>(rl_smem);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = i * 32 + rl, slot = wave * 16 + j * 8 + 2 * q + hh;
        *reinterpret_cast<float4*>(stage + row * RL_N + ((slot ^ (row & 15)) << 2)) =
            make_float4(acc[j][i][4 * q], acc[j][i][4 * q + 1], acc[j][i][4 * q + 2], acc[j][i][4 * q + 3]);
      }
  __syncthreads();
  // one wave per row, eight rows per wave; lane l: columns 4 l .. + 3 (c = 0) and 256 + 4 l .. + 3 (c = 1)
  float4 bs[2], gm[2], bt[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int n = c * 256 + 4 * lane;
    bs[c] = p.bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    bs[c] = make_float4((float)(bf16)bs[c].x, (float)(bf16)bs[c].y, (float)(bf16)bs[c].z, (float)(bf16)bs[c].w);   // autocast rounds the bias
    gm[c] = *reinterpret_cast<const float4*>(p.gamma + n);
    bt[c] = *reinterpret_cast<const float4*>(p.beta + n);
  }
  const uint32_t shi = p.thr ? seed_mix(p.seed_hi, p.ctr) : 0u;
  // all eight rows' residual loads first (16 x 16 B per lane in flight: the rows are then processed without a memory round trip each)
  const float* rbase = p.resid_in ? p.resid_in : p.x_out;
  float4 rsd[8][2];
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {
    const int m = min(m0 + wave * 8 + rr, p.M - 1);
#pragma unroll
    for (int c = 0; c < 2; ++c) rsd[rr][c] = nt_load16<float4>(rbase + (int64_t)m * RL_N + c * 256 + 4 * lane);
  }
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {
    const int row = wave * 8 + rr, m = m0 + row;
    if (m >= p.M) break;                             // (wave-uniform)
    float4 v[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int n = c * 256 + 4 * lane;
      const float4 a = *reinterpret_cast<const float4*>(stage + row * RL_N + (((c * 64 + lane) ^ (row & 15)) << 2));
      const float4 r = rsd[rr][c];
      float y0 = (float)(bf16)(a.x + bs[c].x), y1 = (float)(bf16)(a.y + bs[c].y), y2 = (float)(bf16)(a.z + bs[c].z), y3 = (float)(bf16)(a.w + bs[c].w);
      if (p.thr) {                                   // resid_pdrop: element index m * N + n, 16 random bits per element (two elements per hash)
        const uint32_t lin = (uint32_t)(((int64_t)m * RL_N + n) >> 1);
        const uint32_t h0 = hash32(lin, p.seed_lo, shi), h1 = hash32(lin + 1, p.seed_lo, shi);
        y0 = (h0 & 0xFFFFu) >= p.thr ? y0 * p.inv_keep : 0.f;
        y1 = (h0 >> 16) >= p.thr ? y1 * p.inv_keep : 0.f;
        y2 = (h1 & 0xFFFFu) >= p.thr ? y2 * p.inv_keep : 0.f;
        y3 = (h1 >> 16) >= p.thr ? y3 * p.inv_keep : 0.f;
      }
      v[c] = make_float4(r.x + y0, r.y + y1, r.z + y2, r.w + y3);
      nt_store16<float4>(p.x_out + (int64_t)m * RL_N + n, v[c]);
    }
    // LayerNorm of the row: ln_fwd_kernel's own arithmetic (common.hpp), same element -> lane map, hence the same bits
    const bool ok2[2] = {true, true};
    float mean, rstd;
    ln_row_stats<2>(v, ok2, RL_N, p.eps, mean, rstd);
    if (lane == 0) {
      p.mean[m] = mean;
      p.rstd[m] = rstd;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int n = c * 256 + 4 * lane;
      const float4 o = ln_row_apply(v[c], mean, rstd, gm[c], bt[c]);
      if (p.y_is_bf16) {
        bf16x4 ob;
        ob[0] = (bf16)o.x; ob[1] = (bf16)o.y; ob[2] = (bf16)o.z; ob[3] = (bf16)o.w;
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.y) + (int64_t)m * RL_N + n) = ob;
      } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + (int64_t)m * RL_N + n) = o;
      }
    }
  }
}

// ---- NT, register-staged main loop (any K % 8 == 0; zero-fills ragged K) ---------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmNtParams p) {
  __shared__ __attribute__((aligned(16))) bf16 smem[2 * 2 * BM * NT_LDS_STRIDE];  // 72 KB
  auto As = [&](int buf) { return smem + (2 * buf) * BM * NT_LDS_STRIDE; };
  auto Bs = [&](int buf) { return smem + (2 * buf + 1) * BM * NT_LDS_STRIDE; };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (p.e.N + BN - 1) / BN;
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const int nk = (p.K + BK - 1) / BK;

  // staging map: chunk c = tid + i*256 (i < 4): row = c >> 3, 16-byte k-chunk = c & 7
  bf16x8 ra[4], rb[4];
  auto load_regs = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 256, row = c >> 3, kc = (c & 7) * 8;
      const int k = kt * BK + kc;
      const int gm = m0 + row, gn = n0 + row;
      ra[i] = (gm < p.e.M && k < p.K) ? *reinterpret_cast<const bf16x8*>(p.A + (int64_t)gm * p.lda + k) : zero8();
      rb[i] = (gn < p.e.N && k < p.K) ? *reinterpret_cast<const bf16x8*>(p.B + (int64_t)gn * p.ldb + k) : zero8();
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 256, row = c >> 3, kc = (c & 7) * 8;
      *reinterpret_cast<bf16x8*>(As(buf) + row * NT_LDS_STRIDE + kc) = ra[i];
      *reinterpret_cast<bf16x8*>(Bs(buf) + row * NT_LDS_STRIDE + kc) = rb[i];
    }
  };

  f32x16 acc[2][2];
  ZERO_ACC(acc)
  load_regs(0);
  store_lds(0);
  __syncthreads();
  const int frow = lane & 31, fk = (lane >> 5) * 8;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_regs(kt + 1);
    const bf16* as = As(buf) + (wm * 64 + frow) * NT_LDS_STRIDE + fk;
    const bf16* bs = Bs(buf) + (wn * 64 + frow) * NT_LDS_STRIDE + fk;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = *reinterpret_cast<const bf16x8*>(as + i * 32 * NT_LDS_STRIDE + ks * 16);
        bfr[i] = *reinterpret_cast<const bf16x8*>(bs + i * 32 * NT_LDS_STRIDE + ks * 16);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[j][i] = mfma32(bfr[j], af[i], acc[j][i]);
    }
    if (kt + 1 < nk) store_lds(buf ^ 1);
    __syncthreads();
  }
  tile_epilogue<EPI>(p.e, acc, reinterpret_cast<float*>(smem), m0, n0, tid);
}

// -------------------------------------------------------------------------------------------------------
constexpr int BK_TN = 32;  // reduction rows per LDS tile: 40 KB of LDS per workgroup -> 3 workgroups per CU

struct GemmTnParams {
  const bf16* At; int64_t ldat;
  const bf16* Bt; int64_t ldbt;
  float* C; int64_t ldc;
  float* partial;      // [splits][Mo][ldp] fp32 slabs when splits > 1
  int64_t ldp;
  int Mo, No, Kr, k_chunk, splits;
};

__global__ __launch_bounds__(256, 3) void gemm_tn_kernel(GemmTnParams p) {
  __shared__ __attribute__((aligned(16))) bf16 smem[2 * 2 * BK_TN * TN_LDS_STRIDE];  // 40 KB; reused as the stage
  auto As = [&](int buf) { return smem + (2 * buf) * BK_TN * TN_LDS_STRIDE; };      // [k][m]
  auto Bs = [&](int buf) { return smem + (2 * buf + 1) * BK_TN * TN_LDS_STRIDE; };  // [k][n]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (p.No + BN - 1) / BN;
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const int kbeg = blockIdx.y * p.k_chunk;
  const int kend = min(p.Kr, kbeg + p.k_chunk);
  const int nk = (kend - kbeg + BK_TN - 1) / BK_TN;

  // staging map: chunk c = tid + i*256 (i < 2): k-row = c >> 4, 16-byte chunk along m/n = c & 15
  bf16x8 ra[2], rb[2];
  auto load_regs = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + i * 256, kr = c >> 4, mc = (c & 15) * 8;
      const int k = kbeg + kt * BK_TN + kr;
      const bool kv = k < kend;
      ra[i] = (kv && m0 + mc < p.Mo) ? *reinterpret_cast<const bf16x8*>(p.At + (int64_t)k * p.ldat + m0 + mc) : zero8();
      rb[i] = (kv && n0 + mc < p.No) ? *reinterpret_cast<const bf16x8*>(p.Bt + (int64_t)k * p.ldbt + n0 + mc) : zero8();
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + i * 256, kr = c >> 4, mc = (c & 15) * 8;
      *reinterpret_cast<bf16x8*>(As(buf) + kr * TN_LDS_STRIDE + mc) = ra[i];
      *reinterpret_cast<bf16x8*>(Bs(buf) + kr * TN_LDS_STRIDE + mc) = rb[i];
    }
  };

  f32x16 acc[2][2];
  ZERO_ACC(acc)
  if (nk > 0) {
    load_regs(0);
    store_lds(0);
  }
  __syncthreads();
  // transposed-read lane map: 16-lane group g = lane >> 4 covers columns 16*(g & 1) + 0..15 of a 32-wide block and
  // k-rows 8*(g >> 1) + {0..3} (first read) / + {4..7} (second read); lane i' = lane & 15 addresses
  // [k + (i' >> 2)][col + 4*(i' & 3)] and receives column (lane & 31) of the block.
  const int g = lane >> 4, ip = lane & 15;
  const int tr_off = (8 * (g >> 1) + (ip >> 2)) * TN_LDS_STRIDE + 16 * (g & 1) + 4 * (ip & 3);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_regs(kt + 1);
    const bf16* as = As(buf) + tr_off + wm * 64;
    const bf16* bs = Bs(buf) + tr_off + wn * 64;
#pragma unroll
    for (int ks = 0; ks < BK_TN / 16; ++ks) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bf16* pa = as + ks * 16 * TN_LDS_STRIDE + i * 32;
        const bf16* pb = bs + ks * 16 * TN_LDS_STRIDE + i * 32;
        af[i] = cat4(lds_tr_b64(pa), lds_tr_b64(pa + 4 * TN_LDS_STRIDE));
        bfr[i] = cat4(lds_tr_b64(pb), lds_tr_b64(pb + 4 * TN_LDS_STRIDE));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[j][i] = mfma32(bfr[j], af[i], acc[j][i]);
    }
    if (kt + 1 < nk) store_lds(buf ^ 1);
    __syncthreads();
  }
  // one split: C += acc (this workgroup owns the tile); several: store the fp32 slab, gemm_tn_reduce_kernel sums the
  // slabs in a fixed order (deterministic; device-scope fp32 atomics were ~6x slower here)
  GemmEpi e{};
  e.M = p.Mo;
  e.N = p.No;
  if (p.splits > 1) {
    e.C = p.partial + (int64_t)blockIdx.y * p.Mo * p.ldp;
    e.ldc = p.ldp;
    tile_epilogue<EPI_SLAB_F32>(e, acc, reinterpret_cast<float*>(smem), m0, n0, tid);
  } else {
    e.C = p.C;
    e.ldc = p.ldc;
    tile_epilogue<EPI_ACCUM_F32>(e, acc, reinterpret_cast<float*>(smem), m0, n0, tid);
  }
}

// ---- TN, LDS-DMA main loop (k range a multiple of 64 rows) -----------------------------------------------------------
// LDS image per operand and buffer: [64 k-rows][128 cols] bf16, 256-byte rows, no padding (DMA is lane-linear:
// instruction (wave w, i) covers k-rows w*16 + i*4 .. +4, lane l -> row += l >> 4, 16-byte slot l & 15).  The four
// 64-byte segments of row k are XOR-permuted by (k & 3): a transposed read touches 4 consecutive k-rows x one 64-byte
// segment per half-wave, which the permutation spreads over all 64 banks.
__global__ __launch_bounds__(256, 2) void gemm_tn_glds_kernel(GemmTnParams p) {
  __shared__ __attribute__((aligned(16))) bf16 smem[2 * 2 * 64 * 128];  // 64 KB: [buf][A|B][64*128]; reused as stage
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (p.No + BN - 1) / BN;
  // XCD-aware order over (split, tile): each XCD walks a contiguous range, i.e. (mostly) one reduction split with all
  // its output tiles -- the At / Bt panels of that split are then fetched into that XCD's L2 once
  const int lin = xcd_tile(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int tile = lin % gridDim.x, split = lin / gridDim.x;
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int kbeg = split * p.k_chunk;
  const int kend = min(p.Kr, kbeg + p.k_chunk);
  const int nk = (kend - kbeg) / 64;

  // DMA sources: columns beyond Mo / No are clamped to the last valid 16-byte chunk (those outputs are never stored)
  const int ca_max = max(0, ((p.Mo - m0 + 7) >> 3) - 1), cb_max = max(0, ((p.No - n0 + 7) >> 3) - 1);
  const bf16* ga[4];
  const bf16* gb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wave * 16 + i * 4 + (lane >> 4);
    const int lc = (lane & 15) ^ ((r & 3) << 2);
    ga[i] = p.At + (int64_t)(kbeg + r) * p.ldat + m0 + min(lc, ca_max) * 8;
    gb[i] = p.Bt + (int64_t)(kbeg + r) * p.ldbt + n0 + min(lc, cb_max) * 8;
  }
  // (inline-asm DMA: see lds_dma16_untracked in common.hpp -- the builtin would serialise prefetch and fragment reads)
  const uint32_t lds0 = lds_byte_addr(smem);
  auto issue = [&](int kt, int buf) {
    const uint32_t as = lds0 + (uint32_t)(((buf * 2 + 0) * 64 * 128 + wave * 16 * 128) * sizeof(bf16));
    const uint32_t bs = lds0 + (uint32_t)(((buf * 2 + 1) * 64 * 128 + wave * 16 * 128) * sizeof(bf16));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      lds_dma16_untracked(ga[i] + (int64_t)kt * 64 * p.ldat, as + (uint32_t)(i * 4 * 128 * sizeof(bf16)));
      lds_dma16_untracked(gb[i] + (int64_t)kt * 64 * p.ldbt, bs + (uint32_t)(i * 4 * 128 * sizeof(bf16)));
    }
  };

  f32x16 acc[2][2];
  ZERO_ACC(acc)
  if (nk > 0) issue(0, 0);
  lds_dma_wait_all();
  __syncthreads();
  // transposed-read lane map (see gemm_tn_kernel) on the permuted image: lane i' = lane & 15 of group g reads k-row
  // 8*(g >> 1) + (i' >> 2) (+4), logical column block + 16*(g & 1) + 4*(i' & 3)
  const int g = lane >> 4, ip = lane & 15;
  const int krow = 8 * (g >> 1) + (ip >> 2);
  const int sw = ((ip >> 2) & 3) << 2;  // (k & 3) << 2 for both reads (k-row offsets are multiples of 4)
  int offa[2], offb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ca = wm * 64 + i * 32 + 16 * (g & 1) + 4 * (ip & 3), cb = wn * 64 + i * 32 + 16 * (g & 1) + 4 * (ip & 3);
    offa[i] = krow * 128 + (((ca >> 3) ^ sw) << 3) + (ca & 7);
    offb[i] = krow * 128 + (((cb >> 3) ^ sw) << 3) + (cb & 7);
  }
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
    const bf16* as = smem + (buf * 2 + 0) * 64 * 128;
    const bf16* bs = smem + (buf * 2 + 1) * 64 * 128;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bf16* pa = as + ks * 16 * 128 + offa[i];
        const bf16* pb = bs + ks * 16 * 128 + offb[i];
        af[i] = cat4(lds_tr_b64(pa), lds_tr_b64(pa + 4 * 128));
        bfr[i] = cat4(lds_tr_b64(pb), lds_tr_b64(pb + 4 * 128));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[j][i] = mfma32(bfr[j], af[i], acc[j][i]);
    }
    lds_dma_wait_all();  // the next tile has landed (this wave's share; the barrier publishes everyone's)
    __syncthreads();     // ... and everyone is done reading this one
  }
  GemmEpi e{};
  e.M = p.Mo;
  e.N = p.No;
  if (p.splits > 1) {
    e.C = p.partial + (int64_t)split * p.Mo * p.ldp;
    e.ldc = p.ldp;
    tile_epilogue<EPI_SLAB_F32>(e, acc, reinterpret_cast<float*>(smem), m0, n0, tid);
  } else {
    e.C = p.C;
    e.ldc = p.ldc;
    tile_epilogue<EPI_ACCUM_F32>(e, acc, reinterpret_cast<float*>(smem), m0, n0, tid);
  }
}

// ---- TN, grouped: several weight-gradient problems in ONE launch, every tile over its FULL reduction ----------------
// Why: one dW GEMM of the GPT step has 16..64 output tiles, so it needs a split reduction (8-10 slabs, 25-40 MB of fp32
// written and re-read, a second launch) to fill 256 CUs, and its main loop is 15-19 k-tiles long.  All dW GEMMs of a
// backward pass together have >= 1000 tiles: launched as one grid each workgroup owns a whole tile (C += acc, no slabs,
// no reduce kernel) with a 145-k-tile main loop.  The main loop is gemm_tn_glds_kernel's; a workgroup finds its problem
// from the descriptors' tile prefix sums (one vector load + ballot), and the XCD-aware order keeps the tiles of one
// problem -- which share the At / Bt panels -- on one XCD's L2.
__global__ __launch_bounds__(256, 2) void gemm_tn_grouped_kernel(const ttts_tn_desc* __restrict__ desc, int n_desc) {
  __shared__ __attribute__((aligned(16))) bf16 smem[2 * 2 * 64 * 128];  // 64 KB: [buf][A|B][64*128]; reused as stage
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lin = xcd_tile(blockIdx.x, gridDim.x);
  // problem lookup: lane i holds descriptor i's first tile (n_desc <= 64); the workgroup's problem is the last one that
  // starts at or before its tile
  const int tb = lane < n_desc ? desc[lane].tile_begin : 0x7fffffff;
  const int q = __popcll(__ballot(lin >= tb)) - 1;
  const ttts_tn_desc d = desc[q];
  const bf16* At = reinterpret_cast<const bf16*>(d.At);
  const bf16* Bt = reinterpret_cast<const bf16*>(d.Bt);
  const int64_t ldat = d.ldat, ldbt = d.ldbt;
  const int Mo = d.Mo, No = d.No;
  const int tiles_n = (No + BN - 1) / BN;
  const int tile = lin - d.tile_begin;
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int nk = d.Kr / 64;

  // DMA sources: columns beyond Mo / No are clamped to the last valid 16-byte chunk (those outputs are never stored)
  const int ca_max = max(0, ((Mo - m0 + 7) >> 3) - 1), cb_max = max(0, ((No - n0 + 7) >> 3) - 1);
  const bf16* ga[4];
  const bf16* gb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wave * 16 + i * 4 + (lane >> 4);
    const int lc = (lane & 15) ^ ((r & 3) << 2);
    ga[i] = At + (int64_t)r * ldat + m0 + min(lc, ca_max) * 8;
    gb[i] = Bt + (int64_t)r * ldbt + n0 + min(lc, cb_max) * 8;
  }
  const uint32_t lds0 = lds_byte_addr(smem);
  auto issue = [&](int kt, int buf) {   // inline-asm DMA (lds_dma16_untracked): the prefetch overlaps the fragment reads
    const uint32_t as = lds0 + (uint32_t)(((buf * 2 + 0) * 64 * 128 + wave * 16 * 128) * sizeof(bf16));
    const uint32_t bs = lds0 + (uint32_t)(((buf * 2 + 1) * 64 * 128 + wave * 16 * 128) * sizeof(bf16));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      lds_dma16_untracked(ga[i] + (int64_t)kt * 64 * ldat, as + (uint32_t)(i * 4 * 128 * sizeof(bf16)));
      lds_dma16_untracked(gb[i] + (int64_t)kt * 64 * ldbt, bs + (uint32_t)(i * 4 * 128 * sizeof(bf16)));
    }
  };

  f32x16 acc[2][2];
  ZERO_ACC(acc)
  if (nk > 0) issue(0, 0);
  lds_dma_wait_all();
  __syncthreads();
  // transposed-read lane map on the permuted image: see gemm_tn_glds_kernel
  const int g = lane >> 4, ip = lane & 15;
  const int krow = 8 * (g >> 1) + (ip >> 2);
  const int sw = ((ip >> 2) & 3) << 2;
  int offa[2], offb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ca = wm * 64 + i * 32 + 16 * (g & 1) + 4 * (ip & 3), cb = wn * 64 + i * 32 + 16 * (g & 1) + 4 * (ip & 3);
    offa[i] = krow * 128 + (((ca >> 3) ^ sw) << 3) + (ca & 7);
    offb[i] = krow * 128 + (((cb >> 3) ^ sw) << 3) + (cb & 7);
  }
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
    const bf16* as = smem + (buf * 2 + 0) * 64 * 128;
    const bf16* bs = smem + (buf * 2 + 1) * 64 * 128;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bf16* pa = as + ks * 16 * 128 + offa[i];
        const bf16* pb = bs + ks * 16 * 128 + offb[i];
        af[i] = cat4(lds_tr_b64(pa), lds_tr_b64(pa + 4 * 128));
        bfr[i] = cat4(lds_tr_b64(pb), lds_tr_b64(pb + 4 * 128));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[j][i] = mfma32(bfr[j], af[i], acc[j][i]);
    }
    lds_dma_wait_all();  // the next tile has landed (this wave's share; the barrier publishes everyone's)
    __syncthreads();     // ... and everyone is done reading this one
  }
  GemmEpi e{};
  e.M = Mo;
  e.N = No;
  e.C = d.C;
  e.ldc = d.ldc;
  tile_epilogue<EPI_ACCUM_F32>(e, acc, reinterpret_cast<float*>(smem), m0, n0, tid);
}

// C[m][n] += sum_s partial[s][m][n]   (float4 per thread, slabs read in split order)
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(const float* __restrict__ partial, int64_t ldp, int splits,
                                                             float* __restrict__ C, int64_t ldc, int Mo, int No) {
  const int n4 = (int)(ldp >> 2);
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)Mo * n4) return;
  const int m = (int)(i / n4), n = (int)(i % n4) * 4;
  if (n >= No) return;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int sp = 0; sp < splits; ++sp) {
    const float4 v = *reinterpret_cast<const float4*>(partial + ((int64_t)sp * Mo + m) * ldp + n);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  float* c = C + (int64_t)m * ldc + n;
  const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (n + e < No) c[e] += sv[e];
}

}  // namespace ttts

using namespace ttts;

// Which NT kernel runs a shape, and on what grid (host-side; ttts_gemm_nt_plan_query exposes it to tests and benchmarks).
// All kernels produce the same bits for a shape (tools/ubench/nt_phase.cpp checks that); the rules are measurements on the 256 CUs
// of an MI355X:
//  * RING160: 160 x 128 tiles + a 4-slot ring, when the 128 x 128 tiling has about one tile per CU (tiles in (CUs, 1.6 CUs]) and
//    the 160-row tiles fit one round (a 128 x 64 tile for the N = 512 GEMMs was measured slower -- mlp c_proj 40.2 -> 47.5 us --
//    and removed; so were the 128 x 128 kernels for the K = 512, N = 512 pair: attn c_proj 19.0 -> 21.2 / 21.8 us);
//  * WAVE8: eight waves on a 256 x 128 tile (four 64-row wave groups sharing every B stage; 32-deep stages, two workgroups per
//    CU): 27 % fewer L2 -> LDS fill bytes per flop than two 128 x 128 workgroups, and the fill is what a k-step of the 128 x 128
//    kernel waits for.  Taken where its tile count fills one round of the 2 x CUs slots or makes many rounds:
//    c_attn 23.8 -> 21.9 us = 664 TF/s (37 x 12 = 444 tiles, was 876 in 1.7 rounds), an 8192 x 8194 head 140 -> 113 us (with the
//    start-up stagger); NOT for N = 2048 at M = 9248 (592 tiles = 1.16 rounds: c_fc 38 -> 45 us).  Also measured and dropped: a
//    64-deep-stage version at one workgroup per CU (+8 %), ten waves on 320 x 128 (464 tiles for N = 2048, but 96 VGPRs: spills,
//    +30 ... 58 %), s_setprio around the main loop (within noise), four 32-deep workgroups per CU (no change);
//  * WAVE8_SPLIT, dGELU at N = 2048 (592 eight-wave tiles: 1.16 rounds): the first 2 x CUs workgroups take 256-row tiles (one
//    full round), the remaining rows follow as 64-row tiles on the slots that free up, each a quarter of the work with three of
//    its four wave groups idle: 49.7 -> 45.3 us.  Only for dGELU: the same grid costs the GELU / store / residual epilogues
//    2 ... 10 %, and the tile bookkeeping costs the plain eight-wave kernel 4 %, hence the separate instantiation;
//  * DMA32 + stagger for any other dGELU (32-deep stages, three workgroups per CU, staggered: 52.2 -> 48.7 us);
//  * DMA64 (128 x 128, 64-deep stages, two workgroups per CU) / DMA32 (K % 64 != 0) / REG (register-staged, ragged K) otherwise.
//  * WREG (round 4): weights in registers, one persistent workgroup per CU (gemm_nt_wreg_kernel) for K = 512 and wide N whose
//    256-column panels waste < 10 % (c_attn, c_fc + GELU; not the 1026-column mel head), from 4096 rows up; never dGELU (its
//    saved pre-activation arrives from HBM in a train step: measured no faster there than the split grid, see the kernel).
// The thresholds scale with the CU count of the current device (256 on an MI355X; a host without a GPU plans for 256).
static int device_cus() {
  static int cached[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { (void)hipGetLastError(); return 256; }
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
    cached[dev] = n;   // (benign race: every thread writes the same value)
  }
  return cached[dev];
}
static ttts_gemm_nt_plan plan_nt(int M, int N, int K, int epilogue) {
  const int CUS = device_cus();
  ttts_gemm_nt_plan pl{};
  pl.block = 256; pl.tile_m = 128; pl.tile_n = 128;
  const int grid = (int)(cdiv(M, BM) * cdiv(N, BN));
  pl.grid = grid;
  if (K % 64 != 0) {
    pl.kernel = K % 32 == 0 ? TTTS_NT_KERNEL_DMA32 : TTTS_NT_KERNEL_REG;
    return pl;
  }
  if (epilogue != TTTS_EPI_DGELU_BF16 && K == WR_K && M >= 4096 && N >= 1024 && N % 8 == 0) {
    const int np = (int)cdiv(N, WR_PN);
    if (np <= CUS && (int64_t)np * WR_PN * 10 <= (int64_t)N * 11) {
      const int groups = std::max(1, std::min(CUS / np, (int)cdiv(M, WR_TM)));
      pl.kernel = TTTS_NT_KERNEL_WREG; pl.grid = np * groups; pl.block = 512; pl.tile_m = WR_TM; pl.tile_n = WR_PN;
      pl.main_row_tiles = groups;
      return pl;
    }
  }
  const int tall_grid = (int)(cdiv(M, 160) * cdiv(N, 128));
  if (grid > CUS && grid <= CUS * 8 / 5 && tall_grid <= CUS) {
    pl.kernel = TTTS_NT_KERNEL_RING160; pl.grid = tall_grid; pl.tile_m = 160;
    return pl;
  }
  const int tiles8 = (int)(cdiv(M, 256) * cdiv(N, 128));
  if (tiles8 > CUS && (tiles8 <= 2 * CUS || tiles8 >= 8 * CUS)) {
    pl.kernel = TTTS_NT_KERNEL_WAVE8; pl.grid = tiles8; pl.block = 512; pl.tile_m = 256;
    pl.phase = tiles8 >= 8 * CUS ? 8 : 0;
    return pl;
  }
  if (epilogue == TTTS_EPI_DGELU_BF16) {
    if (tiles8 > 2 * CUS && tiles8 < 8 * CUS) {
      constexpr int TAIL_H = 64;
      const int ncol = (int)cdiv(N, 128), main_rt = 2 * CUS / ncol;
      const int tail_rows = M - main_rt * 256, tail_tiles = (int)cdiv(tail_rows, TAIL_H) * ncol;
      if (main_rt >= 1 && tail_rows > 0 && main_rt * ncol >= CUS * 3 / 2 && tail_tiles <= 2 * CUS) {
        pl.kernel = TTTS_NT_KERNEL_WAVE8_SPLIT; pl.grid = main_rt * ncol + tail_tiles; pl.block = 512; pl.tile_m = 256;
        pl.main_row_tiles = main_rt; pl.tail_tile_rows = TAIL_H;
        return pl;
      }
    }
    pl.kernel = TTTS_NT_KERNEL_DMA32; pl.phase = 3;
    return pl;
  }
  pl.kernel = TTTS_NT_KERNEL_DMA64;
  return pl;
}

// Kernels that need more than 64 KB of dynamic LDS carry a per-function opt-in (hipFuncSetAttribute).  It is set once per
// (function, device) -- an idempotent driver-side attribute of the code object, not state the results depend on -- and a refusal
// is reported instead of being left to fail the launch.  slot: a small id per kernel instantiation.
static bool nt_func_lds(const void* fn, int bytes, int slot) {
  static bool done[16][16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  if (done[dev][slot & 15]) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
    fail(TTTS_EHIP, "gemm_nt: the device refused %d bytes of dynamic LDS: %s", bytes, hipGetErrorString(hipGetLastError()));
    return false;
  }
  done[dev][slot & 15] = true;
  return true;
}

template <int EPI>
static int launch_nt(const GemmNtParams& p, hipStream_t s) {
  const ttts_gemm_nt_plan pl = plan_nt(p.e.M, p.e.N, p.K, EPI);
  GemmNtParams q = p;
  q.phase = pl.phase; q.main_rt = pl.main_row_tiles; q.tail_h = pl.tail_tile_rows;
  switch (pl.kernel) {
    case TTTS_NT_KERNEL_RING160: {
      constexpr size_t smem = (size_t)4 * (160 + 128) * 64 * sizeof(bf16);   // the 4-slot ring: 144 KB
      if (!nt_func_lds(reinterpret_cast<const void*>(gemm_nt_tall_kernel<EPI, 5, 4, 4>), (int)smem, 2 * EPI)) return TTTS_EHIP;
      gemm_nt_tall_kernel<EPI, 5, 4, 4><<<pl.grid, 256, smem, s>>>(q);
      break;
    }
    case TTTS_NT_KERNEL_WREG:
      if constexpr (EPI != TTTS_EPI_DGELU_BF16) {   // (plan_nt never sends dGELU here: see its comment)
        if (!nt_func_lds(reinterpret_cast<const void*>(gemm_nt_wreg_kernel<EPI>), WR_LDS, 1 + 2 * EPI)) return TTTS_EHIP;
        gemm_nt_wreg_kernel<EPI><<<pl.grid, 512, WR_LDS, s>>>(q);
      }
      break;
    case TTTS_NT_KERNEL_WAVE8: gemm_nt_glds_kernel<EPI, 32, 2, 4><<<pl.grid, 512, 0, s>>>(q); break;
    case TTTS_NT_KERNEL_WAVE8_SPLIT: gemm_nt_glds_kernel<EPI, 32, 2, 4, true><<<pl.grid, 512, 0, s>>>(q); break;
    case TTTS_NT_KERNEL_DMA64: gemm_nt_glds_kernel<EPI, 64><<<pl.grid, 256, 0, s>>>(q); break;
    case TTTS_NT_KERNEL_DMA32: gemm_nt_glds_kernel<EPI, 32><<<pl.grid, 256, 0, s>>>(q); break;
    default: gemm_nt_kernel<EPI><<<pl.grid, 256, 0, s>>>(q); break;
  }
  return TTTS_OK;
}

extern "C" int ttts_gemm_nt_plan_query(int32_t M, int32_t N, int32_t K, int32_t epilogue, ttts_gemm_nt_plan* out) {
  TTTS_REQUIRE(out, "gemm_nt_plan_query: null pointer");
  TTTS_REQUIRE(M > 0 && N > 0 && K > 0 && K % 8 == 0, "gemm_nt_plan_query: bad shape M=%d N=%d K=%d", M, N, K);
  TTTS_REQUIRE(epilogue >= TTTS_EPI_STORE_BF16 && epilogue <= TTTS_EPI_STORE_F32, "gemm_nt_plan_query: unknown epilogue %d", epilogue);
  *out = plan_nt(M, N, K, epilogue);
  return TTTS_OK;
}

extern "C" int ttts_gemm_nt_bf16_ex(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                    const float* bias, void* aux, int32_t M, int32_t N, int32_t K, int32_t epilogue,
                                    const float* resid_in, float dropout_p, uint64_t seed, const uint32_t* dropout_counter,
                                    float* colsum, void* stream) {
  TTTS_REQUIRE(A && B && C, "gemm_nt: null pointer");
  TTTS_REQUIRE(!colsum || epilogue == TTTS_EPI_STORE_BF16 || epilogue == TTTS_EPI_DGELU_BF16, "gemm_nt: colsum only with the bf16 STORE / DGELU epilogues");
  TTTS_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_nt: bad shape M=%d N=%d K=%d", M, N, K);
  TTTS_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K, "gemm_nt: K, lda, ldb must be multiples of 8 (K=%d lda=%lld ldb=%lld)", K, (long long)lda, (long long)ldb);
  TTTS_REQUIRE(ldc % 4 == 0 && ldc >= ((N + 3) / 4) * 4, "gemm_nt: ldc must be a multiple of 4 and >= roundup4(N)");
  TTTS_REQUIRE(aligned16(A) && aligned16(B) && aligned16(C), "gemm_nt: 16-byte aligned bases required");
  TTTS_REQUIRE((epilogue != TTTS_EPI_GELU_BF16 && epilogue != TTTS_EPI_DGELU_BF16) || (aux && aligned16(aux)), "gemm_nt: epilogue needs a 16-byte aligned aux");
  TTTS_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "gemm_nt: dropout_p out of range");
  TTTS_REQUIRE(dropout_p == 0.f || (epilogue == TTTS_EPI_RESID_ADD_F32 && N % 8 == 0), "gemm_nt: dropout only with RESID_ADD and N %% 8 == 0");
  TTTS_REQUIRE(!resid_in || aligned16(resid_in), "gemm_nt: resid_in must be 16-byte aligned");
  GemmNtParams p{(const bf16*)A, lda, (const bf16*)B, ldb, K,
                 GemmEpi{C, ldc, bias, (bf16*)aux, resid_in, M, N, dropout_threshold(dropout_p), 1.0f, (uint32_t)seed,
                         (uint32_t)(seed >> 32), dropout_counter, colsum}};
  if (p.e.thr) p.e.inv_keep = 65536.0f / (65536.0f - (float)p.e.thr);
  hipStream_t s = as_stream(stream);
  int rc = TTTS_OK;
  switch (epilogue) {
    case TTTS_EPI_STORE_BF16: rc = launch_nt<TTTS_EPI_STORE_BF16>(p, s); break;
    case TTTS_EPI_GELU_BF16: rc = launch_nt<TTTS_EPI_GELU_BF16>(p, s); break;
    case TTTS_EPI_RESID_ADD_F32: rc = launch_nt<TTTS_EPI_RESID_ADD_F32>(p, s); break;
    case TTTS_EPI_DGELU_BF16: rc = launch_nt<TTTS_EPI_DGELU_BF16>(p, s); break;
    case TTTS_EPI_STORE_F32: rc = launch_nt<TTTS_EPI_STORE_F32>(p, s); break;
    default: return fail(TTTS_EUNSUPPORTED, "gemm_nt: unknown epilogue %d", epilogue);
  }
  if (rc != TTTS_OK) return rc;
  return check_launch("gemm_nt");
}

extern "C" int ttts_gemm_nt_resid_ln_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, const float* bias,
                                          const float* resid_in, float* x_out, int32_t M, int32_t N, int32_t K, float dropout_p,
                                          uint64_t seed, const uint32_t* dropout_counter, const float* gamma, const float* beta,
                                          float eps, void* y, int32_t y_is_bf16, float* mean, float* rstd, void* stream) {
  TTTS_REQUIRE(A && B && x_out && gamma && beta && y && mean && rstd, "gemm_nt_resid_ln: null pointer");
  TTTS_REQUIRE(N == RL_N, "gemm_nt_resid_ln: whole rows of N = %d only (got %d): use ttts_gemm_nt_bf16_ex + ttts_layernorm_fwd", RL_N, N);
  TTTS_REQUIRE(M > 0 && K > 0 && K % RL_BK == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K,
               "gemm_nt_resid_ln: K must be a multiple of %d, lda / ldb multiples of 8 (K=%d lda=%lld ldb=%lld)", RL_BK, K, (long long)lda, (long long)ldb);
  TTTS_REQUIRE(aligned16(A) && aligned16(B) && aligned16(x_out) && aligned16(y) && aligned16(gamma) && aligned16(beta) &&
               (!bias || aligned16(bias)) && (!resid_in || aligned16(resid_in)), "gemm_nt_resid_ln: 16-byte aligned bases required");
  TTTS_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "gemm_nt_resid_ln: dropout_p out of range");
  RowLnParams p{(const bf16*)A, lda, (const bf16*)B, ldb, M, K, bias, resid_in, x_out, dropout_threshold(dropout_p), 1.0f,
                (uint32_t)seed, (uint32_t)(seed >> 32), dropout_counter, gamma, beta, eps, y, y_is_bf16, mean, rstd};
  if (p.thr) p.inv_keep = 65536.0f / (65536.0f - (float)p.thr);
  if (!nt_func_lds(reinterpret_cast<const void*>(gemm_nt_rowln_kernel), RL_LDS, 10)) return TTTS_EHIP;
  gemm_nt_rowln_kernel<<<(int)cdiv(M, RL_TM), 512, RL_LDS, as_stream(stream)>>>(p);
  return check_launch("gemm_nt_resid_ln");
}

extern "C" int ttts_gemm_nt_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                 const float* bias, void* aux, int32_t M, int32_t N, int32_t K, int32_t epilogue,
                                 void* stream) {
  return ttts_gemm_nt_bf16_ex(A, lda, B, ldb, C, ldc, bias, aux, M, N, K, epilogue, nullptr, 0.f, 0, nullptr, nullptr, stream);
}

static void tn_plan(int Mo, int No, int Kr, int& splits, int& k_chunk) {
  const int tiles = (int)(cdiv(Mo, BM) * cdiv(No, BN));
  // enough workgroups to fill 256 CUs (~1.5 per CU), as few slabs as possible, >= 256 reduction rows per split;
  // k_chunk is a multiple of 64 so that the LDS-DMA kernel can take every split whole
  // measured (tools/kernel_bench.py gemm, MI355X): 384 workgroups beats 256 by 10-15 % and ties or beats 512 (more slabs)
  const int target = 384;
  splits = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(Kr, 256), (target + tiles / 2) / tiles));
  k_chunk = (int)(cdiv(cdiv(Kr, splits), 64) * 64);
  splits = (int)cdiv(Kr, k_chunk);
}

extern "C" int64_t ttts_gemm_tn_workspace_bytes(int32_t Mo, int32_t No, int32_t Kr) {
  int splits, k_chunk;
  tn_plan(Mo, No, Kr, splits, k_chunk);
  return splits > 1 ? (int64_t)splits * Mo * (((int64_t)No + 7) / 8 * 8) * (int64_t)sizeof(float) : 0;
}

extern "C" int32_t ttts_tn_desc_tiles(int32_t Mo, int32_t No) {
  return (Mo > 0 && No > 0) ? (int32_t)(cdiv(Mo, BM) * cdiv(No, BN)) : 0;
}

// Validates HOST copies of the descriptors and fills their tile_begin fields; *total_tiles = the launch's grid.
extern "C" int ttts_tn_desc_prepare(ttts_tn_desc* host_desc, int32_t n_desc, int32_t* total_tiles) {
  TTTS_REQUIRE(host_desc && total_tiles, "tn_desc_prepare: null pointer");
  TTTS_REQUIRE(n_desc > 0 && n_desc <= 64, "tn_desc_prepare: 1..64 descriptors (got %d)", n_desc);
  int64_t tiles = 0;
  for (int i = 0; i < n_desc; ++i) {
    ttts_tn_desc& d = host_desc[i];
    TTTS_REQUIRE(d.At && d.Bt && d.C, "tn_desc[%d]: null pointer", i);
    TTTS_REQUIRE(d.Mo > 0 && d.No > 0 && d.Kr > 0, "tn_desc[%d]: bad shape", i);
    TTTS_REQUIRE(d.Kr % 64 == 0, "tn_desc[%d]: Kr must be a multiple of 64 (zero-pad the operands' rows); got %d", i, d.Kr);
    TTTS_REQUIRE(d.ldat % 8 == 0 && d.ldbt % 8 == 0 && d.ldat >= ((d.Mo + 7) / 8) * 8 && d.ldbt >= ((d.No + 7) / 8) * 8,
                 "tn_desc[%d]: ldat/ldbt must be multiples of 8 and cover roundup8(Mo/No)", i);
    TTTS_REQUIRE(d.ldc % 4 == 0 && d.ldc >= ((d.No + 3) / 4) * 4, "tn_desc[%d]: ldc must be a multiple of 4 and >= roundup4(No)", i);
    TTTS_REQUIRE(aligned16(d.At) && aligned16(d.Bt) && aligned16(d.C), "tn_desc[%d]: 16-byte aligned bases required", i);
    d.tile_begin = (int32_t)tiles;
    tiles += ttts_tn_desc_tiles(d.Mo, d.No);
    TTTS_REQUIRE(tiles < (int64_t)1 << 30, "tn_desc: too many tiles");
  }
  *total_tiles = (int32_t)tiles;
  return TTTS_OK;
}

extern "C" int ttts_gemm_tn_grouped_bf16_accum_f32(const ttts_tn_desc* desc_dev, int32_t n_desc, int32_t total_tiles,
                                                   void* stream) {
  TTTS_REQUIRE(desc_dev, "gemm_tn_grouped: null descriptor table");
  TTTS_REQUIRE(n_desc > 0 && n_desc <= 64 && total_tiles > 0, "gemm_tn_grouped: bad counts (n_desc=%d tiles=%d)", n_desc, total_tiles);
  gemm_tn_grouped_kernel<<<total_tiles, 256, 0, as_stream(stream)>>>(desc_dev, n_desc);
  return check_launch("gemm_tn_grouped");
}

extern "C" int ttts_gemm_tn_bf16_accum_f32(const void* At, int64_t ldat, const void* Bt, int64_t ldbt, float* C,
                                           int64_t ldc, int32_t Mo, int32_t No, int32_t Kr, void* workspace,
                                           void* stream) {
  TTTS_REQUIRE(At && Bt && C, "gemm_tn: null pointer");
  TTTS_REQUIRE(Mo > 0 && No > 0 && Kr > 0, "gemm_tn: bad shape");
  TTTS_REQUIRE(ldat % 8 == 0 && ldbt % 8 == 0 && ldat >= ((Mo + 7) / 8) * 8 && ldbt >= ((No + 7) / 8) * 8,
               "gemm_tn: ldat/ldbt must be multiples of 8 and cover roundup8(Mo/No)");
  TTTS_REQUIRE(aligned16(At) && aligned16(Bt) && aligned16(C), "gemm_tn: 16-byte aligned bases required");
  int splits, k_chunk;
  tn_plan(Mo, No, Kr, splits, k_chunk);
  TTTS_REQUIRE(splits == 1 || (workspace && aligned16(workspace)), "gemm_tn: workspace (ttts_gemm_tn_workspace_bytes) required");
  const int tiles = (int)(cdiv(Mo, BM) * cdiv(No, BN));
  const int64_t ldp = ((int64_t)No + 7) / 8 * 8;
  hipStream_t s = as_stream(stream);
  const int kr_main = Kr / 64 * 64;  // whole 64-row tiles: LDS-DMA kernel
  if (kr_main > 0) {
    const int sp = (int)cdiv(kr_main, k_chunk);
    GemmTnParams p{(const bf16*)At, ldat, (const bf16*)Bt, ldbt, C, ldc, (float*)workspace, ldp, Mo, No, kr_main, k_chunk, sp};
    gemm_tn_glds_kernel<<<dim3(tiles, sp), 256, 0, s>>>(p);
    int rc = check_launch("gemm_tn");
    if (rc) return rc;
    if (sp > 1) {
      gemm_tn_reduce_kernel<<<(int)cdiv((int64_t)Mo * (ldp / 4), 256), 256, 0, s>>>((const float*)workspace, ldp, sp, C, ldc, Mo, No);
      rc = check_launch("gemm_tn_reduce");
      if (rc) return rc;
    }
  }
  if (kr_main < Kr) {  // ragged tail (< 64 rows, or everything in the debug path): register-staged kernel, C += directly
    const int kt = Kr - kr_main;
    if (kr_main > 0 || splits == 1) {
      GemmTnParams p{(const bf16*)At + (int64_t)kr_main * ldat, ldat, (const bf16*)Bt + (int64_t)kr_main * ldbt, ldbt, C, ldc,
                     nullptr, ldp, Mo, No, kt, kt, 1};
      gemm_tn_kernel<<<dim3(tiles, 1), 256, 0, s>>>(p);
      return check_launch("gemm_tn_tail");
    }
    GemmTnParams p{(const bf16*)At, ldat, (const bf16*)Bt, ldbt, C, ldc, (float*)workspace, ldp, Mo, No, Kr, k_chunk, splits};
    gemm_tn_kernel<<<dim3(tiles, splits), 256, 0, s>>>(p);
    int rc = check_launch("gemm_tn");
    if (rc) return rc;
    gemm_tn_reduce_kernel<<<(int)cdiv((int64_t)Mo * (ldp / 4), 256), 256, 0, s>>>((const float*)workspace, ldp, splits, C, ldc, Mo, No);
    return check_launch("gemm_tn_reduce");
  }
  return TTTS_OK;
}
