// Persistent, wave-specialised bf16 NT GEMM of the GPT train step: C[M,N] = epi(A[M,K] . B[N,K]^T), K % 64 == 0.
//
// STATUS (round 2, measured on MI355X, tools/gemm_persist_bench.py): CORRECT (tests/test_gpu_kernels.py::test_gemm_nt_epilogues
// with persist=True, incl. tiles finished from up to 7 partial slots) but SLOWER than the one-tile-per-workgroup kernel at
// the GPT shapes (c_attn 46 vs 28 us, c_fc + GELU 61 vs 42, mlp c_proj 52 vs 38), so it is OPT-IN (a workspace argument
// selects it) and the engine does not use it.  Ablations (MFMA off / DMA off / epilogue off) showed where the time goes:
//   * the phase-machine skeleton alone -- no MFMA, no DMA, no stores -- costs 0.55 us per k-step interval (~1300 cycles of
//     scalar, branch-heavy control per barrier on both roles; an s_barrier loop by itself is 50 cycles), and MFMA, DMA and
//     epilogue time ADD to it instead of overlapping;
//   * the stream-K hand-over costs ~8-10 us per launch (agent-scope release / acquire fences are ~2-6 us each on this part,
//     plus a 64 KB slot read), more than the imbalance it removes on 20-40 us kernels; whole-tile ranges were 6 us faster;
//   * reading the bias at the hand-over exposed one memory latency per tile on the barrier path (fixed: converted at use).
// What a faster version needs (round 3): straight-line per-tile code (fixed-trip k-loop, the previous tile's epilogue passes
// statically interleaved) instead of a general phase machine, whole-tile ranges, and no fences on the critical path.
//
// Why (round-1 profiles, MI355X): the 128 x 128 one-tile-per-workgroup kernel runs its K = 512 main loop at ~800 TF/s, but
// a tile is only 8 k-steps long -- load-latency prologue and the LDS-staged store epilogue cost as much as the main loop
// (c_fc + GELU: 23.6 us without the epilogue, 43.8 with it), co-resident workgroups run in lock-step instead of covering each
// other, and 292-tile launches (N = 512) fill 256 CUs 1.14 times.  This kernel removes all three:
//
//  * ONE workgroup per CU (grid = CU count), 8 waves, all 160 KB of LDS: waves 0-3 are COMPUTE waves (one per SIMD: LDS-DMA
//    issue, fragment reads, MFMA), waves 4-7 are EPILOGUE waves (bias / activation / residual / dropout / conversion and the
//    global stores).  Accumulators change hands through a 64 KB fp32 LDS stage, so the matrix cores never wait for a store
//    phase: tile i's epilogue runs beside tile i + 1's MFMAs on the same SIMDs (MFMA and VALU/VMEM pipes are separate).
//  * the operand stream is CONTINUOUS across tiles: the unit of work is one 64-deep k-step of one tile, a workgroup walks a
//    contiguous range of units, and the 3-stage LDS ring is filled two units ahead by global_load_lds_dwordx4 with counted
//    s_waitcnt vmcnt(8) + one raw s_barrier per unit -- a tile boundary costs nothing but an accumulator hand-over.
//  * units, not tiles, are split evenly over the workgroups (stream-K): every workgroup gets total/G units (+-1).  A tile cut
//    by a range boundary is finished by the workgroup that holds its FIRST k-step (it reaches it last, at the end of its
//    range); the workgroups holding the rest reach their piece first, at the start of their range, and hand their partial
//    accumulators over through a caller-owned fp32 workspace (64 KB slot + flag per workgroup; agent-scope release / acquire,
//    MI355X guide section "Workgroup dispatch, XCD placement & inter-workgroup visibility").
//
// LDS map (bytes): ring stage s in {0,1,2}: A tile [128 rows][64 k] bf16 at s * 32768, B tile at s * 32768 + 16384 (DMA image
// lane-linear, 16-byte slot of row r holds logical chunk slot ^ ((r >> 1) & 7): conflict-free ds_read_b128); accumulator stage
// [128 m][128 n] fp32 at 98304, 16-byte chunk c of row m stored at chunk c ^ (m & 7).  Total 163840 = the whole LDS.
//
// Barrier discipline: every wave executes the same sequence of s_barrier instructions (one shared control skeleton with
// role-specific bodies).  Barrier number `bc` is a plain counter all waves agree on; the epilogue of a staged tile (8 passes of
// 16 rows) is spread over the SEVEN barrier intervals after its hand-over (2 + 1 + ... + 1 passes), and the next hand-over
// waits until bc >= last hand-over + 8 -- exactly one K = 512 tile later, so the steady state has no idle barrier.
#include <algorithm>

#include "common.hpp"

namespace ttts {

constexpr int PBM = 128, PBN = 128, PBK = 64;
constexpr int P_STAGE_BYTES = 32768;                  // one ring stage: A 16 KB + B 16 KB
constexpr int P_ACC_OFF = 3 * P_STAGE_BYTES;          // fp32 accumulator stage
constexpr int P_LDS_BYTES = P_ACC_OFF + PBM * PBN * 4;  // 163840
constexpr int P_SLOT_FLOATS = PBM * PBN;              // one partial-accumulator slot of the workspace
constexpr int P_FLAG_BYTES = 4096;                    // flags (one int per workgroup) at the head of the workspace
constexpr int P_SPIN_LIMIT = 1 << 22;                 // bounded flag wait (~1 s): never hang the GPU on a protocol error

struct GemmPersistParams {
  const bf16* A; int64_t lda;
  const bf16* B; int64_t ldb;
  void* C; int64_t ldc;
  const float* bias;
  bf16* aux;
  const float* resid_in;
  int M, N, K;
  uint32_t thr; float inv_keep; uint32_t seed_lo, seed_hi;
  const uint32_t* ctr;
  int* flags;          // workspace head: flags[v] = 1 while slot v holds an unconsumed partial
  float* slots;        // workspace + P_FLAG_BYTES: [G][128 * 128] fp32
  int tiles_n, nk, total_units;
};

__device__ __forceinline__ int p_xcd_index(int bid, int nblk) {   // contiguous virtual ranges per XCD (bijective)
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// One role's program.  Both instantiations run the SAME phase machine on the same wave-uniform scalars, so they execute the
// same sequence of s_barrier instructions; role-specific bodies are compiled in with `if constexpr` (each role keeps only
// its own registers live: accumulators + fragments for COMPUTE, epilogue inputs for the other).
template <int EPI, bool COMPUTE>
__device__ __forceinline__ void persist_program(const GemmPersistParams& p, unsigned char* p_smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x;
  const int v = p_xcd_index(blockIdx.x, G);
  const int nk = p.nk;
  const int base = p.total_units / G, rem = p.total_units % G;
  const int U0 = v * base + min(v, rem), n_units = base + (v < rem ? 1 : 0);
  if (n_units == 0) return;
  float* const stage = reinterpret_cast<float*>(p_smem + P_ACC_OFF);

  // ---- compute-wave state -------------------------------------------------------------------------------------------
  const int cw = wave & 3, wm = cw >> 1, wn = cw & 1, hh = lane >> 5;
  f32x16 acc[2][2];
  int aoff[2], boff[2], swa[2], swb[2];
  const bf16* ga[2];
  const bf16* gb[2];
  // cursors over the unit stream: (tm, tn, k) of the unit being issued / computed, advanced incrementally -- a runtime integer
  // division costs ~200 cycles on this machine and the first version did four per unit (0.5 us per unit of pure overhead)
  const int tile0 = U0 / nk, k0 = U0 - tile0 * nk;
  const int tm0 = tile0 / p.tiles_n, tn0 = tile0 - tm0 * p.tiles_n;
  int iss_u = 0, iss_tm = tm0, iss_tn = tn0, iss_k = k0;
  bool iss_new_tile = true;
  if constexpr (COMPUTE) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ra = wm * 64 + i * 32 + (lane & 31), rb = wn * 64 + i * 32 + (lane & 31);
      aoff[i] = ra * PBK; boff[i] = rb * PBK;
      swa[i] = (ra >> 1) & 7; swb[i] = (rb >> 1) & 7;
    }
  }
  // LDS-DMA of the operand stream: every unit is 32 one-KB pieces (16 of A, 16 of B; a piece = 8 rows x 128 bytes).  ALL
  // eight waves issue, four pieces each (rows 16 w .. 16 w + 15 of both tiles): a piece costs its wave ~100-190 cycles of issue
  // time, so one wave issuing a whole unit (or the four compute waves issuing 8 each, the first version of this kernel:
  // 3700 cycles per unit) starves the matrix cores.  Compute waves slot their four pieces between MFMA groups.
  auto issue_setup = [&]() {   // per-lane source pointers for the tile of unit iss_u
    if (iss_new_tile) {
      iss_new_tile = false;
      const int im0 = iss_tm * PBM, in0 = iss_tn * PBN;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = wave * 16 + i * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        ga[i] = p.A + (int64_t)min(im0 + r, p.M - 1) * p.lda + chunk * 8;
        gb[i] = p.B + (int64_t)min(in0 + r, p.N - 1) * p.ldb + chunk * 8;
      }
    }
  };
  auto issue_piece = [&](int which) {   // which: 0, 1 = this wave's A pieces, 2, 3 = its B pieces (of unit iss_u)
    const int k_i = iss_k;
    unsigned char* st = p_smem + (iss_u % 3) * P_STAGE_BYTES + wave * (16 * PBK * 2) + (which & 1) * (8 * PBK * 2);
    if (which < 2)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga[which] + k_i * PBK),
                                       (__attribute__((address_space(3))) void*)(st), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb[which - 2] + k_i * PBK),
                                       (__attribute__((address_space(3))) void*)(st + 16384), 16, 0, 0);
  };
  auto issue_advance = [&]() {
    ++iss_u;
    if (++iss_k == nk) {
      iss_k = 0; iss_new_tile = true;
      if (++iss_tn == p.tiles_n) { iss_tn = 0; ++iss_tm; }
    }
  };
  auto issue_next = [&]() {
    issue_setup();
#pragma unroll
    for (int w = 0; w < 4; ++w) issue_piece(w);
    issue_advance();
  };
  auto compute_unit = [&](int u, bool issue) {
    if (issue) issue_setup();
    const bf16* as = reinterpret_cast<const bf16*>(p_smem + (u % 3) * P_STAGE_BYTES);
    const bf16* bs = as + 8192;
    bf16x8 af[2][2], bfr[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      af[0][i] = *reinterpret_cast<const bf16x8*>(as + aoff[i] + ((hh ^ swa[i]) << 3));
      bfr[0][i] = *reinterpret_cast<const bf16x8*>(bs + boff[i] + ((hh ^ swb[i]) << 3));
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < 3) {
        const int lc = (ks + 1) * 2 + hh;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          af[(ks + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(as + aoff[i] + ((lc ^ swa[i]) << 3));
          bfr[(ks + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(bs + boff[i] + ((lc ^ swb[i]) << 3));
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[j][i] = mfma32(bfr[ks & 1][j], af[ks & 1][i], acc[j][i]);
      if (issue) issue_piece(ks);                  // one DMA piece behind every four MFMAs (its issue time hides under them)
    }
    if (issue) issue_advance();
  };
  // accumulators <-> a workspace slot: [wave][16 quads][64 lanes] float4, every wave instruction moves 1 KB contiguous
  auto slot_store = [&](float* slot_base) {
    float* slot = slot_base + (cw * 16) * 256 + lane * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          *reinterpret_cast<float4*>(slot + ((j * 2 + i) * 4 + q) * 256) =
              make_float4(acc[j][i][4 * q], acc[j][i][4 * q + 1], acc[j][i][4 * q + 2], acc[j][i][4 * q + 3]);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[j][i][4 * q + e] = 0.f;
        }
  };
  auto slot_add = [&](const float* slot_base) {
    const float* slot = slot_base + (cw * 16) * 256 + lane * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 t = *reinterpret_cast<const float4*>(slot + ((j * 2 + i) * 4 + q) * 256);
          acc[j][i][4 * q] += t.x; acc[j][i][4 * q + 1] += t.y; acc[j][i][4 * q + 2] += t.z; acc[j][i][4 * q + 3] += t.w;
        }
  };
  auto stage_store = [&]() {   // accumulators -> the fp32 LDS stage (chunk-XOR swizzle: conflict-free 16-byte writes)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row_l = wm * 64 + i * 32 + (lane & 31);
          const int chunk = (wn * 64 + j * 32 + 8 * q + 4 * hh) >> 2;
          *reinterpret_cast<float4*>(stage + row_l * PBN + ((chunk ^ (row_l & 7)) << 2)) =
              make_float4(acc[j][i][4 * q], acc[j][i][4 * q + 1], acc[j][i][4 * q + 2], acc[j][i][4 * q + 3]);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[j][i][4 * q + e] = 0.f;
        }
  };

  // ---- epilogue-wave state --------------------------------------------------------------------------------------------
  struct EpiIn { float4 r0, r1; bf16x8 h; };      // inputs of one pass: resid_in (RESID_ADD) or the pre-activation (DGELU)
  const int et = tid - 256;                       // 0..255 among the epilogue waves
  const int e_col = (et & 15) * 8;                // this thread's 8 columns inside the tile
  const int e_row = et >> 4;                      // row inside a 16-row pass
  int pend_m0 = 0, pend_n0 = 0, passes_done = 8;  // the staged tile being stored (8 passes of 16 rows)
  float bias8[8];
  EpiIn cur, nxt, nx2;
  const bool bf16_out = (EPI == TTTS_EPI_STORE_BF16 || EPI == TTTS_EPI_GELU_BF16 || EPI == TTTS_EPI_DGELU_BF16);
  const bool vec_ok = bf16_out ? ((p.ldc & 7) == 0) : ((p.ldc & 3) == 0);
  auto epi_load = [&](int ps) -> EpiIn {           // (clamped addresses: rows / columns beyond the matrix are never stored)
    EpiIn r;
    if (EPI == TTTS_EPI_RESID_ADD_F32 || EPI == TTTS_EPI_DGELU_BF16) {
      const int m = min(pend_m0 + ps * 16 + e_row, p.M - 1);
      const int64_t off = (int64_t)m * p.ldc + min(pend_n0 + e_col, (int)p.ldc - 8);
      if (EPI == TTTS_EPI_RESID_ADD_F32) {
        const float* rin = p.resid_in ? p.resid_in + off : reinterpret_cast<const float*>(p.C) + off;
        r.r0 = *reinterpret_cast<const float4*>(rin);
        r.r1 = *reinterpret_cast<const float4*>(rin + 4);
      } else {
        r.h = *reinterpret_cast<const bf16x8*>(p.aux + off);
      }
    }
    return r;
  };
  auto epi_rows = [&](int ps, const EpiIn& in) {
    const int row_l = ps * 16 + e_row;
    const int m = pend_m0 + row_l, n = pend_n0 + e_col;
    if (m >= p.M || n >= p.N) return;
    const int c0 = (e_col >> 2) ^ (row_l & 7);
    const float4 a = *reinterpret_cast<const float4*>(stage + row_l * PBN + c0 * 4);
    const float4 b = *reinterpret_cast<const float4*>(stage + row_l * PBN + (c0 ^ 1) * 4);
    float vv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int t = 0; t < 8; ++t) vv[t] += (EPI == TTTS_EPI_STORE_F32) ? bias8[t] : (float)(bf16)bias8[t];   // autocast rounds the bias to bf16
    const int64_t off = (int64_t)m * p.ldc + n;
    const bool full = vec_ok && (n + 8 <= p.N);
    if (EPI == TTTS_EPI_STORE_BF16) {
      bf16* c = reinterpret_cast<bf16*>(p.C) + off;
      if (full) {
        bf16x8 o;
#pragma unroll
        for (int t = 0; t < 8; ++t) o[t] = (bf16)vv[t];
        *reinterpret_cast<bf16x8*>(c) = o;
      } else {
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (n + t < p.N) c[t] = (bf16)vv[t];
      }
    } else if (EPI == TTTS_EPI_GELU_BF16) {
      bf16* c = reinterpret_cast<bf16*>(p.C) + off;
      bf16* ax = p.aux + off;
      bf16x8 pre, act;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        pre[t] = (bf16)vv[t];
        act[t] = (bf16)gelu_new_f((float)pre[t]);
      }
      if (full) {
        *reinterpret_cast<bf16x8*>(ax) = pre;
        *reinterpret_cast<bf16x8*>(c) = act;
      } else {
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (n + t < p.N) { ax[t] = pre[t]; c[t] = act[t]; }
      }
    } else if (EPI == TTTS_EPI_RESID_ADD_F32) {
      float* c = reinterpret_cast<float*>(p.C) + off;
      float y[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) y[t] = (float)(bf16)vv[t];
      if (p.thr) {  // resid_pdrop: element index m*N + n, 16 random bits per element (two elements per hash)
        const uint32_t lin = (uint32_t)(((int64_t)m * p.N + n) >> 1);
        const uint32_t shi = seed_mix(p.seed_hi, p.ctr);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const uint32_t r = hash32(lin + t, p.seed_lo, shi);
          y[2 * t] = (r & 0xFFFFu) >= p.thr ? y[2 * t] * p.inv_keep : 0.f;
          y[2 * t + 1] = (r >> 16) >= p.thr ? y[2 * t + 1] * p.inv_keep : 0.f;
        }
      }
      if (full) {
        *reinterpret_cast<float4*>(c) = make_float4(in.r0.x + y[0], in.r0.y + y[1], in.r0.z + y[2], in.r0.w + y[3]);
        *reinterpret_cast<float4*>(c + 4) = make_float4(in.r1.x + y[4], in.r1.y + y[5], in.r1.z + y[6], in.r1.w + y[7]);
      } else {
        const float* rin = p.resid_in ? p.resid_in + off : c;
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (n + t < p.N) c[t] = rin[t] + y[t];
      }
    } else if (EPI == TTTS_EPI_DGELU_BF16) {
      bf16* c = reinterpret_cast<bf16*>(p.C) + off;
      if (full) {
        bf16x8 o;
#pragma unroll
        for (int t = 0; t < 8; ++t) o[t] = (bf16)(vv[t] * gelu_new_grad_f((float)in.h[t]));
        *reinterpret_cast<bf16x8*>(c) = o;
      } else {
        const bf16* ax = p.aux + off;
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (n + t < p.N) c[t] = (bf16)(vv[t] * gelu_new_grad_f((float)ax[t]));
      }
    } else {  // STORE_F32
      float* c = reinterpret_cast<float*>(p.C) + off;
      if (full) {
        *reinterpret_cast<float4*>(c) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        *reinterpret_cast<float4*>(c + 4) = make_float4(vv[4], vv[5], vv[6], vv[7]);
      } else {
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (n + t < p.N) c[t] = vv[t];
      }
    }
  };

  // ---- the phase machine (identical in both roles) ---------------------------------------------------------------------
  enum { PH_UNIT, PH_FIX, PH_DRAIN, PH_TAIL, PH_DONE };
  int ph = PH_UNIT, bc = 0, last_stage_bc = -100, tail_until = 0;
  bool publish_pending = false;
  int u = 0, k = k0, k_end = 0, m0 = 0, n0 = 0, fix_c = 0, fix_cov = 0, ctm = tm0, ctn = tn0;
  bool first_part = true;
  auto setup_part = [&]() {                        // the part starting at unit u: the first one may start inside a tile
    if (!first_part) {
      k = 0;
      if (++ctn == p.tiles_n) { ctn = 0; ++ctm; }
    }
    first_part = false;
    k_end = min(nk, k + (n_units - u));
    m0 = ctm * PBM;
    n0 = ctn * PBN;
  };
  setup_part();
  bool owned = k == 0;
  issue_next();                                    // both roles: two units in flight before the first barrier
  if (n_units > 1) issue_next();

  while (ph != PH_DONE) {
    // -- before the barrier
    if (ph == PH_UNIT) {
      // this wave's pieces of unit u have landed; its four pieces of unit u + 1 -- always the LAST vector-memory operations it
      // issued (an epilogue wave issues them after its stores and loads of the interval) -- may stay in flight
      if (u + 1 < n_units) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (ph == PH_FIX) {
      if (COMPUTE && tid == 0) {                   // wait for the workgroup that holds the next piece of this tile
        int spins = 0;
        while (__hip_atomic_load(p.flags + fix_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && ++spins < P_SPIN_LIMIT)
          __builtin_amdgcn_s_sleep(8);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
    }
    __builtin_amdgcn_s_barrier();
    // -- right after the barrier: publish (compute) / one epilogue interval (epilogue)
    if constexpr (COMPUTE) {
      if (publish_pending && tid == 0) {           // every compute wave drained its slot stores before this barrier
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(p.flags + v, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      const int t = bc - last_stage_bc;            // interval number after the hand-over: passes t (t < 6), then 6 and 7
      if (t >= 0 && t < 7) {
        if (t < 5) {
          nxt = epi_load(t + 1);
          epi_rows(t, cur);
          cur = nxt;
        } else if (t == 5) {
          nxt = epi_load(6);
          nx2 = epi_load(7);
          epi_rows(5, cur);
        } else {
          epi_rows(6, nxt);
          epi_rows(7, nx2);
          passes_done = 8;
        }
      }
    }
    publish_pending = false;
    ++bc;
    // -- the interval's work and the phase transition
    bool check_handover = false;
    if (ph == PH_UNIT) {
      if constexpr (COMPUTE) compute_unit(u, iss_u < n_units);
      else if (iss_u < n_units) issue_next();       // (after this interval's epilogue pass: keeps the DMA pieces youngest)
      ++u; ++k;
      if (k == k_end) {
        if (!owned) {                              // a later piece of a tile another workgroup owns: hand the partial over
          if constexpr (COMPUTE) {
            slot_store(p.slots + (int64_t)v * P_SLOT_FLOATS);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          publish_pending = true;                  // flag goes up after the next barrier
          if (u < n_units) { setup_part(); owned = true; }
          else { ph = PH_TAIL; tail_until = bc + 1; }
        } else if (k_end < nk) {                   // the owner's piece ends early: collect the rest from the next workgroups
          ph = PH_FIX; fix_c = v + 1; fix_cov = k_end;
        } else {
          check_handover = true;
        }
      }
    } else if (ph == PH_FIX) {
      if constexpr (COMPUTE) {
        slot_add(p.slots + (int64_t)fix_c * P_SLOT_FLOATS);
        if (tid == 0) __hip_atomic_store(p.flags + fix_c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      fix_cov += min(base + (fix_c < rem ? 1 : 0), nk - fix_cov);
      ++fix_c;
      if (fix_cov >= nk) check_handover = true;
    } else if (ph == PH_DRAIN) {
      check_handover = true;
    } else {                                       // PH_TAIL
      if (bc >= tail_until) ph = PH_DONE;
    }
    if (check_handover) {
      if (bc >= last_stage_bc + 8) {               // the previous tile has left the stage (7 intervals + 1)
        if constexpr (COMPUTE) {
          stage_store();
        } else {
          pend_m0 = m0; pend_n0 = n0; passes_done = 0;
          const int n = n0 + e_col;
#pragma unroll
          for (int t = 0; t < 8; ++t) bias8[t] = (p.bias && n + t < p.N) ? p.bias[n + t] : 0.f;   // (rounded at use: no wait here)
          cur = epi_load(0);
        }
        last_stage_bc = bc;
        if (u < n_units) { setup_part(); owned = true; ph = PH_UNIT; }
        else { ph = PH_TAIL; tail_until = bc + 7; }
      } else {
        ph = PH_DRAIN;
      }
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm_nt_persist_kernel(GemmPersistParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char p_smem[];
  if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 4) persist_program<EPI, true>(p, p_smem);
  else persist_program<EPI, false>(p, p_smem);
}

}  // namespace ttts

using namespace ttts;

namespace ttts {

static int device_cu_count() {
  static const int n = [] {
    int dev = 0, cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) return 256;
    return std::min(cu, 1024);
  }();
  return n;
}

template <int EPI>
static int launch_persist(const GemmPersistParams& p, int grid, hipStream_t s) {
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_persist_kernel<EPI>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS_BYTES);
  if (attr != hipSuccess) return fail(TTTS_EHIP, "gemm_nt_persist: hipFuncSetAttribute: %s", hipGetErrorString(attr));
  gemm_nt_persist_kernel<EPI><<<grid, 512, P_LDS_BYTES, s>>>(p);
  return check_launch("gemm_nt_persist");
}

// Called by ttts_gemm_nt_bf16_ex (gemm.hip) when the caller passed a workspace.  *handled = false leaves the launch to the
// one-tile-per-workgroup kernels (ragged K, tiny problems).
int gemm_nt_persist_try(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const float* bias,
                        void* aux, int M, int N, int K, int epilogue, const float* resid_in, uint32_t thr, float inv_keep,
                        uint64_t seed, const uint32_t* dropout_counter, void* workspace, hipStream_t s, bool* handled) {
  *handled = false;
  if (!workspace || K % PBK != 0 || K < 4 * PBK) return TTTS_OK;
  const int tiles_n = (int)cdiv(N, PBN), tiles = (int)cdiv(M, PBM) * tiles_n, nk = K / PBK;
  const int64_t total = (int64_t)tiles * nk;
  if (total < 64 || total > (1 << 30)) return TTTS_OK;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(device_cu_count(), total / 8));
  GemmPersistParams p{(const bf16*)A, lda, (const bf16*)B, ldb, C, ldc, bias, (bf16*)aux, resid_in, M, N, K,
                      thr, inv_keep, (uint32_t)seed, (uint32_t)(seed >> 32), dropout_counter,
                      reinterpret_cast<int*>(workspace),
                      reinterpret_cast<float*>(static_cast<char*>(workspace) + P_FLAG_BYTES), tiles_n, nk, (int)total};
  *handled = true;
  switch (epilogue) {
    case TTTS_EPI_STORE_BF16: return launch_persist<TTTS_EPI_STORE_BF16>(p, grid, s);
    case TTTS_EPI_GELU_BF16: return launch_persist<TTTS_EPI_GELU_BF16>(p, grid, s);
    case TTTS_EPI_RESID_ADD_F32: return launch_persist<TTTS_EPI_RESID_ADD_F32>(p, grid, s);
    case TTTS_EPI_DGELU_BF16: return launch_persist<TTTS_EPI_DGELU_BF16>(p, grid, s);
    case TTTS_EPI_STORE_F32: return launch_persist<TTTS_EPI_STORE_F32>(p, grid, s);
    default: *handled = false; return TTTS_OK;
  }
}

}  // namespace ttts

extern "C" int64_t ttts_gemm_nt_workspace_bytes(void) {
  return (int64_t)P_FLAG_BYTES + (int64_t)device_cu_count() * P_SLOT_FLOATS * (int64_t)sizeof(float);
}
