// VQ codebook kernels (ttts/vqvae/core_vq.py:174-230, 303-322): exact-fp32 nearest code, commitment loss,
// EMA codebook update by scatter-add (the reference's 4096 x 1024 one-hot matrix is never materialised).
//
// Nearest code: distances must be IEEE fp32 in the reference's expression order
//     dist = -((|x|^2 - 2*dot) + |e|^2),  argmax, ties -> lowest index          (core_vq.py:176-181)
// so the contraction runs on the f32-input MFMA v_mfma_f32_32x32x2_f32, which is bit-for-bit a k-ordered
// fmaf chain (no reduced-precision path exists on gfx950).  The MFMA is issued with the CODEBOOK as the A
// operand so that each lane owns one x-row (column) and 16 codes per 32-code tile in its accumulator
// registers: the running arg-max is an in-register scan in increasing code order.
#include <algorithm>

#include "common.hpp"

namespace ttts {

constexpr int VQ_ROWS = 32;    // x rows per workgroup
constexpr int VQ_CODES = 128;  // codes staged per iteration (4 waves x 32)

// |e|^2 per code: sequential fmaf chain in k order (defines the reference value used by every row)
__global__ __launch_bounds__(256) void vq_code_norm_kernel(const float* __restrict__ e, float* __restrict__ e2, int K,
                                                           int D) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  float s = 0.f;
  for (int d = 0; d < D; ++d) s = fmaf(e[(int64_t)k * D + d], e[(int64_t)k * D + d], s);
  e2[k] = s;
}

__global__ __launch_bounds__(256) void vq_nearest_kernel(const float* __restrict__ x, const float* __restrict__ cb,
                                                         const float* __restrict__ e2, int64_t* __restrict__ idx,
                                                         float* __restrict__ xq, float* __restrict__ best_dist, int N,
                                                         int K, int D) {
  extern __shared__ __attribute__((aligned(16))) float vq_smem[];
  const int LD = D + 1;  // +1 float pad: lanes (rows) hit distinct banks at a fixed k
  float* xs = vq_smem;                       // [VQ_ROWS][LD]
  float* es = xs + VQ_ROWS * LD;             // [VQ_CODES][LD]
  float* red_d = es + VQ_CODES * LD;         // [4][VQ_ROWS]
  int* red_i = reinterpret_cast<int*>(red_d + 4 * VQ_ROWS);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  const int r0 = blockIdx.x * VQ_ROWS;
  for (int i = tid; i < VQ_ROWS * D; i += 256) {
    const int r = i / D, d = i % D;
    xs[r * LD + d] = (r0 + r < N) ? x[(int64_t)(r0 + r) * D + d] : 0.f;
  }
  __syncthreads();
  const float* xrow = xs + (lane & 31) * LD;
  float x2 = 0.f;
  for (int d = 0; d < D; ++d) x2 = fmaf(xrow[d], xrow[d], x2);
  float best = -INFINITY;
  int best_i = 0;
  for (int c0 = 0; c0 < K; c0 += VQ_CODES) {
    __syncthreads();
    for (int i = tid; i < VQ_CODES * D; i += 256) {
      const int c = i / D, d = i % D;
      es[c * LD + d] = (c0 + c < K) ? cb[(int64_t)(c0 + c) * D + d] : 0.f;
    }
    __syncthreads();
    const int cw = c0 + wave * 32;  // this wave's 32 codes
    if (cw < K) {
      const float* erow = es + (wave * 32 + (lane & 31)) * LD;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      for (int d = 0; d < D; d += 2) {
        // A[i = code][k = hh], B[k = hh][j = x row]: D[code][row] += e[code][d+hh] * x[row][d+hh]
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(erow[d + hh], xrow[d + hh], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int code = cw + acc_row(r, hh);
        if (code < K) {
          const float dist = -((x2 - 2.0f * acc[r]) + e2[code]);
          if (dist > best) { best = dist; best_i = code; }
        }
      }
    }
  }
  // merge the two lane halves, then the four waves (ties -> lowest index)
  {
    const float od = __shfl_xor(best, 32, 64);
    const int oi = __shfl_xor(best_i, 32, 64);
    if (od > best || (od == best && oi < best_i)) { best = od; best_i = oi; }
  }
  if (lane < 32) {
    red_d[wave * VQ_ROWS + lane] = best;
    red_i[wave * VQ_ROWS + lane] = best_i;
  }
  __syncthreads();
  if (tid < VQ_ROWS) {
    float bd = red_d[tid];
    int bi = red_i[tid];
    for (int w = 1; w < 4; ++w) {
      const float od = red_d[w * VQ_ROWS + tid];
      const int oi = red_i[w * VQ_ROWS + tid];
      if (od > bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
    }
    red_i[tid] = bi;
    if (r0 + tid < N) {
      idx[r0 + tid] = (int64_t)bi;
      if (best_dist) best_dist[r0 + tid] = bd;
    }
  }
  __syncthreads();
  if (xq) {
    for (int i = tid; i < VQ_ROWS * D; i += 256) {
      const int r = i / D, d = i % D;
      if (r0 + r < N) xq[(int64_t)(r0 + r) * D + d] = cb[(int64_t)red_i[r] * D + d];
    }
  }
}

// ---- nearest code, second generation (D % 4 == 0) -------------------------------------------------------------------------
// The first kernel above needed 233 us for the BASELINE search (4096 rows x 1024 codes x 192: 1.6 GFLOP, 7 MB) -- 24 GB/s,
// 0.3 % of the HBM roof -- because (a) only N / 32 = 128 workgroups existed for 256 CUs and (b) its LDS staging loop issued
// ONE 4-byte global load per iteration and waited for it (768 serialised L2 round trips per workgroup).  This one:
//   * splits the codebook into slices of 256 codes: grid = (N / 32) x (K / 256) = 512 workgroups, each 32 rows x 256 codes;
//   * stages with 16-byte loads, eight in flight per lane, one row per wave instruction (no index division);
//   * computes |e|^2 of its codes itself (same k-ordered fmaf chain as vq_code_norm_kernel, from LDS: no extra launch);
//   * writes (best distance, best index) per (row, slice); vq_nearest_final_kernel merges the slices in slice order (ties ->
//     lowest index, as before) and gathers the winning code rows.
// The arithmetic per (row, code) pair is unchanged: one k-ordered fmaf chain on v_mfma_f32_32x32x2_f32, then
// -((|x|^2 - 2 dot) + |e|^2) in that expression order, so indices AND distances stay bit-exact (tests/test_gpu_kernels.py).
constexpr int VQ_SLICE = 256;  // codes per workgroup: two 32-code tiles per wave (128: same 38 us -- the code loads are not what bounds it;
                               // note hipcc waits vmcnt(0) at the first use of a prefetched group, so the in-wave prefetch overlaps little)

__device__ __forceinline__ void vq_stage_rows(float* dst, int LD, const float* __restrict__ src, int row0, int nrows_valid,
                                              int rows, int D4, int wave, int lane) {
  // rows `wave, wave + 4, ...` of a [rows][D] block -> LDS with row pitch LD (odd: conflict-free column reads); lane = float4 column
  for (int r0 = wave; r0 < rows; r0 += 32) {
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = r0 + 4 * j;
      v[j] = (r < rows && r < nrows_valid && lane < D4) ? reinterpret_cast<const float4*>(src + (int64_t)(row0 + r) * (D4 * 4))[lane]
                                                        : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = r0 + 4 * j;
      if (r < rows && lane < D4) {
        float* d = dst + r * LD + lane * 4;
        d[0] = v[j].x; d[1] = v[j].y; d[2] = v[j].z; d[3] = v[j].w;
      }
    }
  }
}

// Third step (round 2): the codes never touch LDS.  A lane of the A operand needs e[code = lane & 31][d + hh] for even d -- its
// own code row, every other element -- so each lane streams ITS row with 16-byte global loads (lanes l and l + 32 fetch the same
// row; a 32-code tile is 24 KB and stays L1-resident), eight loads in flight while the previous eight feed the MFMAs, and the
// |e|^2 chain runs on the same registers.  No barrier after the one that publishes the x tile: the four waves of a workgroup
// (and the 2 workgroups per CU) overlap freely.  52 -> ~25 us at the BASELINE search.
// (Round 3 measured unguarded, clamped-index code-row loads with a branch-free full-group path: 37.7 vs 38.4 us -- inside the
// run-to-run spread, removed.)
__global__ __launch_bounds__(256) void vq_nearest_slice_kernel(const float* __restrict__ x, const float* __restrict__ cb,
                                                               float2* __restrict__ part, int N, int K, int D, int SL) {
  extern __shared__ __attribute__((aligned(16))) float vq_smem[];
  const int LD = D + 1, D4 = D >> 2;
  float* xs = vq_smem;                       // [VQ_ROWS][LD]
  float* red_d = xs + VQ_ROWS * LD;          // [4][VQ_ROWS]
  int* red_i = reinterpret_cast<int*>(red_d + 4 * VQ_ROWS);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  const int r0 = blockIdx.x * VQ_ROWS, sl = blockIdx.y, cs0 = sl * VQ_SLICE;
  vq_stage_rows(xs, LD, x, r0, N - r0, VQ_ROWS, D4, wave, lane);
  __syncthreads();
  const float* xrow = xs + (lane & 31) * LD;
  float x2 = 0.f;
  for (int d = 0; d < D; ++d) x2 = fmaf(xrow[d], xrow[d], x2);
  float best = -INFINITY;
  int best_i = 0;
  for (int t = 0; t < VQ_SLICE / 128; ++t) {   // this wave's tiles: codes cs0 + t * 128 + wave * 32 + 0..31
    const int cw = cs0 + t * 128 + wave * 32;
    if (cw >= K) break;                        // wave-uniform
    const int code_l = min(cw + (lane & 31), K - 1);                 // (clamped: rows beyond K are computed and ignored)
    const float4* erow = reinterpret_cast<const float4*>(cb + (int64_t)code_l * D);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float e2 = 0.f;                            // |e|^2 of this lane's code: the k-ordered chain every row uses
    float4 cur[8], nxt[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      cur[j] = (j < D4) ? erow[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j0 = 0; j0 < D4; j0 += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        nxt[j] = (j0 + 8 + j < D4) ? erow[j0 + 8 + j] : make_float4(0.f, 0.f, 0.f, 0.f);
      float xa[8], xb[8];                      // this group's 16 x operands up front: their LDS latency must not sit between MFMAs
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int d = min(j0 + j, D4 - 1) * 4;
        xa[j] = xrow[d + hh];
        xb[j] = xrow[d + 2 + hh];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j0 + j < D4) {                   // (D4 % 8 != 0: the last group is partial)
          const float4 v = cur[j];
          e2 = fmaf(v.x, v.x, e2); e2 = fmaf(v.y, v.y, e2); e2 = fmaf(v.z, v.z, e2); e2 = fmaf(v.w, v.w, e2);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hh ? v.y : v.x, xa[j], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hh ? v.w : v.z, xb[j], acc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) cur[j] = nxt[j];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cl = acc_row(r, hh);           // code inside the tile: its |e|^2 lives in lane cl (either half)
      const float e2c = __shfl(e2, cl, 64);
      const int code = cw + cl;
      if (code < K) {
        const float dist = -((x2 - 2.0f * acc[r]) + e2c);
        if (dist > best) { best = dist; best_i = code; }
      }
    }
  }
  {
    const float od = __shfl_xor(best, 32, 64);
    const int oi = __shfl_xor(best_i, 32, 64);
    if (od > best || (od == best && oi < best_i)) { best = od; best_i = oi; }
  }
  if (lane < 32) {
    red_d[wave * VQ_ROWS + lane] = best;
    red_i[wave * VQ_ROWS + lane] = best_i;
  }
  __syncthreads();
  if (tid < VQ_ROWS && r0 + tid < N) {
    float bd = red_d[tid];
    int bi = red_i[tid];
    for (int w = 1; w < 4; ++w) {
      const float od = red_d[w * VQ_ROWS + tid];
      const int oi = red_i[w * VQ_ROWS + tid];
      if (od > bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
    }
    part[(int64_t)(r0 + tid) * SL + sl] = make_float2(bd, __int_as_float(bi));
  }
}

// ---- nearest code, third generation (round 4; D in {64, 128, 192}) -----------------------------------------------------------
// What held the slice kernel at 38 us (27 % of the exact-f32 matrix-core rate) was not the MFMA chain but how its operands
// arrived: every lane fetched ITS code row with 16-byte global loads -- 32 rows x 16 B per wave instruction, i.e. 32 cache-line
// requests for 512 useful bytes, 6 M line requests per launch: the L2 request rate, not its bandwidth (the same thing cost the
// register-resident GEMM's weight load 8 us: csrc/gemm.hip).  Here nothing is loaded in fragment shape:
//   * a workgroup = 128 x rows (one 32-row block per wave) x 128 codes (four 32-code tiles); grid (N / 128) x (K / 128) = 256
//     workgroups at the BASELINE search, one per CU, one wave per SIMD;
//   * x blocks and code tiles (32 rows x D floats = one "tile image") come in by LDS-DMA, whole rows, 16-byte chunk c of row r
//     at chunk (c & ~15) | ((c & 15) ^ (r & 15)): the ds_read_b128 of 16 lanes = 16 rows then hits 16 distinct slots;
//   * a wave reads its x block ONCE into registers (D / 2 per lane: the B operand of every MFMA it will issue) and its |x|^2;
//   * the four code tiles are read by all four waves; per 16-byte chunk one ds_read_b128 feeds two MFMAs and the |e|^2 chain;
//     the chunks are fetched four ahead of the MFMA chain, which is then never waited for (one dependent chain IS the pipe rate:
//     tools/ubench/mfma_f32.hip);
//   * the arg-max runs in registers across the four tiles; (best, index) per (row, slice) go to the same merge kernel.
// Per (row, code) pair the arithmetic is unchanged -- one k-ordered fmaf chain on v_mfma_f32_32x32x2_f32, |e|^2 and |x|^2 as
// k-ordered fmaf chains, -((|x|^2 - 2 dot) + |e|^2) in that order -- so indices AND distances stay bit-exact.
constexpr int VQ3_ROWS = 128, VQ3_SLICE = 128;
template <int D4>   // D / 4: 16-byte chunks per row
__global__ __launch_bounds__(256, 1) void vq_nearest_tile_kernel(const float* __restrict__ x, const float* __restrict__ cb,
                                                                 float2* __restrict__ part, int N, int K, int SL) {
  constexpr int D = 4 * D4, TILE_B = 32 * D * 4, NDMA = D4 / 2;   // DMA instructions (1 KB) per tile image
  extern __shared__ __attribute__((aligned(16))) unsigned char vq3_smem[];   // region A: 4 tile images (x blocks, then code tiles 2, 3) | region B: code tiles 0, 1
  const uint32_t lds0 = lds_byte_addr(vq3_smem);
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, rl = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = blockIdx.x * VQ3_ROWS, sl = blockIdx.y, c0 = sl * VQ3_SLICE;
  // one tile image: 32 rows of `src` starting at row0 (clamped to row_max) -> LDS at `dst`; the workgroup's 4 waves share the NDMA pieces
  auto dma_tile = [&](const float* src, int row0, int row_max, uint32_t dst) {
    for (int i = wave; i < NDMA; i += 4) {
      const int L = 64 * i + lane, row = L / D4, pc = L % D4;
      const int c = (pc & ~15) | ((pc & 15) ^ (row & 15));
      lds_dma16_untracked(src + (int64_t)min(row0 + row, row_max) * D + c * 4, lds0 + dst + i * 1024);
    }
  };
  for (int b = 0; b < 4; ++b) dma_tile(x, r0 + 32 * b, N - 1, b * TILE_B);
  for (int t = 0; t < 2; ++t) dma_tile(cb, c0 + 32 * t, K - 1, (4 + t) * TILE_B);
  constexpr int PPW = NDMA / 4;                      // pieces per wave and image (NDMA is a multiple of 4 for D = 64, 128, 192)
  static_assert(NDMA % 4 == 0, "pieces are dealt evenly to the four waves");
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PPW) : "memory");   // the x blocks have landed; the two code tiles may still be in flight
  __syncthreads();
  // chunk c of row rl of a tile image
  auto chunk = [&](uint32_t img, int c) {
    return *reinterpret_cast<const float4*>(vq3_smem + img + rl * (D * 4) + (((c & ~15) | ((c & 15) ^ (rl & 15))) << 4));
  };
  float xr[2 * D4];
  float x2 = 0.f;
#pragma unroll
  for (int c = 0; c < D4; ++c) {
    const float4 v = chunk(wave * TILE_B, c);
    x2 = fmaf(v.x, v.x, x2); x2 = fmaf(v.y, v.y, x2); x2 = fmaf(v.z, v.z, x2); x2 = fmaf(v.w, v.w, x2);
    xr[2 * c] = hh ? v.y : v.x;
    xr[2 * c + 1] = hh ? v.w : v.z;
  }
  lds_dma_wait_all();                                // code tiles 0, 1 (issued before the x reads)
  __syncthreads();                                   // every wave holds its x block: region A is free; tiles 0, 1 are published
  for (int t = 2; t < 4; ++t) dma_tile(cb, c0 + 32 * t, K - 1, (t - 2) * TILE_B);
  float best = -INFINITY;
  int best_i = 0;
#pragma unroll 1
  for (int t = 0; t < 4; ++t) {
    const int cw = c0 + 32 * t;
    if (cw >= K) break;                              // (workgroup-uniform)
    if (t == 2) {                                    // tiles 2, 3 have had two tiles of MFMAs to land
      lds_dma_wait_all();
      __syncthreads();
    }
    const uint32_t img = (t < 2 ? 4 + t : t - 2) * TILE_B;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float e2 = 0.f;                                  // |e|^2 of this lane's code (both lane halves run the same chain)
    float4 pf[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) pf[c] = chunk(img, c);
#pragma unroll
    for (int c = 0; c < D4; ++c) {
      const float4 v = pf[c & 3];
      if (c + 4 < D4) pf[c & 3] = chunk(img, c + 4);
      __builtin_amdgcn_sched_barrier(0);
      e2 = fmaf(v.x, v.x, e2); e2 = fmaf(v.y, v.y, e2); e2 = fmaf(v.z, v.z, e2); e2 = fmaf(v.w, v.w, e2);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hh ? v.y : v.x, xr[2 * c], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hh ? v.w : v.z, xr[2 * c + 1], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cl = acc_row(r, hh);                 // code inside the tile: its |e|^2 lives in lane cl (either half)
      const float e2c = __shfl(e2, cl, 64);
      const int code = cw + cl;
      if (code < K) {
        const float dist = -((x2 - 2.0f * acc[r]) + e2c);
        if (dist > best) { best = dist; best_i = code; }
      }
    }
  }
  {
    const float od = __shfl_xor(best, 32, 64);
    const int oi = __shfl_xor(best_i, 32, 64);
    if (od > best || (od == best && oi < best_i)) { best = od; best_i = oi; }
  }
  const int row = r0 + wave * 32 + rl;
  if (lane < 32 && row < N) part[(int64_t)row * SL + sl] = make_float2(best, __int_as_float(best_i));
}

__global__ __launch_bounds__(256) void vq_nearest_final_kernel(const float2* __restrict__ part, const float* __restrict__ cb,
                                                               int64_t* __restrict__ idx, float* __restrict__ xq,
                                                               float* __restrict__ best_dist, int N, int D, int SL) {
  __shared__ int win[VQ_ROWS];
  const int tid = threadIdx.x, r0 = blockIdx.x * VQ_ROWS;
  if (tid < VQ_ROWS && r0 + tid < N) {
    float2 b = part[(int64_t)(r0 + tid) * SL];
    for (int s = 1; s < SL; ++s) {             // slices hold increasing code ranges: strict > keeps the lowest index on ties
      const float2 o = part[(int64_t)(r0 + tid) * SL + s];
      if (o.x > b.x) b = o;
    }
    win[tid] = __float_as_int(b.y);
    idx[r0 + tid] = (int64_t)__float_as_int(b.y);
    if (best_dist) best_dist[r0 + tid] = b.x;
  }
  __syncthreads();
  if (xq) {
    const int D4 = D >> 2;
    for (int i = tid; i < VQ_ROWS * D4; i += 256) {
      const int r = i / D4, c = i - r * D4;
      if (r0 + r < N)
        reinterpret_cast<float4*>(xq + (int64_t)(r0 + r) * D)[c] = reinterpret_cast<const float4*>(cb + (int64_t)win[r] * D)[c];
    }
  }
}

// commitment: t = x + (xq - x); loss = mean((t - x)^2); dx += gs * 2 (x - t) / n
__global__ __launch_bounds__(256) void vq_commit_kernel(const float* __restrict__ x, const float* __restrict__ xq,
                                                        float* __restrict__ dx, float gscale, int64_t n,
                                                        double* __restrict__ partial) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float xv = x[i];
    const float t = xv + (xq[i] - xv);
    const float df = t - xv;
    s += df * df;
    if (dx) dx[i] += gscale * 2.0f * (xv - t) / (float)n;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ((double)sh[0] + sh[1]) + ((double)sh[2] + sh[3]);
}
__global__ __launch_bounds__(256) void vq_commit_final_kernel(const double* __restrict__ partial, int nblk, int64_t n,
                                                              float* __restrict__ loss) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) s += partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *loss = (float)(((sh[0] + sh[1]) + (sh[2] + sh[3])) / (double)n);
}

// EMA update ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vq_ema_scatter_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx,
                                                             float* __restrict__ counts, float* __restrict__ sums,
                                                             int N, int D) {
  const int D4 = D >> 2;
  const int64_t total = (int64_t)N * D4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int n = (int)(i / D4), d4 = (int)(i % D4);
    const int64_t k = idx[n];
    const float4 v = reinterpret_cast<const float4*>(x + (int64_t)n * D)[d4];
    float* dst = sums + k * D + d4 * 4;
    atomicAdd(dst + 0, v.x);
    atomicAdd(dst + 1, v.y);
    atomicAdd(dst + 2, v.z);
    atomicAdd(dst + 3, v.w);
    if (d4 == 0) atomicAdd(counts + k, 1.0f);
  }
}
// single block: cluster_size EMA + Laplace-smoothed sizes -> smoothed[K] (workspace)
__global__ __launch_bounds__(1024) void vq_ema_sizes_kernel(float* __restrict__ cluster_size,
                                                            const float* __restrict__ counts,
                                                            float* __restrict__ smoothed, int K, float decay,
                                                            float eps) {
  __shared__ double sh[16];
  __shared__ float total_s;
  double s = 0.0;
  for (int k = threadIdx.x; k < K; k += 1024) {
    const float cs = cluster_size[k] * decay + counts[k] * (1.0f - decay);
    cluster_size[k] = cs;
    s += (double)cs;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 16; ++i) t += sh[i];
    total_s = (float)t;
  }
  __syncthreads();
  const float tot = total_s;
  for (int k = threadIdx.x; k < K; k += 1024)
    smoothed[k] = (cluster_size[k] + eps) / (tot + (float)K * eps) * tot;
}
__global__ __launch_bounds__(256) void vq_ema_embed_kernel(float* __restrict__ embed_avg, float* __restrict__ embed,
                                                           const float* __restrict__ sums,
                                                           const float* __restrict__ smoothed, int K, int D,
                                                           float decay) {
  const int64_t total = (int64_t)K * D;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i / D);
    const float a = embed_avg[i] * decay + sums[i] * (1.0f - decay);
    embed_avg[i] = a;
    embed[i] = a / smoothed[k];
  }
}

// EMA update in ONE pass without atomics on global memory (round 3; the scatter / sizes / embed kernels above stay as the path for
// D > 256 or unaligned shapes).  A workgroup owns EMA_CODES codes, one per wave.  Every workgroup builds the full histogram of
// idx in LDS (integer atomics: exact) and from it the SAME new cluster sizes and their total; a wave then walks idx 64 rows at a
// time, ballots the rows of its code and adds them in ascending row order (lane = column d, d + 64, ...): deterministic sums,
// no zero-filled workspace.  New cluster sizes go to `cs_new` and are copied over cluster_size by vq_ema_commit_kernel (other
// workgroups still read the old ones here).
constexpr int EMA_CODES = 4;                            // one code per wave
__global__ __launch_bounds__(256) void vq_ema_fused_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx,
                                                           const float* __restrict__ cluster_size, float* __restrict__ embed_avg,
                                                           float* __restrict__ embed, float* __restrict__ cs_new, int N, int K,
                                                           int D, float decay, float eps) {
  extern __shared__ int ema_hist[];                    // [K] histogram, then [N rounded up to 256] the indices as int32
  int* ema_idx = ema_hist + K;
  __shared__ double red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Np = (N + 255) & ~255;
  for (int k = tid; k < K; k += 256) ema_hist[k] = 0;
  __syncthreads();
  for (int n = tid; n < Np; n += 256) {
    const int k = n < N ? (int)idx[n] : -1;
    ema_idx[n] = k;
    if (k >= 0 && k < K) atomicAdd(&ema_hist[k], 1);
  }
  __syncthreads();
  double part = 0.0;
  for (int k = tid; k < K; k += 256) part += (double)(cluster_size[k] * decay + (float)ema_hist[k] * (1.0f - decay));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if (lane == 0) red[wave] = part;
  __syncthreads();
  const float tot = (float)((red[0] + red[1]) + (red[2] + red[3]));
  // Every wave scans a quarter of the rows (64-row chunks, interleaved) for ALL four codes of the workgroup -- a crowded code
  // (198 of 4096 rows on one code in the benchmark's random data) is then shared by the four waves instead of being one wave's
  // chain.  Rows of a code are collected in ascending order (`mine[c]`: lane i holds the i-th, up to 64) and added sixteen at a
  // time with all their loads in flight; the four partial sums per code meet in LDS and are added in wave order.
  const int k0 = blockIdx.x * EMA_CODES;
  float acc[EMA_CODES][4];
  int mine[EMA_CODES], have[EMA_CODES];
  bool live[EMA_CODES];
#pragma unroll
  for (int c = 0; c < EMA_CODES; ++c) {
    mine[c] = 0; have[c] = 0;
    live[c] = k0 + c < K && ema_hist[min(k0 + c, K - 1)] > 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[c][j] = 0.f;
  }
  auto flush = [&](int (&mine_c), int& have_c, float (&acc_c)[4]) {
    constexpr int RB = 16;
    for (int i = 0; i < have_c; i += RB) {
      float v[RB][4];
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int r = __builtin_amdgcn_readlane(mine_c, min(i + u, have_c - 1));
        const float* xr = x + (int64_t)r * D;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[u][j] = (lane + 64 * j < D) ? xr[lane + 64 * j] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < RB; ++u)
        if (i + u < have_c) {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc_c[j] += v[u][j];
        }
    }
    have_c = 0;
  };
  for (int base = wave * 64; base < Np; base += 256) {
    const int kk = ema_idx[base + lane];
#pragma unroll
    for (int c = 0; c < EMA_CODES; ++c) {
      if (!live[c]) continue;
      const uint64_t m = __builtin_amdgcn_ballot_w64(kk == k0 + c);
      if (m) {
        const int n = __builtin_popcountll(m);
        if (have[c] + n > 64) flush(mine[c], have[c], acc[c]);
        const int want = lane - have[c];                 // lane have + j takes the j-th set bit of m
        uint64_t t = m;
        int r = -1;
        for (int j = 0; j < n; ++j) {
          const int b = __builtin_ctzll(t);
          t &= t - 1;
          if (j == want) r = base + b;
        }
        if (want >= 0 && want < n) mine[c] = r;
        have[c] += n;
      }
    }
  }
  // The rows left after the scan (all of them, unless a code overflowed its 64 slots above): the four codes TOGETHER, four rows
  // of each per round with all 64 loads in flight -- with ~1 row per code and wave (N / K = 4) the per-code flushes were four
  // dependent HBM latencies in a row, most of this kernel's 21 us.  Per code the rows are still added in ascending order.
  // A crowded code (more than four rows in this wave's quarter: 198 of 4096 rows fall on one code in the benchmark's data) is
  // flushed alone first, sixteen rows a round -- four a round would be a dozen dependent latencies for it.
#pragma unroll
  for (int c = 0; c < EMA_CODES; ++c)
    if (live[c] && have[c] > 4) flush(mine[c], have[c], acc[c]);
  {
    constexpr int RB = 4;
    int most = 0;
#pragma unroll
    for (int c = 0; c < EMA_CODES; ++c) most = max(most, live[c] ? have[c] : 0);
    for (int i = 0; i < most; i += RB) {
      float v[EMA_CODES][RB][4];
#pragma unroll
      for (int c = 0; c < EMA_CODES; ++c)
#pragma unroll
        for (int u = 0; u < RB; ++u) {
          const int r = __builtin_amdgcn_readlane(mine[c], max(min(i + u, have[c] - 1), 0));
          const float* xr = x + (int64_t)((live[c] && have[c] > 0) ? r : 0) * D;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[c][u][j] = (lane + 64 * j < D) ? xr[lane + 64 * j] : 0.f;
        }
#pragma unroll
      for (int c = 0; c < EMA_CODES; ++c)
#pragma unroll
        for (int u = 0; u < RB; ++u)
          if (live[c] && i + u < have[c]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[c][j] += v[c][u][j];
          }
    }
  }
  __syncthreads();                                       // everyone is done with ema_idx: it becomes the [wave][code][256] stage
  float* stagep = reinterpret_cast<float*>(ema_idx);
#pragma unroll
  for (int c = 0; c < EMA_CODES; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) stagep[(wave * EMA_CODES + c) * 256 + lane + 64 * j] = acc[c][j];
  __syncthreads();
  const int k = k0 + wave;
  if (k >= K) return;
  const float cs = cluster_size[k] * decay + (float)ema_hist[k] * (1.0f - decay);
  const float smoothed = (cs + eps) / (tot + (float)K * eps) * tot;
  if (lane == 0) cs_new[k] = cs;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int d = lane + 64 * j;
    if (d < D) {
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) sum += stagep[(w * EMA_CODES + wave) * 256 + d];
      const int64_t i = (int64_t)k * D + d;
      const float a = embed_avg[i] * decay + sum * (1.0f - decay);
      embed_avg[i] = a;
      embed[i] = a / smoothed;
    }
  }
}
__global__ __launch_bounds__(256) void vq_ema_commit_kernel(float* __restrict__ cluster_size, const float* __restrict__ cs_new, int K) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < K) cluster_size[k] = cs_new[k];
}

}  // namespace ttts

using namespace ttts;

extern "C" int64_t ttts_vq_workspace_bytes(int32_t N, int32_t K) {
  // code norms (generic kernel) / commitment partials, then the (distance, index) pairs of the sliced search
  return cdiv(K, 4) * 16 + 1024 * (int64_t)sizeof(double) + (int64_t)N * cdiv(K, VQ3_SLICE) * (int64_t)sizeof(float2);   // (the finer of the two slicings)
}

extern "C" int ttts_vq_nearest_f32(const float* x, const float* codebook, int64_t* idx, float* xq, float* best_dist,
                                   void* workspace, int32_t N, int32_t K, int32_t D, void* stream) {
  TTTS_REQUIRE(x && codebook && idx && workspace, "vq_nearest: null pointer");
  TTTS_REQUIRE(N > 0 && K > 0 && D > 0 && D % 2 == 0 && D <= 256, "vq_nearest: need even D <= 256 (D=%d)", D);
  hipStream_t s = as_stream(stream);
  const size_t smem = ((size_t)(VQ_ROWS + VQ_CODES) * (D + 1) + 8 * VQ_ROWS) * sizeof(float);
  static OnceFlag once_generic, once_slice;
  hipError_t attr = lds_opt_in(once_generic, reinterpret_cast<const void*>(vq_nearest_kernel));
  if (attr == hipSuccess) attr = lds_opt_in(once_slice, reinterpret_cast<const void*>(vq_nearest_slice_kernel));
  if (attr != hipSuccess) return fail(TTTS_EHIP, "vq_nearest: hipFuncSetAttribute: %s", hipGetErrorString(attr));
  if ((D == 64 || D == 128 || D == 192) && aligned16(x) && aligned16(codebook) && (!xq || aligned16(xq))) {
    const int SL = (int)cdiv(K, VQ3_SLICE);
    float2* part = reinterpret_cast<float2*>(static_cast<char*>(workspace) + cdiv(K, 4) * 16 + 1024 * sizeof(double));
    const dim3 grid((unsigned)cdiv(N, VQ3_ROWS), (unsigned)SL);
    const size_t lds = (size_t)6 * 32 * D * 4;       // six tile images: 144 KB at D = 192
    hipError_t e = hipSuccess;
#define VQ3_LAUNCH(D4_)                                                                                                          \
    do {                                                                                                                          \
      static OnceFlag once_tile;                                                                                                  \
      e = lds_opt_in(once_tile, reinterpret_cast<const void*>(vq_nearest_tile_kernel<D4_>), 6 * 32 * 4 * D4_ * 4);                \
      if (e == hipSuccess) vq_nearest_tile_kernel<D4_><<<grid, 256, lds, s>>>(x, codebook, part, N, K, SL);                      \
    } while (0)
    if (D == 64) VQ3_LAUNCH(16); else if (D == 128) VQ3_LAUNCH(32); else VQ3_LAUNCH(48);   // (D = 256: six images would need 192 KB of LDS)
#undef VQ3_LAUNCH
    if (e != hipSuccess) return fail(TTTS_EHIP, "vq_nearest: hipFuncSetAttribute: %s", hipGetErrorString(e));
    int rc = check_launch("vq_nearest_tile");
    if (rc) return rc;
    vq_nearest_final_kernel<<<(int)cdiv(N, VQ_ROWS), 256, 0, s>>>(part, codebook, idx, xq, best_dist, N, D, SL);
    return check_launch("vq_nearest_final");
  }
  if (D % 4 == 0 && aligned16(x) && aligned16(codebook) && (!xq || aligned16(xq))) {
    const int SL = (int)cdiv(K, VQ_SLICE);
    float2* part = reinterpret_cast<float2*>(static_cast<char*>(workspace) + cdiv(K, 4) * 16 + 1024 * sizeof(double));
    const size_t smem_x = ((size_t)VQ_ROWS * (D + 1) + 8 * VQ_ROWS) * sizeof(float);
    vq_nearest_slice_kernel<<<dim3((unsigned)cdiv(N, VQ_ROWS), (unsigned)SL), 256, smem_x, s>>>(x, codebook, part, N, K, D, SL);
    int rc = check_launch("vq_nearest_slice");
    if (rc) return rc;
    vq_nearest_final_kernel<<<(int)cdiv(N, VQ_ROWS), 256, 0, s>>>(part, codebook, idx, xq, best_dist, N, D, SL);
    return check_launch("vq_nearest_final");
  }
  float* e2 = reinterpret_cast<float*>(workspace);   // generic path (D % 4 != 0): first-generation kernel
  vq_code_norm_kernel<<<(int)cdiv(K, 256), 256, 0, s>>>(codebook, e2, K, D);
  int rc = check_launch("vq_code_norm");
  if (rc) return rc;
  vq_nearest_kernel<<<(int)cdiv(N, VQ_ROWS), 256, smem, s>>>(x, codebook, e2, idx, xq, best_dist, N, K, D);
  return check_launch("vq_nearest");
}

extern "C" int ttts_vq_commit_f32(const float* x, const float* xq, float* loss, float* dx, float grad_scale, int32_t N,
                                  int32_t D, void* workspace, void* stream) {
  TTTS_REQUIRE(x && xq && loss && workspace && N > 0 && D > 0, "vq_commit: bad arguments");
  const int64_t n = (int64_t)N * D;
  const int nblk = (int)std::min<int64_t>(1024, cdiv(n, 256));
  double* partial = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + 0);
  hipStream_t s = as_stream(stream);
  vq_commit_kernel<<<nblk, 256, 0, s>>>(x, xq, dx, grad_scale, n, partial);
  int rc = check_launch("vq_commit");
  if (rc) return rc;
  vq_commit_final_kernel<<<1, 256, 0, s>>>(partial, nblk, n, loss);
  return check_launch("vq_commit_final");
}

extern "C" int64_t ttts_vq_ema_workspace_bytes(int32_t K, int32_t D) {
  return ((int64_t)K * D + 2 * (int64_t)K) * (int64_t)sizeof(float);
}

extern "C" int ttts_vq_ema_update_f32(const float* x, const int64_t* idx, float* cluster_size, float* embed_avg,
                                      float* embed, void* workspace, int32_t N, int32_t K, int32_t D, float decay,
                                      float epsilon, void* stream) {
  TTTS_REQUIRE(x && idx && cluster_size && embed_avg && embed && workspace, "vq_ema: null pointer");
  TTTS_REQUIRE(N > 0 && K > 0 && D > 0 && D % 4 == 0, "vq_ema: need D %% 4 == 0");
  hipStream_t s = as_stream(stream);
  // one pass, no global atomics, no zero fill (+ the cluster-size commit) when its dynamic LDS -- the histogram and the index /
  // accumulator area, sized by whichever of the two is larger -- fits the 64 KB a kernel gets without an opt-in (+ 32 B static)
  const size_t fused_lds = ((size_t)K + std::max<size_t>((size_t)N + 256, (size_t)4 * EMA_CODES * 256)) * sizeof(int);
  if (D <= 256 && fused_lds + 32 <= 64 * 1024) {
    float* cs_new = reinterpret_cast<float*>(workspace);
    vq_ema_fused_kernel<<<(int)cdiv(K, EMA_CODES), 256, fused_lds, s>>>(x, idx, cluster_size, embed_avg, embed, cs_new,
                                                                         N, K, D, decay, epsilon);
    int rc0 = check_launch("vq_ema_fused");
    if (rc0) return rc0;
    vq_ema_commit_kernel<<<(int)cdiv(K, 256), 256, 0, s>>>(cluster_size, cs_new, K);
    return check_launch("vq_ema_commit");
  }
  float* sums = reinterpret_cast<float*>(workspace);
  float* counts = sums + (int64_t)K * D;
  float* smoothed = counts + K;
  hipError_t e = hipMemsetAsync(workspace, 0, ((size_t)K * D + K) * sizeof(float), s);
  if (e != hipSuccess) return fail(TTTS_EHIP, "vq_ema: memset: %s", hipGetErrorString(e));
  vq_ema_scatter_kernel<<<(int)std::min<int64_t>(2048, cdiv((int64_t)N * (D / 4), 256)), 256, 0, s>>>(x, idx, counts, sums, N, D);
  int rc = check_launch("vq_ema_scatter");
  if (rc) return rc;
  vq_ema_sizes_kernel<<<1, 1024, 0, s>>>(cluster_size, counts, smoothed, K, decay, epsilon);
  rc = check_launch("vq_ema_sizes");
  if (rc) return rc;
  vq_ema_embed_kernel<<<(int)std::min<int64_t>(2048, cdiv((int64_t)K * D, 256)), 256, 0, s>>>(embed_avg, embed, sums, smoothed, K, D, decay);
  return check_launch("vq_ema_embed");
}
