// bf16 MFMA GEMMs of the GPT train step (gfx950, v_mfma_f32_32x32x16_bf16, fp32 accumulation).
//
//  * gemm_nt: C[M,N] = epi(A[M,K] . B[N,K]^T)   -- forward projections and dX (both operands K-contiguous;
//    the bf16 "shadow" weights are kept in both layouts so that no operand ever needs a transposed read).
//  * gemm_tn: C[Mo,No] += At[Kr,Mo]^T . Bt[Kr,No] -- weight gradients; the reduction runs over ROWS of both
//    operands, so fragments are fetched with the LDS transpose read (ds_read_b64_tr_b16) and the reduction is
//    split across workgroups (fp32 atomics into the gradient arena).
//
// Tiling: 128x128 block tile, 4 waves (2x2), each wave a 64x64 sub-tile = 2x2 MFMA 32x32 tiles (64 fp32
// accumulator VGPRs), K staged 64 deep through a double-buffered LDS tile; global->register->LDS staging with
// the next tile's loads in flight under the current tile's MFMAs, ONE barrier per K-tile.
// LDS row stride 144 B (NT) makes every ds_read_b128 16-lane service group hit 64 distinct banks;
// 320 B (TN) does the same for the 4-row x 64-byte footprint of a transposed read.
// The MFMA is issued "swapped" (weights as the A operand) so that every lane ends up owning 4 CONSECUTIVE
// output columns of one row per accumulator quad: epilogues use 8-byte bf16 / 16-byte fp32 accesses.
#include <algorithm>

#include "common.hpp"

namespace ttts {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NT_LDS_STRIDE = BK + 8;   // elements (144 B)
constexpr int TN_LDS_STRIDE = 128 + 32; // elements (320 B)

struct GemmNtParams {
  const bf16* A; int64_t lda;
  const bf16* B; int64_t ldb;
  void* C; int64_t ldc;
  const float* bias;
  bf16* aux;
  const float* resid_in;  // RESID_ADD: C = resid_in + dropout(bf16(acc + bias)); NULL -> in place (C += ...)
  int M, N, K;
  uint32_t thr; float inv_keep; uint32_t seed_lo, seed_hi;  // residual dropout (RESID_ADD only; thr = 0: off)
};

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmNtParams p) {
  __shared__ __attribute__((aligned(16))) bf16 As[2][BM * NT_LDS_STRIDE];
  __shared__ __attribute__((aligned(16))) bf16 Bs[2][BN * NT_LDS_STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (p.N + BN - 1) / BN;
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
      ra[i] = (gm < p.M && k < p.K) ? *reinterpret_cast<const bf16x8*>(p.A + (int64_t)gm * p.lda + k) : zero8();
      rb[i] = (gn < p.N && k < p.K) ? *reinterpret_cast<const bf16x8*>(p.B + (int64_t)gn * p.ldb + k) : zero8();
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 256, row = c >> 3, kc = (c & 7) * 8;
      *reinterpret_cast<bf16x8*>(&As[buf][row * NT_LDS_STRIDE + kc]) = ra[i];
      *reinterpret_cast<bf16x8*>(&Bs[buf][row * NT_LDS_STRIDE + kc]) = rb[i];
    }
  };

  f32x16 acc[2][2];  // [j: 32-col block of N][i: 32-row block of M]; D = Btile . Atile^T (rows = n, cols = m)
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  load_regs(0);
  store_lds(0);
  __syncthreads();
  const int frow = lane & 31, fk = (lane >> 5) * 8;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_regs(kt + 1);
    const bf16* as = &As[buf][(wm * 64 + frow) * NT_LDS_STRIDE + fk];
    const bf16* bs = &Bs[buf][(wn * 64 + frow) * NT_LDS_STRIDE + fk];
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

  // epilogue: lane owns row m = .. + (lane & 31); accumulator quad q covers columns n = .. + 8q + 4h + 0..3
  const int h = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 64 + i * 32 + (lane & 31);
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * 64 + j * 32 + 8 * q + 4 * h;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[j][i][4 * q + e];
        if (p.bias) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < p.N) v[e] += (EPI == TTTS_EPI_STORE_F32) ? p.bias[n + e] : (float)(bf16)p.bias[n + e];
        }
        const int64_t off = (int64_t)m * p.ldc + n;
        if (EPI == TTTS_EPI_STORE_BF16) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
          *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.C) + off) = o;
        } else if (EPI == TTTS_EPI_GELU_BF16) {
          bf16x4 pre, act;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            pre[e] = (bf16)v[e];
            act[e] = (bf16)gelu_new_f((float)pre[e]);
          }
          *reinterpret_cast<bf16x4*>(p.aux + off) = pre;
          *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.C) + off) = act;
        } else if (EPI == TTTS_EPI_RESID_ADD_F32) {
          float* c = reinterpret_cast<float*>(p.C) + off;
          const float* rin = p.resid_in ? p.resid_in + off : c;
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = (float)(bf16)v[e];
          if (p.thr) {  // resid_pdrop: element index m*N + n, 16 random bits per element (pair hash)
            const uint32_t lin = (uint32_t)(((int64_t)m * p.N + n) >> 1);
            const uint32_t r0 = hash32(lin, p.seed_lo, p.seed_hi), r1 = hash32(lin + 1, p.seed_lo, p.seed_hi);
            y[0] = (r0 & 0xFFFFu) >= p.thr ? y[0] * p.inv_keep : 0.f;
            y[1] = (r0 >> 16) >= p.thr ? y[1] * p.inv_keep : 0.f;
            y[2] = (r1 & 0xFFFFu) >= p.thr ? y[2] * p.inv_keep : 0.f;
            y[3] = (r1 >> 16) >= p.thr ? y[3] * p.inv_keep : 0.f;
          }
          if (n + 3 < p.N) {
            const float4 r = *reinterpret_cast<const float4*>(rin);
            *reinterpret_cast<float4*>(c) = make_float4(r.x + y[0], r.y + y[1], r.z + y[2], r.w + y[3]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < p.N) c[e] = rin[e] + y[e];
          }
        } else if (EPI == TTTS_EPI_DGELU_BF16) {
          const bf16x4 pre = *reinterpret_cast<const bf16x4*>(p.aux + off);
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16)(v[e] * gelu_new_grad_f((float)pre[e]));
          *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.C) + off) = o;
        } else {  // STORE_F32
          float* c = reinterpret_cast<float*>(p.C) + off;
          if (n + 3 < p.N) {
            *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < p.N) c[e] = v[e];
          }
        }
      }
    }
  }
}

// -------------------------------------------------------------------------------------------------------
struct GemmTnParams {
  const bf16* At; int64_t ldat;
  const bf16* Bt; int64_t ldbt;
  float* C; int64_t ldc;
  int Mo, No, Kr, k_chunk;
};

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(GemmTnParams p) {
  __shared__ __attribute__((aligned(16))) bf16 As[2][BK * TN_LDS_STRIDE];  // [k][m]
  __shared__ __attribute__((aligned(16))) bf16 Bs[2][BK * TN_LDS_STRIDE];  // [k][n]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (p.No + BN - 1) / BN;
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const int kbeg = blockIdx.y * p.k_chunk;
  const int kend = min(p.Kr, kbeg + p.k_chunk);
  const int nk = (kend - kbeg + BK - 1) / BK;
  if (nk <= 0) return;

  // staging map: chunk c = tid + i*256 (i < 4): k-row = c >> 4, 16-byte chunk along m/n = c & 15
  bf16x8 ra[4], rb[4];
  auto load_regs = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 256, kr = c >> 4, mc = (c & 15) * 8;
      const int k = kbeg + kt * BK + kr;
      const bool kv = k < kend;
      ra[i] = (kv && m0 + mc < p.Mo) ? *reinterpret_cast<const bf16x8*>(p.At + (int64_t)k * p.ldat + m0 + mc) : zero8();
      rb[i] = (kv && n0 + mc < p.No) ? *reinterpret_cast<const bf16x8*>(p.Bt + (int64_t)k * p.ldbt + n0 + mc) : zero8();
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 256, kr = c >> 4, mc = (c & 15) * 8;
      *reinterpret_cast<bf16x8*>(&As[buf][kr * TN_LDS_STRIDE + mc]) = ra[i];
      *reinterpret_cast<bf16x8*>(&Bs[buf][kr * TN_LDS_STRIDE + mc]) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  load_regs(0);
  store_lds(0);
  __syncthreads();
  // transposed-read lane map: 16-lane group g = lane >> 4 covers columns 16*(g & 1) + 0..15 of a 32-wide block and
  // k-rows 8*(g >> 1) + {0..3} (first read) / + {4..7} (second read); lane i' = lane & 15 addresses
  // [k + (i' >> 2)][col + 4*(i' & 3)] and receives column (lane & 31) of the block.
  const int g = lane >> 4, ip = lane & 15;
  const int tr_off = (8 * (g >> 1) + (ip >> 2)) * TN_LDS_STRIDE + 16 * (g & 1) + 4 * (ip & 3);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_regs(kt + 1);
    const bf16* as = &As[buf][tr_off + wm * 64];
    const bf16* bs = &Bs[buf][tr_off + wn * 64];
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
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
  const int h = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 64 + i * 32 + (lane & 31);
    if (m >= p.Mo) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * 64 + j * 32 + 8 * q + 4 * h;
        float* c = p.C + (int64_t)m * p.ldc + n;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < p.No) atomicAdd(c + e, acc[j][i][4 * q + e]);
      }
  }
}

}  // namespace ttts

using namespace ttts;

extern "C" int ttts_gemm_nt_bf16_ex(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                    const float* bias, void* aux, int32_t M, int32_t N, int32_t K, int32_t epilogue,
                                    const float* resid_in, float dropout_p, uint64_t seed, void* stream) {
  TTTS_REQUIRE(A && B && C, "gemm_nt: null pointer");
  TTTS_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_nt: bad shape M=%d N=%d K=%d", M, N, K);
  TTTS_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K, "gemm_nt: K, lda, ldb must be multiples of 8 (K=%d lda=%lld ldb=%lld)", K, (long long)lda, (long long)ldb);
  TTTS_REQUIRE(ldc % 4 == 0 && ldc >= ((N + 3) / 4) * 4, "gemm_nt: ldc must be a multiple of 4 and >= roundup4(N)");
  TTTS_REQUIRE(aligned16(A) && aligned16(B) && aligned16(C), "gemm_nt: 16-byte aligned bases required");
  TTTS_REQUIRE((epilogue != TTTS_EPI_GELU_BF16 && epilogue != TTTS_EPI_DGELU_BF16) || aux, "gemm_nt: epilogue needs aux");
  TTTS_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "gemm_nt: dropout_p out of range");
  TTTS_REQUIRE(dropout_p == 0.f || (epilogue == TTTS_EPI_RESID_ADD_F32 && N % 4 == 0), "gemm_nt: dropout only with RESID_ADD and N %% 4 == 0");
  GemmNtParams p{(const bf16*)A, lda, (const bf16*)B, ldb, C, ldc, bias, (bf16*)aux, resid_in, M, N, K,
                 dropout_threshold(dropout_p), 1.0f, (uint32_t)seed, (uint32_t)(seed >> 32)};
  if (p.thr) p.inv_keep = 65536.0f / (65536.0f - (float)p.thr);
  const int grid = (int)(cdiv(M, BM) * cdiv(N, BN));
  hipStream_t s = as_stream(stream);
  switch (epilogue) {
    case TTTS_EPI_STORE_BF16: gemm_nt_kernel<TTTS_EPI_STORE_BF16><<<grid, 256, 0, s>>>(p); break;
    case TTTS_EPI_GELU_BF16: gemm_nt_kernel<TTTS_EPI_GELU_BF16><<<grid, 256, 0, s>>>(p); break;
    case TTTS_EPI_RESID_ADD_F32: gemm_nt_kernel<TTTS_EPI_RESID_ADD_F32><<<grid, 256, 0, s>>>(p); break;
    case TTTS_EPI_DGELU_BF16: gemm_nt_kernel<TTTS_EPI_DGELU_BF16><<<grid, 256, 0, s>>>(p); break;
    case TTTS_EPI_STORE_F32: gemm_nt_kernel<TTTS_EPI_STORE_F32><<<grid, 256, 0, s>>>(p); break;
    default: return fail(TTTS_EUNSUPPORTED, "gemm_nt: unknown epilogue %d", epilogue);
  }
  return check_launch("gemm_nt");
}

extern "C" int ttts_gemm_nt_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                 const float* bias, void* aux, int32_t M, int32_t N, int32_t K, int32_t epilogue,
                                 void* stream) {
  return ttts_gemm_nt_bf16_ex(A, lda, B, ldb, C, ldc, bias, aux, M, N, K, epilogue, nullptr, 0.f, 0, stream);
}

extern "C" int ttts_gemm_tn_bf16_accum_f32(const void* At, int64_t ldat, const void* Bt, int64_t ldbt, float* C,
                                           int64_t ldc, int32_t Mo, int32_t No, int32_t Kr, void* stream) {
  TTTS_REQUIRE(At && Bt && C, "gemm_tn: null pointer");
  TTTS_REQUIRE(Mo > 0 && No > 0 && Kr > 0, "gemm_tn: bad shape");
  TTTS_REQUIRE(ldat % 8 == 0 && ldbt % 8 == 0 && ldat >= ((Mo + 7) / 8) * 8 && ldbt >= ((No + 7) / 8) * 8,
               "gemm_tn: ldat/ldbt must be multiples of 8 and cover roundup8(Mo/No)");
  TTTS_REQUIRE(aligned16(At) && aligned16(Bt), "gemm_tn: 16-byte aligned bases required");
  const int tiles = (int)(cdiv(Mo, BM) * cdiv(No, BN));
  // split the reduction so that ~3 workgroups per CU are in flight (256 CUs)
  int splits = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(Kr, 2 * BK), cdiv(768, tiles)));
  const int k_chunk = (int)(cdiv(cdiv(Kr, splits), BK) * BK);
  splits = (int)cdiv(Kr, k_chunk);
  GemmTnParams p{(const bf16*)At, ldat, (const bf16*)Bt, ldbt, C, ldc, Mo, No, Kr, k_chunk};
  gemm_tn_kernel<<<dim3(tiles, splits), 256, 0, as_stream(stream)>>>(p);
  return check_launch("gemm_tn");
}
