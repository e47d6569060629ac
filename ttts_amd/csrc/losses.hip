// Loss reductions of the VQ-VAE-GAN step (ttts/vqvae/losses.py:7-61 and F.l1_loss in ttts/vqvae/train.py:389):
// HBM-bound streaming reductions with deterministic two-stage sums (block partials -> one finishing block), the value
// written to a device scalar so the trainer never synchronises (the reference calls .item() 12x per step).
#include <algorithm>

#include "common.hpp"

namespace ttts {

constexpr int RED_BLOCKS = 1024;

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__device__ __forceinline__ float red_term(int mode, float a, float b) {
  switch (mode) {
    case TTTS_RED_ABSDIFF: return fabsf(a - b);
    case TTTS_RED_SQ_ONE_MINUS: { const float t = 1.f - a; return t * t; }
    default: return a * a;
  }
}

__global__ __launch_bounds__(256) void reduce_loss_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                  int64_t n, int mode, float* __restrict__ partial) {
  __shared__ float sh[4];
  float s = 0.f;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 va = reinterpret_cast<const float4*>(a)[i];
    float4 vb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mode == TTTS_RED_ABSDIFF) vb = reinterpret_cast<const float4*>(b)[i];
    s += (red_term(mode, va.x, vb.x) + red_term(mode, va.y, vb.y)) + (red_term(mode, va.z, vb.z) + red_term(mode, va.w, vb.w));
  }
  if (blockIdx.x == 0) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    if (i < n) s += red_term(mode, a[i], mode == TTTS_RED_ABSDIFF ? b[i] : 0.f);
  }
  s = block_sum_256(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void reduce_loss_finish_kernel(const float* __restrict__ partial, int nblocks, float scale,
                                                                 float* __restrict__ out, int accumulate) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[i];
  s = block_sum_256(s, sh);
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + s * scale;
}

// gradient of scale * sum term(a, b):  ABSDIFF -> w.r.t. b (the generated branch; the real branch is detached in
// feature_loss / the target in l1_loss), SQ_ONE_MINUS / SQ -> w.r.t. a
__global__ __launch_bounds__(256) void reduce_loss_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              int64_t n, int mode, float scale,
                                                              const float* __restrict__ gout, float* __restrict__ d,
                                                              int accumulate) {
  const float g = scale * (gout ? gout[0] : 1.f);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float v;
    if (mode == TTTS_RED_ABSDIFF) {
      const float t = a[i] - b[i];
      v = t > 0.f ? -g : (t < 0.f ? g : 0.f);
    } else if (mode == TTTS_RED_SQ_ONE_MINUS) {
      v = -2.f * g * (1.f - a[i]);
    } else {
      v = 2.f * g * a[i];
    }
    d[i] = accumulate ? d[i] + v : v;
  }
}

// kl = sum((logs_p - logs_q - 0.5 + 0.5 (z_p - m_p)^2 exp(-2 logs_p)) * mask) / sum(mask)   (losses.py:45-61)
// tensors [B, C, T], mask [B, 1, T];  partial[2*blk] = kl sum, partial[2*blk+1] = mask sum (counted once per (b,t))
__global__ __launch_bounds__(256) void kl_partial_kernel(const float* __restrict__ z_p, const float* __restrict__ logs_q,
                                                         const float* __restrict__ m_p, const float* __restrict__ logs_p,
                                                         const float* __restrict__ mask, int B, int C, int T,
                                                         float* __restrict__ partial) {
  __shared__ float sh[4];
  const int64_t n = (int64_t)B * C * T;
  float s = 0.f, ms = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int t = (int)(i % T);
    const int64_t bc = i / T;
    const int c = (int)(bc % C), b = (int)(bc / C);
    const float mk = mask[(int64_t)b * T + t];
    const float lp = logs_p[i], df = z_p[i] - m_p[i];
    float kl = lp - logs_q[i] - 0.5f;
    kl += 0.5f * (df * df) * expf(-2.f * lp);
    s += kl * mk;
    if (c == 0) ms += mk;
  }
  s = block_sum_256(s, sh);
  ms = block_sum_256(ms, sh);
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = s; partial[2 * blockIdx.x + 1] = ms; }
}
__global__ __launch_bounds__(256) void kl_finish_kernel(const float* __restrict__ partial, int nblocks, float* __restrict__ out) {
  __shared__ float sh[4];
  float s = 0.f, ms = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) { s += partial[2 * i]; ms += partial[2 * i + 1]; }
  s = block_sum_256(s, sh);
  ms = block_sum_256(ms, sh);
  if (threadIdx.x == 0) { out[0] = s / ms; out[1] = ms; }
}
__global__ __launch_bounds__(256) void kl_bwd_kernel(const float* __restrict__ z_p, const float* __restrict__ logs_q,
                                                     const float* __restrict__ m_p, const float* __restrict__ logs_p,
                                                     const float* __restrict__ mask, const float* __restrict__ out,
                                                     const float* __restrict__ gout, int B, int C, int T,
                                                     float* __restrict__ dz_p, float* __restrict__ dlogs_q,
                                                     float* __restrict__ dm_p, float* __restrict__ dlogs_p) {
  const int64_t n = (int64_t)B * C * T;
  const float g = (gout ? gout[0] : 1.f) / out[1];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int t = (int)(i % T);
    const int b = (int)(i / T / C);
    const float gm = g * mask[(int64_t)b * T + t];
    const float lp = logs_p[i], df = z_p[i] - m_p[i], e = expf(-2.f * lp);
    if (dz_p) dz_p[i] = gm * df * e;
    if (dm_p) dm_p[i] = -gm * df * e;
    if (dlogs_q) dlogs_q[i] = -gm;
    if (dlogs_p) dlogs_p[i] = gm * (1.f - df * df * e);
  }
}

}  // namespace ttts

using namespace ttts;

extern "C" int64_t ttts_loss_workspace_bytes(void) { return (int64_t)RED_BLOCKS * 2 * sizeof(float); }

extern "C" int ttts_reduce_loss_f32(const float* a, const float* b, int64_t n, int32_t mode, float scale, float* out,
                                    int32_t accumulate, void* workspace, void* stream) {
  TTTS_REQUIRE(a && out && workspace && n > 0, "reduce_loss: bad arguments");
  TTTS_REQUIRE(mode >= 0 && mode <= 2, "reduce_loss: unknown mode %d", mode);
  TTTS_REQUIRE(mode != TTTS_RED_ABSDIFF || b, "reduce_loss: ABSDIFF needs two inputs");
  TTTS_REQUIRE(aligned16(a) && (!b || aligned16(b)), "reduce_loss: inputs must be 16-byte aligned");
  const int blocks = (int)std::min<int64_t>(RED_BLOCKS, std::max<int64_t>(1, cdiv(n / 4, 256)));
  reduce_loss_partial_kernel<<<blocks, 256, 0, as_stream(stream)>>>(a, b, n, mode, static_cast<float*>(workspace));
  reduce_loss_finish_kernel<<<1, 256, 0, as_stream(stream)>>>(static_cast<const float*>(workspace), blocks, scale, out, accumulate);
  return check_launch("reduce_loss");
}

extern "C" int ttts_reduce_loss_bwd_f32(const float* a, const float* b, int64_t n, int32_t mode, float scale,
                                        const float* gout, float* d, int32_t accumulate, void* stream) {
  TTTS_REQUIRE(a && d && n > 0, "reduce_loss_bwd: bad arguments");
  TTTS_REQUIRE(mode >= 0 && mode <= 2, "reduce_loss_bwd: unknown mode %d", mode);
  TTTS_REQUIRE(mode != TTTS_RED_ABSDIFF || b, "reduce_loss_bwd: ABSDIFF needs two inputs");
  reduce_loss_bwd_kernel<<<(int)std::min<int64_t>(cdiv(n, 256), 4096), 256, 0, as_stream(stream)>>>(a, b, n, mode, scale, gout, d, accumulate);
  return check_launch("reduce_loss_bwd");
}

extern "C" int ttts_kl_loss_fwd_f32(const float* z_p, const float* logs_q, const float* m_p, const float* logs_p,
                                    const float* mask, int32_t B, int32_t C, int32_t T, float* out, void* workspace,
                                    void* stream) {
  TTTS_REQUIRE(z_p && logs_q && m_p && logs_p && mask && out && workspace && B > 0 && C > 0 && T > 0, "kl_loss_fwd: bad arguments");
  const int64_t n = (int64_t)B * C * T;
  const int blocks = (int)std::min<int64_t>(RED_BLOCKS, cdiv(n, 256));
  kl_partial_kernel<<<blocks, 256, 0, as_stream(stream)>>>(z_p, logs_q, m_p, logs_p, mask, B, C, T, static_cast<float*>(workspace));
  kl_finish_kernel<<<1, 256, 0, as_stream(stream)>>>(static_cast<const float*>(workspace), blocks, out);
  return check_launch("kl_loss_fwd");
}

extern "C" int ttts_kl_loss_bwd_f32(const float* z_p, const float* logs_q, const float* m_p, const float* logs_p,
                                    const float* mask, const float* out, const float* gout, int32_t B, int32_t C,
                                    int32_t T, float* dz_p, float* dlogs_q, float* dm_p, float* dlogs_p, void* stream) {
  TTTS_REQUIRE(z_p && logs_q && m_p && logs_p && mask && out && B > 0 && C > 0 && T > 0, "kl_loss_bwd: bad arguments");
  const int64_t n = (int64_t)B * C * T;
  kl_bwd_kernel<<<(int)std::min<int64_t>(cdiv(n, 256), 4096), 256, 0, as_stream(stream)>>>(z_p, logs_q, m_p, logs_p, mask, out, gout, B, C, T,
                                                                                         dz_p, dlogs_q, dm_p, dlogs_p);
  return check_launch("kl_loss_bwd");
}
