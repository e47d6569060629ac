// Parametric-equaliser waveform augmentation of the VQ-VAE step (ttts/vqvae/augment/__init__.py:37-97, peq.py:19-116):
//   stft(center=True, hann) -> multiply every frame by a per-clip product of biquad responses -> istft -> clamp(-1, 1)
//   -> divide by the clip's peak.
// HBM-bound (about 40 bytes per sample), so the two FFTs of a frame never leave LDS:
//   peq_response : H[b][k] = prod_f fir_f(k) / iir_f(k), the closed form of rfft([c0, c1, c2], n_fft) (peq.py:19-30)
//   peq_frames   : one workgroup per 4 frames, transformed together; per frame: reflect-padded windowed samples packed as an n_fft/2-point
//                  complex sequence -> radix-2 Stockham FFT (as in stft.hip) -> one-sided spectrum * H -> re-packed ->
//                  inverse FFT through the conjugate trick -> * window -> frames[b][t][n]
//   istft_ola    : gather form of overlap-add (deterministic, no atomics on samples): out[i] = sum_t frame_t[i + pad - t hop]
//                  / sum_t window^2, trimmed to hop * (frames - 1) samples, clamped; per-clip max |.| through an integer
//                  atomicMax on the float bits (order independent; NaN bit patterns dominate, as torch's amax propagates NaN)
//   peak_scale   : out /= max(peak, eps)
#include <math.h>

#include <algorithm>

#include "common.hpp"

namespace ttts {

constexpr int PEQ_FR = 4;  // frames per workgroup

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// kind 0: peaking equaliser at `freq` (peq.py:97-116); 1: low shelf, 2: high shelf with cutoff `freq` (peq.py:32-95).
// Evaluated in double: near DC the 3-tap sums cancel to ~1e-4 of their terms (60 Hz corner at 32 kHz), where the
// reference's fp32 rfft keeps only 2-3 digits.  Coefficients once per (clip, filter) in LDS, then one thread per bin.
constexpr int PEQ_MAX_FILTERS = 64;
__global__ __launch_bounds__(256) void peq_response_kernel(const float* __restrict__ freq, const float* __restrict__ gain,
                                                           const float* __restrict__ q, const int* __restrict__ kind,
                                                           float2* __restrict__ H, int NF, int n_fft, double sr) {
  __shared__ double co[PEQ_MAX_FILTERS][6];   // b0 b1 b2 a0 a1 a2 of this clip's filters (they do not depend on the bin)
  const int bins = n_fft / 2 + 1;
  const int b = blockIdx.y;
  const int f = threadIdx.x;
  if (f < NF) {
    const double qq = (double)q[b * NF + f];
    const double A = exp((double)gain[b * NF + f] / 40.0 * 2.302585092994046);
    const double w0 = 2.0 * M_PI * (double)freq[b * NF + f] / sr;
    const double alpha = sin(w0) / 2.0 / qq, c = cos(w0), sA = sqrt(A);
    double b0, b1, b2, a0, a1, a2;
    if (kind[f] == 0) {
      b0 = 1.0 + alpha * A; b1 = -2.0 * c; b2 = 1.0 - alpha * A;
      a0 = 1.0 + alpha / A; a1 = -2.0 * c; a2 = 1.0 - alpha / A;
    } else if (kind[f] == 1) {
      b0 = A * ((A + 1.0) - (A - 1.0) * c + 2.0 * sA * alpha);
      b1 = 2.0 * A * ((A - 1.0) - (A + 1.0) * c);
      b2 = A * ((A + 1.0) - (A - 1.0) * c - 2.0 * sA * alpha);
      a0 = (A + 1.0) + (A - 1.0) * c + 2.0 * sA * alpha;
      a1 = -2.0 * ((A - 1.0) + (A + 1.0) * c);
      a2 = (A + 1.0) + (A - 1.0) * c - 2.0 * sA * alpha;
    } else {
      b0 = A * ((A + 1.0) + (A - 1.0) * c + 2.0 * sA * alpha);
      b1 = -2.0 * A * ((A - 1.0) + (A + 1.0) * c);
      b2 = A * ((A + 1.0) + (A - 1.0) * c - 2.0 * sA * alpha);
      a0 = (A + 1.0) - (A - 1.0) * c + 2.0 * sA * alpha;
      a1 = 2.0 * ((A - 1.0) - (A + 1.0) * c);
      a2 = (A + 1.0) - (A - 1.0) * c - 2.0 * sA * alpha;
    }
    co[f][0] = b0; co[f][1] = b1; co[f][2] = b2; co[f][3] = a0; co[f][4] = a1; co[f][5] = a2;
  }
  __syncthreads();
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= bins) return;
  double s1, c1, s2, c2;   // z1 = exp(-2 pi i k / n_fft), z2 = z1^2
  sincospi(-2.0 * (double)k / (double)n_fft, &s1, &c1);
  sincospi(-4.0 * (double)k / (double)n_fft, &s2, &c2);
  double ar = 1.0, ai = 0.0;
  for (int g = 0; g < NF; ++g) {
    const double fr = co[g][0] + co[g][1] * c1 + co[g][2] * c2, fi = co[g][1] * s1 + co[g][2] * s2;
    const double ir = co[g][3] + co[g][4] * c1 + co[g][5] * c2, ii = co[g][4] * s1 + co[g][5] * s2;
    const double d = 1.0 / (ir * ir + ii * ii);
    const double hr = (fr * ir + fi * ii) * d, hi = (fi * ir - fr * ii) * d;
    const double nr = ar * hr - ai * hi, ni = ar * hi + ai * hr;
    ar = nr; ai = ni;
  }
  H[(int64_t)b * bins + k] = make_float2((float)ar, (float)ai);
}

// in-LDS radix-2 Stockham FFT of length L (forward, exp(-i ...)) of NFR independent sequences at once (sequence f lives at
// src + f * 2L / dst + f * 2L): one barrier per pass for the whole batch; returns the buffer that holds the results
template <int NFR>
__device__ __forceinline__ float2* fft_stockham(float2* src, float2* dst, int L, int log2L, const float2* tw, int n_fft,
                                                int tid) {
  for (int ps = 0; ps < log2L; ++ps) {
    const int Ns = 1 << ps;
    for (int j = tid; j < (L >> 1); j += 256) {
      const int k = j & (Ns - 1);
      const float2 t = tw[k * (n_fft >> (ps + 1))];
      const int j0 = (j << 1) - k;
#pragma unroll
      for (int f = 0; f < NFR; ++f) {
        const float2 a = src[f * 2 * L + j];
        const float2 bt = cmul(src[f * 2 * L + j + (L >> 1)], t);
        dst[f * 2 * L + j0] = make_float2(a.x + bt.x, a.y + bt.y);
        dst[f * 2 * L + j0 + Ns] = make_float2(a.x - bt.x, a.y - bt.y);
      }
    }
    __syncthreads();
    float2* tmp = src; src = dst; dst = tmp;
  }
  return src;
}

// LDS: PEQ_FR frames x (two ping-pong buffers of L complex) = 64 KB at n_fft 2048; all frames of the workgroup move through
// every phase together (frames past the end of the clip are computed on zeros and not stored)
__global__ __launch_bounds__(256) void peq_frames_kernel(const float* __restrict__ wav, const float* __restrict__ window,
                                                         const float2* __restrict__ tw, const float2* __restrict__ H,
                                                         float* __restrict__ frames_out, int T, int n_fft, int hop,
                                                         int frames, int log2L) {
  extern __shared__ __attribute__((aligned(16))) float peq_smem[];
  const int L = n_fft >> 1;
  float2* buf0 = reinterpret_cast<float2*>(peq_smem);   // frame f: buf0 + f * 2L (ping), buf0 + f * 2L + L (pong)
  float2* buf1 = buf0 + L;
  float2* tws = buf0 + PEQ_FR * 2 * L;                   // the twiddle table, read 20 x per frame: keep it in LDS
  const int tid = threadIdx.x;
  const int fblocks = (frames + PEQ_FR - 1) / PEQ_FR;
  const int b = blockIdx.x / fblocks;
  const int f0 = (blockIdx.x % fblocks) * PEQ_FR;
  const int pad = n_fft / 2;  // torch.stft(center=True)
  for (int n = tid; n < L; n += 256) tws[n] = tw[n];
  const float* w = wav + (int64_t)b * T;
  const float2* Hb = H ? H + (int64_t)b * (L + 1) : nullptr;
  const float invL = 1.0f / (float)L;

  for (int n = tid; n < L; n += 256) {
    const float2 wv = *reinterpret_cast<const float2*>(window + 2 * n);
#pragma unroll
    for (int ff = 0; ff < PEQ_FR; ++ff) {
      float v[2] = {0.f, 0.f};
      if (f0 + ff < frames) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          int i = (f0 + ff) * hop + 2 * n + e - pad;
          if (i < 0) i = -i;
          if (i >= T) i = 2 * (T - 1) - i;
          v[e] = w[i];
        }
      }
      buf0[ff * 2 * L + n] = make_float2(v[0] * wv.x, v[1] * wv.y);
    }
  }
  __syncthreads();
  float2* Z = fft_stockham<PEQ_FR>(buf0, buf1, L, log2L, tws, n_fft, tid);
  float2* W = (Z == buf0) ? buf1 : buf0;
  // one-sided spectrum X[k], X[L-k] from Z[k], Z[L-k]; Y = X H; re-pack conj(Z') for the inverse transform:
  //   E' = (Y[k] + conj Y[L-k]) / 2, O' = (Y[k] - conj Y[L-k]) / 2 * conj(w_k), Z' = E' + i O'
  for (int k = tid; k < L; k += 256) {
    const float2 t = tws[k];
    float2 hk = make_float2(1.f, 0.f), hc = make_float2(1.f, 0.f);
    if (Hb) { hk = Hb[k]; hc = Hb[L - k]; }
#pragma unroll
    for (int ff = 0; ff < PEQ_FR; ++ff) {
      const float2* Zf = Z + ff * 2 * L;
      float2 yk, yc;
      if (k == 0) {
        const float2 z0 = Zf[0];
        // irfft ignores the imaginary parts of the DC and Nyquist bins
        yk = make_float2((z0.x + z0.y) * hk.x, 0.f);
        yc = make_float2((z0.x - z0.y) * hc.x, 0.f);
      } else {
        const float2 zk = Zf[k], zc = Zf[L - k];
        const float er = 0.5f * (zk.x + zc.x), ei = 0.5f * (zk.y - zc.y);
        const float orr = 0.5f * (zk.y + zc.y), oi = -0.5f * (zk.x - zc.x);
        const float2 ot = cmul(make_float2(orr, oi), t);
        yk = cmul(make_float2(er + ot.x, ei + ot.y), hk);     // X[k] H[k]
        // X[L-k] = conj(E[k]) + w_{L-k} conj(O[k]),  w_{L-k} = -conj(w_k)
        const float2 oc = cmul(make_float2(orr, -oi), make_float2(-t.x, t.y));
        const float2 xc = cmul(make_float2(er + oc.x, -ei + oc.y), hc);
        yc = make_float2(xc.x, -xc.y);                         // conj Y[L-k]
      }
      const float2 e2 = make_float2(0.5f * (yk.x + yc.x), 0.5f * (yk.y + yc.y));
      float2 o2 = make_float2(0.5f * (yk.x - yc.x), 0.5f * (yk.y - yc.y));
      if (k != 0) o2 = cmul(o2, make_float2(t.x, -t.y));
      W[ff * 2 * L + k] = make_float2(e2.x - o2.y, -(e2.y + o2.x));   // conj(E' + i O')
    }
  }
  __syncthreads();
  float2* other = (W == buf0) ? buf1 : buf0;
  float2* R = fft_stockham<PEQ_FR>(W, other, L, log2L, tws, n_fft, tid);
  for (int n = tid; n < L; n += 256) {
    const float2 wv = *reinterpret_cast<const float2*>(window + 2 * n);
#pragma unroll
    for (int ff = 0; ff < PEQ_FR; ++ff) {
      if (f0 + ff >= frames) break;
      const float2 r = R[ff * 2 * L + n];
      float* fo = frames_out + ((int64_t)b * frames + f0 + ff) * n_fft;
      *reinterpret_cast<float2*>(fo + 2 * n) = make_float2(r.x * invL * wv.x, -r.y * invL * wv.y);
    }
  }
}

constexpr int OLA_PER_THREAD = 16;   // one peak atomic per 4096 samples: 20 k same-line atomics per launch serialised in L2
__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ fr, const float* __restrict__ window,
                                                        float* __restrict__ out, uint32_t* __restrict__ peak_bits, int frames,
                                                        int n_fft, int hop, int Tout, int do_clamp) {
  __shared__ uint32_t sh[4];
  const int b = blockIdx.y;
  uint32_t bits = 0;
#pragma unroll 4
  for (int it = 0; it < OLA_PER_THREAD; ++it) {
    const int i = (blockIdx.x * OLA_PER_THREAD + it) * 256 + threadIdx.x;
    if (i >= Tout) break;
    const int p = i + n_fft / 2;
    int t_hi = p / hop;
    if (t_hi > frames - 1) t_hi = frames - 1;
    int t_lo = p - n_fft + 1;
    t_lo = t_lo <= 0 ? 0 : (t_lo + hop - 1) / hop;
    float v = 0.f, env = 0.f;
    for (int t = t_lo; t <= t_hi; ++t) {
      const int n = p - t * hop;
      const float wv = window[n];
      v += fr[((int64_t)b * frames + t) * n_fft + n];
      env += wv * wv;
    }
    v = v / env;
    if (do_clamp) v = v != v ? v : fminf(fmaxf(v, -1.f), 1.f);
    out[(int64_t)b * Tout + i] = v;
    bits = max(bits, __float_as_uint(v) & 0x7FFFFFFFu);
  }
  for (int o = 32; o > 0; o >>= 1) bits = max(bits, (uint32_t)__shfl_xor((int)bits, o, 64));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = bits;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(peak_bits + b, max(max(sh[0], sh[1]), max(sh[2], sh[3])));
}

__global__ __launch_bounds__(256) void peak_scale_kernel(float* __restrict__ x, const uint32_t* __restrict__ peak_bits, int Tout,
                                                         float eps) {
  const int b = blockIdx.y;
  const float pk = __uint_as_float(peak_bits[b]);
  const float d = pk != pk ? pk : fmaxf(pk, eps);   // clamp_min keeps NaN
  for (int i = blockIdx.x * 256 + threadIdx.x; i < Tout; i += gridDim.x * 256) x[(int64_t)b * Tout + i] /= d;
}

}  // namespace ttts

using namespace ttts;

static int peq_log2(int L) {
  int l = 0;
  while ((1 << l) < L) ++l;
  return l;
}

extern "C" int ttts_peq_response_f32(const float* freq, const float* gain, const float* q, const int32_t* kind, float* H,
                                     int32_t B, int32_t n_filters, int32_t n_fft, float sample_rate, void* stream) {
  TTTS_REQUIRE(freq && gain && q && kind && H, "peq_response: null pointer");
  TTTS_REQUIRE(B > 0 && n_filters > 0 && sample_rate > 0.f, "peq_response: bad sizes");
  TTTS_REQUIRE(n_fft >= 64 && n_fft <= 4096 && (n_fft & (n_fft - 1)) == 0, "peq_response: n_fft must be a power of two in [64, 4096]");
  TTTS_REQUIRE(n_filters <= PEQ_MAX_FILTERS, "peq_response: at most %d filters", PEQ_MAX_FILTERS);
  peq_response_kernel<<<dim3((unsigned)cdiv(n_fft / 2 + 1, 256), (unsigned)B), 256, 0, as_stream(stream)>>>(
      freq, gain, q, kind, reinterpret_cast<float2*>(H), n_filters, n_fft, (double)sample_rate);
  return check_launch("peq_response");
}

extern "C" int32_t ttts_stft_center_frames(int32_t T, int32_t hop) { return hop > 0 && T >= 0 ? 1 + T / hop : -1; }

extern "C" int ttts_stft_filter_frames_f32(const float* wav, const float* window, const float* twiddle, const float* H,
                                           float* frames_out, int32_t B, int32_t T, int32_t n_fft, int32_t hop, void* stream) {
  TTTS_REQUIRE(wav && window && twiddle && frames_out, "stft_filter_frames: null pointer");
  TTTS_REQUIRE(n_fft >= 64 && n_fft <= 4096 && (n_fft & (n_fft - 1)) == 0, "stft_filter_frames: n_fft must be a power of two in [64, 4096]");
  TTTS_REQUIRE(hop > 0 && hop <= n_fft && B > 0, "stft_filter_frames: bad hop / batch");
  TTTS_REQUIRE(T > n_fft / 2, "stft_filter_frames: reflect padding needs T > n_fft/2");
  const int frames = 1 + T / hop;
  const int L = n_fft / 2;
  const size_t smem = (size_t)(PEQ_FR * 2 + 1) * L * sizeof(float2);
  static OnceFlag attr_set;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(peq_frames_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return fail(TTTS_EHIP, "stft_filter_frames: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  peq_frames_kernel<<<B * (int)cdiv(frames, PEQ_FR), 256, smem, as_stream(stream)>>>(
      wav, window, reinterpret_cast<const float2*>(twiddle), reinterpret_cast<const float2*>(H), frames_out, T, n_fft, hop,
      frames, peq_log2(L));
  return check_launch("stft_filter_frames");
}

extern "C" int ttts_istft_ola_f32(const float* frames_in, const float* window, float* out, void* peak_bits, int32_t B,
                                  int32_t frames, int32_t n_fft, int32_t hop, int32_t clamp, void* stream) {
  TTTS_REQUIRE(frames_in && window && out && peak_bits, "istft_ola: null pointer");
  TTTS_REQUIRE(B > 0 && frames > 1 && hop > 0 && hop <= n_fft, "istft_ola: bad sizes");
  const int Tout = hop * (frames - 1);
  hipError_t e = hipMemsetAsync(peak_bits, 0, (size_t)B * sizeof(uint32_t), as_stream(stream));
  if (e != hipSuccess) return fail(TTTS_EHIP, "istft_ola: hipMemsetAsync: %s", hipGetErrorString(e));
  istft_ola_kernel<<<dim3((unsigned)cdiv(Tout, 256 * OLA_PER_THREAD), (unsigned)B), 256, 0, as_stream(stream)>>>(
      frames_in, window, out, static_cast<uint32_t*>(peak_bits), frames, n_fft, hop, Tout, clamp);
  return check_launch("istft_ola");
}

extern "C" int ttts_peak_scale_f32(float* x, const void* peak_bits, int32_t B, int32_t T, float eps, void* stream) {
  TTTS_REQUIRE(x && peak_bits && B > 0 && T > 0, "peak_scale: bad arguments");
  peak_scale_kernel<<<dim3((unsigned)std::min<int64_t>(cdiv(T, 256), 1024), (unsigned)B), 256, 0, as_stream(stream)>>>(
      x, static_cast<const uint32_t*>(peak_bits), T, eps);
  return check_launch("peak_scale");
}
