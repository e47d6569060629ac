// Mel / STFT front-end (ttts/utils/data_utils.py:52-103): HBM-bound, so a real FFT in LDS (not a DFT-as-GEMM).
//  stft_mag: workgroup = 16 consecutive frames of one clip.  Per frame: reflect-padded, hann-windowed samples are
//  packed as an n_fft/2-point complex sequence, transformed by a radix-2 Stockham autosort FFT ping-ponging between
//  two LDS buffers (one butterfly per thread and pass for n_fft = 1024, two for 2048), unpacked to the n_fft/2+1
//  one-sided real spectrum, and sqrt(re^2 + im^2 + 1e-6) is parked in an LDS [bin][16 frames] tile so that HBM
//  writes are 64-byte runs along the frame axis of spec[b][bin][frame].
//  Twiddles come from a host-computed (double precision) table: fp32 sincos on device would cost ~1e-6 accuracy.
#include <math.h>

#include <algorithm>

#include "common.hpp"

namespace ttts {

constexpr int STFT_FR = 8;  // frames per workgroup (8: 59 KB of LDS -> two workgroups per CU cover each other's barrier stalls)

__global__ __launch_bounds__(256) void stft_mag_kernel(const float* __restrict__ wav, const float* __restrict__ window,
                                                       const float2* __restrict__ tw, float* __restrict__ spec, int T,
                                                       int n_fft, int hop, int frames, int log2L, int span_lds) {
  extern __shared__ __attribute__((aligned(16))) float stft_smem[];
  const int L = n_fft >> 1;  // complex FFT length
  float2* buf0 = reinterpret_cast<float2*>(stft_smem);
  float2* buf1 = buf0 + L;
  float* outs = reinterpret_cast<float*>(buf1 + L);  // [L + 1][STFT_FR + 1]
  // (round 2) everything a frame needs sits in LDS before the frame loop: the twiddle table (a global load per butterfly
  // and pass exposed one L2 latency in each of the log2 L barrier intervals) and -- when it fits (span_lds) -- the 16
  // frames' whole sample span, reflect-padded, loaded once with coalesced reads (consecutive frames overlap by n_fft - hop)
  float2* tws = reinterpret_cast<float2*>(outs + (L + 1) * (STFT_FR + 1) + ((L + 1) & 1));   // [L], 8-byte aligned
  float* span = reinterpret_cast<float*>(tws + L);  // [(STFT_FR - 1) * hop + n_fft] when span_lds
  const int tid = threadIdx.x;
  const int fblocks = (frames + STFT_FR - 1) / STFT_FR;
  const int b = blockIdx.x / fblocks;
  const int f0 = (blockIdx.x % fblocks) * STFT_FR;
  const int pad = (n_fft - hop) / 2;
  const float* w = wav + (int64_t)b * T;
  for (int k = tid; k < L; k += 256) tws[k] = tw[k];
  if (span_lds) {
    const int nspan = (min(STFT_FR, frames - f0) - 1) * hop + n_fft;
    for (int sidx = tid; sidx < nspan; sidx += 256) {
      int i = f0 * hop + sidx - pad;
      if (i < 0) i = -i;
      if (i >= T) i = 2 * (T - 1) - i;
      span[sidx] = w[i];
    }
  }
  __syncthreads();

  for (int ff = 0; ff < STFT_FR; ++ff) {
    const int frame = f0 + ff;
    if (frame >= frames) break;  // block-uniform
    // 1. windowed, reflect-padded frame packed as z[n] = x[2n] + i x[2n+1]
    for (int n = tid; n < L; n += 256) {
      float v[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float xv;
        if (span_lds) {
          xv = span[ff * hop + 2 * n + e];
        } else {
          int i = frame * hop + 2 * n + e - pad;
          if (i < 0) i = -i;
          if (i >= T) i = 2 * (T - 1) - i;
          xv = w[i];
        }
        v[e] = xv * window[2 * n + e];
      }
      buf0[n] = make_float2(v[0], v[1]);
    }
    __syncthreads();
    // 2. radix-2 Stockham passes: Ns = 1, 2, ..., L/2
    float2* src = buf0;
    float2* dst = buf1;
    for (int ps = 0; ps < log2L; ++ps) {
      const int Ns = 1 << ps;
      for (int j = tid; j < (L >> 1); j += 256) {
        const int k = j & (Ns - 1);
        const float2 a = src[j];
        const float2 bb = src[j + (L >> 1)];
        const float2 t = tws[k * (n_fft >> (ps + 1))];  // exp(-2 pi i k / (2 Ns))
        const float2 bt = make_float2(bb.x * t.x - bb.y * t.y, bb.x * t.y + bb.y * t.x);
        const int j0 = (j << 1) - k;
        dst[j0] = make_float2(a.x + bt.x, a.y + bt.y);
        dst[j0 + Ns] = make_float2(a.x - bt.x, a.y - bt.y);
      }
      __syncthreads();
      float2* tmp = src; src = dst; dst = tmp;
    }
    // 3. unpack the real spectrum: X[k] = E + w_k O, E = (Z[k] + conj Z[L-k])/2, O = -i (Z[k] - conj Z[L-k])/2
    for (int k = tid; k <= L; k += 256) {
      float re, im;
      if (k == 0 || k == L) {
        const float2 z0 = src[0];
        re = (k == 0) ? z0.x + z0.y : z0.x - z0.y;
        im = 0.f;
      } else {
        const float2 zk = src[k];
        const float2 zc = src[L - k];
        const float er = 0.5f * (zk.x + zc.x), ei = 0.5f * (zk.y - zc.y);
        const float orr = 0.5f * (zk.y + zc.y), oi = -0.5f * (zk.x - zc.x);
        const float2 t = tws[k];
        re = er + (orr * t.x - oi * t.y);
        im = ei + (orr * t.y + oi * t.x);
      }
      outs[k * (STFT_FR + 1) + ff] = sqrtf(re * re + im * im + 1e-6f);
    }
    __syncthreads();
  }
  const int nf = min(STFT_FR, frames - f0);
  for (int i = tid; i < (L + 1) * STFT_FR; i += 256) {
    const int k = i / STFT_FR, ff = i % STFT_FR;
    if (ff < nf) spec[((int64_t)b * (L + 1) + k) * frames + f0 + ff] = outs[k * (STFT_FR + 1) + ff];
  }
}

// (round 2, second version) The radix-2 kernel above spends its time in barriers: 12 barrier intervals per frame, frames
// one after the other (96 per workgroup), each interval a dependent LDS round trip with two butterflies per thread in it.
// Here the passes are radix-4 (5 instead of 10 for n_fft = 2048; one radix-2 pass first when log2 L is odd) and NF frames
// go through every pass together: (5 + 2) x 8 / NF barrier intervals per workgroup and NF independent butterflies per
// thread to cover the LDS latency inside an interval.
template <int NF, int FR>
__global__ __launch_bounds__(256) void stft_mag_r4_kernel(const float* __restrict__ wav, const float* __restrict__ window,
                                                          const float2* __restrict__ tw, float* __restrict__ spec, int T,
                                                          int n_fft, int hop, int frames, int log2L) {
  extern __shared__ __attribute__((aligned(16))) float stft_smem[];
  const int L = n_fft >> 1, Q = L >> 2, lq = log2L - 2;
  float2* bufA = reinterpret_cast<float2*>(stft_smem);            // [NF][L]
  float2* bufB = bufA + NF * L;
  float* outs = reinterpret_cast<float*>(bufB + NF * L);           // [L + 1][FR + 1]
  float2* tws = reinterpret_cast<float2*>(outs + (L + 1) * (FR + 1) + (((L + 1) * (FR + 1)) & 1));   // [L]
  const int tid = threadIdx.x;
  const int fblocks = (frames + FR - 1) / FR;
  const int b = blockIdx.x / fblocks;
  const int f0 = (blockIdx.x % fblocks) * FR;
  const int pad = (n_fft - hop) / 2;
  const float* w = wav + (int64_t)b * T;
  for (int k = tid; k < L; k += 256) tws[k] = tw[k];
  // this thread's packed points n = tid + 256 i of a frame group: frame n >> log2L, point n & (L - 1); up to NPT of them
  constexpr int NPT = NF * 2048 / 256;                    // L <= 2048
  const int npt = (NF * L) >> 8;
  float2 win[NPT], xr[NPT];
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const int nn = (tid + 256 * i) & (L - 1);
    win[i] = i < npt ? make_float2(window[2 * nn], window[2 * nn + 1]) : make_float2(0.f, 0.f);
  }
  // samples of group g (reflect-padded; frames beyond the clip read clamped positions and are never stored): unconditional
  // loads, issued one group ahead so that they land behind the previous group's passes
  auto load_group = [&](int ff0) {
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      if (i < npt) {
        const int n = tid + 256 * i, frame = min(f0 + ff0 + (n >> log2L), frames - 1), nn = n & (L - 1);
        int i0 = frame * hop + 2 * nn - pad, i1 = i0 + 1;
        i0 = i0 < 0 ? -i0 : i0; i1 = i1 < 0 ? -i1 : i1;
        i0 = i0 >= T ? 2 * (T - 1) - i0 : i0; i1 = i1 >= T ? 2 * (T - 1) - i1 : i1;
        xr[i] = make_float2(w[i0], w[i1]);
      }
    }
  };
  auto W = [&](int t) {                                  // exp(-2 pi i t / n_fft), t < n_fft; table holds t < n_fft / 2
    const float2 v = tws[t & (L - 1)];
    return t >= L ? make_float2(-v.x, -v.y) : v;
  };
  auto cmul = [](float2 a, float2 t) { return make_float2(a.x * t.x - a.y * t.y, a.x * t.y + a.y * t.x); };
  load_group(0);
  __syncthreads();
  for (int ff0 = 0; ff0 < FR; ff0 += NF) {
    if (f0 + ff0 >= frames) break;                       // block-uniform
    // 1. windowed frames packed as z[n] = x[2n] + i x[2n+1]
#pragma unroll
    for (int i = 0; i < NPT; ++i)
      if (i < npt) bufA[tid + 256 * i] = make_float2(xr[i].x * win[i].x, xr[i].y * win[i].y);
    __syncthreads();
    if (ff0 + NF < FR && f0 + ff0 + NF < frames) load_group(ff0 + NF);
    float2* src = bufA;
    float2* dst = bufB;
    int Ns = 1;
    if (log2L & 1) {                                     // one radix-2 pass (Ns = 1: unit twiddles)
      for (int j = tid; j < NF * (L >> 1); j += 256) {
        const int fq = j >> (log2L - 1), jj = j & ((L >> 1) - 1);
        const float2 a = src[fq * L + jj], c = src[fq * L + jj + (L >> 1)];
        dst[fq * L + 2 * jj] = make_float2(a.x + c.x, a.y + c.y);
        dst[fq * L + 2 * jj + 1] = make_float2(a.x - c.x, a.y - c.y);
      }
      __syncthreads();
      float2* tmp = src; src = dst; dst = tmp;
      Ns = 2;
    }
    for (; Ns < L; Ns <<= 2) {                           // radix-4 Stockham passes
      const int tstep = n_fft / (4 * Ns);                // twiddle index step: exp(-2 pi i r k / (4 Ns)) = W(r k tstep)
      for (int j = tid; j < NF * Q; j += 256) {
        const int fq = j >> lq, jj = j & (Q - 1), k = jj & (Ns - 1);
        const float2* sp = src + fq * L + jj;
        const float2 a = sp[0];
        const float2 bb = cmul(sp[Q], W(k * tstep));
        const float2 cc = cmul(sp[2 * Q], W(2 * k * tstep));
        const float2 dd = cmul(sp[3 * Q], W(3 * k * tstep));
        const float2 s0 = make_float2(a.x + cc.x, a.y + cc.y), s1 = make_float2(a.x - cc.x, a.y - cc.y);
        const float2 s2 = make_float2(bb.x + dd.x, bb.y + dd.y), s3 = make_float2(bb.x - dd.x, bb.y - dd.y);
        float2* dp = dst + fq * L + ((jj - k) << 2) + k;
        dp[0] = make_float2(s0.x + s2.x, s0.y + s2.y);
        dp[Ns] = make_float2(s1.x + s3.y, s1.y - s3.x);          // s1 - i s3
        dp[2 * Ns] = make_float2(s0.x - s2.x, s0.y - s2.y);
        dp[3 * Ns] = make_float2(s1.x - s3.y, s1.y + s3.x);      // s1 + i s3
      }
      __syncthreads();
      float2* tmp = src; src = dst; dst = tmp;
    }
    // 3. unpack the real spectra: X[k] = E + w_k O, E = (Z[k] + conj Z[L-k])/2, O = -i (Z[k] - conj Z[L-k])/2
    for (int i = tid; i < NF * (L + 1); i += 256) {
      const int fq = i / (L + 1), k = i - fq * (L + 1);
      const float2* z = src + fq * L;
      float re, im;
      if (k == 0 || k == L) {
        const float2 z0 = z[0];
        re = (k == 0) ? z0.x + z0.y : z0.x - z0.y;
        im = 0.f;
      } else {
        const float2 zk = z[k];
        const float2 zc = z[L - k];
        const float er = 0.5f * (zk.x + zc.x), ei = 0.5f * (zk.y - zc.y);
        const float orr = 0.5f * (zk.y + zc.y), oi = -0.5f * (zk.x - zc.x);
        const float2 t = tws[k];
        re = er + (orr * t.x - oi * t.y);
        im = ei + (orr * t.y + oi * t.x);
      }
      outs[k * (FR + 1) + ff0 + fq] = sqrtf(re * re + im * im + 1e-6f);
    }
    __syncthreads();
  }
  const int nf = min(FR, frames - f0);
  for (int i = tid; i < (L + 1) * FR; i += 256) {
    const int k = i / FR, ff = i % FR;
    if (ff < nf) spec[((int64_t)b * (L + 1) + k) * frames + f0 + ff] = outs[k * (FR + 1) + ff];
  }
}

// (round 3) The same radix-4 Stockham passes with everything that does not depend on the frame hoisted out of the frame loop.
// PMC on the kernel above at n_fft 2048: 23.6 M VALU wave-instructions per launch = 144 per butterfly, of which ~34 are the
// butterfly's arithmetic -- the rest is index arithmetic (j -> frame / point / k, Stockham output position), the twiddle lookups
// with their range selects, the division by L + 1 of the unpack loop; VALU 52 % busy, LDS 32 %.  A thread's butterflies are the
// same in every frame group, so its LDS offsets and twiddles are computed ONCE into registers (LOG2L even and compile-time:
// LOG2L / 2 passes x NB butterflies x (1 read offset, 4 write offsets, 3 twiddles)); pass 0 (unit twiddles) skips its complex
// multiplications; the index swizzle SW (8-byte bank pair ^= index bits 5..6: the strided Stockham writes of the first passes
// hit 8 of 32 bank pairs otherwise, 57 % of the LDS-active cycles were conflict cycles) costs nothing at run time any more.
template <int NF, int FR, int LOG2L>
__global__ __launch_bounds__(256) void stft_mag_r4p_kernel(const float* __restrict__ wav, const float* __restrict__ window,
                                                           const float2* __restrict__ tw, float* __restrict__ spec, int T,
                                                           int hop, int frames) {
  static_assert((LOG2L & 1) == 0 && LOG2L >= 8, "even log2 L: radix-4 passes only");
  constexpr int L = 1 << LOG2L, Q = L >> 2, NP = LOG2L / 2, n_fft = 2 * L;
  constexpr int NB = NF * Q / 256;                       // butterflies per thread and pass
  constexpr int NPT = NF * L / 256;                      // packed points per thread and frame group
  static_assert(NF * Q % 256 == 0, "whole butterflies per thread");
  extern __shared__ __attribute__((aligned(16))) float stft_smem[];
  float2* bufA = reinterpret_cast<float2*>(stft_smem);            // [NF][L]
  float2* bufB = bufA + NF * L;
  float* outs = reinterpret_cast<float*>(bufB + NF * L);           // [L + 1][FR + 1]
  const int tid = threadIdx.x;
  const int fblocks = (frames + FR - 1) / FR;
  const int b = blockIdx.x / fblocks;
  const int f0 = (blockIdx.x % fblocks) * FR;
  const int pad = (n_fft - hop) / 2;
  const float* w = wav + (int64_t)b * T;
  auto SW = [](int i) {
    const int bb = (i >> 5) & 3;
    return i ^ bb ^ (bb << 2) ^ (((i >> 6) & 1) << 4);
  };
  auto Wg = [&](int t) {                                 // exp(-2 pi i t / n_fft), t < n_fft, from the global table (t < L)
    const float2 v = tw[t & (L - 1)];
    return t >= L ? make_float2(-v.x, -v.y) : v;
  };
  // per-thread tables
  int rd[NP][NB], wr[NP][NB][4];
  float2 tw1[NP][NB], tw2[NP][NB], tw3[NP][NB];
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int Ns = 1 << (2 * ps), tstep = n_fft / (4 * Ns);
#pragma unroll
    for (int m = 0; m < NB; ++m) {
      const int j = tid + 256 * m, fq = j / Q, jj = j & (Q - 1), k = jj & (Ns - 1);
      rd[ps][m] = fq * L + SW(jj);                       // + c Q, c < 4: Q is a multiple of 128, the swizzle bits do not move
      const int o = ((jj - k) << 2) + k;
#pragma unroll
      for (int c = 0; c < 4; ++c) wr[ps][m][c] = fq * L + SW(o + c * Ns);
      tw1[ps][m] = Wg(k * tstep); tw2[ps][m] = Wg(2 * k * tstep); tw3[ps][m] = Wg(3 * k * tstep);
    }
  }
  float2 win[NPT], xr[NPT];
  int pk[NPT];
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const int n = tid + 256 * i, nn = n & (L - 1);
    win[i] = make_float2(window[2 * nn], window[2 * nn + 1]);
    pk[i] = (n & ~(L - 1)) + SW(nn);
  }
  // unpack: this thread's bins k = tid + 256 u (u < L / 256) of every frame of the group, + bin L on thread 0
  constexpr int NU = L / 256;
  float2 twu[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) twu[u] = tw[tid + 256 * u];
  auto load_group = [&](int ff0) {
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int n = tid + 256 * i, frame = min(f0 + ff0 + (n >> LOG2L), frames - 1), nn = n & (L - 1);
      int i0 = frame * hop + 2 * nn - pad, i1 = i0 + 1;
      i0 = i0 < 0 ? -i0 : i0; i1 = i1 < 0 ? -i1 : i1;
      i0 = i0 >= T ? 2 * (T - 1) - i0 : i0; i1 = i1 >= T ? 2 * (T - 1) - i1 : i1;
      xr[i] = make_float2(w[i0], w[i1]);
    }
  };
  auto cmul = [](float2 a, float2 t) { return make_float2(a.x * t.x - a.y * t.y, a.x * t.y + a.y * t.x); };
  load_group(0);
  for (int ff0 = 0; ff0 < FR; ff0 += NF) {
    if (f0 + ff0 >= frames) break;                       // block-uniform
#pragma unroll
    for (int i = 0; i < NPT; ++i) bufA[pk[i]] = make_float2(xr[i].x * win[i].x, xr[i].y * win[i].y);
    __syncthreads();
    if (ff0 + NF < FR && f0 + ff0 + NF < frames) load_group(ff0 + NF);
    float2* src = bufA;
    float2* dst = bufB;
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
#pragma unroll
      for (int m = 0; m < NB; ++m) {
        const float2* sp = src + rd[ps][m];
        const float2 a = sp[0];
        float2 bb = sp[Q], cc = sp[2 * Q], dd = sp[3 * Q];
        if (ps > 0) { bb = cmul(bb, tw1[ps][m]); cc = cmul(cc, tw2[ps][m]); dd = cmul(dd, tw3[ps][m]); }
        const float2 s0 = make_float2(a.x + cc.x, a.y + cc.y), s1 = make_float2(a.x - cc.x, a.y - cc.y);
        const float2 s2 = make_float2(bb.x + dd.x, bb.y + dd.y), s3 = make_float2(bb.x - dd.x, bb.y - dd.y);
        dst[wr[ps][m][0]] = make_float2(s0.x + s2.x, s0.y + s2.y);
        dst[wr[ps][m][1]] = make_float2(s1.x + s3.y, s1.y - s3.x);          // s1 - i s3
        dst[wr[ps][m][2]] = make_float2(s0.x - s2.x, s0.y - s2.y);
        dst[wr[ps][m][3]] = make_float2(s1.x - s3.y, s1.y + s3.x);          // s1 + i s3
      }
      __syncthreads();
      float2* tmp = src; src = dst; dst = tmp;
    }
    // unpack the real spectra: X[k] = E + w_k O, E = (Z[k] + conj Z[L-k]) / 2, O = -i (Z[k] - conj Z[L-k]) / 2
#pragma unroll
    for (int fq = 0; fq < NF; ++fq) {
      const float2* z = src + fq * L;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int k = tid + 256 * u;
        float re, im;
        if (k == 0) {
          const float2 z0 = z[0];
          re = z0.x + z0.y; im = 0.f;
          outs[L * (FR + 1) + ff0 + fq] = sqrtf((z0.x - z0.y) * (z0.x - z0.y) + 1e-6f);   // bin L
        } else {
          const float2 zk = z[SW(k)];
          const float2 zc = z[SW(L - k)];
          const float er = 0.5f * (zk.x + zc.x), ei = 0.5f * (zk.y - zc.y);
          const float orr = 0.5f * (zk.y + zc.y), oi = -0.5f * (zk.x - zc.x);
          re = er + (orr * twu[u].x - oi * twu[u].y);
          im = ei + (orr * twu[u].y + oi * twu[u].x);
        }
        outs[k * (FR + 1) + ff0 + fq] = sqrtf(re * re + im * im + 1e-6f);
      }
    }
    __syncthreads();
  }
  const int nf = min(FR, frames - f0);
  for (int i = tid; i < (L + 1) * FR; i += 256) {
    const int k = i / FR, ff = i % FR;
    if (ff < nf) spec[((int64_t)b * (L + 1) + k) * frames + f0 + ff] = outs[k * (FR + 1) + ff];
  }
}

// (round 6) n_fft = 2048 with ONE WAVE PER FRAME and ONE LDS transpose.  The radix-4 kernel above sends every frame through five
// barrier-separated LDS passes (write + read of the whole 8-KB frame with stride-Ns addressing each): 50 us for 32 clips = 13.6 % of
// the HBM roof, neither HBM (6.8 us) nor VALU (6.4 us) bound but LDS-pass / barrier bound.  Here the packed 1024-point complex
// transform is factored 32 x 32 (n = 32 n1 + n2, k = k1 + 32 k2):
//     Z[k1 + 32 k2] = sum_n2 W_1024^(n2 k1) W_32^(n2 k2) [ sum_n1 z[32 n1 + n2] W_32^(n1 k1) ]
// and BOTH 32-point transforms run in registers: lane (c = lane & 31, h = lane >> 5) holds the 16 points n1 = h + 2 m of column c,
// does a radix-16 (two radix-4 stages) on them, and the two halves of a column combine through one cross-half exchange
// (F[j + 16 s] = G_0[j] + (-1)^s W_32^j G_1[j]).  Between the two transforms the 32 x 32 matrix is transposed through a
// WAVE-PRIVATE 8-KB LDS buffer (pitch 33: conflict-free both ways) -- no workgroup barrier anywhere in a frame; the real-spectrum
// unpacking fetches Z[L - k] through the same buffer.  Per frame: 2 x (8 KB written + 8 KB read) of LDS instead of 5 x, zero
// barriers instead of 7.  Magnitudes are parked in a [bin][16 frames] tile and leave as 64-byte runs along the frame axis.
// Twiddles: the host table (exp(-2 pi i t / 2048), t < 1024) staged in LDS; W_16 / W_32 powers are literals.
namespace w32 {
#ifndef TTTS_STFT_W32_FR
#define TTTS_STFT_W32_FR 16
#define TTTS_STFT_W32_WAVES 8
#endif
// FR frames per workgroup (the output leaves in FR x 4-byte runs), WAVES waves, FR / WAVES frames per wave.  Default 16 x 8 with the
// twiddle / window tables in LDS: 153 KB, one workgroup per CU, 64-byte runs.  -DTTTS_STFT_W32_FR=8 -DTTTS_STFT_W32_WAVES=4 reads the
// tables from global memory (L2 / L1-resident, 8 KB each) and fits TWO workgroups per CU (70.7 KB): measured level, 41.3 vs 40.7 us
// (tools/gpu_r6_af.sh) -- the kernel is bound by its ~1 500 VALU instructions per frame, not by the overlap of its phases
constexpr int FR = TTTS_STFT_W32_FR, WAVES = TTTS_STFT_W32_WAVES, L = 1024, PITCH = 33;
constexpr bool TABLES_IN_LDS = FR >= 16;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 t) { return make_float2(a.x * t.x - a.y * t.y, a.x * t.y + a.y * t.x); }
// forward 4-point DFT in place: (a, b, c, d) -> (y0, y1, y2, y3)
__device__ __forceinline__ void fft4(float2& a, float2& b, float2& c, float2& d) {
  const float2 s0 = cadd(a, c), s1 = csub(a, c), s2 = cadd(b, d), s3 = csub(b, d);
  a = cadd(s0, s2); c = csub(s0, s2);
  b = make_float2(s1.x + s3.y, s1.y - s3.x);          // s1 - i s3
  d = make_float2(s1.x - s3.y, s1.y + s3.x);          // s1 + i s3
}
// forward 16-point DFT in place; X[k] ends up in x[(k >> 2) + 4 (k & 3)]
__device__ __forceinline__ void fft16(float2 (&x)[16]) {
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R = 0.70710678118654752f;
#pragma unroll
  for (int n1 = 0; n1 < 4; ++n1) fft4(x[n1], x[n1 + 4], x[n1 + 8], x[n1 + 12]);     // x[n1 + 4 k2] = A[n1][k2]
  // A[n1][k2] *= W_16^(n1 k2)
  x[1 + 4] = cmul(x[1 + 4], make_float2(C1, -S1));   x[1 + 8] = cmul(x[1 + 8], make_float2(R, -R));    x[1 + 12] = cmul(x[1 + 12], make_float2(S1, -C1));
  x[2 + 4] = cmul(x[2 + 4], make_float2(R, -R));     x[2 + 8] = make_float2(x[2 + 8].y, -x[2 + 8].x);  x[2 + 12] = cmul(x[2 + 12], make_float2(-R, -R));
  x[3 + 4] = cmul(x[3 + 4], make_float2(S1, -C1));   x[3 + 8] = cmul(x[3 + 8], make_float2(-R, -R));   x[3 + 12] = cmul(x[3 + 12], make_float2(-C1, S1));
#pragma unroll
  for (int k2 = 0; k2 < 4; ++k2) fft4(x[4 * k2], x[4 * k2 + 1], x[4 * k2 + 2], x[4 * k2 + 3]);       // x[k1 + 4 k2] = X[4 k1 + k2]
}
// The 32-point transform of a column held by a lane pair: in x[m] = column point h + 2 m; out y[j] = F[j + 16 h] (natural order).
__device__ __forceinline__ void fft32_pair(float2 (&x)[16], float2 (&y)[16], int h) {
  // W_32^j = exp(-2 pi i j / 32), j < 16
  constexpr float C32[16] = {1.f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f,
                             0.55557023301960218f, 0.38268343236508977f, 0.19509032201612825f, 0.f, -0.19509032201612825f,
                             -0.38268343236508977f, -0.55557023301960218f, -0.70710678118654752f, -0.83146961230254524f,
                             -0.92387953251128674f, -0.98078528040323043f};
  constexpr float S32[16] = {0.f, 0.19509032201612825f, 0.38268343236508977f, 0.55557023301960218f, 0.70710678118654752f,
                             0.83146961230254524f, 0.92387953251128674f, 0.98078528040323043f, 1.f, 0.98078528040323043f,
                             0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f, 0.55557023301960218f,
                             0.38268343236508977f, 0.19509032201612825f};
  fft16(x);
  const float sgn = h ? -1.f : 1.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    // the odd half carries W_32^j (a per-lane twiddle (1, 0) on the even half: no divergent multiply, no select of products)
    const float2 g = cmul(x[(j >> 2) + 4 * (j & 3)], make_float2(h ? C32[j] : 1.f, h ? -S32[j] : 0.f));
    // the other half's value: one v_permlane32_swap per component (swap(v, v) = ([lo, lo], [hi, hi])) instead of a ds_bpermute round trip
    const u32x2 sx = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, g.x), __builtin_bit_cast(uint32_t, g.x), false, false);
    const u32x2 sy = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, g.y), __builtin_bit_cast(uint32_t, g.y), false, false);
    const uint32_t sxl = sx[0], sxh = sx[1], syl = sy[0], syh = sy[1];
    const float ox = __builtin_bit_cast(float, h ? sxl : sxh), oy = __builtin_bit_cast(float, h ? syl : syh);
    y[j] = make_float2(fmaf(sgn, g.x, ox), fmaf(sgn, g.y, oy));   // F[j] = G0 + W G1 (even half) ;  F[j + 16] = G0 - W G1 (odd half)
  }
}
}  // namespace w32

__global__ __launch_bounds__(64 * w32::WAVES, 2) void stft_mag_w32_kernel(const float* __restrict__ wav, const float* __restrict__ window,
                                                           const float2* __restrict__ tw, float* __restrict__ spec, int T, int hop,
                                                           int frames) {
  using namespace w32;
  constexpr int n_fft = 2 * L;
  extern __shared__ __attribute__((aligned(16))) float stft_smem[];
  constexpr int NT = 64 * WAVES;
  float2* tws_l = reinterpret_cast<float2*>(stft_smem);           // [L] exp(-2 pi i t / n_fft)        } only when TABLES_IN_LDS
  float2* wins_l = tws_l + L;                                     // [L] (w[2 n], w[2 n + 1])          }
  float2* bufs = TABLES_IN_LDS ? wins_l + L : tws_l;              // [WAVES][32 * PITCH]
  float* outs = reinterpret_cast<float*>(bufs + WAVES * 32 * PITCH);   // [L + 1][FR + 1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, h = lane >> 5;
  if (TABLES_IN_LDS) {
    for (int i = tid; i < L; i += NT) {
      tws_l[i] = tw[i];
      wins_l[i] = make_float2(window[2 * i], window[2 * i + 1]);
    }
    __syncthreads();
  }
  const float2* tws = TABLES_IN_LDS ? tws_l : tw;
  const float2* wins = TABLES_IN_LDS ? wins_l : reinterpret_cast<const float2*>(window);
  const int fblocks = (frames + FR - 1) / FR;
  const int b = blockIdx.x / fblocks, f0 = (blockIdx.x % fblocks) * FR;
  const int pad = (n_fft - hop) / 2;
  const float* w = wav + (int64_t)b * T;
  float2* buf = bufs + wave * 32 * PITCH;
  auto W2048 = [&](int t) {                                       // exp(-2 pi i t / 2048), any t >= 0
    t &= n_fft - 1;
    const float2 v = tws[t & (L - 1)];
    return t >= L ? make_float2(-v.x, -v.y) : v;
  };
  // per-lane twiddles, the same for every frame: between the transforms W_1024^(c k1), k1 = j + 16 h; unpacking: w_k, k = c + 32 (j + 16 h)
  float2 twa[16], twu[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    twa[j] = W2048(2 * c * (j + 16 * h));
    twu[j] = tws[c + 32 * (j + 16 * h)];
  }
  auto load_frame = [&](int f, float2 (&x)[16]) {
    const int frame = min(f, frames - 1);
    // (an interior-frame fast path without the reflection folding -- ~190 VALU instructions per frame -- doubled the load code and
    // spilled 35 VGPRs at the 256-register budget: 48.8 us; removed)
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int n = 64 * m + lane;                                // packed point 32 (h + 2 m) + c
      int i0 = frame * hop + 2 * n - pad, i1 = i0 + 1;
      i0 = i0 < 0 ? -i0 : i0; i1 = i1 < 0 ? -i1 : i1;
      i0 = i0 >= T ? 2 * (T - 1) - i0 : i0; i1 = i1 >= T ? 2 * (T - 1) - i1 : i1;
      x[m] = make_float2(w[i0], w[i1]);
    }
  };
  constexpr int FPW = FR / WAVES;                                 // frames per wave
  float2 x[16], y[16], nx[16];
  load_frame(f0 + wave * FPW, nx);
#pragma unroll
  for (int q = 0; q < FPW; ++q) {
    const int fi = wave * FPW + q;                                // frame slot of the tile
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const float2 wn = wins[64 * m + lane];
      x[m] = make_float2(nx[m].x * wn.x, nx[m].y * wn.y);
    }
    if (q + 1 < FPW) load_frame(f0 + fi + 1, nx);                 // in flight under this frame's arithmetic
    fft32_pair(x, y, h);                                          // y[j] = F[k1 = j + 16 h] of column n2 = c
    __builtin_amdgcn_wave_barrier();                              // (the previous frame's reads of the buffer are done: same wave, in order)
#pragma unroll
    for (int j = 0; j < 16; ++j) buf[(j + 16 * h) * PITCH + c] = cmul(y[j], twa[j]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < 16; ++m) x[m] = buf[c * PITCH + h + 2 * m];   // row k1 = c, points n2 = h + 2 m
    fft32_pair(x, y, h);                                          // y[j] = Z[c + 32 (j + 16 h)]
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 16; ++j) buf[c + 32 * (j + 16 * h)] = y[j];   // natural order for the Z[L - k] fetch
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int k = c + 32 * (j + 16 * h);
      const float2 zk = y[j], zc = buf[(L - k) & (L - 1)];
      float re, im;
      if (k == 0) {
        re = zk.x + zk.y; im = 0.f;
        outs[L * (FR + 1) + fi] = sqrtf((zk.x - zk.y) * (zk.x - zk.y) + 1e-6f);   // bin L
      } else {
        const float er = 0.5f * (zk.x + zc.x), ei = 0.5f * (zk.y - zc.y);
        const float orr = 0.5f * (zk.y + zc.y), oi = -0.5f * (zk.x - zc.x);
        re = er + (orr * twu[j].x - oi * twu[j].y);
        im = ei + (orr * twu[j].y + oi * twu[j].x);
      }
      outs[k * (FR + 1) + fi] = __builtin_amdgcn_sqrtf(re * re + im * im + 1e-6f);
    }
  }
  __syncthreads();
  const int nf = min(FR, frames - f0);
  for (int i = tid; i < (L + 1) * FR; i += NT) {
    const int k = i / FR, ff = i % FR;
    if (ff < nf) spec[((int64_t)b * (L + 1) + k) * frames + f0 + ff] = outs[k * (FR + 1) + ff];
  }
}

// mel[b][m][f] = log(max(sum_k basis[m][k] spec[b][k][f], 1e-5)).
// (round 2) A mel filterbank row is a narrow band (Slaney triangles: ~2 x 1025 non-zeros in 128 x 1025), so the dense
// 32 x 64 x 32 tiled product of round 1 (166 us for 32 clips: 2.9 % of the HBM roof, all of it multiplying zeros) is replaced
// by a band-limited row kernel: each wave finds the first / last non-zero of its basis row (one pass over the row, no
// assumption about the basis beyond "zeros are zeros"), then accumulates basis[m][k] * spec[b][k][f] over that band in
// increasing k -- the same fmaf chain as the dense loop minus its exact-zero terms, i.e. bit-identical results -- with the
// 64 lanes on 64 consecutive frames (256-byte coalesced spec reads, eight in flight).  A workgroup = (clip, 64 frames,
// MEL_MT mels, one per wave: 4096 small workgroups measured 13.8 us against 25.6 with 32 mels per workgroup).
constexpr int MEL_MT = 4;    // mels per workgroup (interleaved over its 4 waves)
__global__ __launch_bounds__(256) void mel_log_kernel(const float* __restrict__ spec, const float* __restrict__ basis,
                                                      const int* __restrict__ bands, float* __restrict__ mel, int n_bins,
                                                      int n_mels, int frames) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ftiles = (frames + 63) / 64, mtiles = (n_mels + MEL_MT - 1) / MEL_MT;
  const int b = blockIdx.x / (ftiles * mtiles);
  const int rem = blockIdx.x % (ftiles * mtiles);
  const int mg = rem / ftiles, f = (rem % ftiles) * 64 + lane;
  const float* sp = spec + (int64_t)b * n_bins * frames + min(f, frames - 1);
  for (int i = 0; i < MEL_MT / 4; ++i) {
    const int m = mg * MEL_MT + i * 4 + wave;       // wave-uniform
    if (m >= n_mels) break;
    const float* brow = basis + (int64_t)m * n_bins;
    int lo = n_bins, hi = -1;
    if (bands) {                                   // caller-computed (once per basis)
      lo = bands[2 * m]; hi = bands[2 * m + 1];
    } else {                                       // one pass over the row (17 dependent L2 round trips: measurably slower)
      for (int k = lane; k < n_bins; k += 64)
        if (brow[k] != 0.f) { lo = min(lo, k); hi = max(hi, k); }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o, 64));
        hi = max(hi, __shfl_xor(hi, o, 64));
      }
    }
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    float acc = 0.f;
    int k = lo;
    for (; k + 8 <= hi + 1; k += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = sp[(int64_t)(k + j) * frames];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = fmaf(brow[k + j], v[j], acc);
    }
    for (; k <= hi; ++k) acc = fmaf(brow[k], sp[(int64_t)k * frames], acc);
    if (f < frames) mel[((int64_t)b * n_mels + m) * frames + f] = logf(fmaxf(acc, 1e-5f));
  }
}

// ---- backward ---------------------------------------------------------------------------------------------------
// d spec[b][k][f] = sum_m basis[m][k] * g[m][f],  g = dmel / max(v, 1e-5) where the clamp passed (v = exp(mel) >= 1e-5)
// (torch.clamp(min) passes the gradient where x >= min; log'(v) = 1/v).  Tile: 32 bins x 64 frames, m chunks of 32.
__global__ __launch_bounds__(256) void mel_log_bwd_kernel(const float* __restrict__ dmel, const float* __restrict__ mel,
                                                          const float* __restrict__ basis, float* __restrict__ dspec,
                                                          int n_bins, int n_mels, int frames) {
  __shared__ float As[32][33];  // [bin][m]
  __shared__ float Bs[32][64];  // [m][frame]
  const int tid = threadIdx.x, tm = tid >> 6, tf = tid & 63;
  const int ftiles = (frames + 63) / 64, ktiles = (n_bins + 31) / 32;
  const int b = blockIdx.x / (ftiles * ktiles);
  const int rem = blockIdx.x % (ftiles * ktiles);
  const int k0 = (rem / ftiles) * 32, f0 = (rem % ftiles) * 64;
  const float LOG_CLIP = -11.512925464970229f;  // log(1e-5)
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int m0 = 0; m0 < n_mels; m0 += 32) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256, m = e >> 5, k = e & 31;  // coalesced along bins
      As[k][m] = (m0 + m < n_mels && k0 + k < n_bins) ? basis[(int64_t)(m0 + m) * n_bins + k0 + k] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + i * 256, m = e >> 6, f = e & 63;
      float g = 0.f;
      if (m0 + m < n_mels && f0 + f < frames) {
        const int64_t idx = ((int64_t)b * n_mels + m0 + m) * frames + f0 + f;
        const float ml = mel[idx];
        // forward value log(max(v, clip)): ml > log(clip) <=> v > clip; at equality torch passes the gradient too
        g = ml >= LOG_CLIP ? dmel[idx] * __expf(-ml) : 0.f;
      }
      Bs[m][f] = g;
    }
    __syncthreads();
#pragma unroll 8
    for (int m = 0; m < 32; ++m) {
      const float bv = Bs[m][tf];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = fmaf(As[tm * 8 + i][m], bv, acc[i]);
    }
    __syncthreads();
  }
  if (f0 + tf < frames) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = k0 + tm * 8 + i;
      if (k < n_bins) dspec[((int64_t)b * n_bins + k) * frames + f0 + tf] += acc[i];
    }
  }
}

// STFT-magnitude backward, one workgroup per frame: recompute the frame's spectrum (same packed real FFT as the forward),
// G_k = dmag_k * (re, im) / mag, then the adjoint of the real DFT,
//   dframe[n] = Re( sum_{k=0..N/2} G_k e^{+2 pi i k n / N} ) = Re( FFT_N(conj(H)) )[n],  H_k = G_k (k <= N/2), 0 otherwise,
// as an N-point complex Stockham FFT, and scatter-add window[n] * dframe[n] onto the (reflect-folded) waveform gradient.
// tw2: twiddle table for 2N (N entries exp(-2 pi i t / 2N)); exp(-2 pi i t / N) = tw2[2t].
__global__ __launch_bounds__(256) void stft_mag_bwd_kernel(const float* __restrict__ wav, const float* __restrict__ window,
                                                           const float2* __restrict__ tw2, const float* __restrict__ dspec,
                                                           float* __restrict__ dwav, int T, int n_fft, int hop,
                                                           int frames, int log2L) {
  extern __shared__ __attribute__((aligned(16))) float stft_smem[];
  const int N = n_fft, L = n_fft >> 1;
  float2* buf0 = reinterpret_cast<float2*>(stft_smem);  // N complex each
  float2* buf1 = buf0 + N;
  const int tid = threadIdx.x;
  const int b = blockIdx.x / frames, frame = blockIdx.x % frames;
  const int pad = (n_fft - hop) / 2;
  const float* w = wav + (int64_t)b * T;
  // 1. forward: packed real FFT of the windowed frame (length L complex)
  for (int n = tid; n < L; n += 256) {
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      int i = frame * hop + 2 * n + e - pad;
      if (i < 0) i = -i;
      if (i >= T) i = 2 * (T - 1) - i;
      v[e] = w[i] * window[2 * n + e];
    }
    buf0[n] = make_float2(v[0], v[1]);
  }
  __syncthreads();
  float2* src = buf0;
  float2* dst = buf1;
  for (int ps = 0; ps < log2L; ++ps) {
    const int Ns = 1 << ps;
    for (int j = tid; j < (L >> 1); j += 256) {
      const int k = j & (Ns - 1);
      const float2 a = src[j];
      const float2 bb = src[j + (L >> 1)];
      const float2 t = tw2[2 * (k * (n_fft >> (ps + 1)))];
      const float2 bt = make_float2(bb.x * t.x - bb.y * t.y, bb.x * t.y + bb.y * t.x);
      const int j0 = (j << 1) - k;
      dst[j0] = make_float2(a.x + bt.x, a.y + bt.y);
      dst[j0 + Ns] = make_float2(a.x - bt.x, a.y - bt.y);
    }
    __syncthreads();
    float2* tmp = src; src = dst; dst = tmp;
  }
  // 2. H = conj(G) into dst (length N, upper half zero)
  const float* dsp = dspec + (int64_t)b * (L + 1) * frames + frame;
  for (int k = tid; k < N; k += 256) {
    float2 h = make_float2(0.f, 0.f);
    if (k <= L) {
      float re, im;
      if (k == 0 || k == L) {
        const float2 z0 = src[0];
        re = (k == 0) ? z0.x + z0.y : z0.x - z0.y;
        im = 0.f;
      } else {
        const float2 zk = src[k];
        const float2 zc = src[L - k];
        const float er = 0.5f * (zk.x + zc.x), ei = 0.5f * (zk.y - zc.y);
        const float orr = 0.5f * (zk.y + zc.y), oi = -0.5f * (zk.x - zc.x);
        const float2 t = tw2[2 * k];
        re = er + (orr * t.x - oi * t.y);
        im = ei + (orr * t.y + oi * t.x);
      }
      const float mag = sqrtf(re * re + im * im + 1e-6f);
      const float gk = dsp[(int64_t)k * frames] / mag;
      h = make_float2(gk * re, -gk * im);
    }
    dst[k] = h;
  }
  __syncthreads();
  // 3. N-point complex FFT of H (dst -> ...), log2(N) = log2L + 1 passes
  {
    float2* s2 = dst;
    float2* d2 = src;
    for (int ps = 0; ps <= log2L; ++ps) {
      const int Ns = 1 << ps;
      for (int j = tid; j < L; j += 256) {  // N/2 butterflies
        const int k = j & (Ns - 1);
        const float2 a = s2[j];
        const float2 bb = s2[j + L];
        const float2 t = tw2[k * (n_fft >> ps)];  // exp(-2 pi i k / (2 Ns)) = tw2[k * 2N / (2 Ns)]
        const float2 bt = make_float2(bb.x * t.x - bb.y * t.y, bb.x * t.y + bb.y * t.x);
        const int j0 = (j << 1) - k;
        d2[j0] = make_float2(a.x + bt.x, a.y + bt.y);
        d2[j0 + Ns] = make_float2(a.x - bt.x, a.y - bt.y);
      }
      __syncthreads();
      float2* tmp = s2; s2 = d2; d2 = tmp;
    }
    // 4. overlap-add onto the waveform gradient through the reflect padding
    float* dw = dwav + (int64_t)b * T;
    for (int n = tid; n < N; n += 256) {
      int i = frame * hop + n - pad;
      if (i < 0) i = -i;
      if (i >= T) i = 2 * (T - 1) - i;
      atomicAdd(dw + i, s2[n].x * window[n]);
    }
  }
}

}  // namespace ttts

using namespace ttts;

extern "C" int ttts_stft_twiddle_host(float* host_out, int32_t n_fft) {
  TTTS_REQUIRE(host_out && n_fft >= 4 && (n_fft & (n_fft - 1)) == 0, "stft_twiddle: n_fft must be a power of two");
  for (int k = 0; k < n_fft / 2; ++k) {
    const double a = -2.0 * M_PI * (double)k / (double)n_fft;
    host_out[2 * k] = (float)cos(a);
    host_out[2 * k + 1] = (float)sin(a);
  }
  return TTTS_OK;
}

extern "C" int ttts_stft_mag_fwd_f32(const float* wav, const float* window, const float* twiddle, float* spec,
                                     int32_t B, int32_t T, int32_t n_fft, int32_t hop, void* stream) {
  TTTS_REQUIRE(wav && window && twiddle && spec, "stft: null pointer");
  TTTS_REQUIRE(n_fft >= 64 && n_fft <= 4096 && (n_fft & (n_fft - 1)) == 0, "stft: n_fft must be a power of two in [64, 4096]");
  TTTS_REQUIRE(hop > 0 && hop <= n_fft && B > 0, "stft: bad hop / batch");
  const int pad = (n_fft - hop) / 2;
  TTTS_REQUIRE(T > pad, "stft: reflect padding needs T > (n_fft - hop)/2");
  TTTS_REQUIRE(T + 2 * pad >= n_fft, "stft: clip shorter than one frame");
  const int frames = (T + 2 * pad - n_fft) / hop + 1;
  const int L = n_fft / 2;
  int log2L = 0;
  while ((1 << log2L) < L) ++log2L;
  constexpr int NF = 2, FR4 = 8;                         // radix-4 kernel: frames per pass, frames per workgroup (16 frames =
  const bool r4 = L >= 256 && L <= 2048;                 // 64-byte output runs, but 110 KB of LDS = one workgroup per CU: 120 us vs 91); short transforms keep radix-2
  static const bool r4_forced = getenv("TTTS_STFT_R4") != nullptr;   // (A/B switch: the round-3 radix-4 kernel)
  if (L == 1024 && !r4_forced) {             // n_fft 2048 (the training configuration): one wave per frame, 32 x 32 in registers
    const size_t smemw = (w32::TABLES_IN_LDS ? (size_t)2 * L * sizeof(float2) : 0) + (size_t)w32::WAVES * 32 * w32::PITCH * sizeof(float2) +
                         (size_t)(L + 1) * (w32::FR + 1) * sizeof(float);
    static OnceFlag attrw_once;
    const hipError_t attrw = lds_opt_in(attrw_once, reinterpret_cast<const void*>(stft_mag_w32_kernel), 160 * 1024);
    if (attrw != hipSuccess) return fail(TTTS_EHIP, "stft: hipFuncSetAttribute: %s", hipGetErrorString(attrw));
    stft_mag_w32_kernel<<<B * (int)cdiv(frames, w32::FR), 64 * w32::WAVES, smemw, as_stream(stream)>>>(
        wav, window, reinterpret_cast<const float2*>(twiddle), spec, T, hop, frames);
    return check_launch("stft_mag_fwd");
  }
  if (L == 1024) {                                        // (TTTS_STFT_R4=1: the round-3 radix-4 kernel) per-thread tables, LDS swizzle
    const size_t smemp = (size_t)2 * NF * L * sizeof(float2) + (size_t)(L + 1) * (FR4 + 1) * sizeof(float);
    static OnceFlag attrp_once;
    const hipError_t attrp = lds_opt_in(attrp_once, reinterpret_cast<const void*>(stft_mag_r4p_kernel<NF, FR4, 10>), 160 * 1024);
    if (attrp != hipSuccess) return fail(TTTS_EHIP, "stft: hipFuncSetAttribute: %s", hipGetErrorString(attrp));
    stft_mag_r4p_kernel<NF, FR4, 10><<<B * (int)cdiv(frames, FR4), 256, smemp, as_stream(stream)>>>(
        wav, window, reinterpret_cast<const float2*>(twiddle), spec, T, hop, frames);
    return check_launch("stft_mag_fwd");
  }
  if (r4) {
    const size_t smem4 = (size_t)2 * NF * L * sizeof(float2) + ((size_t)(L + 1) * (FR4 + 1) + (((L + 1) * (FR4 + 1)) & 1)) * sizeof(float) +
                         (size_t)L * sizeof(float2);
    static OnceFlag attr4_once;
    const hipError_t attr4 = lds_opt_in(attr4_once, reinterpret_cast<const void*>(stft_mag_r4_kernel<NF, FR4>), 160 * 1024);
    if (attr4 != hipSuccess) return fail(TTTS_EHIP, "stft: hipFuncSetAttribute: %s", hipGetErrorString(attr4));
    stft_mag_r4_kernel<NF, FR4><<<B * (int)cdiv(frames, FR4), 256, smem4, as_stream(stream)>>>(
        wav, window, reinterpret_cast<const float2*>(twiddle), spec, T, n_fft, hop, frames, log2L);
    return check_launch("stft_mag_fwd");
  }
  size_t smem = (size_t)2 * L * sizeof(float2) + ((size_t)(L + 1) * (STFT_FR + 1) + ((L + 1) & 1)) * sizeof(float) + (size_t)L * sizeof(float2);
  const size_t span_bytes = ((size_t)(STFT_FR - 1) * hop + n_fft) * sizeof(float);
  const int span_lds = smem + span_bytes <= 80 * 1024;   // only where it does not cost the second workgroup per CU
  if (span_lds) smem += span_bytes;
  static OnceFlag attr_once;
    const hipError_t attr = lds_opt_in(attr_once, reinterpret_cast<const void*>(stft_mag_kernel), 160 * 1024);
  if (attr != hipSuccess) return fail(TTTS_EHIP, "stft: hipFuncSetAttribute: %s", hipGetErrorString(attr));
  const int grid = B * (int)cdiv(frames, STFT_FR);
  stft_mag_kernel<<<grid, 256, smem, as_stream(stream)>>>(wav, window, reinterpret_cast<const float2*>(twiddle), spec, T,
                                                          n_fft, hop, frames, log2L, span_lds);
  return check_launch("stft_mag_fwd");
}

extern "C" int ttts_mel_log_fwd_f32(const float* spec, const float* basis, const int32_t* bands, float* mel, int32_t B,
                                    int32_t n_bins, int32_t n_mels, int32_t frames, void* stream) {
  TTTS_REQUIRE(spec && basis && mel && B > 0 && n_bins > 0 && n_mels > 0 && frames > 0, "mel_log: bad arguments");
  const int grid = B * (int)cdiv(frames, 64) * (int)cdiv(n_mels, MEL_MT);
  mel_log_kernel<<<grid, 256, 0, as_stream(stream)>>>(spec, basis, bands, mel, n_bins, n_mels, frames);
  return check_launch("mel_log_fwd");
}

extern "C" int ttts_mel_log_bwd_f32(const float* dmel, const float* mel, const float* basis, float* dspec, int32_t B,
                                    int32_t n_bins, int32_t n_mels, int32_t frames, void* stream) {
  TTTS_REQUIRE(dmel && mel && basis && dspec && B > 0 && n_bins > 0 && n_mels > 0 && frames > 0, "mel_log_bwd: bad arguments");
  const int grid = B * (int)cdiv(frames, 64) * (int)cdiv(n_bins, 32);
  mel_log_bwd_kernel<<<grid, 256, 0, as_stream(stream)>>>(dmel, mel, basis, dspec, n_bins, n_mels, frames);
  return check_launch("mel_log_bwd");
}

extern "C" int ttts_stft_mag_bwd_f32(const float* wav, const float* window, const float* twiddle2, const float* dspec,
                                     float* dwav, int32_t B, int32_t T, int32_t n_fft, int32_t hop, void* stream) {
  TTTS_REQUIRE(wav && window && twiddle2 && dspec && dwav, "stft_bwd: null pointer");
  TTTS_REQUIRE(n_fft >= 64 && n_fft <= 4096 && (n_fft & (n_fft - 1)) == 0, "stft_bwd: n_fft must be a power of two in [64, 4096]");
  TTTS_REQUIRE(hop > 0 && hop <= n_fft && B > 0, "stft_bwd: bad hop / batch");
  const int pad = (n_fft - hop) / 2;
  TTTS_REQUIRE(T > pad && T + 2 * pad >= n_fft, "stft_bwd: clip too short");
  const int frames = (T + 2 * pad - n_fft) / hop + 1;
  int log2L = 0;
  while ((1 << log2L) < n_fft / 2) ++log2L;
  const size_t smem = (size_t)2 * n_fft * sizeof(float2);
  static OnceFlag attr_set;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(stft_mag_bwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return fail(TTTS_EHIP, "stft_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  stft_mag_bwd_kernel<<<B * frames, 256, smem, as_stream(stream)>>>(wav, window, reinterpret_cast<const float2*>(twiddle2),
                                                                   dspec, dwav, T, n_fft, hop, frames, log2L);
  return check_launch("stft_mag_bwd");
}
