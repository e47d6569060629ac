// Shared host/device helpers for libttts_hip.so (gfx950 only -- no CUDA / multi-backend paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

#include "../../include/ttts_hip.h"

namespace ttts {

// ---- error plumbing (thread-local message, no exceptions across the ABI) ---------------------------
char* error_buffer();
inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}
// "done on the CURRENT device": the dynamic-LDS opt-in of hipFuncSetAttribute is a per-device attribute, so a once-flag next to a
// launch site must be per device as well (one process may drive several devices).  Drop-in for a `static bool`.
// Device ordinal the CURRENT C-ABI call runs on, when the caller told us (ttts_conv_ctx::device: the convolution family launches
// thousands of kernels per step and hipGetDevice is an API call), else -1: ask the runtime.  Set for the duration of an entry point.
inline int& device_hint() { static thread_local int d = -1; return d; }
struct DeviceHint {
  explicit DeviceHint(int d) { device_hint() = d; }
  ~DeviceHint() { device_hint() = -1; }
};
struct OnceFlag {
  std::atomic<uint64_t> mask{0};
  static int dev() {
    int d = device_hint();
    if (d < 0) (void)hipGetDevice(&d);
    return d & 63;
  }
  operator bool() const { return (mask.load(std::memory_order_acquire) >> dev()) & 1; }
  OnceFlag& operator=(bool v) { if (v) mask.fetch_or(1ull << dev(), std::memory_order_release); return *this; }
};
inline hipError_t lds_opt_in(OnceFlag& f, const void* fn, int bytes = 160 * 1024) {
  if (f) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) f = true;
  return e;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(TTTS_EHIP, "%s: %s", what, hipGetErrorString(e));
  return TTTS_OK;
}
#define TTTS_REQUIRE(cond, ...) \
  do {                          \
    if (!(cond)) return ::ttts::fail(TTTS_EINVAL, __VA_ARGS__); \
  } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// The convolution family's caller-owned context (include/ttts_hip.h: ttts_conv_ctx), as the internal launchers see it.
struct ConvCtx {
  void* ws = nullptr;      // scratch for split-bf16 operand copies / weight-gradient slabs; NULL: exact-fp32 kernels only
  int64_t ws_bytes = 0;
  int flags = 0;           // TTTS_CONV_EXACT_F32 | heuristic overrides (tests, tools/conv_bench.py)
  void* const* handles = nullptr;   // the caller's weight-split caches / weight-gradient arenas this call may use (ABI v10)
  int n_handles = 0;
  int device = -1;                  // device ordinal of the call (ABI v10), -1: unknown
};
// Every object handed out through ttts_conv_ctx::handles starts with this tag (the two kinds share one list).
enum : uint32_t { TTTS_HANDLE_WSPLIT = 0x4c505357u /* "WSPL" */, TTTS_HANDLE_SLAB = 0x42414c53u /* "SLAB" */ };
inline ConvCtx conv_ctx_of(const ttts_conv_ctx* c) {
  ConvCtx cx;
  if (c) {
    cx.ws = c->workspace;
    cx.ws_bytes = c->workspace ? c->workspace_bytes : 0;
    cx.flags = c->flags;
    if (c->handles && c->n_handles > 0) { cx.handles = c->handles; cx.n_handles = c->n_handles; }
    cx.device = c->device;
  }
  return cx;
}

// ---- device types ----------------------------------------------------------------------------------
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int WAVE = 64;  // CDNA wavefront

// XCD-aware workgroup order: the hardware deals workgroup ids round-robin over the 8 XCDs (private L2s); this maps id -> logical
// tile so that each XCD walks a CONTIGUOUS range of logical tiles (tiles that share an operand then hit the same L2).
// Bijective for any count.
__device__ __forceinline__ int xcd_order(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// logical (x, y, z) of this workgroup of a 3-D grid, x fastest, through xcd_order: an XCD's L2 then sees a contiguous run of tiles
// instead of every eighth one (operands shared by neighbouring tiles are fetched from the fabric once, not once per XCD)
__device__ __forceinline__ void xcd_tile(int& x, int& y, int& z) {
  const int gx = gridDim.x, gy = gridDim.y;
  const int lin = xcd_order(blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z), gx * gy * gridDim.z);
  x = lin % gx; y = (lin / gx) % gy; z = lin / (gx * gy);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// LayerNorm row arithmetic shared by ln_fwd_kernel (elementwise.hip) and the fused residual-GEMM + LayerNorm kernel (gemm.hip): one
// wave per row, a lane holds VPL float4 chunks.  Contraction is OFF in here so that both kernels form exactly the same roundings
// whatever surrounds the call (the fused kernel's outputs are asserted bit-identical to the two-launch path).
template <int VPL>
__device__ __forceinline__ void ln_row_stats(const float4 (&v)[VPL], const bool (&ok)[VPL], int D, float eps, float& mean, float& rstd) {
#pragma clang fp contract(off)
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    if (ok[i]) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
      q += (a * a + b * b) + (c * c + e * e);
    }
  }
  const float var = wave_sum(q) / (float)D;
  rstd = 1.0f / sqrtf(var + eps);
}
__device__ __forceinline__ float4 ln_row_apply(const float4& v, float mean, float rstd, const float4& g, const float4& b) {
#pragma clang fp contract(off)
  float4 o;
  o.x = (v.x - mean) * rstd * g.x + b.x;
  o.y = (v.y - mean) * rstd * g.y + b.y;
  o.z = (v.z - mean) * rstd * g.z + b.z;
  o.w = (v.w - mean) * rstd * g.w + b.w;
  return o;
}

// v_mfma_f32_32x32x16_bf16: D[32x32] += A[32x16] * B[16x32].
//   A operand: lane l holds A[row = l & 31][k = 8*(l >> 5) + 0..7]
//   B operand: lane l holds B[k = 8*(l >> 5) + 0..7][col = l & 31]
//   C/D:       lane l, reg r holds D[row = (r & 3) + 8*(r >> 2) + 4*(l >> 5)][col = l & 31]
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// v_mfma_f32_32x32x16_f16 on operands that are STORED in bf16-typed arrays (same 2 bytes per element, same fragment layout): the
// single-pass "TF32-class" convolution mode (fp16 has the 11 significant bits TF32 has; fp32 accumulation).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma32_f16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// fp32 -> fp16 (round to nearest even, saturating at +-65504 instead of overflowing to inf), returned in a bf16-typed slot.
// NaN stays NaN (fmaxf(NaN, x) == x would turn a diverged tensor into -65504 and hide the divergence from every finite-loss check).
__device__ __forceinline__ bf16 f16_slot(float v) {
  const float c = fminf(fmaxf(v, -65504.f), 65504.f);
  return __builtin_bit_cast(bf16, (_Float16)(v != v ? v : c));
}
// the same with range bookkeeping: bit 0 of `ev` = |v| above fp16's largest finite value (saturated, +-inf included), bit 1 = a
// non-zero v that rounds to fp16 zero (|v| <= 2^-25).  Both are RARE in a healthy run, which is what lets the split kernels OR them
// per thread and add to the per-device counters with one global atomic per affected thread.  (Values below fp16's smallest normal
// 2^-14, which merely keep fewer than 11 significant bits, are NOT counted: they are common -- millions per step -- and one atomic
// per thread on a single address turned the 109 ms tf32class step into 183 ms when they were.)
__device__ __forceinline__ bf16 f16_slot_ev(float v, unsigned& ev) {
  const float a = fabsf(v);
  ev |= (a > 65504.f ? 1u : 0u) | ((a > 0.f && a <= 2.98023224e-8f) ? 2u : 0u);
  return f16_slot(v);
}
// row index inside a 32x32 accumulator tile held by (reg r, lane-half h = lane >> 5)
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ds_read_b64_tr_b16: within each 16-lane group the 16 lanes x 4 bf16 they address are transposed:
// result lane i (0..15), element j = the element (i & 3) of the 8-byte chunk addressed by lane 4*j + (i >> 2).
// With lane i' addressing &tile[k0 + (i' >> 2)][c0 + 4*(i' & 3)] every lane i receives column c0 + i of the
// 4(k) x 16(c) block: elements tile[k0 + 0..3][c0 + i].
__device__ __forceinline__ bf16x4 lds_tr_b64(const bf16* p) {
  s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
  return __builtin_bit_cast(bf16x4, t);
}
// LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane, lane-linear destination from a wave-uniform LDS byte address)
// issued through INLINE ASM, for kernels that read the staged tile with ds_read_b64_tr_b16.  Why not the builtin there:
// hipcc orders every LDS read that carries no alias-scope metadata -- and the transposed-read intrinsic never does --
// behind ALL pending LDS-DMA with an s_waitcnt vmcnt(0), i.e. the prefetch of tile t + 1 is waited for before the first
// fragment of tile t is read and nothing overlaps (seen in the ISA of the TN GEMM kernels; plain ds_read_b128 readers get
// alias scopes from the LDS lowering pass and are not affected).  The asm form is invisible to that bookkeeping, so the
// CALLER owns the protocol: lds_dma_wait_all() before the barrier that publishes a stage, and no read of a stage before
// that barrier.  (Untracked VMEM ops only make the compiler's own vmcnt waits more conservative: loads return in order.)
__device__ __forceinline__ uint32_t lds_byte_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void lds_dma16_untracked(const void* gsrc, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
               :: "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_base)) : "memory", "m0");
}
#pragma clang diagnostic pop
__device__ __forceinline__ void lds_dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ bf16x8 cat4(bf16x4 a, bf16x4 b) {
  bf16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
  r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return r;
}
__device__ __forceinline__ bf16x8 zero8() {
  bf16x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = (bf16)0.0f;
  return r;
}

// counter hash -> 32 random bits (dropout masks, sampler draws; deterministic in (seed, index), identical in forward and backward).
// Two multiply / xor-shift rounds over index ^ seed_lo, offset by seed_hi, with 24-BIT multiplies: v_mul_u32_u24 is full rate
// on CDNA while v_mul_lo_u32 is quarter rate -- with the 32-bit "lowbias32" finaliser the 32 multiplies per 32 x 64 score tile
// cost as many cycles (512) as the tile's 16 MFMAs.  The xor-shifts fold the bits a 24-bit multiply ignores back in.  Not a
// bijection (90 % distinct outputs over 2^22 consecutive indices) -- irrelevant for masks; keep rate, 256-bin chi-square of
// both 16-bit halves, serial / cross-seed correlation and avalanche (0.494 .. 0.516 over all 32 input bits) are on par with
// lowbias32 (numpy emulation, tests/test_host_cpu.py::test_dropout_threshold_and_hash_reference).
__device__ __forceinline__ uint32_t hash32(uint32_t x, uint32_t seed_lo, uint32_t seed_hi) {
  x = (x ^ seed_lo) + seed_hi;
  x ^= x >> 16;
  x = __umul24(x, 0x9E3779u);      // low 24 bits of both operands, low 32 bits of the product
  x ^= x >> 15;
  x = __umul24(x, 0x85EBCBu);
  x ^= x >> 16;
  return x;
}
// Dropout stream counter: a uint32 in DEVICE memory owned by the caller, passed to every entry point that takes a dropout
// seed.  Kernels add it to their seed at run time, so a step captured once in a hipGraph draws fresh masks on every
// replay (kernel arguments are frozen by capture, device memory is not).  NULL = disabled.
__device__ __forceinline__ uint32_t seed_mix(uint32_t seed_hi, const uint32_t* ctr) {
  return ctr ? seed_hi + *ctr * 0x9E3779B1u : seed_hi;
}
// dropout threshold on 16 random bits: keep iff r16 >= thr, thr = round(p * 65536)
__host__ __device__ inline uint32_t dropout_threshold(float p) {
  float t = p * 65536.0f + 0.5f;
  return t <= 0.f ? 0u : (t >= 65535.f ? 65535u : (uint32_t)t);
}

// tanh via one v_exp_f32 and one reciprocal: tanh(u) = 1 - 2 / (1 + e^(2u)); saturates correctly at +-inf.
// (tanhf() costs ~40 instructions per element and made the GELU epilogues as long as the GEMM main loop.)
__device__ __forceinline__ float fast_tanh(float u) {
  const float e = __builtin_amdgcn_exp2f(u * 2.885390081777927f);  // e^(2u) = 2^(2u log2 e)
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
}
// gelu_new(x) = 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3)  ==  x * sigmoid(2u): seven instructions
// (x^2, fma, mul, exp2, add, rcp, mul) instead of eleven -- the GELU epilogue is VALU-bound beside the MFMA stream.
//   -2u log2(e) = x * (GC1 + GC2 x^2)
constexpr float GELU_C1 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;             // -2 sqrt(2/pi) log2(e)
constexpr float GELU_C2 = GELU_C1 * 0.044715f;
__device__ __forceinline__ float gelu_new_f(float x) {
  const float t = x * x;
  const float e = __builtin_amdgcn_exp2f(x * fmaf(t, GELU_C2, GELU_C1));                  // e^(-2u); inf for very negative x
  return x * __builtin_amdgcn_rcpf(1.0f + e);                                               // x * sigmoid(2u); -> 0 as e -> inf
}
// d/dx gelu_new = s + x s (1 - s) d(2u)/dx with s = sigmoid(2u), d(2u)/dx = 2 sqrt(2/pi) (1 + 3 * 0.044715 x^2)
__device__ __forceinline__ float gelu_new_grad_f(float x) {
  const float t = x * x;
  const float e = __builtin_amdgcn_exp2f(x * fmaf(t, GELU_C2, GELU_C1));
  const float s = __builtin_amdgcn_rcpf(1.0f + e);
  const float dz = fmaf(t, 3.0f * 0.044715f * 2.0f * 0.7978845608028654f, 2.0f * 0.7978845608028654f);
  return s * fmaf((1.0f - s) * x, dz, 1.0f);
}

}  // namespace ttts
