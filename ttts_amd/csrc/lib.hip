// Library glue: error buffer, version / device query, and the hardware-layout probe used by the GPU tests.
#include "common.hpp"

namespace ttts {
char* error_buffer() {
  static thread_local char buf[512] = "";
  return buf;
}

// Probe: dumps (a) the raw accumulators of one v_mfma_f32_32x32x16_bf16 fed with a row-tagged A and a
// column-tagged B, (b) what ds_read_b64_tr_b16 returns for the address pattern the kernels use.
__global__ __launch_bounds__(64) void probe_kernel(float* out_c, int* out_tr) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[16 * 64];
  const int lane = threadIdx.x, hh = lane >> 5, g = lane >> 4, ip = lane & 15;
  for (int i = lane; i < 16 * 64; i += 64) tile[i] = (unsigned short)i;
  __syncthreads();
  bf16x8 a = zero8(), b = zero8();
  if (hh == 0) {
    a[0] = (bf16)(float)((lane & 31) + 1);  // k-slot 0: A row tag
    a[1] = (bf16)1.0f;                      // k-slot 1
    b[0] = (bf16)1.0f;
    b[1] = (bf16)(float)(64 * ((lane & 31) + 1));  // B column tag
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = mfma32(a, b, acc);  // D[i][j] = (i + 1) + 64 (j + 1)
#pragma unroll
  for (int r = 0; r < 16; ++r) out_c[lane * 16 + r] = acc[r];
  const int off = (8 * (g >> 1) + (ip >> 2)) * 64 + 16 * (g & 1) + 4 * (ip & 3);
  const bf16* base = reinterpret_cast<const bf16*>(tile);
  const bf16x4 t0 = lds_tr_b64(base + off);
  const bf16x4 t1 = lds_tr_b64(base + off + 4 * 64);
  const s16x4 s0 = __builtin_bit_cast(s16x4, t0), s1 = __builtin_bit_cast(s16x4, t1);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    out_tr[lane * 8 + j] = (int)(unsigned short)s0[j];
    out_tr[lane * 8 + 4 + j] = (int)(unsigned short)s1[j];
  }
}
}  // namespace ttts

using namespace ttts;

extern "C" int ttts_abi_version(void) { return TTTS_ABI_VERSION; }
extern "C" const char* ttts_last_error(void) { return error_buffer(); }

extern "C" int ttts_device_info(int32_t out[4]) {
  TTTS_REQUIRE(out, "device_info: null pointer");
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return fail(TTTS_EHIP, "hipGetDevice: %s", hipGetErrorString(e));
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) return fail(TTTS_EHIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
  int arch = 0;
  sscanf(prop.gcnArchName, "gfx%d", &arch);
  out[0] = arch;
  out[1] = prop.multiProcessorCount;
  out[2] = prop.warpSize;
  out[3] = (int32_t)prop.maxSharedMemoryPerMultiProcessor;
  return TTTS_OK;
}

extern "C" int ttts_probe_mfma_layout(float* out_c, int32_t* out_tr, void* stream) {
  TTTS_REQUIRE(out_c && out_tr, "probe: null pointer");
  probe_kernel<<<1, 64, 0, as_stream(stream)>>>(out_c, out_tr);
  return check_launch("probe");
}
