// Stand-alone A/B harness for compile-time variants of the NT GEMM (no Python, no torch: a run costs seconds of GPU time).
// Every variant is a small shared library built from csrc/gemm.hip + csrc/lib.hip with its own -D flags
// (tools/ubench/nt_phase_build.sh); this program loads each, runs the GPT step's NT shapes through ttts_gemm_nt_bf16_ex, checks
// the output bit for bit against the first library's (the baseline) and prints launch times measured in interleaved rounds.
//   hipcc -O2 -o nt_phase nt_phase.cpp -ldl ;  ./nt_phase base.so v1.so v2.so ...
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <algorithm>
#include <vector>

typedef int (*nt_fn)(const void*, int64_t, const void*, int64_t, void*, int64_t, const float*, void*, int32_t, int32_t, int32_t,
                     int32_t, const float*, float, uint64_t, const uint32_t*, float*, void*);
typedef const char* (*err_fn)(void);
struct NtPlan { int32_t kernel, grid, block, tile_m, tile_n, phase, main_row_tiles, tail_tile_rows; };   // ttts_gemm_nt_plan
typedef int (*plan_fn)(int32_t, int32_t, int32_t, int32_t, NtPlan*);

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

static uint16_t bf16_of(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
static uint32_t rng = 12345u;
static float urand() {
  rng = rng * 1664525u + 1013904223u;
  return ((rng >> 8) & 0xFFFF) / 65536.0f - 0.5f;
}

struct Shape { const char* name; int M, N, K, epi; };

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s base.so [variant.so ...]\n", argv[0]); return 1; }
  const Shape shapes[] = {
      {"c_attn   9248x1536x512  store", 9248, 1536, 512, 0},
      {"c_fc     9248x2048x512  gelu ", 9248, 2048, 512, 1},
      {"d(c_proj)9248x2048x512  dgelu", 9248, 2048, 512, 3},
      {"         9248x2048x512  store", 9248, 2048, 512, 0},
      {"         9248x2048x512  resid", 9248, 2048, 512, 2},
      {"         9248x2048x512  f32  ", 9248, 2048, 512, 4},
      {"ragged   9001x2000x192  dgelu", 9001, 2000, 192, 3},
      {"ragged   9001x2000x192  resid", 9001, 2000, 192, 2},
      {"mel head 8208x1026x512  store", 8208, 1026, 512, 0},
      {"text head 1040x257x512  store", 1040, 257, 512, 0},
      {"d(mel head) 8208x512x1032 store", 8208, 512, 1032, 0},
      {"d(mel head) 8208x512x1088 store", 8208, 512, 1088, 0},
      {"d(text head) 1040x512x264 store", 1040, 512, 264, 0},
      {"d(text head) 1040x512x320 store", 1040, 512, 320, 0},
      {"big head 8192x8194x512  store", 8192, 8194, 512, 0},
      {"d(c_fc)  9248x512x2048  store", 9248, 512, 2048, 0},
      {"d(c_attn)9248x512x1536  store", 9248, 512, 1536, 0},
      {"attn c_proj 9248x512x512 resid", 9248, 512, 512, 2},
      {"d(attn c_proj) 9248x512x512 store", 9248, 512, 512, 0},
      {"c_proj   9248x512x2048  resid", 9248, 512, 2048, 2},
  };
  const int NS = sizeof(shapes) / sizeof(shapes[0]);
  const int Mx = 9248, Nx = 8200, Kx = 2048;
  std::vector<uint16_t> hA((size_t)Mx * Kx), hB((size_t)Nx * Kx);
  for (auto& v : hA) v = bf16_of(urand());
  for (auto& v : hB) v = bf16_of(urand() * 0.1f);
  std::vector<float> hbias(Nx);
  for (auto& v : hbias) v = urand();
  void *dA, *dB, *dC, *dAux;
  float* dbias;
  const size_t cbytes = (size_t)Mx * Nx * 4;
  HIP(hipMalloc(&dA, hA.size() * 2));
  HIP(hipMalloc(&dB, hB.size() * 2));
  HIP(hipMalloc(&dC, cbytes));
  HIP(hipMalloc(&dAux, (size_t)Mx * 2048 * 2));
  HIP(hipMalloc(&dbias, Nx * 4));
  HIP(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
  HIP(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
  HIP(hipMemcpy(dbias, hbias.data(), Nx * 4, hipMemcpyHostToDevice));
  HIP(hipMemcpy(dAux, hA.data(), (size_t)Mx * 2048 * 2, hipMemcpyHostToDevice));
  hipStream_t st;
  HIP(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  HIP(hipEventCreate(&e0));
  HIP(hipEventCreate(&e1));
  std::vector<unsigned char> hout(cbytes);

  // all libraries first; then per shape: one correctness launch each, and ROUNDS interleaved timing passes (library order
  // rotates inside a round, so clock ramps and thermal drift hit every variant alike); min and median over the rounds
  struct Lib { const char* path; nt_fn nt; err_fn last; plan_fn plan; };
  std::vector<Lib> libs;
  for (int li = 1; li < argc; ++li) {
    void* h = dlopen(argv[li], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", argv[li], dlerror()); return 2; }
    Lib l{argv[li], (nt_fn)dlsym(h, "ttts_gemm_nt_bf16_ex"), (err_fn)dlsym(h, "ttts_last_error"),
          (plan_fn)dlsym(h, "ttts_gemm_nt_plan_query")};   // (libraries older than ABI v6 have no plan query)
    if (!l.nt || !l.last) { fprintf(stderr, "%s: symbols missing\n", argv[li]); return 2; }
    libs.push_back(l);
  }
  const int ROUNDS = 7, IT = 40, NL = (int)libs.size();
  std::vector<unsigned char> ref;
  float* dcs;
  HIP(hipMalloc(&dcs, Nx * 4));
  for (int si = 0; si < NS; ++si) {
    const Shape& s = shapes[si];
    const int ldc = (s.N + 7) / 8 * 8;
    const size_t out_b = (size_t)s.M * ldc * ((s.epi == 2 || s.epi == 4) ? 4 : 2);
    auto call = [&](const Lib& l) {
      int rc = l.nt(dA, s.K, dB, s.K, dC, ldc, dbias, dAux, s.M, s.N, s.K, s.epi, nullptr, 0.f, 0, nullptr, nullptr, st);
      if (rc) { fprintf(stderr, "gemm_nt rc %d: %s\n", rc, l.last()); exit(3); }
    };
    std::vector<int> same(NL, 1);
    std::vector<float> cs_ref;
    std::vector<double> cs_dev(NL, 0.0);
    for (int li = 0; li < NL; ++li) {            // one launch on a zeroed C (the resid-add epilogue accumulates)
      HIP(hipMemsetAsync(dC, 0, out_b, st));
      call(libs[li]);
      HIP(hipStreamSynchronize(st));
      HIP(hipMemcpy(hout.data(), dC, out_b, hipMemcpyDeviceToHost));
      if (li == 0) ref.assign(hout.begin(), hout.begin() + out_b);
      else same[li] = memcmp(ref.data(), hout.data(), out_b) == 0;
      if (s.epi == 0 || s.epi == 3) {           // column sums taken in the epilogue (fp32 atomics: compared with a tolerance)
        HIP(hipMemsetAsync(dcs, 0, Nx * 4, st));
        int rc = libs[li].nt(dA, s.K, dB, s.K, dC, ldc, dbias, dAux, s.M, s.N, s.K, s.epi, nullptr, 0.f, 0, nullptr, dcs, st);
        if (rc) { fprintf(stderr, "gemm_nt colsum rc %d: %s\n", rc, libs[li].last()); exit(3); }
        HIP(hipStreamSynchronize(st));
        std::vector<float> cs(s.N);
        HIP(hipMemcpy(cs.data(), dcs, s.N * 4, hipMemcpyDeviceToHost));
        if (li == 0) cs_ref = cs;
        else {
          double worst = 0, scale = 1e-30;
          for (int n = 0; n < s.N; ++n) { worst = std::max(worst, (double)fabsf(cs[n] - cs_ref[n])); scale = std::max(scale, (double)fabsf(cs_ref[n])); }
          if (worst > 1e-4 * scale) same[li] = 0;
          cs_dev[li] = worst / scale;
        }
      }
    }
    std::vector<std::vector<float>> t(NL);
    for (int r = 0; r < ROUNDS + 1; ++r)
      for (int k = 0; k < NL; ++k) {
        const int li = (k + r) % NL;
        HIP(hipEventRecord(e0, st));
        for (int i = 0; i < IT; ++i) call(libs[li]);
        HIP(hipEventRecord(e1, st));
        HIP(hipEventSynchronize(e1));
        float ms;
        HIP(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) t[li].push_back(ms * 1e3f / IT);
      }
    printf("== %s\n", s.name);
    for (int li = 0; li < NL; ++li) {
      std::sort(t[li].begin(), t[li].end());
      const double mn = t[li][0], med = t[li][ROUNDS / 2], base = t[0][ROUNDS / 2];
      char kern[64] = "";
      NtPlan pl;
      static const char* kname[] = {"reg", "dma64", "dma32", "ring160", "wave8", "wave8-split", "wreg"};
      if (libs[li].plan && libs[li].plan(s.M, s.N, s.K, s.epi, &pl) == 0)
        snprintf(kern, sizeof kern, "  [%s x%d%s]", kname[pl.kernel], pl.grid, pl.phase ? " staggered" : "");
      printf("  %-44s min %7.2f  median %7.2f us  %6.1f TF/s  %+5.1f %%  %s%s\n", libs[li].path, mn, med,
             2.0 * s.M * s.N * s.K / med * 1e-6, (med / base - 1.0) * 100.0, same[li] ? "bit-identical" : "OUTPUT DIFFERS", kern);
      if (cs_dev[li] > 0) printf("      (column sums: max deviation %.2e of the largest)\n", cs_dev[li]);
    }
    fflush(stdout);
  }
  // chains: a GEMM whose output is the next GEMM's A operand, timed as a pair (does the producer's store policy pay twice?)
  struct Chain { const char* name; Shape a, b; };
  const Chain chains[] = {
      {"c_fc GELU -> mlp c_proj resid", {"", 9248, 2048, 512, 1}, {"", 9248, 512, 2048, 2}},
      {"dGELU -> dX of c_fc", {"", 9248, 2048, 512, 3}, {"", 9248, 512, 2048, 0}},
      {"c_attn -> (its output re-read as an A operand)", {"", 9248, 1536, 512, 0}, {"", 9248, 512, 1536, 0}},
  };
  void* dC2;
  HIP(hipMalloc(&dC2, (size_t)Mx * 512 * 4));
  for (const Chain& c : chains) {
    std::vector<std::vector<float>> t(NL);
    for (int r = 0; r < ROUNDS + 1; ++r)
      for (int k = 0; k < NL; ++k) {
        const Lib& l = libs[(k + r) % NL];
        HIP(hipEventRecord(e0, st));
        for (int i = 0; i < IT; ++i) {
          int rc = l.nt(dA, c.a.K, dB, c.a.K, dC, c.a.N, dbias, dAux, c.a.M, c.a.N, c.a.K, c.a.epi, nullptr, 0.f, 0, nullptr, nullptr, st);
          rc |= l.nt(dC, c.b.K, dB, c.b.K, dC2, c.b.N, dbias, nullptr, c.b.M, c.b.N, c.b.K, c.b.epi, nullptr, 0.f, 0, nullptr, nullptr, st);
          if (rc) { fprintf(stderr, "chain rc %d: %s\n", rc, l.last()); exit(3); }
        }
        HIP(hipEventRecord(e1, st));
        HIP(hipEventSynchronize(e1));
        float ms;
        HIP(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) t[(k + r) % NL].push_back(ms * 1e3f / IT);
      }
    printf("== chain: %s\n", c.name);
    for (int li = 0; li < NL; ++li) {
      std::sort(t[li].begin(), t[li].end());
      printf("  %-44s min %7.2f  median %7.2f us  %+5.1f %%\n", libs[li].path, t[li][0], t[li][ROUNDS / 2],
             (t[li][ROUNDS / 2] / t[0][ROUNDS / 2] - 1.0) * 100.0);
    }
    fflush(stdout);
  }
  return 0;
}
