// Stand-alone A/B harness for builds of the VQ-VAE-GAN convolution family (csrc/conv*.hip), the sibling of nt_phase.cpp and
// attn_variants.cpp: every library argument is a build of the conv sources (tools/ubench/conv_variants_build.sh); the program
// replays the call signatures of tools/ubench/conv_cases.txt (the heaviest signatures of one train step, from tools/conv_census.py
// via conv_cases.py) through ttts_conv1d_fwd_f32 / _dgrad_f32 / _wgrad_f32 with the 1.5 GB scratch the product passes, compares
// every output with the first library's (bit for bit; a relative deviation is printed otherwise -- some weight-gradient kernels
// add with atomics) and prints per-signature times from interleaved rounds plus the count-weighted sum per library.
//   ./conv_variants cases.txt base.so [variant.so ...]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct ConvCtx { void* workspace; int64_t workspace_bytes; int32_t flags, reserved; };   // include/ttts_hip.h ttts_conv_ctx
typedef int (*fwd_fn)(const float*, const float*, const float*, const float*, const float*, const float*, const float*, float*, int32_t,
                      int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, float, float, int32_t, float,
                      float, int32_t, const ConvCtx*, void*);
typedef int (*dgrad_fn)(const float*, const float*, const float*, const float*, const float*, const float*, float*, int32_t, int32_t,
                        int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, float, float, float, int32_t,
                        const ConvCtx*, void*);
typedef int (*wgrad_fn)(const float*, const float*, float*, float*, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t,
                        int32_t, int32_t, int32_t, float, float, const ConvCtx*, void*);
typedef const char* (*err_fn)(void);

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

struct Lib { const char* path; fwd_fn fwd; dgrad_fn dgrad; wgrad_fn wgrad; err_fn last; };
struct Case {
  char kind;            // 'f', 'd', 'w'
  int count, B, Cin, Lin, Cout, Lout, K, stride, pad, dil, groups, gate, act;
  float s0, s1;         // fwd / dgrad: in_slope, -; wgrad: x_slope, dy_slope
  std::string text;
};

__global__ void fill_kernel(float* p, size_t n, uint32_t seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    p[i] = ((h >> 8) / 16777216.0f - 0.5f) * 2.0f * scale;
  }
}
static float* dev_rand(size_t n, uint32_t seed, float scale, hipStream_t st) {
  float* p;
  HIP(hipMalloc(&p, std::max<size_t>(n, 1) * 4));
  fill_kernel<<<1024, 256, 0, st>>>(p, n, seed, scale);
  return p;
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s cases.txt base.so [variant.so ...]\n", argv[0]); return 1; }
  std::vector<Case> cases;
  {
    FILE* f = fopen(argv[1], "r");
    if (!f) { perror(argv[1]); return 1; }
    char line[512];
    while (fgets(line, sizeof line, f)) {
      if (line[0] == '#' || line[0] == '\n') continue;
      Case c{};
      char kind[16];
      int n = 0;
      c.text = line;
      c.text.erase(c.text.find_last_not_of("\n") + 1);
      if (sscanf(line, "%15s", kind) != 1) continue;
      if (!strcmp(kind, "fwd")) {
        c.kind = 'f';
        n = sscanf(line, "%*s %d %d %d %d %d %d %d %d %d %d %f %d %d", &c.count, &c.B, &c.Cin, &c.Lin, &c.Cout, &c.K, &c.stride, &c.pad,
                   &c.dil, &c.groups, &c.s0, &c.gate, &c.act);
        c.Lout = (c.Lin + 2 * c.pad - c.dil * (c.K - 1) - 1) / c.stride + 1;
        if (n != 13) { fprintf(stderr, "bad line: %s", line); return 1; }
      } else if (!strcmp(kind, "dgrad")) {
        c.kind = 'd';
        n = sscanf(line, "%*s %d %d %d %d %d %d %d %d %d %d %d %f %d", &c.count, &c.B, &c.Cout, &c.Lout, &c.Cin, &c.Lin, &c.K, &c.stride,
                   &c.pad, &c.dil, &c.groups, &c.s0, &c.gate);
        if (n != 13) { fprintf(stderr, "bad line: %s", line); return 1; }
      } else if (!strcmp(kind, "wgrad")) {
        c.kind = 'w';
        n = sscanf(line, "%*s %d %d %d %d %d %d %d %d %d %d %d %f %f", &c.count, &c.B, &c.Cout, &c.Lout, &c.Cin, &c.Lin, &c.K, &c.stride,
                   &c.pad, &c.dil, &c.groups, &c.s0, &c.s1);
        if (n != 13) { fprintf(stderr, "bad line: %s", line); return 1; }
      } else {
        continue;
      }
      cases.push_back(c);
    }
    fclose(f);
  }
  std::vector<Lib> libs;
  for (int li = 2; li < argc; ++li) {
    void* h = dlopen(argv[li], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", argv[li], dlerror()); return 2; }
    Lib l{argv[li], (fwd_fn)dlsym(h, "ttts_conv1d_fwd_f32"), (dgrad_fn)dlsym(h, "ttts_conv1d_dgrad_f32"),
          (wgrad_fn)dlsym(h, "ttts_conv1d_wgrad_f32"), (err_fn)dlsym(h, "ttts_last_error")};
    if (!l.fwd || !l.dgrad || !l.wgrad || !l.last) { fprintf(stderr, "%s: symbols missing\n", argv[li]); return 2; }
    libs.push_back(l);
  }
  const int NL = (int)libs.size();
  hipStream_t st;
  HIP(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  HIP(hipEventCreate(&e0));
  HIP(hipEventCreate(&e1));
  ConvCtx ctx{nullptr, (int64_t)1536 << 20, 0, 0};
  HIP(hipMalloc(&ctx.workspace, (size_t)ctx.workspace_bytes));
  const char* fl = getenv("CONV_FLAGS");
  if (fl) ctx.flags = atoi(fl);

  const int ROUNDS = 5, IT = 3;
  std::vector<double> total(NL, 0.0);
  int differ = 0;
  for (size_t ci = 0; ci < cases.size(); ++ci) {
    const Case& c = cases[ci];
    const int cpg = c.Cin / c.groups;
    const size_t nx = (size_t)c.B * c.Cin * c.Lin, ny = (size_t)c.B * c.Cout * c.Lout, nw = (size_t)c.Cout * cpg * c.K;
    float *x = nullptr, *w = nullptr, *y = nullptr, *gate = nullptr, *out = nullptr;
    size_t nout = 0;
    if (c.kind == 'f') {
      x = dev_rand(nx, 1, 1.0f, st); w = dev_rand(nw, 2, 0.05f, st); nout = ny;
      if (c.gate) gate = dev_rand(ny, 3, 1.0f, st);
    } else if (c.kind == 'd') {
      y = dev_rand(ny, 4, 1.0f, st); w = dev_rand(nw, 5, 0.05f, st); nout = nx;
      if (c.gate) gate = dev_rand(nx, 6, 1.0f, st);
    } else {
      y = dev_rand(ny, 7, 1.0f, st); x = dev_rand(nx, 8, 1.0f, st); nout = nw;
    }
    HIP(hipMalloc(&out, nout * 4));
    auto call = [&](const Lib& l) {
      int rc;
      if (c.kind == 'f')
        rc = l.fwd(x, w, nullptr, nullptr, nullptr, gate, nullptr, out, c.B, c.Cin, c.Lin, c.Cout, c.Lout, c.K, c.stride, c.pad, c.dil,
                   c.groups, c.s0, 0.1f, c.act, 0.1f, 1.0f, 0, &ctx, st);
      else if (c.kind == 'd')
        rc = l.dgrad(y, w, nullptr, nullptr, gate, nullptr, out, c.B, c.Cin, c.Lin, c.Cout, c.Lout, c.K, c.stride, c.pad, c.dil, c.groups,
                     c.s0, 0.1f, 1.0f, 0, &ctx, st);
      else
        rc = l.wgrad(y, x, out, nullptr, c.B, c.Cin, c.Lin, c.Cout, c.Lout, c.K, c.stride, c.pad, c.dil, c.groups, c.s1, c.s0, &ctx, st);
      if (rc) { fprintf(stderr, "%s\n  rc %d: %s\n", c.text.c_str(), rc, l.last()); exit(3); }
    };
    std::vector<float> ref, cur(nout);
    std::vector<double> dev(NL, 0.0);
    for (int li = 0; li < NL; ++li) {
      HIP(hipMemsetAsync(out, 0, nout * 4, st));     // the weight gradient accumulates
      call(libs[li]);
      HIP(hipStreamSynchronize(st));
      HIP(hipMemcpy(cur.data(), out, nout * 4, hipMemcpyDeviceToHost));
      if (li == 0) ref = cur;
      else if (memcmp(ref.data(), cur.data(), nout * 4)) {
        double num = 0, den = 1e-30;
        for (size_t i = 0; i < nout; ++i) { const double d = (double)cur[i] - ref[i]; num += d * d; den += (double)ref[i] * ref[i]; }
        dev[li] = std::max(std::sqrt(num / den), 1e-12);
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
    printf("%-60s", c.text.c_str());
    for (int li = 0; li < NL; ++li) {
      std::sort(t[li].begin(), t[li].end());
      const double med = t[li][ROUNDS / 2];
      total[li] += med * c.count;
      if (li == 0) printf(" %8.1f us", med);
      else {
        printf(" | %8.1f %+5.1f%%", med, (med / t[0][ROUNDS / 2] - 1) * 100);
        if (dev[li] > 0) { printf(" (rel %.1e)", dev[li]); if (dev[li] > 1e-5) ++differ; }
      }
    }
    printf("\n");
    fflush(stdout);
    for (float* p : {x, w, y, gate, out})
      if (p) HIP(hipFree(p));
  }
  printf("\ncount-weighted sum over %zu signatures (ms of one step):\n", cases.size());
  for (int li = 0; li < NL; ++li) printf("  %-44s %8.2f ms  %+5.1f %%\n", libs[li].path, total[li] / 1e3, (total[li] / total[0] - 1) * 100);
  if (differ) printf("!! %d outputs deviate from the baseline by more than 1e-5 relative\n", differ);
  return 0;
}
