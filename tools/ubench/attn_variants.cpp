// Stand-alone A/B harness for builds of the causal self-attention kernels (csrc/attn.hip + csrc/attn_dh64.hip), the sibling of
// nt_phase.cpp: every argument is a small shared library (tools/ubench/attn_variants_build.sh); the program runs the GPT step's
// attention shape (B 8, H 8, S 1156, head_dim 64, packed [B, S, 3 H dh] q/k/v, dropout 0.1) and two corner shapes through
// ttts_attn_causal_fwd_bf16 / ttts_attn_causal_bwd_bf16, compares O, lse, dQ, dK, dV with the first library's bit for bit (the
// kernels are deterministic: no atomics) and prints forward / backward launch times from interleaved rounds.
//   ./attn_variants base.so [variant.so ...]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int (*fwd_fn)(const void*, const void*, const void*, void*, float*, int32_t, int32_t, int32_t, int32_t, int64_t, int64_t,
                      int64_t, int64_t, float, float, uint64_t, const uint32_t*, void*);
typedef int (*bwd_fn)(const void*, const void*, const void*, const void*, const void*, const float*, void*, void*, void*, void*,
                      int32_t, int32_t, int32_t, int32_t, int64_t, int64_t, int64_t, int64_t, float, float, uint64_t,
                      const uint32_t*, void*);
typedef int64_t (*ws_fn)(int32_t, int32_t, int32_t);
typedef const char* (*err_fn)(void);

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

static uint16_t bf16_of(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
static uint32_t rng = 2463534242u;
static float urand() {
  rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5;
  return (rng >> 8) / 16777216.0f - 0.5f;
}

struct Shape { const char* name; int B, H, S; float p; };
struct Lib { const char* path; fwd_fn fwd; bwd_fn bwd; ws_fn ws; err_fn last; };

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s base.so [variant.so ...]\n", argv[0]); return 1; }
  const Shape shapes[] = {
      {"GPT step  B 8 H 8 S 1156 dropout 0.1", 8, 8, 1156, 0.1f},
      {"ragged    B 2 H 8 S 1000 dropout 0.1", 2, 8, 1000, 0.1f},
      {"no mask   B 8 H 8 S 1156 dropout 0  ", 8, 8, 1156, 0.0f},
  };
  const int DH = 64, Bx = 8, Hx = 8, Sx = 1156, D = Hx * DH;
  std::vector<Lib> libs;
  for (int li = 1; li < argc; ++li) {
    void* h = dlopen(argv[li], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", argv[li], dlerror()); return 2; }
    Lib l{argv[li], (fwd_fn)dlsym(h, "ttts_attn_causal_fwd_bf16"), (bwd_fn)dlsym(h, "ttts_attn_causal_bwd_bf16"),
          (ws_fn)dlsym(h, "ttts_attn_bwd_workspace_bytes"), (err_fn)dlsym(h, "ttts_last_error")};
    if (!l.fwd || !l.bwd || !l.ws || !l.last) { fprintf(stderr, "%s: symbols missing\n", argv[li]); return 2; }
    libs.push_back(l);
  }
  const int NL = (int)libs.size();
  const size_t n_qkv = (size_t)Bx * Sx * 3 * D, n_o = (size_t)Bx * Sx * D, n_lse = (size_t)Bx * Hx * Sx;
  std::vector<uint16_t> hq(n_qkv), hdo(n_o);
  for (auto& v : hq) v = bf16_of(urand() * 2.0f);
  for (auto& v : hdo) v = bf16_of(urand());
  void *qkv, *o, *d_o, *dqkv, *ws;
  float* lse;
  uint32_t* ctr;
  HIP(hipMalloc(&qkv, n_qkv * 2));
  HIP(hipMalloc(&dqkv, n_qkv * 2));
  HIP(hipMalloc(&o, n_o * 2));
  HIP(hipMalloc(&d_o, n_o * 2));
  HIP(hipMalloc(&lse, n_lse * 4));
  HIP(hipMalloc(&ctr, 4));
  HIP(hipMemset(ctr, 0, 4));
  HIP(hipMalloc(&ws, (size_t)libs[0].ws(Bx, Hx, Sx) + 256));
  HIP(hipMemcpy(qkv, hq.data(), n_qkv * 2, hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_o, hdo.data(), n_o * 2, hipMemcpyHostToDevice));
  hipStream_t st;
  HIP(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  HIP(hipEventCreate(&e0));
  HIP(hipEventCreate(&e1));
  const uint64_t seed = 0x1234567887654321ull;
  const float scale = 0.125f;

  const int ROUNDS = 7, IT = 20;
  for (const Shape& s : shapes) {
    const int64_t ss = 3 * D, sb = (int64_t)s.S * ss, oss = D, osb = (int64_t)s.S * oss;
    const char* q = (const char*)qkv;
    auto fwd = [&](const Lib& l) {
      int rc = l.fwd(q, q + D * 2, q + 2 * D * 2, o, lse, s.B, s.H, s.S, DH, sb, ss, osb, oss, scale, s.p, seed, ctr, st);
      if (rc) { fprintf(stderr, "attn fwd rc %d: %s\n", rc, l.last()); exit(3); }
    };
    auto bwd = [&](const Lib& l) {
      char* g = (char*)dqkv;
      int rc = l.bwd(q, q + D * 2, q + 2 * D * 2, o, d_o, lse, g, g + D * 2, g + 2 * D * 2, ws, s.B, s.H, s.S, DH, sb, ss, osb, oss,
                     scale, s.p, seed, ctr, st);
      if (rc) { fprintf(stderr, "attn bwd rc %d: %s\n", rc, l.last()); exit(3); }
    };
    const size_t b_o = (size_t)s.B * s.S * D * 2, b_lse = (size_t)s.B * s.H * s.S * 4, b_g = (size_t)s.B * s.S * 3 * D * 2;
    std::vector<unsigned char> ref_o, ref_lse, ref_g, cur(std::max(b_g, b_o));
    std::vector<int> same(NL, 1);
    for (int li = 0; li < NL; ++li) {
      HIP(hipMemsetAsync(o, 0, b_o, st));
      HIP(hipMemsetAsync(lse, 0, b_lse, st));
      HIP(hipMemsetAsync(dqkv, 0, b_g, st));
      fwd(libs[li]);
      bwd(libs[li]);
      HIP(hipStreamSynchronize(st));
      auto fetch = [&](void* src, size_t n, std::vector<unsigned char>& ref) {
        HIP(hipMemcpy(cur.data(), src, n, hipMemcpyDeviceToHost));
        if (li == 0) ref.assign(cur.begin(), cur.begin() + n);
        else if (memcmp(ref.data(), cur.data(), n)) same[li] = 0;
      };
      fetch(o, b_o, ref_o);
      fetch(lse, b_lse, ref_lse);
      fetch(dqkv, b_g, ref_g);
    }
    // leave the baseline's forward outputs in place for the timed backward launches
    fwd(libs[0]);
    std::vector<std::vector<float>> tf(NL), tb(NL);
    for (int r = 0; r < ROUNDS + 1; ++r)
      for (int k = 0; k < NL; ++k) {
        const int li = (k + r) % NL;
        float ms;
        HIP(hipEventRecord(e0, st));
        for (int i = 0; i < IT; ++i) fwd(libs[li]);
        HIP(hipEventRecord(e1, st));
        HIP(hipEventSynchronize(e1));
        HIP(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) tf[li].push_back(ms * 1e3f / IT);
        HIP(hipEventRecord(e0, st));
        for (int i = 0; i < IT; ++i) bwd(libs[li]);
        HIP(hipEventRecord(e1, st));
        HIP(hipEventSynchronize(e1));
        HIP(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) tb[li].push_back(ms * 1e3f / IT);
      }
    printf("== %s\n", s.name);
    const double pairs = (double)s.B * s.H * DH * s.S * (s.S + 1) / 2;     // causal-useful multiply-adds per matmul
    for (int li = 0; li < NL; ++li) {
      std::sort(tf[li].begin(), tf[li].end());
      std::sort(tb[li].begin(), tb[li].end());
      const double f = tf[li][ROUNDS / 2], b = tb[li][ROUNDS / 2], f0 = tf[0][ROUNDS / 2], b0 = tb[0][ROUNDS / 2];
      printf("  %-40s fwd %7.2f us (%5.1f TF/s, %+5.1f %%)   bwd %7.2f us (%5.1f TF/s, %+5.1f %%)   %s\n", libs[li].path, f,
             4.0 * pairs / f * 1e-6, (f / f0 - 1) * 100, b, 10.0 * pairs / b * 1e-6, (b / b0 - 1) * 100,
             same[li] ? "bit-identical" : "OUTPUT DIFFERS");
    }
    fflush(stdout);
  }
  return 0;
}
