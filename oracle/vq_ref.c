/* ORACLE (test infrastructure, NOT product code) -- plain-C restatement of the reference VQ nearest-code search
 * EuclideanCodebook.quantize, /root/reference/ttts/vqvae/core_vq.py:174-182:
 *     dist = -(x.pow(2).sum(1, keepdim=True) - 2 * x @ embed + embed.pow(2).sum(0, keepdim=True))
 *     embed_ind = dist.max(dim=-1).indices            (first maximum: ties -> lowest index)
 * with the one thing the reference leaves unspecified -- the fp32 summation order of the contraction (BLAS sgemm) --
 * pinned to a k-ordered fmaf chain, which is what the f32-input MFMA of the HIP kernel computes bit-for-bit.
 * Used by tests/ (and smoke) as the bit-exact checker for indices AND distances; pinned against the reference's own
 * outputs through tests/golden/vq.npz (identical indices on all fixtures, no near-ties).
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC oracle/vq_ref.c -o oracle/_build/libvq_ref.so -lm
 * (done by __graft_entry__.build()).
 */
#include <math.h>
#include <stdint.h>

void vq_nearest_ref(const float* x, const float* cb, int64_t N, int64_t K, int64_t D, int64_t* idx, float* best_dist) {
  for (int64_t n = 0; n < N; ++n) {
    const float* xr = x + n * D;
    float x2 = 0.f;
    for (int64_t d = 0; d < D; ++d) x2 = fmaf(xr[d], xr[d], x2);
    float best = -INFINITY;
    int64_t bi = 0;
    for (int64_t k = 0; k < K; ++k) {
      const float* e = cb + k * D;
      float e2 = 0.f, dot = 0.f;
      for (int64_t d = 0; d < D; ++d) e2 = fmaf(e[d], e[d], e2);
      for (int64_t d = 0; d < D; ++d) dot = fmaf(e[d], xr[d], dot);
      const float dist = -((x2 - 2.0f * dot) + e2);
      if (dist > best) { best = dist; bi = k; }
    }
    idx[n] = bi;
    if (best_dist) best_dist[n] = best;
  }
}
