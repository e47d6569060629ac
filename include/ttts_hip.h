/*
 * ttts_hip.h -- C ABI of libttts_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * training hot path of adelacvg/ttts (GPT train step, VQ codebook, mel/STFT).
 *
 * The reference (/root/reference) is 100 % Python and has NO plugin / operator / FFI layer
 * (SURVEY.md F1, section 8b2): each entry point below replaces a *PyTorch op sequence* of the
 * reference, cited per function as `file:line` relative to /root/reference (or to the installed
 * `transformers` GPT-2 implementation the reference instantiates, ttts/gpt/model.py:245-265).
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *  - plain C: raw DEVICE pointers (tensor.data_ptr()), explicit sizes/strides in ELEMENTS, scalars by value;
 *    no torch / C++ types cross the boundary.
 *  - the CALLER owns all memory (inputs, outputs, workspaces; sizes from the *_workspace_bytes helpers);
 *    the library never allocates or frees device memory and keeps NO mutable global state (only a
 *    thread-local error string): dropout stream counters and the convolution scratch are explicit arguments.
 *  - launches are asynchronous on `stream` (a hipStream_t passed as void*; NULL = the default stream);
 *    no internal synchronisation; every kernel is hipGraph-capturable (no host reads of device data).
 *  - return value: TTTS_OK (0) or a negative error code; ttts_last_error() gives the message.
 *  - "bf16" = bfloat16 stored as uint16 bit patterns; "f32" = IEEE float; indices are int64 unless noted.
 *  - gradient outputs documented as "+=" ACCUMULATE into their buffer (zero it at the start of a step).
 */
#ifndef TTTS_HIP_H
#define TTTS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TTTS_OK 0
#define TTTS_EINVAL (-1)       /* bad shape / alignment / null pointer */
#define TTTS_EHIP (-2)         /* a HIP runtime call or kernel launch failed */
#define TTTS_EUNSUPPORTED (-3) /* valid request this build has no kernel for */

#define TTTS_ABI_VERSION 11

/* ---- library ------------------------------------------------------------------------------------ */
int ttts_abi_version(void);
const char* ttts_last_error(void);
/* Dropout stream counter (argument `dropout_counter` of every entry point that takes a dropout `seed`): a pointer to a
 * uint32 in DEVICE memory owned by the caller, or NULL.  The kernel adds it to `seed` at run time; incrementing it once
 * per step (on the stream) gives fresh masks on each replay of a captured hipGraph (kernel arguments are frozen by
 * capture, device memory is not).  Forward and backward of one step must see the same value.  The library itself holds
 * no implicit state: every entry point is re-entrant across threads and streams.  The only state kept between calls sits
 * behind handles the CALLER creates over its own storage (the convolutions' weight-split cache and slab arena, ABI v8); a
 * process-wide, mutex-guarded list of live handles is how convolution calls find them, by pointer range. */
/* Device query: writes {gfx arch number (950), CU count, wavefront size, LDS bytes/CU}. */
int ttts_device_info(int32_t out[4]);

/* ---- GEMM (bf16 operands, fp32 MFMA accumulation) ------------------------------------------------
 * Replaces: HF Conv1D addmm (transformers/pytorch_utils.py Conv1D.forward) at
 * modeling_gpt2.py:103-107,229-243, nn.Linear heads at ttts/gpt/model.py:348-349,432-438, and their
 * autograd backward (dX = dY W^T, dW = X^T dY, db = colsum dY) under autocast(bf16).
 */
enum {
  TTTS_EPI_STORE_BF16 = 0,      /* C = bf16(acc + bias)                                              */
  TTTS_EPI_GELU_BF16 = 1,       /* aux = bf16(acc + bias) (pre-activation), C = bf16(gelu_new(aux)) */
  TTTS_EPI_RESID_ADD_F32 = 2,   /* resid[m][n] += float(bf16(acc + bias))   (fp32 residual stream)  */
  TTTS_EPI_DGELU_BF16 = 3,      /* C = bf16(acc * gelu_new'(aux[m][n]))      (aux = saved pre-act)   */
  TTTS_EPI_STORE_F32 = 4        /* Cf = acc + bias (fp32 output)                                     */
};
/* C[M,N] = epilogue(A[M,K] . B[N,K]^T): both operands K-contiguous ("NT").  lda/ldb/ldc in elements,
 * K % 8 == 0, lda % 8 == 0, ldb % 8 == 0, ldc % 4 == 0, 16-byte aligned bases.
 * bias: f32[N] or NULL.  C: bf16 (or f32 for STORE_F32 / RESID_ADD_F32).  aux: bf16 [M, ldc] or NULL. */
int ttts_gemm_nt_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                      const float* bias, void* aux, int32_t M, int32_t N, int32_t K, int32_t epilogue,
                      void* stream);
/* Same, plus: resid_in (RESID_ADD_F32 only): C = resid_in + dropout(bf16(acc + bias)) out of place (NULL: C += ...);
 * dropout_p/seed: residual dropout (GPT-2 resid_pdrop, modeling_gpt2.py:223,241) on element index m*N + n;
 * colsum (ABI v5; STORE_BF16 / DGELU_BF16 only, or NULL): colsum[n] += sum_m C[m][n] of the bf16 output -- the bias gradient
 * of the layer whose dY this GEMM produces, taken in the epilogue instead of re-reading C (fp32 atomics, one per column and tile). */
int ttts_gemm_nt_bf16_ex(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                         const float* bias, void* aux, int32_t M, int32_t N, int32_t K, int32_t epilogue,
                         const float* resid_in, float dropout_p, uint64_t seed, const uint32_t* dropout_counter,
                         float* colsum, void* stream);
/* The residual GEMM AND the LayerNorm that consumes its output, one launch (ABI v7; N = 512 = whole rows of the GPT model width):
 *   x_out[M,N] = resid_in + dropout(bf16(A[M,K] . B[N,K]^T + bias))        (resid_in NULL: x_out += ..., as RESID_ADD_F32)
 *   y[M,N]     = LayerNorm(x_out; gamma, beta, eps)   (bf16 or f32),  mean[M], rstd[M]
 * Replaces GPT2Block's  `hidden = attn_out + residual; hidden = ln_2(hidden)` and `hidden = residual + mlp_out;` + the NEXT block's
 * ln_1 / the final ln_f (modeling_gpt2.py:229-309 via ttts/gpt/model.py:422).  Every output is bit-identical to
 * ttts_gemm_nt_bf16_ex(RESID_ADD_F32) followed by ttts_layernorm_fwd (same dropout stream, same summation order).
 * K % 64 == 0; N must be 512 (TTTS_EINVAL otherwise: use the two calls). */
int ttts_gemm_nt_resid_ln_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, const float* bias,
                               const float* resid_in, float* x_out, int32_t M, int32_t N, int32_t K, float dropout_p,
                               uint64_t seed, const uint32_t* dropout_counter, const float* gamma, const float* beta,
                               float eps, void* y, int32_t y_is_bf16, float* mean, float* rstd, void* stream);
/* Which kernel and grid ttts_gemm_nt_bf16(_ex) uses for a shape (ABI v6; host-side query, launches nothing, needs no GPU).
 * The kernels differ in tile shape and pipeline only -- a shape's output bits are the same whichever runs -- and the choice is a
 * table of measurements on the 256 CUs of an MI355X (csrc/gemm.hip plan_nt); tests pin the rules, benchmarks report them. */
enum {
  TTTS_NT_KERNEL_REG = 0,         /* 128 x 128, register-staged: any K % 8 == 0                                        */
  TTTS_NT_KERNEL_DMA64 = 1,       /* 128 x 128, LDS-DMA, 64-deep stages, 2 workgroups per CU                           */
  TTTS_NT_KERNEL_DMA32 = 2,       /* 128 x 128, LDS-DMA, 32-deep stages, 3 workgroups per CU                           */
  TTTS_NT_KERNEL_RING160 = 3,     /* 160 x 128, 4-slot LDS-DMA ring, 1 workgroup per CU (narrow N)                     */
  TTTS_NT_KERNEL_WAVE8 = 4,       /* 256 x 128, eight waves sharing each B stage, 2 workgroups per CU                  */
  TTTS_NT_KERNEL_WAVE8_SPLIT = 5, /* the same, grid = main_row_tiles rows of 256-row tiles + tail_tile_rows-row tiles  */
  TTTS_NT_KERNEL_WREG = 6         /* K = 512, wide N: 256-column weight panel in REGISTERS, one persistent 8-wave workgroup
                                     per CU walking 64-row tiles; grid = panels x main_row_tiles row groups               */
};
typedef struct {
  int32_t kernel, grid, block;    /* TTTS_NT_KERNEL_*, workgroups, threads per workgroup                                */
  int32_t tile_m, tile_n;
  int32_t phase;                  /* start-up stagger of co-resident workgroups, units of 1024 cycles (0: none)        */
  int32_t main_row_tiles, tail_tile_rows;   /* WAVE8_SPLIT; WREG: main_row_tiles = row groups                            */
} ttts_gemm_nt_plan;
int ttts_gemm_nt_plan_query(int32_t M, int32_t N, int32_t K, int32_t epilogue, ttts_gemm_nt_plan* out);
/* C[Mo,No] += At[Kr,Mo]^T . Bt[Kr,No]: the weight-gradient GEMM.  The reduction is split over workgroups into fp32
 * slabs in `workspace` (ttts_gemm_tn_workspace_bytes; may be 0 -> NULL) that a second kernel sums in a fixed order
 * (deterministic; no atomics); both operands are row-major with the REDUCTION dimension as rows ("TN").
 * Mo % 8 == 0 or ldat-padded, ldat % 8 == 0, ldbt % 8 == 0. */
int64_t ttts_gemm_tn_workspace_bytes(int32_t Mo, int32_t No, int32_t Kr);
int ttts_gemm_tn_bf16_accum_f32(const void* At, int64_t ldat, const void* Bt, int64_t ldbt, float* C,
                                int64_t ldc, int32_t Mo, int32_t No, int32_t Kr, void* workspace, void* stream);
/* Grouped form of the weight-gradient GEMM: several problems C_i[Mo_i,No_i] += At_i[Kr_i,Mo_i]^T . Bt_i[Kr_i,No_i] in ONE
 * launch, one workgroup per 128x128 output tile over the tile's WHOLE reduction (no slabs, no second kernel; the result
 * is deterministic).  Meant for a backward pass that keeps its dY buffers and runs all dW GEMMs together: the tiles of
 * all problems fill the GPU where a single dW GEMM cannot (same reference call sites as ttts_gemm_tn_bf16_accum_f32:
 * the Conv1D / Linear weight gradients of the GPT blocks, ttts/gpt/model.py:422 via autograd).
 * desc_dev: DEVICE array of n_desc (<= 64) descriptors that ttts_tn_desc_prepare validated and completed on the host.
 * Requirements per problem: Kr % 64 == 0 (zero-pad the operands' rows), ldat/ldbt % 8 == 0 and >= roundup8(Mo/No),
 * ldc % 4 == 0, 16-byte aligned bases. */
typedef struct {
  const void* At; /* [Kr, ldat] bf16 */
  const void* Bt; /* [Kr, ldbt] bf16 */
  float* C;       /* [Mo, ldc] fp32, accumulated into */
  int64_t ldat, ldbt, ldc;
  int32_t Mo, No, Kr;
  int32_t tile_begin; /* exclusive prefix sum of the 128x128 output tiles of earlier descriptors (filled by _prepare) */
} ttts_tn_desc;
int32_t ttts_tn_desc_tiles(int32_t Mo, int32_t No);
int ttts_tn_desc_prepare(ttts_tn_desc* host_desc, int32_t n_desc, int32_t* total_tiles);
int ttts_gemm_tn_grouped_bf16_accum_f32(const ttts_tn_desc* desc_dev, int32_t n_desc, int32_t total_tiles, void* stream);
/* out[n] += sum_m X[m][n]   (bias gradients; X bf16 [M, ldx]) */
int ttts_colsum_bf16_accum_f32(const void* X, int64_t ldx, float* out, int32_t M, int32_t N, void* stream);
/* Batched fp32 -> bf16 cast (+ optional transposed copy) of parameter matrices.
 * desc: DEVICE array of n_desc ttts_cast_desc; total_tiles = sum over descriptors of their 32x32 tiles
 * (ttts_cast_desc_tiles()).  Produces the bf16 "shadow" weights both GEMM operand layouts need. */
typedef struct {
  const float* src; /* [rows, cols] fp32 row-major                 */
  void* dst;        /* [rows, cols] bf16 or NULL                    */
  void* dst_t;      /* [cols, rows] bf16 (transposed copy) or NULL  */
  int32_t rows, cols;
  int32_t tile_begin; /* exclusive prefix sum of tiles of earlier descriptors */
  int32_t ldt;        /* leading dimension of dst_t in elements (0 = rows) */
} ttts_cast_desc;
int32_t ttts_cast_desc_tiles(int32_t rows, int32_t cols);
int ttts_cast_bf16_batched(const ttts_cast_desc* desc, int32_t n_desc, int32_t total_tiles, void* stream);
/* Batched column sums (ABI v5): every bias gradient of a backward section in one launch -- the caller keeps its dY buffers
 * until the end of the section, as for ttts_gemm_tn_grouped_bf16_accum_f32.  desc: DEVICE array, at most 64 entries;
 * tile_begin = exclusive prefix sum of ttts_colsum_desc_tiles(M, N); same arithmetic per problem as the single call. */
typedef struct {
  const void* X;  /* bf16 [M, ldx]                         */
  float* out;     /* f32 [N], accumulated                  */
  int64_t ldx;    /* ldx % 8 == 0, ldx >= roundup8(N)      */
  int32_t M, N;
  int32_t tile_begin, reserved;
} ttts_colsum_desc;
int32_t ttts_colsum_desc_tiles(int32_t M, int32_t N);
int ttts_colsum_bf16_accum_f32_batched(const ttts_colsum_desc* desc, int32_t n_desc, int32_t total_tiles, void* stream);
/* Batched bf16 transpose (ABI v5): dst[c][r] = src[r][c] for up to 64 matrices in one launch (the second GEMM operand layout
 * of the bf16 shadow weights, read from the bf16 shadow AdamW wrote instead of the fp32 master: half the read traffic of
 * ttts_cast_bf16_batched's transposed path).  tile_begin = exclusive prefix sum of ttts_transpose_desc_tiles(rows, cols). */
typedef struct {
  const void* src; /* bf16 [rows, cols] row-major                    */
  void* dst;       /* bf16 [cols, ldd] row-major, ldd >= rows        */
  int32_t rows, cols;
  int32_t ldd, tile_begin;
} ttts_transpose_desc;
int32_t ttts_transpose_desc_tiles(int32_t rows, int32_t cols);
int ttts_transpose_bf16_batched(const ttts_transpose_desc* desc, int32_t n_desc, int32_t total_tiles, void* stream);

/* ---- causal self-attention (flash-style, bf16 I/O, fp32 softmax) -----------------------------------
 * Replaces: GPT2Attention core, modeling_gpt2.py:53-72 (eager) / sdpa, reached from ttts/gpt/model.py:422.
 * q/k/v/o element address = base + b*stride_b + s*stride_s + h*head_dim + d  (so a packed [B,S,3*H*dh]
 * c_attn output is consumed in place: q=base, k=base+H*dh, v=base+2*H*dh, stride_s = 3*H*dh).
 * lse: f32 [B,H,S] = log(sum exp(scaled scores)).  head_dim in {32, 64, 128}.  scale = dh^-0.5.
 * dropout_p in [0,1): keep-mask from a counter hash of (seed, b, h, q, k), quantised to 1/65536;
 * the same (seed) must be given to the backward. */
int ttts_attn_causal_fwd_bf16(const void* q, const void* k, const void* v, void* o, float* lse,
                              int32_t B, int32_t H, int32_t S, int32_t head_dim,
                              int64_t qkv_stride_b, int64_t qkv_stride_s, int64_t o_stride_b, int64_t o_stride_s,
                              float scale, float dropout_p, uint64_t seed, const uint32_t* dropout_counter, void* stream);
/* delta: f32 workspace [B,H,S] (ttts_attn_bwd_workspace_bytes).  dq/dk/dv use the qkv strides (a packed
 * [B,S,3*H*dh] gradient buffer is written in place); do/o use the o strides. */
int64_t ttts_attn_bwd_workspace_bytes(int32_t B, int32_t H, int32_t S);
int ttts_attn_causal_bwd_bf16(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                              const float* lse, void* dq, void* dk, void* dv, void* workspace,
                              int32_t B, int32_t H, int32_t S, int32_t head_dim,
                              int64_t qkv_stride_b, int64_t qkv_stride_s, int64_t o_stride_b, int64_t o_stride_s,
                              float scale, float dropout_p, uint64_t seed, const uint32_t* dropout_counter, void* stream);
/* ---- fused fp32 cross-attention (ABI v5) ------------------------------------------------------------
 * Replaces: the text<->audio cross-attention of MRTE, vc_utils.MultiHeadAttention.attention (ttts/utils/vc_utils.py:597-627,
 * reached from ttts/vqvae/vq2.py:41-43), and any other non-windowed, dropout-free attentions.MultiHeadAttention.attention call
 * (ttts/vqvae/attentions.py:231-290) -- forward and backward in three kernels without the [B, H, Tq, Tk] score tensor.
 * q [B, H*dk, Tq], k / v [B, H*dk, Tk], out [B, H*dk, Tq]: the reference's (B, C, T) tensors, head h = channel rows h*dk ..;
 * qmask [B, Tq] / kmask [B, Tk] (or NULL): a score is replaced by `fill` (-1e4) where qmask * kmask == 0 (masked_fill), then
 * softmax over all Tk keys; stats f32 [B, H, Tq][2] (ttts_attn_cross_stats_bytes) = a query's running (max, 1 / sum), kept for
 * the backward.
 * dk in {64, 96, 128} (else TTTS_EUNSUPPORTED).  Exact-fp32 products (v_mfma_f32_32x32x2_f32). */
int64_t ttts_attn_cross_stats_bytes(int32_t B, int32_t H, int32_t Tq);
int ttts_attn_cross_fwd_f32(const float* q, const float* k, const float* v, const float* qmask, const float* kmask,
                            float* out, float* stats, int32_t B, int32_t H, int32_t dk, int32_t Tq, int32_t Tk,
                            float scale, float fill, void* stream);
int64_t ttts_attn_cross_bwd_workspace_bytes(int32_t B, int32_t H, int32_t Tq);
int ttts_attn_cross_bwd_f32(const float* q, const float* k, const float* v, const float* qmask, const float* kmask,
                            const float* out, const float* dout, const float* stats, float* dq, float* dk_out, float* dv,
                            void* workspace, int32_t B, int32_t H, int32_t dk, int32_t Tq, int32_t Tk, float scale,
                            float fill, void* stream);
/* Debug/test aid: materialise the attention-dropout keep mask (uint8 [B,H,S,S], 1 = keep). */
int ttts_attn_dropout_mask_u8(uint8_t* mask, int32_t B, int32_t H, int32_t S, float dropout_p, uint64_t seed,
                              const uint32_t* dropout_counter, void* stream);

/* ---- LayerNorm (fp32 statistics) -----------------------------------------------------------------
 * Replaces: nn.LayerNorm in GPT2Block (modeling_gpt2.py:254,256), ln_f, final_norm (ttts/gpt/model.py:347,427).
 * x f32 [M,D]; y bf16 or f32 [M,D] (y_is_bf16); mean/rstd f32 [M] saved for backward.
 * split_S/split_T > 0: output row of input row (b*split_S + t) is the "text|mel split layout"
 *   t < split_T ? b*split_T + t : B*split_T + b*(split_S-split_T) + (t-split_T)      (B = M / split_S)
 * which makes the rows each head GEMM consumes contiguous (ttts/gpt/model.py:432-437 slices). */
int ttts_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, int32_t y_is_bf16,
                       float* mean, float* rstd, int32_t M, int32_t D, float eps,
                       int32_t split_S, int32_t split_T, void* stream);
/* dx = [dx_in +] LN'(dy); dgamma += sum dy*xhat; dbeta += sum dy.   dy bf16 or f32 (same row remap as fwd).
 * dx_in may be NULL (then dx = LN'(dy)) or alias dx.  dx_bf16: optional bf16 copy of dx (or NULL).
 * workspace: ttts_layernorm_bwd_workspace_bytes(M, D). */
int64_t ttts_layernorm_bwd_workspace_bytes(int32_t M, int32_t D);
int ttts_layernorm_bwd(const void* dy, int32_t dy_is_bf16, const float* x, const float* gamma,
                       const float* mean, const float* rstd, const float* dx_in, float* dx, void* dx_bf16,
                       float* dgamma, float* dbeta, void* workspace, int32_t M, int32_t D,
                       int32_t split_S, int32_t split_T, void* stream);
/* Same, with the bf16 copy dx_bf16 = bf16(dropout_mask(seed, element row*D + d) * dx / (1 - p)): the gradient that
 * enters the residual-dropout site consuming it (the mask ttts_gemm_nt_bf16_ex applied in the forward), and
 * dcolsum[d] += sum_rows dx_bf16[row][d] (the bias gradient of the projection that consumes dx_bf16; may be NULL). */
int ttts_layernorm_bwd_ex(const void* dy, int32_t dy_is_bf16, const float* x, const float* gamma,
                          const float* mean, const float* rstd, const float* dx_in, float* dx, void* dx_bf16,
                          float* dgamma, float* dbeta, float* dcolsum, void* workspace, int32_t M, int32_t D,
                          int32_t split_S, int32_t split_T, float bf16_dropout_p, uint64_t bf16_dropout_seed,
                          const uint32_t* dropout_counter, void* stream);
/* Deferred parameter gradients (ABI v5): with dgamma == dbeta == NULL the call leaves its per-workgroup partial sums in
 * `workspace` (which then must be this call's own until the finalize) and ttts_layernorm_bwd_finalize_batched adds up the
 * partial sums of up to 65535 such calls (same M, D) in one launch: 14 launches -> 1 in the GPT train step. */
typedef struct {
  const void* workspace;   /* the workspace a deferred ttts_layernorm_bwd_ex call wrote */
  float* dgamma;           /* f32 [D], accumulated */
  float* dbeta;            /* f32 [D], accumulated */
  float* dcolsum;          /* f32 [D], accumulated; or NULL */
} ttts_ln_finalize_desc;
int ttts_layernorm_bwd_finalize_batched(const ttts_ln_finalize_desc* desc /* device */, int32_t n_desc, int32_t M, int32_t D,
                                        void* stream);

/* ---- embeddings ----------------------------------------------------------------------------------
 * Replaces: ttts/gpt/model.py:488,494-495,418 -- token + learned-position embedding sums of the text
 * and mel streams concatenated to x f32 [B, Tt+Tm, D].  dropout_p: GPT2Model.drop (embd_pdrop). */
int ttts_gpt_embed_fwd(const int64_t* text_inp, const int64_t* mel_inp, const float* text_emb,
                       const float* text_pos, const float* mel_emb, const float* mel_pos, float* x,
                       int32_t B, int32_t Tt, int32_t Tm, int32_t D, int32_t n_text, int32_t n_mel,
                       float dropout_p, uint64_t seed, const uint32_t* dropout_counter, void* stream);
int ttts_gpt_embed_bwd(const int64_t* text_inp, const int64_t* mel_inp, const float* dx, float* d_text_emb,
                       float* d_text_pos, float* d_mel_emb, float* d_mel_pos,
                       int32_t B, int32_t Tt, int32_t Tm, int32_t D, float dropout_p, uint64_t seed,
                       const uint32_t* dropout_counter, void* stream);
/* Token plumbing of UnifiedVoice.forward in ONE launch -- replaces ttts/gpt/model.py:474-489 (clip to the batch maximum,
 * set_mel_padding: mel positions >= wav_len / mel_length_compression + 1 become STOP, append STOP) and
 * build_aligned_inputs_and_targets (:397-414: inp = [START, seq], tar = [seq, STOP]).
 * text int64 [B, ld_text], mel int64 [B, ld_mel] (device); Tt / Tm = the clipped lengths the caller computed on the host;
 * mel_valid_host: HOST array of B ints (wav_len / compression + 1), copied into the launch (B <= 256).
 * Outputs (device, int64): text_inp, text_tar [B, Tt + 2]; mel_inp, mel_tar [B, Tm + 2]. */
int ttts_gpt_prepare_tokens(const int64_t* text, int64_t ld_text, const int64_t* mel, int64_t ld_mel,
                            const int32_t* mel_valid_host, int32_t B, int32_t Tt, int32_t Tm,
                            int32_t start_text, int32_t stop_text, int32_t start_mel, int32_t stop_mel,
                            int64_t* text_inp, int64_t* text_tar, int64_t* mel_inp, int64_t* mel_tar, void* stream);

/* ---- cross-entropy -------------------------------------------------------------------------------
 * Replaces: F.cross_entropy(logits.permute, targets) mean reduction, ttts/gpt/model.py:508-509.
 * logits bf16 [R, ldl] (C valid classes per row), targets int64 [R].
 * fwd: row_loss f32 [R] = lse - logit[target]; row_lse f32 [R]; loss_mean (f32 scalar) = mean(row_loss).
 * bwd: dlogits bf16 [R, ldl] = (softmax - onehot) * grad_scale * (*grad_scale_dev if non-NULL) / R. */
int ttts_ce_fwd_bf16(const void* logits, int64_t ldl, const int64_t* targets, float* row_loss, float* row_lse,
                     float* loss_mean, int32_t R, int32_t C, void* stream);
int ttts_ce_bwd_bf16(const void* logits, int64_t ldl, const int64_t* targets, const float* row_lse,
                     void* dlogits, float grad_scale, const float* grad_scale_dev, int32_t R, int32_t C,
                     void* stream);

/* ---- optimizer -----------------------------------------------------------------------------------
 * Replaces: get_grad_norm (ttts/gpt/train.py:22-31, 84 .item() syncs), accelerator.clip_grad_norm_(1.0)
 * (:115), torch.optim.AdamW (:56,118), LambdaLR(warmup) (:36-40,57,120) -- on ONE flat fp32 arena.
 *
 * state: f32[8] device buffer {step, lr, bias_corr1, bias_corr2_sqrt, grad_norm, clip_coef, skip, -}.
 * skip (ABI v11): non-zero makes THIS optimizer step a no-op -- ttts_adamw_schedule does not advance, ttts_adamw_f32 leaves the
 *   parameters, moments and shadow alone and only consumes (zeroes) the gradient: `GradScaler.step`'s behaviour on an overflowed
 *   step (ttts/vqvae/train.py:262,356-372 with fp16_run), decided on the device by ttts_loss_scale_check.  Zero: as before.
 * ttts_adamw_schedule: step += 1 (step counts optimizer steps taken, starts at 0), lr = base_lr *
 *   (warmup_steps > 0 ? min(1, (step-1)/warmup_steps) : 1)  [LambdaLR value used by THIS step],
 *   bias corrections in double precision.  Device-side so the whole step stays graph-capturable. */
int ttts_adamw_schedule(float* state, float base_lr, float beta1, float beta2, int32_t warmup_steps, void* stream);
/* grad_norm = ||g||_2, clip_coef = min(1, max_norm / (norm + 1e-6)) (max_norm <= 0: coef = 1) -> state[4..5].
 * workspace: ttts_gradnorm_workspace_bytes(n). */
int64_t ttts_gradnorm_workspace_bytes(int64_t n);
int ttts_gradnorm_f32(const float* g, int64_t n, float max_norm, float* state, void* workspace, void* stream);
/* AdamW on n elements (n % 4 == 0, 16-byte aligned), torch single-tensor formula order; the gradient is
 * scaled by state[5] (clip_coef); zero_grad != 0 writes g = 0 after use; shadow: optional bf16 [n] copy of
 * the updated parameters (same layout). */
int ttts_adamw_f32(float* p, float* g, float* m, float* v, void* shadow_bf16, int64_t n, const float* state,
                   float beta1, float beta2, float eps, float weight_decay, int32_t zero_grad, void* stream);

/* ---- VQ codebook ---------------------------------------------------------------------------------
 * Replaces: EuclideanCodebook.quantize / dequantize (ttts/vqvae/core_vq.py:174-189), the straight-through
 * + commitment MSE of VectorQuantization.forward (:303-322) and the EMA update (:212-228).
 * x f32 [N,D] row-major, codebook f32 [K,D]; idx int64 [N]; xq f32 [N,D] = codebook[idx].
 * Distances are evaluated in IEEE fp32 exactly as the reference's expression
 *   -((|x|^2 - 2*dot) + |e|^2), dot = k-ordered fmaf chain; first maximum wins (ties -> lowest index).
 * workspace: ttts_vq_workspace_bytes(N, K) (code norms + per-row scratch). */
int64_t ttts_vq_workspace_bytes(int32_t N, int32_t K);
int ttts_vq_nearest_f32(const float* x, const float* codebook, int64_t* idx, float* xq, float* best_dist,
                        void* workspace, int32_t N, int32_t K, int32_t D, void* stream);
/* loss (f32 scalar) = mean((xq - x)^2) computed as mse(x + (xq - x), x) like the reference (:311-317);
 * dx f32 [N,D] += grad_scale * 2 (x - q)/(N*D)  (gradient of the commitment term w.r.t. x). */
int ttts_vq_commit_f32(const float* x, const float* xq, float* loss, float* dx, float grad_scale,
                       int32_t N, int32_t D, void* workspace, void* stream);
/* EMA update: cluster_size, embed_avg, embed updated in place from (x, idx) -- scatter-add, no one-hot.
 * workspace: ttts_vq_ema_workspace_bytes(K, D). */
int64_t ttts_vq_ema_workspace_bytes(int32_t K, int32_t D);
int ttts_vq_ema_update_f32(const float* x, const int64_t* idx, float* cluster_size, float* embed_avg,
                           float* embed, void* workspace, int32_t N, int32_t K, int32_t D, float decay,
                           float epsilon, void* stream);

/* ---- mel / STFT front-end --------------------------------------------------------------------------
 * Replaces: spectrogram_torch (ttts/utils/data_utils.py:52-87): reflect-pad (n_fft-hop)/2, hann window,
 * onesided STFT (center=False), sqrt(re^2 + im^2 + 1e-6).   wav f32 [B,T]; spec f32 [B, n_fft/2+1, frames],
 * frames = (T + 2*pad - n_fft)/hop + 1.  n_fft = win_size in {1024, 2048}.
 * twiddle: f32 [n_fft] device table (cos, sin interleaved pairs for k < n_fft/2) from ttts_stft_twiddle_host. */
int ttts_stft_twiddle_host(float* host_out, int32_t n_fft);
int ttts_stft_mag_fwd_f32(const float* wav, const float* window, const float* twiddle, float* spec,
                          int32_t B, int32_t T, int32_t n_fft, int32_t hop, void* stream);
/* spec_to_mel_torch (:90-103): mel f32 [B, n_mels, frames] = log(clamp(basis[n_mels, n_bins] @ spec, 1e-5)).
 * bands: optional DEVICE int32 [n_mels][2] = {first, last} non-zero column of every basis row (a mel filterbank row is a
 * narrow band; the caller computes this once per basis).  NULL: every workgroup finds the bands itself (slower).  Results
 * do not depend on it: the exact-zero terms it skips leave an fmaf chain unchanged. */
int ttts_mel_log_fwd_f32(const float* spec, const float* basis, const int32_t* bands, float* mel, int32_t B, int32_t n_bins,
                         int32_t n_mels, int32_t frames, void* stream);

/* Backward of the two functions above (mel_spectrogram_torch on the generated audio is differentiated in the VQ-VAE
 * step, ttts/vqvae/train.py:362-371,394).  dspec f32 [B, n_bins, frames] += basis^T . (dmel / v) where the clamp passed;
 * dwav f32 [B, T] += adjoint(STFT magnitude)(dspec) (the frame spectra are recomputed, not stored).
 * twiddle2: f32 [2*n_fft] table from ttts_stft_twiddle_host(out, 2*n_fft). */
int ttts_mel_log_bwd_f32(const float* dmel, const float* mel, const float* basis, float* dspec, int32_t B,
                         int32_t n_bins, int32_t n_mels, int32_t frames, void* stream);
int ttts_stft_mag_bwd_f32(const float* wav, const float* window, const float* twiddle2, const float* dspec,
                          float* dwav, int32_t B, int32_t T, int32_t n_fft, int32_t hop, void* stream);

/* ---- diffusion mel-denoiser step ----------------------------------------------------------------
 * The pieces of AA_diffusion / SpacedDiffusion.training_losses that the conv and attention families above do not cover
 * (ttts/diffusion/aa_model.py:32-287, ttts/utils/utils.py:113-215, ttts/utils/xtransformers.py:146-185,
 * ttts/utils/diffusion.py:17-82,243-282,903-1014).  Tensors are f32 (B, C, T).
 *  groupnorm_fwd: GroupNorm32 (utils.py:113-133): y = act(((x - mean_g) rstd_g gamma + beta) (1 + scale) + shift);
 *    scale_shift f32 [B, 2C] (scale | shift: the ResBlock's timestep modulation, aa_model.py:121-126) or NULL; silu != 0
 *    applies x sigmoid(x) last.  mean / rstd f32 [B, groups] are saved for the backward.
 *  groupnorm_bwd: dx, dgamma / dbeta [C] (accumulate != 0: +=), d_scale_shift [B, 2C]; workspace f32 [2 B C].
 *  relpos_bias_fwd: RelativePositionBias.forward: bias f32 [H, Tq, Tk] = table[bucket[j - i + offset]][h] * scale with
 *    table f32 [num_buckets, H] and bucket i32 [2 offset + 1] (host-built with `_relative_position_bucket`);
 *    relpos_bias_bwd: dtable from dS f32 [B, H, Tq, Tk] (workspace: ttts_relpos_bias_bwd_workspace_bytes).
 *  softmax_bias_fwd: in place softmax_j(scores[b,h,i,j] + bias[h,i,j]) (QKVAttentionLegacy, utils.py:157-162); backward =
 *    ttts_attn_softmax_bwd_f32 without masks.
 *  interp_nearest_fwd/bwd: F.interpolate(mode='nearest') along T for `rows` = B*C rows and its adjoint.
 *  timestep_embedding: aa_model.py:32-51, t i64 [N], freqs f32 [dim/2] (= exp(-ln(max_period) k / (dim/2))) -> emb f32 [N, dim].
 *  select_rows_fwd/bwd: out[b] = use[b] ? vec (C, broadcast over T) : a[b] (the unconditioned-embedding mask, :246-250).
 *  q_sample: x_t = tab[t][0] x_0 + tab[t][1] noise.  table f32 [steps, 8]: sqrt_alphas_cumprod, sqrt_one_minus_alphas_cumprod,
 *    sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod, posterior_mean_coef1, posterior_mean_coef2,
 *    posterior_log_variance_clipped, log(betas) (built in float64 on the host as GaussianDiffusion.__init__ does).
 *  diffusion_loss_fwd: model_out f32 [B, 2C, T] = (eps | var values): terms f32 [B, 3] = (mse, vb, mse + vb) of
 *    training_losses (epsilon / learned_range / mse; the vb term sees eps detached), loss_mean = mean_b loss;
 *    diffusion_loss_bwd: d loss_mean / d model_out (times gout[0] if given). */
int ttts_groupnorm_fwd_f32(const float* x, const float* gamma, const float* beta, const float* scale_shift, float* y,
                           float* mean, float* rstd, int32_t B, int32_t C, int32_t T, int32_t groups, float eps,
                           int32_t silu, void* stream);
int ttts_groupnorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* beta, const float* scale_shift,
                           const float* mean, const float* rstd, float* dx, float* dgamma, float* dbeta,
                           float* d_scale_shift, float* workspace, int32_t B, int32_t C, int32_t T, int32_t groups,
                           int32_t silu, int32_t accumulate, void* stream);
int ttts_relpos_bias_fwd_f32(const float* table, const int32_t* bucket, float* bias, int32_t H, int32_t Tq, int32_t Tk,
                             int32_t bucket_offset, float scale, void* stream);
int64_t ttts_relpos_bias_bwd_workspace_bytes(int32_t B, int32_t H, int32_t Tq, int32_t Tk);
int ttts_relpos_bias_bwd_f32(const float* dS, const int32_t* bucket, float* dtable, float* workspace, int32_t B, int32_t H,
                             int32_t Tq, int32_t Tk, int32_t bucket_offset, int32_t num_buckets, float scale,
                             int32_t accumulate, void* stream);
int ttts_softmax_bias_fwd_f32(float* scores, const float* bias, int32_t B, int32_t H, int32_t Tq, int32_t Tk, void* stream);
/* ABI v11: the attention of AttentionBlock as ONE forward and THREE backward launches (csrc/attn_relpos.hip): no (B, H, T, T) tensor.
 * Replaces: QKVAttentionLegacy.forward (ttts/utils/utils.py:136-169) + RelativePositionBias.forward (xtransformers.py:146-185)
 *   w = softmax(q^T k / sqrt(ch) + table[bucket(j - i)][h] * bias_scale),  out = v w^T
 * qkv f32 (B, H, 3, ch, T) (the qkv convolution's output viewed per head), ch == 32; table f32 (num_buckets, H); bucket int32
 * [2 bucket_off + 1] with the bucket of d = j - i at index d + bucket_off (bucket_off >= T - 1); out f32 (B, H, ch, T);
 * lse f32 (B, H, T): log2-domain log-sum-exp of the scores, the backward's only saved statistic.
 * products: 3 = split-bf16 operands (hi*hi + hi*lo + lo*hi, fp32-equivalent), 1 = plain bf16 operands (autocast arithmetic);
 * fp32 accumulation and softmax either way.  T <= ttts_attn_relpos_max_t(products) (all keys of a head live in LDS): 448 / 896.
 * bwd: dqkv f32 (B, H, 3, ch, T) written; dtable (num_buckets, H) written or (accumulate_dtable) added to, NULL: not computed --
 * (per-workgroup diagonal sums by LDS atomics, then a fixed-order reduction: reproducible to fp32 summation noise; dqkv bit for bit).
 * workspace: ttts_attn_relpos_workspace_bytes(B, H, T). */
int32_t ttts_attn_relpos_max_t(int32_t products);
int64_t ttts_attn_relpos_workspace_bytes(int32_t B, int32_t H, int32_t T);
int ttts_attn_relpos_fwd_f32(const float* qkv, const float* table, const int32_t* bucket, int32_t bucket_off, float* out, float* lse,
                             int32_t B, int32_t H, int32_t T, int32_t ch, float bias_scale, int32_t products, void* stream);
int ttts_attn_relpos_bwd_f32(const float* qkv, const float* table, const int32_t* bucket, int32_t bucket_off, const float* out,
                             const float* dout, const float* lse, float* dqkv, float* dtable, int32_t accumulate_dtable,
                             void* workspace, int32_t B, int32_t H, int32_t T, int32_t ch, int32_t num_buckets, float bias_scale,
                             int32_t products, void* stream);
int ttts_interp_nearest_fwd_f32(const float* x, float* y, int64_t rows, int32_t Tin, int32_t Tout, void* stream);
int ttts_interp_nearest_bwd_f32(const float* dy, float* dx, int64_t rows, int32_t Tin, int32_t Tout, void* stream);
int ttts_timestep_embedding_f32(const int64_t* t, const float* freqs, float* emb, int32_t N, int32_t dim, void* stream);
int ttts_select_rows_fwd_f32(const uint8_t* use_vec, const float* a, const float* vec, float* out, int32_t B, int32_t C,
                             int32_t T, void* stream);
int ttts_select_rows_bwd_f32(const uint8_t* use_vec, const float* dout, float* da, float* dvec, int32_t B, int32_t C,
                             int32_t T, int32_t accumulate, void* stream);
int ttts_q_sample_f32(const float* x_start, const float* noise, const int64_t* t, const float* table, float* x_t, int32_t B,
                      int64_t per_sample, void* stream);
int64_t ttts_diffusion_loss_workspace_bytes(int32_t B);
int ttts_diffusion_loss_fwd_f32(const float* model_out, const float* x_start, const float* x_t, const float* noise,
                                const int64_t* t, const float* table, float* terms, float* loss_mean, float* workspace,
                                int32_t B, int32_t C, int32_t T, void* stream);
int ttts_diffusion_loss_bwd_f32(const float* model_out, const float* x_start, const float* x_t, const float* noise,
                                const int64_t* t, const float* table, const float* gout, float* d_model_out, int32_t B,
                                int32_t C, int32_t T, void* stream);

/* ---- autoregressive decoding of the GPT --------------------------------------------------------
 * Replaces: GPT2InferenceModel.forward with a KV cache (ttts/gpt/model.py:34-184), inference_speech (:533-562) and the
 * sample loop + logits processors of transformers' GenerationMixin it calls (RepetitionPenaltyLogitsProcessor,
 * TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper) plus ttts/utils/typical_sampling.py:5-35.
 * All step-dependent positions come from `ctr`, an int32[4] block in DEVICE memory owned by the caller:
 *   ctr[0] = tokens already in the cache (= sequence index of the token being processed), ctr[1] = tokens generated,
 *   ctr[2] = sequences still running (written by ttts_decode_advance) -- so one captured hipGraph replays every step.
 *  ttts_decode_embed_f32: x f32 [M, D] = emb[tokens[m]] + pos[ctr[0] + pos_offset]   (emb [V, D], pos [P, D]).
 *  ttts_kv_cache_fill_bf16: K / V of a prefill pass, qkv bf16 [B*S, 3*H*dh] -> caches bf16 [B*rep, H, S_max, dh];
 *    sequence b*rep + r is a copy of b (num_return_sequences, repeat_interleave order).
 *  ttts_attn_decode_bf16: qkv bf16 [M, 3*H*dh] of the new token: appends K / V at index ctr[0], out bf16 [M, H*dh] =
 *    softmax(scale q.K^T) V over keys 0..ctr[0].  head_dim in {32, 64, 128}.
 *  ttts_sample_logits_f32: logits f32 row (m / row_div) of [*, ldl], vocabulary V <= 2048.  Order as HF builds it:
 *    repetition penalty over history[m][0 .. hist_base + ctr[1]) -> typical filter (typical_mass > 0) -> and, when
 *    do_sample: / temperature -> top_k (> 0) -> top_p (< 1) -> softmax -> inverse-CDF draw with a counter hash of
 *    (seed, m, ctr[1]); do_sample 0: argmax (lowest id on ties).  Finished rows emit pad_token; a row finishes when it
 *    draws eos_token.  Writes tokens[m], history[m][hist_base + ctr[1]], out[m][ctr[1]], finished[m]; optional
 *    probs_out f32 [M, V] (sampling: final probabilities; greedy: processed scores) and u_out f32 [M] for tests.
 *  ttts_decode_advance: ctr[0]++, ctr[1]++, ctr[2] = #unfinished. */
/* Skinny linear layer of a decode step, M <= 16 rows: out[M, N] = epilogue(LN2(LN1(x))[M, K] . W[N, K]^T + bias) with the
 * rounding points of ttts_gemm_nt_bf16 (operands bf16, bf16(acc + bias) before GELU / residual add).  x: bf16 [M, ldx], or
 * f32 when x_is_f32 (then optionally layer-normed, eps 1e-5, by ln1 and ln2 -- NULL = skip); W bf16 [N, ldw];
 * epilogue in {STORE_BF16, GELU_BF16 (out = gelu only, no aux), RESID_ADD_F32 (out = resid + ..), STORE_F32}. */
int ttts_linear_decode_bf16(const void* x, int64_t ldx, int32_t x_is_f32, const float* ln1_gamma, const float* ln1_beta,
                            const float* ln2_gamma, const float* ln2_beta, const void* W, int64_t ldw, const float* bias,
                            void* out, int64_t ldc, const float* resid, int32_t M, int32_t N, int32_t K, int32_t epilogue,
                            void* stream);
int ttts_decode_embed_f32(const int64_t* tokens, const float* emb, const float* pos, const int32_t* ctr,
                          int32_t pos_offset, float* x, int32_t M, int32_t D, int32_t V, int32_t P, void* stream);
int ttts_kv_cache_fill_bf16(const void* qkv, void* k_cache, void* v_cache, int32_t B, int32_t S, int32_t H,
                            int32_t head_dim, int32_t S_max, int32_t rep, void* stream);
int ttts_attn_decode_bf16(const void* qkv, void* k_cache, void* v_cache, const int32_t* ctr, void* out, int32_t M,
                          int32_t H, int32_t head_dim, int32_t S_max, float scale, void* stream);
int ttts_sample_logits_f32(const float* logits, int64_t ldl, int32_t row_div, int32_t M, int32_t V, int64_t* history,
                           int64_t hist_stride, int32_t hist_base, const int32_t* ctr, int64_t* tokens, int64_t* out,
                           int64_t out_stride, uint8_t* finished, float repetition_penalty, float typical_mass,
                           float temperature, int32_t top_k, float top_p, int32_t do_sample, int32_t eos_token,
                           int32_t pad_token, uint64_t seed, float* probs_out, float* u_out, void* stream);
int ttts_decode_advance(int32_t* ctr, const uint8_t* finished, int32_t M, void* stream);

/* ---- parametric-equaliser augmentation -------------------------------------------------------
 * Replaces Augment.forward's PEQ path (ttts/vqvae/augment/__init__.py:37-97) and ParametricEqualizer
 * (ttts/vqvae/augment/peq.py:19-116): torch.stft(center=True, hann) -> per-clip biquad product -> torch.istft ->
 * clamp(-1, 1) -> divide by the clip's peak.  The Praat stage (augment/praat.py) is a CPU library and stays outside.
 *  ttts_peq_response_f32: H c64 [B, n_fft/2+1] (interleaved re, im) = prod_f fir_f / iir_f with
 *    freq / gain (dB) / q f32 [B, n_filters], kind i32 [n_filters] (TTTS_PEQ_PEAK | _LOW_SHELF | _HIGH_SHELF);
 *    closed form of rfft([c0, c1, c2], n_fft) (peq.py:19-30), evaluated in double (the 3-tap sums cancel near DC).
 *  ttts_stft_filter_frames_f32: frames_out f32 [B, frames, n_fft], frames = ttts_stft_center_frames(T, hop) = 1 + T/hop:
 *    frame t = window * irfft(H * rfft(window * reflect_pad(wav, n_fft/2)[t hop : t hop + n_fft])); H NULL = identity.
 *  ttts_istft_ola_f32: out f32 [B, hop (frames - 1)] = overlap-add / window-envelope, trimmed as torch.istft does,
 *    optionally clamped to [-1, 1]; peak_bits u32 [B] receives the float bits of max |out| per clip (zeroed here).
 *  ttts_peak_scale_f32: x[b] /= max(peak[b], eps) in place. */
#define TTTS_PEQ_PEAK 0
#define TTTS_PEQ_LOW_SHELF 1
#define TTTS_PEQ_HIGH_SHELF 2
int ttts_peq_response_f32(const float* freq, const float* gain, const float* q, const int32_t* kind, float* H,
                          int32_t B, int32_t n_filters, int32_t n_fft, float sample_rate, void* stream);
int32_t ttts_stft_center_frames(int32_t T, int32_t hop);
int ttts_stft_filter_frames_f32(const float* wav, const float* window, const float* twiddle, const float* H,
                                float* frames_out, int32_t B, int32_t T, int32_t n_fft, int32_t hop, void* stream);
int ttts_istft_ola_f32(const float* frames_in, const float* window, float* out, void* peak_bits, int32_t B,
                       int32_t frames, int32_t n_fft, int32_t hop, int32_t clamp, void* stream);
int ttts_peak_scale_f32(float* x, const void* peak_bits, int32_t B, int32_t T, float eps, void* stream);

/* ---- 1-D convolution family (fp32, (B, C, L) layout, groups = 1) ---------------------------------------------
 * Replaces: nn.Conv1d (incl. groups) / Conv2d with (k,1) kernels / ConvTranspose1d / weight_norm + the leaky-relu, bias, residual-add and tanh around them in
 * ResBlock1 (ttts/vqvae/modules.py:224-318), Generator (ttts/vqvae/vq2.py:341-415), PosteriorAudioEncoder (:667-745),
 * WN (modules.py:136-221), and their autograd.  w: [Cout, Cin, K].
 * fwd:   y = [y +] out_scale * omask[b][l] * act_out(lrelu'(gate) * (bias[co] + bbias[b][co] + conv(lrelu(x, in_slope), w)) + resid)
 *        Lout = (Lin + 2 pad - dil (K-1) - 1)/stride + 1;  lrelu'(gate) = gate > 0 ? 1 : gate_slope (gate NULL: 1);
 *        out_act: 0 none, 1 tanh, 2 leaky-relu(out_slope); omask [B, L] (sequence mask) or NULL.  Cin/Cout are TOTAL channel counts; w: [Cout, Cin/groups, K].
 * dgrad: dx = [dx +] out_scale * omask[b][l] * (lrelu'(gate) * (bias[ci] + conv^T(lrelu(dy, in_slope), w)) + resid) -- with a
 *        ConvTranspose1d weight [Cin_t, Cout_t, K] passed as w (Cout := Cin_t, Cin := Cout_t, Lin := output length) this
 *        IS the transposed convolution's forward (and fwd is its data gradient); stride > 1 requires dil == 1.
 * wgrad: dw += sum_{b,l} lrelu(dy, dy_slope) * lrelu(x, x_slope) (shifted).   bias_grad: db[c] += sum_{b,l} dy.
 * weight_norm (dim 0): w[r] = g[r] v[r] / ||v[r]||, norm[r] saved; bwd accumulates dv, dg. */
/* Caller-owned context of the convolution family (plain data; the library keeps no copy and no global of it).
 *  workspace: optional 16-byte aligned scratch (>= 32 MB covers every layer of the path; 1.5 GB also holds the pre-split
 *             activations and weight-gradient slabs of the BASELINE batch).  With it conv1d_fwd / dgrad / wgrad run the
 *             split-bf16 matrix-core kernels (fp32 operands carried as hi + lo bf16, x*w accumulated in fp32 as
 *             hi*hi + hi*lo + lo*hi, relative error ~2^-16) and stage their pre-split operands / partial sums there;
 *             convolutions sharing one workspace must be ordered on one stream.  NULL (or ctx == NULL): exact kernels.
 *  handles:   see the struct; the caches / arenas are caller-owned objects too, so that all state a call can depend on is
 *             reachable from its arguments (no global state in the library except immutable kernel tables).
 *  flags:     TTTS_CONV_EXACT_F32 keeps the exact-fp32 MFMA kernels (bit-for-bit fmaf chains) even with a workspace;
 *             the remaining bits override tile / kernel heuristics for experiments (tools/conv_bench.py) and are 0 in
 *             normal operation. */
typedef struct ttts_conv_ctx {
  void* workspace;
  int64_t workspace_bytes;
  int32_t flags;
  int32_t n_handles;        /* ABI v10 (was `reserved`): entries of `handles` */
  void* const* handles;     /* ABI v10: the weight-split caches (ttts_conv_wsplit_cache_create) and weight-gradient arenas
                             * (ttts_conv_wgrad_arena_create) this call may use, in any order; NULL / 0: none.  The library keeps
                             * NO registry of them: a convolution call sees exactly the objects its context names. */
  int32_t device;           /* ABI v10: ordinal of the device the call runs on (the one `stream` belongs to), or -1: the library asks
                             * the runtime where it has to (per-device kernel attributes) -- a hint that saves an API call per launch */
  int32_t reserved;
} ttts_conv_ctx;
#define TTTS_CONV_EXACT_F32 4096
#define TTTS_CONV_F16X1 1024             /* ABI v10: single-pass "TF32-class" arithmetic for the matrix-core convolutions: operands
                                          * rounded to fp16 (11 significant bits, what the reference's TF32 cuDNN convolutions carry,
                                          * ttts/vqvae/train.py:34-36; saturating at +-65504), ONE fp16 MFMA product per pair with fp32
                                          * accumulation instead of the three bf16 products of the split form.  Forward and data
                                          * gradient; weight gradients keep the split form (wider, 14 % of the family's time).  The
                                          * data gradient's input needs the caller's loss scaling to stay inside fp16's range
                                          * (ttts_amd.vqvae.train applies 2^10).  Ignored with TTTS_CONV_EXACT_F32. */
#define TTTS_CONV_DIRECT_ONLY 256        /* experiments: every convolution on the direct (non-MFMA) kernels */
#define TTTS_CONV_SMALL_TILES 2048       /* experiments: allow the small MFMA tile shapes */
#define TTTS_CONV_FORCE_SPLIT_WGRAD 8192 /* tests: split-bf16 weight gradient for every shape */
#define TTTS_CONV_NO_TAPS_WGRAD 16384    /* experiments: disable the all-taps weight-gradient kernel */
/* further experiment bits (A/B switches of tools/conv_bench.py, documented where they are tested in csrc/conv*.hip):
 * 32768 DMA kernel from one output-channel tile, 65536 never the DMA kernel, 131072 / 262144 never / always its 64 x 256 tile,
 * 1048576 no narrow-layer weight-gradient kernels, 2097152 pre-split instead of fused narrow weight gradient, 8388608 no
 * thin-layer (1-channel) kernels, 16777216 direct instead of MFMA grouped kernels, 33554432 run-time tap loops, 67108864
 * one-tap instead of all-taps stride-3 weight gradient, 134217728 segment folding instead of virtual rows for short rows;
 * round 4: 4194304 per-phase launches instead of the phase-merged strided data gradient / forward, 536870912 / 1073741824 split-K
 * target of 256 / 1024 workgroups instead of 512, 1 small 1 x 1 weight gradients on the exact kernel, 2 phase-merged forward at
 * stride 3 too -- every one of them measured and left off (HISTORY.md 17.3); round 6: 512 switches conv1x1_b3_kernel (1 x 1
 * convolutions without the operand pre-pass: ON by default) off, 1024 its 128-row tiles for wide layers (M % 128 == 0, M >= 512),
 * 2048 the one-pass 1 x 1 weight gradient (conv1x1_wgrad_fused_kernel), 524288 power-of-two segments instead of one virtual row
 * for the short rows of phase-merged strided data gradients */
int ttts_conv1d_fwd_f32(const float* x, const float* w, const float* bias, const float* bbias, const float* resid,
                        const float* gate, const float* omask, float* y, int32_t B, int32_t Cin, int32_t Lin, int32_t Cout, int32_t Lout,
                        int32_t K, int32_t stride, int32_t pad, int32_t dil, int32_t groups, float in_slope,
                        float gate_slope, int32_t out_act, float out_slope, float out_scale, int32_t accumulate,
                        const ttts_conv_ctx* ctx, void* stream);
/* Dual-destination forward (ABI v9; stride 1, groups 1): ONE convolution whose first Cout1 output channels go to y [B, Cout1, Lout]
 * (y = (conv + bias + resid) * omask) and whose remaining channels go to y2 [B, Cout - Cout1, Lout] (y2 [+]= (conv + bias) * omask,
 * accumulate2).  WaveNet's res/skip 1 x 1 convolution (`res_skip_acts[:, :H]` / `[:, H:]`, ttts/vqvae/modules.py:96-104) is two
 * destinations of one GEMM over the same input; results are those of the two ttts_conv1d_fwd_f32 calls it replaces. */
int ttts_conv1d_fwd_dual_f32(const float* x, const float* w, const float* bias, const float* resid, const float* omask, float* y,
                             float* y2, int32_t B, int32_t Cin, int32_t Lin, int32_t Cout, int32_t Cout1, int32_t Lout, int32_t K,
                             int32_t pad, int32_t dil, float in_slope, int32_t accumulate2, const ttts_conv_ctx* ctx, void* stream);
int ttts_conv1d_dgrad_f32(const float* dy, const float* w, const float* bias, const float* resid,
                          const float* gate, const float* omask, float* dx, int32_t B, int32_t Cin, int32_t Lin, int32_t Cout,
                          int32_t Lout, int32_t K, int32_t stride, int32_t pad, int32_t dil, int32_t groups,
                          float in_slope, float gate_slope, float out_scale, int32_t accumulate,
                          const ttts_conv_ctx* ctx, void* stream);
/* dw += weight gradient; db (optional, NULL to skip; requires dy_slope == 1): db[co] += sum_{b,l} dy[b][co][l], the bias
 * gradient of the same layer, folded into the kernels that stream dy anyway (ABI v3; it was a separate pass over dy) */
int ttts_conv1d_wgrad_f32(const float* dy, const float* x, float* dw, float* db, int32_t B, int32_t Cin, int32_t Lin,
                          int32_t Cout, int32_t Lout, int32_t K, int32_t stride, int32_t pad, int32_t dil,
                          int32_t groups, float dy_slope, float x_slope, const ttts_conv_ctx* ctx, void* stream);
int ttts_conv1d_bias_grad_f32(const float* dy, float* db, int32_t B, int32_t C, int32_t L, void* stream);
/* zero_out (ABI v5; or NULL): a second [rows][n] buffer cleared in the same pass -- the buffer the layer's weight-gradient call
 * will accumulate into (ttts_conv1d_wgrad_f32 adds), so that the backward needs no fill launch for it */
int ttts_weight_norm_fwd_f32(const float* v, const float* g, float* w, float* norm, float* zero_out, int32_t rows, int32_t n,
                             void* stream);
int ttts_weight_norm_bwd_f32(const float* dw, const float* v, const float* g, const float* norm, float* dv,
                             float* dg, int32_t rows, int32_t n, void* stream);
/* Batched forms (ABI v7): every weight-normed layer of a network in ONE launch each (one workgroup per weight row; bit-identical
 * to the per-layer calls).  Descriptors live in device memory, ordered by row_begin (the layer's first row in the launch's grid);
 * forward: w = g v / |v|, norm, and dw (if not NULL) cleared; backward: dv += ..., dg += ... from dw (fields the direction does
 * not use may be NULL).  Replaces torch.nn.utils.weight_norm / parametrizations.weight_norm recomputation per forward
 * (ttts/vqvae/vq2.py:10,364, ttts/vqvae/modules.py:8) and its autograd. */
typedef struct {
  const void* v; const void* g; void* w; void* norm; void* dw; void* dv; void* dg;
  int32_t rows, n, row_begin, reserved;
} ttts_wn_desc;
int ttts_weight_norm_fwd_batched_f32(const ttts_wn_desc* desc_dev, int32_t n_desc, int32_t total_rows, void* stream);
int ttts_weight_norm_bwd_batched_f32(const ttts_wn_desc* desc_dev, int32_t n_desc, int32_t total_rows, void* stream);
/* Weight-split cache (ABI v8).  The split-bf16 convolutions consume their weights as bf16 hi / lo arrays in a per-launch layout
 * (transposed / tap-flipped / polyphase for data gradients, padded to the tile): formerly one ~5 us split launch in front of
 * every forward and data-gradient call (1300 of them per VQ-VAE-GAN step).  The host registers an array the weights live in
 * (the optimizer's flat parameter arena, a WeightNormBank's flat effective weights: [w_base, w_base + w_bytes)) with caller-owned
 * `storage` (256-byte aligned; it holds the descriptor table and the split arrays).  While the cache is ARMED, the first
 * convolution call with a given (weight pointer, layout) records a descriptor and splits into a persistent slot; later calls
 * launch nothing, and ttts_conv_wsplit_cache_refresh -- to be called after every change of the weights, before they are next
 * used -- rewrites all recorded splits in ONE launch and arms the cache.  Disarmed (initially, and after _disarm: call it when
 * the step ends, since the arrays may then change without a refresh), when storage or max_entries run out, on a first
 * sighting during stream capture, or -- between an entry's first split and the next refresh -- on any stream other than the one
 * that issued that split (only that stream is ordered behind it), calls split per launch into the workspace as before: the
 * cache changes launch counts, never results.  The handle goes into ttts_conv_ctx::handles of the calls that may use it (v10).  No reference counterpart (cudnn picks its own weight layouts inside F.conv1d, ttts/vqvae/vq2.py:364-403). */
int ttts_conv_wsplit_cache_create(const void* w_base, int64_t w_bytes, void* storage, int64_t storage_bytes, int32_t max_entries,
                                  void** cache_out);
int ttts_conv_wsplit_cache_refresh(void* cache, void* stream);
int ttts_conv_wsplit_cache_disarm(void* cache);
int ttts_conv_wsplit_cache_stats(void* cache, int64_t* out4 /* entries, storage bytes used, hits, misses */);
int ttts_conv_wsplit_cache_destroy(void* cache);
/* Deferred weight-gradient reduction (ABI v8).  The split-bf16 weight-gradient kernels write per-split partial sums ("slabs") and a
 * reduce launch adds them into dw (and db); with the array the gradients accumulate into registered here ([dw_base, dw_base +
 * dw_bytes): a WeightNormBank's flat dW, the optimizer's gradient arena) and the arena armed by _begin, each such call writes
 * its slabs to a persistent slot in caller-owned `storage` (256-byte aligned) and launches no reduce; _reduce adds every slab
 * set written since _begin into its dw / db in ONE launch and disarms.  Call _reduce before anything reads the gradients.  A
 * second gradient into the same dw within a phase, exhausted storage, a first sighting during stream capture, or a bias gradient
 * outside every arena of the call's context reduce immediately as before.  The handle goes into ttts_conv_ctx::handles (v10). */
int ttts_conv_wgrad_arena_create(const void* dw_base, int64_t dw_bytes, void* storage, int64_t storage_bytes, int32_t max_entries,
                                 void** arena_out);
int ttts_conv_wgrad_arena_begin(void* arena);
int ttts_conv_wgrad_arena_reduce(void* arena, void* stream);
int ttts_conv_wgrad_arena_disarm(void* arena);      /* abandon the phase: nothing is added, later calls reduce immediately */
/* ABI v10.  An entry handed out DURING A STREAM CAPTURE belongs to the recorded graph (its weight-gradient launch holds the slab
 * pointer and split count, the reduce launch reads both from the device-side table at replay), so it is frozen: later calls that
 * would need another split count or a larger slot reduce immediately (a fallback, counted) instead of rewriting it.  The caller
 * declares "no graph recorded over this arena will be replayed any more" with _release_graphs (e.g. before it re-records its step
 * for a new batch shape); `generation` (stats[5]) counts every change of the table, for callers that prefer to check. */
int ttts_conv_wgrad_arena_release_graphs(void* arena);
int ttts_conv_wgrad_arena_stats(void* arena, int64_t* out6 /* entries, storage bytes used, deferred, fallbacks, partial reduces, generation */);
int ttts_conv_wgrad_arena_destroy(void* arena);
/* ABI v11: range bookkeeping of TTTS_CONV_F16X1 and the dynamic loss scale built on it.  Replaces: torch.cuda.amp.GradScaler
 * (ttts/vqvae/train.py:262 `GradScaler(enabled=hps.train.fp16_run)`, :356-372 scale / unscale_ / step / update) for the
 * single-pass fp16 convolution mode, whose data gradients are loss-scaled into fp16's range.
 * Every fp32 -> fp16 operand conversion of that mode counts, per device, the threads (8-16 neighbouring elements each) that saw
 *   [0] a value above 65504 (saturated; +-inf included; NaN stays NaN and is not counted), [1] a non-zero value that became zero
 *   (|v| <= 2^-25), [2] reserved, always 0 (values below fp16's smallest normal 2^-14 keep fewer than 11 significant bits; they
 *   are common -- millions per step -- and are deliberately not counted: the mode's stated accuracy, 1.5e-3 of a convolution's
 *   output range, is measured WITH them, tests/test_gpu_vqvae.py::test_tf32class_conv_accuracy).
 * ttts_conv_f16_events: events3[i] += counter[i] on `stream` (device memory, int32[3]); reset != 0 clears the counters.  The
 *   counters belong to the device `stream` runs on.  Graph-capturable; no host sync.
 * ttts_loss_scale_check: one per backward, after ttts_conv_f16_events (and after an all-reduce of events3 under data parallelism):
 *   events3[0] != 0 -> *skip = 1 (pass &optimizer_state[6], see ttts_adamw_schedule) and "overflow seen" in ls8; totals += events;
 *   events3 is cleared.  ls8 = f32[8] {scale, 1 / scale, clean steps in a row, overflow since the last update, saturation total,
 *   flush total, skipped optimizer steps, reserved}; the caller initialises {scale, 1 / scale, 0...}.
 * ttts_loss_scale_update: one per step: overflow -> scale *= backoff (not below 1), else after growth_interval clean steps in a
 *   row scale *= growth (not above 2^24); 1 / scale follows.  GradScaler's defaults: backoff 0.5, growth 2, interval 2000. */
int ttts_conv_f16_events(int32_t* events3, int32_t reset, void* stream);
int ttts_loss_scale_check(float* ls8, int32_t* events3, float* skip, void* stream);
int ttts_loss_scale_update(float* ls8, int32_t growth_interval, float backoff, float growth, void* stream);
int ttts_tanh_bwd_f32(const float* dy, const float* y, float* dx, int64_t n, void* stream);
int ttts_lrelu_bwd_f32(const float* dy, const float* y, float* dx, float slope, int64_t n, void* stream);
/* y = scale * (a + b + c + d), b/c/d optional (NULL): `xs / num_kernels` of Generator.forward (vq2.py:396-403) and its
 * gradient (a = dy, scale = 1/num_kernels).  Pointers 16-byte aligned. */
int ttts_add4_scale_f32(const float* a, const float* b, const float* c, const float* d, float scale, float* y,
                        int64_t n, void* stream);

/* ---- elementwise / small kernels of the VQ-VAE-GAN generator stacks (fp32, (B, C, T)) ---------------------------
 * gate: x [B,2H,T] -> y [B,H,T]; kind 0 = tanh(a) sigmoid(b) (commons.fused_add_tanh_sigmoid_multiply,
 *       ttts/utils/commons.py:103-109; the add is the conv's bbias), kind 1 = a sigmoid(b) (GLU, modules.py Conv1dGLU).
 * mul_mask: y = x * mask[b][t] (x_mask multiplies of vq2.py / modules.py).
 * gauss_sample: stats [B,2C,T] = (m | logs); z = (m + eps exp(logs)) mask (vq2.py:742-744); bwd writes dstats.
 * upsample2: F.interpolate(scale 2, nearest) (vq2.py:853-855); n = number of INPUT elements.
 * act: relu / mish;  dropout: keep iff 16 hash bits >= round(p 65536), scaled 1/(1-p); same call on dy = backward.
 * snake_aa: Activation1d(SnakeBeta(alpha_logscale)) -- kaiser-sinc x2 upsample, x + sin^2(x e^alpha)/(e^beta + 1e-9),
 *       low-pass x2 downsample (alias_free_torch/act.py:8-28, resample.py, filter.py, activations.py:62-119);
 *       filters are the modules' 12-tap buffers; bwd accumulates dalpha/dbeta [C].  8 <= T <= 3072.
 * layernorm_ch: modules.LayerNorm (modules.py:19-31): LayerNorm over C at every (b, t); mean/rstd [B*T] saved. */
#define TTTS_ACT_RELU 0
#define TTTS_ACT_MISH 1
#define TTTS_ACT_SILU 2   /* x sigmoid(x): the diffusion model's nn.SiLU (ttts/diffusion/aa_model.py:98,103,112) */
int ttts_gate_fwd_f32(const float* x, float* y, int32_t B, int32_t H, int32_t T, int32_t kind, void* stream);
int ttts_gate_bwd_f32(const float* dy, const float* x, float* dx, int32_t B, int32_t H, int32_t T, int32_t kind,
                      void* stream);
/* (v11 addition) gate_bwd that also writes rowsum f32 [B, 2H] = sum_t dx[b][c][t]: the conditioning gradient of a WaveNet layer
 * (ttts/vqvae/modules.py:194-201: g_l is added to x_in before the gate, so d g_l = sum over t of d x_in) without a second pass. */
int ttts_gate_bwd_rowsum_f32(const float* dy, const float* x, float* dx, float* rowsum, int32_t B, int32_t H, int32_t T,
                             int32_t kind, void* stream);
int ttts_mul_mask_f32(const float* x, const float* mask, float* y, int32_t B, int32_t C, int32_t T, void* stream);
int ttts_gauss_sample_fwd_f32(const float* stats, const float* eps, const float* mask, float* z, int32_t B, int32_t C,
                              int32_t T, void* stream);
int ttts_gauss_sample_bwd_f32(const float* dz, const float* stats, const float* eps, const float* mask, float* dstats,
                              int32_t B, int32_t C, int32_t T, int32_t accumulate, void* stream);
int ttts_upsample2_fwd_f32(const float* x, float* y, int64_t n, void* stream);
int ttts_upsample2_bwd_f32(const float* dy, float* dx, int64_t n, void* stream);
int ttts_act_fwd_f32(const float* x, float* y, int64_t n, int32_t op, void* stream);
int ttts_act_bwd_f32(const float* dy, const float* x, float* dx, int64_t n, int32_t op, void* stream);
int ttts_dropout_f32(const float* x, float* y, int64_t n, float p, uint64_t seed, const uint32_t* dropout_counter,
                     void* stream);
int ttts_snake_aa_fwd_f32(const float* x, const float* alpha, const float* beta, const float* up_filter,
                          const float* down_filter, float* y, int32_t B, int32_t C, int32_t T, void* stream);
int ttts_snake_aa_bwd_f32(const float* dy, const float* x, const float* alpha, const float* beta,
                          const float* up_filter, const float* down_filter, float* dx, float* dalpha, float* dbeta,
                          int32_t B, int32_t C, int32_t T, void* stream);
int ttts_layernorm_ch_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                              int32_t B, int32_t C, int32_t T, float eps, void* stream);
int ttts_layernorm_ch_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                              float* dx, float* dgamma, float* dbeta, int32_t B, int32_t C, int32_t T, void* stream);

/* embedding_ct: nn.Embedding + transpose(1, 2) (vq2.py:156): y[b][c][t] = table[idx[b][t]][c]; bwd accumulates dtable.
 * masked_mean: MelStyleEncoder.temporal_avg_pool (modules.py:726-734): y[b][c] = sum_t x m / sum_t m (mask NULL: mean). */
int ttts_embedding_ct_fwd_f32(const int64_t* idx, const float* table, float* y, int32_t B, int32_t C, int32_t T,
                              void* stream);
int ttts_embedding_ct_bwd_f32(const int64_t* idx, const float* dy, float* dtable, int32_t B, int32_t C, int32_t T,
                              void* stream);
int ttts_masked_mean_fwd_f32(const float* x, const float* mask, float* y, int32_t B, int32_t C, int32_t T, void* stream);
int ttts_masked_mean_bwd_f32(const float* dy, const float* mask, float* dx, int32_t B, int32_t C, int32_t T, void* stream);

/* ---- fp32 attention pieces of the VQ-VAE text / style encoders ------------------------------------------------------
 * Replaces: attentions.MultiHeadAttention.attention (ttts/vqvae/attentions.py:239-289, relative window 4), the MRTE
 * cross-attention (ttts/utils/vc_utils.py:571-627) and ScaledDotProductAttention (ttts/vqvae/modules.py:664-683).
 * bgemm:  C[z][m][n] = alpha sum_k A[z][m][k] B[z][k][n] + beta C, every operand addressed by element strides
 *         (s_m, s_k / s_k, s_n / s_m, s_n) and a two-level batch z = (outer, inner) with its own strides -- heads are
 *         contiguous d_k-row blocks of (B, C, T) tensors, so no transposes are materialised.
 * attn_softmax_fwd: in place on scores [B,H,Tq,Tk]: add the relative-key logits scale <q_i, Ek[j-i+w]> (|j-i| <= w),
 *         fill where qmask[b][i] kmask[b][j] == 0, softmax over j.  q [B, H dk, Tq]; emb_rel_k [heads_rel, 2w+1, dk].
 * attn_softmax_bwd: in place on dP: dS = [mask] P (dP - sum_j dP P).
 * attn_rel (T = Tq = Tk): mode 0: X[b,h,d,i] += scale sum_r W[i][i+r] E[r+w][d];  mode 1: W[i][i+r] += scale
 *         sum_d X[b,h,d,i] E[r+w][d];  mode 2: E[r+w][d] += scale sum_{b,h,i} W[i][i+r] X[b,h,d,i]. */
int ttts_bgemm_f32(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K, int64_t a_sm,
                   int64_t a_sk, int64_t b_sk, int64_t b_sn, int64_t c_sm, int64_t c_sn, int32_t batch_outer,
                   int32_t batch_inner, int64_t a_so, int64_t a_si, int64_t b_so, int64_t b_si, int64_t c_so,
                   int64_t c_si, float alpha, float beta, void* stream);
int ttts_attn_softmax_fwd_f32(float* scores, const float* q, const float* emb_rel_k, const float* qmask,
                              const float* kmask, int32_t B, int32_t H, int32_t Tq, int32_t Tk, int32_t dk,
                              int32_t window, int32_t heads_rel, float scale, float fill, void* stream);
int ttts_attn_softmax_bwd_f32(float* dP, const float* P, const float* qmask, const float* kmask, int32_t B, int32_t H,
                              int32_t Tq, int32_t Tk, void* stream);
int ttts_attn_rel_f32(float* W, float* X, float* E, int32_t B, int32_t H, int32_t T, int32_t dk, int32_t window,
                      int32_t heads_rel, float scale, int32_t mode, void* stream);

/* ---- loss reductions of the VQ-VAE-GAN step ---------------------------------------------------------------------
 * Replaces: feature_loss / discriminator_loss / generator_loss / kl_loss (ttts/vqvae/losses.py:7-61) and
 * F.l1_loss(y_mel, y_hat_mel) (ttts/vqvae/train.py:389).  Deterministic two-stage sums; results stay on the device.
 * reduce_loss: out[0] = [out[0] +] scale * sum_i term(a_i, b_i); term = |a-b| (ABSDIFF), (1-a)^2, a^2.
 * reduce_loss_bwd: d = [d +] gout[0] * scale * dterm  -- ABSDIFF: w.r.t. b; the squares: w.r.t. a; gout NULL = 1.
 * kl_loss: tensors [B,C,T], mask [B,1,T]; out[0] = sum(kl * mask) / sum(mask), out[1] = sum(mask) (kept for bwd).
 * workspace: ttts_loss_workspace_bytes() bytes. */
#define TTTS_RED_ABSDIFF 0
#define TTTS_RED_SQ_ONE_MINUS 1
#define TTTS_RED_SQ 2
int64_t ttts_loss_workspace_bytes(void);
int ttts_reduce_loss_f32(const float* a, const float* b, int64_t n, int32_t mode, float scale, float* out,
                         int32_t accumulate, void* workspace, void* stream);
int ttts_reduce_loss_bwd_f32(const float* a, const float* b, int64_t n, int32_t mode, float scale, const float* gout,
                             float* d, int32_t accumulate, void* stream);
int ttts_kl_loss_fwd_f32(const float* z_p, const float* logs_q, const float* m_p, const float* logs_p,
                         const float* mask, int32_t B, int32_t C, int32_t T, float* out, void* workspace, void* stream);
int ttts_kl_loss_bwd_f32(const float* z_p, const float* logs_q, const float* m_p, const float* logs_p,
                         const float* mask, const float* out, const float* gout, int32_t B, int32_t C, int32_t T,
                         float* dz_p, float* dlogs_q, float* dm_p, float* dlogs_p, void* stream);

/* ---- probes (tests only): dump hardware fragment layouts the kernels rely on ------------------------ */
/* out_c f32 [64 lanes][16 regs]: raw accumulators of one 32x32x16 bf16 MFMA with D[i][j] = (i+1) + 64*(j+1);
 * out_tr i32 [64 lanes][8]: the uint16 LDS element indices two ds_read_b64_tr_b16 return for the kernels' address map. */
int ttts_probe_mfma_layout(float* out_c, int32_t* out_tr, void* stream);

/* ---- FP8 (OCP e4m3) matrix-core GEMMs (ABI v10; csrc/fp8_gemm.hip) ------------------------------------------------------------
 * The 1 x 1 convolutions / linear layers of the diffusion mel-denoiser step in BASELINE config #5's arithmetic ("bf16 + fp8 MFMA
 * GEMMs"): nn.Conv1d(k = 1) of AttentionBlock.qkv / .proj_out (ttts/utils/utils.py:172-215), ResBlock.in_layers[2]
 * (ttts/diffusion/aa_model.py:70-131) and AA_diffusion.integrating_conv (:228), forward + data gradient + weight gradient.
 * Per-TENSOR current scaling: amax = max |x| of the tensor being quantised (ttts_fp8_amax_f32, a device scalar), q = e4m3(x * 448 /
 * amax) (round to nearest even, clamped to +-448; amax == 0: scale 1), products on v_mfma_f32_32x32x16_fp8_fp8, fp32 accumulation,
 * result scaled by amax_a amax_b / 448^2 (read from device memory by the GEMM: no host round trip, capturable).
 *  _quant:            q [rows][cols_pad] from x [rows][cols] (reduction axis already contiguous; zero padded)
 *  _quant_transpose:  q [B][T][Cp] from x [B][C][T] (the reduction axis of a (B, C, T) activation made contiguous; zero padded)
 *  _gemm_nt:          Y[go][m][n] (+)= alpha sum_{gi} sum_k A[go][gi][m][k] B[go][gi][n][k] (+ bias[m]) (+ resid[go][m][n]);
 *                     K a multiple of 64, operands / pitches / group strides 16-byte aligned (strides in bytes = elements),
 *                     Y / resid strides in elements: Y[go * y_stride_outer + m * y_stride_m + n * y_stride_n].
 *  _quant_both:       both layouts of one (B, C, T) tensor in one pass (Cp, Tp multiples of 64)
 *  _gemm_nt workspace: weight-gradient shaped calls (few output tiles, many inner groups) split the inner groups over several
 *                     workgroups per tile when `workspace` (>= ttts_fp8_gemm_nt_workspace_bytes(...), 16-byte aligned) is given; the
 *                     partial slabs are summed in a fixed order by a second launch (deterministic).  NULL: never split.
 * Oracle: oracle/fp8_ref.py (same scales and rounding, exact sums): results differ by the matrix core's internal summation of the
 * 16 products of an instruction (measured 1.6e-5 of the output range; the tests hold 6e-5). */
/* out_is_zero != 0: *amax_out is known to be zero (e.g. a fresh word of a pool the caller cleared in bulk): no clearing launch */
int ttts_fp8_amax_f32(const float* x, int64_t n, float* amax_out, int32_t out_is_zero, void* stream);
int ttts_fp8_quant_f32(const float* x, void* q, const float* amax, int64_t rows, int32_t cols, int32_t cols_pad, void* stream);
int ttts_fp8_quant_transpose_f32(const float* x, void* q, const float* amax, int32_t B, int32_t C, int32_t T, int32_t Cp, void* stream);
int ttts_fp8_quant_both_f32(const float* x, void* q_rows, void* q_t, const float* amax, int32_t B, int32_t C, int32_t T, int32_t Cp,
                            int32_t Tp, void* stream);
int64_t ttts_fp8_gemm_nt_workspace_bytes(int32_t M, int32_t N, int32_t groups_outer, int32_t groups_inner);
int ttts_fp8_gemm_nt(const void* a, const void* b, float* y, const float* bias, const float* resid, const float* amax_a,
                     const float* amax_b, int32_t M, int32_t N, int32_t K, int32_t groups_outer, int32_t groups_inner,
                     int64_t lda, int64_t ldb, int64_t a_stride_outer, int64_t a_stride_inner, int64_t b_stride_outer,
                     int64_t b_stride_inner, int64_t y_stride_outer, int64_t y_stride_m, int64_t y_stride_n,
                     int32_t accumulate, void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TTTS_HIP_H */
