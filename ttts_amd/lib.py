"""ctypes binding of libttts_hip.so (C ABI: include/ttts_hip.h) and the in-tree build recipe.

The library is built IN-TREE (`ttts_amd/libttts_hip.so`, git-ignored, shipped to the GPU box with the snapshot) by
`build()`: one `hipcc --offload-arch=gfx950 -c` per kernel file, then a shared link.  `get()` loads it and FAILS
LOUDLY if it is missing -- there is no fallback path.
"""
import ctypes
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(CSRC, "build")
SO_PATH = os.environ.get("TTTS_LIB") or os.path.join(HERE, "libttts_hip.so")   # TTTS_LIB: an alternate build of the same ABI (same-box A/B runs: tools/gpu_ab_lib.sh)
SOURCES = ["lib.hip", "elementwise.hip", "gemm.hip", "attn.hip", "attn_dh64.hip", "vq.hip", "stft.hip", "conv.hip", "conv_mfma.hip", "conv_thin.hip", "conv_grouped.hip", "losses.hip", "vqvae_ops.hip", "attn_f32.hip", "attn_cross.hip", "peq.hip", "decode.hip", "diffusion_ops.hip", "fp8_gemm.hip", "attn_relpos.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC", "-Wall", "-Wno-unused-variable"]

# per-file additions (reasons in the file headers)
EXTRA_FLAGS = {"attn_dh64.hip": ["-fno-honor-nans", "-fno-slp-vectorize"]}

TTTS_OK = 0
EPI_STORE_BF16, EPI_GELU_BF16, EPI_RESID_ADD_F32, EPI_DGELU_BF16, EPI_STORE_F32 = range(5)


class TttsError(RuntimeError):
    pass


def _needs_rebuild(obj, deps):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link ttts_amd/libttts_hip.so (cross-compiles without a GPU)."""
    os.makedirs(BUILD, exist_ok=True)
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp"))   # every source sees every header
    headers.append(os.path.join(os.path.dirname(HERE), "include", "ttts_hip.h"))
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(BUILD, src.replace(".hip", ".o"))
        if force or _needs_rebuild(o, [s] + headers):
            jobs.append([HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise TttsError("hipcc failed: %s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(BUILD, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _needs_rebuild(SO_PATH, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO_PATH] + objs)
    return SO_PATH


_c = ctypes
_P, _I32, _I64, _U64, _F = _c.c_void_p, _c.c_int32, _c.c_int64, _c.c_uint64, _c.c_float


ABI_VERSION = 11


class ConvCtx(ctypes.Structure):
    """include/ttts_hip.h: ttts_conv_ctx (caller-owned; the library keeps no copy)."""
    _fields_ = [("workspace", _P), ("workspace_bytes", _I64), ("flags", _I32), ("n_handles", _I32),
                ("handles", ctypes.POINTER(ctypes.c_void_p)), ("device", _I32), ("reserved", _I32)]


class LnFinalizeDesc(ctypes.Structure):
    _fields_ = [("workspace", _P), ("dgamma", _P), ("dbeta", _P), ("dcolsum", _P)]


class ColsumDesc(ctypes.Structure):
    _fields_ = [("X", _P), ("out", _P), ("ldx", _I64), ("M", _I32), ("N", _I32), ("tile_begin", _I32), ("reserved", _I32)]


class WnDesc(ctypes.Structure):
    """include/ttts_hip.h: ttts_wn_desc."""
    _fields_ = [("v", _P), ("g", _P), ("w", _P), ("norm", _P), ("dw", _P), ("dv", _P), ("dg", _P),
                ("rows", _I32), ("n", _I32), ("row_begin", _I32), ("reserved", _I32)]


class GemmNtPlan(ctypes.Structure):
    _fields_ = [("kernel", _I32), ("grid", _I32), ("block", _I32), ("tile_m", _I32), ("tile_n", _I32), ("phase", _I32),
                ("main_row_tiles", _I32), ("tail_tile_rows", _I32)]


NT_KERNEL_REG, NT_KERNEL_DMA64, NT_KERNEL_DMA32, NT_KERNEL_RING160, NT_KERNEL_WAVE8, NT_KERNEL_WAVE8_SPLIT, NT_KERNEL_WREG = range(7)


class TransposeDesc(ctypes.Structure):
    _fields_ = [("src", _P), ("dst", _P), ("rows", _I32), ("cols", _I32), ("ldd", _I32), ("tile_begin", _I32)]


class CastDesc(ctypes.Structure):
    _fields_ = [("src", _P), ("dst", _P), ("dst_t", _P), ("rows", _I32), ("cols", _I32), ("tile_begin", _I32),
                ("ldt", _I32)]


class TnDesc(ctypes.Structure):
    """include/ttts_hip.h: ttts_tn_desc (one problem of the grouped weight-gradient GEMM)."""
    _fields_ = [("At", _P), ("Bt", _P), ("C", _P), ("ldat", _I64), ("ldbt", _I64), ("ldc", _I64), ("Mo", _I32),
                ("No", _I32), ("Kr", _I32), ("tile_begin", _I32)]


# name -> (restype, argtypes); every symbol include/ttts_hip.h declares
SIGNATURES = {
    "ttts_abi_version": (_I32, []),
    "ttts_last_error": (_c.c_char_p, []),
    "ttts_device_info": (_I32, [_P]),
    "ttts_gemm_nt_bf16": (_I32, [_P, _I64, _P, _I64, _P, _I64, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "ttts_gemm_nt_bf16_ex": (_I32, [_P, _I64, _P, _I64, _P, _I64, _P, _P, _I32, _I32, _I32, _I32, _P, _F, _U64, _P, _P, _P]),
    "ttts_gemm_nt_plan_query": (_I32, [_I32, _I32, _I32, _I32, _P]),
    "ttts_gemm_nt_resid_ln_bf16": (_I32, [_P, _I64, _P, _I64, _P, _P, _P, _I32, _I32, _I32, _F, _U64, _P, _P, _P, _F, _P, _I32, _P, _P, _P]),
    "ttts_gemm_tn_workspace_bytes": (_I64, [_I32, _I32, _I32]),
    "ttts_gemm_tn_bf16_accum_f32": (_I32, [_P, _I64, _P, _I64, _P, _I64, _I32, _I32, _I32, _P, _P]),
    "ttts_tn_desc_tiles": (_I32, [_I32, _I32]),
    "ttts_tn_desc_prepare": (_I32, [_P, _I32, _P]),
    "ttts_gemm_tn_grouped_bf16_accum_f32": (_I32, [_P, _I32, _I32, _P]),
    "ttts_colsum_bf16_accum_f32": (_I32, [_P, _I64, _P, _I32, _I32, _P]),
    "ttts_layernorm_bwd_finalize_batched": (_I32, [_P, _I32, _I32, _I32, _P]),
    "ttts_colsum_desc_tiles": (_I32, [_I32, _I32]),
    "ttts_colsum_bf16_accum_f32_batched": (_I32, [_P, _I32, _I32, _P]),
    "ttts_transpose_desc_tiles": (_I32, [_I32, _I32]),
    "ttts_transpose_bf16_batched": (_I32, [_P, _I32, _I32, _P]),
    "ttts_cast_desc_tiles": (_I32, [_I32, _I32]),
    "ttts_cast_bf16_batched": (_I32, [_P, _I32, _I32, _P]),
    "ttts_attn_causal_fwd_bf16": (_I32, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I64, _I64, _I64, _I64, _F, _F,
                                         _U64, _P, _P]),
    "ttts_attn_bwd_workspace_bytes": (_I64, [_I32, _I32, _I32]),
    "ttts_attn_causal_bwd_bf16": (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I64, _I64,
                                         _I64, _I64, _F, _F, _U64, _P, _P]),
    "ttts_attn_cross_fwd_f32": (_I32, [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _F, _F, _P]),
    "ttts_attn_cross_bwd_workspace_bytes": (_I64, [_I32, _I32, _I32]),
    "ttts_attn_cross_stats_bytes": (_I64, [_I32, _I32, _I32]),
    "ttts_attn_cross_bwd_f32": (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _F, _F, _P]),
    "ttts_attn_dropout_mask_u8": (_I32, [_P, _I32, _I32, _I32, _F, _U64, _P, _P]),
    "ttts_layernorm_fwd": (_I32, [_P, _P, _P, _P, _I32, _P, _P, _I32, _I32, _F, _I32, _I32, _P]),
    "ttts_layernorm_bwd_workspace_bytes": (_I64, [_I32, _I32]),
    "ttts_layernorm_bwd": (_I32, [_P, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "ttts_layernorm_bwd_ex": (_I32, [_P, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _F,
                                     _U64, _P, _P]),
    "ttts_gpt_embed_fwd": (_I32, [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _F, _U64, _P, _P]),
    "ttts_gpt_prepare_tokens": (_I32, [_P, _I64, _P, _I64, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P]),
    "ttts_gpt_embed_bwd": (_I32, [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _F, _U64, _P, _P]),
    "ttts_ce_fwd_bf16": (_I32, [_P, _I64, _P, _P, _P, _P, _I32, _I32, _P]),
    "ttts_ce_bwd_bf16": (_I32, [_P, _I64, _P, _P, _P, _F, _P, _I32, _I32, _P]),
    "ttts_adamw_schedule": (_I32, [_P, _F, _F, _F, _I32, _P]),
    "ttts_gradnorm_workspace_bytes": (_I64, [_I64]),
    "ttts_gradnorm_f32": (_I32, [_P, _I64, _F, _P, _P, _P]),
    "ttts_adamw_f32": (_I32, [_P, _P, _P, _P, _P, _I64, _P, _F, _F, _F, _F, _I32, _P]),
    "ttts_vq_workspace_bytes": (_I64, [_I32, _I32]),
    "ttts_vq_nearest_f32": (_I32, [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P]),
    "ttts_vq_commit_f32": (_I32, [_P, _P, _P, _P, _F, _I32, _I32, _P, _P]),
    "ttts_vq_ema_workspace_bytes": (_I64, [_I32, _I32]),
    "ttts_vq_ema_update_f32": (_I32, [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _F, _F, _P]),
    "ttts_stft_twiddle_host": (_I32, [_P, _I32]),
    "ttts_stft_mag_fwd_f32": (_I32, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "ttts_mel_log_fwd_f32": (_I32, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "ttts_mel_log_bwd_f32": (_I32, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "ttts_stft_mag_bwd_f32": (_I32, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "ttts_groupnorm_fwd_f32": (_I32, [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _F, _I32, _P]),
    "ttts_groupnorm_bwd_f32": (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "ttts_relpos_bias_fwd_f32": (_I32, [_P, _P, _P, _I32, _I32, _I32, _I32, _F, _P]),
    "ttts_relpos_bias_bwd_workspace_bytes": (_I64, [_I32, _I32, _I32, _I32]),
    "ttts_relpos_bias_bwd_f32": (_I32, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _F, _I32, _P]),
    "ttts_softmax_bias_fwd_f32": (_I32, [_P, _P, _I32, _I32, _I32, _I32, _P]),
    "ttts_attn_relpos_max_t": (_I32, [_I32]),
    "ttts_attn_relpos_workspace_bytes": (_I64, [_I32, _I32, _I32]),
    "ttts_attn_relpos_fwd_f32": (_I32, [_P, _P, _P, _I32, _P, _P, _I32, _I32, _I32, _I32, _F, _I32, _P]),
    "ttts_attn_relpos_bwd_f32": (_I32, [_P, _P, _P, _I32, _P, _P, _P, _P, _P, _I32, _P, _I32, _I32, _I32, _I32, _I32, _F, _I32, _P]),
    "ttts_interp_nearest_fwd_f32": (_I32, [_P, _P, _I64, _I32, _I32, _P]),
    "ttts_interp_nearest_bwd_f32": (_I32, [_P, _P, _I64, _I32, _I32, _P]),
    "ttts_timestep_embedding_f32": (_I32, [_P, _P, _P, _I32, _I32, _P]),
    "ttts_select_rows_fwd_f32": (_I32, [_P, _P, _P, _P, _I32, _I32, _I32, _P]),
    "ttts_select_rows_bwd_f32": (_I32, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "ttts_q_sample_f32": (_I32, [_P, _P, _P, _P, _P, _I32, _I64, _P]),
    "ttts_diffusion_loss_workspace_bytes": (_I64, [_I32]),
    "ttts_diffusion_loss_fwd_f32": (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P]),
    "ttts_diffusion_loss_bwd_f32": (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P]),
    "ttts_linear_decode_bf16": (_I32, [_P, _I64, _I32, _P, _P, _P, _P, _P, _I64, _P, _P, _I64, _P, _I32, _I32, _I32, _I32, _P]),
    "ttts_decode_embed_f32": (_I32, [_P, _P, _P, _P, _I32, _P, _I32, _I32, _I32, _I32, _P]),
    "ttts_kv_cache_fill_bf16": (_I32, [_P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "ttts_attn_decode_bf16": (_I32, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _F, _P]),
    "ttts_sample_logits_f32": (_I32, [_P, _I64, _I32, _I32, _I32, _P, _I64, _I32, _P, _P, _P, _I64, _P, _F, _F, _F, _I32, _F,
                                      _I32, _I32, _I32, _U64, _P, _P, _P]),
    "ttts_decode_advance": (_I32, [_P, _P, _I32, _P]),
    "ttts_peq_response_f32": (_I32, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _F, _P]),
    "ttts_stft_center_frames": (_I32, [_I32, _I32]),
    "ttts_stft_filter_frames_f32": (_I32, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "ttts_istft_ola_f32": (_I32, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "ttts_peak_scale_f32": (_I32, [_P, _P, _I32, _I32, _F, _P]),
    "ttts_conv1d_fwd_f32": (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32,
                                   _F, _F, _I32, _F, _F, _I32, _P, _P]),
    "ttts_conv1d_dgrad_f32": (_I32, [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32,
                                     _F, _F, _F, _I32, _P, _P]),
    "ttts_conv1d_wgrad_f32": (_I32, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _F, _F,
                                     _P, _P]),
    "ttts_lrelu_bwd_f32": (_I32, [_P, _P, _P, _F, _I64, _P]),
    "ttts_conv1d_bias_grad_f32": (_I32, [_P, _P, _I32, _I32, _I32, _P]),
    "ttts_weight_norm_fwd_f32": (_I32, [_P, _P, _P, _P, _P, _I32, _I32, _P]),
    "ttts_weight_norm_bwd_f32": (_I32, [_P, _P, _P, _P, _P, _P, _I32, _I32, _P]),
    "ttts_weight_norm_fwd_batched_f32": (_I32, [_P, _I32, _I32, _P]),
    "ttts_weight_norm_bwd_batched_f32": (_I32, [_P, _I32, _I32, _P]),
    "ttts_conv1d_fwd_dual_f32": (_I32, [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _F, _I32, _P, _P]),
    "ttts_conv_wsplit_cache_create": (_I32, [_P, _I64, _P, _I64, _I32, ctypes.POINTER(ctypes.c_void_p)]),
    "ttts_conv_wsplit_cache_refresh": (_I32, [_P, _P]),
    "ttts_conv_wsplit_cache_disarm": (_I32, [_P]),
    "ttts_conv_wsplit_cache_stats": (_I32, [_P, ctypes.POINTER(ctypes.c_int64)]),
    "ttts_conv_wsplit_cache_destroy": (_I32, [_P]),
    "ttts_conv_wgrad_arena_create": (_I32, [_P, _I64, _P, _I64, _I32, ctypes.POINTER(ctypes.c_void_p)]),
    "ttts_conv_wgrad_arena_begin": (_I32, [_P]),
    "ttts_conv_wgrad_arena_reduce": (_I32, [_P, _P]),
    "ttts_conv_wgrad_arena_disarm": (_I32, [_P]),
    "ttts_conv_wgrad_arena_stats": (_I32, [_P, ctypes.POINTER(ctypes.c_int64)]),
    "ttts_conv_wgrad_arena_release_graphs": (_I32, [_P]),
    "ttts_conv_wgrad_arena_destroy": (_I32, [_P]),
    "ttts_conv_f16_events": (_I32, [_P, _I32, _P]),
    "ttts_loss_scale_check": (_I32, [_P, _P, _P, _P]),
    "ttts_loss_scale_update": (_I32, [_P, _I32, _F, _F, _P]),
    "ttts_tanh_bwd_f32": (_I32, [_P, _P, _P, _I64, _P]),
    "ttts_add4_scale_f32": (_I32, [_P, _P, _P, _P, _F, _P, _I64, _P]),
    "ttts_gate_fwd_f32": (_I32, [_P, _P, _I32, _I32, _I32, _I32, _P]),
    "ttts_gate_bwd_f32": (_I32, [_P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "ttts_gate_bwd_rowsum_f32": (_I32, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "ttts_mul_mask_f32": (_I32, [_P, _P, _P, _I32, _I32, _I32, _P]),
    "ttts_gauss_sample_fwd_f32": (_I32, [_P, _P, _P, _P, _I32, _I32, _I32, _P]),
    "ttts_gauss_sample_bwd_f32": (_I32, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "ttts_upsample2_fwd_f32": (_I32, [_P, _P, _I64, _P]),
    "ttts_upsample2_bwd_f32": (_I32, [_P, _P, _I64, _P]),
    "ttts_act_fwd_f32": (_I32, [_P, _P, _I64, _I32, _P]),
    "ttts_act_bwd_f32": (_I32, [_P, _P, _P, _I64, _I32, _P]),
    "ttts_dropout_f32": (_I32, [_P, _P, _I64, _F, _U64, _P, _P]),
    "ttts_snake_aa_fwd_f32": (_I32, [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P]),
    "ttts_snake_aa_bwd_f32": (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P]),
    "ttts_layernorm_ch_fwd_f32": (_I32, [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _F, _P]),
    "ttts_layernorm_ch_bwd_f32": (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P]),
    "ttts_embedding_ct_fwd_f32": (_I32, [_P, _P, _P, _I32, _I32, _I32, _P]),
    "ttts_embedding_ct_bwd_f32": (_I32, [_P, _P, _P, _I32, _I32, _I32, _P]),
    "ttts_masked_mean_fwd_f32": (_I32, [_P, _P, _P, _I32, _I32, _I32, _P]),
    "ttts_masked_mean_bwd_f32": (_I32, [_P, _P, _P, _I32, _I32, _I32, _P]),
    "ttts_bgemm_f32": (_I32, [_P, _P, _P, _I32, _I32, _I32, _I64, _I64, _I64, _I64, _I64, _I64, _I32, _I32, _I64, _I64, _I64,
                              _I64, _I64, _I64, _F, _F, _P]),
    "ttts_attn_softmax_fwd_f32": (_I32, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _F, _F, _P]),
    "ttts_attn_softmax_bwd_f32": (_I32, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "ttts_attn_rel_f32": (_I32, [_P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _F, _I32, _P]),
    "ttts_loss_workspace_bytes": (_I64, []),
    "ttts_reduce_loss_f32": (_I32, [_P, _P, _I64, _I32, _F, _P, _I32, _P, _P]),
    "ttts_reduce_loss_bwd_f32": (_I32, [_P, _P, _I64, _I32, _F, _P, _P, _I32, _P]),
    "ttts_kl_loss_fwd_f32": (_I32, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _P, _P, _P]),
    "ttts_kl_loss_bwd_f32": (_I32, [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P, _P, _P, _P, _P]),
    "ttts_probe_mfma_layout": (_I32, [_P, _P, _P]),
    "ttts_fp8_amax_f32": (_I32, [_P, _I64, _P, _I32, _P]),
    "ttts_fp8_quant_f32": (_I32, [_P, _P, _P, _I64, _I32, _I32, _P]),
    "ttts_fp8_quant_transpose_f32": (_I32, [_P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "ttts_fp8_quant_both_f32": (_I32, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "ttts_fp8_gemm_nt_workspace_bytes": (_I64, [_I32, _I32, _I32, _I32]),
    "ttts_fp8_gemm_nt": (_I32, [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I64, _I64, _I64, _I64, _I64, _I64, _I64,
                                _I64, _I64, _I32, _P, _P]),
}

_lib = None


def get():
    """Load libttts_hip.so (once).  Raises TttsError if it has not been built -- no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise TttsError("libttts_hip.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'`"
                            % SO_PATH)
        # The process must hold ONE HIP runtime: torch ships its own libamdhip64 and every tensor / stream we are handed lives in
        # it.  If this library were loaded first (build() then smoke() in one process) its kernels would bind to the system
        # runtime under /opt/rocm and launch into a runtime that has no context ("no ROCm-capable device is detected").
        try:
            import torch  # noqa: F401  (loads torch's HIP runtime first; harmless when it is already imported)
        except ImportError:
            pass
        lib = ctypes.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        if lib.ttts_abi_version() != ABI_VERSION:
            raise TttsError("libttts_hip.so ABI version mismatch (rebuild: python __graft_entry__.py)")
        _lib = lib
    return _lib


def check(rc, what=""):
    if rc != TTTS_OK:
        raise TttsError("%s failed (%d): %s" % (what or "ttts call", rc, get().ttts_last_error().decode()))
