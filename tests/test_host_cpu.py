"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol include/ttts_hip.h declares, argument
validation works without touching the device, host logic (token plumbing, parameter layout, FFT index math,
data-parallel exchange under gloo) matches the oracle / the reference-generated fixtures."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from ttts_amd import lib as l
    return l


def test_header_symbols_exported_and_bound(lib):
    hdr = open(os.path.join(ROOT, "include", "ttts_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ttts_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    handle = ctypes.CDLL(lib.SO_PATH)
    for name in declared:
        assert hasattr(handle, name), "header declares %s but libttts_hip.so does not export it" % name
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    assert lib.get().ttts_abi_version() == lib.ABI_VERSION == 11


def test_gemm_nt_dispatch_table(lib):
    """ttts_gemm_nt_plan_query: the kernel / grid the NT GEMM picks for the GPT step's shapes (B 8 x S 1156 = 9248 rows, d 512) and for
    the corner cases of each rule.  The rules are measurements (csrc/gemm.hip plan_nt, profiles/r03_ubench_nt_variants.txt);
    every kernel gives a shape the same output bits, so this table is about speed only -- but a silent change of it is a
    performance regression no numerics test would see."""
    def plan(M, N, K, epi=0):
        pl = lib.GemmNtPlan()
        assert lib.get().ttts_gemm_nt_plan_query(M, N, K, epi, ctypes.byref(pl)) == 0, lib.get().ttts_last_error()
        return pl
    E = lib
    # K = 512, wide N, many rows: the weights-in-registers kernel -- one persistent eight-wave workgroup per CU, 256-column panels x
    # (CUs / panels) row groups (round 4; c_attn 24.4 -> 22.8 us, c_fc + GELU 42.0 -> 33.4 us inside the step, same output bits)
    pl = plan(9248, 1536, 512)
    assert (pl.kernel, pl.grid, pl.block, pl.tile_m, pl.tile_n, pl.main_row_tiles) == (E.NT_KERNEL_WREG, 6 * 42, 512, 64, 256, 42)
    pl = plan(9248, 2048, 512, E.EPI_GELU_BF16)
    assert (pl.kernel, pl.grid, pl.block, pl.main_row_tiles) == (E.NT_KERNEL_WREG, 8 * 32, 512, 32)
    assert plan(9001, 2000, 512, E.EPI_GELU_BF16).kernel == E.NT_KERNEL_WREG       # last panel 208 of 256 columns: 2.4 % waste
    assert plan(8208, 1026, 512).kernel != E.NT_KERNEL_WREG                        # 5 panels for 1026 columns would waste 25 %
    assert plan(4000, 2048, 512).kernel != E.NT_KERNEL_WREG and plan(9248, 2048, 576).kernel != E.NT_KERNEL_WREG
    # dGELU stays on the tiled kernels (in the train step its saved pre-activation comes from HBM: no gain measured):
    # split grid = 32 x 16 tiles of 256 rows (one full round) + the remaining 1056 rows in 64-row tiles
    pl = plan(9248, 2048, 512, E.EPI_DGELU_BF16)
    assert (pl.kernel, pl.grid, pl.main_row_tiles, pl.tail_tile_rows) == (E.NT_KERNEL_WAVE8_SPLIT, 512 + 17 * 16, 32, 64)
    # the tiled kernels' rules, at a reduction length the register kernel does not take:
    # eight-wave 256 x 128 tiles, one round of the 512 slots, no stagger
    pl = plan(9248, 1536, 1024)
    assert (pl.kernel, pl.grid, pl.block, pl.tile_m, pl.phase) == (E.NT_KERNEL_WAVE8, 37 * 12, 512, 256, 0)
    # N = 2048: 592 eight-wave tiles would be 1.16 rounds -> the 128 x 128 kernel, 64-deep stages
    pl = plan(9248, 2048, 1024, E.EPI_GELU_BF16)
    assert (pl.kernel, pl.grid, pl.block) == (E.NT_KERNEL_DMA64, 73 * 16, 256)
    # dGELU at N = 2048: split grid = 32 x 16 tiles of 256 rows (one full round) + the remaining 1056 rows in 64-row tiles
    pl = plan(9248, 2048, 1024, E.EPI_DGELU_BF16)
    assert (pl.kernel, pl.grid, pl.main_row_tiles, pl.tail_tile_rows) == (E.NT_KERNEL_WAVE8_SPLIT, 512 + 17 * 16, 32, 64)
    assert pl.main_row_tiles * 256 + 17 * 64 >= 9248
    # a dGELU the split grid does not fit (few rows): 32-deep stages, staggered
    pl = plan(2000, 2048, 512, E.EPI_DGELU_BF16)
    assert (pl.kernel, pl.phase, pl.grid) == (E.NT_KERNEL_DMA32, 3, 16 * 16)
    # the four N = 512 GEMMs of a layer: 160 x 128 ring kernel, 58 x 4 tiles, one workgroup per CU
    for K, epi in ((2048, E.EPI_RESID_ADD_F32), (512, E.EPI_RESID_ADD_F32), (2048, 0), (1536, 0)):
        pl = plan(9248, 512, K, epi)
        assert (pl.kernel, pl.grid, pl.tile_m) == (E.NT_KERNEL_RING160, 58 * 4, 160)
    # heads: mel head forward (1026 classes) on eight waves; its dX GEMM runs over the logits pitch (1088 = up64(1026)) -> LDS-DMA
    assert plan(8208, 1026, 512).kernel == E.NT_KERNEL_WAVE8 and plan(8208, 1026, 512).grid == 33 * 9
    assert plan(8208, 512, 1088).kernel == E.NT_KERNEL_RING160
    assert plan(8208, 512, 1032).kernel == E.NT_KERNEL_REG            # the old pitch (up8): register-staged kernel
    # many rounds of eight-wave tiles: staggered start
    pl = plan(8192, 8194, 512)
    assert (pl.kernel, pl.grid, pl.phase) == (E.NT_KERNEL_WAVE8, 32 * 65, 8)
    # ragged K
    assert plan(1000, 264, 40).kernel == E.NT_KERNEL_REG and plan(1000, 264, 96).kernel == E.NT_KERNEL_DMA32
    # small launches stay on the 128 x 128 kernel
    assert plan(300, 200, 128).kernel == E.NT_KERNEL_DMA64 and plan(300, 200, 128).grid == 3 * 2
    # validation
    pl = lib.GemmNtPlan()
    assert lib.get().ttts_gemm_nt_plan_query(16, 16, 12, 0, ctypes.byref(pl)) == -1 and b"bad shape" in lib.get().ttts_last_error()
    assert lib.get().ttts_gemm_nt_plan_query(16, 16, 16, 9, ctypes.byref(pl)) == -1 and b"unknown epilogue" in lib.get().ttts_last_error()
    assert lib.get().ttts_gemm_nt_plan_query(16, 16, 16, 0, None) == -1


def test_gemm_nt_split_grid_covers_every_output_once(lib):
    """The split grid of the eight-wave NT kernel (csrc/gemm.hip, gemm_nt_glds_kernel<..., SPLIT>): workgroup -> (first row, rows,
    first column) restated here from the plan the library reports; every output element must belong to exactly one workgroup,
    the first `main` workgroups must be the 256-row tiles, and the XCD remap must stay a bijection inside each part."""
    def xcd_tile(bid, nblk):
        q, r, x = nblk >> 3, nblk & 7, bid & 7
        return (x * (q + 1) if x < r else r * (q + 1) + (x - r) * q) + (bid >> 3)
    for M, N in ((9248, 2048), (9001, 2000), (8200, 1990), (16385, 1024)):
        pl = lib.GemmNtPlan()
        assert lib.get().ttts_gemm_nt_plan_query(M, N, 512, lib.EPI_DGELU_BF16, ctypes.byref(pl)) == 0
        assert pl.kernel == lib.NT_KERNEL_WAVE8_SPLIT, (M, N, pl.kernel)
        tiles_n = -(-N // 128)
        main = pl.main_row_tiles * tiles_n
        assert main <= 512 and pl.grid - main <= 512          # one round of the chip's slots each
        cover = np.zeros((-(-M // 32), tiles_n), np.int32)     # 32-row granules x column tiles
        for bid in range(pl.grid):
            if bid < main:
                t = xcd_tile(bid, main); m0 = (t // tiles_n) * 256; rows = 256
            else:
                t = xcd_tile(bid - main, pl.grid - main); m0 = pl.main_row_tiles * 256 + (t // tiles_n) * pl.tail_tile_rows
                rows = pl.tail_tile_rows
            m1 = min(M, m0 + rows)
            assert m0 < M and m0 % 32 == 0
            cover[m0 // 32:-(-m1 // 32), t % tiles_n] += 1
        assert (cover == 1).all(), (M, N)


def test_gemm_nt_register_kernel_row_groups_cover_every_row_once(lib):
    """The weights-in-registers NT kernel (csrc/gemm.hip, gemm_nt_wreg_kernel): workgroup -> (column panel, row range) restated
    here from the plan the library reports -- rows are dealt to the row groups in 32-row units; every (row, panel) must belong to
    exactly one workgroup, no group may be empty, and the grid must fit one workgroup per CU."""
    for M, N in ((9248, 1536), (9248, 2048), (9001, 2000), (4096, 1024), (5000, 1280), (70000, 4096)):
        pl = lib.GemmNtPlan()
        assert lib.get().ttts_gemm_nt_plan_query(M, N, 512, 0, ctypes.byref(pl)) == 0
        assert pl.kernel == lib.NT_KERNEL_WREG, (M, N, pl.kernel)
        npan, groups = -(-N // 256), pl.main_row_tiles
        assert pl.grid == npan * groups and pl.grid <= 256 and pl.block == 512
        units = -(-M // 32)
        seen = np.zeros(M, np.int32)
        for g in range(groups):
            r0, r1 = (g * units // groups) * 32, min(M, ((g + 1) * units // groups) * 32)
            assert r0 < r1, (M, N, g)
            seen[r0:r1] += 1
        assert (seen == 1).all(), (M, N)


def test_argument_validation_without_gpu(lib):
    l = lib.get()
    assert l.ttts_gemm_nt_bf16(None, 8, None, 8, None, 8, None, None, 4, 4, 8, 0, None) == -1
    assert b"null pointer" in l.ttts_last_error()
    buf = (ctypes.c_char * 4096)()
    p = ctypes.addressof(buf)
    p = (p + 15) // 16 * 16
    assert l.ttts_gemm_nt_bf16(p, 12, p, 8, p, 8, None, None, 4, 4, 12, 0, None) == -1   # K % 8 != 0
    assert b"multiples of 8" in l.ttts_last_error()
    assert l.ttts_attn_causal_fwd_bf16(p, p, p, p, p, 1, 1, 16, 48, 16, 16, 16, 16, 1.0, 0.0, 0, None, None) == -1
    assert b"head_dim" in l.ttts_last_error()
    assert l.ttts_vq_nearest_f32(p, p, p, None, None, p, 8, 8, 7, None) == -1                # odd D
    assert l.ttts_layernorm_bwd_workspace_bytes(9248, 512) == 1156 * 3 * 512 * 4
    assert l.ttts_gemm_tn_workspace_bytes(512, 1536, 9248) == 8 * 512 * 1536 * 4   # 48 tiles -> 8 slabs
    assert l.ttts_gemm_tn_workspace_bytes(128, 128, 200) == 0                       # single split: no workspace
    assert l.ttts_cast_desc_tiles(257, 512) == 9 * 16
    # grouped weight-gradient GEMM: host-side descriptor preparation (validation + tile prefix sums), no launch
    assert l.ttts_tn_desc_tiles(512, 1536) == 48 and l.ttts_tn_desc_tiles(257, 130) == 3 * 2
    arr = (lib.TnDesc * 3)()
    for i, (mo, no, kr) in enumerate([(512, 1536, 9280), (2048, 512, 9280), (136, 1000, 640)]):
        arr[i].At, arr[i].Bt, arr[i].C = p, p, p
        arr[i].ldat, arr[i].ldbt, arr[i].ldc = mo, no, no
        arr[i].Mo, arr[i].No, arr[i].Kr = mo, no, kr
    total = ctypes.c_int32(0)
    assert l.ttts_tn_desc_prepare(arr, 3, ctypes.byref(total)) == 0
    assert [arr[i].tile_begin for i in range(3)] == [0, 48, 112] and total.value == 112 + 2 * 8
    arr[2].Kr = 650
    assert l.ttts_tn_desc_prepare(arr, 3, ctypes.byref(total)) == -1 and b"multiple of 64" in l.ttts_last_error()
    assert l.ttts_gemm_tn_grouped_bf16_accum_f32(None, 3, 128, None) == -1 and b"null descriptor" in l.ttts_last_error()
    tw = np.empty(2048, np.float32)
    assert l.ttts_stft_twiddle_host(tw.ctypes.data_as(ctypes.c_void_p), 2048) == 0
    np.testing.assert_allclose(tw[2 * 512:2 * 512 + 2], [0.0, -1.0], atol=1e-7)


def test_token_kernel_index_contract_matches_prepare_tokens():
    """Executable statement of ttts_gpt_prepare_tokens' per-thread rule (csrc/elementwise.hip: gpt_prepare_tokens_kernel),
    mirrored in Python and compared with model.prepare_tokens (the torch form of ttts/gpt/model.py:474-489,397-414).
    The kernel itself is compared bit-for-bit on the GPU (test_fused_token_plumbing_equals_prepare_tokens)."""
    from ttts_amd.gpt.engine import resolve_config
    from ttts_amd.gpt.model import prepare_tokens
    c = resolve_config({})
    comp = c["mel_length_compression"]
    st, sp, sm, pm = c["start_text_token"], c["stop_text_token"], c["start_mel_token"], c["stop_mel_token"]

    def mirror(text, tl, mel, wl, clip):
        B, Tt, Tm = text.shape[0], text.shape[1], mel.shape[1]
        if clip:
            Tt, Tm = min(Tt, max(tl)), min(Tm, max(wl) // comp)
        valid = [w // comp + 1 for w in wl]
        ti = torch.zeros(B, Tt + 2, dtype=torch.int64); tt = ti.clone()
        mi = torch.zeros(B, Tm + 2, dtype=torch.int64); mt = mi.clone()
        W = Tt + Tm + 4
        for t in range(B * W):                       # one "thread" per (sample, position of either pair)
            b, i = divmod(t, W)
            if i < Tt + 2:
                ti[b, i] = st if i == 0 else (text[b, i - 1] if i - 1 < Tt else sp)
                tt[b, i] = text[b, i] if i < Tt else sp
            else:
                i -= Tt + 2
                m1 = lambda j: mel[b, j] if (j < Tm and j < valid[b]) else pm   # noqa: E731
                mi[b, i] = sm if i == 0 else m1(i - 1)
                mt[b, i] = m1(i) if i <= Tm else pm
        return ti, tt, mi, mt
    g = torch.Generator().manual_seed(3)
    for B, Tt, Tm, tl, wl, clip in [(4, 40, 300, [40, 33, 12, 25], [300 * 1024 + 7, 257 * 1024, 100 * 1024 + 1023, 299 * 1024], True),
                                    (3, 50, 120, [20, 31, 7], [64 * 1024, 100 * 1024 + 5, 17 * 1024], True),
                                    (2, 16, 64, [16, 9], [64 * 1024, 30 * 1024], False), (1, 8, 8, [8], [8 * 1024], True)]:
        text = torch.randint(1, 255, (B, Tt), generator=g); mel = torch.randint(0, 1024, (B, Tm), generator=g)
        ref = prepare_tokens(c, text, torch.tensor(tl), mel, torch.tensor(wl), clip_inputs=clip)
        for r, m in zip(ref, mirror(text, tl, mel, wl, clip)):
            assert torch.equal(r.reshape(m.shape), m)


def test_grouped_dw_left_out_selection():
    """GptEngine._dw_plan's tile-count quantisation rule (pure host logic): which dW problems leave the grouped launch."""
    from ttts_amd.gpt.engine import left_out_problems
    layer = [64, 64, 48, 16]                       # c_fc, mlp c_proj, c_attn, attn c_proj at d 512
    out = left_out_problems(layer * 6, 512)        # 1152 tiles: 2.25 rounds -> two 64-tile problems out, 1024 stay
    assert sum((layer * 6)[j] for j in out) == 128 and len(out) == 2
    out = left_out_problems(layer * 3, 512)        # one backward section: 576 -> 512 + one 64-tile problem
    assert [(layer * 3)[j] for j in out] == [64]
    assert left_out_problems(layer * 2, 512) == set()          # 384 tiles: fewer than one round, all grouped
    assert left_out_problems([64] * 16, 512) == set()          # exact multiple
    assert left_out_problems([64] * 15, 512) == set()          # 960: remainder 448 >= 5/8 of a round: keep the third round
    tiles = [48, 16, 64, 64, 48, 16, 64, 64, 48]
    out = left_out_problems(tiles, 256)            # 432 = 256 + 176 (>= 160 = 5/8): stays
    assert out == set()
    out = left_out_problems(tiles, 384)            # 432 = 384 + 48: exactly one 48-tile problem leaves
    assert [tiles[j] for j in out] == [48]


def test_product_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ttts_amd.gpt import GptEngine, UnifiedVoice
    from ttts_amd.lib import TttsError
    with pytest.raises(TttsError):
        GptEngine({"layers": 1, "model_dim": 64, "heads": 2}, "cpu")
    with pytest.raises(TttsError):
        UnifiedVoice(layers=1, model_dim=64, heads=2, device="cpu")
    from ttts_amd import ops
    with pytest.raises(TttsError):
        ops.vq_nearest(torch.zeros(4, 8), torch.zeros(4, 8))


def test_product_never_imports_oracle_or_reference():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ttts_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
                assert "/root/reference" not in src, f
    bench = open(os.path.join(ROOT, "bench.py")).read() if os.path.exists(os.path.join(ROOT, "bench.py")) else ""
    assert "/root/reference" not in bench


def test_param_spec_matches_reference_surface(golden_dir):
    from ttts_amd.gpt.engine import param_spec, resolve_config
    surf = json.load(open(os.path.join(golden_dir, "surface.json")))
    spec = param_spec(resolve_config(surf["gpt_config"]))
    assert [[k, list(s)] for k, s in spec] == [[k, s] for k, s, _ in surf["gpt"]]


def test_prepare_tokens_matches_oracle(golden_dir):
    from oracle import gpt_ref
    from ttts_amd.gpt.engine import resolve_config
    from ttts_amd.gpt.model import prepare_tokens
    g = np.load(os.path.join(golden_dir, "gpt_tiny.npz"))
    cfg = json.loads(str(g["cfg_json"]))
    args = [torch.from_numpy(g[k]) for k in ("text", "text_lengths", "mel", "wav_lengths")]
    want = gpt_ref.prepare_tokens(*args, cfg)
    before = args[2].clone()
    got = prepare_tokens(resolve_config(cfg), *args)
    assert torch.equal(args[2], before)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    # ragged: clip by the batch maxima, empty padding region, maximum-length sample untouched
    gen = torch.Generator().manual_seed(1)
    text = torch.randint(1, 255, (3, 20), generator=gen); mel = torch.randint(0, 1024, (3, 50), generator=gen)
    tl = torch.tensor([7, 20, 1]); wl = torch.tensor([50 * 1024, 3 * 1024 + 1000, 0])
    for clip in (True, False):
        want = gpt_ref.prepare_tokens(text, tl, mel, wl, cfg, clip)
        got = prepare_tokens(resolve_config(cfg), text, tl, mel, wl, clip)
        for a, b in zip(got, want):
            assert torch.equal(a, b)


def test_stockham_fft_index_math_mirror():
    """numpy mirror of stft_mag_kernel's index arithmetic (packing, radix-2 Stockham passes, real-spectrum unpack)."""
    for n_fft in (1024, 2048, 64):
        rng = np.random.default_rng(n_fft)
        x = rng.standard_normal(n_fft)
        L = n_fft // 2
        tw = np.exp(-2j * np.pi * np.arange(L) / n_fft)
        src = x[0::2] + 1j * x[1::2]
        log2L = int(np.log2(L))
        for ps in range(log2L):
            Ns = 1 << ps
            dst = np.empty(L, complex)
            j = np.arange(L // 2)
            k = j & (Ns - 1)
            a, b = src[j], src[j + L // 2] * tw[k * (n_fft >> (ps + 1))]
            j0 = (j << 1) - k
            dst[j0], dst[j0 + Ns] = a + b, a - b
            src = dst
        X = np.empty(L + 1, complex)
        X[0], X[L] = src[0].real + src[0].imag, src[0].real - src[0].imag
        kk = np.arange(1, L)
        zk, zc = src[kk], src[L - kk]
        E = 0.5 * (zk + np.conj(zc))
        O = -0.5j * (zk - np.conj(zc))
        X[kk] = E + tw[kk] * O
        np.testing.assert_allclose(X, np.fft.rfft(x), rtol=1e-10, atol=1e-10)


def test_stft_backward_index_math_mirror():
    """numpy mirror of stft_mag_bwd_kernel: adjoint of the one-sided real DFT as an N-point Stockham FFT of conj(H)
    with twiddles taken from the 2N table."""
    for n_fft in (64, 1024):
        rng = np.random.default_rng(n_fft + 1)
        L = n_fft // 2
        g = rng.standard_normal(L + 1) + 1j * rng.standard_normal(L + 1)       # G_k = dRe + i dIm
        n = np.arange(n_fft)
        k = np.arange(L + 1)
        th = 2 * np.pi * np.outer(k, n) / n_fft
        want = (g.real[:, None] * np.cos(th) - g.imag[:, None] * np.sin(th)).sum(0)
        tw2 = np.exp(-2j * np.pi * np.arange(n_fft) / (2 * n_fft))
        src = np.zeros(n_fft, complex)
        src[:L + 1] = np.conj(g)
        log2L = int(np.log2(L))
        for ps in range(log2L + 1):
            Ns = 1 << ps
            dst = np.empty(n_fft, complex)
            j = np.arange(L)
            kk = j & (Ns - 1)
            a, b = src[j], src[j + L] * tw2[kk * (n_fft >> ps)]
            j0 = (j << 1) - kk
            dst[j0], dst[j0 + Ns] = a + b, a - b
            src = dst
        np.testing.assert_allclose(src.real, want, rtol=1e-9, atol=1e-9)


def test_mel_basis_matches_oracle():
    from oracle import mel_ref
    from ttts_amd.utils.data_utils import slaney_mel_basis
    for args in ((32000, 2048, 128, 0, None), (22050, 1024, 80, 0, 8000)):
        assert np.array_equal(slaney_mel_basis(*args), mel_ref.slaney_mel_basis(*args))


def test_dropout_threshold_and_hash_reference():
    """The dropout keep rule documented in DESIGN.md: 16 random bits per element from hash32(e >> 1), hash32 = two rounds of
    24-bit multiply + xor-shift (csrc/common.hpp).  numpy emulation of the device function: keep rate of both halves, independence
    of the two halves and of neighbouring elements, fresh masks for the next stream-counter value."""
    M = np.uint64(0xFFFFFFFF)

    def hash32(x, lo, hi):
        x = ((x ^ np.uint64(lo)) + np.uint64(hi)) & M
        x ^= x >> np.uint64(16)
        x = ((x & np.uint64(0xFFFFFF)) * np.uint64(0x9E3779)) & M
        x ^= x >> np.uint64(15)
        x = ((x & np.uint64(0xFFFFFF)) * np.uint64(0x85EBCB)) & M
        x ^= x >> np.uint64(16)
        return x
    thr = int(0.1 * 65536 + 0.5)
    n = 1 << 20
    r = hash32(np.arange(n, dtype=np.uint64), 12345, 678)
    lo16, hi16 = (r & np.uint64(0xFFFF)).astype(np.int64), (r >> np.uint64(16)).astype(np.int64)
    ka, kb = lo16 >= thr, hi16 >= thr
    sig = (0.09 / n) ** 0.5
    assert abs(ka.mean() - 0.9) < 5 * sig and abs(kb.mean() - 0.9) < 5 * sig
    corr = lambda u, v: float(np.corrcoef(u.astype(float), v.astype(float))[0, 1])   # noqa: E731
    assert abs(corr(ka, kb)) < 5e-3 and abs(corr(ka[:-1], ka[1:])) < 5e-3 and abs(corr(ka[:-578], ka[578:])) < 5e-3
    r2 = hash32(np.arange(n, dtype=np.uint64), 12345, (678 + 0x9E3779B1) & 0xFFFFFFFF)      # seed_mix: next counter value
    k2 = (r2 & np.uint64(0xFFFF)).astype(np.int64) >= thr
    assert abs((ka == k2).mean() - 0.82) < 5e-3
    for half in (lo16, hi16):                                                              # 256-bin uniformity of the 16 bits
        c = np.bincount(half >> 8, minlength=256)
        assert ((c - n / 256) ** 2 / (n / 256)).sum() < 400


# ---- data-parallel exchange under gloo, world_size 2 --------------------------------------------------------------
WORKER = r"""
import os, sys, torch
sys.path.insert(0, %r)
from ttts_amd.parallel import FlatDataParallel, init_distributed, shard_indices
rank, world, _ = init_distributed("gloo")
dp = FlatDataParallel()
assert dp.enabled and dp.world == 2 and dp.rank == rank
params = torch.full((1000,), float(rank + 1))
dp.broadcast_(params)
assert torch.all(params == 1.0)
# each rank: gradient of its own micro-batch, weights pre-scaled by loss_scale(): SUM all-reduce == mean gradient
g = torch.Generator().manual_seed(100 + rank)
local = torch.randn(1000, generator=g)
grads = local * dp.loss_scale()
dp.allreduce_grads_(grads)
want = sum(torch.randn(1000, generator=torch.Generator().manual_seed(100 + r)) for r in range(2)) / 2
assert torch.allclose(grads, want, atol=1e-6)
# ranged asynchronous exchange (the overlapped schedule of GptEngine.train_step): same sums, range by range
g2 = local * dp.loss_scale()
handles = [dp.allreduce_range_(g2, lo, hi) for lo, hi in ((400, 1000), (0, 400), (5, 5))]
assert handles[2] is None
for h in handles[:2]:
    h.wait()
assert torch.equal(g2, grads)
assert abs(dp.max_over_ranks(float(rank)) - 1.0) < 1e-12
assert shard_indices(7, rank, world) == list(range(7))[rank::2]
dp.barrier()
sys.stdout.write("rank" + str(rank) + "-ok\n")
"""


def test_flat_data_parallel_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29000 + os.getpid() % 1000))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"], str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "rank0-ok" in r.stdout and "rank1-ok" in r.stdout


def test_precision_environment_typos_are_refused():
    """TTTS_CONV_PRECISION / TTTS_DIFFUSION_PRECISION are validated at import: 'tf32' (for 'tf32class') used to select the default
    kernels silently while conv_precision() reported the bogus string."""
    for var, mod in (("TTTS_CONV_PRECISION", "ttts_amd.ops"), ("TTTS_DIFFUSION_PRECISION", "ttts_amd.diffusion.aa_model")):
        r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import %s" % (ROOT, mod)],
                           capture_output=True, text=True, env=dict(os.environ, **{var: "tf32"}), timeout=120)
        assert r.returncode != 0 and var in r.stderr, r.stderr[-500:]
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from ttts_amd import ops; "
                        "assert ops.conv_precision() == 'tf32class'" % ROOT],
                       capture_output=True, text=True, env=dict(os.environ, TTTS_CONV_PRECISION="tf32class"), timeout=120)
    assert r.returncode == 0, r.stderr[-500:]


def test_forced_world1_group_runs_the_collectives(tmp_path):
    """TTTS_DP_FORCE=1 at world size 1 (gloo here; `nccl` in the -m gpu tests): the group is created with no launcher
    environment, FlatDataParallel is enabled and its collectives are the identity."""
    script = tmp_path / "w1.py"
    script.write_text("import sys, torch\nsys.path.insert(0, %r)\n" % ROOT + """
import torch.distributed as dist
from ttts_amd.parallel import FlatDataParallel, init_distributed
rank, world, _ = init_distributed("gloo")
assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_world_size() == 1
dp = FlatDataParallel()
assert dp.enabled and dp.world == 1 and dp.loss_scale() == 1.0
g = torch.arange(100, dtype=torch.float32); want = g.clone()
dp.allreduce_grads_(g)
h = dp.allreduce_range_(g, 10, 60); h.wait()
dp.broadcast_(g)
assert torch.equal(g, want) and dp.all_ranks_ok(True) and not dp.all_ranks_ok(False) and dp.max_over_ranks(2.5) == 2.5
dp.barrier()
print("w1-ok")
""")
    env = dict(os.environ, TTTS_DP_FORCE="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and "w1-ok" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]
    env.pop("TTTS_DP_FORCE")
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\nfrom ttts_amd.parallel import FlatDataParallel, "
                        "init_distributed\ninit_distributed('gloo')\nimport torch.distributed as d\n"
                        "assert not d.is_initialized() and not FlatDataParallel().enabled" % ROOT],
                       capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]


WORKER_N = r"""
import os, sys, json
sys.path.insert(0, %r)
import torch
import torch.distributed as dist
from ttts_amd.parallel import FlatDataParallel, init_distributed
from ttts_amd.gpt.engine import arena_layout, exchange_ranges, param_spec, resolve_config
rank, world, _ = init_distributed("gloo")
assert dist.get_world_size() == world and world == int(os.environ["WANT_WORLD"])
cfg = resolve_config(json.load(open(os.path.join(%r, "ttts_amd", "gpt", "config.json")))["gpt"])
offsets, n = arena_layout(param_spec(cfg))
n_small = 20000                                   # the exchange protocol on a prefix-scaled arena (same four-range structure)
split, first, second = exchange_ranges(offsets, n, cfg["layers"])
scale = lambda r: tuple(int(v * n_small // n) for v in r)
first_s, second_s = [scale(r) for r in first], [scale(r) for r in second]
for bf in (False, True):
    dp = FlatDataParallel(grad_dtype=torch.bfloat16 if bf else None)
    assert dp.enabled and dp.world == world
    params = torch.randn(n_small, generator=torch.Generator().manual_seed(7 + rank))   # ranks start DIFFERENT ...
    dp.broadcast_(params)                                                            # ... rank 0's state wins
    seeds = [1234 + r for r in range(world)]                                         # bench.py / Trainer: per-rank data + dropout seeds
    assert len(set(seeds)) == world
    for step in range(3):
        g = torch.Generator().manual_seed(seeds[rank] * 100 + step)
        grads = (torch.randn(n_small, generator=g) + params * 0.01) * dp.loss_scale()   # a "local gradient" that depends on rank data
        local = grads.clone()
        # the engine's schedule: ranges final after backward part 0 go out first, the rest after part 1; all are waited for before the optimizer
        pending = [dp.allreduce_range_(grads, lo, hi) for lo, hi in first_s]
        pending += [dp.allreduce_range_(grads, lo, hi) for lo, hi in second_s]
        for h in pending:
            if h is not None:
                h.wait()
        whole = local.clone()
        FlatDataParallel().allreduce_grads_(whole)                                   # fp32 single all-reduce of the same data
        if bf:
            # every addend is rounded to bf16 (2^-8 relative) and so is every partial sum of the reduction: |error| <= world * 2^-8 * sum_r |g_r| (bf16 unit roundoff 2^-8)
            absum = local.abs()
            FlatDataParallel().allreduce_grads_(absum)
            assert bool(((grads - whole).abs() <= world * 2.0 ** -8 * absum + 1e-7).all()), "bf16 exchange outside its bound"
        else:
            # same addends, but a collective sums a chunk in an order that depends on where the chunk sits in its message: bitwise
            # equal to the whole-arena all-reduce only at world size 2 (one addition); what must hold is fp32-summation noise
            assert float((grads - whole).abs().max()) <= 4e-7 * float(whole.abs().max()) * world, "ranged exchange differs from the whole-arena all-reduce"
        params -= 0.1 * grads
        gathered = [torch.empty_like(params) for _ in range(world)]
        dist.all_gather(gathered, params)
        assert all(torch.equal(gathered[0], t) for t in gathered), "replicas diverged (bf16=%%s, step %%d)" %% (bf, step)
        gl = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gl, local)
        assert not torch.equal(gl[0], gl[1]), "ranks drew the same batch"
assert FlatDataParallel().all_ranks_ok(True) and not FlatDataParallel().all_ranks_ok(rank != world - 1)
dist.barrier()
sys.stdout.write("rank" + str(rank) + "-ok\n")
"""


def test_exchange_ranges_tile_the_arena_exactly_once():
    """GptEngine.grad_exchange_plan (pure host arithmetic, ttts_amd.gpt.engine.exchange_ranges): the four ranges of the overlapped
    gradient exchange cover every element of the flat arena exactly once, for the shipped model and for other depths."""
    from ttts_amd.gpt.engine import arena_layout, exchange_ranges, param_spec, resolve_config
    base = json.load(open(os.path.join(ROOT, "ttts_amd", "gpt", "config.json")))["gpt"]
    for layers in (base["layers"], 2, 7, 30):
        cfg = resolve_config(dict(base, layers=layers))
        offsets, n = arena_layout(param_spec(cfg))
        for split in (None, 1, layers - 1):
            sp, first, second = exchange_ranges(offsets, n, layers, split)
            cover = np.zeros(n, np.int8)
            for lo, hi in first + second:
                assert 0 <= lo < hi <= n
                cover[lo:hi] += 1
            assert (cover == 1).all(), (layers, split)
            # what part 0 releases is exactly layers sp .. L-1, ln_f, final_norm and the heads
            early = np.zeros(n, bool)
            for lo, hi in first:
                early[lo:hi] = True
            for k, o in offsets.items():
                is_early = (k.startswith("gpt.h.") and int(k.split(".")[2]) >= sp) or k.startswith(("gpt.ln_f", "final_norm", "text_head", "mel_head"))
                assert bool(early[o]) == is_early, (k, layers, split)


@pytest.mark.parametrize("world", [4, 8])
def test_ranged_exchange_keeps_replicas_identical_gloo(tmp_path, world):
    """The data-parallel protocol of the GPT step at world sizes 4 and 8 (gloo, CPU): rank-0 broadcast, per-rank seeds, the four
    ranged asynchronous all-reduces == one whole-arena all-reduce up to the summation order, replicas bit-identical after 3 steps; the optional
    bf16 gradient exchange keeps replicas bit-identical too and stays within its rounding bound of the fp32 sum."""
    script = tmp_path / "worker_n.py"
    script.write_text(WORKER_N % (ROOT, ROOT))
    port = str(29000 + (os.getpid() * 7 + world) % 1000)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, WANT_WORLD=str(world), OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert all(("rank%d-ok" % k) in r.stdout for k in range(world))


def test_gpt_collater_and_vq_file_format(tmp_path):
    """ttts/gpt/dataset.py:65-97 semantics: None items filtered, zero right-padding, length tensors; `.vq.pth` holds a plain
    list of ints (ttts/prepare/extract_vq.py:22) that the dataset reads back."""
    from ttts_amd.gpt.dataset import GptTtsCollater, GptTtsDataset
    from ttts_amd.prepare.extract_vq import save_vq
    paths = []
    for i, n in enumerate((5, 9)):
        p = str(tmp_path / ("utt%d.wav" % i))
        out = save_vq(p, torch.arange(n) * 3 + i)
        assert out == p + ".vq.pth" and torch.load(out) == [int(v) for v in (torch.arange(n) * 3 + i)]
        paths.append(p)
    jl = tmp_path / "data.jsonl"
    jl.write_text("\n".join(json.dumps({"path": p, "text_ids": list(range(1, 4 + i)), "wav_length": 24000 * (i + 1)})
                            for i, p in enumerate(paths)) + "\n" + json.dumps({"path": "missing", "text_ids": [1], "wav_length": 1}) + "\n")
    ds = GptTtsDataset(str(jl))
    assert len(ds) == 3 and ds[2] is None
    batch = GptTtsCollater()([ds[0], ds[1], ds[2]])
    assert batch["padded_text"].tolist() == [[1, 2, 3, 0], [1, 2, 3, 4]]
    assert batch["text_lengths"].tolist() == [3, 4] and batch["qmel_lengths"].tolist() == [5, 9]
    assert batch["padded_qmel"].shape == (2, 9) and batch["padded_qmel"][0, 5:].eq(0).all()
    assert batch["wav_lens"].tolist() == [24000, 48000]
    assert GptTtsCollater()([None, None]) is None


def test_bucket_sampler_matches_reference_batches(golden_dir):
    """DistributedBucketSampler: every batch of every rank, two epochs, five configurations, against the batches the
    reference's own class produced (tools/make_goldens.py gen_sampler)."""
    from ttts_amd.vqvae.dataset import DistributedBucketSampler, VQVAECollater
    fx = json.load(open(os.path.join(golden_dir, "sampler.json")))

    class DS:
        lengths = fx["lengths"]

        def __len__(self):
            return len(self.lengths)
    for case in fx["cases"]:
        seen = []
        for rank, want in enumerate(case["ranks"]):
            smp = DistributedBucketSampler(DS(), case["batch_size"], list(case["boundaries"]), num_replicas=case["world"],
                                           rank=rank, shuffle=case["shuffle"])
            assert len(smp) == want["len"] and list(smp.boundaries) == want["boundaries_after"]
            assert list(smp.num_samples_per_bucket) == want["num_samples_per_bucket"]
            for epoch in ("1", "2"):
                smp.set_epoch(int(epoch))
                got = [list(b) for b in smp]
                assert got == want["batches"][epoch], (case["world"], rank, epoch)
            seen.append(got)
        # data-parallel property: same number of batches on every rank, k-th batches come from the same bucket
        assert len({len(s) for s in seen}) == 1
    # collater: zero padding, rows by decreasing wav length, None items dropped
    items = [(torch.ones(1, 5), torch.tensor([1, 2])), None, (torch.ones(1, 9) * 2, torch.tensor([3])), (torch.ones(1, 7) * 3, torch.tensor([4, 5, 6]))]
    out = VQVAECollater()(items)
    assert out["wav"].shape == (3, 9) and out["wav_lengths"].tolist() == [9, 7, 5] and out["text_lengths"].tolist() == [1, 3, 2]
    assert out["wav"][1, 7:].abs().sum() == 0 and out["text"][0].tolist() == [3, 0, 0]


def test_reference_import_names_and_signatures():
    """The drop-in boundary: the reference's module paths and entry-point signatures resolve to this build."""
    import inspect
    import ttts.gpt.model, ttts.gpt.train, ttts.vqvae.train, ttts.vqvae.vq2, ttts.vqvae.modules, ttts.vqvae.attentions  # noqa: E401
    import ttts.vqvae.losses, ttts.vqvae.quantize, ttts.vqvae.core_vq, ttts.vqvae.dataset, ttts.utils.data_utils  # noqa: E401
    import ttts.utils.commons, ttts.utils.vc_utils, ttts.utils.utils, ttts.diffusion.aa_model, ttts.prepare.extract_vq  # noqa: E401
    import ttts_amd
    assert ttts.gpt.model.UnifiedVoice is ttts_amd.gpt.UnifiedVoice
    assert list(inspect.signature(ttts.vqvae.train.train_and_evaluate).parameters) == [
        "rank", "epoch", "hps", "nets", "optims", "schedulers", "scaler", "loaders", "logger", "writers", "aug"]   # ttts/vqvae/train.py:298-300
    assert list(inspect.signature(ttts.vqvae.train.run).parameters)[:3] == ["rank", "n_gpus", "hps"]                 # :119
    assert callable(ttts.vqvae.train.main) and callable(ttts.gpt.train.Trainer.train)
    p = inspect.signature(ttts.gpt.train.Trainer.__init__).parameters
    assert list(p)[:2] == ["self", "cfg_path"]
    fwd = list(inspect.signature(ttts.gpt.model.UnifiedVoice.forward).parameters)
    assert fwd == ["self", "text_inputs", "text_lengths", "mel_codes", "wav_lengths", "types", "text_first", "raw_mels",
                   "return_attentions", "return_latent", "clip_inputs"]
    for name in ("SynthesizerTrn", "MultiPeriodDiscriminator", "Generator", "PosteriorAudioEncoder", "TextEncoder", "MRTE",
                 "ResidualCouplingBlock", "DiscriminatorP", "DiscriminatorS"):
        assert hasattr(ttts.vqvae.vq2, name), name
    for name in ("spectrogram_torch", "spec_to_mel_torch", "mel_spectrogram_torch", "HParams"):
        assert hasattr(ttts.utils.data_utils, name), name
    # `python -m ttts.gpt.train` / `python -m ttts.vqvae.train` exist as runnable modules
    import importlib.util
    for mod in ("ttts.gpt.train", "ttts.vqvae.train"):
        assert importlib.util.find_spec(mod) is not None


def test_attention_dropout_product_scheme_statistics():
    """numpy emulation of the attention keep-mask (csrc/attn_common.hpp: row hash R with bit 23 set, odd 24-bit column multiplier M with the top bit
    set, keep iff (R[23:0] * M + R) mod 2^32 >= thr << 16): keep rate, 256-bin chi-square of the compared word, lag-1..8 row /
    column correlations, the 2 x 2 interaction and per-row / per-column drop rates all at the level of independent draws."""
    def hash32(x, lo, hi):
        x = (np.uint32(x) ^ np.uint32(lo)) + np.uint32(hi)
        x ^= x >> np.uint32(16)
        x = ((x & np.uint32(0xFFFFFF)).astype(np.uint64) * np.uint64(0x9E3779) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        x ^= x >> np.uint32(15)
        x = ((x & np.uint32(0xFFFFFF)).astype(np.uint64) * np.uint64(0x85EBCB) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        x ^= x >> np.uint32(16)
        return x

    def corr(a, b):
        a = a - a.mean(); b = b - b.mean()
        return float((a * b).mean() / np.sqrt((a * a).mean() * (b * b).mean()))
    S, thr = 1156, 6554
    ids = np.arange(S, dtype=np.uint32) + np.uint32(3 * S)              # (b*H + h)*S + position for some (b, h)
    with np.errstate(over="ignore"):
        for seed_lo, seed_hi in ((0x1234567, 0x89ABCDE), (7, 0xC0FFEE), (0xDEADBEEF, 1)):
            R = (hash32(ids, seed_lo, seed_hi) | np.uint32(0x800000)).astype(np.uint64)   # drop_row_hash: bit 23 forced
            M = ((hash32(ids, seed_lo ^ 0x5BD1E995, seed_hi) & np.uint32(0xFFFFFF)) | np.uint32(0x800001)).astype(np.uint64)
            word = ((R[:, None] & 0xFFFFFF) * M[None, :] + R[:, None]) & 0xFFFFFFFF
            keep = word >= (thr << 16)
            assert abs(keep.mean() - 0.9) < 2e-3
            hist = np.bincount((word >> 24).ravel().astype(np.int64), minlength=256)
            chi = float(((hist - word.size / 256) ** 2 / (word.size / 256)).sum())
            assert chi < 400, chi                                        # 255 degrees of freedom
            d = (~keep).astype(np.float64)
            for lag in range(1, 9):
                assert abs(corr(d[:, :-lag].ravel(), d[:, lag:].ravel())) < 5e-3
                assert abs(corr(d[:-lag].ravel(), d[lag:].ravel())) < 5e-3
            x = (keep[:-1, :-1] ^ keep[:-1, 1:]).astype(np.float64); y = (keep[1:, :-1] ^ keep[1:, 1:]).astype(np.float64)
            assert abs(corr(x.ravel(), y.ravel())) < 5e-3
            binom = np.sqrt(0.09 / S)
            assert 0.8 * binom < d.mean(1).std() < 1.25 * binom and 0.8 * binom < d.mean(0).std() < 1.25 * binom


def test_phase_merged_strided_convolution_identities():
    """The index algebra behind ConvMfmaParams::rowS (csrc/conv_mfma.hip: merged_phase_weight / merged_fwd_weight), restated with
    torch on the CPU: a strided convolution's data gradient equals ONE stride-1 convolution whose rows are (channel, phase) pairs,
    and its forward equals a stride-1 convolution over the phase-de-interleaved input -- for the step's k16 stride-10 / stride-8
    layers, a row length that is not a multiple of the stride, K = stride and K < 2 stride."""
    import torch
    import torch.nn.functional as F
    torch.manual_seed(0)
    for cin, cout, k, s, pad, L in [(4, 6, 16, 10, 7, 300), (3, 5, 16, 8, 4, 203), (5, 4, 5, 3, 2, 101), (2, 3, 4, 4, 0, 43), (3, 2, 7, 2, 3, 57)]:
        x = torch.randn(2, cin, L, dtype=torch.float64); w = torch.randn(cout, cin, k, dtype=torch.float64)
        y = F.conv1d(x, w, stride=s, padding=pad)
        dy = torch.randn_like(y)
        # ---- data gradient: rows m = (ci, r), dx[ci][s j + r] = sum_{co, e} A[m][co][e] dy[co][j + e - vpad]
        emax = (s - 1 + pad) // s
        emin = min((r + pad) // s - (k - 1 - (r + pad) % s) // s for r in range(s))
        kv, vpad, T = emax - emin + 1, -emin, -(-L // s)
        A = torch.zeros(cin * s, cout, kv, dtype=torch.float64)
        for m in range(cin * s):
            ci, r = divmod(m, s)
            for e in range(kv):
                t = (r + pad) // s - (e - vpad)
                kk = (r + pad) % s + s * t
                if t >= 0 and kk < k:
                    A[m, :, e] = w[:, ci, kk]
        out = F.conv1d(F.pad(dy, (vpad, T + kv)), A)[:, :, :T]
        dx = out.view(2, cin, s, T).permute(0, 1, 3, 2).reshape(2, cin, T * s)[:, :, :L]
        ref = torch.nn.grad.conv1d_input(x.shape, w, dy, stride=s, padding=pad)
        assert (dx - ref).abs().max() < 1e-10
        # ---- forward: x'[(ci, r)][j] = x[ci][s j + r], y[co][l] = sum W[co][(ci, r)][t'] x'[(ci, r)][l + t' - vp]
        tmin = -(-pad // s) * -1 if pad > 0 else 0
        tmax = (k - 1 - pad) // s
        kf, vp, Lv = tmax - tmin + 1, -tmin, -(-L // s)
        xp = F.pad(x, (0, Lv * s - L)).view(2, cin, Lv, s).permute(0, 1, 3, 2).reshape(2, cin * s, Lv)
        W = torch.zeros(cout, cin * s, kf, dtype=torch.float64)
        for n in range(cin * s):
            ci, r = divmod(n, s)
            for t in range(kf):
                kk = s * (t - vp) + r + pad
                if 0 <= kk < k:
                    W[:, n, t] = w[:, ci, kk]
        yv = F.conv1d(F.pad(xp, (vp, y.shape[2] + kf)), W)[:, :, :y.shape[2]]
        assert (yv - y).abs().max() < 1e-10


def test_bench_respawn_command_line(monkeypatch):
    """bench.py --gpus N without a torchrun environment re-executes itself as `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` (the driver's verb is
    `python bench.py --gpus N`); inside a torchrun environment (RANK set) it does not."""
    import importlib
    import subprocess
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    monkeypatch.delenv("RANK", raising=False)
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-7].endswith("bench.py") and cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_fp8_entry_points_validate_without_gpu_and_oracle_properties(lib):
    """ttts_fp8_* argument validation runs before anything touches the device; the fp8 oracle's quantiser is the OCP e4m3 cast
    (idempotent, saturating at 448, round-to-nearest-even, zero tensors), and its 1 x 1 convolution sits within e4m3's error of fp32."""
    from oracle import fp8_ref as F8
    l = lib.get()
    buf = (ctypes.c_char * 4096)()
    p = (ctypes.addressof(buf) + 15) // 16 * 16
    assert l.ttts_fp8_gemm_nt(p, p, p, None, None, p, p, 64, 64, 96, 1, 1, 96, 96, 0, 0, 0, 0, 0, 64, 1, 0, None, None) == -1      # K % 64 != 0
    assert b"multiple of 64" in l.ttts_last_error()
    assert l.ttts_fp8_gemm_nt(None, p, p, None, None, p, p, 64, 64, 64, 1, 1, 64, 64, 0, 0, 0, 0, 0, 64, 1, 0, None, None) == -1
    assert b"null pointer" in l.ttts_last_error()
    assert l.ttts_fp8_gemm_nt(p, p, p, None, None, p, p, 64, 64, 64, 1, 1, 60, 64, 0, 0, 0, 0, 0, 64, 1, 0, None, None) == -1    # pitch not 16-aligned
    assert l.ttts_fp8_quant_f32(p, p, p, 4, 30, 30, None) == -1 and b"multiple of 4" in l.ttts_last_error()
    assert l.ttts_fp8_quant_transpose_f32(p, p, p, 1, 30, 8, 24, None) == -1
    assert l.ttts_fp8_amax_f32(None, 8, p, 0, None) == -1
    assert l.ttts_fp8_quant_both_f32(p, p, p, p, 1, 30, 8, 60, 64, None) == -1 and b"multiples of 64" in l.ttts_last_error()
    # weight-gradient shapes split their inner groups (slabs + ordered reduce); output-rich shapes never do
    assert l.ttts_fp8_gemm_nt_workspace_bytes(512, 512, 1, 16) == 16 * 512 * 512 * 4 and l.ttts_fp8_gemm_nt_workspace_bytes(1536, 432, 16, 1) == 0
    x = torch.tensor([0.0, 1.0, -448.0, 17.0, 18.0, 19.0, 1e-9, 300.0])
    q, a = F8.quant(x)
    assert float(a) == 448.0 and torch.equal(q, F8.quant(q)[0])                      # idempotent on representable values
    assert q.tolist()[:3] == [0.0, 1.0, -448.0] and q[3].item() == 16.0 and q[4].item() == 18.0 and q[5].item() == 20.0   # ties to even (step 2 at 16..32)
    assert F8.quant(torch.zeros(5))[0].abs().sum() == 0 and F8.alpha(torch.tensor(0.0), torch.tensor(0.0)) == 1.0
    g = torch.Generator().manual_seed(0)
    xx, w = torch.randn(2, 256, 40, generator=g), torch.randn(96, 256, generator=g) / 16
    ref = torch.einsum("oc,bct->bot", w, xx)
    assert float((F8.conv1x1_fwd(xx, w) - ref).norm() / ref.norm()) < 6e-2
    assert float((F8.conv1x1_dgrad(ref, w) - torch.einsum("oc,bot->bct", w, ref)).norm() / torch.einsum("oc,bot->bct", w, ref).norm()) < 6e-2
    assert float((F8.conv1x1_wgrad(ref, xx) - torch.einsum("bot,bct->oc", ref, xx)).norm() / torch.einsum("bot,bct->oc", ref, xx).norm()) < 6e-2
