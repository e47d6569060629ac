"""Register budgets of the GPT step's hot kernels, read from the compiler's code-object metadata (tools/kernel_resources.py: hipcc
cross-compiles for gfx950 without a GPU).  The kernels are written for a fixed number of co-resident workgroups per CU; a source
change that pushes one over its budget (or into spills) is a performance regression no numerics test notices."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)

import kernel_resources as kr  # noqa: E402


def _table(src):
    res = kr.resources(src)
    dm = kr.demangle([n for n, _ in res])
    return {dm[n].split("(")[0].replace("void ", "").replace("ttts::", ""): r for n, r in res}


def test_nt_gemm_kernels_fit_their_occupancy():
    t = _table("gemm.hip")
    assert all(r["spill"] == 0 and r["scratch"] == 0 for r in t.values()), {k: r for k, r in t.items() if r["spill"]}
    for k, r in t.items():
        if k.startswith("gemm_nt_glds_kernel<"):
            epi, bkt, nj, nwm = [int(x) for x in k[k.index("<") + 1:k.index(">")].split(",")[:4]]
            if nwm == 4:          # eight waves, two workgroups per CU = four waves per SIMD
                assert r["wg"] == 512 and r["vgpr"] <= 128 and 2 * r["lds"] <= 160 * 1024, (k, r)
            elif bkt == 32:       # three workgroups of four waves per CU
                assert r["wg"] == 256 and r["vgpr"] <= 168 and 3 * r["lds"] <= 160 * 1024, (k, r)
            else:                 # two workgroups per CU
                assert r["wg"] == 256 and r["vgpr"] <= 256 and 2 * r["lds"] <= 160 * 1024, (k, r)
        if k.startswith("gemm_nt_wreg_kernel<"):     # one eight-wave workgroup per CU: 128 weight registers + accumulators + fragments
            assert r["wg"] == 512 and r["vgpr"] <= 256 and r["lds"] == 0, (k, r)   # (its 160 KB of LDS are dynamic)
        if k.startswith("gemm_nt_tall_kernel<"):     # one workgroup per CU; the 144 KB ring is dynamic LDS
            assert r["vgpr"] <= 256 and r["lds"] == 0, (k, r)
        if k.startswith("gemm_tn_grouped_kernel") or k.startswith("gemm_tn_glds_kernel"):
            assert r["vgpr"] <= 256 and 2 * r["lds"] <= 160 * 1024, (k, r)


def test_dh64_attention_kernels_fit_their_occupancy():
    t = _table("attn_dh64.hip")
    assert all(r["spill"] == 0 and r["scratch"] == 0 for r in t.values()), {k: r for k, r in t.items() if r["spill"]}
    for k, r in t.items():
        if "attn_fwd_kernel" in k:        # three workgroups per CU
            assert r["vgpr"] <= 168, (k, r)
        else:                             # dQ, dK/dV: two per CU
            assert r["vgpr"] <= 256, (k, r)


def test_split_bf16_conv_kernels_keep_two_waves_per_simd():
    """The split-bf16 forward / data-gradient kernels run two workgroups per CU (the DMA kernel's 80 KB stages are sized for it):
    the pipelined stage loop (second fragment set + per-tap addresses) must stay within 256 registers per lane -- which is why the
    64 x 64 wave tile with 11 taps keeps the plain loop (b3_pipe)."""
    t = _table("conv_mfma.hip")
    seen = 0
    for k, r in t.items():
        if k.startswith("conv1d_bf16x3_dma_kernel<") or k.startswith("conv1d_bf16x3_kernel<"):
            seen += 1
            assert r["spill"] == 0 and r["scratch"] == 0, (k, r)
            assert r["wg"] == 256 and r["vgpr"] <= 256, (k, r)      # (the DMA kernel asks for two waves per SIMD: amdgpu_waves_per_eu)
    assert seen >= 20


def test_run_branches_is_a_plain_loop_without_a_gpu():
    """vqvae.modules.run_branches / join_side_streams on a CPU device: results in order, no streams involved."""
    import torch
    from ttts_amd.vqvae import modules as M
    dev = torch.device("cpu")
    order = []
    outs = M.run_branches([lambda i=i: order.append(i) or i * i for i in range(4)], dev)
    assert outs == [0, 1, 4, 9] and order == [0, 1, 2, 3]
    assert M.side_streams("mrf", dev) == [] and M.wgrad_side_stream(dev) is None
    M.join_side_streams(dev)
