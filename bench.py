#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: GPT train audio-tokens/sec on MI355X (configs[1]: VALL-E GPT train step,
batch 8 per GPU, 1024 audio tokens + 128 text tokens, bf16, full ttts/gpt/config.json model).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ..., or
     plainly as above: with no RANK in the environment the script re-executes itself under torch.distributed.run with
     one rank per GPU on 127.0.0.1 and a free port, and rank 0's JSON line is the only thing on stdout)

A "step" is one pass of the hot path over one synthetic batch per rank: token plumbing, forward, backward, gradient
all-reduce (N > 1, RCCL), grad-norm + clip, AdamW, LR schedule -- everything ttts/gpt/train.py:96-121 does per
iteration, in the reference's training mode (GPT-2 dropouts 0.1 active).  Inputs are resident in HBM before the timed
region.  Prints ONE JSON line with `roofline` (dominant kernel, HIP-event timed) and `cpu_baseline` (the oracle's
fp32 train step on the host cores, rank 0, N = 1 only).
"""
import argparse
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA peak (/opt/skills/guides/MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0      # HBM3E peak (same guide; ~6300 GB/s is what a float4 copy reaches)
B_PER_GPU, TEXT_LEN, MEL_LEN = 8, 128, 1024


def _event_time_us(fn, reps=20):
    """Average DEVICE time of `fn` (one or more launches on torch's current stream): `reps` calls are captured into one
    hipGraph and the replay is bracketed by HIP events, so that host-side launch / allocation time (which exceeds the device
    time of the 10-50 us kernels when they are launched one by one from Python) does not enter."""
    fn(); fn(); torch.cuda.synchronize()
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        run = g.replay
        run(); torch.cuda.synchronize()
    except Exception:                      # noqa: BLE001 -- capture refused: time the eager loop
        torch.cuda.synchronize()

        def run():
            for _ in range(reps):
                fn()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps * 1e3
        best = t if best is None else min(best, t)
    return best


def hbm_kernel_table(dev):
    """The HBM-bound kernels north_star names, each at its BASELINE shape: algorithmic bytes / measured
    device time -> GB/s and fraction of the 8 TB/s peak.  Measured live with HIP events on the launch stream."""
    from ttts_amd import ops
    from ttts_amd.utils.data_utils import spec_to_mel_torch, spectrogram_torch
    rows = {}

    def add(name, nbytes, fn, reps=20):
        us = _event_time_us(fn, reps)
        rows[name] = {"us": round(us, 1), "algorithmic_MB": round(nbytes / 1e6, 2), "GBps": round(nbytes / us / 1e3, 1),
                      "frac_of_8TBps": round(nbytes / us / 1e3 / PEAK_HBM_GBS, 4)}
    g = torch.Generator().manual_seed(0)
    wav = (torch.rand(32, 163840, generator=g) * 2 - 1).to(dev)
    add("stft_mag (32 x 163840 -> 32 x 1025 x 256)", 32 * (163840 + 1025 * 256) * 4, lambda: spectrogram_torch(wav, 2048, 640, 2048))
    spec = spectrogram_torch(wav, 2048, 640, 2048)
    add("mel_log (32 x 1025 x 256 -> 32 x 128 x 256)", 32 * (1025 + 128) * 256 * 4, lambda: spec_to_mel_torch(spec, 2048, 128, 32000, 0.0, None))
    x = torch.randn(4096, 192, generator=g).to(dev); cb = torch.randn(1024, 192, generator=g).to(dev)
    add("vq_nearest (4096 x 192 vs 1024 codes)", (4096 * 192 * 2 + 1024 * 192) * 4 + 4096 * 8, lambda: ops.vq_nearest(x, cb))
    idx = ops.vq_nearest(x, cb, want_xq=False)[0]
    cs, ea, em = torch.full((1024,), 4.0, device=dev), cb.clone() * 4, cb.clone()
    add("vq_ema_update (4096 rows, 1024 codes)", (4096 * 192 + 3 * 1024 * 192 + 2 * 1024) * 4 + 4096 * 8,
        lambda: ops.vq_ema_update(x, idx, cs, ea, em, 0.99, 1e-5))
    # (the nearest codes of Gaussian rows against a Gaussian codebook are skewed: the fullest code takes ~200 of the 4096 rows and
    # its gather chain sets the time; evenly used codes -- a trained codebook's regime -- are the second row)
    idx_u = torch.randint(0, 1024, (4096,), generator=g).to(dev)
    add("vq_ema_update (4096 rows, 1024 codes, evenly used)", (4096 * 192 + 3 * 1024 * 192 + 2 * 1024) * 4 + 4096 * 8,
        lambda: ops.vq_ema_update(x, idx_u, cs, ea, em, 0.99, 1e-5))
    M, D = 9248, 512
    xs = torch.randn(M, D, generator=g).to(dev); gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    y = torch.empty(M, D, dtype=torch.bfloat16, device=dev); mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    add("layernorm_fwd (9248 x 512, f32 -> bf16)", M * D * 6 + M * 8, lambda: ops.layernorm_fwd(xs, gam, bet, y, mean, rstd))
    n = 21_460_000 // 8 * 8
    pp, gg, m1, m2 = (torch.zeros(n, device=dev) for _ in range(4))
    sh = torch.zeros(n, dtype=torch.bfloat16, device=dev); state = torch.zeros(8, device=dev)
    ops.adamw_schedule(state, 1e-4, 0.9, 0.96, 500)
    add("adamw (21.46 M parameters, + bf16 shadow, zero-grad)", n * 34, lambda: ops.adamw(pp, gg, m1, m2, sh, state, 0.9, 0.96, 1e-8, 0.01, zero_grad=True))
    return rows


def vqvae_leg(dev, steps, warmup, cpu_leg=True):
    """BASELINE.json metric, second clause: spectrogram frames/s of the full two-phase VQ-VAE-GAN step (config #3: B 32 x
    163 840 samples = 256 frames, fp32 arithmetic with split-bf16 matrix-core convolutions).  Returns the "vqvae" object."""
    from ttts_amd import ops
    from ttts_amd.vqvae.train import SyntheticVqvaeBatches, VqvaeTrainer, get_hparams
    B, NS = 32, 163840
    hps = get_hparams()

    def make_trainer():
        torch.manual_seed(4321)        # (same initial state for the default and the TF32-class trainer: their losses are an A/B)
        t_ = VqvaeTrainer(hps, device=dev)
        cb = t_.net_g.quantizer.vq.layers[0]._codebook
        with torch.no_grad():          # codebook pre-initialised (k-means excluded from timing)
            cb.inited.fill_(1); cb.embed.normal_(0, 0.3); cb.embed_avg.copy_(cb.embed * 4); cb.cluster_size.fill_(4.0)
        torch.manual_seed(8765)        # slice starts / posterior noise of the steps
        return t_
    tr = make_trainer()
    data = next(iter(SyntheticVqvaeBatches(B, n_samples=NS, device=dev)))       # resident in HBM before the timed region
    def timed(step):
        for _ in range(warmup):
            o = step(data)
        torch.cuda.synchronize()
        # The eager step creates ~10^5 Python objects (autograd nodes, tensors, ctypes arguments); with the GPT leg's objects still alive
        # the cyclic collector's full passes land inside the timed steps.  Everything alive now is long-lived: park it in the permanent
        # generation for the duration of the timing (TTTS_BENCH_GC_FREEZE=0 to see the difference).
        frozen = os.environ.get("TTTS_BENCH_GC_FREEZE", "1") == "1"
        if frozen:
            gc.collect(); gc.freeze()
        t0 = time.perf_counter()
        for _ in range(steps):
            o = step(data)
        torch.cuda.synchronize()
        dt_ = (time.perf_counter() - t0) / steps
        if frozen:
            gc.unfreeze()
        return dt_, o
    # two ways to issue the same step: eager launches with the independent branches (sub-discriminators, prior / posterior paths,
    # the three ResBlocks of every MRF stage) on side streams, and one hipGraph replay (which keeps the sub-discriminator and
    # prior / posterior fan-outs: modules.side_streams, TTTS_CAPTURE_POOLS).  `value` is the faster one; both are reported.
    dt_eager, out = timed(tr.train_step)
    vals = {k: float(v) for k, v in out.items()}          # after warmup + steps eager steps: the count the TF32-class column is taken at
    assert all(v == v for v in vals.values()), vals
    dt_graph, out_g = timed(tr.train_step_graphed)
    graphed = tr._graph_state["graph"] is not None
    if dt_graph < dt_eager and graphed:
        dt, mode = dt_graph, "one hipGraph replay per step"
        vg = {k: float(v) for k, v in out_g.items()}
        assert all(v == v for v in vg.values()), vg
    else:
        dt, mode = dt_eager, "eager launches, independent branches on side streams"
    # dominant kernel family: the convolutions (implicit GEMM on the matrix cores), timed eagerly with HIP events
    fam = {"conv1d_fwd": [0, 0.0, 0.0], "conv1d_dgrad": [0, 0.0, 0.0], "conv1d_wgrad": [0, 0.0, 0.0]}
    saved, recs = {}, []

    def wrap(name, flops_fn):
        fn = saved[name]

        def w(*a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r = fn(*a, **kw); e1.record()
            recs.append((name, flops_fn(r, *a), e0, e1))
            return r
        setattr(ops, name, w)
    # algorithmic FLOPs = 2 x (output elements) x (reduction length C_in / groups x taps)
    def fl_fwd(y, x, w_, *a): return 2.0 * y.numel() * w_.shape[1] * w_.shape[2]
    def fl_dgrad(dx, dy, w_, *a): return 2.0 * dy.numel() * w_.shape[1] * w_.shape[2]
    def fl_wgrad(dw, dy, x, *a): return 2.0 * dy.shape[0] * dy.shape[2] * dw.numel()
    for name in ("conv1d_fwd", "conv1d_dgrad", "conv1d_wgrad"):
        saved[name] = getattr(ops, name)
    env_prev = {k: os.environ.get(k) for k in ("TTTS_BRANCH_STREAMS", "TTTS_D_STREAMS")}
    timing_note, dt_one, host_issue_s, park_s = None, None, None, None
    try:
        # (one stream for this pass: with the branches on side streams, overlapping launches would share the chip and every
        # event pair would read longer than the kernel alone)
        os.environ.update({"TTTS_BRANCH_STREAMS": "0", "TTTS_D_STREAMS": "0"})
        # An event pair measures DEVICE time only while the host runs ahead of the device: the device is parked (a spin kernel)
        # for longer than the host needs to issue the whole step.  How long that is depends on the box, so it is measured: the
        # un-instrumented one-stream step first (its wall time bounds the host's issue time from above), then instrumented pass 1
        # (parked 2 x that; absorbs first-use allocations; its host time is read off the clock), then pass 2 parked 1.3 x pass 1's
        # host time -- pass 2 is the one that counts.
        for k, v in saved.items():
            setattr(ops, k, v)                      # un-instrumented for the plain one-stream step
        tr.train_step(data); torch.cuda.synchronize()
        t0 = time.perf_counter(); tr.train_step(data); torch.cuda.synchronize(); dt_one = time.perf_counter() - t0
        spin = int(1e7)                             # calibrate the spin kernel: cycles -> seconds
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(spin); torch.cuda.synchronize()
        e0.record(); torch.cuda._sleep(spin); e1.record(); torch.cuda.synchronize()
        cyc_per_s = spin / max(e0.elapsed_time(e1) * 1e-3, 1e-6)
        for name in list(saved):
            wrap(name, {"conv1d_fwd": fl_fwd, "conv1d_dgrad": fl_dgrad, "conv1d_wgrad": fl_wgrad}[name])
        park_s = 2.0 * dt_one
        for attempt in range(2):
            recs.clear()
            torch.cuda._sleep(int(park_s * cyc_per_s))
            t0 = time.perf_counter(); tr.train_step(data); host_issue_s = time.perf_counter() - t0
            torch.cuda.synchronize()
            if attempt == 0:
                park_s = 1.3 * host_issue_s
    finally:
        for k, v in saved.items():
            setattr(ops, k, v)
        for k, v in env_prev.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    for name, fl, e0, e1 in recs:
        f = fam[name]; f[0] += 1; f[1] += e0.elapsed_time(e1) * 1e-3; f[2] += fl
    tot_s, tot_f, tot_n = sum(f[1] for f in fam.values()), sum(f[2] for f in fam.values()), sum(f[0] for f in fam.values())
    ach = tot_f / tot_s / 1e12
    peak = PEAK_BF16_TFLOPS / 3.0               # an fp32 product costs three bf16 MFMA products (hi*hi + hi*lo + lo*hi)
    # the convolution launches are a SUBSET of the one-stream step: their summed device time cannot exceed that step's wall time.
    # If it does, the event pairs measured host gaps (the host fell behind the device) and no fraction is reported.
    timing_invalid = bool(dt_one is None or tot_s > 1.02 * dt_one)
    try:
        conv_traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["vqvae_conv_family"]["bytes_per_step"]
    except Exception:
        conv_traffic = None
    # the same step with the forward / data-gradient convolutions in the reference's own GPU arithmetic class (TF32: 11 significant
    # bits; here fp16 x fp16 single-pass MFMA products with fp32 accumulation, weight gradients still split-bf16, dynamic loss scale
    # starting at 2^10): reported BESIDE the fp32-equivalent default, never as `value`.  A FRESH trainer with the default trainer's
    # initial state, timed over the same warmup + steps eager steps, so that its losses stand beside the default's at the same step
    # count (an A/B up to the step's own run-to-run noise: a few float-atomic kernels, ~5 % on these losses after a handful of steps)
    max_mem = torch.cuda.max_memory_allocated()
    tr = out = out_g = None
    recs.clear()
    gc.collect(); torch.cuda.empty_cache()
    tf32 = None
    prev_prec = ops.set_conv_precision("tf32class")
    try:
        tr2 = make_trainer()
        dt_te, out_t = timed(tr2.train_step)
        vt = {k: float(v) for k, v in out_t.items()}
        assert all(v == v for v in vt.values()), vt
        # (the losses above are taken after the eager steps: the same count as the default column; the replay is timed afterwards)
        dt_tg, out_tg = timed(tr2.train_step_graphed)
        tg_ok = tr2._graph_state["graph"] is not None and all(float(v) == float(v) for v in out_tg.values())
        dt_t = dt_tg if (tg_ok and dt_tg < dt_te) else dt_te
        tf32 = {"ms_per_step": round(dt_t * 1e3, 2), "ms_per_step_eager_streams": round(dt_te * 1e3, 2),
                "ms_per_step_graph_replay": round(dt_tg * 1e3, 2) if tg_ok else None,
                "value": round(B * 256 / dt_t, 1), "unit": "frames/s",
                "dtype": "forward / data-gradient convolution products as ONE fp16 x fp16 MFMA (11 significant bits = TF32's, the reference's "
                         "cuDNN arithmetic, ttts/vqvae/train.py:34-36; 5 exponent bits: operands saturate at 65504, flush to zero at 3e-8 -- both "
                         "counted on the device -- and keep fewer bits below 6.1e-5), fp32 accumulation; weight gradients split-bf16 x3; dynamic loss scale (GradScaler's rule, "
                         "initial 2^10) divided out of the gradient arenas; everything else as the default",
                "parity": "tests/test_gpu_vqvae.py::test_tf32class_conv_accuracy (1.5e-3 of the output range per convolution), "
                          "::test_full_step_in_tf32class_mode_against_the_reference_fixture (losses within 1e-3 of the reference fixture, "
                          "gradient norms 2e-3, quantizer input 3e-3 of its range; code flips only on audited near ties: 1 of 50 rows), "
                          "::test_dynamic_loss_scale_skips_on_overflow_and_grows_after_clean_steps, ::test_tf32class_and_fp8_keep_nan_and_count_range_events",
                "roof_note": "one product per pair: this mode's matrix-core roof is the full 2500 TFLOP/s, not 2500 / 3",
                "algorithmic_tflops": round(1.97e9 * B * 256 / dt_t / 1e12, 1),
                "losses_after_steps": steps + warmup,
                "losses": {k: round(v, 4) for k, v in vt.items() if k.startswith(("loss", "kl", "grad"))},
                "loss_scale": vt.get("loss_scale"), "f16_saturated": vt.get("f16_saturated"), "f16_flushed": vt.get("f16_flushed"),
                "skipped_steps": vt.get("skipped_steps")}
        max_mem = max(max_mem, torch.cuda.max_memory_allocated())
        tr2 = None
    finally:
        ops.set_conv_precision(prev_prec)
        gc.collect(); torch.cuda.empty_cache()
    res = {"metric": "vqvae_gan_train_frames_per_sec", "value": round(B * 256 / dt, 1), "unit": "frames/s",
           "ms_per_step": round(dt * 1e3, 2), "ms_per_step_eager_streams": round(dt_eager * 1e3, 2),
           "ms_per_step_graph_replay": round(dt_graph * 1e3, 2) if graphed else None, "steps": steps, "warmup": warmup, "dtype": "f32 (conv products as split-bf16 x3 on the bf16 MFMA, fp32 accumulate; VQ distances on the exact f32 MFMA)",
           "vq_code_parity": "nearest-code kernel bit-exact vs the C oracle; through the assembled model the split-bf16 convolutions perturb the "
                             "quantizer input by <= 1.1e-5 of its range: code indices are held to 'differ only on audited near-tie rows, "
                             "<= max(2, 1 %) of frames' (tests/test_gpu_fullsize.py, test_gpu_vqvae.py) -- measured 0 of 50 clips; exact-conv mode is bit-equal to the reference fixture",
           "config": {"workload": "VQ-VAE-GAN two-phase step (spectrograms, SynthesizerTrn, mel, MPD x2, 6 losses, 2 x AdamW, codebook EMA), "
                                  "batch 32 x 163 840 samples (256 frames), %s" % mode},
           "algorithmic_tflops": round(1.97e9 * B * 256 / dt / 1e12, 1),
           "algorithmic_tflops_note": "1.97 GFLOP per frame = the REFERENCE step, which also computes (and discards) the discriminator's "
                                      "parameter gradients in the generator phase; this build skips them, so executed FLOPs are lower",
           "roofline": {"bound": "mfma", "kernel": "conv1d_{fwd,dgrad,wgrad} (split-bf16 implicit GEMM; %d launches per step)" % tot_n,
                        "achieved": None if timing_invalid else round(ach, 1), "peak": round(peak, 1), "unit": "TFLOP/s",
                        "frac": None if timing_invalid else round(ach / peak, 4), "timing_invalid": timing_invalid,
                        "family_gflop_per_step": round(tot_f / 1e9, 1),
                        "one_stream_step_ms": None if dt_one is None else round(dt_one * 1e3, 2),
                        "host_issue_ms": None if host_issue_s is None else round(host_issue_s * 1e3, 2),
                        "park_ms": None if park_s is None else round(park_s * 1e3, 2),
                        "traffic": conv_traffic, "traffic_note": "HBM-side bytes of the whole family per STEP (incl. operand pre-passes "
                        "and slab reductions) from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json: FETCH_SIZE x 2 + WRITE_SIZE), "
                        "not measured in this run",
                        "ms_per_step": round(tot_s * 1e3, 2), "timing": "HIP events around every convolution launch of one eager ONE-STREAM step, device parked 1.3 x the "
                        "measured host issue time so the host stays ahead; second instrumented pass (the first absorbs allocations); "
                        "timing_invalid when the family's summed time exceeds the un-instrumented one-stream step",
                        "families_ms": {k: round(v[1] * 1e3, 2) for k, v in fam.items()}},
           "losses_after_steps": steps + warmup, "losses": {k: round(v, 4) for k, v in vals.items()},
           "max_mem_gb": round(max_mem / 2 ** 30, 1), "tf32class": tf32}
    if cpu_leg:
        res["cpu_baseline"] = vqvae_cpu_baseline()
    return res


def vqvae_cpu_baseline():
    """The oracle's restatement of the same two-phase step (oracle/vqvae_ref.gan_step_losses + CPU autograd + AdamW on the
    discriminator between the phases) on the host cores, bounded sample: ONE clip of BASELINE config #3's length (256 frames =
    163 840 samples, 32-frame decoder segment) instead of the batch of 32 -- frames/s is per clip, so the figure scales linearly."""
    from oracle import vqvae_ref
    from ttts_amd.vqvae.train import get_hparams
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    surf = json.load(open(os.path.join(ROOT, "tests", "golden", "surface.json")))
    hps = get_hparams()
    cfg = {k: getattr(hps.vqvae, k) for k in ("n_heads", "n_layers", "kernel_size", "inter_channels", "hidden_channels", "resblock",
                                               "resblock_kernel_sizes", "resblock_dilation_sizes", "upsample_rates",
                                               "upsample_initial_channel", "upsample_kernel_sizes")}
    h = {k: getattr(hps.data, k) for k in ("filter_length", "hop_length", "win_length", "n_mel_channels", "sampling_rate", "mel_fmin", "mel_fmax")}
    h.update({k: getattr(hps.train, k) for k in ("segment_size", "c_mel", "c_kl", "learning_rate", "betas", "eps")})
    sd_g = {k: vqvae_ref.det_fill(k, s, 0.4) for k, s, *_ in surf["vqvae_g"] if not k.startswith("quantizer.") and not k.endswith("filter")}
    for k, v in sd_g.items():
        if v.is_floating_point():
            v.requires_grad_(True)
    sd_d = {k: vqvae_ref.det_fill(k, s, 0.6).requires_grad_(True) for k, s, *_ in surf["vqvae_d"]}
    embed = vqvae_ref.det_fill("codebook.embed", (1024, 192)) * 2.0
    frames = int(os.environ.get("TTTS_CPU_BASELINE_FRAMES", "256"))   # BASELINE config #3's clip length; one clip instead of 32
    g = torch.Generator().manual_seed(7)
    wav = (torch.rand(1, frames * 640, generator=g) - 0.5)
    opt_d = torch.optim.AdamW(list(sd_d.values()), h["learning_rate"], betas=h["betas"], eps=h["eps"])
    opt_g = torch.optim.AdamW([v for v in sd_g.values() if v.requires_grad], h["learning_rate"], betas=h["betas"], eps=h["eps"])

    def one_step():
        buffers = {"embed": embed.clone(), "embed_avg": embed * 4.0, "cluster_size": torch.full((1024,), 4.0)}

        def d_phase(ld):
            ld.backward(); opt_d.step(); opt_d.zero_grad()
        _, lg, _ = vqvae_ref.gan_step_losses(sd_g, sd_d, cfg, h, buffers, wav, torch.tensor([frames * 640]), torch.randint(1, 255, (1, 16), generator=g),
                                             torch.tensor([16]), torch.randn(1, 192, frames, generator=g), torch.randn(1, 192, frames, generator=g),
                                             torch.tensor([frames - 32]), d_update=d_phase)
        lg.backward(); opt_g.step(); opt_g.zero_grad(); opt_d.zero_grad()
    t0 = time.time(); one_step(); warm = time.time() - t0
    n, t0 = 0, time.time()
    while n < 1 or (time.time() - t0 + 1.5 * warm < 20.0 and n < 3):
        one_step(); n += 1
    dt = (time.time() - t0) / n
    return {"value": round(frames / dt, 2), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "%d full two-phase steps (fwd, D backward + AdamW, G backward + AdamW) of the oracle on ONE clip of %d frames "
                      "(%d samples, 32-frame decoder segment; config #3 is 32 such clips), fp32, %d threads, after 1 warm-up step; "
                      "%.2f s/step" % (n, frames, frames * 640, threads, dt)}


class DiffusionLeg:
    """BASELINE config #5: the diffusion mel-denoiser train step (AA_diffusion, ttts/diffusion/train.py:156-203) at batch 16,
    x_start (16,100,400), latent (16,512,100), refer (16,100,200).  Arithmetic as built: fp32 with split-bf16 matrix-core
    convolutions and attention -- WIDER than the config's "bf16 + fp8" (stated in `dtype`) -- and, beside it, the
    reduced-precision modes (fp8 GEMMs + bf16 attention; the same with TF32-class k = 3 convolutions).
    Two phases, because order matters for eager timings (HISTORY 18.4: a process that has recorded hipGraphs issues eager launches
    more slowly): eager() times every mode launch by launch BEFORE anything in the process has recorded a graph; graphs(), called
    after the VQ-VAE-GAN leg's eager timing, replays the same steps from recorded hipGraphs."""
    MODES = ("default", "fp8", "fp8_tf32class")

    def __init__(self, dev, steps, warmup):
        from ttts_amd.diffusion.train import DiffusionTrainer
        self.B, self.C, self.T, self.Tl, self.Tr = 16, 512, 400, 100, 200
        self.steps, self.warmup = steps, warmup
        cfg = {"train": {"lr": 1e-4, "timesteps": 1000},
               "aa_diffusion": dict(in_channels=100, out_channels=200, model_channels=self.C, num_heads=16, num_layers=6, in_latent_channels=512,
                                    dropout=0, layer_drop=0.1)}
        self.tr = DiffusionTrainer(cfg, device=dev)
        with torch.no_grad():      # the reference zero-initialises every attention output projection: give them signal
            for k, p in self.tr.diffusion.named_parameters():
                if k.endswith("proj_out.weight"):
                    p.normal_(0, 0.02)
        g = torch.Generator().manual_seed(0)
        B, T, Tr, Tl = self.B, self.T, self.Tr, self.Tl
        self.mel = (torch.randn(B, 100, T, generator=g) * 2 - 4).to(dev); self.ref = (torch.randn(B, 100, Tr, generator=g) * 2 - 4).to(dev)
        self.lat = torch.randn(B, 512, Tl, generator=g).to(dev)                # inputs resident in HBM before the timed region
        self.dt_e, self.dt_g, self.out, self.loss = {}, {}, {}, {}

    def _mode(self, mode):
        """context: the precision switches of a mode"""
        import contextlib
        from ttts_amd import ops as _ops
        from ttts_amd.diffusion import aa_model as _aa

        @contextlib.contextmanager
        def cm():
            prev_mode = _aa.set_precision("fp8" if mode != "default" else "f32")
            prev_conv = _ops.set_conv_precision("tf32class" if mode == "fp8_tf32class" else "split_bf16")
            try:
                yield
            finally:
                _ops.set_conv_precision(prev_conv)
                _aa.set_precision(prev_mode)
        return cm()

    def _timed(self, fn, n_warm):
        for _ in range(n_warm):
            o = fn(self.mel, self.ref, self.lat)
        torch.cuda.synchronize(); t0_ = time.perf_counter()
        for _ in range(self.steps):
            o = fn(self.mel, self.ref, self.lat)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0_) / self.steps, o

    def eager(self):
        for mode in self.MODES:
            with self._mode(mode):
                self.dt_e[mode], o = self._timed(self.tr.train_step, self.warmup)
                self.out[mode] = {k: (float(v) if torch.is_tensor(v) and v.numel() == 1 else None) for k, v in o.items() if k != "terms"}
                self.loss[mode] = self.out[mode]["loss"]
                assert self.loss[mode] == self.loss[mode], "non-finite diffusion loss (%s)" % mode
        return self

    def graphs(self):
        """The same steps replayed from recorded hipGraphs: one per layer-drop pattern with at most one skipped layer (85 % of the
        steps), recorded by the first call, ahead of the clock; rarer patterns run launch by launch inside the timed region."""
        for mode in self.MODES:
            with self._mode(mode):
                try:
                    dt_g, o = self._timed(self.tr.train_step_graphed, 3)
                    ok = not self.tr._gstate.get("failed") and len(self.tr._gstate["graphs"]) > 0
                    lv = float(o["loss"])
                    assert lv == lv, "non-finite diffusion loss in graph replay (%s)" % mode
                    self.dt_g[mode] = dt_g if ok else None
                    self.n_graphs = len(self.tr._gstate["graphs"])
                except Exception as err:                  # noqa: BLE001 -- report, keep the eager number
                    print("bench: graphed diffusion step (%s) failed: %s" % (mode, str(err).splitlines()[0][:160]), file=sys.stderr, flush=True)
                    self.dt_g[mode] = None
        return self

    def _pick(self, mode):
        e, g = self.dt_e[mode], self.dt_g.get(mode)
        return e if g is None else min(e, g)

    def _times(self, mode):
        g = self.dt_g.get(mode)
        return {"ms_per_step": round(self._pick(mode) * 1e3, 2), "ms_per_step_eager": round(self.dt_e[mode] * 1e3, 2),
                "ms_per_step_graph_replay": None if g is None else round(g * 1e3, 2)}

    def result(self, cpu_leg=True):
        B, C, T, Tl, Tr = self.B, self.C, self.T, self.Tl, self.Tr
        dt = self._pick("default")
        fp8 = dict(self._times("fp8"))
        fp8.update({"value": round(B * T / self._pick("fp8"), 1), "unit": "frames/s", "loss": round(self.loss["fp8"], 4),
                    "dtype": "f32 activations; the 1 x 1 convolutions / linear layers (qkv, proj_out, ResBlock input conv, integrating conv, "
                             "timestep MLP: forward, data gradient and weight gradient) as e4m3 x e4m3 on v_mfma_f32_32x32x16_fp8_fp8 with "
                             "per-tensor current scaling and fp32 accumulation; attention core (fused, csrc/attn_relpos.hip) on plain bf16 "
                             "operands with fp32 softmax / accumulation; k = 3 convolutions split-bf16",
                    "parity": "tests/test_gpu_fp8.py: kernels within 6e-5 of the output range of the oracle's quantised arithmetic (measured 1.6e-5); "
                              "step vs the reference fixture: loss within 2 %, model output within 15 % relative L2, gradient cosines >= 0.95; "
                              "tests/test_gpu_diffusion.py::test_fused_relpos_attention_vs_fp64 (bf16 operands: 2e-2 of range)"})
        o9 = self.out["fp8_tf32class"]
        sub = dict(self._times("fp8_tf32class"))
        sub.update({"value": round(B * T / self._pick("fp8_tf32class"), 1), "loss": round(self.loss["fp8_tf32class"], 4),
                    "loss_scale": o9.get("loss_scale"), "f16_saturated": o9.get("f16_saturated"), "f16_flushed": o9.get("f16_flushed"),
                    "skipped_steps": o9.get("skipped_steps"),
                    "dtype": "as above, and the k = 3 convolutions' forward / data gradient as ONE fp16 x fp16 MFMA product "
                             "(11 significant bits, fp32 accumulation; weight gradients split-bf16), dynamic loss scale "
                             "(GradScaler's rule on device counters of the fp16 conversions' range events)",
                    "parity": "tools/exp/tf32_diffusion_check.py (config #5 shapes, same inputs): loss equal to 7 digits, "
                              "gradient arena 6.9e-4 relative L2 of the split-bf16 default's, worst tensor 1.8e-3"})
        fp8["with_tf32class_convs"] = sub

        def attn(t):      # qkv + proj 1x1 convs, QK^T and PV
            return 2 * t * C * 3 * C + 2 * t * C * C + 4 * t * t * C

        def resb(t):
            return 2 * t * C * C + 2 * t * C * C * 3
        fwd = (2 * Tl * 512 * C * 3 + 3 * attn(Tl)) + (2 * Tr * 100 * C * 3 + 3 * attn(Tr) + 2 * (Tr + 32) * C * C * 3 + 4 * attn(Tr + 32)) \
            + 3 * (resb(T) + attn(T)) + 2 * T * 100 * C * 3 + 2 * T * 2 * C * C + 6 * (resb(T) + attn(T)) + 3 * resb(T) + 2 * T * C * 200 * 3
        ach = 3.0 * fwd * B / dt / 1e12
        peak = PEAK_BF16_TFLOPS / 3.0
        # the reduced-precision leg's own roof: its GEMMs are ONE product per pair on v_mfma_f32_32x32x16_fp8_fp8 -- the NON-scaled fp8
        # form, which issues at the bf16 rate (MI355X_MICROARCH.md: only the MX-scaled K = 64 / 128 instructions reach ~5 PF) -- and its
        # attention one bf16 product; the k = 3 convolutions stay three bf16 products (one in the tf32class sub-leg)
        a8 = 3.0 * fwd * B / self._pick("fp8") / 1e12
        fp8["roofline"] = {"bound": "mfma", "kernel": "whole step (fp8 1 x 1 GEMMs + bf16 attention + split-bf16 k = 3 convolutions)",
                           "achieved": round(a8, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(a8 / PEAK_BF16_TFLOPS, 4),
                           "traffic": None,
                           "note": "peak = the dense bf16 rate: the instruction used, v_mfma_f32_32x32x16_fp8_fp8 (non-scaled), issues at "
                                   "the bf16 rate; the ~5 PF fp8 roof needs v_mfma_scale_f32_32x32x64_f8f6f4, which this build does not use"}
        try:
            diff_traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["diffusion_step"]["bytes_per_step"]
        except Exception:
            diff_traffic = None
        g = self.dt_g.get("default")
        res = {"metric": "diffusion_train_mel_frames_per_sec", "value": round(B * T / dt, 1), "unit": "frames/s"}
        res.update(self._times("default"))
        res.update({"graphs_recorded": getattr(self, "n_graphs", 0), "steps": self.steps, "warmup": self.warmup,
                    "dtype": "f32 (conv / linear / attention products as split-bf16 x3 on the bf16 MFMA, fp32 accumulation and softmax) -- wider than config #5's bf16 + fp8",
                    "config": {"workload": "AA_diffusion train step (q_sample, model, mse + learned-range VB, backward, clip 1.0, AdamW), batch 16 x "
                                           "(100 x 400 mel, 512 x 100 latent, 100 x 200 reference), 43.2 M parameters, %s"
                                           % ("hipGraph replay (one recording per layer-drop pattern)" if (g is not None and g <= self.dt_e["default"]) else "eager launches")},
                    "roofline": {"bound": "mfma", "kernel": "whole step (convolution family + fused attention)", "achieved": round(ach, 2),
                                 "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": diff_traffic,
                                 "traffic_note": "HBM-side bytes of ALL kernels of one step from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json: "
                                                 "FETCH_SIZE x 2 + WRITE_SIZE), not measured in this run",
                                 "note": "algorithmic FLOPs = 3 x forward (%.1f GFLOP per sample) / step time; peak = bf16 MFMA / 3 (an fp32 product "
                                         "costs three bf16 products)" % (fwd / 1e9)},
                    "loss": round(self.loss["default"], 4), "grad_norm": round(self.out["default"]["grad_norm"], 4), "fp8_gemms": fp8})
        if cpu_leg:
            res["cpu_baseline"] = diffusion_cpu_baseline()
        self.tr = None
        return res


def diffusion_cpu_baseline():
    """The oracle's restatement of the same step (oracle/diffusion_ref: q_sample, AA_diffusion forward, training_losses, CPU autograd,
    clip, AdamW) on the host cores: bounded sample = ONE item of config #5's shapes instead of the batch of 16 (frames/s per item)."""
    from oracle import diffusion_ref as D
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    cfg = dict(in_channels=100, out_channels=200, model_channels=512, num_heads=16, num_layers=6, in_latent_channels=512, dropout=0, layer_drop=0.1)
    sd = {k: D.det_fill(k, shp, 0.5).requires_grad_(True) for k, shp in D.param_spec(cfg)}
    opt = torch.optim.AdamW(list(sd.values()), 1e-4, betas=(0.9, 0.999), weight_decay=0.01)
    tab = D.diffusion_tables(1000)
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(1, 100, 400, generator=g).clamp(-1, 1); lat = torch.randn(1, 512, 100, generator=g); ref = torch.randn(1, 100, 200, generator=g)

    def one_step():
        t = torch.randint(0, 1000, (1,), generator=g)
        noise = torch.randn(x0.shape, generator=g)
        x_t = D.q_sample(tab, x0, t, noise)
        out = D.aa_diffusion_forward(sd, cfg, x_t, t, lat, ref)
        loss = D.training_losses(tab, out, x0, x_t, t, noise)["loss"].mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(sd.values()), 1.0)
        opt.step(); opt.zero_grad()
    t0 = time.time(); one_step(); warm = time.time() - t0
    n, t0 = 0, time.time()
    while n < 1 or (time.time() - t0 + 1.5 * warm < 15.0 and n < 25):
        one_step(); n += 1
    dt = (time.time() - t0) / n
    return {"value": round(400 / dt, 2), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "%d train steps of the oracle on ONE item of config #5's shapes (100 x 400 mel, 512 x 100 latent, 100 x 200 reference; "
                      "the config is 16 such items), fp32, %d threads, after 1 warm-up step; %.2f s/step" % (n, threads, dt)}


class KernelTimer:
    """Brackets every launch of the instrumented ops with HIP events on the launch stream (torch's current stream,
    which is the stream handed to the C ABI) and accumulates (time, algorithmic flops) per kernel family."""

    def __init__(self, ops, k_true=None):
        self.ops, self.records, self.saved, self.step = ops, [], {}, 0
        self.k_true = k_true or {}     # padded reduction length -> algorithmic one (the heads' dX GEMMs run over the logits pitch)

    def next_step(self):
        self.step += 1

    def _wrap(self, name, flops_fn):
        fn = getattr(self.ops, name)
        self.saved[name] = fn

        def wrapped(*a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **kw)
            e1.record()
            fam, fl, by = flops_fn(*a, **kw)
            self.records.append((fam, fl, e0, e1, self.step, by))
            return r
        setattr(self.ops, name, wrapped)

    def __enter__(self):
        epi = {0: "store_bf16", 1: "gelu", 2: "resid_add", 3: "dgelu", 4: "store_f32"}

        def f_nt(a, b, c, bias=None, aux=None, epilogue=0, n=None, k=None, **kw):
            M, K, N = a.shape[0], (a.shape[1] if k is None else k), (b.shape[0] if n is None else n)
            kt = self.k_true.get(K, K)
            # algorithmic HBM bytes: both operands once, the output(s) once, what the epilogue reads once
            out_b = {0: 2 * M * N, 1: 4 * M * N, 2: 8 * M * N, 3: 4 * M * N, 4: 4 * M * N}[epilogue]
            return "gemm_nt_kernel<%s>" % epi[epilogue], 2.0 * M * N * kt, 2.0 * (M + N) * kt + out_b

        def f_tn(at, bt, c, mo=None, no=None, **kw):
            mo_, no_ = (at.shape[1] if mo is None else mo), (bt.shape[1] if no is None else no)
            return "gemm_tn_kernel", 2.0 * at.shape[0] * mo_ * no_, 2.0 * at.shape[0] * (mo_ + no_) + 8.0 * mo_ * no_

        def f_af(q, k, v, o, lse, B, H, S, dh, *a, **kw):
            return "attn_fwd_kernel", 4.0 * B * H * dh * S * (S + 1) / 2, 2.0 * 4 * B * H * S * dh      # causal-useful QK^T + PV; q, k, v in, o out

        def f_ab(q, k, v, o, d_o, lse, dq, dk, dv, ws, B, H, S, dh, *a, **kw):
            return "attn_bwd(delta+dkdv+dq)", 10.0 * B * H * dh * S * (S + 1) / 2, 2.0 * 13 * B * H * S * dh  # 5 causal-useful matmuls; q, k, v, o, dO x2 reads + 3 writes
        self._wrap("gemm_nt", f_nt)
        self._wrap("gemm_tn_accum", f_tn)
        self._wrap("attn_fwd", f_af)
        self._wrap("attn_bwd", f_ab)
        # the grouped weight-gradient launch (all dW GEMMs of a backward section in one grid) is a method of its plan object
        plan_run = self.ops.TnPlan.run
        self._plan_run = plan_run
        timer = self

        def timed_run(plan):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            plan_run(plan)
            e1.record()
            timer.records.append(("gemm_tn_grouped_kernel", plan.flops, e0, e1, timer.step, plan.bytes))
        self.ops.TnPlan.run = timed_run
        return self

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            setattr(self.ops, k, v)
        self.ops.TnPlan.run = self._plan_run

    def summary(self):
        """{family: [launches, seconds, flops]} over all instrumented steps.  Robust to one-off stalls: launch k of a family is
        timed in every instrumented step, and the MINIMUM over the steps is what counts (a host hiccup while the device waits
        for work, or a clock ramp, lands between one event pair and would otherwise dominate a whole family)."""
        torch.cuda.synchronize()
        per = {}                                  # (family, position within its step) -> [flops, [durations over steps], bytes]
        pos = {}
        for fam, fl, e0, e1, st, by in self.records:
            k = pos.get((fam, st), 0)
            pos[(fam, st)] = k + 1
            per.setdefault((fam, k), [fl, [], by])[1].append(e0.elapsed_time(e1) * 1e-3)
        nsteps = max(1, self.step)
        agg = {}
        for (fam, _k), (fl, durs, by) in per.items():
            a = agg.setdefault(fam, [0, 0.0, 0.0, 0.0])
            a[0] += nsteps
            a[1] += min(durs) * nsteps
            a[2] += fl * nsteps
            # the launch's own roof: whichever of the matrix cores (dense bf16 peak) and HBM (8 TB/s) its algorithmic work needs longer
            a[3] += max(fl / (PEAK_BF16_TFLOPS * 1e12), by / (PEAK_HBM_GBS * 1e9)) * nsteps
        return agg


def cpu_baseline(seconds_budget=24.0):
    """The oracle (CPU restatement of the reference train step, eager) on the host cores, BASELINE.md section 3's protocol inside a
    bounded budget: the main line is fp32 on all cores (capped at 32 threads: torch's intra-op pool degrades beyond that here) at
    the LARGEST batch of {8, 4, 2, 1} x (128 text + 1024 audio tokens) whose projected step fits the budget (tokens/s is per
    step, so a smaller batch is the same workload at a lower arithmetic intensity -- stated in `sample`); two more lines at
    batch 1 give the 8-thread and the bf16-autocast figures of the survey's table."""
    from oracle import gpt_ref
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    sd = gpt_ref.det_state_dict(None)
    opt = gpt_ref.new_opt_state(sd)

    def timed(bs, bf16=False, reps=1):
        batch = gpt_ref.synthetic_batch(B=bs, seed=1234)
        t0 = time.time()
        for _ in range(reps):
            gpt_ref.gpt_train_step(sd, opt, batch, None, None, bf16=bf16, dropout_p=0.1)
        return (time.time() - t0) / reps
    timed(1)                                   # warm-up (allocations, thread pool)
    t1 = timed(1)
    bs = 1
    for cand in (8, 4, 2):                     # a batch-B step costs at most B x the batch-1 step
        if cand * t1 <= seconds_budget * 0.5:
            bs = cand
            break
    dt = timed(bs) if bs > 1 else t1
    out = {"value": bs * MEL_LEN / dt, "unit": "audio-tokens/s", "cores": threads, "kind": "port",
           "sample": "one fp32 train step (fwd+bwd+clip+AdamW, dropout 0.1) of the oracle at batch %d x (128 text + 1024 audio tokens), "
                     "full 6-layer model, %d threads, after a warm-up step; %.2f s/step (batch 1: %.2f s/step)" % (bs, threads, dt, t1)}
    t_bf = timed(1, bf16=True)
    out["bf16_autocast"] = {"value": MEL_LEN / t_bf, "unit": "audio-tokens/s", "cores": threads,
                            "sample": "one step at batch 1 with the bf16 rounding points of torch.autocast; %.2f s/step" % t_bf}
    if threads > 8:
        torch.set_num_threads(8)
        timed(1)
        t8 = timed(1)
        out["threads8"] = {"value": MEL_LEN / t8, "unit": "audio-tokens/s", "cores": 8,
                           "sample": "one fp32 step at batch 1 on 8 threads (the survey container's core count); %.2f s/step" % t8}
        torch.set_num_threads(threads)
    return out


def _respawn_under_torchrun(n):
    """`python bench.py --gpus N` with no torchrun environment: start N ranks (one per GPU) of this very command line under
    torch.distributed.run on the loopback address and a free port, pass their stdout / stderr through, return the exit code.
    (ttts/gpt/train.py is launched by `accelerate launch`, which does the same: one process per GPU.)"""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL's cross-process buffers need it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", default="train", choices=["train", "eager", "graph_nodropout"],
                    help="train: reference training mode (dropout 0.1), whole step replayed from one hipGraph (N = 1) -- "
                         "the headline; eager: same, launch by launch; graph_nodropout: dropout off (diagnostic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--exchange", default="ranged", choices=["ranged", "whole"],
                    help="N > 1: 'ranged' = four range all-reduces, the upper layers' overlapped with the lower layers' backward "
                         "(three graphs); 'whole' = one all-reduce of the whole gradient arena after the backward (two graphs)")
    ap.add_argument("--no-vqvae", action="store_true", help="skip the VQ-VAE-GAN leg (second clause of the metric; N = 1 only)")
    ap.add_argument("--vqvae-steps", type=int, default=8)
    ap.add_argument("--no-diffusion", action="store_true", help="skip the diffusion leg (BASELINE config #5; N = 1 only)")
    ap.add_argument("--diffusion-steps", type=int, default=10)
    ap.add_argument("--grad-dtype", default="f32", choices=["f32", "bf16"],
                    help="N > 1: element type of the gradient all-reduce (bf16 halves the bytes on the xGMI links; replicas stay bit-identical)")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(_respawn_under_torchrun(args.gpus))

    from ttts_amd import ops
    from ttts_amd.gpt import GptEngine, prepare_tokens
    from ttts_amd.parallel import FlatDataParallel, init_distributed

    rank, world, local = init_distributed()
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    # The two eager legs run FIRST (N = 1 only): their host side is part of what they measure (5 800 / 3 000 launches per step issued
    # from Python), and a process that has already recorded and replayed the GPT step's hipGraphs issues eager launches more slowly
    # (same box, same code: the VQ-VAE-GAN leg reads 123.7 ms alone and 128.8 ms after the GPT leg, while its hipGraph replay reads
    # the same 129 ms either way: HISTORY 18.4).  The GPT leg is a graph replay and does not care what ran before it.
    pre = {}
    dleg = None
    if world == 1 and not args.no_diffusion:           # (eager timings first: nothing in the process has recorded a graph yet)
        dleg = DiffusionLeg(dev, args.diffusion_steps, 3).eager()
    if world == 1 and not args.no_vqvae:               # (eager timing, then its own graph recording)
        pre["vqvae"] = vqvae_leg(dev, args.vqvae_steps, 2, cpu_leg=not args.no_cpu_baseline)
    if dleg is not None:
        pre["diffusion"] = dleg.graphs().result(cpu_leg=not args.no_cpu_baseline)
        dleg = None
        torch.cuda.empty_cache()
    if pre:
        ops.release_conv_ctxs(keep_current=False)          # the legs' per-stream convolution scratch (1.5 GB each)
        gc.collect()
        torch.cuda.empty_cache()
    cfg = json.load(open(os.path.join(ROOT, "ttts_amd", "gpt", "config.json")))
    dropout = 0.0 if args.mode == "graph_nodropout" else 0.1
    eng = GptEngine(cfg["gpt"], dev, dropout_p=dropout, seed=rank)
    # random-init weights of the named architecture (no checkpoints offline): GPT-2 init via the module surface
    from ttts_amd.gpt import UnifiedVoice  # noqa: F401  (initialisation rule lives there)
    torch.manual_seed(0)
    with torch.no_grad():
        for k, shp in eng.spec:
            p = eng.view(eng.params, k)
            if len(shp) == 1:
                p.fill_(1.0 if (k.endswith("weight")) else 0.0)
            else:
                p.normal_(0.0, 0.02)
    dp = FlatDataParallel(grad_dtype=torch.bfloat16 if args.grad_dtype == "bf16" else None)
    dp.broadcast_(eng.params)
    eng.refresh_shadows()

    g = torch.Generator().manual_seed(1234 + rank)
    text = torch.randint(1, 255, (B_PER_GPU, TEXT_LEN), generator=g)
    mel = torch.randint(0, 1024, (B_PER_GPU, MEL_LEN), generator=g)
    tl = torch.full((B_PER_GPU,), TEXT_LEN)
    wl = torch.full((B_PER_GPU,), MEL_LEN * 1024)
    text_d, mel_d = text.to(dev), mel.to(dev)     # inputs resident in HBM before the timed region
    tr = cfg["train"]
    w_text, w_mel = tr["text_weight"] * dp.loss_scale(), tr["mel_weight"] * dp.loss_scale()

    def step():
        # token plumbing of UnifiedVoice.forward (clip, mel padding -> STOP, START / STOP framing), one launch into the engine's
        # static token buffers; lengths are host tensors: no sync.  (prepare_tokens is the ~20-launch torch form of the same.)
        eng.set_tokens_raw(text_d, tl, mel_d, wl)
        toks = None
        if args.mode != "eager":
            # N > 1: the gradient all-reduce of the upper layers + heads overlaps the backward of the lower layers
            if dp.enabled and args.exchange == "whole":
                eng.train_step(toks, w_text, w_mel, capture=True, lr=tr["lr"], exchange=lambda: dp.allreduce_grads_(eng.grads))
            else:
                eng.train_step(toks, w_text, w_mel, capture=True, lr=tr["lr"],
                               exchange_range=(lambda lo, hi: dp.allreduce_range_(eng.grads, lo, hi)) if dp.enabled else None)
            return
        eng.forward()
        eng.backward(w_text, w_mel)
        dp.allreduce_grads_(eng.grads)
        eng.optimizer_step(lr=tr["lr"], max_norm=1.0, warmup_steps=500)
        eng.step_count += 1

    # W untimed warm-up steps -- and, when W is small, enough further untimed steps that the timed region starts with the chip at
    # its steady clock (the first replays after an idle stretch run below it; 50 steps = 0.17 s)
    for _ in range(max(args.warmup, 50)):
        step()
    dp.barrier()
    torch.cuda.synchronize()
    # per-step device times beside the wall clock: one HIP event after every step (on the launch stream; an event record costs the
    # device nothing and the host ~1 us) -> the MEDIAN step, SURVEY 8(d)'s statistic.  `value` stays K steps / wall time.
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    dp.barrier()
    torch.cuda.synchronize()
    my_elapsed = time.perf_counter() - t0
    elapsed = dp.max_over_ranks(my_elapsed)
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = dp.max_over_ranks(step_ms[len(step_ms) // 2])
    # per-rank step times (every rank's own clock around the same barrier-bracketed region) and, for N > 1, the cost of the
    # exchange by itself: one SUM all-reduce of the whole fp32 gradient arena, HIP events on the current stream, 5 repetitions
    per_rank_ms = [round(my_elapsed / args.steps * 1e3, 3)]
    allreduce_ms = None
    if dp.enabled:
        import torch.distributed as dist
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank_ms[0])
        per_rank_ms = gathered
        scratch = torch.zeros_like(eng.grads)
        dp.allreduce_grads_(scratch)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dp.allreduce_grads_(scratch)
        e1.record()
        torch.cuda.synchronize()
        allreduce_ms = dp.max_over_ranks(e0.elapsed_time(e1) / 5)
        del scratch
    lt, lm = eng.losses()
    assert lt == lt and lm == lm, "non-finite loss"
    graphed = args.mode != "eager" and eng._graph is not None and eng._graph[0] is not None   # False after a refused capture

    # per-kernel HIP-event timing (same kernels, shapes and data; eager launches), right after the timed steps
    # EVERY rank runs the instrumented steps (they contain the gradient all-reduce); only rank 0 records events.
    roof = None
    saved_mode = args.mode
    args.mode = "eager"
    saved_overlap, eng.overlap_dw = eng.overlap_dw, False   # one stream: an event pair then brackets exactly one kernel
    step()                                    # one eager step untimed: absorbs first-eager-launch costs on every rank
    if rank == 0:
        with KernelTimer(ops, {eng.ld_m: eng.nm, eng.ld_t: eng.nt}) as kt:
            for _ in range(args.profile_steps):
                kt.next_step()
                # park the GPU for ~20 ms first: the host then runs ahead of the device while it enqueues the step's ~300
                # launches + event pairs, so an event pair measures device time only (an eager step is host-bound, and an
                # op that launches two kernels would otherwise include the host's launch gap)
                if hasattr(torch.cuda, "_sleep"):
                    torch.cuda._sleep(int(4e7))
                step()
            agg = kt.summary()
        fam, (cnt, secs, flops, _roof) = max(agg.items(), key=lambda kv: kv[1][1])
        ach = flops / secs / 1e12
        # HBM-side bytes per launch from the separate rocprofv3 --pmc passes (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE), see
        # profiles/pmc_traffic.json; null when no counter run exists for this kernel family
        traffic, attn_busy = None, None
        try:
            pmc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")))
            traffic = pmc["traffic_bytes_per_launch"].get(fam)
            attn_busy = pmc.get("attention_mfma_busy")
        except Exception:
            traffic = None
        step_flops = sum(v[2] for v in agg.values()) / args.profile_steps      # useful FLOPs of one step: every instrumented matrix-core kernel
        step_s = elapsed / args.steps
        roof = {"bound": "mfma", "kernel": fam, "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                "timing": "HIP events on the launch stream around every launch, min over %d instrumented eager steps; traffic from the "
                          "committed rocprofv3 --pmc passes (profiles/pmc_traffic.json), not measured in this run" % args.profile_steps,
                "launches_per_step": cnt // args.profile_steps, "avg_launch_us": round(secs / cnt * 1e6, 2),
                "avg_gflop_per_launch": round(flops / cnt / 1e9, 3),
                "all_kernels_ms_per_step": {k: round(v[1] / args.profile_steps * 1e3, 3) for k, v in sorted(agg.items())},
                "all_kernels_tflops": {k: round(v[2] / v[1] / 1e12, 1) for k, v in sorted(agg.items())},
                # per family: the sum over its launches of max(MFMA time at 2.5 PF, HBM time at 8 TB/s of the algorithmic bytes) / measured time
                "all_kernels_frac_of_own_roof": {k: round(v[3] / v[1], 3) for k, v in sorted(agg.items())},
                # the number the north star is about: useful matrix-core FLOPs of the WHOLE step (attention counted causal-useful)
                # / the headline step time / dense bf16 peak
                "step_gflop": round(step_flops / 1e9, 1), "step_tflops": round(step_flops / step_s / 1e12, 1),
                "step_frac": round(step_flops / step_s / 1e12 / PEAK_BF16_TFLOPS, 4),
                "attention_mfma_busy": attn_busy}
    else:
        for _ in range(args.profile_steps):
            step()
    args.mode = saved_mode
    eng.overlap_dw = saved_overlap
    torch.cuda.synchronize()
    if dp.enabled:
        dp.barrier()

    if rank == 0:
        tokens = world * B_PER_GPU * MEL_LEN * args.steps
        out = {"metric": "gpt_train_audio_tokens_per_sec", "value": round(tokens / elapsed, 1), "unit": "audio-tokens/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / args.steps * 1e3, 3), "ms_per_step_median": round(median_ms, 3),
               "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": "VALL-E GPT train step (ttts/gpt/config.json model: 6 layers, d512, 8 heads, 21.46 M "
                                      "params), batch 8 per GPU x (128 text + 1024 audio tokens) = S 1156, fwd+bwd+clip+AdamW, "
                                      "dropout %.1f, %s" % (dropout, "eager launches" if not graphed else ("hipGraph replay" if not dp.enabled else
                                                                       "hipGraph replay, RCCL all-reduce of the upper layers overlapped with the lower layers' backward")),
                          "global_batch": world * B_PER_GPU, "seq_len": TEXT_LEN + 2 + MEL_LEN + 2, "parallelism": "dp%d" % world,
                          "mode": args.mode, "graph_replay": bool(graphed),
                          "exchange": (args.exchange if dp.enabled else None),
                          "dist_backend": (torch.distributed.get_backend() if dp.enabled else None), "world_size": world,
                          "ranks_seen_by_backend": (torch.distributed.get_world_size() if dp.enabled else 1),
                          "grad_exchange_dtype": (args.grad_dtype if dp.enabled else None),
                          "capture_note": getattr(eng, "_capture_error", None),
                          # the multi-GPU communication budget of DESIGN section 6 rests on an ASSUMED RCCL bus bandwidth (250 GB/s
                          # on the 7-link xGMI mesh): this line's allreduce_arena_ms is the first measurement of it
                          "exchange_budget_note": ("DESIGN section 6 budgets the exchange with an assumed 250 GB/s RCCL bus bandwidth; "
                                                   "allreduce_arena_ms / allreduce_arena_mb in this line are the measured figures"
                                                   if dp.enabled else None)},
               "final_loss_mel": round(lm, 4), "per_rank_ms_per_step": per_rank_ms,
               "allreduce_arena_ms": (None if allreduce_ms is None else round(allreduce_ms, 3)),
               "allreduce_arena_mb": (None if allreduce_ms is None else round(eng.grads.numel() * 4 / 2 ** 20, 1)), "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        if world == 1:
            out["hbm_kernels"] = hbm_kernel_table(dev)
        eng = None                                 # release the GPT replica before the other models' legs
        torch.cuda.empty_cache()
        out.update(pre)
        # flat scalars LAST: a record that keeps only the tail of this line still holds the second clause of the metric
        vq, df = pre.get("vqvae"), pre.get("diffusion")
        if vq is not None:
            out["vqvae_ms_per_step"] = vq["ms_per_step"]
            out["vqvae_frames_per_s"] = vq["value"]
            out["vqvae_ms_eager"] = vq["ms_per_step_eager_streams"]
            out["vqvae_ms_graph"] = vq["ms_per_step_graph_replay"]
            out["vqvae_conv_frac"] = vq["roofline"]["frac"]
            out["vqvae_tf32class_ms"] = None if vq.get("tf32class") is None else vq["tf32class"]["ms_per_step"]
        if df is not None:
            out["diffusion_ms_per_step"] = df["ms_per_step"]
            out["diffusion_ms_eager"] = df["ms_per_step_eager"]
            out["diffusion_ms_graph"] = df["ms_per_step_graph_replay"]
            out["diffusion_fp8_ms"] = None if df.get("fp8_gemms") is None else df["fp8_gemms"]["ms_per_step"]
            out["diffusion_fp8_tf32class_ms"] = (None if not (df.get("fp8_gemms") or {}).get("with_tf32class_convs")
                                                 else df["fp8_gemms"]["with_tf32class_convs"]["ms_per_step"])
        out["gpt_ms_per_step"] = out["ms_per_step"]
        out["gpt_tokens_per_s"] = out["value"]
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
