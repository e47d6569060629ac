#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: GPT train audio-tokens/sec on MI355X (configs[1]: VALL-E GPT train step,
batch 8 per GPU, 1024 audio tokens + 128 text tokens, bf16, full ttts/gpt/config.json model).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one synthetic batch per rank: token plumbing, forward, backward, gradient
all-reduce (N > 1, RCCL), grad-norm + clip, AdamW, LR schedule -- everything ttts/gpt/train.py:96-121 does per
iteration, in the reference's training mode (GPT-2 dropouts 0.1 active).  Inputs are resident in HBM before the timed
region.  Prints ONE JSON line with `roofline` (dominant kernel, HIP-event timed) and `cpu_baseline` (the oracle's
fp32 train step on the host cores, rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA peak (/opt/skills/guides/MI355X_MICROARCH.md)
B_PER_GPU, TEXT_LEN, MEL_LEN = 8, 128, 1024


class KernelTimer:
    """Brackets every launch of the instrumented ops with HIP events on the launch stream (torch's current stream,
    which is the stream handed to the C ABI) and accumulates (time, algorithmic flops) per kernel family."""

    def __init__(self, ops):
        self.ops, self.records, self.saved, self.step = ops, [], {}, 0

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
            fam, fl = flops_fn(*a, **kw)
            self.records.append((fam, fl, e0, e1, self.step))
            return r
        setattr(self.ops, name, wrapped)

    def __enter__(self):
        epi = {0: "store_bf16", 1: "gelu", 2: "resid_add", 3: "dgelu", 4: "store_f32"}

        def f_nt(a, b, c, bias=None, aux=None, epilogue=0, n=None, k=None, **kw):
            M, K, N = a.shape[0], (a.shape[1] if k is None else k), (b.shape[0] if n is None else n)
            return "gemm_nt_kernel<%s>" % epi[epilogue], 2.0 * M * N * K

        def f_tn(at, bt, c, mo=None, no=None, **kw):
            return "gemm_tn_kernel", 2.0 * at.shape[0] * (at.shape[1] if mo is None else mo) * (bt.shape[1] if no is None else no)

        def f_af(q, k, v, o, lse, B, H, S, dh, *a, **kw):
            return "attn_fwd_kernel", 4.0 * B * H * dh * S * (S + 1) / 2      # causal-useful QK^T + PV

        def f_ab(q, k, v, o, d_o, lse, dq, dk, dv, ws, B, H, S, dh, *a, **kw):
            return "attn_bwd(delta+dkdv+dq)", 10.0 * B * H * dh * S * (S + 1) / 2  # 5 causal-useful matmuls
        self._wrap("gemm_nt", f_nt)
        self._wrap("gemm_tn_accum", f_tn)
        self._wrap("attn_fwd", f_af)
        self._wrap("attn_bwd", f_ab)
        return self

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            setattr(self.ops, k, v)

    def summary(self):
        """{family: [launches, seconds, flops]} over all instrumented steps.  Robust to one-off stalls: launch k of a family is
        timed in every instrumented step, and the MINIMUM over the steps is what counts (a host hiccup while the device waits
        for work, or a clock ramp, lands between one event pair and would otherwise dominate a whole family)."""
        torch.cuda.synchronize()
        per = {}                                  # (family, position within its step) -> [flops, [durations over steps]]
        pos = {}
        for fam, fl, e0, e1, st in self.records:
            k = pos.get((fam, st), 0)
            pos[(fam, st)] = k + 1
            per.setdefault((fam, k), [fl, []])[1].append(e0.elapsed_time(e1) * 1e-3)
        nsteps = max(1, self.step)
        agg = {}
        for (fam, _k), (fl, durs) in per.items():
            a = agg.setdefault(fam, [0, 0.0, 0.0])
            a[0] += nsteps
            a[1] += min(durs) * nsteps
            a[2] += fl * nsteps
        return agg


def cpu_baseline(seconds_budget=25.0):
    """The oracle (CPU restatement of the reference train step: fp32, eager, all host cores) on a bounded sample."""
    from oracle import gpt_ref
    threads = min(os.cpu_count() or 1, 32)   # torch's intra-op pool stops scaling (and degrades) beyond ~32 threads here
    torch.set_num_threads(threads)
    sd = gpt_ref.det_state_dict(None)
    opt = gpt_ref.new_opt_state(sd)
    bs = 1
    batch = gpt_ref.synthetic_batch(B=bs, seed=1234)
    t0 = time.time()
    gpt_ref.gpt_train_step(sd, opt, batch, None, None, bf16=False, dropout_p=0.1)   # warm-up (allocations, threads)
    warm = time.time() - t0
    n, t0 = 0, time.time()
    while n < 1 or (time.time() - t0 + 1.5 * warm < seconds_budget and n < 8):
        gpt_ref.gpt_train_step(sd, opt, batch, None, None, bf16=False, dropout_p=0.1)
        n += 1
    dt = (time.time() - t0) / n
    return {"value": bs * MEL_LEN / dt, "unit": "audio-tokens/s", "cores": threads, "kind": "port",
            "sample": "%d fp32 train steps (fwd+bwd+clip+AdamW, dropout 0.1) of the oracle at batch %d x (128 text + 1024 "
                      "audio tokens), full 6-layer model, after 1 warm-up step; %.2f s/step" % (n, bs, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", default="train", choices=["train", "eager", "graph_nodropout"],
                    help="train: reference training mode (dropout 0.1), whole step replayed from one hipGraph (N = 1) -- "
                         "the headline; eager: same, launch by launch; graph_nodropout: dropout off (diagnostic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=3)
    args = ap.parse_args()

    from ttts_amd import ops
    from ttts_amd.gpt import GptEngine, prepare_tokens
    from ttts_amd.parallel import FlatDataParallel, init_distributed

    rank, world, local = init_distributed()
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
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
    dp = FlatDataParallel()
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
        toks = prepare_tokens(eng.c, text_d, tl, mel_d, wl)   # token plumbing (lengths are host tensors: no sync)
        if args.mode != "eager":
            # N > 1: the gradient all-reduce of the upper layers + heads overlaps the backward of the lower layers
            eng.train_step(toks, w_text, w_mel, capture=True, lr=tr["lr"],
                           exchange_range=(lambda lo, hi: dp.allreduce_range_(eng.grads, lo, hi)) if world > 1 else None)
            return
        eng.set_tokens(*toks)
        eng.forward()
        eng.backward(w_text, w_mel)
        dp.allreduce_grads_(eng.grads)
        eng.optimizer_step(lr=tr["lr"], max_norm=1.0, warmup_steps=500)
        eng.step_count += 1

    for _ in range(args.warmup):
        step()
    dp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dp.barrier()
    torch.cuda.synchronize()
    elapsed = dp.max_over_ranks(time.perf_counter() - t0)
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
        with KernelTimer(ops) as kt:
            for _ in range(args.profile_steps):
                kt.next_step()
                # park the GPU for ~20 ms first: the host then runs ahead of the device while it enqueues the step's ~300
                # launches + event pairs, so an event pair measures device time only (an eager step is host-bound, and an
                # op that launches two kernels would otherwise include the host's launch gap)
                if hasattr(torch.cuda, "_sleep"):
                    torch.cuda._sleep(int(4e7))
                step()
            agg = kt.summary()
        fam, (cnt, secs, flops) = max(agg.items(), key=lambda kv: kv[1][1])
        ach = flops / secs / 1e12
        # HBM-side bytes per launch from the separate rocprofv3 --pmc passes (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE), see
        # profiles/pmc_traffic.json; null when no counter run exists for this kernel family
        traffic = None
        try:
            pmc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")))
            traffic = pmc["traffic_bytes_per_launch"].get(fam)
        except Exception:
            traffic = None
        roof = {"bound": "mfma", "kernel": fam, "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                "launches_per_step": cnt // args.profile_steps, "avg_launch_us": round(secs / cnt * 1e6, 2),
                "avg_gflop_per_launch": round(flops / cnt / 1e9, 3),
                "all_kernels_ms_per_step": {k: round(v[1] / args.profile_steps * 1e3, 3) for k, v in sorted(agg.items())},
                "all_kernels_tflops": {k: round(v[2] / v[1] / 1e12, 1) for k, v in sorted(agg.items())}}
    else:
        for _ in range(args.profile_steps):
            step()
    args.mode = saved_mode
    eng.overlap_dw = saved_overlap
    torch.cuda.synchronize()
    if world > 1:
        dp.barrier()

    if rank == 0:
        tokens = world * B_PER_GPU * MEL_LEN * args.steps
        out = {"metric": "gpt_train_audio_tokens_per_sec", "value": round(tokens / elapsed, 1), "unit": "audio-tokens/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": "VALL-E GPT train step (ttts/gpt/config.json model: 6 layers, d512, 8 heads, 21.46 M "
                                      "params), batch 8 per GPU x (128 text + 1024 audio tokens) = S 1156, fwd+bwd+clip+AdamW, "
                                      "dropout %.1f, %s" % (dropout, "eager launches" if not graphed else ("hipGraph replay" if world == 1 else
                                                                       "hipGraph replay, RCCL all-reduce of the upper layers overlapped with the lower layers' backward")),
                          "global_batch": world * B_PER_GPU, "seq_len": TEXT_LEN + 2 + MEL_LEN + 2, "parallelism": "dp%d" % world,
                          "mode": args.mode},
               "final_loss_mel": round(lm, 4), "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
