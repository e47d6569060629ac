"""Run under torchrun (2 ranks; TTTS_SHARE_GPU=1 puts both on cuda:0 over gloo): three captured train steps of a small GPT
with (a) the overlapped ranged gradient exchange and (b) one whole-arena all-reduce must leave bit-identical parameters, and
the replicas must stay identical across ranks."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ttts_amd.gpt import GptEngine, prepare_tokens  # noqa: E402
from ttts_amd.parallel import FlatDataParallel, init_distributed  # noqa: E402

rank, world, local = init_distributed()
dev = torch.device("cuda", local)
torch.cuda.set_device(dev)
dp = FlatDataParallel()
cfg = dict(layers=4, model_dim=128, heads=4, max_text_tokens=40, max_mel_tokens=100, number_text_tokens=256, start_text_token=255,
           number_mel_codes=1026, start_mel_token=1024, stop_mel_token=1025, mel_length_compression=1024)
g = torch.Generator().manual_seed(10 + rank)
text = torch.randint(1, 255, (2, 24), generator=g); mel = torch.randint(0, 1024, (2, 60), generator=g)
tl = torch.full((2,), 24); wl = torch.full((2,), 60 * 1024)


def run(mode):
    eng = GptEngine(cfg, dev, dropout_p=0.1, seed=rank)
    eng.seed_ctr.zero_()                      # the dropout stream counter is process-wide: both runs start from step 0
    torch.manual_seed(0)
    with torch.no_grad():
        for k, shp in eng.spec:
            p = eng.view(eng.params, k)
            p.fill_(1.0 if k.endswith("weight") else 0.0) if len(shp) == 1 else p.normal_(0.0, 0.05)
    dp.broadcast_(eng.params)
    eng.refresh_shadows()
    toks = prepare_tokens(eng.c, text.to(dev), tl, mel.to(dev), wl)
    for _ in range(3):
        if mode == "range":
            eng.train_step(toks, 0.01 / world, 1.0 / world, capture=True, lr=1e-3,
                           exchange_range=lambda lo, hi: dp.allreduce_range_(eng.grads, lo, hi))
        else:
            eng.train_step(toks, 0.01 / world, 1.0 / world, capture=True, lr=1e-3, exchange=lambda: dp.allreduce_grads_(eng.grads))
    torch.cuda.synchronize()
    return eng.params.clone(), eng.losses()


pa, la = run("range")
pb, lb = run("whole")
assert torch.equal(pa, pb), "ranged exchange != whole exchange: max diff %g" % (pa - pb).abs().max().item()
assert la == lb
lo, hi = pa.clone(), pa.clone()
dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
assert torch.equal(lo, hi), "replicas diverged"
split, first, second = GptEngine(cfg, dev).grad_exchange_plan()
cover = sorted(first + second)
assert cover[0][0] == 0 and all(cover[i][1] == cover[i + 1][0] for i in range(len(cover) - 1)), cover
dp.barrier()
print("rank%d-consistent loss_mel %.4f" % (rank, la[1]), flush=True)
