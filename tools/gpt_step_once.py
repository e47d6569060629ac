"""A few eager forward+backward passes of the GPT engine at the BASELINE shape (B 8 x (128 text + 1024 audio tokens), the
ttts/gpt/config.json model): the workload `tools/gpt_pmc.sh` wraps in rocprofv3 --pmc passes (counters per kernel, e.g. of
the grouped weight-gradient GEMM).  GPT_PASSES = number of passes (default 2)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import __graft_entry__ as ge

ge.build()
from ttts_amd.gpt import GptEngine, prepare_tokens

dev = torch.device("cuda:0")
cfg = json.load(open(os.path.join(ROOT, "ttts_amd", "gpt", "config.json")))
eng = GptEngine(cfg["gpt"], dev, dropout_p=0.1, seed=0)
torch.manual_seed(0)
with torch.no_grad():
    for k, shp in eng.spec:
        p = eng.view(eng.params, k)
        if len(shp) == 1:
            p.fill_(1.0 if k.endswith("weight") else 0.0)
        else:
            p.normal_(0.0, 0.02)
eng.refresh_shadows()
g = torch.Generator().manual_seed(0)
B, Tt, Tm = 8, 128, 1024
text = torch.randint(1, 255, (B, Tt), generator=g).to(dev)
mel = torch.randint(0, 1024, (B, Tm), generator=g).to(dev)
toks = prepare_tokens(eng.c, text, torch.full((B,), Tt), mel, torch.full((B,), Tm * 1024))
eng.set_tokens(*toks)
for _ in range(int(os.environ.get("GPT_PASSES", "2"))):
    eng.zero_grad()
    eng.forward()
    eng.backward()
torch.cuda.synchronize()
print("losses", eng.losses())
