"""Decode throughput of the full-config GPT (ttts/gpt/config.json): prompt = 130 text + 1 + DB_PROMPT mel tokens, DB_NEW sampled
tokens, DB_B sequences.  Reports tokens/s, us per step and the HBM floor of one step (bf16 weights once + the KV cache)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ttts_amd.gpt as g  # noqa: E402

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
cfg = json.load(open(os.path.join(ROOT, "ttts_amd", "gpt", "config.json")))["gpt"]
B, NEW, PROMPT = int(os.environ.get("DB_B", 1)), int(os.environ.get("DB_NEW", 256)), int(os.environ.get("DB_PROMPT", 100))
model = g.UnifiedVoice(**cfg, device="cuda:0", dropout_p=0.0)
model.eval()
with torch.no_grad():
    model.mel_head.bias[model.stop_mel_token] = -30.0        # never stop early: time exactly NEW steps
model.engine.refresh_shadows()
model.post_init_gpt2_config()
gen = torch.Generator().manual_seed(0)
text = torch.randint(1, 255, (B, 128), generator=gen).cuda()
prompt = torch.randint(0, 1024, (B, PROMPT), generator=gen).cuda()
kw = dict(do_sample=True, top_p=0.8, temperature=0.8, repetition_penalty=2.0, max_generate_length=NEW)
for rep in range(3):                                         # first call captures the step graph; the last one is reported
    torch.cuda.synchronize(); t0 = time.perf_counter()
    codes = model.inference_speech(text, prompt, seed=rep, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
us_step = dt / NEW * 1e6
c = model.engine.c
D, L = c["model_dim"], c["layers"]
w_bytes = 2 * (L * 12 * D * D + c["number_mel_codes"] * D)
S_avg = 131 + PROMPT + NEW / 2
kv_bytes = B * L * 2 * S_avg * D * 2
out = {"B": B, "new_tokens": NEW, "prompt": PROMPT, "wall_s": round(dt, 4), "us_per_step": round(us_step, 1),
       "tokens_per_s": round(B * NEW / dt, 1), "hbm_floor_us_at_8TBps": round((w_bytes + kv_bytes) / 8e12 * 1e6, 2),
       "weight_MB": round(w_bytes / 1e6, 1), "kv_MB_avg": round(kv_bytes / 1e6, 1), "codes_shape": list(codes.shape)}
print(json.dumps(out))
