"""Run the GPT forward + backward N times on the same engine / tokens / dropout counter and compare every activation buffer and
the gradient arena bitwise with the first run: which buffer differs first, how often.  (python tools/exp/fwd_determinism.py [N])"""
import sys
import torch
sys.path.insert(0, ".")
from oracle import gpt_ref
from ttts_amd import gpt

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
eng = gpt.GptEngine(gpt_ref.GPT_CONFIG, dev, dropout_p=0.1, seed=5)
eng.load_state_dict(gpt_ref.det_state_dict(None))
toks = gpt.prepare_tokens(eng.c, *gpt_ref.synthetic_batch(B=8, seed=99))
eng.set_tokens(*toks)


def snapshot():
    out = {}
    for k, v in eng.b.items():
        if torch.is_tensor(v):
            out[k] = v.clone()
        elif isinstance(v, (list, tuple)):
            for i, t in enumerate(v):
                if torch.is_tensor(t):
                    out["%s[%d]" % (k, i)] = t.clone()
                elif isinstance(t, (list, tuple)):
                    for j, u in enumerate(t):
                        if torch.is_tensor(u):
                            out["%s[%d][%d]" % (k, i, j)] = u.clone()
    return out


ref = None
bad = {}
for it in range(N):
    eng.seed_ctr.zero_()
    eng.zero_grad()
    eng.forward()
    torch.cuda.synchronize()
    fwd = snapshot()
    eng.backward()
    torch.cuda.synchronize()
    g = eng.grads.clone()
    if ref is None:
        ref, gref = fwd, g
        continue
    for k in ("xs[1]", "xs[2]", "qkv[0]", "att[0]", "fc_act[0]", "fc_pre[0]", "logits_m", "losses"):
        if k in fwd and not torch.equal(fwd[k], ref[k]):
            d = (fwd[k].float() - ref[k].float()).abs()
            bad.setdefault(k, []).append((it, int((d > 0).sum()), float(d.max())))
    names = [k for k in fwd if k in ref and fwd[k].shape == ref[k].shape and not torch.equal(fwd[k], ref[k])]
    if names:
        print("iter", it, "forward buffers that differ:", names[:12])
    if not torch.equal(g, gref):
        d = (g - gref).abs()
        print("iter", it, "grads differ: n", int((d > 0).sum()), "max", float(d.max()), "rel", float(d.norm() / gref.norm()))
print("forward mismatches:", bad if bad else "none")
print("losses", eng.losses())
