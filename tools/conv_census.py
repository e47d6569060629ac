"""Where the VQ-VAE-GAN step's convolution time goes: records every conv-family call of one training step (config #3,
B x 163 840 samples) by signature, then times each distinct signature alone and prints count x time, sorted.
usage: python tools/conv_census.py [B] [top]     -> table on stdout + gpurun_out/conv_census.json"""
import json
import os
import sys
from collections import OrderedDict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as ge

ge.build()
from ttts_amd import ops
from ttts_amd.vqvae.train import SyntheticVqvaeBatches, VqvaeTrainer, get_hparams

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
TOP = int(sys.argv[2]) if len(sys.argv) > 2 else 60
hps = get_hparams()
tr = VqvaeTrainer(hps)
cb = tr.net_g.quantizer.vq.layers[0]._codebook
with torch.no_grad():
    cb.inited.fill_(1); cb.embed.normal_(0, 0.3); cb.embed_avg.copy_(cb.embed * 4); cb.cluster_size.fill_(4.0)
data = next(iter(SyntheticVqvaeBatches(B, device=tr.device)))
tr.train_step(data)
torch.cuda.synchronize()

calls = OrderedDict()      # signature -> [count, replay closure factory]
orig = {n: getattr(ops, n) for n in ("conv1d_fwd", "conv1d_dgrad", "conv1d_wgrad", "conv1d_bias_grad")}


def shp(t):
    return None if t is None else tuple(t.shape)


def rec(kind, sig, make):
    key = (kind,) + sig
    if key not in calls:
        calls[key] = [0, make]
    calls[key][0] += 1


def fwd(x, w, bias=None, resid=None, stride=1, pad=0, dil=1, in_slope=1.0, out_act=None, out_scale=1.0, out=None,
        accumulate=False, bbias=None, gate=None, gate_slope=1.0, groups=1, out_slope=1.0, omask=None):
    sig = (shp(x), shp(w), stride, pad, dil, groups, bias is not None, resid is not None, gate is not None, omask is not None,
           in_slope != 1.0, out_act)
    xs, ws = x.shape, w.shape

    def make():
        xx = torch.randn(xs, device=x.device); ww = torch.randn(ws, device=x.device) * 0.05
        g = torch.randn_like(orig["conv1d_fwd"](xx, ww, None, None, stride, pad, dil, groups=groups)) if gate is not None else None
        return lambda: orig["conv1d_fwd"](xx, ww, None, None, stride, pad, dil, in_slope, out_act, groups=groups, gate=g,
                                          gate_slope=gate_slope)
    rec("fwd", sig, make)
    return orig["conv1d_fwd"](x, w, bias, resid, stride, pad, dil, in_slope, out_act, out_scale, out, accumulate, bbias, gate,
                              gate_slope, groups, out_slope, omask)


def dgrad(dy, w, lin, stride=1, pad=0, dil=1, gate=None, gate_slope=1.0, bias=None, in_slope=1.0, resid=None, out_scale=1.0,
          out=None, accumulate=False, groups=1, omask=None):
    sig = (shp(dy), shp(w), lin, stride, pad, dil, groups, gate is not None, bias is not None, resid is not None, in_slope != 1.0)
    ds, ws = dy.shape, w.shape

    def make():
        dd = torch.randn(ds, device=dy.device); ww = torch.randn(ws, device=dy.device) * 0.05
        g = torch.randn(ds[0], ws[1] * groups, lin, device=dy.device) if gate is not None else None
        return lambda: orig["conv1d_dgrad"](dd, ww, lin, stride, pad, dil, gate=g, gate_slope=gate_slope, in_slope=in_slope,
                                            groups=groups)
    rec("dgrad", sig, make)
    return orig["conv1d_dgrad"](dy, w, lin, stride, pad, dil, gate, gate_slope, bias, in_slope, resid, out_scale, out, accumulate,
                                groups, omask)


def wgrad(dy, x, k, stride=1, pad=0, dil=1, x_slope=1.0, dy_slope=1.0, out=None, groups=1, db=None):
    sig = (shp(dy), shp(x), k, stride, pad, dil, groups, x_slope != 1.0, dy_slope != 1.0)
    ds, xs = dy.shape, x.shape

    def make():
        dd = torch.randn(ds, device=dy.device); xx = torch.randn(xs, device=dy.device)
        o = torch.zeros(ds[1], xs[1] // groups, k, device=dy.device)
        return lambda: orig["conv1d_wgrad"](dd, xx, k, stride, pad, dil, x_slope, dy_slope, out=o, groups=groups)
    rec("wgrad", sig, make)
    return orig["conv1d_wgrad"](dy, x, k, stride, pad, dil, x_slope, dy_slope, out, groups, db=db)


def bgrad(dy, out=None):
    ds = dy.shape

    def make():
        dd = torch.randn(ds, device=dy.device); o = torch.zeros(ds[1], device=dy.device)
        return lambda: orig["conv1d_bias_grad"](dd, out=o)
    rec("bias", (shp(dy),), make)
    return orig["conv1d_bias_grad"](dy, out)


ops.conv1d_fwd, ops.conv1d_dgrad, ops.conv1d_wgrad, ops.conv1d_bias_grad = fwd, dgrad, wgrad, bgrad
tr.train_step(data)
torch.cuda.synchronize()
for n, f in orig.items():
    setattr(ops, n, f)
del tr, data
torch.cuda.empty_cache()


def timeit(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3   # us


def flops_bytes(key):
    kind = key[0]
    if kind == "fwd":
        (b, cin, lin), (cout, cpg, k) = key[1], key[2]
        stride, pad, dil = key[3], key[4], key[5]
        lout = ops.conv_out_len(lin, k, stride, pad, dil)
        return 2.0 * b * cout * cpg * k * lout, 4.0 * (b * cin * lin + b * cout * lout)
    if kind == "dgrad":
        (b, cout, lout), (_, cpg, k), lin, groups = key[1], key[2], key[3], key[7]
        return 2.0 * b * cout * cpg * k * lout, 4.0 * (b * cpg * groups * lin + b * cout * lout)
    if kind == "wgrad":
        (b, cout, lout), (_, cin, lin), k, groups = key[1], key[2], key[3], key[7]
        return 2.0 * b * cout * (cin // groups) * k * lout, 4.0 * (b * cin * lin + b * cout * lout)
    (b, c, l) = key[1]
    return 1.0 * b * c * l, 4.0 * b * c * l


rows = []
for key, (count, make) in calls.items():
    fn = make()
    us = timeit(fn)
    del fn
    fl, by = flops_bytes(key)
    rows.append({"sig": repr(key), "kind": key[0], "count": count, "us": us, "total_ms": count * us / 1e3, "tf": fl / us / 1e6,
                 "hbm_floor_us": by / 8e6, "gflop": fl / 1e9})
    torch.cuda.empty_cache()
rows.sort(key=lambda r: -r["total_ms"])
tot = sum(r["total_ms"] for r in rows)
by_kind = {}
for r in rows:
    by_kind[r["kind"]] = by_kind.get(r["kind"], 0.0) + r["total_ms"]
print("distinct signatures %d, calls %d, summed time %.1f ms; by kind: %s" % (
    len(rows), sum(r["count"] for r in rows), tot, {k: round(v, 1) for k, v in by_kind.items()}))
for r in rows[:TOP]:
    print("%6.2f ms = %3d x %8.1f us | %6.1f TF/s | hbm floor %7.1f us | %s" % (r["total_ms"], r["count"], r["us"], r["tf"],
                                                                              r["hbm_floor_us"], r["sig"]))
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"batch": B, "total_ms": tot, "by_kind": by_kind, "rows": rows}, open("gpurun_out/conv_census.json", "w"), indent=1)
