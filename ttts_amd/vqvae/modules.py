"""Convolutional building blocks of the VQ-VAE-GAN path on the HIP conv family (csrc/conv.hip).

Mirrors, with the reference's constructor arguments and state-dict keys:
  * `Conv1d` / `ConvTranspose1d` (+ both weight-norm styles: `torch.nn.utils.weight_norm` -> `weight_g`/`weight_v`
    (ttts/vqvae/vq2.py:10,364) and `torch.nn.utils.parametrizations.weight_norm` -> `parametrizations.weight.original0/1`
    (ttts/vqvae/modules.py:8))
  * `ResBlock1`   ttts/vqvae/modules.py:224-318
  * `Generator`   ttts/vqvae/vq2.py:341-415 (HiFi-GAN decoder)

Every convolution, its leaky-relu prologue, bias, residual add and tanh epilogue, the weight-norm reparametrisation and
all of their gradients run in `libttts_hip.so`; torch provides autograd bookkeeping and storage only.  There is no CPU
path: the ops raise `TttsError` off-GPU.
"""
import math
import os

import torch
from torch import nn

from .. import ops

LRELU_SLOPE = 0.1   # ttts/vqvae/modules.py:16


def get_padding(kernel_size, dilation=1):
    """ttts/utils/commons.py:12-13."""
    return int((kernel_size * dilation - dilation) / 2)


# ---- autograd glue ------------------------------------------------------------------------------------------------------
def _take_dw_buffer(w):
    """The pre-cleared weight-gradient buffer _WeightNormFn.forward attached to its output (once), else None."""
    z = getattr(w, "_ttts_dw_zero", None)
    if z is not None:
        w._ttts_dw_zero = None
    return z


_branch_pools = {}
_dirty_streams = []
_fork_depth = [0]            # > 0 while run_branches is issuing a branch on a side stream (host-side nesting level)
_env_cache = {}


def _env_int(name, default):
    """int(os.environ[name]) for switches read on hot paths: os.environ.get costs ~1.5 us (key encode, lookup, value decode); the raw
    value is compared with the one parsed last time instead of being parsed again."""
    try:
        raw = os.environ._data.get(_env_keys.get(name) or _env_keys.setdefault(name, os.environ.encodekey(name)))
    except AttributeError:                               # (an os.environ without CPython's internals: the plain, slower way)
        v = os.environ.get(name)
        return default if v in (None, "") else int(v)
    hit = _env_cache.get(name)
    if hit is not None and hit[0] is raw:
        return hit[1]
    val = default if raw is None else int(os.environ.decodevalue(raw) or default)
    _env_cache[name] = (raw, val)
    return val


_env_keys = {}


def _eager_pools():
    """Fan-out levels in use outside a capture (TTTS_EAGER_POOLS, default all: disc, synth, mrf, wgrad)."""
    raw = os.environ.get("TTTS_EAGER_POOLS")
    return ("disc", "synth", "mrf", "wgrad") if raw is None else raw.split(",")


def side_streams(pool, device, n=None):
    """The named set of side streams of `device` (created on first use); [] when disabled (TTTS_BRANCH_STREAMS=0) or not a GPU."""
    n = int(os.environ.get("TTTS_BRANCH_STREAMS", "3")) if n is None else n
    if n <= 0 or device.type != "cuda":
        return []
    if torch.cuda.is_current_stream_capturing():
        # hipStreamEndCapture crashes (SIGSEGV, ROCm 7.2) on a capture that holds NESTED fan-outs (a fork made on a forked stream:
        # synth -> mrf; tools/exp/capture_debug.sh).  The step's fan-outs are siblings since round 6 (run_branches(inline=...): the
        # branch that fans out again stays on the caller's stream), so every level CAN be recorded; a branch that does sit on a side
        # stream (_fork_depth > 0: the prior path) runs its own inner fan-out sequentially while recording.  Which levels pay under
        # replay was measured (HISTORY 19.8, tools/gpu_r6_n.sh; B = 32): none 136.4 ms, disc 136.8, synth 120.6, disc + synth 114.5,
        # disc + mrf 127.6, synth + mrf 128.4, all three 122.4 -- the three-way ResBlock fan-out that helps eager launches costs a
        # replay 8 ms (every fork / join is a cross-stream dependency the graph executor resolves with barrier packets)
        if _fork_depth[0] > 0 or pool not in os.environ.get("TTTS_CAPTURE_POOLS", "disc,synth").split(","):
            return []
    elif pool not in _eager_pools():
        return []
    key = (pool, device.index if device.index is not None else torch.cuda.current_device(), n)
    streams = _branch_pools.get(key)
    if streams is None:
        streams = _branch_pools[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
    return streams


def fork_to(stream, main):
    """`stream` continues from everything queued on `main` so far; remembered for join_side_streams()."""
    stream.wait_stream(main)
    if stream not in _dirty_streams:
        _dirty_streams.append(stream)


def join_side_streams(device):
    """The caller's stream waits for every side stream used since the last call.  Needed after a BACKWARD pass: autograd replays
    each node on its forward stream and only synchronises along gradient edges and with the streams of AccumulateGrad nodes --
    a node that writes its parameter gradients straight into the optimizer's arena (`_grad_slot`) and whose input needs no
    gradient (a sub-discriminator's first layer in the discriminator phase) hands nothing to anybody, so nothing would ever
    wait for its weight-gradient kernels (and a stream capture would end with an un-joined fork)."""
    if device.type != "cuda":
        return
    main = torch.cuda.current_stream(device)
    while _dirty_streams:
        main.wait_stream(_dirty_streams.pop())


_join_pending = {}          # device -> {stream handle: stream}: side streams the running backward pass has touched


def _join_pending_streams():
    """Engine callback (end of a backward pass, on the thread and stream backward() was called from)."""
    for device, streams in list(_join_pending.items()):
        if streams:
            main = torch.cuda.current_stream(device)
            for st in streams.values():
                main.wait_stream(st)
            streams.clear()


class _JoinAfterBackward(torch.autograd.Function):
    """Identity on ONE output tensor of a fan-out branch.  Its backward node sits downstream of the branch, so it runs before the
    branch's nodes; it notes the branch's stream and queues an engine callback: when the whole backward pass is over, the stream
    backward() was called on waits for every side stream noted.  This makes the join part of every backward through run_branches /
    MultiPeriodDiscriminator -- user loops, tests, torch.autograd.grad -- instead of a rule the caller has to know (VqvaeStep still
    calls join_side_streams() itself: waiting twice is free).  One node PER TENSOR, deliberately: a single node over all outputs
    of a fan-out is a barrier -- it runs only when every one of its gradients has arrived, so no branch's backward could start
    before the slowest loss term was differentiated (measured: + 2.7 ms on the 132 ms VQ-VAE-GAN step, HISTORY 18.3)."""

    @staticmethod
    def forward(ctx, holder, x):
        ctx.holder = holder
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        device, streams = ctx.holder
        pend = _join_pending.setdefault(device, {})
        for st in streams:
            pend[st.cuda_stream] = st
        # (queued every time: a backward pass that raised never ran its callbacks, so "already queued" cannot be remembered)
        torch.autograd.Variable._execution_engine.queue_callback(_join_pending_streams)
        return None, g


def join_after_backward(outs, device, streams):
    """`outs` (tensor or nested tuples / lists of tensors) with every gradient-carrying tensor routed through its own
    _JoinAfterBackward node."""
    if not streams or not torch.is_grad_enabled() or _env_int("TTTS_AUTOJOIN", 1) != 1:
        return outs
    holder = (device, list(streams))

    def rebuild(o):
        if torch.is_tensor(o):
            return _JoinAfterBackward.apply(holder, o) if o.requires_grad else o
        if isinstance(o, (tuple, list)):
            return type(o)(rebuild(e) for e in o)
        return o
    return rebuild(outs)


_wgrad_rr = [0]


def wgrad_side_stream(device):
    """A side stream (forked from the caller's stream, remembered for join_side_streams) for a weight-gradient launch of a
    backward node, or None: TTTS_WGRAD_STREAMS streams, taken in turn.  Default 0 (off): measured 146.6 ms against 136-139 ms per step with 2 streams
    (profiles/r04_ab_wgrad_streams.txt) -- with the branch streams already filling the chip the extra fork per convolution costs more
    than the overlap returns."""
    n = _env_int("TTTS_WGRAD_STREAMS", 0)          # (once per convolution backward: ~1 100 times a step)
    if n <= 0:
        return None
    streams = side_streams("wgrad", device, n)
    if not streams:
        return None
    _wgrad_rr[0] = (_wgrad_rr[0] + 1) % len(streams)
    st = streams[_wgrad_rr[0]]
    fork_to(st, torch.cuda.current_stream(device))
    return st


def run_branches(fns, device, pool="mrf", inline=None):
    """Independent sub-graphs (callables) on side streams, results in order.  Most convolution launches of the VQ-VAE-GAN step
    either under-fill the 256 CUs or end in a nearly empty last round of workgroups; issued on separate streams, one branch's
    tail overlaps another branch's next launch (and autograd replays every node on its forward stream, so the backward overlaps
    the same way -- call join_side_streams() after it).  Every side stream first waits for the caller's stream and the caller's
    stream waits for all of them before returning, so the call is ordered like a plain sequential one.  `pool` names a set of
    TTTS_BRANCH_STREAMS (default 3; 0: run sequentially on the caller's stream) streams; nested fan-outs use different pools.
    `inline` = index of a branch that runs on the CALLER's stream (after the earlier branches were issued to their side streams):
    a branch that fans out again then forks from the caller's stream, i.e. the fan-outs are siblings instead of nested -- which is
    what lets a stream capture keep both levels (hipStreamEndCapture crashes on a fork made from a forked stream, ROCm 7.2)."""
    streams = side_streams(pool, device) if len(fns) >= 2 else []
    if not streams:
        return [f() for f in fns]
    main = torch.cuda.current_stream(device)
    outs, used = [], []
    for i, f in enumerate(fns):
        if i == inline:                  # this branch stays on the caller's stream: whatever it forks is forked from there
            outs.append(f())
            continue
        st = streams[(i if inline is None or i < inline else i - 1) % len(streams)]
        fork_to(st, main)
        _fork_depth[0] += 1
        try:
            with torch.cuda.stream(st):
                outs.append(f())
        finally:
            _fork_depth[0] -= 1
        if st not in used:
            used.append(st)
    for st in used:
        main.wait_stream(st)
    return join_after_backward(outs, device, used)


def _grad_slot(p):
    """A leaf parameter whose `.grad` already exists (FlatAdamW gives every parameter a persistent view of the flat
    gradient arena): the backward kernels, which all ACCUMULATE, then write straight into it and the Function returns
    None for that input -- no zero-filled temporary and no autograd `grad += temp` launch per parameter (the two were
    ~3000 of the ~14000 launches of a VQ-VAE-GAN step).  Returns None when the ordinary autograd route must be used."""
    if p is not None and p.is_leaf and p.requires_grad and p.grad is not None and p.grad.is_contiguous():
        return p.grad
    return None


class _WeightNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, g):
        # The same launch clears the buffer this layer's weight-gradient kernels will accumulate into (they add; w is not a leaf,
        # so there is no .grad slot for it): the first convolution backward that uses `w` takes it from the tensor
        # (_take_dw_buffer), later uses of the same w fall back to a fresh zero-filled buffer.
        want = any(ctx.needs_input_grad[:2])          # (grad mode is off inside a Function's forward: ask the ctx)
        if want:
            w, norm, zero = ops.weight_norm_fwd(v, g, want_zero=True)
            w._ttts_dw_zero = zero
        else:
            w, norm = ops.weight_norm_fwd(v, g)
        ctx.save_for_backward(v, g, norm)
        ctx.refs = (v, g)
        return w

    @staticmethod
    def backward(ctx, dw):
        v, g, norm = ctx.saved_tensors
        sv, sg = _grad_slot(ctx.refs[0]), _grad_slot(ctx.refs[1])
        if sv is not None and sg is not None:
            ops.weight_norm_bwd(dw, v, g, norm, dv=sv, dg=sg)
            return None, None
        dv, dg = ops.weight_norm_bwd(dw, v, g, norm)
        return dv, dg


class WeightNormBank:
    """All weight-normed layers of one network (generator or discriminators) normalised by ONE launch per phase, and their
    (dv, dg) formed from the accumulated effective-weight gradients by ONE launch at the end of the phase's backward
    (ops.WeightNormPlan -> ttts_weight_norm_{fwd,bwd}_batched_f32) -- instead of one weight_norm_fwd per convolution call and one
    weight_norm_bwd per layer (~920 launches a step at the 5 us launch floor).  Per layer the bank owns a LEAF tensor `w` (a view
    of one flat buffer) whose .grad is a view of a flat dW buffer: the convolution backward kernels, which all accumulate, add
    straight into it through the existing `_grad_slot` route.  `refresh()` rewrites every w from (v, g) and clears dW;
    `finish()` turns dW into accumulations on v.grad / g.grad (views of the optimizer's flat gradient arena).  Outside a step
    (`active` False) the layers recompute their weight per call as before."""

    def __init__(self, net):
        self.layers = [m for m in net.modules() if isinstance(m, _ConvBase) and m._wn is not None]
        self.active = False
        if not self.layers:
            return
        dev = self.layers[0]._v().device
        sizes = [m._v().numel() for m in self.layers]
        rows = [m._v().shape[0] for m in self.layers]
        self.flat_w = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        self.flat_dw = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        self.flat_norm = torch.zeros(sum(rows), dtype=torch.float32, device=dev)
        self.w, entries_f, entries_b = [], [], []
        o = r = 0
        for i, m in enumerate(self.layers):
            v, g = m._v(), m._g()
            shape = m._eff_shape(v.shape)
            w = self.flat_w[o:o + sizes[i]].view(shape)
            w.requires_grad_(v.requires_grad)
            dw = self.flat_dw[o:o + sizes[i]].view(shape)
            if v.requires_grad:
                w.grad = dw
            norm = self.flat_norm[r:r + rows[i]]
            self.w.append(w)
            e = {"v": v.detach(), "g": g.detach().reshape(-1), "w": w.detach(), "norm": norm, "dw": dw if v.requires_grad else None}
            entries_f.append(e)
            if v.requires_grad:
                if v.grad is None or g.grad is None:
                    raise ops.TttsError("WeightNormBank needs the parameters' persistent .grad views (build the FlatAdamW first)")
                entries_b.append(dict(e, dv=v.grad, dg=g.grad.reshape(-1)))
            object.__setattr__(m, "_bank_ref", (self, i))
            o += sizes[i]; r += rows[i]
        self.plan_f = ops.WeightNormPlan(entries_f, dev)
        self.plan_b = ops.WeightNormPlan(entries_b, dev) if entries_b else None
        self._trainable = [w.requires_grad for w in self.w]

    def weight(self, i):
        return self.w[i]

    def refresh(self, requires_grad=True):
        """Recompute every effective weight (and clear the dW buffer).  requires_grad False: a phase in which this network's
        parameters are frozen (the generator phase's discriminator): no weight gradients are formed."""
        if not self.layers:
            return
        self.plan_f.forward()
        for w, t in zip(self.w, self._trainable):
            w.requires_grad_(bool(requires_grad and t))
        self.active = True

    def finish(self):
        """dW -> (dv, dg), accumulated into the parameters' gradient slots."""
        if self.layers and self.plan_b is not None:
            self.plan_b.backward()

    def release(self):
        self.active = False


class _Conv1dFn(torch.autograd.Function):
    """y = act(bias + bbias + conv1d(lrelu(x, in_slope), w) + resid);  out_act None | 'tanh' | 'lrelu'."""

    @staticmethod
    def forward(ctx, x, w, bias, resid, bbias, stride, pad, dil, in_slope, out_act, groups=1, out_slope=LRELU_SLOPE,
                omask=None):
        if omask is not None:
            omask = omask.reshape(x.shape[0], -1).contiguous()
        y = ops.conv1d_fwd(x, w, bias, resid, stride, pad, dil, in_slope, out_act, bbias=bbias, groups=groups,
                           out_slope=out_slope, omask=omask)
        ctx.save_for_backward(x, w, y if out_act else None, omask)
        ctx.cfg = (stride, pad, dil, in_slope, out_act, groups, out_slope, bias is not None, resid is not None,
                   bbias is not None)
        ctx.refs = (w, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y, omask = ctx.saved_tensors
        stride, pad, dil, in_slope, out_act, groups, out_slope, has_b, has_r, has_bb = ctx.cfg
        dy = dy.contiguous()
        if omask is not None:
            dy = ops.mul_mask(dy, omask)
        if out_act == "tanh":
            dy = ops.tanh_bwd(dy, y)
        elif out_act == "lrelu":
            dy = ops.lrelu_bwd(dy, y, out_slope)
        need = ctx.needs_input_grad
        dx = dw = db = dbb = None
        want_b = has_b and need[2]
        bslot = _grad_slot(ctx.refs[1]) if want_b else None
        # The weight gradient only feeds the optimizer: when it lands in persistent slots (nothing returns to autograd) it goes to
        # a side stream and overlaps the data-gradient chain, whose launches leave CUs idle (wgrad_side_stream).
        if need[1] and need[0] and _grad_slot(ctx.refs[0]) is not None and (not want_b or bslot is not None):
            ws = wgrad_side_stream(dy.device)
            if ws is not None:
                with torch.cuda.stream(ws):
                    ops.conv1d_wgrad(dy, x, w.shape[2], stride, pad, dil, x_slope=in_slope, groups=groups,
                                     out=_grad_slot(ctx.refs[0]), db=bslot if want_b else None)
                dy.record_stream(ws); x.record_stream(ws)
                gate = x if in_slope != 1.0 else None
                dx = ops.conv1d_dgrad(dy, w, x.shape[2], stride, pad, dil, gate=gate, gate_slope=in_slope, groups=groups)
                if has_bb and need[4]:
                    B, C, L = dy.shape
                    dbb = ops.conv1d_bias_grad(dy.view(1, B * C, L)).view(B, C)
                return dx, None, None, (dy if has_r and need[3] else None), dbb, None, None, None, None, None, None, None, None
        if need[0]:
            gate = x if in_slope != 1.0 else None
            dx = ops.conv1d_dgrad(dy, w, x.shape[2], stride, pad, dil, gate=gate, gate_slope=in_slope, groups=groups)
        if need[1]:
            slot = _grad_slot(ctx.refs[0])
            if slot is None:
                slot_w = _take_dw_buffer(ctx.refs[0])      # weight-normed: the buffer its forward cleared (None: a fresh one)
            if want_b:      # the bias gradient rides on the weight-gradient call (its kernels stream dy anyway)
                db = bslot if bslot is not None else torch.zeros(dy.shape[1], dtype=torch.float32, device=dy.device)
            dw = ops.conv1d_wgrad(dy, x, w.shape[2], stride, pad, dil, x_slope=in_slope, groups=groups,
                                  out=slot if slot is not None else slot_w, db=db if want_b else None)
            if slot is not None:
                dw = None
        elif want_b:
            db = ops.conv1d_bias_grad(dy, out=bslot)
        if bslot is not None:
            db = None
        if has_bb and need[4]:
            B, C, L = dy.shape
            dbb = ops.conv1d_bias_grad(dy.view(1, B * C, L)).view(B, C)
        return dx, dw, db, (dy if has_r and need[3] else None), dbb, None, None, None, None, None, None, None, None


class _ConvTranspose1dFn(torch.autograd.Function):
    """y = bias + conv_transpose1d(lrelu(x, in_slope), w);  w (Cin, Cout, K)."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, in_slope):
        lout = (x.shape[2] - 1) * stride - 2 * pad + w.shape[2]
        y = ops.conv1d_dgrad(x, w, lout, stride, pad, 1, bias=bias, in_slope=in_slope)
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, in_slope, bias is not None)
        ctx.refs = (w, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad, in_slope, has_b = ctx.cfg
        dy = dy.contiguous()
        need = ctx.needs_input_grad
        dx = dw = db = None
        if need[0]:
            dx = ops.conv1d_fwd(dy, w, stride=stride, pad=pad, gate=x if in_slope != 1.0 else None, gate_slope=in_slope)
        if need[1]:
            slot = _grad_slot(ctx.refs[0])
            dw = ops.conv1d_wgrad(x, dy, w.shape[2], stride, pad, 1, dy_slope=in_slope,
                                  out=slot if slot is not None else _take_dw_buffer(ctx.refs[0]))
            if slot is not None:
                dw = None
        if has_b and need[2]:
            slot = _grad_slot(ctx.refs[1])
            db = ops.conv1d_bias_grad(dy, out=slot)
            if slot is not None:
                db = None
        return dx, dw, db, None, None, None


_alive = []


def _keep_until_backward_ends(t):
    """Hold a reference to a gradient that ONE backward node hands to several consumers until the backward pass is over.  The
    summands of an add may have been computed on different streams (run_branches); autograd then replays their backward nodes
    on those streams, all reading this tensor.  Without the extra reference the last consumer to be dispatched sees a uniquely
    owned tensor and the engine accumulates the next gradient INTO it in place -- while the other streams' reads are still
    queued (seen as deterministic 10-40 % gradient errors in PosteriorAudioEncoder).  With it the engine adds out of place."""
    torch.autograd.Variable._execution_engine.queue_callback(_alive.clear)   # (every time: a backward that raised never ran its callbacks)
    _alive.append(t)


class _AddScaleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scale, *xs):
        ctx.scale, ctx.n = scale, len(xs)
        return ops.add_scale(xs, scale)

    @staticmethod
    def backward(ctx, dy):
        g = ops.add_scale([dy], ctx.scale) if ctx.scale != 1.0 else dy
        _keep_until_backward_ends(g)
        return (None,) + (g,) * ctx.n


def add_scale(xs, scale=1.0):
    return _AddScaleFn.apply(scale, *xs)


# ---- modules ------------------------------------------------------------------------------------------------------------
class _Originals(nn.Module):
    """Holder giving the `parametrizations.weight.original0/original1` keys of the new-style weight norm."""

    def __init__(self, g, v):
        super().__init__()
        self.original0 = nn.Parameter(g)
        self.original1 = nn.Parameter(v)


class _ConvBase(nn.Module):
    def _init_params(self, wshape, fan_in, bias_ch, bias):
        w = torch.empty(wshape)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))           # torch's Conv default (reset_parameters)
        self.weight = nn.Parameter(w)
        if bias:
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            self.bias = nn.Parameter(torch.empty(bias_ch).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)
        self._wn = None

    def apply_weight_norm(self, style):
        """style 'old': torch.nn.utils.weight_norm (weight_g, weight_v); 'new': parametrizations.weight_norm."""
        w = self.weight.detach()
        g = w.flatten(1).norm(dim=1).view(-1, *([1] * (w.dim() - 1))).clone()
        del self.weight
        if style == "old":
            self.weight_g = nn.Parameter(g)
            self.weight_v = nn.Parameter(w.clone())
        else:
            self.parametrizations = nn.ModuleDict({"weight": _Originals(g, w.clone())})
        self._wn = style
        return self

    def _g(self):
        return self.weight_g if self._wn == "old" else self.parametrizations["weight"].original0

    def _v(self):
        return self.weight_v if self._wn == "old" else self.parametrizations["weight"].original1

    def _banked_weight(self):
        """The WeightNormBank leaf holding this layer's effective weight while a step is running, else None."""
        bank = getattr(self, "_bank_ref", None)
        if self._wn is not None and bank is not None and bank[0].active:
            return bank[0].weight(bank[1])
        return None

    def effective_weight(self):
        if self._wn is None:
            return self.weight
        w = self._banked_weight()       # the network's weights were all normalised by ONE launch at the phase start
        if w is not None:
            return w
        return _WeightNormFn.apply(self._v(), self._g())

    def _eff_shape(self, shape):
        """Shape the convolution consumes the effective weight in (Conv2dK1 drops its trailing unit axis)."""
        return tuple(shape)


class Conv1d(_ConvBase):
    """nn.Conv1d(in, out, k, stride, padding, dilation, bias) (groups = 1) on the HIP kernels.  `forward` takes the fused
    neighbours: `in_slope` (leaky-relu on the input), `resid` (added to the output), `bbias` (per-sample bias (B, C)),
    `out_act` ('tanh' | 'lrelu' on the output), `omask` (sequence mask (B, 1, T) multiplied last)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.stride, self.padding, self.dilation, self.groups = stride, padding, dilation, groups
        self._init_params((out_channels, in_channels // groups, kernel_size), in_channels // groups * kernel_size,
                          out_channels, bias)

    def forward(self, x, in_slope=1.0, resid=None, bbias=None, out_act=None, out_slope=LRELU_SLOPE, omask=None):
        return _Conv1dFn.apply(x, self.effective_weight(), self.bias, resid, bbias, self.stride, self.padding, self.dilation,
                               float(in_slope), out_act, self.groups, float(out_slope), omask)


class Conv2dK1(_ConvBase):
    """nn.Conv2d(in, out, (k, 1), (stride, 1), padding=(pad, 0)) -- the only Conv2d shape on the path (DiscriminatorP,
    vq2.py:425-472).  Parameters keep the 4-D reference shapes ((out, in, k, 1), weight_g (out, 1, 1, 1)); the module
    computes on (B*W, C, H) tensors, i.e. as a 1-D convolution along H with the period axis folded into the batch."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.stride, self.padding = stride, padding
        self._init_params((out_channels, in_channels, kernel_size, 1), in_channels * kernel_size, out_channels, bias)

    def _eff_shape(self, shape):
        return tuple(shape[:3])

    def forward(self, x, in_slope=1.0, out_act=None, out_slope=LRELU_SLOPE):
        w = self.effective_weight()
        return _Conv1dFn.apply(x, w.squeeze(-1) if w.dim() == 4 else w, self.bias, None, None, self.stride, self.padding, 1,
                               float(in_slope), out_act, 1, float(out_slope))


class ConvTranspose1d(_ConvBase):
    """nn.ConvTranspose1d(in, out, k, stride, padding) (output_padding 0, dilation 1, groups 1)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.stride, self.padding = stride, padding
        self._init_params((in_channels, out_channels, kernel_size), out_channels * kernel_size, out_channels, bias)

    def forward(self, x, in_slope=1.0):
        return _ConvTranspose1dFn.apply(x, self.effective_weight(), self.bias, self.stride, self.padding, float(in_slope))


def _wn_conv_new(channels, k, d):
    # the reference's `convs.apply(init_weights)` runs AFTER weight_norm and writes `.weight.data.normal_` into the
    # recomputed temporary, i.e. it changes no parameter: the effective init is torch's Conv default with g = ||v||
    return Conv1d(channels, channels, k, 1, dilation=d, padding=get_padding(k, d)).apply_weight_norm("new")


def _wgrad_into_slots(dy, x, w_ref, b_ref, k, pad, dil, x_slope, need_w, need_b):
    """Weight (+ bias) gradient of one stride-1 convolution the way _Conv1dFn.backward routes it: straight into the parameters'
    persistent .grad slots where they exist (then None is returned for autograd), else into the weight-norm buffer / a fresh tensor."""
    dw = db = None
    want_b = b_ref is not None and need_b
    bslot = _grad_slot(b_ref) if want_b else None
    if need_w:
        slot = _grad_slot(w_ref)
        slot_w = None if slot is not None else _take_dw_buffer(w_ref)
        if want_b:
            db = bslot if bslot is not None else torch.zeros(dy.shape[1], dtype=torch.float32, device=dy.device)
        dw = ops.conv1d_wgrad(dy, x, k, 1, pad, dil, x_slope=x_slope, out=slot if slot is not None else slot_w, db=db if want_b else None)
        if slot is not None:
            dw = None
    elif want_b:
        db = ops.conv1d_bias_grad(dy, out=bslot)
    if bslot is not None:
        db = None
    return dw, db


class _ResPairFn(torch.autograd.Function):
    """y = x + c2(lrelu(c1(lrelu(x)))) -- one (dilated, plain) pair of ResBlock1 -- as ONE autograd node.  As two nodes, x has two
    consumers (c1's input and c2's fused residual), so the engine finished every pair's backward with `grad_x = dgrad(c1) + dy`: an
    elementwise launch over the activation (135 of them per VQ-VAE-GAN step, ~11 GB of traffic at config #3).  Here dy rides into
    c1's data gradient as its `resid` epilogue operand: the same two fp32 numbers are added, in the kernel -- bit-identical."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, pad1, dil1, pad2, slope):
        x = x.contiguous()
        xt = ops.conv1d_fwd(x, w1, b1, None, 1, pad1, dil1, slope)
        y = ops.conv1d_fwd(xt, w2, b2, x, 1, pad2, 1, slope)
        ctx.save_for_backward(x, xt, w1, w2)
        ctx.cfg = (pad1, dil1, pad2, slope)
        ctx.refs = (w1, b1, w2, b2)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, xt, w1, w2 = ctx.saved_tensors
        pad1, dil1, pad2, slope = ctx.cfg
        r1, rb1, r2, rb2 = ctx.refs
        need = ctx.needs_input_grad
        dy = dy.contiguous()
        gate_t = xt if slope != 1.0 else None
        d_xt = ops.conv1d_dgrad(dy, w2, xt.shape[2], 1, pad2, 1, gate=gate_t, gate_slope=slope)
        dw2, db2 = _wgrad_into_slots(dy, xt, r2, rb2, w2.shape[2], pad2, 1, slope, need[3], need[4])
        dx = None
        if need[0]:
            dx = ops.conv1d_dgrad(d_xt, w1, x.shape[2], 1, pad1, dil1, gate=x if slope != 1.0 else None, gate_slope=slope, resid=dy)
        dw1, db1 = _wgrad_into_slots(d_xt, x, r1, rb1, w1.shape[2], pad1, dil1, slope, need[1], need[2])
        return dx, dw1, db1, dw2, db2, None, None, None, None


class ResBlock1(nn.Module):
    """ttts/vqvae/modules.py:224-318: x <- x + c2(lrelu(c1(lrelu(x)))) for three (dilated, plain) conv pairs.  The two
    leaky-relus and the residual add are fused into the convolutions."""

    def __init__(self, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        self.convs1 = nn.ModuleList([_wn_conv_new(channels, kernel_size, d) for d in dilation])
        self.convs2 = nn.ModuleList([_wn_conv_new(channels, kernel_size, 1) for _ in dilation])

    def forward(self, x, x_mask=None):
        if x_mask is not None:
            raise NotImplementedError("ResBlock1 with x_mask is not on the training path (vq2.py never passes one)")
        # one autograd node per pair (default since round 6; TTTS_RESPAIR=0: two nodes).  Round 5 measured it level with the two-node
        # form: the residual's gradient rode into c1's data gradient through the per-element epilogue loop (a chain of dependent
        # round trips).  With the batched gate + residual epilogue (csrc/conv_mfma.hip: conv_store_col16) the 135 engine-side adds it
        # removes show: 104.8 / 105.7 -> 103.2 / 103.1 ms replayed, 107.3 -> 105.2 ms eager (tools/gpu_r6_o.sh)
        fused = _env_int("TTTS_RESPAIR", 1) == 1
        for c1, c2 in zip(self.convs1, self.convs2):
            if fused:          # one autograd node per pair: the residual's gradient is added in c1's data-gradient epilogue
                x = _ResPairFn.apply(x, c1.effective_weight(), c1.bias, c2.effective_weight(), c2.bias, c1.padding, c1.dilation,
                                     c2.padding, LRELU_SLOPE)
            else:
                xt = c1(x, in_slope=LRELU_SLOPE)
                x = c2(xt, in_slope=LRELU_SLOPE, resid=x)
        return x


class Generator(nn.Module):
    """HiFi-GAN decoder, ttts/vqvae/vq2.py:341-415 (resblock type "1")."""

    def __init__(self, initial_channel, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                 upsample_initial_channel, upsample_kernel_sizes, gin_channels=0):
        super().__init__()
        if str(resblock) != "1":
            raise NotImplementedError("only ResBlock1 (vqvae/config.json: resblock '1') is built")
        self.num_kernels = len(resblock_kernel_sizes)
        self.num_upsamples = len(upsample_rates)
        self.conv_pre = Conv1d(initial_channel, upsample_initial_channel, 7, 1, padding=3)
        self.ups = nn.ModuleList()
        for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)):
            up = ConvTranspose1d(upsample_initial_channel // (2 ** i), upsample_initial_channel // (2 ** (i + 1)), k, u,
                                 padding=(k - u) // 2)
            self.ups.append(up.apply_weight_norm("old"))    # `self.ups.apply(init_weights)` is a no-op, see _wn_conv_new
        self.resblocks = nn.ModuleList()
        ch = upsample_initial_channel
        for i in range(len(self.ups)):
            ch = upsample_initial_channel // (2 ** (i + 1))
            for k, d in zip(resblock_kernel_sizes, resblock_dilation_sizes):
                self.resblocks.append(ResBlock1(ch, k, d))
        self.conv_post = Conv1d(ch, 1, 7, 1, padding=3, bias=False)
        if gin_channels != 0:
            self.cond = Conv1d(gin_channels, upsample_initial_channel, 1)

    def forward(self, x, g=None):
        bbias = None
        if g is not None:
            bbias = self.cond(g).squeeze(-1)               # (B, C, 1) broadcast over time -> per-sample bias of conv_pre
        x = self.conv_pre(x, bbias=bbias)
        for i in range(self.num_upsamples):
            x = self.ups[i](x, in_slope=LRELU_SLOPE)
            xs = run_branches([lambda j=j: self.resblocks[i * self.num_kernels + j](x) for j in range(self.num_kernels)], x.device)
            x = add_scale(xs, 1.0 / self.num_kernels)
        return self.conv_post(x, in_slope=0.01, out_act="tanh")   # F.leaky_relu default slope (vq2.py:404)


# ---- small fused pieces -------------------------------------------------------------------------------------------------
class _MulMaskFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return ops.mul_mask(x, mask)

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        return ops.mul_mask(dy, mask), None


def mul_mask(x, mask):
    """x (B, C, T) * mask (B, 1, T)."""
    return _MulMaskFn.apply(x, mask)


class _GaussSampleFn(torch.autograd.Function):
    """z = (m + eps * exp(logs)) * mask with stats = (m | logs) stacked on the channel axis."""

    @staticmethod
    def forward(ctx, stats, eps, mask):
        ctx.save_for_backward(stats, eps, mask)
        return ops.gauss_sample_fwd(stats, eps, mask)

    @staticmethod
    def backward(ctx, dz):
        stats, eps, mask = ctx.saved_tensors
        return ops.gauss_sample_bwd(dz, stats, eps, mask), None, None


class _Upsample2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return ops.upsample2_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        return ops.upsample2_bwd(dy)


def upsample_nearest2(x):
    """F.interpolate(x, size=2*T, mode='nearest') (vq2.py:853-855)."""
    return _Upsample2Fn.apply(x)


class _SnakeAAFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha, beta, fup, fdn):
        ctx.save_for_backward(x, alpha, beta, fup, fdn)
        return ops.snake_aa_fwd(x, alpha, beta, fup, fdn)

    @staticmethod
    def backward(ctx, dy):
        x, alpha, beta, fup, fdn = ctx.saved_tensors
        dx, da, db = ops.snake_aa_bwd(dy, x, alpha, beta, fup, fdn)
        return dx, da, db, None, None


def kaiser_sinc_filter1d(cutoff, half_width, kernel_size):
    """Windowed-sinc low-pass prototype of alias_free_torch/filter.py:28-56, shape (1, 1, kernel_size)."""
    half = kernel_size // 2
    delta_f = 4 * half_width
    A = 2.285 * (half - 1) * math.pi * delta_f + 7.95
    if A > 50.0:
        beta = 0.1102 * (A - 8.7)
    elif A >= 21.0:
        beta = 0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0)
    else:
        beta = 0.0
    window = torch.kaiser_window(kernel_size, beta=beta, periodic=False)
    time = (torch.arange(-half, half) + 0.5) if kernel_size % 2 == 0 else (torch.arange(kernel_size) - half)
    filt = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    filt = filt / filt.sum()
    return filt.view(1, 1, kernel_size)


class _FilterHolder(nn.Module):
    def __init__(self, filt):
        super().__init__()
        self.register_buffer("filter", filt)


class _DownSampleHolder(nn.Module):
    def __init__(self, filt):
        super().__init__()
        self.lowpass = _FilterHolder(filt)


class SnakeBeta(nn.Module):
    """activations.SnakeBeta(in_features, alpha_logscale=True) parameters (activations.py:62-119)."""

    def __init__(self, in_features, alpha_logscale=True):
        super().__init__()
        if not alpha_logscale:
            raise NotImplementedError("only alpha_logscale=True is on the path (vq2.py:703)")
        self.alpha = nn.Parameter(torch.zeros(in_features))
        self.beta = nn.Parameter(torch.zeros(in_features))


class Activation1d(nn.Module):
    """alias_free_torch.Activation1d(SnakeBeta) with ratio 2 / 12-tap filters (act.py:8-28) as one fused kernel.
    State-dict keys as in the reference: act.alpha, act.beta, upsample.filter, downsample.lowpass.filter."""

    def __init__(self, activation, up_ratio=2, down_ratio=2, up_kernel_size=12, down_kernel_size=12):
        super().__init__()
        if (up_ratio, down_ratio, up_kernel_size, down_kernel_size) != (2, 2, 12, 12) or not isinstance(activation, SnakeBeta):
            raise NotImplementedError("Activation1d: only SnakeBeta with ratio 2 and 12-tap filters is built")
        self.act = activation
        self.upsample = _FilterHolder(kaiser_sinc_filter1d(0.5 / up_ratio, 0.6 / up_ratio, up_kernel_size))
        self.downsample = _DownSampleHolder(kaiser_sinc_filter1d(0.5 / down_ratio, 0.6 / down_ratio, down_kernel_size))

    def forward(self, x):
        return _SnakeAAFn.apply(x, self.act.alpha, self.act.beta, self.upsample.filter.view(-1),
                                self.downsample.lowpass.filter.view(-1))


# ---- WaveNet stack (modules.WN, ttts/vqvae/modules.py:136-221) as ONE autograd node ------------------------------------
_WN_DUAL = os.environ.get("TTTS_WN_DUAL", "1") == "1"      # (A/B switch: 0 = the two launches the dual convolution replaces)


class _WNFn(torch.autograd.Function):
    """params = [in_v, in_g, in_b, rs_v, rs_g, rs_b] * n_layers (old-style weight norm tensors).
    Per layer: x_in = conv_k(x) + g_l ; acts = tanh*sigmoid ; (res | skip) = 1x1(acts) ; x = (x + res) * mask ;
    out += skip ; finally out * mask.  The mask multiplies, the conditioning add, the residual add and the skip
    accumulation are epilogues of the convolutions."""

    @staticmethod
    def forward(ctx, x, mask, gcond, n_layers, K, dil_rate, *params):
        B, H, T = x.shape
        m2 = mask.reshape(B, T).contiguous() if mask is not None else None
        out = torch.empty_like(x)
        saved, xi = [], x.contiguous()
        want_dw = any(ctx.needs_input_grad[6:])
        dwz = []
        ctx.dwz = dwz
        # the per-layer conditioning biases as rows of ONE (n_layers, B, 2H) copy (a strided slice + .contiguous() per layer was 16
        # small copy launches per stack)
        gc = gcond[:, :, 0].reshape(B, n_layers, 2 * H).permute(1, 0, 2).contiguous() if gcond is not None else None
        for i in range(n_layers):
            in_v, in_g, in_b, rs_v, rs_g, rs_b = params[6 * i:6 * i + 6]
            dil = dil_rate ** i
            pad = int((K * dil - dil) / 2)
            if in_g is None:  # banked: in_v / rs_v ARE the effective weights (WeightNormBank leaves, normalised once per phase)
                w_in, n_in, w_rs, n_rs = in_v, None, rs_v, None
            elif want_dw:     # the same launches clear the buffers the backward's weight-gradient kernels accumulate into
                w_in, n_in, z_in = ops.weight_norm_fwd(in_v, in_g, want_zero=True)
                w_rs, n_rs, z_rs = ops.weight_norm_fwd(rs_v, rs_g, want_zero=True)
                dwz += [z_in, z_rs]
            else:
                w_in, n_in = ops.weight_norm_fwd(in_v, in_g)
                w_rs, n_rs = ops.weight_norm_fwd(rs_v, rs_g)
            bb = gc[i] if gc is not None else None
            x_in = ops.conv1d_fwd(xi, w_in, in_b, None, 1, pad, dil, bbias=bb)
            acts = ops.gate_fwd(x_in, ops.GATE_TANH_SIGMOID)
            if i < n_layers - 1:
                if _WN_DUAL:          # res | skip: one launch, two destinations (x_next = (x + res) * mask, out += skip * mask)
                    x_next = torch.empty_like(xi)
                    ops.conv1d_fwd_dual(acts, w_rs, rs_b, xi, m2, x_next, out, H, accumulate2=i > 0)
                else:
                    x_next = ops.conv1d_fwd(acts, w_rs[:H], rs_b[:H], xi, omask=m2)
                    ops.conv1d_fwd(acts, w_rs[H:], rs_b[H:], omask=m2, out=out, accumulate=i > 0)
            else:
                x_next = None
                ops.conv1d_fwd(acts, w_rs, rs_b, omask=m2, out=out, accumulate=i > 0)
            saved += [xi, x_in, acts, w_in, n_in, w_rs, n_rs]
            xi = x_next
        ctx.save_for_backward(m2, *saved, *params)
        ctx.cfg = (n_layers, K, dil_rate, gcond is not None, H)
        ctx.prefs = params
        return out

    @staticmethod
    def backward(ctx, dout):
        n_layers, K, dil_rate, has_g, H = ctx.cfg
        m2 = ctx.saved_tensors[0]
        saved = ctx.saved_tensors[1:1 + 7 * n_layers]
        params = ctx.saved_tensors[1 + 7 * n_layers:]
        dout = dout.contiguous()
        B, _, T = dout.shape
        dsk = ops.mul_mask(dout, m2) if m2 is not None else dout
        dres = None
        pgrads = [None] * (6 * n_layers)
        # the conditioning gradients of all layers land in rows of ONE buffer (a zero-filled temporary + a bias-gradient launch per
        # layer and a concatenation were 33 launches per stack)
        dg_all = torch.empty(n_layers, B * 2 * H, dtype=dout.dtype, device=dout.device) if has_g else None
        for i in reversed(range(n_layers)):
            xi, x_in, acts, w_in, n_in, w_rs, n_rs = saved[7 * i:7 * i + 7]
            in_v, in_g, in_b, rs_v, rs_g, rs_b = params[6 * i:6 * i + 6]
            dil = dil_rate ** i
            pad = int((K * dil - dil) / 2)
            banked = in_g is None
            pre = (None, None)                                   # the buffers the forward cleared (used once)
            if banked:                                           # the bank's dW views (cleared by its refresh)
                pre = (_grad_slot(ctx.prefs[6 * i]), _grad_slot(ctx.prefs[6 * i + 3]))
            elif len(ctx.dwz) == 2 * n_layers and ctx.dwz[2 * i] is not None:
                pre = (ctx.dwz[2 * i], ctx.dwz[2 * i + 1])
                ctx.dwz[2 * i] = ctx.dwz[2 * i + 1] = None
            dw_rs = pre[1] if pre[1] is not None else torch.zeros_like(w_rs)
            slots = [_grad_slot(t) if t is not None else None for t in ctx.prefs[6 * i:6 * i + 6]]   # in_v, in_g, in_b, rs_v, ..
            direct = all(sl is not None for sl, t in zip(slots, ctx.prefs[6 * i:6 * i + 6]) if t is not None)
            if i < n_layers - 1:
                dacts = ops.conv1d_dgrad(dres, w_rs[:H], T)
                ops.conv1d_dgrad(dsk, w_rs[H:], T, out=dacts, accumulate=True)
                # (bias gradients ride on the weight-gradient calls: the kernels stream dy anyway)
                db_t = slots[5] if direct else torch.zeros_like(rs_b)
                ops.conv1d_wgrad(dres, acts, 1, out=dw_rs[:H], db=db_t[:H])
                ops.conv1d_wgrad(dsk, acts, 1, out=dw_rs[H:], db=db_t[H:])
                db_rs = None if direct else db_t
            else:
                dacts = ops.conv1d_dgrad(dsk, w_rs, T)
                db_t = slots[5] if direct else torch.zeros_like(rs_b)
                ops.conv1d_wgrad(dsk, acts, 1, out=dw_rs, db=db_t)
                db_rs = None if direct else db_t
            # (the conditioning gradient d g_l[b][c] = sum_t d x_in rides on the gate's backward: one launch)
            dx_in = ops.gate_bwd(dacts, x_in, ops.GATE_TANH_SIGMOID, rowsum=dg_all[i] if has_g else None)
            db_in = slots[2] if direct else torch.zeros_like(in_b)
            dw_in = ops.conv1d_wgrad(dx_in, xi, K, 1, pad, dil, db=db_in, out=pre[0])
            dres = ops.conv1d_dgrad(dx_in, w_in, T, 1, pad, dil, resid=dres, omask=m2 if i > 0 else None)
            if banked:
                pass                                             # WeightNormBank.finish() forms (dv, dg) for every layer at once
            elif direct:
                ops.weight_norm_bwd(dw_in, in_v, in_g, n_in, dv=slots[0], dg=slots[1])
                ops.weight_norm_bwd(dw_rs, rs_v, rs_g, n_rs, dv=slots[3], dg=slots[4])
            else:
                dv_in, dg_in = ops.weight_norm_bwd(dw_in, in_v, in_g, n_in)
                dv_rs, dg_rs = ops.weight_norm_bwd(dw_rs, rs_v, rs_g, n_rs)
                pgrads[6 * i:6 * i + 6] = [dv_in, dg_in, db_in, dv_rs, dg_rs, db_rs]
        dg = dg_all.view(n_layers, B, 2 * H).permute(1, 0, 2).reshape(B, n_layers * 2 * H).unsqueeze(-1) if has_g else None
        return (dres, None, dg, None, None, None, *pgrads)


class WN(nn.Module):
    """modules.WN(hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels, p_dropout)."""

    def __init__(self, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=0, p_dropout=0):
        super().__init__()
        assert kernel_size % 2 == 1
        if p_dropout != 0:
            raise NotImplementedError("WN dropout is 0 everywhere on the path (vq2.py:708, modules.py:425)")
        self.hidden_channels, self.kernel_size, self.dilation_rate = hidden_channels, (kernel_size,), dilation_rate
        self.n_layers, self.gin_channels, self.p_dropout = n_layers, gin_channels, p_dropout
        self.in_layers = nn.ModuleList()
        self.res_skip_layers = nn.ModuleList()
        if gin_channels != 0:
            self.cond_layer = Conv1d(gin_channels, 2 * hidden_channels * n_layers, 1).apply_weight_norm("old")
        for i in range(n_layers):
            dilation = dilation_rate ** i
            padding = int((kernel_size * dilation - dilation) / 2)
            self.in_layers.append(Conv1d(hidden_channels, 2 * hidden_channels, kernel_size, dilation=dilation,
                                         padding=padding).apply_weight_norm("old"))
            rs = 2 * hidden_channels if i < n_layers - 1 else hidden_channels
            self.res_skip_layers.append(Conv1d(hidden_channels, rs, 1).apply_weight_norm("old"))

    def forward(self, x, x_mask, g=None, **kwargs):
        gcond = self.cond_layer(g) if g is not None else None
        params = []
        for a, b in zip(self.in_layers, self.res_skip_layers):
            wa, wb = a._banked_weight(), b._banked_weight()
            if wa is not None and wb is not None and all(
                    _grad_slot(t) is not None for t in (wa, wb, a.bias, b.bias)):
                params += [wa, None, a.bias, wb, None, b.bias]
            else:
                params += [a.weight_v, a.weight_g, a.bias, b.weight_v, b.weight_g, b.bias]
        return _WNFn.apply(x, x_mask, gcond, self.n_layers, self.kernel_size[0], self.dilation_rate, *params)


class Flip(nn.Module):
    """modules.Flip (modules.py:377-385)."""

    def forward(self, x, *args, reverse=False, **kwargs):
        x = torch.flip(x, [1])
        if not reverse:
            return x, torch.zeros(x.size(0), dtype=x.dtype, device=x.device)
        return x


class ResidualCouplingLayer(nn.Module):
    """modules.ResidualCouplingLayer (modules.py:403-459), both directions; `mean_only=True` on the path."""

    def __init__(self, channels, hidden_channels, kernel_size, dilation_rate, n_layers, p_dropout=0, gin_channels=0,
                 mean_only=False):
        assert channels % 2 == 0, "channels should be divisible by 2"
        super().__init__()
        if not mean_only:
            raise NotImplementedError("only mean_only=True coupling layers are on the path (vq2.py:240)")
        self.channels, self.hidden_channels, self.half_channels = channels, hidden_channels, channels // 2
        self.mean_only = mean_only
        self.pre = Conv1d(self.half_channels, hidden_channels, 1)
        self.enc = WN(hidden_channels, kernel_size, dilation_rate, n_layers, p_dropout=p_dropout, gin_channels=gin_channels)
        self.post = Conv1d(hidden_channels, self.half_channels, 1)
        with torch.no_grad():
            self.post.weight.zero_()
            self.post.bias.zero_()

    def forward(self, x, x_mask, g=None, reverse=False):
        x0, x1 = torch.split(x, [self.half_channels] * 2, 1)
        h = self.pre(x0.contiguous(), omask=x_mask)
        h = self.enc(h, x_mask, g=g)
        if reverse:
            # inference direction (modules.py:456-459, logs = 0): x1 <- (x1 - m) * mask with m = post(h) * mask -- the subtraction is
            # the convolution's own epilogue: y = x1 * mask, then y += -1 * (bias + conv(h)) * mask.  No autograd (infer / decode).
            from .. import ops
            y = mul_mask(x1.contiguous(), x_mask).contiguous()
            ops.conv1d_fwd(h.detach(), self.post.effective_weight().detach(), self.post.bias.detach(), pad=self.post.padding, out_scale=-1.0,
                           out=y, accumulate=True, omask=x_mask)
            return torch.cat([x0, y], 1)
        x1 = self.post(h, resid=x1.contiguous(), omask=x_mask)      # m + x1 * mask  (logs = 0)
        x = torch.cat([x0, x1], 1)
        return x, torch.zeros(x.size(0), dtype=x.dtype, device=x.device)
