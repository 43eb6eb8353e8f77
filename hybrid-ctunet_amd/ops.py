"""Autograd glue between torch tensors and the libctunet_hip.so kernels.

Every op is a torch.autograd.Function whose forward/backward launch hand-written HIP kernels through the C ABI on
torch's current HIP stream.  Activations are channels-last: a volume is a contiguous [B, D, H, W, C] tensor
(== row-major [rows, C] matrix), dtype float32 (parity mode) or bfloat16.  Parameters stay fp32 in the layout of
the reference's state_dict; packed panels for the MFMA kernels are cached per (parameter version, weights epoch).
"""
from __future__ import annotations

import math
import os
import time
import weakref
from typing import Optional, Sequence, Tuple

import torch

from . import _lib as L
from ._lib import AttnGeom, Epilogue, Geom, call, dcode, ptr, stream

LRELU_SLOPE = 0.01

# ---------------------------------------------------------------------------------------------------------------
# side streams (CTUNet runs its two independent encoder branches on two HIP streams; autograd replays each node on the
# stream its forward ran on).  Whoever consumes results on the main stream joins them first.
# ---------------------------------------------------------------------------------------------------------------
if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
    # (parameters of the two branches are touched from different streams step after step: expected here)
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)

_side_streams = {}
_used_side = set()   # keys of side streams handed out since the last join_side_streams()
_ws_epoch = 0        # bumped by new_workspace_epoch(): a HIP-graph capture gets per-stream workspaces of its own


def new_workspace_epoch():
    """Fresh per-stream workspaces from now on.  A graph capture must not share them with eager steps: their state at the
    end of the captured step has to equal their state at its start, and workspaces created INSIDE the capture are
    zero-filled by every replay."""
    global _ws_epoch
    _ws_epoch += 1


# HIP stream priorities (range on this device: 0 = normal, -1 = high).  Measured with every combination for the branch /
# weight-gradient streams: 49.1 - 49.8 ms per step, no trend - all streams stay at the default.
WGRAD_PRIORITY = int(os.environ.get("CTU_WGRAD_PRIORITY", "0"))
BRANCH_PRIORITY = int(os.environ.get("CTU_BRANCH_PRIORITY", "0"))


def side_stream(device, tag="branch"):
    key = (device, tag)
    st = _side_streams.get(key)
    if st is None:
        prio = WGRAD_PRIORITY if isinstance(tag, tuple) and tag[0] == "wgrad" else BRANCH_PRIORITY
        st = _side_streams[key] = torch.cuda.Stream(device=device, priority=prio)
    _used_side.add(key)
    return st


def side_streams(device):
    """Side streams that carried work since the last join."""
    return [_side_streams[k] for k in _used_side if k[0] == device]


# Weight-gradient kernels of a layer depend only on (dY, X), not on the data gradient that the rest of the backward pass
# waits for: with a training harness that owns the gradient storage (direct sinks) they are queued on a companion stream
# of the layer's compute stream and overlap the data-gradient chain - the small-volume stages and the 864-token ViT
# trunk are launch-latency bound, two kernels side by side fill what one leaves idle.  The optimizer joins the streams.
# Round 3: on (CTU_NO_WGRAD_STREAM=1 switches them off).  Same box, python bench.py --steps 20 --warmup 5: 45.86 / 45.86 ms per step
# with the companion streams, 46.89 / 46.84 without at four hardware queues; 46.8 - 47.0 against 48.3 at eight, with or without
# an RCCL communicator in the process (profiles/r03_bench_rccl_group_vs_hw_queues.log).  (An earlier "46.86 vs 46.89, no gain" in
# the history of this comment compared two runs of a bench.py that switched them on in both.)  What made the step fragile against
# the stream-to-queue mapping was the optimizer's OWN update stream (66 ms with >= 5 queues, profiles/r03_bench_hw_queues_sweep.log),
# which is gone: the per-bucket updates ride on the branch stream.
WGRAD_STREAM = not os.environ.get("CTU_NO_WGRAD_STREAM")


class _WgradSide:
    def __init__(self, enabled, device, *tensors):
        self.on = bool(enabled and WGRAD_STREAM and device.type == "cuda" and not torch.cuda.is_current_stream_capturing())
        self.device, self.tensors = device, tensors

    def __enter__(self):
        if self.on:
            cur = torch.cuda.current_stream()
            ws = side_stream(self.device, ("wgrad", cur.cuda_stream))
            ws.wait_stream(cur)                      # dY (and everything before it) is ordered in front of the side work
            for t in self.tensors:
                if t is not None:
                    t.record_stream(ws)              # freed by the caller's scope while the side stream may still read it
            self._cm = torch.cuda.stream(ws)
            self._cm.__enter__()
            ensure_join_after_backward()
        return self

    def __exit__(self, *exc):
        if self.on:
            self._cm.__exit__(*exc)
        return False


_join_queued = False


_stash_watch = []   # slots a GradStash parked a gradient in during the backward pass that is running


def _join_after_backward():
    global _join_queued
    _join_queued = False
    join_side_streams()
    lost = [s for s in _stash_watch if _parked(s)]
    _stash_watch.clear()
    if lost:
        # the consumer the stash relies on (the layer whose data-gradient kernel adds the parked gradient in its epilogue) did
        # not run in this backward pass - e.g. a loss over a deep-supervision head alone: plain autograd would have propagated
        # that gradient, so dropping it silently would be wrong
        for s in lost:
            s.clear()
        raise RuntimeError(f"{len(lost)} gradient(s) were parked by ops.GradStash but their consuming layer never ran in this "
                           "backward pass (backpropagating through one head of a shared tensor only?); run such a backward with "
                           "ops.STASH_SHORTCUT_CONV = False / without the model's gradient stashes")


def ensure_join_after_backward():
    """Called from a backward node that queued work on a companion stream: when the whole backward pass has been enqueued, the
    stream backward() was called from waits for every side stream - so that whatever reads .grad next on that stream
    (clip_grad_norm_, GradScaler.unscale_, a torch optimizer over FlatParams views, a test's .cpu()) is ordered behind the
    weight-gradient kernels without knowing about them.  One engine callback per backward pass."""
    global _join_queued
    if not _join_queued:
        _join_queued = True
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_join_after_backward)
        except RuntimeError:      # not inside a backward pass (a Function's backward called by hand)
            _join_queued = False


def join_side_streams():
    """Make torch's current stream wait for everything queued so far on the side streams (weight-gradient kernels write
    straight into the flat gradient buffer from whichever stream their layer ran on; the optimizer / the all-reduce of the
    last bucket must not start before them)."""
    if not torch.cuda.is_available():
        return
    cur = torch.cuda.current_stream()
    for key in list(_used_side):
        if key[0] == cur.device:
            cur.wait_stream(_side_streams[key])
            _used_side.discard(key)

# ---------------------------------------------------------------------------------------------------------------
# packed-weight cache
# ---------------------------------------------------------------------------------------------------------------
_weights_epoch = 0
_pack_cache = {}  # id(nn.Parameter) -> (weakref to it, {(kind, dtype): (version key, packed tensor)})


def bump_weights_epoch():
    """Call after parameters were modified by something torch's version counter cannot see (the fused AdamW)."""
    global _weights_epoch
    _weights_epoch += 1


class _Built:
    """Where and when a cached panel was produced: a consumer on another stream waits for the event once."""
    __slots__ = ("event", "sid", "waited")

    def __init__(self, device):
        self.sid, self.event, self.waited = 0, None, None
        if device.type == "cuda":
            self.sid = stream()
            self.event = torch.cuda.Event()
            self.event.record()
            self.waited = {self.sid}

    def fence(self):
        if self.event is not None:
            sid = stream()
            if sid not in self.waited:
                torch.cuda.current_stream().wait_event(self.event)
                self.waited.add(sid)


def _packed(param: torch.Tensor, kind: str, dtype: torch.dtype, builder):
    """Packed panel of a parameter, rebuilt when the parameter changes.  Only nn.Parameter objects are cached (keyed
    by object identity, weakly): temporaries could alias a freed tensor's address."""
    if not isinstance(param, torch.nn.Parameter) or (param.is_cuda and torch.cuda.is_current_stream_capturing()):
        # (inside a graph capture the packing kernel must be part of the graph: a replay runs no Python, and a panel cached
        # now would be read stale after the next optimizer step)
        with torch.no_grad():
            return builder()
    ver = (param._version, _weights_epoch, param.data_ptr(), tuple(param.shape))
    ent = _pack_cache.get(id(param))
    if ent is None or ent[0]() is not param:
        pid = id(param)
        ent = (weakref.ref(param, lambda _r, pid=pid: _pack_cache.pop(pid, None)), {})
        _pack_cache[pid] = ent
    slot = ent[1]
    hit = slot.get((kind, dtype))
    if hit is not None and hit[0] == ver:
        hit[2].fence()
        return hit[1]
    with torch.no_grad():
        t = builder()
    slot[(kind, dtype)] = (ver, t, _Built(t.device))
    return t


# ---------------------------------------------------------------------------------------------------------------
# bf16 parameter mirrors: train.FusedAdamW keeps a bf16 copy of every parameter current from inside its update kernel
# and registers the per-parameter views here; the bf16 GEMMs then take Linear / 1x1x1 conv weights from the mirror
# instead of casting them every step.  A mirror is used only while the parameter object is untouched since the last
# sync (version counter) and no invisible write happened (generation, bumped by invalidate_bf16_mirrors()).
# ---------------------------------------------------------------------------------------------------------------
_bf16_mirrors = {}
_mirror_gen = 0


def mirror_generation() -> int:
    return _mirror_gen


def invalidate_bf16_mirrors():
    """Call after writing parameters behind torch's back other than through FusedAdamW.step() (e.g. a broadcast into
    the flat buffer)."""
    global _mirror_gen
    _mirror_gen += 1


def register_bf16_mirror(param: torch.Tensor, view: torch.Tensor):
    pid = id(param)
    _bf16_mirrors[pid] = (weakref.ref(param, lambda _r, pid=pid: _bf16_mirrors.pop(pid, None)), view,
                          (param._version, param.data_ptr(), _mirror_gen))


def _bf16_weight(param: torch.Tensor):
    ent = _bf16_mirrors.get(id(param))
    if ent is not None and ent[0]() is param and ent[2] == (param._version, param.data_ptr(), _mirror_gen):
        return ent[1]
    return None


# ---------------------------------------------------------------------------------------------------------------
# direct gradient sinks: a training harness that owns pre-zeroed, persistent fp32 gradient storage (train.FlatParams)
# registers (parameter storage address -> callback).  Weight-gradient kernels whose output layout equals the
# parameter layout (Linear / 1x1x1 conv weights, biases) then accumulate straight into `param.grad` and call the
# callback, instead of materialising a zero-filled temporary that autograd adds into .grad afterwards.
# ---------------------------------------------------------------------------------------------------------------
_grad_sinks = {}


def register_grad_sink(param: torch.Tensor, callback):
    _grad_sinks[param.data_ptr()] = (weakref.ref(param), callback)


def unregister_grad_sink(param: torch.Tensor):
    _grad_sinks.pop(param.data_ptr(), None)


def clear_grad_sinks():
    _grad_sinks.clear()
    _sink_uses.clear()


# A parameter may be used by several ops of one graph (weight sharing).  Every forward that will accumulate into a sink
# announces itself (sink_expect); the sink's callback - "this gradient is complete" - fires only when the last announced
# use has accumulated.  Counts are per storage address and are cleared at the step boundary (reset_grad_sink_counts:
# FlatParams.zero_grad / DataParallel.finish), so a forward whose backward never ran cannot poison the next step.
_sink_uses = {}


def sink_expect(param, needed: bool = True):
    """Called from an autograd.Function's forward (where grad mode is off: `needed` is ctx.needs_input_grad[i])."""
    if needed and param is not None and param.data_ptr() in _grad_sinks:
        k = param.data_ptr()
        _sink_uses[k] = _sink_uses.get(k, 0) + 1


def reset_grad_sink_counts():
    _sink_uses.clear()


def _direct_grad(param):
    """param.grad if a sink is registered for this parameter and its .grad can be accumulated into in place; the second
    value is the 'accumulated' notification, which reports the gradient complete once every counted use has run."""
    ent = _grad_sinks.get(param.data_ptr()) if param is not None else None
    if ent is None:
        return None, None
    p = ent[0]()
    if p is None or p.grad is None or p.grad.dtype != torch.float32 or not p.grad.is_contiguous() or \
            p.grad.shape != param.shape:
        return None, None

    def done():
        k = p.data_ptr()
        left = _sink_uses.get(k, 1) - 1
        if left > 0:
            _sink_uses[k] = left
            return
        _sink_uses.pop(k, None)
        ent[1](p)
    return p.grad, done


def permute3(src: torch.Tensor, dst: torch.Tensor, n, s, d, accumulate=False):
    """dst[i0*d0+i1*d1+i2*d2] (+)= src[i0*s0+i1*s1+i2*s2]; src fp32."""
    assert src.dtype == torch.float32
    call("ctu_permute3", ptr(src), ptr(dst), dcode(dst.dtype), n[0], n[1], n[2], s[0], s[1], s[2], d[0], d[1], d[2],
         int(accumulate), stream())


def _t3(v) -> Tuple[int, int, int]:
    return tuple(int(x) for x in v) if isinstance(v, (tuple, list)) else (int(v),) * 3


def _geom(B, din, dout, C1, C2, N, k=(1, 1, 1), s=(1, 1, 1), p=(0, 0, 0), mode=0) -> Geom:
    return Geom(B, din[0], din[1], din[2], dout[0], dout[1], dout[2], C1, C2, N, k[0], k[1], k[2], s[0], s[1], s[2],
                p[0], p[1], p[2], mode)


def _plain_geom(M, K, N) -> Geom:
    return _geom(1, (M, 1, 1), (M, 1, 1), K, 0, N)


def _splitk_for(M: int, N: int, K: int, dma: bool = False) -> int:
    """Split-K factor for a plain GEMM: few output tiles (the 864-token ViT trunk) and a long reduction.  The LDS-DMA
    kernel (bf16, K % 64 == 0) first drops to 64 x 64 tiles; it splits only when even those leave the chip idle."""
    if dma:
        items = ((M + 63) // 64) * ((N + 63) // 64)
        if items >= 128 or K < 1024:
            return 1
        return max(1, min(K // 256, (255 + items) // items))
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if tiles >= 128 or K < 512:
        return 1
    return max(1, min(16, K // 128, 512 // tiles))


def _conv_splitk(M: int, N: int, K: int, taps: int) -> int:
    """Split of the (tap, 32-channel chunk) loop of a gathered conv on the generic implicit-GEMM kernel: the strided
    3x3x3 convs at 6x6x12 / 12x12x24 have 14-108 output tiles and a reduction of 27 x 256."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128 if N > 64 else 1)
    its = taps * ((K + 31) // 32)
    if tiles >= 128 or its < 32:
        return 1
    return max(1, min(its // 8, (255 + tiles) // tiles))


def _wskey(device):
    """Persistent workspaces are private to a (device, stream) pair: kernels of one stream run in order, so a workspace
    handed back clean by one launch is clean for the next; two streams (the two encoder branches of CTUNet run on two,
    see networks/hybrid_CTUNet.py) must never share one."""
    return (device, stream(), _ws_epoch) if device.type == "cuda" else (device, 0, 0)


_SPLITK_WS = {}


def _splitk_workspace(device, n):
    """Persistent fp32 split-K workspace (zero between calls: the finish kernel hands it back zeroed)."""
    key = _wskey(device)
    ws = _SPLITK_WS.get(key)
    if ws is None or ws.numel() < n:
        ws = _SPLITK_WS[key] = torch.zeros(max(n, 1 << 22), dtype=torch.float32, device=device)
    return ws


def _epi(ldc, bias=None, residual=None, act=0, out2=None, n_split=0, ldc2=0, scatter=None, splitk_ws=None,
         splitk=1, w_kn=0, in_acc=None, in_rows=0, pre_out=None) -> Epilogue:
    e = Epilogue()
    e.pre_out = ptr(pre_out)
    e.w_kn = w_kn
    e.in_acc = ptr(in_acc)
    e.in_rows = in_rows
    e.splitk = splitk if splitk_ws is not None else 1
    e.splitk_ws = ptr(splitk_ws)
    e.bias = ptr(bias)
    e.residual = ptr(residual)
    e.act = act
    e.ldc = ldc
    e.out2 = ptr(out2)
    e.n_split = n_split
    e.ldc2 = ldc2
    if scatter is not None:
        e.scatter = 1
        (e.n_per_tap, e.sc_D, e.sc_H, e.sc_W, e.sc_kd, e.sc_kh, e.sc_kw) = scatter
    else:
        e.scatter = 0
    return e


def _check_act(x: torch.Tensor):
    L.require_device(x)
    if not x.is_contiguous():
        raise RuntimeError("hybrid-ctunet_amd ops expect contiguous channels-last activations")


# ---------------------------------------------------------------------------------------------------------------
# GEMM-shaped ops
# ---------------------------------------------------------------------------------------------------------------
def _igemm_nt(x1, x2, w, out, g: Geom, e: Epilogue):
    call("ctu_igemm_nt", dcode(x1.dtype), ptr(x1), ptr(x2), ptr(w), ptr(out), g, e, stream())


_TN_WS = {}


def _tn_workspace(device):
    """Persistent scratch (16 Mi floats) for the two-stage reduction of small weight-gradient panels; contents are
    irrelevant between calls (every partial is written before it is read), single compute stream."""
    key = _wskey(device)
    ws = _TN_WS.get(key)
    if ws is None:
        ws = _TN_WS[key] = torch.empty(1 << 24, dtype=torch.float32, device=device)
    return ws


def _igemm_tn(p, ldp, q1, q2, dw, g: Geom, bias_grad=None):
    ws = _tn_workspace(p.device)
    call("ctu_igemm_tn", dcode(p.dtype), ptr(p), ldp, ptr(q1), ptr(q2), ptr(dw), ptr(bias_grad), g, ptr(ws), ws.numel(),
         stream())


def _plain_gemm(x, w, out, M, K, N, bias=None, residual=None, act=0, w_kn=0, in_acc=None, in_rows=0, pre_out=None):
    """out[M,N] = act(x[M,K] @ w[N,K]^T + bias) + residual, with split-K when there are few tiles and a long K."""
    sk = 1 if (in_acc is not None or pre_out is not None) else \
        _splitk_for(M, N, K, dma=x.dtype == torch.bfloat16 and K % 32 == 0)
    ws = _splitk_workspace(x.device, M * N) if sk > 1 else None
    _igemm_nt(x, None, w, out, _plain_geom(M, K, N),
              _epi(N, bias=bias, residual=residual, act=act, splitk_ws=ws, splitk=sk, w_kn=w_kn, in_acc=in_acc,
                   in_rows=in_rows, pre_out=pre_out))


USE_W_KN = True  # tests clear this together with the "generic_gemm" hook (the generic kernels need W transposed)


def _linear_weight(weight, w2, dtype):
    """[N][K] weight in the activation dtype: the parameter itself (fp32), its optimizer-maintained bf16 mirror, or a
    cached cast."""
    if dtype == torch.float32 and w2.is_contiguous():
        return w2
    if dtype == torch.bfloat16:
        m = _bf16_weight(weight)
        if m is not None:
            return m
    return _packed(weight, "lin_f", dtype, lambda: w2.detach().to(dtype).contiguous())


class LinearFn(torch.autograd.Function):
    """y = act(x @ W^T + b) (+ residual).  x: [..., K]; W: [N, K] fp32 (nn.Linear / 1x1x1 conv layout).
    Reference: nn.Linear at vit.py:36,39,59,62,117 and hybrid_CTUNet.py:402,457,465,519,522,632-633,641,679;
    1x1x1 nn.Conv3d at resnet.py:96,100 and hybrid_CTUNet.py:75-83."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, act: int, in_stats: bool = False, grad_stash=None):
        _check_act(x)
        ctx.grad_stash = grad_stash
        N, K = weight.shape[0], weight[0].numel()
        M = x.numel() // K
        acc, rows = None, 0
        if (in_stats and FUSE_IN_STATS and USE_W_KN and x.dtype == torch.bfloat16 and K % 32 == 0 and x.dim() >= 3
                and bias is None and residual is None and act == 0):
            rows = M // x.shape[0]  # rows per batch item: a 1x1x1 conv whose output feeds an InstanceNorm
            if rows % 128 == 0 and rows * x.shape[0] == M:
                acc = _in_acc_take(x.device, x.shape[0] * N * 2)
        w2 = weight.reshape(N, K)
        wf = _linear_weight(weight, w2, x.dtype)
        pre = None
        out = torch.empty((*x.shape[:-1], N), dtype=x.dtype, device=x.device)
        if act == 1 and any(ctx.needs_input_grad[:3]) and USE_W_KN and x.dtype == torch.bfloat16 and K % 32 == 0:
            # GELU in the GEMM epilogue, the pre-activation (for GELU') stored alongside: no separate activation pass
            pre = torch.empty_like(out)
            _plain_gemm(x, wf, out, M, K, N, bias=bias, residual=residual, act=1, pre_out=pre)
        elif act == 1 and any(ctx.needs_input_grad[:3]):
            # keep the pre-activation for GELU'
            pre = torch.empty_like(out)
            _plain_gemm(x, wf, pre, M, K, N, bias=bias)
            call("ctu_gelu_fwd", dcode(x.dtype), ptr(pre), ptr(out), out.numel(), stream())
            if residual is not None:
                call("ctu_add", dcode(x.dtype), ptr(out), ptr(residual), ptr(out), out.numel(), stream())
        else:
            _plain_gemm(x, wf, out, M, K, N, bias=bias, residual=residual, act=act,
                        in_acc=acc[1] if acc is not None else None, in_rows=rows)
            if acc is not None:
                global _last_in_acc
                _last_in_acc = acc
        ctx.save_for_backward(x, weight, pre, bias)
        sink_expect(weight, ctx.needs_input_grad[1])
        sink_expect(bias, ctx.needs_input_grad[2])
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        ctx.act = act
        return out

    @staticmethod
    def backward(ctx, gy):
        x, weight, pre, bias = ctx.saved_tensors
        gy = gy.contiguous()
        N, K = weight.shape[0], weight[0].numel()
        M = x.numel() // K
        gres = gy if ctx.has_res else None
        g = gy
        if ctx.act == 1:
            g = torch.empty_like(gy)
            call("ctu_gelu_bwd", dcode(gy.dtype), ptr(gy), ptr(pre), ptr(g), gy.numel(), stream())
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            # gradient that reached x through another branch (GradStash): added in this GEMM's epilogue instead of by
            # a separate autograd accumulation pass over three tensors
            extra = _take(ctx.grad_stash)
            if extra is not None and (extra.shape != x.shape or extra.dtype != x.dtype or not extra.is_contiguous()):
                extra = extra.to(x.dtype).contiguous().view_as(x)
            if USE_W_KN and x.dtype == torch.bfloat16 and N % 64 == 0 and K % 8 == 0:
                # dX = dY @ W: the LDS-DMA GEMM reads the forward weight [N][K] reduction-major, no transposed copy
                _plain_gemm(g, _linear_weight(weight, weight.reshape(N, K), x.dtype), gx, M, N, K, w_kn=1, residual=extra)
            else:
                wd = _packed(weight, "lin_d", x.dtype, lambda: weight.detach().reshape(N, K).t().to(x.dtype).contiguous())
                _plain_gemm(g, wd, gx, M, N, K, residual=extra)
        want_gb = ctx.has_bias and ctx.needs_input_grad[2]
        # accumulate straight into persistent .grad storage when the harness registered it (see register_grad_sink)
        gw_buf, gw_done = _direct_grad(weight) if ctx.needs_input_grad[1] else (None, None)
        gb_buf, gb_done = _direct_grad(bias) if want_gb else (None, None)
        if want_gb and gb_buf is None:
            gb = gb_buf = torch.zeros(N, dtype=torch.float32, device=x.device)
        direct = (gw_done is not None or not ctx.needs_input_grad[1]) and (gb_done is not None or not want_gb)
        with _WgradSide(direct, x.device, g, x):
            if ctx.needs_input_grad[1]:
                if gw_buf is None:
                    gw = gw_buf = torch.zeros(weight.shape, dtype=torch.float32, device=x.device)
                _igemm_tn(g, N, x, None, gw_buf, _plain_geom(M, K, N), bias_grad=gb_buf if want_gb else None)
            elif want_gb:
                call("ctu_colsum", dcode(g.dtype), ptr(g), None, M, N, N, ptr(gb_buf), stream())
            if gw_done is not None:
                gw_done()
            if gb_done is not None:
                gb_done()
        return gx, gw, gb, gres, None, None, None


STASH_SHORTCUT_CONV = True  # ResBlock with a conv shortcut: conv3's data gradients added inside conv1's data-gradient kernel


_CONSUMED = object()   # left in a stash slot by its consumer: a gradient parked afterwards would be lost


def _take(slot):
    """The gradient parked in `slot` (None if there is none), and the slot marked as read: GradStash.backward raises if it
    comes later - autograd replayed the two nodes in the other order and the parked gradient would silently be dropped."""
    if slot is None:
        return None
    g = None
    if slot and slot[-1] is not _CONSUMED:
        g = slot.pop()
    if not slot or slot[-1] is not _CONSUMED:
        slot.append(_CONSUMED)
    return g


def _parked(slot) -> bool:
    return bool(slot) and slot[-1] is not _CONSUMED


class GradStash(torch.autograd.Function):
    """Identity whose backward parks the incoming gradient in `slot` (a list) instead of returning it.  Used on the
    residual branch of a bottleneck: the branch's gradient w.r.t. the block input is then added inside the data-gradient
    GEMM of the block's first conv (LinearFn, grad_stash=slot), which autograd runs later, rather than by a separate
    elementwise pass.  The node must be created AFTER that conv's node so that it runs first in backward."""

    @staticmethod
    def forward(ctx, x, slot):
        ctx.slot = slot
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if ctx.slot and ctx.slot[-1] is _CONSUMED:
            raise RuntimeError("GradStash: the consumer of this slot ran before the gradient was parked (the stash node must be "
                               "created after the consuming layer's node, on the same stream)")
        ctx.slot.append(g)
        _stash_watch.append(ctx.slot)
        ensure_join_after_backward()
        return None, None


# Measurement only (tools/step_trace.py): with TRACE a list, trace_point(t, label) records a HIP event on the current stream where the
# forward pass queues it and - through an identity autograd node - where the backward pass reaches the same point, each with the host
# time of the enqueue.  TRACE is None in production: trace_point returns its argument and nothing is recorded.
TRACE = None


def mark(label: str):
    if TRACE is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        TRACE.append((label, torch.cuda.current_stream().cuda_stream, ev, time.perf_counter()))


class _TraceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, label):
        ctx.label = label
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        mark("bwd:" + ctx.label)
        return g, None


def trace_point(t, label: str):
    if TRACE is None:
        return t
    mark(label)
    return _TraceFn.apply(t, label) if (torch.is_grad_enabled() and t.requires_grad) else t


def linear(x, weight, bias=None, residual=None, act: int = 0, in_stats: bool = False, grad_stash=None):
    """in_stats=True: the caller will feed the result to instance_norm (a 1x1x1 conv of a ResNet bottleneck): its
    statistics are then summed in the GEMM epilogue when the shape allows it."""
    global _last_in_acc
    _last_in_acc = None
    out = LinearFn.apply(x, weight, bias, residual, act, in_stats, grad_stash)
    if _last_in_acc is not None:
        out._ctu_in_acc = _last_in_acc
        _last_in_acc = None
    return out


class ConvFn(torch.autograd.Function):
    """nn.Conv3d(bias=False) on channels-last volumes, input optionally the channel-concat of two tensors
    (torch.cat at hybrid_CTUNet.py:199,618 folded into the K loop).  x: [B,D,H,W,C]; weight: [N, C1+C2, kd,kh,kw].
    Reference: get_conv_layer, networks/resnet.py:17-50 (3x3x3 s1/s2, 1x1x1 s2)."""

    @staticmethod
    def forward(ctx, x1, x2, weight, stride, padding, grad_stash=None, x1_b16=False, link=None, grad_stash2=None):
        _check_act(x1)
        ctx.grad_stash = grad_stash
        ctx.grad_stash2 = grad_stash2 if x2 is not None else None
        ctx.x1_b16 = bool(x1_b16)
        ctx.link = link
        B, D, H, W, C1 = x1.shape
        C2 = 0 if x2 is None else x2.shape[-1]
        N = weight.shape[0]
        k = tuple(weight.shape[2:])
        taps = k[0] * k[1] * k[2]
        K = C1 + C2
        assert weight.shape[1] == K
        dout = tuple((n + 2 * p - kk) // s + 1 for n, p, kk, s in zip((D, H, W), padding, k, stride))
        out = torch.empty((B, *dout, N), dtype=x1.dtype, device=x1.device)
        if _halo_ok(k, stride, padding) and C1 % 32 == 0 and C2 % 32 == 0:
            wfr = _packed_frag(weight, "conv_hf", x1.dtype, N, K, taps, K * taps, taps, 1, 0)
            acc = None
            if FUSE_IN_STATS and x1.dtype == torch.bfloat16 and B * D * H * W * max(C1, C2) < (1 << 31):
                # InstanceNorm statistics of the output from the conv epilogue (consumed by instance_norm, if it follows)
                acc = _in_acc_take(x1.device, B * N * 2)
            ws = _tn_workspace(x1.device)
            call("ctu_conv3_halo", dcode(x1.dtype), ptr(x1), ptr(x2), ptr(wfr), ptr(out), None, B, D, H, W, C1, C2, N, 0,
                 N, 0, ptr(acc[1]) if acc is not None else None, None, None, ptr(ws), ws.numel(), int(ctx.x1_b16), stream())
            ctx.in_acc = acc
            # the InstanceNorm that follows may hand this conv's data- and weight-gradient kernels their dY in the blocked
            # layout (they are its only readers)
            ctx.b16_grad_ok = B16_LAYOUT and x1.dtype == torch.bfloat16 and N % 32 == 0 and \
                B * D * H * W * max(N, K) < (1 << 31)
        else:
            if ctx.x1_b16:
                raise RuntimeError("a CTU_LAYOUT_B16 tensor reached a convolution that is not on the 3x3x3 halo kernel")
            wf = _packed(weight, "conv_f", x1.dtype,
                         lambda: _pack(weight, (taps, N, K), (1, K * taps, taps), x1.dtype))
            g = _geom(B, (D, H, W), dout, C1, C2, N, k, stride, padding, 0)
            M = B * dout[0] * dout[1] * dout[2]
            sk = _conv_splitk(M, N, K, taps)
            _igemm_nt(x1, x2, wf, out, g, _epi(N, splitk=sk, splitk_ws=_splitk_workspace(x1.device, M * N) if sk > 1 else None))
        ctx.save_for_backward(x1, x2, weight)
        if taps > 1:
            sink_expect(weight, ctx.needs_input_grad[2])
        ctx.cfg = (stride, padding, k, dout)
        global _last_in_acc, _last_b16_grad_ok
        _last_b16_grad_ok = bool(getattr(ctx, "b16_grad_ok", False))
        if getattr(ctx, "in_acc", None) is not None:
            _last_in_acc = ctx.in_acc
            ctx.in_acc = None
        return out

    @staticmethod
    def backward(ctx, gy):
        x1, x2, weight = ctx.saved_tensors
        stride, padding, k, dout = ctx.cfg
        gy = gy.contiguous()
        gy_b16 = ctx.link is not None and ctx.link.gy_b16   # dY written blocked by the InstanceNorm backward behind this conv
        x1_b16 = ctx.x1_b16
        B, D, H, W, C1 = x1.shape
        C2 = 0 if x2 is None else x2.shape[-1]
        N, K = weight.shape[0], C1 + C2
        taps = k[0] * k[1] * k[2]
        g1 = g2 = gw = None
        if (gy_b16 or x1_b16) and not (_halo_ok(k, stride, padding) and N % 32 == 0 and C1 % 32 == 0 and C2 % 32 == 0):
            raise RuntimeError("CTU_LAYOUT_B16 operand outside the halo convolution kernels")
        if ctx.needs_input_grad[0] or (x2 is not None and ctx.needs_input_grad[1]):
            g1 = torch.empty_like(x1)
            g2 = torch.empty_like(x2) if x2 is not None else None
            if _halo_ok(k, stride, padding) and N % 32 == 0 and (x2 is None or C1 % 32 == 0):
                # dX = conv(dY, W flipped, in/out channels swapped): W'(n'=cin, c'=cout, t') = W[cout][cin][26 - t']
                wfr = _packed_frag(weight, "conv_hd", x1.dtype, K, N, taps, taps, K * taps, 1, 1)
                ws = _tn_workspace(x1.device)
                extra = extra2 = None
                if x1.dtype == torch.bfloat16 and B * D * H * W * max(N, K) < (1 << 31):
                    # gradients of x1 / x2 through another branch (GradStash): added in the epilogue
                    extra = _take(ctx.grad_stash)
                    if extra is not None:
                        if extra.shape != x1.shape or extra.dtype != x1.dtype or not extra.is_contiguous():
                            extra = extra.to(x1.dtype).contiguous().view_as(x1)
                    extra2 = _take(ctx.grad_stash2) if x2 is not None else None
                    if extra2 is not None:
                        if extra2.shape != x2.shape or extra2.dtype != x2.dtype or not extra2.is_contiguous():
                            extra2 = extra2.to(x2.dtype).contiguous().view_as(x2)
                call("ctu_conv3_halo", dcode(x1.dtype), ptr(gy), None, ptr(wfr), ptr(g1), ptr(g2), B, D, H, W, N, 0, K,
                     C1 if x2 is not None else 0, C1, C2, None, ptr(extra), ptr(extra2), ptr(ws), ws.numel(), int(gy_b16),
                     stream())
            else:
                # dX[v][c] = sum_t sum_n dY[(v + p - t)/s][n] W[n][c][t]  ->  panel [t][c][n]
                wd = _packed(weight, "conv_d", x1.dtype,
                             lambda: _pack(weight, (taps, K, N), (1, taps, K * taps), x1.dtype))
                gd = _geom(B, dout, (D, H, W), N, 0, K, k, stride, padding, 1)
                Mi = B * D * H * W
                sk = _conv_splitk(Mi, K, N, taps) if x2 is None else 1
                extra = None
                if sk == 1:
                    # gradient of x1 through another consumer (GradStash): added in this GEMM's epilogue (to the x1 columns)
                    extra = _take(ctx.grad_stash)
                if extra is not None:
                    if extra.shape != x1.shape or extra.dtype != x1.dtype or not extra.is_contiguous():
                        extra = extra.to(x1.dtype).contiguous().view_as(x1)
                _igemm_nt(gy, None, wd, g1, gd, _epi(C1, residual=extra, out2=g2, n_split=C1 if x2 is not None else 0, ldc2=C2,
                                                     splitk=sk,
                                                     splitk_ws=_splitk_workspace(x1.device, Mi * K) if sk > 1 else None))
        if ctx.needs_input_grad[2]:
            gw_buf, gw_done = _direct_grad(weight) if taps > 1 else (None, None)
            with _WgradSide(gw_buf is not None, x1.device, gy, x1, x2):
                if gw_buf is not None:
                    # persistent zeroed scratch panel: the wgrad kernel accumulates into it, permute3 adds it into the
                    # parameter's gradient storage in the parameter's layout and hands the panel back zeroed
                    panel = _panel_scratch(x1.device, taps * N * K).view(taps, N, K)
                else:
                    panel = torch.zeros((taps, N, K), dtype=torch.float32, device=x1.device)
                if _halo_ok(k, stride, padding) and C1 % 32 == 0 and C2 % 32 == 0:
                    wws = _wgrad_workspace(x1.device) if x1.dtype == torch.bfloat16 else None
                    call("ctu_conv3_halo_wgrad", dcode(x1.dtype), ptr(gy), ptr(x1), ptr(x2), ptr(panel), B, D, H, W, C1, C2,
                         N, int(x1_b16), int(gy_b16), ptr(wws), wws.numel() if wws is not None else 0, stream())
                else:
                    gq = _geom(B, (D, H, W), dout, C1, C2, N, k, stride, padding, 0)
                    _igemm_tn(gy, N, x1, x2, panel, gq)
                if taps == 1:
                    gw = panel.view(weight.shape)
                elif gw_buf is not None:
                    permute3(panel, gw_buf, (N, K, taps), (K, 1, N * K), (K * taps, taps, 1), accumulate=2)
                    gw_done()
                else:
                    gw = torch.empty(weight.shape, dtype=torch.float32, device=x1.device)
                    permute3(panel, gw, (N, K, taps), (K, 1, N * K), (K * taps, taps, 1))
        if _parked(ctx.grad_stash):  # not consumed by a fused epilogue (generic path): add it here
            g1 = g1 + _take(ctx.grad_stash).to(g1.dtype)
        if _parked(ctx.grad_stash2):
            g2 = g2 + _take(ctx.grad_stash2).to(g2.dtype)
        return g1, g2, gw, None, None, None, None, None, None


_PANEL_SCRATCH = {}


WGRAD_PARTIALS = True  # halo weight gradient: per-split partial panels + a reduce kernel instead of fp32 atomics
_WGRAD_WS = {}


def _wgrad_workspace(device):
    """Scratch for the partial panels of ctu_conv3_halo_wgrad: 256 workgroups x 54 tiles x 1024 floats (57 MB) cover every shape
    (splits x 27 x N x K <= that for all of them); one per (device, stream, epoch) like the other workspaces.  Contents irrelevant."""
    if not WGRAD_PARTIALS:
        return None
    key = _wskey(device)
    t = _WGRAD_WS.get(key)
    if t is None:
        t = _WGRAD_WS[key] = torch.empty(256 * 54 * 1024 + 27 * 64 * 1024, dtype=torch.float32, device=device)
    return t


def _panel_scratch(device, n):
    """Persistent fp32 scratch for multi-tap weight-gradient panels; zero between uses (permute3 accumulate=2 clears
    what it read), single compute stream."""
    key = _wskey(device)
    t = _PANEL_SCRATCH.get(key)
    if t is None or t.numel() < n:
        t = _PANEL_SCRATCH[key] = torch.zeros(max(n, 27 * 512 * 512), dtype=torch.float32, device=device)
    return t[:n]


USE_HALO_CONV = True  # tests flip this to run the generic implicit GEMM on the same shapes

# CTU_LAYOUT_B16 (include/ctunet_hip.h): the tensor between an InstanceNorm and the 3x3x3 halo convolution that is its only
# consumer - forward (norm output -> conv input) and backward (norm input gradient -> the producing conv's dY) - is stored
# as [C/16][voxels][16].  Same bytes, same shape attribute.  Forward: instance_norm(out_b16=True) marks its result
# (`_ctu_b16`), conv3d reads the mark and raises if the convolution cannot run on the halo kernel.  Backward: the conv's
# autograd node and the norm applied to its output share a _B16Link (set in the norm's forward, read in the conv's
# backward) - the blocked gradient travels along exactly one graph edge.
B16_LAYOUT = not os.environ.get("CTU_NO_B16")


class _B16Link:
    """Shared between a halo convolution's autograd node and the InstanceNorm applied to its output: the norm's forward
    sets `gy_b16` when its backward will write that conv's dY in CTU_LAYOUT_B16; the conv's backward reads it."""
    __slots__ = ("gy_b16",)

    def __init__(self):
        self.gy_b16 = False


def wants_b16(conv_weight, x: torch.Tensor, stride, padding) -> bool:
    """Should the InstanceNorm in front of this convolution write its output blocked?  (x: the norm's input = the conv's
    future input, channels-last [B, D, H, W, C].)"""
    if not (B16_LAYOUT and USE_HALO_CONV and x.dtype == torch.bfloat16 and x.dim() == 5):
        return False
    k = tuple(conv_weight.shape[2:])
    C, N = x.shape[-1], conv_weight.shape[0]
    return (_halo_ok(k, _t3(stride), _t3(padding)) and C % 32 == 0 and conv_weight.shape[1] == C and
            x.numel() // C * max(C, N) < (1 << 31))


def _halo_ok(k, stride, padding) -> bool:
    return USE_HALO_CONV and tuple(k) == (3, 3, 3) and tuple(stride) == (1, 1, 1) and tuple(padding) == (1, 1, 1)


# Batched fragment packing.  Every 3x3x3 conv repacks its weights twice per step (forward panel, flipped / transposed
# data-gradient panel): 94 launches of ~7 us for CTUNet, all latency.  The first step registers each (parameter, kind)
# job as it packs it singly; from then on the first stale lookup of a step repacks ALL registered panels with one launch
# (job table in device memory, two alternating output buffers so that a graph built before the update keeps its panels
# alive one more round).
BATCH_PACK = not os.environ.get("CTU_NO_BATCH_PACK")
_frag_jobs = {}      # (id(param), kind, dtype) -> [weakref(param), (N, K, taps, sn, sc, st, flip), numel]
_frag_state = {"side": 0, "tables": [None, None]}  # per side: (signature, job keys, table tensor, buffer, views)


def _frag_numel(N, K, taps):
    return (K // 32) * taps * 2 * ((N + 31) // 32) * 512


def _packed_frag(param, kind, dtype, N, K, taps, sn, sc, st, flip):
    """Fragment panel of a conv weight (cached per parameter like _packed; batched repacking, see above)."""
    single = lambda: _pack_frag(param, N, K, taps, sn, sc, st, flip, dtype)  # noqa: E731
    if not (BATCH_PACK and isinstance(param, torch.nn.Parameter) and param.is_contiguous() and param.dtype == torch.float32) \
            or (param.is_cuda and torch.cuda.is_current_stream_capturing()):
        return _packed(param, kind, dtype, single)
    key = (id(param), kind, dtype)
    job = _frag_jobs.get(key)
    if job is None or job[0]() is not param or job[1] != (N, K, taps, sn, sc, st, flip):
        _frag_jobs[key] = [weakref.ref(param, lambda _r, key=key: _frag_jobs.pop(key, None)), (N, K, taps, sn, sc, st, flip),
                           _frag_numel(N, K, taps)]
        return _packed(param, kind, dtype, single)       # first sight: pack it alone
    ent = _pack_cache.get(id(param))
    ver = (param._version, _weights_epoch, param.data_ptr(), tuple(param.shape))
    hit = ent[1].get((kind, dtype)) if ent is not None and ent[0]() is param else None
    if hit is not None and hit[0] == ver:
        hit[2].fence()
        return hit[1]
    _repack_all(param.device)
    ent = _pack_cache.get(id(param))
    hit = ent[1].get((kind, dtype)) if ent is not None and ent[0]() is param else None
    if hit is not None and hit[0] == ver:
        hit[2].fence()
        return hit[1]
    return _packed(param, kind, dtype, single)           # (not covered by the batch after all)


def _repack_all(device):
    """One launch for every registered panel of live parameters on `device`; results go into the _packed cache."""
    live = []
    for key, job in list(_frag_jobs.items()):
        prm = job[0]()
        if prm is None or prm.device != device or not prm.is_contiguous():
            continue
        live.append((key, prm, job))
    if not live:
        return
    st = _frag_state
    side = st["side"] = 1 - st["side"]
    sig = tuple((key, prm.data_ptr()) for key, prm, _ in live)
    tab = st["tables"][side]
    if tab is None or tab[0] != sig:
        import numpy as np
        offs, total_bytes = [], 0
        for key, prm, job in live:
            offs.append(total_bytes)
            total_bytes += (job[2] * (2 if key[2] == torch.bfloat16 else 4) + 255) // 256 * 256
        for sd in (side, 1 - side):   # both sides now: a later capture of the step finds its table on the device
            buf = torch.empty(total_bytes, dtype=torch.uint8, device=device)
            rows = np.zeros((len(live), 10), dtype=np.int64)    # 80-byte ctu_pack_job records
            views = []
            nblocks = 0
            for r, ((key, prm, job), off) in enumerate(zip(live, offs)):
                N, K, taps, sn, sc, stt, flip = job[1]
                nbytes = job[2] * (2 if key[2] == torch.bfloat16 else 4)
                view = buf[off:off + nbytes].view(key[2])
                views.append(view)
                rows[r, 0] = prm.data_ptr()
                rows[r, 1] = view.data_ptr()
                rows[r, 2:7] = (sn, sc, stt, job[2], nblocks)
                nblocks += max(1, min(2048, (job[2] + 2047) // 2048))   # ~8 elements per thread, at most 2048 workgroups a job
                i32 = np.array([N, K, taps, flip, (N + 31) // 32, dcode(key[2])], dtype=np.int32)
                rows[r, 7:10] = i32.view(np.int64)
            table = torch.from_numpy(rows.view(np.uint8).reshape(-1)).to(device)
            st["tables"][sd] = (sig, [k for k, _, _ in live], table, buf, views, nblocks)
        tab = st["tables"][side]
    call("ctu_pack_frag_batched", ptr(tab[2]), len(live), tab[5], stream())
    built = _Built(device)   # one event for the whole batch
    for (key, prm, job), view in zip(live, tab[4]):
        ver = (prm._version, _weights_epoch, prm.data_ptr(), tuple(prm.shape))
        ent = _pack_cache.get(id(prm))
        if ent is None or ent[0]() is not prm:
            pid = id(prm)
            ent = (weakref.ref(prm, lambda _r, pid=pid: _pack_cache.pop(pid, None)), {})
            _pack_cache[pid] = ent
        ent[1][(key[1], key[2])] = (ver, view, built)


def _pack_frag(weight, N, K, taps, sn, sc, st, flip, dtype):
    """MFMA-fragment-order panel [K/32][taps][2][ceil(N/32)][64][8] of W(n, c, t) = weight.flat[n*sn + c*sc + t*st]."""
    w = weight.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    out = torch.empty((K // 32) * taps * 2 * ((N + 31) // 32) * 512, dtype=dtype, device=w.device)
    call("ctu_pack_frag", ptr(w), ptr(out), dcode(dtype), N, K, taps, sn, sc, st, flip, stream())
    return out


def _pack(weight, n, src_strides, dtype):
    """out[i0][i1][i2] = weight.flat[i0*s0 + i1*s1 + i2*s2] cast to dtype (weight fp32, contiguous)."""
    w = weight.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    out = torch.empty(n, dtype=dtype, device=w.device)
    permute3(w, out, n, src_strides, (n[1] * n[2], n[2], 1))
    return out


def conv3d(x1, weight, stride=1, padding=0, x2=None, grad_stash=None, grad_stash2=None):
    global _last_in_acc
    _last_in_acc = None
    global _last_b16_grad_ok
    _last_b16_grad_ok = False
    link = _B16Link()
    out = ConvFn.apply(x1, x2, weight, _t3(stride), _t3(padding), grad_stash, bool(getattr(x1, "_ctu_b16", False)), link,
                       grad_stash2)
    if _last_in_acc is not None:
        out._ctu_in_acc = _last_in_acc  # instance_norm(out, ...) picks the statistics up instead of re-reading `out`
        _last_in_acc = None
    if _last_b16_grad_ok:
        out._ctu_b16_link = link        # instance_norm(out, ...) may write this conv's dY in CTU_LAYOUT_B16
        _last_b16_grad_ok = False
    return out


# fused InstanceNorm statistics: a small ring of persistent fp64 accumulators (zero when free).  A conv epilogue fills
# one; instance_norm on that very output finalises it (which zeroes it again).  An accumulator whose output never met
# an InstanceNorm is re-zeroed when the ring comes round to it.
FUSE_IN_STATS = True
_last_in_acc = None
_IN_ACC_RING = {}


def _in_acc_take(device, n):
    key = _wskey(device)
    ring = _IN_ACC_RING.get(key)
    if ring is None or ring[0][0][0].numel() < n:
        size = max(n, 1 << 13)
        ring = _IN_ACC_RING[key] = [[[torch.zeros(size, dtype=torch.float64, device=device), False] for _ in range(4)], 0]
    slots, nxt = ring
    slot = slots[nxt]
    ring[1] = (nxt + 1) % len(slots)
    if slot[1]:
        slot[0].zero_()
    slot[1] = True
    return slot, slot[0]


class ConvTransposeFn(torch.autograd.Function):
    """nn.ConvTranspose3d with kernel == stride, padding 0, bias=False (hybrid_CTUNet.py:177-185,232-240,286-294):
    a GEMM [M, Cin] x [Cin, taps*Cout] whose epilogue scatters each tap to its output voxel.
    weight: [Cin, Cout, kd, kh, kw]."""

    @staticmethod
    def forward(ctx, x, weight, grad_stash=None):
        _check_act(x)
        ctx.grad_stash = grad_stash   # gradient of x through another consumer (GradStash): added in the data-gradient epilogue
        B, D, H, W, Cin = x.shape
        Cout = weight.shape[1]
        k = tuple(weight.shape[2:])
        taps = k[0] * k[1] * k[2]
        M = B * D * H * W
        wf = _packed(weight, "convt_f", x.dtype,
                     lambda: _pack(weight, (taps, Cout, Cin), (1, taps, Cout * taps), x.dtype))
        out = torch.empty((B, D * k[0], H * k[1], W * k[2], Cout), dtype=x.dtype, device=x.device)
        g = _geom(B, (D, H, W), (D, H, W), Cin, 0, taps * Cout)
        _igemm_nt(x, None, wf, out, g, _epi(Cout, scatter=(Cout, D, H, W, k[0], k[1], k[2])))
        ctx.save_for_backward(x, weight)
        return out

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        B, D, H, W, Cin = x.shape
        Cout = weight.shape[1]
        k = tuple(weight.shape[2:])
        taps = k[0] * k[1] * k[2]
        dbig = (D * k[0], H * k[1], W * k[2])
        gx = gw = None
        if ctx.needs_input_grad[0]:
            # dX[v][ci] = sum_t sum_co dY[v*k + t][co] W[ci][co][t]: a stride-k conv over dY; panel [t][ci][co]
            wd = _packed(weight, "convt_d", x.dtype,
                         lambda: _pack(weight, (taps, Cin, Cout), (1, Cout * taps, taps), x.dtype))
            gx = torch.empty_like(x)
            extra = _take(ctx.grad_stash)
            if extra is not None and (extra.shape != x.shape or extra.dtype != x.dtype or not extra.is_contiguous()):
                extra = extra.to(x.dtype).contiguous().view_as(x)
            _igemm_nt(gy, None, wd, gx, _geom(B, dbig, (D, H, W), Cout, 0, Cin, k, k, (0, 0, 0), 0), _epi(Cin, residual=extra))
        if ctx.needs_input_grad[1]:
            # dW[ci][co][t] = sum_v x[v][ci] dY[v*k + t][co]: P = x (N := Cin), Q = dY gathered (C := Cout)
            panel = torch.zeros((taps, Cin, Cout), dtype=torch.float32, device=x.device)
            _igemm_tn(x, Cin, gy, None, panel, _geom(B, dbig, (D, H, W), Cout, 0, Cin, k, k, (0, 0, 0), 0))
            gw = torch.empty(weight.shape, dtype=torch.float32, device=x.device)
            permute3(panel, gw, (Cin, Cout, taps), (Cout, 1, Cin * Cout), (Cout * taps, taps, 1))
        return gx, gw, None


def conv_transpose3d(x, weight, grad_stash=None):
    return ConvTransposeFn.apply(x, weight, grad_stash)


class ConvCin1Fn(torch.autograd.Function):
    """nn.Conv3d with one input channel (vit_encoder0.conv1, hybrid_CTUNet.py:57-65; stem, resnet.py:150-155).
    x: [B, D, H, W, 1]; weight [N, 1, kd, kh, kw]; direct VALU kernels; no input gradient (x is the image)."""

    @staticmethod
    def forward(ctx, x, weight, stride, padding):
        _check_act(x)
        B, D, H, W, _ = x.shape
        N = weight.shape[0]
        k = tuple(weight.shape[2:])
        taps = k[0] * k[1] * k[2]
        dout = tuple((n + 2 * p - kk) // s + 1 for n, p, kk, s in zip((D, H, W), padding, k, stride))
        out = torch.empty((B, *dout, N), dtype=x.dtype, device=x.device)
        g = _geom(B, (D, H, W), dout, 8, 0, N, k, stride, padding, 0)
        P = None
        if CIN1_AS_GEMM and x.dtype == torch.bfloat16 and taps > 1 and N % 8 == 0:
            # patch matrix + LDS-DMA GEMM (the direct VALU kernel reaches a tenth of the VALU peak on the 7x7x7 stem)
            kpad = (taps + 63) // 64 * 64
            M = B * dout[0] * dout[1] * dout[2]
            P = torch.empty((M, kpad), dtype=x.dtype, device=x.device)
            call("ctu_im2col_cin1", ptr(x), ptr(P), g, kpad, stream())

            def build():
                w = torch.zeros((N, kpad), dtype=x.dtype, device=x.device)
                w[:, :taps] = weight.detach().reshape(N, taps)
                return w
            rows = M // B
            acc = _in_acc_take(x.device, B * N * 2) if (FUSE_IN_STATS and rows % 128 == 0) else None
            _plain_gemm(P, _packed(weight, "cin1_g", x.dtype, build), out, M, kpad, N,
                        in_acc=acc[1] if acc is not None else None, in_rows=rows)
            if acc is not None:  # these convs feed an InstanceNorm: its sums come from the GEMM epilogue
                global _last_in_acc
                _last_in_acc = acc
        elif taps == 1 and tuple(stride) == (1, 1, 1) and N % 8 == 0:
            # out[m][n] = x[m] * w[n]: a pure store stream
            call("ctu_outer_rows", dcode(x.dtype), ptr(x), ptr(weight.detach().reshape(N)), ptr(out), x.numel(), N, stream())
        else:
            wf = _packed(weight, "cin1_f", torch.float32, lambda: _pack(weight, (1, taps, N), (0, 1, taps), torch.float32))
            call("ctu_conv_cin1_fwd", dcode(x.dtype), ptr(x), ptr(wf), ptr(out), g, stream())
        ctx.save_for_backward(x, weight, P)
        ctx.cfg = (stride, padding, k, dout)
        return out

    @staticmethod
    def backward(ctx, gy):
        x, weight, P = ctx.saved_tensors
        stride, padding, k, dout = ctx.cfg
        gy = gy.contiguous()
        B, D, H, W, _ = x.shape
        N = weight.shape[0]
        taps = k[0] * k[1] * k[2]
        gw = None
        if ctx.needs_input_grad[1] and P is not None:
            # dW[n][k] = sum_m dY[m][n] P[m][k] on the saved patch matrix
            M, kpad = P.shape
            dwp = torch.zeros((N, kpad), dtype=torch.float32, device=x.device)
            _igemm_tn(gy, N, P, None, dwp, _plain_geom(M, kpad, N))
            gw = dwp[:, :taps].reshape(weight.shape)
        elif ctx.needs_input_grad[1] and taps == 1 and tuple(stride) == (1, 1, 1) and N % 8 == 0:
            # dW[n] = sum_m x[m] dY[m][n]: column sums of dY weighted by the image
            gw = torch.zeros(weight.shape, dtype=torch.float32, device=x.device)
            call("ctu_colsum", dcode(gy.dtype), ptr(gy), ptr(x), x.numel(), N, N, ptr(gw), stream())
        elif ctx.needs_input_grad[1]:
            panel = torch.zeros((taps, N), dtype=torch.float32, device=x.device)
            g = _geom(B, (D, H, W), dout, 8, 0, N, k, stride, padding, 0)
            call("ctu_conv_cin1_wgrad", dcode(x.dtype), ptr(x), ptr(gy), ptr(panel), g, stream())
            gw = torch.empty(weight.shape, dtype=torch.float32, device=x.device)
            permute3(panel, gw, (1, N, taps), (0, 1, N), (0, taps, 1))
        return None, gw, None, None


CIN1_AS_GEMM = True  # tests clear this to run the direct kernels


def conv3d_cin1(x, weight, stride, padding):
    global _last_in_acc
    _last_in_acc = None
    out = ConvCin1Fn.apply(x, weight, _t3(stride), _t3(padding))
    if _last_in_acc is not None:
        out._ctu_in_acc = _last_in_acc
        _last_in_acc = None
    return out


# ---------------------------------------------------------------------------------------------------------------
# normalisation / activation
# ---------------------------------------------------------------------------------------------------------------
SIGN_MASK = not os.environ.get("CTU_NO_SIGN_MASK")


class InstanceNormFn(torch.autograd.Function):
    """y = act(InstanceNorm3d(x) + residual), eps 1e-5, no affine, LeakyReLU(0.01)
    (resnet.py:97-124,156-157,198; hybrid_CTUNet.py:84-104)."""

    @staticmethod
    def forward(ctx, x, residual, act: bool, out_b16: bool = False):
        _check_act(x)
        B, C = x.shape[0], x.shape[-1]
        S = x.numel() // (B * C)
        stats = torch.empty((B, C, 2), dtype=torch.float32, device=x.device)
        y = torch.empty_like(x)
        dc = dcode(x.dtype)
        fused = getattr(x, "_ctu_in_acc", None)
        if fused is not None and fused[0][1]:
            # the producing conv already summed (y, y^2) in its epilogue
            call("ctu_in_finalize", B, S, C, ptr(fused[1]), ptr(stats), stream())
            fused[0][1] = False
            x._ctu_in_acc = None
        else:
            acc = _in_workspace(x.device, B * C * 2)[0]  # zero on entry, handed back zeroed by ctu_in_stats
            call("ctu_in_stats", dc, ptr(x), B, S, C, ptr(acc), ptr(stats), stream())
        if out_b16 and (residual is not None or x.dtype != torch.bfloat16 or C % 16):
            raise RuntimeError("CTU_LAYOUT_B16 output: bf16, C % 16 == 0, no residual")
        # with a residual the LeakyReLU's argument sign survives only in y: record it as one byte per 8 channels, so the two
        # backward kernels read 1/16 of the bytes a second and third pass over y would cost
        mask = None
        if SIGN_MASK and act and residual is not None and C % 8 == 0:
            mask = torch.empty(x.numel() // 8, dtype=torch.uint8, device=x.device)
        call("ctu_in_apply", dc, ptr(x), ptr(stats), ptr(residual), ptr(y), B, S, C, int(act), int(out_b16), ptr(mask),
             stream())
        # x is the output of a halo conv: its gradient has one reader, that conv's backward, which takes it blocked
        link = getattr(x, "_ctu_b16_link", None)
        ctx.dx_b16 = link is not None and x.dtype == torch.bfloat16 and C % 16 == 0 and ctx.needs_input_grad[0]
        if ctx.dx_b16:
            link.gy_b16 = True
            x._ctu_b16_link = None   # one norm per conv output: a second consumer would read a blocked gradient sum
        ctx.has_res = residual is not None
        # without a residual sign(y) == sign(xhat): backward recomputes the LeakyReLU mask from x and needs no y
        ctx.save_for_backward(x, (mask if mask is not None else y) if ctx.has_res else None, stats)
        ctx.has_mask = mask is not None
        ctx.act = int(act)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, stats = ctx.saved_tensors
        mask = None
        if ctx.has_mask:
            y, mask = None, y
        gy = gy.contiguous()
        B, C = x.shape[0], x.shape[-1]
        S = x.numel() // (B * C)
        ws = _in_workspace(x.device, B * C * 2)
        # two sums buffers alternate: this call reduces into the clean one and its apply kernel zeroes the other,
        # which the previous call left dirty and nothing reads any more (stream order)
        sums, dirty, dirty_n = ws[1 + ws[3]], ws[2 - ws[3]], ws[4]
        gx = torch.empty_like(x)
        gres = torch.empty_like(x) if ctx.has_res else None
        dc = dcode(x.dtype)
        # (measured: reducing and applying one batch item at a time, hoping the second read hits the 256 MB Infinity
        # Cache, is 2 % slower than one pass over the whole batch)
        call("ctu_in_bwd_reduce", dc, ptr(gy), ptr(x), ptr(y), ptr(stats), ptr(sums), B, S, C, ctx.act, ptr(mask), stream())
        call("ctu_in_bwd_apply", dc, ptr(gy), ptr(x), ptr(y), ptr(stats), ptr(sums), ptr(gx), ptr(gres), B, S, C,
             ctx.act, ptr(dirty), dirty_n, int(ctx.dx_b16), ptr(mask), stream())
        ws[3] ^= 1
        ws[4] = B * C * 2
        return gx, gres, None, None


_IN_WS = {}


def _in_workspace(device, n):
    """Persistent fp64 InstanceNorm accumulators of one device (single compute stream): [fwd acc, bwd sums A, bwd sums B,
    index of the clean bwd buffer, dirty entries of the other].  The kernels keep them zero between uses."""
    key = _wskey(device)
    ws = _IN_WS.get(key)
    if ws is None or ws[0].numel() < n:
        size = max(n, 1 << 15)
        ws = [torch.zeros(size, dtype=torch.float64, device=device) for _ in range(3)] + [0, 0]
        _IN_WS[key] = ws
    return ws


def instance_norm(x, residual=None, act=False, out_b16: bool = False):
    """out_b16: write the result in CTU_LAYOUT_B16 - only when its one consumer is a 3x3x3 halo convolution (wants_b16)."""
    y = InstanceNormFn.apply(x, residual, act, out_b16)
    if out_b16:
        y._ctu_b16 = True   # ops.conv3d reads the mark; nothing else may consume y
    return y


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm(dim), eps 1e-5 (vit.py:35,55,116,118; hybrid_CTUNet.py:456,518,630-631)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, grad_stash=None):
        _check_act(x)
        ctx.grad_stash = grad_stash
        dim = x.shape[-1]
        rows = x.numel() // dim
        y = torch.empty_like(x)
        mr = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
        call("ctu_layernorm_fwd", dcode(x.dtype), ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mr), rows, dim, stream())
        ctx.save_for_backward(x, gamma, mr, beta)
        sink_expect(gamma, ctx.needs_input_grad[1])
        sink_expect(beta, ctx.needs_input_grad[2])
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, mr, beta = ctx.saved_tensors
        gy = gy.contiguous()
        dim = x.shape[-1]
        rows = x.numel() // dim
        gx = torch.empty_like(x)
        gg_buf, gg_done = _direct_grad(gamma)
        gb_buf, gb_done = _direct_grad(beta)
        gg = gb = None
        if gg_buf is None or gb_buf is None:
            gg_buf = gg = torch.zeros(dim, dtype=torch.float32, device=x.device)
            gb_buf = gb = torch.zeros(dim, dtype=torch.float32, device=x.device)
            gg_done = gb_done = None
        ws = torch.empty(1024 * 2 * dim, dtype=torch.float32, device=x.device)  # CTU_LN_BWD_MAX_BLOCKS partial rows
        # gradient that reached x around the norm (residual branch, GradStash): added while dx is written
        extra = _take(ctx.grad_stash)
        if extra is not None and (extra.shape != x.shape or extra.dtype != x.dtype or not extra.is_contiguous()):
            extra = extra.to(x.dtype).contiguous().view_as(x)
        call("ctu_layernorm_bwd_add", dcode(x.dtype), ptr(gy), ptr(x), ptr(gamma), ptr(mr), ptr(extra), ptr(gx), ptr(gg_buf),
             ptr(gb_buf), ptr(ws), rows, dim, stream())
        if gg_done is not None:
            gg_done()
            gb_done()
        return gx, gg, gb, None


def layer_norm(x, gamma, beta, grad_stash=None):
    return LayerNormFn.apply(x, gamma, beta, grad_stash)


LN_STASH = not os.environ.get("CTU_NO_LN_STASH")  # (A/B switch for measurements)


def norm_with_residual(x, residual, gamma, beta):
    """(LayerNorm(x), residual') for the pre-norm residual pattern `f(LN(x)) + x` (residual is x): the residual is routed
    through GradStash so that its gradient joins the LayerNorm backward's dx instead of being added by autograd.  The
    stash node is created after the norm's node; its backward runs when the block's last op hands back the residual
    gradient, long before the norm's own backward."""
    if residual is x and x.requires_grad and torch.is_grad_enabled() and LN_STASH:
        stash = []
        h = layer_norm(x, gamma, beta, stash)
        return h, GradStash.apply(x, stash)
    return layer_norm(x, gamma, beta), residual


class AddBcastFn(torch.autograd.Function):
    """x + pos_embedding (vit.py:133): x [B, n, C], pos fp32 [1, n, C]."""

    @staticmethod
    def forward(ctx, x, pos):
        _check_act(x)
        B = x.shape[0]
        rows, cols = x.numel() // x.shape[-1], x.shape[-1]
        y = torch.empty_like(x)
        call("ctu_add_bcast", dcode(x.dtype), ptr(x), ptr(pos), ptr(y), rows, cols, rows // B, stream())
        ctx.B = B
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        gp = None
        if ctx.needs_input_grad[1]:
            per = gy.numel() // ctx.B
            gp = torch.zeros((1, *gy.shape[1:]), dtype=torch.float32, device=gy.device)
            call("ctu_colsum", dcode(gy.dtype), ptr(gy), None, ctx.B, per, per, ptr(gp), stream())
        return gy, gp


def add_bcast(x, pos):
    return AddBcastFn.apply(x, pos)


# ---------------------------------------------------------------------------------------------------------------
# attention / fusion / layout
# ---------------------------------------------------------------------------------------------------------------
# ---------------------------------------------------------------------------------------------------------------
# dropout.  Masks come from a counter RNG (csrc/philox.h): a dropout call is identified by
# (seed, offset); backward replays the same pair, nothing is stored.  The state below hands out offsets in call order.
# ---------------------------------------------------------------------------------------------------------------
_drop_seed = None
_drop_offset = 0


def manual_seed(seed: int, offset: int = 0):
    """Seed of the dropout masks (default: torch.initial_seed() at the first dropout call) and the offset of the next
    dropout call."""
    global _drop_seed, _drop_offset
    _drop_seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    _drop_offset = int(offset)


def dropout_state() -> Tuple[int, int]:
    """(seed, offset of the next dropout call)."""
    global _drop_seed
    if _drop_seed is None:
        # default: torch's seed, with the data-parallel rank mixed in - the reference seeds every rank alike
        # (torch.manual_seed in main_worker) and still gets different masks per GPU from its per-device generators
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        _drop_seed = (int(torch.initial_seed()) ^ (rank * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF
    return _drop_seed, _drop_offset


def _next_drop_key() -> Tuple[int, int]:
    global _drop_offset
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        # the (seed, offset) key is a kernel ARGUMENT taken from host state at launch time: a captured launch would replay the same
        # mask forever, silently (GraphedStep / graph_stages with dropout_rate > 0)
        raise RuntimeError("dropout with p > 0 cannot be captured into a HIP graph: the Philox key is a launch argument; run the "
                           "step eagerly (the default) or set dropout_rate = 0")
    seed, off = dropout_state()
    _drop_offset = off + 1
    return seed, off


class DropoutFn(torch.autograd.Function):
    """nn.Dropout(p) in training mode, optionally followed by `+ residual` (the reference's `x = attn(x) + x`)."""

    @staticmethod
    def forward(ctx, x, residual, p: float):
        _check_act(x)
        x = x.contiguous()
        if residual is not None:
            residual = residual.contiguous()
            assert residual.shape == x.shape and residual.dtype == x.dtype
        ctx.key = (float(p),) + _next_drop_key()
        ctx.has_res = residual is not None
        y = torch.empty_like(x)
        call("ctu_dropout", dcode(x.dtype), ptr(x), ptr(residual), ptr(y), x.numel(), *ctx.key, stream())
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        gx = torch.empty_like(gy)
        call("ctu_dropout", dcode(gy.dtype), ptr(gy), None, ptr(gx), gy.numel(), *ctx.key, stream())
        return gx, (gy if ctx.has_res else None), None


def dropout(x, p: float, training: bool = True, residual=None):
    """y = dropout(x) [+ residual]; identity [+ residual] when p == 0 or not training (nn.Dropout semantics)."""
    if p < 0.0 or p >= 1.0:
        raise ValueError(f"dropout probability has to be in [0, 1), got {p}")
    if p == 0.0 or not training:
        return x if residual is None else x + residual
    return DropoutFn.apply(x, residual, p)


class AttentionFn(torch.autograd.Function):
    """softmax(scale * q k^T + rel_pos_bias) v on a fused qkv matrix (vit.py:66-78; hybrid_CTUNet.py:481-511).
    qkv: [B, D, H, W, 3*heads*dh] (part 1/2: window partitions of the volume) or [B, n, 3*heads*dh] (part 0).
    drop_p > 0: dropout of the softmax probabilities (vit.py:74, hybrid_CTUNet.py:459-462), mask regenerated in backward."""

    @staticmethod
    def forward(ctx, qkv, bias_table, part: int, win: int, heads: int, scale: float, drop_p: float = 0.0):
        _check_act(qkv)
        dim = qkv.shape[-1] // 3
        dh = dim // heads
        if part == 0:
            B, n = qkv.shape[0], qkv.shape[1]
            geo = AttnGeom(0, B, n, 1, 1, 0, heads, dh, scale)
            groups, ntok = B, n
        else:
            B, D, H, W = qkv.shape[:4]
            geo = AttnGeom(part, B, D, H, W, win, heads, dh, scale)
            groups, ntok = B * (D // win) * (H // win) * (W // win), win ** 3
        out = torch.empty((*qkv.shape[:-1], dim), dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty((groups * heads, ntok), dtype=torch.float32, device=qkv.device)
        ctx.drop = None
        if drop_p > 0.0:
            ctx.drop = (float(drop_p),) + _next_drop_key()
            call("ctu_attn_fwd_dropout", dcode(qkv.dtype), ptr(qkv), ptr(bias_table), ptr(out), ptr(lse), geo, *ctx.drop,
                 stream())
        else:
            call("ctu_attn_fwd", dcode(qkv.dtype), ptr(qkv), ptr(bias_table), ptr(out), ptr(lse), geo, stream())
        ctx.save_for_backward(qkv, bias_table, out, lse)
        ctx.geo = geo
        return out

    @staticmethod
    def backward(ctx, gout):
        qkv, bias_table, out, lse = ctx.saved_tensors
        gout = gout.contiguous()
        gqkv = torch.empty_like(qkv)
        gbias = torch.zeros_like(bias_table) if bias_table is not None else None
        if ctx.drop is not None:
            call("ctu_attn_bwd_dropout", dcode(qkv.dtype), ptr(qkv), ptr(bias_table), ptr(out), ptr(gout), ptr(lse),
                 ptr(gqkv), ptr(gbias), ctx.geo, *ctx.drop, stream())
        else:
            call("ctu_attn_bwd", dcode(qkv.dtype), ptr(qkv), ptr(bias_table), ptr(out), ptr(gout), ptr(lse), ptr(gqkv),
                 ptr(gbias), ctx.geo, stream())
        return gqkv, gbias, None, None, None, None, None


def attention(qkv, heads, scale, bias_table=None, part=0, win=0, dropout_p: float = 0.0, training: bool = True):
    return AttentionFn.apply(qkv, bias_table, part, win, heads, scale, dropout_p if training else 0.0)


def attention_dropout_mask(pairs: int, ntok: int, p: float, seed: int, offset: int, device) -> torch.Tensor:
    """Verification hook: the keep flags [pairs = groups*heads, ntok, ntok] (uint8) the attention kernels apply for the
    dropout call (p, seed, offset)."""
    keep = torch.empty((pairs, ntok, ntok), dtype=torch.uint8, device=device)
    call("ctu_attn_dropout_mask", ptr(keep), pairs, ntok, float(p), seed, offset, stream())
    return keep


class PwaFn(torch.autograd.Function):
    """Binary cross-weight core (hybrid_CTUNet.py:651-665): a1 = sigmoid(scale*(<q2,k1>-<q1,k2>)) per head of 32."""

    @staticmethod
    def forward(ctx, qkv1, qkv2, scale: float):
        _check_act(qkv1)
        C = qkv1.shape[-1] // 3
        rows = qkv1.numel() // (3 * C)
        out = torch.empty((*qkv1.shape[:-1], C), dtype=qkv1.dtype, device=qkv1.device)
        call("ctu_pwa_fwd", dcode(qkv1.dtype), ptr(qkv1), ptr(qkv2), ptr(out), rows, C, scale, stream())
        ctx.save_for_backward(qkv1, qkv2)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, gout):
        qkv1, qkv2 = ctx.saved_tensors
        gout = gout.contiguous()
        C = qkv1.shape[-1] // 3
        rows = qkv1.numel() // (3 * C)
        g1, g2 = torch.empty_like(qkv1), torch.empty_like(qkv2)
        call("ctu_pwa_bwd", dcode(qkv1.dtype), ptr(qkv1), ptr(qkv2), ptr(gout), ptr(g1), ptr(g2), rows, C, ctx.scale,
             stream())
        return g1, g2, None


def pwa(qkv1, qkv2, scale):
    return PwaFn.apply(qkv1, qkv2, scale)


class PixelShuffleFn(torch.autograd.Function):
    """'b (c p1 p2 p3) h w f -> b (h p1) (w p2) (f p3) c' in channels-last form (hybrid_CTUNet.py:420-428)."""

    @staticmethod
    def forward(ctx, x, factor):
        _check_act(x)
        B, D, H, W, Cbig = x.shape
        p1, p2, p3 = factor
        c = Cbig // (p1 * p2 * p3)
        y = torch.empty((B, D * p1, H * p2, W * p3, c), dtype=x.dtype, device=x.device)
        call("ctu_pixel_shuffle", dcode(x.dtype), ptr(x), ptr(y), B, D, H, W, c, p1, p2, p3, 0, stream())
        ctx.cfg = (B, D, H, W, c, p1, p2, p3)
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        B, D, H, W, c, p1, p2, p3 = ctx.cfg
        gx = torch.empty((B, D, H, W, c * p1 * p2 * p3), dtype=gy.dtype, device=gy.device)
        call("ctu_pixel_shuffle", dcode(gy.dtype), ptr(gy), ptr(gx), B, D, H, W, c, p1, p2, p3, 1, stream())
        return gx, None


def pixel_shuffle(x, factor):
    return PixelShuffleFn.apply(x, tuple(factor))


def patchify(x, p1, p2, p3):
    """x [B, H, W, F] (single channel) -> tokens [B, (H/p1)(W/p2)(F/p3), p1*p2*p3]  (vit.py:115); no gradient."""
    _check_act(x)
    B, H, W, Fr = x.shape
    tok = torch.empty((B, (H // p1) * (W // p2) * (Fr // p3), p1 * p2 * p3), dtype=x.dtype, device=x.device)
    call("ctu_patchify", dcode(x.dtype), ptr(x), ptr(tok), B, H, W, Fr, p1, p2, p3, stream())
    return tok


# ---------------------------------------------------------------------------------------------------------------
# classification heads: per-voxel Linear(C -> n_cls <= 16) + bias, logits computed into 16-column rows
# ---------------------------------------------------------------------------------------------------------------
LOGIT_PAD = 16
_head_state = {}      # id(weight) -> {"ref", dtype -> [version key, w16 [16][K], wT16 [K][16]], "b16", "gw16", "gb16", built}
_padded_grads = {}    # data_ptr of a [.., 16] gradient buffer whose pad columns are zero (train.DiceCEFn.backward) -> True


def note_padded_grad(buf: torch.Tensor):
    """The producer of a logits gradient (the fused DiceCE backward) announces a channels-last buffer with LOGIT_PAD
    columns whose pad columns are written as zeros: HeadFn.backward then multiplies the buffer as it stands."""
    if len(_padded_grads) > 64:   # announcements nobody collected (a loss on logits that are not a head's)
        _padded_grads.clear()
    _padded_grads[buf.data_ptr()] = True


def _head_panels(weight, bias, dtype):
    """Zero-padded panels of a head: W [16][K] (forward) and W^T [K][16] (data gradient) in the activation dtype, bias
    fp32 [16].  The buffers persist (pad rows / columns stay zero for ever); after a parameter update only the 14 real
    rows are rewritten: one cast and one permute launch per head and step, no fills."""
    n_cls, K = weight.shape[0], weight[0].numel()
    st = _head_state.get(id(weight))
    if st is None or st["ref"]() is not weight:
        wid = id(weight)
        st = _head_state[wid] = {"ref": weakref.ref(weight, lambda _r, wid=wid: _head_state.pop(wid, None))}
    ver = (weight._version, bias._version if bias is not None else -1, _weights_epoch, weight.data_ptr())
    ent = st.get(dtype)
    if ent is None:
        dev = weight.device
        ent = st[dtype] = [None, torch.zeros((LOGIT_PAD, K), dtype=dtype, device=dev),
                           torch.zeros((K, LOGIT_PAD), dtype=dtype, device=dev),
                           torch.zeros(LOGIT_PAD, dtype=torch.float32, device=dev), None]
    if ent[0] != ver or torch.cuda.is_current_stream_capturing():
        with torch.no_grad():
            w = weight.detach().reshape(n_cls, K)
            call("ctu_cast", ptr(w), dcode(torch.float32), ptr(ent[1]), dcode(dtype), n_cls * K, stream())
            permute3(w, ent[2], (K, n_cls, 1), (1, K, 0), (LOGIT_PAD, 1, 0))
            if bias is not None:
                call("ctu_cast", ptr(bias.detach()), dcode(torch.float32), ptr(ent[3]), dcode(torch.float32), n_cls, stream())
        ent[0] = ver
        ent[4] = _Built(weight.device)
    else:
        ent[4].fence()
    return ent[1], ent[2], ent[3]


def _head_scratch(weight, K):
    """Persistent zeroed fp32 [16][K] + [16] accumulators of a head's weight / bias gradient (direct-sink path)."""
    st = _head_state[id(weight)]
    sc = st.get("scratch")
    if sc is None:
        sc = st["scratch"] = (torch.zeros((LOGIT_PAD, K), dtype=torch.float32, device=weight.device),
                              torch.zeros(LOGIT_PAD, dtype=torch.float32, device=weight.device))
    return sc


class HeadFn(torch.autograd.Function):
    """UnetOutBlock (1x1x1 conv with bias, hybrid_CTUNet.py:781-783,810) / DecoderLinear.head (:679,685): logits
    [B, D, H, W, 16] with the first n_cls columns meaningful, returned as the [B, n_cls, D, H, W] view the caller's loss
    indexes.  Forward and backward stay in the padded layout: no slicing, padding or re-layout pass anywhere."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _check_act(x)
        n_cls, K = weight.shape[0], weight[0].numel()
        if n_cls > LOGIT_PAD:
            raise NotImplementedError(f"out_channels > {LOGIT_PAD} not supported by the logits/loss kernels")
        M = x.numel() // K
        w16, _, b16 = _head_panels(weight, bias, x.dtype)
        out = torch.empty((*x.shape[:-1], LOGIT_PAD), dtype=x.dtype, device=x.device)
        _plain_gemm(x, w16, out, M, K, LOGIT_PAD, bias=b16 if bias is not None else None)
        ctx.save_for_backward(x, weight, bias)
        sink_expect(weight, ctx.needs_input_grad[1])
        sink_expect(bias, bias is not None and ctx.needs_input_grad[2])
        return out[..., :n_cls].permute(0, 4, 1, 2, 3)

    @staticmethod
    def backward(ctx, g):
        x, weight, bias = ctx.saved_tensors
        n_cls, K = weight.shape[0], weight[0].numel()
        M = x.numel() // K
        B, _, D, H, W = g.shape
        st = g.stride()
        padded = (g.dtype == x.dtype and st[1] == 1 and st[4] == LOGIT_PAD and st[3] == W * LOGIT_PAD and
                  st[2] == H * W * LOGIT_PAD and st[0] == D * H * W * LOGIT_PAD and
                  _padded_grads.pop(g.data_ptr(), None) is not None)
        if padded:
            g16 = torch.as_strided(g, (B, D, H, W, LOGIT_PAD), (st[0], st[2], st[3], st[4], 1))
        else:  # a foreign gradient: re-lay it out (plumbing; the fused loss never takes this path)
            g16 = torch.zeros((B, D, H, W, LOGIT_PAD), dtype=x.dtype, device=x.device)
            g16[..., :n_cls] = g.permute(0, 2, 3, 4, 1)
        _, wT16, _ = _head_panels(weight, bias, x.dtype)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            _plain_gemm(g16, wT16, gx, M, LOGIT_PAD, K)            # dX = dY16 @ W16
        want_gb = bias is not None and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] or want_gb:
            gw_buf, gw_done = _direct_grad(weight) if ctx.needs_input_grad[1] else (None, None)
            gb_buf, gb_done = _direct_grad(bias) if want_gb else (None, None)
            direct = (gw_buf is not None or not ctx.needs_input_grad[1]) and (gb_buf is not None or not want_gb)
            if direct:
                gw16, gb16 = _head_scratch(weight, K)              # zero on entry, handed back zeroed below
            else:
                gw16 = torch.zeros((LOGIT_PAD, K), dtype=torch.float32, device=x.device)
                gb16 = torch.zeros(LOGIT_PAD, dtype=torch.float32, device=x.device)
            if ctx.needs_input_grad[1]:
                _igemm_tn(g16, LOGIT_PAD, x, None, gw16, _plain_geom(M, K, LOGIT_PAD), bias_grad=gb16 if want_gb else None)
            else:
                call("ctu_colsum", dcode(g16.dtype), ptr(g16), None, M, LOGIT_PAD, LOGIT_PAD, ptr(gb16), stream())
            if direct:
                if gw_buf is not None:
                    permute3(gw16, gw_buf, (n_cls * K, 1, 1), (1, 0, 0), (1, 0, 0), accumulate=2)
                    gw_done()
                if gb_buf is not None:
                    permute3(gb16, gb_buf, (n_cls, 1, 1), (1, 0, 0), (1, 0, 0), accumulate=2)
                    gb_done()
            else:
                gw = gw16[:n_cls].view(weight.shape) if ctx.needs_input_grad[1] else None
                gb = gb16[:n_cls] if want_gb else None
        return gx, gw, gb


def head(x, weight, bias):
    return HeadFn.apply(x, weight, bias)


def cast(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    if x.dtype == dtype:
        return x
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    call("ctu_cast", ptr(x), dcode(x.dtype), ptr(out), dcode(dtype), x.numel(), stream())
    return out
