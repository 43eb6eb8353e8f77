"""Caller-side counterparts of the reference's training step (trainer_CTUNet.py:76-133, main_CTUNet.py:156-204):
fused DiceCE with on-device deep-supervision targets, flat fused AdamW, and a one-process-per-GPU data-parallel
wrapper that all-reduces gradient buckets over RCCL on a side stream while backward is still running.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops
from ._lib import call, dcode, ptr, require_device, stream
import ctypes as C
import weakref

# ---------------------------------------------------------------------------------------------------------------
# DiceCE
# ---------------------------------------------------------------------------------------------------------------
_idx_cache: Dict[Tuple, torch.Tensor] = {}


def zoom_nearest_index(n_in: int, n_out: int) -> np.ndarray:
    """Source index of scipy.ndimage.zoom(order=0) along one axis (trainer_CTUNet.py:93-94):
    floor(i*(n_in-1)/(n_out-1) + 0.5).  NOT a stride-2/4 subsample (96->48 jumps by 3 at i=24)."""
    if n_out == 1:
        return np.zeros(1, dtype=np.int32)
    zoom = float(n_in - 1) / float(n_out - 1)
    coord = np.arange(n_out, dtype=np.float64) * zoom
    idx = np.floor(coord + 0.5).astype(np.int32)
    idx[(coord < 0) | (coord > n_in - 1)] = -1  # scipy: out of range -> cval 0 (e.g. last index of 48->24)
    return idx


def _index_map(n_in: int, n_out: int, device) -> torch.Tensor:
    key = (n_in, n_out, str(device))
    t = _idx_cache.get(key)
    if t is None:
        t = torch.from_numpy(zoom_nearest_index(n_in, n_out)).to(device)
        _idx_cache[key] = t
    return t


class DiceCEFn(torch.autograd.Function):
    """weight * DiceCELoss(to_onehot_y, softmax, squared_pred, smooth_nr, smooth_dr)(logits, zoom(target)).
    logits: the [B, n_cls, D, H, W] view a model head returns (channels-last storage padded to 16 columns) or any
    tensor of that shape; target: float class ids [B, 1, LD, LH, LW] at full resolution."""

    @staticmethod
    def forward(ctx, logits, target, weight: float, smooth_nr: float, smooth_dr: float):
        require_device(logits)
        require_device(target)
        B, n_cls, D, H, W = logits.shape
        st = logits.stride()
        padded = (st[1] == 1 and st[4] in (8, 16) and st[4] >= n_cls and st[3] == W * st[4] and st[2] == H * W * st[4]
                  and st[0] == D * H * W * st[4] and logits.dtype in (torch.float32, torch.bfloat16))
        if padded:
            ldl = st[4]
            buf = logits
        else:  # repack into the padded channels-last layout (plumbing; the model heads never take this path)
            ldl = 16
            dt = logits.dtype if logits.dtype in (torch.float32, torch.bfloat16) else torch.float32
            tmp = torch.zeros((B, D, H, W, ldl), dtype=dt, device=logits.device)
            tmp[..., :n_cls] = logits.permute(0, 2, 3, 4, 1)
            buf = tmp
        tgt = target.detach()
        if tgt.dtype != torch.float32:
            tgt = tgt.float()
        tgt = tgt.contiguous()
        LD, LH, LW = tgt.shape[2:]
        dev = logits.device
        idx = (_index_map(LD, D, dev), _index_map(LH, H, dev), _index_map(LW, W, dev))
        acc = torch.zeros(B * n_cls * 3 + 1, dtype=torch.float32, device=dev)
        loss = torch.zeros(1, dtype=torch.float32, device=dev)
        dc = dcode(buf.dtype)
        call("ctu_dicece_fwd", dc, ptr(buf), ldl, ptr(tgt), ptr(idx[0]), ptr(idx[1]), ptr(idx[2]), B, D, H, W, LD, LH,
             LW, n_cls, ptr(acc), stream())
        call("ctu_dicece_finalize", ptr(acc), B, n_cls, D * H * W, smooth_nr, smooth_dr, weight, ptr(loss), stream())
        ctx.save_for_backward(buf, tgt, acc, *idx)
        ctx.cfg = (B, n_cls, D, H, W, LD, LH, LW, ldl, weight, smooth_nr, smooth_dr, padded, logits.dtype)
        return loss[0]

    @staticmethod
    def backward(ctx, gloss):
        buf, tgt, acc, i0, i1, i2 = ctx.saved_tensors
        B, n_cls, D, H, W, LD, LH, LW, ldl, weight, nr, dr, padded, in_dtype = ctx.cfg
        gscale = gloss.detach().float().reshape(1).contiguous()
        dl = torch.empty((B, D, H, W, ldl), dtype=buf.dtype, device=buf.device)
        call("ctu_dicece_bwd", dcode(buf.dtype), ptr(buf), ldl, ptr(tgt), ptr(i0), ptr(i1), ptr(i2), B, D, H, W, LD, LH,
             LW, n_cls, ptr(acc), nr, dr, weight, ptr(gscale), ptr(dl), stream())
        g = dl[..., :n_cls].permute(0, 4, 1, 2, 3)
        if ldl == ops.LOGIT_PAD:
            ops.note_padded_grad(dl)   # pad columns are zero: the head's backward multiplies the buffer as it stands
        if not padded:
            g = g.to(in_dtype)
        return g, None, None, None, None


def dice_ce_loss(logits, target, weight: float = 1.0, smooth_nr: float = 0.0, smooth_dr: float = 1e-6):
    return DiceCEFn.apply(logits, target, float(weight), float(smooth_nr), float(smooth_dr))


def ctunet_loss(outputs, target, smooth_nr=0.0, smooth_dr=1e-6):
    """trainer_CTUNet.py:92-103: L(o1a,t) + 0.5*(L(o1b,t/2) + 0.5*L(o1c,t/4)) + 0.5*(L(o2a,t) + L(o2b,t)); the
    nearest-neighbour target downsampling happens inside the loss kernel (no host round trip)."""
    (o1a, o1b, o1c), (o2a, o2b) = outputs
    a = (smooth_nr, smooth_dr)
    return (dice_ce_loss(o1a, target, 1.0, *a) + dice_ce_loss(o1b, target, 0.5, *a) + dice_ce_loss(o1c, target, 0.25, *a)
            + dice_ce_loss(o2a, target, 0.5, *a) + dice_ce_loss(o2b, target, 0.5, *a))


def cunet_loss(outputs, target, smooth_nr=0.0, smooth_dr=1e-6):
    """trainer_CUNet.py:91-100."""
    o0, o1, o2 = outputs
    a = (smooth_nr, smooth_dr)
    return dice_ce_loss(o0, target, 1.0, *a) + dice_ce_loss(o1, target, 0.5, *a) + dice_ce_loss(o2, target, 0.25, *a)


def tunet_loss(outputs, target, smooth_nr=0.0, smooth_dr=1e-6):
    """trainer_TUNet.py:80-82."""
    o0, o1 = outputs
    return dice_ce_loss(o0, target, 1.0, smooth_nr, smooth_dr) + dice_ce_loss(o1, target, 1.0, smooth_nr, smooth_dr)


LOSSES = {"ctunet": ctunet_loss, "cunet": cunet_loss, "tunet": tunet_loss}


# ---------------------------------------------------------------------------------------------------------------
# flat parameter / gradient storage + fused AdamW
# ---------------------------------------------------------------------------------------------------------------
class FlatParams:
    """Re-homes a model's parameters and gradients into two flat fp32 buffers (64-element aligned segments) so the
    optimizer is one kernel and gradient all-reduce works on contiguous slices.  ``order`` lets the caller lay
    parameters out in expected gradient-ready order (buckets then become ready front to back)."""

    ALIGN = 64

    def __init__(self, params: Iterable[nn.Parameter]):
        self.params: List[nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        for p in self.params:
            # One owner per parameter: a second FlatParams would leave the first one's hooks and gradient sinks registered,
            # and both would fold every gradient (2x, 3x, ... the true value).  Share the existing one instead
            # (FlatParams.of(params), DataParallel(...).flat, FusedAdamW(..., flat=...)) or release() it first.
            if getattr(p, "_ctu_flat", None) is not None and p._ctu_flat() is not None:
                raise RuntimeError("parameter already belongs to a FlatParams: pass that object (`flat=`) instead of "
                                   "building a second one, or call its release() first")
        dev = self.params[0].device
        self.offsets = []
        off = 0
        for p in self.params:
            if p.dtype != torch.float32:
                raise TypeError("master parameters must be float32")
            self.offsets.append(off)
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.total = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                self.flat[o:o + p.numel()].copy_(p.detach().reshape(-1))
                p.data = self.flat[o:o + p.numel()].view(p.shape)
                p.grad = self.grad[o:o + p.numel()].view(p.shape)
        self.touched = [False] * len(self.params)
        self.always_touched = set()   # parameters whose gradients are written by replayed graphs (graphs.graph_stages)
        self.listeners = []  # callables(i): "the gradient of parameter i is complete for this backward"
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(i)) for i, p in enumerate(self.params)]
        me = weakref.ref(self)
        for p in self.params:
            p._ctu_flat = me
        if self.flat.is_cuda:
            # let weight-gradient kernels accumulate straight into the flat buffer (no zero-filled temporaries, no adds)
            for i, p in enumerate(self.params):
                ops.register_grad_sink(p, self._make_sink(i))
        ops.bump_weights_epoch()

    @staticmethod
    def of(params: Iterable[nn.Parameter]) -> Optional["FlatParams"]:
        """The FlatParams that already owns exactly these trainable parameters (any order), or None."""
        ps = [p for p in params if p.requires_grad]
        owners = {id(o): o for o in (getattr(p, "_ctu_flat", None) and p._ctu_flat() for p in ps) if o is not None}
        if len(owners) != 1:
            return None
        owner = next(iter(owners.values()))
        return owner if {id(p) for p in ps} == {id(p) for p in owner.params} else None

    def release(self):
        """Give the parameters up: hooks and gradient sinks are removed (the parameters keep their storage in the flat
        buffers, which stay alive through them)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.params:
            ops.unregister_grad_sink(p)
            p._ctu_flat = None
        self.listeners = []

    def _ready(self, i):
        self.touched[i] = True
        for fn in self.listeners:
            fn(i)

    def _make_sink(self, i):
        return lambda p: self._ready(i)

    def _make_hook(self, i):
        def hook(p):
            # .grad is not the flat view: the caller reset it (`param.grad = None`, trainer_CTUNet.py:88-89, or
            # torch.optim's zero_grad()) and autograd installed a fresh tensor holding everything accumulated since that
            # reset.  COPY it in (the slice still holds the previous step's gradient - adding would accumulate across
            # steps) and re-point .grad at the slice, so later accumulations of this backward land in place.
            g = p.grad
            o = self.offsets[i]
            view = self.grad[o:o + p.numel()].view(p.shape)
            if g is not None and g.data_ptr() != view.data_ptr():
                view.copy_(g)
                p.grad = view
            self._ready(i)
        return hook

    def zero_grad(self):
        """Equivalent of `param.grad = None` (trainer_CTUNet.py:88-89) without giving up the flat views."""
        self.grad.zero_()
        self.touched = [False] * len(self.params)
        ops.reset_grad_sink_counts()
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)

    def untouched_ranges(self) -> List[Tuple[int, int]]:
        out = []
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            if not self.touched[i] and i not in self.always_touched:
                end = o + (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
                if out and out[-1][1] == o:
                    out[-1] = (out[-1][0], end)
                else:
                    out.append((o, end))
        return out


class FusedAdamW:
    """torch.optim.AdamW(lr, betas, eps, weight_decay) semantics (main_CTUNet.py:192-193) as ONE kernel over the flat
    buffers.  Parameters whose gradient was never produced in this step (7 ResBlock.conv3 tensors in CTUNet,
    SURVEY.md section 8a row D) are skipped exactly like torch skips `grad is None`."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, flat: Optional[FlatParams] = None,
                 capturable: bool = False, overlap: bool = False, bucket_mb: float = 64.0):
        """capturable: keep the learning rate, the step count and the bias corrections in four device words (advanced by
        a one-thread kernel in front of the update), so that step() can be captured into a HIP graph and replayed
        (GraphedStep); change the learning rate with set_lr().

        overlap: update a bucket of parameters (~bucket_mb of the flat buffer, which lies in gradient-ready order) as soon
        as the backward pass has finished ITS gradients, on a stream of its own - the update is pure HBM traffic and runs
        under the matrix-bound kernels of the rest of the backward pass; step() then only updates what is left and joins.
        Same arithmetic, element for element.  Not for a loop that inspects or rescales gradients between backward() and
        step() (GradScaler.unscale_/step, gradient clipping): there the update has to wait for all of them - leave it
        off.  With DataParallel (world > 1) the reducer drives it: DataParallel(..., optimizer=opt) updates a bucket
        behind its all-reduce, on the communication stream."""
        if flat is None:
            params = list(params)
            flat = FlatParams.of(params) or FlatParams(params)   # e.g. the one DataParallel(model) built
        self.flat = flat
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.m = torch.zeros_like(self.flat.flat)
        self.v = torch.zeros_like(self.flat.flat)
        self.step_count = 0
        self.param_groups = [{"lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay}]
        self._static_skip = None
        # bf16 copy of the parameters, kept current by the AdamW kernel itself: the bf16 GEMMs read layer weights from it
        self.mirror = torch.empty(self.flat.total, dtype=torch.bfloat16, device=self.flat.flat.device) \
            if self.flat.flat.is_cuda else None
        self._mirror_stamp = None
        self.sync_mirror()
        self.capturable = bool(capturable) and self.flat.flat.is_cuda
        self.hyper = None
        if self.capturable:
            self.hyper = torch.zeros(4, dtype=torch.float32, device=self.flat.flat.device)
            self.hyper[0] = lr
        # ---- overlapped update (see the docstring)
        self._done: List[Tuple[int, int]] = []   # flat ranges already updated in the step that is open
        self._open = False                       # step_count already advanced for this iteration
        self._ov = None
        self.reducer_driven = False              # DataParallel(optimizer=self) calls update_range() itself
        self.overlap_enabled = True              # (a harness may switch the per-bucket updates off for single-stream profiles)
        if overlap and self.flat.flat.is_cuda and not self.capturable:
            f = self.flat
            target = int(bucket_mb * 1024 * 1024 / 4)
            buckets, begin, members = [], 0, []
            for i in range(len(f.params)):
                members.append(i)
                end = f.offsets[i + 1] if i + 1 < len(f.params) else f.total
                if end - begin >= target or i + 1 == len(f.params):
                    buckets.append((begin, end, members))
                    begin, members = end, []
            self._ov = {"buckets": buckets, "of": {i: b for b, (_, _, mem) in enumerate(buckets) for i in mem},
                        "pending": [len(mem) for _, _, mem in buckets], "seen": [False] * len(f.params),
                        # The updates are queued on the model's BRANCH stream (ops.side_stream: CTUNet's second encoder stream), not
                        # on a stream of their own: in line with the ViT branch's backward, beside the main stream.  A dedicated
                        # stream is 0.15 ms per step faster while HIP happens to map it onto a hardware queue it shares with
                        # another stream (the default four queues) - and 19 ms slower when it gets a queue of its own
                        # (GPU_MAX_HW_QUEUES >= 5: 66 ms per step, profiles/r03_bench_hw_queues_sweep.log): the 5.2 GB of
                        # update traffic then run truly beside the backward pass.
                        "stream": ops.side_stream(f.flat.device), "home": torch.cuda.current_stream()}
            f.listeners.append(self._on_ready)

    def set_lr(self, lr: float):
        self.lr = self.param_groups[0]["lr"] = float(lr)
        if self.hyper is not None:
            self.hyper[0:1].fill_(float(lr))

    def device_step_count(self) -> int:
        """Steps taken, read back from the device in capturable mode (replayed graphs advance it without the host)."""
        if self.hyper is None:
            return self.step_count
        return int(self.hyper[1:2].view(torch.int32).item())

    def sync_mirror(self):
        """Full fp32 -> bf16 refresh + (re)registration of the per-parameter views; needed only after the parameters
        were modified by something other than step() (load_state_dict, a broadcast into the flat buffer, ...)."""
        if self.mirror is None:
            return
        f = self.flat
        self.mirror.copy_(f.flat)
        for p, o in zip(f.params, f.offsets):
            ops.register_bf16_mirror(p, self.mirror[o:o + p.numel()])
        self._mirror_stamp = (ops.mirror_generation(), tuple(p._version for p in f.params))

    def zero_grad(self, set_to_none: bool = False):
        self.flat.zero_grad()

    def state_dict(self, params: Optional[Iterable[nn.Parameter]] = None):
        """Optimizer state.  With `params` (the parameters in the order a torch.optim.AdamW would have received them,
        e.g. model.parameters()): torch.optim.AdamW's own format - `state` {index: step, exp_avg, exp_avg_sq} +
        `param_groups` - which the reference's tools and torch.optim.AdamW.load_state_dict read
        (trainer_CTUNet.py:311-312 saves exactly that).  Without: the flat form (step count, hyper-parameters, the two
        moment buffers in FlatParams order with the shapes that define the order)."""
        step = self.device_step_count()
        if params is None:
            return {"step": step, "param_groups": [dict(g) for g in self.param_groups],
                    "shapes": [tuple(p.shape) for p in self.flat.params], "offsets": list(self.flat.offsets),
                    "exp_avg": self.m.detach().cpu(), "exp_avg_sq": self.v.detach().cpu()}
        where = {id(p): (o, p.numel()) for p, o in zip(self.flat.params, self.flat.offsets)}
        skipped = self._static_skip if self._static_skip is not None else []
        state, idx = {}, []
        for i, p in enumerate(params):
            idx.append(i)
            if id(p) not in where:
                continue
            o, n = where[id(p)]
            if step == 0 or any(a <= o < b for a, b in skipped):
                continue   # torch keeps no state for a parameter that never saw a gradient
            state[i] = {"step": torch.tensor(float(step)), "exp_avg": self.m[o:o + n].view(p.shape).detach().cpu().clone(),
                        "exp_avg_sq": self.v[o:o + n].view(p.shape).detach().cpu().clone()}
        g0 = self.param_groups[0]
        group = {"lr": g0["lr"], "betas": tuple(g0["betas"]), "eps": g0["eps"], "weight_decay": g0["weight_decay"],
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "decoupled_weight_decay": True, "params": idx}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd, params: Optional[Iterable[nn.Parameter]] = None):
        """Accepts the flat form and torch.optim.AdamW's form (a checkpoint written by the reference,
        trainer_CTUNet.py:311-312); the latter needs `params` = the parameters in the order that optimizer was built
        with (model.parameters() in main_CTUNet.py:192)."""
        if "state" in sd:
            if params is None:
                raise ValueError("a torch.optim.AdamW state dict is indexed by parameter position: pass params=model.parameters()")
            where = {id(p): (o, p.numel()) for p, o in zip(self.flat.params, self.flat.offsets)}
            plist = list(params)
            index_of = {}
            for g in sd["param_groups"]:
                for pos in g["params"]:
                    index_of[pos] = plist[pos] if pos < len(plist) else None
            self.m.zero_()
            self.v.zero_()
            steps = []
            for pos, st in sd["state"].items():
                p = index_of.get(int(pos))
                if p is None or id(p) not in where:
                    raise ValueError(f"optimizer state entry {pos} has no matching parameter")
                o, n = where[id(p)]
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"optimizer state entry {pos}: shape {tuple(st['exp_avg'].shape)} != {tuple(p.shape)}")
                self.m[o:o + n].copy_(st["exp_avg"].reshape(-1))
                self.v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                steps.append(int(float(st["step"])))
            if steps and min(steps) != max(steps):
                raise ValueError("per-parameter step counts differ: the fused optimizer keeps one")
            self.step_count = steps[0] if steps else 0
            g0 = sd["param_groups"][0]
            self.param_groups = [{"lr": g0["lr"], "betas": tuple(g0["betas"]), "eps": g0["eps"],
                                  "weight_decay": g0["weight_decay"]}]
        else:
            if [tuple(s) for s in sd["shapes"]] != [tuple(p.shape) for p in self.flat.params]:
                raise ValueError("optimizer state was saved for a different parameter order / model")
            self.step_count = int(sd["step"])
            self.param_groups = [dict(g) for g in sd["param_groups"]]
            self.m.copy_(sd["exp_avg"])
            self.v.copy_(sd["exp_avg_sq"])
        if self.hyper is not None:
            self.hyper[1:2].view(torch.int32).fill_(self.step_count)
        g0 = self.param_groups[0]
        self.lr, self.betas, self.eps, self.weight_decay = g0["lr"], tuple(g0["betas"]), g0["eps"], g0["weight_decay"]
        if self.hyper is not None:
            self.hyper[0:1].fill_(float(self.lr))

    def freeze_skip_ranges(self):
        """After one real backward: remember which parameters never receive a gradient, so later steps (e.g. replayed
        from a HIP graph, where Python hooks do not run) skip the same ranges."""
        self._static_skip = self.flat.untouched_ranges()
        return self._static_skip

    def _on_ready(self, i):
        """FlatParams listener (overlap=True, no reducer): the gradient of parameter i is final for this backward."""
        ov = self._ov
        if ov is None or self.reducer_driven or not self.overlap_enabled or ov["seen"][i]:
            return
        ov["seen"][i] = True
        b = ov["of"][i]
        ov["pending"][b] -= 1
        if ov["pending"][b] > 0:
            return
        begin, end, _ = ov["buckets"][b]
        # The update stream has to see: the kernel that wrote the last gradient (queued on the stream this callback runs
        # under - a weight-gradient companion stream, usually), every other side stream (earlier members of the bucket), and
        # the compute streams themselves - the data-gradient GEMM of a layer reads W (its bf16 mirror, which the update
        # rewrites) and is queued before the layer reports its weight gradient.
        st = ov["stream"]
        st.wait_stream(torch.cuda.current_stream())
        st.wait_stream(ov["home"])
        for s2 in ops.side_streams(self.flat.flat.device):
            st.wait_stream(s2)
        with torch.cuda.stream(st):
            self.update_range(begin, end)

    def update_range(self, begin: int, end: int, skip: Sequence[Tuple[int, int]] = ()):
        """AdamW on flat[begin:end) on torch's current stream; the first call of an iteration opens the step (advances
        the step count every range of this iteration shares).  step() updates whatever no call has covered."""
        if not self._open:
            self.step_count += 1
            self._open = True
        if len(skip) > 16:
            raise RuntimeError(f"{len(skip)} disjoint gradient-less parameter ranges; the fused AdamW handles up to 16")
        arr = (C.c_int64 * (2 * max(1, len(skip))))()
        for k, (a, b) in enumerate(skip):
            arr[2 * k], arr[2 * k + 1] = a - begin, b - begin
        f = self.flat
        lr = self.param_groups[0]["lr"]
        off = lambda t, es: (t.data_ptr() + es * begin) if t is not None else None
        call("ctu_adamw", off(f.flat, 4), off(f.grad, 4), off(self.m, 4), off(self.v, 4), off(self.mirror, 2), end - begin, lr,
             self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step_count, arr, len(skip), ptr(self.hyper), stream())
        self._done.append((begin, end))

    def step(self):
        if self.flat.flat.is_cuda:
            ops.join_side_streams()   # weight gradients written from the ViT branch's stream
        f = self.flat
        skip = self._static_skip if self._static_skip is not None else f.untouched_ranges()
        if self.hyper is not None:
            if not self._open:
                self.step_count += 1
                self._open = True
            call("ctu_adamw_tick", ptr(self.hyper), self.betas[0], self.betas[1], stream())
        # what the overlapped updates (update_range) have not covered yet: everything, without them
        pos = 0
        for a, b in sorted(self._done) + [(f.total, f.total)]:
            if a > pos:
                self.update_range(pos, a, [(max(x, pos), min(y, a)) for x, y in skip if x < a and y > pos])
            pos = max(pos, b)
        if self._ov is not None:
            ov = self._ov
            ov["home"] = torch.cuda.current_stream()
            ov["home"].wait_stream(ov["stream"])
            ov["pending"] = [len(mem) for _, _, mem in ov["buckets"]]
            ov["seen"] = [False] * len(f.params)
        self._done = []
        self._open = False
        ops.bump_weights_epoch()
        if self.mirror is not None and self._mirror_stamp != (ops.mirror_generation(), tuple(p._version for p in f.params)):
            self.sync_mirror()  # someone else wrote parameters since the last sync (grad-less ones would stay stale)


# ---------------------------------------------------------------------------------------------------------------
# whole-step HIP graph
# ---------------------------------------------------------------------------------------------------------------
class GraphedStep:
    """One training step - zero grads, forward, loss, backward, optimizer - captured into ONE HIP graph and replayed.

    A step is ~2 300 kernel launches on three to five streams; enqueued from Python it costs the host ~40 ms, and wherever
    the launching thread is busy with one stream the others starve.  The replayed graph hands the device the whole
    dependency DAG at once.  Requirements: static input tensors (copy new batches into them), FusedAdamW(capturable=True),
    no host synchronisation inside `fn`.  `fn` returns the loss tensor (static as well: read it after a replay).  The
    constructor runs `warmup` eager steps, then records one more call of `fn` (recording executes nothing).

    Measured on ROCm 7.2 / MI355X (CTUNet d101, B=2): replay 58.2 ms per step against 49.3 ms for the eager step on its four
    streams - the graph executor runs the captured branches one after the other, i.e. replay gives back exactly what the
    two-branch / weight-gradient stream overlap gained.  bench.py therefore enqueues eagerly unless --graph is given; the
    class stays for hosts whose launch thread is the limiter (host enqueue 40 ms per step here).

    Workspaces: the capture starts a new workspace epoch (ops.new_workspace_epoch), so every per-stream workspace it uses
    is created - and zero-filled - inside the graph; each replay therefore starts from the state the capture started from."""

    def __init__(self, fn, optimizer: Optional["FusedAdamW"] = None, warmup: int = 2, stream=None):
        if optimizer is not None and not optimizer.capturable:
            raise ValueError("GraphedStep needs FusedAdamW(capturable=True)")
        self.fn = fn
        # Eager warm-up steps run on the very stream the capture will use: autograd's per-parameter AccumulateGrad nodes and
        # every cache built on first use (index maps, weight-panel job tables, side streams) then belong to that stream.
        self.stream = stream if stream is not None else torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            for i in range(max(1, warmup)):
                out = fn()
                if optimizer is not None and optimizer._static_skip is None:
                    optimizer.freeze_skip_ranges()   # which parameters never get a gradient: fixed from here on
                del out
        torch.cuda.synchronize()
        ops.new_workspace_epoch()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):
            self.out = fn()
        torch.cuda.synchronize()

    def __call__(self):
        self.graph.replay()
        return self.out


# ---------------------------------------------------------------------------------------------------------------
# data parallel: one process per GPU, bucketed gradient all-reduce over RCCL/xGMI overlapped with backward
# ---------------------------------------------------------------------------------------------------------------
class DataParallel(nn.Module):
    """Replacement for DistributedDataParallel(find_unused_parameters=True) at main_CTUNet.py:187-189.

    Gradients live in one flat buffer (FlatParams) laid out in *expected ready order*; the buffer is cut into
    buckets of ~bucket_mb.  A post-accumulate hook counts arrivals per bucket; when a bucket is complete its slice is
    all-reduced (SUM, pre-scaled by 1/world) on a side stream fenced by events, so communication overlaps the rest of
    backward.  `finish()` (called by the optimizer wrapper / step function) flushes buckets that hold never-touched
    parameters and makes the compute stream wait for the side stream.  Exposes `.module` like DDP
    (trainer_CTUNet.py:309 unwraps it).  Works with any backend: "nccl" (= RCCL) on GPUs, "gloo" in CPU tests."""

    def __init__(self, module: nn.Module, flat: Optional[FlatParams] = None, bucket_mb: float = 32.0,
                 process_group=None, ready_order: Optional[Sequence[nn.Parameter]] = None, broadcast: bool = True,
                 payload: str = "fp32", comm=None, optimizer: Optional["FusedAdamW"] = None, static_unused: bool = False):
        """optimizer: a FusedAdamW over the same FlatParams; every bucket is then UPDATED right behind its all-reduce, on the
        communication stream, while the backward pass goes on (FusedAdamW(overlap=...) explains when that is allowed);
        optimizer.step() afterwards only covers buckets flushed by finish().
        static_unused: the parameters that produced no gradient in the FIRST backward never will (DDP's static_graph; true of
        the reference models: the seven ResBlock.conv3 of CTUNet with in == out channels are built and never called).  From
        the second step on a bucket then goes out as soon as its USED members are ready instead of waiting for finish() -
        in CTUNet the three buckets that finish first (110 MB) hold such parameters and would otherwise be reduced behind
        the backward pass, exposed.  A gradient reported for one of them later raises."""
        super().__init__()
        self._static_unused = bool(static_unused)
        self._unused = None   # set of parameter indices, learnt in the first finish()
        if payload not in ("fp32", "bf16"):
            raise ValueError("payload must be 'fp32' or 'bf16'")
        self.module = module
        self.pg = process_group
        self.payload = payload
        self._comm = comm   # comm.Communicator (RCCL through the C ABI) for the bf16 payload path
        if payload == "bf16" and comm is None:
            raise ValueError("payload='bf16' needs a comm.Communicator (hybrid_ctunet_amd.comm.Communicator.from_torch())")
        self.world = dist.get_world_size(self.pg) if dist.is_initialized() else 1
        # rehearsal on a one-GPU box (bench.py CTU_BENCH_FORCE_DP=1 CTU_DP_REDUCE_AT_WORLD1=1): run every bucket's collective at
        # world 1 as well, so that the event fences, RCCL's stream and finish() carry the real choreography
        self._alone = self.world == 1 and dist.is_initialized() and bool(os.environ.get("CTU_DP_REDUCE_AT_WORLD1"))
        params = list(ready_order) if ready_order is not None else [p for p in module.parameters() if p.requires_grad]
        if flat is None:
            flat = FlatParams.of(params) or FlatParams(params)
        self.flat = flat
        f = self.flat
        # bucket boundaries on parameter boundaries
        target = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets: List[Tuple[int, int, List[int]]] = []  # (begin, end, param indices)
        begin, members = 0, []
        for i, (p, o) in enumerate(zip(f.params, f.offsets)):
            members.append(i)
            end = f.offsets[i + 1] if i + 1 < len(f.params) else f.total
            if end - begin >= target or i + 1 == len(f.params):
                self.buckets.append((begin, end, members))
                begin, members = end, []
        self._bucket_of = {}
        for b, (_, _, mem) in enumerate(self.buckets):
            for i in mem:
                self._bucket_of[i] = b
        self._pending = [len(mem) for _, _, mem in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._works = []
        self._is_cuda = f.flat.is_cuda
        self._wgrad_stream_before = None
        if (self.world > 1 or self._alone) and self._is_cuda and not os.environ.get("CTU_WGRAD_STREAM"):
            # HIP streams run on hardware queues that share FOUR hardware pipes (queue id mod 4), and a queue parked on a barrier
            # packet - a stream waiting for an event - holds its pipe against the other queues on it.  The step alone uses four
            # streams (main, branch, two weight-gradient companions: queues 4 - 7, one pipe each); the exchange stream is the
            # fifth and lands on the pipe of one of them, where its bucket fences (one wait per bucket, each for work queued deep
            # in the backward pass) stall that stream: 65 ms per step on main's pipe, 66 on the branch's, 74 / 83 on a companion's,
            # against 47.6 with the companions off - and moved k queues along the pattern repeats with period 4
            # (profiles/r04_five_streams.log).  So under data parallelism the step keeps to the pipes that are left: main, branch,
            # exchange (+ RCCL's own stream); the weight gradients stay on their layer's stream.  release() puts the switch back.
            self._wgrad_stream_before = ops.WGRAD_STREAM
            ops.WGRAD_STREAM = False
        # measurement only (profiles/r04_five_streams.log): CTU_DP_QUEUE_SHIFT=k creates k throw-away streams first, so the exchange
        # stream is handed the hardware queue k places further on
        self._dummies = []
        if self._is_cuda:
            for _ in range(int(os.environ.get("CTU_DP_QUEUE_SHIFT", "0"))):
                st = torch.cuda.Stream()
                with torch.cuda.stream(st):
                    torch.zeros(1, device=f.flat.device)
                self._dummies.append(st)
        self._side = torch.cuda.Stream() if self._is_cuda else None
        self._seen = [False] * len(f.params)
        self._opt = None
        self._flushing = False
        if optimizer is not None:
            self.attach_optimizer(optimizer)
        self._home = torch.cuda.current_stream() if self._is_cuda else None
        self.exposed = None   # a list: finish() appends an event pair around the compute stream's wait for the exchange
        f.listeners.append(self._on_ready)  # fires for autograd-accumulated and for directly accumulated gradients
        if broadcast and self.world > 1:
            dist.broadcast(f.flat, src=0, group=self.pg)  # DDP ctor semantics: rank 0's parameters win
            ops.bump_weights_epoch()
            ops.invalidate_bf16_mirrors()

    def forward(self, *a, **kw):
        return self.module(*a, **kw)

    def attach_optimizer(self, optimizer: "FusedAdamW"):
        """Update every bucket right behind its all-reduce (see the constructor); a no-op for one rank or on the CPU, where
        the optimizer's own overlap (FusedAdamW(overlap=True)) or its plain step() applies."""
        if optimizer.flat is not self.flat:
            raise ValueError("optimizer and DataParallel must share one FlatParams")
        if self.world > 1 and self._is_cuda and not optimizer.capturable:
            self._opt = optimizer
            optimizer.reducer_driven = True

    def _on_ready(self, i):
        # One report per parameter and backward: autograd's AccumulateGrad node runs once however often the parameter is
        # used (the engine sums the contributions first), and the direct gradient sinks report only after the LAST of the
        # uses counted in forward (ops.sink_expect / sink_done).  A duplicate report is ignored all the same.
        if self._seen[i]:
            if self._unused is not None and i in self._unused:
                raise RuntimeError(f"DataParallel(static_unused=True): parameter {i} produced no gradient in the first step "
                                   "but does now; its bucket may already have been reduced")
            return
        self._seen[i] = True
        b = self._bucket_of[i]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def _launch(self, b):
        if self._launched[b] or (self.world == 1 and not self._alone):
            self._launched[b] = True
            return
        self._launched[b] = True
        begin, end, _ = self.buckets[b]
        sl = self.flat.grad[begin:end]
        if self._is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._side.wait_event(ev)
            for st in ops.side_streams(sl.device):   # members of this bucket may come from the other branch's stream
                self._side.wait_stream(st)
            if self._opt is not None and not self._flushing:
                self._side.wait_stream(self._home)   # data-gradient GEMMs of these layers read the weights the update rewrites
            with torch.cuda.stream(self._side):
                self._reduce_cuda(sl)
                if self._opt is not None and not self._flushing:  # (a flushed bucket holds gradient-less parameters: step() skips them)
                    # static_unused: this bucket may hold members that never receive a gradient; torch.optim.AdamW leaves
                    # `grad is None` parameters untouched (no weight decay, no state) and so does the fused update
                    self._opt.update_range(begin, end, self._unused_ranges(b))
        else:
            sl.mul_(1.0 / self.world)   # gloo (CPU tests) has no AVG
            self._works.append(dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def _unused_ranges(self, b) -> List[Tuple[int, int]]:
        """Aligned flat ranges of the never-used members of bucket b (merged where adjacent)."""
        if not self._unused:
            return []
        f, out = self.flat, []
        for i in self.buckets[b][2]:
            if i in self._unused:
                o = f.offsets[i]
                end = o + (f.params[i].numel() + f.ALIGN - 1) // f.ALIGN * f.ALIGN
                if out and out[-1][1] == o:
                    out[-1] = (out[-1][0], end)
                else:
                    out.append((o, end))
        return out

    def _reduce_cuda(self, sl):
        """Mean of one bucket slice over the ranks, on the side stream.  payload "fp32": one RCCL all-reduce with the
        average folded into the collective (ncclAvg: no scaling pass).  payload "bf16": the slice is rounded to bf16
        once, every rank receives the other ranks' chunk of it (reduce-scatter as an all-to-all, one message per peer:
        all 7 xGMI links at once), sums its chunk in fp32, and the bf16 means are all-gathered - half the bytes of the
        fp32 ring, accumulation still fp32 (csrc/comm.hip via ctu_allreduce_bucket)."""
        backend = dist.get_backend(self.pg)
        if self.payload == "bf16" and self._comm is not None:
            self._comm.allreduce_mean_bf16(sl)
        elif backend == "nccl":
            dist.all_reduce(sl, op=dist.ReduceOp.AVG, group=self.pg)
        else:  # gloo moving device buffers (one-GPU rehearsals)
            sl.mul_(1.0 / self.world)
            dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.pg)

    def release(self):
        """Detach from the gradient buffer and give back what the constructor changed process-wide (the weight-gradient
        companion streams): a model that goes on training without the wrapper gets its four-stream schedule back."""
        if self._on_ready in self.flat.listeners:
            self.flat.listeners.remove(self._on_ready)
        if self._wgrad_stream_before is not None:
            ops.WGRAD_STREAM = self._wgrad_stream_before
            self._wgrad_stream_before = None

    def finish(self):
        """Call after backward, before the optimizer step."""
        if self._is_cuda:
            ops.join_side_streams()
        self._flushing = True
        for b in range(len(self.buckets)):
            if not self._launched[b]:
                self._launch(b)  # buckets containing parameters that never produced a gradient
        self._flushing = False
        if self._is_cuda:
            self._home = torch.cuda.current_stream()
        for w in self._works:
            w.wait()
        self._works = []
        if self._is_cuda and (self.world > 1 or self._alone):
            cur = torch.cuda.current_stream()
            if self.exposed is not None:   # measurement: how long the compute stream stands waiting for the exchange
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                cur.wait_stream(self._side)
                e1.record(cur)
                self.exposed.append((e0, e1))
            else:
                cur.wait_stream(self._side)
        if self._static_unused and self._unused is None:
            self._unused = {i for i, s in enumerate(self._seen) if not s}
        unused = self._unused or ()
        self._pending = [sum(1 for i in mem if i not in unused) for _, _, mem in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._seen = [i in unused for i in range(len(self.flat.params))]
        ops.reset_grad_sink_counts()


def gradient_ready_order(model: nn.Module) -> List[nn.Parameter]:
    """Parameters in the order autograd finishes them for CTUNet/CUNet/TUNet as built here: res heads -> res decoders
    0..3 -> vit heads/decoder -> vit_encoder -> vit_encoder0 -> vit -> convnet (deep to shallow).  (The reference builds
    the ViT branch before the ResNet and therefore finishes it last; CTUNet.forward here swaps the two
    independent encoders so that the parameter-heavy ViT trunk is not the tail of the backward pass.)  Modules not present
    are skipped; anything unlisted goes last."""
    prefixes = ["res_out_24x24", "res_out_48x48", "res_out", "res_decoder0", "res_decoder1", "res_decoder2",
                "res_decoder3", "decoder_linear_96x96", "vit_out", "vit_decoder0", "vit_encoder.", "vit_encoder0", "vit.",
                "convnet.layer4", "convnet.layer3", "convnet.layer2", "convnet.layer1", "convnet"]
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    taken, out = set(), []
    for pre in prefixes:
        for n, p in reversed(named):
            if n not in taken and n.startswith(pre):
                taken.add(n)
                out.append(p)
    out += [p for n, p in named if n not in taken]
    return out
