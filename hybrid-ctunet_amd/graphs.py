"""HIP graphs for the launch-latency-bound stages of a training step.

The small-volume stages (ResNet layer3 / layer4 at 12x12x24 and 6x6x12, the 864-token ViT trunk, the first window-attention
stages) consist of hundreds of 5-30 us kernels: enqueued from Python (~17 us of host time per launch) the launch thread, not
the device, sets their pace, and while it feeds one stream the other streams of the step starve.  Each such stage is
captured ONCE - forward and backward separately, torch.cuda.make_graphed_callables - and from then on costs two graph
launches per step.  The large-volume stages stay eager: they are device-bound, and they are where the two encoder
branches and the weight-gradient kernels overlap on several streams (a replayed graph executes its nodes in one order).

Inside a capture
  * weight panels are packed inside the graph (ops._packed rebuilds instead of trusting its per-step cache: nobody would
    refresh a cached panel on replay),
  * weight-gradient kernels stay on the stage's own stream (no companion stream to join),
  * per-stream workspaces belong to the capture (created, i.e. zero-filled, inside it).
Weight gradients of a graphed stage go straight into the flat gradient buffer (train.FlatParams sinks, static addresses);
Python-side "gradient ready" notifications do not fire on replay, so the stage's parameters are declared always-touched."""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence

import torch
import torch.nn as nn

from . import ops


class _Seq(nn.Module):
    """Detached container (NOT registered in the model tree: state_dict keys stay the reference's) running blocks in order."""

    def __init__(self, blocks: Sequence[nn.Module]):
        super().__init__()
        self.blocks = nn.ModuleList(list(blocks))

    def forward(self, x):
        for b in self.blocks:
            x = b(x)
        return x


class GraphedStage:
    """`module(x)` for one tensor argument, replayed from graphs when x has the captured shape / dtype, in training mode and
    with gradients enabled; the original forward otherwise (inference batches, eval mode, other precisions)."""

    def __init__(self, module: nn.Module, sample: torch.Tensor):
        self.module = module
        self.eager_forward = module.forward
        self.shape, self.dtype = tuple(sample.shape), sample.dtype
        wg = ops.WGRAD_STREAM
        ops.WGRAD_STREAM = False
        try:
            torch.cuda.make_graphed_callables(module, (sample,), allow_unused_input=True)
        finally:
            ops.WGRAD_STREAM = wg
        self.graphed_forward = module.forward   # (patched by make_graphed_callables)
        module.forward = self.__call__

    def __call__(self, x, *a, **kw):
        if (a or kw or tuple(x.shape) != self.shape or x.dtype != self.dtype or not self.module.training
                or not torch.is_grad_enabled() or not x.requires_grad):
            return self.eager_forward(x, *a, **kw)
        return self.graphed_forward(x)

    def release(self):
        self.module.forward = self.eager_forward


DEFAULT_STAGES = ("convnet.layer3", "convnet.layer4", "vit.transformer", "vit_encoder.layers.0", "vit_encoder.layers.1")


def graph_stages(model: nn.Module, x_in: torch.Tensor, stages: Iterable[str] = DEFAULT_STAGES,
                 autocast_dtype: Optional[torch.dtype] = torch.bfloat16, flat=None) -> List[GraphedStage]:
    """Capture the named stages of `model` (those it has) for inputs shaped like the ones a forward of `x_in` feeds them.
    Call after the parameters have their final storage (train.FlatParams) and after one eager step; pass `flat` so the
    stages' parameters count as always-touched for the optimizer's "no gradient this step" logic.  No autograd graph of
    an earlier step may be alive (drop the last loss tensor first): its AccumulateGrad nodes are bound to the streams of
    that step, and the engine would tie them into the capture (ROCm 7.2 then crashes in hipStreamEndCapture)."""
    import gc
    gc.collect()
    # The capture's warm-up passes run backward with uninitialised upstream gradients; whoever listens to "gradient ready"
    # (FusedAdamW(overlap=True) would UPDATE the weights from them, DataParallel would all-reduce them and keep the buckets
    # marked as launched) must not hear them: the owners' listeners are detached for the duration of the capture.
    owners = {}
    for p in model.parameters():
        o = getattr(p, "_ctu_flat", None)
        o = o() if o is not None else None
        if o is not None:
            owners[id(o)] = o
    if flat is not None:
        owners[id(flat)] = flat
    saved_listeners = {k: o.listeners for k, o in owners.items()}
    for o in owners.values():
        o.listeners = []
    try:
        return _graph_stages(model, x_in, stages, autocast_dtype, flat)
    finally:
        for k, o in owners.items():
            o.listeners = saved_listeners[k]
        ops.reset_grad_sink_counts()


def _graph_stages(model, x_in, stages, autocast_dtype, flat):
    mods = dict(model.named_modules())
    targets: Dict[str, nn.Module] = {}
    installs = {}
    firsts = {}
    for name in stages:
        m = mods.get(name)
        if m is None:
            continue
        owner_name, _, attr = name.rpartition(".")
        owner = mods[owner_name] if owner_name else model
        firsts[name] = m
        if isinstance(m, nn.ModuleList):      # vit.transformer: a list of blocks run in order by ViT.forward
            firsts[name] = m[0]
            m = _Seq(m)
            installs[name] = (owner, "_graphed_" + attr)
        elif owner_name.endswith(".layers") and hasattr(mods[owner_name.rpartition(".")[0]], "stage_runner"):
            up = mods[owner_name.rpartition(".")[0]]          # UpAttentionBlock: stage <attr> of its pyramid
            firsts[name] = m[0][1]
            m = up.stage_runner(int(attr))
            installs[name] = (up, "_graphed_stage_" + attr)
        targets[name] = m
    samples: Dict[str, torch.Tensor] = {}
    hooks = []
    for name, m in targets.items():
        hooks.append(firsts[name].register_forward_pre_hook(
            lambda mod, args, name=name: samples.setdefault(name, args[0].detach().clone())))
    was_training = model.training
    model.train()
    try:
        with torch.no_grad():
            if autocast_dtype is not None:
                with torch.autocast("cuda", dtype=autocast_dtype):
                    model(x_in)
            else:
                model(x_in)
    finally:
        for h in hooks:
            h.remove()
    torch.cuda.synchronize()
    out = []
    for name, m in targets.items():
        if name not in samples:
            continue
        st = GraphedStage(m, samples[name].requires_grad_(True))
        if name in installs:                      # the owner looks the stage up under this attribute before looping itself
            object.__setattr__(installs[name][0], installs[name][1], st)
        out.append(st)
        if flat is not None:
            ids = {id(p) for p in m.parameters()}
            flat.always_touched.update(i for i, p in enumerate(flat.params) if id(p) in ids)
    torch.cuda.synchronize()
    if flat is not None:
        flat.zero_grad()   # the capture's warm-up passes accumulated into the flat gradient buffer
    model.train(was_training)
    return out
