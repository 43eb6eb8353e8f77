"""Module-level autograd Functions replayed from C-side launch lists (csrc/plan.hip, _plan.py).

The per-op path (ops.py) crosses the Python / C boundary once per kernel and builds one autograd node per op: ~17 us of
host time per launch, 38 ms per 48 ms step.  Here one autograd node covers a whole block of the reference -
`Bottleneck.forward` (networks/resnet.py:106-126), `ResBlock.forward` (networks/hybrid_CTUNet.py:93-105), ... - and its
forward / backward are launch lists recorded on first use per (shapes, routing) and replayed by one `ctu_plan_run` call.
The kernels, their arguments and their order are those of the per-op path (tests compare the two); what disappears is
the host work between them.

Only the production routing is covered: bf16, channels-last, the LDS-DMA GEMM / halo kernels, fused InstanceNorm
statistics.  Anything else (fp32 parity mode, the A/B switches that pin fallback kernels, a HIP-graph capture, the
per-launch profiler of bench.py) takes the per-op path: `usable()` decides.

Intermediates of a block are plain byte buffers whose sizes are fixed at record time; the block's output is its own tensor.  Workspaces with state (InstanceNorm accumulators, alternating backward sums) are private to the
fused path per (device, stream), so the two paths can be mixed freely within a step.
"""
from __future__ import annotations

import os
import weakref
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib as L
from . import ops
from ._lib import Epilogue, Geom
from ._plan import Recorder, Ref

BF16 = L.CTU_BF16
ENABLED = not os.environ.get("CTU_NO_PLANS")
# A/B switches for measurements (CTU_OPT="gelu2=0,acc=0": comma-separated name=0/1; read once at import, part of every plan key)
OPT = {"gelu2": 1,    # GELU backward inside the data-gradient GEMM of the Linear behind it (ctu_epilogue.act = 2)
       "acc": 1,      # InstanceNorm finalize folded into the apply kernel (ctu_in_apply_acc)
       "wparam": 1,   # halo weight gradients reduced straight into the parameter layout (ctu_conv3_halo_wgrad_param)
       "s2": 1,       # stride-2 data gradients of the stage transitions: 3x3x3 on the halo kernel over the zero-upsampled dY,
                      # 1x1x1 as a plain GEMM over the output rows + ctu_add_strided2 (instead of the generic implicit GEMM)
       "nogres": 1,   # blocks with a conv + norm shortcut: that norm's backward reads the block's gradient and sign mask itself
                      # instead of a copy with the LeakyReLU slope applied, written by the main norm's backward (one tensor pass)
       "dual": 0,     # blocks with a conv + norm shortcut: the block's last norm and the shortcut's norm applied by ONE kernel
                      # (ctu_in_apply_dual) - the normalised shortcut tensor is never written and read back.  Built and tested (10
                      # launches and 0.9 GB per step fewer; 45.08 / 45.25 against 45.32 / 45.25 ms per step: inside the noise).  Off:
                      # its results differ from the two-launch form by one bf16 ulp in 0.003 % of the elements (tools/dualdbg.py), and
                      # CUNet-101 amplifies that into a 1e-4 shift of the coarsest head's Dice term - across the gate that the
                      # whole-model bf16 test holds at twice the reference's own bf16 error (profiles/r04_bench_ab_dual_norm.log)
       "ff1": 1,      # FeedForward forward of the 128-wide stages as ONE kernel (ctu_ff_fwd: LayerNorm, both products and GELU fused;
                      # the backward pass re-derives LayerNorm(x) - the operand of W1's weight gradient - with one LayerNorm launch)
       "pwa1": 1,     # pixelweight_attention.forward of the 128-wide stages WITHOUT autograd (inference) as ONE kernel (ctu_pwa_block_fwd,
                      # nothing saved): 191 us against 480 - 520 for the six launches per 442 368-row call.  Training keeps the six
                      # launches: with the projections saved the kernel takes 395 - 415 us and the backward pass would still have to
                      # re-derive both LayerNorms (2 x 45 us) - no gain without a backward kernel that recomputes the projections
                      # (profiles/r04_pwa_block_fwd.txt)
       "in1": 0}      # InstanceNorm backward of tensors up to IN_FUSED_BYTES as ONE launch (ctu_in_bwd_fused: reduce, meet at a counter, apply).
                      # Built, parity-tested and measured SLOWER: 21 - 57 us against 13 - 43 for the pair in kbench, 47.4 - 47.8 against
                      # 45.8 - 45.9 ms per step (profiles/r04_experiment_in_bwd_one_launch.log) - two launches of one stream pipeline, a
                      # grid-wide meet (returning atomics, counter poll, sums read past L2) is a 10 us chain.  Off.
for _kv in filter(None, os.environ.get("CTU_OPT", "").split(",")):
    _k, _, _v = _kv.partition("=")
    if _k not in OPT:
        raise ValueError(f"CTU_OPT: unknown switch {_k!r}")
    OPT[_k] = int(_v or 1)
_cache: Dict[tuple, object] = {}

_EPI_OFF = {n: getattr(Epilogue, n).offset for n in ("bias", "residual", "out2", "splitk_ws", "in_acc", "pre_out")}


def clear_cache():
    _cache.clear()


def usable(x: torch.Tensor) -> bool:
    return (ENABLED and x.is_cuda and x.dtype == torch.bfloat16 and L.PROFILER is None and ops.USE_HALO_CONV
            and ops.USE_W_KN and ops.FUSE_IN_STATS and ops.SIGN_MASK and x.is_contiguous()
            and not torch.cuda.is_current_stream_capturing())


def _flags():
    # (autograd on / off is part of every plan key: a forward recorded without it writes nothing a backward pass would read)
    return (ops.B16_LAYOUT, ops.WGRAD_PARTIALS, ops.WGRAD_STREAM, tuple(OPT.values()), torch.is_grad_enabled())


# ---------------------------------------------------------------------------------------------------------------
# per-(device, stream) state of the fused path
# ---------------------------------------------------------------------------------------------------------------
class _WS:
    __slots__ = ("acc", "apar", "adirty", "sums", "par", "dirty_n", "insync", "acc2", "apar2", "adirty2")

    def __init__(self, device):
        # InstanceNorm forward statistics: two fp64 accumulators alternate from norm to norm - a producer's epilogue sums into
        # the clean one, the norm's apply kernel reads it and zeroes the other (ctu_in_apply_acc: no finalize launch)
        self.acc = [torch.zeros(1 << 14, dtype=torch.float64, device=device) for _ in range(2)]
        self.apar = 0       # index of the clean accumulator
        self.adirty = 0     # entries the last norm left dirty in the other one
        # statistics of shortcut convolutions (ctu_in_apply_dual): a pair of their own, alternating from one dual norm to the next -
        # the shortcut's sums have to survive the block's other norms, which go through `acc`
        self.acc2 = [torch.zeros(1 << 14, dtype=torch.float64, device=device) for _ in range(2)]
        self.apar2 = 0
        self.adirty2 = 0
        self.sums = [torch.zeros(1 << 14, dtype=torch.float64, device=device) for _ in range(2)]  # IN backward sums, same scheme
        self.par = 0
        self.dirty_n = 0
        self.insync = torch.zeros(128, dtype=torch.int32, device=device)   # arrival / departure counters of ctu_in_bwd_fused


_FWS: Dict[tuple, _WS] = {}
# (mean, rstd) tables of a plan sit 32 KiB apart and the fp64 accumulators hold 1 << 14 entries: a block whose widest norm has
# more than 8192 / 2 (batch item, channel) pairs takes the per-op path, whose workspaces grow on demand (bottleneck_ok, resblock_ok)
STATS_MAX = 8192
IN_FUSED_BYTES = 32 << 20   # norms whose tensor is larger keep the reduce + apply pair (byte-bound: 8 192 workgroups stream better than 256)


def _fws(device, sid) -> _WS:
    key = (device, sid, ops._ws_epoch)
    w = _FWS.get(key)
    if w is None:
        w = _FWS[key] = _WS(device)
    return w


class _Bufs:
    """The intermediates of one pass.  Buffers of equal size share one allocation (the twelve `h` tensors of the ViT trunk are
    one [12 x size] tensor): a handful of allocator calls and record_stream marks per pass instead of one per buffer, in
    sizes that repeat from step to step so torch's caching allocator serves them from its free lists.  (ONE arena per
    pass was measured at 200 us per allocation: a size nothing else shares gets carved out of, and merged back into,
    differently sized cached blocks every step.)"""

    def __init__(self, prefix: str):
        self.prefix = prefix
        self.names: List[str] = []
        self.sizes: List[int] = []
        self._groups = None

    def add(self, name: str, nbytes: int) -> str:
        assert self._groups is None
        self.names.append(name)
        self.sizes.append((max(int(nbytes), 16) + 255) & ~255)
        return name

    def remove(self, name: str) -> int:
        i = self.names.index(name)
        self.names.pop(i)
        return self.sizes.pop(i)

    def groups(self):
        """[(slot name, element size, [buffer names])], by descending size."""
        if self._groups is None:
            by = {}
            for n, s in zip(self.names, self.sizes):
                by.setdefault(s, []).append(n)
            self._groups = [(f"{self.prefix}{k}", s, by[s]) for k, s in enumerate(sorted(by, reverse=True))]
        return self._groups

    def slot_names(self):
        return [g[0] for g in self.groups()]

    def bind(self, R: Recorder):
        for slot, size, names in self.groups():
            for k, n in enumerate(names):
                R.alias(n, R[slot] + k * size)

    def alloc(self, device):
        return [torch.empty(size * len(names), dtype=torch.uint8, device=device) for _, size, names in self.groups()]


# ---------------------------------------------------------------------------------------------------------------
# emitters: the calls ops.py makes for one layer, written against a Recorder (pointers are Refs)
# ---------------------------------------------------------------------------------------------------------------
def _epi(ldc, bias=None, residual=None, act=0, out2=None, n_split=0, ldc2=0, splitk_ws=None, splitk=1, w_kn=0,
         in_acc=None, in_rows=0, pre_out=None) -> Epilogue:
    e = Epilogue()
    e.act, e.ldc, e.n_split, e.ldc2, e.scatter = act, ldc, n_split, ldc2, 0
    e.w_kn, e.in_rows = w_kn, in_rows
    e.splitk = splitk if splitk_ws is not None else 1
    e._refs = {_EPI_OFF["bias"]: bias, _EPI_OFF["residual"]: residual, _EPI_OFF["out2"]: out2,
               _EPI_OFF["splitk_ws"]: splitk_ws if splitk > 1 else None, _EPI_OFF["in_acc"]: in_acc, _EPI_OFF["pre_out"]: pre_out}
    return e


class _Need:
    """Sizes of shared workspaces a plan relies on (checked / grown per call, their pointers are slots)."""

    def __init__(self):
        self.skws = 0       # split-K workspace floats
        self.panel = 0      # multi-tap weight-gradient panel floats (on the weight-gradient stream)


def em_gemm(R: Recorder, need: _Need, x, w, out, M, K, N, *, bias=None, residual=None, act=0, w_kn=0, in_acc=None,
            in_rows=0, pre_out=None, stream=0):
    """ops._plain_gemm: out[M,N] = act(x[M,K] @ w^T + bias) + residual."""
    sk = 1 if (in_acc is not None or pre_out is not None or act != 0) else ops._splitk_for(M, N, K, dma=K % 32 == 0)
    if sk > 1:
        need.skws = max(need.skws, M * N)
    e = _epi(N, bias=bias, residual=residual, act=act, splitk_ws=R["skws"] if sk > 1 else None, splitk=sk, w_kn=w_kn,
             in_acc=in_acc, in_rows=in_rows, pre_out=pre_out)
    R.call("ctu_igemm_nt", BF16, x, None, w, out, ops._plain_geom(M, K, N), e, stream=stream)


class _InFwd:
    """ops.InstanceNormFn.forward for the k-th norm of a forward plan.  `acc()` is the accumulator the norm's producer sums
    into from its epilogue; the apply kernel derives (mean, rstd) from it and zeroes the other accumulator (the previous
    norm's).  A producer without fused sums gets a statistics pass over its output instead."""

    def __init__(self, R: Recorder):
        self.R, self.k, self.prev = R, 0, None

    def acc(self):
        return self.R["ia0"] if self.k % 2 == 0 else self.R["ia1"]

    def emit(self, y, stats, out, B, S, C, *, fused: bool, residual=None, act=1, b16=0, mask=None):
        R = self.R
        acc, other = (R["ia0"], R["ia1"]) if self.k % 2 == 0 else (R["ia1"], R["ia0"])
        clear_n = R.ival("iadirty") if self.k == 0 else self.prev
        if not fused:
            R.call("ctu_in_stats", BF16, y, B, S, C, acc, stats)     # (leaves its accumulator zeroed)
        elif not OPT["acc"]:
            R.call("ctu_in_finalize", B, S, C, acc, stats)           # (A/B: the separate finalize launch; zeroes acc)
            fused = False
        R.call("ctu_in_apply_acc", BF16, y, acc if fused else None, stats, residual, out, B, S, C, int(act), int(b16), mask,
               other, clear_n)
        self.k += 1
        self.prev = B * C * 2
        if self.prev > STATS_MAX:
            raise RuntimeError(f"fused path: {B} x {C} InstanceNorm statistics exceed the fixed tables (the *_ok guards route such shapes to the per-op path)")


def _emit_dual(nf, y, stats, y2, stats2, out, B, S, C, *, fused: bool, fused2: bool, act=1, mask=None):
    """The block's last norm (k-th of the plan, producer sums in nf.acc() when `fused`) together with the shortcut's norm, whose
    producer summed into R["ib0"] when `fused2` (else stats2 already holds (mean, rstd)).  Returns what the shortcut accumulators'
    bookkeeping needs: 1 = ib0 was used and is dirty now, 0 = untouched (ib1 was cleared either way)."""
    R = nf.R
    acc, other = (R["ia0"], R["ia1"]) if nf.k % 2 == 0 else (R["ia1"], R["ia0"])
    clear_n = R.ival("iadirty") if nf.k == 0 else nf.prev
    if not fused:
        R.call("ctu_in_stats", BF16, y, B, S, C, acc, stats)
    R.call("ctu_in_apply_dual", BF16, y, acc if fused else None, stats, y2, R["ib0"] if fused2 else None, stats2, out, B, S, C, int(act),
           mask, other, clear_n, R["ib1"], R.ival("ibdirty"))
    nf.k += 1
    nf.prev = B * C * 2
    if nf.prev > STATS_MAX:
        raise RuntimeError(f"fused path: {B} x {C} InstanceNorm statistics exceed the fixed tables")
    return 1 if fused2 else 0


class _InBwd:
    """ops.InstanceNormFn.backward for the k-th norm of a backward plan: the two sums buffers alternate."""

    def __init__(self, R: Recorder):
        self.R, self.k, self.prev = R, 0, None

    def emit(self, gy, x, y, stats, gx, gres, B, S, C, act, dx_b16=0, mask=None):
        R = self.R
        sums, clear = (R["s0"], R["s1"]) if self.k % 2 == 0 else (R["s1"], R["s0"])
        clear_n = R.ival("dirty0") if self.k == 0 else self.prev
        if OPT["in1"] and B * S * C * 2 <= IN_FUSED_BYTES and B <= 64:
            R.call("ctu_in_bwd_fused", BF16, gy, x, y, stats, sums, gx, gres, B, S, C, int(act), clear, clear_n, int(dx_b16), mask,
                   R["insync"])
        else:
            R.call("ctu_in_bwd_reduce", BF16, gy, x, y, stats, sums, B, S, C, int(act), mask)
            R.call("ctu_in_bwd_apply", BF16, gy, x, y, stats, sums, gx, gres, B, S, C, int(act), clear, clear_n, int(dx_b16), mask)
        self.k += 1
        self.prev = B * C * 2
        if self.prev > STATS_MAX:
            raise RuntimeError(f"fused path: {B} x {C} InstanceNorm statistics exceed the fixed tables (the *_ok guards route such shapes to the per-op path)")


class ConvSpec:
    """Static description of one convolution of a block and which kernel family serves it."""

    def __init__(self, B, din, C1, C2, N, k, stride, padding):
        self.B, self.din, self.C1, self.C2, self.N = B, tuple(din), C1, C2, N
        self.k, self.stride, self.padding = tuple(k), tuple(stride), tuple(padding)
        self.K = C1 + C2
        self.taps = k[0] * k[1] * k[2]
        self.dout = tuple((n + 2 * p - kk) // s + 1 for n, p, kk, s in zip(din, padding, k, stride))
        self.Mi = B * din[0] * din[1] * din[2]
        self.Mo = B * self.dout[0] * self.dout[1] * self.dout[2]
        self.So = self.Mo // B
        if self.taps == 1 and self.stride == (1, 1, 1) and C2 == 0:
            self.kind = "lin"
        elif ops._halo_ok(self.k, self.stride, self.padding) and C1 % 32 == 0 and C2 % 32 == 0:
            self.kind = "halo"
        else:
            self.kind = "gen"
        if self.kind == "halo" and N % 32:
            raise NotImplementedError("fused path: halo convolution with N % 32 != 0")
        big = self.Mi * max(C1, C2, N) < (1 << 31)
        # InstanceNorm sums of the output from the producing kernel's epilogue (else a pass over the output)
        self.fused_stats = (self.kind == "lin" and self.K % 32 == 0 and self.So % 128 == 0) or (self.kind == "halo" and big)
        # may the norm behind this conv hand its input gradient over in CTU_LAYOUT_B16 (this conv's backward is its one reader)
        self.gy_b16 = bool(self.kind == "halo" and ops.B16_LAYOUT and big)
        self.wkn_d = self.kind == "lin" and N % 64 == 0 and self.K % 8 == 0
        # stride-2 data gradients off the generic implicit GEMM (which masks 7 of 8 taps per input voxel: 25 - 47 TFLOP/s)
        self.dgrad_via = None
        even = all(n % 2 == 0 for n in din) and self.stride == (2, 2, 2) and C2 == 0 and OPT["s2"]
        if self.kind == "gen" and even and self.k == (3, 3, 3) and self.padding == (1, 1, 1) and C1 % 32 == 0 and N % 32 == 0 \
                and self.Mi * max(C1, N) < (1 << 31):
            self.dgrad_via = "halo_up2"      # dX = conv3_halo(zero-upsampled dY, flipped W)
        elif self.kind == "gen" and even and self.taps == 1 and self.padding == (0, 0, 0) and C1 % 8 == 0 and N % 32 == 0:
            self.dgrad_via = "gemm_s2"       # compact dX = dY W over the output rows; the caller adds it at the even voxels
            self.wkn_d = N % 64 == 0 and self.K % 8 == 0

    def wants_b16_input(self) -> bool:
        return bool(self.kind == "halo" and ops.B16_LAYOUT and self.C2 == 0 and self.Mi * max(self.C1, self.N) < (1 << 31))


def conv_weights(spec: ConvSpec, weight, fwd=True, dgrad=True):
    """Per call: the operand panels of this convolution (cached per parameter version by ops), forward and data gradient."""
    N, K, taps = spec.N, spec.K, spec.taps
    dt = torch.bfloat16
    wf = wd = None
    if spec.kind == "lin":
        wf = ops._linear_weight(weight, weight.reshape(N, K), dt)
        if dgrad:
            wd = wf if spec.wkn_d else ops._packed(weight, "lin_d", dt, lambda: weight.detach().reshape(N, K).t().to(dt).contiguous())
    elif spec.kind == "halo":
        if fwd:
            wf = ops._packed_frag(weight, "conv_hf", dt, N, K, taps, K * taps, taps, 1, 0)
        if dgrad:
            wd = ops._packed_frag(weight, "conv_hd", dt, K, N, taps, taps, K * taps, 1, 1)
    else:
        if fwd:
            wf = ops._packed(weight, "conv_f", dt, lambda: ops._pack(weight, (taps, N, K), (1, K * taps, taps), dt))
        if dgrad and spec.dgrad_via == "halo_up2":
            wd = ops._packed_frag(weight, "conv_hd", dt, K, N, taps, taps, K * taps, 1, 1)
        elif dgrad and spec.dgrad_via == "gemm_s2":
            wd = ops._linear_weight(weight, weight.reshape(N, K), dt) if spec.wkn_d else \
                ops._packed(weight, "lin_d", dt, lambda: weight.detach().reshape(N, K).t().to(dt).contiguous())
        elif dgrad:
            wd = ops._packed(weight, "conv_d", dt, lambda: ops._pack(weight, (taps, K, N), (1, taps, K * taps), dt))
    return wf, wd


def em_conv_fwd(R, need, spec: ConvSpec, x1, x2, w, out, *, x1_b16=0, acc=None):
    """ops.LinearFn / ops.ConvFn forward of `spec`; returns whether the InstanceNorm sums of `out` were added into `acc`."""
    s = spec
    if s.kind == "lin":
        em_gemm(R, need, x1, w, out, s.Mo, s.K, s.N, in_acc=acc if s.fused_stats else None,
                in_rows=s.So if s.fused_stats else 0)
    elif s.kind == "halo":
        B, (D, H, W) = s.B, s.din
        R.call("ctu_conv3_halo", BF16, x1, x2, w, out, None, B, D, H, W, s.C1, s.C2, s.N, 0, s.N, 0,
               acc if s.fused_stats else None, None, None, R["tnws"], 1 << 24, int(x1_b16))
    else:
        assert not x1_b16
        g = ops._geom(s.B, s.din, s.dout, s.C1, s.C2, s.N, s.k, s.stride, s.padding, 0)
        sk = ops._conv_splitk(s.Mo, s.N, s.K, s.taps)
        if sk > 1:
            need.skws = max(need.skws, s.Mo * s.N)
        R.call("ctu_igemm_nt", BF16, x1, x2, w, out, g, _epi(s.N, splitk=sk, splitk_ws=R["skws"] if sk > 1 else None))
    return s.fused_stats and acc is not None


def em_conv_dgrad(R, need, spec: ConvSpec, gy, wd, g1, g2, *, extra=None, extra2=None, gy_b16=0, up2=None):
    """Data gradient of `spec`: g1 [Mi, C1] (and g2 [Mi, C2]) from gy [Mo, N]; extra / extra2 are added in the epilogue.
    up2: scratch [Mi, N] of a "halo_up2" spec.  A "gemm_s2" spec writes the COMPACT gradient [Mo, C1] into g1 (no extra)."""
    s = spec
    if s.dgrad_via == "halo_up2":
        assert g2 is None and extra2 is None and not gy_b16 and up2 is not None
        B, (D, H, W), (Do, Ho, Wo) = s.B, s.din, s.dout
        R.call("ctu_upsample2_zeros", BF16, gy, up2, B, Do, Ho, Wo, s.N)
        R.call("ctu_conv3_halo", BF16, up2, None, wd, g1, None, B, D, H, W, s.N, 0, s.K, 0, s.C1, 0,
               None, extra, None, R["tnws"], 1 << 24, 0)
        return
    if s.dgrad_via == "gemm_s2":
        assert g2 is None and extra is None and extra2 is None and not gy_b16
        em_gemm(R, need, gy, wd, g1, s.Mo, s.N, s.K, w_kn=1 if s.wkn_d else 0)
        return
    if s.kind == "lin":
        assert g2 is None and extra2 is None
        em_gemm(R, need, gy, wd, g1, s.Mo, s.N, s.K, w_kn=1 if s.wkn_d else 0, residual=extra)
    elif s.kind == "halo":
        B, (D, H, W) = s.B, s.din
        R.call("ctu_conv3_halo", BF16, gy, None, wd, g1, g2, B, D, H, W, s.N, 0, s.K, s.C1 if g2 is not None else 0, s.C1, s.C2,
               None, extra, extra2, R["tnws"], 1 << 24, int(gy_b16))
    else:
        assert not gy_b16 and extra2 is None
        gd = ops._geom(s.B, s.dout, s.din, s.N, 0, s.K, s.k, s.stride, s.padding, 1)
        sk = ops._conv_splitk(s.Mi, s.K, s.N, s.taps) if g2 is None else 1
        if sk > 1:
            if extra is not None:
                raise NotImplementedError("fused path: split-K data gradient with an epilogue residual")
            need.skws = max(need.skws, s.Mi * s.K)
        R.call("ctu_igemm_nt", BF16, gy, None, wd, g1, gd,
               _epi(s.C1, residual=extra, out2=g2, n_split=s.C1 if g2 is not None else 0, ldc2=s.C2, splitk=sk,
                    splitk_ws=R["skws"] if sk > 1 else None))


def em_conv_wgrad(R, need, spec: ConvSpec, gy, x1, x2, gw, *, stream, x1_b16=0, gy_b16=0, up2=None):
    """Weight gradient of `spec` accumulated into gw (fp32, the parameter's layout).  up2: the zero-upsampled dY of a "halo_up2"
    spec (written by em_conv_dgrad): its weight gradient is the stride-1 halo weight gradient of (up2, x1)."""
    s = spec
    if s.dgrad_via == "halo_up2" and up2 is not None and ops.WGRAD_PARTIALS and OPT["wparam"]:
        assert not x1_b16 and not gy_b16 and x2 is None
        B, (D, H, W) = s.B, s.din
        R.call("ctu_conv3_halo_wgrad_param", BF16, up2, x1, None, gw, B, D, H, W, s.C1, 0, s.N, 0, 0,
               R["wgws"], 256 * 54 * 1024 + 27 * 64 * 1024, stream=stream)
        return
    if s.kind == "lin":
        R.call("ctu_igemm_tn", BF16, gy, s.N, x1, None, gw, None, ops._plain_geom(s.Mo, s.K, s.N), R["tnws1"], 1 << 24, stream=stream)
        return
    if s.taps == 1:   # strided 1x1x1 (downsample): the panel IS the parameter layout [N][K]
        gq = ops._geom(s.B, s.din, s.dout, s.C1, s.C2, s.N, s.k, s.stride, s.padding, 0)
        R.call("ctu_igemm_tn", BF16, gy, s.N, x1, x2, gw, None, gq, R["tnws1"], 1 << 24, stream=stream)
        return
    N, K, taps = s.N, s.K, s.taps
    if s.kind == "halo" and ops.WGRAD_PARTIALS and OPT["wparam"]:
        # per-split partial panels summed AND transposed into the parameter layout by one reduce kernel: no panel, no permute
        B, (D, H, W) = s.B, s.din
        R.call("ctu_conv3_halo_wgrad_param", BF16, gy, x1, x2, gw, B, D, H, W, s.C1, s.C2, N, int(x1_b16), int(gy_b16),
               R["wgws"], 256 * 54 * 1024 + 27 * 64 * 1024, stream=stream)
        return
    need.panel = max(need.panel, taps * N * K)
    if s.kind == "halo":
        B, (D, H, W) = s.B, s.din
        R.call("ctu_conv3_halo_wgrad", BF16, gy, x1, x2, R["panel"], B, D, H, W, s.C1, s.C2, N, int(x1_b16), int(gy_b16),
               R["wgws"] if ops.WGRAD_PARTIALS else None, (256 * 54 * 1024 + 27 * 64 * 1024) if ops.WGRAD_PARTIALS else 0,
               stream=stream)
    else:
        gq = ops._geom(s.B, s.din, s.dout, s.C1, s.C2, N, s.k, s.stride, s.padding, 0)
        R.call("ctu_igemm_tn", BF16, gy, N, x1, x2, R["panel"], None, gq, R["tnws1"], 1 << 24, stream=stream)
    # panel [taps][N][K] -> += parameter layout [N][K][taps]; the panel is handed back zeroed
    R.call("ctu_permute3", R["panel"], gw, L.CTU_F32, N, K, taps, K, 1, N * K, K * taps, taps, 1, 2, stream=stream)


# ---------------------------------------------------------------------------------------------------------------
# shared run-time plumbing
# ---------------------------------------------------------------------------------------------------------------
WS_NAMES = ("ia0", "ia1", "iadirty", "tnws", "skws", "ib0", "ib1", "ibdirty")
BWD_WS_NAMES = ("s0", "s1", "dirty0", "tnws", "skws", "tnws1", "panel", "wgws", "insync")


def _fwd_ws_values(device, sid, need: _Need, n_norms: int, last_n: int, dual=None):
    """Slot values of WS_NAMES for one forward replay, and the accumulator bookkeeping of its n_norms norms.  dual: None = the plan
    has no ctu_in_apply_dual, 1 = its shortcut sums went through ib0 (dirty now), 0 = a dual norm that left ib0 untouched."""
    w = _fws(device, sid)
    vals = [w.acc[w.apar].data_ptr(), w.acc[1 - w.apar].data_ptr(), w.adirty, ops._tn_workspace(device).data_ptr(),
            ops._splitk_workspace(device, need.skws).data_ptr(), w.acc2[w.apar2].data_ptr(), w.acc2[1 - w.apar2].data_ptr(), w.adirty2]
    w.apar ^= n_norms & 1
    w.adirty = last_n
    if dual is not None:
        if dual:
            w.apar2 ^= 1
            w.adirty2 = last_n      # (the dual norm is the plan's last norm: same B x C)
        else:
            w.adirty2 = 0
    return vals


def _grad_targets(weights, needs):
    """Per weight: (fp32 buffer the weight gradient is accumulated into, completion callback or None, tensor to return to
    autograd or None).  With a registered gradient sink (train.FlatParams) the buffer is the parameter's persistent .grad."""
    out = []
    for w, need in zip(weights, needs):
        if w is None or not need:
            out.append((None, None, None))
            continue
        buf, done = ops._direct_grad(w)
        if buf is None:
            buf = torch.zeros(w.shape, dtype=torch.float32, device=w.device)
            out.append((buf, None, buf))
        else:
            out.append((buf, done, None))
    return out


class _BwdRun:
    """Streams, workspaces and bookkeeping of one backward replay."""

    def __init__(self, device, targets):
        self.device = device
        cur = torch.cuda.current_stream()
        self.cur = cur
        self.s0 = cur.cuda_stream
        self.direct = all(t[1] is not None or t[0] is None for t in targets)
        self.side = None
        if self.direct and ops.WGRAD_STREAM and any(t[0] is not None for t in targets):
            self.side = ops.side_stream(device, ("wgrad", self.s0))
            ops.ensure_join_after_backward()
        self.s1 = self.side.cuda_stream if self.side is not None else self.s0

    def ws_values(self, need: _Need):
        dev = self.device
        w = _fws(dev, self.s0)
        key1 = (dev, self.s1, ops._ws_epoch)
        vals = [w.sums[w.par].data_ptr(), w.sums[1 - w.par].data_ptr(), w.dirty_n,
                ops._tn_workspace(dev).data_ptr(), ops._splitk_workspace(dev, need.skws).data_ptr(),
                _tn_ws_for(key1).data_ptr(), _panel_for(key1, need.panel).data_ptr(),
                _wgws_for(key1).data_ptr() if ops.WGRAD_PARTIALS else 0, w.insync.data_ptr()]
        return vals, w

    def keep_alive(self, *tensors):
        """Tensors the weight-gradient stream may still read after the caller's scope has dropped them."""
        if self.side is not None:
            for t in tensors:
                if t is not None:
                    t.record_stream(self.side)

    def finish(self, w: _WS, n_norms: int, last_n: int, targets):
        w.par ^= n_norms & 1
        w.dirty_n = last_n
        # "gradient complete", last layer first: the order the flat gradient buffer is laid out in (train.gradient_ready_order)
        if any(t[1] is not None for t in targets):
            if self.side is not None:
                with torch.cuda.stream(self.side):   # reported on the stream that wrote it
                    for t in reversed(targets):
                        if t[1] is not None:
                            t[1]()
            else:
                for t in reversed(targets):
                    if t[1] is not None:
                        t[1]()


def _tn_ws_for(key):
    ws = ops._TN_WS.get(key)
    if ws is None:
        ws = ops._TN_WS[key] = torch.empty(1 << 24, dtype=torch.float32, device=key[0])
    return ws


def _panel_for(key, n):
    t = ops._PANEL_SCRATCH.get(key)
    if t is None or t.numel() < n:
        t = ops._PANEL_SCRATCH[key] = torch.zeros(max(n, 27 * 512 * 512), dtype=torch.float32, device=key[0])
    return t


def _wgws_for(key):
    t = ops._WGRAD_WS.get(key)
    if t is None:
        t = ops._WGRAD_WS[key] = torch.empty(256 * 54 * 1024 + 27 * 64 * 1024, dtype=torch.float32, device=key[0])
    return t


def _wg_stream(R: Recorder, side: bool):
    """Order the weight-gradient stream behind everything queued so far on the compute stream; returns its index."""
    if not side:
        return 0
    ev = R.new_event()
    R.event_record(ev, 0)
    R.stream_wait(1, ev)
    return 1


# ---------------------------------------------------------------------------------------------------------------
# Bottleneck (networks/resnet.py:82-126)
# ---------------------------------------------------------------------------------------------------------------
class _BneckPlan:
    """Everything static about one bottleneck shape: conv specs, arena layouts, the two recorded plans."""

    def __init__(self, B, D, H, W, Cin, P, stride, has_down):
        self.B, self.Cin, self.P, self.N4 = B, Cin, P, 4 * P
        one = (1, 1, 1)
        self.c1 = ConvSpec(B, (D, H, W), Cin, 0, P, one, one, (0, 0, 0))
        self.c2 = ConvSpec(B, (D, H, W), P, 0, P, (3, 3, 3), stride, (1, 1, 1))
        self.c3 = ConvSpec(B, self.c2.dout, P, 0, 4 * P, one, one, (0, 0, 0))
        self.cd = ConvSpec(B, (D, H, W), Cin, 0, 4 * P, one, stride, (0, 0, 0)) if has_down else None
        if not has_down and (Cin != 4 * P or tuple(stride) != one):
            raise ValueError("identity shortcut needs matching shapes")
        self.a1_b16 = self.c2.wants_b16_input()
        Mi, Mo, N4 = self.c1.Mi, self.c2.Mo, 4 * P
        F = self.fbufs = _Bufs("F")    # forward intermediates kept for backward
        F.add("y1", Mi * P * 2), F.add("a1", Mi * P * 2), F.add("y2", Mo * P * 2), F.add("a2", Mo * P * 2)
        F.add("y3", Mo * N4 * 2)
        if has_down:
            F.add("yd", Mo * N4 * 2)
        F.add("st", 4 * 8192 * 4)      # four (mean, rstd) tables, 32 KiB apart
        F.add("mask", Mo * N4 // 8)
        G = self.gbufs = _Bufs("G")    # backward temporaries
        G.add("g3", Mo * N4 * 2), G.add("ga2", Mo * P * 2), G.add("g2", Mo * P * 2)
        G.add("gres", Mo * N4 * 2 if not (has_down and OPT["nogres"]) else 256)
        G.add("ga1", Mi * P * 2), G.add("g1", Mi * P * 2)
        if has_down:
            G.add("gd", Mo * N4 * 2), G.add("gxd", (Mo if self.cd.dgrad_via == "gemm_s2" else Mi) * Cin * 2)
        if self.c2.dgrad_via == "halo_up2":
            G.add("up2", Mi * P * 2)       # conv2's dY zero-upsampled to its input's size
        if B * N4 * 2 > STATS_MAX:
            raise RuntimeError("fused Bottleneck: statistics tables too small for this batch x width (bottleneck_ok guards this)")
        self.need = _Need()
        self.fwd = None
        self.bwd: Dict[tuple, object] = {}
        self.out_shape = (B, *self.c2.dout, N4)

    FWD_SLOTS = ("x", "out", "rd", "w1", "w2", "w3", "wd") + WS_NAMES

    def record_fwd(self):
        R = Recorder(self.FWD_SLOTS + tuple(self.fbufs.slot_names()))
        self.fbufs.bind(R)
        B, P, N4 = self.B, self.P, self.N4
        c1, c2, c3, cd = self.c1, self.c2, self.c3, self.cd
        y1, a1, y2, a2, y3 = R["y1"], R["a1"], R["y2"], R["a2"], R["y3"]
        st = [R["st"] + 32768 * i for i in range(4)]
        need = self.need
        nf = _InFwd(R)
        res = R["x"]
        self.f_dual = None
        dual = cd is not None and OPT["dual"]
        f_sc = False
        if dual:             # shortcut conv: its sums go to the shortcut accumulator and wait there for the block's last norm
            f_sc = em_conv_fwd(R, need, cd, R["x"], None, R["wd"], R["yd"], acc=R["ib0"])
            if not f_sc:
                R.call("ctu_in_stats", BF16, R["yd"], B, cd.So, N4, R["ib0"], st[3])   # (leaves ib0 zeroed)
        elif cd is not None:   # the shortcut branch first: its norm is done before conv3's sums need an accumulator
            f = em_conv_fwd(R, need, cd, R["x"], None, R["wd"], R["yd"], acc=nf.acc())
            nf.emit(R["yd"], st[3], R["rd"], B, cd.So, N4, fused=f, act=0)
            res = R["rd"]
        f = em_conv_fwd(R, need, c1, R["x"], None, R["w1"], y1, acc=nf.acc())
        nf.emit(y1, st[0], a1, B, c1.So, P, fused=f, act=1, b16=self.a1_b16)
        f = em_conv_fwd(R, need, c2, a1, None, R["w2"], y2, x1_b16=self.a1_b16, acc=nf.acc())
        nf.emit(y2, st[1], a2, B, c2.So, P, fused=f, act=1)
        f = em_conv_fwd(R, need, c3, a2, None, R["w3"], y3, acc=nf.acc())
        if dual:
            self.f_dual = _emit_dual(nf, y3, st[2], R["yd"], st[3], R["out"], B, c3.So, N4, fused=f, fused2=f_sc, act=1, mask=R["mask"])
        else:
            nf.emit(y3, st[2], R["out"], B, c3.So, N4, fused=f, residual=res, act=1, mask=R["mask"])
        self.f_norms, self.f_last = nf.k, nf.prev
        self.fwd = R.finish()

    BWD_SLOTS = ("x", "gy", "gx", "w1d", "w2d", "w3d", "wdd", "gw1", "gw2", "gw3", "gwd") + BWD_WS_NAMES

    def record_bwd(self, key):
        need_w, side = key
        R = Recorder(self.BWD_SLOTS + tuple(self.fbufs.slot_names()) + tuple(self.gbufs.slot_names()), nstreams=2)
        self.fbufs.bind(R)
        self.gbufs.bind(R)
        B, P, N4 = self.B, self.P, self.N4
        c1, c2, c3, cd = self.c1, self.c2, self.c3, self.cd
        y1, a1, y2, a2, y3 = R["y1"], R["a1"], R["y2"], R["a2"], R["y3"]
        st = [R["st"] + 32768 * i for i in range(4)]
        g3, gres, ga2, g2, ga1, g1 = (R[n] for n in ("g3", "gres", "ga2", "g2", "ga1", "g1"))
        need = self.need
        nb = _InBwd(R)
        # gn3 (+ residual + LeakyReLU): dy -> g3 (w.r.t. conv3's output) and gres (w.r.t. the shortcut)
        direct = cd is not None and OPT["nogres"]   # the shortcut norm applies mask and slope to gy itself: no gres tensor
        nb.emit(R["gy"], y3, None, st[2], g3, None if direct else gres, B, c3.So, N4, 1, 0, R["mask"])
        em_conv_dgrad(R, need, c3, g3, R["w3d"], ga2, None)
        if need_w[2]:
            em_conv_wgrad(R, need, c3, g3, a2, None, R["gw3"], stream=_wg_stream(R, side))
        nb.emit(ga2, y2, None, st[1], g2, None, B, c2.So, P, 1, c2.gy_b16)
        em_conv_dgrad(R, need, c2, g2, R["w2d"], ga1, None, gy_b16=c2.gy_b16, up2=R["up2"] if c2.dgrad_via == "halo_up2" else None)
        if need_w[1]:
            em_conv_wgrad(R, need, c2, g2, a1, None, R["gw2"], stream=_wg_stream(R, side), x1_b16=self.a1_b16, gy_b16=c2.gy_b16,
                          up2=R["up2"] if c2.dgrad_via == "halo_up2" else None)
        nb.emit(ga1, y1, None, st[0], g1, None, B, c1.So, P, 1, 0)
        extra, compact = gres, None
        if cd is not None:
            gd, gxd = R["gd"], R["gxd"]
            if direct:
                nb.emit(R["gy"], R["yd"], None, st[3], gd, None, B, cd.So, N4, 1, 0, R["mask"])
            else:
                nb.emit(gres, R["yd"], None, st[3], gd, None, B, cd.So, N4, 0, 0)
            em_conv_dgrad(R, need, cd, gd, R["wdd"], gxd, None)
            if need_w[3]:
                em_conv_wgrad(R, need, cd, gd, R["x"], None, R["gwd"], stream=_wg_stream(R, side))
            extra = gxd
            if cd.dgrad_via == "gemm_s2":   # gxd holds the output rows only: added at the even voxels behind conv1's data gradient
                extra, compact = None, gxd
        # conv1's data gradient takes the shortcut's gradient in its epilogue
        em_conv_dgrad(R, need, c1, g1, R["w1d"], R["gx"], None, extra=extra)
        if compact is not None:
            R.call("ctu_add_strided2", BF16, R["gx"], compact, B, *cd.dout, self.Cin)
        if need_w[0]:
            em_conv_wgrad(R, need, c1, g1, R["x"], None, R["gw1"], stream=_wg_stream(R, side))
        self.n_norms, self.last_n = nb.k, nb.prev
        self.bwd[key] = R.finish()
        return self.bwd[key]


class BottleneckFn(torch.autograd.Function):
    """Bottleneck.forward (networks/resnet.py:106-126) as one autograd node: 1x1x1 -> IN -> LReLU -> 3x3x3(stride) -> IN ->
    LReLU -> 1x1x1(x4) -> IN -> (+ identity | IN(1x1x1 strided)) -> LReLU.  x: [B, D, H, W, Cin] bf16."""

    @staticmethod
    def forward(ctx, x, w1, w2, w3, wd, stride):
        B, D, H, W, Cin = x.shape
        P = w1.shape[0]
        key = ("bneck", B, D, H, W, Cin, P, stride, wd is not None, _flags())
        pl = _cache.get(key)
        if pl is None:
            pl = _cache[key] = _BneckPlan(B, D, H, W, Cin, P, stride, wd is not None)
        if pl.fwd is None:
            pl.record_fwd()
        dev = x.device
        out = torch.empty(pl.out_shape, dtype=x.dtype, device=dev)
        bufs = pl.fbufs.alloc(dev)
        rd = torch.empty(pl.out_shape, dtype=x.dtype, device=dev) if (wd is not None and pl.f_dual is None) else None
        w1f, _ = conv_weights(pl.c1, w1, dgrad=False)
        w2f, _ = conv_weights(pl.c2, w2, dgrad=False)
        w3f, _ = conv_weights(pl.c3, w3, dgrad=False)
        wdf = conv_weights(pl.cd, wd, dgrad=False)[0] if wd is not None else None
        sid = L.stream()
        vals = [x.data_ptr(), out.data_ptr(), rd.data_ptr() if rd is not None else 0, w1f.data_ptr(),
                w2f.data_ptr(), w3f.data_ptr(), wdf.data_ptr() if wdf is not None else 0] + \
            _fwd_ws_values(dev, sid, pl.need, pl.f_norms, pl.f_last, pl.f_dual) + [t.data_ptr() for t in bufs]
        pl.fwd.run(vals, (sid,))
        ctx.pl = pl
        ctx.bufs = bufs
        ctx.save_for_backward(x, w1, w2, w3, wd)
        for i, w in enumerate((w1, w2, w3, wd)):
            ops.sink_expect(w, ctx.needs_input_grad[1 + i])
        return out

    @staticmethod
    def backward(ctx, gy):
        x, w1, w2, w3, wd = ctx.saved_tensors
        pl, bufs = ctx.pl, ctx.bufs
        gy = gy.contiguous()
        dev = x.device
        weights = (w1, w2, w3, wd)
        need_w = tuple(bool(ctx.needs_input_grad[1 + i]) and w is not None for i, w in enumerate(weights))
        targets = _grad_targets(weights, need_w)
        run = _BwdRun(dev, targets)
        key = (need_w, run.side is not None)
        plan = pl.bwd.get(key) or pl.record_bwd(key)
        gx = torch.empty_like(x)
        G = pl.gbufs.alloc(dev)
        wds = [conv_weights(s, w, fwd=False)[1] if w is not None else None
               for s, w in zip((pl.c1, pl.c2, pl.c3, pl.cd), weights)]
        wsv, w = run.ws_values(pl.need)
        vals = [x.data_ptr(), gy.data_ptr(), gx.data_ptr()] + \
               [t.data_ptr() if t is not None else 0 for t in wds] + \
               [t[0].data_ptr() if t[0] is not None else 0 for t in targets] + wsv + \
               [t.data_ptr() for t in bufs] + [t.data_ptr() for t in G]
        run.keep_alive(x, *bufs, *G)
        plan.run(vals, (run.s0, run.s1))
        run.finish(w, pl.n_norms, pl.last_n, targets)
        ctx.bufs = None
        return (gx,) + tuple(t[2] for t in targets) + (None,)


def bottleneck(blk, x):
    """resnet.Bottleneck.forward through the fused path."""
    wd = blk.downsample[0].conv.weight if blk.downsample is not None else None
    stride = ops._t3(blk.conv2.stride)
    return BottleneckFn.apply(x, blk.conv1.conv.weight, blk.conv2.conv.weight, blk.conv3.conv.weight, wd, stride)


def bottleneck_ok(blk, x) -> bool:
    if not usable(x) or x.dim() != 5:
        return False
    Cin, P = x.shape[-1], blk.conv1.out_channels
    return Cin % 32 == 0 and P % 32 == 0 and x.numel() < (1 << 31) and x.shape[0] * 4 * P * 2 <= STATS_MAX


# ---------------------------------------------------------------------------------------------------------------
# token pipelines: ViT trunk (networks/vit.py:46-139), window-attention stages (networks/hybrid_CTUNet.py:442-591)
# ---------------------------------------------------------------------------------------------------------------
# A pipeline is a chain of layers over a [rows, dim] activation: x_0 -> layer 0 -> x_1 -> ... -> x_L.  Each layer kind
# knows its parameters, the buffers it keeps for backward, and how to emit its forward / backward launches (the calls
# ops.LayerNormFn / LinearFn / AttentionFn / PixelShuffleFn make, in the same order).
LNWS_FLOATS = 1024 * 2 * 2048   # CTU_LN_BWD_MAX_BLOCKS partial rows of up to 2048 columns


class _Layer:
    params: Tuple[str, ...] = ()      # parameter names, in the order the Function receives them
    lin: Tuple[str, ...] = ()         # ... those that are Linear weights (forward operand = bf16 mirror [N][K])
    packed: Tuple[Tuple[str, str], ...] = ()   # (slot suffix, parameter name): re-ordered copies of a weight the forward list reads

    def __init__(self, tag):
        self.tag = tag

    def n(self, name):
        return f"{self.tag}.{name}"

    def t(self, R, name):
        """Transposed copy of a Linear weight, present only where the data gradient cannot read it reduction-major."""
        nm = self.n(name) + ".t"
        return R[nm] if R.has(nm) else None


def _lin_dgrad(R, need, gy, w_nk, w_t, gx, M, N, K, residual=None):
    """dX[M,K] = dY[M,N] @ W[N,K] (ops.LinearFn.backward): the forward weight read reduction-major, or a transposed copy."""
    if N % 64 == 0 and K % 8 == 0:
        em_gemm(R, need, gy, w_nk, gx, M, N, K, w_kn=1, residual=residual)
    else:
        em_gemm(R, need, gy, w_t, gx, M, N, K, residual=residual)


def _lin_wgrad(R, gy, x, gw, gb, M, N, K, stream):
    """dW[N,K] += dY^T X (+ bias gradient from the same pass)."""
    if gw is not None:
        R.call("ctu_igemm_tn", BF16, gy, N, x, None, gw, gb, ops._plain_geom(M, K, N), R["tnws1"], 1 << 24, stream=stream)
    elif gb is not None:
        R.call("ctu_colsum", BF16, gy, None, M, N, N, gb, stream=stream)


class AttnRes(_Layer):
    """x + to_out(attention(to_qkv(LayerNorm(x)))): vit.Attention (vit.py:46-78) / MultiAxisAttention
    (hybrid_CTUNet.py:481-511) under Residual (:434-440)."""

    def __init__(self, tag, geo, M, dim, out_bias, rel_bias, groups, ntok):
        super().__init__(tag)
        self.geo, self.M, self.dim, self.groups, self.ntok = geo, M, dim, groups, ntok
        self.params = ("g", "b", "wqkv", "wo") + (("bo",) if out_bias else ()) + (("tab",) if rel_bias else ())
        self.lin = ("wqkv", "wo")
        self.out_bias, self.rel_bias = out_bias, rel_bias

    def declare(self, F, G):
        M, D = self.M, self.dim
        for nm, nb in (("h", M * D * 2), ("mr", M * 8), ("qkv", M * 3 * D * 2), ("o", M * D * 2),
                       ("lse", self.groups * self.geo.heads * self.ntok * 4), ("y", M * D * 2)):
            F.add(self.n(nm), nb)
        for nm, nb in (("go", M * D * 2), ("gqkv", M * 3 * D * 2), ("gh", M * D * 2), ("gx", M * D * 2)):
            G.add(self.n(nm), nb)

    def fwd(self, R, need, x):
        n, M, D = self.n, self.M, self.dim
        R.call("ctu_layernorm_fwd", BF16, x, R[n("g")], R[n("b")], R[n("h")], R[n("mr")], M, D)
        em_gemm(R, need, R[n("h")], R[n("wqkv")], R[n("qkv")], M, D, 3 * D)
        R.call("ctu_attn_fwd", BF16, R[n("qkv")], R[n("tab")] if self.rel_bias else None, R[n("o")], R[n("lse")], self.geo)
        em_gemm(R, need, R[n("o")], R[n("wo")], R[n("y")], M, D, D, bias=R[n("bo")] if self.out_bias else None, residual=x)
        return R[n("y")]

    def bwd(self, R, need, x, gy, wg):
        n, M, D = self.n, self.M, self.dim
        _lin_dgrad(R, need, gy, R[n("wo")], self.t(R, "wo"), R[n("go")], M, D, D)
        _lin_wgrad(R, gy, R[n("o")], R[n("wo") + ".g"], R[n("bo") + ".g"] if self.out_bias else None, M, D, D, wg())
        R.call("ctu_attn_bwd", BF16, R[n("qkv")], R[n("tab")] if self.rel_bias else None, R[n("o")], R[n("go")], R[n("lse")],
               R[n("gqkv")], R[n("tab") + ".g"] if self.rel_bias else None, self.geo)
        _lin_dgrad(R, need, R[n("gqkv")], R[n("wqkv")], self.t(R, "wqkv"), R[n("gh")], M, 3 * D, D)
        _lin_wgrad(R, R[n("gqkv")], R[n("h")], R[n("wqkv") + ".g"], None, M, 3 * D, D, wg())
        R.call("ctu_layernorm_bwd_add", BF16, R[n("gh")], x, R[n("g")], R[n("mr")], gy, R[n("gx")], R[n("g") + ".g"],
               R[n("b") + ".g"], R["lnws"], M, D)
        return R[n("gx")]

    def needs_t(self):
        D = self.dim
        return {"wo": not (D % 64 == 0 and D % 8 == 0), "wqkv": not ((3 * D) % 64 == 0 and D % 8 == 0)}


class FFRes(_Layer):
    """x + Linear(GELU(Linear(LayerNorm(x)))): vit.FeedForward (vit.py:31-44) / hybrid_CTUNet.FeedForward (:513-526)."""

    def __init__(self, tag, M, dim, hidden):
        super().__init__(tag)
        self.M, self.dim, self.hidden = M, dim, hidden
        self.params = ("g", "b", "w1", "b1", "w2", "b2")
        self.lin = ("w1", "w2")
        # the forward as ONE kernel (ctu_ff_fwd) where its tile shapes fit: width 128, 256-row tiles, 64-unit hidden chunks
        self.fused_fwd = bool(OPT["ff1"] and dim == 128 and hidden % 64 == 0 and 128 <= hidden <= 4096 and M % 256 == 0)
        self.save = torch.is_grad_enabled()   # (without autograd the fused kernel writes neither `pre` nor `u`: 2.3 GB per 442 k-row call)
        if self.fused_fwd:
            self.packed = (("w2f", "w2"),)

    def declare(self, F, G):
        M, D, Hd = self.M, self.dim, self.hidden
        for nm, nb in (("h", M * D * 2), ("mr", M * 8), ("pre", M * Hd * 2), ("u", M * Hd * 2), ("y", M * D * 2)):
            if nm == "h" and self.fused_fwd:
                G.add(self.n(nm), nb)     # LayerNorm(x) is never written forward: the backward pass re-derives it
            elif not self.save and (nm == "pre" or (nm == "u" and self.fused_fwd)):
                pass                      # read by a backward pass only
            else:
                F.add(self.n(nm), nb)
        fused_gelu = D % 64 == 0 and Hd % 8 == 0 and OPT["gelu2"]
        for nm, nb in ((() if fused_gelu else (("gu", M * Hd * 2),)) + (("gpre", M * Hd * 2), ("gh", M * D * 2), ("gx", M * D * 2))):
            G.add(self.n(nm), nb)

    def fwd(self, R, need, x):
        n, M, D, Hd = self.n, self.M, self.dim, self.hidden
        if self.fused_fwd:
            R.call("ctu_ff_fwd", BF16, x, R[n("g")], R[n("b")], R[n("w1")], R[n("b1")], R[n("w2f")], R[n("b2")], R[n("y")],
                   R[n("pre")] if self.save else None, R[n("u")] if self.save else None, R[n("mr")], M, D, Hd)
            return R[n("y")]
        R.call("ctu_layernorm_fwd", BF16, x, R[n("g")], R[n("b")], R[n("h")], R[n("mr")], M, D)
        em_gemm(R, need, R[n("h")], R[n("w1")], R[n("u")], M, D, Hd, bias=R[n("b1")], act=1, pre_out=R[n("pre")] if self.save else None)
        em_gemm(R, need, R[n("u")], R[n("w2")], R[n("y")], M, Hd, D, bias=R[n("b2")], residual=x)
        return R[n("y")]

    def bwd(self, R, need, x, gy, wg):
        n, M, D, Hd = self.n, self.M, self.dim, self.hidden
        if D % 64 == 0 and Hd % 8 == 0 and OPT["gelu2"]:
            # dY.W2 leaves the GEMM already multiplied by GELU'(pre): no [M, hidden] gradient tensor written and re-read
            em_gemm(R, need, gy, R[n("w2")], R[n("gpre")], M, D, Hd, w_kn=1, act=2, residual=R[n("pre")])
        else:
            _lin_dgrad(R, need, gy, R[n("w2")], self.t(R, "w2"), R[n("gu")], M, D, Hd)
            R.call("ctu_gelu_bwd", BF16, R[n("gu")], R[n("pre")], R[n("gpre")], M * Hd)
        _lin_wgrad(R, gy, R[n("u")], R[n("w2") + ".g"], R[n("b2") + ".g"], M, D, Hd, wg())
        _lin_dgrad(R, need, R[n("gpre")], R[n("w1")], self.t(R, "w1"), R[n("gh")], M, Hd, D)
        if self.fused_fwd:   # h = LayerNorm(x) for W1's weight gradient (the same statistics land in mr again)
            R.call("ctu_layernorm_fwd", BF16, x, R[n("g")], R[n("b")], R[n("h")], R[n("mr")], M, D)
        _lin_wgrad(R, R[n("gpre")], R[n("h")], R[n("w1") + ".g"], R[n("b1") + ".g"], M, Hd, D, wg())
        R.call("ctu_layernorm_bwd_add", BF16, R[n("gh")], x, R[n("g")], R[n("mr")], gy, R[n("gx")], R[n("g") + ".g"],
               R[n("b") + ".g"], R["lnws"], M, D)
        return R[n("gx")]

    def needs_t(self):
        D, Hd = self.dim, self.hidden
        return {"w2": not (D % 64 == 0 and Hd % 8 == 0), "w1": not (Hd % 64 == 0 and D % 8 == 0)}


class Shuffle(_Layer):
    """PixelShuffle (hybrid_CTUNet.py:388-432): channel -> space rearrangement, then Linear(c -> out) with bias."""

    def __init__(self, tag, B, D, H, W, cbig, factor, cout):
        super().__init__(tag)
        self.B, self.D, self.H, self.W, self.factor, self.cout = B, D, H, W, tuple(factor), cout
        self.c = cbig // (factor[0] * factor[1] * factor[2])
        self.M = B * D * H * W * factor[0] * factor[1] * factor[2]
        self.Min, self.cbig = B * D * H * W, cbig
        self.params = ("w", "b")
        self.lin = ("w",)

    def declare(self, F, G):
        F.add(self.n("ps"), self.M * self.c * 2)
        F.add(self.n("y"), self.M * self.cout * 2)
        G.add(self.n("gps"), self.M * self.c * 2)
        G.add(self.n("gx"), self.Min * self.cbig * 2)

    def fwd(self, R, need, x):
        n, f = self.n, self.factor
        R.call("ctu_pixel_shuffle", BF16, x, R[n("ps")], self.B, self.D, self.H, self.W, self.c, f[0], f[1], f[2], 0)
        em_gemm(R, need, R[n("ps")], R[n("w")], R[n("y")], self.M, self.c, self.cout, bias=R[n("b")])
        return R[n("y")]

    def bwd(self, R, need, x, gy, wg):
        n, f = self.n, self.factor
        _lin_dgrad(R, need, gy, R[n("w")], self.t(R, "w"), R[n("gps")], self.M, self.cout, self.c)
        _lin_wgrad(R, gy, R[n("ps")], R[n("w") + ".g"], R[n("b") + ".g"], self.M, self.cout, self.c, wg())
        R.call("ctu_pixel_shuffle", BF16, R[n("gps")], R[n("gx")], self.B, self.D, self.H, self.W, self.c, f[0], f[1], f[2], 1)
        return R[n("gx")]

    def needs_t(self):
        return {"w": not (self.cout % 64 == 0 and self.c % 8 == 0)}


class _PipePlan:
    FWD_WS = ("skws",)
    BWD_WS = ("skws", "tnws1", "lnws")

    def __init__(self, layers: List[_Layer], x_bytes: int):
        self.layers = layers
        self.F, self.G = _Bufs("F"), _Bufs("G")
        for ly in layers:
            ly.declare(self.F, self.G)
        # the last layer's output / the first layer's input gradient are the Function's own tensors, not buffers
        self.out_name = layers[-1].n("y")
        self.F.remove(self.out_name)
        self.gx_name = layers[0].n("gx")
        self.G.remove(self.gx_name)
        self.pnames = [ly.n(p) for ly in layers for p in ly.params]
        self.is_lin = [p in ly.lin for ly in layers for p in ly.params]
        self.tnames = [ly.n(p) + ".t" for ly in layers for p, need in ly.needs_t().items() if need]
        self.t_index = [self.pnames.index(t[:-2]) for t in self.tnames]
        self.xnames = [ly.n(sfx) for ly in layers for sfx, _ in ly.packed]          # re-ordered weight copies (forward list)
        self.x_index = [self.pnames.index(ly.n(src)) for ly in layers for _, src in ly.packed]
        self.need = _Need()
        self.fwd = None
        self.bwd: Dict[tuple, object] = {}

    def record_fwd(self):
        R = Recorder(["x", self.out_name] + self.pnames + self.xnames + list(self.FWD_WS) + self.F.slot_names())
        self.F.bind(R)
        x = R["x"]
        for ly in self.layers:
            x = ly.fwd(R, self.need, x)
        self.fwd = R.finish()

    def record_bwd(self, key):
        side = key
        gnames = [p + ".g" for p in self.pnames]
        R = Recorder(["x", "gy", self.gx_name] + self.pnames + self.tnames + gnames + list(self.BWD_WS) + self.F.slot_names() +
                     [self.out_name] + self.G.slot_names(), nstreams=2)
        self.F.bind(R)
        self.G.bind(R)
        xs = [R["x"]] + [R[ly.n("y")] for ly in self.layers[:-1]]
        g = R["gy"]
        wg = lambda: _wg_stream(R, side)   # noqa: E731  (every weight-gradient launch waits for what precedes it)
        for ly, x in zip(reversed(self.layers), reversed(xs)):
            g = ly.bwd(R, self.need, x, g, wg)
        self.bwd[key] = R.finish()
        return self.bwd[key]


class PipeFn(torch.autograd.Function):
    """A chain of AttnRes / FFRes / Shuffle layers as one autograd node.  apply(x, plan, out_shape, *params)."""

    @staticmethod
    def forward(ctx, x, pl: _PipePlan, out_shape, *params):
        if pl.fwd is None:
            pl.record_fwd()
        dev = x.device
        out = torch.empty(out_shape, dtype=x.dtype, device=dev)
        bufs = pl.F.alloc(dev)
        pv = [(ops._linear_weight(p, p, torch.bfloat16) if lin else p).data_ptr() for p, lin in zip(params, pl.is_lin)]
        xv = [ops._packed(params[i], "ff_w2f", torch.bfloat16, lambda p=params[i]: _ff_w2_frag(p)).data_ptr() for i in pl.x_index]
        sid = L.stream()
        vals = [x.data_ptr(), out.data_ptr()] + pv + xv + [ops._splitk_workspace(dev, pl.need.skws).data_ptr()] + \
               [t.data_ptr() for t in bufs]
        pl.fwd.run(vals, (sid,))
        ctx.pl, ctx.bufs = pl, bufs
        ctx.save_for_backward(x, out, *params)
        for i, p in enumerate(params):
            ops.sink_expect(p, ctx.needs_input_grad[3 + i])
        return out

    @staticmethod
    def backward(ctx, gy):
        x, out, *params = ctx.saved_tensors
        pl, bufs = ctx.pl, ctx.bufs
        gy = gy.contiguous()
        dev = x.device
        need_w = [bool(ctx.needs_input_grad[3 + i]) for i in range(len(params))]
        if not all(need_w):
            raise NotImplementedError("fused token pipeline: every parameter is expected to require a gradient")
        targets = _grad_targets(params, need_w)
        run = _BwdRun(dev, targets)
        key = run.side is not None
        plan = pl.bwd.get(key) or pl.record_bwd(key)
        gx = torch.empty_like(x)
        G = pl.G.alloc(dev)
        pv = [(ops._linear_weight(p, p, torch.bfloat16) if lin else p).data_ptr() for p, lin in zip(params, pl.is_lin)]
        tv = [ops._packed(params[i], "lin_d", torch.bfloat16,
                          lambda p=params[i]: p.detach().t().to(torch.bfloat16).contiguous()).data_ptr() for i in pl.t_index]
        key1 = (dev, run.s1, ops._ws_epoch)
        vals = [x.data_ptr(), gy.data_ptr(), gx.data_ptr()] + pv + tv + [t[0].data_ptr() for t in targets] + \
               [ops._splitk_workspace(dev, pl.need.skws).data_ptr(), _tn_ws_for(key1).data_ptr(), _lnws(dev, run.s0).data_ptr()] + \
               [t.data_ptr() for t in bufs] + [out.data_ptr()] + [t.data_ptr() for t in G]
        run.keep_alive(x, gy, out, *bufs, *G)
        plan.run(vals, (run.s0, run.s1))
        # "gradient complete", last layer first (the flat gradient buffer's order); main-stream and weight-gradient-stream
        # gradients alike: whoever listens joins the side streams
        for t in reversed(targets):
            if t[1] is not None:
                t[1]()
        ctx.bufs = None
        return (gx, None, None) + tuple(t[2] for t in targets)


def _ff_w2_frag(w2):
    """W2 [D][hidden] of a FeedForward in the fragment order ctu_ff_fwd reads (cached per parameter version by ops._packed)."""
    m = ops._linear_weight(w2, w2, torch.bfloat16)   # (the optimizer's bf16 mirror is a flat view: shapes from the parameter)
    out = torch.empty(w2.shape, dtype=torch.bfloat16, device=w2.device)
    L.call("ctu_ff_pack_w2", m.data_ptr(), out.data_ptr(), w2.shape[0], w2.shape[1], L.stream())
    return out


_LNWS: Dict[tuple, torch.Tensor] = {}


def _lnws(device, sid):
    key = (device, sid, ops._ws_epoch)
    t = _LNWS.get(key)
    if t is None:
        t = _LNWS[key] = torch.empty(LNWS_FLOATS, dtype=torch.float32, device=device)
    return t


def _attn_params(m, out_bias, rel_bias):
    ps = [m.norm.weight, m.norm.bias, m.to_qkv.weight, m.to_out[0].weight]
    if out_bias:
        ps.append(m.to_out[0].bias)
    if rel_bias:
        ps.append(m.rel_pos_bias.weight)
    return ps


def _ff_params(net):
    return [net[0].weight, net[0].bias, net[1].weight, net[1].bias, net[4].weight, net[4].bias]


def _all_trainable(module) -> bool:
    """The fused backward produces every parameter gradient of the block; a partly frozen block takes the per-op path."""
    return not torch.is_grad_enabled() or all(p.requires_grad for p in module.parameters())


def vit_trunk_ok(vit, x) -> bool:
    b0 = vit.transformer[0]
    drop = max(b0.attn.dropout.p, b0.attn.to_out[1].p, b0.ff.net[3].p, b0.ff.net[5].p)
    if not (usable(x) and x.dim() == 3 and not (vit.training and drop > 0.0) and _all_trainable(vit.transformer)):
        return False
    # AttnRes sizes qkv / o and their GEMMs from `dim`: heads * dim_head has to equal it (num_heads is a reference CLI flag,
    # main_CTUNet.py:59; hidden_size 768 with 8 heads of 64 gives an inner width of 512 - the per-op path is shape-generic)
    dim = x.shape[-1]
    return all(tuple(b.attn.to_qkv.weight.shape) == (3 * dim, dim) and tuple(b.attn.to_out[0].weight.shape) == (dim, dim)
               for b in vit.transformer)


def vit_trunk(blocks, x):
    """The TransformerBlocks of vit.ViT.transformer (vit.py:80-96) over tokens [B, n, dim] as one autograd node."""
    B, n, dim = x.shape
    a0 = blocks[0].attn
    heads, dh = a0.heads, a0.to_qkv.weight.shape[0] // (3 * a0.heads)
    hidden = blocks[0].ff.net[1].weight.shape[0]
    key = ("vit", B, n, dim, heads, dh, hidden, len(blocks), _flags())
    pl = _cache.get(key)
    if pl is None:
        M = B * n
        geo = L.AttnGeom(0, B, n, 1, 1, 0, heads, dh, a0.scale)
        layers = []
        for i in range(len(blocks)):
            layers.append(AttnRes(f"a{i}", geo, M, dim, True, False, B, n))
            layers.append(FFRes(f"f{i}", M, dim, hidden))
        pl = _cache[key] = _PipePlan(layers, M * dim * 2)
    params = []
    for blk in blocks:
        params += _attn_params(blk.attn, True, False)
        params += _ff_params(blk.ff.net)
    return PipeFn.apply(x, pl, tuple(x.shape), *params)


def up_stage_ok(x, blk, ind, training) -> bool:
    if not (usable(x) and x.dim() == 5):
        return False
    mods = (blk[1], blk[2], blk[5], blk[6]) if ind <= 2 else (blk[1], blk[2])
    C = x.shape[-1]
    for m in mods:
        f = m.fn
        p = (f.to_out[1].p if hasattr(f, "to_out") else max(f.net[3].p, f.net[5].p))
        if training and p > 0.0:
            return False
        if hasattr(f, "to_qkv") and (tuple(f.to_qkv.weight.shape) != (3 * C, C) or tuple(f.to_out[0].weight.shape) != (C, C)):
            return False   # (inner width != dim: AttnRes sizes its buffers from dim)
    return _all_trainable(blk)


def up_stage(blk, ind, x):
    """One stage of UpAttentionBlock (hybrid_CTUNet.py:528-591): block attention + FF + grid attention + FF + PixelShuffle
    (stage 3: FF + FF + PixelShuffle) over the channels-last volume x [B, D, H, W, C] as one autograd node."""
    B, D, H, W, C = x.shape
    sh = blk[8] if ind <= 2 else blk[4]
    cout = sh.to_out.weight.shape[0]
    f = sh.scale_factor
    key = ("upstage", ind <= 2, B, D, H, W, C, cout, f, _flags())
    pl = _cache.get(key)
    M = B * D * H * W
    if pl is None:
        layers = []
        if ind <= 2:
            a = blk[1].fn
            win = a.window_size
            groups = B * (D // win) * (H // win) * (W // win)
            for tag, part in (("a1", 1), ("a2", 2)):
                geo = L.AttnGeom(part, B, D, H, W, win, a.heads, C // a.heads, a.scale)
                layers.append(AttnRes(tag, geo, M, C, False, True, groups, win ** 3))
                layers.append(FFRes("f" + tag[1], M, C, blk[2].fn.net[1].weight.shape[0]))
        else:
            layers.append(FFRes("f1", M, C, blk[1].fn.net[1].weight.shape[0]))
            layers.append(FFRes("f2", M, C, blk[2].fn.net[1].weight.shape[0]))
        layers.append(Shuffle("ps", B, D, H, W, C, f, cout))
        pl = _cache[key] = _PipePlan(layers, M * C * 2)
    if ind <= 2:
        params = _attn_params(blk[1].fn, False, True) + _ff_params(blk[2].fn.net) + \
            _attn_params(blk[5].fn, False, True) + _ff_params(blk[6].fn.net)
    else:
        params = _ff_params(blk[1].fn.net) + _ff_params(blk[2].fn.net)
    params += [sh.to_out.weight, sh.to_out.bias]
    return PipeFn.apply(x, pl, (B, D * f[0], H * f[1], W * f[2], cout), *params)


# ---------------------------------------------------------------------------------------------------------------
# ResBlock (networks/hybrid_CTUNet.py:29-105), stride 1, optionally over a channel-concatenated pair of inputs
# ---------------------------------------------------------------------------------------------------------------
class _ResBlockPlan:
    def __init__(self, B, D, H, W, C1, C2, N, has_down):
        self.B, self.N, self.C1, self.C2, self.has_down = B, N, C1, C2, has_down
        one, three = (1, 1, 1), (3, 3, 3)
        self.c1 = ConvSpec(B, (D, H, W), C1, C2, N, three, one, one)
        self.c2 = ConvSpec(B, (D, H, W), N, 0, N, three, one, one)
        self.c3 = ConvSpec(B, (D, H, W), C1, C2, N, one, one, (0, 0, 0)) if has_down else None
        if self.c1.kind != "halo" or self.c2.kind != "halo":
            raise NotImplementedError("fused ResBlock: both 3x3x3 convolutions on the halo kernel")
        self.a1_b16 = self.c2.wants_b16_input()
        M = self.c1.Mo
        self.S = self.c1.So
        F = self.fbufs = _Bufs("F")
        F.add("y1", M * N * 2), F.add("a1", M * N * 2), F.add("y2", M * N * 2)
        if has_down:
            F.add("y3", M * N * 2)
        F.add("st", 3 * 8192 * 4)
        F.add("mask", M * N // 8)
        G = self.gbufs = _Bufs("G")
        G.add("g2", M * N * 2), G.add("ga1", M * N * 2), G.add("g1", M * N * 2)
        G.add("gres", M * N * 2 if not (has_down and OPT["nogres"]) else 256)
        if has_down:
            G.add("g3", M * N * 2), G.add("gs1", M * C1 * 2)
            if C2:
                G.add("gs2", M * C2 * 2)
        if B * N * 2 > STATS_MAX:
            raise RuntimeError("fused ResBlock: statistics tables too small for this batch x width (resblock_ok guards this)")
        self.need = _Need()
        self.fwd = None
        self.bwd: Dict[tuple, object] = {}
        self.out_shape = (B, D, H, W, N)

    FWD_SLOTS = ("x1", "x2", "out", "rd", "w1", "w2", "w3") + WS_NAMES

    def record_fwd(self):
        R = Recorder(self.FWD_SLOTS + tuple(self.fbufs.slot_names()))
        self.fbufs.bind(R)
        B, N, S = self.B, self.N, self.S
        c1, c2, c3 = self.c1, self.c2, self.c3
        x2 = R["x2"] if self.C2 else None
        st = [R["st"] + 32768 * i for i in range(3)]
        need = self.need
        nf = _InFwd(R)
        res = R["x1"]
        self.f_dual = None
        dual = c3 is not None and OPT["dual"]
        f_sc = False
        if dual:             # (see _BneckPlan.record_fwd)
            f_sc = em_conv_fwd(R, need, c3, R["x1"], x2, R["w3"], R["y3"], acc=R["ib0"])
            if not f_sc:
                R.call("ctu_in_stats", BF16, R["y3"], B, S, N, R["ib0"], st[2])
        elif c3 is not None:   # the shortcut branch first (see _BneckPlan.record_fwd)
            f = em_conv_fwd(R, need, c3, R["x1"], x2, R["w3"], R["y3"], acc=nf.acc())
            nf.emit(R["y3"], st[2], R["rd"], B, S, N, fused=f, act=0)
            res = R["rd"]
        f = em_conv_fwd(R, need, c1, R["x1"], x2, R["w1"], R["y1"], acc=nf.acc())
        nf.emit(R["y1"], st[0], R["a1"], B, S, N, fused=f, act=1, b16=self.a1_b16)
        f = em_conv_fwd(R, need, c2, R["a1"], None, R["w2"], R["y2"], x1_b16=self.a1_b16, acc=nf.acc())
        if dual:
            self.f_dual = _emit_dual(nf, R["y2"], st[1], R["y3"], st[2], R["out"], B, S, N, fused=f, fused2=f_sc, act=1, mask=R["mask"])
        else:
            nf.emit(R["y2"], st[1], R["out"], B, S, N, fused=f, residual=res, act=1, mask=R["mask"])
        self.f_norms, self.f_last = nf.k, nf.prev
        self.fwd = R.finish()

    BWD_SLOTS = ("x1", "x2", "gy", "gx1", "gx2", "ext", "w1d", "w2d", "w3d", "gw1", "gw2", "gw3") + BWD_WS_NAMES

    def record_bwd(self, key):
        need_w, side, has_ext = key
        R = Recorder(self.BWD_SLOTS + tuple(self.fbufs.slot_names()) + tuple(self.gbufs.slot_names()), nstreams=2)
        self.fbufs.bind(R)
        self.gbufs.bind(R)
        B, N, S = self.B, self.N, self.S
        c1, c2, c3 = self.c1, self.c2, self.c3
        x2 = R["x2"] if self.C2 else None
        gx2 = R["gx2"] if self.C2 else None
        st = [R["st"] + 32768 * i for i in range(3)]
        need = self.need
        nb = _InBwd(R)
        # norm2 (+ residual + LeakyReLU)
        direct = c3 is not None and OPT["nogres"]   # (as in the bottleneck: norm3's backward takes gy + mask)
        nb.emit(R["gy"], R["y2"], None, st[1], R["g2"], None if direct else R["gres"], B, S, N, 1, c2.gy_b16, R["mask"])
        em_conv_dgrad(R, need, c2, R["g2"], R["w2d"], R["ga1"], None, gy_b16=c2.gy_b16)
        if need_w[1]:
            em_conv_wgrad(R, need, c2, R["g2"], R["a1"], None, R["gw2"], stream=_wg_stream(R, side), x1_b16=self.a1_b16,
                          gy_b16=c2.gy_b16)
        nb.emit(R["ga1"], R["y1"], None, st[0], R["g1"], None, B, S, N, 1, c1.gy_b16)
        ext = R["ext"] if has_ext else None
        if c3 is not None:
            # conv shortcut: norm3 (no activation) -> conv3's data gradients, folded into conv1's data-gradient epilogue
            if direct:
                nb.emit(R["gy"], R["y3"], None, st[2], R["g3"], None, B, S, N, 1, 0, R["mask"])
            else:
                nb.emit(R["gres"], R["y3"], None, st[2], R["g3"], None, B, S, N, 0, 0)
            gs2 = R["gs2"] if self.C2 else None
            em_conv_dgrad(R, need, c3, R["g3"], R["w3d"], R["gs1"], gs2, extra=ext)
            if need_w[2]:
                em_conv_wgrad(R, need, c3, R["g3"], R["x1"], x2, R["gw3"], stream=_wg_stream(R, side))
            em_conv_dgrad(R, need, c1, R["g1"], R["w1d"], R["gx1"], gx2, extra=R["gs1"], extra2=gs2, gy_b16=c1.gy_b16)
        else:
            assert not has_ext
            em_conv_dgrad(R, need, c1, R["g1"], R["w1d"], R["gx1"], gx2, extra=R["gres"], gy_b16=c1.gy_b16)
        if need_w[0]:
            em_conv_wgrad(R, need, c1, R["g1"], R["x1"], x2, R["gw1"], stream=_wg_stream(R, side), gy_b16=c1.gy_b16)
        self.n_norms, self.last_n = nb.k, nb.prev
        self.bwd[key] = R.finish()
        return self.bwd[key]


class ResBlockFn(torch.autograd.Function):
    """ResBlock.forward (networks/hybrid_CTUNet.py:93-105) as one autograd node; x2 = second half of a channel concat
    (torch.cat at :199,618) or None; w3 = the 1x1x1 shortcut convolution's weight when in != out channels, else None.
    grad_stash: a list another consumer of x1 parks its gradient in (ops.GradStash); it joins conv3's data gradient."""

    @staticmethod
    def forward(ctx, x1, x2, w1, w2, w3, grad_stash):
        B, D, H, W, C1 = x1.shape
        C2 = x2.shape[-1] if x2 is not None else 0
        N = w1.shape[0]
        key = ("resblock", B, D, H, W, C1, C2, N, w3 is not None, _flags())
        pl = _cache.get(key)
        if pl is None:
            pl = _cache[key] = _ResBlockPlan(B, D, H, W, C1, C2, N, w3 is not None)
        if pl.fwd is None:
            pl.record_fwd()
        dev = x1.device
        out = torch.empty(pl.out_shape, dtype=x1.dtype, device=dev)
        bufs = pl.fbufs.alloc(dev)
        rd = torch.empty(pl.out_shape, dtype=x1.dtype, device=dev) if (w3 is not None and pl.f_dual is None) else None
        w1f = conv_weights(pl.c1, w1, dgrad=False)[0]
        w2f = conv_weights(pl.c2, w2, dgrad=False)[0]
        w3f = conv_weights(pl.c3, w3, dgrad=False)[0] if w3 is not None else None
        sid = L.stream()
        vals = [x1.data_ptr(), x2.data_ptr() if x2 is not None else 0, out.data_ptr(), rd.data_ptr() if rd is not None else 0,
                w1f.data_ptr(), w2f.data_ptr(), w3f.data_ptr() if w3f is not None else 0] + \
            _fwd_ws_values(dev, sid, pl.need, pl.f_norms, pl.f_last, pl.f_dual) + [t.data_ptr() for t in bufs]
        pl.fwd.run(vals, (sid,))
        ctx.pl, ctx.bufs, ctx.grad_stash = pl, bufs, grad_stash
        ctx.save_for_backward(x1, x2, w1, w2, w3)
        for i, w in enumerate((w1, w2, w3)):
            ops.sink_expect(w, ctx.needs_input_grad[2 + i])
        return out

    @staticmethod
    def backward(ctx, gy):
        x1, x2, w1, w2, w3 = ctx.saved_tensors
        pl, bufs = ctx.pl, ctx.bufs
        gy = gy.contiguous()
        dev = x1.device
        weights = (w1, w2, w3)
        need_w = tuple(bool(ctx.needs_input_grad[2 + i]) and w is not None for i, w in enumerate(weights))
        targets = _grad_targets(weights, need_w)
        run = _BwdRun(dev, targets)
        ext = ops._take(ctx.grad_stash)
        if ext is not None and (ext.shape != x1.shape or ext.dtype != x1.dtype or not ext.is_contiguous()):
            ext = ext.to(x1.dtype).contiguous().view_as(x1)
        if ext is not None and w3 is None:
            raise NotImplementedError("an outside gradient stash and an identity shortcut share conv1's one residual input")
        key = (need_w, run.side is not None, ext is not None)
        plan = pl.bwd.get(key) or pl.record_bwd(key)
        gx1 = torch.empty_like(x1)
        gx2 = torch.empty_like(x2) if x2 is not None else None
        G = pl.gbufs.alloc(dev)
        wds = [conv_weights(s, w, fwd=False)[1] if w is not None else None for s, w in zip((pl.c1, pl.c2, pl.c3), weights)]
        wsv, w = run.ws_values(pl.need)
        vals = [x1.data_ptr(), x2.data_ptr() if x2 is not None else 0, gy.data_ptr(), gx1.data_ptr(),
                gx2.data_ptr() if gx2 is not None else 0, ext.data_ptr() if ext is not None else 0] + \
               [t.data_ptr() if t is not None else 0 for t in wds] + \
               [t[0].data_ptr() if t[0] is not None else 0 for t in targets] + wsv + \
               [t.data_ptr() for t in bufs] + [t.data_ptr() for t in G]
        run.keep_alive(x1, x2, *bufs, *G)
        plan.run(vals, (run.s0, run.s1))
        run.finish(w, pl.n_norms, pl.last_n, targets)
        ctx.bufs = None
        return (gx1, gx2) + tuple(t[2] for t in targets) + (None,)


def resblock_ok(blk, x1, x2, grad_stash) -> bool:
    if not usable(x1) or x1.dim() != 5 or (x2 is not None and not (x2.is_contiguous() and x2.dtype == x1.dtype)):
        return False
    c1 = blk.conv1
    if c1.kernel_size != (3, 3, 3) or c1.stride != (1, 1, 1) or blk.conv2.kernel_size != (3, 3, 3):
        return False
    C1, C2, N = x1.shape[-1], (x2.shape[-1] if x2 is not None else 0), c1.out_channels
    if C1 % 32 or C2 % 32 or N % 32 or x1.numel() // C1 * max(C1, C2, N) >= (1 << 31) or x1.shape[0] * N * 2 > STATS_MAX:
        return False
    if not blk.downsample and (grad_stash is not None or x2 is not None):
        return False
    return True


def resblock(blk, x1, x2, grad_stash):
    w3 = blk.conv3.conv.weight if blk.downsample else None
    return ResBlockFn.apply(x1, x2, blk.conv1.conv.weight, blk.conv2.conv.weight, w3, grad_stash)


# ---------------------------------------------------------------------------------------------------------------
# pixelweight_attention: binary cross-weight fusion (networks/hybrid_CTUNet.py:622-669)
# ---------------------------------------------------------------------------------------------------------------
class _PwaPlan:
    PNAMES = ("g1", "b1", "g2", "b2", "wq1", "wq2", "wo")
    LIN = (False, False, False, False, True, True, True)

    def __init__(self, M, C, scale):
        self.M, self.C, self.scale = M, C, float(scale)
        F = self.F = _Bufs("F")
        for i in (1, 2):
            F.add(f"h{i}", M * C * 2), F.add(f"mr{i}", M * 8), F.add(f"qkv{i}", M * 3 * C * 2)
        F.add("o", M * C * 2)
        G = self.G = _Bufs("G")
        G.add("go", M * C * 2)
        for i in (1, 2):
            G.add(f"gq{i}", M * 3 * C * 2), G.add(f"gh{i}", M * C * 2)
        self.need = _Need()
        self.fwd = None
        self.bwd: Dict[tuple, object] = {}
        self.t_names = [n + ".t" for n, bad in (("wq1", (3 * C) % 64 or C % 8), ("wq2", (3 * C) % 64 or C % 8), ("wo", C % 64 or C % 8)) if bad]

    def record_fwd(self):
        R = Recorder(["x1", "x2", "out"] + list(self.PNAMES) + ["skws"] + self.F.slot_names())
        self.F.bind(R)
        M, C = self.M, self.C
        for i in (1, 2):
            R.call("ctu_layernorm_fwd", BF16, R[f"x{i}"], R[f"g{i}"], R[f"b{i}"], R[f"h{i}"], R[f"mr{i}"], M, C)
            em_gemm(R, self.need, R[f"h{i}"], R[f"wq{i}"], R[f"qkv{i}"], M, C, 3 * C)
        R.call("ctu_pwa_fwd", BF16, R["qkv1"], R["qkv2"], R["o"], M, C, self.scale)
        em_gemm(R, self.need, R["o"], R["wo"], R["out"], M, C, C)
        self.fwd = R.finish()

    def record_bwd(self, side):
        gnames = [p + ".g" for p in self.PNAMES]
        R = Recorder(["x1", "x2", "gy", "gx1", "gx2"] + list(self.PNAMES) + self.t_names + gnames + ["skws", "tnws1", "lnws"] +
                     self.F.slot_names() + self.G.slot_names(), nstreams=2)
        self.F.bind(R)
        self.G.bind(R)
        M, C = self.M, self.C
        t = lambda n: R[n + ".t"] if R.has(n + ".t") else None   # noqa: E731
        _lin_dgrad(R, self.need, R["gy"], R["wo"], t("wo"), R["go"], M, C, C)
        _lin_wgrad(R, R["gy"], R["o"], R["wo.g"], None, M, C, C, _wg_stream(R, side))
        R.call("ctu_pwa_bwd", BF16, R["qkv1"], R["qkv2"], R["go"], R["gq1"], R["gq2"], M, C, self.scale)
        for i in (2, 1):
            _lin_dgrad(R, self.need, R[f"gq{i}"], R[f"wq{i}"], t(f"wq{i}"), R[f"gh{i}"], M, 3 * C, C)
            _lin_wgrad(R, R[f"gq{i}"], R[f"h{i}"], R[f"wq{i}.g"], None, M, 3 * C, C, _wg_stream(R, side))
            R.call("ctu_layernorm_bwd_add", BF16, R[f"gh{i}"], R[f"x{i}"], R[f"g{i}"], R[f"mr{i}"], None, R[f"gx{i}"],
                   R[f"g{i}.g"], R[f"b{i}.g"], R["lnws"], M, C)
        self.bwd[side] = R.finish()
        return self.bwd[side]


class PwaBlockFn(torch.autograd.Function):
    """pixelweight_attention.forward (hybrid_CTUNet.py:645-669): to_out(cross_weight(to_qkv1(LN1(x1)), to_qkv2(LN2(x2))))."""

    @staticmethod
    def forward(ctx, x1, x2, scale, *params):
        C = x1.shape[-1]
        M = x1.numel() // C
        key = ("pwa", M, C, float(scale), _flags())
        pl = _cache.get(key)
        if pl is None:
            pl = _cache[key] = _PwaPlan(M, C, scale)
        if pl.fwd is None:
            pl.record_fwd()
        dev = x1.device
        out = torch.empty_like(x1)
        bufs = pl.F.alloc(dev)
        pv = [(ops._linear_weight(p, p, torch.bfloat16) if lin else p).data_ptr() for p, lin in zip(params, pl.LIN)]
        sid = L.stream()
        vals = [x1.data_ptr(), x2.data_ptr(), out.data_ptr()] + pv + [ops._splitk_workspace(dev, pl.need.skws).data_ptr()] + \
               [t.data_ptr() for t in bufs]
        pl.fwd.run(vals, (sid,))
        ctx.pl, ctx.bufs = pl, bufs
        ctx.save_for_backward(x1, x2, *params)
        for i, p in enumerate(params):
            ops.sink_expect(p, ctx.needs_input_grad[3 + i])
        return out

    @staticmethod
    def backward(ctx, gy):
        x1, x2, *params = ctx.saved_tensors
        pl, bufs = ctx.pl, ctx.bufs
        gy = gy.contiguous()
        dev = x1.device
        need_w = [bool(ctx.needs_input_grad[3 + i]) for i in range(len(params))]
        if not all(need_w):
            raise NotImplementedError("fused cross-weight block: every parameter is expected to require a gradient")
        targets = _grad_targets(params, need_w)
        run = _BwdRun(dev, targets)
        side = run.side is not None
        plan = pl.bwd.get(side) or pl.record_bwd(side)
        gx1, gx2 = torch.empty_like(x1), torch.empty_like(x2)
        G = pl.G.alloc(dev)
        pv = [(ops._linear_weight(p, p, torch.bfloat16) if lin else p).data_ptr() for p, lin in zip(params, pl.LIN)]
        tv = [ops._packed(params[pl.PNAMES.index(n[:-2])], "lin_d", torch.bfloat16,
                          lambda p=params[pl.PNAMES.index(n[:-2])]: p.detach().t().to(torch.bfloat16).contiguous()).data_ptr()
              for n in pl.t_names]
        key1 = (dev, run.s1, ops._ws_epoch)
        vals = [x1.data_ptr(), x2.data_ptr(), gy.data_ptr(), gx1.data_ptr(), gx2.data_ptr()] + pv + tv + \
               [t[0].data_ptr() for t in targets] + \
               [ops._splitk_workspace(dev, pl.need.skws).data_ptr(), _tn_ws_for(key1).data_ptr(), _lnws(dev, run.s0).data_ptr()] + \
               [t.data_ptr() for t in bufs] + [t.data_ptr() for t in G]
        run.keep_alive(x1, x2, gy, *bufs, *G)
        plan.run(vals, (run.s0, run.s1))
        for t in reversed(targets):
            if t[1] is not None:
                t[1]()
        ctx.bufs = None
        return (gx1, gx2, None) + tuple(t[2] for t in targets)


def pwa_block_ok(m, x1, x2) -> bool:
    return (usable(x1) and x2.is_contiguous() and x2.dtype == x1.dtype and x1.shape == x2.shape and x1.shape[-1] % 32 == 0
            and _all_trainable(m))


_PWA_PACKED: Dict[int, tuple] = {}
_PWA_STATS: Dict[tuple, torch.Tensor] = {}


def _pwa_packed(m):
    """The three weight matrices of a pixelweight_attention in the fragment order ctu_pwa_block_fwd streams (ctu_pwa_pack), rebuilt
    when any of them changes; cached per module (weakly)."""
    ws = (m.to_qkv1.weight, m.to_qkv2.weight, m.to_out[0].weight)
    ver = tuple((w._version, w.data_ptr()) for w in ws) + (ops._weights_epoch,)
    ent = _PWA_PACKED.get(id(m))
    if ent is not None and ent[0]() is m and ent[1] == ver:
        ent[3].fence()
        return ent[2]
    with torch.no_grad():
        mir = [ops._linear_weight(w, w, torch.bfloat16) for w in ws]
        t = torch.empty(4 * 56 * 512, dtype=torch.bfloat16, device=ws[0].device)
        L.call("ctu_pwa_pack", mir[0].data_ptr(), mir[1].data_ptr(), mir[2].data_ptr(), t.data_ptr(), 128, L.stream())
    mid = id(m)
    _PWA_PACKED[mid] = (weakref.ref(m, lambda _r, mid=mid: _PWA_PACKED.pop(mid, None)), ver, t, ops._Built(t.device))
    return t


def pwa_block(m, x1, x2):
    C = x1.shape[-1]
    M = x1.numel() // C
    if OPT["pwa1"] and C == 128 and M % 128 == 0 and not torch.is_grad_enabled():
        out = torch.empty_like(x1)
        mr = _PWA_STATS.get((x1.device, M))     # the LayerNorm statistics nobody reads without a backward pass: one scratch per size
        if mr is None:
            mr = _PWA_STATS[(x1.device, M)] = torch.empty((2, M, 2), dtype=torch.float32, device=x1.device)
        L.call("ctu_pwa_block_fwd", BF16, x1.data_ptr(), x2.data_ptr(), m.norm1.weight.data_ptr(), m.norm1.bias.data_ptr(),
               m.norm2.weight.data_ptr(), m.norm2.bias.data_ptr(), _pwa_packed(m).data_ptr(), out.data_ptr(), None, None,
               mr[0].data_ptr(), mr[1].data_ptr(), M, C, float(m.scale), L.stream())
        return out
    return PwaBlockFn.apply(x1, x2, m.scale, m.norm1.weight, m.norm1.bias, m.norm2.weight, m.norm2.bias, m.to_qkv1.weight,
                            m.to_qkv2.weight, m.to_out[0].weight)
