"""CPU suite, part 2: the C-ABI library loads and exports every symbol include/ctunet_hip.h declares (no compute calls
without a GPU), the host-side mirror keeps the reference's interface (state_dict manifests, constructor errors, loud
failure without a HIP device), and the data-parallel wrapper all-reduces correctly over gloo with world_size 2."""
import ctypes
import json
import os
import re
import socket

import numpy as np
import pytest
import torch
import torch.nn as nn

import hybrid_ctunet_amd as H
from hybrid_ctunet_amd import _lib, train

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ctunet_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(ctu_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 28
    path = _lib.build()  # no-op when the in-tree .so is newer than its sources
    lib = ctypes.CDLL(path)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in ctunet_hip.h but not exported"
    assert sorted(_lib.EXPORTED) == declared, "python binding table and header disagree"
    lib.ctu_abi_version.restype = ctypes.c_int
    assert lib.ctu_abi_version() == 8


def test_plan_dispatch_table_is_current_and_packs_every_prototype():
    """csrc/plan_dispatch.inc (the launch-list interpreter's switch) is generated from the header; the recorder packs argument
    words by the same parse.  A plan can be recorded without a GPU: check the word layout of a call with inline structs."""
    from hybrid_ctunet_amd import _plan
    assert open(_plan.DISPATCH_INC).read() == _plan.emit_dispatch(), "run `python -m hybrid_ctunet_amd._plan --emit`"
    assert len(_plan.SIGS) >= 40 and "ctu_adamw" not in _plan.SIGS and "ctu_plan_run" not in _plan.SIGS
    for name, sig in _plan.SIGS.items():          # the binding table and the header parse agree on the argument count
        assert len(sig) == len(_lib._SIGS[name]), name
    R = _plan.Recorder(["x", "w", "y", "acc"], nstreams=2)
    e = _lib.Epilogue()
    e.ldc, e.in_rows = 64, 128
    e._refs = {_lib.Epilogue.in_acc.offset: R["acc"] + 256}
    R.call("ctu_igemm_nt", _lib.CTU_BF16, R["x"], None, R["w"], R["y"] + 1024, _lib.Geom(B=1, Di=4096, N=64), e, stream=1)
    ev = R.new_event()
    R.event_record(ev, 1)
    R.stream_wait(0, ev)
    assert R.words[0] == _plan.OPS["ctu_igemm_nt"] and R.words[1] == 1 and R.words[2] == 5 + 10 + 14
    pos = {p[0]: (p[1], p[2]) for p in R.patches}
    assert pos[3 + 1] == (0, 0) and pos[3 + 3] == (1, 0) and pos[3 + 4] == (2, 1024)
    assert pos[3 + 5 + 10 + _lib.Epilogue.in_acc.offset // 8] == (3, 256)
    assert R.words[3 + 5] & 0xFFFFFFFF == 1 and R.words[3 + 5] >> 32 == 4096      # ctu_geom inline: B, Di
    assert R.words[-8:] == [_plan.OP_EVENT_RECORD, 1, 1, 0, _plan.OP_STREAM_WAIT, 0, 1, 0]
    with pytest.raises(TypeError):
        R.call("ctu_add", _lib.CTU_BF16, 12345, R["x"], R["y"], 64)     # a raw address cannot enter a plan


def test_ctypes_struct_layouts_match_header():
    # 20 int32 fields; epilogue: 2 pointers, 2 ints, pointer, 10 ints
    assert ctypes.sizeof(_lib.Geom) == 80
    assert ctypes.sizeof(_lib.AttnGeom) == 36
    e = _lib.Epilogue
    assert e.bias.offset == 0 and e.residual.offset == 8 and e.act.offset == 16 and e.ldc.offset == 20
    assert e.out2.offset == 24 and e.n_split.offset == 32 and e.sc_kw.offset == 32 + 4 * 9
    assert e.splitk.offset == 72 and e.w_kn.offset == 76 and e.splitk_ws.offset == 80 and e.in_acc.offset == 88
    assert e.in_rows.offset == 96 and e.pre_out.offset == 104 and ctypes.sizeof(e) == 112


@pytest.mark.parametrize("kind,depth,man", [("ctunet", 101, "ctunet101"), ("cunet", 101, "cunet101"),
                                            ("cunet", 50, "cunet50"), ("tunet", 101, "tunet")])
def test_state_dict_keys_and_shapes_match_reference_manifest(golden_dir, kind, depth, man):
    with torch.device("meta"):
        m = H.build_model(kind, model_depth=depth)
    ref = json.load(open(os.path.join(golden_dir, f"manifest_{man}.json")))
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == ref
    assert len(list(m.buffers())) == (6 if kind != "cunet" else 0)  # rel_pos_indices, non-persistent


def test_constructor_contract():
    with pytest.raises(AssertionError):
        H.CUNet(out_channels=14, model_depth=34)
    with torch.device("meta"):
        with pytest.raises(ValueError):
            H.CTUNet(in_channels=1, dim_conv_stem=64, out_channels=14, model_depth=50, img_size=(96, 96), frames=96,
                     patch_frame=16)
        with pytest.raises(NotImplementedError):
            H.CUNet(out_channels=14, model_depth=50, norm_name="batch")
        with pytest.raises(ValueError):  # torch's nn.Dropout raises ValueError outside [0, 1]
            H.TUNet(in_channels=1, dim_conv_stem=64, out_channels=14, img_size=(96, 96), frames=96, patch_frame=8,
                    dropout_rate=1.2)
        t = H.TUNet(in_channels=1, dim_conv_stem=64, out_channels=14, img_size=(96, 96), frames=96, patch_frame=8,
                    dropout_rate=0.2)   # the authors' dr0.2 recipe (test_CTUNet_final.py:448)
        assert t.vit.dropout.p == 0.2
        with pytest.raises(AssertionError):  # vit.py:108-109
            H.TUNet(in_channels=1, dim_conv_stem=64, out_channels=14, img_size=(100, 96), frames=96, patch_frame=8)


def test_no_cpu_fallback():
    m = H.CUNet(out_channels=14, model_depth=50)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 1, 96, 96, 96))
    with pytest.raises(RuntimeError):
        H.dice_ce_loss(torch.zeros(1, 14, 4, 4, 4), torch.zeros(1, 1, 4, 4, 4))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "hybrid-ctunet_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f


def test_zoom_index_map_host_equals_oracle():
    from oracle import ctunet_oracle as O
    for n_in, n_out in ((96, 48), (96, 24), (48, 24), (24, 12), (10, 5)):
        assert np.array_equal(train.zoom_nearest_index(n_in, n_out), O.zoom_nearest_index(n_in, n_out))
    a = train.zoom_nearest_index(96, 48)
    assert a[23] == 46 and a[24] == 49  # the jump SURVEY 8a-H documents: not a stride-2 subsample


def test_gradient_ready_order_covers_all_parameters():
    with torch.device("meta"):
        m = H.build_model("ctunet")
    order = train.gradient_ready_order(m)
    assert len(order) == 412 and len({id(p) for p in order}) == 412
    names = {id(p): n for n, p in m.named_parameters()}
    first, last = names[id(order[0])], names[id(order[-1])]
    assert first.startswith("res_out_24x24") and last.startswith("convnet.")   # the stem: last and smallest
    # the parameter-heavy ViT trunk sits in the middle of the order (CTUNet.forward builds the ResNet first), so its
    # all-reduce overlaps the convnet's backward
    pos = [i for i, p in enumerate(order) if names[id(p)].startswith("vit.transformer")]
    conv = [i for i, p in enumerate(order) if names[id(p)].startswith("convnet.")]
    assert max(pos) < min(conv)


class _Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(12, 40)
        self.unused = nn.Linear(3, 3)  # never receives a gradient (like ResBlock.conv3 when in == out)
        self.b = nn.Linear(40, 5)

    def forward(self, x):
        return self.b(torch.tanh(self.a(x)))


def test_flat_params_views_and_untouched_ranges():
    torch.manual_seed(0)
    m = _Toy()
    ref = [p.detach().clone() for p in m.parameters()]
    fp = train.FlatParams(m.parameters())
    for p, r in zip(m.parameters(), ref):
        assert torch.equal(p, r) and p.data_ptr() >= fp.flat.data_ptr()
    m(torch.randn(7, 12)).sum().backward()
    rng = fp.untouched_ranges()
    assert len(rng) == 1 and rng[0][1] - rng[0][0] == 64 + 64  # weight (9->64) + bias (3->64) of `unused`
    assert fp.grad.abs().sum() > 0
    for p, o in zip(fp.params, fp.offsets):
        assert p.grad.data_ptr() == fp.grad.data_ptr() + 4 * o
    fp.zero_grad()
    assert fp.grad.abs().sum() == 0 and fp.untouched_ranges()[0][0] == 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(123 + rank)  # different init per rank: the wrapper must broadcast rank 0's parameters
    m = _Toy()
    dp = train.DataParallel(m, bucket_mb=0.0005, static_unused=True)  # tiny buckets -> several of them
    assert hasattr(dp, "module") and len(dp.buckets) >= 2
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 12, generator=g)
    xs = x[rank * 4:(rank + 1) * 4]
    early = []
    for step in range(3):
        dp.flat.zero_grad()
        dp(xs).pow(2).mean().backward()
        early.append(sum(dp._launched))   # buckets that went out during backward, before finish() flushes the rest
        dp.finish()
    # static_unused: from the second step on the bucket(s) holding the never-used Linear no longer wait for finish()
    assert early[0] < len(dp.buckets) and early[1] == early[2] == len(dp.buckets), early
    assert dp._unused == {i for i, p in enumerate(dp.flat.params) if any(p is q for q in m.unused.parameters())}
    torch.save({"grad": dp.flat.grad.clone(), "flat": dp.flat.flat.clone()}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_gloo_world2(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "r0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "r1.pt"))
    assert torch.equal(r0["flat"], r1["flat"])            # parameters were broadcast from rank 0
    assert torch.allclose(r0["grad"], r1["grad"], atol=1e-7)  # every bucket (incl. the one with unused params) reduced
    # reference: single process, full batch, rank 0's init
    torch.manual_seed(123)
    m = _Toy()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 12, generator=g)
    # mean over the 2 half-batches of the per-half mean loss == what DP averages
    loss = 0.5 * (m(x[:4]).pow(2).mean() + m(x[4:]).pow(2).mean())
    loss.backward()
    ref = torch.cat([torch.nn.functional.pad(p.grad.reshape(-1) if p.grad is not None else torch.zeros(p.numel()),
                                             (0, (-p.numel()) % 64)) for p in m.parameters()])
    assert torch.allclose(r0["grad"], ref, atol=1e-6)


def test_reference_grad_reset_idiom_does_not_accumulate_across_steps():
    """the reference loop clears gradients with `param.grad = None` (trainer_CTUNet.py:88-89) and
    drives torch.optim.AdamW; with DataParallel(model) in between, autograd then installs a fresh .grad every step and
    the FlatParams hook must COPY it into the flat slice - adding it (round 1) made the flat gradient, and therefore the
    all-reduced bucket and the update, grow 1x, 2x, 3x ... over the steps."""
    torch.manual_seed(0)
    plain = _Toy()
    wrapped = _Toy()
    wrapped.load_state_dict(plain.state_dict())
    dp = train.DataParallel(wrapped)                       # world 1: no process group needed
    o1 = torch.optim.AdamW(plain.parameters(), lr=1e-2, weight_decay=1e-5)
    o2 = torch.optim.AdamW(wrapped.parameters(), lr=1e-2, weight_decay=1e-5)
    g = torch.Generator().manual_seed(3)
    for step in range(3):
        x = torch.randn(6, 12, generator=g)
        for p in plain.parameters():
            p.grad = None
        for p in wrapped.parameters():
            p.grad = None
        plain(x).pow(2).mean().backward()
        dp(x).pow(2).mean().backward()
        dp.finish()
        for (n, a), b in zip(plain.named_parameters(), wrapped.parameters()):
            if a.grad is None:
                assert b.grad is None, n
            else:
                assert torch.allclose(a.grad, b.grad, atol=1e-7), (step, n)
        o1.step()
        o2.step()
    for a, b in zip(plain.parameters(), wrapped.parameters()):
        assert torch.allclose(a, b, atol=1e-6)
    # torch's own zero_grad(set_to_none=False) idiom keeps working in place too
    o2.zero_grad(set_to_none=False)
    assert dp.flat.grad[:dp.flat.offsets[1]].abs().sum() == 0


def test_one_flat_owner_per_parameter():
    """DataParallel(model) followed by FusedAdamW(model.parameters()) used to build a second
    FlatParams over the same parameters (both folding every gradient).  Now the optimizer adopts the existing owner, and
    an explicit second FlatParams raises until the first is released."""
    m = _Toy()
    dp = train.DataParallel(m)
    assert train.FlatParams.of(m.parameters()) is dp.flat
    opt = train.FusedAdamW(m.parameters(), lr=1e-3)
    assert opt.flat is dp.flat
    with pytest.raises(RuntimeError, match="already belongs"):
        train.FlatParams(m.parameters())
    dp.flat.release()
    assert train.FlatParams.of(m.parameters()) is None
    fp2 = train.FlatParams(m.parameters())               # re-homing after an explicit release is allowed
    m(torch.randn(4, 12)).sum().backward()
    assert fp2.grad.abs().sum() > 0


class _Shared(nn.Module):
    """One weight used twice in a graph (weight sharing)."""

    def __init__(self):
        super().__init__()
        self.w = nn.Linear(8, 8)
        self.tail = nn.Linear(8, 2)

    def forward(self, x):
        return self.tail(self.w(torch.tanh(self.w(x))))


def test_parameter_used_twice_reports_once_with_the_full_gradient():
    """The bucket of a parameter may only be reduced after its LAST contribution: autograd sums the uses before
    AccumulateGrad runs once, so the hook's single report already sees the complete gradient."""
    torch.manual_seed(1)
    m = _Shared()
    ref = _Shared()
    ref.load_state_dict(m.state_dict())
    dp = train.DataParallel(m, bucket_mb=0.00001)
    seen = []
    at_report = {}

    def listener(i):
        seen.append(i)
        p = dp.flat.params[i]
        at_report[i] = p.grad.detach().clone()
    dp.flat.listeners.append(listener)
    x = torch.randn(5, 8)
    dp(x).pow(2).sum().backward()
    ref(x).pow(2).sum().backward()
    dp.finish()
    assert sorted(seen) == list(range(len(dp.flat.params)))          # exactly one report per parameter
    for i, p in enumerate(dp.flat.params):
        r = dict(ref.named_parameters())[[n for n, q in m.named_parameters() if q is p][0]]
        assert torch.allclose(at_report[i], r.grad, atol=1e-6)       # and it was complete when reported


def test_package_synthetic_batch_equals_the_oracle_generator():
    from oracle import ctunet_oracle as O
    a, b = H.synthetic_batch(1, size=(8, 8, 8), seed=1003), O.synthetic_batch(1, size=(8, 8, 8), seed=1003)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_data_parallel_release_restores_the_stream_switch():
    """the wrapper switches the weight-gradient companion streams off process-wide while it exchanges
    gradients (four hardware pipes: main, branch, exchange, RCCL's stream - train.DataParallel.__init__); release() must put the
    caller's setting back and stop listening to the gradient buffer."""
    from hybrid_ctunet_amd import ops
    m = _Toy()
    dp = train.DataParallel(m)
    before = ops.WGRAD_STREAM
    try:
        ops.WGRAD_STREAM = False
        dp._wgrad_stream_before = True          # what the constructor records on a GPU with world > 1
        assert dp._on_ready in dp.flat.listeners
        dp.release()
        assert ops.WGRAD_STREAM is True and dp._wgrad_stream_before is None
        assert dp._on_ready not in dp.flat.listeners
        dp.release()                            # idempotent
        assert ops.WGRAD_STREAM is True
    finally:
        ops.WGRAD_STREAM = before
        dp.flat.release()


def test_plan_keys_separate_forward_with_and_without_autograd():
    """A forward plan recorded under torch.no_grad() leaves out what only a backward pass reads (ops_fused.FFRes.save, the fused
    cross-weight forward): the autograd mode therefore has to be part of every plan key."""
    import torch
    from hybrid_ctunet_amd import ops_fused as F
    on = F._flags()
    with torch.no_grad():
        off = F._flags()
    assert on != off and on[:-1] == off[:-1] and on[-1] is True and off[-1] is False
    with torch.no_grad():
        ff = F.FFRes("f", 512, 128, 512)
    assert ff.fused_fwd and not ff.save
    assert F.FFRes("f", 512, 128, 512).save
