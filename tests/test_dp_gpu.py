"""Two data-parallel ranks sharing the one GPU of the test box (gloo moves the buckets; the stream / event / hook logic
is the same code that runs over RCCL on a multi-GPU node): the bench's own step function - DataParallel with direct
gradient sinks, bucketed all-reduce on the side stream overlapped with backward, finish(), fused AdamW with the frozen
skip ranges - on different inputs per rank.  If every bucket is reduced exactly once and after its last contribution,
both ranks apply identical updates and their parameters stay BIT-identical; any missed, early or doubled bucket shows
up as a divergence."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, kind, reducer_opt=False):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hybrid_ctunet_amd as H
    from oracle.ctunet_oracle import synthetic_batch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(100 + rank)                      # different initial weights: the constructor's broadcast must win
    model = H.build_model(kind, model_depth=50).to(dev)
    flat = H.FlatParams(H.gradient_ready_order(model))
    dp = H.DataParallel(model, flat=flat, bucket_mb=8.0, static_unused=reducer_opt)
    assert len(dp.buckets) >= 4
    opt = H.FusedAdamW(None, lr=1e-3, weight_decay=1e-5, flat=flat)
    if reducer_opt:
        dp.attach_optimizer(opt)   # every bucket updated right behind its all-reduce, on the communication stream
        assert dp._opt is opt
    flat0 = flat.flat.detach().cpu().clone()           # (after the constructor's broadcast)
    x, y = synthetic_batch(1, seed=1000 + rank)        # different data per rank
    x, y = x.to(dev), y.to(dev)
    grads = []
    for step in range(3):
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = H.LOSSES[kind](model(x), y)
        loss.backward()
        dp.finish()
        if step == 0:
            opt.freeze_skip_ranges()
        torch.cuda.synchronize()
        grads.append(flat.grad.detach().cpu().clone())
        opt.step()
    torch.cuda.synchronize()
    tag = "o" if reducer_opt else "r"
    torch.save({"flat": flat.flat.detach().cpu(), "grads": grads, "loss": float(loss), "m": opt.m.detach().cpu(),
                "v": opt.v.detach().cpu(), "skip": opt._static_skip, "flat0": flat0,
                "unused": sorted(dp._unused) if dp._unused else []}, os.path.join(out_dir, f"{tag}{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["cunet", "tunet", "ctunet"])  # ctunet = BASELINE config 5's model (depth 50 here)
def test_two_ranks_on_one_gpu_stay_identical(tmp_path, kind):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), kind), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "r0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "r1.pt"))
    for g0, g1 in zip(r0["grads"], r1["grads"]):
        assert torch.equal(g0, g1)                      # the all-reduced flat gradient, every bucket of it
        assert torch.isfinite(g0).all() and g0.abs().max() > 0
    assert torch.equal(r0["flat"], r1["flat"])          # three identical AdamW updates on the broadcast parameters
    assert r0["loss"] != r1["loss"]                     # ... although the ranks saw different data


def test_reducer_driven_optimizer_leaves_gradient_less_parameters_alone(tmp_path):
    """DataParallel(static_unused=True, optimizer=opt): from the second step on a bucket holding never-used parameters goes out
    during backward and is UPDATED behind its all-reduce.  torch.optim.AdamW (the reference's optimizer) does not touch a
    parameter whose grad is None - no weight decay, no state - and neither may the per-bucket update: those ranges
    keep their initial values bit for bit and their moments stay zero; everything else follows the plain finish() + step()
    sequence (two separate runs: the weight-gradient atomics are not bit-reproducible, and Adam's first steps move a weight by
    ~lr whatever the gradient's magnitude, so the comparison is bounded by steps x lr, not bit-exact)."""
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), "ctunet", False), nprocs=2, join=True)
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), "ctunet", True), nprocs=2, join=True)
    plain = torch.load(os.path.join(tmp_path, "r0.pt"))
    over = torch.load(os.path.join(tmp_path, "o0.pt"))
    over1 = torch.load(os.path.join(tmp_path, "o1.pt"))
    assert over["unused"], "CTUNet builds ResBlock.conv3 tensors it never calls"
    assert torch.equal(over["flat"], over1["flat"])
    assert over["skip"] == plain["skip"] and over["skip"]
    for a, b in over["skip"]:                            # the gradient-less ranges: no decay, no moments, in both schedules
        for r in (over, plain):
            assert torch.equal(r["flat"][a:b], r["flat0"][a:b])
            assert (r["m"][a:b] == 0).all() and (r["v"][a:b] == 0).all()
    assert torch.equal(plain["flat0"], over["flat0"])
    d = (plain["flat"] - over["flat"]).abs()
    assert d.max().item() <= 2 * 3 * 1e-3 + 1e-4, d.max().item()
    moved = (over["flat"] - over["flat0"]).abs()
    assert moved.max().item() > 1e-3                     # ... and the used parameters did move


def test_ctunet_gradients_become_ready_in_bucket_order():
    """Bucket layout = gradient_ready_order; what matters for overlap is that the real backward agrees with it: the
    parameter-heavy ViT trunk must be done well before the end (CTUNet.forward builds the ResNet branch first for exactly
    this), the last gradients to appear must be the small stem of the convnet, and buckets must complete roughly in
    order."""
    import hybrid_ctunet_amd as H
    from oracle.ctunet_oracle import synthetic_batch
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = H.build_model("ctunet", model_depth=50).to(dev)
    order = H.gradient_ready_order(model)
    flat = H.FlatParams(order)
    dp = H.DataParallel(model, flat=flat, bucket_mb=32.0)
    fired = []
    flat.listeners.append(lambda i: fired.append(i))
    names = {id(p): n for n, p in model.named_parameters()}
    x, y = synthetic_batch(1, seed=1000)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = H.ctunet_loss(model(x.to(dev)), y.to(dev))
    loss.backward()
    dp.finish()
    seen, seq = set(), []
    for i in fired:                       # first report of every parameter
        if i not in seen:
            seen.add(i)
            seq.append(i)
    pos = {i: k for k, i in enumerate(seq)}
    n = len(seq)
    vit = [pos[i] for i, p in enumerate(flat.params) if i in pos and names[id(p)].startswith("vit.transformer")]
    conv = [pos[i] for i, p in enumerate(flat.params) if i in pos and names[id(p)].startswith("convnet.")]
    assert vit and conv and max(vit) < min(conv)                       # the whole convnet backward still follows the trunk
    assert names[id(flat.params[seq[-1]])].startswith("convnet.")      # the tail is the ResNet's shallow end
    # buckets complete (last member reported) in nearly increasing order: count inversions between neighbours
    done = []
    for begin, end, members in dp.buckets:
        ks = [pos[i] for i in members if i in pos]
        if ks:
            done.append(max(ks))
    inv = sum(1 for a, b in zip(done, done[1:]) if b < a)
    assert inv <= max(1, len(done) // 5), (inv, done)


def test_rccl_bucket_exchange_world1():
    """ctu_comm_init / ctu_allreduce_bucket over a real RCCL communicator (one rank: the box has one GPU): the fp32 payload
    (ncclAllReduce, ncclAvg) must leave a bucket unchanged, the bf16 payload (cast -> all-to-all -> fp32 sum -> all-gather
    -> expand) must return exactly the bf16 rounding of every element, ragged tail included; on a side stream."""
    from hybrid_ctunet_amd.comm import Communicator
    comm = Communicator.from_torch()
    assert comm.world == 1
    g = torch.Generator().manual_seed(0)
    for n in (1000003, 8 * 1024 * 1024, 64):
        x = torch.randn(n, generator=g).cuda()
        ref = x.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            comm.allreduce_mean(x, "fp32")
        torch.cuda.current_stream().wait_stream(side)
        assert torch.equal(x, ref)
        with torch.cuda.stream(side):
            comm.allreduce_mean(x, "bf16")
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert torch.equal(x, ref.to(torch.bfloat16).float())
    comm.close()


def test_shared_weight_gradient_sink_reports_after_the_last_use():
    """A parameter used by two ops of one graph accumulates twice into its gradient sink; 'gradient complete' (what lets
    DataParallel launch the bucket's all-reduce) may fire only after the second accumulation (ops.sink_expect)."""
    import hybrid_ctunet_amd as H
    from hybrid_ctunet_amd import ops
    torch.manual_seed(0)
    w = torch.nn.Parameter((torch.randn(64, 64) * 0.1).cuda())
    flat = H.FlatParams([w])
    reports = []
    flat.listeners.append(lambda i: reports.append(flat.grad.clone()))
    x = torch.randn(512, 64, device="cuda")
    flat.zero_grad()
    h = ops.linear(ops.linear(x, w), w)
    h.sum().backward()
    torch.cuda.synchronize()
    wr = w.detach().clone().requires_grad_(True)
    ((x @ wr.t()) @ wr.t()).sum().backward()
    # the FIRST report already carries both contributions (a later duplicate is harmless: DataParallel counts a parameter once)
    assert len(reports) >= 1
    for r in reports:
        assert torch.allclose(r[:4096].view(64, 64), wr.grad, rtol=2e-3, atol=2e-3 * wr.grad.abs().max().item())


@pytest.mark.parametrize("world", [2, 3, 8])
def test_bf16_bucket_dataflow_with_virtual_ranks(world):
    """The bf16 payload of ctu_allreduce_bucket - cast -> all-to-all -> fp32 sum x 1/world -> all-gather -> expand - at world
    sizes the one-GPU box cannot run over RCCL: the three kernels are the real ones (ctu_allreduce_bucket_stage), the two
    collectives are played by copies between the scratch buffers of `world` virtual ranks.  Checks chunking, the padding to
    8 elements, the ragged tail, and that every rank - the chunk's owner included - ends with the same bf16-rounded means."""
    from hybrid_ctunet_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(world)
    for n in (1000003, 8 * 4096, 13):
        xs = [torch.randn(n, generator=g).cuda() for _ in range(world)]
        ref = torch.zeros(n, device="cuda")
        for x in xs:
            ref = ref + x.to(torch.bfloat16).float()          # the kernel's order: rank 0 first, fp32 accumulate
        ref = (ref * torch.tensor(1.0 / world, dtype=torch.float32, device="cuda")).to(torch.bfloat16).float()
        nbytes = L.ctu_allreduce_scratch_bytes(world, n)
        chunk = ((n + world - 1) // world + 7) // 8 * 8
        scr = [torch.empty(nbytes // 2, dtype=torch.bfloat16, device="cuda") for _ in range(world)]
        bufs = [x.clone() for x in xs]
        st = _lib.stream()
        for r in range(world):
            _lib.call("ctu_allreduce_bucket_stage", 0, world, bufs[r].data_ptr(), n, scr[r].data_ptr(), nbytes, st)
        send = [s[:world * chunk].view(world, chunk) for s in scr]
        recv = [s[world * chunk:2 * world * chunk].view(world, chunk) for s in scr]
        mean = [s[2 * world * chunk:2 * world * chunk + chunk] for s in scr]
        for r in range(world):                                  # all-to-all: rank r receives chunk r of every rank
            for w in range(world):
                recv[r][w].copy_(send[w][r])
        for r in range(world):
            _lib.call("ctu_allreduce_bucket_stage", 1, world, bufs[r].data_ptr(), n, scr[r].data_ptr(), nbytes, st)
        for r in range(world):                                  # all-gather of the mean chunks into the send region
            for w in range(world):
                send[r][w].copy_(mean[w])
        for r in range(world):
            _lib.call("ctu_allreduce_bucket_stage", 2, world, bufs[r].data_ptr(), n, scr[r].data_ptr(), nbytes, st)
        torch.cuda.synchronize()
        for r in range(world):
            assert torch.equal(bufs[r], ref), (world, n, r)
