#!/usr/bin/env python3
"""Headline benchmark: training volumes/second of CTUNet (ResNet d101 + ViT, patch_frame 8) on 96^3 bf16 patches.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch per GPU (per-GPU batch 2, BASELINE.json configs[3]/[4]):
zero grads -> forward under autocast(bf16) -> 5-head DiceCE with on-device deep-supervision targets -> backward
(+ RCCL gradient all-reduce overlapped on a side stream when N > 1) -> fused AdamW.  Synthetic inputs (image
U[0,1), labels randint(0,14), generator seed 1000+rank) are resident in HBM before the timed region.  W warm-up steps,
then exactly K steps bracketed by barrier + torch.cuda.synchronize() on both sides; the time is the MAX over ranks;
rank 0 prints ONE JSON line.

Extra objects in that line (DESIGN.md section "Measurement"):
  roofline     live HIP-event timing of the dominant kernel (the implicit-GEMM family) over instrumented steps run
               right after the timed region: achieved = algorithmic FLOPs of its launches / their summed duration.
  cpu_baseline the oracle (CPU restatement of the reference, kind "port") timed on this box's host cores on a bounded
               sample: ONE 96^3 volume through forward + loss + backward + AdamW.
"""
import argparse
import json
import os
import sys
import time

# Before HIP initialises (ROCclr reads it once): eight hardware queues instead of the default four.  The step runs on five to six
# HIP streams (two encoder branches, weight-gradient companions, the gradient-exchange stream); RCCL's communicator adds its own,
# and with four queues two of OUR streams then share one: 47.5 -> 55.4 ms per step from nothing but dist.init_process_group("nccl")
# (profiles/r03_bench_rccl_group_vs_hw_queues.log; 5, 6 and 8 queues: 47.3 - 47.5 with and without the group).  A value the caller
# has set stays.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
# measured on the pool's MI355X with tools/peaks (profiles/r02_peaks.log): back-to-back v_mfma_f32_32x32x16_bf16 on random
# data sustain 1 867 TFLOP/s at the 1.84 - 1.87 GHz the chip holds under that load (32.0 cycles per MFMA per SIMD);
# HBM 16-B-per-lane streams: read 6.45, write 4.55, copy 4.85 TB/s
PEAK_BF16_TFLOPS_MEASURED = 1867.0
FLOP_PER_VOLUME = {"ctunet": 10.26e12, "cunet": 5.278e12, "tunet": 3.496e12}  # fwd+bwd, SURVEY.md section 8d


class KernelTimer:
    """Collects (entry point, algorithmic FLOPs, event pair) for every implicit-GEMM launch."""

    names = {"ctu_igemm_nt", "ctu_igemm_tn", "ctu_conv3_halo", "ctu_conv3_halo_wgrad"}

    def __init__(self):
        self.rec = []

    def add(self, name, args, e0, e1):
        if name == "ctu_conv3_halo":       # (dtype, x1, x2, w, out, out2, B, D, H, W, C1, C2, N, ...)
            B, D, H, W, C1, C2, N = args[6:13]
            self.rec.append((name, 2.0 * B * D * H * W * N * (C1 + C2) * 27, e0, e1))
            self.shape[len(self.rec) - 1] = f"conv3_halo {C1}+{C2}->{N} @{D}x{H}x{W} B{B}"
            return
        if name == "ctu_conv3_halo_wgrad":  # (dtype, dy, x1, x2, dw, B, D, H, W, C1, C2, N, layouts, ws, ws_floats, stream)
            B, D, H, W, C1, C2, N = args[5:12]
            self.rec.append((name, 2.0 * B * D * H * W * N * (C1 + C2) * 27, e0, e1))
            self.shape[len(self.rec) - 1] = f"conv3_halo_wgrad {C1}+{C2}->{N} @{D}x{H}x{W} B{B}"
            return
        g = args[5] if name == "ctu_igemm_nt" else args[7]  # (dtype, p, ldp, q1, q2, dw, bias_grad, geom, stream)
        taps = g.kd * g.kh * g.kw
        rows_out = g.B * g.Do * g.Ho * g.Wo
        rows_in = g.B * g.Di * g.Hi * g.Wi
        k = g.C1 + g.C2
        if name == "ctu_igemm_nt" and g.mode == 1:
            # data gradient of a (possibly strided) conv: algorithmic MACs = those of the forward conv it differentiates
            flops = 2.0 * rows_in * g.N * k * taps
        else:
            flops = 2.0 * rows_out * g.N * k * taps
        self.rec.append((name, flops, e0, e1))
        self.shape[len(self.rec) - 1] = f"{name[4:]} M={rows_out} N={g.N} K={k} taps={taps} s={g.sd}{g.sh}{g.sw} mode={g.mode}"

    shape = {}

    def by_shape(self, top=25):
        agg = {}
        for i, (name, flops, e0, e1) in enumerate(self.rec):
            key = self.shape.get(i, name)
            a = agg.setdefault(key, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1)
            a[2] += flops
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]
        return [f"{v[1]:8.2f} ms  x{v[0]:3d}  {v[2] / max(v[1], 1e-9) / 1e9:8.1f} TFLOP/s  {k}" for k, v in rows]

    def summary(self):
        out = {}
        for name, flops, e0, e1 in self.rec:
            ms = e0.elapsed_time(e1)
            s = out.setdefault(name, {"launches": 0, "flops": 0.0, "ms": 0.0})
            s["launches"] += 1
            s["flops"] += flops
            s["ms"] += ms
        return out


def source_hash():
    """Hash of what decides a kernel's HBM traffic: the HIP sources, the C ABI and this file.  tools/pmc_traffic.py stores it
    with a PMC profile; a profile taken from other sources is not quoted."""
    import glob
    import hashlib
    h = hashlib.sha256()
    pkg = os.path.join(ROOT, "hybrid-ctunet_amd")
    for f in sorted(glob.glob(os.path.join(pkg, "csrc", "*.hip")) + glob.glob(os.path.join(pkg, "csrc", "*.h")) +
                    [os.path.join(ROOT, "include", "ctunet_hip.h"), os.path.join(pkg, "ops.py"), os.path.join(pkg, "ops_fused.py")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(entry_point):
    """(bytes, source): HBM bytes per launch of the dominant entry point's kernels from the newest committed rocprofv3 PMC
    passes over this very command (profiles/r*_pmc_hbm_traffic.json: FETCH_SIZE x 2 + WRITE_SIZE, separate passes, per
    the guide's gfx950 correction).  Counters cannot be read from inside the process, so this is the last profiled value,
    not a live one: the file records the hash of the sources it was taken from (source_hash) and the commit; when the
    kernels or the host path changed since, the figure is dropped (None) and the source field says so."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))
    path = files[-1] if files else ""
    key = {"ctu_conv3_halo": "conv3_halo_dma_kernel", "ctu_conv3_halo_wgrad": "conv3_halo_wgrad_dma_kernel",
           "ctu_igemm_nt": "gemm_nt_dma_kernel", "ctu_igemm_tn": "gemm_tn_dma_kernel"}.get(entry_point)
    if key is None or not os.path.exists(path):
        return None, None
    prof = json.load(open(path))
    if prof.get("source_hash") != source_hash():
        return None, f"{os.path.basename(path)} (commit {prof.get('commit', '?')}) is older than the sources: not quoted"
    ks = prof["kernels"]
    n = b = 0.0
    for name, v in ks.items():
        if key in name:
            n += v["launches_per_step"]
            b += v["launches_per_step"] * (v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"])
    return (round(b / n) if n else None), f"{os.path.basename(path)} (commit {prof.get('commit', '?')})"


def _cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(model_name, threads):
    """Oracle on the host cores: one 96^3 volume per iteration through fwd + DiceCE + bwd + AdamW in
    fp32, 1 warm-up + 2 timed iterations (the bounded sample: ~3 x 16 s for CTUNet on 16 threads)."""
    from oracle import ctunet_oracle as O
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    m = O.build(model_name)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-5)
    x, y = O.synthetic_batch(1, seed=1000)
    times = []
    for it in range(3):
        t0 = time.time()
        opt.zero_grad(set_to_none=True)
        loss = O.LOSSES[model_name](m(x), y)
        loss.backward()
        opt.step()
        times.append(time.time() - t0)
        if it == 0:
            first_loss = loss.item()
    dt = sum(times[1:]) / 2
    return {"value": 1.0 / dt, "unit": "volumes/s", "cores": threads, "kind": "port", "cpu": _cpu_model_name(),
            "sample": f"1 volume (1x1x96x96x96) per iteration, {model_name} d101 step fwd+DiceCE+bwd+AdamW, fp32 oracle, "
                      f"torch {torch.__version__} CPU, 1 warm-up ({times[0]:.1f} s) + 2 timed ({times[1]:.1f}, {times[2]:.1f} s), "
                      f"first loss {first_loss:.4f}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="ctunet", choices=["ctunet", "cunet", "tunet"])
    ap.add_argument("--batch", type=int, default=2, help="per-GPU batch (BASELINE: 2)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--bucket-mb", type=float, default=32.0)
    ap.add_argument("--payload", default="fp32", choices=["fp32", "bf16"],
                    help="gradient exchange payload for N > 1: fp32 = one ncclAllReduce(ncclAvg) per bucket (torch's RCCL binding); "
                         "bf16 = ctu_allreduce_bucket (cast -> all-to-all -> fp32 sum -> all-gather: half the bytes on every xGMI "
                         "link, csrc/comm.hip) over a communicator bootstrapped from torch.distributed")
    ap.add_argument("--stage-graphs", action="store_true",
                    help="replay the launch-latency-bound stages (ResNet layer3/4, ViT trunk, first window stages) from HIP "
                         "graphs (graphs.graph_stages).  Off by default: measured 50.8 ms / step against 49.4 ms without - "
                         "inside a graph the stage loses the weight-gradient stream overlap, and the launch thread was not "
                         "what limited those stages")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as one HIP graph (train.GraphedStep).  Off by default: on ROCm 7.2 the graph executor "
                         "serialises the captured streams (58 ms / step against 49 ms enqueued eagerly on four streams)")
    ap.add_argument("--opt-overlap", dest="opt_overlap", action="store_true", default=True,
                    help="per-bucket AdamW updates queued under the backward pass (FusedAdamW(overlap=True)) instead of one launch "
                         "after backward(): same arithmetic, element for element.  Default since round 3: with the launch lists the "
                         "host no longer paces the backward pass and the early updates fill its gaps (47.1 / 47.0 against 47.6 / "
                         "47.5 ms per step, two A/B pairs on one box, profiles/r03_bench_ab_stream_options.log)")
    ap.add_argument("--no-opt-overlap", dest="opt_overlap", action="store_false", help="one AdamW launch after backward()")
    ap.add_argument("--enc0-stream", action="store_true",
                    help="A/B: vit_encoder0 on a third HIP stream from the start of the forward pass (measured slower: off)")
    ap.add_argument("--wgrad-stream", action=argparse.BooleanOptionalAction, default=None,
                    help="weight-gradient kernels on companion HIP streams (ops.WGRAD_STREAM).  Default: on at N = 1 (-1.0 to "
                         "-1.5 ms per step), off under data parallelism (train.DataParallel switches them off: with them every "
                         "bucket's event fence costs ~1 ms, profiles/r03_bench_dp_rehearsal_world1.log)")
    ap.add_argument("--route", type=int, default=0,
                    help="A/B measurements: ctu_set_option(\"route\", N) bit set (include/ctunet_hip.h); 0 = the shipped routing")
    ap.add_argument("--serial", action="store_true",
                    help="one HIP stream (no branch / weight-gradient overlap): per-kernel profiles without co-running kernels")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    # rehearsal switches (one-GPU boxes): CTU_BENCH_BACKEND=gloo CTU_BENCH_ONE_GPU=1 run all ranks on cuda:0 and move the
    # buckets with gloo - the same control flow (barriers, instrumented steps, max-over-ranks timing) without RCCL
    one_gpu = bool(os.environ.get("CTU_BENCH_ONE_GPU"))
    backend = os.environ.get("CTU_BENCH_BACKEND", "nccl")
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # CTU_BENCH_FORCE_DP=1 (one-GPU boxes): a one-rank RCCL process group and the DataParallel wrapper at N = 1 - every bucket goes
    # through ncclAllReduce on the side stream, so RCCL's communicator, its streams and the bucket schedule are in the step
    force_mode = os.environ.get("CTU_BENCH_FORCE_DP", "") if world == 1 else ""   # "1": group + wrapper; "pg": group only; "dp": wrapper only
    force_dp = force_mode in ("1", "dp")
    force_pg = force_mode in ("1", "pg")
    if force_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1 or force_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import hybrid_ctunet_amd as H
    from hybrid_ctunet_amd import _lib
    from hybrid_ctunet_amd.synthetic import synthetic_batch

    if a.route:
        _lib.call("ctu_set_option", b"route", a.route)
    torch.manual_seed(0)  # identical default init on every rank (and DataParallel broadcasts rank 0 anyway)
    model = H.build_model(a.model).to(dev)
    n_params = sum(p.numel() for p in model.parameters())
    order = H.gradient_ready_order(model)
    flat = H.FlatParams(order)
    # static_unused: the parameters without a gradient in the first (warm-up) step never get one (seven ResBlock.conv3 that the
    # models build and never call); their buckets then go out during backward instead of behind it (train.DataParallel)
    comm = None
    if (world > 1 or force_dp) and a.payload == "bf16":
        from hybrid_ctunet_amd.comm import Communicator
        comm = Communicator.from_torch()
    dp = H.DataParallel(model, flat=flat, bucket_mb=a.bucket_mb, static_unused=a.warmup > 0, payload=a.payload,
                        comm=comm) if (world > 1 or force_dp) else None
    use_graph = world == 1 and a.graph and not a.serial
    # the optimizer updates a bucket of parameters as soon as its gradients are final (behind the bucket's all-reduce for N > 1):
    # same arithmetic as one update after backward(), queued under the rest of the backward pass (train.FusedAdamW)
    # (N > 1: the per-bucket update behind each bucket's all-reduce - DataParallel.attach_optimizer - has run over gloo on one GPU
    #  only; the first runs over RCCL keep the plain finish() + step() sequence unless --opt-overlap is given explicitly)
    explicit = "--opt-overlap" in sys.argv
    opt_overlap = a.opt_overlap and not (a.serial or use_graph) and (world == 1 or explicit)
    opt = H.FusedAdamW(None, lr=1e-4, weight_decay=1e-5, flat=flat, capturable=use_graph, overlap=opt_overlap and world == 1)
    if dp is not None and opt_overlap:
        dp.attach_optimizer(opt)
    loss_fn = H.LOSSES[a.model]
    x, y = synthetic_batch(a.batch, seed=1000 + rank)
    x, y = x.to(dev), y.to(dev)
    use_bf16 = a.precision == "bf16"
    from hybrid_ctunet_amd import ops as _ops, ops_fused as _fused

    def set_serial(flag):
        """Serial = every kernel on one stream.  The timed region overlaps the two encoder branches and the weight-gradient
        kernels on companion streams; the per-kernel roofline figures are taken with the overlap off, so a launch's HIP
        events bracket that kernel alone."""
        wg = a.wgrad_stream if a.wgrad_stream is not None else dp is None
        _ops.WGRAD_STREAM = bool(wg) and not flag   # (companion streams of the weight-gradient kernels)
        opt.overlap_enabled = not flag
        if hasattr(model, "overlap_branches"):
            model.overlap_branches = not flag
            model.enc0_stream = bool(a.enc0_stream) and not flag
    set_serial(a.serial)

    def step():
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=use_bf16):
            out = model(x)
            loss = loss_fn(out, y)
        loss.backward()
        if dp is not None:
            dp.finish()
        opt.step()
        return loss

    for _ in range(max(a.warmup, 1) if use_graph else a.warmup):
        loss = step()
    if a.warmup > 0 or use_graph:
        opt.freeze_skip_ranges()
    eager_step, graph_note = step, "eager"
    if a.stage_graphs and not a.serial and use_bf16:
        # the launch-latency-bound stages (ResNet layer3 / layer4, ViT trunk, first window stages) as HIP graphs
        loss = None   # (no autograd graph of an earlier step may be alive during a capture: its AccumulateGrad nodes carry streams)
        names = os.environ.get("CTU_STAGES")
        stages = H.graph_stages(model, x, flat=flat, **({"stages": names.split(",")} if names else {}))
        graph_note = f"eager, {len(stages)} small stages replayed from HIP graphs"
        for _ in range(2):
            loss = step()
    if use_graph:
        # The whole step as one HIP graph (train.GraphedStep): same kernels, same streams, no per-launch host work.  If the
        # capture is refused the run continues eagerly and says so.
        try:
            del loss   # (drops the last eager step's autograd graph)
            graphed = H.GraphedStep(step, opt, warmup=2)
            for _ in range(2):
                loss = graphed()
            torch.cuda.synchronize()
            step, graph_note = graphed, "hip-graph replay"
        except Exception as e:  # noqa: BLE001
            print(f"[bench] HIP-graph capture failed ({type(e).__name__}: {e}); continuing eagerly", file=sys.stderr, flush=True)
            graph_note = f"eager (graph capture failed: {type(e).__name__})"
            torch.cuda.synchronize()
            loss = step()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    final_loss = loss.item()

    # After the timed region, per rank: (1) host time to ENQUEUE one step into an empty queue (no back-pressure from a full
    # HIP queue: the launch thread's own cost), (2) with N > 1 the time the compute stream stood waiting for the gradient
    # exchange in DataParallel.finish() (event pair around that wait: communication the backward pass did not hide).
    enq, exposed = [], []
    for _ in range(3):
        fence()
        if dp is not None:
            dp.exposed = []
        h0 = time.perf_counter()
        loss = step()
        enq.append(1e3 * (time.perf_counter() - h0))
        torch.cuda.synchronize()
        if dp is not None and dp.exposed:
            exposed.append(sum(e0.elapsed_time(e1) for e0, e1 in dp.exposed))
    if dp is not None:
        dp.exposed = None
    host_ms = sorted(enq)[1]
    exposed_ms = sorted(exposed)[len(exposed) // 2] if exposed else 0.0
    per_rank = [[round(host_ms, 2), round(exposed_ms, 3)]]
    if world > 1:
        t = torch.tensor([host_ms, exposed_ms], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allr, t)
        per_rank = [[round(v[0].item(), 2), round(v[1].item(), 3)] for v in allr]

    roofline = None
    step = eager_step
    if not a.no_roofline:
        set_serial(True)
        step()   # (workspaces of the single-stream schedule are created outside the instrumented steps)
        torch.cuda.synchronize()
    if not a.no_roofline and rank != 0:
        for _ in range(2):  # the instrumented steps below contain the gradient all-reduce: every rank takes part
            step()
        torch.cuda.synchronize()
    if not a.no_roofline and rank == 0:
        timer = KernelTimer()
        _lib.PROFILER = timer
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        _lib.PROFILER = None
        s = timer.summary()
        if os.environ.get("CTU_BENCH_SHAPES"):
            print("\n".join(timer.by_shape(int(os.environ["CTU_BENCH_SHAPES"]) if os.environ["CTU_BENCH_SHAPES"].isdigit() and int(os.environ["CTU_BENCH_SHAPES"]) > 1 else 25)), file=sys.stderr, flush=True)
        # dominant kernel = the entry point that carries most of the algorithmic work (the 3x3x3 conv: 80 % of the FLOPs,
        # SURVEY.md 2.2 row K1); by time it ties with the 360 mostly HBM-bound plain GEMM launches
        dom = max(s, key=lambda k: s[k]["flops"])
        tot = {k: round(v["ms"] / 2, 3) for k, v in s.items()}
        ach = s[dom]["flops"] / (s[dom]["ms"] * 1e-3) / 1e12
        traffic, traffic_src = pmc_traffic(dom)
        roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "peak_measured": PEAK_BF16_TFLOPS_MEASURED,
                    "frac_of_measured_peak": round(ach / PEAK_BF16_TFLOPS_MEASURED, 4), "kernel": dom,
                    "launches_per_step": s[dom]["launches"] // 2,
                    "avg_launch_ms": round(s[dom]["ms"] / s[dom]["launches"], 4),
                    "igemm_ms_per_step": tot,
                    "note": "HIP events around every implicit-GEMM launch over 2 instrumented single-stream steps after the "
                            "timed region (the timed steps overlap kernels on several streams)"}
    if world > 1:
        dist.barrier()

    if rank == 0:
        vols = a.batch * world * a.steps
        value = vols / elapsed
        whole_path = value * FLOP_PER_VOLUME[a.model] / world / 1e12
        res = {
            "metric": "train volumes/sec (96^3 patches) CTUNet d101 pf8" if a.model == "ctunet"
                      else f"train volumes/sec (96^3 patches) {a.model}",
            "value": round(value, 4), "unit": "volumes/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * elapsed / a.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if use_bf16 else "f32", "data": "synthetic",
            "config": {"workload": f"{a.model} d101 pf8, per-GPU batch {a.batch} x 1x96x96x96, fwd + DiceCE "
                                   f"(deep supervision, on-device targets) + bwd + fused AdamW"
                                   + (" + RCCL bucketed grad all-reduce" if world > 1 else ""),
                       "global_batch": a.batch * world, "parallelism": f"dp{world}", "launch": graph_note, "host_path": "launch lists (ctu_plan_run)" if _fused.ENABLED else "per-op",
                       **({"grad_payload": a.payload} if world > 1 else {}), "optimizer": "per-bucket updates under backward" if opt_overlap else "one update after backward",
                       **({"route": a.route} if a.route else {}),
                       "params_M": round(n_params / 1e6, 2),
                       "final_loss": round(final_loss, 5)},
            "host_enqueue_ms_per_step": max(r[0] for r in per_rank),
            "per_rank": {"host_enqueue_ms": [r[0] for r in per_rank], "exposed_comm_ms": [r[1] for r in per_rank],
                         "note": "after the timed region, median of 3 steps: host time to enqueue one step into an empty queue; "
                                 "time the compute stream waited for the gradient exchange in DataParallel.finish()"},
            "whole_path_tflops_per_gpu": round(whole_path, 2),
            "whole_path_frac_of_mfma_peak": round(whole_path / PEAK_BF16_TFLOPS, 4),
        }
        if roofline is not None:
            res["roofline"] = roofline
        if world == 1 and not a.no_cpu_baseline:
            try:
                threads = len(os.sched_getaffinity(0))
            except AttributeError:
                threads = os.cpu_count() or 1
            threads = max(1, min(threads, 16))  # the one-GPU box's CPU share
            res["cpu_baseline"] = cpu_baseline(a.model, threads)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
