"""Where does a wave of the halo convolution spend its cycles?  Runs the STAMP build (ctu_set_option("nt_debug", 16)) of
ctu_conv3_halo on the large shapes and prints, per wave index, the median cycles before the first barrier opens, waiting at
stage barriers, working between barriers and in the epilogue."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import hybrid_ctunet_amd  # noqa: F401
from hybrid_ctunet_amd import ops
from hybrid_ctunet_amd._lib import call, dcode, ptr, stream
dev = "cuda"
for (B, D, H, W, C, N) in ([] if os.environ.get('STAMPS_WGRAD_ONLY') else [(2, 96, 96, 96, 64, 64), (2, 48, 48, 96, 128, 128), (2, 24, 24, 48, 256, 256)]):
    x = torch.randn(B, D, H, W, C, device=dev, dtype=torch.bfloat16)
    w = torch.nn.Parameter(torch.randn(N, C, 3, 3, 3, device=dev) * 0.05)
    wfr = ops._pack_frag(w, N, C, 27, C * 27, 27, 1, 0, torch.bfloat16)
    out = torch.empty(B, D, H, W, N, device=dev, dtype=torch.bfloat16)
    ws = torch.zeros(1 << 24, device=dev, dtype=torch.float32)
    for dbg in (16, 16 | 8, 16 | 4, 16 | 12):
        call("ctu_set_option", b"nt_debug", dbg)
        for _ in range(3):
            call("ctu_conv3_halo", dcode(x.dtype), ptr(x), None, ptr(wfr), ptr(out), None, B, D, H, W, C, 0, N, 0, N, 0, None, None, None,
                 ptr(ws), ws.numel(), 0, stream())
        torch.cuda.synchronize()
        call("ctu_set_option", b"nt_debug", 0)
        ntn = N // 32
        NT = 4 if ntn % 4 == 0 else 2
        nwg = B * ((D + 3) // 4) * ((H + 7) // 8) * ((W + 7) // 8) * (ntn // NT)
        st = ws.view(torch.int64)[:nwg * 16].view(nwg, 4, 4).double().cpu()
        med = st.median(dim=0).values
        tot = med.sum(dim=1)
        tag = {16: "full", 24: "no halo DMA", 20: "no weight DMA", 28: "no DMA"}[dbg]
        print(f"{C}->{N} @{D}x{H}x{W} [{tag:13s}] median cycles per wave: " + "  ".join(
            f"w{i}: pro {med[i,0]:7.0f} wait {med[i,1]:7.0f} work {med[i,2]:7.0f} epi {med[i,3]:6.0f} (wait {100*med[i,1]/tot[i]:4.1f} %)" for i in range(4)))


# weight-gradient kernel: ticks until the first operands have landed, waiting for own DMA, at barriers, in the k loops, epilogue
print("weight gradient (median clock ticks per wave over all workgroups)")
for (B, D, H, W, C, N) in [(2, 96, 96, 96, 64, 64), (2, 48, 48, 96, 128, 128), (2, 24, 24, 48, 256, 256)]:
    x = torch.randn(B, D, H, W, C, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(B, D, H, W, N, device=dev, dtype=torch.bfloat16)
    panel = torch.zeros(27, N, C, device=dev)
    ws = torch.zeros(256 * 54 * 1024 + 27 * 64 * 1024, device=dev, dtype=torch.float32)
    tiles = ((N + 63) // 64) * (C // 32)
    nbricks = B * ((D + 3) // 4) * ((H + 7) // 8) * ((W + 7) // 8)
    splits = min((256 + tiles - 1) // tiles, nbricks)
    bpb = (nbricks + splits - 1) // splits
    splits = (nbricks + bpb - 1) // bpb
    for dbg in (16, 16 | 4):
        call("ctu_set_option", b"nt_debug", dbg)
        for _ in range(int(os.environ.get("STAMPS_LAUNCHES", "3"))):  # (hundreds: the clock the governor settles on under this load)
            call("ctu_conv3_halo_wgrad", dcode(x.dtype), ptr(dy), ptr(x), None, ptr(panel), B, D, H, W, C, 0, N, 0, 0, ptr(ws), ws.numel(),
                 stream())
        torch.cuda.synchronize()
        call("ctu_set_option", b"nt_debug", 0)
        nwg = tiles * splits
        raw = ws[splits * 27 * N * C:].view(torch.int64)[:nwg * 64].view(nwg, 8, 8).double().cpu()
        st = raw[:, :, :5]
        med = st.median(dim=0).values
        ghz = (raw[:, 0, 5] / raw[:, 0, 6] * 0.1).median().item()
        start = (raw[:, 0, 7] - raw[:, 0, 7].min()) * 0.01   # us
        dur = raw[:, 0, 6] * 0.01
        end = start + dur
        q = lambda t, f: t.sort().values[int(f * (len(t) - 1))].item()
        tag = {16: "full", 20: "no operand DMA", 17: "no epilogue"}[dbg]
        if dbg & 1:
            continue  # (the no-epilogue build returns before the stamps are written)
        print(f"{C}->{N} @{D}x{H}x{W} [{tag:14s}] ({bpb} bricks per workgroup; shader clock while the kernel ran: {ghz:.2f} GHz)")
        print(f"    workgroups: start spread {q(start, 0.5):.1f} / {q(start, 0.9):.1f} / {start.max().item():.1f} us (median / p90 / max), "
              f"duration {q(dur, 0.1):.1f} / {q(dur, 0.5):.1f} / {q(dur, 0.9):.1f} / {dur.max().item():.1f} us (p10 / median / p90 / max), "
              f"last end {end.max().item():.1f} us")
        for wv in range(8):
            m = med[wv]
            print(f"    wave {wv}: first operands {m[0]:6.0f}  own DMA {m[1]:6.0f}  barriers {m[2]:6.0f}  k loops {m[3]:6.0f}  epilogue {m[4]:6.0f}")
