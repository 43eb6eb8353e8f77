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
for (B, D, H, W, C, N) in [(2, 96, 96, 96, 64, 64), (2, 48, 48, 96, 128, 128), (2, 24, 24, 48, 256, 256)]:
    x = torch.randn(B, D, H, W, C, device=dev, dtype=torch.bfloat16)
    w = torch.nn.Parameter(torch.randn(N, C, 3, 3, 3, device=dev) * 0.05)
    wfr = ops._pack_frag(w, N, C, 27, C * 27, 27, 1, 0, torch.bfloat16)
    out = torch.empty(B, D, H, W, N, device=dev, dtype=torch.bfloat16)
    ws = torch.zeros(1 << 24, device=dev, dtype=torch.float32)
    for dbg in (16, 16 | 8, 16 | 4, 16 | 12):
        call("ctu_set_option", b"nt_debug", dbg)
        for _ in range(3):
            call("ctu_conv3_halo", dcode(x.dtype), ptr(x), None, ptr(wfr), ptr(out), None, B, D, H, W, C, 0, N, 0, N, 0, None, None,
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
