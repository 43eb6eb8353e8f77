"""Run-to-run bit equality of kernels whose outputs involve no atomics (a race would show as a sporadic difference):
3x3x3 halo convs forward / data gradient, the 64 x 64 x 128-deep trunk GEMMs, the channel-split small-volume conv."""
import math
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa: E402

from hybrid_ctunet_amd import ops  # noqa: E402

torch.manual_seed(0)
bad = 0
dt = torch.bfloat16
for (B, D, H, W, C, N, reps) in [(2, 48, 48, 96, 128, 128, 60), (2, 96, 96, 96, 64, 64, 40), (2, 12, 12, 24, 512, 512, 100),
                                 (2, 48, 48, 96, 32, 32, 100), (1, 20, 22, 30, 64, 160, 100)]:
    x = torch.randn(B, D, H, W, C, device="cuda").to(dt).requires_grad_(True)
    w = (torch.randn(N, C, 3, 3, 3, device="cuda") / math.sqrt(27 * C)).requires_grad_(True)
    gy = torch.randn(B, D, H, W, N, device="cuda").to(dt)
    ref = None
    for it in range(reps):
        x.grad = None
        y = ops.conv3d(x, w, 1, 1)
        y.backward(gy)
        cur = (y.detach().clone(), x.grad.clone())
        if ref is None:
            ref = cur
        elif not (torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1])):
            bad += 1
            print("CONV MISMATCH", (B, D, H, W, C, N), it)
    print("conv", (B, D, H, W, C, N), "ok")
for (M, K, N) in [(864, 3072, 768), (864, 768, 768), (864, 768, 3072)]:
    x = torch.randn(M, K, device="cuda").to(dt)
    w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(dt)
    b = torch.randn(N, device="cuda")
    ref = None
    for it in range(200):
        out = torch.empty(M, N, device="cuda", dtype=dt)
        ops._plain_gemm(x, w, out, M, K, N, bias=b, act=1)
        if ref is None:
            ref = out
        elif not torch.equal(out, ref):
            bad += 1
            print("GEMM MISMATCH", (M, K, N), it)
    print("gemm", (M, K, N), "ok")
print("bad =", bad)
