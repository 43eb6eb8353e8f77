"""Repeat the InstanceNorm-sum variant of gemm_nt_stream 150 times per shape against the general kernel: outputs must be
bit-equal every time (a rare race would show as a sporadic mismatch), the sums equal to fp32 summation order."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from hybrid_ctunet_amd import ops, _lib
torch.manual_seed(0)
bad = 0
for (M, K, N, B) in [(442368, 128, 512, 2), (442368, 32, 128, 2), (55296, 64, 256, 2), (36864 * 3, 128, 128, 3)]:
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    ref = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    _lib.call("ctu_set_option", b"route", 1)
    acc0 = torch.zeros(B * N * 2, device="cuda", dtype=torch.float64)
    ops._plain_gemm(x, w, ref, M, K, N, in_acc=acc0, in_rows=M // B)
    _lib.call("ctu_set_option", b"route", 0)
    torch.cuda.synchronize()
    for it in range(150):
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        acc = torch.zeros(B * N * 2, device="cuda", dtype=torch.float64)
        ops._plain_gemm(x, w, out, M, K, N, in_acc=acc, in_rows=M // B)
        if not torch.equal(out, ref):
            bad += 1
            print("MISMATCH", M, K, N, it, (out.float() - ref.float()).abs().max().item())
        if not torch.allclose(acc, acc0, rtol=1e-6, atol=1e-3):   # fp32 partial sums in another order
            bad += 1
            print("STATS MISMATCH", M, K, N, it, (acc - acc0).abs().max().item())
    print("shape", M, K, N, "ok")
print("bad =", bad)
