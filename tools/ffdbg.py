import math, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import hybrid_ctunet_amd
from hybrid_ctunet_amd._lib import call, dcode, ptr, stream
torch.manual_seed(0)
M, D, Hd = 512, 128, 512
dt = torch.bfloat16
x = (torch.randn(M, D) * 1.0).to(dt).cuda()
g = torch.ones(D).cuda(); b = torch.zeros(D).cuda()
w1 = (torch.randn(Hd, D) / math.sqrt(D)).to(dt).cuda(); b1 = torch.zeros(Hd).cuda()
w2 = (torch.randn(D, Hd) / math.sqrt(Hd)).to(dt).cuda(); b2 = torch.zeros(D).cuda()
w2f = torch.empty(D * Hd, device="cuda", dtype=dt)
call("ctu_ff_pack_w2", ptr(w2), ptr(w2f), D, Hd, stream())
y = torch.zeros(M, D, device="cuda", dtype=dt); pre = torch.full((M, Hd), 7.0, device="cuda", dtype=dt); u = torch.full((M, Hd), 7.0, device="cuda", dtype=dt)
mr = torch.empty(M, 2, device="cuda")
call("ctu_ff_fwd", dcode(dt), ptr(x), ptr(g), ptr(b), ptr(w1), ptr(b1), ptr(w2f), ptr(b2), ptr(y), ptr(pre), ptr(u), ptr(mr), M, D, Hd, stream())
torch.cuda.synchronize()
xd = x.double()
h = torch.nn.functional.layer_norm(xd, (D,)).to(dt).double()
pre_ref = h @ w1.double().t()
err = (pre.double() - pre_ref).abs()
bad = err > 0.05
print("bad fraction", bad.float().mean().item(), "untouched (==7)", (pre == 7).float().mean().item())
print("bad per 64-col chunk", bad.view(M, 8, 64).float().mean((0, 2)).tolist())
print("bad per 32-row group", bad.view(16, 32, Hd).float().mean((1, 2)).tolist())
print("bad per col mod 32 (first chunk)", bad[:, :64].float().mean(0).view(2, 32).tolist())
u_ref = torch.nn.functional.gelu(pre_ref)
print("u bad", ((u.double() - u_ref).abs() > 0.05).float().mean().item())
y_ref = xd + u_ref.to(dt).double() @ w2.double().t()
print("y bad", ((y.double() - y_ref).abs() > 0.1).float().mean().item(), (y.double()-y_ref).abs().max().item())
print(pre[0, :16].tolist()); print(pre_ref[0, :16].tolist())
print("bad per row mod 32", bad.view(16, 32, Hd).float().mean((0, 2)).tolist())
rows = bad.any(1).nonzero().flatten()[:3].tolist()
for rr in rows:
    cols = bad[rr].nonzero().flatten()[:8].tolist()
    print("row", rr, "cols", cols, "got", [pre[rr, c].item() for c in cols], "ref", [round(pre_ref[rr, c].item(), 3) for c in cols],
          "u at those", [u[rr, c].item() for c in cols])
