"""cProfile of the launch thread over five eager CTUNet steps (forward side only: autograd runs the backward Functions on its own
thread, where cProfile does not look): which Python frames the ~40 ms of host time per step are spent in."""
import sys, os, cProfile, pstats, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import hybrid_ctunet_amd as H
from hybrid_ctunet_amd.synthetic import synthetic_batch
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = H.build_model("ctunet").to(dev)
flat = H.FlatParams(H.gradient_ready_order(model))
opt = H.FusedAdamW(None, lr=1e-4, weight_decay=1e-5, flat=flat)
x, y = synthetic_batch(2, seed=1000)
x, y = x.to(dev), y.to(dev)
def step():
    opt.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(x)
        loss = H.ctunet_loss(out, y)
    loss.backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5): step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(70)
print(s.getvalue()[:14000])
