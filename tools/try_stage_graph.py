"""python tools/try_stage_graph.py <stage> [<stage> ...]: capture the named stages, run two steps, print the losses."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import hybrid_ctunet_amd as H
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = H.build_model("ctunet").to(dev)
flat = H.FlatParams(H.gradient_ready_order(model))
opt = H.FusedAdamW(None, lr=1e-4, weight_decay=1e-5, flat=flat)
x, y = H.synthetic_batch(2, seed=1000)
x, y = x.to(dev), y.to(dev)
def step():
    opt.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = H.ctunet_loss(model(x), y)
    loss.backward()
    opt.step()
    return loss
l0 = [float(step()) for _ in range(2)]
opt.freeze_skip_ranges()
print("eager", l0, flush=True)
st = H.graph_stages(model, x, stages=sys.argv[1:], flat=flat)
print("captured", len(st), flush=True)
print("graphed", [float(step()) for _ in range(3)], flush=True)
