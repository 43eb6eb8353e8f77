import sys, os, collections
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
for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.name in ("aten::add", "aten::add_", "aten::copy_", "aten::contiguous", "aten::clone", "aten::zeros", "aten::fill_", "aten::zero_", "aten::mul", "aten::to", "aten::_to_copy"):
        k = (e.name, str(e.input_shapes)[:90])
        agg[k][0] += 1
        agg[k][1] += e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{v[1]:9.1f} us  x{v[0]:3d}  {k[0]:16s} {k[1]}")
