"""Loss curve of the benchmark configuration (CTUNet d101 pf8, batch 2, synthetic volumes, fused DiceCE + AdamW lr 1e-4) over 40
steps in bf16 (the bench path) and, from the same initial state, over the first 12 steps in fp32 parity mode: the two must track
each other and the loss must fall - end-to-end evidence that forward, loss, backward (all stashes / sinks / streams) and the
optimizer update fit together at the size the bench measures."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import hybrid_ctunet_amd as H
from hybrid_ctunet_amd.synthetic import synthetic_batch
dev = torch.device("cuda", 0)
x, y = synthetic_batch(2, seed=1000)
x, y = x.to(dev), y.to(dev)


def run(precision, steps):
    torch.manual_seed(0)
    model = H.build_model("ctunet").to(dev)
    if precision == "fp32":
        model.set_precision("fp32")
    flat = H.FlatParams(H.gradient_ready_order(model))
    opt = H.FusedAdamW(None, lr=1e-4, weight_decay=1e-5, flat=flat)
    out = []
    for i in range(steps):
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=precision == "bf16"):
            loss = H.ctunet_loss(model(x), y)
        loss.backward()
        opt.step()
        out.append(loss.item())
    flat.release()
    return out


b = run("bf16", 40)
f = run("fp32", 12)
print("step   bf16      fp32")
for i, v in enumerate(b):
    print(f"{i:4d}  {v:8.5f}  {f[i]:8.5f}" if i < len(f) else f"{i:4d}  {v:8.5f}")
assert b[-1] < 0.9 * b[0] and all(v < u for u, v in zip(b, b[1:])), "the loss does not fall step by step"
assert all(abs(u - v) <= 0.03 * abs(v) for u, v in zip(b, f)), "bf16 and fp32 trajectories diverge"
print("ok")
