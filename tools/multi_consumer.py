"""Which tensors of the step receive their gradient from more than one consumer (autograd then sums the contributions with an
aten::add_ pass over three tensors)?  Walks the autograd graph of one CTUNet step, counts the edges into every (node, output) and
prints the nodes with two or more, with the shape of the gradient they receive."""
import os, sys, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import hybrid_ctunet_amd as H
from hybrid_ctunet_amd.synthetic import synthetic_batch
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = H.build_model(sys.argv[1] if len(sys.argv) > 1 else "ctunet").to(dev)
x, y = synthetic_batch(2, seed=1000)
x, y = x.to(dev), y.to(dev)
node2mod = {}
def fh(name):
    def hook(mod, inp, out):
        outs = out if isinstance(out, (tuple, list)) else (out,)
        for o in outs:
            if torch.is_tensor(o) and o.grad_fn is not None:
                node2mod.setdefault(o.grad_fn, name)   # inner modules fire first: the innermost owner wins
    return hook
for name, mod in model.named_modules():
    mod.register_forward_hook(fh(name))
with torch.autocast("cuda", dtype=torch.bfloat16):
    out = model(x)
    loss = H.LOSSES[sys.argv[1] if len(sys.argv) > 1 else "ctunet"](out, y)
edges = collections.Counter()
parents = collections.defaultdict(list)
seen, stack = set(), [loss.grad_fn]
while stack:
    n = stack.pop()
    if n is None or n in seen:
        continue
    seen.add(n)
    for nxt, idx in n.next_functions:
        if nxt is not None:
            edges[(nxt, idx)] += 1
            parents[(nxt, idx)].append(type(n).__name__.replace("Backward", "") + "@" + node2mod.get(n, "?"))
            stack.append(nxt)
multi = {k: v for k, v in edges.items() if v >= 2 and "AccumulateGrad" not in type(k[0]).__name__}
print(f"{len(multi)} (node, output) pairs with two or more consumers")
rows = []
def mk(node, idx, cnt):
    def hook(grad_inputs, grad_outputs):
        g = grad_outputs[idx] if idx < len(grad_outputs) else None
        rows.append((g.numel() if g is not None else 0, type(node).__name__ + "@" + node2mod.get(node, "?"), cnt, tuple(g.shape) if g is not None else None,
                     sorted(parents[(node, idx)])))
    return hook
for (node, idx), cnt in multi.items():
    node.register_hook(mk(node, idx, cnt))
loss.backward()
torch.cuda.synchronize()
for numel, name, cnt, shape, par in sorted(rows, key=lambda r: -r[0])[:60]:
    if sum(1 for q in par if not q.startswith("GradStash")) <= 1:
        continue   # the other gradients are parked and added inside the remaining consumer's data-gradient kernel: no pass
    print(f"{numel * 2 / 1e6:8.1f} MB  {name}  {shape}\n             <- {par}")
