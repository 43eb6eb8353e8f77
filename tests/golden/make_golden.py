#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing the REFERENCE's own networks/*.py unchanged.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py [--skip-models]

What it does
  * puts the test-only MONAI stand-in (tests/golden/_monai_standin, see its docstring: parity unpinned at the
    MONAI wrapper boundary) and /root/reference on sys.path and imports networks.hybrid_CTUNet / resnet / vit;
  * loads deterministic synthetic weights (oracle.ctunet_oracle.synthetic_tensor, keyed by state_dict name);
  * writes
      manifest_<model>.json   state_dict key -> shape  (412 / 126 / 126 / 235 tensors)
      blocks.npz              block-level goldens at tiny shapes: inputs, outputs, input-grads, weight-grads
      model_<model>.npz       whole-model 96^3 goldens: 4096 sampled logits per output + moments, per-sample loss,
                              B=2 loss and per-parameter gradient norms
      loss.npz                scipy.ndimage.zoom(order=0) index maps + DiceCE values on small random logits
      model_<model>_bf16.npz  (--mode bf16) the reference under torch.autocast("cpu", bfloat16): the same samples
      model_<model>_vox{32,64}.npz  (--mode f32|f64) whole class vectors of 2048 voxels per output + Dice / CE terms
Nothing from the reference's source text is stored: only numbers it computed.
"""
import argparse
import json
import os
import sys
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "_monai_standin"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import scipy.ndimage as ndimage  # noqa: E402
import torch  # noqa: E402

from networks import hybrid_CTUNet as R  # noqa: E402  (the reference)
from networks import resnet as RR  # noqa: E402
from networks import vit as RV  # noqa: E402
from oracle import ctunet_oracle as O  # noqa: E402

torch.set_num_threads(os.cpu_count() or 8)
KW = dict(in_channels=1, dim_conv_stem=64, out_channels=14, img_size=(96, 96), frames=96, patch_frame=8)
N_SAMPLES = 4096


def flat_outputs(o):
    return [t for g in o for t in (g if isinstance(g, tuple) else (g,))]


def load_synth(module, prefix):
    sd = {k: O.synthetic_tensor(f"{prefix}.{k}", v.shape) for k, v in module.state_dict().items()}
    module.load_state_dict(sd, strict=True)
    return sd


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def run_block(out, name, module, inputs, seed):
    """forward + backward with a fixed upstream gradient; store everything under '<name>/...'."""
    load_synth(module, name)
    ins = [t.clone().requires_grad_(True) for t in inputs]
    y = module(*ins)
    gy = rnd(y.shape, seed + 77)
    y.backward(gy)
    for i, t in enumerate(ins):
        out[f"{name}/in{i}"] = t.detach().numpy()
        out[f"{name}/gin{i}"] = t.grad.numpy()
    out[f"{name}/out"] = y.detach().numpy()
    out[f"{name}/gout"] = gy.numpy()
    for k, p in module.named_parameters():
        out[f"{name}/gw/{k}"] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
        out[f"{name}/gw_isnone/{k}"] = np.array(p.grad is None)


class _Win(torch.nn.Module):
    """partition -> Residual(MultiAxisAttention) -> Residual(FeedForward) -> un-partition, built from the reference's
    own classes exactly as hybrid_CTUNet.py:558-567 composes them."""

    def __init__(self, dim, mode):
        super().__init__()
        w = 6
        if mode == "block":
            a = R.Rearrange('b c (h h1) (w w1) (f f1) -> b h w f h1 w1 f1 c', h1=w, w1=w, f1=w)
            b = R.Rearrange('b h w f h1 w1 f1 c -> b c (h h1) (w w1) (f f1)')
        else:
            a = R.Rearrange('b c (h1 h) (w1 w) (f1 f) -> b h w f h1 w1 f1 c', h1=w, w1=w, f1=w)
            b = R.Rearrange('b h w f h1 w1 f1 c -> b c (h1 h) (w1 w) (f1 f)')
        self.seq = torch.nn.Sequential(a, R.Residual(R.MultiAxisAttention(dim=dim, dim_head=32, dropout=0.0, window_size=w)),
                                       R.Residual(R.FeedForward(dim, dropout=0.0)), b)

    def forward(self, x):
        return self.seq(x)


def make_blocks():
    out = {}
    run_block(out, "resblock_same", R.ResBlock(3, 16, 16, 3, 1, "instance"), [rnd((2, 16, 6, 8, 8), 1)], 1)
    run_block(out, "resblock_proj", R.ResBlock(3, 32, 16, 3, 1, "instance"), [rnd((1, 32, 8, 6, 8), 2)], 2)
    run_block(out, "resblock_in1", R.ResBlock(3, 1, 64, 3, 1, "instance"), [rnd((1, 1, 8, 8, 12), 3)], 3)  # N=64: the width every Cin=1 conv of the models has
    ds = torch.nn.Sequential(RR.get_conv_layer(3, 32, 64, kernel_size=1, stride=(2, 2, 2), conv_only=True),
                             RR.get_norm_layer(name="instance", spatial_dims=3, channels=64))
    run_block(out, "bottleneck_s2", RR.Bottleneck(32, 16, stride=(2, 2, 2), downsample=ds), [rnd((1, 32, 8, 8, 8), 4)], 4)
    run_block(out, "bottleneck_id", RR.Bottleneck(64, 16), [rnd((2, 64, 4, 6, 8), 5)], 5)
    run_block(out, "stem", RR.get_conv_layer(3, 1, 64, kernel_size=(7, 7, 7), stride=(2, 2, 1), conv_only=True),
              [rnd((1, 1, 12, 12, 10), 6)], 6)
    run_block(out, "convt222", RR.get_conv_layer(3, 32, 16, kernel_size=(2, 2, 2), stride=(2, 2, 2), conv_only=True,
                                                 is_transposed=True), [rnd((1, 32, 3, 4, 5), 7)], 7)
    run_block(out, "convt221", RR.get_conv_layer(3, 32, 16, kernel_size=(2, 2, 1), stride=(2, 2, 1), conv_only=True,
                                                 is_transposed=True), [rnd((2, 32, 3, 4, 5), 8)], 8)
    run_block(out, "upcat", R.UpCatConvBlock(3, 32, 16, 3, (2, 2, 2), "instance"),
              [rnd((1, 32, 3, 3, 4), 9), rnd((1, 16, 6, 6, 8), 10)], 9)
    run_block(out, "pwa", R.pixelweight_attention(64), [rnd((2, 64, 3, 4, 5), 11), rnd((2, 64, 3, 4, 5), 12)], 11)
    run_block(out, "fusion", R.Up_2Fusion_Block(3, 64, 32, 3, (2, 2, 2), "instance"),
              [rnd((1, 64, 2, 3, 3), 13), rnd((1, 32, 4, 6, 6), 14), rnd((1, 32, 4, 6, 6), 15)], 13)
    run_block(out, "win_block", _Win(64, "block"), [rnd((1, 64, 6, 12, 12), 16)], 16)
    run_block(out, "win_grid", _Win(64, "grid"), [rnd((1, 64, 6, 12, 12), 17)], 17)
    run_block(out, "pixelshuffle222", R.PixelShuffle(3, (2, 2, 2), 64, 24), [rnd((2, 64, 2, 3, 4), 18)], 18)
    run_block(out, "pixelshuffle221", R.PixelShuffle(3, (2, 2, 1), 32, 16), [rnd((1, 32, 3, 2, 5), 19)], 19)
    run_block(out, "feedforward", R.FeedForward(32, dropout=0.0), [rnd((2, 3, 4, 5, 32), 20)], 20)
    run_block(out, "vit_block", RV.TransformerBlock(64, 2, 32, 128), [rnd((2, 50, 64), 21)], 21)
    small_vit = RV.ViT(image_size=(32, 32), image_patch_size=16, frames=16, frame_patch_size=8, dim=64, depth=2,
                       heads=2, mlp_dim=128, dim_head=32)
    run_block(out, "vit_small", small_vit, [rnd((2, 1, 32, 32, 16), 22)], 22)
    np.savez_compressed(os.path.join(HERE, "blocks.npz"), **out)
    print("blocks.npz:", len(out), "arrays,", sum(v.nbytes for v in out.values()) / 1e6, "MB raw")


def make_loss():
    out = {}
    for n_in, z in ((96, 0.5), (96, 0.25), (48, 0.5), (24, 0.5), (10, 0.5), (7, 0.5)):
        a = np.arange(n_in, dtype=np.float64)
        out[f"zoom/{n_in}/{z}"] = ndimage.zoom(a, z, order=0, prefilter=False).astype(np.int64)
    g = torch.Generator().manual_seed(5)
    t = torch.randint(0, 14, (2, 1, 96, 96, 96), generator=g).float()
    t1 = torch.from_numpy(ndimage.zoom(t.numpy(), (1, 1, 0.5, 0.5, 1), order=0, prefilter=False))
    t2 = torch.from_numpy(ndimage.zoom(t.numpy(), (1, 1, 0.25, 0.25, 0.5), order=0, prefilter=False))
    idx = torch.randint(0, t1.numel(), (2048,), generator=g)
    out["zoom3d/t_seed5_half_idx"] = idx.numpy()
    out["zoom3d/t_seed5_half_val"] = t1.flatten()[idx].numpy()
    idx2 = torch.randint(0, t2.numel(), (2048,), generator=g)
    out["zoom3d/t_seed5_quarter_idx"] = idx2.numpy()
    out["zoom3d/t_seed5_quarter_val"] = t2.flatten()[idx2].numpy()
    out["zoom3d/shapes"] = np.array([list(t1.shape), list(t2.shape)])
    # DiceCE on small logits: values from the oracle restatement in float64 ("parity unpinned": MONAI absent)
    lg = rnd((2, 14, 6, 8, 10), 31, 3.0).double().requires_grad_(True)
    tg = torch.randint(0, 14, (2, 1, 6, 8, 10), generator=g).float()
    loss, dice, ce = O.dice_ce_loss(lg, tg, return_parts=True)
    loss.backward()
    out["dicece/logits"] = lg.detach().numpy()
    out["dicece/target"] = tg.numpy()
    out["dicece/loss_dice_ce"] = np.array([loss.item(), dice.item(), ce.item()])
    out["dicece/grad"] = lg.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "loss.npz"), **out)
    print("loss.npz written")


MODELS = {
    "cunet50": (lambda: R.CUNet(out_channels=14, model_depth=50), "cunet"),
    "cunet101": (lambda: R.CUNet(out_channels=14, model_depth=101), "cunet"),
    "tunet": (lambda: R.TUNet(**KW), "tunet"),
    "ctunet101": (lambda: R.CTUNet(model_depth=101, **KW), "ctunet"),
}


def make_model(name):
    t0 = time.time()
    ctor, loss_name = MODELS[name]
    model = ctor()
    manifest = {k: list(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(HERE, f"manifest_{name}.json"), "w") as f:
        json.dump(manifest, f, indent=0)
    model.load_state_dict({k: O.synthetic_tensor(k, s) for k, s in manifest.items()}, strict=True)
    out = {}
    losses = []
    for s in range(2):
        x, y = O.synthetic_batch(1, seed=1000 + s)
        outs = flat_outputs(model(x))
        loss = O.LOSSES[loss_name](model_outputs_regroup(name, outs), y)
        (0.5 * loss).backward()  # B=2 batch == mean of two B=1 samples (InstanceNorm is per-sample)
        losses.append(loss.item())
        for i, o in enumerate(outs):
            g = torch.Generator().manual_seed(4242 + i)
            idx = torch.randint(0, o.numel(), (N_SAMPLES,), generator=g)
            of = o.detach().flatten()
            out[f"s{s}/out{i}/idx"] = idx.numpy()
            out[f"s{s}/out{i}/val"] = of[idx].numpy()
            out[f"s{s}/out{i}/shape"] = np.array(o.shape)
            out[f"s{s}/out{i}/moments"] = np.array([of.mean().item(), of.std().item(), of.abs().max().item()])
        del outs, loss
    out["loss_per_sample"] = np.array(losses)
    out["loss_b2"] = np.array(sum(losses) / 2)
    keys, norms, isnone = [], [], []
    for k, p in model.named_parameters():
        keys.append(k)
        isnone.append(p.grad is None)
        norms.append(0.0 if p.grad is None else p.grad.double().norm().item())
    out["grad/keys"] = np.array(keys)
    out["grad/norm_b2"] = np.array(norms)
    out["grad/isnone"] = np.array(isnone)
    # a few sampled gradient entries of the largest tensors, for a sharper check than norms alone
    big = sorted(((p.numel(), k) for k, p in model.named_parameters() if p.grad is not None), reverse=True)[:8]
    for j, (_, k) in enumerate(big):
        p = dict(model.named_parameters())[k]
        g = torch.Generator().manual_seed(99 + j)
        idx = torch.randint(0, p.numel(), (256,), generator=g)
        out[f"grad/sample{j}/key"] = np.array(k)
        out[f"grad/sample{j}/idx"] = idx.numpy()
        out[f"grad/sample{j}/val"] = p.grad.flatten()[idx].numpy()
    np.savez_compressed(os.path.join(HERE, f"model_{name}.npz"), **out)
    print(f"model_{name}.npz: losses {losses}, {len(manifest)} keys, {time.time() - t0:.0f}s", flush=True)


def make_model_f64(name):
    """The same reference modules run in float64 (`model.double()`): the 'exact arithmetic' value of the reference's
    algorithm at the same sample points as model_<name>.npz.  This deep InstanceNorm network amplifies fp32 rounding
    noise ~1000x (the fp32 reference itself is 2-4e-4 away from these values on CUNet-101), so the GPU parity tests
    gate on the distance to THESE numbers and report the distance to the fp32 goldens beside it."""
    t0 = time.time()
    ctor, loss_name = MODELS[name]
    z32 = np.load(os.path.join(HERE, f"model_{name}.npz"), allow_pickle=False)
    model = ctor()
    model.load_state_dict({k: O.synthetic_tensor(k, v.shape) for k, v in model.state_dict().items()}, strict=True)
    model = model.double()
    out, losses = {}, []
    for s in range(2):
        x, y = O.synthetic_batch(1, seed=1000 + s)
        outs = flat_outputs(model(x.double()))
        loss = _loss64(loss_name, model_outputs_regroup(name, outs), y)
        (0.5 * loss).backward()
        losses.append(loss.item())
        for i, o in enumerate(outs):
            idx = torch.from_numpy(z32[f"s{s}/out{i}/idx"])
            out[f"s{s}/out{i}/val64"] = o.detach().flatten()[idx].numpy()
        del outs, loss
    out["loss_per_sample64"] = np.array(losses)
    out["loss_b2_64"] = np.array(sum(losses) / 2)
    pr = dict(model.named_parameters())
    out["grad/norm_b2_64"] = np.array([0.0 if pr[str(k)].grad is None else pr[str(k)].grad.norm().item()
                                       for k in z32["grad/keys"]])
    for j in range(8):
        k = str(z32[f"grad/sample{j}/key"])
        out[f"grad/sample{j}/val64"] = pr[k].grad.flatten()[torch.from_numpy(z32[f"grad/sample{j}/idx"])].numpy()
    np.savez_compressed(os.path.join(HERE, f"model_{name}_f64.npz"), **out)
    print(f"model_{name}_f64.npz: losses {losses}, {time.time() - t0:.0f}s", flush=True)


N_VOX = 2048


def _dice_ce_parts64(logits, target):
    """(Dice term, CE term) of SURVEY 8a row H evaluated in float64 from the given logits (any dtype)."""
    import torch.nn.functional as F
    lg = logits.detach().double()
    n_cls = lg.shape[1]
    labels = target.squeeze(1).long()
    p = torch.softmax(lg, dim=1)
    yy = F.one_hot(labels, n_cls).permute(0, 4, 1, 2, 3).to(p.dtype)
    inter = (p * yy).sum((2, 3, 4))
    denom = (yy * yy).sum((2, 3, 4)) + (p * p).sum((2, 3, 4))
    return (1.0 - 2.0 * inter / (denom + 1e-6)).mean().item(), F.cross_entropy(lg, labels).item()


def _targets_for(name, y):
    t1 = O.downsample_target(y, (0.5, 0.5, 1.0))
    t2 = O.downsample_target(y, (0.25, 0.25, 0.5))
    if name.startswith("ctunet"):
        return [y, t1, t2, y, y]
    if name.startswith("cunet"):
        return [y, t1, t2]
    return [y, y]


def make_model_mode(name, mode):
    """One more pass of the REFERENCE modules over the two golden samples in `mode`:
         'bf16'  under torch.autocast('cpu', dtype=torch.bfloat16) exactly where the trainer wraps model(data)
                 (trainer_CTUNet.py:90-91); the loss is evaluated on the up-cast logits (DiceCELoss runs in fp32 under
                 autocast) -> model_<name>_bf16.npz: sampled logits at the fp32 golden's indices, loss, gradient norms
                 and sampled gradient entries;
         'f32' / 'f64'  plain / .double() -> model_<name>_vox32.npz / _vox64.npz.
       Every mode also stores, per sample and output, the complete 14-logit vectors of N_VOX fixed voxels (argmax
       agreement) and the Dice and CE terms of that output against its deep-supervision target (float64 evaluation of
       the stored-precision logits): 'Dice within 1e-4' of the north star is checked on these."""
    t0 = time.time()
    ctor, loss_name = MODELS[name]
    z32 = np.load(os.path.join(HERE, f"model_{name}.npz"), allow_pickle=False)
    model = ctor()
    model.load_state_dict({k: O.synthetic_tensor(k, v.shape) for k, v in model.state_dict().items()}, strict=True)
    if mode == "f64":
        model = model.double()
    out, losses = {}, []
    for s in range(2):
        x, y = O.synthetic_batch(1, seed=1000 + s)
        if mode == "bf16":
            with torch.autocast("cpu", dtype=torch.bfloat16):
                outs = flat_outputs(model(x))
            assert all(o.dtype == torch.bfloat16 for o in outs)
            loss = O.LOSSES[loss_name](model_outputs_regroup(name, outs), y)   # casts the logits up itself
        elif mode == "f64":
            outs = flat_outputs(model(x.double()))
            loss = _loss64(loss_name, model_outputs_regroup(name, outs), y)
        else:
            outs = flat_outputs(model(x))
            loss = O.LOSSES[loss_name](model_outputs_regroup(name, outs), y)
        (0.5 * loss).backward()
        losses.append(loss.item())
        for i, (o, t) in enumerate(zip(outs, _targets_for(name, y))):
            idx = torch.from_numpy(z32[f"s{s}/out{i}/idx"])
            of = o.detach()
            if mode == "bf16":
                out[f"s{s}/out{i}/val_bf16"] = of.flatten()[idx].float().numpy()
            g = torch.Generator().manual_seed(777 + i)
            nvox = of[0, 0].numel()
            vidx = torch.randint(0, nvox, (N_VOX,), generator=g)
            out[f"s{s}/out{i}/vox_idx"] = vidx.numpy()
            vox = of[0].reshape(of.shape[1], -1)[:, vidx].t()
            out[f"s{s}/out{i}/vox"] = (vox.double() if mode == "f64" else vox.float()).numpy()
            d, c = _dice_ce_parts64(of, t)
            out[f"s{s}/out{i}/dice_ce"] = np.array([d, c])
        del outs, loss
    out["loss_per_sample"] = np.array(losses)
    out["loss_b2"] = np.array(sum(losses) / 2)
    if mode == "bf16":
        pr = dict(model.named_parameters())
        out["grad/norm_b2"] = np.array([0.0 if pr[str(k)].grad is None else pr[str(k)].grad.double().norm().item()
                                        for k in z32["grad/keys"]])
        for j in range(8):
            k = str(z32[f"grad/sample{j}/key"])
            out[f"grad/sample{j}/val"] = pr[k].grad.flatten()[torch.from_numpy(z32[f"grad/sample{j}/idx"])].float().numpy()
    fn = {"bf16": f"model_{name}_bf16.npz", "f32": f"model_{name}_vox32.npz", "f64": f"model_{name}_vox64.npz"}[mode]
    np.savez_compressed(os.path.join(HERE, fn), **out)
    print(f"{fn}: losses {losses}, {time.time() - t0:.0f}s", flush=True)


def _loss64(loss_name, outs, y):
    """oracle loss composition evaluated in float64 (O.dice_ce_loss casts logits to float32, so restate the cast-free
    form here)."""
    import torch.nn.functional as F

    def dice_ce(logits, target):
        n_cls = logits.shape[1]
        labels = target.squeeze(1).long()
        p = torch.softmax(logits, dim=1)
        yy = F.one_hot(labels, n_cls).permute(0, 4, 1, 2, 3).to(p.dtype)
        inter = (p * yy).sum((2, 3, 4))
        denom = (yy * yy).sum((2, 3, 4)) + (p * p).sum((2, 3, 4))
        return (1.0 - 2.0 * inter / (denom + 1e-6)).mean() + F.cross_entropy(logits, labels)

    t1 = O.downsample_target(y, (0.5, 0.5, 1.0))
    t2 = O.downsample_target(y, (0.25, 0.25, 0.5))
    if loss_name == "ctunet":
        (a, b, c), (d, e) = outs
        return dice_ce(a, y) + 0.5 * (dice_ce(b, t1) + 0.5 * dice_ce(c, t2)) + 0.5 * (dice_ce(d, y) + dice_ce(e, y))
    if loss_name == "cunet":
        a, b, c = outs
        return dice_ce(a, y) + 0.5 * (dice_ce(b, t1) + 0.5 * dice_ce(c, t2))
    a, b = outs
    return dice_ce(a, y) + dice_ce(b, y)


def model_outputs_regroup(name, outs):
    if name.startswith("ctunet"):
        return ((outs[0], outs[1], outs[2]), (outs[3], outs[4]))
    return tuple(outs)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-models", action="store_true")
    ap.add_argument("--only", default=None)
    ap.add_argument("--f64", action="store_true", help="write model_<name>_f64.npz (reference run in float64)")
    ap.add_argument("--mode", default=None, choices=["bf16", "f32", "f64"],
                    help="write model_<name>_bf16.npz (reference under CPU autocast) / _vox32.npz / _vox64.npz")
    a = ap.parse_args()
    if a.mode:
        for n in MODELS:
            if a.only is None or a.only == n:
                make_model_mode(n, a.mode)
        sys.exit(0)
    if a.f64:
        for n in MODELS:
            if a.only is None or a.only == n:
                make_model_f64(n)
        sys.exit(0)
    if a.only is None:
        make_blocks()
        make_loss()
    if not a.skip_models:
        for n in MODELS:
            if a.only is None or a.only == n:
                make_model(n)
