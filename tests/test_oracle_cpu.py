"""CPU suite, part 1: the oracle restatement against the golden vectors the REFERENCE produced
(tests/golden/make_golden.py), and the scipy index map.  No GPU, no /root/reference."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ctunet_oracle as O



@pytest.fixture(autouse=True, scope="module")
def _cpu_threads():
    """Use the cores this process may actually run on (at most 16) for the oracle - and only while THESE tests run: a
    module-level torch.set_num_threads(os.cpu_count()) also hit the GPU tests collected in the same session, where the
    box reports every host core but the job owns 16 (the whole GPU suite took 15 minutes of thrashing instead of 1)."""
    before = torch.get_num_threads()
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 8
    torch.set_num_threads(max(1, min(16, n)))
    yield
    torch.set_num_threads(before)


def _npz(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _load_block(module, name):
    module.load_state_dict({k: O.synthetic_tensor(f"{name}.{k}", v.shape) for k, v in module.state_dict().items()})


class _Win(torch.nn.Module):
    def __init__(self, dim, mode):
        super().__init__()
        self.mode = mode
        self.attn = O.Residual(O.MultiAxisAttention(dim, 32, 6))
        self.ff = O.Residual(O.FeedForward(dim, dim * 4))

    def forward(self, x):
        return O._unpartition(self.ff(self.attn(O._partition(x, 6, self.mode))), self.mode)


# (golden name, oracle module factory, key remap golden->oracle or None)
def _blocks():
    return {
        "resblock_same": lambda: O.ResBlock(16, 16, 3, 1),
        "resblock_proj": lambda: O.ResBlock(32, 16, 3, 1),
        "resblock_in1": lambda: O.ResBlock(1, 64, 3, 1),
        "bottleneck_s2": lambda: O.Bottleneck(32, 16, (2, 2, 2), O._Downsample(32, 64, (2, 2, 2))),
        "bottleneck_id": lambda: O.Bottleneck(64, 16),
        "stem": lambda: O.ConvLayer(1, 64, (7, 7, 7), (2, 2, 1)),
        "convt222": lambda: O.ConvLayer(32, 16, (2, 2, 2), (2, 2, 2), is_transposed=True),
        "convt221": lambda: O.ConvLayer(32, 16, (2, 2, 1), (2, 2, 1), is_transposed=True),
        "upcat": lambda: O.UpCatConvBlock(32, 16, 3, (2, 2, 2)),
        "pwa": lambda: O.PixelweightAttention(64),
        "fusion": lambda: O.Up2FusionBlock(64, 32, 3, (2, 2, 2)),
        "pixelshuffle222": lambda: O.PixelShuffle((2, 2, 2), 64, 24),
        "pixelshuffle221": lambda: O.PixelShuffle((2, 2, 1), 32, 16),
        "feedforward": lambda: O.FeedForward(32, 128),
        "vit_block": lambda: O.TransformerBlock(64, 2, 32, 128),
        "vit_small": lambda: O.ViT((32, 32), 16, 16, 8, 64, 2, 2, 128, dim_head=32),
    }


@pytest.mark.parametrize("name", sorted(_blocks()))
def test_oracle_block_matches_reference_golden(golden_dir, name):
    z = _npz(golden_dir, "blocks.npz")
    m = _blocks()[name]()
    _load_block(m, name)
    ins = []
    i = 0
    while f"{name}/in{i}" in z:
        ins.append(torch.from_numpy(z[f"{name}/in{i}"]).requires_grad_(True))
        i += 1
    y = m(*ins)
    ref = torch.from_numpy(z[f"{name}/out"])
    assert y.shape == ref.shape
    assert (y - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    y.backward(torch.from_numpy(z[f"{name}/gout"]))
    for j, t in enumerate(ins):
        g = torch.from_numpy(z[f"{name}/gin{j}"])
        assert (t.grad - g).abs().max().item() <= 5e-5 * max(1.0, g.abs().max().item())
    for k, p in m.named_parameters():
        g = torch.from_numpy(z[f"{name}/gw/{k}"])
        isnone = bool(z[f"{name}/gw_isnone/{k}"])
        assert (p.grad is None) == isnone, k
        if not isnone:
            assert (p.grad - g).abs().max().item() <= 1e-4 * max(1.0, g.abs().max().item()), k


@pytest.mark.parametrize("mode", ["block", "grid"])
def test_oracle_window_attention_matches_reference_golden(golden_dir, mode):
    z = _npz(golden_dir, "blocks.npz")
    name = f"win_{mode}"
    m = _Win(64, mode)
    # golden keys: seq.1.fn.* (attention), seq.2.fn.* (ff)
    sd = {}
    for k, v in m.state_dict().items():
        gk = k.replace("attn.", "seq.1.").replace("ff.", "seq.2.")
        sd[k] = O.synthetic_tensor(f"{name}.{gk}", v.shape)
    m.load_state_dict(sd)
    x = torch.from_numpy(z[f"{name}/in0"]).requires_grad_(True)
    y = m(x)
    ref = torch.from_numpy(z[f"{name}/out"])
    assert (y - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    y.backward(torch.from_numpy(z[f"{name}/gout"]))
    g = torch.from_numpy(z[f"{name}/gin0"])
    assert (x.grad - g).abs().max().item() <= 5e-5 * max(1.0, g.abs().max().item())
    gb = torch.from_numpy(z[f"{name}/gw/seq.1.fn.rel_pos_bias.weight"])
    assert (m.attn.fn.rel_pos_bias.weight.grad - gb).abs().max().item() <= 1e-4 * max(1.0, gb.abs().max().item())


def test_zoom_index_map_matches_scipy_golden(golden_dir):
    z = _npz(golden_dir, "loss.npz")
    for key in z.files:
        if key.startswith("zoom/"):
            _, n_in, zf = key.split("/")
            n_in, zf = int(n_in), float(zf)
            got = O.zoom_nearest_index(n_in, int(round(n_in * zf)))
            # golden = scipy.ndimage.zoom(arange(n_in)): the value IS the source index; -1 (out of range) reads cval 0
            assert np.array_equal(np.where(got < 0, 0, got), z[key]), key
            if n_in == 96:
                assert (got >= 0).all()  # the trainer's cases have no out-of-range slot
    # the 3-D composition used by the trainer (trainer_CTUNet.py:93-94)
    g = torch.Generator().manual_seed(5)
    t = torch.randint(0, 14, (2, 1, 96, 96, 96), generator=g).float()
    t1 = O.downsample_target(t, (0.5, 0.5, 1.0))
    t2 = O.downsample_target(t, (0.25, 0.25, 0.5))
    assert list(t1.shape) == list(z["zoom3d/shapes"][0]) and list(t2.shape) == list(z["zoom3d/shapes"][1])
    assert np.array_equal(t1.flatten()[torch.from_numpy(z["zoom3d/t_seed5_half_idx"])].numpy(), z["zoom3d/t_seed5_half_val"])
    assert np.array_equal(t2.flatten()[torch.from_numpy(z["zoom3d/t_seed5_quarter_idx"])].numpy(),
                          z["zoom3d/t_seed5_quarter_val"])


def test_dicece_restatement_fixture(golden_dir):
    z = _npz(golden_dir, "loss.npz")
    lg = torch.from_numpy(z["dicece/logits"]).requires_grad_(True)
    loss, dice, ce = O.dice_ce_loss(lg, torch.from_numpy(z["dicece/target"]), return_parts=True)
    assert np.allclose([loss.item(), dice.item(), ce.item()], z["dicece/loss_dice_ce"], rtol=1e-5)
    loss.backward()
    assert np.allclose(lg.grad.numpy(), z["dicece/grad"], rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("name,model,depth", [("cunet50", "cunet", 50), ("tunet", "tunet", 101)])
def test_oracle_whole_model_forward_matches_reference_golden(golden_dir, name, model, depth):
    """Forward-only on CPU (keeps the suite in minutes); gradients are covered block-wise above and whole-model on
    the GPU against the same goldens."""
    z = _npz(golden_dir, f"model_{name}.npz")
    man = json.load(open(os.path.join(golden_dir, f"manifest_{name}.json")))
    m = O.build(model, model_depth=depth)
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == man
    m.load_state_dict(O.synthetic_state_dict(m))
    x, y = O.synthetic_batch(1, seed=1000)
    with torch.no_grad():
        outs = m(x)
        loss = O.LOSSES[model](outs, y)
    for i, o in enumerate(outs):
        ref = z[f"s0/out{i}/val"]
        got = o.flatten()[torch.from_numpy(z[f"s0/out{i}/idx"])].numpy()
        assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max(), i
    assert abs(loss.item() - float(z["loss_per_sample"][0])) <= 1e-5 * abs(loss.item())
