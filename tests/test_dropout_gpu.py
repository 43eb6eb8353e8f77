"""GPU parity of the dropout row.  Which elements fall is a property of the generator, so parity is
established in two steps: (1) the device masks equal oracle/dropout_oracle.py's numpy Philox bit for bit (itself pinned on
the Random123 known-answer vectors); (2) GIVEN those masks, outputs and gradients of the HIP blocks equal the reference's
formulae (oracle blocks with mask-driven stand-ins at the reference's nn.Dropout positions).  fp32 parity mode: 5e-4;
bf16: 3e-2 (the op-level gates of test_ops_gpu.py)."""
import numpy as np
import pytest
import torch

import hybrid_ctunet_amd as H
from hybrid_ctunet_amd import ops
from hybrid_ctunet_amd.networks import hybrid_CTUNet as N
from hybrid_ctunet_amd.networks import vit as V
from oracle import ctunet_oracle as O
from oracle import dropout_oracle as D

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(got, ref):
    return (got.detach().float().cpu() - ref.detach().float()).abs().max().item() / max(ref.abs().max().item(), 1e-12)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n", [8, 1003, 65536 + 13])
@pytest.mark.parametrize("p", [0.2, 0.5])
def test_flat_dropout_mask_and_values_are_bit_exact(dtype, n, p):
    torch.manual_seed(0)
    x = torch.randn(n).to(dtype)
    r = torch.randn(n).to(dtype)
    ops.manual_seed(0x1234567899, 41)
    xd = x.to(DEV).requires_grad_(True)
    rd = r.to(DEV).requires_grad_(True)
    y = ops.dropout(xd, p)
    y2 = ops.dropout(xd, p, residual=rd)
    assert ops.dropout_state() == (0x1234567899, 43)
    k1 = torch.from_numpy(D.flat_keep(n, p, 0x1234567899, 41))
    k2 = torch.from_numpy(D.flat_keep(n, p, 0x1234567899, 42))
    s = D.scale(p)
    assert torch.equal(y.detach().cpu(), (x.float() * s * k1).to(dtype))
    assert torch.equal(y2.detach().cpu(), (x.float() * s * k2 + r.float()).to(dtype))
    g = torch.randn(n).to(dtype)
    y2.backward(g.to(DEV))
    assert torch.equal(xd.grad.cpu(), (g.float() * s * k2).to(dtype))        # backward regenerates the same mask
    assert torch.equal(rd.grad.cpu(), g)


@pytest.mark.parametrize("pairs,ntok", [(5, 216), (3, 50), (2, 432)])
def test_attention_mask_equals_oracle(pairs, ntok):
    got = ops.attention_dropout_mask(pairs, ntok, 0.2, 77, 9, DEV).cpu().numpy().astype(bool)
    assert np.array_equal(got, D.attn_keep(pairs, ntok, 0.2, 77, 9))


def _masked_attention(qkv, heads, scale, keep, drop_scale, bias=None):
    """softmax(q k^T scale (+ bias)) -> dropout -> . v on [G, n, 3 * heads * dh] in float64-free plain torch."""
    G, n, _ = qkv.shape
    q, k, v = (t.reshape(G, n, heads, -1).transpose(1, 2) for t in qkv.chunk(3, -1))
    sim = (q @ k.transpose(-1, -2)) * scale
    if bias is not None:
        sim = sim + bias
    attn = torch.softmax(sim, -1) * keep.reshape(G, heads, n, n).to(qkv.dtype) * drop_scale
    return (attn @ v).transpose(1, 2).reshape(G, n, -1)


@pytest.mark.parametrize("dtype,n,heads,dh,tol", [(torch.float32, 96, 2, 64, 5e-4), (torch.float32, 216, 3, 32, 5e-4),
                                                 (torch.bfloat16, 432, 12, 64, 3e-2), (torch.bfloat16, 200, 2, 32, 3e-2)])
def test_attention_core_with_dropout_equals_masked_math(dtype, n, heads, dh, tol):
    torch.manual_seed(1)
    G, p = 2, 0.2
    qkv = (torch.randn(G, n, 3 * heads * dh) * 0.7).to(dtype)
    go = torch.randn(G, n, heads * dh).to(dtype)
    ops.manual_seed(99, 3)
    qd = qkv.to(DEV).requires_grad_(True)
    out = ops.attention(qd, heads, dh ** -0.5, dropout_p=p)
    out.backward(go.to(DEV))
    keep = torch.from_numpy(D.attn_keep(G * heads, n, p, 99, 3))
    qr = qkv.float().requires_grad_(True)
    ref = _masked_attention(qr, heads, dh ** -0.5, keep, D.scale(p))
    ref.backward(go.float())
    assert _rel(out, ref) <= tol
    assert _rel(qd.grad, qr.grad) <= tol
    # and it is not the dropout-free result
    plain = ops.attention(qd.detach(), heads, dh ** -0.5)
    assert _rel(plain, ref) > 0.05


def test_attention_dropout_needs_the_mfma_kernels():
    qkv = torch.randn(1, 432, 3 * 64, device=DEV)            # fp32, 432 tokens: VALU kernels only
    with pytest.raises(RuntimeError, match="MFMA"):
        ops.attention(qkv, 1, 0.125, dropout_p=0.2)


def _load(prod, orac):
    prod.load_state_dict(orac.state_dict(), strict=True)
    return prod.to(DEV).train()


def _grads_close(prod, orac, tol):
    pg = dict(prod.named_parameters())
    for k, v in orac.named_parameters():
        assert _rel(pg[k].grad, v.grad) <= tol, k


def test_vit_transformer_block_with_dropout_equals_masked_reference():
    """networks/vit.py:80-96 with dropout 0.2 at its four sites (attention probabilities, to_out, FF twice)."""
    torch.manual_seed(2)
    orac = O.TransformerBlock(128, 2, 64, 256)
    prod = _load(V.TransformerBlock(128, 2, 64, 256, dropout=0.2), orac)
    x = torch.randn(2, 96, 128)
    g = torch.randn(2, 96, 128)
    ops.manual_seed(5, 100)
    xd = x.to(DEV).requires_grad_(True)
    y = prod(xd)
    y.backward(g.to(DEV))
    assert ops.dropout_state() == (5, 104)                   # four dropout calls, in the reference's module order
    D.install(orac, D.Provider(0.2, 5, 100))
    xr = x.clone().requires_grad_(True)
    yr = orac(xr)
    yr.backward(g)
    assert _rel(y, yr) <= 5e-4 and _rel(xd.grad, xr.grad) <= 5e-4
    _grads_close(prod, orac, 1e-3)
    prod.eval()                                              # eval: nn.Dropout is the identity
    D.install(orac, D.Provider(0.0, 0))
    assert _rel(prod(xd), orac(xr)) <= 5e-4
    assert ops.dropout_state() == (5, 104)                   # ... and consumes no offsets


@pytest.mark.parametrize("part,mode", [(1, "block"), (2, "grid")])
def test_window_attention_and_ff_with_dropout_equal_masked_reference(part, mode):
    """hybrid_CTUNet.py:442-526 inside Residual, block and grid partitions: attention-probability dropout under the
    relative-position bias, to_out dropout and both FeedForward dropouts, masks laid out on the un-partitioned volume."""
    torch.manual_seed(3)
    C, w = 64, 6
    o_att, o_ff = O.Residual(O.MultiAxisAttention(C, 32, w)), O.Residual(O.FeedForward(C, 4 * C))
    p_att = _load(N.Residual(N.MultiAxisAttention(C, 32, dropout=0.2, window_size=w)), o_att)
    p_ff = _load(N.Residual(N.FeedForward(C, dropout=0.2)), o_ff)
    x = torch.randn(2, C, 12, 6, 18)                         # 2 x 1 x 3 windows per item
    g = torch.randn(2, C, 12, 6, 18)
    ops.manual_seed(8, 0)
    xd = x.permute(0, 2, 3, 4, 1).contiguous().to(DEV).requires_grad_(True)
    y = p_ff(p_att(xd, part=part))
    y.backward(g.permute(0, 2, 3, 4, 1).contiguous().to(DEV))
    assert ops.dropout_state() == (8, 4)
    prov = D.Provider(0.2, 8, 0)
    D.install(o_att, prov, mode)
    D.install(o_ff, prov, mode)
    xr = x.clone().requires_grad_(True)
    yr = O._unpartition(o_ff(o_att(O._partition(xr, w, mode))), mode)
    yr.backward(g)
    assert _rel(y.permute(0, 4, 1, 2, 3), yr) <= 5e-4
    assert _rel(xd.grad.permute(0, 4, 1, 2, 3), xr.grad) <= 5e-4
    _grads_close(p_att, o_att, 1e-3)
    _grads_close(p_ff, o_ff, 1e-3)


def test_tunet_trains_with_dropout_and_is_reproducible():
    """The authors' dr0.2 recipe end to end (bf16): finite loss and gradients; the same (seed, offset) replays the same
    masks, another offset does not; eval mode is the dropout-free function."""
    torch.manual_seed(4)
    net = H.build_model("tunet", dropout_rate=0.2).to(DEV).set_precision("bf16").train()
    x = torch.rand(1, 1, 96, 96, 96, device=DEV)
    y = torch.randint(0, 14, (1, 1, 96, 96, 96), device=DEV).float()

    def run(offset):
        ops.manual_seed(21, offset)
        net.zero_grad()
        out = net(x)
        loss = H.tunet_loss(out, y)
        loss.backward()
        return out[0].detach().float(), loss.item(), ops.dropout_state()[1] - offset

    a, la, used = run(0)
    b, lb, _ = run(0)
    c, lc, _ = run(1000)
    assert used == 1 + 12 * 4 + 3 * 8 + 4                    # emb + 12 ViT blocks x 4 + 3 window stages x 8 + FF-only stage
    assert np.isfinite(la) and all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    def l2(u, v):   # relative L2: bf16 run-to-run noise is ~1e-2, the max norm over 10^7 logits wanders more
        return ((u - v).norm() / v.norm()).item()

    assert l2(b, a) <= 8e-2                                  # same masks (bf16 run-to-run noise only)
    assert l2(c, a) >= 0.15                                  # other masks
    net.eval()
    with torch.no_grad():
        e1 = net(x)[0].float()
        state = ops.dropout_state()
        e2 = net(x)[0].float()
    assert ops.dropout_state() == state and l2(e2, e1) <= 8e-2
    ref = H.build_model("tunet").to(DEV).set_precision("bf16").eval()
    ref.load_state_dict(net.state_dict())
    with torch.no_grad():
        assert l2(ref(x)[0].float(), e1) <= 8e-2


def test_dropout_refuses_graph_capture():
    """The Philox (seed, offset) key is a launch argument taken from host state: captured into a HIP graph, every replay would
    reuse one mask without any error.  A dropout call with p > 0 on a capturing stream raises instead."""
    x = torch.randn(64, 128, device=DEV).to(torch.bfloat16)
    ops.dropout(x, 0.2)   # (warm: workspaces, seed)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    raised = False
    with torch.cuda.stream(s):
        g.capture_begin()
        try:
            ops.dropout(x, 0.2)
        except RuntimeError as e:
            raised = "cannot be captured" in str(e)
        finally:
            g.capture_end()
    torch.cuda.current_stream().wait_stream(s)
    assert raised
