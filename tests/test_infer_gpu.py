"""GPU parity of SURVEY.md 8f rows 1-3: the on-device sliding-window blend and the hybrid complementation against
oracle/infer_oracle.py (same seeded inputs, same predictor arithmetic), size-independent properties at the reference's
96^3 window, one real CTUNet pass, and checkpoint round trips through the device model."""
import pytest
import torch

import hybrid_ctunet_amd as H
from oracle import infer_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel_l2(a, b):
    """|a - b|_2 / |b|_2: bf16 run-to-run noise (reordered atomics feed rounding flips) is ~1e-2 here, while the max norm over
    10^7 logits wanders up to ~5 %; wrong weights / windows / masks give O(1) in either norm."""
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


def _pointwise(outputs, channels_last=False, dtype=torch.float32):
    """A predictor that is the same fp32 arithmetic on host and device: per-voxel functions of the window."""
    def predictor(w):
        w = w.float()
        a = torch.cat([w * 2 + 1, -w, w * w, w.abs()], 1)
        b = torch.cat([w, w + 3, 0.5 * w, w - 1], 1)
        if w.is_cuda:
            a, b = a.to(dtype), b.to(dtype)
            if channels_last:   # the models hand back channels-last storage viewed as NCDHW
                a = a.permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3)
                b = b.permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3)
        return ((a, None), (b, None)) if outputs == "multi" else (a, None)
    return predictor


@pytest.mark.parametrize("outputs", ["multi", "single"])
@pytest.mark.parametrize("mode", ["constant", "gaussian"])
@pytest.mark.parametrize("shape,roi,ov,swb", [((2, 1, 21, 12, 30), (16, 16, 16), 0.5, 3),     # padded in one dim, ragged
                                              ((1, 2, 40, 33, 17), (16, 8, 17), 0.25, 4),   # one window wide in w
                                              ((1, 1, 8, 8, 8), (8, 8, 8), 0.7, 1)])        # single window
def test_sliding_window_equals_oracle(outputs, mode, shape, roi, ov, swb):
    torch.manual_seed(0)
    x = torch.randn(shape)
    if shape[1] != 1:
        def wrap(p):
            return lambda w: p(w[:, :1] + w[:, 1:])
    else:
        def wrap(p):
            return p
    ref = O.sliding_window_inference(x, roi, swb, wrap(_pointwise(outputs)), overlap=ov, mode=mode, outputs=outputs)
    got = H.sliding_window_inference(x.to(DEV), roi, swb, wrap(_pointwise(outputs, channels_last=True)), overlap=ov,
                                     mode=mode, outputs=outputs)
    ref = ref if outputs == "multi" else (ref,)
    got = got if outputs == "multi" else (got,)
    assert len(ref) == len(got)
    for r, g in zip(ref, got):
        assert g.dtype == torch.float32 and g.shape == r.shape
        assert torch.allclose(g.cpu(), r, rtol=1e-5, atol=1e-5)


def test_sliding_window_neighbourhood_predictor_equals_oracle():
    """A predictor that is NOT pointwise (a fixed 3x3x3 box filter with zero padding inside the window): results depend
    on window placement and blend weights, so this pins window order, batching and the gaussian map, not just coverage."""
    torch.manual_seed(3)
    x = torch.randn(1, 1, 37, 30, 26)
    k = torch.ones(3, 1, 3, 3, 3) * torch.tensor([1.0, -0.5, 0.25]).view(3, 1, 1, 1, 1) / 27

    def pred(w):
        y = torch.nn.functional.conv3d(w.float(), k.to(w.device), padding=1)
        return ((y, None), (y * y, None))

    ref = O.sliding_window_inference(x, (16, 16, 16), 4, pred, overlap=0.7, mode="gaussian")
    got = H.sliding_window_inference(x.to(DEV), (16, 16, 16), 4, pred, overlap=0.7, mode="gaussian")
    for r, g in zip(ref, got):
        assert torch.allclose(g.cpu(), r, rtol=1e-4, atol=1e-5)


def test_sliding_window_bf16_logits_and_kwargs():
    torch.manual_seed(4)
    x = torch.randn(1, 1, 20, 20, 20)
    seen = {}

    def pred(w, scale, flag=False):
        seen["args"] = (scale, flag)
        return ((w * scale).to(torch.bfloat16), None)

    got = H.sliding_window_inference(x.to(DEV), 16, 2, pred, 0.5, "gaussian", 0.125, "constant", 0.0, None, None, 2.0,
                                     outputs="single", flag=True)
    assert seen["args"] == (2.0, True)
    ref = O.sliding_window_inference(x, 16, 2, lambda w: pred(w, 2.0), overlap=0.5, mode="gaussian", outputs="single")
    assert torch.allclose(got.cpu(), ref, rtol=1e-5, atol=1e-5)


def test_sliding_window_full_size_partition_of_unity():
    """The reference's evaluation geometry (roi 96^3, overlap 0.7 as test_CTUNet_final.py, gaussian) on a ragged case-sized
    volume: a pointwise predictor must be reproduced exactly up to fp32 rounding - the size-independent property."""
    torch.manual_seed(5)
    x = torch.randn(1, 1, 150, 131, 97, device=DEV)
    r1, r2 = H.sliding_window_inference(x, (96, 96, 96), 4, _pointwise("multi", channels_last=True, dtype=torch.bfloat16),
                                        overlap=0.7, mode="gaussian")
    xf = x.float()
    e1 = torch.cat([xf * 2 + 1, -xf, xf * xf, xf.abs()], 1).to(torch.bfloat16).float()
    e2 = torch.cat([xf, xf + 3, 0.5 * xf, xf - 1], 1).to(torch.bfloat16).float()
    assert r1.shape == e1.shape
    assert torch.allclose(r1, e1, rtol=1e-5, atol=1e-5) and torch.allclose(r2, e2, rtol=1e-5, atol=1e-5)


def test_sliding_window_argument_errors():
    x = torch.zeros(1, 1, 8, 8, 8, device=DEV)
    with pytest.raises(AssertionError):
        H.sliding_window_inference(x, 4, 1, _pointwise("single"), overlap=1.0, outputs="single")
    with pytest.raises(ValueError):
        H.sliding_window_inference(x, 4, 1, _pointwise("single"), mode="linear", outputs="single")
    with pytest.raises(ValueError):
        H.sliding_window_inference(x[0], 4, 1, _pointwise("single"), outputs="single")
    with pytest.raises(ValueError):   # predictor returning the wrong spatial size
        H.sliding_window_inference(x, 4, 1, lambda w: (w[..., :2],), outputs="single")
    with pytest.raises(ValueError):
        H.sliding_window_inference(x, 4, 1, _pointwise("single"), sw_device="cpu", outputs="single")


@pytest.mark.parametrize("C,shape", [(14, (19, 23, 31)), (2, (5, 5, 5)), (32, (3, 4, 130))])
def test_hybrid_complement_equals_oracle(C, shape):
    torch.manual_seed(6)
    p1, p2 = torch.randn(C, *shape) * 4, torch.randn(C, *shape) * 4
    p2[:, 0] = p1[:, 0]                       # a slab where both models agree exactly
    p1[:, 1, 0] = 0.0                         # ties: all classes equal -> first index
    r = O.hybrid_complement(p1, p2)
    g = H.hybrid_complement(p1.to(DEV), p2.to(DEV))
    for name, a, b in zip(("model1", "model2", "hybrid"), g, r):
        assert a.dtype == torch.int64
        a = a.cpu()
        if name != "hybrid":
            assert torch.equal(a, b), name
        else:
            # the averaged softmax can tie to 1 ulp between expf implementations: accept a different label only where
            # the oracle's two best averaged probabilities are within 1e-6
            s = (torch.softmax(p1, 0) + torch.softmax(p2, 0)) / 2
            diff = a != b
            if diff.any():
                top2 = s.topk(2, 0).values
                assert ((top2[0] - top2[1])[diff] < 1e-6).all()
            assert diff.float().mean() < 1e-3
    with pytest.raises(ValueError):
        H.hybrid_complement(p1.to(DEV), p2[:, :2].to(DEV))


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-4), ("bf16", 8e-2)])
def test_ctunet_through_sliding_window_and_checkpoint(tmp_path, precision, tol):
    """One real model: CTUNet -> save_checkpoint -> fresh CTUNet + load_checkpoint -> the same sliding-window outputs
    (fp32 mode: to rounding of reordered atomics; bf16 mode: to bf16 noise - a model with other weights is off by O(1));
    both heads finite, shaped [1, 14, D, H, W]; and labels/dice computed from them."""
    torch.manual_seed(7)
    net = H.build_model("ctunet", model_depth=50).to(DEV).set_precision(precision).eval()
    x = torch.rand(1, 1, 96, 96, 120, device=DEV)
    with torch.no_grad():
        a1, a2 = H.sliding_window_inference(x, (96, 96, 96), 1, net, overlap=0.5, mode="gaussian")
    assert a1.shape == (1, 14, 96, 96, 120) and a2.shape == a1.shape
    assert torch.isfinite(a1).all() and torch.isfinite(a2).all()
    f = H.save_checkpoint(net, 5, str(tmp_path / "m.pt"), best_acc=0.5)
    torch.manual_seed(8)
    net2 = H.build_model("ctunet", model_depth=50).to(DEV).set_precision(precision).eval()
    with torch.no_grad():
        c1, _ = H.sliding_window_inference(x, (96, 96, 96), 1, net2, overlap=0.5, mode="gaussian")
    assert _rel_l2(c1, a1) > 0.3                               # other weights: a different function
    assert H.load_checkpoint(net2, f, strict=True) == (5, 0.5)
    with torch.no_grad():
        b1, b2 = H.sliding_window_inference(x, (96, 96, 96), 1, net2, overlap=0.5, mode="gaussian")
    for a, b_ in ((a1, b1), (a2, b2)):
        assert _rel_l2(b_, a) <= tol
    l1, l2, lh = H.hybrid_complement(a1[0], a2[0])
    for lab, logit in ((l1, a1[0]), (l2, a2[0])):
        # the reference takes argmax of the SOFTMAX: logits closer than an fp32 ulp of exp() tie there and the first class
        # wins, so a label may differ from the logits' argmax only across such a gap
        diff = lab != logit.argmax(0)
        top2 = logit.topk(2, 0).values
        assert ((top2[0] - top2[1])[diff] < 1e-6).all() and diff.float().mean() < 1e-2
    d = H.dice_per_organ(lh, l1, 14)
    assert len(d) == 13 and all(0.0 <= v <= 1.0 for v in d)


def test_single_output_models_through_sliding_window():
    """trainer_CUNet.py / trainer_TUNet.py form: predictor(...)[0] of CUNet and TUNet."""
    x = torch.rand(1, 1, 96, 100, 96, device=DEV)
    for kind in ("cunet", "tunet"):
        torch.manual_seed(11)
        net = H.build_model(kind, model_depth=50).to(DEV).set_precision("bf16").eval()
        with torch.no_grad():
            out = H.sliding_window_inference(x, (96, 96, 96), 1, net, overlap=0.5, mode="gaussian", outputs="single")
            one = net(x[:, :, :, :96].contiguous())[0].float()
        assert out.shape == (1, 14, 96, 100, 96) and torch.isfinite(out).all()
        # rows 0..3 of dim H are covered by the first window only: the blend returns that window's logits unchanged
        # (bf16 run-to-run noise from reordered atomics; a wrong window or weight map would be off by O(1))
        assert _rel_l2(out[:, :, :, :4], one[:, :, :, :4]) <= 8e-2


def test_optimizer_state_round_trip(tmp_path):
    torch.manual_seed(9)
    net = H.build_model("cunet", model_depth=50).to(DEV).set_precision("bf16")
    opt = H.FusedAdamW(net.parameters(), lr=1e-3, weight_decay=1e-5)
    x = torch.rand(1, 1, 96, 96, 96, device=DEV)
    y = torch.randint(0, 14, (1, 1, 96, 96, 96), device=DEV).float()
    for _ in range(2):
        opt.zero_grad()
        H.cunet_loss(net(x), y).backward()
        opt.step()
    f = H.save_checkpoint(net, 2, str(tmp_path / "o.pt"), optimizer=opt)
    net2 = H.build_model("cunet", model_depth=50).to(DEV).set_precision("bf16")
    opt2 = H.FusedAdamW(net2.parameters(), lr=5e-2, weight_decay=0.0)
    assert H.load_checkpoint(net2, f, strict=True, optimizer=opt2)[0] == 2
    assert opt2.step_count == opt.step_count and torch.equal(opt2.m, opt.m) and torch.equal(opt2.v, opt.v)
    assert opt2.param_groups[0]["lr"] == pytest.approx(1e-3) and opt2.weight_decay == pytest.approx(1e-5)
    for p, q in zip(net.parameters(), net2.parameters()):
        assert torch.equal(p, q)
    # the next step lands on the same parameters (atomics in the weight gradients reorder fp32 sums: tolerance, not bits)
    for n_, o_ in ((net, opt), (net2, opt2)):
        o_.zero_grad()
        H.cunet_loss(n_(x), y).backward()
        o_.step()
    for p, q in zip(net.parameters(), net2.parameters()):
        assert torch.allclose(p, q, rtol=0, atol=5e-3) and (p - q).abs().mean() < 1e-4
    with pytest.raises(ValueError):
        H.FusedAdamW(H.build_model("cunet", model_depth=101).to(DEV).parameters()).load_state_dict(opt.state_dict())


def test_optimizer_state_interop_with_torch_adamw(tmp_path):
    """a checkpoint the reference wrote holds torch.optim.AdamW's state dict (trainer_CTUNet.py:311-312);
    FusedAdamW must resume from it, and torch.optim.AdamW must resume from what save_checkpoint writes for FusedAdamW."""
    torch.manual_seed(3)
    net = H.build_model("cunet", model_depth=50).to(DEV).set_precision("bf16")
    x = torch.rand(1, 1, 96, 96, 96, device=DEV)
    y = torch.randint(0, 14, (1, 1, 96, 96, 96), device=DEV).float()
    ref_opt = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=1e-5)
    for _ in range(2):
        for p in net.parameters():
            p.grad = None
        H.cunet_loss(net(x), y).backward()
        ref_opt.step()
    ck = {"epoch": 2, "best_acc": 0.5, "state_dict": {k: v.detach().cpu() for k, v in net.state_dict().items()},
          "optimizer": ref_opt.state_dict()}                      # what trainer_CTUNet.py:308-317 writes
    torch.save(ck, str(tmp_path / "ref.pt"))
    net2 = H.build_model("cunet", model_depth=50).to(DEV).set_precision("bf16")
    fused = H.FusedAdamW(net2.parameters(), lr=1.0, weight_decay=0.5)
    assert H.load_checkpoint(net2, str(tmp_path / "ref.pt"), strict=True, optimizer=fused) == (2, 0.5)
    assert fused.step_count == 2 and fused.lr == pytest.approx(1e-3) and fused.weight_decay == pytest.approx(1e-5)
    where = {id(p): (o, p.numel()) for p, o in zip(fused.flat.params, fused.flat.offsets)}
    for (k, p2), p1 in zip(net2.named_parameters(), net.parameters()):
        st = ref_opt.state.get(p1)
        o, n = where[id(p2)]
        if st:
            assert torch.equal(fused.m[o:o + n].view(p2.shape), st["exp_avg"]), k
            assert torch.equal(fused.v[o:o + n].view(p2.shape), st["exp_avg_sq"]), k
        else:
            assert float(fused.m[o:o + n].abs().max()) == 0.0, k
    # and back: torch.optim.AdamW reads the file save_checkpoint writes for the fused optimizer
    f = H.save_checkpoint(net2, 3, str(tmp_path / "fused.pt"), optimizer=fused)
    back = torch.load(f, map_location="cpu", weights_only=True)
    opt3 = torch.optim.AdamW(net.parameters(), lr=0.3)
    opt3.load_state_dict(back["optimizer"])
    assert opt3.param_groups[0]["lr"] == pytest.approx(1e-3)
    for p1 in net.parameters():
        st = ref_opt.state.get(p1)
        if st:
            assert torch.equal(opt3.state[p1]["exp_avg"].cpu(), st["exp_avg"].cpu())
            assert float(opt3.state[p1]["step"]) == 2.0
