"""CPU tests of SURVEY.md 8f rows 1-3: the inference oracle's closed forms and size-independent properties (the MONAI
helpers the reference calls are absent here, so this row is "parity unpinned" at that boundary - see oracle/infer_oracle.py),
the product's host-side window geometry against the oracle, and checkpoint interop."""
import math

import numpy as np
import pytest
import torch

import hybrid_ctunet_amd as H
from hybrid_ctunet_amd import checkpoint as ck
from hybrid_ctunet_amd import inference as inf
from oracle import infer_oracle as O


# ---- oracle: closed forms ------------------------------------------------------------------------------------------
def test_scan_interval_follows_reference_rule():
    # trainer_CTUNet.py:560-581: roi when the image is one window wide, else int(roi*(1-overlap)), at least 1
    assert O.get_scan_interval((96, 96, 144), (96, 96, 96), 0.5) == (96, 96, 48)
    assert O.get_scan_interval((100, 100, 100), (96, 96, 96), 0.7) == (28, 28, 28)
    assert O.get_scan_interval((10, 10), (2, 2), 0.99) == (1, 1)


@pytest.mark.parametrize("n,sigma_scale", [(96, 0.125), (32, 0.125), (9, 0.25), (7, 0.125)])
def test_gaussian_importance_is_the_separable_closed_form(n, sigma_scale):
    m = O.compute_importance_map((n, n + 1, n + 2), "gaussian", sigma_scale)
    assert m.max().item() == pytest.approx(1.0)
    assert m.min().item() > 0
    c = (n // 2, (n + 1) // 2, (n + 2) // 2)
    assert m[c].item() == pytest.approx(1.0)
    # separable: m[i,j,k] * m[c] == m[i,cj,ck] * m[ci,j,ck] * m[ci,cj,k] away from the replaced zeros
    a, b, d = m[:, c[1], c[2]], m[c[0], :, c[2]], m[c[0], c[1], :]
    outer = a[:, None, None] * b[None, :, None] * d[None, None, :]
    big = outer > 1e-20
    assert torch.allclose(m[big], outer[big], rtol=2e-5, atol=0)
    # 1-D profile: unit-cell integral of N(0, sigma) normalised by the centre cell
    sigma = n * sigma_scale
    x = np.arange(n) - n // 2
    g = np.array([0.5 * (math.erf((xi + 0.5) / (sigma * math.sqrt(2))) - math.erf((xi - 0.5) / (sigma * math.sqrt(2)))) for xi in x])
    tail = int(max(sigma * 4.0, 0.5) + 0.5)
    g[np.abs(x) > tail] = 0
    ref = g / g[n // 2]
    keep = ref > 1e-12
    assert np.allclose(a.numpy()[keep], ref[keep], rtol=2e-3, atol=1e-7)   # fp32 erf differences cancel badly in the tails


@pytest.mark.parametrize("img,roi,ov", [((96, 96, 144), (96, 96, 96), 0.5), ((100, 50, 70), (32, 32, 32), 0.7),
                                        ((33, 33, 33), (32, 32, 32), 0.5), ((40, 41, 17), (16, 8, 17), 0.0)])
def test_windows_cover_the_volume_and_stay_inside(img, roi, ov):
    sl = O.dense_patch_slices(img, roi, O.get_scan_interval(img, roi, ov))
    cover = np.zeros(img, dtype=np.int32)
    for s in sl:
        assert all(x.start >= 0 and x.stop <= n and x.stop - x.start == r for x, n, r in zip(s, img, roi))
        cover[s] += 1
    assert cover.min() >= 1
    assert len(set(tuple(x.start for x in s) for s in sl)) == len(sl)


@pytest.mark.parametrize("mode", ["constant", "gaussian"])
@pytest.mark.parametrize("outputs", ["multi", "single"])
def test_oracle_blend_is_a_partition_of_unity(mode, outputs):
    """A predictor that is a pointwise function of the window reproduces that function of the whole volume whatever the
    overlap / blend - and padding is cropped away again (image smaller than the roi in one dim)."""
    torch.manual_seed(0)
    x = torch.randn(2, 1, 21, 12, 30)

    def predictor(w):
        a = torch.cat([w * 2 + 1, -w, w * w], 1)
        b = torch.cat([w, w + 3, 0.5 * w], 1)
        return ((a, None), (b, None)) if outputs == "multi" else (a, None)

    res = O.sliding_window_inference(x, (16, 16, 16), 3, predictor, overlap=0.5, mode=mode, outputs=outputs)
    exp_a = torch.cat([x * 2 + 1, -x, x * x], 1)
    if outputs == "multi":
        assert res[0].shape == exp_a.shape
        assert torch.allclose(res[0], exp_a, atol=1e-5)
        assert torch.allclose(res[1], torch.cat([x, x + 3, 0.5 * x], 1), atol=1e-5)
    else:
        assert torch.allclose(res, exp_a, atol=1e-5)


def test_oracle_hybrid_and_dice():
    torch.manual_seed(1)
    p1, p2 = torch.randn(14, 5, 6, 7) * 3, torch.randn(14, 5, 6, 7) * 3
    l1, l2, lh = O.hybrid_complement(p1, p2)
    assert torch.equal(l1, p1.argmax(0)) and torch.equal(l2, p2.argmax(0))
    agree = l1 == l2
    assert torch.equal(lh[agree], l1[agree])   # where both models agree the hybrid agrees
    y = (np.arange(24).reshape(2, 3, 4) % 3 == 0).astype(np.float64)
    assert O.dice(y, y) == 1.0 and O.dice(1 - y, y) == 0.0 and O.dice(y, np.zeros_like(y)) == 0.0


# ---- product host logic against the oracle ------------------------------------------------------------------------
@pytest.mark.parametrize("img,roi,ov", [((96, 96, 144), (96, 96, 96), 0.5), ((226, 180, 97), (96, 96, 96), 0.7),
                                        ((33, 33, 33), (32, 32, 32), 0.5), ((40, 41, 17), (16, 8, 17), 0.0),
                                        ((10, 10, 10), (2, 3, 4), 0.99)])
def test_window_starts_equal_oracle(img, roi, ov):
    mine = inf.window_starts(img, roi, ov)
    ref = [tuple(x.start for x in s) for s in O.dense_patch_slices(img, roi, O.get_scan_interval(img, roi, ov))]
    assert mine == ref


@pytest.mark.parametrize("roi,ss", [((96, 96, 96), 0.125), ((32, 48, 20), 0.125), ((7, 8, 9), 0.3), ((5, 64, 6), (0.1, 0.2, 0.5))])
def test_importance_map_closed_form_equals_oracle_filtering(roi, ss):
    a = inf.importance_map(roi, "gaussian", ss)
    b = O.compute_importance_map(roi, "gaussian", ss)
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-30)
    assert torch.equal(inf.importance_map(roi, "constant"), O.compute_importance_map(roi, "constant"))
    with pytest.raises(ValueError):
        inf.importance_map(roi, "linear")


def test_sliding_window_refuses_host_tensors():
    with pytest.raises(RuntimeError):
        H.sliding_window_inference(torch.zeros(1, 1, 8, 8, 8), (4, 4, 4), 1, lambda w: (w,), outputs="single")
    with pytest.raises(RuntimeError):
        H.hybrid_complement(torch.zeros(3, 2, 2, 2), torch.zeros(3, 2, 2, 2))


def test_dice_per_organ_equals_oracle():
    rng = np.random.default_rng(0)
    p, l = rng.integers(0, 14, (8, 9, 10)), rng.integers(0, 12, (8, 9, 10))
    mine = H.dice_per_organ(torch.from_numpy(p), torch.from_numpy(l), 14)
    ref = [O.dice((p == c).astype(np.float64), (l == c).astype(np.float64)) for c in range(1, 14)]
    assert mine == pytest.approx(ref)
    assert mine[12] == 0.0          # class 13 absent from the label


# ---- checkpoint interop ---------------------------------------------------------------------------------------------
class _Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(3, 4)
        self.b = torch.nn.Conv3d(1, 2, 3)


class _Wrap(torch.nn.Module):
    def __init__(self, m):
        super().__init__()
        self.module = m


def test_checkpoint_round_trip_reference_format(tmp_path):
    torch.manual_seed(0)
    src, dst = _Tiny(), _Tiny()
    f = ck.save_checkpoint(_Wrap(src), 17, str(tmp_path / "model.pt"), best_acc=0.83)
    raw = torch.load(f, weights_only=True)     # the reference's reader: torch.load + ["state_dict"]
    assert set(raw) == {"epoch", "best_acc", "state_dict"}
    assert list(raw["state_dict"]) == list(src.state_dict())      # unwrapped keys, reference order
    epoch, best = ck.load_checkpoint(dst, f)
    assert (epoch, best) == (17, pytest.approx(0.83))
    for (k, v), (_, w) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(v, w), k


def test_checkpoint_prefix_stripping_and_bare_state_dict(tmp_path):
    torch.manual_seed(1)
    src, dst, dst2 = _Tiny(), _Tiny(), _Tiny()
    # main_CTUNet.py:166-178: keys may carry "backbone."; a DDP-saved dict carries "module."; unknown keys are ignored
    sd = {"module.backbone." + k if i % 2 else "backbone." + k: v for i, (k, v) in enumerate(src.state_dict().items())}
    sd["swin_vit.extra"] = torch.zeros(1)
    assert ck.load_checkpoint(_Wrap(dst), {"state_dict": sd, "epoch": 3}) == (3, 0.0)
    assert all(torch.equal(v, dst.state_dict()[k]) for k, v in src.state_dict().items())
    with pytest.raises(RuntimeError):
        ck.load_checkpoint(dst, {"state_dict": sd}, strict=True)
    # --resume_ckpt form: a bare state dict on disk (main_CTUNet.py:145-148)
    torch.save(src.state_dict(), tmp_path / "bare.pt")
    assert ck.load_checkpoint(dst2, tmp_path / "bare.pt", strict=True) == (0, 0.0)
    assert all(torch.equal(v, dst2.state_dict()[k]) for k, v in src.state_dict().items())


def test_checkpoint_load_executes_nothing(tmp_path):
    """Files are read with weights_only=True: a pickle that would run code is refused, not executed."""
    import pickle

    class Boom:
        def __reduce__(self):
            return (pytest.fail, ("checkpoint loader executed pickled code",))

    with open(tmp_path / "evil.pt", "wb") as fh:
        pickle.dump({"state_dict": Boom()}, fh)
    with pytest.raises(Exception):
        ck.load_checkpoint(_Tiny(), str(tmp_path / "evil.pt"))
