"""GPU parity, op level: every HIP kernel family (called through the C ABI via hybrid_ctunet_amd.ops) against a plain
PyTorch float64 CPU computation of the same op on the same seeded inputs.  Tolerances: fp32 mode 2e-4 of the
reference's max magnitude (exact-f32 MFMA, different summation order); bf16 mode 3e-2 (8-bit mantissa operands)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 2e-4, torch.bfloat16: 3e-2}


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import hybrid_ctunet_amd  # noqa: F401
    from hybrid_ctunet_amd import ops as o
    return o


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return ((torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * scale)


def dev(t, dtype, grad=False):
    """CPU float64 -> device tensor of `dtype`; returns (device tensor, the float64 value it actually holds)."""
    d = t.to(dtype).cuda().contiguous()
    held = d.detach().cpu().double()
    if grad:
        d.requires_grad_(True)
    return d, held


def close(got, ref, dtype, what="", scale=None):
    got = got.detach().float().cpu().double()
    ref = ref.detach().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    s = ref.abs().max().item() if scale is None else scale
    err = (got - ref).abs().max().item()
    assert math.isfinite(err) and err <= TOL[dtype] * max(s, 1e-6), f"{what}: max|d|={err:.3e} ref max={s:.3e} ({dtype})"


def cl(t):  # NCDHW -> NDHWC
    return t.permute(0, 2, 3, 4, 1).contiguous()


def cf(t):  # NDHWC -> NCDHW
    return t.permute(0, 4, 1, 2, 3).contiguous()


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,K,N,bias,act,res", [(300, 96, 72, True, 0, False), (257, 32, 136, True, 1, True),
                                                (1000, 768, 64, False, 0, True), (130, 16, 16, True, 0, False),
                                                (64, 3072, 768, True, 1, False), (300, 160, 200, True, 0, False),
                                                (2000, 256, 384, True, 0, True), (300, 192, 200, True, 1, True),
                                                (1500, 128, 520, False, 0, False), (700, 32, 128, True, 0, True),
                                                (260, 96, 64, False, 0, False)])
@pytest.mark.parametrize("gemm", ["dma", "generic"])
def test_linear(ops, dtype, M, K, N, bias, act, res, gemm):
    """gemm="dma": plain bf16 GEMMs with K % 64 == 0 run on the LDS-DMA kernels (gemm_dma.hip); "generic" pins them to
    the implicit-GEMM kernels (igemm.hip) through the ctu_set_option test hook."""
    from hybrid_ctunet_amd import _lib
    if gemm == "generic" and (dtype == torch.float32 or K % 32 or N % 64):
        pytest.skip("this case runs on the generic kernels anyway")
    _lib.call("ctu_set_option", b"generic_gemm", 1 if gemm == "generic" else 0)
    ops.USE_W_KN = gemm != "generic"
    try:
        _linear_case(ops, dtype, M, K, N, bias, act, res)
    finally:
        _lib.call("ctu_set_option", b"generic_gemm", 0)
        ops.USE_W_KN = True


@pytest.mark.parametrize("M,K,N,res", [(32768, 128, 256, False), (32768 + 128 * 37, 64, 128, True), (40960, 32, 384, False),
                                       (32768, 128, 128, True)])
def test_linear_stream_kernel(ops, M, K, N, res):
    """The short-K long-M layers (M >= 32768, K in {32, 64, 128}, N % 128 == 0) run on gemm_nt_stream: forward (plain and
    with a residual in the epilogue) and, through the data gradient of a Linear whose forward weight is read reduction-major,
    its w_kn form (here: K_bwd = N_fwd = 128 for the last case; the other cases' data gradients have K_bwd = N_fwd >= 256
    and stay on the general kernel).  Checked against float64 math like every other GEMM case, and against the general
    kernel (ctu_set_option "route" bit 1) for equal rounding."""
    from hybrid_ctunet_amd import _lib
    dtype = torch.bfloat16
    _linear_case(ops, dtype, M, K, N, False, 0, res)
    x = rnd((M, K), 1).to(dtype).cuda()
    w = rnd((N, K), 2, 1 / math.sqrt(K)).float().cuda()
    r = rnd((M, N), 4).to(dtype).cuda() if res else None
    with torch.no_grad():
        a = ops.linear(x, w, None, r, 0)
        _lib.call("ctu_set_option", b"route", 1)
        try:
            b = ops.linear(x, w, None, r, 0)
        finally:
            _lib.call("ctu_set_option", b"route", 0)
    assert torch.equal(a, b)      # same products, same fp32 accumulation order over k, same single rounding to bf16


def _linear_case(ops, dtype, M, K, N, bias, act, res):
    x, xh = dev(rnd((M, K), 1), dtype, True)
    w, wh = dev(rnd((N, K), 2, 1 / math.sqrt(K)), torch.float32, True)
    b, bh = dev(rnd((N,), 3), torch.float32, True) if bias else (None, None)
    r, rh = dev(rnd((M, N), 4), dtype, True) if res else (None, None)
    gy, gyh = dev(rnd((M, N), 5), dtype)
    y = ops.linear(x, w, b, r, act)
    wq = wh if dtype == torch.float32 else wh.to(dtype).double()
    xr, wr = xh.clone().requires_grad_(True), wq.clone().requires_grad_(True)
    ref = xr @ wr.t()
    br = rr = None
    if bias:
        br = bh.clone().requires_grad_(True)
        ref = ref + br
    if act:
        ref = F.gelu(ref)
    if res:
        rr = rh.clone().requires_grad_(True)
        ref = ref + rr
    close(y, ref, dtype, "y")
    y.backward(gy)
    ref.backward(gyh)
    close(x.grad, xr.grad, dtype, "gx")
    close(w.grad, wr.grad, dtype, "gw")
    if bias:
        close(b.grad, br.grad, dtype, "gb")
    if res:
        close(r.grad, rr.grad, dtype, "gres")


CONV_CASES = [  # B, D,H,W, C1, C2, N, k, s, p
    (2, 5, 6, 7, 32, 0, 64, 3, 1, 1),
    (1, 6, 8, 10, 32, 32, 48, 3, 1, 1),
    (1, 8, 8, 6, 64, 0, 32, 3, 2, 1),
    (1, 8, 6, 8, 64, 0, 136, 1, 2, 0),
    (2, 4, 5, 6, 64, 64, 32, 1, 1, 0),
    (1, 9, 10, 11, 16, 0, 24, 3, 1, 1),
    (1, 8, 8, 8, 32, 0, 32, 3, (2, 2, 1), 1),
    (1, 4, 8, 8, 64, 0, 128, 3, 1, 1),    # halo kernel: NT=4, two channel chunks, exact brick
    (2, 7, 9, 17, 32, 32, 64, 3, 1, 1),   # halo kernel: concat forward + split data gradient, ragged bricks
    (1, 12, 12, 24, 96, 0, 32, 3, 1, 1),  # halo kernel: NT=1, three chunks
    (1, 5, 16, 8, 64, 64, 160, 3, 1, 1),  # halo kernel: N=160 -> 5 n tiles (NT=1 path), concat
    (1, 8, 8, 16, 256, 0, 256, 3, 1, 1),  # halo kernel: 8 n tiles -> two n blocks, 16 half chunks, several bricks/workgroup
    (2, 12, 12, 24, 128, 0, 128, 3, 1, 1),  # halo kernel: batch-pair bricks (12 rows: 3 x 4 for two items), channel split + finish pass
    (2, 5, 12, 9, 64, 64, 128, 3, 1, 1),    # batch-pair bricks, ragged d and w, concat forward + split data gradient
    (4, 4, 4, 8, 32, 0, 256, 3, 1, 1),      # batch-pair bricks, two pairs, two n blocks of four tiles
]


@pytest.mark.parametrize("halo", [True, False])
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3d(ops, dtype, case, halo):
    B, D, H, W, C1, C2, N, k, s, p = case
    if not halo and not (k == 3 and s == 1 and C1 % 32 == 0 and C2 % 32 == 0):
        pytest.skip("generic kernel already exercised by the halo=True run of this case")
    ops.USE_HALO_CONV = halo
    x1, x1h = dev(cl(rnd((B, C1, D, H, W), 1)), dtype, True)
    x2 = x2h = None
    if C2:
        x2, x2h = dev(cl(rnd((B, C2, D, H, W), 2)), dtype, True)
    w, wh = dev(rnd((N, C1 + C2, k, k, k), 3, 1 / math.sqrt((C1 + C2) * k ** 3)), torch.float32, True)
    y = ops.conv3d(x1, w, s, p, x2)
    wq = wh if dtype == torch.float32 else wh.to(dtype).double()
    xr1 = cf(x1h).requires_grad_(True)
    xr2 = cf(x2h).requires_grad_(True) if C2 else None
    wr = wq.clone().requires_grad_(True)
    xin = torch.cat((xr1, xr2), 1) if C2 else xr1
    ref = F.conv3d(xin, wr, stride=s, padding=p)
    close(y, cl(ref), dtype, "y")
    gy, gyh = dev(cl(rnd(tuple(ref.shape), 4)), dtype)
    y.backward(gy)
    ops.USE_HALO_CONV = True
    ref.backward(cf(gyh))
    close(x1.grad, cl(xr1.grad), dtype, "gx1")
    if C2:
        close(x2.grad, cl(xr2.grad), dtype, "gx2")
    close(w.grad, wr.grad, dtype, "gw")


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("case", [(2, 7, 9, 17, 32, 32, 64), (1, 8, 8, 16, 64, 0, 160), (2, 4, 8, 8, 32, 0, 32),
                                  (2, 12, 12, 24, 128, 0, 128), (2, 5, 12, 9, 64, 64, 128)])   # (the last two: batch-pair bricks, sums per item)
def test_conv_instance_norm_fused_statistics(ops, case, fused):
    """bf16 conv3x3x3 -> InstanceNorm -> LeakyReLU: the statistics come from the conv epilogue (fp32 accumulators summed per
    brick, ctu_conv3_halo in_acc + ctu_in_finalize) or, with fused=False, from the separate pass over the bf16 output.
    Both must match the float64 reference; ragged bricks and a partial last n tile included."""
    B, D, H, W, C1, C2, N = case
    dtype = torch.bfloat16
    ops.FUSE_IN_STATS = fused
    try:
        x1, x1h = dev(cl(rnd((B, C1, D, H, W), 1)), dtype, True)
        x2, x2h = dev(cl(rnd((B, C2, D, H, W), 2)), dtype, True) if C2 else (None, None)
        w, wh = dev(rnd((N, C1 + C2, 3, 3, 3), 3, 1 / math.sqrt((C1 + C2) * 27)), torch.float32, True)
        conv = ops.conv3d(x1, w, 1, 1, x2)
        assert (getattr(conv, "_ctu_in_acc", None) is not None) == fused
        y = ops.instance_norm(conv, None, True)
        xin = torch.cat((cf(x1h), cf(x2h)), 1) if C2 else cf(x1h)
        xin.requires_grad_(True)
        wr = wh.to(dtype).double().requires_grad_(True)
        ref = F.leaky_relu(F.instance_norm(F.conv3d(xin, wr, padding=1), eps=1e-5), 0.01)
        close(y, cl(ref), dtype, "y")
        gy, gyh = dev(cl(rnd(tuple(ref.shape), 4)), dtype)
        y.backward(gy)
        ref.backward(cf(gyh))
        close(w.grad, wr.grad, dtype, "gw")
        close(x1.grad, cl(xin.grad[:, :C1]), dtype, "gx1")
    finally:
        ops.FUSE_IN_STATS = True


@pytest.mark.parametrize("case", [(2, 4, 8, 8, 64, 256), (2, 8, 8, 8, 128, 72), (1, 2, 8, 8, 192, 64),
                                  (2, 16, 32, 32, 128, 256), (3, 16, 16, 48, 32, 128), (2, 24, 32, 32, 64, 384)])
def test_conv1x1_instance_norm_fused_statistics(ops, case):
    """bf16 1x1x1 conv (plain GEMM) -> InstanceNorm: statistics summed in the LDS-DMA GEMM epilogue (ctu_epilogue.in_acc)
    vs the float64 reference; rows per batch item are multiples of 128, N with a partial last tile included.  The last three
    cases (M >= 32768, K in {128, 32, 64}, N % 128 == 0) run on gemm_nt_stream: per-lane partial sums across a workgroup's
    tile range, batch-item switches inside a range (3 items over 288 tiles), DPP reduction at the flush."""
    B, D, H, W, K, N = case
    dtype = torch.bfloat16
    x, xh = dev(cl(rnd((B, K, D, H, W), 1)) + 0.25, dtype, True)
    w, wh = dev(rnd((N, K, 1, 1, 1), 2, 1 / math.sqrt(K)), torch.float32, True)
    conv = ops.linear(x, w, in_stats=True)
    assert getattr(conv, "_ctu_in_acc", None) is not None
    y = ops.instance_norm(conv, None, False)  # no LeakyReLU: its mask flips on near-zero values would dominate gx
    xr = cf(xh).requires_grad_(True)
    wr = wh.to(dtype).double().requires_grad_(True)
    ref = F.instance_norm(F.conv3d(xr, wr), eps=1e-5)
    close(y, cl(ref), dtype, "y")
    gy, gyh = dev(cl(rnd(tuple(ref.shape), 3)), dtype)
    y.backward(gy)
    ref.backward(cf(gyh))
    close(x.grad, cl(xr.grad), dtype, "gx")
    close(w.grad, wr.grad, dtype, "gw")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("k", [(2, 2, 2), (2, 2, 1)])
def test_conv_transpose(ops, dtype, k):
    B, D, H, W, Ci, Co = 2, 3, 4, 5, 64, 32
    x, xh = dev(cl(rnd((B, Ci, D, H, W), 1)), dtype, True)
    w, wh = dev(rnd((Ci, Co, *k), 2, 1 / math.sqrt(Ci)), torch.float32, True)
    y = ops.conv_transpose3d(x, w)
    wq = wh if dtype == torch.float32 else wh.to(dtype).double()
    xr, wr = cf(xh).requires_grad_(True), wq.clone().requires_grad_(True)
    ref = F.conv_transpose3d(xr, wr, stride=k)
    close(y, cl(ref), dtype, "y")
    gy, gyh = dev(cl(rnd(tuple(ref.shape), 3)), dtype)
    y.backward(gy)
    ref.backward(cf(gyh))
    close(x.grad, cl(xr.grad), dtype, "gx")
    close(w.grad, wr.grad, dtype, "gw")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("k,s,p,dims", [((3, 3, 3), (1, 1, 1), (1, 1, 1), (9, 10, 12)), ((7, 7, 7), (2, 2, 1), (3, 3, 3), (12, 14, 10)),
                                        ((1, 1, 1), (1, 1, 1), (0, 0, 0), (5, 6, 7))])
@pytest.mark.parametrize("as_gemm", [True, False])
def test_conv_cin1(ops, dtype, k, s, p, dims, as_gemm):
    """as_gemm: bf16 multi-tap Cin=1 convs run as patch matrix (ctu_im2col_cin1) + LDS-DMA GEMMs; False pins the direct
    VALU kernels (the only path in fp32 / for 1x1x1)."""
    if as_gemm and (dtype == torch.float32 or k == (1, 1, 1)):
        pytest.skip("direct kernel in this configuration")
    ops.CIN1_AS_GEMM = as_gemm
    try:
        _conv_cin1_case(ops, dtype, k, s, p, dims)
    finally:
        ops.CIN1_AS_GEMM = True


def _conv_cin1_case(ops, dtype, k, s, p, dims):
    B, N = 2, 64
    x, xh = dev(rnd((B, *dims, 1), 1), dtype)
    w, wh = dev(rnd((N, 1, *k), 2, 1 / math.sqrt(k[0] * k[1] * k[2])), torch.float32, True)
    y = ops.conv3d_cin1(x, w, s, p)
    xr, wr = cf(xh), wh.clone().requires_grad_(True)
    ref = F.conv3d(xr, wr, stride=s, padding=p)
    close(y, cl(ref), dtype, "y")
    gy, gyh = dev(cl(rnd(tuple(ref.shape), 3)), dtype)
    y.backward(gy)
    ref.backward(cf(gyh))
    close(w.grad, wr.grad, dtype, "gw")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("C,act,res", [(16, True, False), (64, True, True), (32, False, False), (96, True, True), (1024, False, True)])
def test_instance_norm(ops, dtype, C, act, res):
    B, D, H, W = 2, 5, 6, 7
    x, xh = dev(cl(rnd((B, C, D, H, W), 1, 2.0) + 0.3), dtype, True)
    r, rh = dev(cl(rnd((B, C, D, H, W), 2)), dtype, True) if res else (None, None)
    y = ops.instance_norm(x, r, act)
    xr = cf(xh).requires_grad_(True)
    ref = F.instance_norm(xr, eps=1e-5)
    rr = None
    if res:
        rr = cf(rh).requires_grad_(True)
        ref = ref + rr
    if act:
        ref = F.leaky_relu(ref, 0.01)
    close(y, cl(ref), dtype, "y")
    gy, gyh = dev(cl(rnd((B, C, D, H, W), 3)), dtype)
    y.backward(gy)
    ref.backward(cf(gyh))
    close(x.grad, cl(xr.grad), dtype, "gx")
    if res:
        close(r.grad, cl(rr.grad), dtype, "gres")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("rows,dim", [(50, 32), (100, 64), (333, 128), (77, 768), (33, 2048), (40, 512)])
def test_layer_norm(ops, dtype, rows, dim):
    x, xh = dev(rnd((rows, dim), 1, 2.0) + 0.2, dtype, True)
    g, gh = dev(1 + 0.1 * rnd((dim,), 2), torch.float32, True)
    b, bh = dev(0.1 * rnd((dim,), 3), torch.float32, True)
    y = ops.layer_norm(x, g, b)
    xr, gr, br = xh.clone().requires_grad_(True), gh.clone().requires_grad_(True), bh.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (dim,), gr, br, 1e-5)
    close(y, ref, dtype, "y")
    gy, gyh = dev(rnd((rows, dim), 4), dtype)
    y.backward(gy)
    ref.backward(gyh)
    close(x.grad, xr.grad, dtype, "gx")
    close(g.grad, gr.grad, dtype, "ggamma")
    close(b.grad, br.grad, dtype, "gbeta")


def _attn_ref(qkv, heads, scale, bias=None):
    # qkv: [G, n, 3*dim] float64
    G, n, d3 = qkv.shape
    dim = d3 // 3
    q, k, v = (t.reshape(G, n, heads, dim // heads).transpose(1, 2) for t in qkv.chunk(3, -1))
    sim = (q @ k.transpose(-1, -2)) * scale
    if bias is not None:
        sim = sim + bias
    return (torch.softmax(sim, -1) @ v).transpose(1, 2).reshape(G, n, dim)


def _set_attn_impl(valu):
    from hybrid_ctunet_amd import _lib
    _lib.call("ctu_set_option", b"attn_valu", 1 if valu else 0)


@pytest.mark.parametrize("valu", [False, True])
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,n,heads,dh", [(2, 50, 2, 32), (1, 432, 3, 64), (2, 130, 2, 64), (1, 216, 2, 32), (1, 300, 1, 32)])
def test_attention_global(ops, dtype, B, n, heads, dh, valu):
    _set_attn_impl(valu)
    dim = heads * dh
    qkv, qh = dev(rnd((B, n, 3 * dim), 1), dtype, True)
    scale = dh ** -0.5
    y = ops.attention(qkv, heads, scale)
    qr = qh.clone().requires_grad_(True)
    ref = _attn_ref(qr, heads, scale)
    close(y, ref, dtype, "y")
    gy, gyh = dev(rnd((B, n, dim), 2), dtype)
    y.backward(gy)
    _set_attn_impl(False)
    ref.backward(gyh)
    close(qkv.grad, qr.grad, dtype, "gqkv")


@pytest.mark.parametrize("valu", [False, True])
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("part", [1, 2])
def test_attention_window(ops, dtype, part, valu):
    from oracle import ctunet_oracle as O
    _set_attn_impl(valu)
    B, D, H, W, heads, dh, win = 1, 6, 12, 12, 2, 32, 6
    dim = heads * dh
    qkv, qh = dev(rnd((B, D, H, W, 3 * dim), 1), dtype, True)
    table, th = dev(rnd(((2 * win - 1) ** 3, heads), 2, 0.5), torch.float32, True)
    scale = dh ** -0.5
    y = ops.attention(qkv, heads, scale, table, part, win)
    mode = "block" if part == 1 else "grid"
    qr, tr = qh.clone().requires_grad_(True), th.clone().requires_grad_(True)
    xp = O._partition(qr.permute(0, 4, 1, 2, 3), win, mode)  # b X Y Z w1 w2 w3 c
    b, X, Y, Z = xp.shape[:4]
    bias = tr[O.rel_pos_indices(win)].permute(2, 0, 1)
    o = _attn_ref(xp.reshape(b * X * Y * Z, win ** 3, 3 * dim), heads, scale, bias)
    ref = O._unpartition(o.reshape(b, X, Y, Z, win, win, win, dim), mode).permute(0, 2, 3, 4, 1)
    close(y, ref, dtype, "y")
    gy, gyh = dev(rnd((B, D, H, W, dim), 3), dtype)
    y.backward(gy)
    _set_attn_impl(False)
    ref.backward(gyh)
    close(qkv.grad, qr.grad, dtype, "gqkv")
    close(table.grad, tr.grad, dtype, "gbias")


@pytest.mark.parametrize("dtype", DT)
def test_pwa(ops, dtype):
    rows, C = 203, 64
    scale = 32 ** -0.5
    a, ah = dev(rnd((rows, 3 * C), 1), dtype, True)
    b, bh = dev(rnd((rows, 3 * C), 2), dtype, True)
    y = ops.pwa(a, b, scale)
    ar, br = ah.clone().requires_grad_(True), bh.clone().requires_grad_(True)
    q1, k1, v1 = (t.reshape(rows, C // 32, 32) for t in ar.chunk(3, -1))
    q2, k2, v2 = (t.reshape(rows, C // 32, 32) for t in br.chunk(3, -1))
    d1 = (q2 * k1).sum(-1, keepdim=True) * scale
    d2 = (q1 * k2).sum(-1, keepdim=True) * scale
    att = torch.softmax(torch.cat((d1, d2), -1), -1)
    ref = (att[..., 0:1] * v1 + att[..., 1:2] * v2).reshape(rows, C)
    close(y, ref, dtype, "y")
    gy, gyh = dev(rnd((rows, C), 3), dtype)
    y.backward(gy)
    ref.backward(gyh)
    close(a.grad, ar.grad, dtype, "g1")
    close(b.grad, br.grad, dtype, "g2")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("f,c", [((2, 2, 2), 8), ((2, 2, 1), 16), ((2, 2, 2), 32), ((2, 2, 2), 96)])
def test_pixel_shuffle_and_patchify(ops, dtype, f, c):
    from oracle import ctunet_oracle as O
    B, D, H, W = 2, 3, 4, 5
    Cb = c * f[0] * f[1] * f[2]
    x, xh = dev(cl(rnd((B, Cb, D, H, W), 1)), dtype, True)
    y = ops.pixel_shuffle(x, f)
    xr = cf(xh).requires_grad_(True)
    ps = O.PixelShuffle(f, Cb, c)
    xx = xr.reshape(B, c, *f, D, H, W).permute(0, 5, 2, 6, 3, 7, 4, 1).reshape(B, D * f[0], H * f[1], W * f[2], c)
    close(y, xx, dtype, "y")
    gy, gyh = dev(rnd(tuple(xx.shape), 2), dtype)
    y.backward(gy)
    xx.backward(gyh)
    close(x.grad, cl(xr.grad), dtype, "gx")
    img, ih = dev(rnd((2, 32, 32, 16), 3), dtype)
    tok = ops.patchify(img, 16, 16, 8)
    vit = O.ViT((32, 32), 16, 16, 8, 64, 1, 2, 128, dim_head=32)
    close(tok, vit.patchify(ih[:, None]), dtype, "patchify")


@pytest.mark.parametrize("dtype", DT)
def test_add_bcast_and_cast(ops, dtype):
    x, xh = dev(rnd((3, 20, 64), 1), dtype, True)
    p, ph = dev(rnd((1, 20, 64), 2), torch.float32, True)
    y = ops.add_bcast(x, p)
    close(y, xh + ph, dtype, "y")
    gy, gyh = dev(rnd((3, 20, 64), 3), dtype)
    y.backward(gy)
    close(x.grad, gyh, dtype, "gx")
    close(p.grad, gyh.sum(0, keepdim=True), dtype, "gpos")
    f = torch.rand(1000, device="cuda")
    assert torch.equal(ops.cast(f, torch.bfloat16), f.to(torch.bfloat16))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape,tshape,weight", [((2, 14, 6, 8, 10), (6, 8, 10), 1.0), ((1, 14, 12, 12, 24), (24, 24, 24), 0.5),
                                                 ((2, 14, 6, 6, 12), (24, 24, 24), 0.25)])
def test_dice_ce(ops, dtype, shape, tshape, weight):
    from oracle import ctunet_oracle as O
    from hybrid_ctunet_amd.train import dice_ce_loss
    B, n_cls = shape[0], shape[1]
    lg64 = rnd(shape, 1, 3.0)
    # the head layout: channels-last, padded to 16 columns
    buf = torch.zeros((B, *shape[2:], 16), dtype=dtype, device="cuda")
    buf[..., :n_cls] = cl(lg64).to(dtype).cuda()
    buf.requires_grad_(True)
    view = buf[..., :n_cls].permute(0, 4, 1, 2, 3)
    g = torch.Generator().manual_seed(7)
    tgt = torch.randint(0, n_cls, (B, 1, *tshape), generator=g).float()
    loss = dice_ce_loss(view, tgt.cuda(), weight)
    held = cf(buf.detach()[..., :n_cls].cpu().double()).requires_grad_(True)
    zoom = tuple(o / i for o, i in zip(shape[2:], tshape))
    ref = weight * O.dice_ce_loss(held, O.downsample_target(tgt, zoom))
    assert abs(loss.item() - ref.item()) <= 1e-4 * abs(ref.item()), (loss.item(), ref.item())
    (loss * 1.5).backward()
    (ref * 1.5).backward()
    close(buf.grad[..., :n_cls], cl(held.grad), dtype, "dlogits")
    assert buf.grad[..., n_cls:].abs().max().item() == 0.0
    # non-padded input path (plain contiguous NCDHW logits)
    plain = cf(buf.detach()[..., :n_cls]).contiguous().requires_grad_(True)
    loss2 = dice_ce_loss(plain, tgt.cuda(), weight)
    assert abs(loss2.item() - ref.item()) <= 1e-4 * abs(ref.item())
    loss2.backward()
    close(plain.grad, held.grad / 1.5, dtype, "dlogits-plain")


def test_direct_gradient_sink_matches_autograd_path(ops):
    """With train.FlatParams the Linear/LayerNorm weight-gradient kernels accumulate straight into the flat buffer
    (ops.register_grad_sink); the result must equal the ordinary autograd path, accumulate over repeated backwards,
    and mark the parameters as touched."""
    from hybrid_ctunet_amd.train import FlatParams
    torch.manual_seed(0)
    dim, hid = 64, 160
    params = dict(w1=torch.randn(hid, dim) * 0.1, b1=torch.randn(hid) * 0.1, w2=torch.randn(dim, hid) * 0.1,
                  g=1 + 0.1 * torch.randn(dim), be=0.1 * torch.randn(dim), unused=torch.randn(5))
    x = torch.randn(300, dim)
    gy = torch.randn(300, dim)

    def run(ps, xx):
        h = ops.layer_norm(xx, ps["g"], ps["be"])
        h = ops.linear(h, ps["w1"], ps["b1"], None, 1)
        return ops.linear(h, ps["w2"], None, xx, 0)

    ref = {k: torch.nn.Parameter(v.clone().cuda()) for k, v in params.items()}
    for _ in range(2):
        run(ref, x.cuda()).backward(gy.cuda())
    direct = {k: torch.nn.Parameter(v.clone().cuda()) for k, v in params.items()}
    fp = FlatParams(direct.values())
    for _ in range(2):
        run(direct, x.cuda()).backward(gy.cuda())
    for k in ("w1", "b1", "w2", "g", "be"):
        close(direct[k].grad, ref[k].grad.cpu().double(), torch.float32, k)
        assert direct[k].grad.data_ptr() >= fp.grad.data_ptr()
    assert fp.touched == [True, True, True, True, True, False]
    ops.clear_grad_sinks()

    # 27-tap conv weights: wgrad panel in the persistent scratch -> permute3(accumulate=2) into the flat gradient
    wc = torch.randn(64, 32, 3, 3, 3) * 0.05
    xc = torch.randn(2, 5, 9, 8, 32).cuda()
    gc = torch.randn(2, 5, 9, 8, 64).cuda()
    for dtype in DT:
        refw = torch.nn.Parameter(wc.clone().cuda())
        for _ in range(2):
            ops.conv3d(xc.to(dtype), refw, 1, 1).backward(gc.to(dtype))
        dirw = torch.nn.Parameter(wc.clone().cuda())
        fp = FlatParams([dirw])
        for _ in range(2):
            ops.conv3d(xc.to(dtype), dirw, 1, 1).backward(gc.to(dtype))
        close(dirw.grad, refw.grad.cpu().double(), torch.float32, f"conv weight {dtype}")
        assert dirw.grad.data_ptr() == fp.grad.data_ptr() and fp.touched == [True]
        ops.clear_grad_sinks()


def test_fused_adamw_matches_torch(ops):
    from hybrid_ctunet_amd.train import FusedAdamW
    torch.manual_seed(0)
    shapes = [(33, 7), (128,), (5, 3, 3, 3, 3), (1000,), (17,)]
    ps = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt = FusedAdamW(ps, lr=1e-2, weight_decay=1e-2)
    ropt = torch.optim.AdamW(ref, lr=1e-2, weight_decay=1e-2)
    for step in range(4):
        opt.zero_grad()
        for r in ref:
            r.grad = None
        for i, (p, r) in enumerate(zip(ps, ref)):
            if i == 2:
                continue  # never produces a gradient -> must be skipped like torch skips grad=None
            g = torch.randn(p.shape, device="cuda")
            (p * g).sum().backward()
            r.grad = g.clone()
        opt.step()
        ropt.step()
        for p, r in zip(ps, ref):
            assert torch.allclose(p, r, rtol=1e-5, atol=1e-6), step


def test_batched_fragment_packing_equals_single(ops):
    """From the second sight of a conv weight on, all registered 3x3x3 panels (forward and data-gradient form) are repacked
    by ONE launch per weight update; outputs and gradients must be bit-identical to the one-launch-per-panel path."""
    torch.manual_seed(0)
    dtype = torch.bfloat16
    shapes = [(64, 32), (32, 96), (128, 64)]           # (N, C): different panel sizes in one batch
    ws = [torch.nn.Parameter((torch.randn(n, c, 3, 3, 3) / math.sqrt(27 * c)).cuda()) for n, c in shapes]
    xs = [torch.randn(1, 8, 8, 8, c, device="cuda").to(dtype).requires_grad_(True) for _, c in shapes]

    def run(batched):
        ops.BATCH_PACK = batched
        outs = []
        for rnd_ in range(3):                          # round 0 registers, rounds 1-2 take the batched path
            with torch.no_grad():
                for w in ws:
                    w.mul_(1.0 + 0.01 * (rnd_ + 1))    # a weight update the version counter sees
            ops.bump_weights_epoch()                   # ... and one it does not (fused AdamW)
            for x in xs:
                x.grad = None
            ys = [ops.conv3d(x, w, 1, 1) for x, w in zip(xs, ws)]
            for y in ys:
                y.float().square().mean().backward()
            outs.append([y.detach().clone() for y in ys] + [x.grad.clone() for x in xs])
        return outs

    try:
        ref_w = [w.detach().clone() for w in ws]
        a = run(False)
        with torch.no_grad():
            for w, r in zip(ws, ref_w):
                w.copy_(r)
        b = run(True)
    finally:
        ops.BATCH_PACK = True
    for ra, rb in zip(a, b):
        for ta, tb in zip(ra, rb):
            assert torch.equal(ta, tb)


@pytest.mark.parametrize("case", [(2, 8, 16, 16, 32, 32), (1, 5, 9, 11, 64, 96), (2, 4, 8, 8, 128, 64),
                                  (2, 12, 12, 24, 128, 128)])   # (the last: batch-pair bricks reading the blocked layout)
def test_b16_layout_chain_equals_channels_last(ops, case):
    """CTU_LAYOUT_B16 (the 16-channel-blocked tensor between an InstanceNorm and the 3x3x3 halo convolution behind it, and
    between an InstanceNorm backward and the convolution in front of it) changes where bytes live, not what is computed:
    conv1 -> IN+LReLU -> conv2 -> IN+LReLU with the layout on and off must give the same output, input gradient and
    weight gradients (the only differences allowed: the arrival order of fp64 / fp32 atomics)."""
    B, D, H, W, C, N = case
    dtype = torch.bfloat16
    x0 = rnd((B, D, H, W, C), 11).to(dtype).cuda()
    w1 = torch.nn.Parameter(rnd((N, C, 3, 3, 3), 12, 1 / math.sqrt(27 * C)).float().cuda())
    w2 = torch.nn.Parameter(rnd((N, N, 3, 3, 3), 13, 1 / math.sqrt(27 * N)).float().cuda())
    gout = rnd((B, D, H, W, N), 14).to(dtype).cuda()

    def run(b16):
        ops.B16_LAYOUT = b16
        try:
            x = x0.clone().requires_grad_(True)
            for w in (w1, w2):
                w.grad = None
            y1 = ops.conv3d(x, w1, 1, 1)
            use = ops.wants_b16(w2, y1, 1, 1)
            assert use == b16
            a1 = ops.instance_norm(y1, None, True, out_b16=use)
            y2 = ops.conv3d(a1, w2, 1, 1)
            out = ops.instance_norm(y2, None, True)
            out.backward(gout)
            torch.cuda.synchronize()
            return out.detach().float(), x.grad.float(), w1.grad.clone(), w2.grad.clone()
        finally:
            ops.B16_LAYOUT = True
    ref = run(False)
    got = run(True)
    for name, a, b in zip(("out", "dx", "dw1", "dw2"), got, ref):
        scale = b.abs().max().item()
        if name.startswith("dw"):   # fp32 atomics into the weight-gradient panel arrive in any order
            assert (a - b).abs().max().item() <= 2e-3 * scale, name
        else:
            assert (a - b).abs().max().item() <= 1e-2 * scale, name
            assert (a == b).float().mean().item() > 0.99, name


@pytest.mark.parametrize("case", [(2, 8, 16, 16, 64, 64), (1, 5, 9, 11, 32, 32), (2, 4, 8, 8, 96, 64)])
def test_conv3d_halo_long_weight_stages_bit_equal(ops, case):
    """ctu_set_option("route", 16): the halo kernels with one or two n tiles fetch their weights in 9-tap stages (three per half
    chunk) instead of 3-tap stages.  Same products in the same order into the same accumulators: forward output and data
    gradient must be bit-equal to the default kernel's."""
    from hybrid_ctunet_amd import _lib
    B, D, H, W, C, N = case
    x0 = rnd((B, D, H, W, C), 21).to(torch.bfloat16).cuda()
    w = torch.nn.Parameter(rnd((N, C, 3, 3, 3), 22, 1 / math.sqrt(27 * C)).float().cuda())
    gout = rnd((B, D, H, W, N), 23).to(torch.bfloat16).cuda()

    def run(route):
        _lib.call("ctu_set_option", b"route", route)
        try:
            x = x0.clone().requires_grad_(True)
            w.grad = None
            y = ops.conv3d(x, w, 1, 1)
            y.backward(gout)
            torch.cuda.synchronize()
            return y.detach().clone(), x.grad.clone()
        finally:
            _lib.call("ctu_set_option", b"route", 0)
    a, b = run(0), run(16)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.parametrize("route", [32])
@pytest.mark.parametrize("case", [(2, 8, 16, 16, 64, 64), (1, 5, 9, 11, 128, 128), (2, 4, 8, 8, 96, 64), (1, 9, 17, 10, 64, 256),
                                  (1, 6, 9, 9, 64, 32)])
def test_conv3d_halo_spread_gather_bit_equal(ops, case, route):
    """The halo gather of the LDS-DMA kernels is dealt over all four waves (7 + 3 x 4 pieces, counted vmcnt waits on the
    weight-loader waves); ctu_set_option("route", 32) restores the rule it replaced (wave 0 issues all 19 pieces).  Same LDS image,
    same products: forward output and data gradient must be bit-equal (ragged volumes, one / two / four n tiles per workgroup,
    several half chunks)."""
    from hybrid_ctunet_amd import _lib
    B, D, H, W, C, N = case
    x0 = rnd((B, D, H, W, C), 41).to(torch.bfloat16).cuda()
    w = torch.nn.Parameter(rnd((N, C, 3, 3, 3), 42, 1 / math.sqrt(27 * C)).float().cuda())
    gout = rnd((B, D, H, W, N), 43).to(torch.bfloat16).cuda()

    def run(rt):
        _lib.call("ctu_set_option", b"route", rt)
        try:
            x = x0.clone().requires_grad_(True)
            w.grad = None
            y = ops.conv3d(x, w, 1, 1)
            y.backward(gout)
            torch.cuda.synchronize()
            return y.detach().clone(), x.grad.clone()
        finally:
            _lib.call("ctu_set_option", b"route", 0)
    a, b = run(0), run(route)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.parametrize("case", [(2, 8, 16, 16, 64, 64), (1, 5, 9, 11, 128, 96), (2, 4, 8, 8, 32, 32), (1, 9, 17, 10, 64, 256)])
def test_conv3d_halo_weight_gradient_partial_panels(ops, case):
    """The weight-gradient halo kernel stores one partial panel per brick split and a second kernel adds them in a fixed order
    (ops.WGRAD_PARTIALS, workspace argument of ctu_conv3_halo_wgrad); ctu_set_option("route", 128) keeps the fp32 atomics.  Two
    runs with partial panels must be bit-equal (no atomics left in the path), and both routes agree to fp32 rounding of the
    split sums.  Ragged volumes, one- and two-n-tile workgroups, N not a multiple of 64."""
    from hybrid_ctunet_amd import _lib
    B, D, H, W, C, N = case
    x0 = rnd((B, D, H, W, C), 51).to(torch.bfloat16).cuda()
    w = torch.nn.Parameter(rnd((N, C, 3, 3, 3), 52, 1 / math.sqrt(27 * C)).float().cuda())
    gout = rnd((B, D, H, W, N), 53).to(torch.bfloat16).cuda()

    def run(route):
        _lib.call("ctu_set_option", b"route", route)
        try:
            w.grad = None
            y = ops.conv3d(x0, w, 1, 1)
            y.backward(gout)
            torch.cuda.synchronize()
            return w.grad.clone()
        finally:
            _lib.call("ctu_set_option", b"route", 0)
    a, b, c = run(0), run(0), run(128)
    assert torch.equal(a, b)
    scale = c.abs().max().item()
    assert (a - c).abs().max().item() <= 2e-5 * scale + 1e-30


@pytest.mark.parametrize("case", [(2, 8, 16, 16, 64, 64), (1, 5, 9, 11, 128, 128), (2, 4, 8, 8, 32, 32), (1, 8, 8, 16, 96, 72)])
def test_conv3d_halo_weight_gradient_param_layout(ops, case):
    """ctu_conv3_halo_wgrad_param adds the sum of the per-split partial panels straight into a gradient in the PARAMETER layout
    [N][C][27] (one contiguous run of PAIRS x 27 elements per workgroup; 16 pairs per workgroup below 16 384 (n, c) pairs, else 32).
    Against ctu_conv3_halo_wgrad (panel [27][N][C]) on the same operands and workspace: equal to fp32 rounding of the split sums,
    the previous content of the gradient kept (+=), two runs bit-equal."""
    from hybrid_ctunet_amd import _lib
    from hybrid_ctunet_amd._lib import call, dcode, ptr, stream
    B, D, H, W, C, N = case
    x = rnd((B, D, H, W, C), 61).to(torch.bfloat16).cuda()
    dy = rnd((B, D, H, W, N), 62).to(torch.bfloat16).cuda()
    ws = torch.empty(64 << 20, device="cuda")
    panel = torch.zeros(27, N, C, device="cuda")
    call("ctu_conv3_halo_wgrad", dcode(torch.bfloat16), ptr(dy), ptr(x), None, ptr(panel), B, D, H, W, C, 0, N, 0, 0, ptr(ws),
         ws.numel(), stream())
    base = rnd((N, C, 27), 63).float().cuda()
    outs = []
    for _ in range(2):
        g = base.clone()
        call("ctu_conv3_halo_wgrad_param", dcode(torch.bfloat16), ptr(dy), ptr(x), None, ptr(g), B, D, H, W, C, 0, N, 0, 0, ptr(ws),
             ws.numel(), stream())
        torch.cuda.synchronize()
        outs.append(g)
    assert torch.equal(outs[0], outs[1])
    want = base + panel.permute(1, 2, 0)
    scale = panel.abs().max().item()
    assert (outs[0] - want).abs().max().item() <= 2e-5 * scale + 1e-6


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(2, 3, 5, 4, 32), (1, 6, 6, 12, 72)])
def test_stride2_helpers(ops, dtype, shape):
    """ctu_upsample2_zeros (y[2d][2h][2w] = x, zero elsewhere) and ctu_add_strided2 (y[2d][2h][2w] += x), and what the launch-list
    path builds from the first: the data gradient of a 3x3x3 stride-2 padding-1 convolution = the stride-1 data gradient kernel
    (ops.conv3d backward on the halo route) applied to the zero-upsampled dY."""
    from hybrid_ctunet_amd._lib import call, dcode, ptr, stream
    B, D, H, W, C = shape
    x = rnd((B, D, H, W, C), 81).to(dtype).cuda()
    up = torch.full((B, 2 * D, 2 * H, 2 * W, C), 7.0, dtype=dtype, device="cuda")
    call("ctu_upsample2_zeros", dcode(dtype), ptr(x), ptr(up), B, D, H, W, C, stream())
    want = torch.zeros_like(up)
    want[:, ::2, ::2, ::2] = x
    assert torch.equal(up, want)
    y0 = rnd((B, 2 * D, 2 * H, 2 * W, C), 82).to(dtype).cuda()
    y = y0.clone()
    call("ctu_add_strided2", dcode(dtype), ptr(y), ptr(x), B, D, H, W, C, stream())
    ref = y0.float()
    ref[:, ::2, ::2, ::2] += x.float()
    assert torch.equal(y, ref.to(dtype))
    if C % 32 == 0 and dtype == torch.bfloat16:
        w = torch.nn.Parameter(rnd((C, C, 3, 3, 3), 83, 1 / math.sqrt(27 * C)).float().cuda())
        xin = rnd((B, 2 * D, 2 * H, 2 * W, C), 84).to(dtype).cuda().requires_grad_(True)
        ops.conv3d(xin, w, 2, 1).backward(x)                     # generic implicit GEMM, strided data gradient
        x1 = xin.detach().clone().requires_grad_(True)
        ops.conv3d(x1, w, 1, 1).backward(up)                     # stride-1 (halo) data gradient of the upsampled dY
        scale = xin.grad.float().abs().max().item()
        assert (xin.grad.float() - x1.grad.float()).abs().max().item() <= 2 ** -7 * scale


@pytest.mark.parametrize("two", [False, True])
def test_resblock_conv_shortcut_gradients_folded_into_conv1(ops, two):
    """ResBlock with a conv shortcut (in != out channels): conv3 reads the same input (or channel-concatenated pair of inputs) as
    conv1.  Its data gradients are parked (GradStash) and added in the epilogue of conv1's data-gradient halo kernel - for a
    pair of inputs through `residual` AND `residual2` of ctu_conv3_halo - instead of by autograd's accumulation passes.  Input
    gradients must agree with the accumulated ones to one bf16 rounding (the folded form rounds once), weight gradients
    exactly (they do not depend on where the sum is taken)."""
    from hybrid_ctunet_amd.networks.hybrid_CTUNet import ResBlock
    torch.manual_seed(0)
    C1, C2, N = 64, (64 if two else 0), 32
    blk = ResBlock(3, C1 + C2, N, 3, 1, "instance").cuda()
    a0 = rnd((1, 6, 9, 10, C1), 61).to(torch.bfloat16).cuda()
    b0 = rnd((1, 6, 9, 10, C2), 62).to(torch.bfloat16).cuda() if two else None
    gy = rnd((1, 6, 9, 10, N), 63).to(torch.bfloat16).cuda()

    def run(flag):
        ops.STASH_SHORTCUT_CONV = flag
        try:
            a = a0.clone().requires_grad_(True)
            b = b0.clone().requires_grad_(True) if two else None
            for p in blk.parameters():
                p.grad = None
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = blk(a, b)
            y.backward(gy)
            torch.cuda.synchronize()
            return [a.grad.clone()] + ([b.grad.clone()] if two else []), [blk.conv1.conv.weight.grad.clone(), blk.conv3.conv.weight.grad.clone()]
        finally:
            ops.STASH_SHORTCUT_CONV = True
    (gi_f, gw_f), (gi_a, gw_a) = run(True), run(False)
    for u, v in zip(gi_f, gi_a):
        scale = v.float().abs().max().item()
        assert (u.float() - v.float()).abs().max().item() <= 2 ** -7 * scale
        assert (u.float() - v.float()).abs().mean().item() <= 2 ** -9 * v.float().abs().mean().item()
    for u, v in zip(gw_f, gw_a):
        assert torch.allclose(u, v, rtol=1e-5, atol=1e-6 * v.abs().max().item())


@pytest.mark.parametrize("dtype", DT)
def test_instance_norm_sign_mask_equals_reading_y(ops, dtype):
    """With a residual, the backward kernels take the LeakyReLU mask from the byte-per-8-channels sign mask the forward wrote
    (ops.SIGN_MASK) instead of re-reading y: same gradients, bit for bit (the mask records sign(pre-activation), y > 0 says
    the same thing)."""
    B, D, H, W, C = 2, 6, 7, 9, 64
    x0 = cl(rnd((B, C, D, H, W), 31, 2.0) + 0.2).to(dtype).cuda()
    r0 = cl(rnd((B, C, D, H, W), 32)).to(dtype).cuda()
    gy = cl(rnd((B, C, D, H, W), 33)).to(dtype).cuda()

    def run(flag):
        ops.SIGN_MASK = flag
        try:
            x, r = x0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
            y = ops.instance_norm(x, r, True)
            y.backward(gy)
            torch.cuda.synchronize()
            return y.detach().clone(), x.grad.clone(), r.grad.clone()
        finally:
            ops.SIGN_MASK = True
    a, b = run(True), run(False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
    assert (a[1].float() - b[1].float()).abs().max() <= 1e-2 * b[1].float().abs().max()   # fp64 atomics order in the sums
    assert (a[1] == b[1]).float().mean() > 0.99


def test_grad_stash_without_its_consumer_raises(ops):
    """A gradient parked by GradStash is added by a specific later layer's data-gradient kernel.  If that layer is not part of
    the backward pass (here: only the stashed branch is differentiated) the gradient must not vanish silently."""
    w = torch.nn.Parameter(torch.randn(32, 32, device="cuda") * 0.1)
    x = torch.randn(64, 32, device="cuda", requires_grad=True)
    slot = []
    main = ops.linear(x, w, grad_stash=slot)          # the consumer: would add the parked gradient in its epilogue
    side = ops.GradStash.apply(x, slot)                # the other consumer of x parks its gradient
    with pytest.raises(RuntimeError, match="parked"):
        (side * 2.0).sum().backward()                  # ... but only the side branch is backpropagated
    slot2 = []
    main = ops.linear(x, w, grad_stash=slot2)
    side = ops.GradStash.apply(x, slot2)
    x.grad = None
    (main.sum() + (side * 2.0).sum()).backward()       # both: the parked gradient arrives through the consumer
    torch.cuda.synchronize()
    ref = (torch.ones(64, 32, device="cuda") @ w.detach()) + 2.0
    assert torch.allclose(x.grad, ref, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,S,C,act,use_mask,use_y,want_res,b16", [
    (2, 3456, 128, 1, False, False, False, 0),     # layer3 gn1 / gn2
    (2, 3456, 512, 1, True, False, True, 0),       # layer3 gn3: sign mask + gradient of the shortcut
    (2, 432, 1024, 1, False, True, False, 0),      # layer4 gn3 reading y; 27 rows of 16 per item
    (2, 27648, 64, 1, False, False, False, 1),     # layer2 gn1: dx in CTU_LAYOUT_B16
    (2, 1000, 96, 0, False, False, False, 0),      # no activation (shortcut norm), ragged rows, C / 8 not a power of two
    (3, 777, 2048, 1, False, False, False, 0),     # widest row, three items
    (1, 221184, 32, 1, False, False, False, 1),    # 864 rows per workgroup
])
def test_instance_norm_backward_in_one_launch(ops, dtype, B, S, C, act, use_mask, use_y, want_res, b16):
    """ctu_in_bwd_fused (reduce, meet at a counter, apply: one launch) against the ctu_in_bwd_reduce + ctu_in_bwd_apply pair on the
    same tensors - same arithmetic, only the order of the fp64 atomics is free - and against float64 math; run three times on one
    sync workspace (the counters hand themselves back zeroed) with the two sums buffers alternating as the launch lists do."""
    from hybrid_ctunet_amd._lib import call, dcode, lib, ptr, stream
    if b16 and dtype != torch.bfloat16:
        pytest.skip("CTU_LAYOUT_B16 is a bf16 layout")
    g = torch.Generator().manual_seed(S + C)
    x = (torch.randn(B, S, C, generator=g) * 1.5 + 0.3).to(dtype).cuda()
    dy = torch.randn(B, S, C, generator=g).to(dtype).cuda()
    res = torch.randn(B, S, C, generator=g).to(dtype).cuda() if (use_mask or use_y) else None
    xd = x.double()
    mean, var = xd.mean(1, keepdim=True), xd.var(1, unbiased=False, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    stats = torch.stack([mean.squeeze(1), rstd.squeeze(1)], dim=-1).float().contiguous()
    xhat = (xd - mean) * rstd
    pre = xhat + (res.double() if res is not None else 0.0)
    yv = torch.where(pre > 0, pre, 0.01 * pre).to(dtype) if act else pre.to(dtype)
    mask = None
    if use_mask:
        bits = (pre > 0).view(B, S, C // 8, 8).to(torch.int32) << torch.arange(8, device="cuda", dtype=torch.int32)
        mask = bits.sum(-1).to(torch.uint8).contiguous()
    ypass = yv if use_y else None
    gref = dy.double() * (torch.where(pre > 0, 1.0, 0.01) if act else 1.0)
    dx_ref = rstd * (gref - gref.mean(1, keepdim=True) - xhat * (gref * xhat).mean(1, keepdim=True))

    def unblock(t):   # CTU_LAYOUT_B16 [C/16][B*S][16] -> [B, S, C]
        return t.view(C // 16, B * S, 16).permute(1, 0, 2).reshape(B, S, C) if b16 else t
    sums = [torch.zeros(B * C * 2, dtype=torch.float64, device="cuda") for _ in range(2)]
    sync = torch.zeros(128, dtype=torch.int32, device="cuda")
    dx_p, dres_p = torch.empty_like(x), (torch.empty_like(x) if want_res else None)
    call("ctu_in_bwd_reduce", dcode(dtype), ptr(dy), ptr(x), ptr(ypass), ptr(stats), ptr(sums[0]), B, S, C, act, ptr(mask), stream())
    call("ctu_in_bwd_apply", dcode(dtype), ptr(dy), ptr(x), ptr(ypass), ptr(stats), ptr(sums[0]), ptr(dx_p), ptr(dres_p), B, S, C, act,
         ptr(sums[1]), B * C * 2, b16, ptr(mask), stream())
    sums_pair = sums[0].clone()
    sums[0].zero_()
    for rep in range(3):
        cur, other = sums[rep & 1], sums[1 - (rep & 1)]
        dx_f, dres_f = torch.full_like(x, float("nan")), (torch.full_like(x, float("nan")) if want_res else None)
        call("ctu_in_bwd_fused", dcode(dtype), ptr(dy), ptr(x), ptr(ypass), ptr(stats), ptr(cur), ptr(dx_f), ptr(dres_f), B, S, C, act,
             ptr(other), B * C * 2, b16, ptr(mask), ptr(sync), stream())
        torch.cuda.synchronize()
        assert int(sync.abs().sum()) == 0, "the counters come back zeroed"
        assert int(other.abs().sum()) == 0, "the other sums buffer was cleared"
        assert torch.allclose(cur, sums_pair, rtol=1e-9, atol=1e-9 * float(sums_pair.abs().max()) + 1e-12)
        close(unblock(dx_f), dx_ref.cpu(), dtype, f"dx[{rep}] vs float64", scale=float(dx_ref.abs().max()))
        d = (unblock(dx_f).float() - unblock(dx_p).float()).abs().max().item()
        # (fp32 outputs show the last bit of the differently ordered fp64 sums; bf16 outputs round it away almost everywhere)
        lim, same = (2e-2, 0.97) if dtype == torch.bfloat16 else (1e-5, 0.5)
        assert d <= lim * dx_p.float().abs().max().item() and (dx_f == dx_p).float().mean().item() > same, ("vs the pair", d)
        if want_res:
            assert torch.equal(dres_f, dres_p)
    assert lib().ctu_sync_timeouts() == 0


@pytest.mark.parametrize("case", [(2, 8, 16, 16, 128, 128, 0), (1, 5, 9, 11, 128, 128, 0), (2, 4, 8, 8, 96, 256, 0), (1, 9, 17, 10, 64, 128, 64),
                                  (1, 6, 9, 9, 32, 128, 0), (2, 6, 6, 12, 256, 256, 0)])
def test_conv3d_halo_n_split_bit_equal(ops, case):
    """conv3_halo_ns_kernel (ctu_set_option("route", 4096): a wave owns one n tile for the whole brick, wave-private weight rings, one
    barrier per half chunk) against conv3_halo_dma_kernel: the same products summed in the same order per accumulator, so forward
    output and data gradient are bit-equal; the fused InstanceNorm sums agree to fp32 rounding of the per-brick partial sums.
    Ragged volumes, concat input (second source), one to sixteen half chunks, the channel-split small-volume path, residual inputs
    (GradStash route of the data gradient)."""
    from hybrid_ctunet_amd import _lib
    from hybrid_ctunet_amd._lib import call, dcode, ptr, stream
    B, D, H, W, C, N, C2 = case
    x0 = rnd((B, D, H, W, C), 61).to(torch.bfloat16).cuda()
    x2 = rnd((B, D, H, W, C2), 64).to(torch.bfloat16).cuda() if C2 else None
    w = torch.nn.Parameter(rnd((N, C + C2, 3, 3, 3), 62, 1 / math.sqrt(27 * (C + C2))).float().cuda())
    gout = rnd((B, D, H, W, N), 63).to(torch.bfloat16).cuda()

    def run(rt):
        _lib.call("ctu_set_option", b"route", rt)
        try:
            x = x0.clone().requires_grad_(True)
            xb = x2.clone().requires_grad_(True) if x2 is not None else None
            w.grad = None
            y = ops.conv3d(x, w, 1, 1, x2=xb) if xb is not None else ops.conv3d(x, w, 1, 1)
            y.backward(gout)
            torch.cuda.synchronize()
            return y.detach().clone(), x.grad.clone(), (xb.grad.clone() if xb is not None else None)
        finally:
            _lib.call("ctu_set_option", b"route", 0)
    a, b = run(0), run(4096)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    if x2 is not None:
        assert torch.equal(a[2], b[2])
    # fused statistics + a residual in the epilogue, straight through the C ABI
    wfr = ops._pack_frag(w, N, C + C2, 27, (C + C2) * 27, 27, 1, 0, torch.bfloat16)
    res = rnd((B, D, H, W, N), 65).to(torch.bfloat16).cuda()
    ws = torch.zeros(1 << 22, device="cuda")
    outs, accs = [], []
    for rt in (0, 4096):
        _lib.call("ctu_set_option", b"route", rt)
        try:
            out = torch.empty(B, D, H, W, N, device="cuda", dtype=torch.bfloat16)
            acc = torch.zeros(B * N * 2, device="cuda", dtype=torch.float64)
            call("ctu_conv3_halo", dcode(torch.bfloat16), ptr(x0), ptr(x2), ptr(wfr), ptr(out), None, B, D, H, W, C, C2, N, 0, N, 0,
                 ptr(acc), None, None, None, 0, 0, stream())
            out_r = torch.empty_like(out)
            call("ctu_conv3_halo", dcode(torch.bfloat16), ptr(x0), ptr(x2), ptr(wfr), ptr(out_r), None, B, D, H, W, C, C2, N, 0, N, 0,
                 None, ptr(res), None, None, 0, 0, stream())
            torch.cuda.synchronize()
            outs.append((out, out_r))
            accs.append(acc)
        finally:
            _lib.call("ctu_set_option", b"route", 0)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.allclose(accs[0], accs[1], rtol=1e-5, atol=1e-5 * float(accs[0].abs().max()))
    del ws


@pytest.mark.parametrize("M,Hd", [(512, 512), (2048, 128), (768, 1024)])
def test_fused_feedforward_forward(ops, M, Hd):
    """ctu_ff_fwd (LayerNorm -> W1 -> GELU -> W2 + residual in one kernel, hybrid_CTUNet.py:513-526) against float64 math on the
    values the device holds: y, the stored pre-activation, gelu(pre) and the LayerNorm statistics."""
    from hybrid_ctunet_amd._lib import call, dcode, ptr, stream
    D = 128
    dt = torch.bfloat16
    x, xh = dev(rnd((M, D), 71, 2.0) + 0.3, dt)
    g, gh = dev(1 + 0.2 * rnd((D,), 72), torch.float32)
    b, bh = dev(0.1 * rnd((D,), 73), torch.float32)
    w1, w1h = dev(rnd((Hd, D), 74, 1 / math.sqrt(D)), dt)
    b1, b1h = dev(0.2 * rnd((Hd,), 75), torch.float32)
    w2, w2h = dev(rnd((D, Hd), 76, 1 / math.sqrt(Hd)), dt)
    b2, b2h = dev(0.2 * rnd((D,), 77), torch.float32)
    w2f = torch.empty(D * Hd, device="cuda", dtype=dt)
    call("ctu_ff_pack_w2", ptr(w2), ptr(w2f), D, Hd, stream())
    y = torch.empty(M, D, device="cuda", dtype=dt)
    pre = torch.empty(M, Hd, device="cuda", dtype=dt)
    u = torch.empty(M, Hd, device="cuda", dtype=dt)
    mr = torch.empty(M, 2, device="cuda")
    for _ in range(2):   # (second call: the stage ring starts from the state the first left)
        call("ctu_ff_fwd", dcode(dt), ptr(x), ptr(g), ptr(b), ptr(w1), ptr(b1), ptr(w2f), ptr(b2), ptr(y), ptr(pre), ptr(u), ptr(mr),
             M, D, Hd, stream())
    torch.cuda.synchronize()
    mean = xh.mean(1, keepdim=True)
    var = xh.var(1, unbiased=False, keepdim=True)
    rstd = 1 / torch.sqrt(var + 1e-5)
    h = ((xh - mean) * rstd * gh + bh).to(dt).double()       # the kernel rounds the normalised rows to bf16 (MFMA operand)
    pre_ref = h @ w1h.t() + b1h
    u_ref = F.gelu(pre_ref)
    y_ref = xh + u_ref.to(dt).double() @ w2h.t() + b2h       # (and gelu(pre) likewise)
    close(mr[:, 0], mean.squeeze(1), torch.float32, "mean")
    close(mr[:, 1], rstd.squeeze(1), torch.float32, "rstd")
    close(pre, pre_ref, dt, "pre")
    close(u, u_ref, dt, "u")
    close(y, y_ref, dt, "y")


@pytest.mark.parametrize("M,K,N,variant", [(36864, 128, 32, "stats"), (36864, 128, 32, "plain"), (36864, 128, 32, "dgrad"),
                                           (55296, 64, 64, "stats"), (32768, 32, 32, "plain"), (36864, 64, 32, "dgrad")])
def test_narrow_output_gemm(ops, M, K, N, variant):
    """gemm_nt_narrow (N = 32 / 64 output columns, M >= 32 768 rows: conv1 of the ResNet bottlenecks, resnet.py:96, and the data
    gradient of their conv3, :100): result and fused InstanceNorm sums against float64, and bit-equal to the general LDS-DMA
    kernel (ctu_set_option("route", 16384)) - the same products in the same order per accumulator."""
    from hybrid_ctunet_amd import _lib
    dt = torch.bfloat16
    x, xh = dev(rnd((M, K), 91), dt)
    B = 2
    if variant == "dgrad":       # the forward weight of a K_fwd = N -> N_fwd = K layer, read reduction-major
        w, wh = dev(rnd((K, N), 92, 1 / math.sqrt(K)), dt)
        ref = xh @ wh
    else:
        w, wh = dev(rnd((N, K), 92, 1 / math.sqrt(K)), dt)
        ref = xh @ wh.t()
    outs, accs = [], []
    for route in (65536, 16384):    # (65536: the streaming kernel also where statistics are summed - off by default, gemm_narrow.hip)
        _lib.call("ctu_set_option", b"route", route)
        try:
            out = torch.empty(M, N, device="cuda", dtype=dt)
            acc = torch.zeros(B * N * 2, device="cuda", dtype=torch.float64) if variant == "stats" else None
            if variant == "dgrad":
                ops._plain_gemm(x, w, out, M, K, N, w_kn=1)
            else:
                ops._plain_gemm(x, w, out, M, K, N, in_acc=acc, in_rows=M // B if acc is not None else 0)
            torch.cuda.synchronize()
            outs.append(out)
            accs.append(acc)
        finally:
            _lib.call("ctu_set_option", b"route", 0)
    close(outs[0], ref, dt, "out")
    assert torch.equal(outs[0], outs[1])
    if variant == "stats":
        o = outs[0].double().cpu().view(B, M // B, N)      # (the sums are taken from the fp32 accumulators, before rounding)
        s_ref = torch.stack([ref.view(B, M // B, N).sum(1), (ref.view(B, M // B, N) ** 2).sum(1)], -1).reshape(-1)
        got = accs[0].cpu()
        assert torch.allclose(got, s_ref, rtol=2e-3, atol=2e-3 * float(s_ref.abs().max())), (got[:4], s_ref[:4])
        assert torch.allclose(accs[0], accs[1], rtol=1e-5, atol=1e-6 * float(accs[1].abs().max()))
        del o


@pytest.mark.parametrize("B,din,k,s,C,N", [
    (2, (12, 12, 24), (2, 2, 2), (2, 2, 2), 128, 256),     # a patch convolution of the ViT branch (hybrid_CTUNet.py:66-83)
    (1, (8, 12, 20), (2, 2, 1), (2, 2, 1), 64, 128),       # data / weight gradient shape of ConvTranspose3d (2,2,1) (:286-294)
    (2, (6, 10, 14), (1, 1, 1), (2, 2, 2), 192, 96),       # 1x1x1 stride-2 shortcut (resnet.py:197)
    (1, (7, 9, 11), (2, 2, 2), (2, 2, 2), 64, 40),         # grid not covered by the taps; N tail
    (3, (16, 16, 16), (2, 2, 2), (2, 2, 2), 256, 512),     # several m tiles per workgroup column
])
def test_gathered_gemm_in_grid_taps(ops, B, din, k, s, C, N):
    """Convolutions whose taps never leave the grid (no padding) run on the LDS-DMA GEMM kernels with a gathered operand
    (GatherGeom, gemm_dma.h): forward / data-gradient form (ctu_igemm_nt) and weight-gradient form (ctu_igemm_tn) against
    float64, and against the generic implicit-GEMM kernels (ctu_set_option("route", 131072))."""
    from hybrid_ctunet_amd import _lib
    dt = torch.bfloat16
    dout = tuple((n - kk) // ss + 1 for n, kk, ss in zip(din, k, s))
    taps = k[0] * k[1] * k[2]
    M = B * dout[0] * dout[1] * dout[2]
    x, xh = dev(rnd((B, *din, C), 5), dt)
    w, wh = dev(rnd((taps, N, C), 6, 1 / math.sqrt(taps * C)), dt)
    gy, gyh = dev(rnd((M, N), 7), dt)
    # gathered rows [M][taps][C] in float64
    cols = []
    for td in range(k[0]):
        for th in range(k[1]):
            for tw in range(k[2]):
                cols.append(xh[:, td:td + (dout[0] - 1) * s[0] + 1:s[0], th:th + (dout[1] - 1) * s[1] + 1:s[1],
                               tw:tw + (dout[2] - 1) * s[2] + 1:s[2], :].reshape(M, C))
    G = torch.stack(cols, 1)
    out_ref = torch.einsum("mtc,tnc->mn", G, wh)
    dw_ref = torch.einsum("mn,mtc->tnc", gyh, G)
    g = ops._geom(B, din, dout, C, 0, N, k, s, (0, 0, 0), 0)
    res = []
    for route in (0, 131072):
        _lib.call("ctu_set_option", b"route", route)
        try:
            out = torch.empty(M, N, device="cuda", dtype=dt)
            ops._igemm_nt(x, None, w, out, g, ops._epi(N))
            dw = torch.zeros(taps, N, C, device="cuda")
            ops._igemm_tn(gy, N, x, None, dw, g)
            torch.cuda.synchronize()
            res.append((out, dw))
        finally:
            _lib.call("ctu_set_option", b"route", 0)
    close(res[0][0], out_ref, dt, "out")
    close(res[0][1], dw_ref, torch.float32, "dw", scale=float(dw_ref.abs().max()) * 4)
    close(res[1][0], out_ref, dt, "out (generic)")
    close(res[0][0], res[1][0].double().cpu(), dt, "out: DMA against generic")
    # the epilogues the model puts behind these layers: split K through a workspace (patch convolutions at 864 rows), residual
    sk = 4 if taps * C >= 1024 else 1
    ws = torch.zeros(M * N, device="cuda") if sk > 1 else None
    resid, residh = dev(rnd((M, N), 8), dt)
    out2 = torch.empty(M, N, device="cuda", dtype=dt)
    ops._igemm_nt(x, None, w, out2, g, ops._epi(N, residual=resid, splitk_ws=ws, splitk=sk))
    torch.cuda.synchronize()
    close(out2, out_ref + residh, dt, f"out + residual, split K {sk}")


@pytest.mark.parametrize("M,save", [(512, True), (2304, True), (768, False)])
def test_fused_cross_weight_forward(ops, M, save):
    """ctu_pwa_block_fwd (pixelweight_attention.forward in one kernel, hybrid_CTUNet.py:645-669) against float64 math on the values
    the device holds: the output, the saved projections and both LayerNorm statistics; and against the six-launch path."""
    from hybrid_ctunet_amd._lib import call, dcode, ptr, stream
    C = 128
    dt = torch.bfloat16
    scale = 32 ** -0.5
    x1, x1h = dev(rnd((M, C), 81, 2.0) + 0.3, dt)
    x2, x2h = dev(rnd((M, C), 82, 1.5) - 0.2, dt)
    par = {}
    for i, n in enumerate(("g1", "g2")):
        par[n] = dev(1 + 0.2 * rnd((C,), 83 + i), torch.float32)
    for i, n in enumerate(("b1", "b2")):
        par[n] = dev(0.1 * rnd((C,), 85 + i), torch.float32)
    wq1, wq1h = dev(rnd((3 * C, C), 87, 2 / math.sqrt(C)), dt)
    wq2, wq2h = dev(rnd((3 * C, C), 88, 2 / math.sqrt(C)), dt)
    wo, woh = dev(rnd((C, C), 89, 1 / math.sqrt(C)), dt)
    wpk = torch.empty(4 * 56 * 512, device="cuda", dtype=dt)
    call("ctu_pwa_pack", ptr(wq1), ptr(wq2), ptr(wo), ptr(wpk), C, stream())
    out = torch.empty(M, C, device="cuda", dtype=dt)
    q1 = torch.empty(M, 3 * C, device="cuda", dtype=dt) if save else None
    q2 = torch.empty(M, 3 * C, device="cuda", dtype=dt) if save else None
    mr1, mr2 = torch.empty(M, 2, device="cuda"), torch.empty(M, 2, device="cuda")
    for _ in range(2):   # (second call: the stage ring starts from the state the first left)
        call("ctu_pwa_block_fwd", dcode(dt), ptr(x1), ptr(x2), ptr(par["g1"][0]), ptr(par["b1"][0]), ptr(par["g2"][0]), ptr(par["b2"][0]),
             ptr(wpk), ptr(out), ptr(q1), ptr(q2), ptr(mr1), ptr(mr2), M, C, scale, stream())
    torch.cuda.synchronize()

    def ln(xh, g, b):
        mean = xh.mean(1, keepdim=True)
        rstd = 1 / torch.sqrt(xh.var(1, unbiased=False, keepdim=True) + 1e-5)
        return ((xh - mean) * rstd * g + b).to(dt).double(), mean.squeeze(1), rstd.squeeze(1)   # (rounded: MFMA operand)

    h1, m1, r1 = ln(x1h, par["g1"][1], par["b1"][1])
    h2, m2, r2 = ln(x2h, par["g2"][1], par["b2"][1])
    p1 = (h1 @ wq1h.t()).to(dt).double()      # the projections are rounded to bf16 before the mix, as the stored copies are
    p2 = (h2 @ wq2h.t()).to(dt).double()
    qa, ka, va = (t.view(M, 4, 32) for t in p1.split(C, 1))
    qb, kb, vb = (t.view(M, 4, 32) for t in p2.split(C, 1))
    z = ((qb * ka).sum(-1) - (qa * kb).sum(-1)) * scale
    a1 = torch.sigmoid(z).unsqueeze(-1)
    o = (a1 * va + (1 - a1) * vb).reshape(M, C).to(dt).double()
    out_ref = o @ woh.t()
    close(mr1[:, 0], m1, torch.float32, "mean1"), close(mr1[:, 1], r1, torch.float32, "rstd1")
    close(mr2[:, 0], m2, torch.float32, "mean2"), close(mr2[:, 1], r2, torch.float32, "rstd2")
    if save:
        close(q1, p1, dt, "qkv1"), close(q2, p2, dt, "qkv2")
    close(out, out_ref, dt, "out")
    # the six-launch path on the same inputs
    hh1, hh2 = torch.empty_like(x1), torch.empty_like(x2)
    t1, t2 = torch.empty(M, 2, device="cuda"), torch.empty(M, 2, device="cuda")
    call("ctu_layernorm_fwd", dcode(dt), ptr(x1), ptr(par["g1"][0]), ptr(par["b1"][0]), ptr(hh1), ptr(t1), M, C, stream())
    call("ctu_layernorm_fwd", dcode(dt), ptr(x2), ptr(par["g2"][0]), ptr(par["b2"][0]), ptr(hh2), ptr(t2), M, C, stream())
    s1, s2 = torch.empty(M, 3 * C, device="cuda", dtype=dt), torch.empty(M, 3 * C, device="cuda", dtype=dt)
    ops._plain_gemm(hh1, wq1, s1, M, C, 3 * C)
    ops._plain_gemm(hh2, wq2, s2, M, C, 3 * C)
    oo, out6 = torch.empty(M, C, device="cuda", dtype=dt), torch.empty(M, C, device="cuda", dtype=dt)
    call("ctu_pwa_fwd", dcode(dt), ptr(s1), ptr(s2), ptr(oo), M, C, scale, stream())
    ops._plain_gemm(oo, wo, out6, M, C, C)
    torch.cuda.synchronize()
    close(out, out6.double().cpu(), dt, "out against the six-launch path")
    if save:
        assert (q1 == s1).float().mean().item() > 0.99 and (q2 == s2).float().mean().item() > 0.99
