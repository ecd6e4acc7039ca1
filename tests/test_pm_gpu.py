"""GPU parity: fused point-major kernels vs the CPU oracle (through layout transposes) and, for the
dense layers, vs a plain fp32 restatement evaluated in float64 numpy."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def close(a, b, rtol=1e-4, atol=1e-5):
    return np.allclose(a, b, rtol=rtol, atol=atol)


def relerr(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def _cloud(rng, B, N, K, oracle):
    xyz = rng.random((B, N, 3), dtype=np.float32)
    nn, _ = oracle.knn_bruteforce(np.ascontiguousarray(xyz.transpose(0, 2, 1)), K)
    return xyz, nn  # nn [B,N,K]


def test_mfma_fragment_layout_identity_probe(dev):
    """A = I against an ASYMMETRIC W: catches row/col swaps in the MFMA C/D mapping or the packing."""
    from dh3d_amd import pm
    Kd, Dout = 64, 64
    W = (np.arange(Kd)[:, None] * 1000 + np.arange(Dout)[None, :]).astype(np.float32)
    x = np.eye(64, Kd, dtype=np.float32)
    out = pm.linear(T(x, dev), pm.pack_weight(T(W, dev)), Dout)
    assert np.array_equal(out.cpu().numpy(), W)
    x = np.random.default_rng(0).integers(-3, 4, (200, Kd)).astype(np.float32)  # exact in fp32
    W2 = np.random.default_rng(1).integers(-3, 4, (Kd, 128)).astype(np.float32)
    out = pm.linear(T(x, dev), pm.pack_weight(T(W2, dev)), 128)
    assert np.array_equal(out.cpu().numpy(), x @ W2)


@pytest.mark.parametrize("Din,Dout,N", [(32, 64, 1000), (64, 64, 777), (64, 128, 512), (128, 128, 300), (128, 256, 512)])
def test_flex_conv_pm_vs_oracle(dev, oracle, Din, Dout, N):
    from dh3d_amd import pm
    rng = np.random.default_rng(Din + Dout)
    B, K = 2, 8
    xyz, nn = _cloud(rng, B, N, K, oracle)
    f = rng.standard_normal((B, N, Din)).astype(np.float32)
    theta = (rng.standard_normal((3, Din, Dout)) / np.sqrt(Din)).astype(np.float32)
    bias = (rng.standard_normal((Din, Dout)) / np.sqrt(8 * Din)).astype(np.float32)
    fb = rng.standard_normal(Dout).astype(np.float32)
    sc = (0.5 + rng.random(Dout)).astype(np.float32)
    sh = rng.standard_normal(Dout).astype(np.float32)
    wp = pm.pack_flex_weight(T(theta, dev), T(bias, dev))
    raw = pm.flex_conv(T(f, dev), T(xyz, dev), T(nn, dev), wp, Dout).cpu().numpy()
    exp = oracle.flex_convolution(f.transpose(0, 2, 1), xyz.transpose(0, 2, 1), nn.transpose(0, 2, 1), theta, bias,
                                  True).transpose(0, 2, 1)
    assert relerr(raw, exp) < 2e-6 and close(raw, exp, 1e-4, 1e-4 * np.abs(exp).max())
    fused = pm.flex_conv(T(f, dev), T(xyz, dev), T(nn, dev), wp, Dout, pre_bias=T(fb, dev), scale=T(sc, dev),
                         shift=T(sh, dev), act=pm.ACT_RELU).cpu().numpy()
    assert close(fused, np.maximum((exp + fb) * sc + sh, 0), 1e-4, 1e-4 * np.abs(exp).max())


@pytest.mark.parametrize("Din,Dout,B,N", [(32, 64, 2, 999), (64, 64, 2, 999), (64, 64, 1, 40), (32, 64, 1, 96),
                                          (32, 64, 3, 2048), (64, 64, 3, 2048)])
def test_flex_conv_x6_vs_oracle(dev, oracle, Din, Dout, B, N):
    """Persistent bf16x6 flex_conv: f32-accurate against the oracle (ragged tiles, tiles straddling clouds,
    one and several tiles per workgroup, odd tile counts)."""
    from dh3d_amd import pm
    rng = np.random.default_rng(Din + Dout + N)
    K = 8
    xyz, nn = _cloud(rng, B, N, K, oracle)
    f = rng.standard_normal((B, N, Din)).astype(np.float32)
    theta = (rng.standard_normal((3, Din, Dout)) / np.sqrt(Din)).astype(np.float32)
    bias = (rng.standard_normal((Din, Dout)) / np.sqrt(8 * Din)).astype(np.float32)
    fb = rng.standard_normal(Dout).astype(np.float32)
    sc = (0.5 + rng.random(Dout)).astype(np.float32)
    sh = rng.standard_normal(Dout).astype(np.float32)
    assert pm.flex_x6_supported(Din, Dout, K)
    wp3 = pm.pack_flex_weight_x3(T(theta, dev), T(bias, dev))
    raw = pm.flex_conv_x6(T(f, dev), T(xyz, dev), T(nn, dev), wp3, Dout).cpu().numpy()
    exp = oracle.flex_convolution(f.transpose(0, 2, 1), xyz.transpose(0, 2, 1), nn.transpose(0, 2, 1), theta, bias,
                                  True).transpose(0, 2, 1)
    assert relerr(raw, exp) < 2e-6 and close(raw, exp, 1e-4, 1e-4 * np.abs(exp).max())
    fused = pm.flex_conv_x6(T(f, dev), T(xyz, dev), T(nn, dev), wp3, Dout, pre_bias=T(fb, dev), scale=T(sc, dev),
                            shift=T(sh, dev), act=pm.ACT_RELU).cpu().numpy()
    assert close(fused, np.maximum((exp + fb) * sc + sh, 0), 1e-4, 1e-4 * np.abs(exp).max())


@pytest.mark.parametrize("Din,Dout,B,N", [(64, 128, 2, 512), (128, 128, 2, 300), (128, 256, 2, 512), (64, 128, 3, 45),
                                          (128, 128, 1, 17), (128, 256, 3, 1000)])
def test_flex_conv_tile_x6_vs_oracle(dev, oracle, Din, Dout, B, N):
    """32-point tiles with the tile GEMM on the bf16 pipe (the sampled levels): f32-accurate against the oracle -- ragged
    last tiles, tiles straddling clouds, clouds smaller than a tile -- with and without the fused epilogue."""
    from dh3d_amd import pm
    rng = np.random.default_rng(Din + Dout + N)
    K = 8
    xyz, nn = _cloud(rng, B, N, K, oracle)
    f = rng.standard_normal((B, N, Din)).astype(np.float32)
    theta = (rng.standard_normal((3, Din, Dout)) / np.sqrt(Din)).astype(np.float32)
    bias = (rng.standard_normal((Din, Dout)) / np.sqrt(8 * Din)).astype(np.float32)
    fb = rng.standard_normal(Dout).astype(np.float32)
    sc = (0.5 + rng.random(Dout)).astype(np.float32)
    sh = rng.standard_normal(Dout).astype(np.float32)
    assert pm.flex_tile_x6_supported(Din, Dout, K)
    wp3 = pm.pack_flex_weight_x3(T(theta, dev), T(bias, dev))
    raw = pm.flex_conv_tile_x6(T(f, dev), T(xyz, dev), T(nn, dev), wp3, Dout).cpu().numpy()
    exp = oracle.flex_convolution(f.transpose(0, 2, 1), xyz.transpose(0, 2, 1), nn.transpose(0, 2, 1), theta, bias,
                                  True).transpose(0, 2, 1)
    assert relerr(raw, exp) < 2e-6 and close(raw, exp, 1e-4, 1e-4 * np.abs(exp).max())
    fused = pm.flex_conv_tile_x6(T(f, dev), T(xyz, dev), T(nn, dev), wp3, Dout, pre_bias=T(fb, dev), scale=T(sc, dev),
                                 shift=T(sh, dev), act=pm.ACT_RELU).cpu().numpy()
    assert close(fused, np.maximum((exp + fb) * sc + sh, 0), 1e-4, 1e-4 * np.abs(exp).max())


def test_flex_conv_tile_x6_post_linear_and_k12(dev):
    """(a) the global step's 128 -> 256 with NetVLAD's cluster logits riding in the launch == the exact-f32 kernel's two
    outputs to rounding; (b) cfg 5's 128 -> 128, K = 12 at 16384 points == the exact-f32 kernel; both deterministic."""
    from dh3d_amd import pm
    g = torch.Generator().manual_seed(11)
    for B, N in ((4, 512), (20, 512)):  # 64 tiles; 320 tiles (more than CUs: the half-K planes, two workgroups per CU)
        _tile_x6_post_case(dev, g, B, N)
    B, N, K, Din, Dout = 10, 1024, 8, 128, 128  # 320 tiles, half-K planes
    xyz = torch.rand(B, N, 3, generator=g).to(dev)
    nbr, _ = pm.knn_xyz(xyz, K)
    f = torch.randn(B, N, Din, generator=g).to(dev)
    theta = (torch.randn(3, Din, Dout, generator=g) / Din ** 0.5).to(dev)
    bias = (torch.randn(Din, Dout, generator=g) / (K * Din) ** 0.5).to(dev)
    a = pm.flex_conv(f, xyz, nbr, pm.pack_flex_weight(theta, bias), Dout)
    b = pm.flex_conv_tile_x6(f, xyz, nbr, pm.pack_flex_weight_x3(theta, bias), Dout)
    assert (a - b).abs().max().item() / a.abs().max().item() < 2e-6
    # cfg 5 (512 tiles, half-K planes): both kernels against the float64 generic kernel -- two f32-accurate results may
    # differ from each other by the sum of their errors
    from dh3d_amd import ops
    B, N, K, Din, Dout = 1, 16384, 12, 128, 128
    xyz = torch.rand(B, N, 3, generator=g).to(dev)
    nbr, _ = pm.knn_xyz(xyz, K)
    f = torch.randn(B, N, Din, generator=g).to(dev)
    theta = (torch.randn(3, Din, Dout, generator=g) / Din ** 0.5).to(dev)
    bias = (torch.randn(Din, Dout, generator=g) / (K * Din) ** 0.5).to(dev)
    a = pm.flex_conv(f, xyz, nbr, pm.pack_flex_weight(theta, bias), Dout)
    b = pm.flex_conv_tile_x6(f, xyz, nbr, pm.pack_flex_weight_x3(theta, bias), Dout)
    ref = ops.flex_convolution(f.double().transpose(1, 2).contiguous(), xyz.double().transpose(1, 2).contiguous(),
                               nbr.transpose(1, 2).contiguous(), theta.double(), bias.double()).transpose(1, 2)
    scale = ref.abs().max().item()
    assert (b.double() - ref).abs().max().item() / scale < 2e-6
    assert (a.double() - ref).abs().max().item() / scale < 2e-6
    assert torch.equal(b, pm.flex_conv_tile_x6(f, xyz, nbr, pm.pack_flex_weight_x3(theta, bias), Dout))


def _tile_x6_post_case(dev, g, B, N):
    from dh3d_amd import pm
    K, Din, Dout = 8, 128, 256
    xyz = torch.rand(B, N, 3, generator=g).to(dev)
    nbr, _ = pm.knn_xyz(xyz, K)
    f = torch.randn(B, N, Din, generator=g).to(dev)
    theta = (torch.randn(3, Din, Dout, generator=g) / Din ** 0.5).to(dev)
    bias = (torch.randn(Din, Dout, generator=g) / (8 * Din) ** 0.5).to(dev)
    wpost = pm.pack_weight((torch.randn(Dout, 64, generator=g) / Dout ** 0.5).to(dev))
    fb, sc = torch.randn(Dout, generator=g).to(dev), (0.5 + torch.rand(Dout, generator=g)).to(dev)
    kw = dict(pre_bias=fb, scale=sc, shift=fb, act=pm.ACT_RELU)
    a, a2 = pm.flex_conv_post(f, xyz, nbr, pm.pack_flex_weight(theta, bias), Dout, wpost, 64, **kw)
    wp3 = pm.pack_flex_weight_x3(theta, bias)
    b, b2 = pm.flex_conv_tile_x6(f, xyz, nbr, wp3, Dout, wpost_packed=wpost, Dpost=64, **kw)
    c, c2 = pm.flex_conv_tile_x6(f, xyz, nbr, wp3, Dout, wpost_packed=wpost, Dpost=64, **kw)
    assert torch.equal(b, c) and torch.equal(b2, c2)
    assert (a - b).abs().max().item() / a.abs().max().item() < 2e-6
    assert (a2 - b2).abs().max().item() / a2.abs().max().item() < 4e-6


@pytest.mark.parametrize("Din", [32, 64])
def test_flex_conv_x6_full_size_matches_f32_kernel(dev, Din):
    """BASELINE shape (B=8, N=8192, K=8): eight tiles per workgroup; bf16x6 == exact-f32 MFMA kernel to rounding."""
    from dh3d_amd import pm
    g = torch.Generator().manual_seed(Din)
    B, N, K, Dout = 8, 8192, 8, 64
    xyz = torch.rand(B, N, 3, generator=g).to(dev)
    nbr, _ = pm.knn_xyz(xyz, K)
    f = torch.randn(B, N, Din, generator=g).to(dev)
    theta = (torch.randn(3, Din, Dout, generator=g) / Din ** 0.5).to(dev)
    bias = (torch.randn(Din, Dout, generator=g) / (8 * Din) ** 0.5).to(dev)
    a = pm.flex_conv(f, xyz, nbr, pm.pack_flex_weight(theta, bias), Dout)
    b = pm.flex_conv_x6(f, xyz, nbr, pm.pack_flex_weight_x3(theta, bias), Dout)
    b2 = pm.flex_conv_x6(f, xyz, nbr, pm.pack_flex_weight_x3(theta, bias), Dout)
    assert torch.equal(b, b2)  # deterministic
    err = (a - b).abs().max().item() / a.abs().max().item()
    assert err < 2e-6, err


def test_flex_conv_cfg5_shape_matches_reference_formulation(dev, oracle):
    """BASELINE config 5: one flex_conv 128 -> 128 at B=1, N=16384, K=12 (localdesc_extract.py:146,166).  The fused
    factorised kernel (run-time K) against (a) the drop-in kernel that keeps the reference's formulation and summation
    order -- with ops.FAST_PATH off, so that it is NOT the fused kernel behind two transposes -- and (b) the ORACLE on a
    strided sample of the 16384 rows (the C loop is 9*K*Din*Dout flop per point: the full launch would take a minute)."""
    from dh3d_amd import ops, pm
    g = torch.Generator().manual_seed(5)
    B, N, K, Din, Dout = 1, 16384, 12, 128, 128
    xyz = torch.rand(B, N, 3, generator=g).to(dev)
    nbr, _ = pm.knn_xyz(xyz, K)
    f = torch.randn(B, N, Din, generator=g).to(dev)
    theta = (torch.randn(3, Din, Dout, generator=g) / Din ** 0.5).to(dev)
    bias = (torch.randn(Din, Dout, generator=g) / (K * Din) ** 0.5).to(dev)
    fused = pm.flex_conv(f, xyz, nbr, pm.pack_flex_weight(theta, bias), Dout)
    fast = ops.FAST_PATH
    try:
        ops.FAST_PATH = False  # the reference-formulation kernel (csrc/flex_generic.hip), not the fused one re-wrapped
        ref = ops.flex_convolution(f.transpose(1, 2).contiguous(), xyz.transpose(1, 2).contiguous(),
                                   nbr.transpose(1, 2).contiguous(), theta, bias).transpose(1, 2)
    finally:
        ops.FAST_PATH = fast
    err = (fused - ref).abs().max().item() / ref.abs().max().item()
    assert err < 5e-6, err
    assert nbr.shape == (B, N, K) and bool((nbr[:, :, 0] == torch.arange(N, device=dev)).all())
    # (b) the oracle on every 97th row: gather those rows' neighbourhoods into a compact cloud of 169 * 12 points
    rows = np.arange(0, N, 97)
    nb = nbr[0].cpu().numpy()[rows]                                     # [R, K] ids into the full cloud
    ids = nb.reshape(-1)
    sub_f = f[0].cpu().numpy()[ids].T[None]                             # [1, Din, R*K]
    sub_p = xyz[0].cpu().numpy()[ids].T[None]
    R = len(rows)
    sub_nb = np.zeros((1, K, R * K), np.int32)                          # row r*K is query r (its rank-0 neighbour = itself)
    sub_nb[0, :, ::K] = (np.arange(R)[None, :] * K + np.arange(K)[:, None])
    exp = oracle.flex_convolution(np.ascontiguousarray(sub_f), np.ascontiguousarray(sub_p), sub_nb,
                                  theta.cpu().numpy(), bias.cpu().numpy(), center_self=True)[0][:, ::K].T   # [R, Dout]
    got = fused[0].cpu().numpy()[rows]
    assert np.abs(got - exp).max() <= 1e-5 * np.abs(exp).max() + 1e-4 * np.abs(exp).mean(), np.abs(got - exp).max()


def test_flex_pool_pm_exact(dev, oracle):
    from dh3d_amd import pm
    rng = np.random.default_rng(3)
    xyz, nn = _cloud(rng, 2, 999, 8, oracle)
    f = rng.standard_normal((2, 999, 64)).astype(np.float32)
    out, arg = pm.flex_pool(T(f, dev), T(nn, dev), want_argmax=True)
    eo, ea = oracle.flex_pooling(f.transpose(0, 2, 1), nn.transpose(0, 2, 1))
    assert np.array_equal(out.cpu().numpy(), eo.transpose(0, 2, 1))
    assert np.array_equal(arg.cpu().numpy(), ea.transpose(0, 2, 1))


def test_conv_pointset_pm_vs_oracle(dev, oracle):
    from dh3d_amd import pm
    rng = np.random.default_rng(4)
    xyz, nn = _cloud(rng, 2, 1234, 8, oracle)
    theta = rng.standard_normal((3, 32)).astype(np.float32)
    bias = rng.standard_normal(32).astype(np.float32)
    out = pm.conv_pointset_xyz(T(xyz, dev), T(nn, dev), T(theta, dev), T(bias, dev)).cpu().numpy()
    exp = oracle.convolution_pointset(xyz.transpose(0, 2, 1), nn.transpose(0, 2, 1), theta, bias).transpose(0, 2, 1)
    assert close(out, exp, 1e-4, 1e-5)


@pytest.mark.parametrize("B,N,scale", [(2, 1234, 1.0), (3, 8192, 40.0), (1, 64, 1.0)])
def test_conv_pointset_pool_fused_vs_oracle(dev, oracle, B, N, scale):
    """conv_pointset 3 -> 32 + BatchNorm (negative scales included: max and BN do not commute) + ReLU + flex_pool in the
    fused two-launch form against the oracle's two operators (conv_pointset_kernel.cc:46-64, flex_pool_kernel.cc:41-57),
    and against the two separate HIP kernels; duplicated points (rank-0 neighbour not the point itself) included."""
    from dh3d_amd import pm
    rng = np.random.default_rng(B * 1000 + N)
    xyz = (rng.random((B, N, 3), dtype=np.float32) * scale - scale / 2).astype(np.float32)
    xyz[:, N // 2] = xyz[:, N // 3]  # an exact duplicate in every cloud
    nn, _ = oracle.knn_bruteforce(np.ascontiguousarray(xyz.transpose(0, 2, 1)), 8)
    theta = (rng.standard_normal((3, 32)) * 4 / scale).astype(np.float32)
    bias = rng.standard_normal(32).astype(np.float32)
    sc = rng.standard_normal(32).astype(np.float32)  # both signs
    sh = (0.5 * rng.standard_normal(32)).astype(np.float32)
    conv = oracle.convolution_pointset(xyz.transpose(0, 2, 1), nn.transpose(0, 2, 1), theta, bias)       # [B,32,N]
    act = np.maximum(conv * sc[None, :, None] + sh[None, :, None], 0).astype(np.float32)
    exp, _ = oracle.flex_pooling(act, np.ascontiguousarray(nn.transpose(0, 2, 1)))
    exp = exp.transpose(0, 2, 1)
    args = (T(xyz, dev), T(nn, dev), T(theta, dev), T(bias, dev))
    kw = dict(scale=T(sc, dev), shift=T(sh, dev), act=pm.ACT_RELU)
    got = pm.conv_pointset_pool_xyz(*args, **kw).cpu().numpy()
    tol = 1e-5 * np.abs(conv).max()
    assert np.abs(got - exp).max() <= tol + 1e-4 * np.abs(exp).max(), np.abs(got - exp).max()
    two = pm.flex_pool(pm.conv_pointset_xyz(*args, **kw), T(nn, dev)).cpu().numpy()
    assert np.abs(got - two).max() <= tol, np.abs(got - two).max()


@pytest.mark.parametrize("C1,C2,Dout", [(64, 0, 64), (64, 0, 128), (128, 64, 128), (128, 0, 256), (256, 0, 192)])
def test_linear_pm_vs_fp64(dev, C1, C2, Dout):
    from dh3d_amd import pm
    rng = np.random.default_rng(C1 + Dout)
    R = 1000  # not a multiple of the 64-row tile
    x1 = rng.standard_normal((R, C1)).astype(np.float32)
    x2 = rng.standard_normal((R, C2)).astype(np.float32) if C2 else None
    W = (rng.standard_normal((C1 + C2, Dout)) / np.sqrt(C1 + C2)).astype(np.float32)
    b = rng.standard_normal(Dout).astype(np.float32)
    sc = (0.5 + rng.random(Dout)).astype(np.float32)
    sh = rng.standard_normal(Dout).astype(np.float32)
    res = rng.standard_normal((R, Dout)).astype(np.float32)
    out = pm.linear(T(x1, dev), pm.pack_weight(T(W, dev)), Dout, x2=T(x2, dev) if C2 else None, pre_bias=T(b, dev),
                    scale=T(sc, dev), shift=T(sh, dev), act=pm.ACT_RELU, residual=T(res, dev)).cpu().numpy()
    xin = np.concatenate([x1, x2], 1) if C2 else x1
    exp = np.maximum((xin.astype(np.float64) @ W + b) * sc + sh, 0) + res
    assert close(out, exp, 1e-4, 1e-5)


@pytest.mark.parametrize("C1,C2,Dout,R", [(128, 64, 128, 1000), (64, 0, 128, 4097), (128, 128, 256, 777), (256, 0, 256, 128)])
def test_linear_x6_vs_fp64(dev, C1, C2, Dout, R):
    """Tiled bf16x6 GEMM: f32-accurate against float64 (ragged row counts, concat input, both tile widths)."""
    from dh3d_amd import pm
    rng = np.random.default_rng(C1 + Dout + R)
    x1 = rng.standard_normal((R, C1)).astype(np.float32)
    x2 = rng.standard_normal((R, C2)).astype(np.float32) if C2 else None
    W = (rng.standard_normal((C1 + C2, Dout)) / np.sqrt(C1 + C2)).astype(np.float32)
    b = rng.standard_normal(Dout).astype(np.float32)
    sc = (0.5 + rng.random(Dout)).astype(np.float32)
    sh = rng.standard_normal(Dout).astype(np.float32)
    res = rng.standard_normal((R, Dout)).astype(np.float32)
    kw = dict(x2=T(x2, dev) if C2 else None, pre_bias=T(b, dev), scale=T(sc, dev), shift=T(sh, dev), act=pm.ACT_RELU,
              residual=T(res, dev))
    out = pm.linear_x6(T(x1, dev), pm.pack_weight_x3(T(W, dev)), Dout, **kw).cpu().numpy()
    ref32 = pm.linear(T(x1, dev), pm.pack_weight(T(W, dev)), Dout, **kw).cpu().numpy()
    xin = np.concatenate([x1, x2], 1) if C2 else x1
    exp = np.maximum((xin.astype(np.float64) @ W + b) * sc + sh, 0) + res
    assert close(out, exp, 1e-5, 2e-5)
    assert np.abs(out - exp).max() <= 2 * np.abs(ref32 - exp).max() + 1e-6  # as accurate as the exact-f32 pipe


def test_upsample_linear_x6_is_interpolate_then_linear(dev, oracle):
    """The fused up-sampling GEMM == three_interpolate (inverse-distance weights) followed by linear_x6, bit for bit."""
    from dh3d_amd import ops, pm
    g = torch.Generator().manual_seed(11)
    B, n, m, C1, C2, Dout = 2, 4096, 512, 128, 64, 128
    xyz = torch.rand(B, n, 3, generator=g).to(dev)
    xyz_s = xyz[:, ::8].contiguous()
    d3, i3 = ops.three_nn(xyz, xyz_s)
    coarse = torch.randn(B, m, C1, generator=g).to(dev)
    fine = torch.randn(B, n, C2, generator=g).to(dev)
    W = (torch.randn(C1 + C2, Dout, generator=g) / (C1 + C2) ** 0.5).to(dev)
    b = torch.randn(Dout, generator=g).to(dev)
    res = torch.randn(B, n, Dout, generator=g).to(dev)
    wp3 = pm.pack_weight_x3(W)
    up = pm.three_interpolate_idw(coarse, i3, d3)
    two = pm.linear_x6(up, wp3, Dout, x2=fine, pre_bias=b, act=pm.ACT_RELU, residual=res)
    one = pm.upsample_linear_x6(coarse, i3, d3, wp3, Dout, x2=fine, pre_bias=b, act=pm.ACT_RELU, residual=res)
    assert torch.equal(one, two)


@pytest.mark.parametrize("C", [64, 128])
def test_se_res_pm(dev, C):
    from dh3d_amd import pm
    rng = np.random.default_rng(C)
    R = 333
    x = rng.standard_normal((R, C)).astype(np.float32)
    pool = rng.standard_normal((R, C)).astype(np.float32)
    W1 = (rng.standard_normal((C, C // 4)) / 8).astype(np.float32); b1 = rng.standard_normal(C // 4).astype(np.float32)
    W2 = (rng.standard_normal((C // 4, C)) / 4).astype(np.float32); b2 = rng.standard_normal(C).astype(np.float32)
    out = pm.se_res(T(x, dev), T(pool, dev), T(W1, dev), T(b1, dev), T(W2, dev), T(b2, dev)).cpu().numpy()
    h = np.maximum(pool.astype(np.float64) @ W1 + b1, 0)
    g = 1 / (1 + np.exp(-(h @ W2 + b2)))
    assert close(out, np.maximum(x + x * g, 0), 1e-4, 1e-5)
    packed = pm.se_res_pack(T(W1, dev), T(b1, dev), T(W2, dev))
    out2 = pm.se_res_packed(T(x, dev), T(pool, dev), *packed, T(b2, dev)).cpu().numpy()
    assert close(out2, np.maximum(x + x * g, 0), 1e-4, 1e-5)


@pytest.mark.parametrize("C,B,N", [(64, 2, 1000), (128, 3, 333), (64, 8, 8192), (128, 8, 1024), (128, 5, 4100)])
def test_se_res_on_max_pool_fused_equals_two_kernels(dev, C, B, N):
    """SE-residual block on flex_pool(x) in one launch (pooled rows formed while staging) == flex_pool_pm then
    se_res_packed, bit for bit (a maximum is order-independent, the rest is the same code)."""
    from dh3d_amd import pm
    g = torch.Generator().manual_seed(C + N)
    x = torch.randn(B, N, C, generator=g).to(dev)
    xyz = torch.rand(B, N, 3, generator=g).to(dev)
    nbr, _ = pm.knn_xyz(xyz, 8)
    W1 = (torch.randn(C, C // 4, generator=g) / 8).to(dev); b1 = torch.randn(C // 4, generator=g).to(dev)
    W2 = (torch.randn(C // 4, C, generator=g) / 4).to(dev); b2 = torch.randn(C, generator=g).to(dev)
    packed = pm.se_res_pack(W1, b1, W2)
    two = pm.se_res_packed(x, pm.flex_pool(x, nbr), *packed, b2)
    one = pm.se_res_pool_packed(x, nbr, *packed, b2)
    assert torch.equal(one, two)


@pytest.mark.parametrize("B,N,C", [(3, 2000, 64), (3, 701, 128), (8, 1024, 128), (5, 4100, 128)])
def test_se_res_pool_conv_fused_equals_separate_kernels(dev, B, N, C):
    """... and with the following C -> C conv (+ BatchNorm + ReLU) in the same launch: the block's output bit for bit,
    the conv's output equal to linear() of it (same MFMA code on the same tile) -- 64-row tiles and, for C = 128 on fewer
    rows than 64-row tiles would fill the chip with, 32-row tiles."""
    from dh3d_amd import pm
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, N, C, generator=g).to(dev)
    nbr, _ = pm.knn_xyz(torch.rand(B, N, 3, generator=g).to(dev), 8)
    W1 = (torch.randn(C, C // 4, generator=g) / 8).to(dev); b1 = torch.randn(C // 4, generator=g).to(dev)
    W2 = (torch.randn(C // 4, C, generator=g) / 4).to(dev); b2 = torch.randn(C, generator=g).to(dev)
    Wc = (torch.randn(C, C, generator=g) / 8).to(dev); bc = torch.randn(C, generator=g).to(dev)
    sc = (0.5 + torch.rand(C, generator=g)).to(dev); sh = torch.randn(C, generator=g).to(dev)
    packed = pm.se_res_pack(W1, b1, W2)
    y_ref = pm.se_res_pool_packed(x, nbr, *packed, b2)
    z_ref = pm.linear(y_ref, pm.pack_weight(Wc), C, pre_bias=bc, scale=sc, shift=sh, act=pm.ACT_RELU)
    y, z = pm.se_res_pool_conv(x, nbr, *packed, b2, pm.pack_weight(Wc), bc, sc, sh)
    assert torch.equal(y, y_ref)
    assert torch.equal(z, z_ref)


@pytest.mark.parametrize("B,N", [(3, 2000), (8, 8192), (2, 77)])
def test_se_res_pool_conv_with_two_tails_equals_separate_kernels(dev, B, N):
    """Stage 1's SE block + before_stage2_conv1d + the shortcut conv (on the block's output) + the commuted concat conv's
    lower block (on the conv's output) in ONE launch == the fused SE + conv kernel followed by linear_x6 twice: the conv's
    output bit for bit, the tails to rounding of the same bf16x6 products (ragged last tile, clouds smaller than a tile)."""
    from dh3d_amd import pm
    g = torch.Generator().manual_seed(N)
    C = 64
    x = torch.randn(B, N, C, generator=g).to(dev)
    nbr, _ = pm.knn_xyz(torch.rand(B, N, 3, generator=g).to(dev), 8)
    W1 = (torch.randn(C, C // 4, generator=g) / 8).to(dev); b1 = torch.randn(C // 4, generator=g).to(dev)
    W2 = (torch.randn(C // 4, C, generator=g) / 4).to(dev); b2 = torch.randn(C, generator=g).to(dev)
    Wc = (torch.randn(C, C, generator=g) / 8).to(dev); bc = torch.randn(C, generator=g).to(dev)
    sc = (0.5 + torch.rand(C, generator=g)).to(dev); sh = torch.randn(C, generator=g).to(dev)
    Wa = (torch.randn(C, 128, generator=g) / 8).to(dev); ba = torch.randn(128, generator=g).to(dev)
    sa = (0.5 + torch.rand(128, generator=g)).to(dev); ha = torch.randn(128, generator=g).to(dev)
    Wb = (torch.randn(C, 128, generator=g) / 8).to(dev)
    packed = pm.se_res_pack(W1, b1, W2)
    y_ref, z_ref = pm.se_res_pool_conv(x, nbr, *packed, b2, pm.pack_weight(Wc), bc, sc, sh)
    a_ref = pm.linear_x6(y_ref, pm.pack_weight_x3(Wa), 128, pre_bias=ba, scale=sa, shift=ha, act=pm.ACT_RELU)
    b_ref = pm.linear_x6(z_ref, pm.pack_weight_x3(Wb), 128)
    ta = (pm.pack_weight_x3(Wa), ba, sa, ha, pm.ACT_RELU)
    tb = (pm.pack_weight_x3(Wb), None, None, None, pm.ACT_NONE)
    y, z, oa, ob = pm.se_res_pool_conv_tails(x, nbr, *packed, b2, pm.pack_weight(Wc), bc, sc, sh, ta, tb)
    assert torch.equal(y, y_ref) and torch.equal(z, z_ref)
    for got, ref in ((oa, a_ref), (ob, b_ref)):
        assert (got - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    y2, z2, oa2, ob2 = pm.se_res_pool_conv_tails(x, nbr, *packed, b2, pm.pack_weight(Wc), bc, sc, sh, ta, tb, store_y=False)
    assert y2 is None and torch.equal(z2, z) and torch.equal(oa2, oa) and torch.equal(ob2, ob)


def test_interpolate_idw_l2norm_and_head(dev, oracle):
    from dh3d_amd import ops, pm
    rng = np.random.default_rng(6)
    xyz = rng.random((2, 800, 3), dtype=np.float32)
    sub = xyz[:, ::8].copy()
    pts = rng.standard_normal((2, 100, 128)).astype(np.float32)
    d, i = ops.three_nn(T(xyz, dev), T(sub, dev))
    out = pm.three_interpolate_idw(T(pts, dev), i, d).cpu().numpy()
    ed, ei = oracle.three_nn(xyz, sub)
    dist = np.maximum(ed, np.float32(1e-10))
    w = (1.0 / dist) / np.sum(1.0 / dist, axis=2, keepdims=True)
    assert close(out, oracle.three_interpolate(pts, ei, w.astype(np.float32)), 1e-4, 1e-5)
    # l2norm + concat (core/model.py:177-181), incl. an all-zero row (eps rule) and a tiny row
    x = rng.standard_normal((500, 128)).astype(np.float32)
    x[3] = 0; x[4] *= 1e-6
    pre = rng.standard_normal((500, 3)).astype(np.float32)
    o = pm.l2norm_concat(T(x, dev), 1e-8, prefix=T(pre, dev)).cpu().numpy()
    exp = x / np.sqrt(np.maximum((x.astype(np.float64) ** 2).sum(1, keepdims=True), 1e-8))
    assert np.array_equal(o[:, :3], pre) and close(o[:, 3:], exp, 1e-4, 1e-6)
    # MLP head: 256 -> 1024 (BN, ReLU) -> 1 -> sigmoid
    R, C, H = 300, 256, 1024
    h = rng.standard_normal((R, C)).astype(np.float32)
    W = (rng.standard_normal((C, H)) / 16).astype(np.float32); b = rng.standard_normal(H).astype(np.float32)
    sc = (0.5 + rng.random(H)).astype(np.float32); sh = rng.standard_normal(H).astype(np.float32)
    wfc = (rng.standard_normal(H) / 32).astype(np.float32)
    att = pm.mlp_head(T(h, dev), pm.pack_weight(T(W, dev)), H, T(wfc, dev), 0.125, pre_bias=T(b, dev),
                      scale=T(sc, dev), shift=T(sh, dev)).cpu().numpy()
    hid = np.maximum((h.astype(np.float64) @ W + b) * sc + sh, 0)
    assert close(att[:, 0], 1 / (1 + np.exp(-(hid @ wfc + 0.125))), 1e-4, 1e-5)


@pytest.mark.parametrize("B,N", [(2, 512), (3, 1000), (32, 128)])
def test_netvlad_vs_restatement(dev, B, N):
    from oracle import model_np
    from dh3d_amd import pm
    rng = np.random.default_rng(B * N)
    D, C, O = 256, 64, 256
    x = rng.standard_normal((B, N, D)).astype(np.float32)
    att = rng.random((B, N, 1), dtype=np.float32)
    w = {"cluster_weights": (rng.standard_normal((D, C)) / 16).astype(np.float32),
         "cluster_weights2": (rng.standard_normal((1, D, C)) / 16).astype(np.float32),
         "hidden1_weights": (rng.standard_normal((D * C, O)) / 8).astype(np.float32),
         "gating_weights": (rng.standard_normal((O, O)) / 16).astype(np.float32)}
    for s, n in (("cluster_bn", C), ("bn", O), ("gating_bn", O)):
        w[s + "/gamma"] = (0.5 + rng.random(n)).astype(np.float32)
        w[s + "/beta"] = (0.1 * rng.standard_normal(n)).astype(np.float32)
        w[s + "/moving_mean"] = (0.1 * rng.standard_normal(n)).astype(np.float32)
        w[s + "/moving_variance"] = (0.5 + rng.random(n)).astype(np.float32)
    exp = model_np.global_netvlad_block(x, att, w, 1e-3)

    def fold(s):
        sc = w[s + "/gamma"] / np.sqrt(w[s + "/moving_variance"] + np.float32(1e-3))
        return T(sc, dev), T(w[s + "/beta"] - w[s + "/moving_mean"] * sc, dev)
    cs, ch = fold("cluster_bn"); s1, h1 = fold("bn"); s2, h2 = fold("gating_bn")
    vlad = pm.netvlad_aggregate(T(x, dev), T(att, dev), pm.pack_weight(T(w["cluster_weights"], dev)), cs, ch,
                                T(w["cluster_weights2"].reshape(D, C), dev))
    out = pm.netvlad_head(vlad, T(w["hidden1_weights"], dev), s1, h1, T(w["gating_weights"], dev), s2, h2).cpu().numpy()
    assert close(out, exp, 1e-4, 1e-5), relerr(out, exp)
    v = vlad.cpu().numpy()
    assert np.allclose(np.linalg.norm(v, axis=1), 1, atol=1e-5)  # whole-vector L2 norm
    out2 = pm.netvlad_head(vlad, T(w["hidden1_weights"], dev), s1, h1, T(w["gating_weights"], dev), s2, h2,
                           l2_eps=1e-8).cpu().numpy()
    assert close(out2, exp / np.linalg.norm(exp, axis=1, keepdims=True), 1e-4, 1e-5)


def test_mlp_head_bf16x6_is_f32_accurate(dev):
    """The bf16x6 head against float64 and against the f32-MFMA head: same answer to f32 rounding."""
    from dh3d_amd import pm
    rng = np.random.default_rng(77)
    R, C, H = 1000, 256, 1024
    h = (rng.standard_normal((R, C)) * np.exp(rng.standard_normal((R, 1)))).astype(np.float32)  # wide dynamic range
    W = (rng.standard_normal((C, H)) / 16).astype(np.float32)
    b = rng.standard_normal(H).astype(np.float32)
    sc = (0.5 + rng.random(H)).astype(np.float32); sh = rng.standard_normal(H).astype(np.float32)
    wfc = (rng.standard_normal(H) / 32).astype(np.float32)
    kw = dict(pre_bias=T(b, dev), scale=T(sc, dev), shift=T(sh, dev))
    a6 = pm.mlp_head_x6(T(h, dev), pm.pack_weight_x3(T(W, dev)), H, T(wfc, dev), 0.125, **kw).cpu().numpy()[:, 0]
    a1 = pm.mlp_head(T(h, dev), pm.pack_weight(T(W, dev)), H, T(wfc, dev), 0.125, **kw).cpu().numpy()[:, 0]
    hid = np.maximum((h.astype(np.float64) @ W + b) * sc + sh, 0)
    logit = hid @ wfc + 0.125
    exp = 1 / (1 + np.exp(-logit))
    e6, e1 = np.abs(a6 - exp).max(), np.abs(a1 - exp).max()
    assert e6 < 2e-6 and e6 < 4 * e1 + 1e-6, (e6, e1)
    # the pre-sigmoid sums themselves: compare a linear head (no activation, unit fc) at full relative precision
    ones = np.ones(H, np.float32)
    lin6 = pm.mlp_head_x6(T(h, dev), pm.pack_weight_x3(T(W, dev)), H, T(ones, dev), 0.0, act=pm.ACT_NONE).cpu().numpy()[:, 0]
    s = (h.astype(np.float64) @ W).sum(1)
    mag = (np.abs(h.astype(np.float64)) @ np.abs(W)).sum(1)
    z6 = np.log(lin6 / (1 - lin6))  # undo the sigmoid
    ok = np.abs(s) < 10
    assert np.all(np.abs(z6[ok] - s[ok]) <= 5e-7 * mag[ok] + 1e-5)


@pytest.mark.parametrize("B,n,l2", [(2, 4096, False), (1, 4100, True), (3, 4160, True)])
def test_upsample_linear_shortcut_fused_matches_two_kernels(dev, B, n, l2):
    """Concat conv + shortcut conv in one kernel == shortcut conv, then concat conv with the residual added in its
    store (same bf16x6 products, same epilogue arithmetic: equal to rounding of the final sum), with and without
    the fused l2-normalise + xyz concat."""
    from dh3d_amd import pm, ops
    g = torch.Generator().manual_seed(n)
    m, C1, C2, C3, Dout = n // 8, 128, 64, 64, 128
    fine = torch.rand(B, n, 3, generator=g).to(dev)
    coarse_xyz = fine[:, :m].contiguous()
    d3, i3 = ops.three_nn(fine, coarse_xyz)
    pts = torch.randn(B, m, C1, generator=g).to(dev)
    x2 = torch.randn(B, n, C2, generator=g).to(dev)
    x3 = torch.randn(B, n, C3, generator=g).to(dev)
    W = (torch.randn(C1 + C2, Dout, generator=g) / (C1 + C2) ** 0.5).to(dev)
    Ws = (torch.randn(C3, Dout, generator=g) / C3 ** 0.5).to(dev)
    mk = lambda: [torch.randn(Dout, generator=g).to(dev), (0.5 + torch.rand(Dout, generator=g)).to(dev),
                  torch.randn(Dout, generator=g).to(dev)]
    e1, e2 = mk(), mk()
    sc = pm.linear_x6(x3, pm.pack_weight_x3(Ws), Dout, pre_bias=e2[0], scale=e2[1], shift=e2[2], act=pm.ACT_RELU)
    l2cat = (fine, 1e-8) if l2 else None
    ref = pm.upsample_linear_x6(pts, i3, d3, pm.pack_weight_x3(W), Dout, x2=x2, pre_bias=e1[0], scale=e1[1],
                                shift=e1[2], act=pm.ACT_RELU, residual=sc, l2cat=l2cat)
    got = pm.upsample_linear_shortcut_x6(pts, i3, d3, pm.pack_weight_x3(torch.cat([W, Ws], 0).contiguous()), Dout, x2,
                                         x3, (e1[0], e1[1], e1[2], pm.ACT_RELU), (e2[0], e2[1], e2[2], pm.ACT_RELU),
                                         l2cat=l2cat)
    assert got.shape == ref.shape
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err < 1e-6, err


@pytest.mark.parametrize("B,n,m", [(2, 4096, 512), (1, 1001, 125), (3, 514, 64)])
def test_interp_combine_vs_fp64(dev, B, n, m):
    """The tail of the commuted concat conv (gather of the coarse product, + partial, epilogue, + residual, optional
    [xyz | l2_normalize] rows through LDS) against float64 -- also with an odd number of rows (the last wave holds one)."""
    from dh3d_amd import pm, ops
    g = torch.Generator().manual_seed(n + m)
    fine = torch.rand(B, n, 3, generator=g).to(dev)
    d3, i3 = ops.three_nn(fine, fine[:, :m].contiguous())
    cw = torch.randn(B, m, 128, generator=g).to(dev)
    part, res = torch.randn(B, n, 128, generator=g).to(dev), torch.randn(B, n, 128, generator=g).to(dev)
    pb, sc, sh = torch.randn(128, generator=g).to(dev), (0.5 + torch.rand(128, generator=g)).to(dev), torch.randn(128, generator=g).to(dev)
    w = 1.0 / d3.double().clamp_min(1e-10); w = w / w.sum(2, keepdim=True)
    up = (torch.gather(cw.double(), 1, i3.long().reshape(B, -1, 1).expand(-1, -1, 128)).reshape(B, n, 3, 128) * w[..., None]).sum(2)
    v = torch.relu((up + part.double() + pb.double()) * sc.double() + sh.double()) + res.double()
    got = pm.interp_combine(cw, i3, d3, partial=part, pre_bias=pb, scale=sc, shift=sh, act=pm.ACT_RELU, residual=res)
    assert got.shape == (B, n, 128) and float((got.double() - v).abs().max()) < 2e-5
    got = pm.interp_combine(cw, i3, d3, partial=part, pre_bias=pb, scale=sc, shift=sh, act=pm.ACT_RELU, residual=res,
                            l2cat=(fine, 1e-8))
    ref = torch.cat([fine.double(), v * torch.rsqrt(torch.clamp((v * v).sum(2, keepdim=True), min=1e-8))], 2)
    assert got.shape == (B, n, 131)
    assert torch.equal(got[:, :, :3], fine) and float((got.double() - ref).abs().max()) < 2e-6


@pytest.mark.parametrize("B,n", [(2, 4096), (1, 4100), (3, 5000)])
def test_interp_head_equals_head_on_upsampled_rows(dev, B, n):
    """The attention head with its wide conv commuted through the 3-NN interpolation == the head run on the
    materialised up-sampled tensor (same weights; a different association of the same sums)."""
    from dh3d_amd import pm, ops
    g = torch.Generator().manual_seed(n + B)
    m, C, Hd = n // 8, 256, 1024
    fine = torch.rand(B, n, 3, generator=g).to(dev)
    d3, i3 = ops.three_nn(fine, fine[:, :m].contiguous())
    coarse = torch.randn(B, m, C, generator=g).to(dev)
    W = (torch.randn(C, Hd, generator=g) / C ** 0.5).to(dev)
    wfc = (torch.randn(Hd, generator=g) / Hd ** 0.5).to(dev)
    b = torch.randn(Hd, generator=g).to(dev); sc = (0.5 + torch.rand(Hd, generator=g)).to(dev); sh = torch.randn(Hd, generator=g).to(dev)
    up = pm.three_interpolate_idw(coarse, i3, d3)
    ref = pm.mlp_head_x6(up, pm.pack_weight_x3(W), Hd, wfc, 0.2, pre_bias=b, scale=sc, shift=sh, act=pm.ACT_RELU)
    slices = torch.cat([pm.pack_weight_x3(W[:, j:j + 256].contiguous()) for j in range(0, Hd, 256)])
    got = pm.interp_head(coarse, i3, d3, slices, Hd, wfc, 0.2, pre_bias=b, scale=sc, shift=sh, act=pm.ACT_RELU)
    assert got.shape == ref.shape == (B, n, 1)
    assert (got - ref).abs().max().item() < 2e-6
    # and against float64
    w = 1.0 / d3.double().clamp_min(1e-10); w = w / w.sum(2, keepdim=True)
    upd = (torch.gather(coarse.double(), 1, i3.long().reshape(B, -1, 1).expand(-1, -1, C)).reshape(B, n, 3, C) * w[..., None]).sum(2)
    z = torch.relu((upd @ W.double() + b.double()) * sc.double() + sh.double()) @ wfc.double() + 0.2
    assert (got.double().squeeze(-1) - torch.sigmoid(z)).abs().max().item() < 2e-6


@pytest.mark.parametrize("B,n,clustered", [(2, 4096, False), (3, 5000, False), (9, 4096, False), (1, 8192, False), (2, 4100, True)])
def test_interp_head_lds_staged_equals_gather_kernel(dev, B, n, clustered):
    """interp_head with the fine points walked in Morton order and the distinct coarse rows staged in LDS (per
    256-channel slice) == the plain gather kernel: same per-element arithmetic, only the 1024-term row dot is summed in
    another order.  `clustered`: a cloud squeezed into a thin slab, where a block of 128 points can touch more distinct
    coarse rows than the 64 LDS slots hold (the overflow path reads them from global memory)."""
    from dh3d_amd import pm, ops
    g = torch.Generator().manual_seed(n + B)
    m, C, Hd = n // 8, 256, 1024
    fine = torch.rand(B, n, 3, generator=g)
    if clustered is True:
        fine[:, :, 2] *= 1e-3
        fine[:, :, 1] *= 0.05
    fine = fine.to(dev)
    samp = ops.farthest_point_sample(m, fine)
    coarse_xyz = torch.gather(fine, 1, samp.long()[:, :, None].expand(-1, -1, 3)).contiguous()
    d3, i3 = ops.three_nn(fine, coarse_xyz)
    coarse = torch.randn(B, m, C, generator=g).to(dev)
    W = (torch.randn(C, Hd, generator=g) / C ** 0.5).to(dev)
    wfc = (torch.randn(Hd, generator=g) / Hd ** 0.5).to(dev)
    b = torch.randn(Hd, generator=g).to(dev); sc = (0.5 + torch.rand(Hd, generator=g)).to(dev); sh = torch.randn(Hd, generator=g).to(dev)
    slices = torch.cat([pm.pack_weight_x3(W[:, j:j + 256].contiguous()) for j in range(0, Hd, 256)])
    kw = dict(pre_bias=b, scale=sc, shift=sh, act=pm.ACT_RELU)
    ref = pm.interp_head(coarse, i3, d3, slices, Hd, wfc, 0.2, **kw)
    srt, _ = pm.spatial_sort(fine)
    got = pm.interp_head(coarse, i3, d3, slices, Hd, wfc, 0.2, order=srt, **kw)
    assert got.shape == ref.shape == (B, n, 1)
    assert (got - ref).abs().max().item() < 1e-6, (got - ref).abs().max().item()


def test_flex_conv_with_fused_group_point_is_bit_identical(dev):
    """pm.flex_conv(full_map, ..., remap=idx) == pm.flex_conv(group_point(full_map, idx), ...): the sampled level's
    feature gather fused into the neighbour gather (compile-time K = 8 and run-time K paths, both tile sizes)."""
    from dh3d_amd import pm, ops
    from dh3d_amd.backbones import gather_rows
    g = torch.Generator().manual_seed(5)
    for (B, N, Din, Dout, K) in [(3, 4096, 64, 128, 8), (2, 2048, 128, 256, 8), (2, 1000, 128, 128, 12), (1, 8192, 64, 128, 8)]:
        M = N // 8
        xyz = torch.rand(B, N, 3, generator=g).to(dev)
        feat = torch.randn(B, N, Din, generator=g).to(dev)
        idx = ops.farthest_point_sample(M, xyz)
        xyz_s = gather_rows(xyz, idx)
        nbr_s, _ = pm.knn_xyz(xyz_s, K)
        wp = pm.pack_flex_weight((torch.randn(3, Din, Dout, generator=g) / Din ** 0.5).to(dev),
                                 (torch.randn(Din, Dout, generator=g) / (8 * Din) ** 0.5).to(dev))
        fb = torch.randn(Dout, generator=g).to(dev)
        kw = dict(pre_bias=fb, scale=fb * 0 + 1.5, shift=fb, act=pm.ACT_RELU)
        a = pm.flex_conv(gather_rows(feat, idx), xyz_s, nbr_s, wp, Dout, **kw)
        b = pm.flex_conv(feat, xyz_s, nbr_s, wp, Dout, remap=idx, **kw)
        assert torch.equal(a, b), (B, N, Din, Dout, K)


@pytest.mark.parametrize("B,N", [(2, 512), (32, 4096), (3, 1000)])
def test_netvlad_fused_equals_two_calls(dev, B, N):
    """dh3d_netvlad_fused_fwd (no separate whole-vector normalisation kernel) == aggregate + head."""
    from dh3d_amd import pm
    g = torch.Generator().manual_seed(B + N)
    D, C, O = 256, 64, 256
    x = torch.randn(B, N, D, generator=g).to(dev); att = torch.rand(B, N, 1, generator=g).to(dev)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    wc = pm.pack_weight((r(D, C) / 16).contiguous()); W2 = (r(D, C) / 16).contiguous()
    Wh, Wg = (r(D * C, O) / 8).contiguous(), (r(O, O) / 16).contiguous()
    cs, ch, s1, h1, s2, h2 = [0.5 + torch.rand(n, generator=g).to(dev) if i % 2 == 0 else 0.1 * r(n)
                              for i, n in enumerate((C, C, O, O, O, O))]
    for l2 in (0.0, 1e-8):
        two = pm.netvlad_head(pm.netvlad_aggregate(x, att, wc, cs, ch, W2), Wh, s1, h1, Wg, s2, h2, l2_eps=l2)
        one = pm.netvlad_fused(x, att, wc, cs, ch, W2, Wh, s1, h1, Wg, s2, h2, l2_eps=l2)
        assert float((one - two).abs().max()) <= 2e-6 * float(two.abs().max()), float((one - two).abs().max())
    nog = pm.netvlad_fused(x, att, wc, cs, ch, W2, Wh, s1, h1, None, None, None)
    ref = pm.netvlad_head(pm.netvlad_aggregate(x, att, wc, cs, ch, W2), Wh, s1, h1, None, None, None)
    assert float((nog - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


@pytest.mark.parametrize("B,n,clustered", [(2, 4096, False), (9, 4096, False), (3, 5000, False), (1, 8192, False), (2, 4100, True),
                                           (2, 4096, "scrambled")])
def test_global_tail_equals_upsample_attention_netvlad(dev, B, n, clustered):
    """pm.global_tail (one walk over the fine points, everything else on the coarse rows) == three_interpolate ->
    attention head -> NetVLAD + gating on the materialised up-sampled map; same math reassociated (f32 atomics: the
    result is reproducible to ~1e-7, compared at 2e-6 of the descriptor's largest entry), attention weights too."""
    from dh3d_amd import pm, ops
    g = torch.Generator().manual_seed(n * 3 + B)
    m, C, Hd, Cl, O = n // 8, 256, 1024, 64, 256
    fine = torch.rand(B, n, 3, generator=g)
    if clustered:
        fine[:, :, 2] *= 1e-3
        fine[:, :, 1] *= 0.05
    fine = fine.to(dev)
    samp = ops.farthest_point_sample(m, fine)
    cxyz = torch.gather(fine, 1, samp.long()[:, :, None].expand(-1, -1, 3)).contiguous()
    d3, i3 = ops.three_nn(fine, cxyz)
    if clustered == "scrambled":  # neighbours with no spatial coherence: a 128-point block touches ~350 coarse rows,
        # far more than the 64 staged in LDS -- every block takes the overflow path (rows read from / added to memory)
        i3 = torch.randint(0, m, (B, n, 3), generator=g, dtype=torch.int32).to(dev)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    coarse = r(B, m, C)
    W = (r(C, Hd) / C ** 0.5).contiguous(); wfc = r(Hd) / Hd ** 0.5
    b, sc, sh = r(Hd), (0.5 + torch.rand(Hd, generator=g)).to(dev), r(Hd)
    slices = torch.cat([pm.pack_weight_x3(W[:, j:j + 256].contiguous()) for j in range(0, Hd, 256)])
    Wc = (r(C, Cl) / 16).contiguous(); wc = pm.pack_weight(Wc); W2 = (r(C, Cl) / 16).contiguous()
    Wh, Wg = (r(C * Cl, O) / 8).contiguous(), (r(O, O) / 16).contiguous()
    cs, ch = (0.5 + torch.rand(Cl, generator=g)).to(dev), 0.1 * r(Cl)
    s1, h1, s2, h2 = (0.5 + torch.rand(O, generator=g)).to(dev), 0.1 * r(O), (0.5 + torch.rand(O, generator=g)).to(dev), 0.1 * r(O)
    srt, _ = pm.spatial_sort(fine)
    # reference: materialised up-sampling
    up = pm.three_interpolate_idw(coarse, i3, d3)
    att_ref = pm.interp_head(coarse, i3, d3, slices, Hd, wfc, 0.2, pre_bias=b, scale=sc, shift=sh, act=pm.ACT_RELU)
    ref = pm.netvlad_fused(up, att_ref, wc, cs, ch, W2, Wh, s1, h1, Wg, s2, h2, l2_eps=1e-8)
    got, att = pm.global_tail(coarse, i3, d3, srt, slices, Hd, wfc, 0.2, (b, sc, sh, pm.ACT_RELU), wc, cs, ch, W2, Wh, s1,
                              h1, Wg, s2, h2, l2_eps=1e-8, want_att=True)
    assert float((att - att_ref).abs().max()) < 1e-6
    assert float((got - ref).abs().max()) <= 2e-6 * float(ref.abs().max()) + 1e-7, float((got - ref).abs().max())
    again = pm.global_tail(coarse, i3, d3, srt, slices, Hd, wfc, 0.2, (b, sc, sh, pm.ACT_RELU), wc, cs, ch, W2, Wh, s1, h1,
                           Wg, s2, h2, l2_eps=1e-8)
    assert float((again - got).abs().max()) <= 1e-6 * float(got.abs().max())
    # round 6: the walk's slot tables built ahead of it (pm.walk_plan, behind three_nn in the model): same tables, same walk
    plan = pm.walk_plan(i3, d3, srt, m)
    planned, att_p = pm.global_tail(coarse, i3, d3, srt, slices, Hd, wfc, 0.2, (b, sc, sh, pm.ACT_RELU), wc, cs, ch, W2, Wh,
                                    s1, h1, Wg, s2, h2, l2_eps=1e-8, want_att=True, plan=plan)
    assert torch.equal(att_p, att)                       # no atomics on the attention path: bit-equal
    assert float((planned - got).abs().max()) <= 1e-6 * float(got.abs().max())
    with pytest.raises(ValueError):
        pm.global_tail(coarse, i3, d3, srt, slices, Hd, wfc, 0.2, (b, sc, sh, pm.ACT_RELU), wc, cs, ch, W2, Wh, s1, h1, Wg, s2,
                       h2, plan=plan[:-4])


def test_local_tail_fused_vs_float64_and_vs_the_three_launch_form(dev):
    """csrc/dense_tail.hip: [xyz | l2norm(relu(BN_c(interp3(cw) + x2 Wl + b_c)) + relu(BN_s(x1 Ws + b_s)))] in one launch
    against a float64 restatement and against the three launches it replaces (two 64 -> 128 GEMMs + interp_combine)."""
    from dh3d_amd import pm
    rng = np.random.default_rng(4242)
    B, N, M = 2, 4128, 516          # (N % 32 == 0, not a multiple of the 8-wave workgroup's 256 rows)
    x1 = rng.standard_normal((B, N, 64)).astype(np.float32)
    x2 = rng.standard_normal((B, N, 64)).astype(np.float32)
    cw = rng.standard_normal((B, M, 128)).astype(np.float32)
    Ws = (rng.standard_normal((64, 128)) / 8).astype(np.float32)
    Wl = (rng.standard_normal((64, 128)) / 8).astype(np.float32)
    idx = rng.integers(0, M, (B, N, 3)).astype(np.int32)
    dist = (rng.random((B, N, 3)) * 0.01 + 1e-4).astype(np.float32)
    dist[0, :5, 0] = 0.0             # coincident sample: weight 1e10 / (1e10 + ...)
    xyz = rng.random((B, N, 3)).astype(np.float32)
    eps = [(0.1 * rng.standard_normal(128)).astype(np.float32), (0.5 + rng.random(128)).astype(np.float32),
           (0.1 * rng.standard_normal(128)).astype(np.float32)]
    epc = [(0.1 * rng.standard_normal(128)).astype(np.float32), (0.5 + rng.random(128)).astype(np.float32),
           (0.1 * rng.standard_normal(128)).astype(np.float32)]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    got = pm.local_tail_fused(T(x1), T(x2), pm.pack_weight_x3(T(Ws)), pm.pack_weight_x3(T(Wl)), tuple(map(T, eps)),
                              tuple(map(T, epc)), T(cw), T(idx), T(dist), T(xyz), 1e-8).cpu().numpy()
    # float64 restatement
    d = np.maximum(dist.astype(np.float64), 1e-10)
    w = (1.0 / d) / (1.0 / d).sum(-1, keepdims=True)
    up = sum(w[..., t:t + 1] * np.take_along_axis(cw.astype(np.float64), idx[..., t:t + 1].astype(np.int64), 1) for t in range(3))
    yc = np.maximum((up + x2.astype(np.float64) @ Wl + epc[0]) * epc[1] + epc[2], 0)
    ys = np.maximum((x1.astype(np.float64) @ Ws + eps[0]) * eps[1] + eps[2], 0)
    y = yc + ys
    exp = np.concatenate([xyz, y / np.sqrt(np.maximum((y * y).sum(-1, keepdims=True), 1e-8))], -1)
    assert got.shape == (B, N, 131) and np.array_equal(got[..., :3], xyz)
    assert np.abs(got - exp).max() < 2e-6, np.abs(got - exp).max()
    # the three-launch form
    sc = pm.linear_x6(T(x1), pm.pack_weight_x3(T(Ws)), 128, pre_bias=T(eps[0]), scale=T(eps[1]), shift=T(eps[2]), act=pm.ACT_RELU)
    lo = pm.linear_x6(T(x2), pm.pack_weight_x3(T(Wl)), 128)
    ref = pm.interp_combine(T(cw), T(idx), T(dist), lo, pre_bias=T(epc[0]), scale=T(epc[1]), shift=T(epc[2]),
                            act=pm.ACT_RELU, residual=sc, l2cat=(T(xyz), 1e-8)).cpu().numpy()
    assert np.abs(got - ref).max() < 2e-6, np.abs(got - ref).max()
    # no prefix: the plain sum (what the global path consumes)
    plain = pm.local_tail_fused(T(x1), T(x2), pm.pack_weight_x3(T(Ws)), pm.pack_weight_x3(T(Wl)), tuple(map(T, eps)),
                                tuple(map(T, epc)), T(cw), T(idx), T(dist), None, 0.0).cpu().numpy()
    assert plain.shape == (B, N, 128)
    assert np.abs(plain - y).max() <= 2e-6 * np.abs(y).max(), np.abs(plain - y).max()
