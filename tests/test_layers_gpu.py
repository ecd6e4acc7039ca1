"""GPU: the Layer API (dh3d_amd/layers.py, mirrors core/layers.py:49-707) -- initconv + stage 1 of the local backbone
assembled from the Layer classes in the reference's channels-first convention (as core/backbones.py:104-116 with
core/tf_utils.py:48-83 does) must reproduce the fused point-major path, and be differentiable."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bn(x, bn):  # inference BatchNorm over axis 1 (data_format NCHW, core/tf_utils.py:61)
    sh = (1, -1, 1)
    return (x - bn.mean_EMA.reshape(sh)) * torch.rsqrt(bn.variance_EMA.reshape(sh) + bn.eps) * bn.gamma.reshape(sh) \
        + bn.beta.reshape(sh)


def test_stage1_from_layer_classes_matches_fused_path(dev):
    from dh3d_amd import ConfigFactory, layers, pm
    from dh3d_amd.model import DH3D
    m = DH3D(ConfigFactory("basic_config").getconfig()).init_synthetic(9)
    g = torch.Generator().manual_seed(10)
    with torch.no_grad():
        for name, buf in m.named_buffers():
            if name.endswith("mean_EMA"):
                buf.copy_(0.1 * torch.randn(buf.shape, generator=g))
            elif name.endswith("variance_EMA"):
                buf.copy_(0.5 + torch.rand(buf.shape, generator=g))
    m = m.to(dev).eval().prepare()
    pts = torch.rand(2, 1500, 3, generator=g).to(dev)

    # ---- fused point-major path
    with torch.no_grad():
        geo = m._geometry(pts)
        m._join_side(geo)
        p = m._local._prep
        init = pm.conv_pointset_xyz(geo.xyz, geo.nbr, p["theta"], p["bias"], scale=p["scale"], shift=p["shift"],
                                    act=pm.ACT_RELU)
        init = pm.flex_pool(init, geo.nbr)
        x1 = m.stage1(geo, init, nbr=geo.nbr)                                  # [B,N,64]

    # ---- the same graph from the Layer classes, channels-first
    pts_T = pts.transpose(1, 2).contiguous()
    nn_, _ = layers.KnnBruteforce(8)(pts_T)                                     # [B,K,N]
    assert torch.equal(nn_.transpose(1, 2), geo.nbr)
    nn_e, _ = layers.knn_bruteforce(pts_T.unsqueeze(2), 8, data_format="expanded")
    assert torch.equal(nn_e.squeeze(2), nn_)
    conv0 = layers.ConvolutionPointset(3, 32).to(dev)
    fc0, fc1 = layers.FlexConvolution(32, 64).to(dev), layers.FlexConvolution(64, 64).to(dev)
    with torch.no_grad():
        conv0.position_theta.copy_(m.initconv.position_theta); conv0.position_bias.copy_(m.initconv.position_bias)
        for lay, src in ((fc0, m.stage1.flexconv_0), (fc1, m.stage1.flexconv_1)):
            lay.position_theta.copy_(src.position_theta); lay.position_bias.copy_(src.position_bias)
            lay.feature_bias.copy_(src.feature_bias)
    feats = pts_T.clone().requires_grad_(True)
    x = torch.relu(_bn(conv0(feats, nn_), m.initconv_bn))                      # tf_utils.py:67-83
    x = layers.flex_pooling(x, nn_)
    x = torch.relu(_bn(fc0(x, pts_T, nn_), m.stage1.flexconv_0_bn))            # tf_utils.py:48-64
    x = torch.relu(_bn(fc1(x, pts_T, nn_), m.stage1.flexconv_1_bn))
    pool = layers.FlexPooling()(x, nn_)
    se = m.stage1.se                                                            # backbones.py:45-55
    W1 = se.f1.tfconv0.W.reshape(64, 16); W2 = se.f2.tfconv0.W.reshape(16, 64)
    sq = torch.relu(pool.transpose(1, 2) @ W1 + se.f1.tfconv0.b)
    sq = torch.sigmoid(sq @ W2 + se.f2.tfconv0.b).transpose(1, 2)
    y = torch.relu(x + x * sq)                                                  # [B,64,N]
    ref = y.transpose(1, 2)
    scale = float(ref.abs().max())
    assert float((ref - x1).abs().max()) <= 1e-4 * scale, float((ref - x1).abs().max()) / scale
    # differentiable through the registered gradients (FlexConvGrad / FlexPoolGrad / ConvPointsetGrad)
    y.sum().backward()
    for t in (feats.grad, fc0.position_theta.grad, fc1.position_bias.grad, conv0.position_theta.grad):
        assert t is not None and torch.isfinite(t).all() and float(t.abs().sum()) > 0


def test_flex_avg_layer_vs_oracle_flex_conv_with_zero_theta(dev, oracle):
    """Flex_Avg (core/layers.py:342-436) = the reference's FlexConv functor with theta = 0 and bias = eye(Dout): the
    layer's neighbour-sum kernel against exactly that call of the oracle (flex_conv_kernel.cc:48-63); a non-zero theta
    (a loaded variable) takes the flex_conv operator and matches the oracle too; gradients flow to the features."""
    import torch
    from dh3d_amd.layers import Flex_Avg, flex_avg
    rng = np.random.default_rng(8)
    B, N, C, K = 2, 777, 32, 8
    xyz = rng.random((B, N, 3), dtype=np.float32)
    nn, _ = oracle.knn_bruteforce(np.ascontiguousarray(xyz.transpose(0, 2, 1)), K)
    f = rng.standard_normal((B, C, N)).astype(np.float32)
    p_cf = np.ascontiguousarray(xyz.transpose(0, 2, 1))
    nb_cf = np.ascontiguousarray(nn.transpose(0, 2, 1))
    exp = oracle.flex_convolution(f, p_cf, nb_cf, np.zeros((3, C, C), np.float32), np.eye(C, dtype=np.float32), True)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    layer = Flex_Avg(C, C).to(dev)
    assert [n for n, p in layer.named_parameters() if p.requires_grad] == []  # position_theta is not trainable (:384)
    got = layer(T(f), T(p_cf), T(nb_cf))
    assert got.shape == (B, C, N) and np.array_equal(got.cpu().numpy(), exp)  # a sum in neighbour order: exact
    assert np.array_equal(flex_avg(T(f), T(p_cf), T(nb_cf), C).cpu().numpy(), exp)
    exp4 = layer.__class__(C, C, data_format="expanded").to(dev)(T(f).unsqueeze(2), T(p_cf).unsqueeze(2), T(nb_cf).unsqueeze(2))
    assert exp4.shape == (B, C, 1, N) and torch.equal(exp4.squeeze(2), got)
    # gradients: d(sum out)/d f[c, j] = number of neighbourhoods j appears in
    fr = T(f).requires_grad_(True)
    layer(fr, T(p_cf), T(nb_cf)).sum().backward()
    cnt = np.stack([np.bincount(nn[b].reshape(-1), minlength=N) for b in range(B)]).astype(np.float32)
    assert np.allclose(fr.grad.cpu().numpy(), np.broadcast_to(cnt[:, None, :], (B, C, N)), atol=1e-4)
    # a loaded, non-zero theta
    theta = (0.1 * rng.standard_normal((3, C, C))).astype(np.float32)
    with torch.no_grad():
        layer.position_theta.copy_(T(theta))
    exp2 = oracle.flex_convolution(f, p_cf, nb_cf, theta, np.eye(C, dtype=np.float32), True)
    got2 = layer(T(f), T(p_cf), T(nb_cf)).cpu().numpy()
    assert np.abs(got2 - exp2).max() <= 1e-4 * np.abs(exp2).max()
    # a write through .data (what loaders do) changes neither data_ptr nor _version: the layer must still notice
    layer2 = Flex_Avg(C, C).to(dev)
    assert np.array_equal(layer2(T(f), T(p_cf), T(nb_cf)).cpu().numpy(), exp)
    layer2.position_theta.data.copy_(T(theta))
    got3 = layer2(T(f), T(p_cf), T(nb_cf)).cpu().numpy()
    assert np.abs(got3 - exp2).max() <= 1e-4 * np.abs(exp2).max()
    with pytest.raises(ValueError):
        Flex_Avg(32, 64)
