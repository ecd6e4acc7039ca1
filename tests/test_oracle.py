"""CPU: the oracle (oracle/dh3d_oracle.c) against every pin the reference offers."""
import os

import numpy as np
import pytest
from scipy.spatial.distance import pdist, squareform

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(G, name)))


def close(a, b, rtol=1e-4, atol=1e-6):  # the reference's own criterion, user_ops/misc.py:89-97
    return np.allclose(a, b, rtol=rtol, atol=atol)


def test_knn_vs_scipy_reference_recipe(oracle):
    """user_ops/test_knn_bruteforce.py:32-56 on the FakePointCloud(2,32,4,2,6,3) fixture, k=4."""
    c = load("fake_pointcloud.npz")
    nn, d = oracle.knn_bruteforce(c["position"], 4)
    assert np.array_equal(nn, c["knn_scipy_ids"])
    assert close(d, c["knn_scipy_dist"], 1e-6, 1e-6)
    # the neighbourhood the reference fixture feeds to the flex ops is the same thing, [B,K,N]
    assert np.array_equal(nn.transpose(0, 2, 1), c["neighborhood"])


def test_knn_n4_whole_cloud_sort(oracle):
    """The reference test's effective case: N=4 and k=4 (test_knn_bruteforce.py:29,49-52)."""
    np.random.seed(42)
    pos = np.random.randn(1, 3, 4).astype(np.float32)
    nn, d = oracle.knn_bruteforce(pos, 4)
    D = squareform(pdist(pos[0].T.astype(np.float64)))
    assert np.array_equal(nn[0], np.argsort(D, axis=1))
    assert close(d[0], np.sort(D, axis=1), 1e-6, 1e-6)


def test_knn_tie_rule_is_cub_blocked_rank(oracle):
    """Equal distances break by (id % C_THREADS, id / C_THREADS) with C_THREADS from the N ladder."""
    assert oracle.knn_ladder(300) == (128, 4) and oracle.knn_ladder(8192) == (1024, 8)
    N = 300
    pos = np.zeros((1, 3, N), np.float32)  # all points coincide: pure tie-break order
    nn, d = oracle.knn_bruteforce(pos, 8)
    ct, cv = oracle.knn_ladder(N)
    order = sorted(range(N), key=lambda x: (x % ct) * cv + x // ct)[:8]
    assert np.array_equal(nn[0, 0], order) and np.array_equal(nn[0, 17], order)
    assert np.all(d == 0)
    # K > N pads with id -1 / FLT_MAX (knn_bruteforce_kernel_gpu.cu.cc:110-111)
    nn, d = oracle.knn_bruteforce(np.random.rand(1, 3, 3).astype(np.float32), 4)
    assert np.all(nn[:, :, 3] == -1) and np.all(d[:, :, 3] == np.finfo(np.float32).max)


def test_knn_golden_ties_frozen(oracle):
    c = load("knn_ties.npz")
    for name in ("lat300", "lat1100"):
        nn, d = oracle.knn_bruteforce(c[name + "_pos"], 8)
        assert np.array_equal(nn, c[name + "_nn"]) and np.array_equal(d, c[name + "_dist"])


def test_flex_pool_known_answer(oracle):
    """user_ops/test_flex_pooling.py:76-98: x=[1,2,5,3], cyclic neighbourhoods -> grad 4 on index 2."""
    c = load("flex_pool_kat.npz")
    out, arg = oracle.flex_pooling(c["x"], c["nbr"])
    assert np.array_equal(out, c["out"]) and np.array_equal(arg, c["argmax"])
    g = oracle.flex_pooling_grad(np.ones_like(out), arg)
    assert np.array_equal(g, c["grad"])
    g[0, 0, 2] -= 4
    assert g.sum() == 0


def test_flex_pool_is_first_max_global_id(oracle):
    c = load("fake_pointcloud.npz")
    out, arg = oracle.flex_pooling(c["features"], c["neighborhood"])
    f, nb = c["features"], c["neighborhood"]
    gathered = np.stack([np.take_along_axis(f, np.repeat(nb[:, k:k + 1, :], f.shape[1], 1), 2) for k in range(4)], 0)
    assert np.array_equal(out, gathered.max(0))
    first = gathered.argmax(0)  # first max in neighbour order
    assert np.array_equal(arg, np.take_along_axis(nb, first, 1)[:, :f.shape[1]] if False else
                          np.stack([nb[b][first[b], np.arange(f.shape[2])[None, :]] for b in range(2)]))
    assert np.array_equal(out, c["flex_pool"]) and np.array_equal(arg, c["flex_pool_argmax"])


def _flex_conv_f64(c):
    f, p, nb, th, bi = [c[k].astype(np.float64) if c[k].dtype != np.int32 else c[k]
                        for k in ("features", "position", "neighborhood", "theta", "bias")]
    B, Din, N = f.shape
    out = np.zeros((B, th.shape[2], N))
    for b in range(B):
        for n in range(N):
            for k in nb[b, :, n]:
                delta = p[b, :, k] - p[b, :, n]
                W = bi + np.einsum("d,dio->io", delta, th)
                out[b, :, n] += f[b, :, k] @ W
    return out


def test_flex_conv_closed_form_and_factorisation(oracle):
    c = load("fake_pointcloud.npz")
    out = oracle.flex_convolution(c["features"], c["position"], c["neighborhood"], c["theta"], c["bias"], True)
    assert close(out, _flex_conv_f64(c))
    # rank-0 neighbour is the point itself on duplicate-free data: CPU and GPU centre rules coincide
    out0 = oracle.flex_convolution(c["features"], c["position"], c["neighborhood"], c["theta"], c["bias"], False)
    assert np.array_equal(out, out0)
    # the factorised form the HIP kernel uses: out = [S0|Sx|Sy|Sz] @ [bias; theta]
    f, p, nb = c["features"].astype(np.float64), c["position"].astype(np.float64), c["neighborhood"]
    B, Din, N = f.shape
    S = np.zeros((B, N, 4, Din))
    for b in range(B):
        for n in range(N):
            for k in nb[b, :, n]:
                S[b, n, 0] += f[b, :, k]
                for d in range(3):
                    S[b, n, 1 + d] += (p[b, d, k] - p[b, d, n]) * f[b, :, k]
    Wcat = np.concatenate([c["bias"][None], c["theta"]], 0).astype(np.float64)  # [4,Din,Dout]
    fac = np.einsum("bnci,cio->bon", S, Wcat)
    assert close(out, fac)
    assert np.array_equal(out, c["flex_conv"])


def _numeric_grad(fn, x, top, eps=1e-2):
    g = np.zeros_like(x, dtype=np.float64)
    it = np.nditer(x, flags=["multi_index"])
    while not it.finished:
        i = it.multi_index
        xp, xm = x.copy(), x.copy()
        xp[i] += eps; xm[i] -= eps
        g[i] = ((fn(xp).astype(np.float64) - fn(xm).astype(np.float64)) * top).sum() / (2 * eps)
        it.iternext()
    return g


def test_flex_conv_gradients_numeric(oracle):
    """The reference's own style of check (test_flex_convolution.py:93-115): analytic vs numeric."""
    c = load("fake_pointcloud.npz")
    top = c["topdiff"]
    gf, gt, gb = oracle.flex_convolution_grad(c["features"], c["position"], c["neighborhood"], c["theta"], c["bias"], top)
    fc = lambda f=None, t=None, b=None: oracle.flex_convolution(
        c["features"] if f is None else f, c["position"], c["neighborhood"], c["theta"] if t is None else t,
        c["bias"] if b is None else b, False)
    assert np.allclose(gt, _numeric_grad(lambda t: fc(t=t), c["theta"], top), rtol=2e-2, atol=2e-2)
    assert np.allclose(gb, _numeric_grad(lambda b: fc(b=b), c["bias"], top), rtol=2e-2, atol=2e-2)
    assert np.allclose(gf, _numeric_grad(lambda f: fc(f=f), c["features"], top), rtol=2e-2, atol=2e-2)
    assert np.array_equal(gf, c["flex_conv_gf"]) and np.array_equal(gt, c["flex_conv_gt"])


def test_conv_pointset_closed_form_and_gradients(oracle):
    c = load("fake_pointcloud.npz")
    f, nb, th, bi = c["features"], c["neighborhood"], c["theta_rel"], c["bias_rel"]
    out = oracle.convolution_pointset(f, nb, th, bi)
    f64 = f.astype(np.float64)
    D = sum(np.take_along_axis(f64, np.repeat(nb[:, k:k + 1, :], 2, 1), 2) -
            np.take_along_axis(f64, np.repeat(nb[:, 0:1, :], 2, 1), 2) for k in range(4))
    ref = np.einsum("io,bin->bon", th.astype(np.float64), D) + bi[None, :, None]
    assert close(out, ref) and np.array_equal(out, c["conv_pointset"])
    top = c["topdiff"]
    gf, gt, gb = oracle.convolution_pointset_grad(f, nb, th, top)
    assert np.allclose(gt, _numeric_grad(lambda t: oracle.convolution_pointset(f, nb, t, bi), th, top), rtol=2e-2, atol=2e-2)
    assert np.allclose(gf, _numeric_grad(lambda x: oracle.convolution_pointset(x, nb, th, bi), f, top), rtol=2e-2, atol=2e-2)
    assert close(gb, top.sum((0, 2)), 1e-5, 1e-5)


def test_pointnet2_ops_vs_reference_twin_outputs(oracle):
    """Golden outputs produced by the reference's own compiled twins (tests/golden/make_golden.py)."""
    c = load("twins.npz")
    assert np.array_equal(oracle.three_interpolate(c["points"], c["idx3"], c["w3"]), c["interp"])
    assert np.array_equal(oracle.three_interpolate_grad(c["points"].shape, c["idx3"], c["w3"], c["grad_out"]),
                          c["interp_grad"])
    assert np.array_equal(oracle.group_point(c["points"], c["gidx"]), c["group"])
    assert np.array_equal(oracle.group_point_grad(c["points"].shape, c["gidx"], c["ggrad_out"]), c["group_grad"])
    d, i = oracle.three_nn(np.zeros((2, 5, 3), np.float32), c["xyz2"])
    assert np.array_equal(d, c["nn_origin_dist"]) and np.array_equal(i, c["nn_origin_idx"])


def test_three_nn_vs_reference_twin_away_from_the_origin(oracle):
    """tests/golden/twins_nn_lattice.npz: the reference twin on pre-translated lattice candidates (exact arithmetic) ==
    tf_interpolate.cpp's three_nn for queries anywhere, ties (dozens on a 17^3 lattice) included."""
    c = load("twins_nn_lattice.npz")
    d, i = oracle.three_nn(c["xyz1"], c["xyz2"])
    assert np.array_equal(d, c["dist"]) and np.array_equal(i, c["idx"])
    assert int((c["dist"][:, :, 0] == c["dist"][:, :, 1]).sum()) > 10   # the fixture does exercise the tie rule


def test_live_reference_twins_if_built(oracle):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference)")
    rng = np.random.default_rng(5)
    pts = rng.standard_normal((3, 64, 16), dtype=np.float32)
    idx = rng.integers(0, 64, (3, 200, 3)).astype(np.int32)
    w = rng.random((3, 200, 3), dtype=np.float32)
    assert np.array_equal(oracle.three_interpolate(pts, idx, w), oracle.ref_three_interpolate(pts, idx, w))
    gi = rng.integers(0, 64, (3, 10, 5)).astype(np.int32)
    assert np.array_equal(oracle.group_point(pts, gi), oracle.ref_group_point(pts, gi))


def test_three_nn_semantics(oracle):
    rng = np.random.default_rng(3)
    a = rng.random((2, 50, 3), dtype=np.float32)
    b = rng.random((2, 20, 3), dtype=np.float32)
    d, i = oracle.three_nn(a, b)
    D = ((a[:, :, None, :].astype(np.float64) - b[:, None, :, :]) ** 2).sum(-1)
    assert np.array_equal(i, np.argsort(D, axis=2, kind="stable")[:, :, :3])
    assert close(d, np.sort(D, axis=2)[:, :, :3], 1e-5, 1e-7)
    # fewer than 3 candidates: 1e40 -> inf, index 0 (tf_interpolate.cpp:66-67,91-96)
    d, i = oracle.three_nn(a[:, :4], b[:, :2])
    assert np.all(np.isinf(d[:, :, 2])) and np.all(i[:, :, 2] == 0)


def test_fps_properties_and_golden(oracle):
    c = load("fps.npz")
    idx = oracle.farthest_point_sample(128, c["xyz"])
    assert np.array_equal(idx, c["idx"]) and np.all(idx[:, 0] == 0)
    assert all(len(set(r)) == 128 for r in idx)
    # greedy property: each pick maximises the distance to the already-picked set
    x = c["xyz"][0].astype(np.float64)
    md = np.full(1024, np.inf)
    for j in range(1, 40):
        md = np.minimum(md, ((x - x[idx[0, j - 1]]) ** 2).sum(1))
        assert md[idx[0, j]] >= md.max() * (1 - 1e-6)
    assert np.array_equal(oracle.farthest_point_sample(64, c["lat"]), c["lat_idx"])
    # tie rule: all-equal distances -> smallest (k % 512, k / 512)
    z = np.zeros((1, 1500, 3), np.float32)
    z[0, 700] = 1.0; z[0, 188] = 1.0  # two equidistant farthest points: 188 = (188,0) beats 700 = (188,1)
    assert oracle.farthest_point_sample(2, z)[0, 1] == 188


def test_training_mode_batchnorm_of_the_numpy_graph():
    """oracle/model_np._bn in training mode against torch's batch_norm on the CPU: batch statistics with the biased
    variance in the normalisation, moving buffers updated with decay 0.9 / 0.999 and the Bessel-corrected variance
    (the fused kernel's rule; torch's is the same), the biased one for cluster_bn; a cloud mask drops padding clouds."""
    import torch
    from oracle import model_np
    rng = np.random.default_rng(3)
    x = rng.standard_normal((6, 50, 16)).astype(np.float32) * 2 + 1
    w = {"s/gamma": rng.random(16).astype(np.float32) + 0.5, "s/beta": rng.standard_normal(16).astype(np.float32),
         "s/mean/EMA": rng.standard_normal(16).astype(np.float32), "s/variance/EMA": rng.random(16).astype(np.float32) + 0.5}
    st = model_np.TrainState()
    y = model_np._bn(x, w, "s", 2, 1e-5, train=st, decay=0.9)
    rm, rv = torch.from_numpy(w["s/mean/EMA"].copy()), torch.from_numpy(w["s/variance/EMA"].copy())
    yt = torch.nn.functional.batch_norm(torch.from_numpy(x).reshape(-1, 16), rm, rv, torch.from_numpy(w["s/gamma"]),
                                        torch.from_numpy(w["s/beta"]), training=True, momentum=0.1, eps=1e-5)
    assert np.allclose(y.reshape(-1, 16), yt.numpy(), rtol=1e-5, atol=1e-5)
    assert np.allclose(st.updates["s/mean/EMA"], rm.numpy(), rtol=1e-5, atol=1e-6)
    assert np.allclose(st.updates["s/variance/EMA"], rv.numpy(), rtol=1e-5, atol=1e-6)
    # inference mode is untouched by the switch
    y0 = model_np._bn(x, w, "s", 2, 1e-5)
    assert np.allclose(y0, (x - w["s/mean/EMA"]) / np.sqrt(w["s/variance/EMA"] + np.float32(1e-5)) * w["s/gamma"] + w["s/beta"])
    # biased moving variance (cluster_bn, fused=False upstream): n/(n-1) apart
    st2 = model_np.TrainState()
    model_np._bn(x, w, "s", 2, 1e-5, train=st2, decay=0.9, bessel=False)
    n = 300.0
    dv_b = st2.updates["s/variance/EMA"] - 0.9 * w["s/variance/EMA"]
    dv_u = st.updates["s/variance/EMA"] - 0.9 * w["s/variance/EMA"]
    assert np.allclose(dv_u, dv_b * n / (n - 1), rtol=1e-4)
    # mask: padding clouds do not count
    mask = np.array([1, 1, 1, 1, 0, 0], bool)
    st3 = model_np.TrainState(mask=mask)
    ym = model_np._bn(x, w, "s", 2, 1e-5, train=st3, decay=0.9)
    st4 = model_np.TrainState()
    y4 = model_np._bn(x[:4], w, "s", 2, 1e-5, train=st4, decay=0.9)
    assert np.allclose(ym[:4], y4, rtol=1e-5, atol=1e-6) and np.allclose(st3.updates["s/mean/EMA"], st4.updates["s/mean/EMA"])
