"""Generates tests/golden/*.npz.  Run in the build container (needs /root/reference for the twins):

    python tests/golden/make_golden.py

Fixtures are DATA (inputs + expected outputs):
  fake_pointcloud.npz   the reference's own test recipe FakePointCloud(B=2,N=32,K=4,Din=2,Dout=6,Dp=3)
                        with np.random.seed(42) (user_ops/misc.py:27,31-68; the recipe needs only
                        numpy + scipy), the scipy kNN answer its test compares against
                        (user_ops/test_knn_bruteforce.py:32-40), and the oracle's outputs frozen.
  twins_nn_lattice.npz  three_nn of lattice queries away from the origin through the reference twin (pre-translated
                        candidates: exact arithmetic)
  twins.npz             outputs of the REFERENCE's stand-alone twins (oracle/_ref, compiled from
                        tf_ops/interpolation/interpolate.cpp and tf_ops/grouping/test/query_ball_point.cpp).
  flex_pool_kat.npz     the hand-made 4-point known-answer test (user_ops/test_flex_pooling.py:76-98).
  knn_ties.npz / fps.npz  oracle outputs on adversarial / seeded inputs (frozen; see DESIGN.md on pinning).
"""
import os
import sys

import numpy as np
from scipy.spatial.distance import pdist, squareform

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import cpu as O  # noqa: E402


def fake_pointcloud(B, N, K, Din, Dout, Dp):
    """Data recipe of user_ops/misc.py:31-68 (draw order matters for the seed)."""
    np.random.seed(42)
    rv = lambda shape: np.random.randn(*shape).astype(np.float32)
    theta = rv([Dp, Din, Dout]); bias = rv([Din, Dout])
    theta_rel = rv([Din, Dout]); bias_rel = rv([Dout])
    position = rv([B, Dp, N]); features = rv([B, Din, N])
    nbr = []
    for batch in position.astype(np.float64):
        d = squareform(pdist(batch.T, "euclidean"))
        nbr.append(np.argsort(d, axis=1)[:, :K])
    nbr = np.array(nbr).transpose(0, 2, 1).astype(np.int32)
    return dict(theta=theta, bias=bias, theta_rel=theta_rel, bias_rel=bias_rel, position=position,
                features=features, neighborhood=nbr)


def main():
    O.build()
    c = fake_pointcloud(2, 32, 4, 2, 6, 3)
    nn_exp, d_exp = [], []
    for batch in c["position"].astype(np.float64):  # python_bruteforce, test_knn_bruteforce.py:32-40
        d = squareform(pdist(batch.T, "euclidean"))
        nn_exp.append(np.argsort(d, axis=1)[:, :4]); d_exp.append(np.sort(d, axis=1)[:, :4])
    top = np.random.RandomState(7).randn(2, 6, 32).astype(np.float32)
    fc = O.flex_convolution(c["features"], c["position"], c["neighborhood"], c["theta"], c["bias"], True)
    gf, gt, gb = O.flex_convolution_grad(c["features"], c["position"], c["neighborhood"], c["theta"], c["bias"], top)
    cp = O.convolution_pointset(c["features"], c["neighborhood"], c["theta_rel"], c["bias_rel"])
    cgf, cgt, cgb = O.convolution_pointset_grad(c["features"], c["neighborhood"], c["theta_rel"], top)
    fp, fa = O.flex_pooling(c["features"], c["neighborhood"])
    np.savez(os.path.join(HERE, "fake_pointcloud.npz"), knn_scipy_ids=np.array(nn_exp).astype(np.int32),
             knn_scipy_dist=np.array(d_exp), topdiff=top, flex_conv=fc, flex_conv_gf=gf, flex_conv_gt=gt,
             flex_conv_gb=gb, conv_pointset=cp, conv_pointset_gf=cgf, conv_pointset_gt=cgt, conv_pointset_gb=cgb,
             flex_pool=fp, flex_pool_argmax=fa, **c)

    rng = np.random.default_rng(11)
    pts = rng.standard_normal((2, 24, 8), dtype=np.float32)
    idx3 = rng.integers(0, 24, (2, 50, 3)).astype(np.int32)
    w3 = rng.random((2, 50, 3), dtype=np.float32)
    go = rng.standard_normal((2, 50, 8), dtype=np.float32)
    gidx = rng.integers(0, 24, (2, 9, 4)).astype(np.int32)
    ggo = rng.standard_normal((2, 9, 4, 8), dtype=np.float32)
    xyz2 = rng.random((2, 40, 3), dtype=np.float32)
    d0, i0 = O.ref_three_nn_origin(xyz2, 5)
    np.savez(os.path.join(HERE, "twins.npz"), points=pts, idx3=idx3, w3=w3, grad_out=go, gidx=gidx, ggrad_out=ggo,
             xyz2=xyz2, interp=O.ref_three_interpolate(pts, idx3, w3),
             interp_grad=O.ref_three_interpolate_grad(pts.shape, idx3, w3, go),
             group=O.ref_group_point(pts, gidx), group_grad=O.ref_group_point_grad(pts.shape, gidx, ggo),
             nn_origin_dist=d0, nn_origin_idx=i0)

    # three_nn with the query away from the origin, still pinned by the reference twin: the twin measures |xyz2|^2 (it
    # drops xyz1, interpolate.cpp:34), so it is handed xyz2 - q -- on an integer lattice that subtraction, and every
    # product / sum of the distance, is exact in float32, i.e. twin(xyz2 - q) IS tf_interpolate.cpp's three_nn(q, xyz2)
    # bit for bit, ties (many on a lattice) included.
    lrng = np.random.default_rng(1111)
    lat1 = lrng.integers(-8, 9, (2, 48, 3)).astype(np.float32)
    lat2 = lrng.integers(-8, 9, (2, 300, 3)).astype(np.float32)
    nd = np.empty((2, 48, 3), np.float32)
    ni = np.empty((2, 48, 3), np.int32)
    for b in range(2):
        for j in range(48):
            d1, i1 = O.ref_three_nn_origin((lat2[b:b + 1] - lat1[b, j]).astype(np.float32), 1)
            nd[b, j], ni[b, j] = d1[0, 0], i1[0, 0]
    np.savez(os.path.join(HERE, "twins_nn_lattice.npz"), xyz1=lat1, xyz2=lat2, dist=nd, idx=ni)

    x = np.array([[[1], [2], [5], [3]]], np.float32).transpose(0, 2, 1)
    n = np.array([[[0, 1, 2, 3], [1, 2, 3, 0], [2, 3, 0, 1], [3, 0, 1, 2]]]).transpose(0, 2, 1).astype(np.int32)
    np.savez(os.path.join(HERE, "flex_pool_kat.npz"), x=x, nbr=n, out=np.full((1, 1, 4), 5, np.float32),
             argmax=np.full((1, 1, 4), 2, np.int32), grad=np.array([[[0, 0, 4, 0]]], np.float32))

    # kNN with exact ties: integer lattice (many equal distances) + duplicated points, N spanning two ladder rungs
    cases = {}
    for name, N in (("lat300", 300), ("lat1100", 1100)):
        g = rng.integers(0, 6, (1, 3, N)).astype(np.float32)
        nn, dd = O.knn_bruteforce(g, 8)
        cases[name + "_pos"], cases[name + "_nn"], cases[name + "_dist"] = g, nn, dd
    np.savez(os.path.join(HERE, "knn_ties.npz"), **cases)

    xyz = rng.random((2, 1024, 3), dtype=np.float32)
    lat = rng.integers(0, 5, (1, 600, 3)).astype(np.float32)  # ties in the FPS argmax
    np.savez(os.path.join(HERE, "fps.npz"), xyz=xyz, idx=O.farthest_point_sample(128, xyz), lat=lat,
             lat_idx=O.farthest_point_sample(64, lat))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
