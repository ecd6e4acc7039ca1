"""GPU parity (through the C ABI): drop-in operators in the reference layouts vs the CPU oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def close(a, b, rtol=1e-4, atol=1e-6):  # reference criterion, user_ops/misc.py:89-97
    return np.allclose(a, b, rtol=rtol, atol=atol)


def close_sum(a, b, rtol=1e-4, k=2e-6):
    """Same criterion with the absolute floor scaled to the output magnitude: for the larger shapes an output
    is a sum of K*Din terms, and a different (equally valid) summation order moves it by ~1e-7 * sum|terms|
    even where the sum itself cancels to ~0 (the reference only tests Din=2, where 1e-6 is enough)."""
    return np.allclose(a, b, rtol=rtol, atol=max(1e-6, k * float(np.abs(b).max())))


def load(name):
    return dict(np.load(os.path.join(G, name)))


# ------------------------------------------------------------------------------ kNN
@pytest.mark.parametrize("B,N,K", [(2, 32, 4), (1, 4, 4), (2, 300, 8), (1, 1100, 8), (1, 2048, 16), (2, 4096, 8),
                                   (1, 100, 50)])
def test_knn_bitexact_vs_oracle(dev, oracle, B, N, K):
    from dh3d_amd import ops
    rng = np.random.default_rng(1000 + N + K)
    pos = rng.random((B, 3, N), dtype=np.float32) * 40 - 20
    nn, d = ops.knn_bruteforce(T(pos, dev), K)
    enn, ed = oracle.knn_bruteforce(pos, K)
    assert np.array_equal(nn.cpu().numpy(), enn)
    assert np.array_equal(d.cpu().numpy(), ed)  # IEEE sqrt + explicit fma chain: bit-exact distances too


def test_cfg1_exact_workload(dev, oracle):
    """BASELINE config 1 as bench.py's cpu_baseline times it on the host (oracle.cpu_worker.cfg1: seed 1001, ONE cloud of
    N=1024, K=8, knn_bruteforce + one flex_conv 32->32) -- the same arrays through the HIP drop-in operators: ids and
    distance bits equal, the convolution within the reference's own criterion (test_flex_convolution.py:45)."""
    from dh3d_amd import ops
    from oracle.cpu_worker import cfg1_inputs
    pos, feat, theta, bias = cfg1_inputs()
    assert pos.shape == (1, 3, 1024) and feat.shape == (1, 32, 1024) and theta.shape == (3, 32, 32)
    nn, d = ops.knn_bruteforce(T(pos, dev), 8)                                # [1, N, K]
    enn, ed = oracle.knn_bruteforce(pos, 8)
    assert np.array_equal(nn.cpu().numpy(), enn) and np.array_equal(d.cpu().numpy(), ed)
    nbr = nn.transpose(1, 2).contiguous()                                      # every consumer wants [B, K, N]
    out = ops.flex_convolution(T(feat, dev), T(pos, dev), nbr, T(theta, dev), T(bias, dev))
    exp = oracle.flex_convolution(feat, pos, np.ascontiguousarray(enn.transpose(0, 2, 1)), theta, bias, True)
    assert out.shape == (1, 32, 1024)
    assert close_sum(out.cpu().numpy(), exp)


def test_knn_golden_and_ties(dev, oracle):
    from dh3d_amd import ops
    c = load("fake_pointcloud.npz")
    nn, d = ops.knn_bruteforce(T(c["position"], dev), 4)
    assert np.array_equal(nn.cpu().numpy(), c["knn_scipy_ids"])  # scipy answer of test_knn_bruteforce.py
    assert close(d.cpu().numpy(), c["knn_scipy_dist"], 1e-6, 1e-6)
    t = load("knn_ties.npz")
    for name in ("lat300", "lat1100"):
        nn, d = ops.knn_bruteforce(T(t[name + "_pos"], dev), 8)
        assert np.array_equal(nn.cpu().numpy(), t[name + "_nn"]), name
        assert np.array_equal(d.cpu().numpy(), t[name + "_dist"]), name
    # duplicates + K > N padding
    z = np.zeros((1, 3, 300), np.float32)
    nn, _ = ops.knn_bruteforce(T(z, dev), 8)
    assert np.array_equal(nn.cpu().numpy(), oracle.knn_bruteforce(z, 8)[0])
    p3 = np.random.default_rng(0).random((1, 3, 3), dtype=np.float32)
    nn, d = ops.knn_bruteforce(T(p3, dev), 4)
    enn, ed = oracle.knn_bruteforce(p3, 4)
    assert np.array_equal(nn.cpu().numpy(), enn) and np.array_equal(d.cpu().numpy(), ed)


def test_knn_full_size_properties(dev, oracle):
    """N = 8192 (BASELINE cfg2 size): sortedness, self at rank 0, and an oracle spot check."""
    from dh3d_amd import ops, pm
    rng = np.random.default_rng(2002)
    xyz = rng.random((2, 8192, 3), dtype=np.float32)
    nn, d = pm.knn_xyz(T(xyz, dev), 8)
    nn2, d2 = ops.knn_bruteforce(T(xyz.transpose(0, 2, 1), dev), 8)
    assert torch.equal(nn, nn2) and torch.equal(d, d2)  # both layouts, same kernel family
    nn, d = nn.cpu().numpy(), d.cpu().numpy()
    assert np.all(np.diff(d, axis=2) >= 0) and np.all(d[:, :, 0] == 0)
    assert np.array_equal(nn[:, :, 0], np.broadcast_to(np.arange(8192), (2, 8192)))
    sub = np.ascontiguousarray(xyz[:1].transpose(0, 2, 1))
    enn, ed = oracle.knn_bruteforce(sub, 8)  # ~1.5 s on CPU
    assert np.array_equal(nn[:1], enn) and np.array_equal(d[:1], ed)


def test_knn_beyond_reference_cap(dev, oracle):
    from dh3d_amd import pm
    xyz = np.random.default_rng(5005).random((1, 9000, 3), dtype=np.float32)
    nn, d = pm.knn_xyz(T(xyz, dev), 12)
    enn, ed = oracle.knn_bruteforce(np.ascontiguousarray(xyz.transpose(0, 2, 1)), 12)
    assert np.array_equal(nn.cpu().numpy(), enn) and np.array_equal(d.cpu().numpy(), ed)


# ------------------------------------------------------------------------------ FPS
@pytest.mark.parametrize("B,Dp,N,K", [(2, 2, 700, 8), (1, 5, 300, 16), (2, 8, 1030, 4), (1, 16, 260, 33), (1, 4, 6, 8)])
def test_knn_any_position_dimension_bitexact_vs_oracle(dev, oracle, B, Dp, N, K):
    """knn_bruteforce on positions of any dimension (the reference loops over Dp, knn_bruteforce_kernel_gpu.cu.cc:
    98-107): ids and distances bit-equal to the oracle, exact ties (a lattice) ordered by the CUB rank, N < K padded
    with -1 / FLT_MAX."""
    from dh3d_amd import ops
    rng = np.random.default_rng(Dp * 1000 + N)
    pos = rng.random((B, Dp, N), dtype=np.float32)
    pos[0] = np.round(pos[0] * 4) / 4          # a coarse lattice: many exact distance ties
    nn, d = ops.knn_bruteforce(T(pos, dev), K)
    enn, ed = oracle.knn_bruteforce(pos, K)
    assert np.array_equal(nn.cpu().numpy(), enn) and np.array_equal(d.cpu().numpy(), ed)


@pytest.mark.parametrize("B,N,m", [(2, 1024, 128), (1, 600, 64), (2, 4096, 512), (1, 8192, 1024), (1, 100, 100),
                                   (1, 10000, 300)])
def test_fps_bitexact_vs_oracle(dev, oracle, B, N, m):
    from dh3d_amd import ops
    xyz = np.random.default_rng(N + m).random((B, N, 3), dtype=np.float32)
    idx = ops.farthest_point_sample(m, T(xyz, dev)).cpu().numpy()
    assert np.array_equal(idx, oracle.farthest_point_sample(m, xyz))


def test_fps_golden_and_ties(dev, oracle):
    from dh3d_amd import ops
    c = load("fps.npz")
    assert np.array_equal(ops.farthest_point_sample(128, T(c["xyz"], dev)).cpu().numpy(), c["idx"])
    assert np.array_equal(ops.farthest_point_sample(64, T(c["lat"], dev)).cpu().numpy(), c["lat_idx"])
    z = np.zeros((1, 1500, 3), np.float32)
    z[0, 700] = 1.0; z[0, 188] = 1.0
    assert int(ops.farthest_point_sample(2, T(z, dev))[0, 1]) == 188


# ------------------------------------------------------------------------------ flex ops (reference layout)
def _case(rng, B, N, K, Din, Dout, oracle):
    pos = rng.standard_normal((B, 3, N)).astype(np.float32)
    nn, _ = oracle.knn_bruteforce(pos, K)
    return dict(position=pos, neighborhood=np.ascontiguousarray(nn.transpose(0, 2, 1)),
                features=rng.standard_normal((B, Din, N)).astype(np.float32),
                theta=rng.standard_normal((3, Din, Dout)).astype(np.float32),
                bias=rng.standard_normal((Din, Dout)).astype(np.float32),
                theta_rel=rng.standard_normal((Din, Dout)).astype(np.float32),
                bias_rel=rng.standard_normal((Dout,)).astype(np.float32),
                topdiff=rng.standard_normal((B, Dout, N)).astype(np.float32))


@pytest.mark.parametrize("shape", [(2, 32, 4, 2, 6), (2, 200, 8, 16, 24), (1, 513, 8, 32, 40), (1, 700, 12, 100, 36),
                                   (2, 1024, 8, 48, 96)])
def test_flex_conv_fwd_bwd(dev, oracle, shape):
    """The drop-in op at shapes no DH3D layer has: channel counts that are multiples of four run the factorisation in
    two launches (S, then S @ [bias; theta] on the GEMM kernels; csrc/flex_bwd.hip), anything else the reference
    formulation (csrc/flex_generic.hip) -- both against the oracle, forward and the three gradients."""
    from dh3d_amd import ops
    c = load("fake_pointcloud.npz") if shape == (2, 32, 4, 2, 6) else _case(np.random.default_rng(7), *shape, oracle)
    f = T(c["features"], dev).requires_grad_()
    th = T(c["theta"], dev).requires_grad_()
    bi = T(c["bias"], dev).requires_grad_()
    out = ops.flex_convolution(f, T(c["position"], dev), T(c["neighborhood"], dev), th, bi)
    exp = oracle.flex_convolution(c["features"], c["position"], c["neighborhood"], c["theta"], c["bias"], True)
    chk = close if shape == (2, 32, 4, 2, 6) else close_sum  # the reference's exact criterion on its own fixture
    assert chk(out.detach().cpu().numpy(), exp)
    out.backward(T(c["topdiff"], dev))
    gf, gt, gb = oracle.flex_convolution_grad(c["features"], c["position"], c["neighborhood"], c["theta"],
                                              c["bias"], c["topdiff"])
    # atomics / reduction order differ: the reference allows 1e-3 here (test_flex_convolution.py:120-133)
    assert close(f.grad.cpu().numpy(), gf, 1e-3, 1e-4)
    assert close(th.grad.cpu().numpy(), gt, 1e-3, 1e-3)
    assert close(bi.grad.cpu().numpy(), gb, 1e-3, 1e-3)


@pytest.mark.parametrize("shape", [(2, 32, 4, 2, 6), (2, 300, 8, 3, 32)])
def test_conv_pointset_fwd_bwd(dev, oracle, shape):
    from dh3d_amd import ops
    c = load("fake_pointcloud.npz") if shape == (2, 32, 4, 2, 6) else _case(np.random.default_rng(8), *shape, oracle)
    f = T(c["features"], dev).requires_grad_()
    th = T(c["theta_rel"], dev).requires_grad_()
    bi = T(c["bias_rel"], dev).requires_grad_()
    out = ops.convolution_pointset(f, T(c["neighborhood"], dev), th, bi)
    chk = close if shape == (2, 32, 4, 2, 6) else close_sum
    assert chk(out.detach().cpu().numpy(), oracle.convolution_pointset(c["features"], c["neighborhood"],
                                                                       c["theta_rel"], c["bias_rel"]))
    out.backward(T(c["topdiff"], dev))
    gf, gt, gb = oracle.convolution_pointset_grad(c["features"], c["neighborhood"], c["theta_rel"], c["topdiff"])
    assert close(f.grad.cpu().numpy(), gf, 1e-3, 1e-4)
    assert close(th.grad.cpu().numpy(), gt, 1e-3, 1e-3)
    assert close(bi.grad.cpu().numpy(), gb, 1e-3, 1e-3)


def test_flex_pool_exact_and_known_answer(dev, oracle):
    from dh3d_amd import ops
    k = load("flex_pool_kat.npz")
    x = T(k["x"], dev).requires_grad_()
    out, arg = ops.flex_pooling(x, T(k["nbr"], dev))
    assert np.array_equal(out.detach().cpu().numpy(), k["out"]) and np.array_equal(arg.cpu().numpy(), k["argmax"])
    out.sum().backward()
    assert np.array_equal(x.grad.cpu().numpy(), k["grad"])  # all 4 units of gradient land on index 2
    c = _case(np.random.default_rng(9), 2, 257, 8, 40, 8, oracle)
    c["features"][0, :, 5] = c["features"][0, :, 9]  # exact value ties: first max in neighbour order wins
    f = T(c["features"], dev).requires_grad_()
    out, arg = ops.flex_pooling(f, T(c["neighborhood"], dev))
    eo, ea = oracle.flex_pooling(c["features"], c["neighborhood"])
    assert np.array_equal(out.detach().cpu().numpy(), eo) and np.array_equal(arg.cpu().numpy(), ea)
    top = np.random.default_rng(1).standard_normal(eo.shape).astype(np.float32)
    out.backward(T(top, dev))
    assert close(f.grad.cpu().numpy(), oracle.flex_pooling_grad(top, ea), 1e-5, 1e-5)


# ------------------------------------------------------------------------------ PointNet++ ops
def test_group_point_and_interpolate_vs_reference_twin_golden(dev):
    from dh3d_amd import ops
    c = load("twins.npz")
    p = T(c["points"], dev).requires_grad_()
    out = ops.group_point(p, T(c["gidx"], dev))
    assert np.array_equal(out.detach().cpu().numpy(), c["group"])
    out.backward(T(c["ggrad_out"], dev))
    assert close(p.grad.cpu().numpy(), c["group_grad"], 1e-5, 1e-5)
    p2 = T(c["points"], dev).requires_grad_()
    o2 = ops.three_interpolate(p2, T(c["idx3"], dev), T(c["w3"], dev))
    assert close(o2.detach().cpu().numpy(), c["interp"], 1e-6, 1e-6)
    o2.backward(T(c["grad_out"], dev))
    assert close(p2.grad.cpu().numpy(), c["interp_grad"], 1e-5, 1e-5)
    d, i = ops.three_nn(torch.zeros(2, 5, 3, device=dev), T(c["xyz2"], dev))
    assert np.array_equal(i.cpu().numpy(), c["nn_origin_idx"]) and np.array_equal(d.cpu().numpy(), c["nn_origin_dist"])


@pytest.mark.parametrize("b,n,m", [(2, 50, 20), (2, 4096, 512), (1, 1000, 2), (1, 3000, 1100)])
def test_three_nn_bitexact(dev, oracle, b, n, m):
    from dh3d_amd import ops
    rng = np.random.default_rng(n + m)
    x1 = rng.random((b, n, 3), dtype=np.float32)
    x2 = rng.random((b, m, 3), dtype=np.float32)
    d, i = ops.three_nn(T(x1, dev), T(x2, dev))
    ed, ei = oracle.three_nn(x1, x2)
    assert np.array_equal(i.cpu().numpy(), ei) and np.array_equal(d.cpu().numpy(), ed)


def test_error_behaviour(dev):
    from dh3d_amd import ops
    with pytest.raises(ValueError):
        ops.farthest_point_sample(0, torch.zeros(1, 8, 3, device=dev))
    with pytest.raises(ValueError):
        ops.farthest_point_sample(4, torch.zeros(1, 8, 2, device=dev))
    with pytest.raises(ValueError):
        ops.flex_pooling(torch.zeros(1, 4, 8, device=dev), torch.zeros(2, 3, 8, dtype=torch.int32, device=dev))
    with pytest.raises(ValueError):
        ops.knn_bruteforce(torch.zeros(1, 3, 8, device=dev, dtype=torch.float64), 2)


# ------------------------------------------------------------------------------ spatially ordered variants
@pytest.mark.parametrize("B,N,K", [(2, 64, 4), (1, 100, 8), (2, 1000, 8), (2, 4096, 8), (1, 8192, 8), (1, 8192, 16),
                                   (1, 9000, 12), (1, 16384, 8), (3, 777, 50)])
def test_knn_sorted_identical_to_bruteforce(dev, oracle, B, N, K):
    from dh3d_amd import pm
    rng = np.random.default_rng(N * 7 + K)
    xyz = (rng.random((B, N, 3), dtype=np.float32) * 40 - 20)
    xyz[0, N // 2:, :] *= 0.05  # strongly non-uniform density: half the cloud in a tiny blob
    t = T(xyz, dev)
    srt, gbox = pm.spatial_sort(t)
    perm = srt[:, :, 3].contiguous().view(torch.int32).cpu().numpy()
    assert all(sorted(p.tolist()) == list(range(N)) for p in perm)  # a permutation
    assert np.array_equal(srt[:, :, :3].cpu().numpy(), np.take_along_axis(xyz, perm[:, :, None].astype(np.int64), 1))
    nn, d = pm.knn_sorted(srt, gbox, K)
    nn0, d0 = pm.knn_xyz(t, K)
    assert torch.equal(nn, nn0) and torch.equal(d, d0)
    if N <= 4096:
        enn, ed = oracle.knn_bruteforce(np.ascontiguousarray(xyz.transpose(0, 2, 1)), K)
        assert np.array_equal(nn.cpu().numpy(), enn) and np.array_equal(d.cpu().numpy(), ed)


def test_knn_sorted_ties_duplicates_lattice(dev, oracle):
    from dh3d_amd import pm
    t = load("knn_ties.npz")
    for name in ("lat300", "lat1100"):
        xyz = np.ascontiguousarray(t[name + "_pos"].transpose(0, 2, 1))
        srt, gbox = pm.spatial_sort(T(xyz, dev))
        nn, d = pm.knn_sorted(srt, gbox, 8)
        assert np.array_equal(nn.cpu().numpy(), t[name + "_nn"]) and np.array_equal(d.cpu().numpy(), t[name + "_dist"])
    z = np.zeros((1, 300, 3), np.float32)  # all points identical: degenerate boxes, pure tie order
    srt, gbox = pm.spatial_sort(T(z, dev))
    nn, _ = pm.knn_sorted(srt, gbox, 8)
    assert np.array_equal(nn.cpu().numpy(), oracle.knn_bruteforce(np.ascontiguousarray(z.transpose(0, 2, 1)), 8)[0])
    p3 = np.random.default_rng(0).random((1, 3, 3), dtype=np.float32)  # K > N padding
    srt, gbox = pm.spatial_sort(T(p3, dev))
    nn, d = pm.knn_sorted(srt, gbox, 4)
    enn, ed = oracle.knn_bruteforce(np.ascontiguousarray(p3.transpose(0, 2, 1)), 4)
    assert np.array_equal(nn.cpu().numpy(), enn) and np.array_equal(d.cpu().numpy(), ed)


@pytest.mark.parametrize("B,N,m", [(2, 1024, 128), (1, 600, 64), (2, 4096, 512), (2, 8192, 1024), (1, 100, 100),
                                   (1, 10000, 300), (1, 12288, 1536)])
def test_fps_sorted_identical(dev, oracle, B, N, m):
    from dh3d_amd import ops, pm
    rng = np.random.default_rng(N + m)
    xyz = rng.random((B, N, 3), dtype=np.float32)
    xyz[0, : N // 3] = xyz[0, : N // 3] * 0.02 + 0.5  # dense blob
    t = T(xyz, dev)
    srt, gbox = pm.spatial_sort(t)
    idx = pm.fps_sorted(srt, gbox, m)
    assert torch.equal(idx, ops.farthest_point_sample(m, t))
    assert np.array_equal(idx.cpu().numpy(), oracle.farthest_point_sample(m, xyz))
    idx2, xyz_s = pm.fps_sorted(srt, gbox, m, with_xyz=True)  # + the sampled coordinates from the same kernel
    assert torch.equal(idx2, idx)
    assert np.array_equal(xyz_s.cpu().numpy(), np.take_along_axis(xyz, idx.cpu().numpy()[:, :, None].astype(np.int64), 1))


@pytest.mark.parametrize("case", ["uniform", "scene", "dup", "exhausted", "small"])
def test_fps_sorted_ordered_sampled_set(dev, oracle, case):
    """dh3d_fps_sorted_ordered: the same picks as the plain op, plus the sampled set in the cloud's Morton order out of the
    same launch -- records are a permutation of the picks (coordinates of pick r under rank r), every box is exactly the
    min / max of its 64 records, the subset's cell table partitions the records exactly as the cloud's table partitions the
    picked positions (duplicate picks of one point included), and the consumers agree with the oracle on it: three_nn of the
    cloud against the sampled set and the sampled set's own kNN (cell lists on the inherited grid), ids and distance bits."""
    from dh3d_amd import ops, pm
    rng = np.random.default_rng(606)
    B, N, m = 3, 8192, 1024
    xyz = rng.random((B, N, 3), dtype=np.float32)
    if case == "scene":
        xyz = (xyz * np.array([60, 60, 6], np.float32) - np.array([30, 30, 3], np.float32)).astype(np.float32)
        xyz[:, : N // 2, 2] = -3 + 0.05 * rng.random((B, N // 2), dtype=np.float32)     # ground plane
    elif case == "dup":
        B, N, m = 2, 5000, 625
        xyz = np.repeat(rng.random((B, N // 2, 3), dtype=np.float32), 2, 1)[:, rng.permutation(N)]
    elif case == "exhausted":       # 300 distinct points, 512 picks: the tail of the picks repeats one point
        B, N, m = 2, 4096, 512
        base = rng.random((B, 300, 3), dtype=np.float32)
        xyz = base[:, rng.integers(0, 300, N)]
    elif case == "small":
        B, N, m = 2, 4096, 512
        xyz = rng.random((B, N, 3), dtype=np.float32)
    t = T(xyz, dev)
    srt, gbox, cells = pm.spatial_sort_cells(t)
    idx, xyz_s, srt_s, gbox_s, cells_s = pm.fps_sorted_ordered(srt, gbox, m, cells=cells)
    torch.cuda.synchronize()
    assert torch.equal(idx, pm.fps_sorted(srt, gbox, m))
    idx_h, xs, rec = idx.cpu().numpy(), xyz_s.cpu().numpy(), srt_s.cpu().numpy()
    assert np.array_equal(idx_h, oracle.farthest_point_sample(m, xyz))
    assert np.array_equal(xs, np.take_along_axis(xyz, idx_h[:, :, None].astype(np.int64), 1))
    rank = rec[:, :, 3].view(np.int32)
    srt_h, ct, cs = srt.cpu().numpy(), cells.cpu().numpy(), cells_s.cpu().numpy()
    gb = gbox_s.cpu().numpy()
    for b in range(B):
        assert np.array_equal(np.sort(rank[b]), np.arange(m)), "records are not a permutation of the picks"
        assert np.array_equal(rec[b, :, :3], xs[b][rank[b]])
        # Morton order = the order of the picked positions in the sorted cloud (ties: the same position picked again)
        pos_of = np.empty(N, np.int64)
        pos_of[srt_h[b, :, 3].view(np.int32)] = np.arange(N)
        ppos = pos_of[idx_h[b][rank[b]]]
        assert np.all(np.diff(ppos) >= 0), "records are not in the cloud's order"
        for g in range((m + 63) // 64):
            blk = rec[b, g * 64:(g + 1) * 64, :3]
            assert np.array_equal(gb[b, g, 0:3], blk.min(0)) and np.array_equal(gb[b, g, 4:7], blk.max(0)), g
        # the subset's table: cell c holds the records whose cloud position lies in the cloud's cell c
        assert cs[b, 0] == 0 and cs[b, 4096] == m and np.all(np.diff(cs[b, :4097]) >= 0)
        cell_of_pos = np.searchsorted(ct[b, 1:4097], np.arange(N), side="right")     # cloud position -> cell
        assert np.array_equal(np.bincount(cell_of_pos[ppos], minlength=4096), np.diff(cs[b, :4097]))
        assert np.array_equal(cs[b, 4100:4106], ct[b, 4100:4106]) and cs[b, 4107] == ct[b, 4107]   # the cloud's grid
        assert cs[b, 4106] in (0, 1)
    # consumers
    d3, i3 = pm.three_nn_sorted(srt, gbox, srt_s, gbox_s)
    ed, ei = oracle.three_nn(xyz, xs)
    assert np.array_equal(i3.cpu().numpy(), ei) and np.array_equal(d3.cpu().numpy(), ed)
    nn, dd = pm.knn_grid(srt_s, gbox_s, cells_s, 8)
    enn, edd = oracle.knn_bruteforce(np.ascontiguousarray(xs.transpose(0, 2, 1)), 8)
    assert np.array_equal(nn.cpu().numpy(), enn) and np.array_equal(dd.cpu().numpy(), edd)
    nn2, dd2 = pm.knn_sorted(srt_s, gbox_s, 8)
    assert np.array_equal(nn2.cpu().numpy(), enn) and np.array_equal(dd2.cpu().numpy(), edd)


def test_fps_sorted_batched_rounds_adversarial(dev, oracle):
    """Several picks per synchronisation must stay the sequential picks under exact ties, duplicates and exhaustion."""
    from dh3d_amd import ops, pm
    rng = np.random.default_rng(5)
    g = np.stack(np.meshgrid(*[np.arange(16, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(1, -1, 3)
    lattice = g[:, rng.permutation(4096)]                                # every distance tied many times over
    dup = np.repeat(rng.random((1, 2500, 3), dtype=np.float32), 2, 1)    # each point twice: zero distances early
    dup = dup[:, rng.permutation(5000)]
    plane = rng.random((2, 6000, 3), dtype=np.float32); plane[:, :, 2] *= 1e-3
    same = np.zeros((1, 1500, 3), np.float32)
    for xyz, m in ((lattice, 4096), (lattice, 700), (dup, 3000), (plane, 1500), (same, 20),
                   (rng.random((3, 70, 3), dtype=np.float32), 70), (rng.random((1, 5000, 3), dtype=np.float32), 5000)):
        t = T(xyz, dev)
        srt, gbox = pm.spatial_sort(t)
        idx = pm.fps_sorted(srt, gbox, m)
        assert torch.equal(idx, ops.farthest_point_sample(m, t))
        if xyz.shape[1] * m <= 4096 * 1024:
            assert np.array_equal(idx.cpu().numpy(), oracle.farthest_point_sample(m, xyz))


def test_fps_sorted_golden_and_ties(dev):
    from dh3d_amd import pm
    c = load("fps.npz")
    for pts, m, key in ((c["xyz"], 128, "idx"), (c["lat"], 64, "lat_idx")):
        srt, gbox = pm.spatial_sort(T(pts, dev))
        assert np.array_equal(pm.fps_sorted(srt, gbox, m).cpu().numpy(), c[key])
    z = np.zeros((1, 1500, 3), np.float32)
    z[0, 700] = 1.0; z[0, 188] = 1.0
    srt, gbox = pm.spatial_sort(T(z, dev))
    assert int(pm.fps_sorted(srt, gbox, 2)[0, 1]) == 188


@pytest.mark.parametrize("contract", [1, 0])
@pytest.mark.parametrize("B,N,m", [(2, 777, 100), (1, 4096, 512), (1, 20000, 300)])
def test_fps_mode_any_n_both_contractions(dev, oracle, B, N, m, contract):
    """dh3d_farthest_point_sample_mode: the any-N kernel (running distances in scratch, as upstream) in both
    roundings of tf_sampling_g.cu:141 -- each bit-equal to the oracle in the same mode; N > 16384 included."""
    from dh3d_amd import ops
    rng = np.random.default_rng(N + contract)
    xyz = rng.random((B, N, 3), dtype=np.float32) * 30 - 15
    got = ops.farthest_point_sample(m, T(xyz, dev), contract=contract).cpu().numpy()
    assert np.array_equal(got, oracle.farthest_point_sample(m, xyz, contract=bool(contract)))
    if contract == 1 and N <= 16384:  # the default kernels compute the contracted form
        assert np.array_equal(got, ops.farthest_point_sample(m, T(xyz, dev)).cpu().numpy())


def test_fps_modes_differ_somewhere_and_ties(dev, oracle):
    """The two roundings are not the same function (otherwise the switch would be untestable), and the tie rule
    holds in the any-N kernel: lattice clouds where many points share the maximum."""
    from dh3d_amd import ops
    rng = np.random.default_rng(99)
    differ = 0
    for s in range(6):
        xyz = (rng.random((1, 3000, 3), dtype=np.float32) * 50 - 25)
        a = ops.farthest_point_sample(400, T(xyz, dev), contract=1).cpu().numpy()
        b = ops.farthest_point_sample(400, T(xyz, dev), contract=0).cpu().numpy()
        differ += int(not np.array_equal(a, b))
    lat = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(12), indexing="ij"), -1).reshape(1, -1, 3)
    lat = lat.astype(np.float32)[:, rng.permutation(1728)]
    for c in (0, 1):
        assert np.array_equal(ops.farthest_point_sample(200, T(lat, dev), contract=c).cpu().numpy(),
                              oracle.farthest_point_sample(200, lat, contract=bool(c)))
    print("clouds on which the two FPS roundings pick differently: %d / 6" % differ)


@pytest.mark.parametrize("b,n,m", [(2, 8192, 1024), (3, 4096, 512), (1, 5000, 700), (1, 16384, 2048), (2, 300, 256), (1, 1000, 2)])
def test_three_nn_sorted_identical_to_bruteforce(dev, oracle, b, n, m):
    """The box-pruned three_nn on Morton-ordered sets: ids and (squared) distances bit-equal to the brute-force op and
    to the oracle, incl. a sampled set that is a subset of the cloud (distance-0 hits) and exact ties on a lattice."""
    from dh3d_amd import ops, pm
    rng = np.random.default_rng(n + m)
    xyz1 = (rng.random((b, n, 3), dtype=np.float32) * 30 - 15)
    sel = np.stack([rng.choice(n, m, replace=False) for _ in range(b)])
    xyz2 = np.take_along_axis(xyz1, sel[:, :, None], 1)            # the model's case: samples of the cloud itself
    for (a1, a2) in ((xyz1, xyz2), (np.round(xyz1), np.round(xyz2))):   # second: integer lattice -> many exact ties
        t1, t2 = T(a1, dev), T(a2, dev)
        d0, i0 = ops.three_nn(t1, t2)
        s1, g1 = pm.spatial_sort(t1)
        s2, g2 = pm.spatial_sort(t2)
        d1, i1 = pm.three_nn_sorted(s1, g1, s2, g2)
        assert torch.equal(i0, i1) and torch.equal(d0, d1)
    if n <= 5000:
        de, ie = oracle.three_nn(xyz1, xyz2)
        s1, g1 = pm.spatial_sort(T(xyz1, dev)); s2, g2 = pm.spatial_sort(T(xyz2, dev))
        dd, ii = pm.three_nn_sorted(s1, g1, s2, g2)
        assert np.array_equal(ii.cpu().numpy(), ie) and np.array_equal(dd.cpu().numpy(), de)


@pytest.mark.parametrize("B,N,m", [(1, 12289, 1536), (2, 16384, 2048), (1, 13000, 700)])
def test_fps_sorted_above_the_lds_table_limit(dev, oracle, B, N, m):
    """Clouds of 12289..16384 points: the batched-round FPS without its LDS coordinate table (winners read from the
    cloud) -- same picks as the plain op, and as the oracle."""
    from dh3d_amd import ops, pm
    rng = np.random.default_rng(N)
    xyz = rng.random((B, N, 3), dtype=np.float32) * 40 - 20
    t = T(xyz, dev)
    srt, gbox = pm.spatial_sort(t)
    idx, xyz_s = pm.fps_sorted(srt, gbox, m, with_xyz=True, xyz=t)
    assert torch.equal(idx, ops.farthest_point_sample(m, t))
    assert torch.equal(xyz_s, torch.gather(t, 1, idx.long()[:, :, None].expand(-1, -1, 3)))
    if B == 1:
        assert np.array_equal(idx.cpu().numpy(), oracle.farthest_point_sample(m, xyz))
    with pytest.raises(ValueError):
        pm.fps_sorted(srt, gbox, m)          # no cloud given: the table does not fit


def test_three_nn_vs_reference_twin_lattice_golden(dev):
    """Both device kernels against outputs of the reference's own twin (queries away from the origin, exact ties)."""
    from dh3d_amd import ops, pm
    c = load("twins_nn_lattice.npz")
    t1, t2 = T(c["xyz1"], dev), T(c["xyz2"], dev)
    d, i = ops.three_nn(t1, t2)
    assert np.array_equal(d.cpu().numpy(), c["dist"]) and np.array_equal(i.cpu().numpy(), c["idx"])
    s1, g1 = pm.spatial_sort(t1); s2, g2 = pm.spatial_sort(t2)
    d2, i2 = pm.three_nn_sorted(s1, g1, s2, g2)
    assert np.array_equal(d2.cpu().numpy(), c["dist"]) and np.array_equal(i2.cpu().numpy(), c["idx"])


def _f64_case(dev, B=2, N=32, K=4, Din=2, Dout=6, Dp=3, seed=42):
    """The reference's FakePointCloud (user_ops/misc.py:32-69; B=2, N=32, K=4, Din=2, Dout=6, Dp=3 in every op test) in
    float64, neighbours from the exact distance matrix."""
    rng = np.random.default_rng(seed)
    pos = rng.standard_normal((B, Dp, N))
    d = ((pos[:, :, :, None] - pos[:, :, None, :]) ** 2).sum(1)
    nbr = np.argsort(d, axis=2, kind="stable")[:, :, :K].transpose(0, 2, 1).astype(np.int32)   # [B,K,N], rank 0 = self
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return dict(pos=t(pos), nbr=t(nbr), feat=t(rng.standard_normal((B, Din, N))), theta=t(rng.standard_normal((Dp, Din, Dout))),
                bias=t(rng.standard_normal((Din, Dout))), theta_rel=t(rng.standard_normal((Din, Dout))),
                bias_rel=t(rng.standard_normal((Dout,))))


def test_flex_ops_float64_forward_and_numeric_gradients(dev):
    """The reference registers double kernels for the flex operators (flex_conv_op.cc:97-106) and checks its gradients in
    float64 against numeric differentiation (test_flex_convolution.py:93-115, test_conv_pointset.py, test_flex_pooling.py;
    tf.test.compute_gradient).  Same recipe on the *_f64 entry points: forward against a float64 numpy restatement of
    the operator, analytic against numeric Jacobians with torch.autograd.gradcheck at the reference's case size."""
    from dh3d_amd import ops
    c = _f64_case(dev)
    f, p, nb, th, bi = c["feat"], c["pos"], c["nbr"], c["theta"], c["bias"]
    # forward, flex_conv: out[b,o,n] = sum_k sum_i (bias[i,o] + sum_d theta[d,i,o] (p[nk] - p[n])_d) f[b,i,nk]
    fn, pn, nbn, thn, bin_ = (x.cpu().numpy() for x in (f, p, nb, th, bi))
    B, Din, N = fn.shape
    exp = np.zeros((B, thn.shape[2], N))
    for b in range(B):
        for n in range(N):
            for k in range(nbn.shape[1]):
                nk = nbn[b, k, n]
                w = bin_ + np.einsum("d,dio->io", pn[b, :, nk] - pn[b, :, n], thn)
                exp[b, :, n] += fn[b, :, nk] @ w
    out = ops.flex_convolution(f, p, nb, th, bi)
    assert out.dtype == torch.float64 and np.abs(out.cpu().numpy() - exp).max() < 1e-12
    # conv_pointset: out[b,o,n] = bias[o] + sum_k sum_i theta[i,o] (f[nk] - f[n0]),  n0 = rank-0 neighbour
    thr, br = c["theta_rel"], c["bias_rel"]
    exp = np.zeros((B, thr.shape[1], N))
    for b in range(B):
        for n in range(N):
            n0 = nbn[b, 0, n]
            for k in range(nbn.shape[1]):
                exp[b, :, n] += (fn[b, :, nbn[b, k, n]] - fn[b, :, n0]) @ thr.cpu().numpy()
            exp[b, :, n] += br.cpu().numpy()
    out = ops.convolution_pointset(f, nb, thr, br)
    assert out.dtype == torch.float64 and np.abs(out.cpu().numpy() - exp).max() < 1e-12
    # flex_pool: max over the neighbourhood, argmax = point id
    out, arg = ops.flex_pooling(f, nb)
    g = fn[np.arange(B)[:, None, None, None], np.arange(Din)[None, :, None, None], nbn[:, None, :, :]]   # [B,Din,K,N]
    assert out.dtype == torch.float64 and np.array_equal(out.cpu().numpy(), g.max(2))
    assert np.array_equal(arg.cpu().numpy(), np.take_along_axis(np.broadcast_to(nbn[:, None], g.shape), g.argmax(2)[:, :, None], 2)[:, :, 0])
    # numeric against analytic Jacobians, every differentiable input
    req = lambda x: x.clone().requires_grad_(True)
    assert torch.autograd.gradcheck(lambda a, t_, b_: ops.flex_convolution(a, p, nb, t_, b_), (req(f), req(th), req(bi)),
                                    eps=1e-6, atol=1e-7, rtol=1e-6, nondet_tol=1e-12)
    assert torch.autograd.gradcheck(lambda a, t_, b_: ops.convolution_pointset(a, nb, t_, b_), (req(f), req(thr), req(br)),
                                    eps=1e-6, atol=1e-7, rtol=1e-6, nondet_tol=1e-12)
    assert torch.autograd.gradcheck(lambda a: ops.flex_pooling(a, nb)[0], (req(f),), eps=1e-6, atol=1e-7, rtol=1e-6,
                                    nondet_tol=1e-12)
    # mixed dtypes are refused like the op registration would
    with pytest.raises(ValueError):
        ops.flex_convolution(f, p.float(), nb, th, bi)


def _knn_clouds(name, B, N, rng):
    if name == "uniform":
        return rng.random((B, N, 3), dtype=np.float32)
    if name == "oxford_extent":
        return (rng.random((B, N, 3), dtype=np.float32) * np.array([60, 60, 8], np.float32) - 30).astype(np.float32)
    if name == "clusters":  # a few dense blobs + sparse background: cells with hundreds of points next to empty ones
        c = rng.random((B, 6, 3), dtype=np.float32)
        pts = c[:, rng.integers(0, 6, N)] + rng.normal(0, 0.004, (B, N, 3)).astype(np.float32)
        pts[:, : N // 8] = rng.random((B, N // 8, 3), dtype=np.float32)
        return pts.astype(np.float32)
    if name == "lattice":   # exact ties everywhere: the CUB (id mod C_THREADS, id div C_THREADS) order decides
        g = np.stack(np.meshgrid(*[np.arange(32)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
        pts = np.stack([g[rng.permutation(len(g))[:N]] for _ in range(B)])
        return (pts * 0.25).astype(np.float32)
    if name == "duplicates":
        pts = rng.random((B, N, 3), dtype=np.float32)
        pts[:, N // 2:] = pts[:, : N - N // 2]            # every point twice
        return pts
    if name == "plane":     # degenerate extent along z
        pts = rng.random((B, N, 3), dtype=np.float32)
        pts[:, :, 2] = 0.5
        return pts
    if name == "one_point":  # everything in one cell: the brute-force limit of the cell search
        return np.tile(rng.random((B, 1, 3), dtype=np.float32), (1, N, 1))
    if name == "outliers":  # a tight cloud and a few far points: the grid is almost empty
        pts = (rng.random((B, N, 3), dtype=np.float32) * 0.01).astype(np.float32)
        pts[:, :5] = rng.random((B, 5, 3), dtype=np.float32) * 100
        return pts
    if name == "scene":     # a street scene normalised to [-1, 1]: ground plane + walls + clutter, z extent a tenth of x / y --
        out = np.empty((B, N, 3), np.float32)  # the grid's cells are a tenth as high as wide, occupied ones hold dozens of points
        for b in range(B):
            n_g, n_w = int(N * 0.55), int(N * 0.35)
            g = np.stack([rng.uniform(-1, 1, n_g), rng.uniform(-1, 1, n_g), rng.normal(-0.08, 0.004, n_g)], 1)
            walls = []
            for _ in range(6):
                m = n_w // 6
                x0, y0, ang, ln = rng.uniform(-0.8, 0.8), rng.uniform(-0.8, 0.8), rng.uniform(0, np.pi), rng.uniform(0.3, 0.9)
                tt = rng.uniform(0, ln, m)
                walls.append(np.stack([x0 + tt * np.cos(ang), y0 + tt * np.sin(ang), rng.uniform(-0.08, 0.12, m)], 1)
                             + rng.normal(0, 0.003, (m, 3)))
            w = np.concatenate(walls)
            c = rng.uniform(-1, 1, (N - n_g - len(w), 3)) * np.array([1, 1, 0.1])
            out[b] = np.clip(np.concatenate([g, w, c])[rng.permutation(N)], -1, 1)
        return out
    raise KeyError(name)


@pytest.mark.parametrize("name,B,N,K", [
    ("scene", 4, 8192, 8), ("scene", 4, 4096, 8), ("scene", 2, 16384, 5), ("scene", 3, 1000, 8),
    ("uniform", 8, 8192, 8), ("uniform", 3, 4097, 8), ("uniform", 2, 16384, 8), ("uniform", 2, 9000, 5),
    ("oxford_extent", 4, 4096, 8), ("clusters", 2, 8192, 8), ("lattice", 2, 8192, 8), ("lattice", 1, 4096, 3),
    ("duplicates", 2, 4096, 8), ("plane", 2, 4096, 8), ("one_point", 1, 2100, 8), ("outliers", 2, 4096, 8),
    ("uniform", 1, 2049, 1),
    # smaller sets: coarser grids (one bit of the cell code less per halving, down to 4 x 4 x 4 cells)
    ("uniform", 8, 1024, 8), ("uniform", 32, 512, 8), ("uniform", 4, 2048, 8), ("uniform", 3, 300, 8),
    ("uniform", 2, 128, 8), ("uniform", 2, 40, 8), ("uniform", 2, 5, 8), ("clusters", 2, 1024, 8),
    ("lattice", 2, 1000, 8), ("duplicates", 2, 512, 8), ("plane", 2, 1024, 8), ("one_point", 1, 600, 8),
    ("outliers", 2, 1024, 8), ("oxford_extent", 4, 2048, 8), ("oxford_extent", 8, 1024, 8)])
def test_knn_grid_cell_list_search_is_bit_equal_to_brute_force(dev, name, B, N, K):
    """knn_grid (cell lists on the sort's Morton grid, candidates pooled per query) == the brute-force kernel: ids AND
    distance bits, on uniform / anisotropic / clustered / tie-ridden / degenerate clouds at every grid resolution; the
    brute-force kernel itself is pinned on the oracle above."""
    from dh3d_amd import pm
    import zlib
    rng = np.random.default_rng(zlib.crc32(repr((name, B, N, K)).encode()))   # (hash() varies with PYTHONHASHSEED)
    pts = torch.from_numpy(_knn_clouds(name, B, N, rng)).to(dev)
    srt, gbox, cells = pm.spatial_sort_cells(pts)
    s2, g2 = pm.spatial_sort(pts)
    assert torch.equal(srt, s2) and torch.equal(gbox, g2)      # the cell table changes nothing else
    ct = cells[:, :4097].cpu().numpy()
    assert (ct[:, 0] == 0).all() and (ct[:, 4096] == N).all() and (np.diff(ct, axis=1) >= 0).all()
    # the sort's verdict: far fewer occupied cells than a uniform cloud of N points leaves -> the pruned scan serves the cloud
    flag = cells[:, 4106].cpu().numpy()
    occupied = (np.diff(ct, axis=1) > 0).sum(1)
    assert np.array_equal(flag != 0, occupied < int(0.6 * 4096.0 * (1.0 - np.exp(-N / 4096.0)))), (name, flag, occupied)
    if name in ("uniform", "oxford_extent"):
        assert not flag.any()
    if name in ("scene", "one_point", "clusters", "outliers") and N >= 4096:
        assert flag.all()
    nn_g, d_g = pm.knn_grid(srt, gbox, cells, K)
    nn_b, d_b = pm.knn_xyz(pts, K)
    assert torch.equal(nn_g, nn_b), (name, int((nn_g != nn_b).sum()))
    assert torch.equal(d_g.view(torch.int32), d_b.view(torch.int32))
    nn_s, d_s = pm.knn_sorted(srt, gbox, K)
    assert torch.equal(nn_s, nn_b)


def test_knn_grid_vs_oracle_N8192(dev, oracle):
    from dh3d_amd import pm
    pts = np.random.default_rng(8192).random((2, 8192, 3), dtype=np.float32)
    srt, gbox, cells = pm.spatial_sort_cells(torch.from_numpy(pts).to(dev))
    nn_g, d_g = pm.knn_grid(srt, gbox, cells, 8)
    nn_o, d_o = oracle.knn_bruteforce(np.ascontiguousarray(pts.transpose(0, 2, 1)), 8)
    assert np.array_equal(nn_g.cpu().numpy(), nn_o)
    assert np.array_equal(d_g.cpu().numpy().view(np.uint32), d_o.view(np.uint32))


def test_knn_config_size_vs_the_references_own_checker(dev):
    """a1 at config size WITHOUT the oracle in the loop: the reference's python_bruteforce (float64 pdist -> argsort,
    user_ops/test_knn_bruteforce.py:32-40) at N = 8192, K = 8.  HIP ids must equal scipy's on every row whose
    consecutive float64 distances among the first nine are further apart than a few float32 ulps (a float32 kernel
    cannot order closer pairs the float64 way); those rows must be more than 99.9 % of the cloud."""
    from scipy.spatial.distance import cdist
    from dh3d_amd import ops
    rng = np.random.default_rng(81928)
    pos = rng.random((1, 3, 8192), dtype=np.float32)
    nn, dist = ops.knn_bruteforce(torch.from_numpy(pos).to(dev), 8)
    nn, dist = nn[0].cpu().numpy(), dist[0].cpu().numpy()
    p64 = pos[0].T.astype(np.float64)
    exp_i = np.empty((8192, 9), np.int64)
    exp_d = np.empty((8192, 9))
    for s in range(0, 8192, 1024):      # (pdist of the full cloud, a block of rows at a time)
        d = cdist(p64[s:s + 1024], p64, "euclidean")
        order = np.argsort(d, axis=1)[:, :9]
        exp_i[s:s + 1024] = order
        exp_d[s:s + 1024] = np.take_along_axis(d, order, 1)
    gap = np.diff(exp_d, axis=1)
    clear = (gap > 4 * np.finfo(np.float32).eps * np.maximum(exp_d[:, 1:], 1.0)).all(1)
    assert clear.mean() > 0.999, clear.mean()
    assert np.array_equal(nn[clear], exp_i[clear, :8])
    assert np.abs(dist - exp_d[:, :8]).max() < 2e-6      # the reference's own tolerance on the distances is 1e-4
