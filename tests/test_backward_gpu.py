"""GPU: the backward / training kernels (section C of the C ABI) and the fast kernels behind the reference signatures
(section A'): exact-f32 MFMA GEMMs vs float64, layout helpers, the factorised flex_conv backward vs the oracle's
restatement of the reference gradient (flex_conv_kernel.cc:75-164), and ops.* fast path == reference formulation."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


# (products of >= 2^26 multiply-adds with K % 4 == 0 run on the bf16x6 kernel, the rest on the exact-f32 one)
@pytest.mark.parametrize("K,M,N", [(1000, 256, 64), (4097, 512, 256), (22, 16384, 256), (333, 128, 1024), (70, 36, 12),
                                   (11264, 256, 1024), (4100, 516, 260), (90112, 256, 64), (1028, 132, 1000),
                                   (22, 256, 256), (7, 100, 36), (32, 16384, 64)])   # short reductions: gemm_shortk_kernel
def test_gemm_tn_vs_fp64(dev, K, M, N):
    from dh3d_amd import pm
    rng = np.random.default_rng(K + M + N)
    A = rng.standard_normal((K, M)).astype(np.float32)
    B = rng.standard_normal((K, N)).astype(np.float32)
    exp = A.astype(np.float64).T @ B.astype(np.float64)
    mag = np.abs(A.astype(np.float64)).T @ np.abs(B.astype(np.float64))
    got = pm.gemm_tn(T(A, dev), T(B, dev)).cpu().numpy()
    assert np.all(np.abs(got - exp) <= 2e-6 * mag + 1e-6), float(np.abs(got - exp).max())
    # accumulate adds to what is there
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    got2 = pm.gemm_tn(T(A, dev), T(B, dev), out=T(C0, dev), accumulate=True).cpu().numpy()
    assert np.all(np.abs(got2 - (exp + C0)) <= 2e-6 * (mag + np.abs(C0)) + 1e-6)


@pytest.mark.parametrize("M,K,N", [(1000, 256, 512), (11264, 256, 512), (130, 1024, 256), (65, 64, 128), (300, 20, 36),
                                   (11264, 1024, 256), (90112, 64, 256), (4099, 260, 68), (70001, 36, 60),
                                   (22, 256, 256), (22, 256, 16384), (5, 36, 100), (32, 512, 68)])   # few rows: gemm_rows_kernel
def test_gemm_nn_vs_fp64(dev, M, K, N):
    from dh3d_amd import pm
    rng = np.random.default_rng(M + K + N)
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((K, N)).astype(np.float32)
    exp = A.astype(np.float64) @ B.astype(np.float64)
    mag = np.abs(A.astype(np.float64)) @ np.abs(B.astype(np.float64))
    got = pm.gemm_nn(T(A, dev), T(B, dev)).cpu().numpy()
    assert np.all(np.abs(got - exp) <= 2e-6 * mag + 1e-6), float(np.abs(got - exp).max())
    bias = rng.standard_normal((N,)).astype(np.float32)
    got = pm.gemm_nn(T(A, dev), T(B, dev), bias=T(bias, dev)).cpu().numpy()
    assert np.all(np.abs(got - (exp + bias)) <= 2e-6 * (mag + np.abs(bias)) + 1e-6)
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    got = pm.gemm_nn(T(A, dev), T(B, dev), out=T(C0, dev), accumulate=True).cpu().numpy()
    assert np.all(np.abs(got - (exp + C0)) <= 2e-6 * (mag + np.abs(C0)) + 1e-6)


def test_gemm_batched_and_bias_vs_fp64(dev):
    """batched forms (+ column bias, + wide-magnitude operands: the three-way bf16 split is exact for every f32) at
    sizes that take the bf16x6 kernel and at sizes that stay on the f32 one."""
    from dh3d_amd import pm
    rng = np.random.default_rng(11)
    for (b, M, K, N) in [(22, 4096, 256, 64), (22, 4096, 64, 256), (3, 130, 36, 20), (5, 1000, 128, 132)]:
        A = (rng.standard_normal((b, M, K)) * np.exp(rng.uniform(-6, 6, (b, M, 1)))).astype(np.float32)
        B = rng.standard_normal((b, K, N)).astype(np.float32)
        bias = rng.standard_normal((b, N)).astype(np.float32)
        A64, B64 = A.astype(np.float64), B.astype(np.float64)
        exp = A64 @ B64 + bias[:, None, :]
        mag = np.abs(A64) @ np.abs(B64) + np.abs(bias[:, None, :])
        got = pm.gemm_nn_batched(T(A, dev), T(B, dev), bias=T(bias, dev)).cpu().numpy()
        assert np.all(np.abs(got - exp) <= 2e-6 * mag + 1e-6), (b, M, K, N, float(np.abs(got - exp).max()))
        # tn: C[b] = A[b]^T D[b],  A [b, M(reduction), K], D [b, M, N]
        D = rng.standard_normal((b, M, N)).astype(np.float32)
        exp = A64.transpose(0, 2, 1) @ D.astype(np.float64)
        mag = np.abs(A64).transpose(0, 2, 1) @ np.abs(D.astype(np.float64))
        got = pm.gemm_tn_batched(T(A, dev), T(D, dev)).cpu().numpy()
        assert np.all(np.abs(got - exp) <= 2e-6 * mag + 1e-6), (b, M, K, N, float(np.abs(got - exp).max()))
    A = rng.standard_normal((3000, 512)).astype(np.float32)
    B = rng.standard_normal((512, 260)).astype(np.float32)
    bias = rng.standard_normal(260).astype(np.float32)
    got = pm.gemm_nn(T(A, dev), T(B, dev), bias=T(bias, dev)).cpu().numpy()
    exp = A.astype(np.float64) @ B.astype(np.float64) + bias
    assert np.all(np.abs(got - exp) <= 2e-6 * (np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64) + 1) + 1e-6)


def test_transpose_and_colsum(dev):
    from dh3d_amd import pm
    rng = np.random.default_rng(3)
    for shape in [(3, 64, 1000), (2, 8, 4097), (1, 3, 77), (2, 129, 65)]:
        x = rng.standard_normal(shape).astype(np.float32)
        assert np.array_equal(pm.transpose_last2(T(x, dev)).cpu().numpy(), x.transpose(0, 2, 1))
    xi = rng.integers(0, 1 << 30, (2, 8, 513), dtype=np.int32)
    assert np.array_equal(pm.transpose_last2(T(xi, dev)).cpu().numpy(), xi.transpose(0, 2, 1))
    x = rng.standard_normal((70001, 100)).astype(np.float32)
    got = pm.colsum(T(x, dev)).cpu().numpy()
    exp = x.astype(np.float64).sum(0)
    assert np.all(np.abs(got - exp) <= 2e-6 * np.abs(x.astype(np.float64)).sum(0))


def _cloud_case(rng, B, N, K, Din, Dout, oracle):
    pos = rng.random((B, 3, N), dtype=np.float32)
    nn, _ = oracle.knn_bruteforce(pos, K)
    return dict(features=rng.standard_normal((B, Din, N)).astype(np.float32), position=pos,
                neighborhood=np.ascontiguousarray(nn.transpose(0, 2, 1)),
                theta=(rng.standard_normal((3, Din, Dout)) / np.sqrt(Din)).astype(np.float32),
                bias=(rng.standard_normal((Din, Dout)) / np.sqrt(8 * Din)).astype(np.float32),
                topdiff=rng.standard_normal((B, Dout, N)).astype(np.float32))


@pytest.mark.parametrize("B,N,K,Din,Dout", [(2, 300, 8, 32, 64), (1, 513, 8, 64, 64), (2, 256, 8, 128, 256), (1, 200, 12, 128, 128),
                                           (2, 100, 5, 16, 24)])
def test_flex_conv_bwd_factorised_vs_oracle(dev, oracle, B, N, K, Din, Dout):
    """pm.flex_conv_bwd (point-major) against the oracle's reference-order gradient; both centre rules coincide under
    exact kNN (rank-0 neighbour == self)."""
    from dh3d_amd import pm
    c = _cloud_case(np.random.default_rng(B * N + Din), B, N, K, Din, Dout, oracle)
    gf, gt, gb = oracle.flex_convolution_grad(c["features"], c["position"], c["neighborhood"], c["theta"], c["bias"],
                                              c["topdiff"])
    pmf = lambda a: T(np.ascontiguousarray(a.transpose(0, 2, 1)), dev)
    for rank0 in (False, True):
        df, dth, dbi = pm.flex_conv_bwd(pmf(c["features"]), pmf(c["position"]), pmf(c["neighborhood"]), T(c["theta"], dev),
                                        T(c["bias"], dev), pmf(c["topdiff"]), center_rank0=rank0)
        for got, exp in ((df.cpu().numpy().transpose(0, 2, 1), gf), (dth.cpu().numpy(), gt), (dbi.cpu().numpy(), gb)):
            tol = 1e-4 * float(np.abs(exp).max())
            assert float(np.abs(got - exp).max()) <= tol, (rank0, float(np.abs(got - exp).max()), tol)
    # weights-only variant
    df, dth2, _ = pm.flex_conv_bwd(pmf(c["features"]), pmf(c["position"]), pmf(c["neighborhood"]), T(c["theta"], dev),
                                   T(c["bias"], dev), pmf(c["topdiff"]), need_grad_features=False)
    assert df is None and torch.allclose(dth2, dth, rtol=1e-4, atol=1e-4 * float(dth.abs().max()))


def test_flex_conv_bwd_rank0_centre_with_foreign_first_neighbour(dev, oracle):
    """Neighbourhoods whose rank-0 entry is NOT the point itself (arbitrary caller-provided lists): the reference
    backward centres on that neighbour (flex_conv_kernel_gpu.cu.cc:196-202,314) and so must the drop-in gradient."""
    from dh3d_amd import ops
    rng = np.random.default_rng(5)
    B, N, K, Din, Dout = 1, 257, 8, 32, 64
    c = _cloud_case(rng, B, N, K, Din, Dout, oracle)
    c["neighborhood"] = rng.integers(0, N, (B, K, N), dtype=np.int32)   # random lists, rank 0 != self
    gf, gt, gb = oracle.flex_convolution_grad(c["features"], c["position"], c["neighborhood"], c["theta"], c["bias"],
                                              c["topdiff"])
    f, th, bi = T(c["features"], dev).requires_grad_(), T(c["theta"], dev).requires_grad_(), T(c["bias"], dev).requires_grad_()
    out = ops.flex_convolution(f, T(c["position"], dev), T(c["neighborhood"], dev), th, bi)
    exp = oracle.flex_convolution(c["features"], c["position"], c["neighborhood"], c["theta"], c["bias"], True)
    assert np.abs(out.detach().cpu().numpy() - exp).max() <= 1e-4 * np.abs(exp).max()   # forward: centre = point n
    out.backward(T(c["topdiff"], dev))
    for got, e in ((f.grad, gf), (th.grad, gt), (bi.grad, gb)):
        assert float(np.abs(got.cpu().numpy() - e).max()) <= 1e-4 * float(np.abs(e).max())


@pytest.mark.parametrize("B,N,K,Din,Dout", [(2, 999, 8, 32, 64), (1, 4100, 8, 64, 64), (2, 512, 8, 64, 128), (1, 700, 8, 128, 256),
                                           (1, 1024, 12, 128, 128)])
def test_ops_fast_path_equals_reference_formulation(dev, oracle, B, N, K, Din, Dout):
    """ops.flex_convolution / flex_pooling through section A' of the ABI (fused MFMA kernels behind the reference
    signatures) against the reference-order kernels of section A and against the oracle, forward and backward."""
    from dh3d_amd import ops
    c = _cloud_case(np.random.default_rng(N + Dout), B, N, K, Din, Dout, oracle)
    args = [T(c[k], dev) for k in ("features", "position", "neighborhood", "theta", "bias")]
    res = {}
    for fast in (True, False):
        ops.FAST_PATH = fast
        try:
            f, th, bi = args[0].clone().requires_grad_(), args[3].clone().requires_grad_(), args[4].clone().requires_grad_()
            out = ops.flex_convolution(f, args[1], args[2], th, bi)
            out.backward(T(c["topdiff"], dev))
            pool, arg = ops.flex_pooling(args[0], args[2])
            res[fast] = [t.detach().cpu().numpy() for t in (out, f.grad, th.grad, bi.grad, pool, arg)]
        finally:
            ops.FAST_PATH = True
    exp = oracle.flex_convolution(c["features"], c["position"], c["neighborhood"], c["theta"], c["bias"], True)
    for fast in (True, False):
        assert np.abs(res[fast][0] - exp).max() <= 1e-5 * np.abs(exp).max(), fast
    for a, b in zip(res[True][:4], res[False][:4]):
        assert float(np.abs(a - b).max()) <= 1e-4 * float(np.abs(b).max())
    assert np.array_equal(res[True][4], res[False][4]) and np.array_equal(res[True][5], res[False][5])  # pool: exact
    ep, ea = oracle.flex_pooling(c["features"], c["neighborhood"])
    assert np.array_equal(res[True][4], ep) and np.array_equal(res[True][5], ea)
