"""Drop-in operator API: the reference's op names, argument orders and tensor layouts.

Mirrors user_ops/__init__.py (knn_bruteforce :50, flex_convolution :63-89, flex_pooling :115-135,
convolution_pointset :205-225) and tf_ops/{sampling,grouping,interpolation}/tf_*.py
(farthest_point_sample tf_sampling.py:63-71, group_point tf_grouping.py:48-56, three_nn / three_interpolate
tf_interpolate.py:8-34), with torch.autograd.Function standing in for the RegisterGradient hooks
(user_ops/__init__.py:95-111,141-151,231-246; tf_grouping.py:57-61; tf_interpolate.py:29-34).
Where TF raised InvalidArgument these raise ValueError.  Every op runs on the HIP library; there is no
CPU path.
"""
import torch

from . import _lib as L

__all__ = [
    "knn_bruteforce", "flex_convolution", "flex_pooling", "convolution_pointset",
    "farthest_point_sample", "group_point", "three_nn", "three_interpolate",
]


# The reference-layout operators run on the fused MFMA kernels whenever the shape is one they serve (every DH3D
# layer); False forces the reference formulation (csrc/flex_generic.hip) -- the tests compare the two.
FAST_PATH = True


def _same(a, b, what):
    if a != b:
        raise ValueError("%s mismatch: %s vs %s" % (what, a, b))


# --------------------------------------------------------------------------- knn
def knn_bruteforce(positions, k, name=None):
    """positions [B, Dp, N] -> (neighborhood [B, N, K] int32, distances [B, N, K]).

    user_ops/ops/knn_bruteforce.cc:11-35; not differentiable."""
    p = L.require_cuda_f32(positions, "positions", 3)
    if int(k) <= 0:
        raise ValueError("k must be positive")
    B, Dp, N = p.shape
    nn = torch.empty((B, N, int(k)), dtype=torch.int32, device=p.device)
    dist = torch.empty((B, N, int(k)), dtype=torch.float32, device=p.device)
    with torch.cuda.device(p.device):
        L.check(L.lib().dh3d_knn_bruteforce(L.ptr(p), B, Dp, N, int(k), L.ptr(nn), L.ptr(dist),
                                            L.stream_ptr()), "knn_bruteforce")
    return nn, dist


# --------------------------------------------------------------------------- flex_conv
class _FlexConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, theta, bias, neighborhood, position):
        f = L.require_cuda_float(features, "features", 3)
        t = L.require_cuda_float(theta, "theta", 3, like=f)
        bi = L.require_cuda_float(bias, "bias", 2, like=f)
        nb = L.require_cuda_i32(neighborhood, "neighborhood", 3)
        p = L.require_cuda_float(position, "position", 3, like=f)
        f64 = f.dtype == torch.float64   # (flex_conv_op.cc:97-106 registers double too: the reference formulation)
        B, Din, N = f.shape
        Dp, Din_t, Dout = t.shape
        K = nb.shape[1]
        # shape function of user_ops/ops/flex_conv.cc:41-82
        _same(Din_t, Din, "Din(theta/features)")
        _same(tuple(bi.shape), (Din, Dout), "bias shape")
        _same((nb.shape[0], nb.shape[2]), (B, N), "neighborhood [B,_,N]")
        _same(tuple(p.shape), (B, Dp, N), "position shape")
        out = torch.empty((B, Dout, N), dtype=f.dtype, device=f.device)
        with torch.cuda.device(f.device):
            ws_bytes = L.lib().dh3d_flex_conv_fwd_workspace_bytes(B, N, K, Dp, Din, Dout) if FAST_PATH and not f64 else 0
            if f64:
                L.check(L.lib().dh3d_flex_conv_fwd_f64(L.ptr(f), L.ptr(t), L.ptr(bi), L.ptr(nb), L.ptr(p), B, N, K, Dp,
                                                       Din, Dout, L.ptr(out), L.stream_ptr()), "flex_convolution")
            elif ws_bytes:  # the DH3D shapes: fused MFMA kernels behind the reference signature (section A' of the ABI)
                ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=f.device)
                L.check(L.lib().dh3d_flex_conv_fwd_ws(L.ptr(f), L.ptr(t), L.ptr(bi), L.ptr(nb), L.ptr(p), B, N, K, Dp,
                                                      Din, Dout, L.ptr(out), L.ptr(ws), ws_bytes, L.stream_ptr()),
                        "flex_convolution")
            else:       # any other shape: the reference formulation (csrc/flex_generic.hip)
                L.check(L.lib().dh3d_flex_conv_fwd(L.ptr(f), L.ptr(t), L.ptr(bi), L.ptr(nb), L.ptr(p), B, N, K, Dp,
                                                   Din, Dout, L.ptr(out), L.stream_ptr()), "flex_convolution")
        ctx.save_for_backward(f, t, bi, nb, p)
        return out

    @staticmethod
    def backward(ctx, topdiff):
        f, t, bi, nb, p = ctx.saved_tensors
        td = topdiff.contiguous()
        B, Din, N = f.shape
        Dp, _, Dout = t.shape
        K = nb.shape[1]
        gf, gt, gb = torch.empty_like(f), torch.empty_like(t), torch.empty_like(bi)
        with torch.cuda.device(f.device):
            f64 = f.dtype == torch.float64
            ws_bytes = L.lib().dh3d_flex_conv_bwd_workspace_bytes(B, N, K, Dp, Din, Dout) if FAST_PATH and not f64 else 0
            if f64:
                L.check(L.lib().dh3d_flex_conv_bwd_f64(L.ptr(f), L.ptr(t), L.ptr(bi), L.ptr(nb), L.ptr(p), L.ptr(td), B,
                                                       N, K, Dp, Din, Dout, L.ptr(gf), L.ptr(gt), L.ptr(gb),
                                                       L.stream_ptr()), "flex_convolution_grad")
            elif ws_bytes:  # factorised backward on the MFMA pipe + atomics scatter (csrc/flex_bwd.hip)
                ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=f.device)
                L.check(L.lib().dh3d_flex_conv_bwd_ws(L.ptr(f), L.ptr(t), L.ptr(bi), L.ptr(nb), L.ptr(p), L.ptr(td), B,
                                                      N, K, Dp, Din, Dout, L.ptr(gf), L.ptr(gt), L.ptr(gb), L.ptr(ws),
                                                      ws_bytes, L.stream_ptr()), "flex_convolution_grad")
            else:
                L.check(L.lib().dh3d_flex_conv_bwd(L.ptr(f), L.ptr(t), L.ptr(bi), L.ptr(nb), L.ptr(p), L.ptr(td), B,
                                                   N, K, Dp, Din, Dout, L.ptr(gf), L.ptr(gt), L.ptr(gb),
                                                   L.stream_ptr()), "flex_convolution_grad")
        return gf, gt, gb, None, None


def flex_convolution(features, position, neighborhood, theta, bias, name=None):
    """features [B,Din,N], position [B,Dp,N], neighborhood [B,K,N] int32, theta [Dp,Din,Dout],
    bias [Din,Dout] -> [B,Dout,N]   (user_ops/__init__.py:63-89; note its argument re-order)."""
    return _FlexConv.apply(features, theta, bias, neighborhood, position)


# --------------------------------------------------------------------------- flex_pool
class _FlexPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, neighborhood):
        f = L.require_cuda_float(features, "features", 3)
        nb = L.require_cuda_i32(neighborhood, "neighborhood", 3)
        f64 = f.dtype == torch.float64
        B, D, N = f.shape
        K = nb.shape[1]
        _same((nb.shape[0], nb.shape[2]), (B, N), "neighborhood [B,_,N]")  # ops/flex_pool.cc:35-56
        out = torch.empty_like(f)
        argmax = torch.empty((B, D, N), dtype=torch.int32, device=f.device)
        with torch.cuda.device(f.device):
            ws_bytes = L.lib().dh3d_flex_pool_fwd_workspace_bytes(B, N, K, D) if FAST_PATH and not f64 else 0
            if f64:
                L.check(L.lib().dh3d_flex_pool_fwd_f64(L.ptr(f), L.ptr(nb), B, N, K, D, L.ptr(out), L.ptr(argmax),
                                                       L.stream_ptr()), "flex_pooling")
            elif ws_bytes:
                ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=f.device)
                L.check(L.lib().dh3d_flex_pool_fwd_ws(L.ptr(f), L.ptr(nb), B, N, K, D, L.ptr(out), L.ptr(argmax),
                                                      L.ptr(ws), ws_bytes, L.stream_ptr()), "flex_pooling")
            else:
                L.check(L.lib().dh3d_flex_pool_fwd(L.ptr(f), L.ptr(nb), B, N, K, D, L.ptr(out), L.ptr(argmax),
                                                   L.stream_ptr()), "flex_pooling")
        ctx.save_for_backward(argmax)
        ctx.mark_non_differentiable(argmax)
        return out, argmax

    @staticmethod
    def backward(ctx, topdiff, _unused):
        (argmax,) = ctx.saved_tensors
        td = topdiff.contiguous()
        B, D, N = td.shape
        gf = torch.empty_like(td)
        with torch.cuda.device(td.device):
            fn = L.lib().dh3d_flex_pool_bwd_f64 if td.dtype == torch.float64 else L.lib().dh3d_flex_pool_bwd
            L.check(fn(L.ptr(td), L.ptr(argmax), B, N, D, L.ptr(gf), L.stream_ptr()), "flex_pooling_grad")
        return gf, None


def flex_pooling(features, neighborhood, name=None):
    """features [B,D,N], neighborhood [B,K,N] -> (max values [B,D,N], argmax point ids [B,D,N] int32)
    (user_ops/__init__.py:115-135)."""
    return _FlexPool.apply(features, neighborhood)


# --------------------------------------------------------------------------- conv_pointset
class _ConvPointset(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, theta, bias, neighborhood):
        f = L.require_cuda_float(features, "features", 3)
        t = L.require_cuda_float(theta, "theta", 2, like=f)
        bi = L.require_cuda_float(bias, "bias", 1, like=f)
        nb = L.require_cuda_i32(neighborhood, "neighborhood", 3)
        B, Din, N = f.shape
        Din_t, Dout = t.shape
        K = nb.shape[1]
        _same(Din_t, Din, "Din(theta/features)")  # ops/conv_pointset.cc:38-73
        _same(bi.shape[0], Dout, "bias length")
        _same((nb.shape[0], nb.shape[2]), (B, N), "neighborhood [B,_,N]")
        out = torch.empty((B, Dout, N), dtype=f.dtype, device=f.device)
        with torch.cuda.device(f.device):
            fn = L.lib().dh3d_conv_pointset_fwd_f64 if f.dtype == torch.float64 else L.lib().dh3d_conv_pointset_fwd
            L.check(fn(L.ptr(f), L.ptr(t), L.ptr(bi), L.ptr(nb), B, N, K, Din, Dout, L.ptr(out), L.stream_ptr()),
                    "convolution_pointset")
        ctx.save_for_backward(f, t, nb)
        return out

    @staticmethod
    def backward(ctx, topdiff):
        f, t, nb = ctx.saved_tensors
        td = topdiff.contiguous()
        B, Din, N = f.shape
        Dout = t.shape[1]
        K = nb.shape[1]
        gf, gt = torch.empty_like(f), torch.empty_like(t)
        gb = torch.empty((Dout,), dtype=f.dtype, device=f.device)
        with torch.cuda.device(f.device):
            fn = L.lib().dh3d_conv_pointset_bwd_f64 if f.dtype == torch.float64 else L.lib().dh3d_conv_pointset_bwd
            L.check(fn(L.ptr(f), L.ptr(t), L.ptr(nb), L.ptr(td), B, N, K, Din, Dout, L.ptr(gf), L.ptr(gt), L.ptr(gb),
                       L.stream_ptr()), "convolution_pointset_grad")
        return gf, gt, gb, None


def convolution_pointset(features, neighborhood, theta, bias, name=None):
    """features [B,Din,N], neighborhood [B,K,N], theta [Din,Dout], bias [Dout] -> [B,Dout,N]
    (user_ops/__init__.py:205-225)."""
    return _ConvPointset.apply(features, theta, bias, neighborhood)


# --------------------------------------------------------------------------- PointNet++ ops
def farthest_point_sample(npoint, inp, contract=None):
    """inp [B,N,3] -> idx [B,npoint] int32 (tf_sampling.py:63-71); not differentiable.

    contract (extension): None = the default kernels (distance rounded as fma(dz,dz,fma(dx,dx,dy*dy)), the LLVM/NVVM
    contraction of tf_sampling_g.cu:141); 1 / 0 = the any-N kernel with that contraction / the uncontracted
    (dx*dx+dy*dy)+dz*dz of an -fmad=false build (dh3d_farthest_point_sample_mode)."""
    x = L.require_cuda_f32(inp, "inp", 3)
    if x.shape[2] != 3:
        raise ValueError("FarthestPointSample expects (batch_size,num_points,3) inp shape")  # tf_sampling.cpp:105
    if int(npoint) <= 0:
        raise ValueError("FarthestPointSample expects positive npoint")  # tf_sampling.cpp:100
    B, N, _ = x.shape
    out = torch.empty((B, int(npoint)), dtype=torch.int32, device=x.device)
    with torch.cuda.device(x.device):
        if contract is None and N <= 16384:
            L.check(L.lib().dh3d_farthest_point_sample(B, N, int(npoint), L.ptr(x), None, L.ptr(out),
                                                       L.stream_ptr()), "farthest_point_sample")
        else:  # running min-distances in scratch, as the reference's allocate_temp (tf_sampling.cpp:115)
            temp = torch.empty((B, N), dtype=torch.float32, device=x.device)
            L.check(L.lib().dh3d_farthest_point_sample_mode(B, N, int(npoint), L.ptr(x), L.ptr(temp), L.ptr(out),
                                                            1 if contract is None else int(bool(contract)),
                                                            L.stream_ptr()), "farthest_point_sample")
    return out


class _GroupPoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx):
        p = L.require_cuda_f32(points, "points", 3)
        ix = L.require_cuda_i32(idx, "idx", 3)
        b, n, c = p.shape
        _same(ix.shape[0], b, "batch(points/idx)")  # tf_grouping.cpp OP_REQUIRES
        m, ns = ix.shape[1], ix.shape[2]
        out = torch.empty((b, m, ns, c), dtype=torch.float32, device=p.device)
        with torch.cuda.device(p.device):
            L.check(L.lib().dh3d_group_point_fwd(b, n, c, m, ns, L.ptr(p), L.ptr(ix), L.ptr(out), L.stream_ptr()),
                    "group_point")
        ctx.save_for_backward(ix)
        ctx.pshape = (b, n, c)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (ix,) = ctx.saved_tensors
        b, n, c = ctx.pshape
        m, ns = ix.shape[1], ix.shape[2]
        go = grad_out.contiguous()
        gp = torch.empty((b, n, c), dtype=torch.float32, device=go.device)
        with torch.cuda.device(go.device):
            L.check(L.lib().dh3d_group_point_bwd(b, n, c, m, ns, L.ptr(go), L.ptr(ix), L.ptr(gp), L.stream_ptr()),
                    "group_point_grad")
        return gp, None


def group_point(points, idx):
    """points [b,n,c], idx [b,m,nsample] int32 -> [b,m,nsample,c] (tf_grouping.py:48-61)."""
    return _GroupPoint.apply(points, idx)


def three_nn(xyz1, xyz2):
    """xyz1 [b,n,3], xyz2 [b,m,3] -> (dist [b,n,3] squared, idx [b,n,3] int32) (tf_interpolate.py:8-18)."""
    a = L.require_cuda_f32(xyz1, "xyz1", 3)
    q = L.require_cuda_f32(xyz2, "xyz2", 3)
    if a.shape[2] != 3 or q.shape[2] != 3:
        raise ValueError("ThreeNN expects (b,n,3) xyz1 and (b,m,3) xyz2")  # tf_interpolate.cpp:163,168
    _same(q.shape[0], a.shape[0], "batch(xyz1/xyz2)")
    b, n, _ = a.shape
    m = q.shape[1]
    dist = torch.empty((b, n, 3), dtype=torch.float32, device=a.device)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=a.device)
    with torch.cuda.device(a.device):
        L.check(L.lib().dh3d_three_nn(b, n, m, L.ptr(a), L.ptr(q), L.ptr(dist), L.ptr(idx), L.stream_ptr()),
                "three_nn")
    return dist, idx


class _ThreeInterpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx, weight):
        p = L.require_cuda_f32(points, "points", 3)
        ix = L.require_cuda_i32(idx, "idx", 3)
        w = L.require_cuda_f32(weight, "weight", 3)
        b, m, c = p.shape
        n = ix.shape[1]
        _same(tuple(ix.shape), (b, n, 3), "idx shape")  # tf_interpolate.cpp:197-206
        _same(tuple(w.shape), (b, n, 3), "weight shape")
        out = torch.empty((b, n, c), dtype=torch.float32, device=p.device)
        with torch.cuda.device(p.device):
            L.check(L.lib().dh3d_three_interpolate_fwd(b, m, c, n, L.ptr(p), L.ptr(ix), L.ptr(w), L.ptr(out),
                                                       L.stream_ptr()), "three_interpolate")
        ctx.save_for_backward(ix, w)
        ctx.pshape = (b, m, c)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        ix, w = ctx.saved_tensors
        b, m, c = ctx.pshape
        n = ix.shape[1]
        go = grad_out.contiguous()
        gp = torch.empty((b, m, c), dtype=torch.float32, device=go.device)
        with torch.cuda.device(go.device):
            L.check(L.lib().dh3d_three_interpolate_bwd(b, n, c, m, L.ptr(go), L.ptr(ix), L.ptr(w), L.ptr(gp),
                                                       L.stream_ptr()), "three_interpolate_grad")
        return gp, None, None


def three_interpolate(points, idx, weight):
    """points [b,m,c], idx [b,n,3], weight [b,n,3] -> [b,n,c] (tf_interpolate.py:19-34)."""
    return _ThreeInterpolate.apply(points, idx, weight)
