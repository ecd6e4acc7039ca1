"""ctypes front-end of the CPU oracle (oracle/dh3d_oracle.c) and of the reference twins.

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under dh3d_amd/ imports this module.

All functions take / return numpy arrays in the reference op layouts
(user_ops: channels-first [B,C,N]; tf_ops: channels-last [B,N,C]).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF_I = None
_REF_G = None

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.POINTER(ctypes.c_int32)


def build(force=False):
    """Compile the oracle (and, if /root/reference is present, oracle/_ref)."""
    so = os.path.join(_HERE, "libdh3d_oracle.so")
    src = os.path.join(_HERE, "dh3d_oracle.c")
    if force or not os.path.isfile(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libdh3d_oracle.so"])
    subprocess.check_call(["make", "-C", _HERE, "ref"])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libdh3d_oracle.so")
        if not os.path.isfile(so):
            build()
        _LIB = ctypes.CDLL(so)
    return _LIB


def have_ref():
    return os.path.isfile(os.path.join(_HERE, "_ref", "libref_interpolate.so")) and os.path.isfile(
        os.path.join(_HERE, "_ref", "libref_grouping.so"))


def ref_interpolate():
    global _REF_I
    if _REF_I is None:
        _REF_I = ctypes.CDLL(os.path.join(_HERE, "_ref", "libref_interpolate.so"))
    return _REF_I


def ref_grouping():
    global _REF_G
    if _REF_G is None:
        _REF_G = ctypes.CDLL(os.path.join(_HERE, "_ref", "libref_grouping.so"))
    return _REF_G


def _fp(a):
    return a.ctypes.data_as(_f)


def _ip(a):
    return a.ctypes.data_as(_i)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# --------------------------------------------------------------------------- user_ops
def knn_ladder(N):
    t, v = ctypes.c_int(), ctypes.c_int()
    lib().dh3d_oracle_knn_ladder(int(N), ctypes.byref(t), ctypes.byref(v))
    return t.value, v.value


def knn_bruteforce(positions, k):
    """positions [B,Dp,N] -> (nn [B,N,K] int32, dist [B,N,K] f32)."""
    p = _f32(positions)
    B, Dp, N = p.shape
    nn = np.empty((B, N, k), np.int32)
    dist = np.empty((B, N, k), np.float32)
    rc = lib().dh3d_oracle_knn_bruteforce(_fp(p), B, Dp, N, int(k), _ip(nn), _fp(dist))
    if rc:
        raise ValueError("oracle knn_bruteforce: invalid argument")
    return nn, dist


def flex_convolution(features, position, neighborhood, theta, bias, center_self=True):
    f, p, t, bi = _f32(features), _f32(position), _f32(theta), _f32(bias)
    nb = _i32(neighborhood)
    B, Din, N = f.shape
    K = nb.shape[1]
    Dp, _, Dout = t.shape
    out = np.empty((B, Dout, N), np.float32)
    lib().dh3d_oracle_flex_conv_fwd(_fp(f), _fp(t), _fp(bi), _ip(nb), _fp(p), B, N, K, Dp, Din, Dout,
                                    1 if center_self else 0, _fp(out))
    return out


def flex_convolution_grad(features, position, neighborhood, theta, bias, topdiff):
    f, p, t, bi, td = _f32(features), _f32(position), _f32(theta), _f32(bias), _f32(topdiff)
    nb = _i32(neighborhood)
    B, Din, N = f.shape
    K = nb.shape[1]
    Dp, _, Dout = t.shape
    gf = np.empty_like(f)
    gt = np.empty_like(t)
    gb = np.empty_like(bi)
    lib().dh3d_oracle_flex_conv_bwd(_fp(f), _fp(t), _fp(bi), _ip(nb), _fp(p), _fp(td), B, N, K, Dp, Din,
                                    Dout, _fp(gf), _fp(gt), _fp(gb))
    return gf, gt, gb


def flex_pooling(features, neighborhood):
    f = _f32(features)
    nb = _i32(neighborhood)
    B, D, N = f.shape
    K = nb.shape[1]
    out = np.empty_like(f)
    arg = np.empty((B, D, N), np.int32)
    lib().dh3d_oracle_flex_pool_fwd(_fp(f), _ip(nb), B, N, K, D, _fp(out), _ip(arg))
    return out, arg


def flex_pooling_grad(topdiff, argmax):
    td = _f32(topdiff)
    am = _i32(argmax)
    B, D, N = td.shape
    gf = np.empty_like(td)
    lib().dh3d_oracle_flex_pool_bwd(_fp(td), _ip(am), B, N, D, _fp(gf))
    return gf


def convolution_pointset(features, neighborhood, theta, bias):
    f, t, bi = _f32(features), _f32(theta), _f32(bias)
    nb = _i32(neighborhood)
    B, Din, N = f.shape
    K = nb.shape[1]
    Dout = t.shape[1]
    out = np.empty((B, Dout, N), np.float32)
    lib().dh3d_oracle_conv_pointset_fwd(_fp(f), _fp(t), _fp(bi), _ip(nb), B, N, K, Din, Dout, _fp(out))
    return out


def convolution_pointset_grad(features, neighborhood, theta, topdiff):
    f, t, td = _f32(features), _f32(theta), _f32(topdiff)
    nb = _i32(neighborhood)
    B, Din, N = f.shape
    K = nb.shape[1]
    Dout = t.shape[1]
    gf = np.empty_like(f)
    gt = np.empty_like(t)
    gb = np.empty((Dout,), np.float32)
    lib().dh3d_oracle_conv_pointset_bwd(_fp(f), _fp(t), _ip(nb), _fp(td), B, N, K, Din, Dout, _fp(gf),
                                        _fp(gt), _fp(gb))
    return gf, gt, gb


# --------------------------------------------------------------------------- tf_ops
def farthest_point_sample(npoint, inp, contract=True):
    x = _f32(inp)
    B, N, _ = x.shape
    idx = np.empty((B, npoint), np.int32)
    lib().dh3d_oracle_fps(_fp(x), B, N, int(npoint), _ip(idx), 1 if contract else 0)
    return idx


def group_point(points, idx):
    p = _f32(points)
    ix = _i32(idx)
    b, n, c = p.shape
    _, m, ns = ix.shape
    out = np.empty((b, m, ns, c), np.float32)
    lib().dh3d_oracle_group_point_fwd(_fp(p), _ip(ix), b, n, c, m, ns, _fp(out))
    return out


def group_point_grad(points_shape, idx, grad_out):
    ix = _i32(idx)
    go = _f32(grad_out)
    b, n, c = points_shape
    _, m, ns = ix.shape
    gp = np.empty((b, n, c), np.float32)
    lib().dh3d_oracle_group_point_bwd(_fp(go), _ip(ix), b, n, c, m, ns, _fp(gp))
    return gp


def three_nn(xyz1, xyz2):
    a, bq = _f32(xyz1), _f32(xyz2)
    b, n, _ = a.shape
    m = bq.shape[1]
    dist = np.empty((b, n, 3), np.float32)
    idx = np.empty((b, n, 3), np.int32)
    lib().dh3d_oracle_three_nn(_fp(a), _fp(bq), b, n, m, _fp(dist), _ip(idx))
    return dist, idx


def three_interpolate(points, idx, weight):
    p, w = _f32(points), _f32(weight)
    ix = _i32(idx)
    b, m, c = p.shape
    n = ix.shape[1]
    out = np.empty((b, n, c), np.float32)
    lib().dh3d_oracle_three_interpolate_fwd(_fp(p), _ip(ix), _fp(w), b, m, c, n, _fp(out))
    return out


def three_interpolate_grad(points_shape, idx, weight, grad_out):
    w, go = _f32(weight), _f32(grad_out)
    ix = _i32(idx)
    b, m, c = points_shape
    n = ix.shape[1]
    gp = np.empty((b, m, c), np.float32)
    lib().dh3d_oracle_three_interpolate_bwd(_fp(go), _ip(ix), _fp(w), b, n, c, m, _fp(gp))
    return gp


# --------------------------------------------------------------------------- reference twins
def ref_three_nn_origin(xyz2, n):
    """Reference twin threenn_cpu (interpolate.cpp:21-64).  The twin ignores xyz1 (it measures
    |xyz2|^2), i.e. it is tf_interpolate.cpp's three_nn with every query at the origin."""
    bq = _f32(xyz2)
    b, m, _ = bq.shape
    a = np.zeros((b, n, 3), np.float32)
    dist = np.empty((b, n, 3), np.float32)
    idx = np.empty((b, n, 3), np.int32)
    ref_interpolate().ref_threenn_cpu(b, n, m, _fp(a), _fp(bq), _fp(dist), _ip(idx))
    return dist, idx


def ref_three_interpolate(points, idx, weight):
    p, w = _f32(points), _f32(weight)
    ix = _i32(idx)
    b, m, c = p.shape
    n = ix.shape[1]
    out = np.empty((b, n, c), np.float32)
    ref_interpolate().ref_interpolate_cpu(b, m, c, n, _fp(p), _ip(ix), _fp(w), _fp(out))
    return out


def ref_three_interpolate_grad(points_shape, idx, weight, grad_out):
    w, go = _f32(weight), _f32(grad_out)
    ix = _i32(idx)
    b, m, c = points_shape
    n = ix.shape[1]
    gp = np.zeros((b, m, c), np.float32)  # the twin accumulates into caller-zeroed memory
    ref_interpolate().ref_interpolate_grad_cpu(b, n, c, m, _fp(go), _ip(ix), _fp(w), _fp(gp))
    return gp


def ref_group_point(points, idx):
    p = _f32(points)
    ix = _i32(idx)
    b, n, c = p.shape
    _, m, ns = ix.shape
    out = np.empty((b, m, ns, c), np.float32)
    ref_grouping().ref_group_point_cpu(b, n, c, m, ns, _fp(p), _ip(ix), _fp(out))
    return out


def ref_group_point_grad(points_shape, idx, grad_out):
    ix = _i32(idx)
    go = _f32(grad_out)
    b, n, c = points_shape
    _, m, ns = ix.shape
    gp = np.zeros((b, n, c), np.float32)
    ref_grouping().ref_group_point_grad_cpu(b, n, c, m, ns, _fp(go), _ip(ix), _fp(gp))
    return gp
