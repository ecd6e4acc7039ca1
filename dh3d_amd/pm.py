"""Point-major ([B,N,C]) fused kernels used by the model path (section B of include/dh3d_hip.h).

Thin, allocation-only wrappers: each function validates shapes, allocates the output with torch and
enqueues one C-ABI call on the current stream.  No function here has a CPU or eager-torch fallback.
"""
import os

import torch

from . import _lib as L

ACT_NONE, ACT_RELU, ACT_SIGMOID = L.ACT_NONE, L.ACT_RELU, L.ACT_SIGMOID


def _ep(pre_bias, scale, shift, act):
    return L.make_epilogue(pre_bias, scale, shift, act)


def knn_xyz(xyz, k):
    """xyz [B,N,3] -> (nbr [B,N,K] int32, dist [B,N,K]); same function as ops.knn_bruteforce."""
    x = L.require_cuda_f32(xyz, "xyz", 3)
    B, N, _ = x.shape
    nn = torch.empty((B, N, k), dtype=torch.int32, device=x.device)
    dist = torch.empty((B, N, k), dtype=torch.float32, device=x.device)
    L.check(L.lib().dh3d_knn_bruteforce_xyz(L.ptr(x), B, N, k, L.ptr(nn), L.ptr(dist), L.stream_ptr()), "knn_xyz")
    return nn, dist


def spatial_sort(xyz):
    """xyz [B,N,3] -> (sorted [B,N,4] Morton-ordered records (x,y,z,bits(orig idx)), gbox [B,ceil(N/64),8])."""
    x = L.require_cuda_f32(xyz, "xyz", 3)
    B, N, _ = x.shape
    srt = torch.empty((B, N, 4), dtype=torch.float32, device=x.device)
    gbox = torch.empty((B, (N + 63) // 64, 8), dtype=torch.float32, device=x.device)
    L.check(L.lib().dh3d_spatial_sort(L.ptr(x), B, N, L.ptr(srt), L.ptr(gbox), L.stream_ptr()), "spatial_sort")
    return srt, gbox


CELL_INTS = 4112  # include/dh3d_hip.h DH3D_CELL_INTS


# dev A/B switch (DH3D_KNN_GRID=0: the Morton-pruned shared scan / brute force for every K); identical ids either way
KNN_GRID = os.environ.get("DH3D_KNN_GRID", "1") != "0"


def spatial_sort_cells(xyz):
    """spatial_sort + the 16^3 cell table of every cloud: (sorted, gbox, cells [B, CELL_INTS] int32)."""
    x = L.require_cuda_f32(xyz, "xyz", 3)
    B, N, _ = x.shape
    srt = torch.empty((B, N, 4), dtype=torch.float32, device=x.device)
    gbox = torch.empty((B, (N + 63) // 64, 8), dtype=torch.float32, device=x.device)
    cells = torch.empty((B, CELL_INTS), dtype=torch.int32, device=x.device)
    L.check(L.lib().dh3d_spatial_sort_cells(L.ptr(x), B, N, L.ptr(srt), L.ptr(gbox), L.ptr(cells), L.stream_ptr()),
            "spatial_sort_cells")
    return srt, gbox, cells


def knn_grid(srt, gbox, cells, k):
    """kNN by cell lists on the three outputs of spatial_sort_cells(); same (nbr [B,N,K], dist) as knn_xyz bit for bit.
    K <= 8."""
    B, N, _ = srt.shape
    nn = torch.empty((B, N, k), dtype=torch.int32, device=srt.device)
    dist = torch.empty((B, N, k), dtype=torch.float32, device=srt.device)
    L.check(L.lib().dh3d_knn_grid(L.ptr(srt), L.ptr(gbox), L.ptr(cells), B, N, k, L.ptr(nn), L.ptr(dist), L.stream_ptr()), "knn_grid")
    return nn, dist


def knn_sorted(srt, gbox, k):
    """kNN from spatial_sort() output; same (nbr [B,N,K], dist) as knn_xyz, original indexing."""
    B, N, _ = srt.shape
    nn = torch.empty((B, N, k), dtype=torch.int32, device=srt.device)
    dist = torch.empty((B, N, k), dtype=torch.float32, device=srt.device)
    L.check(L.lib().dh3d_knn_sorted(L.ptr(srt), L.ptr(gbox), B, N, k, L.ptr(nn), L.ptr(dist), L.stream_ptr()),
            "knn_sorted")
    return nn, dist


def fps_sorted(srt, gbox, npoint, with_xyz=False, xyz=None):
    """FPS from spatial_sort() output; same idx [B,npoint] (original indexing) as ops.farthest_point_sample.
    with_xyz: also the sampled coordinates [B,npoint,3], written by the same kernel.
    xyz: the cloud the records came from -- required above 12288 points (no room for the LDS coordinate table)."""
    B, N, _ = srt.shape
    out = torch.empty((B, npoint), dtype=torch.int32, device=srt.device)
    xyz_s = torch.empty((B, npoint, 3), dtype=torch.float32, device=srt.device) if with_xyz else None
    if xyz is not None:
        x = L.require_cuda_f32(xyz, "xyz", 3)
        L.check(L.lib().dh3d_fps_sorted_cloud(L.ptr(srt), L.ptr(gbox), L.ptr(x), B, N, npoint, L.ptr(out), L.ptr(xyz_s),
                                              L.stream_ptr()), "fps_sorted_cloud")
    elif with_xyz:
        L.check(L.lib().dh3d_fps_sorted_xyz(L.ptr(srt), L.ptr(gbox), B, N, npoint, L.ptr(out), L.ptr(xyz_s),
                                            L.stream_ptr()), "fps_sorted_xyz")
    else:
        L.check(L.lib().dh3d_fps_sorted(L.ptr(srt), L.ptr(gbox), B, N, npoint, L.ptr(out), L.stream_ptr()),
                "fps_sorted")
    return (out, xyz_s) if with_xyz else out


def fps_sorted_ordered(srt, gbox, npoint, cells=None):
    """fps_sorted(with_xyz=True) + the sampled set in the cloud's Morton order out of the same launch: returns
    (idx [B,m], xyz_s [B,m,3], srt_s [B,m,4], gbox_s [B,ceil(m/64),8], cells_s [B,CELL_INTS] or None) -- srt_s / gbox_s /
    cells_s are what spatial_sort_cells(xyz_s) would hand to three_nn_sorted / knn_grid (cells_s only when the cloud's
    own table `cells` is given).  N <= 8192."""
    B, N, _ = srt.shape
    dev = srt.device
    out = torch.empty((B, npoint), dtype=torch.int32, device=dev)
    xyz_s = torch.empty((B, npoint, 3), dtype=torch.float32, device=dev)
    srt_s = torch.empty((B, npoint, 4), dtype=torch.float32, device=dev)
    gbox_s = torch.empty((B, (npoint + 63) // 64, 8), dtype=torch.float32, device=dev)
    cells_s = torch.empty((B, CELL_INTS), dtype=torch.int32, device=dev) if cells is not None else None
    L.check(L.lib().dh3d_fps_sorted_ordered(L.ptr(srt), L.ptr(gbox), L.ptr(cells), B, N, npoint, L.ptr(out), L.ptr(xyz_s),
                                            L.ptr(srt_s), L.ptr(gbox_s), L.ptr(cells_s), L.stream_ptr()), "fps_sorted_ordered")
    return out, xyz_s, srt_s, gbox_s, cells_s


def three_nn_sorted(srt1, gbox1, srt2, gbox2):
    """three_nn from spatial_sort() outputs of the query cloud and of the candidate set: same (dist [B,n,3] squared,
    idx [B,n,3]) as ops.three_nn, original indexing on both sides."""
    B, n, _ = srt1.shape
    m = srt2.shape[1]
    d3 = torch.empty((B, n, 3), dtype=torch.float32, device=srt1.device)
    i3 = torch.empty((B, n, 3), dtype=torch.int32, device=srt1.device)
    L.check(L.lib().dh3d_three_nn_sorted(B, n, m, L.ptr(srt1), L.ptr(gbox1), L.ptr(srt2), L.ptr(gbox2), L.ptr(d3),
                                         L.ptr(i3), L.stream_ptr()), "three_nn_sorted")
    return d3, i3


def pack_weight(W):
    """W [Kd, Dout] row-major -> MFMA fragment order (see mfma_gemm.h)."""
    W = L.require_cuda_f32(W, "W", 2)
    out = torch.empty_like(W)
    L.check(L.lib().dh3d_pack_weight(L.ptr(W), W.shape[0], W.shape[1], L.ptr(out), L.stream_ptr()), "pack_weight")
    return out


def pack_flex_weight(theta, bias):
    """theta [3,Din,Dout], bias [Din,Dout] -> packed [4*Din, Dout] = [bias; theta_x; theta_y; theta_z]."""
    t = L.require_cuda_f32(theta, "theta", 3)
    b = L.require_cuda_f32(bias, "bias", 2)
    if t.shape[0] != 3:
        raise ValueError("point-major flex_conv needs Dp == 3")
    Din, Dout = b.shape
    out = torch.empty((4 * Din, Dout), dtype=torch.float32, device=t.device)
    L.check(L.lib().dh3d_pack_flex_weight(L.ptr(t), L.ptr(b), Din, Dout, L.ptr(out), L.stream_ptr()),
            "pack_flex_weight")
    return out


def flex_conv(features, xyz, nbr, wpacked, Dout, pre_bias=None, scale=None, shift=None, act=ACT_NONE, remap=None):
    """Fused factorised flex_conv (exact-f32 MFMA).  remap [B,N] int32 (optional): `features` is the [B,Nsrc,Din] map of
    the level above and point j of this level is its row remap[b,j] -- group_point fused into the neighbour gather."""
    f = L.require_cuda_f32(features, "features", 3)
    x = L.require_cuda_f32(xyz, "xyz", 3)
    nb = L.require_cuda_i32(nbr, "nbr", 3)
    B, N = x.shape[0], x.shape[1]
    Din = f.shape[2]
    K = nb.shape[2]
    rm = None
    if remap is not None:
        rm = L.require_cuda_i32(remap, "remap", 2)
        if tuple(rm.shape) != (B, N) or f.shape[0] != B:
            raise ValueError("flex_conv: remap must be [B,N] and features [B,Nsrc,Din]")
    elif tuple(f.shape[:2]) != (B, N):
        raise ValueError("flex_conv: xyz/nbr do not match features [B,N,*]")
    if tuple(x.shape) != (B, N, 3) or tuple(nb.shape[:2]) != (B, N):
        raise ValueError("flex_conv: xyz/nbr do not match [B,N,*]")
    out = torch.empty((B, N, Dout), dtype=torch.float32, device=f.device)
    ep = _ep(pre_bias, scale, shift, act)
    L.check(L.lib().dh3d_flex_conv_pm_gather_fwd(L.ptr(f), L.ptr(rm), f.shape[1], L.ptr(x), L.ptr(nb), L.ptr(wpacked), B,
                                                 N, K, Din, Dout, ep, L.ptr(out), L.stream_ptr()), "flex_conv_pm")
    return out


def flex_conv_post(features, xyz, nbr, wpacked, Dout, wpost_packed, Dpost, pre_bias=None, scale=None, shift=None,
                   act=ACT_NONE):
    """flex_conv() and, from the same launch, out2 = out @ Wpost (a packed [Dout, Dpost] weight, no bias / activation).
    Returns (out [B,N,Dout], out2 [B,N,Dpost]).  128 -> 256 with Dpost = 64, K = 8 (the global path's sampled level)."""
    f = L.require_cuda_f32(features, "features", 3)
    x = L.require_cuda_f32(xyz, "xyz", 3)
    nb = L.require_cuda_i32(nbr, "nbr", 3)
    B, N, Din = f.shape
    if tuple(x.shape) != (B, N, 3) or tuple(nb.shape[:2]) != (B, N):
        raise ValueError("flex_conv_post: xyz/nbr do not match features [B,N,*]")
    out = torch.empty((B, N, Dout), dtype=torch.float32, device=f.device)
    out2 = torch.empty((B, N, Dpost), dtype=torch.float32, device=f.device)
    ep = _ep(pre_bias, scale, shift, act)
    L.check(L.lib().dh3d_flex_conv_pm_post_fwd(L.ptr(f), L.ptr(x), L.ptr(nb), L.ptr(wpacked), B, N, nb.shape[2], Din, Dout, ep,
                                               L.ptr(out), L.ptr(wpost_packed), Dpost, L.ptr(out2), L.stream_ptr()),
            "flex_conv_pm_post")
    return out, out2


def flex_tile_x6_supported(Din, Dout, K):
    """Shapes served by the 32-point-tile bf16x6 flex_conv (csrc/flex_tx6.hip): the sampled levels, cfg 5's K = 12."""
    return (Din, Dout, K) in ((64, 128, 8), (128, 128, 8), (128, 256, 8), (128, 128, 12))


def flex_conv_tile_x6(features, xyz, nbr, wpacked_x3, Dout, pre_bias=None, scale=None, shift=None, act=ACT_NONE,
                      wpost_packed=None, Dpost=0):
    """flex_conv on 32-point tiles, tile GEMM on the bf16 matrix pipe at f32 accuracy.  With wpost_packed (a packed
    [Dout, 64] weight): returns (out, out @ Wpost) from the same launch, like flex_conv_post."""
    f = L.require_cuda_f32(features, "features", 3)
    x = L.require_cuda_f32(xyz, "xyz", 3)
    nb = L.require_cuda_i32(nbr, "nbr", 3)
    B, N, Din = f.shape
    if tuple(x.shape) != (B, N, 3) or tuple(nb.shape[:2]) != (B, N):
        raise ValueError("flex_conv_tile_x6: xyz/nbr do not match features [B,N,*]")
    out = torch.empty((B, N, Dout), dtype=torch.float32, device=f.device)
    out2 = torch.empty((B, N, Dpost), dtype=torch.float32, device=f.device) if wpost_packed is not None else None
    ep = _ep(pre_bias, scale, shift, act)
    L.check(L.lib().dh3d_flex_conv_pm_tile_x6_fwd(L.ptr(f), L.ptr(x), L.ptr(nb), L.ptr(wpacked_x3), B, N, nb.shape[2], Din,
                                                  Dout, ep, L.ptr(out), L.ptr(wpost_packed), int(Dpost), L.ptr(out2),
                                                  L.stream_ptr()), "flex_conv_pm_tile_x6")
    return out if wpost_packed is None else (out, out2)


def flex_post_supported(Din, Dout, K, Dpost):
    return Din == 128 and Dout == 256 and K == 8 and Dpost == 64


def flex_x6_supported(Din, Dout, K):
    """Shapes served by the persistent bf16x6 flex_conv (csrc/flex_x6.hip)."""
    return K == 8 and (Din, Dout) in ((32, 64), (64, 64))


def pack_flex_weight_x3(theta, bias):
    """[bias; theta_x; theta_y; theta_z] as three exact bf16 chunk planes in MFMA fragment order."""
    t = L.require_cuda_f32(theta, "theta", 3)
    b = L.require_cuda_f32(bias, "bias", 2)
    if t.shape[0] != 3:
        raise ValueError("point-major flex_conv needs Dp == 3")
    Din, Dout = b.shape
    out = torch.empty((3 * 4 * Din * Dout,), dtype=torch.int16, device=t.device)
    L.check(L.lib().dh3d_pack_flex_weight_x3(L.ptr(t), L.ptr(b), Din, Dout, L.ptr(out), L.stream_ptr()),
            "pack_flex_weight_x3")
    return out


def flex_conv_x6(features, xyz, nbr, wpacked_x3, Dout, pre_bias=None, scale=None, shift=None, act=ACT_NONE,
                 reserve_cus_per_xcd=0):
    """flex_conv on the bf16 matrix pipe at f32 accuracy (full-resolution layers, K == 8).
    reserve_cus_per_xcd: placement hint -- CUs per XCD another stream's kernel holds (speed only, see the header)."""
    f = L.require_cuda_f32(features, "features", 3)
    x = L.require_cuda_f32(xyz, "xyz", 3)
    nb = L.require_cuda_i32(nbr, "nbr", 3)
    B, N, Din = f.shape
    K = nb.shape[2]
    if tuple(x.shape) != (B, N, 3) or tuple(nb.shape[:2]) != (B, N):
        raise ValueError("flex_conv_x6: xyz/nbr do not match features [B,N,*]")
    out = torch.empty((B, N, Dout), dtype=torch.float32, device=f.device)
    ep = _ep(pre_bias, scale, shift, act)
    L.check(L.lib().dh3d_flex_conv_pm_x6_fwd_r(L.ptr(f), L.ptr(x), L.ptr(nb), L.ptr(wpacked_x3), B, N, K, Din, Dout, ep,
                                               int(reserve_cus_per_xcd), L.ptr(out), L.stream_ptr()), "flex_conv_pm_x6")
    return out


def flex_pool(features, nbr, want_argmax=False):
    f = L.require_cuda_f32(features, "features", 3)
    nb = L.require_cuda_i32(nbr, "nbr", 3)
    B, N, C = f.shape
    K = nb.shape[2]
    out = torch.empty_like(f)
    argmax = torch.empty((B, N, C), dtype=torch.int32, device=f.device) if want_argmax else None
    L.check(L.lib().dh3d_flex_pool_pm_fwd(L.ptr(f), L.ptr(nb), B, N, K, C, L.ptr(out), L.ptr(argmax),
                                          L.stream_ptr()), "flex_pool_pm")
    return (out, argmax) if want_argmax else out


def flex_avg(features, nbr, scale=1.0):
    """Flex_Avg (core/layers.py:342-436): scale * sum_k features[nbr[n,k]]; backbones.py:80-82 uses scale = 1/knn."""
    f = L.require_cuda_f32(features, "features", 3)
    nb = L.require_cuda_i32(nbr, "nbr", 3)
    B, N, C = f.shape
    out = torch.empty_like(f)
    L.check(L.lib().dh3d_flex_avg_pm_fwd(L.ptr(f), L.ptr(nb), B, N, nb.shape[2], C, float(scale), L.ptr(out),
                                         L.stream_ptr()), "flex_avg_pm")
    return out


def conv_pointset_xyz(xyz, nbr, theta, bias, pre_bias=None, scale=None, shift=None, act=ACT_NONE):
    x = L.require_cuda_f32(xyz, "xyz", 3)
    nb = L.require_cuda_i32(nbr, "nbr", 3)
    B, N, _ = x.shape
    K = nb.shape[2]
    Dout = theta.shape[1]
    out = torch.empty((B, N, Dout), dtype=torch.float32, device=x.device)
    ep = _ep(pre_bias, scale, shift, act)
    L.check(L.lib().dh3d_conv_pointset_pm_fwd(L.ptr(x), L.ptr(nb), L.ptr(theta), L.ptr(bias), B, N, K, Dout, ep,
                                              L.ptr(out), L.stream_ptr()), "conv_pointset_pm")
    return out


def conv_pointset_pool_xyz(xyz, nbr, theta, bias, pre_bias=None, scale=None, shift=None, act=ACT_NONE):
    """flex_pool(epilogue(conv_pointset(xyz))) over the same neighbourhoods [B,N,8] in two small launches; the
    [B,N,Dout] map between the two operators is not materialised (core/backbones.py:107-110)."""
    x = L.require_cuda_f32(xyz, "xyz", 3)
    nb = L.require_cuda_i32(nbr, "nbr", 3)
    B, N, _ = x.shape
    Dout = theta.shape[1]
    out = torch.empty((B, N, Dout), dtype=torch.float32, device=x.device)
    scratch = torch.empty((B, N, 4), dtype=torch.float32, device=x.device)
    ep = _ep(pre_bias, scale, shift, act)
    L.check(L.lib().dh3d_conv_pointset_pool_pm_fwd(L.ptr(x), L.ptr(nb), L.ptr(theta), L.ptr(bias), B, N, nb.shape[2], Dout,
                                                   ep, L.ptr(scratch), L.ptr(out), L.stream_ptr()), "conv_pointset_pool_pm")
    return out


def linear(x1, wpacked, Dout, x2=None, pre_bias=None, scale=None, shift=None, act=ACT_NONE, residual=None):
    """out = epilogue([x1 | x2] @ W) (+ residual); x* are [..., C] with identical leading dims."""
    a = L.require_cuda_f32(x1, "x1")
    lead = a.shape[:-1]
    C1 = a.shape[-1]
    R = a.numel() // C1
    C2 = 0
    b = None
    if x2 is not None:
        b = L.require_cuda_f32(x2, "x2")
        C2 = b.shape[-1]
        if b.shape[:-1] != lead:
            raise ValueError("linear: x1/x2 leading dims differ")
    res = None
    if residual is not None:
        res = L.require_cuda_f32(residual, "residual")
    out = torch.empty(lead + (Dout,), dtype=torch.float32, device=a.device)
    ep = _ep(pre_bias, scale, shift, act)
    L.check(L.lib().dh3d_linear_pm_fwd(L.ptr(a), C1, L.ptr(b), C2, L.ptr(wpacked), R, Dout, ep, L.ptr(res),
                                       L.ptr(out), L.stream_ptr()), "linear_pm")
    return out


def se_res(x, pool, W1, b1, W2, b2):
    a = L.require_cuda_f32(x, "x")
    p = L.require_cuda_f32(pool, "pool")
    C = a.shape[-1]
    R = a.numel() // C
    out = torch.empty_like(a)
    L.check(L.lib().dh3d_se_res_pm_fwd(L.ptr(a), L.ptr(p), L.ptr(W1), L.ptr(b1), L.ptr(W2), L.ptr(b2), R, C,
                                       L.ptr(out), L.stream_ptr()), "se_res_pm")
    return out


def se_res_pack(W1, b1, W2):
    """W1 [C,C/4], b1 [C/4], W2 [C/4,C] -> (w1packed, b1pad, w2packed) for se_res_packed (zero padding to 32)."""
    C, Cq = W1.shape
    W1p = torch.zeros((C, 32), dtype=torch.float32, device=W1.device)
    W1p[:, :Cq] = W1
    b1p = torch.zeros((32,), dtype=torch.float32, device=W1.device)
    b1p[:Cq] = b1
    W2p = torch.zeros((32, C), dtype=torch.float32, device=W1.device)
    W2p[:Cq] = W2
    return pack_weight(W1p), b1p, pack_weight(W2p)


def se_res_packed(x, pool, w1packed, b1pad, w2packed, b2):
    """se_res on the matrix pipe (csrc/dense.hip se_res_mfma_kernel)."""
    a = L.require_cuda_f32(x, "x")
    p = L.require_cuda_f32(pool, "pool")
    C = a.shape[-1]
    R = a.numel() // C
    out = torch.empty_like(a)
    L.check(L.lib().dh3d_se_res_pm_packed_fwd(L.ptr(a), L.ptr(p), L.ptr(w1packed), L.ptr(b1pad), L.ptr(w2packed),
                                              L.ptr(b2), R, C, L.ptr(out), L.stream_ptr()), "se_res_pm_packed")
    return out


def se_res_pool_packed(x, nbr, w1packed, b1pad, w2packed, b2):
    """se_res_packed(x, flex_pool(x, nbr)) in one launch (the pooled rows are formed while staging).  x [B,N,C], nbr
    [B,N,K] int32, C = 64 or 128."""
    a = L.require_cuda_f32(x, "x", 3)
    nb = L.require_cuda_i32(nbr, "nbr", 3)
    B, N, C = a.shape
    out = torch.empty_like(a)
    L.check(L.lib().dh3d_se_res_pool_pm_packed_fwd(L.ptr(a), L.ptr(nb), B, N, nb.shape[2], L.ptr(w1packed), L.ptr(b1pad),
                                                   L.ptr(w2packed), L.ptr(b2), C, L.ptr(out), L.stream_ptr()),
            "se_res_pool_pm_packed")
    return out


def se_res_pool_conv(x, nbr, w1packed, b1pad, w2packed, b2, conv_wp, conv_b, conv_scale, conv_shift, act=ACT_RELU):
    """(y, act(bn(y @ Wconv + b))) with y = se_res_packed(x, flex_pool(x, nbr)), one launch; x [B,N,C], conv C -> C
    (conv_wp = pack_weight of its [C,C] matrix), C = 64 or 128; conv_b / conv_scale / conv_shift may be None."""
    a = L.require_cuda_f32(x, "x", 3)
    nb = L.require_cuda_i32(nbr, "nbr", 3)
    B, N, C = a.shape
    out, out2 = torch.empty_like(a), torch.empty((B, N, C), dtype=torch.float32, device=a.device)
    ep = _ep(conv_b, conv_scale, conv_shift, act)
    L.check(L.lib().dh3d_se_res_pool_conv_pm_fwd(L.ptr(a), L.ptr(nb), B, N, nb.shape[2], L.ptr(w1packed), L.ptr(b1pad),
                                                 L.ptr(w2packed), L.ptr(b2), C, L.ptr(out), L.ptr(conv_wp), ep, C,
                                                 L.ptr(out2), L.stream_ptr()), "se_res_pool_conv_pm")
    return out, out2


def se_res_pool_conv_tails(x, nbr, w1packed, b1pad, w2packed, b2, conv_wp, conv_b, conv_scale, conv_shift, tail_a, tail_b,
                           act=ACT_RELU, store_y=True):
    """se_res_pool_conv (C = 64) with two more 1x1 convs 64 -> 128 in the launch (bf16x6): tail_a on the block's output y,
    tail_b on z = conv(y); a tail = (pack_weight_x3 of [64, 128], pre_bias, scale, shift, act) with act NONE or RELU.
    Returns (y or None, z, out_a [B,N,128], out_b [B,N,128])."""
    a = L.require_cuda_f32(x, "x", 3)
    nb = L.require_cuda_i32(nbr, "nbr", 3)
    B, N, C = a.shape
    if C != 64:
        raise ValueError("se_res_pool_conv_tails: C == 64")
    y = torch.empty_like(a) if store_y else None
    z = torch.empty((B, N, C), dtype=torch.float32, device=a.device)
    oa = torch.empty((B, N, 128), dtype=torch.float32, device=a.device)
    ob = torch.empty((B, N, 128), dtype=torch.float32, device=a.device)
    ep = _ep(conv_b, conv_scale, conv_shift, act)
    ea, eb = _ep(*tail_a[1:5]), _ep(*tail_b[1:5])
    L.check(L.lib().dh3d_se_res_pool_conv_tails_pm_fwd(L.ptr(a), L.ptr(nb), B, N, nb.shape[2], L.ptr(w1packed), L.ptr(b1pad),
                                                       L.ptr(w2packed), L.ptr(b2), L.ptr(y), L.ptr(conv_wp), ep, L.ptr(z),
                                                       L.ptr(tail_a[0]), ea, L.ptr(oa), L.ptr(tail_b[0]), eb, L.ptr(ob),
                                                       L.stream_ptr()), "se_res_pool_conv_tails_pm")
    return y, z, oa, ob


def three_interpolate_idw(points, idx, dist):
    p = L.require_cuda_f32(points, "points", 3)
    ix = L.require_cuda_i32(idx, "idx", 3)
    d = L.require_cuda_f32(dist, "dist", 3)
    b, m, c = p.shape
    n = ix.shape[1]
    out = torch.empty((b, n, c), dtype=torch.float32, device=p.device)
    L.check(L.lib().dh3d_three_interpolate_idw_fwd(b, m, c, n, L.ptr(p), L.ptr(ix), L.ptr(d), L.ptr(out),
                                                   L.stream_ptr()), "three_interpolate_idw")
    return out


def l2norm_concat(x, eps, prefix=None):
    a = L.require_cuda_f32(x, "x")
    C = a.shape[-1]
    R = a.numel() // C
    P = 0
    pf = None
    if prefix is not None:
        pf = L.require_cuda_f32(prefix, "prefix")
        P = pf.shape[-1]
    out = torch.empty(a.shape[:-1] + (P + C,), dtype=torch.float32, device=a.device)
    L.check(L.lib().dh3d_l2norm_concat_fwd(L.ptr(a), R, C, float(eps), L.ptr(pf), P, L.ptr(out), L.stream_ptr()),
            "l2norm_concat")
    return out


def mlp_head(h, wpacked, H, w_fc, b_fc, pre_bias=None, scale=None, shift=None, act=ACT_RELU):
    a = L.require_cuda_f32(h, "h")
    C = a.shape[-1]
    R = a.numel() // C
    out = torch.empty(a.shape[:-1] + (1,), dtype=torch.float32, device=a.device)
    ep = _ep(pre_bias, scale, shift, act)
    L.check(L.lib().dh3d_mlp_head_pm_fwd(L.ptr(a), R, C, L.ptr(wpacked), H, ep, L.ptr(w_fc), float(b_fc),
                                         L.ptr(out), L.stream_ptr()), "mlp_head_pm")
    return out


def linear_x6(x1, wpacked_x3, Dout, x2=None, pre_bias=None, scale=None, shift=None, act=ACT_NONE, residual=None):
    """linear() on the bf16 pipe at f32 accuracy (tiled bf16x6 GEMM); Dout in {128, 256}, C1/C2 multiples of 32."""
    a = L.require_cuda_f32(x1, "x1")
    lead = a.shape[:-1]
    C1 = a.shape[-1]
    R = a.numel() // C1
    C2 = 0
    b = None
    if x2 is not None:
        b = L.require_cuda_f32(x2, "x2")
        C2 = b.shape[-1]
        if b.shape[:-1] != lead:
            raise ValueError("linear_x6: x1/x2 leading dims differ")
    res = None
    if residual is not None:
        res = L.require_cuda_f32(residual, "residual")
    out = torch.empty(lead + (Dout,), dtype=torch.float32, device=a.device)
    ep = _ep(pre_bias, scale, shift, act)
    L.check(L.lib().dh3d_linear_pm_x6_fwd(L.ptr(a), C1, L.ptr(b), C2, L.ptr(wpacked_x3), R, Dout, ep, L.ptr(res),
                                          L.ptr(out), L.stream_ptr()), "linear_pm_x6")
    return out


def upsample_linear_x6(points, idx, dist, wpacked_x3, Dout, x2=None, pre_bias=None, scale=None, shift=None,
                       act=ACT_NONE, residual=None, l2cat=None):
    """linear_x6([three_interpolate_idw(points, idx, dist) | x2]) without materialising the up-sampled tensor.
    points [B,m,C1], idx/dist [B,n,3], x2 [B,n,C2].  l2cat = (prefix [B,n,3], eps): return
    [prefix | l2_normalize(result, eps)] [B,n,3+Dout] instead (Dout == 128), the result itself is not written."""
    p = L.require_cuda_f32(points, "points", 3)
    ix = L.require_cuda_i32(idx, "idx", 3)
    d = L.require_cuda_f32(dist, "dist", 3)
    B, m, C1 = p.shape
    n = ix.shape[1]
    C2 = 0
    b = None
    if x2 is not None:
        b = L.require_cuda_f32(x2, "x2", 3)
        C2 = b.shape[-1]
        if tuple(b.shape[:2]) != (B, n):
            raise ValueError("upsample_linear_x6: x2 must be [B,n,C2]")
    res = None
    if residual is not None:
        res = L.require_cuda_f32(residual, "residual", 3)
    ep = _ep(pre_bias, scale, shift, act)
    if l2cat is not None:
        pf = L.require_cuda_f32(l2cat[0], "prefix", 3)
        if tuple(pf.shape) != (B, n, 3):
            raise ValueError("upsample_linear_x6: prefix must be [B,n,3]")
        out = torch.empty((B, n, 3 + Dout), dtype=torch.float32, device=p.device)
        L.check(L.lib().dh3d_upsample_linear_l2cat_pm_x6_fwd(L.ptr(p), L.ptr(ix), L.ptr(d), B, n, m, C1, L.ptr(b), C2,
                                                             L.ptr(wpacked_x3), Dout, ep, L.ptr(res), L.ptr(pf),
                                                             float(l2cat[1]), L.ptr(out), L.stream_ptr()),
                "upsample_linear_l2cat_pm_x6")
        return out
    out = torch.empty((B, n, Dout), dtype=torch.float32, device=p.device)
    L.check(L.lib().dh3d_upsample_linear_pm_x6_fwd(L.ptr(p), L.ptr(ix), L.ptr(d), B, n, m, C1, L.ptr(b), C2,
                                                   L.ptr(wpacked_x3), Dout, ep, L.ptr(res), L.ptr(out), L.stream_ptr()),
            "upsample_linear_pm_x6")
    return out


def upsample_linear_shortcut_x6(points, idx, dist, wstacked_x3, Dout, x2, x3, ep_main, ep_shortcut, l2cat=None):
    """act(BN([three_interpolate_idw(points, idx, dist) | x2] W)) + act_sc(BN_sc(x3 W_sc)) in one kernel (Dout == 128);
    wstacked_x3 = pack_weight_x3([W; W_sc]); ep_* = (pre_bias, scale, shift, act).  l2cat as in upsample_linear_x6."""
    p = L.require_cuda_f32(points, "points", 3)
    ix = L.require_cuda_i32(idx, "idx", 3)
    d = L.require_cuda_f32(dist, "dist", 3)
    B, m, C1 = p.shape
    n = ix.shape[1]
    b = L.require_cuda_f32(x2, "x2", 3)
    c = L.require_cuda_f32(x3, "x3", 3)
    if tuple(b.shape[:2]) != (B, n) or tuple(c.shape[:2]) != (B, n):
        raise ValueError("upsample_linear_shortcut_x6: x2 / x3 must be [B,n,*]")
    pf = None
    width = Dout
    if l2cat is not None:
        pf = L.require_cuda_f32(l2cat[0], "prefix", 3)
        if tuple(pf.shape) != (B, n, 3):
            raise ValueError("upsample_linear_shortcut_x6: prefix must be [B,n,3]")
        width = 3 + Dout
    out = torch.empty((B, n, width), dtype=torch.float32, device=p.device)
    e1, e2 = _ep(*ep_main), _ep(*ep_shortcut)
    L.check(L.lib().dh3d_upsample_linear_shortcut_pm_x6_fwd(
        L.ptr(p), L.ptr(ix), L.ptr(d), B, n, m, C1, L.ptr(b), b.shape[-1], L.ptr(c), c.shape[-1], L.ptr(wstacked_x3),
        Dout, e1, e2, L.ptr(pf), float(l2cat[1]) if l2cat is not None else 0.0, L.ptr(out), L.stream_ptr()),
        "upsample_linear_shortcut_pm_x6")
    return out


def interp_combine(coarse_w, idx, dist, partial=None, pre_bias=None, scale=None, shift=None, act=ACT_NONE, residual=None,
                   l2cat=None):
    """act(BN(interp3(coarse_w) + partial + pre_bias)) + residual -> [B,N,128]; l2cat = (prefix [B,N,3], eps) returns
    [prefix | l2_normalize(.)] [B,N,131] instead (the tail of a concat conv commuted through the up-sampling)."""
    cw = L.require_cuda_f32(coarse_w, "coarse_w", 3)
    ix = L.require_cuda_i32(idx, "idx", 3)
    d = L.require_cuda_f32(dist, "dist", 3)
    B, M, C = cw.shape
    N = ix.shape[1]
    pf, eps = (L.require_cuda_f32(l2cat[0], "prefix", 3), float(l2cat[1])) if l2cat is not None else (None, 0.0)
    out = torch.empty((B, N, C + (3 if pf is not None else 0)), dtype=torch.float32, device=cw.device)
    ep = _ep(pre_bias, scale, shift, act)
    L.check(L.lib().dh3d_interp_combine_fwd(L.ptr(cw), L.ptr(ix), L.ptr(d), L.ptr(partial), B, N, M, C, ep,
                                            L.ptr(residual), L.ptr(pf), eps, L.ptr(out), L.stream_ptr()),
            "interp_combine")
    return out


def local_tail_fused(x1, x2, wp3_shortcut, wp3_lower, ep_shortcut, ep_concat, coarse_w, idx, dist, prefix, l2_eps):
    """[prefix | l2_normalize(relu(BN_c(interp3(coarse_w) + x2 W_lower + b_c)) + relu(BN_s(x1 W_s + b_s)))] [B,N,131] in one
    launch (csrc/dense_tail.hip): interp_combine with the lower-block and shortcut GEMMs done on the fly.
    ep_* = (bias, scale, shift) tuples (activation ReLU)."""
    a = L.require_cuda_f32(x1, "x1", 3)
    b = L.require_cuda_f32(x2, "x2", 3)
    cw = L.require_cuda_f32(coarse_w, "coarse_w", 3)
    ix = L.require_cuda_i32(idx, "idx", 3)
    d = L.require_cuda_f32(dist, "dist", 3)
    pf = L.require_cuda_f32(prefix, "prefix", 3) if prefix is not None else None   # None: the plain [B,N,128] sum
    B, N, C = a.shape
    if C != 64 or b.shape != a.shape or cw.shape[2] != 128 or N % 32:
        raise ValueError("local_tail_fused: x1 / x2 [B,N,64], coarse_w [B,M,128], N % 32 == 0")
    out = torch.empty((B, N, 131 if pf is not None else 128), dtype=torch.float32, device=a.device)
    e1, e2 = _ep(*ep_shortcut, ACT_RELU), _ep(*ep_concat, ACT_RELU)
    L.check(L.lib().dh3d_local_tail_fused_fwd(L.ptr(a), L.ptr(b), L.ptr(wp3_shortcut), L.ptr(wp3_lower), e1, e2, L.ptr(cw),
                                              L.ptr(ix), L.ptr(d), L.ptr(pf), float(l2_eps), B, N, cw.shape[1], L.ptr(out),
                                              L.stream_ptr()), "local_tail_fused")
    return out


def pack_weight_x3(W):
    """W [Kd, Dout] f32 -> three exact bf16 chunk planes in MFMA fragment order (csrc/dense_x6.hip)."""
    W = L.require_cuda_f32(W, "W", 2)
    out = torch.empty((3 * W.shape[0] * W.shape[1],), dtype=torch.int16, device=W.device)
    L.check(L.lib().dh3d_pack_weight_x3(L.ptr(W), W.shape[0], W.shape[1], L.ptr(out), L.stream_ptr()),
            "pack_weight_x3")
    return out


def mlp_head_x6(h, wpacked_x3, H, w_fc, b_fc, pre_bias=None, scale=None, shift=None, act=ACT_RELU):
    """mlp_head with the 256 -> 1024 GEMM on the bf16 pipe at f32 accuracy (six exact-chunk products)."""
    a = L.require_cuda_f32(h, "h")
    C = a.shape[-1]
    R = a.numel() // C
    out = torch.empty(a.shape[:-1] + (1,), dtype=torch.float32, device=a.device)
    ep = _ep(pre_bias, scale, shift, act)
    L.check(L.lib().dh3d_mlp_head_pm_x6_fwd(L.ptr(a), R, C, L.ptr(wpacked_x3), H, ep, L.ptr(w_fc), float(b_fc),
                                            L.ptr(out), L.stream_ptr()), "mlp_head_pm_x6")
    return out


def interp_head(coarse, idx, dist, wslices_x3, Hd, w_fc, b_fc, pre_bias=None, scale=None, shift=None, act=ACT_RELU,
                order=None):
    """mlp_head_x6(three_interpolate_idw(coarse, idx, dist), ...) with the wide conv commuted through the interpolation:
    the [C, Hd] GEMM runs on the coarse rows (its 256-column slices in one launch, `wslices_x3` = their packed images
    back to back), a gather kernel
    interpolates the Hd-wide rows and finishes bias + BN + act + w_fc + sigmoid.  coarse [B,m,C], idx/dist [B,n,3].
    order: spatial_sort records [B,n,4] of the fine cloud -> the LDS-staged kernel (m <= 1024)."""
    x = L.require_cuda_f32(coarse, "coarse", 3)
    ix = L.require_cuda_i32(idx, "idx", 3)
    d = L.require_cuda_f32(dist, "dist", 3)
    B, m, C = x.shape
    n = ix.shape[1]
    ns = Hd // 256
    H = torch.empty((ns, B * m, 256), dtype=torch.float32, device=x.device)
    L.check(L.lib().dh3d_linear_slices_pm_x6_fwd(L.ptr(x), C, L.ptr(wslices_x3), B * m, ns, L.ptr(H), L.stream_ptr()),
            "linear_slices_pm_x6")
    out = torch.empty((B, n, 1), dtype=torch.float32, device=x.device)
    ep = _ep(pre_bias, scale, shift, act)
    if order is not None and m <= 1024:
        # fine points in Morton order (`order` = spatial_sort records of the fine cloud), coarse rows staged in LDS
        L.check(L.lib().dh3d_interp_head_sorted_fwd(L.ptr(H), Hd, L.ptr(ix), L.ptr(d), L.ptr(order), B, n, m, ep,
                                                    L.ptr(w_fc), float(b_fc), L.ptr(out), L.stream_ptr()),
                "interp_head_sorted")
        return out
    L.check(L.lib().dh3d_interp_head_fwd(L.ptr(H), Hd, L.ptr(ix), L.ptr(d), B, n, m, ep, L.ptr(w_fc), float(b_fc),
                                         L.ptr(out), L.stream_ptr()), "interp_head")
    return out


def netvlad_aggregate(x, att, wc_packed, bn_scale, bn_shift, W2):
    a = L.require_cuda_f32(x, "x", 3)
    B, N, D = a.shape
    Cl = W2.shape[-1]
    t = L.require_cuda_f32(att, "att").reshape(B, N)
    ws_bytes = L.lib().dh3d_netvlad_workspace_bytes(B, N, D, Cl)
    if ws_bytes == 0:
        raise ValueError("netvlad_aggregate: unsupported shape D=%d Cl=%d" % (D, Cl))
    ws = torch.empty((ws_bytes // 4,), dtype=torch.float32, device=a.device)
    vlad = torch.empty((B, D * Cl), dtype=torch.float32, device=a.device)
    L.check(L.lib().dh3d_netvlad_aggregate_fwd(L.ptr(a), L.ptr(t), L.ptr(wc_packed), L.ptr(bn_scale),
                                               L.ptr(bn_shift), L.ptr(W2), B, N, D, Cl, L.ptr(ws), ws_bytes,
                                               L.ptr(vlad), L.stream_ptr()), "netvlad_aggregate")
    return vlad


def netvlad_fused(x, att, wc_packed, bn_scale, bn_shift, W2, Wh, bn1_scale, bn1_shift, Wg, bn2_scale, bn2_shift,
                  l2_eps=0.0):
    """netvlad_head(netvlad_aggregate(...), ...) in one C call, without the separate whole-vector normalisation kernel."""
    xx = L.require_cuda_f32(x, "x", 3)
    B, N, D = xx.shape
    Cl, O = W2.shape[1], Wh.shape[1]
    a = L.require_cuda_f32(att, "att").reshape(B, N).contiguous()
    ws_bytes = L.lib().dh3d_netvlad_fused_workspace_bytes(B, N, D, Cl, O)
    if ws_bytes == 0:
        raise ValueError("netvlad_fused: unsupported shape D=%d Cl=%d O=%d" % (D, Cl, O))
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=xx.device)
    out = torch.empty((B, O), dtype=torch.float32, device=xx.device)
    L.check(L.lib().dh3d_netvlad_fused_fwd(L.ptr(xx), L.ptr(a), L.ptr(wc_packed), L.ptr(bn_scale), L.ptr(bn_shift),
                                           L.ptr(W2), L.ptr(Wh), L.ptr(bn1_scale), L.ptr(bn1_shift), L.ptr(Wg),
                                           L.ptr(bn2_scale), L.ptr(bn2_shift), B, N, D, Cl, O, float(l2_eps), L.ptr(ws),
                                           ws_bytes, L.ptr(out), L.stream_ptr()), "netvlad_fused")
    return out


def walk_plan(idx, dist, order, m):
    """The slot tables of pm.global_tail's walk, built ahead of it (csrc/dense_x6.hip walk_plan_kernel): idx / dist [B,n,3]
    from three_nn against m <= 1024 sampled points, order = spatial_sort records [B,n,4] of the fine cloud (or None).
    Returns an opaque int32 tensor, valid for exactly these inputs."""
    ix = L.require_cuda_i32(idx, "idx", 3)
    d = L.require_cuda_f32(dist, "dist", 3)
    B, n, _ = ix.shape
    nbytes = L.lib().dh3d_walk_plan_bytes(B, n)
    plan = torch.empty((nbytes // 4,), dtype=torch.int32, device=ix.device)
    L.check(L.lib().dh3d_walk_plan(L.ptr(ix), L.ptr(d), L.ptr(order), B, n, int(m), L.ptr(plan), L.stream_ptr()), "walk_plan")
    return plan


def global_tail_accum_size(B, m):
    """floats of global_tail's accumulator block [ A' B*m*64 | asum B*64 ]."""
    return B * m * 64 + B * 64


def global_tail(coarse, idx, dist, order, wslices_x3, Hd, w_fc, b_fc, att_ep, wc_packed, cl_scale, cl_shift, W2, Wh,
                bn1_scale, bn1_shift, Wg, bn2_scale, bn2_shift, l2_eps=0.0, want_att=False, accum=None, cw=None, plan=None):
    """three_interpolate -> attention head -> NetVLAD + gating, with the up-sampling commuted through both consumers: the
    fine points are walked once (csrc/dense_x6.hip VladTail), everything else runs on the coarse rows.
    coarse [B,m,256], idx/dist [B,n,3], order = spatial_sort records [B,n,4] of the fine cloud, att_ep =
    (pre_bias, scale, shift, act) of the attention's hidden layer.  accum (optional): a ZEROED float32 tensor of
    global_tail_accum_size(B, m) elements (the caller's fill, issued off the critical chain); cw (optional):
    coarse @ cluster_weights if the caller has it already; plan (optional): walk_plan(idx, dist, order, m) of exactly these
    inputs, built off the critical chain.  Returns the descriptor [B,O] (, att [B,n,1])."""
    x = L.require_cuda_f32(coarse, "coarse", 3)
    ix = L.require_cuda_i32(idx, "idx", 3)
    d = L.require_cuda_f32(dist, "dist", 3)
    B, m, C = x.shape
    n = ix.shape[1]
    ns = Hd // 256
    H = torch.empty((ns, B * m, 256), dtype=torch.float32, device=x.device)
    L.check(L.lib().dh3d_linear_slices_pm_x6_fwd(L.ptr(x), C, L.ptr(wslices_x3), B * m, ns, L.ptr(H), L.stream_ptr()),
            "linear_slices_pm_x6")
    if cw is None:   # (the caller may have it already: pm.flex_conv_post computes it in the flex_conv's launch)
        cw = linear(x, wc_packed, 64)                                              # coarse @ cluster_weights
    att = torch.empty((B, n, 1), dtype=torch.float32, device=x.device) if want_att else None
    zero_here = accum is None
    if accum is None:
        accum = torch.empty((global_tail_accum_size(B, m),), dtype=torch.float32, device=x.device)
    elif accum.numel() != global_tail_accum_size(B, m) or accum.dtype != torch.float32 or not accum.is_cuda:
        raise ValueError("global_tail: accum must be a zeroed float32 GPU tensor of global_tail_accum_size(B, m) elements")
    ep = _ep(*att_ep)
    if plan is not None and (plan.dtype != torch.int32 or plan.numel() * 4 != L.lib().dh3d_walk_plan_bytes(B, n)):
        raise ValueError("global_tail: plan is not walk_plan(idx, dist, order, m) of these shapes")
    L.check(L.lib().dh3d_global_walk_planned_fwd(L.ptr(H), Hd, L.ptr(x), L.ptr(cw), L.ptr(ix), L.ptr(d), L.ptr(order), L.ptr(plan),
                                                 B, n, m, ep, L.ptr(w_fc), float(b_fc), L.ptr(cl_scale), L.ptr(cl_shift),
                                                 L.ptr(att), L.ptr(accum), 1 if zero_here else 0, L.stream_ptr()), "global_walk")
    apart, asum = accum[:B * m * 64], accum[B * m * 64:]
    O = Wh.shape[1]
    ws_bytes = L.lib().dh3d_netvlad_tail_workspace_bytes(B, C, 64, O)
    if ws_bytes == 0:
        raise ValueError("global_tail: unsupported NetVLAD shape")
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=x.device)
    out = torch.empty((B, O), dtype=torch.float32, device=x.device)
    # V = A'^T coarse is formed inside the finalize kernel (no batched GEMM launch)
    L.check(L.lib().dh3d_netvlad_tail_assign_fwd(L.ptr(apart), L.ptr(x), L.ptr(asum), m, L.ptr(W2), L.ptr(Wh), L.ptr(bn1_scale),
                                                 L.ptr(bn1_shift), L.ptr(Wg), L.ptr(bn2_scale), L.ptr(bn2_shift), B, C, 64, O,
                                                 float(l2_eps), L.ptr(ws), ws_bytes, L.ptr(out), L.stream_ptr()),
            "netvlad_tail_assign")
    return (out, att) if want_att else out


def netvlad_head(vlad, Wh, bn1_scale, bn1_shift, Wg, bn2_scale, bn2_shift, l2_eps=0.0):
    v = L.require_cuda_f32(vlad, "vlad", 2)
    B, Kd = v.shape
    O = Wh.shape[1]
    ws_bytes = L.lib().dh3d_netvlad_head_workspace_bytes(B, Kd, O)
    if ws_bytes == 0:
        raise ValueError("netvlad_head: unsupported output dim %d" % O)
    ws = torch.empty((ws_bytes // 4,), dtype=torch.float32, device=v.device)
    out = torch.empty((B, O), dtype=torch.float32, device=v.device)
    L.check(L.lib().dh3d_netvlad_head_fwd(L.ptr(v), L.ptr(Wh), L.ptr(bn1_scale), L.ptr(bn1_shift), L.ptr(Wg),
                                          L.ptr(bn2_scale), L.ptr(bn2_shift), B, Kd, O, float(l2_eps), L.ptr(ws),
                                          ws_bytes, L.ptr(out), L.stream_ptr()), "netvlad_head")
    return out


# --------------------------------------------------------------------------------------------- backward / training
def flex_conv_bwd(features, xyz, nbr, theta, bias, grad_out, center_rank0=False, need_grad_features=True):
    """Factorised flex_conv backward (csrc/flex_bwd.hip): features [B,N,Din], xyz [B,N,3], nbr [B,N,K] int32,
    theta [3,Din,Dout], bias [Din,Dout], grad_out [B,N,Dout] -> (grad_features [B,N,Din] or None, grad_theta, grad_bias)."""
    f = L.require_cuda_f32(features, "features", 3)
    x = L.require_cuda_f32(xyz, "xyz", 3)
    nb = L.require_cuda_i32(nbr, "nbr", 3)
    t = L.require_cuda_f32(theta, "theta", 3)
    bi = L.require_cuda_f32(bias, "bias", 2)
    g = L.require_cuda_f32(grad_out, "grad_out", 3)
    B, N, Din = f.shape
    K, Dout = nb.shape[2], t.shape[2]
    if t.shape[0] != 3 or tuple(bi.shape) != (Din, Dout) or tuple(g.shape) != (B, N, Dout):
        raise ValueError("flex_conv_bwd: inconsistent shapes")
    ws_bytes = L.lib().dh3d_flex_conv_pm_bwd_workspace_bytes(B, N, Din, Dout)
    if ws_bytes == 0:
        raise ValueError("flex_conv_bwd: Din and Dout must be multiples of 4")
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=f.device)
    gf = torch.empty_like(f) if need_grad_features else None
    gt, gb = torch.empty_like(t), torch.empty_like(bi)
    L.check(L.lib().dh3d_flex_conv_pm_bwd(L.ptr(f), L.ptr(x), L.ptr(nb), L.ptr(t), L.ptr(bi), L.ptr(g), B, N, K, Din,
                                          Dout, 1 if center_rank0 else 0, L.ptr(ws), ws_bytes, L.ptr(gf), L.ptr(gt),
                                          L.ptr(gb), L.stream_ptr()), "flex_conv_pm_bwd")
    return gf, gt, gb


def gemm_tn(A, B, out=None, accumulate=False):
    """C[M,N] (+)= A[K,M]^T @ B[K,N], f32-accurate (exact-f32 MFMA for small products, bf16x6 with both operands split on
    the fly from 2^26 multiply-adds; csrc/gemm.hip) -- weight gradients: the reduction runs over rows."""
    A = L.require_cuda_f32(A, "A", 2)
    B = L.require_cuda_f32(B, "B", 2)
    K, M = A.shape
    if B.shape[0] != K:
        raise ValueError("gemm_tn: row counts differ")
    N = B.shape[1]
    if out is None and _ZERO_ARENA[0] is not None and L.lib().dh3d_gemm_is_split(1, M, N, K, 1):
        out, accumulate = zeros((M, N), torch.float32, A.device), True   # split reduction: the arena's zeros, no fill
    C = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=A.device)
    L.check(L.lib().dh3d_gemm_tn_f32(L.ptr(A), L.ptr(B), K, M, N, 1 if accumulate else 0, L.ptr(C), L.stream_ptr()),
            "gemm_tn")
    return C


def gemm_nn(A, B, out=None, accumulate=False, bias=None):
    """C[M,N] (+)= A[M,K] @ B[K,N] (+ bias[N]), f32-accurate (see gemm_tn)."""
    A = L.require_cuda_f32(A, "A", 2)
    B = L.require_cuda_f32(B, "B", 2)
    M, K = A.shape
    if B.shape[0] != K:
        raise ValueError("gemm_nn: inner dimensions differ")
    N = B.shape[1]
    if out is None and bias is None and _ZERO_ARENA[0] is not None and L.lib().dh3d_gemm_is_split(0, M, N, K, 1):
        out, accumulate = zeros((M, N), torch.float32, A.device), True
    C = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=A.device)
    L.check(L.lib().dh3d_gemm_nn_f32(L.ptr(A), L.ptr(B), L.ptr(bias), M, K, N, 1 if accumulate else 0, L.ptr(C),
                                     L.stream_ptr()), "gemm_nn")
    return C


def stage_copy(src, dst):
    """dst <- src on the current stream by a kernel (dh3d_stage_copy): each side a contiguous CUDA tensor or a contiguous
    PINNED CPU tensor (device-addressable), same byte count.  The serving loop's host <-> slot-buffer hand-over
    (engine.Pipeline.submit): no copy engine, no engine-to-engine dependency on the step's chain."""
    for t, nm in ((src, "src"), (dst, "dst")):
        if not (t.is_cuda or t.is_pinned()):
            raise ValueError("stage_copy: %s must be a CUDA tensor or a pinned CPU tensor" % nm)
        if not t.is_contiguous():
            raise ValueError("stage_copy: %s must be contiguous" % nm)
    nbytes = src.numel() * src.element_size()
    if nbytes != dst.numel() * dst.element_size():
        raise ValueError("stage_copy: %d bytes into %d" % (nbytes, dst.numel() * dst.element_size()))
    L.check(L.lib().dh3d_stage_copy(src.data_ptr(), dst.data_ptr(), nbytes, int(src.is_cuda and dst.is_cuda),
                                    L.stream_ptr()), "stage_copy")
    return dst


def transpose_last2(x):
    """[..., R, C] -> [..., C, R] (32-bit elements) through LDS tiles."""
    if not x.is_cuda or x.element_size() != 4:
        raise ValueError("transpose_last2: 32-bit GPU tensors only")
    x = x.contiguous()
    R, C = x.shape[-2], x.shape[-1]
    Bt = x.numel() // (R * C)
    out = torch.empty(tuple(x.shape[:-2]) + (C, R), dtype=x.dtype, device=x.device)
    L.check(L.lib().dh3d_transpose32(L.ptr(x), Bt, R, C, L.ptr(out), L.stream_ptr()), "transpose32")
    return out


def colsum(x, out=None, accumulate=False):
    """x [R, C] -> [C] column sums (bias gradients)."""
    x = L.require_cuda_f32(x, "x", 2)
    R, C = x.shape
    o = out if out is not None else torch.empty((C,), dtype=torch.float32, device=x.device)
    L.check(L.lib().dh3d_colsum_f32(L.ptr(x), R, C, 1 if accumulate else 0, L.ptr(o), L.stream_ptr()), "colsum")
    return o


def _mask_u8(mask):
    return None if mask is None else mask.to(torch.uint8).contiguous()


_ITEMSIZE = {torch.float32: 4, torch.float64: 8, torch.int32: 4, torch.int64: 8, torch.uint8: 1}


class ZeroArena(object):
    """One zeroed device buffer per training step for the accumulators its kernels add into (statistics partials, scatter
    targets: include/dh3d_hip.h "zeroed by the CALLER").  `begin()` clears it with ONE fill and rewinds; `zeros()` below
    hands out views while the arena is active (`with pm.zero_arena(a): ...`) and falls back to torch.zeros otherwise or
    when the arena is too small -- the demand of a step is recorded, the next `begin()` outside a graph capture grows
    the buffer to it.  Views are only valid until the next `begin()`.  Two kinds of tensors that leave a step DO come
    from here and inherit that lifetime: the loss never does, but the parameter gradients of split-reduction GEMMs
    (`gemm_tn` / `gemm_nn` below: dW of the attention head and of NetVLAD's assignment) are arena views -- `p.grad` is
    valid until the next step begins, which is what an optimiser step needs; code that keeps gradients across steps
    (gradient accumulation, inspection) must clone them (QuadrupletTrainer.keep_grads does).

    A buffer is never freed or replaced while a captured hipGraph may still replay into it: an arena built with
    `fixed_bytes` (one per captured step graph, owned by that graph's entry) never reallocates -- `take()` falls back
    to torch.zeros (inside a capture: the graph's own pool) when it is too small; only the growing arena of the eager
    steps reallocates, and no graph records its address."""

    def __init__(self, fixed_bytes=None, device=None):
        self.buf, self.off, self.demand, self.peak = None, 0, 0, 0
        self.fixed = fixed_bytes is not None
        if self.fixed and fixed_bytes > 0:
            self.buf = torch.empty((int(fixed_bytes),), dtype=torch.uint8, device=device)

    def begin(self, device):
        self.peak = max(self.peak, self.demand)
        capturing = torch.cuda.is_current_stream_capturing() if torch.cuda.is_available() else False
        if (not self.fixed and self.peak and not capturing
                and (self.buf is None or self.buf.numel() < self.peak or self.buf.device != device)):
            self.buf = torch.empty((self.peak,), dtype=torch.uint8, device=device)
        if self.buf is not None:
            self.buf.zero_()
        self.off, self.demand = 0, 0

    def take(self, shape, dtype, device):
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * _ITEMSIZE[dtype]
        span = (nbytes + 255) // 256 * 256
        self.demand += span
        if self.buf is None or self.buf.device != device or self.off + span > self.buf.numel():
            return torch.zeros(shape, dtype=dtype, device=device)
        # a tensor of its own on the arena's storage (NOT a view of `buf`: autograd would tie every accumulator that
        # leaves a custom Function to the arena's version counter, which the next begin() bumps)
        esz = nbytes // max(n, 1) if n else 1
        v = torch.empty((0,), dtype=dtype, device=device).set_(self.buf.untyped_storage(), self.off // esz, tuple(shape))
        self.off += span
        return v


_ZERO_ARENA = [None]


class zero_arena(object):
    """Context: accumulators come out of `arena` (a ZeroArena, already begun) instead of one torch.zeros each."""

    def __init__(self, arena):
        self.arena, self.prev = arena, None

    def __enter__(self):
        self.prev, _ZERO_ARENA[0] = _ZERO_ARENA[0], self.arena
        return self.arena

    def __exit__(self, *exc):
        _ZERO_ARENA[0] = self.prev
        return False


def zeros(shape, dtype, device):
    """A zeroed accumulator: a view of the active ZeroArena, or torch.zeros."""
    a = _ZERO_ARENA[0]
    return a.take(tuple(shape), dtype, device) if a is not None else torch.zeros(tuple(shape), dtype=dtype, device=device)


def bn_colstats(x, mask=None, rows_per_cloud=0, out=None):
    """x [R,C] -> (sum [C], sumsq [C]) float64 (views of `out` [>= 2C] if given); mask [clouds] bool excludes padding
    clouds of rows_per_cloud rows."""
    x = L.require_cuda_f32(x, "x", 2)
    R, C = x.shape
    buf = out if out is not None else zeros((2 * C,), torch.float64, x.device)   # `out` must be zeroed by the caller
    s1, s2 = buf[:C], buf[C:2 * C]
    m = _mask_u8(mask)
    L.check(L.lib().dh3d_bn_colstats(L.ptr(x), R, C, L.ptr(m), int(rows_per_cloud), L.ptr(s1), L.ptr(s2),
                                     L.stream_ptr()), "bn_colstats")
    return s1, s2


def bn_finalize(s1, s2, cnt, gamma, beta, eps, momentum, run_mean, run_var, unbiased=True):
    """(sum, sumsq, count) -> [mean, rstd, scale, shift] as rows of one [4, C] float32 tensor; running buffers updated
    (unbiased: the moving variance takes the Bessel-corrected batch variance, as tf.nn.fused_batch_norm feeds it)."""
    C = gamma.numel()
    o = torch.empty((4, C), dtype=torch.float32, device=gamma.device)
    L.check(L.lib().dh3d_bn_finalize(L.ptr(s1), L.ptr(s2), L.ptr(cnt), L.ptr(gamma), L.ptr(beta), float(eps),
                                     float(momentum), 1 if unbiased else 0, L.ptr(run_mean), L.ptr(run_var), C, L.ptr(o[0]), L.ptr(o[1]),
                                     L.ptr(o[2]), L.ptr(o[3]), L.stream_ptr()), "bn_finalize")
    return o


def bn_bwd_finalize(S1, S2, cnt, mean, rstd, gamma):
    """-> [k2, k3] rows of a [2, C] tensor (coefficients of bn_bwd_apply)."""
    C = gamma.numel()
    o = torch.empty((2, C), dtype=torch.float32, device=gamma.device)
    L.check(L.lib().dh3d_bn_bwd_finalize(L.ptr(S1), L.ptr(S2), L.ptr(cnt), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), C,
                                         L.ptr(o[0]), L.ptr(o[1]), L.stream_ptr()), "bn_bwd_finalize")
    return o


def bn_finalize_parts(part, cnt, gamma, beta, eps, momentum, run_mean, run_var, unbiased=True):
    """bn_finalize from per-cloud partial rows part [2, P, C] float64 (sum | sumsq), the reduction over P included."""
    _, P, C = part.shape
    o = torch.empty((4, C), dtype=torch.float32, device=gamma.device)
    L.check(L.lib().dh3d_bn_finalize_parts(L.ptr(part), P, L.ptr(cnt), L.ptr(gamma), L.ptr(beta), float(eps),
                                           float(momentum), 1 if unbiased else 0, L.ptr(run_mean), L.ptr(run_var), C,
                                           L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[2]), L.ptr(o[3]), L.stream_ptr()),
            "bn_finalize_parts")
    return o


def bn_bwd_finalize_parts(part, cnt, mean, rstd, gamma):
    """part [nk, P, C] float64 per-cloud partial sums (nk = 2 or 3) -> ([k2, k3] rows of a [2, C] tensor, grads [nk, C]
    float32 = their sums over P: dbeta, dgamma(, d w_fc)) in one launch."""
    nk, P, C = part.shape
    o = torch.empty((2, C), dtype=torch.float32, device=part.device)
    grads = torch.empty((nk, C), dtype=torch.float32, device=part.device)
    L.check(L.lib().dh3d_bn_bwd_finalize_parts(L.ptr(part), nk, P, L.ptr(cnt), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), C,
                                               L.ptr(o[0]), L.ptr(o[1]), L.ptr(grads), L.stream_ptr()),
            "bn_bwd_finalize_parts")
    return o, grads


def sigmoid_bwd(datt, att, mask=None, rows_per_cloud=0):
    """-> (dlogit = datt*att*(1-att), 0 on the rows of masked clouds; its sum as a [1] tensor)."""
    datt, att = datt.contiguous(), att.contiguous()
    dlogit = torch.empty_like(att)
    tot = torch.zeros((1,), dtype=torch.float32, device=att.device)   # (a parameter gradient: not from the step arena)
    L.check(L.lib().dh3d_sigmoid_bwd(L.ptr(datt), L.ptr(att), L.ptr(_mask_u8(mask)), int(rows_per_cloud), att.numel(),
                                     L.ptr(dlogit), L.ptr(tot), L.stream_ptr()), "sigmoid_bwd")
    return dlogit, tot


def scale_shift_act(x, scale, shift, relu, out=None, residual=None):
    """y = act(x * scale[c] + shift[c]) (+ residual, added after the activation)."""
    x = L.require_cuda_f32(x, "x", 2)
    R, C = x.shape
    y = out if out is not None else torch.empty_like(x)
    if residual is not None:
        r = L.require_cuda_f32(residual, "residual", 2)
        if r.shape != x.shape:
            raise ValueError("scale_shift_act: residual shape differs")
        L.check(L.lib().dh3d_scale_shift_act_res(L.ptr(x), R, C, L.ptr(scale), L.ptr(shift), 1 if relu else 0, L.ptr(r),
                                                 L.ptr(y), L.stream_ptr()), "scale_shift_act_res")
        return y
    L.check(L.lib().dh3d_scale_shift_act(L.ptr(x), R, C, L.ptr(scale), L.ptr(shift), 1 if relu else 0, L.ptr(y),
                                         L.stream_ptr()), "scale_shift_act")
    return y


def row_logit_sigmoid(h, scale, shift, w, b):
    """att[r] = sigmoid(sum_c relu(h*scale+shift)[r,c]*w[c] + b); b: device tensor holding the scalar."""
    h = L.require_cuda_f32(h, "h", 2)
    R, C = h.shape
    att = torch.empty((R,), dtype=torch.float32, device=h.device)
    L.check(L.lib().dh3d_row_logit_sigmoid(L.ptr(h), R, C, L.ptr(scale), L.ptr(shift), L.ptr(w), L.ptr(b), L.ptr(att),
                                           L.stream_ptr()), "row_logit_sigmoid")
    return att


def bn_bwd_sums(x, mean, rstd, gamma, beta, relu, dy=None, rowscale=None, colvec=None, mask=None, rows_per_cloud=0):
    """-> S [3, C] float64: S[0] = sum dz, S[1] = sum dz*xhat, S[2] = sum rowscale*y (rank-one form only; else unused)."""
    x = L.require_cuda_f32(x, "x", 2)
    R, C = x.shape
    S = zeros((3, C), torch.float64, x.device)
    m = _mask_u8(mask)
    L.check(L.lib().dh3d_bn_bwd_sums(L.ptr(x), L.ptr(dy), L.ptr(rowscale), L.ptr(colvec), R, C, L.ptr(mean), L.ptr(rstd),
                                     L.ptr(gamma), L.ptr(beta), 1 if relu else 0, L.ptr(m), int(rows_per_cloud),
                                     L.ptr(S[0]), L.ptr(S[1]), L.ptr(S[2]), L.stream_ptr()), "bn_bwd_sums")
    return S


def bn_bwd_apply(x, scale, shift, k2, k3, relu, dy=None, rowscale=None, colvec=None, mask=None, rows_per_cloud=0,
                 out=None):
    """dx = scale*dz - k2 - k3*x (see include/dh3d_hip.h); out may be x (in place)."""
    x = L.require_cuda_f32(x, "x", 2)
    R, C = x.shape
    dx = out if out is not None else torch.empty_like(x)
    m = _mask_u8(mask)
    L.check(L.lib().dh3d_bn_bwd_apply(L.ptr(x), L.ptr(dy), L.ptr(rowscale), L.ptr(colvec), R, C, L.ptr(scale),
                                      L.ptr(shift), L.ptr(k2), L.ptr(k3), 1 if relu else 0, L.ptr(m),
                                      int(rows_per_cloud), L.ptr(dx), L.stream_ptr()), "bn_bwd_apply")
    return dx


# ---- attention head, training mode, conv commuted through the up-sampling (csrc/interp_train.hip)
def _g_layout(G, idx):
    """G: [ns, B*m, 256] slices or [B*m, Hd] row-major -> (Hd, row_major flag, m)."""
    B = idx.shape[0]
    if G.dim() == 3:
        return G.shape[0] * 256, 0, G.shape[1] // B
    return G.shape[1], 1, G.shape[0] // B


def interp_bn_colstats(G, idx, dist, order, mask=None, out=None, parts=False):
    """G = coarse @ W + b as [ns, B*m, 256] slices or as [B*m, Hd]; idx/dist [B,n,3]; order = spatial_sort records
    [B,n,4] or None -> (sum, sumsq) [Hd] float64 of the virtual rows three_interpolate(G) (views of `out` if given)."""
    Hd, rm, m = _g_layout(G, idx)
    B, n = idx.shape[0], idx.shape[1]
    part = zeros((2, B, Hd), torch.float64, G.device)
    L.check(L.lib().dh3d_interp_bn_colstats(L.ptr(G), Hd, rm, L.ptr(idx), L.ptr(dist), L.ptr(order), B, n, m,
                                            L.ptr(_mask_u8(mask)), L.ptr(part), L.stream_ptr()), "interp_bn_colstats")
    if parts:
        return part
    buf = out if out is not None else torch.empty((2 * Hd,), dtype=torch.float64, device=G.device)
    torch.sum(part, dim=1, out=buf[:2 * Hd].view(2, Hd))
    return buf[:Hd], buf[Hd:2 * Hd]


def interp_head_rows(G, idx, dist, order, scale, shift, w_fc, b_fc_dev):
    """att [B*n] = sigmoid(relu(three_interpolate(G) * scale + shift) . w_fc + b_fc), b_fc a device scalar."""
    Hd, rm, m = _g_layout(G, idx)
    B, n = idx.shape[0], idx.shape[1]
    out = torch.empty((B * n,), dtype=torch.float32, device=G.device)
    ep = _ep(None, scale, shift, ACT_RELU)
    L.check(L.lib().dh3d_interp_head_sorted_fwd_dev(L.ptr(G), Hd, rm, L.ptr(idx), L.ptr(dist), L.ptr(order), B, n, m, ep,
                                                    L.ptr(w_fc), L.ptr(b_fc_dev), L.ptr(out), L.stream_ptr()),
            "interp_head_sorted_dev")
    return out


def interp_bn_bwd_sums(G, idx, dist, order, dlogit, w_fc, mean, rstd, gamma, beta, mask=None, parts=False):
    """-> S [3, Hd] float64 (parts: the per-cloud partials [3, B, Hd] for bn_bwd_finalize_parts)."""
    Hd, rm, m = _g_layout(G, idx)
    B, n = idx.shape[0], idx.shape[1]
    part = zeros((3, B, Hd), torch.float64, G.device)
    L.check(L.lib().dh3d_interp_bn_bwd_sums(L.ptr(G), Hd, rm, L.ptr(idx), L.ptr(dist), L.ptr(order), B, n, m,
                                            L.ptr(_mask_u8(mask)), L.ptr(dlogit), L.ptr(w_fc), L.ptr(mean), L.ptr(rstd),
                                            L.ptr(gamma), L.ptr(beta), L.ptr(part), L.stream_ptr()), "interp_bn_bwd_sums")
    return part if parts else part.sum(1)


def interp_bn_bwd_apply(G, idx, dist, order, dlogit, w_fc, scale, shift, k2, k3, mask=None):
    """-> dG (the layout of G) = interp^T(scale*dz - k2 - k3*h)."""
    Hd, rm, m = _g_layout(G, idx)
    B, n = idx.shape[0], idx.shape[1]
    dG = zeros(G.shape, torch.float32, G.device)
    L.check(L.lib().dh3d_interp_bn_bwd_apply(L.ptr(G), Hd, rm, L.ptr(idx), L.ptr(dist), L.ptr(order), B, n, m,
                                             L.ptr(_mask_u8(mask)), L.ptr(dlogit), L.ptr(w_fc), L.ptr(scale),
                                             L.ptr(shift), L.ptr(k2), L.ptr(k3), L.ptr(dG), L.stream_ptr()),
            "interp_bn_bwd_apply")
    return dG


# ---- NetVLAD's assignment, training mode, rows commuted through the up-sampling (csrc/netvlad_train.hip)
def nv_commuted_fwd_stats(c, cw, idx, dist, order, mask=None, out=None, parts=False):
    """c [B*m,256], cw = c @ Wc [B*m,64] -> s [B*n,64], rinv [B*n] (by original point index) and (sum, sumsq) [64]
    float64 of the columns of s (views of `out` if given)."""
    B, n = idx.shape[0], idx.shape[1]
    m = c.shape[0] // B
    s = torch.empty((B * n, 64), dtype=torch.float32, device=c.device)
    rinv = torch.empty((B * n,), dtype=torch.float32, device=c.device)
    part = zeros((2, B, 64), torch.float64, c.device)
    L.check(L.lib().dh3d_netvlad_commuted_fwd_stats(L.ptr(c), L.ptr(cw), L.ptr(idx), L.ptr(dist), L.ptr(order), B, n, m,
                                                    L.ptr(_mask_u8(mask)), L.ptr(s), L.ptr(rinv), L.ptr(part),
                                                    L.stream_ptr()), "netvlad_commuted_fwd_stats")
    if parts:
        return s, rinv, part
    buf = out if out is not None else torch.empty((128,), dtype=torch.float64, device=c.device)
    torch.sum(part, dim=1, out=buf[:128].view(2, 64))
    return s, rinv, buf[:64], buf[64:128]


def nv_commuted_fwd_assign(s, rinv, att, scale, shift, idx, dist, order, m, mask=None):
    """-> p [B*n,64] = softmax(s*scale + shift), asum [B,64] = sum_n p*att, Ap [B*m,64] = interp^T(p*att*rinv)."""
    B, n = idx.shape[0], idx.shape[1]
    p = torch.empty_like(s)
    asum = zeros((B, 64), torch.float32, s.device)
    Ap = zeros((B * m, 64), torch.float32, s.device)
    L.check(L.lib().dh3d_netvlad_commuted_fwd_assign(L.ptr(s), L.ptr(rinv), L.ptr(att), L.ptr(scale), L.ptr(shift),
                                                     L.ptr(idx), L.ptr(dist), L.ptr(order), B, n, m, L.ptr(_mask_u8(mask)),
                                                     L.ptr(p), L.ptr(asum), L.ptr(Ap), L.stream_ptr()),
            "netvlad_commuted_fwd_assign")
    return p, asum, Ap


def nv_commuted_bwd_sums(E, p, s, att, rinv, dasum, mean, rstd, idx, dist, order, mask=None, parts=False):
    """-> dz [B*n,64], datt [B*n], t2 [B*n], S [2,64] float64 (sum dz, sum dz*shat)."""
    B, n = idx.shape[0], idx.shape[1]
    m = E.shape[0] // B
    dz = torch.empty_like(s)
    datt = torch.empty((B * n,), dtype=torch.float32, device=s.device)
    t2 = torch.empty((B * n,), dtype=torch.float32, device=s.device)
    part = zeros((2, B, 64), torch.float64, s.device)
    L.check(L.lib().dh3d_netvlad_commuted_bwd_sums(L.ptr(E), L.ptr(p), L.ptr(s), L.ptr(att), L.ptr(rinv), L.ptr(dasum),
                                                   L.ptr(mean), L.ptr(rstd), L.ptr(idx), L.ptr(dist), L.ptr(order), B, n, m,
                                                   L.ptr(_mask_u8(mask)), L.ptr(dz), L.ptr(datt), L.ptr(t2), L.ptr(part),
                                                   L.stream_ptr()), "netvlad_commuted_bwd_sums")
    return dz, datt, t2, (part if parts else part.sum(1))


def nv_commuted_bwd_apply(dz, s, rinv, t2, k1, k2, k3, idx, dist, order, m, mask=None):
    """-> q [B*n], dcw [B*m,64] = interp^T(rinv * (k1*dz - k2 - k3*s))."""
    B, n = idx.shape[0], idx.shape[1]
    q = torch.empty((B * n,), dtype=torch.float32, device=s.device)
    dcw = zeros((B * m, 64), torch.float32, s.device)
    L.check(L.lib().dh3d_netvlad_commuted_bwd_apply(L.ptr(dz), L.ptr(s), L.ptr(rinv), L.ptr(t2), L.ptr(k1), L.ptr(k2),
                                                    L.ptr(k3), L.ptr(idx), L.ptr(dist), L.ptr(order), B, n, m,
                                                    L.ptr(_mask_u8(mask)), L.ptr(q), L.ptr(dcw), L.stream_ptr()),
            "netvlad_commuted_bwd_apply")
    return q, dcw


def interp_scatter_scaled(c, q, idx, dist, order, dc, mask=None):
    """dc [B*m,256] += interp^T(-q * interp(c)) in place."""
    B, n = idx.shape[0], idx.shape[1]
    m = c.shape[0] // B
    L.check(L.lib().dh3d_interp_scatter_scaled(L.ptr(c), L.ptr(q), L.ptr(idx), L.ptr(dist), L.ptr(order), B, n, m,
                                               L.ptr(_mask_u8(mask)), L.ptr(dc), L.stream_ptr()), "interp_scatter_scaled")
    return dc


def netvlad_assign_rows(s, scale, shift, att, rows_per_cloud=0):
    """a = softmax(s*scale + shift) * att; with rows_per_cloud (% 64 == 0) also (a, asum [clouds, 64]) -- the per-cloud
    column sums of a from the same pass."""
    s = L.require_cuda_f32(s, "s", 2)
    a = torch.empty_like(s)
    asum = None
    if rows_per_cloud:
        asum = torch.empty((s.shape[0] // rows_per_cloud, s.shape[1]), dtype=torch.float32, device=s.device)
    L.check(L.lib().dh3d_netvlad_assign_rows(L.ptr(s), s.shape[0], s.shape[1], L.ptr(scale), L.ptr(shift), L.ptr(att),
                                             L.ptr(a), L.ptr(asum), int(rows_per_cloud), L.stream_ptr()),
            "netvlad_assign_rows")
    return (a, asum) if rows_per_cloud else a


def idw_weights(dist):
    """three_nn distances [..., 3] -> inverse-distance weights (core/backbones.py:92-95), one launch."""
    dist = L.require_cuda_f32(dist, "dist", dist.dim())
    w = torch.empty_like(dist)
    L.check(L.lib().dh3d_idw_weights(L.ptr(dist), dist.numel() // 3, L.ptr(w), L.stream_ptr()), "idw_weights")
    return w


def context_gate(v, g):
    v, g = L.require_cuda_f32(v, "v", v.dim()), L.require_cuda_f32(g, "g", g.dim())
    y = torch.empty_like(v)
    L.check(L.lib().dh3d_context_gate_fwd(L.ptr(v), L.ptr(g), v.numel(), L.ptr(y), L.stream_ptr()), "context_gate")
    return y


def context_gate_bwd(v, g, dy):
    dv, dg = torch.empty_like(v), torch.empty_like(v)
    L.check(L.lib().dh3d_context_gate_bwd(L.ptr(v), L.ptr(g), L.ptr(dy), v.numel(), L.ptr(dv), L.ptr(dg), L.stream_ptr()),
            "context_gate_bwd")
    return dv, dg


def netvlad_assign_rows_bwd(s, scale, shift, att, da):
    s = L.require_cuda_f32(s, "s", 2)
    dz = torch.empty_like(s)
    datt = torch.empty((s.shape[0],), dtype=torch.float32, device=s.device)
    L.check(L.lib().dh3d_netvlad_assign_rows_bwd(L.ptr(s), s.shape[0], s.shape[1], L.ptr(scale), L.ptr(shift),
                                                 L.ptr(att), L.ptr(da), L.ptr(dz), L.ptr(datt), L.stream_ptr()),
            "netvlad_assign_rows_bwd")
    return dz, datt


def l2norm_rows_bwd(x, dxn, eps):
    x = L.require_cuda_f32(x, "x", 2)
    dx = torch.empty_like(x)
    L.check(L.lib().dh3d_l2norm_rows_bwd(L.ptr(x), L.ptr(dxn), x.shape[0], x.shape[1], float(eps), L.ptr(dx),
                                         L.stream_ptr()), "l2norm_rows_bwd")
    return dx


def gemm_tn_batched(A, B):
    """A [b,K,M], B [b,K,N] -> C [b,M,N] = A^T B per batch entry."""
    A = L.require_cuda_f32(A, "A", 3)
    B = L.require_cuda_f32(B, "B", 3)
    b, K, M = A.shape
    N = B.shape[2]
    C = torch.empty((b, M, N), dtype=torch.float32, device=A.device)
    L.check(L.lib().dh3d_gemm_tn_f32_batched(L.ptr(A), L.ptr(B), b, K, M, N, L.ptr(C), L.stream_ptr()), "gemm_tn_batched")
    return C


def gemm_nn_batched(A, B, bias=None):
    """A [b,M,K], B [b,K,N] (+ bias [b,N]) -> C [b,M,N]."""
    A = L.require_cuda_f32(A, "A", 3)
    B = L.require_cuda_f32(B, "B", 3)
    b, M, K = A.shape
    N = B.shape[2]
    C = torch.empty((b, M, N), dtype=torch.float32, device=A.device)
    L.check(L.lib().dh3d_gemm_nn_f32_batched(L.ptr(A), L.ptr(B), L.ptr(bias), b, M, K, N, L.ptr(C), L.stream_ptr()),
            "gemm_nn_batched")
    return C
