"""Autograd nodes of the training step whose forward AND backward are hand-written HIP kernels (csrc/train.hip,
csrc/gemm.hip, csrc/flex_bwd.hip) -- the counterparts of the TF graph's fused_batch_norm / MatMul / custom-op gradient
nodes that core/model.py:135-255 builds for the trainable global head:

    BatchNormTrain   training-mode BatchNorm (+ReLU) on [R, C] rows (tensorpack BatchNorm, slim batch_norm with
                     is_training; core/tf_utils.py:60-63, core/backbones.py:218-223,271-274,304-309)
    Linear           y = x W (+ b): exact-f32 MFMA GEMMs in both directions (dW = x^T dy, dx = dy W^T)
    AttentionHead    globalatt_block (core/backbones.py:156-173): 256 -> 1024 BNReLU -> 1 -> sigmoid, the rank-one
                     activation gradient never materialised, the pre-activation overwritten by its gradient in place
    NetVLADAssign    rows of NetVLAD (core/backbones.py:207-255): l2-normalise, soft assignment with batch-norm,
                     attention weighting, VLAD contraction -- returns (sum_n a x^T per cloud, sum_n a per cloud)

Sharded batches: `mask` ([clouds] bool) marks this rank's real clouds, padding rows take no part in statistics and
get zero gradients; `sync` all-reduces the statistics (sum, sum of squares, count) and the backward sums over the ranks
so that they equal the reference's single-GPU whole-batch BatchNorm.  Parameter gradients are this rank's partials
(the trainer SUM-all-reduces them).
"""
import torch
import torch.distributed as dist

from . import dist as D
from . import pm


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


_CNT = {}


def _count(R, mask, rows_per_cloud, device):
    """Rows that take part in the statistics, as a float64 device scalar (cached: no host round trip per call)."""
    if mask is None:
        key = (str(device), int(R))
        if key not in _CNT:
            if torch.cuda.is_current_stream_capturing():
                # first sight of this row count inside a graph capture: a fill on the device (no pageable host copy,
                # which a capturing stream refuses) and NOT cached -- the tensor lives in the graph's private pool
                return torch.full((1,), float(R), dtype=torch.float64, device=device)
            _CNT[key] = torch.full((1,), float(R), dtype=torch.float64, device=device)
        return _CNT[key]
    return (mask.sum().to(torch.float64) * float(rows_per_cloud)).reshape(1)


def _once(ctx, what):
    """Several backward nodes overwrite a saved tensor with its gradient through raw pointers (autograd's version
    counters cannot see that): a second backward over the same graph would silently read garbage -- refuse it."""
    if getattr(ctx, "_dh3d_used", False):
        raise RuntimeError("dh3d_amd.train_ops.%s: this node's backward reuses its saved activations in place and can "
                           "run only once per forward (retain_graph / double backward are not supported)" % what)
    ctx._dh3d_used = True


class _BNState(object):
    """Forward statistics of one BatchNorm site: [mean, rstd, scale, shift] rows + the row count."""
    __slots__ = ("stats", "cnt")


def _forward_stats(x, gamma, beta, run_mean, run_var, eps, momentum, mask, rows_per_cloud, sync, unbiased=True):
    """Two launches (+ the all-reduce under sync-BN): column sums, then the finalize kernel."""
    C = x.shape[1]
    packed = pm.zeros((2 * C + 1,), torch.float64, x.device)
    s1, s2 = pm.bn_colstats(x, mask, rows_per_cloud, out=packed)
    cnt = _count(x.shape[0], mask, rows_per_cloud, x.device)
    if sync and D.collectives_active():
        packed[2 * C:] = cnt
        D.all_reduce_sum_(packed)
        cnt = packed[2 * C:]
    st = _BNState()
    st.stats = pm.bn_finalize(s1, s2, cnt, gamma, beta, eps, momentum, run_mean, run_var, unbiased)
    st.cnt = cnt
    return st


def _backward_coeffs_parts(part, st, gamma, sync):
    """part [nk, P, C] f64 per-cloud partials -> (grads [nk, C] float32 local sums: dbeta, dgamma(, d w_fc), [k2, k3]).
    One launch; under sync-BN the sums are all-reduced first (the long way)."""
    if sync and D.collectives_active():
        S = part.sum(1)
        dgamma, dbeta, k = _backward_coeffs(S, st, gamma, sync)
        grads = [dbeta, dgamma] + ([S[2].float()] if part.shape[0] > 2 else [])
        return grads, k
    k, grads = pm.bn_bwd_finalize_parts(part, st.cnt, st.stats[0], st.stats[1], gamma)
    return grads, k


def _backward_coeffs(S, st, gamma, sync):
    """S [3, C] f64 local sums -> (local dgamma, dbeta as float32, [k2, k3]) with the all-reduce under sync-BN."""
    if not (sync and D.collectives_active()):   # one launch: k2 / k3 and the float32 gradients
        k, grads = pm.bn_bwd_finalize_parts(S[:2].reshape(2, 1, S.shape[1]), st.cnt, st.stats[0], st.stats[1], gamma)
        return grads[1], grads[0], k
    local = S[:2].float()                       # this rank's partials: dbeta = S1, dgamma = S2
    if sync and D.collectives_active():
        S = S.clone()
        D.all_reduce_sum_(S[:2])
    k = pm.bn_bwd_finalize(S[0], S[1], st.cnt, st.stats[0], st.stats[1], gamma)
    return local[1], local[0], k


class _BatchNormTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, eps, momentum, relu, sync, mask, rows_per_cloud, unbiased):
        x = x.contiguous()
        g, be = gamma.detach().contiguous(), beta.detach().contiguous()
        st = _forward_stats(x, g, be, run_mean, run_var, eps, momentum, mask, rows_per_cloud, sync, unbiased)
        y = pm.scale_shift_act(x, st.stats[2], st.stats[3], relu)
        ctx.save_for_backward(x, g, be)
        ctx.cfg = (bool(relu), bool(sync), mask, int(rows_per_cloud), st)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, be = ctx.saved_tensors
        relu, sync, mask, rpc, st = ctx.cfg
        dy = dy.contiguous()
        S = pm.bn_bwd_sums(x, st.stats[0], st.stats[1], g, be, relu, dy=dy, mask=mask, rows_per_cloud=rpc)
        dgamma, dbeta, k = _backward_coeffs(S, st, g, sync)
        dx = pm.bn_bwd_apply(x, st.stats[2], st.stats[3], k[0], k[1], relu, dy=dy, mask=mask, rows_per_cloud=rpc)
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, None, None


class _BatchNormTrainSmall(torch.autograd.Function):
    """_BatchNormTrain for a short tensor (R <= 64 rows, one row per cloud): one launch per direction."""

    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, eps, momentum, relu, mask, unbiased):
        from . import _lib as L
        x = x.contiguous()
        R, C = x.shape
        g, be = gamma.detach().contiguous(), beta.detach().contiguous()
        stats = torch.empty((4, C), dtype=torch.float32, device=x.device)
        y = torch.empty_like(x)
        m8 = pm._mask_u8(mask)
        L.check(L.lib().dh3d_bn_small_fwd(L.ptr(x), R, C, L.ptr(g), L.ptr(be), float(eps), float(momentum),
                                          1 if unbiased else 0, 1 if relu else 0, L.ptr(m8), L.ptr(run_mean), L.ptr(run_var), L.ptr(stats),
                                          L.ptr(y), L.stream_ptr()), "bn_small_fwd")
        ctx.save_for_backward(x, g, stats)
        ctx.cfg = (bool(relu), m8)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import _lib as L
        x, g, stats = ctx.saved_tensors
        relu, m8 = ctx.cfg
        R, C = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dgb = torch.empty((2, C), dtype=torch.float32, device=x.device)
        L.check(L.lib().dh3d_bn_small_bwd(L.ptr(x), L.ptr(dy), R, C, L.ptr(g), L.ptr(stats), 1 if relu else 0, L.ptr(m8),
                                          L.ptr(dx), L.ptr(dgb[0]), L.ptr(dgb[1]), L.stream_ptr()), "bn_small_bwd")
        return dx, dgb[0], dgb[1], None, None, None, None, None, None, None


def batch_norm_train(x, bnmod, relu, sync=False, mask=None, rows_per_cloud=0, momentum=None):
    """x [R, C] -> act(BN_train(x)); bnmod: backbones.TPBatchNorm / SlimBatchNorm (running buffers updated in place:
    decay 0.9 / 0.999 like tensorpack / slim; the moving variance takes the Bessel-corrected batch variance where the
    upstream layer is a fused batch norm -- bnmod.ema_unbiased).  mask [clouds] bool with rows_per_cloud rows each, or
    None."""
    from . import backbones as bb
    tp = isinstance(bnmod, bb.TPBatchNorm)
    rm, rv = (bnmod.mean_EMA, bnmod.variance_EMA) if tp else (bnmod.moving_mean, bnmod.moving_variance)
    mom = momentum if momentum is not None else (0.9 if tp else 0.999)
    unb = bool(getattr(bnmod, "ema_unbiased", True))
    if x.shape[0] <= 64 and (mask is None or rows_per_cloud == 1) and not (sync and D.collectives_active()):
        # the [clouds, C] activations behind NetVLAD: one launch per direction (single rank or per-rank statistics)
        return _BatchNormTrainSmall.apply(x, bnmod.gamma, bnmod.beta, rm, rv, bnmod.eps, mom, relu, mask, unb)
    return _BatchNormTrain.apply(x, bnmod.gamma, bnmod.beta, rm, rv, bnmod.eps, mom, relu, sync, mask, rows_per_cloud, unb)


class _AddChannelBias(torch.autograd.Function):
    """x [..., C] + bias [C]; the bias gradient is one column-sum launch (autograd's own is a generic reduction over the
    broadcast dimensions: 25 us for [22*512, 256])."""

    @staticmethod
    def forward(ctx, x, bias):
        return x + bias.reshape((1,) * (x.dim() - 1) + (-1,))

    @staticmethod
    def backward(ctx, dy):
        db = pm.colsum(dy.reshape(-1, dy.shape[-1]).contiguous()) if ctx.needs_input_grad[1] else None
        return dy, db


def add_channel_bias(x, bias):
    return _AddChannelBias.apply(x, bias)


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b):
        x = x.contiguous()
        y = pm.gemm_nn(x, W.detach().contiguous(), bias=None if b is None else b.detach().contiguous())
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = dy.contiguous()
        dx = pm.gemm_nn(dy, pm.transpose_last2(W.detach())) if ctx.needs_input_grad[0] else None
        dW = pm.gemm_tn(x, dy) if ctx.needs_input_grad[1] else None
        db = pm.colsum(dy) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dW, db


def linear(x, W, b=None):
    """x [R, Cin] @ W [Cin, Cout] (+ b) with hand-written GEMMs in both directions (Cin, Cout multiples of 4)."""
    return _Linear.apply(x, W, b)


class _AttentionHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, W, b, gamma, beta, run_mean, run_var, eps, momentum, wfc, bfc, sync, mask, rows_per_cloud):
        X = X.contiguous()
        h = pm.gemm_nn(X, W.detach().contiguous(), bias=b.detach().contiguous())          # [R, H] pre-activation
        g, be = gamma.detach().contiguous(), beta.detach().contiguous()
        st = _forward_stats(h, g, be, run_mean, run_var, eps, momentum, mask, rows_per_cloud, sync)
        att = pm.row_logit_sigmoid(h, st.stats[2], st.stats[3], wfc.detach().reshape(-1).contiguous(),
                                   bfc.detach().reshape(-1).contiguous())
        ctx.save_for_backward(X, W, h, g, be, wfc, att)
        ctx.cfg = (bool(sync), mask, int(rows_per_cloud), st)
        return att

    @staticmethod
    def backward(ctx, datt):
        X, W, h, g, be, wfc, att = ctx.saved_tensors
        sync, mask, rpc, st = ctx.cfg
        _once(ctx, "attention_head")  # h is overwritten by its gradient below
        dlogit = (datt * att * (1.0 - att)).contiguous()                                # sigmoid
        if mask is not None:
            dlogit = dlogit * mask.repeat_interleave(rpc).to(dlogit.dtype)
        wv = wfc.detach().reshape(-1).contiguous()
        S = pm.bn_bwd_sums(h, st.stats[0], st.stats[1], g, be, True, rowscale=dlogit, colvec=wv, mask=mask,
                           rows_per_cloud=rpc)
        dwfc = S[2].float().reshape(wfc.shape)
        dbfc = dlogit.sum().reshape(1)
        dgamma, dbeta, k = _backward_coeffs(S, st, g, sync)
        dh = pm.bn_bwd_apply(h, st.stats[2], st.stats[3], k[0], k[1], True, rowscale=dlogit, colvec=wv, mask=mask,
                             rows_per_cloud=rpc, out=h)                                  # in place: h is dead after this
        dW = pm.gemm_tn(X, dh)
        # d/db: the BatchNorm that follows removes any per-channel constant, the exact gradient is 0 (the column sums
        # of dh are rounding noise)
        db = torch.zeros_like(g)
        dX = pm.gemm_nn(dh, pm.transpose_last2(W.detach())) if ctx.needs_input_grad[0] else None
        return dX, dW, db, dgamma, dbeta, None, None, None, None, dwfc, dbfc, None, None, None


def attention_head(X, conv, wfc, bfc, sync=False, mask=None, rows_per_cloud=0):
    """globalatt_block on rows X [R, Cin]: conv = backbones.Conv2D1x1 (W, b, bn), wfc [H,1] / bfc [1] the final layer.
    Returns att [R]."""
    bn = conv.bn
    return _AttentionHead.apply(X, conv.W.reshape(conv.cin, conv.cout), conv.b, bn.gamma, bn.beta, bn.mean_EMA,
                                bn.variance_EMA, bn.eps, 0.9, wfc, bfc, sync, mask, rows_per_cloud)


class _VladNormalize(torch.autograd.Function):
    """V [Bt, Cl, D], asum [Bt, Cl], W2 [1, D, Cl] -> [Bt, D*Cl]: subtract asum * cluster_weights2, intra-normalise per
    cluster, flatten, L2-normalise (core/backbones.py:241-262) -- one launch per direction (csrc/train.hip) instead of
    ~37 tiny tensor ops."""

    @staticmethod
    def forward(ctx, V, asum, W2):
        from . import _lib as L
        V, asum = V.contiguous(), asum.contiguous()
        W2d = W2.detach().reshape(W2.shape[-2], W2.shape[-1]).contiguous()
        Bt, Cl, Dm = V.shape
        out = torch.empty((Bt, Dm * Cl), dtype=torch.float32, device=V.device)
        aux = torch.empty((Bt * (Cl + 1),), dtype=torch.float32, device=V.device)
        L.check(L.lib().dh3d_vlad_normalize_fwd(L.ptr(V), L.ptr(asum), L.ptr(W2d), Bt, Dm, Cl, 1e-12, L.ptr(out),
                                                L.ptr(aux), L.ptr(aux[Bt * Cl:]), L.stream_ptr()), "vlad_normalize")
        ctx.save_for_backward(V, asum, W2d)
        ctx.w2shape = W2.shape
        return out

    @staticmethod
    def backward(ctx, g):
        from . import _lib as L
        V, asum, W2d = ctx.saved_tensors
        Bt, Cl, Dm = V.shape
        g = g.contiguous()
        dV, dasum, dW2 = torch.empty_like(V), torch.empty_like(asum), torch.empty_like(W2d)
        L.check(L.lib().dh3d_vlad_normalize_bwd(L.ptr(V), L.ptr(asum), L.ptr(W2d), L.ptr(g), Bt, Dm, Cl, 1e-12, L.ptr(dV),
                                                L.ptr(dasum), L.ptr(dW2), L.stream_ptr()), "vlad_normalize_bwd")
        return dV, dasum, dW2.reshape(ctx.w2shape)


def vlad_normalize(V, asum, W2):
    return _VladNormalize.apply(V, asum, W2)


def vlad_normalize_supported(V):
    return V.dim() == 3 and V.shape[1] == 64 and V.shape[2] == 256


class _QuadrupletLoss(torch.autograd.Function):
    """losses.lazy_quadruplet_loss and its gradient in one launch (csrc/train.hip quadruplet_loss_kernel)."""

    @staticmethod
    def forward(ctx, desc, B, P, Ng, m1, m2):
        from . import _lib as L
        d = desc.contiguous()
        loss = torch.empty((1,), dtype=torch.float32, device=d.device)
        grad = torch.empty_like(d)
        L.check(L.lib().dh3d_quadruplet_loss(L.ptr(d), B, P, Ng, d.shape[1], float(m1), float(m2), L.ptr(loss), L.ptr(grad),
                                             L.stream_ptr()), "quadruplet_loss")
        ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None, None


def quadruplet_loss(desc, batch_size, num_pos, num_neg, margin=0.5, margin2=0.2):
    """desc [B*(2 + P + Ng), 256] role-ordered -> scalar loss (core/losses.py:173-200)."""
    if desc.shape[0] != batch_size * (2 + num_pos + num_neg):
        raise ValueError("descriptor rows %d do not match the role sizes" % desc.shape[0])
    return _QuadrupletLoss.apply(desc, batch_size, num_pos, num_neg, margin, margin2)


def quadruplet_loss_supported(desc, num_pos, num_neg):
    return desc.is_cuda and desc.dim() == 2 and desc.shape[1] == 256 and num_pos <= 8 and num_neg <= 64


class _L2NormalizeRows(torch.autograd.Function):
    """x [R, C] -> x * rsqrt(max(sum x^2, eps)) (tf.nn.l2_normalize), one launch per direction."""

    @staticmethod
    def forward(ctx, x, eps):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.eps = eps
        return pm.l2norm_concat(x, eps)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return pm.l2norm_rows_bwd(x, g.contiguous(), ctx.eps), None


def l2_normalize_rows(x, eps):
    return _L2NormalizeRows.apply(x, eps)


class _ThreeInterpolateSorted(torch.autograd.Function):
    """ops.three_interpolate with the backward on the Morton order of the fine cloud (csrc/interp_train.hip MODE 3)."""

    @staticmethod
    def forward(ctx, points, idx, weight, order):
        from . import _lib as L
        p = points.contiguous()
        b, m, c = p.shape
        n = idx.shape[1]
        out = torch.empty((b, n, c), dtype=torch.float32, device=p.device)
        L.check(L.lib().dh3d_three_interpolate_fwd(b, m, c, n, L.ptr(p), L.ptr(idx), L.ptr(weight), L.ptr(out),
                                                   L.stream_ptr()), "three_interpolate")
        ctx.save_for_backward(idx, weight, order)
        ctx.pshape = (b, m, c)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from . import _lib as L
        idx, weight, order = ctx.saved_tensors
        b, m, c = ctx.pshape
        go = grad_out.contiguous()
        gp = torch.empty((b, m, c), dtype=torch.float32, device=go.device)
        L.check(L.lib().dh3d_three_interpolate_bwd_sorted(b, idx.shape[1], c, m, L.ptr(go), L.ptr(idx), L.ptr(weight),
                                                          L.ptr(order), L.ptr(gp), L.stream_ptr()),
                "three_interpolate_bwd_sorted")
        return gp, None, None, None


def three_interpolate_sorted(points, idx, weight, order):
    """points [b,m,256], idx / weight [b,n,3], order = spatial_sort records of the fine cloud [b,n,4] -> [b,n,256]."""
    return _ThreeInterpolateSorted.apply(points, idx.contiguous(), weight.contiguous(), order.contiguous())


class _AttentionHeadCommuted(torch.autograd.Function):
    """The same head on the up-sampled rows three_interpolate(C) WITHOUT building them: conv(interp(C)) = interp(conv(C)),
    so the three big GEMMs run on the sampled rows C [Bt*M, Cin] -- one launch each, G = C W + b in its natural
    [Bt*M, H] layout -- and the [Bt*N, H] pre-activation only exists inside the kernels of csrc/interp_train.hip
    (statistics, forward, backward sums, backward apply + scatter)."""

    @staticmethod
    def forward(ctx, C, W, b, gamma, beta, run_mean, run_var, eps, momentum, wfc, bfc, sync, mask, idx, dist, order):
        C = C.contiguous()
        Wd = W.detach().contiguous()
        H = Wd.shape[1]
        G = pm.gemm_nn(C, Wd, bias=b.detach().contiguous())                             # [Bt*M, H]
        Bt, N = idx.shape[0], idx.shape[1]
        g, be = gamma.detach().contiguous(), beta.detach().contiguous()
        cnt = _count(Bt * N, mask, N, C.device)
        st = _BNState()
        if sync and D.collectives_active():
            packed = torch.empty((2 * H + 1,), dtype=torch.float64, device=C.device)
            s1, s2 = pm.interp_bn_colstats(G, idx, dist, order, mask, out=packed)
            packed[2 * H:] = cnt
            D.all_reduce_sum_(packed)
            cnt = packed[2 * H:]
            st.stats = pm.bn_finalize(s1, s2, cnt, g, be, eps, momentum, run_mean, run_var)
        else:  # the per-cloud partial rows straight into the finalize kernel
            part = pm.interp_bn_colstats(G, idx, dist, order, mask, parts=True)
            st.stats = pm.bn_finalize_parts(part, cnt, g, be, eps, momentum, run_mean, run_var)
        st.cnt = cnt
        wv = wfc.detach().reshape(-1).contiguous()
        att = pm.interp_head_rows(G, idx, dist, order, st.stats[2], st.stats[3], wv, bfc.detach().reshape(-1).contiguous())
        ctx.save_for_backward(C, Wd, G, g, be, wv, att)
        ctx.cfg = (bool(sync), mask, idx, dist, order, st, W.shape, wfc.shape)
        return att

    @staticmethod
    def backward(ctx, datt):
        C, Wd, G, g, be, wv, att = ctx.saved_tensors
        sync, mask, idx, dist, order, st, wshape, wfcshape = ctx.cfg
        N = idx.shape[1]
        dlogit, dbfc = pm.sigmoid_bwd(datt, att, mask, N)
        part = pm.interp_bn_bwd_sums(G, idx, dist, order, dlogit, wv, st.stats[0], st.stats[1], g, be, mask, parts=True)
        grads, k = _backward_coeffs_parts(part, st, g, sync)
        dbeta, dgamma, dwfc = grads[0], grads[1], grads[2].reshape(wfcshape)
        dG = pm.interp_bn_bwd_apply(G, idx, dist, order, dlogit, wv, st.stats[2], st.stats[3], k[0], k[1], mask)
        dW = pm.gemm_tn(C, dG).reshape(wshape)                                           # [Cin, H]
        dC = pm.gemm_nn(dG, pm.transpose_last2(Wd)) if ctx.needs_input_grad[0] else None
        db = torch.zeros_like(g)   # the BatchNorm removes any per-channel constant: exact gradient 0
        return dC, dW, db, dgamma, dbeta, None, None, None, None, dwfc, dbfc, None, None, None, None, None


def attention_head_commuted(coarse, conv, wfc, bfc, idx, dist, order, sync=False, mask=None):
    """globalatt_block on the rows three_interpolate(coarse [Bt*M, Cin]) (inverse-distance weights of `dist`), never
    materialised.  idx / dist [Bt,N,3] int32 / float32, order = spatial_sort records of the fine clouds [Bt,N,4].
    Returns att [Bt*N]."""
    bn = conv.bn
    return _AttentionHeadCommuted.apply(coarse, conv.W.reshape(conv.cin, conv.cout), conv.b, bn.gamma, bn.beta,
                                        bn.mean_EMA, bn.variance_EMA, bn.eps, 0.9, wfc, bfc, sync, mask, idx, dist, order)


def attention_commute_supported(conv, M):
    return conv.cout % 256 == 0 and 256 <= conv.cout <= 1024 and conv.cin % 4 == 0 and M <= 1024


class _NetVLADAssign(torch.autograd.Function):
    """x [Bt, N, D] rows, att [Bt*N] -> (V [Bt, Cl, D] = sum_n a[n,c] xn[n,d],  asum [Bt, Cl] = sum_n a[n,c])."""

    @staticmethod
    def forward(ctx, x, att, Wc, gamma, beta, run_mean, run_var, eps, momentum, sync, mask):
        Bt, N, Dm = x.shape
        x2 = x.reshape(Bt * N, Dm).contiguous()
        xn = pm.l2norm_concat(x2, 1e-12)                                   # tf.nn.l2_normalize(reshaped_input, 1)
        s = pm.gemm_nn(xn, Wc.detach().contiguous())                       # [R, Cl]
        g, be = gamma.detach().contiguous(), beta.detach().contiguous()
        # cluster_bn is the one un-fused batch norm upstream (core/backbones.py:218-223): biased moving variance
        st = _forward_stats(s, g, be, run_mean, run_var, eps, momentum, mask, N, sync, unbiased=False)
        scale, shift = st.stats[2], st.stats[3]
        att = att.contiguous()
        if N % 64 == 0:   # softmax(bn(s)) * att, and its per-cloud column sums from the same pass
            a, asum = pm.netvlad_assign_rows(s, scale, shift, att, rows_per_cloud=N)
        else:
            a = pm.netvlad_assign_rows(s, scale, shift, att)
            asum = a.reshape(Bt, N, a.shape[1]).sum(1)
        Cl = a.shape[1]
        V = pm.gemm_tn_batched(a.reshape(Bt, N, Cl), xn.reshape(Bt, N, Dm))   # [Bt, Cl, D]
        ctx.save_for_backward(x2, xn, s, a, att, Wc, g, be)
        ctx.cfg = (Bt, N, bool(sync), mask, st)
        return V, asum

    @staticmethod
    def backward(ctx, dV, dasum):
        x2, xn, s, a, att, Wc, g, be = ctx.saved_tensors
        Bt, N, sync, mask, st = ctx.cfg
        scale, shift = st.stats[2], st.stats[3]
        Dm, Cl = xn.shape[1], a.shape[1]
        dV = dV.contiguous()
        # da[n,c] = sum_d dV[c,d] xn[n,d] + dasum[c];   dxn[n,d] = sum_c a[n,c] dV[c,d]
        da = pm.gemm_nn_batched(xn.reshape(Bt, N, Dm), pm.transpose_last2(dV), bias=dasum.contiguous())
        dxn = pm.gemm_nn_batched(a.reshape(Bt, N, Cl), dV).reshape(Bt * N, Dm)
        dz, datt = pm.netvlad_assign_rows_bwd(s, scale, shift, att, da.reshape(Bt * N, Cl))
        S = pm.bn_bwd_sums(s, st.stats[0], st.stats[1], g, be, False, dy=dz, mask=mask, rows_per_cloud=N)
        dgamma, dbeta, k = _backward_coeffs(S, st, g, sync)
        ds = pm.bn_bwd_apply(s, scale, shift, k[0], k[1], False, dy=dz, mask=mask, rows_per_cloud=N, out=dz)
        dWc = pm.gemm_tn(xn, ds)
        pm.gemm_nn(ds, pm.transpose_last2(Wc.detach()), out=dxn, accumulate=True)
        dx = pm.l2norm_rows_bwd(x2, dxn, 1e-12).reshape(Bt, N, Dm)
        if mask is not None:
            datt = datt * mask.repeat_interleave(N).to(datt.dtype)
        return dx, datt, dWc, dgamma, dbeta, None, None, None, None, None, None


class _NetVLADAssignCommuted(torch.autograd.Function):
    """_NetVLADAssign on the rows three_interpolate(c) WITHOUT building them (csrc/netvlad_train.hip): c [Bt, M, 256]
    sampled rows, att [Bt*N]; idx / dist [Bt,N,3] three_nn of the fine points, order = spatial_sort records of the fine
    clouds -> (V [Bt, 64, 256], asum [Bt, 64]).  GEMMs on the Bt*M sampled rows, 64-wide walks over the fine points."""

    @staticmethod
    def forward(ctx, c, att, Wc, gamma, beta, run_mean, run_var, eps, momentum, sync, mask, idx, dist, order):
        Bt, M, Dm = c.shape
        N = idx.shape[1]
        c2 = c.reshape(Bt * M, Dm).contiguous()
        Wd = Wc.detach().contiguous()
        cw = pm.gemm_nn(c2, Wd)                                            # [Bt*M, 64]
        g, be = gamma.detach().contiguous(), beta.detach().contiguous()
        cnt = _count(Bt * N, mask, N, c.device)
        st = _BNState()
        # cluster_bn is the one un-fused batch norm upstream (core/backbones.py:218-223): biased moving variance
        if sync and D.collectives_active():
            packed = torch.empty((2 * 64 + 1,), dtype=torch.float64, device=c.device)
            s, rinv, s1, s2 = pm.nv_commuted_fwd_stats(c2, cw, idx, dist, order, mask, out=packed)
            packed[128:] = cnt
            D.all_reduce_sum_(packed)
            cnt = packed[128:]
            st.stats = pm.bn_finalize(s1, s2, cnt, g, be, eps, momentum, run_mean, run_var, unbiased=False)
        else:
            s, rinv, part = pm.nv_commuted_fwd_stats(c2, cw, idx, dist, order, mask, parts=True)
            st.stats = pm.bn_finalize_parts(part, cnt, g, be, eps, momentum, run_mean, run_var, unbiased=False)
        st.cnt = cnt
        att = att.contiguous()
        p, asum, Ap = pm.nv_commuted_fwd_assign(s, rinv, att, st.stats[2], st.stats[3], idx, dist, order, M, mask)
        V = pm.gemm_tn_batched(Ap.reshape(Bt, M, 64), c2.reshape(Bt, M, Dm))   # [Bt, 64, 256]
        ctx.save_for_backward(c2, s, rinv, p, att, Wd, g, Ap)
        ctx.cfg = (Bt, M, bool(sync), mask, st, idx, dist, order)
        return V, asum

    @staticmethod
    def backward(ctx, dV, dasum):
        c2, s, rinv, p, att, Wd, g, Ap = ctx.saved_tensors
        Bt, M, sync, mask, st, idx, dist, order = ctx.cfg
        Dm = c2.shape[1]
        dV = dV.contiguous()
        E = pm.gemm_nn_batched(c2.reshape(Bt, M, Dm), pm.transpose_last2(dV)).reshape(Bt * M, 64)
        dz, datt, t2, part = pm.nv_commuted_bwd_sums(E, p, s, att, rinv, dasum.contiguous(), st.stats[0], st.stats[1],
                                                     idx, dist, order, mask, parts=True)
        (dbeta, dgamma), k = _backward_coeffs_parts(part, st, g, sync)
        q, dcw = pm.nv_commuted_bwd_apply(dz, s, rinv, t2, st.stats[2], k[0], k[1], idx, dist, order, M, mask)
        dWc = pm.gemm_tn(c2, dcw)
        dc = pm.gemm_nn_batched(Ap.reshape(Bt, M, 64), dV).reshape(Bt * M, Dm)
        pm.gemm_nn(dcw, pm.transpose_last2(Wd), out=dc, accumulate=True)
        pm.interp_scatter_scaled(c2, q, idx, dist, order, dc, mask)
        return dc.reshape(Bt, M, Dm), datt, dWc, dgamma, dbeta, None, None, None, None, None, None, None, None, None


def netvlad_assign_commuted(c, att, Wc, bnmod, idx, dist, order, sync=False, mask=None):
    return _NetVLADAssignCommuted.apply(c, att, Wc, bnmod.gamma, bnmod.beta, bnmod.moving_mean, bnmod.moving_variance,
                                        bnmod.eps, 0.999, sync, mask, idx.contiguous(), dist.contiguous(),
                                        order.contiguous())


def netvlad_commute_supported(c, Wc):
    return c.shape[2] == 256 and Wc.shape[1] == 64 and c.shape[1] <= 1024


class _ContextGate(torch.autograd.Function):
    """v * sigmoid(g) (core/backbones.py:271-277): one launch per direction."""

    @staticmethod
    def forward(ctx, v, g):
        v, g = v.contiguous(), g.contiguous()
        ctx.save_for_backward(v, g)
        return pm.context_gate(v, g)

    @staticmethod
    def backward(ctx, dy):
        v, g = ctx.saved_tensors
        return pm.context_gate_bwd(v, g, dy.contiguous())


def context_gate(v, g):
    return _ContextGate.apply(v, g)


def netvlad_assign(x, att, Wc, bnmod, sync=False, mask=None):
    return _NetVLADAssign.apply(x, att, Wc, bnmod.gamma, bnmod.beta, bnmod.moving_mean, bnmod.moving_variance, bnmod.eps,
                                0.999, sync, mask)


# ----------------------------------------------------------------------------------------------------------------------
# Local backbone in training mode (stage 1-2 training: basic_config / detection_config, core/model.py:212-246): the nodes
# the frozen-backbone step never needed -- conv_pointset, flex_pool, the SE gate, a plain ReLU -- point-major, forward and
# backward on HIP kernels (include/dh3d_hip.h: dh3d_pointset_sum_pm, dh3d_flex_pool_pm_bwd, dh3d_se_gate_*, dh3d_relu_*).
from . import _lib as _L


class _ConvPointsetXYZ(torch.autograd.Function):
    """conv_pointset on the coordinates (Din = 3): out = theta^T S + bias with S[n] = sum_k (p[n_k] - p[n_0])
    (conv_pointset_kernel.cc:46-64).  Backward = ConvPointsetGrad (conv_pointset_kernel_gpu.cu.cc:157-347) in its
    factorised form: grad_theta = S^T grad_out (one split-reduction GEMM over the rows), grad_bias = column sums; the
    coordinates are data."""

    @staticmethod
    def forward(ctx, xyz, nbr, theta, bias):
        out = pm.conv_pointset_xyz(xyz, nbr, theta.detach().contiguous(), bias.detach().contiguous())
        ctx.save_for_backward(xyz, nbr)
        return out

    @staticmethod
    def backward(ctx, dout):
        xyz, nbr = ctx.saved_tensors
        B, N, _ = xyz.shape
        d2 = dout.reshape(B * N, -1).contiguous()
        S = torch.empty((B * N, 4), dtype=torch.float32, device=xyz.device)
        _L.check(_L.lib().dh3d_pointset_sum_pm(_L.ptr(xyz), _L.ptr(nbr), B, N, nbr.shape[2], _L.ptr(S), _L.stream_ptr()),
                 "pointset_sum")
        dtheta = pm.gemm_tn(S, d2)[:3]          # [4, Dout]: the fourth row of S is zero padding
        return None, None, dtheta, pm.colsum(d2)


def conv_pointset_xyz(xyz, nbr, theta, bias):
    return _ConvPointsetXYZ.apply(xyz.contiguous(), nbr.contiguous(), theta, bias)


class _FlexPoolPM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, nbr):
        out, arg = pm.flex_pool(x.contiguous(), nbr, want_argmax=True)
        ctx.save_for_backward(arg)
        ctx.n = x.shape[1]
        return out

    @staticmethod
    def backward(ctx, dout):
        (arg,) = ctx.saved_tensors
        dout = dout.contiguous()
        B, N, C = dout.shape
        din = pm.zeros((B, N, C), torch.float32, dout.device)
        _L.check(_L.lib().dh3d_flex_pool_pm_bwd(_L.ptr(dout), _L.ptr(arg), B, N, C, _L.ptr(din), _L.stream_ptr()),
                 "flex_pool_pm_bwd")
        return din, None


def flex_pool(x, nbr):
    """x [B,N,C], nbr [B,N,K] -> max over the neighbourhood, differentiable (FlexPool / FlexPoolGrad)."""
    return _FlexPoolPM.apply(x, nbr)


class _SEGate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, z):
        x, z = x.contiguous(), z.contiguous()
        y = torch.empty_like(x)
        _L.check(_L.lib().dh3d_se_gate_fwd(_L.ptr(x), _L.ptr(z), x.numel(), _L.ptr(y), _L.stream_ptr()), "se_gate_fwd")
        ctx.save_for_backward(x, z)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, z = ctx.saved_tensors
        dy = dy.contiguous()
        dx, dz = torch.empty_like(x), torch.empty_like(z)
        _L.check(_L.lib().dh3d_se_gate_bwd(_L.ptr(x), _L.ptr(z), _L.ptr(dy), x.numel(), _L.ptr(dx), _L.ptr(dz),
                                           _L.stream_ptr()), "se_gate_bwd")
        return dx, dz


def se_gate(x, z):
    """relu(x + x * sigmoid(z))  (core/backbones.py:52-55)."""
    return _SEGate.apply(x, z)


class _ReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        _L.check(_L.lib().dh3d_relu_fwd(_L.ptr(x), x.numel(), _L.ptr(y), _L.stream_ptr()), "relu_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        _L.check(_L.lib().dh3d_relu_bwd(_L.ptr(y), _L.ptr(dy), y.numel(), _L.ptr(dx), _L.stream_ptr()), "relu_bwd")
        return dx


def relu(x):
    return _ReLU.apply(x)


class _PairwiseSqDist(torch.autograd.Function):
    """pairwise_dist of the local losses (core/tf_utils.py:125-136): [B,n,D], [B,m,D] -> [B,n,m].  Forward: the HIP kernel
    (differences formed as upstream).  Backward in closed form on the batched GEMM kernels:
    dA = 2 (rowsum(G) * A - G B),  dB = 2 (colsum(G) * B - G^T A)."""

    @staticmethod
    def forward(ctx, A, Bm):
        A, Bm = A.contiguous(), Bm.contiguous()
        Bt, n, D = A.shape
        m = Bm.shape[1]
        out = torch.empty((Bt, n, m), dtype=torch.float32, device=A.device)
        _L.check(_L.lib().dh3d_pairwise_sqdist(_L.ptr(A), _L.ptr(Bm), Bt, n, m, D, _L.ptr(out), _L.stream_ptr()),
                 "pairwise_sqdist")
        ctx.save_for_backward(A, Bm)
        return out

    @staticmethod
    def backward(ctx, G):
        A, Bm = ctx.saved_tensors
        G = G.contiguous()
        dA = dB = None
        n, m, D = A.shape[1], Bm.shape[1], A.shape[2]
        hip = n % 4 == 0 and m % 4 == 0 and D % 4 == 0   # the batched GEMM kernels' shapes (the losses': 512 x 512 x 128)
        if ctx.needs_input_grad[0]:
            GB = pm.gemm_nn_batched(G, Bm) if hip else torch.einsum("bij,bjd->bid", G, Bm)
            dA = 2.0 * (G.sum(2, keepdim=True) * A - GB)
        if ctx.needs_input_grad[1]:
            GA = pm.gemm_tn_batched(G, A) if hip else torch.einsum("bij,bid->bjd", G, A)
            dB = 2.0 * (G.sum(1).unsqueeze(2) * Bm - GA)
        return dA, dB


def pairwise_sqdist(A, Bm):
    return _PairwiseSqDist.apply(A, Bm)
