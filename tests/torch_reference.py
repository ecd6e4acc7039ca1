"""Plain tensor-op restatements of the training step's blocks -- TEST REFERENCE ONLY.

Until round 5 these lived in dh3d_amd/training.py behind `QuadrupletTrainer(impl="torch")`; a product package must not
carry a second, run-time-selectable implementation of its own hot path, so they live with the tests that compare the
HIP training kernels (dh3d_amd.train_ops) against them.  Nothing under dh3d_amd/ imports this module
(tests/test_abi.py guards it).

  batch_norm_train            training-mode BatchNorm as tensor ops (tensorpack / slim semantics, sync-BN capable)
  backbone_local_batch_stats  the frozen local backbone on BATCH statistics, layer by layer (core/backbones.py:104-127
                              under core/tf_utils.py:145-153)
  global_head_autograd        compute_global (core/model.py:112-133) with torch autograd
  TorchQuadrupletTrainer      QuadrupletTrainer whose head / backbone-BN / normalisation hooks run the functions above
"""
import torch
import torch.nn.functional as F

from dh3d_amd import backbones as bb
from dh3d_amd import dist as D
from dh3d_amd import losses, ops, pm
from dh3d_amd.training import QuadrupletTrainer, _AllReduceSum, flex_conv_factorised


def batch_norm_train(x, channel_dim, gamma, beta, run_mean, run_var, eps, momentum, sync_bn, mask=None, unbiased=True):
    """Training-mode BatchNorm over all dims but `channel_dim`; updates the running buffers in place.
    `momentum` is the EMA decay (tensorpack 0.9, slim 0.999); `unbiased`: the moving variance takes the
    Bessel-corrected batch variance (tf.nn.fused_batch_norm) -- every site but cluster_bn.  `mask` ([leading] bool)
    drops padding clouds."""
    dims = [d for d in range(x.dim()) if d != channel_dim]
    shape = [1] * x.dim()
    shape[channel_dim] = -1
    if mask is not None:
        w = mask.to(x.dtype).reshape([-1] + [1] * (x.dim() - 1))
        cnt = w.sum() * (x.numel() / (x.shape[0] * x.shape[channel_dim]))
        s1 = (x * w).sum(dims)
        s2 = (x * x * w).sum(dims)
    else:
        cnt = torch.tensor(float(x.numel() / x.shape[channel_dim]), device=x.device)
        s1 = x.sum(dims)
        s2 = (x * x).sum(dims)
    if sync_bn and D.collectives_active():
        packed = _AllReduceSum.apply(torch.cat([s1, s2, cnt.reshape(1)]))
        C = s1.numel()
        s1, s2, cnt = packed[:C], packed[C:2 * C], packed[2 * C]
    # a rank that holds only padding clouds (e.g. 22 clouds over 12 or 16 ranks) has cnt == 0 without sync_bn: its
    # statistics are 0/0.  Guard the division and leave its running buffers alone -- every row of x is masked out of
    # the loss there, so its gradients are exact zeros instead of NaN that the SUM all-reduce would spread.
    empty = cnt <= 0
    cnt = cnt.clamp_min(1.0)
    mean = s1 / cnt
    var = (s2 / cnt - mean * mean).clamp_min(0.0)
    with torch.no_grad():
        keep = empty.to(mean.dtype)  # 1 -> buffers unchanged
        run_mean.copy_(keep * run_mean + (1 - keep) * (momentum * run_mean + (1 - momentum) * mean.detach()))
        uvar = var.detach() * (cnt / (cnt - 1).clamp_min(1.0)) if unbiased else var.detach()
        run_var.copy_(keep * run_var + (1 - keep) * (momentum * run_var + (1 - momentum) * uvar))
    return (x - mean.reshape(shape)) * torch.rsqrt(var.reshape(shape) + eps) * gamma.reshape(shape) + beta.reshape(shape)


def _bn(x, channel_dim, bnmod, training, sync_bn, mask):
    tp = isinstance(bnmod, bb.TPBatchNorm)
    rm, rv = (bnmod.mean_EMA, bnmod.variance_EMA) if tp else (bnmod.moving_mean, bnmod.moving_variance)
    if training:
        return batch_norm_train(x, channel_dim, bnmod.gamma, bnmod.beta, rm, rv, bnmod.eps, 0.9 if tp else 0.999,
                                 sync_bn, mask, bool(getattr(bnmod, "ema_unbiased", True)))
    shape = [1] * x.dim()
    shape[channel_dim] = -1
    return (x - rm.reshape(shape)) * torch.rsqrt(rv.reshape(shape) + bnmod.eps) * bnmod.gamma.reshape(shape) + \
        bnmod.beta.reshape(shape)



@torch.no_grad()
def backbone_local_batch_stats(model, points, geo, sync_bn=False, mask=None):
    """The FROZEN local backbone as the reference runs it while global_config trains (core/backbones.py:104-127 under
    core/tf_utils.py:145-153 freeze_variables(stop_gradient=False, skip_collection=True)): frozen means 'not in the
    TRAINABLE collection' -- every BatchNorm still normalises with the statistics of the batch and updates its moving
    averages.  Layer by layer on the fused kernels WITHOUT their folded BatchNorm epilogues (raw flex_conv /
    conv_pointset / pooling outputs from HIP, the 1x1 convs, SE block and BatchNorm as tensor ops); the default
    trainer uses the fused inference path with moving averages instead (a documented deviation),
    this is `QuadrupletTrainer(backbone_bn="batch")`, compared with the oracle's training-mode graph in the tests.
    Returns (localdesc [b,N,128], geometry level)."""

    def bn_relu(x, bnmod, rows_dim):
        return F.relu(batch_norm_train(x, rows_dim, bnmod.gamma, bnmod.beta, bnmod.mean_EMA, bnmod.variance_EMA,
                                        bnmod.eps, 0.9, sync_bn, mask, True))

    def conv_bnrelu(x, fc1d):
        conv = fc1d.tfconv0
        return bn_relu(x @ conv.W.reshape(conv.cin, conv.cout) + conv.b, conv.bn, 2)

    def flex_stack(mod, x, xyz, nbr):
        for i in range(len(mod.outdims)):
            fc, bn = getattr(mod, "flexconv_%d" % i), getattr(mod, "flexconv_%d_bn" % i)
            y = pm.flex_conv(x.contiguous(), xyz, nbr, pm.pack_flex_weight(fc.position_theta.detach(),
                                                                             fc.position_bias.detach()), fc.cout)
            x = bn_relu(y + fc.feature_bias.reshape(1, 1, -1), bn, 2)
        pool = pm.flex_pool(x.contiguous(), nbr)                                       # backbones.py:76-79
        se = mod.se
        sq = F.relu(pool @ se.f1.tfconv0.W.reshape(se.channels, -1) + se.f1.tfconv0.b)
        sq = torch.sigmoid(sq @ se.f2.tfconv0.W.reshape(-1, se.channels) + se.f2.tfconv0.b)
        return F.relu(x + x * sq)                                                       # backbones.py:45-55

    if model._local.featdim < 128 or model.stage1.add_se != "max_pool":
        raise NotImplementedError("backbone_bn='batch' covers the shipped backbone (featdim 128, max-pool SE)")
    model._join_side(geo)  # the kNN of the full cloud runs on the geometry's side stream
    nn_8 = geo.nbr if geo.nbr.shape[2] == 8 else geo.nbr[:, :, 0:8].contiguous()
    ic = model.initconv
    init = pm.conv_pointset_xyz(geo.xyz, nn_8, ic.position_theta.detach().contiguous(), ic.position_bias.detach().contiguous())
    init = pm.flex_pool(bn_relu(init, model.initconv_bn, 2).contiguous(), nn_8)
    x1 = flex_stack(model.stage1, init, geo.xyz, nn_8)
    x2 = conv_bnrelu(x1, model.before_stage2_conv1d)
    lv = geo.level(8, model.knn_num)
    s2 = model.stage2
    feat_s = bb.gather_rows(x2.contiguous(), lv["idx"])
    y = flex_stack(s2, feat_s, lv["xyz_s"], lv["nbr_s"])
    up = ops.three_interpolate(y.contiguous(), lv["nn3_idx"], pm.idw_weights(lv["nn3_dist"]).contiguous())
    x2 = conv_bnrelu(torch.cat([up, x2], 2), s2.concat_conv1d)
    feat = conv_bnrelu(x1, model.local_stage1_shortcut) + x2
    return feat.contiguous(), lv



def global_head_autograd(model, points, localdesc, lv, bn_training=True, sync_bn=False, mask=None):
    """Differentiable restatement of compute_global (core/model.py:112-133) on the parameters of `model`.

    points [Bt,N,3], localdesc [Bt,N,128] (detached backbone output), lv = geometry level dict
    (idx, xyz_s, nbr_s, nn3_dist, nn3_idx).  Returns the un-normalised global descriptor [Bt,256]."""
    if getattr(model, "global_conv1d", False):
        # core/backbones.py:189-197: 1x1 conv + BNReLU on the full-resolution descriptors (only the last conv of the
        # loop reaches the output)
        conv = model._global_front()[-1]
        h = localdesc @ conv.W.reshape(conv.cin, conv.cout) + conv.b
        forglobal = F.relu(_bn(h, 2, conv.bn, bn_training, sync_bn, mask)).contiguous()
    else:
        gba = model.global_before_assemble
        fc, fbn = gba.flexconv_0, gba.flexconv_0_bn
        feat_s = bb.gather_rows(localdesc, lv["idx"])                                   # [Bt,M,128]
        if model.config.concat_xyz:
            # core/backbones.py:180-181: [xyz | descriptors] into the flex_conv -- here literally, through the drop-in
            # operator in the reference's channels-first layout (any channel count)
            xin = torch.cat([lv["xyz_s"], feat_s], 2).transpose(1, 2).contiguous()      # [Bt,131,M]
            x = ops.flex_convolution(xin, lv["xyz_s"].transpose(1, 2).contiguous(),
                                     lv["nbr_s"].transpose(1, 2).contiguous(), fc.position_theta,
                                     fc.position_bias).transpose(1, 2)
        else:
            x = flex_conv_factorised(feat_s, lv["xyz_s"], lv["nbr_s"], fc.position_theta, fc.position_bias)
        x = x + fc.feature_bias.reshape(1, 1, -1)                                       # layers.py:330-331
        new_feat = F.relu(_bn(x, 2, fbn, bn_training, sync_bn, mask)).contiguous()      # tf_utils.py:60-63; [Bt,M,256]
        d = torch.clamp(lv["nn3_dist"], min=1e-10)                                      # backbones.py:92-95
        w = (1.0 / d) / (1.0 / d).sum(2, keepdim=True)
        forglobal = ops.three_interpolate(new_feat, lv["nn3_idx"], w.contiguous())      # [Bt,N,256]

    att_mod = model.globalatt
    h = forglobal
    for i in range(len(att_mod.conv_dims)):
        conv = getattr(att_mod, "detec_conv%d" % i)
        h = h @ conv.W.reshape(conv.cin, conv.cout) + conv.b
        h = F.relu(_bn(h, 2, conv.bn, bn_training, sync_bn, mask))
    fcw = att_mod.detec_conv_fc
    att = torch.sigmoid(h @ fcw.W.reshape(fcw.cin, 1) + fcw.b)                      # [Bt,N,1]

    nv = model._netvlad
    Bt, N, Dm = forglobal.shape
    xr = forglobal.reshape(-1, Dm)
    xr = xr * torch.rsqrt(torch.clamp((xr * xr).sum(1, keepdim=True), min=1e-12))   # tf.nn.l2_normalize
    act = xr @ nv.cluster_weights
    pmask = mask.repeat_interleave(N) if mask is not None else None
    act = _bn(act, 1, nv.cluster_bn, bn_training, sync_bn, pmask)
    act = torch.softmax(act, dim=1) * att.reshape(-1, 1)
    act = act.reshape(Bt, N, nv.C)
    a = act.sum(1, keepdim=True) * nv.cluster_weights2                               # [Bt,D,C]
    vlad = torch.matmul(act.transpose(1, 2), xr.reshape(Bt, N, Dm)).transpose(1, 2) - a
    vlad = vlad * torch.rsqrt(torch.clamp((vlad * vlad).sum(1, keepdim=True), min=1e-12))
    vlad = vlad.reshape(Bt, nv.C * Dm)
    vlad = vlad * torch.rsqrt(torch.clamp((vlad * vlad).sum(1, keepdim=True), min=1e-12))
    v = _bn(vlad @ nv.hidden1_weights, 1, nv.bn, bn_training, sync_bn, mask)
    gates = _bn(v @ nv.gating_weights, 1, nv.gating_bn, bn_training, sync_bn, mask)
    return v * torch.sigmoid(gates)



class TorchQuadrupletTrainer(QuadrupletTrainer):
    """The quadruplet step with the trainable head (and, for backbone_bn="batch", the frozen backbone's BatchNorms) as
    tensor ops + autograd: what the HIP step is compared with.  Always eager."""

    def __init__(self, model, **kw):
        kw["graph_step"] = False
        kw["graph_backbone"] = False
        super().__init__(model, **kw)
        self.impl = "torch"

    def _backbone_batch_stats(self, block, geo, m):
        localdesc, lv = backbone_local_batch_stats(self.model, block, geo, self.sync_bn, m)
        self.model.invalidate()  # the folded copies of the backbone's moving averages are stale now
        return localdesc, lv

    def _head(self, block, localdesc, lv, m):
        return global_head_autograd(self.model, block, localdesc, lv, bn_training=True, sync_bn=self.sync_bn, mask=m)

    def _normalize(self, desc):
        return desc * torch.rsqrt(torch.clamp((desc * desc).sum(1, keepdim=True), min=1e-8))     # model.py:205

    def _loss(self, full):
        cfg = self.cfg
        m1, m2 = cfg.global_triplet_margin or 0.5, cfg.global_quadruplet_margin or 0.2
        return losses.lazy_quadruplet_loss(full, cfg.batch_size, cfg.num_pos, cfg.num_neg, m1, m2)
