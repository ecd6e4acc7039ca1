"""numpy restatement of the DH3D forward graph (TEST INFRASTRUCTURE ONLY).

Follows the reference's Python graph code line by line, in the reference's own tensor layouts (with
its transposes), calling the C oracle for the custom ops:
    core/model.py:135-210 (build_graph), core/backbones.py:45-127,156-320, core/tf_utils.py:48-109.
Weights come in as a dict keyed by TensorFlow variable name (models/*/*.index naming), so this file
shares no code with dh3d_amd.  The third-party BatchNorm constants (tensorpack: eps 1e-5, EMA decay 0.9;
slim / contrib.layers: eps 1e-3, decay 0.999) are parameters -- PARITY UNPINNED at that boundary (see
DESIGN.md): neither library is in the reference tree.
Dense algebra runs in float32 numpy (matmul order differs from TF; tolerance 1e-4 covers it).

Training mode (`train` = a TrainState; core/model.py:135-255 under global_config, BASELINE config 4):
every BatchNorm normalises with the statistics of the WHOLE batch it is given (biased variance) and records
the update of its moving buffers in train.updates.  tensorpack's BatchNorm on the rank-4 conv / flex_conv
outputs (core/tf_utils.py:61,80,99-109) and contrib.layers.batch_norm on rank-2 input ('bn', 'gating_bn':
fused=None -> fused; core/backbones.py:271-274,304-309) go through tf.nn.fused_batch_norm, whose moving-variance
update takes the Bessel-corrected batch variance (n / (n - 1)); 'cluster_bn' is fused=False upstream
(core/backbones.py:218-223) and updates with the biased one.  The FROZEN local backbone of global_config
(core/configs.py:112-113) is frozen by freeze_variables(stop_gradient=False, skip_collection=True)
(core/tf_utils.py:145-153): its variables leave the TRAINABLE collection, but its BatchNorm layers still see
training=True -- batch statistics and moving-average updates; `train.backbone_batch_stats=False` selects the
moving averages there instead (what dh3d_amd's default trainer does, a documented deviation).
"""
import numpy as np

from . import cpu as O


class TrainState(object):
    """Training-mode switch of the graph: batch statistics in every BatchNorm it reaches, moving-buffer updates
    collected in `updates` (TF variable name -> new value).  mask [Bt] bool (optional): clouds that count (padding
    clouds of a sharded block are left out of the statistics)."""

    def __init__(self, tp_decay=0.9, slim_decay=0.999, backbone_batch_stats=True, mask=None):
        self.tp_decay, self.slim_decay = tp_decay, slim_decay
        self.backbone_batch_stats = backbone_batch_stats
        self.mask = mask
        self.updates = {}


def _bn(x, w, scope, axis, eps, names=("gamma", "beta", "mean/EMA", "variance/EMA"), train=None, decay=0.9,
        bessel=True, rows_per_cloud=None):
    """Inference: moving averages.  Training (train = TrainState): statistics over every axis but `axis` in float64
    (biased variance), moving buffers <- decay * old + (1 - decay) * batch value (variance Bessel-corrected when the
    upstream layer is a fused batch norm)."""
    shape = [1] * x.ndim
    shape[axis] = -1
    g, b, m, v = [w["%s/%s" % (scope, n)].reshape(shape) for n in names]
    if train is not None:
        axes = tuple(a for a in range(x.ndim) if a != axis)
        xs = x.astype(np.float64)
        if train.mask is not None:  # leading axis = clouds (or clouds * rows_per_cloud flattened rows)
            mk = np.asarray(train.mask, bool)
            if rows_per_cloud is not None:
                mk = np.repeat(mk, rows_per_cloud)
            xs = xs[mk]
        n = xs.size // xs.shape[axis]
        mu = xs.mean(axis=axes)
        var = xs.var(axis=axes)
        train.updates["%s/%s" % (scope, names[2])] = (decay * m.reshape(-1) + (1 - decay) * mu).astype(np.float32)
        uvar = var * (n / max(n - 1, 1)) if bessel else var
        train.updates["%s/%s" % (scope, names[3])] = (decay * v.reshape(-1) + (1 - decay) * uvar).astype(np.float32)
        m, v = mu.astype(np.float32).reshape(shape), var.astype(np.float32).reshape(shape)
    return (x - m) / np.sqrt(v + np.float32(eps)) * g + b


def _slim_bn(x, w, scope, eps, train=None, bessel=True, rows_per_cloud=None):
    return _bn(x, w, scope, x.ndim - 1, eps, names=("gamma", "beta", "moving_mean", "moving_variance"), train=train,
               decay=train.slim_decay if train is not None else 0.999, bessel=bessel, rows_per_cloud=rows_per_cloud)


def _relu(x):
    return np.maximum(x, 0)


def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(np.float32)


def _conv1x1(x, w, scope, bn_eps=None, act=None, train=None):
    """tensorpack Conv2D(kernel 1) on channels-last x [..., Cin]; scope holds W [1,1,Cin,Cout], b, bn/*."""
    W = w[scope + "/W"].reshape(w[scope + "/W"].shape[2], -1)
    y = x @ W + w[scope + "/b"]
    if bn_eps is not None:
        y = _bn(y, w, scope + "/bn", y.ndim - 1, bn_eps, train=train, decay=train.tp_decay if train is not None else 0.9)
    if act is not None:
        y = act(y)
    return y.astype(np.float32)


def _l2_normalize(x, axis, eps):
    ss = np.sum(x * x, axis=axis, keepdims=True)
    return (x / np.sqrt(np.maximum(ss, np.float32(eps)))).astype(np.float32)


def knn_bruteforce_layer(points_T, k):
    """core/layers.py:85-98: op output [B,N,K] -> [B,K,N]."""
    nn, dist = O.knn_bruteforce(points_T, k)
    return np.ascontiguousarray(nn.transpose(0, 2, 1)), np.ascontiguousarray(dist.transpose(0, 2, 1))


def flexconv_bn(feats_T, points_T, nn, w, scope, eps, train=None):
    """flexconv_withBatchnorm, core/tf_utils.py:48-64 (+ feature_bias, core/layers.py:330-331)."""
    x = O.flex_convolution(feats_T, points_T, nn, w[scope + "/position_theta"], w[scope + "/position_bias"],
                           center_self=True)
    x = x + w[scope + "/feature_bias"].reshape(1, -1, 1)
    return _relu(_bn(x, w, scope + "_bn", 1, eps, train=train,
                     decay=train.tp_decay if train is not None else 0.9)).astype(np.float32)


def se_res_bottleneck(l, pool_l, w, scope):
    """core/backbones.py:45-55."""
    pool_T = pool_l.transpose(0, 2, 1)
    sq = _conv1x1(pool_T, w, scope + "/f1/tfconv0", act=_relu)
    sq = _conv1x1(sq, w, scope + "/f2/tfconv0", act=_sigmoid)
    return _relu(l + l * sq.transpose(0, 2, 1)).astype(np.float32)


def subsample(points, feat, targetnum):
    """core/tf_utils.py:86-96."""
    kp = O.farthest_point_sample(targetnum, points)[:, :, None]
    feat_s = O.group_point(feat, kp)[:, :, 0, :]
    xyz_s = O.group_point(points, kp)[:, :, 0, :]
    return xyz_s, feat_s, kp


def flex_conv_dilate(xyz, feat, dilate, knn, outdims, scope, w, eps, knn_indices=None, concat=True,
                     add_se="max_pool", upsample=True, trace=None, train=None):
    """core/backbones.py:58-101.  trace (dict or None): receives the block's integer intermediates
    ('<scope>/fps_idx' [B,m], '<scope>/knn' [B,K,m], '<scope>/nn3_idx' [B,N,3], '<scope>/nn3_dist')."""
    N = xyz.shape[1]
    if dilate > 1:
        points_s, feat_s, kp = subsample(xyz, feat, N // dilate)
        if trace is not None:
            trace[scope + "/fps_idx"] = kp[:, :, 0]
    else:
        points_s, feat_s = xyz, feat
    feats_T = np.ascontiguousarray(feat_s.transpose(0, 2, 1))
    points_T = np.ascontiguousarray(points_s.transpose(0, 2, 1))
    if knn_indices is None:
        knn_indices, _ = knn_bruteforce_layer(points_T, knn)
        if trace is not None:
            trace[scope + "/knn"] = knn_indices
    x = feats_T
    for i, d in enumerate(outdims):
        x = flexconv_bn(x, points_T, knn_indices, w, "%s/flexconv_%d" % (scope, i), eps, train=train)
    if add_se == "max_pool":
        x_pool, _ = O.flex_pooling(x, knn_indices)
        x = se_res_bottleneck(x, x_pool, w, scope + "/se")
    elif add_se == "avg_pool":  # backbones.py:80-83: flex_avg = FlexConv(theta=0, bias=eye) (layers.py:379-386), * 1/knn
        d = outdims[-1]
        x_pool = O.flex_convolution(x, points_T, knn_indices, np.zeros((3, d, d), np.float32), np.eye(d, dtype=np.float32),
                                    center_self=True)
        x_pool = (x_pool * np.float32(1.0 / knn)).astype(np.float32)
        x = se_res_bottleneck(x, x_pool, w, scope + "/se")
    new_feat = np.ascontiguousarray(x.transpose(0, 2, 1))
    if upsample and dilate > 1:
        dist, idx = O.three_nn(xyz, points_s)
        if trace is not None:
            trace[scope + "/nn3_idx"], trace[scope + "/nn3_dist"] = idx, dist.copy()
        dist = np.maximum(dist, np.float32(1e-10))
        norm = np.sum(np.float32(1.0) / dist, axis=2, keepdims=True)
        weight = (np.float32(1.0) / dist) / norm
        new_feat = O.three_interpolate(new_feat, idx, weight.astype(np.float32))
    if concat:
        new_feat = np.concatenate([new_feat, feat], axis=2)
        new_feat = _conv1x1(new_feat, w, scope + "/concat_conv1d/tfconv0", bn_eps=eps, act=_relu, train=train)
    return xyz, new_feat


def backbone_local_dilate(points, knn_ind, w, eps, trace=None, featdim=128, add_se="max_pool", train=None):
    """core/backbones.py:104-127 (add_se is 'max_pool' upstream; a parameter here to exercise :80-83)."""
    nn_8 = np.ascontiguousarray(knn_ind[:, 0:8, :])
    pts_T = np.ascontiguousarray(points.transpose(0, 2, 1))
    init = O.convolution_pointset(pts_T, nn_8, w["initconv/position_theta"], w["initconv/position_bias"])
    init = _relu(_bn(init, w, "initconv_bn", 1, eps, train=train,
                     decay=train.tp_decay if train is not None else 0.9)).astype(np.float32)
    init, _ = O.flex_pooling(init, nn_8)
    init = np.ascontiguousarray(init.transpose(0, 2, 1))
    _, x1 = flex_conv_dilate(points, init, 1, 8, [64, 64], "stage1", w, eps, knn_indices=nn_8, concat=False,
                             add_se=add_se, train=train)
    x2 = _conv1x1(x1, w, "before_stage2_conv1d/tfconv0", bn_eps=eps, act=_relu, train=train)
    _, x2 = flex_conv_dilate(points, x2, 8, 8, [128, 128], "stage2", w, eps, knn_indices=None, concat=True,
                             trace=trace, add_se=add_se, train=train)
    feat = _conv1x1(x1, w, "local_stage1_shortcut/tfconv0", bn_eps=eps, act=_relu, train=train) + x2
    if featdim < 128:  # :125-126
        feat = _conv1x1(feat, w, "final_fc/tfconv0", bn_eps=eps, act=_relu, train=train)
    return points, feat.astype(np.float32)


def detection_block(features, w, eps, scope="detection_block_reliable", train=None):
    """core/backbones.py:132-151."""
    x = features
    for i in range(3):
        x = _conv1x1(x, w, "%s/detec_conv%d" % (scope, i), bn_eps=eps, act=_relu, train=train)
    return _sigmoid(_conv1x1(x, w, scope + "/detec_conv_fc"))


def globalatt_block(features, w, eps, scope="globalatt", train=None):
    """core/backbones.py:156-173 (featdim <= 256 -> conv_dims [1024])."""
    x = _conv1x1(features, w, scope + "/detec_conv0", bn_eps=eps, act=_relu, train=train)
    return _sigmoid(_conv1x1(x, w, scope + "/detec_conv_fc"))


def global_netvlad_block(features, att, w, slim_eps, cluster_size=64, add_batch_norm=True, gating=True, train=None):
    """core/backbones.py:202-279 + context_gating :282-320."""
    B, N, D = features.shape
    x = _l2_normalize(features.reshape(-1, D), 1, 1e-12)
    act = x @ w["cluster_weights"]
    # cluster_bn: fused=False upstream (:218-223) -> biased variance in the moving-variance update
    act = (_slim_bn(act, w, "cluster_bn", slim_eps, train=train, bessel=False, rows_per_cloud=N) if add_batch_norm
           else act + w["cluster_biases"])
    act = act - act.max(axis=1, keepdims=True)
    act = np.exp(act)
    act = (act / act.sum(axis=1, keepdims=True)).astype(np.float32)
    act = act * att.reshape(-1, 1)
    act = act.reshape(B, N, cluster_size)
    a_sum = act.sum(axis=1, keepdims=True)             # [B,1,C]
    a = a_sum * w["cluster_weights2"]                   # [B,D,C]
    vlad = np.matmul(act.transpose(0, 2, 1), x.reshape(B, N, D))  # [B,C,D]
    vlad = vlad.transpose(0, 2, 1) - a                  # [B,D,C]
    vlad = _l2_normalize(vlad, 1, 1e-12)
    vlad = vlad.reshape(B, cluster_size * D)
    vlad = _l2_normalize(vlad, 1, 1e-12)
    vlad = vlad @ w["hidden1_weights"]
    vlad = _slim_bn(vlad, w, "bn", slim_eps, train=train).astype(np.float32)
    if not gating:
        return vlad.astype(np.float32)
    gates = vlad @ w["gating_weights"]
    gates = _slim_bn(gates, w, "gating_bn", slim_eps, train=train) if add_batch_norm else gates + w["gating_biases"]
    return (vlad * _sigmoid(gates)).astype(np.float32)


def forward(points, w, detection=False, extract_global=False, knn_num=8, tp_eps=1e-5, slim_eps=1e-3,
            knn_inds=None, trace=None, featdim=128, add_batch_norm=True, add_se="max_pool",
            global_backbone="global_before_assemble", gl_dims=(256,), concat_xyz=False, train=None):
    """core/model.py:135-210.  points [Bt,N,3] float32; returns dict of named outputs.
    trace (dict or None) collects the integer intermediates of the sampled levels (see flex_conv_dilate).
    train (TrainState or None): training-mode BatchNorm (module docstring)."""
    points = np.ascontiguousarray(points, np.float32)
    outs = {"pointclouds": points}
    if knn_inds is not None:
        knn_indices = np.ascontiguousarray(knn_inds.transpose(0, 2, 1))
    else:
        knn_indices, _ = knn_bruteforce_layer(np.ascontiguousarray(points.transpose(0, 2, 1)), knn_num)
    outs["knn_indices"] = knn_indices
    bb_train = train if (train is not None and train.backbone_batch_stats) else None
    newpoints, localdesc = backbone_local_dilate(points, knn_indices, w, tp_eps, trace=trace, featdim=featdim,
                                                  add_se=add_se, train=bb_train)
    l2n = _l2_normalize(localdesc, 2, 1e-8)
    outs["feat"] = localdesc
    outs["feat_l2normed"] = l2n
    outs["xyz_feat"] = np.concatenate([newpoints, l2n], -1)
    if detection:
        att = detection_block(localdesc, w, tp_eps, train=bb_train)
        outs["attention"] = att
        outs["xyz_feat_att"] = np.concatenate([newpoints, l2n, att], -1)
    if extract_global:
        if concat_xyz:  # core/backbones.py:180-181
            localdesc = np.concatenate([points, localdesc], -1)
        if global_backbone == "global_before_assemble_conv1d":
            # core/backbones.py:189-197: every conv of the loop reads localdesc, the last one is returned
            for i, d in enumerate(gl_dims):
                forglobal = _conv1x1(localdesc, w, "global_before_assemble_conv1%d" % i, bn_eps=tp_eps, act=_relu,
                                     train=train)
        else:
            _, forglobal = flex_conv_dilate(points, localdesc, 8, knn_num, [256], "global_before_assemble", w, tp_eps,
                                            knn_indices=None, concat=False, upsample=True, add_se="",
                                            trace=trace, train=train)
        gatt = globalatt_block(forglobal, w, tp_eps, train=train)
        g = global_netvlad_block(forglobal, gatt, w, slim_eps, add_batch_norm=add_batch_norm, train=train)
        outs["forglobal"] = forglobal
        outs["global_att"] = gatt
        outs["globaldesc"] = _l2_normalize(g, -1, 1e-8)
    return outs


def training_step_forward(points, w, batch_size=1, num_pos=2, num_neg=18, other_neg=True, margin1=0.5, margin2=0.2,
                          backbone_batch_stats=True, mask=None, **fw):
    """Forward half of the Siamese step under global_config (core/model.py:135-236, core/configs.py:104-144): the
    role-ordered batch [B*(1+P+Ng+1), N, 3] through the graph in TRAINING mode, then the lazy quadruplet loss on the
    l2-normalised descriptors (core/losses.py:173-200, global_loss_weight 1).  Returns (loss, outs, updates): updates
    maps every moving-average variable the step touches to its value after the step.  (The weight-decay term
    regularize_cost('.*/W') of model.py:238-243 is added by the caller: it does not depend on the data.)"""
    from . import losses_np
    st = TrainState(backbone_batch_stats=backbone_batch_stats, mask=mask)
    outs = forward(points, w, extract_global=True, train=st, **fw)
    desc = outs["globaldesc"] if mask is None else outs["globaldesc"][np.asarray(mask, bool)]
    if other_neg:
        loss = losses_np.lazy_quadruplet_loss(desc, batch_size, num_pos, num_neg, margin1, margin2)
    else:
        loss = losses_np.lazy_triplet_loss(desc, batch_size, num_pos, num_neg, margin1)
    return float(loss), outs, st.updates


def local_training_step_forward(points, R, sample_idx, w, config, **fw):
    """Forward half of the stage 1-2 training step (core/model.py:135-236 with basic_config / detection_config,
    core/configs.py:35-102): [anchors | positives] [2B,N,3] through the local backbone (and the detector) in TRAINING mode
    -- every BatchNorm on batch statistics --, l2-normalised descriptors (model.py:177), the rows at the loader's keypoint
    indices `sample_idx` [2B,M] (model.py:159-163,185-196; `backbones.subsample(..., kp_idx=)` does not exist upstream:
    with the indices given it is the gather of those rows), then the losses exactly as compute_loss assembles them
    (model.py:212-237: every loss called with **config, scaled by its *_loss_weight).  config: a mapping with the keys of
    core/configs.py.  Returns (loss, outs, updates): `updates` = every moving-average variable after the step."""
    from . import losses_np
    cfg = dict(config)
    st = TrainState(backbone_batch_stats=True)
    detection = bool(cfg.get("detection"))
    net = forward(points, w, detection=detection, extract_global=False, train=st, **fw)
    kp = np.ascontiguousarray(np.asarray(sample_idx, np.int32)[:, :, None])
    pts = np.ascontiguousarray(points, np.float32)
    outs = {"xyz": pts, "feat": net["feat"], "local_desc": net["feat_l2normed"], "R": np.asarray(R, np.float32),
            "sample_nodes_concat": kp, "xyz_sampled": O.group_point(pts, kp)[:, :, 0, :],
            "feat_sampled": O.group_point(np.ascontiguousarray(net["feat_l2normed"]), kp)[:, :, 0, :]}
    loss = np.float32(0)
    if cfg.get("add_local_loss"):
        w_loc = cfg.get("local_loss_weight")
        loss = loss + np.float32(1.0 if w_loc is None else w_loc) * losses_np.desc_local_loss(outs, **cfg)
    if detection:
        outs["attention"] = net["attention"]
        outs["att_sampled"] = O.group_point(np.ascontiguousarray(net["attention"]), kp)[:, :, 0, :]
        if cfg.get("add_det_loss"):
            w_det = cfg.get("det_loss_weight")
            loss = loss + np.float32(1.0 if w_det is None else w_det) * losses_np.local_detection_loss_nn(outs, **cfg)
    return float(loss), outs, st.updates
