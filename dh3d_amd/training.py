"""Siamese training step of the global-descriptor stage (BASELINE config 4; core/model.py:135-255 with
core/configs.py:104-144 global_config), sharded over GPUs by cloud, and the stage-1 step of the local backbone
(LocalTrainer) -- both on the hand-written HIP kernels of dh3d_amd.train_ops in BOTH directions, each replayed as one
hipGraph per step.

global_config freezes the local backbone (configs.py:112-113), so a quadruplet step is:
  1. frozen backbone + geometry on the fused HIP inference path (dh3d_amd.model, no autograd) -- or, with
     backbone_bn="batch", the same producers with their BatchNorm epilogues replaced by the colstats / finalize / apply
     kernels (backbone_local_batch_stats_hip: the reference's own semantics, batch statistics + EMA updates),
  2. the trainable global head -- global_before_assemble flex_conv (backbones.py:178-186), attention MLP (:156-173),
     NetVLAD + context gating (:202-320) -- as autograd nodes whose forward AND backward are HIP kernels
     (global_head_hip: gemm_x6 / gemm_f32, flex_S / flex_scatter, the commuted walks of interp_train.hip and
     netvlad_train.hip, training-mode BatchNorm in two HBM passes); torch contributes the autograd tape, the optimizer
     and [Bt, 256]-sized element-wise glue, no GEMM,
  3. all-gather of the [clouds_per_rank, 256] descriptors over RCCL (differentiable: backward keeps the rank's own
     slice, every rank evaluates the identical full loss), lazy quadruplet loss (core/losses.py:173-200) on the device,
     backward, ONE SUM all-reduce of the flat gradient arena, fused Adam with the staircase exponential learning rate
     (core/model.py:248-255) and L2 weight decay on '.*/W' (model.py:239-243).

BatchNorm statistics under sharding: `sync_bn=True` (default) all-reduces (sum, sum of squares, count) so the
statistics equal the reference's single-GPU whole-batch statistics; `sync_bn=False` uses per-rank statistics.

There is no tensor-op implementation of the step in this package: the plain-torch restatement the HIP step is compared
with lives in tests/torch_reference.py.
"""
import torch
import torch.distributed as dist

from . import backbones as bb
from . import dist as D
from . import losses, ops, pm


class _AllGatherKeepOwn(torch.autograd.Function):
    """all_gather whose backward returns this rank's slice of the incoming gradient."""

    @staticmethod
    def forward(ctx, local):
        if not D.collectives_active():
            ctx.rank, ctx.n = 0, local.shape[0]
            return local.clone()
        ctx.rank, ctx.n = dist.get_rank(), local.shape[0]
        return D.all_gather_rows(local)

    @staticmethod
    def backward(ctx, grad):
        return grad[ctx.rank * ctx.n:(ctx.rank + 1) * ctx.n].contiguous()


class _AllReduceSum(torch.autograd.Function):
    """SUM all-reduce with the matching backward (every rank's output depends on every rank's input: the gradient
    of the sum is the sum of the gradients)."""

    @staticmethod
    def forward(ctx, t):
        return D.all_reduce_sum_(t.clone())

    @staticmethod
    def backward(ctx, grad):
        return D.all_reduce_sum_(grad.clone())


@torch.no_grad()
def backbone_local_batch_stats_hip(model, points, geo, sync_bn=False, mask=None):
    """The frozen backbone normalising with BATCH statistics and updating its moving
    averages, as the reference's global_config graph does (core/tf_utils.py:60-63,145-153; core/backbones.py:104-127)
    -- built from the HIP kernels only, so that the whole training step stays one replayable hipGraph:

      every BatchNorm site = the producing kernel with its epilogue reduced to the conv bias (raw pre-BN rows)
                             -> dh3d_bn_colstats (f64 column sums) [-> SUM all-reduce under sync-BN]
                             -> dh3d_bn_finalize (mean / rstd / scale / shift on the device + the EMA update of the
                                moving averages, decay 0.9, Bessel-corrected variance)
                             -> dh3d_scale_shift_act (BN + ReLU in one fma pass; the last one adds the shortcut)

    The producers are the inference path's kernels with the SAME packed weights (the backbone is frozen: packed once):
    conv_pointset / flex_conv_x6 / flex_conv (exact-f32 MFMA) / linear (x6) / the fused flex_pool + SE + residual
    kernel / group_point / three_interpolate.  No tensor-op matmul, no host round trip.  The moving averages change every
    step; the FOLDED copies the inference path keeps are marked stale (model._bn_stale) and rebuilt by the next
    inference forward, not here.  Returns (localdesc [b, N, 128], geometry level)."""
    from . import train_ops as T
    if model._local.featdim < 128 or model.stage1.add_se != "max_pool":
        raise NotImplementedError("backbone_bn='batch' covers the shipped backbone (featdim 128, max-pool SE)")

    def bn_relu(x3, bnmod, residual=None):
        b, n, c = x3.shape
        x2 = x3.reshape(b * n, c)
        if residual is None:
            return T.batch_norm_train(x2, bnmod, True, sync=sync_bn, mask=mask, rows_per_cloud=n).reshape(b, n, c)
        st = T._forward_stats(x2, bnmod.gamma.detach().contiguous(), bnmod.beta.detach().contiguous(), bnmod.mean_EMA,
                              bnmod.variance_EMA, bnmod.eps, 0.9, mask, n, sync_bn, bool(getattr(bnmod, "ema_unbiased", True)))
        return pm.scale_shift_act(x2, st.stats[2], st.stats[3], True, residual=residual.reshape(b * n, c)).reshape(b, n, c)

    def conv_raw(x3, fc1d, x2=None):
        conv = fc1d.tfconv0
        p = conv._prep or conv.prepare()
        per_cloud = x3.shape[-2]
        if "wp3" in p and per_cloud >= 4096 and x3.shape[-1] % 32 == 0 and (x2 is None or x2.shape[-1] % 32 == 0):
            return pm.linear_x6(x3, p["wp3"], conv.cout, x2=x2, pre_bias=p["b"])
        return pm.linear(x3, p["wp"], conv.cout, x2=x2, pre_bias=p["b"])

    def flex_stack(mod, x, xyz, nbr):
        prep = mod._prep or mod.prepare()
        for i, p in enumerate(prep):
            bn = getattr(mod, "flexconv_%d_bn" % i)
            if p["wp3"] is not None and nbr.shape[2] == 8:
                y = pm.flex_conv_x6(x, xyz, nbr, p["wp3"], p["dout"], pre_bias=p["fb"],
                                    reserve_cus_per_xcd=getattr(geo, "busy_cus_per_xcd", 0))
            else:
                y = pm.flex_conv(x, xyz, nbr, p["wp"], p["dout"], pre_bias=p["fb"])
            x = bn_relu(y, bn)
        return mod.se.forward_on_max_pool(x, nbr)   # flex_pool + SE + residual + ReLU: no BatchNorm inside (:45-55,76-79)

    model._join_side(geo)  # the kNN of the full cloud runs on the geometry's side stream
    nn_8 = geo.nbr if geo.nbr.shape[2] == 8 else geo.nbr[:, :, 0:8].contiguous()
    lp = model._local._prep or model._local.prepare()
    init = pm.conv_pointset_xyz(geo.xyz, nn_8, lp["theta"], lp["bias"])
    init = pm.flex_pool(bn_relu(init, model.initconv_bn), nn_8)
    x1 = flex_stack(model.stage1, init, geo.xyz, nn_8)
    x2 = bn_relu(conv_raw(x1, model.before_stage2_conv1d), model.before_stage2_conv1d.tfconv0.bn)
    lv = geo.level(8, model.knn_num)
    s2 = model.stage2
    y = flex_stack(s2, bb.gather_rows(x2, lv["idx"]), lv["xyz_s"], lv["nbr_s"])
    up = pm.three_interpolate_idw(y, lv["nn3_idx"], lv["nn3_dist"])
    shortcut = bn_relu(conv_raw(x1, model.local_stage1_shortcut), model.local_stage1_shortcut.tfconv0.bn)
    feat = bn_relu(conv_raw(up, s2.concat_conv1d, x2=x2), s2.concat_conv1d.tfconv0.bn, residual=shortcut)
    model.mark_weights_changed(bn_stale=True)
    return feat, lv


class _FlexConvFactorised(torch.autograd.Function):
    """flex_conv for the training step, point-major, in its factorised form
        out = [S0 | Sx | Sy | Sz] x [bias; theta_x; theta_y; theta_z],  S0 = sum_k f[n_k],  Sd = sum_k dp_d(k) f[n_k]
    (the form the fused inference kernel runs; the drop-in op `ops.flex_convolution` keeps the reference's
    9*K*Din*Dout-flop formulation and its atomics backward: 21 of the 32 ms of a 22-cloud step).  Forward = the fused
    HIP kernel; backward = the same factorisation differentiated, also in HIP (pm.flex_conv_bwd): dW = S^T dOut and
    dS = dOut W^T on the MFMA pipe, dS scattered back over the neighbour lists with f32 atomics.  Gradients w.r.t. features, theta and bias (positions are data).  Centre = the point
    itself (the GPU forward's rule, flex_conv_kernel_gpu.cu.cc:77-79; identical to the backward's rank-0-neighbour
    rule under exact kNN)."""

    @staticmethod
    def forward(ctx, feat, xyz, nbr, theta, bias):
        from . import pm
        out = pm.flex_conv(feat, xyz, nbr, pm.pack_flex_weight(theta.detach(), bias.detach()), theta.shape[2])
        ctx.save_for_backward(feat, xyz, nbr, theta, bias)
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import pm
        feat, xyz, nbr, theta, bias = ctx.saved_tensors
        # dWcat = S^T dOut and dS = dOut Wcat^T on the f32 MFMA pipe, dS scattered over the neighbour lists with
        # hardware f32 atomics (csrc/flex_bwd.hip)
        dfeat, dtheta, dbias = pm.flex_conv_bwd(feat, xyz, nbr, theta.detach(), bias.detach(), dout.contiguous(),
                                                center_rank0=False, need_grad_features=ctx.needs_input_grad[0])
        return dfeat, None, None, dtheta, dbias


def flex_conv_factorised(feat, xyz, nbr, theta, bias):
    """feat [B,M,Din], xyz [B,M,3], nbr [B,M,K] int32, theta [3,Din,Dout], bias [Din,Dout] -> [B,M,Dout]."""
    return _FlexConvFactorised.apply(feat.contiguous(), xyz.contiguous(), nbr.contiguous(), theta, bias)


def global_head_hip(model, points, localdesc, lv, sync_bn=False, mask=None, commute_attention=True,
                    commute_netvlad=True):
    """compute_global (core/model.py:112-133) in training mode with every row-level operator -- flex_conv, the three
    training-mode BatchNorms on rows, three_interpolate, the attention MLP, NetVLAD's assignment / aggregation -- as a
    hand-written HIP kernel in BOTH directions (dh3d_amd.train_ops, csrc/train.hip / gemm.hip / flex_bwd.hip); what is
    left to torch are [Bt, 256]- and [Bt, 16384]-sized per-cloud element-wise ops (microseconds).  The tests compare it
    with tests/torch_reference.global_head_autograd(bn_training=True).  Head shapes the kernels do not cover (more than one
    attention conv or global flex_conv, NetVLAD without BatchNorm / gating: no shipped config) raise."""
    from . import train_ops as T
    gba, att_mod, nv = model.global_before_assemble, model.globalatt, model._netvlad
    conv1d = getattr(model, "global_conv1d", False)
    if (len(att_mod.conv_dims) != 1 or (not conv1d and len(gba.outdims) != 1) or not (nv.add_batch_norm and nv.gating)):
        raise NotImplementedError("global_head_hip covers the shipped global head (one attention conv, one "
                                  "global_before_assemble flex_conv or the conv1d front, NetVLAD with BatchNorm + gating)")
    Bt, N = points.shape[0], points.shape[1]
    if conv1d:
        # global_before_assemble_conv1d (core/backbones.py:189-197): no sampled level, no interpolation -- a 1x1 conv +
        # BNReLU on the full-resolution rows, the attention head on the materialised rows, NetVLAD as below
        conv = model._global_front()[-1]
        h = T.linear(localdesc.reshape(Bt * N, conv.cin), conv.W.reshape(conv.cin, conv.cout), conv.b)
        forglobal = T.batch_norm_train(h, conv.bn, True, sync_bn, mask, N).reshape(Bt, N, conv.cout)
        fcw = att_mod.detec_conv_fc
        att = T.attention_head(forglobal.reshape(Bt * N, conv.cout), att_mod.detec_conv0, fcw.W, fcw.b, sync_bn, mask, N)
        return _netvlad_tail_hip(T, nv, forglobal, att, sync_bn, mask)
    fc, fbn = gba.flexconv_0, gba.flexconv_0_bn
    M = lv["xyz_s"].shape[1]
    feat_s = bb.gather_rows(localdesc, lv["idx"])                                   # [Bt,M,128]
    if model.config.concat_xyz:
        # [xyz | descriptors] input (core/backbones.py:180-181): flex_conv is linear in its input channels, so it is the
        # factorised kernel on the 128 descriptor channels plus the same kernel on the coordinates padded to that
        # width (zero weights in the padding -- the inference path's split, backbones.FlexConvDilate.prepare); the
        # slices / concatenations are autograd's, so position_theta / position_bias receive their whole gradient
        theta, bias = fc.position_theta, fc.position_bias
        cf = theta.shape[1] - 3
        xp = torch.zeros((Bt, M, cf), dtype=torch.float32, device=points.device)
        xp[:, :, :3] = lv["xyz_s"]
        tx = torch.cat([theta[:, :3], theta.new_zeros((3, cf - 3, theta.shape[2]))], 1)
        bx = torch.cat([bias[:3], bias.new_zeros((cf - 3, bias.shape[1]))], 0)
        x = (flex_conv_factorised(feat_s, lv["xyz_s"], lv["nbr_s"], theta[:, 3:].contiguous(), bias[3:].contiguous())
             + flex_conv_factorised(xp, lv["xyz_s"], lv["nbr_s"], tx, bx))
    else:
        x = flex_conv_factorised(feat_s, lv["xyz_s"], lv["nbr_s"], fc.position_theta, fc.position_bias)
    x = T.add_channel_bias(x, fc.feature_bias.reshape(-1))                          # layers.py:330-331
    Dg = x.shape[2]
    new_feat = T.batch_norm_train(x.reshape(Bt * M, Dg), fbn, True, sync_bn, mask, M).reshape(Bt, M, Dg)
    from . import pm
    w = pm.idw_weights(lv["nn3_dist"])                                              # backbones.py:92-95 (no gradient: geometry)
    fcw = att_mod.detec_conv_fc
    sorted_walks = commute_attention and T.attention_commute_supported(att_mod.detec_conv0, M) and Dg == 256
    commuted_vlad = (sorted_walks and commute_netvlad and T.netvlad_commute_supported(new_feat, nv.cluster_weights)
                     and nv.add_batch_norm)
    forglobal = None
    if sorted_walks:
        # the fine clouds' Morton records: from the geometry level if it has them (compute_level stores them for
        # N >= 4096); three_interpolate's backward and the attention head walk the points in that order
        order = lv["_ordered"][0] if "_ordered" in lv else pm.spatial_sort(points)[0]
        if not commuted_vlad:
            forglobal = T.three_interpolate_sorted(new_feat, lv["nn3_idx"], w, order)   # [Bt,N,256]
    else:
        forglobal = ops.three_interpolate(new_feat, lv["nn3_idx"], w.contiguous())
    if sorted_walks:
        # conv(interp(c)) = interp(conv(c)): the head's GEMMs on the Bt*M sampled rows, its [Bt*N, 1024] pre-activation
        # never written (csrc/interp_train.hip)
        att = T.attention_head_commuted(new_feat.reshape(Bt * M, Dg), att_mod.detec_conv0, fcw.W, fcw.b, lv["nn3_idx"],
                                        lv["nn3_dist"], order, sync_bn, mask)
    else:
        att = T.attention_head(forglobal.reshape(Bt * N, Dg), att_mod.detec_conv0, fcw.W, fcw.b, sync_bn, mask, N)
    if commuted_vlad:
        # NetVLAD's assignment commuted through the up-sampling as well: the [Bt,N,256] rows are never built in the
        # training step (csrc/netvlad_train.hip)
        return _netvlad_tail_hip(T, nv, None, att, sync_bn, mask,
                                 commuted=(new_feat, lv["nn3_idx"], lv["nn3_dist"], order))
    return _netvlad_tail_hip(T, nv, forglobal, att, sync_bn, mask)


def _netvlad_tail_hip(T, nv, forglobal, att, sync_bn, mask, commuted=None):
    """NetVLAD + context gating of the training step on HIP nodes (core/backbones.py:202-279): forglobal [Bt,N,D] rows
    (or commuted = (sampled rows [Bt,M,D], three_nn idx, dist, Morton records of the fine clouds)), att [Bt*N] ->
    [Bt, 256]."""
    if commuted is not None:
        coarse, idx3, dist3, order = commuted
        Bt, Dg = coarse.shape[0], coarse.shape[2]
        V, asum = T.netvlad_assign_commuted(coarse, att, nv.cluster_weights, nv.cluster_bn, idx3, dist3, order, sync_bn,
                                            mask)
    else:
        Bt, Dg = forglobal.shape[0], forglobal.shape[2]
        V, asum = T.netvlad_assign(forglobal, att, nv.cluster_weights, nv.cluster_bn, sync_bn, mask)   # [Bt,C,D], [Bt,C]
    if T.vlad_normalize_supported(V):
        vlad = T.vlad_normalize(V, asum, nv.cluster_weights2)                       # [Bt, D*C]  (backbones.py:241-262)
    else:
        vlad = V.transpose(1, 2) - asum.unsqueeze(1) * nv.cluster_weights2          # [Bt,D,C]
        vlad = vlad * torch.rsqrt(torch.clamp((vlad * vlad).sum(1, keepdim=True), min=1e-12))
        vlad = vlad.reshape(Bt, nv.C * Dg)
        vlad = vlad * torch.rsqrt(torch.clamp((vlad * vlad).sum(1, keepdim=True), min=1e-12))
    v = T.batch_norm_train(T.linear(vlad, nv.hidden1_weights), nv.bn, False, sync_bn, mask, 1)
    gates = T.batch_norm_train(T.linear(v, nv.gating_weights), nv.gating_bn, False, sync_bn, mask, 1)
    return T.context_gate(v, gates)


def trainable_head_parameters(model):
    """Parameters that global_config trains (backbone frozen: configs.py:112-113).  With
    global_before_assemble_conv1d only the LAST conv of the loop reaches the descriptor (core/backbones.py:189-197: every
    conv reads localdesc), so only it has a gradient."""
    front = model._global_front()[-1:] if getattr(model, "global_conv1d", False) else [model.global_before_assemble]
    mods = front + [model.globalatt, model._netvlad]
    seen, out = set(), []
    for mod in mods:
        for p in mod.parameters():
            if id(p) not in seen:
                seen.add(id(p))
                out.append(p)
    return out


class QuadrupletTrainer(object):
    """One process per GPU.  `step(points)` takes the role-ordered batch [B*(1+P+Ng+1), N, 3] (identical on
    every rank, e.g. generated from a shared seed), runs this rank's block and returns the loss."""

    def __init__(self, model, start_lr=None, decay_step=None, decay_rate=None, weight_decay=None, sync_bn=True,
                 graph_backbone=True, graph_step=None, backbone_bn="ema"):
        """Schedule / weight decay default to the model's config (core/configs.py:50-54,115-117).  sync_bn=True (the
        default) reproduces the reference's whole-batch BatchNorm statistics under sharding -- and keeps the running
        buffers identical on every rank; sync_bn=False normalises with per-rank statistics (a few clouds of one role
        each under the contiguous role-ordered partition) and lets the buffers diverge.
        backbone_bn: "ema" (default) -- the frozen backbone runs the fused inference path on its moving averages;
        "batch" -- it normalises with batch statistics and updates its moving averages, which is what the reference
        graph does while global_config trains (backbone_local_batch_stats_hip; slower, un-fused)."""
        if backbone_bn not in ("ema", "batch"):
            raise ValueError("backbone_bn must be 'ema' or 'batch'")
        self.model = model
        self.cfg = model.config
        self.impl = "hip"         # (reported by bench.py; the tensor-op subclass of the tests says "torch")
        self.backbone_bn = backbone_bn
        self.graph_backbone = graph_backbone and backbone_bn == "ema"
        self._bb_graphs = {}
        self.keep_grads, self.last_grads = False, None
        self._ev = None
        c = self.cfg
        start_lr = start_lr if start_lr is not None else (c.start_lr or 5e-4)
        decay_step = decay_step if decay_step is not None else (c.decay_step or 20000)
        decay_rate = decay_rate if decay_rate is not None else (c.decay_rate or 0.9)
        if weight_decay is None:
            weight_decay = (c.train_weight_decay or 1e-5) if c.add_weight_decay is not False else 0.0
        self.sync_bn = sync_bn
        self.params = trainable_head_parameters(model)
        self.wd_params = [p for n, p in model.named_parameters() if n.endswith(".W")
                          and any(p is q for q in self.params)]
        self.weight_decay = weight_decay
        # Whole-step hipGraph (forward, loss, backward, weight decay, gradient all-reduce, Adam): a 22-cloud step is ~600
        # launches, about as much host time as GPU time.  The first steps on a batch shape run eagerly (allocator /
        # autograd warm-up), then the step is captured once per shape and replayed.  The learning rate is a device
        # scalar the staircase schedule writes into.  Sharded (RCCL): the collectives -- sync-BN statistics, the
        # descriptor all-gather, ONE all-reduce of the flat gradient arena -- are captured with the kernels between
        # them (torch's NCCL process group enqueues on the capturing stream); gloo's host-staged collectives cannot be
        # captured, so the CPU-test path stays eager.
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        capturable = world == 1 and not D.collectives_active() or (D.collectives_active() and dist.get_backend() == "nccl")
        self.graph_step = capturable if graph_step is None else (bool(graph_step) and capturable)
        self._zarena = pm.ZeroArena()   # the step's accumulators: one fill per step (pm.ZeroArena)
        self._garena = None     # flat gradient arena of the sharded step (all-reduced in place, .grad are views of it)
        self._sched = (float(start_lr), int(decay_step), float(decay_rate))
        self._steps_done = 0
        self._step_graphs = {}
        self._shape_demand = {}  # batch shape -> accumulator bytes one step of it takes from the ZeroArena
        self._eager_seen = {}   # batch shape -> eager steps run on it (every shape warms up before its capture)
        self.keep_desc, self.last_desc = False, None
        if self.graph_step:
            dev = self.params[0].device
            self._lr = torch.tensor(float(start_lr), dtype=torch.float32, device=dev)
            self._lr_value = float(start_lr)
            # (fused: one multi-tensor kernel per step instead of ~8 foreach passes over the 20 parameter tensors)
            self.opt = torch.optim.Adam(self.params, lr=self._lr, capturable=True, fused=True)
            self.sched = None
        else:
            self.opt = torch.optim.Adam(self.params, lr=start_lr, fused=bool(self.params[0].is_cuda))
            self.sched = torch.optim.lr_scheduler.LambdaLR(self.opt, lambda s: decay_rate ** (s // decay_step))

    def forward_loss(self, points):
        cfg = self.cfg
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        Bt = points.shape[0]
        block, mask = D.shard_batch(points, rank, world)
        start, stop = D.local_slice(Bt, rank, world)
        m = mask if (stop - start) < block.shape[0] else None  # (known on the host: no device round trip)
        self.model.eval()
        if self.backbone_bn == "batch":  # the reference's semantics: batch statistics + moving-average updates
            geo = self.model._geometry(block, None)
            localdesc, lv = self._backbone_batch_stats(block, geo, m)
        else:  # frozen backbone on the fused inference path (moving averages)
            localdesc, lv = self._backbone(block)
        self._mark(1)
        desc = self._normalize(self._head(block, localdesc.detach(), lv, m))
        if not D.collectives_active() and desc.shape[0] == Bt:
            full = desc   # one rank, no padding: nothing to gather or slice (a clone, a fill and two copies per step)
        else:
            full = _AllGatherKeepOwn.apply(desc)[:Bt]
        if self.keep_desc:  # tests: the gathered, l2-normalised descriptors of this step
            self.last_desc = full.detach().clone()
        return self._loss(full)

    # The four blocks of a step (tests/torch_reference.TorchQuadrupletTrainer overrides them with tensor ops):
    def _backbone_batch_stats(self, block, geo, m):
        """HIP kernels only, capturable (the moving averages are updated on the device)."""
        localdesc, lv = backbone_local_batch_stats_hip(self.model, block, geo, self.sync_bn, m)
        return localdesc, {k: v for k, v in lv.items() if torch.is_tensor(v) or k == "_ordered"}

    def _head(self, block, localdesc, lv, m):
        return global_head_hip(self.model, block, localdesc, lv, sync_bn=self.sync_bn, mask=m)

    def _normalize(self, desc):
        from . import train_ops as T
        return T.l2_normalize_rows(desc, 1e-8)                                              # model.py:205

    def _loss(self, full):
        from . import train_ops as T
        cfg = self.cfg
        m1, m2 = cfg.global_triplet_margin or 0.5, cfg.global_quadruplet_margin or 0.2
        if T.quadruplet_loss_supported(full, cfg.num_pos, cfg.num_neg):
            return T.quadruplet_loss(full.contiguous(), cfg.batch_size, cfg.num_pos, cfg.num_neg, m1, m2)
        return losses.lazy_quadruplet_loss(full, cfg.batch_size, cfg.num_pos, cfg.num_neg, m1, m2)

    def _backbone(self, block):
        """Frozen backbone + geometry of this rank's block: (localdesc [b,N,128], level dict of tensors).  Its weights
        never change during global_config training, so the two-stream forward is captured into a hipGraph once per
        block shape and replayed (eager, its ~45 launches cost 1.09 ms per step against ~0.75 ms replayed)."""
        if not self.graph_backbone or torch.cuda.is_current_stream_capturing():  # (no replay inside a capture)
            with torch.no_grad():
                geo = self.model._geometry(block, None)
                _, localdesc = self.model.compute_local(block, _geo=geo)
                lv = geo.level(8, self.model.knn_num)
                self.model._join_side(geo)  # nothing of this step may stay open on the side stream (whole-step capture)
            return localdesc, lv
        key = (tuple(block.shape), block.device, getattr(self.model, "_backbone_version", 0))
        ent = self._bb_graphs.get(key)
        if ent is None:
            self._bb_graphs.clear()  # (a reloaded / moved model: the old graph points at freed weight copies)
            static_in = block.clone()

            def body():
                with torch.no_grad():
                    geo = self.model._geometry(static_in, None)
                    _, localdesc = self.model.compute_local(static_in, _geo=geo)
                    lv = geo.level(8, self.model.knn_num)
                keep = {k: v for k, v in lv.items() if torch.is_tensor(v) or k == "_ordered"}  # no capture-time events
                return localdesc, keep

            s = torch.cuda.Stream(device=block.device)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):
                    body()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                outs = body()
            ent = (graph, static_in, outs)
            self._bb_graphs[key] = ent
        graph, static_in, outs = ent
        static_in.copy_(block)
        graph.replay()
        return outs

    # phase timing (bench.py --workload train): events on the current stream around the four phases of a step
    def time_phases(self, on=True):
        self._ev = [] if on else None

    def _mark(self, k):
        if self._ev is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._ev.append((k, e))

    def phase_times_ms(self):
        """Mean ms of [frozen backbone fwd, head fwd + loss, backward, grad all-reduce + Adam] over the timed steps."""
        if not self._ev:
            return None
        torch.cuda.synchronize()
        names = ["backbone_fwd", "head_fwd_loss", "backward", "allreduce_adam"]
        tot, cnt = [0.0] * 4, [0] * 4
        for (k0, e0), (k1, e1) in zip(self._ev[:-1], self._ev[1:]):
            if k1 == k0 + 1:
                tot[k0] += e0.elapsed_time(e1)
                cnt[k0] += 1
        return {n: (t / c if c else None) for n, t, c in zip(names, tot, cnt)}

    def _lr_now(self):
        lr0, dstep, drate = self._sched
        return lr0 * drate ** (self._steps_done // dstep)

    def _set_lr(self):
        lr = self._lr_now()
        if lr != self._lr_value:  # (staircase schedule: changes every decay_step steps)
            self._lr.fill_(lr)
            self._lr_value = lr

    def input_buffer(self, shape, device=None):
        """The replayed step's own input tensor for batches of `shape` (None until that shape has been captured): a
        loader that writes the next batch straight into it and passes it to `step` saves the copy into the graph's
        input (the graph reads this buffer; any other tensor is copied into it first)."""
        for (shp, dev, _), ent in self._step_graphs.items():
            if tuple(shp) == tuple(shape) and (device is None or dev == device):
                return ent[1]
        return None

    def _reduce_gradients(self):
        """Sharded step: SUM all-reduce of the head gradients.  One persistent flat arena (18.7 MB for global_config):
        the gradients are gathered into it by ONE multi-tensor copy, all-reduced in place, and every .grad becomes a
        view of it -- no per-step concatenation, no per-parameter clones."""
        self._ensure_arena()
        have = [(v, p.grad) for v, p in zip(self._gviews, self.params) if p.grad is not None and p.grad is not v]
        none = [v for v, p in zip(self._gviews, self.params) if p.grad is None]
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        if none:
            torch._foreach_zero_(none)
        D.all_reduce_sum_(self._garena)  # every rank holds a partial of the SAME loss
        for v, p in zip(self._gviews, self.params):
            p.grad = v

    def _ensure_arena(self):
        if self._garena is None:  # (never first built inside a capture: it must outlive any one graph's pool)
            n = sum(p.numel() for p in self.params)
            self._garena = torch.zeros(n, dtype=torch.float32, device=self.params[0].device)
            self._gviews, off = [], 0
            for p in self.params:
                self._gviews.append(self._garena[off:off + p.numel()].view(p.shape))
                off += p.numel()

    def _step_graphed(self, points):
        key = (tuple(points.shape), points.device, getattr(self.model, "_backbone_version", 0))
        ent = self._step_graphs.get(key)
        if ent is None:
            for k in [k for k in self._step_graphs if k[2] != key[2]]:  # graphs of an older backbone are stale
                del self._step_graphs[k]
            static_in = points.clone()
            # the graph's OWN accumulator arena, sized by this shape's eager steps and never reallocated: the eager
            # arena grows (and frees its old buffer) whenever a larger shape comes along, and a graph replaying into a
            # freed buffer would corrupt whatever owns that memory by then
            arena = pm.ZeroArena(fixed_bytes=self._shape_demand.get(key[:2], self._zarena.peak), device=points.device)
            self.opt.zero_grad(set_to_none=True)
            if D.collectives_active():
                self._ensure_arena()
            graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(graph), pm.zero_arena(arena):
                    arena.begin(static_in.device)   # ONE fill for every accumulator of the step
                    loss = self.forward_loss(static_in)
                    loss.backward()
                    if self.wd_params and self.weight_decay and (not D.collectives_active() or dist.get_rank() == 0):
                        gs = [p.grad for p in self.wd_params if p.grad is not None]
                        ps = [p.detach() for p in self.wd_params if p.grad is not None]
                        if gs:
                            torch._foreach_add_(gs, ps, alpha=self.weight_decay)
                    if D.collectives_active():
                        self._reduce_gradients()
                    self.opt.step()
            except RuntimeError as e:
                # something in this shape's step cannot be captured: stay eager for good (the parameters are untouched --
                # a capture records, it does not run) -- and say so.  In a sharded run the ranks must not diverge (one
                # replaying captured collectives, one issuing them eagerly), and a capture that died inside a
                # collective leaves the communicator undefined: that is an error, not a fallback.
                if D.collectives_active():
                    raise
                import warnings
                warnings.warn("dh3d_amd: the training step for batches of shape %s could not be captured into a "
                              "hipGraph (%s: %s); continuing with eager steps" % (tuple(points.shape), type(e).__name__, e),
                              RuntimeWarning)
                self.graph_step = False
                self.opt.zero_grad(set_to_none=True)
                return None
            ent = (graph, static_in, loss, arena)
            self._step_graphs[key] = ent
        graph, static_in, loss = ent[:3]
        # (each of these two is a launch of its own in front of the replay -- ~10 us of queue latency apiece: skipped
        # when the batch already sits in the step's input buffer (`input_buffer`) / the rate has not changed)
        if points.data_ptr() != static_in.data_ptr():
            static_in.copy_(points)
        self._set_lr()
        graph.replay()
        self._steps_done += 1
        self.model.invalidate(head_only=True)
        if self.backbone_bn == "batch":   # the replay moved the backbone's moving averages on the device
            self.model._bn_stale = True
        return loss.detach()

    def step(self, points, sync=True):
        """One optimisation step; returns the loss -- as a float (sync=True: one host round trip per step) or as the
        device scalar the step wrote (sync=False: nothing waits for the GPU; a replayed step overwrites it)."""
        loss = self._step(points)
        return float(loss) if sync else loss

    def _step(self, points):
        # eager for the first steps ON EVERY BATCH SHAPE (allocator / autograd / lazily built constants warm up before
        # the capture), while phases are timed or gradients / descriptors are kept for inspection
        shape = (tuple(points.shape), points.device)
        if (self.graph_step and self._eager_seen.get(shape, 0) >= 3 and self._ev is None and not self.keep_grads
                and not self.keep_desc):
            out = self._step_graphed(points)
            if out is not None:
                return out
        self._eager_seen[shape] = self._eager_seen.get(shape, 0) + 1
        if self.graph_step:
            self._set_lr()
        self.opt.zero_grad(set_to_none=True)
        self._mark(0)
        with pm.zero_arena(self._zarena):
            self._zarena.begin(points.device)
            loss = self.forward_loss(points)
            self._mark(2)
            loss.backward()
        self._shape_demand[shape] = max(self._shape_demand.get(shape, 0), self._zarena.demand)
        # L2 weight decay on '.*/W' (regularize_cost, core/model.py:239-243): d/dp [wd/2 * sum p^2] = wd * p, added to the
        # gradients directly (one multi-tensor launch) by rank 0 only -- the SUM all-reduce below then counts it once
        if self.wd_params and self.weight_decay and (not dist.is_initialized() or dist.get_rank() == 0):
            gs = [p.grad for p in self.wd_params if p.grad is not None]
            ps = [p.detach() for p in self.wd_params if p.grad is not None]
            if gs:
                torch._foreach_add_(gs, ps, alpha=self.weight_decay)
        self._mark(3)
        if D.collectives_active():
            self._reduce_gradients()
        if self.keep_grads:  # tests: the reduced gradient of this step (Adam's m/sqrt(v) is sign-like in the first
            self.last_grads = [p.grad.detach().clone() for p in self.params]  # steps: parameters are ill-conditioned)
        self.opt.step()
        if self.sched is not None:
            self.sched.step()
        self._steps_done += 1
        self._mark(4)
        self.model.invalidate(head_only=True)  # the packed / folded weight copies of the fused inference path are stale now
        return loss.detach()


# ======================================================================================================================
# Stage 1-2 training: the LOCAL backbone (+ the detector) trained with desc_local_loss / local_detection_loss_nn
# (core/model.py:135-246 with core/configs.py:35-102 basic_config / detection_config, core/losses.py:29-133).
def local_backbone_train(model, points, geo, sync_bn=False, mask=None):
    """backbone_local_dilate (core/backbones.py:104-127) in TRAINING mode with autograd: every BatchNorm on batch
    statistics (moving averages updated), every operator a HIP kernel in both directions --
      conv_pointset / ConvPointsetGrad (factorised: grad_theta = S^T grad_out), flex_pool / FlexPoolGrad (argmax scatter),
      flex_conv / FlexConvGrad in the factorised form at FULL resolution (S^T dOut, dOut W^T on the MFMA pipe, the
      feature gradient scattered over the neighbour lists), the 1x1 convs as f32-accurate GEMMs (x W, dy W^T, x^T dy),
      BatchNorm-train (two passes per direction), the SE gate, group_point / three_interpolate with their gradients.
    What torch contributes is autograd's bookkeeping, one concatenation and the residual add.
    points [Bt,N,3]; geo: model._geometry(points) (kNN, FPS level -- integer work, no gradient).  Returns the
    un-normalised descriptors [Bt,N,128]."""
    from . import train_ops as T
    if model._local.featdim < 128 or model.stage1.add_se != "max_pool" or model.stage2.add_se != "max_pool":
        raise NotImplementedError("LocalTrainer covers the shipped backbone (featdim 128, max-pool SE)")
    Bt, N, _ = points.shape

    def bn_relu(x3, bnmod):
        b, n, c = x3.shape
        return T.batch_norm_train(x3.reshape(b * n, c), bnmod, True, sync=sync_bn, mask=mask, rows_per_cloud=n).reshape(b, n, c)

    def conv_bnrelu(x3, fc1d):
        conv = fc1d.tfconv0
        b, n, c = x3.shape
        h = T.linear(x3.reshape(b * n, c), conv.W.reshape(conv.cin, conv.cout), conv.b)
        return T.batch_norm_train(h, conv.bn, True, sync=sync_bn, mask=mask, rows_per_cloud=n).reshape(b, n, conv.cout)

    def flex_stack(mod, x, xyz, nbr):
        for i in range(len(mod.outdims)):
            fc, bn = getattr(mod, "flexconv_%d" % i), getattr(mod, "flexconv_%d_bn" % i)
            y = flex_conv_factorised(x, xyz, nbr, fc.position_theta, fc.position_bias)
            x = bn_relu(T.add_channel_bias(y, fc.feature_bias.reshape(-1)), bn)
        b, n, c = x.shape
        se = mod.se
        pool = T.flex_pool(x, nbr)                                                           # backbones.py:76-79
        f1, f2 = se.f1.tfconv0, se.f2.tfconv0
        sq = T.relu(T.linear(pool.reshape(b * n, c), f1.W.reshape(f1.cin, f1.cout), f1.b))
        z = T.linear(sq, f2.W.reshape(f2.cin, f2.cout), f2.b)
        return T.se_gate(x, z.reshape(b, n, c))                                              # relu(x + x * sigmoid(z))

    model._join_side(geo)  # the kNN of the full cloud runs on the geometry's side stream
    nn_8 = geo.nbr if geo.nbr.shape[2] == 8 else geo.nbr[:, :, 0:8].contiguous()
    ic = model.initconv
    init = T.conv_pointset_xyz(geo.xyz, nn_8, ic.position_theta, ic.position_bias)
    init = T.flex_pool(bn_relu(init, model.initconv_bn), nn_8)
    x1 = flex_stack(model.stage1, init, geo.xyz, nn_8)
    x2 = conv_bnrelu(x1, model.before_stage2_conv1d)
    lv = geo.level(8, model.knn_num)
    s2 = model.stage2
    feat_s = ops.group_point(x2, lv["idx"].unsqueeze(2)).squeeze(2)                          # [Bt,M,64], differentiable
    y = flex_stack(s2, feat_s, lv["xyz_s"], lv["nbr_s"])
    up = ops.three_interpolate(y, lv["nn3_idx"], pm.idw_weights(lv["nn3_dist"]))             # backbones.py:89-95
    x2 = conv_bnrelu(torch.cat([up, x2], 2), s2.concat_conv1d)                               # :98-100
    return conv_bnrelu(x1, model.local_stage1_shortcut) + x2                                 # :123


def detection_block_train(model, feat, sync_bn=False, mask=None):
    """detection_block (core/backbones.py:132-151) in training mode on rows feat [Bt,N,128]: Conv2D + BNReLU chain as
    GEMM + BatchNorm-train nodes, the last wide layer + the 1-channel logit + sigmoid as the fused attention-head node
    (its [R, 1024] pre-activation is the only tensor of that width).  Returns att [Bt,N,1]."""
    from . import train_ops as T
    det = model.detection_block_reliable
    Bt, N, C = feat.shape
    x = feat.reshape(Bt * N, C)
    nconv = len(det.conv_dims)
    for i in range(nconv - 1):
        conv = getattr(det, "detec_conv%d" % i)
        x = T.batch_norm_train(T.linear(x, conv.W.reshape(conv.cin, conv.cout), conv.b), conv.bn, True, sync=sync_bn,
                               mask=mask, rows_per_cloud=N)
    last, fcw = getattr(det, "detec_conv%d" % (nconv - 1)), det.detec_conv_fc
    att = T.attention_head(x, last, fcw.W, fcw.b, sync_bn, mask, N)
    return att.reshape(Bt, N, 1)


def local_training_outputs(model, points, R, sample_idx, sync_bn=False, mask=None):
    """The named outputs the local losses read (core/model.py:166-199): 'xyz', 'feat', 'local_desc', 'R',
    'sample_nodes_concat', 'xyz_sampled', 'feat_sampled' (+ 'attention', 'att_sampled' with config.detection).
    points [2B,N,3] = [anchors | positives]; R [B,3,3]; sample_idx [2B,M] int32: the keypoint indices the data loader
    supplies as sample_ind_anchor / sample_ind_pos (model.py:159-163).  `backbones.subsample(points, feat, kpnum,
    kp_idx=...)` that model.py:187 calls does not exist upstream; with the indices given it can only be the gather of
    those rows (group_point), which is what runs here."""
    from . import train_ops as T
    cfg = model.config
    Bt, N, _ = points.shape
    geo = model._geometry(points, None)
    feat = local_backbone_train(model, points, geo, sync_bn, mask)
    desc = T.l2_normalize_rows(feat.reshape(Bt * N, -1), 1e-8).reshape(Bt, N, -1)            # model.py:177
    kp = sample_idx.to(torch.int32).reshape(Bt, -1, 1).contiguous()
    outs = {"xyz": points, "feat": feat, "local_desc": desc, "R": R, "sample_nodes_concat": kp,
            "xyz_sampled": ops.group_point(points, kp).squeeze(2), "feat_sampled": ops.group_point(desc, kp).squeeze(2)}
    if cfg.detection:
        # freezedetection (core/backbones.py:136, tf_utils.py:144-153) is freeze_variables(stop_gradient=False,
        # skip_collection=True): the detector's VARIABLES leave the trainable set (local_trainable_parameters), the
        # detection loss's gradient still flows through the detector into the backbone -- so no detach here
        att = detection_block_train(model, feat, sync_bn, mask)
        outs["attention"] = att
        outs["att_sampled"] = ops.group_point(att, kp).squeeze(2)                            # model.py:196
    return outs


def local_trainable_parameters(model):
    """Everything basic_config / detection_config train (configs.py:41-43): the local backbone unless `freezebackbone`,
    and, with config.detection, the detector unless `freezedetection` (frozen variables are skipped by the optimiser
    and by the weight decay, core/tf_utils.py:144-153)."""
    cfg = model.config
    mods = ([] if cfg.get("freezebackbone") else [model._local]) + \
           ([model.detection_block_reliable] if cfg.detection and not cfg.get("freezedetection") else [])
    seen, out = set(), []
    for mod in mods:
        for p in mod.parameters():
            if id(p) not in seen:
                seen.add(id(p))
                out.append(p)
    return out


class LocalTrainer(object):
    """Stage 1-2 training step on ONE GPU: forward of the whole local backbone in training mode (local_backbone_train),
    desc_local_loss (+ local_detection_loss_nn x det_loss_weight with detection_config) through losses.compute_loss as
    core/model.py:212-237 assembles them, backward, L2 weight decay on '.*/W', Adam with the staircase learning rate
    (model.py:238-255).  After three eager steps on a batch shape the whole step -- ~500 launches -- is captured into one
    hipGraph and replayed from its own input buffers."""

    def __init__(self, model, start_lr=None, decay_step=None, decay_rate=None, weight_decay=None, graph_step=True):
        self.model, self.cfg = model, model.config
        c = self.cfg
        if c.extract_global:
            raise ValueError("LocalTrainer trains basic_config / detection_config (extract_global False); the global "
                             "stage is QuadrupletTrainer")
        start_lr = start_lr if start_lr is not None else (c.start_lr or 5e-4)
        decay_step = decay_step if decay_step is not None else (c.decay_step or 10000)
        decay_rate = decay_rate if decay_rate is not None else (c.decay_rate or 0.5)
        if weight_decay is None:
            weight_decay = (c.train_weight_decay or 1e-5) if c.add_weight_decay is not False else 0.0
        self.weight_decay = weight_decay
        self.params = local_trainable_parameters(model)
        self.wd_params = [p for n, p in model.named_parameters() if n.endswith(".W") and any(p is q for q in self.params)]
        self.graph_step = bool(graph_step) and self.params[0].is_cuda
        self._sched = (float(start_lr), int(decay_step), float(decay_rate))
        self._steps_done, self._graphs, self._eager_seen, self._shape_demand = 0, {}, {}, {}
        self._zarena = pm.ZeroArena()
        dev = self.params[0].device
        self._lr = torch.tensor(float(start_lr), dtype=torch.float32, device=dev)
        self._lr_value = float(start_lr)
        if self.params[0].is_cuda:
            self.opt = torch.optim.Adam(self.params, lr=self._lr, capturable=True, fused=True)
        else:
            self.opt = torch.optim.Adam(self.params, lr=float(start_lr))
        self.keep_grads, self.last_grads, self.last_outs = False, None, None

    def _set_lr(self):
        lr0, dstep, drate = self._sched
        lr = lr0 * drate ** (self._steps_done // dstep)
        if lr != self._lr_value:
            self._lr.fill_(lr)
            self._lr_value = lr

    def forward_loss(self, points, R, sample_idx):
        outs = local_training_outputs(self.model, points, R, sample_idx)
        if self.keep_grads:
            self.last_outs = outs
        return losses.compute_loss(outs, self.cfg)

    def _body(self, arena, points, R, sample_idx):
        with pm.zero_arena(arena):
            arena.begin(points.device)
            loss = self.forward_loss(points, R, sample_idx)
            loss.backward()
        if self.wd_params and self.weight_decay:
            gs = [p.grad for p in self.wd_params if p.grad is not None]
            ps = [p.detach() for p in self.wd_params if p.grad is not None]
            if gs:
                torch._foreach_add_(gs, ps, alpha=self.weight_decay)   # d/dp [wd/2 * sum p^2] (model.py:239-243)
        return loss

    def step(self, points, R, sample_idx, sync=True):
        """One optimisation step on (points [2B,N,3], R [B,3,3], sample_idx [2B,M] int32); returns the loss (a float, or
        the device scalar with sync=False)."""
        self.model.eval()  # (the module flag: the training-mode graph is built explicitly, the fused path stays usable)
        sample_idx = sample_idx.to(torch.int32)
        key = (tuple(points.shape), tuple(R.shape), tuple(sample_idx.shape), points.device)
        graphed = self.graph_step and self._eager_seen.get(key, 0) >= 3 and not self.keep_grads
        if graphed:
            ent = self._graphs.get(key)
            if ent is None:
                static = (points.clone(), R.clone(), sample_idx.clone())
                arena = pm.ZeroArena(fixed_bytes=self._shape_demand.get(key, self._zarena.peak), device=points.device)
                self.opt.zero_grad(set_to_none=True)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    loss = self._body(arena, *static)
                    self.opt.step()
                ent = (graph, static, loss, arena)
                self._graphs[key] = ent
            graph, static, loss, _ = ent
            for dst, src in zip(static, (points, R, sample_idx)):
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src)
            self._set_lr()
            graph.replay()
        else:
            self._eager_seen[key] = self._eager_seen.get(key, 0) + 1
            self._set_lr()
            self.opt.zero_grad(set_to_none=True)
            loss = self._body(self._zarena, points, R, sample_idx)
            self._shape_demand[key] = max(self._shape_demand.get(key, 0), self._zarena.demand)
            if self.keep_grads:
                self.last_grads = [None if p.grad is None else p.grad.detach().clone() for p in self.params]
            self.opt.step()
        self._steps_done += 1
        self.model.mark_weights_changed(bn_stale=True)   # moving averages and weights moved: the inference path re-folds on its next forward; older replays refuse to run
        loss = loss.detach()
        return float(loss) if sync else loss
