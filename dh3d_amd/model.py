"""DH3D forward API on the MI355X kernels (mirrors core/model.py).

`DH3D(config)` keeps the reference's method names and named outputs:
    compute_local(points)  -> (newpoints [Bt,N,3], localdesc [Bt,N,128])      core/model.py:99-108
    compute_global(outs)   -> globaldesc [Bt,256] (un-normalised)             core/model.py:112-133
    forward(points)        -> dict with 'pointclouds', 'knn_inds', 'feat', 'feat_l2normed',
                              'xyz_feat' [Bt,N,131], 'xyz_feat_att' [Bt,N,132] (detection),
                              'globaldesc' [Bt,256] (extract_global)           core/model.py:135-210
Only inference-mode BatchNorm is implemented on the fused path (training-mode statistics: see DESIGN.md).

The forward is launch-bound when driven op by op from Python (about 30 short kernels), so
`DH3D.graphed(example)` captures it into one hipGraph (two streams: feature path + geometry path)
and replays it with static buffers.
"""
import os

import torch
from torch import nn

from . import backbones as bb
from . import pm
from .configs import ConfigFactory, dotdict


def tf_variable_name(state_dict_key):
    """state_dict key -> TensorFlow checkpoint variable name (models/*/*.index)."""
    name = state_dict_key.replace(".", "/")
    name = name.replace("mean_EMA", "mean/EMA").replace("variance_EMA", "variance/EMA")
    return name


KNN_GRID = pm.KNN_GRID


def _copy_in(src, dst):
    """dst (a graph's static input) <- src: device tensors by copy_, pinned host tensors by the staging kernel."""
    if (not src.is_cuda and src.is_pinned() and src.is_contiguous() and src.dtype == dst.dtype
            and src.numel() == dst.numel()):
        pm.stage_copy(src, dst)
    else:
        dst.copy_(src, non_blocking=True)


# Development A/B knobs (tools/single_stream_ab.py, the placement-hint sweeps): read only when DH3D_DEBUG_KNOBS=1, so a
# production process cannot pick one up from a stray environment variable.
_DEBUG_KNOBS = os.environ.get("DH3D_DEBUG_KNOBS") == "1"
_DEV_RESERVE = int(os.environ["DH3D_DEV_RESERVE"]) if _DEBUG_KNOBS and os.environ.get("DH3D_DEV_RESERVE") else None


class DH3D(nn.Module):
    def __init__(self, config=None):
        super().__init__()
        if config is None:
            config = ConfigFactory("basic_config").getconfig()
        self.config = config if isinstance(config, dotdict) else dotdict(config)
        cfg = self.config
        if cfg.local_backbone not in (None, "backbone_local_dilate"):
            raise NotImplementedError("local_backbone %r" % cfg.local_backbone)
        tp_eps = cfg.tp_bn_eps if cfg.tp_bn_eps is not None else 1e-5
        slim_eps = cfg.slim_bn_eps if cfg.slim_bn_eps is not None else 1e-3
        self.input_knn_indices = bool(cfg.num_points and cfg.num_points > 8192)  # core/model.py:38
        self.knn_num = cfg.knn_num or 8

        # ---- local backbone: variables live at the root scope (initconv, stage1, ...)
        local = bb.BackboneLocalDilate(featdim=cfg.featdim or 128, dilate2=8, bn_eps=tp_eps)
        for name, child in list(local.named_children()):
            self.add_module(name, child)
        self._local_names = [n for n, _ in local.named_children()]

        if cfg.detection:
            self.detection_block_reliable = bb.PointMLPHead(cfg.featdim or 128, [128, 256, 1024], fc_bias_init=1.0 / 8,
                                                            bn_eps=tp_eps)
        if cfg.extract_global:
            if cfg.global_backbone not in (None, "global_before_assemble", "global_before_assemble_conv1d"):
                raise NotImplementedError("global_backbone %r (core/backbones.py:178-197 knows 'global_before_assemble' "
                                          "and 'global_before_assemble_conv1d')" % cfg.global_backbone)
            if cfg.concat_xyz and cfg.global_backbone == "global_before_assemble_conv1d":
                raise NotImplementedError("concat_xyz=True with global_before_assemble_conv1d (core/backbones.py:191-192: "
                                          "a 131-channel 1x1 conv) is not built; the shipped global_config sets False")
            if cfg.global_subsample and cfg.global_subsample > 0:
                raise NotImplementedError("global_subsample > 0 (core/model.py:119-121) is not built -- nor upstream: that "
                                          "branch calls backbones.subsample / self.global_subsample, neither of which "
                                          "exists in the reference; the shipped global_config sets -1")
            if (cfg.featdim or 128) != 128:
                raise NotImplementedError("extract_global with featdim < 128 is not built: the global flex_conv and "
                                          "attention kernels are instantiated for the 128-d descriptor of the "
                                          "shipped configs (core/configs.py:58)")
            gl_dims = list(cfg.gl_dims or [256])
            self.global_conv1d = cfg.global_backbone == "global_before_assemble_conv1d"
            if self.global_conv1d:
                # core/backbones.py:189-197 ("conv1d is found to be better than flexconv for global descriptor"): 1x1
                # convs + BNReLU on the full-resolution descriptors; EVERY conv of the loop reads localdesc and only the
                # last one's output is used (the others still exist as variables upstream, so they do here)
                for i, d in enumerate(gl_dims):
                    setattr(self, "global_before_assemble_conv1%d" % i, bb.Conv2D1x1(128, d, bn_eps=tp_eps))
                self.global_before_assemble = None
            else:
                cx = bool(cfg.concat_xyz)  # core/backbones.py:180-181: [points | localdesc] into the flex_conv
                self.global_before_assemble = bb.FlexConvDilate(128 + (3 if cx else 0), gl_dims,
                                                                dilate=cfg.gl_dilate or 8, knn=self.knn_num, concat=False,
                                                                add_se="", upsample=True, bn_eps=tp_eps, xyz_prefix=cx)
            conv_dims = [256, 1024] if gl_dims[-1] > 256 else [1024]  # backbones.py:159-162
            self.globalatt = bb.PointMLPHead(gl_dims[-1], conv_dims, bn_eps=tp_eps)
            nv = bb.NetVLAD(gl_dims[-1], 64, 256, add_batch_norm=cfg.add_batch_norm is not False,
                            slim_bn_eps=slim_eps)
            # NetVLAD variables are created at the root variable scope (backbones.py:212-276)
            for vname in ("cluster_weights", "cluster_bn", "cluster_biases", "cluster_weights2", "hidden1_weights", "bn",
                          "gating_weights", "gating_bn", "gating_biases"):
                if hasattr(nv, vname):
                    setattr(self, vname, getattr(nv, vname))
            object.__setattr__(self, "_netvlad", nv)  # not registered twice
        object.__setattr__(self, "_local", local)
        self._geo_stream = None
        self._prepared = False
        self._head_prepared = False
        self.validate_knn_inds = True  # range-check caller-provided neighbour ids (one device sync per eager call)

    # ------------------------------------------------------------------ weights
    @torch.no_grad()
    def init_synthetic(self, seed=0):
        """Random weights of the benchmark recipe: N(0,1)/sqrt(fan_in), identity BatchNorm."""
        g = torch.Generator().manual_seed(seed)
        for name, p in self.named_parameters():
            leaf = name.split(".")[-1]
            if leaf in ("gamma",):
                p.fill_(1.0)
            elif leaf in ("beta", "b", "feature_bias", "cluster_biases", "gating_biases"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif leaf == "position_theta" and p.dim() == 3:
                p.copy_(torch.randn(p.shape, generator=g) / (p.shape[1] ** 0.5))
            elif leaf == "position_bias" and p.dim() == 2:
                p.copy_(torch.randn(p.shape, generator=g) / ((8 * p.shape[0]) ** 0.5))
            elif leaf == "position_theta":  # initconv [3, 32]
                p.copy_(torch.randn(p.shape, generator=g) * 4.0)
            elif leaf == "position_bias":
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif leaf == "W":
                p.copy_(torch.randn(p.shape, generator=g) / (p.shape[2] ** 0.5))
            elif leaf in ("cluster_weights", "cluster_weights2", "gating_weights"):
                p.copy_(torch.randn(p.shape, generator=g) / (p.shape[-2] ** 0.5))
            elif leaf == "hidden1_weights":
                p.copy_(torch.randn(p.shape, generator=g) / 8.0)
        self.invalidate()
        return self

    def prepare(self):
        """Fold BatchNorm and pack weights for the kernels.  Called lazily by the forward; explicit calls are
        allowed (and harmless) after loading / moving weights."""
        self._local.prepare()
        if self.config.detection:
            self.detection_block_reliable.prepare()
        self._prepared = True
        self._prepare_head()
        return self

    def _prepare_head(self):
        if self.config.extract_global:
            for top in self._global_front():
                top.prepare()
            self.globalatt.prepare()
            self._netvlad.prepare()
        self._head_prepared = True

    def _global_front(self):
        """The modules in front of the attention head / NetVLAD: the sampled-level flex_conv block, or the 1x1 convs of
        global_before_assemble_conv1d."""
        if getattr(self, "global_conv1d", False):
            return [getattr(self, "global_before_assemble_conv1%d" % i) for i in range(len(list(self.config.gl_dims or [256])))]
        return [self.global_before_assemble]

    def invalidate(self, head_only=False):
        """Drop the folded / packed copies of the weights (rebuilt by the next forward): called whenever parameters
        may have changed -- load_state_dict, .to()/.cuda()/.float() (_apply), init_synthetic, an optimiser step
        (head_only: the global head that global_config trains; the frozen backbone's copies stay)."""
        self._head_prepared = False
        self._weights_version = self.__dict__.get("_weights_version", 0) + 1  # replays captured before this are stale
        mods = []
        if self.config.extract_global:
            for top in self._global_front() + [self.globalatt, self.__dict__.get("_netvlad")]:
                mods += [top] + (list(top.modules()) if top is not None else [])
        if not head_only:
            self._prepared = False
            self._backbone_version = getattr(self, "_backbone_version", 0) + 1  # captured graphs of it are stale now
            mods += list(self.modules()) + [self.__dict__.get("_local"), self.__dict__.get("_netvlad")]
        for m in mods:
            if m is not None and getattr(m, "_prep", None) is not None:
                m._prep = None

    def mark_weights_changed(self, bn_stale=True):
        """A trainer moved weights (or BatchNorm moving averages) on the device without going through invalidate():
        every replay captured before this (graphed(), Pipeline) refuses to run again, and -- bn_stale -- the folded
        BatchNorm copies of the inference path are rebuilt by the next forward."""
        self._weights_version = self.__dict__.get("_weights_version", 0) + 1
        if bn_stale:
            self._bn_stale = True

    @property
    def weights_version(self):
        return self.__dict__.get("_weights_version", 0)

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate()
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if "_prepared" in self.__dict__:
            self.invalidate()
        return out

    def _check_mode(self, need_head=True):
        if self.__dict__.get("_bn_stale"):
            # a trainer updated the backbone's moving averages on the device (training.backbone_local_batch_stats_hip):
            # the folded BatchNorm copies of the inference path are rebuilt now, by the first forward that needs them
            self._bn_stale = False
            self.invalidate()
        if self.training:
            raise NotImplementedError(
                "training-mode BatchNorm is not implemented on the fused path; call model.eval()")
        if not self._prepared:
            self.prepare()
        elif need_head and not self._head_prepared:
            # (the trainer invalidates the head's packed copies after every optimiser step and only runs the frozen
            #  backbone through this path: re-packing the head there would be wasted work -- and, inside a whole-step
            #  hipGraph capture, host round trips)
            self._prepare_head()

    # ------------------------------------------------------------------ two streams
    # The step's critical path is  sort -> FPS -> gather -> kNN(N/8) -> stage 2 -> concat conv -> ...  (FPS alone is
    # ~70 % of the local step).  It stays on the CALLER's stream end to end; everything that only needs the full
    # cloud (kNN(N), initconv, stage 1, the two 1x1 convs, three_nn) runs beside it on a side stream and joins
    # where stage 2 starts.  (hipGraph replay charges ~10 us per cross-queue dependency that is last to arrive, so
    # the critical chain must not hop between queues: the earlier arrangement -- FPS on the side stream --
    # paid that twice per step.)
    def _geometry(self, points, knn_inds=None, prezero_tail=False):
        geo = bb.Geometry(points, self.knn_num, fps_contract=self.config.fps_contract)
        main = torch.cuda.current_stream()
        if points.shape[1] <= 16384 and (knn_inds is None or (4096 <= points.shape[1] and self.config.fps_contract is None)):
            # Morton order + group boxes: shared by the kNN (side) and the pruned FPS (here); + the cell table when the
            # kNN is the cell-list search
            geo.ordered(cells=knn_inds is None and self.knn_num <= 8 and KNN_GRID)
        if self._geo_stream is None:
            self._geo_stream = torch.cuda.Stream(device=points.device)
        side = main if (_DEBUG_KNOBS and getattr(self, "_single_stream", False)) else self._geo_stream  # (dev: tools/single_stream_ab.py)
        fork = torch.cuda.Event()
        fork.record()  # the side stream needs the input (and the ordering) only
        geo._side = side
        # stage2 (dilate2=8) and global (gl_dilate=8) share this level.  Enqueued BEFORE the side work: hipGraph keeps
        # a node's first-captured successor on its queue, and the FPS chain is the one that must not hop.
        geo._lv = geo.level(8, self.knn_num, finish=False)
        if prezero_tail:  # the global tail's walk will run: its slot tables are built right behind three_nn (off its chain)
            geo._lv["_want_walk_plan"] = True
        # three_nn needs the sampled coordinates only.  It goes where there is slack: the FPS chain is ~0.05 us per
        # point of a cloud whatever the batch, kNN(N) + stage 1 on the side stream ~2.5 us per 1000 points of the
        # batch -- big batches of small clouds (cfg 3) leave this stream waiting for stage 1, and three_nn beside
        # the N/8 convolutions would slow those down (42 vs 17 us for 64->128 at 32x512); a few big clouds (cfg 2)
        # are the other way round and it runs on the side stream after stage 1 (compute_local).
        if self._three_nn_before_sampled_level(points):
            bb.finish_level(points, geo._lv, same_stream=True)
            if prezero_tail:  # (this stream has slack here: the side stream ends last)
                geo._tail_accum = torch.zeros((pm.global_tail_accum_size(points.shape[0], points.shape[1] // 8),),
                                              dtype=torch.float32, device=points.device)
            if getattr(self, "steps_in_flight", 1) > 1:  # the OTHER steps' FPS kernels hold CUs while stage 1 runs
                geo.busy_cus_per_xcd = min(8, (self.steps_in_flight - 1) * ((points.shape[0] + 7) // 8))
        else:
            # the FPS chain outlasts stage 1: one CU per cloud (block b -> XCD b % 8) is held with ~100 KB of LDS while
            # the persistent flex_conv kernels of stage 1 run -- they leave those CUs out (placement hint, speed only)
            # (steps_in_flight: a serving loop that overlaps consecutive batches has that many FPS kernels on the chip)
            geo.busy_cus_per_xcd = min(8, getattr(self, "steps_in_flight", 1) * ((points.shape[0] + 7) // 8))
        if _DEV_RESERVE is not None:  # dev A/B (DH3D_DEV_RESERVE=n: the placement hint of the persistent kernels)
            geo.busy_cus_per_xcd = _DEV_RESERVE
        side.wait_event(fork)
        with torch.cuda.stream(side):
            if knn_inds is not None:
                geo.nbr = knn_inds.contiguous()
            elif points.shape[1] > 16384:
                # beyond the Morton-ordered kernels' range (14-bit point ids in the sort keys): the brute-force kernel,
                # any N, same ids bit for bit (the reference needs host sklearn indices here: core/model.py:148-155,
                # core/utils.py:53-57)
                geo.nbr, _ = pm.knn_xyz(points, self.knn_num)
            else:
                srt, gbox = geo.ordered()
                if geo.cells is not None:   # cell lists on the sort's grid; crowded clouds: the pruned scan (csrc/knn.hip dh3d_knn_grid)
                    geo.nbr, _ = pm.knn_grid(srt, gbox, geo.cells, self.knn_num)  # core/model.py:157
                else:                       # the pruned shared scan (K > 8)
                    geo.nbr, _ = pm.knn_sorted(srt, gbox, self.knn_num)
            geo.nbr.record_stream(main)
        return geo

    def _stage1_tails(self, n_per_cloud):
        """(tail on stage 1's output, tail on before_stage2_conv1d's output) for FlexConvDilate.forward(post_tails=...), or
        None where the separate launches are what applies (the same rules as Conv2D1x1.forward / commuted_partial: the
        points per cloud only, never the batch)."""
        sc = self.local_stage1_shortcut.tfconv0
        cc = self.stage2.concat_conv1d.tfconv0 if self.stage2.concat else None
        if (cc is None or n_per_cloud <= 4096 or not (self.stage2.upsample and self.stage2.dilate > 1)
                or not cc.commuted_supported(self.stage2.outdims[-1]) or sc.cin != 64 or sc.cout != 128):
            return None
        ps, pc = sc._prep or sc.prepare(), cc._prep or cc.prepare()
        if "wp3" not in ps or "wp3_bot" not in pc or pc["W2"].shape[0] - pc.get("c_top", 0) != 64 or cc.cout != 128:
            return None
        return ((ps["wp3"], ps["b"], ps["scale"], ps["shift"], pm.ACT_RELU), (pc["wp3_bot"], None, None, None, pm.ACT_NONE))

    def _three_nn_before_sampled_level(self, points):
        """Where three_nn (+ the sampled set's sort) is enqueued: on the MAIN stream between the sampled set's kNN and its
        convolutions, or on the side stream behind stage 1 (beside those convolutions).  Placement only -- the results
        are the same bits.  Measured on the same box (tools/side_critical_ab.py, round 5):
                                   one step at a time          steps in flight (global 2, local 4)
            global  main / side    0.7238 / 0.7025 ms          0.5935 / 0.6697 ms
            local   main / side    0.5217 / 0.4996 ms          0.2322 / 0.2797 ms
        One step alone wants it on the side stream (it leaves the critical chain); with other steps in flight the
        side stream's slack belongs to THEIR chip-wide kernels and the shorter side chain wins.  So the engine's mode
        decides (DH3D.steps_in_flight, set by Pipeline while it captures its slots).  Rounds 2-4 chose by a batch / cloud-size
        model of which chain ends last instead; it predicted "main" for the serial global step, where "side" measures
        3 % faster, and was removed.  Shapes outside the two measured workloads (cfg 5's 4 x 16384: tools/side_critical_ab.py
        --workload cfg5) follow the same rule: in flight 0.600 -> 0.565 ms per step on the main stream."""
        return getattr(self, "steps_in_flight", 1) > 1

    def _join_side(self, geo):
        """Everything enqueued on the side stream so far is visible to the current stream."""
        torch.cuda.current_stream().wait_stream(geo._side)

    # ------------------------------------------------------------------ reference API
    def compute_local(self, points, knn_inds=None, _geo=None, _l2cat_eps=None, _prezero_tail=False):
        """(points, local descriptors [Bt,N,featdim]) (core/model.py:157-171).  _l2cat_eps (internal): the second item is
        [points | l2_normalize(descriptors)] instead, written by the last conv's store (forward(fetch=...) when nothing
        needs the raw descriptors)."""
        self._check_mode(need_head=False)
        geo = _geo if _geo is not None else self._geometry(points, knn_inds)
        main = torch.cuda.current_stream()
        p = self._local._prep
        with torch.cuda.stream(geo._side):  # behind kNN(N), beside the FPS chain
            nn_8 = geo.nbr if geo.nbr.shape[2] == 8 else geo.nbr[:, :, 0:8].contiguous()
            # conv_pointset 3 -> 32, BNReLU, flex_pool (backbones.py:107-110) fused: the map between them is not built
            if nn_8.shape[2] == 8 and p["theta"].shape[1] in (32, 64, 128):
                init = pm.conv_pointset_pool_xyz(geo.xyz, nn_8, p["theta"], p["bias"], scale=p["scale"], shift=p["shift"],
                                                 act=pm.ACT_RELU)
            else:   # other init_feat_dim (e.g. 16): the fused pair covers Dout 32 / 64 / 128 -- the two operators apart
                init = pm.flex_pool(pm.conv_pointset_xyz(geo.xyz, nn_8, p["theta"], p["bias"], scale=p["scale"],
                                                         shift=p["shift"], act=pm.ACT_RELU), nn_8)
            # larger clouds: the shortcut conv (on stage 1's output) and the lower block of stage 2's commuted concat conv
            # (on before_stage2_conv1d's output) ride in stage 1's SE kernel -- two launches over tiles it already holds
            fuse_sc = points.shape[1] <= 4096 and self.stage2.shortcut_fusable(points.shape[1])
            tails = None if fuse_sc else self._stage1_tails(points.shape[1])
            r = self.stage1(geo, init, nbr=nn_8, post_conv=self.before_stage2_conv1d, post_tails=tails)
            shortcut = lower = None
            if isinstance(r, tuple) and len(r) == 4:
                x1, x2, shortcut, lower = r
            else:
                x1, x2 = r if isinstance(r, tuple) else (r, self.before_stage2_conv1d(r, act=pm.ACT_RELU))
            # BNReLU(conv(x1)) + stage2 (backbones.py:123).  Large clouds: the shortcut conv runs INSIDE stage 2's last
            # conv (its input x1 is just more K for that GEMM, with its own accumulators and epilogue), so its
            # [Bt,N,128] result is never written or read back.  Otherwise it runs here, beside the FPS chain, and its
            # sum is folded into stage 2's last store.
            # (only for clouds of <= 4096 points, where this stream tends to be the step's critical chain: beside a
            #  longer FPS chain the separate conv is free and the fusion just adds K to the critical tail -- same-box
            #  A/B: cfg 3 -23 us, cfg 2 +6 us.  The rule looks at the points per cloud only, never at the batch: the
            #  two forms differ in the rounding of the final sum and a sharded batch must reproduce the unsharded
            #  result bit for bit.)
            # the step's tail as ONE launch (csrc/dense_tail.hip): when only [xyz | l2-normalised descriptors] are wanted
            # the shortcut conv and the concat conv's lower block run INSIDE the gather / epilogue kernel behind the
            # sampled level -- their two [Bt,N,128] maps (67 MB written and read back at 8 x 8192) never exist
            cconv = self.stage2.concat_conv1d.tfconv0 if getattr(self.stage2, "concat", False) else None
            small = points.shape[1] <= 4096
            fused_tail = ((_l2cat_eps is not None or small) and shortcut is None and lower is None and cconv is not None
                          and self._local.featdim == 128 and cconv.commuted_supported(128)
                          and cconv.tail_fusable(self.local_stage1_shortcut.tfconv0, points.shape[1],
                                                 in_flight=getattr(self, "steps_in_flight", 1) > 1))
            if fused_tail:
                fuse_sc = False   # (the one-launch tail replaces the K = 256 GEMM with the shortcut fused into it)
            if shortcut is None and not fuse_sc and not fused_tail:
                shortcut = self.local_stage1_shortcut(x1, act=pm.ACT_RELU)
            # larger clouds: stage 2's concat conv is commuted through its up-sampling -- its lower weight block meets
            # x2 here, beside the sampling chain; behind the sampled level only a GEMM on the N/8 rows and one
            # gather / epilogue kernel are left (backbones.Conv2D1x1.forward_commuted)
            if lower is None and not fuse_sc and not fused_tail:
                lower = self.stage2.commuted_partial(x2)
            stage1_done = torch.cuda.Event()
            stage1_done.record()
            if _prezero_tail and getattr(geo, "_tail_accum", None) is None:
                # the global tail's accumulators (6 MB at cfg 3), zero-filled HERE, beside the sampling chain: the fill
                # (a ~5 us node + its dependency gap) is off the critical chain when the tail starts.  BEHIND the
                # stage1_done record (round 6): in front of it the sampled level on the main stream waited for the fill too
                # (its consumer, the walk, joins this stream through three_nn's event: geo.finish)
                geo._tail_accum = torch.zeros((pm.global_tail_accum_size(points.shape[0], points.shape[1] // 8),),
                                              dtype=torch.float32, device=points.device)
                geo._tail_accum.record_stream(main)
            geo.start_nn3(geo._lv)  # three_nn: waits for the sampled coordinates, overlaps the N/8 convolutions
            for t in (x2, x1 if (fuse_sc or fused_tail) else shortcut, geo._lv["nn3_dist"], geo._lv["nn3_idx"], lower,
                      geo._lv.get("walk_plan")):
                if t is not None:
                    t.record_stream(main)
        main.wait_event(stage1_done)  # not the whole side stream: three_nn is joined at the interpolation (geo.finish)
        l2cat = (points, _l2cat_eps) if _l2cat_eps is not None else None
        feat = self.stage2(geo, x2, residual=shortcut, l2cat=l2cat,  # gather, N/8 convs, SE, interpolation, concat conv
                           shortcut_src=x1 if fuse_sc else None, lower_partial=lower,
                           fused_tail=(x1, self.local_stage1_shortcut.tfconv0) if fused_tail else None)
        if self._local.featdim < 128:  # core/backbones.py:125-126
            feat = self.final_fc(feat, act=pm.ACT_RELU)
        self._last_geo = geo
        return points, feat

    def compute_global(self, outs, l2_eps=0.0):
        self._check_mode()
        points, localdesc = outs["xyz"], outs["feat"]
        if getattr(self, "global_conv1d", False):
            # full-resolution rows: no sampled level, no interpolation -- the wide 1x1-conv GEMM, the head on the
            # materialised rows, the row-streaming NetVLAD kernel
            last = getattr(self, "global_before_assemble_conv1%d" % (len(list(self.config.gl_dims or [256])) - 1))
            forglobal = last(localdesc, act=pm.ACT_RELU)
            return self._netvlad(forglobal, self.globalatt(forglobal), l2_eps=l2_eps)
        geo = outs.get("_geo")
        if geo is None:
            geo = self._geometry(points, None)
            self._join_side(geo)
        lv = geo.level(self.global_before_assemble.dilate, self.knn_num, finish=False)
        nv, ga = self._netvlad, self.globalatt
        m = lv["xyz_s"].shape[1]
        if ("_ordered" in lv and points.shape[1] >= 4096 and m <= 1024 and nv.add_batch_norm
                and self.global_before_assemble.outdims[-1] == 256
                and "wslices" in (ga._prep or ga.prepare())):
            # Both consumers of the up-sampled map -- the attention MLP and NetVLAD's soft assignment / aggregation --
            # are reached through the interpolation's linearity: the fine points are walked once (Morton order, coarse
            # rows staged in LDS), the [Bt, N, 256] map is never built (rule on the points per cloud only).
            p = nv._prep or nv.prepare()
            coarse = self.global_before_assemble(geo, localdesc, coarse_only=True, post_linear=(p["wc"], 64))
            cw = self.global_before_assemble._last_post  # coarse @ cluster_weights out of the same launch (or None)
            acc = getattr(geo, "_tail_accum", None)
            if acc is not None and acc.numel() != pm.global_tail_accum_size(points.shape[0], m):
                acc = None
            geo._tail_accum = None  # (one use: the tail accumulates into it)
            last = ga.detec_conv0
            lp, gp = last._prep, ga._prep
            return pm.global_tail(coarse, lv["nn3_idx"], lv["nn3_dist"], lv["_ordered"][0], gp["wslices"], last.cout,
                                  gp["w_fc"], gp["b_fc"], (lp["b"], lp["scale"], lp["shift"], pm.ACT_RELU), p["wc"],
                                  p["cs"], p["ch"], p["W2"], p["Wh"], p["s1"], p["h1"], p["Wg"], p["s2"], p["h2"],
                                  l2_eps=l2_eps, accum=acc, cw=cw, plan=lv.get("walk_plan"))
        forglobal = self.global_before_assemble(geo, localdesc)
        coarse, lv = getattr(self.global_before_assemble, "_last_coarse", (None, None))
        if coarse is not None and "nn3_idx" in lv and self.globalatt.interpolated_supported(coarse, lv["nn3_idx"]):
            # the attention MLP's 256 -> 1024 conv commutes with the up-sampling: it runs on the N/8 level
            order = lv["_ordered"][0] if "_ordered" in lv else None  # Morton records of the full cloud
            att = self.globalatt.forward_interpolated(coarse, lv["nn3_idx"], lv["nn3_dist"], order=order)
        else:
            att = self.globalatt(forglobal)
        return self._netvlad(forglobal, att, l2_eps=l2_eps)

    OUTPUT_NAMES = frozenset(("pointclouds", "xyz", "knn_inds", "feat", "xyz_feat", "feat_l2normed", "attention",
                              "xyz_feat_att", "globaldesc", "fps_inds", "sampled_knn_inds", "nn3_inds"))

    def forward(self, points, knn_inds=None, fetch=None):
        """points [Bt, N, 3] float32 on the GPU (anchor/pos/neg already concatenated, core/model.py:139-146).
        knn_inds [Bt, N, K] int32: optional precomputed neighbours (the reference requires them for
        num_points > 8192, core/model.py:148-155; here the device kNN serves every N: the Morton-pruned kernels up to
        16384 points, the brute-force kernel beyond).
        fetch: names of the outputs wanted (None = all).  Like a TF session fetch, tensors nobody asked for are not
        computed: the global-descriptor extraction (globaldesc_extract.py fetches 'globaldesc' only) skips the
        normalised per-point descriptors and the detector."""
        want = (lambda *names: True) if fetch is None else (lambda *names: any(n in fetch for n in names))
        if fetch is not None:
            unknown = sorted(set(fetch) - self.OUTPUT_NAMES)
            if unknown:  # a TF session raises on an unknown fetch too; silently computing nothing would pass for speed
                raise ValueError("unknown output name(s) %s: the forward produces %s" % (unknown, sorted(self.OUTPUT_NAMES)))
        self._check_mode()
        cfg = self.config
        if points.dim() != 3 or points.shape[2] != 3:
            raise ValueError("points must be [Bt, N, 3]")
        if knn_inds is not None:
            # the kernels never bounds-check neighbour ids (nor does the reference): do it here
            if (knn_inds.dim() != 3 or knn_inds.shape[0] != points.shape[0] or knn_inds.shape[1] != points.shape[1]
                    or knn_inds.shape[2] < 8 or knn_inds.dtype != torch.int32 or knn_inds.device != points.device):
                raise ValueError("knn_inds must be int32 [Bt, N, K>=8] on the device of points, got %s %s"
                                 % (tuple(knn_inds.shape), knn_inds.dtype))
            if self.validate_knn_inds and not torch.cuda.is_current_stream_capturing():
                lo, hi = int(knn_inds.min()), int(knn_inds.max())  # one sync; switch off with validate_knn_inds=False
                if lo < 0 or hi >= points.shape[1]:
                    raise ValueError("knn_inds out of range [0, %d): min %d max %d (kNN pads with -1 when N < K)"
                                     % (points.shape[1], lo, hi))
        # num_points > 8192: the reference feeds host (sklearn) kNN indices because its op stops at 8192
        # (core/model.py:38,148-155); they are still accepted, but the device search covers every N itself.
        outs = {"pointclouds": points, "xyz": points}
        prezero = bool(cfg.extract_global and want("globaldesc") and getattr(self, "global_before_assemble", None) is not None
                       and self.global_before_assemble.dilate == 8)
        geo = self._geometry(points, knn_inds, prezero_tail=prezero)
        outs["knn_inds"] = geo.nbr
        needs_raw = want("feat", "attention", "xyz_feat_att", "globaldesc")
        if fetch is not None and not needs_raw and want("xyz_feat", "feat_l2normed") and self._local.featdim == 128:
            # only the normalised descriptors are asked for: the last conv writes [xyz | l2_normalize(feat)] itself
            _, xyz_feat = self.compute_local(points, _geo=geo, _l2cat_eps=1e-8)
            outs["xyz_feat"] = xyz_feat
            outs["feat_l2normed"] = xyz_feat[:, :, 3:]
            self._level_ids(geo, outs, fetch)
            return outs
        newpoints, localdesc = self.compute_local(points, _geo=geo, _prezero_tail=prezero)
        outs["feat"] = localdesc
        self._level_ids(geo, outs, fetch)
        xyz_feat = None
        if want("xyz_feat", "feat_l2normed", "xyz_feat_att"):
            xyz_feat = pm.l2norm_concat(localdesc, 1e-8, prefix=newpoints)  # l2_normalize(dim=2, eps=1e-8) + concat
            outs["xyz_feat"] = xyz_feat
            outs["feat_l2normed"] = xyz_feat[:, :, 3:]
        if cfg.detection and want("attention", "xyz_feat_att"):
            att = self.detection_block_reliable(localdesc)
            outs["attention"] = att
            if xyz_feat is not None:
                outs["xyz_feat_att"] = torch.cat([xyz_feat, att], dim=-1)
        if cfg.extract_global and want("globaldesc"):
            outs["_geo"] = geo
            outs["globaldesc"] = self.compute_global(outs, l2_eps=1e-8)  # model.py:205
            del outs["_geo"]
        return outs

    @staticmethod
    def _level_ids(geo, outs, fetch):
        """The integer results of the shared N/8 level (FPS picks, sampled-set kNN ids, three_nn ids) as named outputs
        -- only on request: the parity tests of graph replays / steps in flight compare them bit for bit."""
        lv = getattr(geo, "_lv", None)
        if lv is None or fetch is None:
            return
        for name, key in (("fps_inds", "idx"), ("sampled_knn_inds", "nbr_s"), ("nn3_inds", "nn3_idx")):
            if key in lv and name in fetch:
                outs[name] = lv[key]

    # ------------------------------------------------------------------ hipGraph replay
    def pipeline(self, example_points, depth=2, outputs=None, example_knn=None, streams=None):
        """`depth` forwards in flight: one hipGraph instance + batch buffers + stream per slot (dh3d_amd/engine.py).
        Returns an engine.Pipeline: submit(batch) -> ticket, result(ticket) -> outputs, map(batches)."""
        from .engine import Pipeline
        return Pipeline(self, example_points, depth=depth, outputs=outputs, example_knn=example_knn, streams=streams)

    def graphed(self, example_points, example_knn=None, outputs=None, warmup=2):
        """Capture forward() for inputs shaped like `example_points` into a hipGraph.

        Returns a callable f(points[, knn_inds]) -> dict of output tensors (static buffers, overwritten by
        the next call).  Zero-copy hand-over: write the batch into `f.static_input` (`f.static_knn`) -- e.g. as the
        destination of the host-to-device copy -- and call f() / f(f.static_input); any other tensor is copied in
        (a ~5 us kernel plus two dependency gaps per step)."""
        self._check_mode()
        static_in = example_points.clone()
        static_knn = example_knn.clone() if example_knn is not None else None
        keep = outputs
        s = torch.cuda.Stream(device=example_points.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self.forward(static_in, static_knn, fetch=keep)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            outs = self.forward(static_in, static_knn, fetch=keep)
        if keep is not None:
            outs = {k: v for k, v in outs.items() if k in keep}
        version = self.weights_version

        def run(points=None, knn_inds=None):
            if self.weights_version != version:
                raise RuntimeError("the model's weights changed (optimiser step / invalidate / load_state_dict) after this "
                                   "forward was captured: the graph holds packed copies of the old ones -- capture again")
            # (a pinned HOST batch is read by a staging KERNEL on the current stream, in front of the replay -- no copy
            # engine on the step's chain; engine.Pipeline.submit hands host batches over this way)
            if points is not None and points is not static_in:
                _copy_in(points, static_in)
            if static_knn is not None and knn_inds is not None and knn_inds is not static_knn:
                _copy_in(knn_inds, static_knn)
            graph.replay()
            return outs

        run.graph = graph
        run.outputs = outs
        run.static_input = static_in
        run.static_knn = static_knn
        return run
