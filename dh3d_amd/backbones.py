"""FlexConv + SE encoder, detector / attention heads and NetVLAD on the fused HIP kernels.

Mirrors core/backbones.py (se_res_bottleneck :45-55, flex_conv_dilate :58-101, backbone_local_dilate
:104-127, detection_block :132-151, globalatt_block :156-173, global_before_assemble :178-186,
global_netvald_block :202-279, context_gating :282-320) and the glue of core/tf_utils.py
(flexconv_withBatchnorm :48-64, convolution_pointset_withBatchnorm :67-83, subsample :86-96,
feature_conv1d_1 :99-109).

Differences that are deliberate:
  * activations stay point-major [B, N, C] end to end -- the reference transposes between [B,C,N] and
    [B,N,C] around every block (backbones.py:35,41,68-69,87,108,110);
  * BatchNorm (inference statistics), biases and activations are folded into the producing kernel;
  * the geometry chain FPS -> gather -> kNN(N/8) -> three_nn depends only on xyz: it runs on a side HIP
    stream concurrently with stage 1, and is computed once where the reference computes it twice
    (stage2 and global_before_assemble use identical xyz / npoint / k).

Parameter names follow the checkpoint variable names ('/' -> '.', 'mean/EMA' -> 'mean_EMA'); see
dh3d_amd.model.tf_variable_name.
"""
import math
import os

import torch
from torch import nn

from . import _lib as L
from . import pm


# --------------------------------------------------------------------------- parameter holders
class TPBatchNorm(nn.Module):
    """tensorpack BatchNorm variables: gamma, beta, mean/EMA, variance/EMA (inference statistics)."""

    ema_unbiased = True  # training: rank-4 inputs go through tf.nn.fused_batch_norm (Bessel-corrected moving variance)

    def __init__(self, channels, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.gamma = nn.Parameter(torch.ones(channels))
        self.beta = nn.Parameter(torch.zeros(channels))
        self.register_buffer("mean_EMA", torch.zeros(channels))
        self.register_buffer("variance_EMA", torch.ones(channels))

    def fold(self):
        scale = self.gamma * torch.rsqrt(self.variance_EMA + self.eps)
        shift = self.beta - self.mean_EMA * scale
        return scale.contiguous(), shift.contiguous()


class SlimBatchNorm(nn.Module):
    """slim / tf.contrib.layers batch_norm variables: gamma, beta, moving_mean, moving_variance.  fused: whether the
    upstream layer runs the fused kernel in training (rank-2 input, fused=None: yes -- its moving variance then takes
    the Bessel-corrected batch variance; cluster_bn is fused=False, core/backbones.py:218-223)."""

    def __init__(self, channels, eps=1e-3, fused=True):
        super().__init__()
        self.eps = eps
        self.ema_unbiased = bool(fused)
        self.gamma = nn.Parameter(torch.ones(channels))
        self.beta = nn.Parameter(torch.zeros(channels))
        self.register_buffer("moving_mean", torch.zeros(channels))
        self.register_buffer("moving_variance", torch.ones(channels))

    def fold(self):
        scale = self.gamma * torch.rsqrt(self.moving_variance + self.eps)
        shift = self.beta - self.moving_mean * scale
        return scale.contiguous(), shift.contiguous()


class Conv2D1x1(nn.Module):
    """tensorpack Conv2D(kernel_shape=1): W [1,1,Cin,Cout], b [Cout], optional BatchNorm 'bn'."""

    def __init__(self, cin, cout, bn=True, bn_eps=1e-5, bias_init=0.0):
        super().__init__()
        self.cin, self.cout = cin, cout
        self.W = nn.Parameter(torch.randn(1, 1, cin, cout) * math.sqrt(2.0 / cin))
        self.b = nn.Parameter(torch.full((cout,), float(bias_init)))
        self.bn = TPBatchNorm(cout, bn_eps) if bn else None
        self._prep = None

    def prepare(self):
        W2 = self.W.detach().reshape(self.cin, self.cout).contiguous()
        p = {"W2": W2, "b": self.b.detach().contiguous()}
        if self.cout % 64 == 0 and self.cin % 8 == 0:
            p["wp"] = pm.pack_weight(W2)
        elif self.cin % 8 == 0 and self.cout > 1:
            # narrow outputs (final_fc with featdim < 128, backbones.py:125-126): the GEMM runs on zero-padded columns
            # and the caller's view drops them
            pad = (-self.cout) % 64
            p["pad"] = pad
            p["wp"] = pm.pack_weight(torch.cat([W2, W2.new_zeros(self.cin, pad)], 1).contiguous())
        if self.cout in (128, 256) and self.cin % 32 == 0:
            p["wp3"] = pm.pack_weight_x3(W2)  # large row counts go to the tiled bf16x6 GEMM
        if self.bn is not None:
            p["scale"], p["shift"] = [t.detach() for t in self.bn.fold()]
        else:
            p["scale"], p["shift"] = None, None
        if p.get("pad"):
            z = W2.new_zeros(p["pad"])
            p["b"] = torch.cat([p["b"], z])
            if p["scale"] is not None:
                p["scale"], p["shift"] = torch.cat([p["scale"], z + 1]), torch.cat([p["shift"], z])
        self._prep = p
        return p

    def forward(self, x, x2=None, act=pm.ACT_RELU, residual=None):
        p = self._prep or self.prepare()
        if p.get("pad"):
            if residual is not None:
                raise NotImplementedError("residual with a padded 1x1 conv")
            y = pm.linear(x, p["wp"], self.cout + p["pad"], x2=x2, pre_bias=p["b"], scale=p["scale"],
                          shift=p["shift"], act=act)
            return y[..., :self.cout].contiguous()
        # the choice depends on the points per cloud only, never on the batch: a sharded batch must reproduce the
        # unsharded result bit for bit
        per_cloud = x.shape[-2] if x.dim() >= 2 else 1
        if "wp3" in p and per_cloud >= 4096 and x.shape[-1] % 32 == 0 and (x2 is None or x2.shape[-1] % 32 == 0):
            return pm.linear_x6(x, p["wp3"], self.cout, x2=x2, pre_bias=p["b"], scale=p["scale"], shift=p["shift"],
                                act=act, residual=residual)
        return pm.linear(x, p["wp"], self.cout, x2=x2, pre_bias=p["b"], scale=p["scale"], shift=p["shift"],
                         act=act, residual=residual)

    def fuse_shortcut(self, other):
        """Let forward_upsampled(shortcut=x3) add act(BN_other(x3 W_other)) in the same kernel: the two weight matrices
        are stacked into one packed operand (both convs must have this cout = 128 and 32-multiple inputs)."""
        p, q = self._prep or self.prepare(), other._prep or other.prepare()
        if self.cout == 128 and other.cout == 128 and "wp3" in p and other.cin % 32 == 0:
            p["wp3_sc"] = pm.pack_weight_x3(torch.cat([p["W2"], q["W2"]], 0).contiguous())
            p["sc_ep"] = (q["b"], q["scale"], q["shift"], pm.ACT_RELU)

    def commuted_supported(self, c_top):
        """forward_commuted available: the conv splits into an upper block applied to c_top up-sampled channels and a
        lower block applied to the full-resolution input (cout = 128, both blocks GEMM-able)."""
        p = self._prep or self.prepare()
        if "wp_top" not in p and self.cout == 128 and 0 < c_top < self.cin and c_top % 8 == 0 and (self.cin - c_top) % 8 == 0:
            p["c_top"] = c_top
            p["wp_top"] = pm.pack_weight(p["W2"][:c_top].contiguous())
            p["wp_bot"] = pm.pack_weight(p["W2"][c_top:].contiguous())
            if (self.cin - c_top) % 32 == 0:  # full-resolution rows: the tiled bf16x6 GEMM (26.6 -> 19.1 us at 8 x 8192)
                p["wp3_bot"] = pm.pack_weight_x3(p["W2"][c_top:].contiguous())
        return p.get("c_top") == c_top

    def lower_partial(self, x2):
        """x2 @ W[c_top:] (no bias / BN / activation): the part of a commuted concat conv that needs the full-resolution
        input only -- the caller runs it off the critical chain."""
        p = self._prep
        if "wp3_bot" in p and x2.shape[-2] >= 4096:  # (points per cloud, never the batch: see forward)
            return pm.linear_x6(x2, p["wp3_bot"], self.cout)
        return pm.linear(x2, p["wp_bot"], self.cout)

    def forward_commuted(self, coarse, idx, dist, partial, act=pm.ACT_RELU, residual=None, l2cat=None, cw=None):
        """forward([three_interpolate_idw(coarse) | x2]) with the upper weight block applied to the COARSE rows (the
        interpolation is linear and acts on rows, so it commutes with the conv) and `partial` = lower_partial(x2):
        a [B*M, c_top] x [c_top, 128] GEMM + one gather / epilogue kernel behind the sampled level instead of the
        [B*N, cin] x [cin, 128] GEMM with the up-sampling fused into its staging."""
        p = self._prep
        if cw is None:  # (the caller may have it already: it rides in the SE kernel of the sampled level)
            cw = pm.linear(coarse, p["wp_top"], self.cout)
        return pm.interp_combine(cw, idx, dist, partial, pre_bias=p["b"], scale=p["scale"], shift=p["shift"], act=act,
                                 residual=residual, l2cat=l2cat)

    def tail_fusable(self, shortcut_conv, n, in_flight=False):
        """forward_commuted_fused available: this (commuted) concat conv and the caller's shortcut conv are both
        64 -> 128 on full-resolution rows of n points per cloud.  Clouds of up to 4096 points (where the tail is the
        step's critical chain and the one-launch form replaces the K = 256 GEMM with the up-sampling fused into its
        staging: global serial -20 us) take it.  Larger clouds by the ENGINE'S MODE, like three_nn's placement
        (DH3D._three_nn_before_sampled_level): one step at a time the two GEMMs are free beside the sampling chain and
        the one-launch tail only lengthens the chain behind it (0.4747 -> 0.4801 ms); with steps in flight that slack
        belongs to the other steps' kernels and 25 us of chip time + 130 MB of traffic less per step win (four in flight
        0.2388 -> 0.2292 ms, round 6; placement only, same values).  DH3D_TAIL_FUSED=0 / 1 forces it off / on."""
        p, q = self._prep or self.prepare(), shortcut_conv._prep or shortcut_conv.prepare()
        big = in_flight if TAIL_FUSED is None else TAIL_FUSED
        return ((big if n > 4096 else TAIL_FUSED_SMALL) and self.cout == 128 and p.get("c_top") == 128
                and self.cin - 128 == 64 and "wp3_bot" in p and shortcut_conv.cin == 64 and shortcut_conv.cout == 128
                and "wp3" in q and n % 32 == 0 and n >= 1024)

    def forward_commuted_fused(self, coarse, idx, dist, x2, x1, shortcut_conv, l2cat, cw=None):
        """forward_commuted(partial = lower_partial(x2), residual = relu(BN(shortcut_conv(x1))), l2cat) with both 64 -> 128
        GEMMs inside the tail kernel: their [B,N,128] outputs are never written (pm.local_tail_fused)."""
        p, q = self._prep, shortcut_conv._prep
        if cw is None:
            cw = pm.linear(coarse, p["wp_top"], self.cout)
        return pm.local_tail_fused(x1, x2, q["wp3"], p["wp3_bot"], (q["b"], q["scale"], q["shift"]),
                                   (p["b"], p["scale"], p["shift"]), cw, idx, dist,
                                   l2cat[0] if l2cat is not None else None, l2cat[1] if l2cat is not None else 0.0)

    def upsampled_supported(self, coarse, idx, x2):
        """Can forward_upsampled serve this call?  (same batch-independent rule as forward's x6 choice)"""
        p = self._prep or self.prepare()
        return ("wp3" in p and idx.shape[1] >= 4096 and coarse.shape[-1] % 32 == 0
                and (x2 is None or x2.shape[-1] % 32 == 0))

    def forward_upsampled(self, coarse, idx, dist, x2=None, act=pm.ACT_RELU, residual=None, l2cat=None, shortcut=None):
        """forward([three_interpolate_idw(coarse, idx, dist) | x2]) with the up-sampling fused into the GEMM
        (l2cat: and the l2-normalise + xyz concat of core/model.py:177-181 fused into its store; shortcut = x3:
        + the conv given to fuse_shortcut applied to x3, instead of `residual`)."""
        p = self._prep or self.prepare()
        if shortcut is not None:
            return pm.upsample_linear_shortcut_x6(coarse, idx, dist, p["wp3_sc"], self.cout, x2, shortcut,
                                                  (p["b"], p["scale"], p["shift"], act), p["sc_ep"], l2cat=l2cat)
        return pm.upsample_linear_x6(coarse, idx, dist, p["wp3"], self.cout, x2=x2, pre_bias=p["b"], scale=p["scale"],
                                     shift=p["shift"], act=act, residual=residual,
                                     l2cat=l2cat if self.cout == 128 else None)


class FeatureConv1d(nn.Module):
    """feature_conv1d_1 (core/tf_utils.py:99-109): variable scope '<name>/tfconv0'."""

    def __init__(self, cin, cout, bn=True, bn_eps=1e-5):
        super().__init__()
        self.tfconv0 = Conv2D1x1(cin, cout, bn=bn, bn_eps=bn_eps)

    def forward(self, x, x2=None, act=pm.ACT_RELU, residual=None):
        return self.tfconv0(x, x2=x2, act=act, residual=residual)


class FlexConvParams(nn.Module):
    """Weights of one FlexConvolution layer (core/layers.py:268-295)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.cin, self.cout = cin, cout
        limit = math.sqrt(6.0 / (cin + cout))
        self.position_theta = nn.Parameter(torch.empty(3, cin, cout).uniform_(-limit, limit))
        self.position_bias = nn.Parameter(torch.zeros(cin, cout))
        self.feature_bias = nn.Parameter(torch.zeros(cout, 1))


class SEBlock(nn.Module):
    """se_res_bottleneck weights: f1 (C -> C/4, ReLU), f2 (C/4 -> C, sigmoid); no BatchNorm."""

    def __init__(self, channels):
        super().__init__()
        self.channels = channels
        self.f1 = FeatureConv1d(channels, channels // 4, bn=False)
        self.f2 = FeatureConv1d(channels // 4, channels, bn=False)
        self._prep = None

    def prepare(self):
        c = self.channels
        W1 = self.f1.tfconv0.W.detach().reshape(c, c // 4).contiguous()
        b1 = self.f1.tfconv0.b.detach().contiguous()
        W2 = self.f2.tfconv0.W.detach().reshape(c // 4, c).contiguous()
        b2 = self.f2.tfconv0.b.detach().contiguous()
        packed = pm.se_res_pack(W1, b1, W2) + (b2,) if c in (64, 128) and c // 4 <= 32 else None
        self._prep = (W1, b1, W2, b2, packed)
        return self._prep

    def forward(self, x, pool):
        W1, b1, W2, b2, packed = self._prep or self.prepare()
        if packed is not None:
            return pm.se_res_packed(x, pool, *packed)
        return pm.se_res(x, pool, W1, b1, W2, b2)

    def forward_on_max_pool(self, x, nbr):
        """forward(x, flex_pool(x, nbr)) (core/backbones.py:76-79); one launch where the matrix-pipe kernel applies."""
        W1, b1, W2, b2, packed = self._prep or self.prepare()
        if packed is not None and x.dim() == 3 and x.shape[2] in (64, 128):
            return pm.se_res_pool_packed(x, nbr, *packed)
        return self.forward(x, pm.flex_pool(x, nbr))

    def forward_on_max_pool_then_conv(self, x, nbr, conv, act=pm.ACT_RELU, tails=None):
        """(y, conv(y)) with y = forward_on_max_pool(x, nbr): one launch when the block is 64 wide and `conv` a 64 -> 64
        Conv2D1x1 (stage 1 -> before_stage2_conv1d, core/backbones.py:115-117); same values either way.
        tails = (tail on y, tail on conv(y)), each (x3-packed [64,128] weight, pre_bias, scale, shift, act): two more 1x1
        convs in the same launch -- returns (None, conv(y), tail outputs...) then (y is not stored), or the 2-tuple when
        the fused kernel does not apply (the caller then runs the tails itself)."""
        W1, b1, W2, b2, packed = self._prep or self.prepare()
        inner = getattr(conv, "tfconv0", conv)  # FeatureConv1d wraps its Conv2D1x1
        cp = inner._prep or inner.prepare()
        if (packed is not None and x.dim() == 3 and x.shape[2] == 64 and inner.cin == 64 and inner.cout == 64
                and "wp" in cp and not cp.get("pad")):
            if tails is not None and SE_TAILS:
                return pm.se_res_pool_conv_tails(x, nbr, *packed, cp["wp"], cp["b"], cp["scale"], cp["shift"], tails[0],
                                                 tails[1], act=act, store_y=False)
            return pm.se_res_pool_conv(x, nbr, *packed, cp["wp"], cp["b"], cp["scale"], cp["shift"], act=act)
        y = self.forward_on_max_pool(x, nbr)
        return y, conv(y, act=act)


# --------------------------------------------------------------------------- geometry (xyz-only work)
class Geometry(object):
    """Everything the forward needs that depends on coordinates only."""

    def __init__(self, xyz, knn_num, nbr=None, fps_contract=None):
        self.fps_contract = fps_contract
        self.xyz = xyz
        self.knn_num = knn_num
        self.nbr = nbr          # [B,N,K] int32
        self.levels = {}        # dilate -> dict(idx, xyz_s, nbr_s, nn3_dist, nn3_idx)
        self.sorted = None      # (records [B,N,4], gbox) from pm.spatial_sort, shared by kNN and FPS
        self.cells = None       # [B, CELL_INTS] cell table of the sort (pm.spatial_sort_cells) or None

    def ordered(self, cells=False):
        """(Morton-ordered records, group boxes) of the cloud; cells=True: the sort also writes the 16^3 cell table the
        cell-list kNN searches (self.cells)."""
        if self.sorted is None:
            if cells:
                srt, gbox, self.cells = pm.spatial_sort_cells(self.xyz)
                self.sorted = (srt, gbox)
            else:
                self.sorted = pm.spatial_sort(self.xyz)
        return self.sorted

    def level(self, dilate, knn, finish=True):
        key = (dilate, knn)
        if key not in self.levels:
            self.levels[key] = compute_level(self.xyz, dilate, knn, ordered=self.sorted,
                                             fps_contract=self.fps_contract, cells=self.cells)
        return self.finish(self.levels[key]) if finish else self.levels[key]

    def start_nn3(self, lv):
        """Enqueue three_nn of a level on the CURRENT stream (it first waits for the sampled coordinates) and leave
        an event for `finish`; meant for a stream that runs beside the consumer's N/8 convolutions."""
        if "_nn3_done" in lv or "nn3_idx" in lv:
            return
        finish_level(self.xyz, lv)
        done = torch.cuda.Event()
        done.record()
        lv["_nn3_done"] = done

    def finish(self, lv):
        """Make nn3_dist / nn3_idx of a level usable on the current stream."""
        if "_nn3_done" in lv:
            # ONE wait per consuming stream: a second wait_event on the same stream is a second cross-queue edge in the
            # captured graph -- it lands on whatever node comes next (the global block's slices GEMM, which does not read
            # three_nn's result) and costs that node the ~6 us of a cross-queue dependency (round 6 timeline)
            cur = torch.cuda.current_stream()
            joined = lv.setdefault("_nn3_joined", [])
            if not any(s == cur for s in joined):
                cur.wait_event(lv["_nn3_done"])
                joined.append(cur)
            return lv
        return finish_level(self.xyz, lv)


def gather_rows(points, idx):
    """points [B,N,C], idx [B,m] -> [B,m,C]  (group_point with nsample = 1, core/tf_utils.py:92-95)."""
    p = L.require_cuda_f32(points, "points", 3)
    ix = L.require_cuda_i32(idx, "idx", 2)
    b, n, c = p.shape
    m = ix.shape[1]
    out = torch.empty((b, m, c), dtype=torch.float32, device=p.device)
    L.check(L.lib().dh3d_group_point_fwd(b, n, c, m, 1, L.ptr(p), L.ptr(ix), L.ptr(out), L.stream_ptr()),
            "group_point")
    return out


# dev A/B switch (DH3D_SE_TAILS=1: the local step's shortcut conv and the lower concat block ride in stage 1's SE kernel.
# Measured slower than the three launches -- DEADENDS.md -- so off by default; the kernel stays tested)
SE_TAILS = os.environ.get("DH3D_SE_TAILS", "0") == "1"
# dev A/B switch (DH3D_FLEX_TX6=0: the exact-f32 MFMA tile kernel for the sampled levels)
FLEX_TX6 = os.environ.get("DH3D_FLEX_TX6", "1") != "0"
# dev A/B switch (DH3D_TAIL_FUSED=0 / 1: the local step's tail -- shortcut conv, the concat conv's lower block, up-sampling,
# epilogue, l2-normalised rows -- as ONE launch, csrc/dense_tail.hip, never / always for clouds of more than 4096 points.
# 27.5 us instead of 52 us of kernels and 130 MB less HBM traffic per step, but the two GEMMs move from beside the sampling
# chain to behind it.  Unset: with steps in flight only -- Conv2D1x1.tail_fusable)
TAIL_FUSED = {"0": False, "1": True}.get(os.environ.get("DH3D_TAIL_FUSED", ""))   # None: by the engine's mode
# (DH3D_TAIL_FUSED_SMALL=0: clouds of <= 4096 points back on the K = 256 GEMM with the fused up-sampling + shortcut)
TAIL_FUSED_SMALL = os.environ.get("DH3D_TAIL_FUSED_SMALL", "1") != "0"


# dev A/B switches (round 6).  DH3D_FPS_ORDERED=0: the sampled set is sorted by its own launch again (spatial_sort_kernel<1>
# behind the sampling) instead of leaving the FPS kernel in Morton order; DH3D_SAMPLED_GRID=0: the sampled set's kNN back on
# the brute-force wave-per-query kernel instead of the cell lists on the table the FPS kernel writes.
FPS_ORDERED = os.environ.get("DH3D_FPS_ORDERED", "1") != "0"
SAMPLED_GRID = os.environ.get("DH3D_SAMPLED_GRID", "1") != "0"


def compute_level(xyz, dilate, knn, ordered=None, fps_contract=None, cells=None):
    """FPS -> gather xyz -> kNN on the sampled set (three_nn back to the full set: finish_level).

    `ordered` = (records, group boxes) of pm.spatial_sort(xyz) if the caller has them: large clouds then use the
    region-pruned FPS with several picks per synchronisation (csrc/fps.hip: 0.38 ms vs 0.82 ms at 8192 -> 1024,
    0.2 vs 0.28 ms at 4096 -> 512; no gain at 2048 and below).
    fps_contract: None = default kernels; 0 / 1 forces the any-N kernel with that distance rounding
    (ops.farthest_point_sample)."""
    B, N, _ = xyz.shape
    npoint = N // dilate
    ordered_s, cells_s = None, None
    if ordered is not None and 4096 <= N <= 8192 and fps_contract is None and FPS_ORDERED:
        # the sampled set leaves the FPS kernel IN MORTON ORDER (a stable compaction of the picked positions of the sorted
        # cloud: records, group boxes and -- with the cloud's cell table -- the subset's table on the cloud's grid):
        # three_nn and the sampled set's kNN need no sort of their own behind the sampling
        idx, xyz_s, srt_s, gbox_s, cells_s = pm.fps_sorted_ordered(ordered[0], ordered[1], npoint, cells=cells)
        ordered_s = (srt_s, gbox_s)
    elif ordered is not None and 4096 <= N <= 16384 and fps_contract is None:
        # (above 12288 points the kernel has no room for its LDS coordinate table and reads winners from the cloud)
        idx, xyz_s = pm.fps_sorted(ordered[0], ordered[1], npoint, with_xyz=True, xyz=xyz if N > 12288 else None)
    else:
        from . import ops
        idx = ops.farthest_point_sample(npoint, xyz, contract=fps_contract)  # any N (scratch distances above 16384)
        xyz_s = gather_rows(xyz, idx)
    ready = torch.cuda.Event()
    ready.record()  # xyz_s exists: three_nn may start on another stream while the sampled-set kNN runs here
    if cells_s is not None and knn <= 8 and pm.KNN_GRID and SAMPLED_GRID and npoint >= 256 and B * npoint >= 16384:
        # cell lists on the table the FPS kernel wrote (a third of the brute-force kernel's instructions per query; with a
        # sort of its own on the chain this lost: DEADENDS.md "the sampled levels on cell lists").  Only where the launch
        # fills the chip: 64 queries per workgroup -- 32 x 512 is 256 workgroups (global serial 0.6732 -> 0.6671 ms, three
        # in flight 0.5424 -> 0.5343), 8 x 1024 only 128 and latency-bound (local serial 0.4996 -> 0.5186: stays on the
        # wave-per-query kernel)
        nbr_s, _ = pm.knn_grid(ordered_s[0], ordered_s[1], cells_s, knn)
    elif npoint <= 2048 or npoint > 16384:  # small sets: the brute-force kernel beats sort + pruned search (launch /
        nbr_s, _ = pm.knn_xyz(xyz_s, knn)  # latency bound); sets beyond the Morton sort's 14-bit ids: it is what serves any N
    elif knn <= 8 and pm.KNN_GRID:
        srt_s, gbox_s, cells_s = pm.spatial_sort_cells(xyz_s)
        nbr_s, _ = pm.knn_grid(srt_s, gbox_s, cells_s, knn)
        ordered_s = (srt_s, gbox_s)
    else:
        srt_s, gbox_s = pm.spatial_sort(xyz_s)
        nbr_s, _ = pm.knn_sorted(srt_s, gbox_s, knn)
        ordered_s = (srt_s, gbox_s)
    level_ready = torch.cuda.Event()
    level_ready.record()  # idx / xyz_s / nbr_s exist
    lv = {"idx": idx, "xyz_s": xyz_s, "nbr_s": nbr_s, "_xyz_ready": ready, "_level_ready": level_ready}
    if ordered is not None:
        lv["_ordered"] = ordered            # Morton records + boxes of the full cloud: the pruned three_nn uses them
        if ordered_s is not None:
            lv["_ordered_s"] = ordered_s
    return lv


def finish_level(xyz, lv, same_stream=False):
    """three_nn of the full cloud against the sampled set (idempotent).  `same_stream`: the caller is on the
    stream that produced the level, so stream order already covers lv["_xyz_ready"]."""
    if "nn3_idx" not in lv:
        if not same_stream:
            torch.cuda.current_stream().wait_event(lv["_xyz_ready"])
        B, N, _ = xyz.shape
        xyz_s = lv["xyz_s"]
        if "_ordered" in lv and 256 <= xyz_s.shape[1] <= 16384:
            # both sets in Morton order: a wave's 64 queries are a compact region and its scan starts at the matching
            # place of the sampled set's order, so the 3-deep lists settle within the first ~128 candidates and the
            # rest of the scan is the bare distance test (bit-identical outputs, csrc/pointnet2.hip
            # three_nn_sorted_kernel); the sampled set is sorted here unless the level's kNN already did
            srt_s, gbox_s = lv["_ordered_s"] if "_ordered_s" in lv else pm.spatial_sort(xyz_s)
            d3, i3 = pm.three_nn_sorted(lv["_ordered"][0], lv["_ordered"][1], srt_s, gbox_s)
        else:
            d3 = torch.empty((B, N, 3), dtype=torch.float32, device=xyz.device)
            i3 = torch.empty((B, N, 3), dtype=torch.int32, device=xyz.device)
            L.check(L.lib().dh3d_three_nn(B, N, xyz_s.shape[1], L.ptr(xyz), L.ptr(xyz_s), L.ptr(d3), L.ptr(i3),
                                          L.stream_ptr()), "three_nn")
        lv["nn3_dist"], lv["nn3_idx"] = d3, i3
        if lv.get("_want_walk_plan") and "_ordered" in lv and xyz_s.shape[1] <= 1024:
            # the global tail's walk over the fine points (pm.global_tail): its per-block slot tables depend on three_nn's
            # result only -- built here, behind three_nn and off the tail's chain (17 us of the walk when built inside it)
            lv["walk_plan"] = pm.walk_plan(i3, d3, lv["_ordered"][0], xyz_s.shape[1])
    return lv


# --------------------------------------------------------------------------- flex_conv_dilate
class FlexConvDilate(nn.Module):
    """flex_conv_dilate (core/backbones.py:58-101)."""

    def __init__(self, cin, outdims, dilate, knn=8, concat=True, add_se="max_pool", upsample=True,
                 bn_eps=1e-5, xyz_prefix=False):
        super().__init__()
        if add_se not in ("max_pool", "avg_pool", ""):
            raise ValueError("add_se=%r (core/backbones.py:76-86 knows 'max_pool', 'avg_pool', anything else = none)"
                             % add_se)
        self.cin, self.outdims, self.dilate, self.knn = cin, list(outdims), dilate, knn
        self.xyz_prefix = xyz_prefix  # the input is [xyz | features] (concat_xyz): cin = 3 + feature channels
        self.concat, self.add_se, self.upsample = concat, add_se, upsample
        c = cin
        for i, d in enumerate(outdims):
            setattr(self, "flexconv_%d" % i, FlexConvParams(c, d))
            setattr(self, "flexconv_%d_bn" % i, TPBatchNorm(d, bn_eps))
            c = d
        if add_se in ("max_pool", "avg_pool"):
            self.se = SEBlock(outdims[-1])
        if concat:
            self.concat_conv1d = FeatureConv1d(outdims[-1] + cin, outdims[-1], bn=True, bn_eps=bn_eps)
        self._prep = None

    def prepare(self):
        prep = []
        for i, d in enumerate(self.outdims):
            fc = getattr(self, "flexconv_%d" % i)
            bn = getattr(self, "flexconv_%d_bn" % i)
            scale, shift = [t.detach() for t in bn.fold()]
            theta, bias = fc.position_theta.detach(), fc.position_bias.detach()
            if i == 0 and self.xyz_prefix:
                # concat_xyz (core/backbones.py:180-181): the input is [xyz | features], 3 + 128 channels.  flex_conv is
                # linear in its input channels, so it splits into the fused kernel on the 128 feature channels plus the
                # same kernel on the coordinates (padded to the feature width, zero weights in the padding); the
                # feature_bias / BatchNorm / ReLU epilogue runs once on the sum
                cf = theta.shape[1] - 3
                fb = fc.feature_bias.detach().reshape(-1)
                tx = torch.zeros((3, cf, d), dtype=theta.dtype, device=theta.device)
                bx = torch.zeros((cf, d), dtype=theta.dtype, device=theta.device)
                tx[:, :3], bx[:3] = theta[:, :3], bias[:3]
                prep.append({
                    "wp": pm.pack_flex_weight(theta[:, 3:].contiguous(), bias[3:].contiguous()), "wp3": None,
                    "wp_xyz": pm.pack_flex_weight(tx, bx), "cf": cf,
                    "fb": None, "scale": scale.contiguous(), "shift": (shift + fb * scale).contiguous(), "dout": d,
                })
                continue
            x6 = pm.flex_x6_supported(theta.shape[1], d, 8)  # full-resolution shapes: bf16x6 pipeline (K == 8)
            tx6 = any(pm.flex_tile_x6_supported(theta.shape[1], d, k) for k in (8, 12))  # 32-point tiles, bf16x6 tile GEMM
            prep.append({
                "wp": pm.pack_flex_weight(theta, bias),
                "wp3": pm.pack_flex_weight_x3(theta, bias) if x6 else None,
                "wp3t": pm.pack_flex_weight_x3(theta, bias) if tx6 else None,
                "fb": fc.feature_bias.detach().reshape(-1).contiguous(),
                "scale": scale, "shift": shift, "dout": d,
            })
        self._prep = prep
        if self.add_se in ("max_pool", "avg_pool"):
            self.se.prepare()
        if self.concat:
            self.concat_conv1d.tfconv0.prepare()
        return prep

    def shortcut_fusable(self, n_points):
        """forward(shortcut_src=...) available: the concat conv takes the fused path at this cloud size and holds the
        stacked weights (Conv2D1x1.fuse_shortcut)."""
        conv = self.concat_conv1d.tfconv0 if self.concat else None
        return (conv is not None and self.upsample and self.dilate > 1 and n_points >= 4096
                and "wp3_sc" in (conv._prep or conv.prepare()))

    def commuted_partial(self, feat):
        """The part of this block's concat conv that needs `feat` only (Conv2D1x1.lower_partial), or None when the
        commuted form does not apply: clouds of more than 4096 points, where the sampling chain is the critical one and
        whatever can run beside it should (rule on the points per cloud only, never on the batch)."""
        conv = self.concat_conv1d.tfconv0 if self.concat else None
        if (conv is None or not (self.upsample and self.dilate > 1) or feat.shape[1] <= 4096
                or not conv.commuted_supported(self.outdims[-1])):
            return None
        return conv.lower_partial(feat)

    def forward(self, geo, feat, nbr=None, residual=None, l2cat=None, shortcut_src=None, lower_partial=None,
                coarse_only=False, post_conv=None, post_linear=None, post_tails=None, fused_tail=None):
        """geo: Geometry; feat [B,N,cin]; nbr [B,N,K] for dilate == 1 (else computed on the sampled set);
        residual [B,N,cout]: added to the concat conv's output in its store (the caller's shortcut branch);
        l2cat = (prefix [B,N,3], eps): return [prefix | l2_normalize(output)] instead of the output;
        lower_partial: commuted_partial(feat), computed earlier by the caller; fused_tail = (x1, shortcut conv): instead of
        lower_partial / residual, both GEMMs run inside the tail kernel (Conv2D1x1.forward_commuted_fused; needs l2cat);
        post_conv: a Conv2D1x1 the caller applies
        to this block's output next -- where it can ride in the SE kernel the result is (output, post_conv(output))."""
        prep = self._prep or self.prepare()
        self._last_post = None
        cw_top = None
        if self.dilate > 1:
            lv = geo.level(self.dilate, self.knn, finish=False)  # three_nn is joined only where it is consumed
            xyz_s, nbr_s = lv["xyz_s"], lv["nbr_s"]
            # group_point (core/tf_utils.py:92-95) as its own tiny kernel: fusing it into the first flex_conv's neighbour
            # gather (pm.flex_conv remap=, bit-identical) was measured and is NOT used -- the 8-fold re-read of every
            # sampled row then goes to the scattered rows of the full map instead of a compact, cache-resident copy
            # (64->128 at 8x1024: 15.6 + 10.8 us -> 29.5 us)
            x, remap = gather_rows(feat, lv["idx"]), None
        else:
            xyz_s, nbr_s, x, remap = geo.xyz, (nbr if nbr is not None else geo.nbr), feat, None
        for p in prep:
            x6 = p["wp3"] is not None and nbr_s.shape[2] == 8 and remap is None
            if "wp_xyz" in p:  # [xyz | features] input: two launches of the fused kernel, one epilogue (see prepare)
                xp = torch.zeros((x.shape[0], x.shape[1], p["cf"]), dtype=torch.float32, device=x.device)
                xp[:, :, :3] = xyz_s
                y = pm.flex_conv(x, xyz_s, nbr_s, p["wp"], p["dout"]) + pm.flex_conv(xp, xyz_s, nbr_s, p["wp_xyz"], p["dout"])
                x = pm.scale_shift_act(y.reshape(-1, p["dout"]), p["scale"], p["shift"], True).reshape(y.shape)
            elif x6:
                x = pm.flex_conv_x6(x, xyz_s, nbr_s, p["wp3"], p["dout"], pre_bias=p["fb"], scale=p["scale"],
                                    shift=p["shift"], act=pm.ACT_RELU,
                                    reserve_cus_per_xcd=getattr(geo, "busy_cus_per_xcd", 0))
            elif (post_linear is not None and coarse_only and p is prep[-1] and remap is None and not self.add_se
                  and pm.flex_post_supported(x.shape[2], p["dout"], nbr_s.shape[2], post_linear[1])):
                # post_linear = (packed [dout, 64] weight, 64): the caller's next linear layer on this block's coarse
                # output rides in the last flex_conv's launch (NetVLAD's cluster logits, model.compute_global)
                if FLEX_TX6 and p.get("wp3t") is not None:
                    x, self._last_post = pm.flex_conv_tile_x6(x, xyz_s, nbr_s, p["wp3t"], p["dout"], pre_bias=p["fb"],
                                                              scale=p["scale"], shift=p["shift"], act=pm.ACT_RELU,
                                                              wpost_packed=post_linear[0], Dpost=post_linear[1])
                else:
                    x, self._last_post = pm.flex_conv_post(x, xyz_s, nbr_s, p["wp"], p["dout"], post_linear[0],
                                                           post_linear[1], pre_bias=p["fb"], scale=p["scale"],
                                                           shift=p["shift"], act=pm.ACT_RELU)
            elif (FLEX_TX6 and p.get("wp3t") is not None and remap is None
                  and pm.flex_tile_x6_supported(x.shape[2], p["dout"], nbr_s.shape[2])):
                # the sampled levels (and cfg 5's K = 12 layer): 32-point tiles, tile GEMM on the bf16 pipe (csrc/flex_tx6.hip)
                x = pm.flex_conv_tile_x6(x, xyz_s, nbr_s, p["wp3t"], p["dout"], pre_bias=p["fb"], scale=p["scale"],
                                         shift=p["shift"], act=pm.ACT_RELU)
            else:
                x = pm.flex_conv(x, xyz_s, nbr_s, p["wp"], p["dout"], pre_bias=p["fb"], scale=p["scale"],
                                 shift=p["shift"], act=pm.ACT_RELU, remap=remap)
            remap = None
        post = None
        if self.add_se == "max_pool":
            cconv = self.concat_conv1d.tfconv0 if self.concat else None
            if (post_conv is not None and not (self.upsample and self.dilate > 1) and not self.concat
                    and residual is None and l2cat is None and shortcut_src is None):
                r = self.se.forward_on_max_pool_then_conv(x, nbr_s, post_conv, tails=post_tails)
                if len(r) == 4:  # (None, post_conv(output), tail on the output, tail on post_conv(output)): one launch
                    return r
                x, post = r  # x is this block's output
            elif (self.upsample and self.dilate > 1 and not coarse_only and cconv is not None
                  and (lower_partial is not None or fused_tail is not None) and shortcut_src is None and x.shape[2] == 128 and cconv.cout == 128
                  and (self.se._prep or self.se.prepare())[4] is not None and cconv._prep.get("c_top") == 128):
                # the commuted concat conv's upper block (coarse @ W_top, no bias / activation: those belong to the sum)
                # rides in the SE kernel: one launch + its dependency gap less on the chain behind the sampling
                sp = self.se._prep
                x, cw_top = pm.se_res_pool_conv(x, nbr_s, *sp[4], cconv._prep["wp_top"], None, None, None, act=pm.ACT_NONE)
            else:
                x = self.se.forward_on_max_pool(x, nbr_s)
        elif self.add_se == "avg_pool":  # flex_avg (theta 0, bias eye: the neighbour sum) * 1/knn, backbones.py:80-83
            x = self.se(x, pm.flex_avg(x, nbr_s, 1.0 / self.knn))
        if self.upsample and self.dilate > 1:
            self._last_coarse = (x, lv)  # the level's features before up-sampling (PointMLPHead.forward_interpolated)
            geo.finish(lv)
            if coarse_only:  # the caller commutes the up-sampling through whatever consumes it (model.compute_global)
                return x
            conv = self.concat_conv1d.tfconv0 if self.concat else None
            if conv is not None and fused_tail is not None:
                return conv.forward_commuted_fused(x, lv["nn3_idx"], lv["nn3_dist"], feat, fused_tail[0], fused_tail[1],
                                                   l2cat, cw=cw_top)
            if conv is not None and lower_partial is not None and shortcut_src is None:
                return conv.forward_commuted(x, lv["nn3_idx"], lv["nn3_dist"], lower_partial, act=pm.ACT_RELU,
                                             residual=residual, l2cat=l2cat, cw=cw_top)
            if conv is not None and conv.upsampled_supported(x, lv["nn3_idx"], feat):
                # up-sampling fused into the concat conv's operand staging: the [B,N,C] tensor is never written
                fuse = l2cat if conv.cout == 128 else None
                y = conv.forward_upsampled(x, lv["nn3_idx"], lv["nn3_dist"], x2=feat, act=pm.ACT_RELU,
                                           residual=residual, l2cat=fuse, shortcut=shortcut_src)
                return y if (l2cat is None or fuse is not None) else pm.l2norm_concat(y, l2cat[1], prefix=l2cat[0])
            x = pm.three_interpolate_idw(x, lv["nn3_idx"], lv["nn3_dist"])
        if shortcut_src is not None:
            raise RuntimeError("shortcut_src was given but the fused up-sample + concat conv path is not available "
                               "for this shape: the caller must check shortcut_fusable() (the stage-1 shortcut "
                               "would be dropped silently otherwise)")
        if self.concat:
            x = self.concat_conv1d(x, x2=feat, act=pm.ACT_RELU, residual=residual)
        elif residual is not None:
            x = x + residual
        if post is not None:
            return x, post  # (block output, post_conv(block output))
        return x if l2cat is None else pm.l2norm_concat(x, l2cat[1], prefix=l2cat[0])


# --------------------------------------------------------------------------- local backbone
class BackboneLocalDilate(nn.Module):
    """backbone_local_dilate (core/backbones.py:104-127)."""

    def __init__(self, featdim=128, dilate2=8, bn_eps=1e-5):
        super().__init__()
        if featdim > 128 or featdim < 4 or featdim % 4:
            raise ValueError("featdim must be a multiple of 4 in [4, 128] (core/backbones.py:125: featdim < 128 adds "
                             "'final_fc', larger values leave the 128-d descriptor as it is upstream)")
        self.featdim = featdim
        self.initconv = nn.Module()
        limit = math.sqrt(6.0 / (3 + 32))
        self.initconv.position_theta = nn.Parameter(torch.empty(3, 32).uniform_(-limit, limit))
        self.initconv.position_bias = nn.Parameter(torch.zeros(32))
        self.initconv_bn = TPBatchNorm(32, bn_eps)
        self.stage1 = FlexConvDilate(32, [64, 64], dilate=1, knn=8, concat=False, add_se="max_pool", bn_eps=bn_eps)
        self.before_stage2_conv1d = FeatureConv1d(64, 64, bn=True, bn_eps=bn_eps)
        self.stage2 = FlexConvDilate(64, [128, 128], dilate=dilate2, knn=8, concat=True, add_se="max_pool",
                                     bn_eps=bn_eps)
        self.local_stage1_shortcut = FeatureConv1d(64, 128, bn=True, bn_eps=bn_eps)
        if featdim < 128:  # backbones.py:125-126: feature_conv1d_1(feat, featdim, 'final_fc') -- Conv2D + BNReLU
            self.final_fc = FeatureConv1d(128, featdim, bn=True, bn_eps=bn_eps)
        self._prep = None

    def prepare(self):
        scale, shift = [t.detach() for t in self.initconv_bn.fold()]
        self._prep = {"theta": self.initconv.position_theta.detach().contiguous(),
                      "bias": self.initconv.position_bias.detach().contiguous(), "scale": scale, "shift": shift}
        self.stage1.prepare()
        self.before_stage2_conv1d.tfconv0.prepare()
        self.stage2.prepare()
        self.local_stage1_shortcut.tfconv0.prepare()
        self.stage2.concat_conv1d.tfconv0.fuse_shortcut(self.local_stage1_shortcut.tfconv0)
        if self.featdim < 128:
            self.final_fc.tfconv0.prepare()
        return self._prep

    def forward(self, geo):
        p = self._prep or self.prepare()
        nn_8 = geo.nbr if geo.nbr.shape[2] == 8 else geo.nbr[:, :, 0:8].contiguous()  # backbones.py:105
        init = pm.conv_pointset_xyz(geo.xyz, nn_8, p["theta"], p["bias"], scale=p["scale"], shift=p["shift"],
                                    act=pm.ACT_RELU)
        init = pm.flex_pool(init, nn_8)
        x1 = self.stage1(geo, init, nbr=nn_8)
        x2 = self.before_stage2_conv1d(x1, act=pm.ACT_RELU)
        # BNReLU(conv(x1)) + stage2 (:123).  The shortcut needs only stage-1 features, so it is issued here, under
        # the farthest-point sampling that stage 2 waits for, and the sum is folded into stage 2's last store.
        shortcut = self.local_stage1_shortcut(x1, act=pm.ACT_RELU)
        feat = self.stage2(geo, x2, residual=shortcut)
        if self.featdim < 128:
            feat = self.final_fc(feat, act=pm.ACT_RELU)
        return geo.xyz, feat


# --------------------------------------------------------------------------- heads
class PointMLPHead(nn.Module):
    """detection_block (:132-151) / globalatt_block (:156-173): Conv2D chain -> 1 logit -> sigmoid."""

    def __init__(self, cin, conv_dims, fc_bias_init=0.0, bn_eps=1e-5):
        super().__init__()
        self.conv_dims = list(conv_dims)
        c = cin
        for i, d in enumerate(conv_dims):
            setattr(self, "detec_conv%d" % i, Conv2D1x1(c, d, bn=True, bn_eps=bn_eps))
            c = d
        self.detec_conv_fc = Conv2D1x1(c, 1, bn=False, bias_init=fc_bias_init)
        self._prep = None

    def prepare(self):
        for i in range(len(self.conv_dims)):
            getattr(self, "detec_conv%d" % i).prepare()
        self._prep = {"w_fc": self.detec_conv_fc.W.detach().reshape(-1).contiguous(),
                      "b_fc": float(self.detec_conv_fc.b.detach().reshape(-1)[0].item()), "wp3": None}
        last = getattr(self, "detec_conv%d" % (len(self.conv_dims) - 1))
        if last.cin % 32 == 0 and last.cout % 256 == 0:  # wide last layer: tiled bf16x6 GEMM (csrc/dense_x6.hip)
            W2 = last.W.detach().reshape(last.cin, last.cout)
            self._prep["wp3"] = pm.pack_weight_x3(W2.contiguous())
            if len(self.conv_dims) == 1 and last.cout <= 1024:  # single wide layer: can be commuted through an up-sampling
                self._prep["wslices"] = torch.cat([pm.pack_weight_x3(W2[:, j:j + 256].contiguous())
                                                   for j in range(0, last.cout, 256)])
        return self._prep

    def interpolated_supported(self, coarse, idx):
        """forward_interpolated available (same batch-independent kind of rule as the convs': cloud size only)."""
        p = self._prep or self.prepare()
        return "wslices" in p and coarse.shape[-1] % 32 == 0 and idx.shape[1] >= 4096

    def forward_interpolated(self, coarse, idx, dist, order=None):
        """forward(three_interpolate_idw(coarse, idx, dist)) without running the wide conv on the up-sampled rows: the
        conv is linear and the interpolation weights sum to one, so it is applied to the coarse rows and its output
        interpolated (csrc/dense_x6.hip interp_head_kernel)."""
        p = self._prep or self.prepare()
        last = getattr(self, "detec_conv0")
        lp = last._prep
        return pm.interp_head(coarse, idx, dist, p["wslices"], last.cout, p["w_fc"], p["b_fc"], pre_bias=lp["b"],
                              scale=lp["scale"], shift=lp["shift"], act=pm.ACT_RELU, order=order)

    def forward(self, x):
        p = self._prep or self.prepare()
        n = len(self.conv_dims)
        for i in range(n - 1):
            x = getattr(self, "detec_conv%d" % i)(x, act=pm.ACT_RELU)
        last = getattr(self, "detec_conv%d" % (n - 1))
        lp = last._prep
        if p["wp3"] is not None:
            return pm.mlp_head_x6(x, p["wp3"], last.cout, p["w_fc"], p["b_fc"], pre_bias=lp["b"], scale=lp["scale"],
                                  shift=lp["shift"], act=pm.ACT_RELU)
        return pm.mlp_head(x, lp["wp"], last.cout, p["w_fc"], p["b_fc"], pre_bias=lp["b"], scale=lp["scale"],
                           shift=lp["shift"], act=pm.ACT_RELU)


class NetVLAD(nn.Module):
    """global_netvald_block + context_gating (core/backbones.py:202-320); variables live at the root scope."""

    def __init__(self, feature_size=256, cluster_size=64, output_dim=256, add_batch_norm=True, gating=True,
                 slim_bn_eps=1e-3):
        super().__init__()
        D, C, O = feature_size, cluster_size, output_dim
        self.D, self.C, self.O = D, C, O
        self.add_batch_norm, self.gating = bool(add_batch_norm), bool(gating)
        self.cluster_weights = nn.Parameter(torch.randn(D, C) / math.sqrt(D))
        if add_batch_norm:
            self.cluster_bn = SlimBatchNorm(C, slim_bn_eps, fused=False)
        else:  # backbones.py:224-229
            self.cluster_biases = nn.Parameter(torch.randn(C) / math.sqrt(D))
        self.cluster_weights2 = nn.Parameter(torch.randn(1, D, C) / math.sqrt(D))
        self.hidden1_weights = nn.Parameter(torch.randn(C * D, O) / math.sqrt(C))
        self.bn = SlimBatchNorm(O, slim_bn_eps)  # 'bn' is applied whatever add_batch_norm says (backbones.py:270-274)
        if gating:
            self.gating_weights = nn.Parameter(torch.randn(O, O) / math.sqrt(O))
            if add_batch_norm:
                self.gating_bn = SlimBatchNorm(O, slim_bn_eps)
            else:  # backbones.py:310-314
                self.gating_biases = nn.Parameter(torch.randn(O) / math.sqrt(O))
        self._prep = None

    def prepare(self):
        if self.add_batch_norm:
            cs, ch = [t.detach() for t in self.cluster_bn.fold()]
        else:  # "+ bias" is the folded form with unit scale
            cs, ch = torch.ones_like(self.cluster_biases.detach()), self.cluster_biases.detach().contiguous()
        s1, h1 = [t.detach() for t in self.bn.fold()]
        Wg = s2 = h2 = None
        if self.gating:
            Wg = self.gating_weights.detach().contiguous()
            if self.add_batch_norm:
                s2, h2 = [t.detach() for t in self.gating_bn.fold()]
            else:
                s2, h2 = torch.ones_like(self.gating_biases.detach()), self.gating_biases.detach().contiguous()
        self._prep = {
            "wc": pm.pack_weight(self.cluster_weights.detach().contiguous()),
            "cs": cs, "ch": ch,
            "W2": self.cluster_weights2.detach().reshape(self.D, self.C).contiguous(),
            "Wh": self.hidden1_weights.detach().contiguous(), "s1": s1, "h1": h1,
            "Wg": Wg, "s2": s2, "h2": h2,
        }
        return self._prep

    def forward(self, features, att, l2_eps=0.0):
        """features [B,N,256], att [B,N,1] -> [B,256] ('final_global'); l2_eps > 0 also applies
        tf.nn.l2_normalize(dim=-1, epsilon=l2_eps) (core/model.py:205)."""
        p = self._prep or self.prepare()
        return pm.netvlad_fused(features, att, p["wc"], p["cs"], p["ch"], p["W2"], p["Wh"], p["s1"], p["h1"], p["Wg"],
                                p["s2"], p["h2"], l2_eps=l2_eps)
