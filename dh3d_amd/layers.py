"""Layer wrappers in the reference's channels-first convention (mirrors core/layers.py).

KnnBruteforce (core/layers.py:49-107), FlexPooling (:110-175), FlexConvolution (:178-339, :439-461),
Flex_Avg (:342-436, :464-480), ConvolutionPointset (:564-707): same constructor arguments that matter, same weight names
(position_theta / position_bias / feature_bias), same tensor layouts ([B, C, N] features,
[B, K, N] neighbourhoods).  They call the drop-in operators of dh3d_amd.ops and are differentiable
through the registered gradients.  The fused point-major model path lives in dh3d_amd.backbones.
"""
import math

import torch
from torch import nn

from . import ops

__all__ = ["KnnBruteforce", "knn_bruteforce", "FlexPooling", "flex_pooling", "FlexConvolution",
           "ConvolutionPointset", "Flex_Avg", "flex_avg"]


class KnnBruteforce(nn.Module):
    """positions [B, Dp, N] -> (neighborhoods [B, K, N] int32, distances [B, K, N])."""

    def __init__(self, k, data_format="simple"):
        super().__init__()
        assert k > 0
        assert data_format in ["simple", "expanded"]
        self.k = k
        self.data_format = data_format

    def forward(self, positions):
        if self.data_format == "expanded":
            positions = positions.squeeze(2)
        nn_, dist = ops.knn_bruteforce(positions, k=self.k)
        nn_ = nn_.transpose(1, 2).contiguous()      # core/layers.py:92
        dist = dist.transpose(1, 2).contiguous()    # core/layers.py:93
        if self.data_format == "expanded":
            nn_ = nn_.unsqueeze(2)
        return nn_, dist


def knn_bruteforce(positions, k, data_format="simple"):
    return KnnBruteforce(k, data_format=data_format)(positions)


class FlexPooling(nn.Module):
    def __init__(self, data_format="simple"):
        super().__init__()
        assert data_format in ["simple", "expanded"]
        self.data_format = data_format

    def forward(self, features, neighborhoods):
        if self.data_format == "expanded":
            features, neighborhoods = features.squeeze(2), neighborhoods.squeeze(2)
        y, _ = ops.flex_pooling(features, neighborhoods)
        if self.data_format == "expanded":
            y = y.unsqueeze(2)
        return y


def flex_pooling(features, neighborhoods, data_format="simple"):
    return FlexPooling(data_format=data_format)(features, neighborhoods)


class FlexConvolution(nn.Module):
    """Weights: position_theta [Dp, Din, Dout], position_bias [Din, Dout], feature_bias [Dout, 1]."""

    def __init__(self, in_channels, filters, dp=3, activation=None, use_feature_bias=True,
                 data_format="simple"):
        super().__init__()
        assert data_format in ["simple", "expanded"]
        self.filters = int(filters)
        self.activation = activation
        self.data_format = data_format
        # glorot-uniform like the Keras default kernel initializer; zeros for both biases (layers.py:231-233)
        limit = math.sqrt(6.0 / (in_channels + filters))
        self.position_theta = nn.Parameter(torch.empty(dp, in_channels, filters).uniform_(-limit, limit))
        self.position_bias = nn.Parameter(torch.zeros(in_channels, filters))
        self.feature_bias = nn.Parameter(torch.zeros(filters, 1)) if use_feature_bias else None

    def forward(self, features, positions, neighborhoods):
        if self.data_format == "expanded":
            features, positions, neighborhoods = features.squeeze(2), positions.squeeze(2), neighborhoods.squeeze(2)
        y = ops.flex_convolution(features, positions, neighborhoods, self.position_theta, self.position_bias)
        if self.feature_bias is not None:
            y = y + self.feature_bias
        if self.activation is not None:
            y = self.activation(y)
        if self.data_format == "expanded":
            y = y.unsqueeze(2)
        return y


class ConvolutionPointset(nn.Module):
    """Weights: position_theta [Din, Dout], position_bias [Dout] (+ feature_bias [Dout,1] if enabled)."""

    def __init__(self, in_channels, filters, activation=None, use_feature_bias=False, data_format="simple"):
        super().__init__()
        assert data_format in ["simple", "expanded"]
        self.filters = int(filters)
        self.activation = activation
        self.data_format = data_format
        limit = math.sqrt(6.0 / (in_channels + filters))
        self.position_theta = nn.Parameter(torch.empty(in_channels, filters).uniform_(-limit, limit))
        self.position_bias = nn.Parameter(torch.zeros(filters))
        self.feature_bias = nn.Parameter(torch.zeros(filters, 1)) if use_feature_bias else None

    def forward(self, features, neighborhoods):
        if self.data_format == "expanded":
            features, neighborhoods = features.squeeze(2), neighborhoods.squeeze(2)
        y = ops.convolution_pointset(features, neighborhoods, self.position_theta, self.position_bias)
        if self.feature_bias is not None:
            y = y + self.feature_bias
        if self.activation is not None:
            y = self.activation(y)
        if self.data_format == "expanded":
            y = y.unsqueeze(2)
        return y


class Flex_Avg(nn.Module):
    """core/layers.py:342-436: FlexConvolution with a NON-trainable `position_theta` [Dp, Din, Dout] (zeros by default,
    :347,379-384) and `position_bias` = eye(Dout) (:386, which makes Din == Dout): with theta = 0 every positional term
    of the flex_conv sum is an exact 0 * f and what is left is the neighbour sum out[b, c, n] = sum_k f[b, c, nbr[b, k, n]]
    (backbones.py:80-83 scales it by 1/knn for `add_se='avg_pool'`).  That case runs on the neighbour-sum kernel
    (dh3d_flex_avg_pm_fwd, point-major, through two LDS-tile transposes); a theta that was loaded with non-zero values,
    or inputs that need gradients, take the flex_conv operator with the same theta / eye -- the layer as the reference
    wrote it."""

    def __init__(self, in_channels, filters, dp=3, activation=None, data_format="simple"):
        super().__init__()
        assert data_format in ["simple", "expanded"]
        if int(in_channels) != int(filters):
            raise ValueError("Flex_Avg: position_bias = eye(filters) needs in_channels == filters (core/layers.py:386)")
        self.filters = int(filters)
        self.activation = activation
        self.data_format = data_format
        self.position_theta = nn.Parameter(torch.zeros(dp, in_channels, filters), requires_grad=False)
        self.register_buffer("position_bias", torch.eye(filters), persistent=False)  # not a variable upstream

    def _theta_is_zero(self):
        """position_theta == 0 (what the reference initialises it to and never trains).  Asking the device costs a host
        sync and cannot happen inside a hipGraph capture, so: OUTSIDE a capture the tensor is re-checked on every call
        (writes through `.data` -- p.data.copy_(...), common in loaders -- change neither data_ptr nor _version, so no
        cache key can see them); UNDER capture the verdict of the last eager call stands -- run one eager forward after
        changing theta, as hipGraph capture needs for warm-up anyway.  load_state_dict / .to() / .float() reset it."""
        if self.__dict__.get("_zero") is None or not torch.cuda.is_current_stream_capturing():
            self._zero = not bool(self.position_theta.any())
        return self._zero

    def reset_theta_cache(self):
        self._zero = None

    def _load_from_state_dict(self, *args, **kwargs):
        self.reset_theta_cache()
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self.reset_theta_cache()
        return super()._apply(fn, *args, **kwargs)

    def forward(self, features, positions, neighborhoods):
        from . import pm
        if self.data_format == "expanded":
            features, positions, neighborhoods = features.squeeze(2), positions.squeeze(2), neighborhoods.squeeze(2)
        plain_sum = (not features.requires_grad and features.dtype == torch.float32 and features.shape[1] % 4 == 0
                     and self._theta_is_zero())
        if plain_sum:
            f_pm = pm.transpose_last2(features)                                   # [B, N, C]
            nb_pm = pm.transpose_last2(neighborhoods.to(torch.int32))             # [B, N, K]
            y = pm.transpose_last2(pm.flex_avg(f_pm, nb_pm, 1.0))                 # [B, C, N]
        else:
            y = ops.flex_convolution(features, positions, neighborhoods, self.position_theta,
                                     self.position_bias.to(features.dtype))
        if self.activation is not None:
            y = self.activation(y)
        if self.data_format == "expanded":
            y = y.unsqueeze(2)
        return y


def flex_avg(features, positions, neighborhoods, filters, activation=None, data_format="simple"):
    """core/layers.py:464-480."""
    cin = features.shape[1]
    layer = Flex_Avg(cin, filters, dp=positions.shape[1], activation=activation, data_format=data_format).to(features.device)
    return layer(features, positions, neighborhoods)
