"""dh3d_amd -- MI355X-native implementation of the DH3D point-cloud feature-extraction hot path.

    dh3d_amd.ops        drop-in operators (reference names / layouts), differentiable
    dh3d_amd.layers     Layer classes mirroring core/layers.py
    dh3d_amd.pm         fused point-major kernels (model path)
    dh3d_amd.backbones  FlexConv+SE encoder, heads, NetVLAD
    dh3d_amd.model      DH3D(config): compute_local / compute_global / forward
    dh3d_amd.dist       batch sharding + RCCL all-gather of global descriptors
Everything computes in libdh3d_hip.so (hand-written gfx950 kernels); there is no CPU fallback.
"""
from .configs import ConfigFactory, dotdict  # noqa: F401

__version__ = "0.1.0"
