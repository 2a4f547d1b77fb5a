"""MI355X-native differentiable 3D Gaussian Splatting rasterizer: a drop-in for
the reference's CUDA extension ``gsplatcu`` (scomup/EasyGaussianSplatting).

    from easygaussiansplatting_amd import gsplatcu as gsc   # the seven reference ops
    from easygaussiansplatting_amd.function import GSFunction  # autograd boundary

Importing this package does not touch the GPU; the HIP library is loaded on the
first op call and its absence is a hard error (no CPU fallback).
"""
__version__ = "0.1.0"
