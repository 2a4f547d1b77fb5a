"""MI355X-native differentiable 3D Gaussian Splatting: a drop-in for the hot path of
scomup/EasyGaussianSplatting (its CUDA extension ``gsplatcu``) and the pieces a training run needs
around it, on hand-written HIP kernels for gfx950 behind a C ABI (``include/egs_hip.h``).

    gsplatcu   the seven reference ops (project ... splat, splatB) + set_policy
    function   GSFunction (autograd boundary of gsmodel.py), GSRawFunction (GSModel.forward in one node), render
    fused      the fused forward / backward behind GSFunction
    loss       gau_loss (0.8 L1 + 0.2 (1 - SSIM)) as HIP kernels
    optim      FusedAdam;  density  DensityControl (prune / clone / split / alpha reset on the device)
    trainer    the train.py loop, one camera view per GPU;  dist_views  the RCCL gradient exchange
    gau_io, colmap, dataset, knn, viewer   file formats, COLMAP scenes, nearest neighbours, viewer preprocess
    scene      deterministic synthetic scenes (tests, bench.py)

Importing this package does not touch the GPU; the HIP library is loaded on the first op call and its
absence is a hard error (no CPU fallback).
"""
__version__ = "0.1.0"
