"""``import gsplat.*`` -- the module names of the reference's Python package, re-exporting the MI355X
implementations so that the reference's own ``train.py`` / ``forward_gpu.py`` run unmodified with this
repository on ``PYTHONPATH`` (next to the ``gsplatcu`` drop-in):

    gsplat.gau_io            -> easygaussiansplatting_amd.gau_io
    gsplat.read_write_model  -> easygaussiansplatting_amd.colmap
    gsplat.gausplat_dataset  -> easygaussiansplatting_amd.dataset
    gsplat.pytorch_ssim      -> easygaussiansplatting_amd.loss
    gsplat.gsmodel           -> GSFunction / GSModel / get_training_params on the fused kernels
    gsplat.utils             -> activations, learning-rate schedule

(The NumPy reference renderer ``gsplat.gausplat`` and the viewer are not provided: ``oracle/`` holds the
CPU restatement as test infrastructure.)
"""
