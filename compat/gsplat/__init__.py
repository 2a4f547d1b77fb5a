"""OPT-IN: ``import gsplat.*`` -- the module names of the reference's Python package,
re-exporting the MI355X implementations so that the reference's own ``train.py`` runs unmodified with
``<repo>/compat`` AHEAD of the reference checkout on ``PYTHONPATH`` (next to the ``gsplatcu`` drop-in, which only
needs the repository root):

    gsplat.gau_io            -> easygaussiansplatting_amd.gau_io
    gsplat.read_write_model  -> easygaussiansplatting_amd.colmap
    gsplat.gausplat_dataset  -> easygaussiansplatting_amd.dataset
    gsplat.pytorch_ssim      -> easygaussiansplatting_amd.loss
    gsplat.gsmodel           -> GSFunction / GSModel / get_training_params on the fused kernels
    gsplat.utils             -> activations, learning-rate schedule

This directory is NOT on the path of the plain drop-in (§1): there the reference keeps its whole ``gsplat`` package.
When it is opted into, the names it does not provide -- the reference's NumPy renderer ``gsplat.gausplat`` and
``gsplat.sh_coef``, which ``forward_gpu.py:6`` / ``backward_gpu.py:5`` import -- must stay importable: the package
path is extended with every other ``gsplat`` directory on ``sys.path`` (the reference's is a namespace package
without ``__init__.py``), this directory first.
"""
import pkgutil

__path__ = pkgutil.extend_path(__path__, __name__)
