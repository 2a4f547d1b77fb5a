"""``import gsplatcu as gsc`` -- the literal drop-in name of the reference's CUDA
extension (reference gsplatcu/setup.py:6, ext.cpp:68-77), re-exporting the
MI355X implementation in easygaussiansplatting_amd.gsplatcu."""
from easygaussiansplatting_amd.gsplatcu import (  # noqa: F401
    project, computeCov3D, computeCov2D, sh2Color, inverseCov2D, splat, splatB,
    set_policy, get_policy, chain_rule)
