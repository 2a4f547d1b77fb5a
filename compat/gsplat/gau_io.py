"""Reference module name for easygaussiansplatting_amd.gau_io (gsplat/gau_io.py)."""
from easygaussiansplatting_amd.gau_io import (  # noqa: F401
    PlyFormatError, get_example_gs, load_gs, load_ply, matrix_to_quaternion, quaternion_to_matrix,
    read_ply_vertices, rotate_gaussian, save_gs, save_ply, save_training_params)
from easygaussiansplatting_amd.scene import gsdata_type  # noqa: F401
from gsplat.utils import *  # noqa: F401,F403  (the reference's gau_io re-exports gsplat.utils)
