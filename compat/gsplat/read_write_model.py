"""Reference module name for easygaussiansplatting_amd.colmap (gsplat/read_write_model.py)."""
from easygaussiansplatting_amd.colmap import (  # noqa: F401
    CAMERA_MODELS, SH_C0_0, BaseImage, Camera, ColmapFormatError, Image, Point3D, points_to_gaussians, qvec2rotmat,
    read_cameras_binary, read_images_binary, read_model, read_points3D_binary, read_points_bin_as_gau)
