"""Reference module name for easygaussiansplatting_amd.dataset (gsplat/gausplat_dataset.py)."""
from easygaussiansplatting_amd.dataset import Camera, GSplatDataset  # noqa: F401
from gsplat.read_write_model import *  # noqa: F401,F403
