"""Registry-visible classes of the hot path (what ``plugin_dir = "projects/mmdet3d_plugin/"``
provides in the reference).  Importing this package fills the registries."""
import torch

from .view_transformer import ViewTransformerLiftSplatShootVoxel  # noqa: F401
from .voxel_encoder import CustomResNet3D, OccHead, SECONDFPN3D  # noqa: F401
from .image_branch import CustomEfficientNet, SECONDFPN  # noqa: F401  (the step before the hot path)
from .detector import BEVDepthOccupancy  # noqa: F401

# Normalisation layers stay on ATen's native HIP kernels (no MIOpen JIT on a fresh box).
torch.backends.cudnn.enabled = False
