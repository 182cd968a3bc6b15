"""Registry-visible classes of the hot path (what ``plugin_dir = "projects/mmdet3d_plugin/"``
provides in the reference).  Importing this package fills the registries."""
from .view_transformer import ViewTransformerLiftSplatShootVoxel  # noqa: F401
from .voxel_encoder import CustomResNet3D, OccHead, SECONDFPN3D  # noqa: F401
from .image_branch import CustomEfficientNet, SECONDFPN  # noqa: F401  (the step before the hot path)
from .detector import BEVDepthOccupancy  # noqa: F401

# No process-wide switches are flipped here: every convolution / normalisation of the path runs on the ssbev kernels, the
# remaining ATen calls (group_norm of the [B, 30] camera vector, linear, softmax) never reach MIOpen.
