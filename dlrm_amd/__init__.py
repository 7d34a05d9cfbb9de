"""dlrm_amd — MI355X-native DLRM forward/backward hot path (hand-written HIP kernels behind a C ABI)
with the module surface of facebookresearch/dlrm's `DLRM_Net`."""
from . import ext_dist  # noqa: F401
from .dlrm_net import DLRM_Net, FusedMLP, FusedBCELoss, FusedMSELoss, set_embedding_init  # noqa: F401

__all__ = ["DLRM_Net", "FusedMLP", "FusedBCELoss", "FusedMSELoss", "set_embedding_init", "ext_dist"]
