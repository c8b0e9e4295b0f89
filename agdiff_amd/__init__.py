"""agdiff_amd: MI355X-native implementation of AGDIFF's diffusion-sampling hot path.

Public surface mirrors the reference for this path only:
    from agdiff_amd import get_model, DualEncoderEpsNetwork
"""
from .config import Config, drugs_model_config, qm9_model_config  # noqa: F401
from .epsnet import DualEncoderEpsNetwork, get_beta_schedule, get_model  # noqa: F401

__all__ = ["get_model", "DualEncoderEpsNetwork", "get_beta_schedule", "Config", "qm9_model_config",
           "drugs_model_config"]
