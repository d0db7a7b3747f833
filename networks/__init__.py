"""``networks`` - the module the reference's JSON configs name as ``_target_``
(3d_ldm/config/config_train_16g.json:8,40: "networks.AutoencoderKL", "networks.DiffusionModelUNet").
The reference ships no such module; this one binds those names to the MI355X-native implementations."""
from ldm3d.networks import AutoencoderKL, DiffusionModelUNet  # noqa: F401

__all__ = ["AutoencoderKL", "DiffusionModelUNet"]
